// NEOHOOKEAN_GS kernels, PRECISE arithmetic: f64 math with f32 stores exactly where Softbody.js rounds.
// Build with -ffp-contract=off (JS never fuses multiply-add).
#define TETSIM_FAST 0
#define TETSIM_MODE_SUFFIX precise
#include "nh_kernels.inc"
