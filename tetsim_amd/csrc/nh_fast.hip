// NEOHOOKEAN_GS kernels, FAST arithmetic: f32 throughout with FMA contraction.  Tolerance-level parity.
#define TETSIM_FAST 1
#define TETSIM_MODE_SUFFIX fast
#include "nh_kernels.inc"
