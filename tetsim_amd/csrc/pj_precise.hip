// POLAR_JACOBI kernels, PRECISE arithmetic.  Build with -ffp-contract=off (IEEE f32, no FMA fusion,
// correctly rounded divide/sqrt) so results track the GLSL-order CPU restatement op for op.
#define TETSIM_FAST 0
#define TETSIM_MODE_SUFFIX precise
#include "pj_kernels.inc"
