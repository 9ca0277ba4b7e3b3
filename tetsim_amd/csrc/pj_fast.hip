// POLAR_JACOBI kernels, FAST arithmetic: FMA contraction, v_rcp/v_rsq/v_sin, algebraically simplified
// rotation columns.  Tolerance-level parity (stated in tests/test_gpu_polar.py).
#define TETSIM_FAST 1
#define TETSIM_MODE_SUFFIX fast
#include "pj_kernels.inc"
