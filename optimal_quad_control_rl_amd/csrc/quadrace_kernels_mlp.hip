// quadrace_kernels_mlp.hip -- the two fused E2E + residual-MLP rollout kernels (rollout_fast_mlp_kernel, rollout_lean_mlp_kernel) and
// their launcher, as a translation unit of their own so that build.py can compile them WITHOUT the SLP vectoriser
// (PER_SOURCE_FLAGS; reasons and measurements next to launch_rollout_mlp's declaration in quadrace_kernels.hip).
// Nothing is written here: it is the same source, the same device functions, the same arithmetic.
#define QR_TU_MLP_ROLLOUT 1
#include "quadrace_kernels.hip"
