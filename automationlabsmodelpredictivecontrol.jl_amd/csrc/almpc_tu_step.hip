// almpc_tu_step.hip -- one translation unit of libalmpc.so: the per-step kernels of a shared model: k_step_fused, k_admm, k_polish, k_polish_sgl, k_rollout.
// Device code only; the launch logic is in almpc_api.hip, which declares these instantiations `extern template` (see there).
#include "almpc_kernels.hip.h"
#define ALMPC_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "instances/step.inc"
