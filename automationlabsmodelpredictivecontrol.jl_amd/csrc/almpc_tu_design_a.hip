// almpc_tu_design_a.hip -- one translation unit of libalmpc.so: the per-instance and time-varying designs k_design_instance_t, k_design_ltv_reg.
// Device code only; the launch logic is in almpc_api.hip, which declares these instantiations `extern template` (see there).
#include "almpc_instance.hip.h"
#define ALMPC_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "instances/design_a.inc"
