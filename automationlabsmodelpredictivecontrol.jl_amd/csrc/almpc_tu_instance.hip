// almpc_tu_instance.hip -- one translation unit of libalmpc.so: the one-wave-per-instance step k_step_inst_wave.
// Device code only; the launch logic is in almpc_api.hip, which declares these instantiations `extern template` (see there).
#include "almpc_instance.hip.h"
#define ALMPC_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "instances/instance.inc"
