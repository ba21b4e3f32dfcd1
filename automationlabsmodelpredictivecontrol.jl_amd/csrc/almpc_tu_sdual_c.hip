// almpc_tu_sdual_c.hip -- one translation unit of libalmpc.so: k_sdual (last third).
// Device code only; the launch logic is in almpc_api.hip, which declares these instantiations `extern template` (see there).
#include "almpc_sdual.hip.h"
#define ALMPC_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "instances/sdual_c.inc"
