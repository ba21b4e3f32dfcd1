// almpc_tu_polish_gen.hip -- one translation unit of libalmpc.so: the state-row finish k_polish_gen / k_polish_gen64 and k_ghat_inst.
// Device code only; the launch logic is in almpc_api.hip, which declares these instantiations `extern template` (see there).
#include "almpc_polish_gen.hip.h"
#define ALMPC_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "instances/polish_gen.inc"
