// almpc_tu_design_b.hip -- one translation unit of libalmpc.so: the batched inverses, k_neg_gm_cols, k_riccati_t, k_fnn_jacobian_w.
// Device code only; the launch logic is in almpc_api.hip, which declares these instantiations `extern template` (see there).
#include "almpc_design.hip.h"
#include "almpc_riccati.hip.h"
#include "almpc_fnn.hip.h"
#define ALMPC_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "instances/design_b.inc"
