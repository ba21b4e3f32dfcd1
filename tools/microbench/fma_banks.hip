// FP64 FMA with three VGPR-pair sources: cycles per instruction by register bank (bank = register number mod 4; a pair takes two banks)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fma_banks tools/microbench/fma_banks.hip && /tmp/fma_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
__global__ void k(int mode, double* out, long long* cyc) {
    long long t0 = 0, t1 = 0;
    // acc pairs: v[40:41] v[44:45] v[48:49] v[52:53] (banks 0,1)  and  v[42:43] v[46:47] v[50:51] v[54:55] (banks 2,3)
    // f: v[60:61] (banks 0,1) / v[62:63] (banks 2,3);  w: v[64:65] (0,1) / v[66:67] (2,3)
    asm volatile(
        "v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0x3ff00000\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0x3ff00000\n"
        "v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3ff00000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0x3ff00000\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0x3ff00000\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0x3ff00000\n"
        "v_mov_b32 v60, 0\n v_mov_b32 v61, 0x3e000000\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0x3e000000\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3ff00000\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0x3ff00000\n"
        ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","v64","v65","v66","v67");
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
#define RUN(BODY) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t0)); \
    for (int i = 0; i < 64; ++i) asm volatile(REP8(BODY) ::: CLOB); \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t1));
    if (mode == 0) {        // acc (0,1), f (0,1), w (0,1): all three in the same banks
        RUN("v_fma_f64 v[40:41], v[60:61], v[64:65], v[40:41]\n v_fma_f64 v[44:45], v[60:61], v[64:65], v[44:45]\n v_fma_f64 v[48:49], v[60:61], v[64:65], v[48:49]\n v_fma_f64 v[52:53], v[60:61], v[64:65], v[52:53]\n")
    } else if (mode == 1) { // acc (0,1), f (0,1), w (2,3)
        RUN("v_fma_f64 v[40:41], v[60:61], v[66:67], v[40:41]\n v_fma_f64 v[44:45], v[60:61], v[66:67], v[44:45]\n v_fma_f64 v[48:49], v[60:61], v[66:67], v[48:49]\n v_fma_f64 v[52:53], v[60:61], v[66:67], v[52:53]\n")
    } else if (mode == 2) { // acc (2,3), f (0,1), w (0,1)
        RUN("v_fma_f64 v[42:43], v[60:61], v[64:65], v[42:43]\n v_fma_f64 v[46:47], v[60:61], v[64:65], v[46:47]\n v_fma_f64 v[50:51], v[60:61], v[64:65], v[50:51]\n v_fma_f64 v[54:55], v[60:61], v[64:65], v[54:55]\n")
    } else if (mode == 3) { // dst differs from src2 (acc (0,1) -> (2,3)), f (0,1), w (2,3)
        RUN("v_fma_f64 v[42:43], v[60:61], v[66:67], v[40:41]\n v_fma_f64 v[46:47], v[60:61], v[66:67], v[44:45]\n v_fma_f64 v[50:51], v[60:61], v[66:67], v[48:49]\n v_fma_f64 v[54:55], v[60:61], v[66:67], v[52:53]\n")
    } else if (mode == 4) { // two VGPR sources + inline constant
        RUN("v_fma_f64 v[40:41], v[60:61], 1.0, v[40:41]\n v_fma_f64 v[44:45], v[60:61], 1.0, v[44:45]\n v_fma_f64 v[48:49], v[60:61], 1.0, v[48:49]\n v_fma_f64 v[52:53], v[60:61], 1.0, v[52:53]\n")
    } else if (mode == 5) { // v_mul_f64 two sources same banks
        RUN("v_mul_f64 v[40:41], v[60:61], v[64:65]\n v_mul_f64 v[44:45], v[60:61], v[64:65]\n v_mul_f64 v[48:49], v[60:61], v[64:65]\n v_mul_f64 v[52:53], v[60:61], v[64:65]\n")
    } else if (mode == 6) { // w repeated as src0 and src1? f same register for both multiplicands
        RUN("v_fma_f64 v[40:41], v[60:61], v[60:61], v[40:41]\n v_fma_f64 v[44:45], v[60:61], v[60:61], v[44:45]\n v_fma_f64 v[48:49], v[60:61], v[60:61], v[48:49]\n v_fma_f64 v[52:53], v[60:61], v[60:61], v[52:53]\n")
    } else if (mode == 7) { // v_fmac_f64 (dst = src2 implicitly), banks as mode 1
        RUN("v_fmac_f64 v[40:41], v[60:61], v[66:67]\n v_fmac_f64 v[44:45], v[60:61], v[66:67]\n v_fmac_f64 v[48:49], v[60:61], v[66:67]\n v_fmac_f64 v[52:53], v[60:61], v[66:67]\n")
    } else if (mode == 8) { // neg modifier on src0
        RUN("v_fma_f64 v[40:41], -v[60:61], v[66:67], v[40:41]\n v_fma_f64 v[44:45], -v[60:61], v[66:67], v[44:45]\n v_fma_f64 v[48:49], -v[60:61], v[66:67], v[48:49]\n v_fma_f64 v[52:53], -v[60:61], v[66:67], v[52:53]\n")
    } else if (mode == 9) { // a different w pair per instruction
        RUN("v_fma_f64 v[40:41], v[60:61], v[64:65], v[40:41]\n v_fma_f64 v[44:45], v[60:61], v[66:67], v[44:45]\n v_fma_f64 v[48:49], v[60:61], v[62:63], v[48:49]\n v_fma_f64 v[52:53], v[60:61], v[54:55], v[52:53]\n")
    } else if (mode == 10) { // eight accumulators (dependent distance 8), a different w pair per instruction
        RUN("v_fma_f64 v[40:41], v[60:61], v[64:65], v[40:41]\n v_fma_f64 v[44:45], v[60:61], v[66:67], v[44:45]\n v_fma_f64 v[48:49], v[60:61], v[62:63], v[48:49]\n v_fma_f64 v[52:53], v[60:61], v[64:65], v[52:53]\n v_fma_f64 v[42:43], v[60:61], v[66:67], v[42:43]\n v_fma_f64 v[46:47], v[60:61], v[62:63], v[46:47]\n v_fma_f64 v[50:51], v[60:61], v[64:65], v[50:51]\n v_fma_f64 v[54:55], v[60:61], v[66:67], v[54:55]\n")
    } else if (mode == 11) { // dst != src2, round robin over two register sets (as the compiler renames)
        RUN("v_fma_f64 v[42:43], v[60:61], v[64:65], v[40:41]\n v_fma_f64 v[46:47], v[60:61], v[66:67], v[44:45]\n v_fma_f64 v[40:41], v[60:61], v[64:65], v[42:43]\n v_fma_f64 v[44:45], v[60:61], v[66:67], v[46:47]\n")
    }
    double r;
    asm volatile("v_add_f64 %0, v[40:41], v[42:43]" : "=v"(r));
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
    const char* n[] = {"acc(0,1) f(0,1) w(0,1)", "acc(0,1) f(0,1) w(2,3)", "acc(2,3) f(0,1) w(0,1)", "dst(2,3) <- acc(0,1), f(0,1) w(2,3)", "two VGPR sources + constant", "v_mul_f64, sources in the same banks", "f used twice + acc(0,1)", "v_fmac_f64 acc(0,1) f(0,1) w(2,3)", "neg modifier on src0", "different w per instruction", "8 accumulators, different w", "dst != src2, alternating"};
    for (int mode = 0; mode < 12; ++mode) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, out, cyc); hipDeviceSynchronize(); }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-40s %.2f s_memtime ticks per instruction\n", n[mode], h / (64.0 * (mode == 10 ? 64 : 32)));
    }
    return 0;
}
