// Do f64 vector FMAs and f64 matrix instructions share the SIMD's double-precision datapath?  (developer tool)
// Blocks of 512 threads put two waves on every SIMD; waves 0-3 run `mode_a`, waves 4-7 run `mode_b`
// (0 = idle, 1 = a stream of independent v_fma_f64, 2 = a stream of independent v_mfma_f64_16x16x4_f64).
// If the pipes are separate, (fma beside mfma) takes as long as the slower of the two alone; if shared, about their sum.
// hipcc --offload-arch=gfx950 -O3 tools/dp_share.hip -o /tmp/dp_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double run_fma(int iters, double a, double b)
{
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)                       // 16 x 8 independent-chain FMAs = 128 vector instructions
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    return s;
}
__device__ __forceinline__ double run_mfma(int iters, double a, double b)
{
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);   // 8 x 64 cycles
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    return s;
}
__global__ __launch_bounds__(512) void k(double *out, int iters, int mode_a, int mode_b, double a, double b)
{
    const int mode = threadIdx.x < 256 ? mode_a : mode_b;
    double s = 0;
    if (mode == 1) s = run_fma(iters, a + threadIdx.x, b);
    else if (mode == 2) s = run_mfma(iters, a + threadIdx.x, b);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main()
{
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * 512);
    const int iters = 20000;
    const int combos[][2] = {{1, 0}, {2, 0}, {1, 1}, {2, 2}, {1, 2}};
    const char *nm[] = {"idle", "fma ", "mfma"};
    for (auto &c : combos) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<256, 512>>>(out, 10, c[0], c[1], 1.0, 0.999);
        hipEventRecord(e0);
        k<<<256, 512>>>(out, iters, c[0], c[1], 1.0, 0.999);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: an fma wave issues iters*128 instructions (4 cycles of DP work each), an mfma wave iters*8 (64 cycles each)
        const double cyc = ms * 1e-3 * 2.4e9;
        const double dp = (c[0] == 1 ? iters * 128.0 * 4 : c[0] == 2 ? iters * 8.0 * 64 : 0) + (c[1] == 1 ? iters * 128.0 * 4 : c[1] == 2 ? iters * 8.0 * 64 : 0);
        printf("waves 0-3 %s | waves 4-7 %s : %8.3f ms = %.3g cycles per SIMD; nominal DP work %.3g cycles (%.0f %%)\n", nm[c[0]], nm[c[1]], ms, cyc, dp,
               100.0 * dp / cyc);
    }
    return 0;
}
