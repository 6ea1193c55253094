// Sustained v_mfma_f64_16x16x4_f64 rate with every SIMD busy (developer tool): hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters)
{
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10, 1.0, 2.0);
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("blocks %5d (x4 waves) acc %d: %.3f ms  %.1f TFLOP/s\n", blocks, NACC, ms, flop / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<8>(256, 20000);      // 1 wave per SIMD
    run<8>(512, 20000);      // 2 waves per SIMD
    run<8>(1024, 20000);     // 4 waves per SIMD
    run<1>(512, 100000);     // dependent chain
    run<2>(512, 100000);
    return 0;
}
