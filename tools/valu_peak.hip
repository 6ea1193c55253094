// Sustained v_fma_f64 rate with every SIMD busy (developer tool): hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters)
{
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10, 0.999, 1e-3);
    (void)hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 0.999, 1e-3);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * NACC;            // wave instructions
    printf("blocks %5d (x4 waves) chains %2d: %.3f ms  %.1f TFLOP/s  %.2f cycles/wave-instr/SIMD at 2.4 GHz\n", blocks, NACC, ms,
           inst * 128 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / inst);
    (void)hipFree(out);
}
int main()
{
    run<8>(256, 200000);
    run<8>(512, 200000);
    run<8>(1024, 200000);
    run<1>(512, 1000000);
    run<1>(1024, 1000000);
    return 0;
}
