// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand/result layout, summation semantics, issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_probe.hip -o /tmp/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_one(const double *A, const double *B, const double *C, double *D)
{
    // A[16][4], B[4][16], C/D[16][16] row-major
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}
__global__ void k_rate(double *out, int n)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * n);
}
int main()
{
    double A[64], B[64], C[256], D[256];
    srand(1);
    int nfail[4] = {0, 0, 0, 0}, total = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dC, sizeof C); hipMalloc(&dD, sizeof D);
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 64; ++i) { A[i] = (rand() / (double)RAND_MAX - 0.5) * pow(10, rand() % 7 - 3); B[i] = (rand() / (double)RAND_MAX - 0.5) * pow(10, rand() % 7 - 3); }
        for (int i = 0; i < 256; ++i) C[i] = (rand() / (double)RAND_MAX - 0.5) * pow(10, rand() % 7 - 3);
        hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice); hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double c = C[i * 16 + j];
                double p[4];
                for (int k = 0; k < 4; ++k) p[k] = A[i * 4 + k];
                // 0: fma chain k ascending  1: fma chain k descending  2: exact products summed pairwise  3: unfused products
                double r0 = c, r1 = c;
                for (int k = 0; k < 4; ++k) r0 = fma(A[i * 4 + k], B[k * 16 + j], r0);
                for (int k = 3; k >= 0; --k) r1 = fma(A[i * 4 + k], B[k * 16 + j], r1);
                long double s = c;
                for (int k = 0; k < 4; ++k) s += (long double)A[i * 4 + k] * (long double)B[k * 16 + j];
                double r2 = (double)s;
                double r3 = c;
                for (int k = 0; k < 4; ++k) r3 = r3 + A[i * 4 + k] * B[k * 16 + j];
                const double d = D[i * 16 + j];
                nfail[0] += memcmp(&d, &r0, 8) != 0; nfail[1] += memcmp(&d, &r1, 8) != 0;
                nfail[2] += memcmp(&d, &r2, 8) != 0; nfail[3] += memcmp(&d, &r3, 8) != 0;
                ++total;
            }
    }
    printf("elements %d  mismatches: fma-chain-k-ascending %d, k-descending %d, wide-sum %d, unfused %d\n", total, nfail[0], nfail[1], nfail[2], nfail[3]);
    double *dout; hipMalloc(&dout, 8 * 256 * 1024);
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, dout, 10000);
    double r; hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost);
    printf("one wave: %.1f cycles per v_mfma_f64_16x16x4_f64 (4 independent accumulators)\n", r);
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(256), 0, 0, dout, 10000);
    hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost);
    printf("four waves on one CU: %.1f cycles per mfma per wave\n", r);
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(512), 0, 0, dout, 10000);
    hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost);
    printf("eight waves on one CU: %.1f cycles per mfma per wave\n", r);
    return 0;
}
