// Random-row reads of a table (developer tool; the access pattern of the 64-lane SCAM step: a wave reads one 8 KB row of the
// eigenvector table per step, the row index a hash of (wave, step)) -- what the L2 / MALL path delivers for tables that fit the
// XCD's 4 MB L2 and for tables that do not.  hipcc --offload-arch=gfx950 -O3 tools/row_gather_bw.hip -o /tmp/row_gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// W16: 16-byte loads (lane l takes bytes 16 l of each 1 KB piece) instead of 8-byte ones (bytes 8 l of each 512 B piece)
template <bool W16, int DEP /* loads in flight per wave: 16 = a whole row */>
__global__ __launch_bounds__(256) void gather(const double *tab, int nrows, int steps, double *out)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = 0.0;
    for (int s = 0; s < steps; ++s) {
        const unsigned k = __builtin_amdgcn_readfirstlane(hash(wave * 7919u + s) % (unsigned)nrows);
        const double *row = tab + (size_t)k * 1024;
        if (W16) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 *r2 = (const d2 *)row + lane;
            d2 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = r2[64 * e];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += v[e].x + v[e].y;
        } else {
            double v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = row[lane + 64 * e];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc += v[e];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main(int argc, char **argv)
{
    const int waves = 32768, steps = 100;
    double *tab, *out;
    (void)hipMalloc(&tab, sizeof(double) * 1024 * 4096);
    (void)hipMemset(tab, 0, sizeof(double) * 1024 * 4096);
    (void)hipMalloc(&out, sizeof(double) * waves * 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int nrows : {128, 256, 384, 512, 768, 1000, 2048, 4096}) {
        for (int w16 = 0; w16 < 2; ++w16) {
            for (int i = 0; i < 2; ++i) {
                if (i == 1) (void)hipEventRecord(e0);
                if (w16) gather<true, 16><<<waves / 4, 256>>>(tab, nrows, steps, out);
                else gather<false, 16><<<waves / 4, 256>>>(tab, nrows, steps, out);
            }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("table %5.1f MB (%4d rows of 8 KB)  %s loads: %.3f ms per %d wave-steps x %d waves = %.1f TB/s\n", nrows * 8192 / 1048576.0, nrows,
                   w16 ? "16-byte" : " 8-byte", ms, steps, waves, (double)waves * steps * 8192 / ms / 1e9);
        }
    }
    return 0;
}
