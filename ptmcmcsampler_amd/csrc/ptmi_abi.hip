// ptmi_abi.hip -- the C ABI of libptmi.so, the engine object and the kernels that are not per-chain templates
// (swap sweep, Welford / pooling, DE ring, self-tests).  See include/ptmi.h for the boundary and DESIGN.md.
#include <math.h>
#include <stdlib.h>
#include <dlfcn.h>

#include <new>
#include <mutex>
#include <vector>

#include <type_traits>

#include <rocblas/internal/rocblas-types.h>          // enum values only (ptmi_eig_sytrd calls the library through dlsym, nothing is linked)
#include <rocsolver/rocsolver-extra-types.h>
#include "ptmi_common.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int ptmi_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// --------------------------------------------------------------------- swap
__global__ void gather_lnl_kernel(const double *lnL, const int32_t *slot_of, double *out, long long n, int nt)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long w = i / nt;
    out[i] = lnL[w * nt + slot_of[i]];
}

// PT:666-686 in two kernels.  swap_prepare_kernel (one thread per position and walker) does everything that does
// not depend on the carried state: the LOGARITHM of the pair's uniform (PT:679's u <= exp(sum) is tested as log u <= sum,
// as the oracle defines it: the transcendental leaves the recurrence) and every quotient of a position's OWN likelihood,
// L[k] / T[k], L[k] / T[k+1] and L[k] / T[k-1].  swap_sweep_kernel (one lane per walker) then runs the hot -> cold
// recurrence with the carried map.  Scratch: one 48-byte record per (position, walker), position-major [n][W], so a pair
// costs the sweep three 16-byte loads per lane from one wave-uniform base and a wave reads 3 KB in a row.  When the whole
// ladder is local (slot_of != nullptr) the slot tables are rewritten in place: position k+1 becomes final at step k and
// positions <= k are still untouched.
struct __attribute__((aligned(16))) SwapPre {
    double lu, L;        // log of the pair's uniform; the position's likelihood
    double a, b;         // -L/T[k], L/T[k+1]
    double c;            // L/T[k-1]
    int32_t row, pad;    // the slot that holds the position (whole ladder local)
};
static_assert(sizeof(SwapPre) == 48, "three 16-byte loads");
// what a record is made from
struct SwapSrc {
    const double *ladder, *lnL_pos, *lnL_rows;
    const int32_t *slot_of;      // whole ladder local: the slot tables (else nullptr: lnL_pos holds the likelihoods by position)
    long long iter;
    u64 seed;
    int walker0;
    int block_nt;                // > 0: lnL_pos is [n / block_nt][W][block_nt], as all-gathered
    const double *u_over;        // TEST HOOK (ptmi_test_replay): the pair uniforms [W][n - 1] as recorded from the reference, or nullptr
};
__device__ __forceinline__ SwapPre swap_record(const SwapSrc &p, int W, int n, int k, int w)
{
    const bool fused = p.slot_of != nullptr;
    const int row = fused ? p.slot_of[(size_t)w * n + k] : 0;
    const double L = fused ? p.lnL_rows[(size_t)w * n + row]
                   : (p.block_nt > 0 ? p.lnL_pos[((size_t)(k / p.block_nt) * W + w) * p.block_nt + k % p.block_nt] : p.lnL_pos[(size_t)w * n + k]);
    double u = 0.0, b = 0.0, c = 0.0;
    if (k < n - 1) {
        const u32 sid = (u32)((u64)(p.walker0 + w) * (u32)n + 0u);    // rank 0's stream (PT:679)
        u64 w0, w1;
        philox_words(p.seed, (u64)p.iter, sid, SLOT_SWAP + (u32)k, w0, w1);
        u = det_log(p.u_over ? p.u_over[(size_t)w * (n - 1) + k] : w2uniform(w0));   // log of the [0,1) uniform; -inf for u = 0: always accepted
        b = L / p.ladder[k + 1];
    }
    if (k > 0) c = L / p.ladder[k - 1];
    SwapPre r;
    r.lu = u; r.L = L; r.a = -L / p.ladder[k]; r.b = b; r.c = c; r.row = fused ? row : k; r.pad = 0;
    return r;
}
__global__ void swap_prepare_kernel(int W, int n, SwapSrc src, SwapPre *pre)
{
    // grid (walkers, positions); the 2 M records of a 512-rank ladder are 100 MB: this kernel is bound by writing them
    const int k = (int)blockIdx.y, w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= W) return;
    pre[(size_t)k * W + (size_t)w] = swap_record(src, W, n, k, w);
}

// The recurrence of pair k (positions k, k+1; carried state of likelihood Lc at k+1) is the reference's four-term sum in
// its order, -L[k]/T[k] - Lc/T[k+1] + Lc/T[k] + L[k]/T[k+1], against log u.  The two quotients of Lc are carried along with
// it: if the pair accepts, Lc moves on and pair k-1 needs Lc/T[k] (this pair's third term) and Lc/T[k-1] -- ONE new division,
// independent of this pair's decision, so it runs in the shadow of the sums and the compare; if it rejects, the new carried
// state is position k's own and both quotients come from the prepared arrays (-(-L[k]/T[k]) and L[k]/T[k-1]; negation is
// exact).  Same operations on the same values as the oracle: bit-identical decisions.  Per pair the serial path is three
// sums, a compare and the selects (0.65 us per pair with two divisions and an exp on it, round 2).
// parity >= 0 (odd/even mode): only the pairs with k = parity (mod 2) are tried; an untried pair never accepts, so
// the carried state is always position k+1's own and the recurrence degenerates into independent pair tests.
// STG: the tables the sweep writes are [walker][position] -- a lane per walker scatters 4-byte stores 4 n bytes apart, 64
// memory transactions per store instruction.  So the block (one wave = 64 walkers) keeps its walkers' forward table and
// acceptance flags in LDS (rows of n + 1 ints: a lane per bank) and writes them out at the end with the lanes along the
// position, building the inverse table there.  2 x wpb x (n + 1) ints: 64 walkers per block up to 319 ranks, 32 / 16 / 8
// for longer ladders (512 ranks of an 8-GPU ladder: 32); beyond that the direct stores (STG = false).
#ifndef PTMI_SWEEP_BATCH
#define PTMI_SWEEP_BATCH 8
#endif
// the AM-buffer row of a swap iteration (PT:624-627, 327-328): the state that sits at rank 0 after the sweep
struct SwapAmRow { const double *X, *lnL, *lp; double *AM, *AMaux; int d, cov_update, am_epl; long long iter; AmFlag *AMflag; };
// the post-swap rows are KEY rows (AM row flags, ptmi_common.h)
__device__ __forceinline__ void swap_am_key(const SwapAmRow &amr, int w0, int nw, int tid, int nthreads)
{
    if (amr.AMflag == nullptr) return;
    const int ring = (int)(amr.iter % amr.cov_update);
    for (int wl = tid; wl < nw; wl += nthreads) amr.AMflag[(size_t)(w0 + wl) * amr.cov_update + (size_t)ring] = AMROW_KEY;
}
template <bool STG>
__global__ __launch_bounds__(STG ? 256 : 64) void swap_sweep_kernel(int W, int n, const double *ladder, const SwapPre *pre,
                                  int32_t *slot_of, int32_t *temp_of, int32_t *map, u64 *nswap, int local0, int nlocal,
                                  int parity, int32_t *inv /* with map: inv[w][map[w][j]] = j */,
                                  int wpb /* walkers per block: 64, fewer when a long ladder's tables would not fit the LDS */,
                                  int hop_nt, int32_t *hop_flag /* hop_nt > 0 (STG, map form): set *hop_flag when a state moves beyond a
                                                                 * neighbouring block of hop_nt ranks (ptmi_exchange_multihop) */,
                                  SwapAmRow amr /* STG, fused: the write-out also stores the swap iteration's AM row (am_write_kernel) */)
{
    // STG blocks have four waves: the first runs the recurrence (a lane per walker), all four write the tables out
    extern __shared__ int32_t sw_lds[];
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int w = (int)blockIdx.x * wpb + lane;
    const bool fused = slot_of != nullptr;
    const int ld = n + 1;
    int32_t *const l0 = sw_lds + (size_t)lane * ld;                    // slot_of / map of this lane's walker
    int32_t *const lf = sw_lds + (size_t)(wpb + lane) * ld;            // pair k accepted
    if (wave == 0 && lane < wpb && w < W) {
    int32_t *fw = STG ? l0 : (fused ? slot_of + (size_t)w * n : map + (size_t)w * n);     // forward table: row (fused) or source position
    int32_t *bw = STG ? nullptr : (fused ? temp_of + (size_t)w * n : inv + (size_t)w * n); // its inverse (STG: built at write-out)
    const SwapPre top = pre[(size_t)(n - 1) * W + w];
    int crow = top.row;                    // what the forward table says about the state carried at k+1 (its slot, or its position)
    double Lc = top.L;
    double q1 = -top.a;                    // Lc / T[k+1]
    double q0 = top.c;                     // Lc / T[k]
    // Only (Lc, q1, q0) are carried from pair to pair.  The scratch of SW pairs is fetched at once into one of two register
    // sets (this kernel runs one wave per SIMD: registers are free), the NEXT batch being requested before the current one
    // is worked through, so that one memory latency is exposed per launch instead of one per batch (round 2's version
    // requested T[k] through the scalar unit, one waited-for load per pair: 0.5 us per pair whatever the arithmetic).
    // Indices below 0 are clamped, not branched around: their values are never used.
    constexpr int SW = PTMI_SWEEP_BATCH;
    struct Batch { double u[SW], L[SW], a[SW], b[SW], c[SW], T[SW]; int r[SW]; };
    // addresses: the lane's record of position 0 (computed once) + a wave-uniform stride per position
    const char *const lane0 = reinterpret_cast<const char *>(pre + ((size_t)blockIdx.x * wpb + (unsigned)lane));
    const size_t kstride = (size_t)W * sizeof(SwapPre);
    auto fetch = [&](int k0, Batch &B) {
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int kk = k0 - j > 0 ? k0 - j : 0;
            const SwapPre r = *reinterpret_cast<const SwapPre *>(lane0 + (size_t)kk * kstride);
            B.u[j] = r.lu; B.L[j] = r.L; B.a[j] = r.a; B.b[j] = r.b; B.c[j] = r.c; B.r[j] = r.row;
            B.T[j] = ladder[kk > 0 ? kk - 1 : 0];                       // T[k-1] (uniform: a scalar load)
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chain = [&](int k0, const Batch &B) {
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int k = k0 - j;
            if (k < 0) break;
            const double spec = Lc / B.T[j];   // Lc / T[k-1]: needed if this pair accepts; does not wait for its decision
            double la = B.a[j];                // -L[k] / T[k]
            la += -q1;                         // -Lc / T[k+1]
            la += q0;                          //  Lc / T[k]
            la += B.b[j];                      //  L[k] / T[k+1]
            const bool acc = (parity < 0 || (k & 1) == parity) && B.u[j] <= la;      // log u <= sum
            // position k+1 is final: it keeps the carried state, or takes position k's
            const int fin = acc ? B.r[j] : crow;
            fw[k + 1] = fin;
            if (!STG) bw[fin] = k + 1;
            if (STG) lf[k] = acc ? 1 : 0;
            else if (acc && k >= local0 && k < local0 + nlocal) atomicAdd((unsigned long long *)&nswap[(size_t)w * n + k], 1ull);   // no-return atomic
            q1 = acc ? q0 : -B.a[j];
            q0 = acc ? spec : B.c[j];
            Lc = acc ? Lc : B.L[j];
            crow = acc ? crow : B.r[j];
        }
    };
    Batch A, B;
    fetch(n - 2, A);
    for (int k0 = n - 2; k0 >= 0; k0 -= 2 * SW) {
        if (k0 - SW >= 0) fetch(k0 - SW, B);
        chain(k0, A);
        if (k0 - SW < 0) break;
        if (k0 - 2 * SW >= 0) fetch(k0 - 2 * SW, A);
        chain(k0 - SW, B);
    }
    fw[0] = crow;
    if (!STG) bw[crow] = 0;
    }
    // block of a position (the multi-hop scan of the write-out): filled by the waves that sit out the recurrence
    int32_t *const blk = sw_lds + (size_t)2 * wpb * ld;
    if (STG && hop_nt > 0 && wave > 0)
        for (int k = (int)threadIdx.x - 64; k < n; k += 192) blk[k] = k / hop_nt;
    if (STG) {
        __syncthreads();
        const int w0 = (int)blockIdx.x * wpb;
        int32_t *g0 = fused ? slot_of : map, *g1 = fused ? temp_of : inv;
        const int nw = W - w0 < wpb ? W - w0 : wpb;
        bool far = false;
        for (int wl = wave; wl < nw; wl += 4) {                        // a wave per walker, the lanes along the position
            const size_t row = (size_t)(w0 + wl) * n;
            for (int k = lane; k < n; k += 64) {
                const int f = sw_lds[(size_t)wl * ld + k];
                g0[row + k] = f;
                g1[row + f] = k;                                       // the inverse table: a scatter inside the walker's own row
                if (hop_nt > 0) { const int hop = blk[f] - blk[k]; far = far || hop > 1 || hop < -1; }
                // a no-return atomic: fire and forget (a read-modify-write would wait for its load in every trip: 37 against 24 us)
                if (k < n - 1 && k >= local0 && k < local0 + nlocal && sw_lds[(size_t)(wpb + wl) * ld + k])
                    atomicAdd((unsigned long long *)&nswap[row + k], 1ull);
            }
        }
        if (hop_nt > 0 && __ballot(far) != 0 && lane == 0) atomicOr(hop_flag, 1);   // once per wave at most
        if (amr.AM != nullptr) {
            // the rows now at rank 0 into the AM ring (am_write_kernel's copy): the block's nw rows as one list of elements, six
            // reads in flight per thread (a wave per walker waited for sixteen round trips in turn)
            constexpr int NB = 6;
            const int tot = nw * amr.d, ring = (int)(amr.iter % amr.cov_update);
            for (int base = (int)threadIdx.x; base < tot; base += 256 * NB) {
                double v[NB];
                size_t dst[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int idx = base + 256 * u, ic = idx < tot ? idx : tot - 1;
                    const int wl = ic / amr.d, i = ic % amr.d;
                    const size_t r = (size_t)(w0 + wl) * n + (size_t)sw_lds[(size_t)wl * ld];
                    v[u] = amr.X[r * amr.d + i];
                    dst[u] = ((size_t)(w0 + wl) * amr.cov_update + (size_t)ring) * amr.d + (size_t)am_pos(i, amr.am_epl);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (base + 256 * u < tot) amr.AM[dst[u]] = v[u];
            }
            if (amr.AMaux)
                for (int wl = (int)threadIdx.x; wl < nw; wl += 256) {
                    const size_t r = (size_t)(w0 + wl) * n + (size_t)sw_lds[(size_t)wl * ld];
                    const size_t arow = (size_t)(w0 + wl) * amr.cov_update + (size_t)ring;
                    amr.AMaux[arow * 2] = amr.lnL[r];
                    amr.AMaux[arow * 2 + 1] = amr.lp[r];
                }
            swap_am_key(amr, w0, nw, (int)threadIdx.x, 256);
        }
    }
}

// The sweep with its records made in the block (no scratch in memory: the 48-byte records of a 512-rank ladder are 100 MB
// written and read back, 34 us of the 160 us a swap epoch takes on one of eight GPUs; with 64 ranks the prepare kernel and its
// launch gap are a third of the epoch).  Blocks of 512 threads: wave 0 runs the recurrence as in swap_sweep_kernel<true>, six
// of the others (not wave 4, which sits on the recurrence's SIMD) make the records of the batch after next (eight pairs) into a
// three-slot LDS ring while it works through the current one and reads the next into its second register set; one barrier
// per batch.  Same records, same recurrence, same write-out: bit-identical.
constexpr int SWF_BLK = 512;
__host__ __device__ inline size_t swf_ring_offset(int wpb, int n) { return ((sizeof(int32_t) * (2 * (size_t)wpb * (size_t)(n + 1) + (size_t)n)) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t swf_lds_bytes(int wpb, int n) { return swf_ring_offset(wpb, n) + sizeof(SwapPre) * (size_t)(3 * PTMI_SWEEP_BATCH + 1) * (size_t)wpb; }
__global__ __launch_bounds__(SWF_BLK) void swap_fused_kernel(int W, int n, SwapSrc src, int32_t *slot_of, int32_t *temp_of, int32_t *map,
                                                          u64 *nswap, int local0, int nlocal, int parity, int32_t *inv, int wpb, int wpb_log2,
                                                          int hop_nt, int32_t *hop_flag, SwapAmRow amr)
{
    extern __shared__ int32_t sw_lds[];
    constexpr int SW = PTMI_SWEEP_BATCH;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w0 = (int)blockIdx.x * wpb, w = w0 + lane;
    const bool fused = slot_of != nullptr;
    const int ld = n + 1;
    int32_t *const fw = sw_lds + (size_t)lane * ld;                    // slot_of / map of this lane's walker
    int32_t *const lf = sw_lds + (size_t)(wpb + lane) * ld;            // pair k accepted
    int32_t *const blk = sw_lds + (size_t)2 * wpb * ld;
    SwapPre *const ring = reinterpret_cast<SwapPre *>(reinterpret_cast<char *>(sw_lds) + swf_ring_offset(wpb, n));   // [3][SW][wpb]
    SwapPre *const topr = ring + (size_t)3 * SW * wpb;                 // [wpb]: the records of position n - 1
    const int NB = (n - 1 + SW - 1) / SW;                              // batches of the pairs n - 2 .. 0
    auto produce = [&](int b, int t0, int nthr) {                      // batch b by the threads t0 .. t0 + nthr - 1
        const int k0 = n - 2 - b * SW;
        for (int idx = tid - t0; idx < SW * wpb; idx += nthr) {
            const int j = idx >> wpb_log2, wl = idx & (wpb - 1), k = k0 - j;
            if (k >= 0 && w0 + wl < W) ring[((size_t)(b % 3) * SW + j) * wpb + wl] = swap_record(src, W, n, k, w0 + wl);
        }
    };
    for (int wl = tid; wl < wpb; wl += SWF_BLK)
        if (w0 + wl < W) topr[wl] = swap_record(src, W, n, n - 1, w0 + wl);
    if (NB > 0) produce(0, 0, SWF_BLK);
    if (NB > 1) produce(1, 0, SWF_BLK);
    if (hop_nt > 0)
        for (int k = tid; k < n; k += SWF_BLK) blk[k] = k / hop_nt;
    __syncthreads();
    const bool chainer = wave == 0 && lane < wpb && w < W;
    if (wave == 0) __builtin_amdgcn_s_setprio(3);                      // the recurrence is the critical path of the block
    int crow = 0;                          // what the forward table says about the state carried at k+1 (its slot, or its position)
    double Lc = 0.0, q1 = 0.0, q0 = 0.0;   // its likelihood, Lc / T[k+1], Lc / T[k]
    if (chainer) {
        const SwapPre top = topr[lane];
        crow = top.row; Lc = top.L; q1 = -top.a; q0 = top.c;
    }
    // The ring holds three batches: while the recurrence works through batch b out of one register set it reads batch b + 1
    // (made during batch b - 1) into the other, and the makers fill the slot of batch b + 2 (last read during batch b - 2).
    struct Batch { SwapPre R[SW]; double T[SW]; };
    auto fetch = [&](int b, Batch &B) {
        const int k0 = n - 2 - b * SW;
        const SwapPre *rb = ring + (size_t)(b % 3) * SW * wpb + lane;
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int kk = k0 - j > 0 ? k0 - j : 0;
            B.R[j] = rb[(size_t)(k0 - j >= 0 ? j : 0) * wpb];
            B.T[j] = src.ladder[kk > 0 ? kk - 1 : 0];                  // T[k-1] (uniform: a scalar load)
        }
    };
    auto chain = [&](int b, const Batch &B) {
        const int k0 = n - 2 - b * SW;
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int k = k0 - j;
            if (k < 0) break;
            const double spec = Lc / B.T[j];     // Lc / T[k-1]: needed if this pair accepts; does not wait for its decision
            double la = B.R[j].a;                // -L[k] / T[k]
            la += -q1;                           // -Lc / T[k+1]
            la += q0;                            //  Lc / T[k]
            la += B.R[j].b;                      //  L[k] / T[k+1]
            const bool acc = (parity < 0 || (k & 1) == parity) && B.R[j].lu <= la;     // log u <= sum
            fw[k + 1] = acc ? B.R[j].row : crow; // position k+1 is final: it keeps the carried state, or takes position k's
            lf[k] = acc ? 1 : 0;
            q1 = acc ? q0 : -B.R[j].a;
            q0 = acc ? spec : B.R[j].c;
            Lc = acc ? Lc : B.R[j].L;
            crow = acc ? crow : B.R[j].row;
        }
    };
    auto turn = [&](int b, Batch &cur, Batch &nxt) {                   // one batch: every wave passes here, one barrier
        if (wave == 0) {
            if (chainer) {
                if (b + 1 < NB) fetch(b + 1, nxt);
                chain(b, cur);
            }
        } else if (wave != 4 && b + 2 < NB) {                          // wave 4 shares the recurrence's SIMD: it sits the batches out
            produce(b + 2, wave < 4 ? 64 : 128, SWF_BLK - 128);
        }
        __syncthreads();
    };
    Batch A, B;
    if (chainer && NB > 0) fetch(0, A);
    for (int b = 0; b < NB; b += 2) {
        turn(b, A, B);
        if (b + 1 < NB) turn(b + 1, B, A);
    }
    if (chainer) fw[0] = crow;
    __syncthreads();
    // write-out: a wave per walker, the lanes along the position (as swap_sweep_kernel<true>)
    int32_t *g0 = fused ? slot_of : map, *g1 = fused ? temp_of : inv;
    const int nw = W - w0 < wpb ? W - w0 : wpb;
    bool far = false;
    for (int wl = wave; wl < nw; wl += SWF_BLK / 64) {
        const size_t row = (size_t)(w0 + wl) * n;
        for (int k = lane; k < n; k += 64) {
            const int f = sw_lds[(size_t)wl * ld + k];
            g0[row + k] = f;
            g1[row + f] = k;                                           // the inverse table: a scatter inside the walker's own row
            if (hop_nt > 0) { const int hop = blk[f] - blk[k]; far = far || hop > 1 || hop < -1; }
            if (k < n - 1 && k >= local0 && k < local0 + nlocal && sw_lds[(size_t)(wpb + wl) * ld + k])
                atomicAdd((unsigned long long *)&nswap[row + k], 1ull);
        }
    }
    if (hop_nt > 0 && __ballot(far) != 0 && lane == 0) atomicOr(hop_flag, 1);   // once per wave at most
    if (amr.AM != nullptr) {                                           // the rows now at rank 0 into the AM ring (as swap_sweep_kernel<true>)
        constexpr int NB6 = 6;
        const int tot = nw * amr.d, ringrow = (int)(amr.iter % amr.cov_update);
        for (int base = tid; base < tot; base += SWF_BLK * NB6) {
            double v[NB6];
            size_t dst[NB6];
#pragma unroll
            for (int u = 0; u < NB6; ++u) {
                const int idx = base + SWF_BLK * u, ic = idx < tot ? idx : tot - 1;
                const int wl = ic / amr.d, i = ic % amr.d;
                const size_t r = (size_t)(w0 + wl) * n + (size_t)sw_lds[(size_t)wl * ld];
                v[u] = amr.X[r * amr.d + i];
                dst[u] = ((size_t)(w0 + wl) * amr.cov_update + (size_t)ringrow) * amr.d + (size_t)am_pos(i, amr.am_epl);
            }
#pragma unroll
            for (int u = 0; u < NB6; ++u)
                if (base + SWF_BLK * u < tot) amr.AM[dst[u]] = v[u];
        }
        if (amr.AMaux)
            for (int wl = tid; wl < nw; wl += SWF_BLK) {
                const size_t r = (size_t)(w0 + wl) * n + (size_t)sw_lds[(size_t)wl * ld];
                const size_t arow = (size_t)(w0 + wl) * amr.cov_update + (size_t)ringrow;
                amr.AMaux[arow * 2] = amr.lnL[r];
                amr.AMaux[arow * 2 + 1] = amr.lp[r];
            }
        swap_am_key(amr, w0, nw, tid, SWF_BLK);
    }
}

// Odd/even mode with the whole ladder local: one thread per (walker, tried pair), the slot tables rewritten in place
// (the pairs are disjoint).  The pair test is the sweep's, term by term.
__global__ void swap_oddeven_kernel(int W, int n, const double *ladder, const double *lnL_rows, int32_t *slot_of,
                                    int32_t *temp_of, u64 *nswap, long long iter, u64 seed, int walker0, int parity)
{
    const int npairs = (n - parity) / 2;                    // k = parity, parity + 2, ... <= n - 2
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)npairs * W) return;
    const int w = (int)(idx / npairs), k = parity + 2 * (int)(idx % npairs);
    int32_t *so = slot_of + (size_t)w * n, *to = temp_of + (size_t)w * n;
    const int rk = so[k], rk1 = so[k + 1];
    const double Lk = lnL_rows[(size_t)w * n + rk], Lk1 = lnL_rows[(size_t)w * n + rk1];
    const u32 sid = (u32)((u64)(walker0 + w) * (u32)n + 0u);
    u64 w0, w1;
    philox_words(seed, (u64)iter, sid, SLOT_SWAP + (u32)k, w0, w1);
    const double Tk = ladder[k], Tk1 = ladder[k + 1];
    double la = -Lk / Tk;
    la += -Lk1 / Tk1;
    la += Lk1 / Tk;
    la += Lk / Tk1;
    if (det_log(w2uniform(w0)) <= la) {
        so[k] = rk1;
        so[k + 1] = rk;
        to[rk1] = k;
        to[rk] = k + 1;
        nswap[(size_t)w * n + k] += 1;
    }
}

// AM-buffer row of a swap iteration: the state that now sits at rank 0 (PT:624-627, 327-328)
__global__ void am_write_kernel(const double *X, const double *lnL, const double *lp, const int32_t *slot_of, double *AM,
                                double *AMaux, int W, int nt, int d, int cov_update, long long iter, int am_epl, AmFlag *AMflag)
{
    const int w = (int)blockIdx.x;
    const size_t r = (size_t)w * nt + slot_of[(size_t)w * nt];
    const double *row = X + r * d;
    double *am = AM + ((size_t)w * cov_update + (size_t)(iter % cov_update)) * d;
    for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) am[am_pos(i, am_epl)] = row[i];
    if (AMaux && threadIdx.x == 0) {
        double *ax = AMaux + ((size_t)w * cov_update + (size_t)(iter % cov_update)) * 2;
        ax[0] = lnL[r];
        ax[1] = lp[r];
    }
    if (AMflag && threadIdx.x == 0) AMflag[(size_t)w * cov_update + (size_t)(iter % cov_update)] = AMROW_KEY;
}

// ------------------------------------------------------------------ Welford
// PT:769-794.  Block (walker, tile_i, tile_j) keeps a 112x112 tile of M2 in registers
// (7x7 per thread, 16x16 threads) and streams the `mem` buffered rows; the running mean
// of the tile's i- and j-slices is recomputed by the first 2x112 threads and handed to
// the tile through LDS.  Every element sees exactly the reference's operation order:
// M2[i][j] += diff[i] * (row[j] - mu_new[j]), one product and one sum, rows ascending.
constexpr int WT = 7, WTILE = 16 * WT;
template <bool FUSED>
__global__ __launch_bounds__(256) void welford_kernel(const double *AM, double *mu, double *M2, double *cov, int d, int mem,
                                                     long long iter, int cov_stride_per_walker, int am_epl)
{
    constexpr int PF = 8;       // rows fetched ahead by the carrier threads (one HBM latency per PF rows)
    constexpr int RB = 4;       // rows handed to the tile per barrier
    __shared__ double sh[2][RB][2][WTILE];  // [buf][row][diff|e][WTILE]
    if (FUSED && blockIdx.y > blockIdx.x) return;       // pooled definition: the lower triangle is mirrored afterwards
    const int w = (int)blockIdx.z;
    const int ti0 = (int)blockIdx.y * WTILE, tj0 = (int)blockIdx.x * WTILE;
    const int tx = (int)(threadIdx.x & 15), ty = (int)(threadIdx.x >> 4);
    const double *am = AM + (size_t)w * mem * d;
    double *muw = mu + (size_t)w * d, *M2w = M2 + (size_t)w * d * d;
    long long it = iter - mem;
    const bool reset = it == 0;

    // threads 0..111 carry mu of the i-slice, 112..223 of the j-slice
    const int role = (int)threadIdx.x / WTILE, ridx = (int)threadIdx.x % WTILE;
    const int rel = role == 0 ? ti0 + ridx : tj0 + ridx;
    const bool carrier = role < 2 && rel < d;
    double m = carrier && !reset ? muw[rel] : 0.0;

    double acc[WT][WT];
#pragma unroll
    for (int p = 0; p < WT; ++p)
#pragma unroll
        for (int r = 0; r < WT; ++r) {
            const int i = ti0 + ty + 16 * p, j = tj0 + tx + 16 * r;
            acc[p][r] = (!reset && i < d && j < d) ? M2w[(size_t)i * d + j] : 0.0;
        }
    int bsel = 0;
    for (int ii0 = 0; ii0 < mem; ii0 += PF) {
        double vpre[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) vpre[u] = (carrier && ii0 + u < mem) ? am[(size_t)(ii0 + u) * d + am_pos(rel < d ? rel : 0, am_epl)] : 0.0;
#pragma unroll
        for (int u0 = 0; u0 < PF; u0 += RB) {
            if (ii0 + u0 >= mem) break;
            // the carriers advance the running mean through RB rows (the recurrence is theirs alone) ...
            if (role < 2) {
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    double df = 0.0, ev = 0.0;
                    if (carrier && ii0 + u0 + u < mem) {
                        const double v = vpre[u0 + u];
                        df = v - m;
                        if (FUSED) m = m + df * (1.0 / (double)(it + 1 + u));
                        else m += df / (double)(it + 1 + u);
                        ev = v - m;
                    }
                    if (role == 0) sh[bsel][u][0][ridx] = df;
                    else sh[bsel][u][1][ridx] = ev;
                }
            }
            __syncthreads();
            // ... and the whole tile applies the RB rank-1 updates, rows ascending (rows past the end carry zeros)
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                double dv[WT], evv[WT];
#pragma unroll
                for (int p = 0; p < WT; ++p) dv[p] = sh[bsel][u][0][ty + 16 * p];
#pragma unroll
                for (int r = 0; r < WT; ++r) evv[r] = sh[bsel][u][1][tx + 16 * r];
                if (ii0 + u0 + u < mem) {
#pragma unroll
                    for (int p = 0; p < WT; ++p)
#pragma unroll
                        for (int r = 0; r < WT; ++r) {
                            if (FUSED) acc[p][r] = __builtin_fma(dv[p], evv[r], acc[p][r]);   // pooled mode: not a reference replica
                            else acc[p][r] += dv[p] * evv[r];                                 // PT:792, one product and one sum
                        }
                }
            }
            const int done = mem - (ii0 + u0) < RB ? mem - (ii0 + u0) : RB;
            it += done;
            bsel ^= 1;
        }
    }
    const double den = (double)(it - 1);
    double *covw = cov ? cov + (size_t)w * cov_stride_per_walker : nullptr;
#pragma unroll
    for (int p = 0; p < WT; ++p)
#pragma unroll
        for (int r = 0; r < WT; ++r) {
            const int i = ti0 + ty + 16 * p, j = tj0 + tx + 16 * r;
            if (i < d && j < d) {
                M2w[(size_t)i * d + j] = acc[p][r];
                if (covw) covw[(size_t)i * d + j] = acc[p][r] / den;
            }
        }
    // Several tiles per walker: mu is advanced by welford_mean_kernel, launched after this one (every tile needs
    // the old mean).  One tile: this block's carriers hold the new mean already.
    if (gridDim.x == 1 && role == 0 && carrier) muw[rel] = m;
}

// One tile per walker (d <= TYN * PR, the per-walker replica mode at ndim <= 100): the same recurrence with the work split by waves.
// welford_kernel's carriers are threads of the tile's own four waves, so every wave pays for the mean's division (a dozen
// instructions beside the 98 of a row) and a quarter of its 7 x 7 accumulators lie outside a 100 x 100 matrix: 5.1 ms per epoch at
// 4096 x 1000 x 100 against 2.1 ms of products and sums.  Here threads 0 .. TYN TXN - 1 hold PR x PC accumulators each (25 x 10
// threads of 4 x 10: no padding at d = 100, the row's factors read from LDS as 16-byte pieces), and two more waves carry the mean:
// thread 256 + i advances mu[i] through the rows one batch AHEAD of the tile (double-buffered in LDS, one barrier per RB rows).
// The division df / n (n = the row count, the same for every lane) is Markstein's: r = 1 / n by a true division once per batch
// and lane, then q = df r corrected twice through the exact remainder fma(-n, q, df) -- the correctly rounded quotient (r is the
// correctly rounded reciprocal of an integer below 2^53; 4e8 random and near-tie cases against a / n: tests/test_welford_div.py
// runs the same arithmetic on the host).
__device__ __forceinline__ double div_by_count(double a, double n, double r)
{
    if (__builtin_fabs(a) < 0x1p-900) return a / n;         // a remainder in the subnormal range would not be exact
    double q = a * r;
    double e = __builtin_fma(-n, q, a);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-n, q, a);
    return __builtin_fma(e, r, q);
}
template <int TYN, int TXN, int PR, int PC>
__global__ __launch_bounds__(768) void welford_rows_kernel(const double *AM, double *mu, double *M2, double *cov, int d, int mem, long long iter,
                                                           int cov_stride_per_walker, int am_epl, int nwalkers)
{
    // a block = TWO walkers: 2 x 4 tile waves + 2 x 2 carrier waves = three waves on every SIMD (up to 170 registers each); a block
    // of one walker's six waves lands 2 + 2 + 1 + 1 on the SIMDs and a second block no longer fits beside it
#ifndef PTMI_WF_RB
#define PTMI_WF_RB 8
#endif
    constexpr int RB = PTMI_WF_RB, LW = TYN * PR > TXN * PC ? TYN * PR : TXN * PC;
    __shared__ __attribute__((aligned(16))) double sh_all[2][2][RB][2][LW];      // [walker of the block][buffer][row][diff | e][element]
    const int tb = (int)threadIdx.x;
    const int slot = tb < 512 ? tb >> 8 : (tb - 512) >> 7;
    const int w = 2 * (int)blockIdx.x + slot;
    const bool wact = w < nwalkers;
    double (*sh)[RB][2][LW] = sh_all[slot];
    const double *am = AM + (size_t)(wact ? w : 0) * mem * d;
    double *muw = mu + (size_t)(wact ? w : 0) * d, *M2w = M2 + (size_t)(wact ? w : 0) * d * d;
    const long long it0 = iter - mem;
    const bool reset = it0 == 0;
    const int t = tb < 512 ? (tb & 255) : 256 + ((tb - 512) & 127);      // 0 .. 255 the tile, 256 .. 383 the carriers of this walker
    const int nb = (mem + RB - 1) / RB;
    if (t >= 256) {
        // ------------------------------------------------ the mean's carriers
        const int rel = t - 256;
        const bool carrier = rel < d && wact;
        const int pos = am_pos(carrier ? rel : 0, am_epl);
        double m = carrier && !reset ? muw[rel] : 0.0;
        double vc[RB], vn[RB];
        auto load = [&](int b, double (&v)[RB]) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int row = b * RB + u;
                v[u] = (carrier && row < mem) ? am[(size_t)row * d + pos] : 0.0;
            }
        };
        auto produce = [&](int b, const double (&v)[RB]) {
            const double nl = (double)(it0 + (long long)b * RB + 1 + (t & (RB - 1))), rl = 1.0 / nl;      // lane u of every RB: row u's count
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const double n = __shfl(nl, u, RB), r = __shfl(rl, u, RB);
                double df = 0.0, ev = 0.0;
#ifdef PTMI_WF_NOCARRY
                df = v[u] + n; ev = v[u] + r;
#else
                if (b * RB + u < mem) {
                    df = v[u] - m;
                    m += div_by_count(df, n, r);            // PT:787 (mean += diff / n)
                    ev = v[u] - m;
                }
#endif
                if (rel < LW) { sh[b & 1][u][0][rel] = carrier ? df : 0.0; sh[b & 1][u][1][rel] = carrier ? ev : 0.0; }
            }
        };
        load(0, vc);
        if (nb > 1) load(1, vn);
        produce(0, vc);
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            if (b + 1 < nb) {
#pragma unroll
                for (int u = 0; u < RB; ++u) vc[u] = vn[u];
                if (b + 2 < nb) load(b + 2, vn);
                produce(b + 1, vc);
            }
            __syncthreads();
        }
        if (carrier && wact) muw[rel] = m;
        return;
    }
    // ---------------------------------------------------- the tile
    const int ty = t / TXN, tx = t % TXN;
    const bool live = t < TYN * TXN && wact;
    double acc[PR][PC];
#pragma unroll
    for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int r = 0; r < PC; ++r) {
            const int i = ty * PR + p, j = tx * PC + r;
            acc[p][r] = (live && !reset && i < d && j < d) ? M2w[(size_t)i * d + j] : 0.0;
        }
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        if (live) {
            // the next row's factors are requested before this row's products (two register sets): an LDS round trip per row is
            // not covered by the one or two other waves of the SIMD
            double dv[2][PR], ev[2][PC];
            auto rd = [&](int u, int k) {
#pragma unroll
                for (int p = 0; p < PR; ++p) dv[k][p] = sh[b & 1][u][0][ty * PR + p];
#pragma unroll
                for (int r = 0; r < PC; ++r) ev[k][r] = sh[b & 1][u][1][tx * PC + r];
            };
            // PT:792, one product and one sum per element, rows ascending (fencing a row's products off from their sums cost 37
            // spilled registers and 2 ms).  Rows past the end of the buffer carry zeros from the carriers (acc + 0 * 0: an accumulator
            // is never -0), so the batch is straight-line code.
            auto row = [&](int k) {
#pragma unroll
                for (int p = 0; p < PR; ++p)
#pragma unroll
                    for (int r = 0; r < PC; ++r) acc[p][r] += dv[k][p] * ev[k][r];
            };
            rd(0, 0);
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (u + 1 < RB) {
                    rd(u + 1, (u + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                row(u & 1);
            }
        }
        __syncthreads();
    }
    if (!live) return;
    const double den = (double)(iter - 1);
    double *covw = cov ? cov + (size_t)w * cov_stride_per_walker : nullptr;
#pragma unroll
    for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int r = 0; r < PC; ++r) {
            const int i = ty * PR + p, j = tx * PC + r;
            if (i < d && j < d) M2w[(size_t)i * d + j] = acc[p][r];
        }
    // the covariance in a rolled loop over the thread's own elements, read back (forty unrolled divisions beside the accumulators
    // cost the main loop 44 spilled registers)
    if (covw) {
#pragma unroll 1
        for (int e = 0; e < PR * PC; ++e) {
            const int i = ty * PR + e / PC, j = tx * PC + e % PC;
            if (i < d && j < d) covw[(size_t)i * d + j] = M2w[(size_t)i * d + j] / den;
        }
    }
}

// ---------------------------------------------------------------- pooled covariance
// cov_mode "pooled" (one covariance adapted from all walkers' rank-0 samples; oracle: orc_pool_update).  The epoch's chunk
// -- the W x cov_update buffered rows as one [rows][d] matrix -- enters as shifted sums T = sum dx dx^T, t = sum dx with
// dx = x - (the running pooled mean): a symmetric rank-k update with no recurrence in it, so the rows go straight from
// memory through LDS into v_mfma_f64_16x16x4_f64, whose k-ascending fma chain is the oracle's row-ascending definition.
// The column sums come out of the same instructions: column d of dx is the constant 1.
// Grid: (macro tile, slab): the tiles of a slab are neighbours in launch order, so that its rows, wanted by every one of them, are
// read from memory once and from the caches after that.  A slab is a contiguous run of rows (pool_slab walkers); a macro tile is 112 x 112 outputs
// (7 x 7 matrix tiles) of the columns [112 I, 112 I + 112) x [112 J, 112 J + 112), I <= J.  DIAG (I == J): the 28 tiles with
// ti <= tj, seven per wave; else all 49, 13 / 12 / 12 / 12.  Rows are staged 32 at a time (eight k-steps) in a double-buffered LDS
// chunk, the next chunk's global loads in flight during the matrix work; one barrier per chunk.  Each block writes its
// slab's partial sums; pool_reduce_kernel adds the slabs in order, pool_finish_kernel applies Chan's formula.
// Round 2 ran a per-walker Welford recurrence on the matrix cores (1.45 ms per epoch at 4096 x 1000 x 100, two carrier waves
// feeding the recurrence) + a 328 MB two-level combination (0.12 ms); d > 112 had no matrix-core path at all (26 ms at d = 1000).
constexpr int PS_W = 112;       // columns of a macro tile
// rows per staged chunk: 32 (8 k-steps x 7 tiles) on the diagonal, 16 (4 k-steps x 13 tiles) off it: some 55 matrix instructions
// per wave cover a memory round trip, and two blocks share a CU (57 KB of LDS each)
constexpr int ps_rc(bool diag) { return diag ? 32 : 16; }
typedef double ps_d4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int pool_groups(int d) { return (d + 1 + PS_W - 1) / PS_W; }          // macro tiles per side (columns 0 .. d)
// walkers per slab: up to 512 slabs when one macro tile covers the matrix, up to 32 beyond (a partial is d (d + 1) doubles);
// part of the definition (summation order): oracle/oracle.py pool_slab is the same rule
static inline int pool_slab(int nwalkers, int d) { const int target = d + 1 <= PS_W ? 512 : 32; const int s = (nwalkers + target - 1) / target; return s < 1 ? 1 : s; }
// RLE (AM row flags, ptmi_common.h): a slab's matrix is not its rows but its STORED rows, each scaled by the square root of the
// length of its run (pool_rle_kernel lists them: PoolEnt = the row inside the slab and sqrt(run length)): sum over runs of n dx dx^T
// as (sqrt(n) dx) (sqrt(n) dx)^T, column d of the staged matrix holding sqrt(n) so that the column sums come out as sum n dx.
// The oracle defines the same sums (orc_pool_update_rle): 43 % fewer rows to read, stage and multiply at the stationary
// acceptance of a SCAM cycle.  A stager needs the list entry before it can ask for the row: the entries of chunk i + 2 are
// requested while the rows of chunk i + 1 are in flight and chunk i is multiplied.
// Round 4, second form (the first took 1.22 ms per epoch at 4096 x 1000 x 100 against 0.42 ms of matrix instructions: 350 vector
// instructions of 64-bit index arithmetic and selects per wave and chunk beside the 56 matrix ones, which the f64 matrix pipe does
// not overlap): the macro tile lives in POSITION space -- column p of the staged matrix is position p of a buffered row, whatever
// parameter am_pos put there; an element's k-ascending fma chain does not care where its column sits, the epilogue maps the
// pair back with am_inv -- so a stager takes 16 bytes of a row per load (PAIR: d even) at a 32-bit offset from the slab's base;
// a dead row is a zero weight, not a select per element; the waves' tiles are compile-time lists (row-major runs: a wave's
// fragments are read from LDS once per k-step on the diagonal, 7 instead of 14).
struct PoolEnt { double wgt; int32_t src; int32_t pad; };
struct PoolRle { const PoolEnt *ent; const int32_t *cnt; };
typedef double ps_d2 __attribute__((ext_vector_type(2)));
// tiles of wave wv: on the diagonal tile rows wv and 7 - wv (7 tiles each wave), off it the row-major run [1 + 12 wv, ...) of the 49
constexpr int ps_nt(bool diag, int wv) { return diag ? 7 : (wv == 0 ? 13 : 12); }
__host__ __device__ constexpr int ps_t0(int wv) { return wv == 0 ? 0 : 1 + 12 * wv; }
constexpr int ps_ti(bool diag, int wv, int n) { return diag ? (n < 7 - wv ? wv : 7 - wv) : (ps_t0(wv) + n) / 7; }
constexpr int ps_tj(bool diag, int wv, int n) { return diag ? (n < 7 - wv ? wv + n : n) : (ps_t0(wv) + n) % 7; }
template <bool DIAG, int WV>
__device__ __forceinline__ void ps_mma(const double *Ab, const double *Bb, ps_d4 (&acc)[DIAG ? 7 : 13])
{
#pragma unroll
    for (int k0 = 0; k0 < ps_rc(DIAG); k0 += 4)
#pragma unroll
        for (int n = 0; n < ps_nt(DIAG, WV); ++n)
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[k0 * PS_W + 16 * ps_ti(DIAG, WV, n)], Bb[k0 * PS_W + 16 * ps_tj(DIAG, WV, n)], acc[n], 0, 0, 0);
}
template <bool DIAG, int WV>
__device__ __forceinline__ void ps_store(const ps_d4 (&acc)[DIAG ? 7 : 13], double *out, int I, int J, int d, int am_epl, int g, int c)
{
#pragma unroll
    for (int n = 0; n < ps_nt(DIAG, WV); ++n) {
        const int ti = ps_ti(DIAG, WV, n), tj = ps_tj(DIAG, WV, n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pi = I * PS_W + 16 * ti + g + 4 * r, pj = J * PS_W + 16 * tj + c;
            if (pi < d && pj <= d && (!(DIAG && ti == tj) || pi <= pj)) {
                const int ci = am_inv(pi, am_epl), cj = pj < d ? am_inv(pj, am_epl) : d;
                const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
                out[(size_t)lo * (d + 1) + hi] = acc[n][r];
            }
        }
    }
}
template <bool DIAG, bool RLE, bool PAIR>
__global__ __launch_bounds__(256, 2) void pool_syrk_kernel(const double *rows, long long nrows, int d, const double *shift,
                                                                     long long rows_per_slab, double *part, int am_epl,
                                                                     int shift_epl /* row format of `shift` (an AM row at the first epoch) */,
                                                                     PoolRle rl)
{
    constexpr int NTW = DIAG ? 7 : 13, NA = DIAG ? 1 : 2, PS_RC = ps_rc(DIAG), NU = PS_RC / 4;
    __shared__ double Dl[NA][2][PS_RC][PS_W];
    const int lane = (int)(threadIdx.x & 63), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 15, g = lane >> 4;
    const int ng = pool_groups(d);
    int I = (int)blockIdx.x, J = (int)blockIdx.x;
    if (!DIAG) {                                            // blockIdx.x enumerates the pairs I < J row by row
        int p = (int)blockIdx.x;
        I = 0;
        while (p >= ng - 1 - I) { p -= ng - 1 - I; ++I; }
        J = I + 1 + p;
    }
    const long long beg = (long long)blockIdx.y * rows_per_slab;
    // the loop runs over the slab's LIST: RLE its stored rows (entries 0 .. count - 1 of ent), else its rows
    const int nlist = RLE ? rl.cnt[blockIdx.y] : (int)(beg + rows_per_slab < nrows ? rows_per_slab : nrows - beg);
    const char *slab = (const char *)(rows + beg * d);      // byte offsets inside a slab fit 32 bits (checked by the host)
    const PoolEnt *ent = RLE ? rl.ent + beg : nullptr;
    ps_d4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = ps_d4{0.0, 0.0, 0.0, 0.0};
    // off the diagonal a wave's tiles are the run [ps_t0(wave), ...) of the 49, their fragments' offsets in scalar registers (four
    // compile-time copies of 13 accumulators' code spilled 93 registers)
    int offa[DIAG ? 1 : NTW], offb[DIAG ? 1 : NTW];
    if constexpr (!DIAG) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int t = ps_t0(wave) + n < 49 ? ps_t0(wave) + n : 48;
            offa[n] = __builtin_amdgcn_readfirstlane(16 * (t / 7));
            offb[n] = __builtin_amdgcn_readfirstlane(16 * (t % 7));
        }
    }
    // staging: a wave's lanes 16 r4 + (q & 15) own the position pair (2 q, 2 q + 1) of the macro tile(s) and rows r4, r4 + 4, ... of a
    // chunk: a load instruction takes 256 contiguous bytes of four rows, an LDS store fills all the banks
    const int q = 16 * wave + c, r4 = g;
    const bool stager = q < PS_W / 2;
    unsigned colb[NA][2];           // byte offset of the elements' positions inside a row (clamped into it)
    double sh[NA][2], one[NA][2];
    bool dat[NA][2];
    bool plain = true;
#pragma unroll
    for (int a2 = 0; a2 < NA; ++a2)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = (a2 == 0 ? I : J) * PS_W + 2 * q + e;
            dat[a2][e] = p < d;
            const int pc = PAIR ? (p - e < d ? p : d - 2 + e) : (p < d ? p : d - 1);
            colb[a2][e] = 8u * (unsigned)pc;
            sh[a2][e] = (stager && p < d) ? shift[am_pos(am_inv(p, am_epl), shift_epl)] : 0.0;
            one[a2][e] = p == d ? 1.0 : 0.0;
            plain = plain && (dat[a2][e] || !stager);
        }
    const bool wave_plain = __all(plain);                   // no ones / padding column among this wave's stagers: no selects at all
    const unsigned rstride = 8u * (unsigned)d;
    // Loads are unconditional (list index and column clamped: a branch per load made every one of them wait for its own round
    // trip, 2.6 ms per epoch); what a slot really holds is decided when it is staged.
    ps_d2 v[NA][NU];
    double wg[NU];                 // weight of the rows in v: sqrt(run length) (RLE) or 1, zero past the end of the list
    int nsrc[RLE ? NU : 1];        // RLE: rows and weights of the chunk after the one in v
    double nwg[RLE ? NU : 1];
    auto list = [&](int j0) {      // RLE: request the list entries of the chunk that starts at entry j0
        if constexpr (RLE) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = j0 + 4 * u + r4, jc = j < nlist ? j : nlist - 1;
                const ps_d2 e2 = *(const ps_d2 *)(ent + jc);
                nwg[u] = e2.x;
                nsrc[u] = (int)(__double_as_longlong(e2.y) & 0xffffffffll);
            }
        }
    };
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = j0 + 4 * u + r4, jc = j < nlist ? j : nlist - 1;
            const unsigned rb = (unsigned)(RLE ? nsrc[u] : jc) * rstride;
            wg[u] = j < nlist ? (RLE ? nwg[u] : 1.0) : 0.0;
#pragma unroll
            for (int a2 = 0; a2 < NA; ++a2) {
                if constexpr (PAIR) v[a2][u] = *(const ps_d2 *)(slab + (rb + colb[a2][0]));
                else {
                    v[a2][u].x = *(const double *)(slab + (rb + colb[a2][0]));
                    v[a2][u].y = *(const double *)(slab + (rb + colb[a2][1]));
                }
            }
        }
    };
    auto stage = [&](int buf) {
        if (!stager) return;
        if (wave_plain) {
#pragma unroll
            for (int a2 = 0; a2 < NA; ++a2)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    ps_d2 x;
                    x.x = (v[a2][u].x - sh[a2][0]) * wg[u];
                    x.y = (v[a2][u].y - sh[a2][1]) * wg[u];
                    *(ps_d2 *)&Dl[a2][buf][4 * u + r4][2 * q] = x;
                }
        } else {
#pragma unroll
            for (int a2 = 0; a2 < NA; ++a2)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    ps_d2 x;
                    x.x = (dat[a2][0] ? v[a2][u].x - sh[a2][0] : one[a2][0]) * wg[u];
                    x.y = (dat[a2][1] ? v[a2][u].y - sh[a2][1] : one[a2][1]) * wg[u];
                    *(ps_d2 *)&Dl[a2][buf][4 * u + r4][2 * q] = x;
                }
        }
    };
    if (nlist > 0) {
        list(0);
        fetch(0);
        list(PS_RC);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (int j0 = 0; j0 < nlist; j0 += PS_RC) {
        const bool more = j0 + PS_RC < nlist;
        if (more) {
            fetch(j0 + PS_RC);
            list(j0 + 2 * PS_RC);
        }
        const double *Ab = &Dl[0][buf][0][0] + g * PS_W + c, *Bb = &Dl[NA - 1][buf][0][0] + g * PS_W + c;
        if constexpr (DIAG) {
            switch (wave) {
            case 0: ps_mma<DIAG, 0>(Ab, Bb, acc); break;
            case 1: ps_mma<DIAG, 1>(Ab, Bb, acc); break;
            case 2: ps_mma<DIAG, 2>(Ab, Bb, acc); break;
            default: ps_mma<DIAG, 3>(Ab, Bb, acc); break;
            }
        } else {
#pragma unroll
            for (int k0 = 0; k0 < PS_RC; k0 += 4)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    if (n < 12 || wave == 0) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[k0 * PS_W + offa[n]], Bb[k0 * PS_W + offb[n]], acc[n], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    double *out = part + (size_t)blockIdx.y * d * (d + 1);
    if constexpr (DIAG) {
        switch (wave) {
        case 0: ps_store<DIAG, 0>(acc, out, I, J, d, am_epl, g, c); break;
        case 1: ps_store<DIAG, 1>(acc, out, I, J, d, am_epl, g, c); break;
        case 2: ps_store<DIAG, 2>(acc, out, I, J, d, am_epl, g, c); break;
        default: ps_store<DIAG, 3>(acc, out, I, J, d, am_epl, g, c); break;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            if (!(n < 12 || wave == 0)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pi = I * PS_W + offa[n] + g + 4 * r, pj = J * PS_W + offb[n] + c;
                if (pi < d && pj <= d) {
                    const int ci = am_inv(pi, am_epl), cj = pj < d ? am_inv(pj, am_epl) : d;
                    const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
                    out[(size_t)lo * (d + 1) + hi] = acc[n][r];
                }
            }
        }
    }
}

// The list of a slab's stored rows (AM row flags) and the square roots of their run lengths: entry j of slab s (at beg + j) is the
// j-th row of the slab whose flag word says NEW or KEY; its run ends where the next stored row begins (ring row 0 of every walker is
// a KEY row, so a run never crosses into another walker's ring) or at the slab's end.  One block per slab, rows in order.
__global__ __launch_bounds__(256) void pool_rle_kernel(const AmFlag *flag, long long nrows, long long rows_per_slab, PoolEnt *ent, int32_t *cnt)
{
    // eight groups of 256 rows per trip: the flags of a trip are requested together and its ballots share two barriers (a load,
    // a ballot and three barriers per 256 rows made the kernel the latency of 31 round trips: 59 us per epoch at 4096 x 1000)
    constexpr int NG = 8;
    __shared__ int wsum[NG][4], base_s;
    const long long beg = (long long)blockIdx.x * rows_per_slab;
    const long long end = beg + rows_per_slab < nrows ? beg + rows_per_slab : nrows;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (long long r0 = beg; r0 < end; r0 += 256 * NG) {
        AmFlag f[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) { const long long r = r0 + 256 * u + threadIdx.x; f[u] = flag[r < end ? r : end - 1]; }
        unsigned long long m[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const long long r = r0 + 256 * u + threadIdx.x;
            m[u] = __ballot(r < end && (f[u] & (AMROW_NEW | AMROW_KEY)) != 0);
            if (lane == 0) wsum[u][wave] = (int)__popcll(m[u]);
        }
        __syncthreads();
        int off = base_s;
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const long long r = r0 + 256 * u + threadIdx.x;
            int mine = off + (int)__popcll(m[u] & ((1ull << lane) - 1ull));
            for (int k = 0; k < wave; ++k) mine += wsum[u][k];
            if ((m[u] >> lane) & 1ull) ent[beg + mine].src = (int32_t)(r - beg);
            off += wsum[u][0] + wsum[u][1] + wsum[u][2] + wsum[u][3];
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = off;
        __syncthreads();
    }
    const int n = base_s;
    if (threadIdx.x == 0) cnt[blockIdx.x] = n;
    for (int j0 = 0; j0 < n; j0 += 256 * NG) {
        int here[NG], next[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int j = j0 + 256 * u + (int)threadIdx.x, jc = j < n ? j : n - 1;
            here[u] = ent[beg + jc].src;
            next[u] = jc + 1 < n ? ent[beg + jc + 1].src : (int)(end - beg);
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int j = j0 + 256 * u + (int)threadIdx.x;
            if (j < n) ent[beg + j].wgt = det_sqrt((double)((long long)next[u] - (long long)here[u]));
        }
    }
}

// AM row flags -> every row: rows that were not stored (rejected steps) are copied forward from the row before them, in place,
// for the readers that want every row (DE history, chain files, the ESS window, tests): iterations it_lo .. it_hi of walkers
// w0 .. w0 + nw - 1.  Both iterations lie in the current covariance period [base, base + cu], base = the last multiple of cov_update
// below it_hi (ring row 0): once the ring wraps, the stored row an older repeat hangs on is overwritten (the statistics read a period
// when it is complete; nothing reads further back).  One block per walker, a thread per parameter, rows in time order.
__global__ __launch_bounds__(128) void am_expand_kernel(double *AM, const AmFlag *flag, int d, int cu, int w0, long long it_lo, long long it_hi,
                                                        long long base)
{
    const int w = w0 + (int)blockIdx.x;
    const AmFlag *fw = flag + (size_t)w * cu;
    double *aw = AM + (size_t)w * cu * d;
    long long it0 = it_lo;                                   // the stored row to start from (uniform)
    while (it0 > base && !(fw[it0 % cu] & (AMROW_NEW | AMROW_KEY))) --it0;
    for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) {       // element i of the buffer's row format, whatever it is
        double x = aw[(size_t)(it0 % cu) * d + i];
        for (long long it = it0 + 1; it <= it_hi; ++it) {
            const int ring = (int)(it % cu);
            if (fw[ring] & (AMROW_NEW | AMROW_KEY)) x = aw[(size_t)ring * d + i];
            else if (it >= it_lo) aw[(size_t)ring * d + i] = x;
        }
    }
}
// sum of the slabs' partials, slabs ascending (plain sums): Tsum[i][j], j in [i, d]; column d holds t
__global__ void pool_reduce_kernel(const double *part, int nslab, int d, double *Tsum)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ld = d + 1;
    if (idx >= (long long)d * ld) return;
    const int i = (int)(idx / ld), j = (int)(idx % ld);
    if (j < i) return;
    // 32 slabs' values requested at once (the sums stay in slab order): ten thousand threads walk 512 partials each, the kernel is
    // the latency of its loads
    const size_t st = (size_t)d * ld;
    double sum = 0.0;
    int s = 0;
    for (; s + 32 <= nslab; s += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = part[(size_t)(s + u) * st + idx];
#pragma unroll
        for (int u = 0; u < 32; ++u) sum += v[u];
    }
    for (; s + 8 <= nslab; s += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(s + u) * st + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; s < nslab; ++s) sum += part[(size_t)s * st + idx];
    Tsum[idx] = sum;
}
// Chan's combination of the chunk (nb samples, sums about the shift) with the running pooled statistics (nprev samples)
__global__ void pool_finish_kernel(const double *Tsum, const double *shift, int shift_epl, double *mu, double *M2, double *cov, int d, int first,
                                   double nb, double f /* nprev nb / (nprev + nb) */, double gw /* nb / (nprev + nb) */, double den)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)d * d) return;
    const int i = (int)(idx / d), j = (int)(idx % d), ld = d + 1;
    if (j < i) return;
    const double ti = Tsum[(size_t)i * ld + d], tj = Tsum[(size_t)j * ld + d];
    const double M2b = Tsum[(size_t)i * ld + j] - (ti * tj) / nb;
    double m;
    if (first) m = M2b;
    else m = (M2[idx] + M2b) + ((ti / nb) * (tj / nb)) * f;
    M2[idx] = m;
    M2[(size_t)j * d + i] = m;
    const double cv = m / den;
    cov[idx] = cv;
    cov[(size_t)j * d + i] = cv;
    if (i == j) mu[i] = first ? shift[am_pos(i, shift_epl)] + ti / nb : mu[i] + (ti / nb) * gw;
}

__global__ void welford_mean_kernel(const double *AM, double *mu, int d, int mem, long long iter, int fused, int am_epl)
{
    const int w = (int)blockIdx.y;
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= d) return;
    const double *am = AM + (size_t)w * mem * d;
    long long it = iter - mem;
    double m = it == 0 ? 0.0 : mu[(size_t)w * d + j];
    for (int ii = 0; ii < mem; ++ii) {
        it += 1;
        const double df = am[(size_t)ii * d + am_pos(j, am_epl)] - m;
        if (fused) m = m + df * (1.0 / (double)it);
        else m += df / (double)it;
    }
    mu[(size_t)w * d + j] = m;
}

// ------------------------------------------------- AM increments ahead of the launch (large ndim)
// The AM proposal q = x + U (cd sqrt(S) z) (PT:879-933) costs 2 ndim^2 flops per pick, and its increment depends on the chain's
// stream, the iteration and the scale branch only -- not on its state.  For the 16- and 64-lane shapes (ndim > 104), where a
// chain is a quarter of a wave or a whole one and the kernel's own product ran on the vector pipe (1.2 s per 100 steps of the
// default mix at ndim = 1000), the increments of a piece of the launch are computed AHEAD of it as one matrix product on the
// matrix cores: the piece's AM picks are listed (am_count / am_scan / am_fill: one thread per chain repeats the kernel's cycle
// draw), am_gemm_kernel computes INC[event][:] = sum_k Ut[k][:] w_event[k] -- 64 events per block, every weight generated in the
// block from the event's stream, v_mfma_f64_16x16x4 accumulating k ascending: the oracle's fma chain -- and the step kernel
// reads its increment instead of computing it.
struct AmEvent { long long it; double cd; u32 sid, pad /* the pick's parameter group */; };
struct AmArgs {
    u64 seed;
    long long iter0, nch;
    int nsteps, nt, ntg, temp0, walker0, w_host, w_scam, w_am, w_de, pick_walker;
    int w_gj;                          // NUTS + HMC entries behind the others in the cycle (propose() with GJ)
    int ngroups;                       // parameter groups (PT:129-145): > 1: the pick's group is drawn as propose() draws it
    int per_walker;                    // per-walker covariances: an event's table is its walker's (key = walker * ngroups + group), else key = group
    const double *gcn;                 // [ngroups] 2.4 / sqrt(2 size of the group) (PT:928)
    const int32_t *temp_of;
    const double *temps_mh;
};
// the cycle draw of propose() (ptmi_mh.inc.h) for one (chain, iteration); cd = 2.4 / sqrt(2 ndim) * scale (PT:846-862, 928)
__device__ __forceinline__ bool am_pick(const AmArgs &p, long long ch, long long it, AmEvent &e)
{
    const int w = (int)(ch / p.nt), t = p.temp_of[ch];
    const u32 sid0 = (u32)((u64)(p.walker0 + w) * (u32)p.ntg), sid = sid0 + (u32)(p.temp0 + t);
    u64 p0, p1;
    philox_words(p.seed, (u64)it, sid, 0u, p0, p1);
    u32 pickw = (u32)(p0 >> 32);
    if (p.pick_walker) {
        u64 q0, q1;
        philox_words(p.seed, (u64)it, sid0, 0u, q0, q1);
        pickw = (u32)(q0 >> 32);
    }
    const int L = p.w_host + p.w_scam + p.w_am + p.w_de + p.w_gj;
    const int ind = (int)__umulhi(pickw, (u32)L) - p.w_host;
    if (ind < p.w_scam || ind >= p.w_scam + p.w_am) return false;
    constexpr u32 T97 = (u32)(0.97 * 4294967296.0), T90 = (u32)(0.9 * 4294967296.0);
    const u32 plo = (u32)p0;
    const double temp = p.temps_mh[t];
    const bool warm = temp <= 100.0;
    const double sT = warm ? det_sqrt(temp) : 1.0;
    const double base = plo > T97 ? 10.0 : (plo > T90 ? 0.2 : 1.0);
    u32 g = 0;
    if (p.ngroups > 1) {               // propose()'s group draw (PT:897): its own Philox call
        u64 g0, g1;
        philox_words(p.seed, (u64)it, sid, 2u, g0, g1);
        g = __umulhi((u32)(g0 >> 32), (u32)p.ngroups);
    }
    e.it = it; e.sid = sid; e.pad = g;
    e.cd = p.gcn[g] * (warm ? base * sT : base);
    return true;
}
// (gtot: with parameter groups the picks per group, [ngroups] zeroed by the caller: the block's histogram first)
constexpr int AM_MAXG = 1024;         // = the ABI's limit on ngroups
__global__ __launch_bounds__(256) void am_count_kernel(const AmArgs p, int32_t *count, int32_t *gtot)
{
    __shared__ int32_t lh[AM_MAXG];
    const bool hist = gtot && !p.per_walker;                             // (per walker: thousands of keys, a handful of picks each: straight to memory)
    if (hist)
        for (int g = (int)threadIdx.x; g < p.ngroups; g += 256) lh[g] = 0;
    if (hist) __syncthreads();
    const long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < p.nch) {
        int n = 0;
        AmEvent e;
        const long long kw = p.per_walker ? (ch / p.nt) * p.ngroups : 0;
        for (int s = 0; s < p.nsteps; ++s)
            if (am_pick(p, ch, p.iter0 + s, e)) {
                n += 1;
                if (hist) atomicAdd(&lh[e.pad], 1);
                else if (gtot) atomicAdd(&gtot[kw + e.pad], 1);
            }
        count[ch] = n;
    }
    if (hist) {
        __syncthreads();
        for (int g = (int)threadIdx.x; g < p.ngroups; g += 256)
            if (lh[g]) atomicAdd(&gtot[g], lh[g]);
    }
}
// exclusive prefix sums of the chains' counts, base[nch] = the number of events; two launches of 1024-chain blocks: the blocks' sums, then
// every block adds up the sums before it and scans its own counts (one block over all chains took 0.44 ms at 262 144 chains: its
// threads walked the counts 256 apart)
__device__ __forceinline__ int am_block_scan(int v, int &total, int *wsum /* [16] */)
{
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < 16; ++w) { const int t = wsum[w]; all += t; if (w < wave) before += t; }
    total = all;
    return before + inc - v;                                             // exclusive
}
__global__ __launch_bounds__(1024) void am_scan_sums_kernel(const int32_t *count, int32_t *part, long long nch)
{
    __shared__ int wsum[16];
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    int total;
    (void)am_block_scan(i < nch ? count[i] : 0, total, wsum);
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void am_scan_kernel(const int32_t *count, const int32_t *part, long long *base, long long nch)
{
    __shared__ int wsum[16];
    __shared__ long long off_s;
    __shared__ long long red[16];
    long long off = 0;
    for (int j = (int)threadIdx.x; j < (int)blockIdx.x; j += 1024) off += part[j];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) off += __shfl_down(off, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = off;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int w = 0; w < 16; ++w) t += red[w]; off_s = t; }
    __syncthreads();
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    int total;
    const int ex = am_block_scan(i < nch ? count[i] : 0, total, wsum);
    if (i < nch) base[i] = off_s + ex;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) base[nch] = off_s + total;
}
// The events in chain order (the step kernel walks its chain's increments in step order); with parameter groups also perm: the
// events' indices listed group by group (group g: perm[gbase[g] ...), in no particular order inside a group -- an event's increment
// does not depend on its neighbours), so that a block of am_gemm_kernel holds 64 events of ONE group and multiplies by that group's
// rows only.  A block reserves its share of every group's list with one atomic per group.
__global__ __launch_bounds__(256) void am_fill_kernel(const AmArgs p, const long long *base, AmEvent *ev, const long long *gbase, int32_t *cursor, int32_t *perm)
{
    __shared__ int32_t lh[AM_MAXG];
    __shared__ long long lb[AM_MAXG];
    const long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    AmEvent e;
    const bool hist = perm && !p.per_walker;
    if (hist) {
        for (int g = (int)threadIdx.x; g < p.ngroups; g += 256) lh[g] = 0;
        __syncthreads();
        if (ch < p.nch)
            for (int s = 0; s < p.nsteps; ++s)
                if (am_pick(p, ch, p.iter0 + s, e)) atomicAdd(&lh[e.pad], 1);
        __syncthreads();
        for (int g = (int)threadIdx.x; g < p.ngroups; g += 256) {
            lb[g] = lh[g] ? gbase[g] + atomicAdd(&cursor[g], lh[g]) : 0;
            lh[g] = 0;
        }
        __syncthreads();
    }
    if (ch >= p.nch) return;
    long long at = base[ch];
    const long long kw = p.per_walker ? (ch / p.nt) * p.ngroups : 0;
    for (int s = 0; s < p.nsteps; ++s)
        if (am_pick(p, ch, p.iter0 + s, e)) {
            if (hist) perm[lb[e.pad] + atomicAdd(&lh[e.pad], 1)] = (int32_t)at;
            else if (perm) perm[gbase[kw + e.pad] + atomicAdd(&cursor[kw + e.pad], 1)] = (int32_t)at;      // per walker: the key's own cursor
            ev[at++] = e;
        }
}
// 64 events per block of four waves; wave v holds the output tiles v, v + 4, ... (16 rows of the increment each) of all four event
// tiles: at most 8 x 4 tiles of 8 registers = the 256 accumulation registers, i.e. 512 rows of the increment per block.  Beyond
// (ndim <= 1024) two blocks (blockIdx.y) share an event tile, each with half of the output rows and its own copy of the weights.
// The weights of 2 G consecutive directions -- G Box-Muller pairs (k, k + G) per event, the pairing of the step kernels -- are
// generated into LDS one super-chunk ahead of the products that use them.
// Parameter groups (PT:129-145, 897): one launch per group with the group's table (its eigenvectors embedded in the full space, rows
// k < nk = the group's size) over the group's list of events (am_fill_kernel's perm): an event's increment is the k-ascending fma
// chain over ITS group's rows, as the step kernel's own product, and lands in the event's row of inc.
template <int G, int MAXT>
__global__ __launch_bounds__(256, 1) void am_gemm_kernel(const AmEvent *ev, const long long *base, long long nch, int d, const double *Ut,
                                                        const double *S, u64 seed, double *inc, int nk, const long long *kbase /* the lists' starts (+ the end) */,
                                                        const int32_t *perm, int grp, int ngroups, int z0 /* first walker of this launch's grid rows */)
{
    constexpr int NEV = 64, K2 = 2 * G;
    extern __shared__ __attribute__((aligned(16))) double Wl[];          // [2][K2][NEV]
    __shared__ int32_t evi[NEV];                                         // parameter groups: the events' indices (their rows of inc)
    // one group (perm == nullptr): the events e0 .. of the chain-ordered list; else entries e0 .. of the group's list
    // the list's key: the group, or (blockIdx.z = the walker, group) with per-walker covariances -- and the key's table
    const long long key = perm ? ((long long)blockIdx.z + z0) * ngroups + grp : 0;
    const long long seg0 = perm ? kbase[key] : 0;
    const long long nev = perm ? kbase[key + 1] - seg0 : base[nch], e0 = (long long)blockIdx.x * NEV;
    if (e0 >= nev) return;
    Ut += (size_t)key * d * d;
    S += (size_t)key * d;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const bool ev_on = e0 + lane < nev;
    const long long mine = ev_on ? e0 + lane : e0;
    const long long ei = perm ? (long long)perm[seg0 + mine] : mine;
    const AmEvent me = ev[ei];
    if (perm && wave == 0) evi[lane] = (int32_t)ei;
    ps_d4 acc[MAXT][4];
#pragma unroll
    for (int tt = 0; tt < MAXT; ++tt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[tt][ct] = ps_d4{0.0, 0.0, 0.0, 0.0};
    const int ntile = (d + 15) / 16, nsup = (nk + K2 - 1) / K2;
    const int tile0 = (int)blockIdx.y * 4 * MAXT + wave;               // this wave's first output tile
    auto gen = [&](int m, int buf) {
        for (int kk = wave; kk < G; kk += 4) {                          // pair (k, k + G) of event `lane`
            const int k = m * K2 + kk;
            double wa = 0.0, wb = 0.0;
            if (ev_on && k < nk) {
                u64 w0, w1;
                philox_words(seed, (u64)me.it, me.sid, SLOT_AM + (u32)k, w0, w1);
                const double r = det_sqrt(-2.0 * unit_log<0>(w0));
                u32 aj;
                double at, sn, cs;
                unit_angle64(w1, aj, at);
                unit_sincos<0>(aj, at, sn, cs);
                wa = (r * cs) * me.cd * det_sqrt(S[k]);                   // PT:930
                if (k + G < nk) wb = (r * sn) * me.cd * det_sqrt(S[k + G]);
            }
            Wl[((size_t)buf * K2 + kk) * NEV + lane] = wa;
            Wl[((size_t)buf * K2 + kk + G) * NEV + lane] = wb;
        }
    };
    gen(0, 0);
    __syncthreads();
    int buf = 0;
    for (int m = 0; m < nsup; ++m) {
        if (m + 1 < nsup) gen(m + 1, buf ^ 1);
        const double *Wb = Wl + (size_t)buf * K2 * NEV + (size_t)g * NEV + c;
        // The table values go four output tiles at a time: the next group's (the next k-step's first group after the last) are
        // requested before the 16 matrix instructions of this one -- one wave per SIMD, nothing else hides the round trip, and
        // the 256 accumulation registers leave no room to hold a whole k-step ahead.
        auto rows_of = [&](int k0, int grp, double (&dst)[4]) {
            const int kr = k0 + g < nk ? k0 + g : nk - 1;                // rows past the end: their weights are zero
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = 16 * (tile0 + 4 * (4 * grp + u)) + c;
                dst[u] = Ut[(size_t)kr * d + (col < d ? col : d - 1)];
            }
        };
        double ga[4], gn[4];
        rows_of(m * K2, 0, ga);
#pragma unroll 1
        for (int ks = 0; ks < K2 / 4; ++ks) {
            const int k0 = m * K2 + 4 * ks;
            if (k0 >= nk) break;                                         // uniform
            double bq[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) bq[ct] = Wb[(size_t)(4 * ks) * NEV + 16 * ct];
#pragma unroll
            for (int grp = 0; grp < MAXT / 4; ++grp) {
                if (grp + 1 < MAXT / 4) rows_of(k0, grp + 1, gn);
                else rows_of(k0 + 4 < nk ? k0 + 4 : k0, 0, gn);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (tile0 + 4 * (4 * grp + u) < ntile) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[4 * grp + u][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[u], bq[ct], acc[4 * grp + u][ct], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u) ga[u] = gn[u];
            }
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int tt = 0; tt < MAXT; ++tt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (tile0 + 4 * tt) + g + 4 * r;
                const long long e = e0 + 16 * ct + c;
                if (i < d && e < nev) inc[(size_t)(perm ? (long long)evi[16 * ct + c] : e) * d + i] = acc[tt][ct][r];
            }
}

// DE history ring: rows [head, head+mem) are the oldest; overwrite them with the AM buffer.
// Row layout (ptmi_de_row_stride): with 4 lanes per chain a row is stored in 16-byte PIECES dealt to the lanes in turn --
// position 8 (e / 2) + 2 lane + e % 2 holds element lane + 4 e, zero where that is past ndim -- so that one read
// instruction of a DE proposal takes 64 contiguous bytes per chain (16 cache lines per wave instruction, each used
// again by the next instruction).  Round-2 history: in element order the four lanes pulled 8 B each out of 26 x 32-B
// pieces (140 us per step of 262 144 chains); lane-major rows (a lane's 26 values contiguous, 208 B) made every
// instruction touch 64 different lines and a piece straddle 2-3 of them (85 us).
__global__ void de_update_kernel(double *DE, const double *AM, int d, int de_size, int mem, int head, int W, int pooled, int ld, int epl, int am_epl)
{
    const int r = (int)blockIdx.x;   // new row index 0..mem-1 (or the tail when mem > de_size)
    const int wc = (int)blockIdx.y;
    const int skip = mem > de_size ? mem - de_size : 0;
    if (r < skip) return;
    const int phys = (head + (r - skip)) % de_size;
    const int src_w = pooled ? r % W : wc;
    const double *src = AM + ((size_t)src_w * mem + r) * d;
    double *dst = DE + ((size_t)wc * de_size + phys) * ld;
    for (int j = (int)threadIdx.x; j < ld; j += (int)blockDim.x) {
        int i = j;                                                    // element stored at position j
        if (epl) {
            const int e = 2 * (j / 8) + (j & 1);
            i = e < epl ? ((j & 7) >> 1) + 4 * e : d;
        }
        dst[j] = i < d ? src[am_pos(i, am_epl)] : 0.0;
    }
}


// ---------------------------------------------------------- eigensolver
// Batched symmetric eigensolver for the per-walker covariances (PT:797-803 calls LAPACK's SVD once per epoch; a batch of
// thousands of walkers would queue thousands of host factorizations).  One block per matrix, one-sided (Hestenes) Jacobi
// on the rows of W = V^T A with W and V^T both in LDS: in every round of the circle-method schedule the n/2 disjoint row
// pairs are rotated at once, eight lanes per pair (lane l owns elements l, l+8, ... of both rows, cached in registers
// for the three dot products and the rotation); one barrier per round.  Operation order = oracle/ptmcmc_oracle.c orc_eig_jacobi,
// so the results are bit-identical to it.  Eigenvalues descending, eigenvectors as rows, largest component positive.
constexpr int JAC_THREADS = 512;                            // 8 lanes per row pair, up to 64 pairs (ndim <= 101 uses 51)
constexpr int JAC_L = 8;
constexpr int JAC_MAX_SWEEPS = 30;
// sum over the eight lanes of a pair: xor 4, xor 2, xor 1 (the oracle's ((s0+s4)+(s2+s6)) + ((s1+s5)+(s3+s7)))
__device__ __forceinline__ double jac_oct_sum(double p)
{
    p = p + __shfl_xor(p, 4, 64);
    p = p + dppf64<0x4E>(p);     // xor 2
    p = p + dppf64<0xB1>(p);     // xor 1
    return p;
}
__global__ __launch_bounds__(JAC_THREADS) void eig_jacobi_kernel(const double *cov, double *Ut, double *S, int d, int ut_stride, int s_stride)
{
    extern __shared__ __attribute__((aligned(16))) double jsm[];     // W[d][d], V[d][d]: all of the CU's LDS at d = 101
    double *W = jsm, *V = jsm + (size_t)d * d;
    constexpr int NE = 13;                                           // elements of a row per lane: l, l + 8, ... < 104
    const int tid = (int)threadIdx.x;
    const double *A = cov + (size_t)blockIdx.x * d * d;
    for (int i = tid; i < d * d; i += JAC_THREADS) {
        W[i] = A[i];
        V[i] = (i / d == i % d) ? 1.0 : 0.0;
    }
    const int n = d + (d & 1), P = n / 2, rounds = n - 1;
    const int pr = tid / JAC_L, l = tid % JAC_L;
    __syncthreads();
    for (int sweep = 0; sweep < JAC_MAX_SWEEPS; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < rounds; ++r) {
            if (pr < P) {                                            // P <= 51 pairs: eight lanes each
                const int k = pr;
                const int a = k == 0 ? n - 1 : (r + k) % (n - 1);
                const int b = k == 0 ? r : (r - k + (n - 1)) % (n - 1);
                const int p = a < b ? a : b, q = a < b ? b : a;
                const bool real = q < d;                             // the bye of an odd dimension
                double *wp = W + (size_t)p * d, *wq = W + (size_t)(real ? q : p) * d;
                // both rows into registers once (zeros beyond the row: fma(0, 0, s) = s leaves the sums untouched)
                double xp[NE], xq[NE];
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int i = l + JAC_L * j;
                    xp[j] = i < d ? wp[i] : 0.0;
                    xq[j] = i < d ? wq[i] : 0.0;
                }
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    al = __builtin_fma(xp[j], xp[j], al);
                    be = __builtin_fma(xq[j], xq[j], be);
                    ga = __builtin_fma(xp[j], xq[j], ga);
                }
                al = jac_oct_sum(al); be = jac_oct_sum(be); ga = jac_oct_sum(ga);
                if (real && __builtin_fabs(ga) > 0x1.0p-50 * det_sqrt(al * be)) {      // uniform over the pair's lanes
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (__builtin_fabs(zeta) + det_sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / det_sqrt(1.0 + t * t), sn = c * t;
                    double *vp = V + (size_t)p * d, *vq = V + (size_t)q * d;
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const int i = l + JAC_L * j;
                        if (i < d) {
                            const double u = vp[i], v = vq[i];
                            wp[i] = c * xp[j] - sn * xq[j];
                            wq[i] = sn * xp[j] + c * xq[j];
                            vp[i] = c * u - sn * v;
                            vq[i] = sn * u + c * v;
                        }
                    }
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
    }
    // norms (eight lanes per row, same summation as above): first in registers, then -- W is dead -- in W[0..d)
    double mynorm[2] = {0.0, 0.0};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int k = pass * (JAC_THREADS / JAC_L) + pr;
        double al = 0.0;
        if (k < d)
            for (int j = 0; j < NE; ++j) { const int i = l + JAC_L * j; const double x = i < d ? W[(size_t)k * d + i] : 0.0; al = __builtin_fma(x, x, al); }
        mynorm[pass] = det_sqrt(jac_oct_sum(al));
    }
    __syncthreads();                                                 // every row of W has been read: W[0..d) now holds the norms
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int k = pass * (JAC_THREADS / JAC_L) + pr;
        if (k < d && l == 0) W[k] = mynorm[pass];
    }
    __syncthreads();
    const double *nrm = W;
    double *Uo = Ut + (size_t)blockIdx.x * ut_stride, *So = S + (size_t)blockIdx.x * s_stride;
    for (int k = pr; k < d; k += JAC_THREADS / JAC_L) {
        const double mine = nrm[k];
        int rank = 0;
        for (int j = 0; j < d; ++j) rank += (nrm[j] > mine) || (nrm[j] == mine && j < k);
        const double *vk = V + (size_t)k * d;
        int im = 0;
        for (int i = 1; i < d; ++i) if (__builtin_fabs(vk[i]) > __builtin_fabs(vk[im])) im = i;
        const double sg = vk[im] < 0.0 ? -1.0 : 1.0;
        for (int i = l; i < d; i += JAC_L) Uo[(size_t)rank * d + i] = sg * vk[i];
        if (l == 0) So[rank] = mine;
    }
}

// sqrt(x) and 1 / sqrt(x)'s partner 1 / r of a rotation, for x in the middle of the exponent range: the compiler's own correctly
// rounded sequences (v_rsq_f64 / v_rcp_f64 + the fma refinements of its sqrt and division lowerings) without their range scaling,
// special-value tests and fix-ups -- 17 instead of 29 instructions on the chain that bounds eig_ql_chain_kernel, the same bits
// wherever no scaling would have been applied; anything else takes the plain operations.
__device__ __forceinline__ void ql_root_and_reciprocal(double x, double &r, double &ri)
{
    if (x > 0x1p-600 && x < 0x1p600) {                              // uniform in the chain kernel
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, hh = 0.5 * y;
        const double r0 = __builtin_fma(-hh, g, 0.5);
        g = __builtin_fma(g, r0, g);
        hh = __builtin_fma(hh, r0, hh);
        double dd = __builtin_fma(-g, g, x);
        g = __builtin_fma(dd, hh, g);
        dd = __builtin_fma(-g, g, x);
        r = __builtin_fma(dd, hh, g);
        double q = __builtin_amdgcn_rcp(r);
        double e = __builtin_fma(-r, q, 1.0);
        q = __builtin_fma(q, e, q);
        e = __builtin_fma(-r, q, 1.0);
        q = __builtin_fma(q, e, q);
        e = __builtin_fma(-r, q, 1.0);
        ri = __builtin_fma(e, q, q);
    } else {
        r = det_sqrt(x);
        ri = 1.0 / r;
    }
}
// ----------------------------------------------------------- tridiagonal QL eigensolver (eig_mode "ql")
// The eigendecomposition of PT:797-803 by Householder tridiagonalization with the transformations accumulated, then implicit QL
// iterations on the tridiagonal matrix (oracle: orc_eig_ql -- the kernel does the oracle's operations in the oracle's order, dot products
// as eight interleaved fma chains, so both give the same bits).  eig_jacobi_kernel needs nine sweeps of n^2 / 2
// rotations, each moving two rows of W and two of V through LDS (4.4 ms per 100 x 100 matrix, one matrix per CU), on the nearly
// degenerate spectra an isotropic target adapts to; here the O(n^3) work is two passes over the matrix and the rest is a chain of
// some 7500 plane rotations whose scalars depend on each other (one sqrt and one division each) while the columns they turn do not.
// One block of two waves per matrix, two blocks per CU at ndim = 100 (the matrix, the subdiagonal and one work row: 81.6 KB):
//  * reduction, row i = n-1 .. 1: every thread forms the row's scalars itself (broadcast reads: no barrier for them); thread j owns
//    row j of the products p = A u / h and of the rank-two update;
//  * accumulation, row i = 0 .. n-1: thread j owns column j of the leading block (its product and its update need nothing else);
//  * QL: ONE wave (64 lanes, rows k and k + 64 of the eigenvector matrix each) runs the scalar recurrence in every lane and turns
//    its rows; nothing is synchronised inside this phase.
constexpr int QL_THREADS = 128;
constexpr int QL_MAXIT = 60;
// the oracle's QL_DOT8: eight interleaved fma chains, term k into chain k mod 8 (a dependent f64 operation costs a lone wave some 20
// cycles: one chain of 100 terms is 2000 cycles, eight side by side 300)
template <class FA, class FB>
__device__ __forceinline__ double ql_dot8(int cnt, FA fa, FB fb)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    int k = 0;
    for (; k + 8 <= cnt; k += 8) {
        s0 = __builtin_fma(fa(k), fb(k), s0);
        s1 = __builtin_fma(fa(k + 1), fb(k + 1), s1);
        s2 = __builtin_fma(fa(k + 2), fb(k + 2), s2);
        s3 = __builtin_fma(fa(k + 3), fb(k + 3), s3);
        s4 = __builtin_fma(fa(k + 4), fb(k + 4), s4);
        s5 = __builtin_fma(fa(k + 5), fb(k + 5), s5);
        s6 = __builtin_fma(fa(k + 6), fb(k + 6), s6);
        s7 = __builtin_fma(fa(k + 7), fb(k + 7), s7);
    }
    if (k < cnt) s0 = __builtin_fma(fa(k), fb(k), s0);
    if (k + 1 < cnt) s1 = __builtin_fma(fa(k + 1), fb(k + 1), s1);
    if (k + 2 < cnt) s2 = __builtin_fma(fa(k + 2), fb(k + 2), s2);
    if (k + 3 < cnt) s3 = __builtin_fma(fa(k + 3), fb(k + 3), s3);
    if (k + 4 < cnt) s4 = __builtin_fma(fa(k + 4), fb(k + 4), s4);
    if (k + 5 < cnt) s5 = __builtin_fma(fa(k + 5), fb(k + 5), s5);
    if (k + 6 < cnt) s6 = __builtin_fma(fa(k + 6), fb(k + 6), s6);
    return ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7));
}
// (Measured and dropped: the matrix in a global scratch with 3 n doubles of LDS per block, sixteen blocks per CU and all 4096 matrices
// resident at once -- every broadcast read became an L2 round trip: 84 ms per epoch against 41.)
__global__ __launch_bounds__(QL_THREADS) void eig_ql_kernel(const double *cov, double *Ut, double *S, int n, int ut_stride, int s_stride, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) double qsm[];
    double *z = qsm, *e = qsm + (((size_t)n * n + 1) & ~(size_t)1), *pq = e + n;     // pq: the products p / h, then q; after the accumulation: the diagonal d
    const int t = (int)threadIdx.x;
    const double *A = cov + (size_t)blockIdx.x * n * n;
#define QZ(i, j) z[(i) * n + (j)]
    for (int i = t; i < n * n; i += QL_THREADS) z[i] = A[i];
#ifdef PTMI_QL_PROFILE
    unsigned long long qt0 = __builtin_readcyclecounter(), qt1, qt2, qt3;
#endif
    unsigned long long hmask[2] = {0ull, 0ull};                   // rows whose reflector exists (the oracle's d[i] != 0), n <= 128
    __syncthreads();
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0;
        if (l > 0) h = ql_dot8(l + 1, [&](int k) { return QZ(i, k); }, [&](int k) { return QZ(i, k); });
        if (l == 0 || h == 0.0) {                                 // uniform
            if (t == 0) e[i] = QZ(i, l);
            __syncthreads();
            continue;
        }
        const double f0 = QZ(i, l);
        const double g0 = f0 >= 0.0 ? -det_sqrt(h) : det_sqrt(h);
        h = h - f0 * g0;
        __syncthreads();                                          // every thread has read Z(i, l)
        if (t == 0) { e[i] = g0; QZ(i, l) = f0 - g0; }
        __syncthreads();
        for (int j = t; j <= l; j += QL_THREADS) {
            QZ(j, i) = QZ(i, j) / h;
            const double g = ql_dot8(l + 1, [&](int k) { return k <= j ? QZ(j, k) : QZ(k, j); }, [&](int k) { return QZ(i, k); });
            pq[j] = g / h;
        }
        __syncthreads();
        const double f = ql_dot8(l + 1, [&](int k) { return pq[k]; }, [&](int k) { return QZ(i, k); });
        const double hh = f / (h + h);
        __syncthreads();                                          // every thread has its f
        for (int j = t; j <= l; j += QL_THREADS) pq[j] = pq[j] - hh * QZ(i, j);
        __syncthreads();
        for (int j = t; j <= l; j += QL_THREADS) {
            const double uj = QZ(i, j), qj = pq[j];
            for (int k = 0; k <= j; ++k) QZ(j, k) = QZ(j, k) - (uj * pq[k] + qj * QZ(i, k));
        }
        hmask[i >> 6] |= 1ull << (i & 63);
        __syncthreads();
    }
    if (t == 0) e[0] = 0.0;
#ifdef PTMI_QL_PROFILE
    qt1 = __builtin_readcyclecounter();
#endif
    // accumulation of the transformations
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if ((hmask[i >> 6] >> (i & 63)) & 1ull) {
            for (int j = t; j <= l; j += QL_THREADS) {
                const double g = ql_dot8(l + 1, [&](int k) { return QZ(i, k); }, [&](int k) { return QZ(k, j); });
                for (int k = 0; k <= l; ++k) QZ(k, j) = QZ(k, j) - g * QZ(k, i);
            }
        }
        __syncthreads();
        if (t == 0) { pq[i] = QZ(i, i); QZ(i, i) = 1.0; }
        for (int j = t; j <= l; j += QL_THREADS) { QZ(j, i) = 0.0; QZ(i, j) = 0.0; }
        __syncthreads();
    }
    // ---- implicit QL: one wave, no barrier; lane `t` turns rows t and t + 64.  The diagonal and the subdiagonal are re-laid as
    // pairs {d[i], e[i]} over the 2 n doubles of e and pq (one 16-byte read and one 16-byte write per rotation); of the two
    // columns a rotation turns, the lower one is the next rotation's upper one and stays in a register.
    typedef double ql_d2 __attribute__((ext_vector_type(2)));
    ql_d2 *de = reinterpret_cast<ql_d2 *>(e);
    int iters = 0, failed = 0;
#ifdef PTMI_QL_PROFILE
    qt2 = __builtin_readcyclecounter();
#endif
    if (t < 64) {
        const int k0 = t, k1 = t + 64;
        const bool r0 = k0 < n, r1 = k1 < n;
        {
            // e[i - 1] = e[i], e[n - 1] = 0, then the pairs: every lane reads its entries before any lane writes
            const int ia = t, ib = t + 64;
            const double da = ia < n ? pq[ia] : 0.0, db = ib < n ? pq[ib] : 0.0;
            const double ea = ia + 1 < n ? e[ia + 1] : 0.0, eb = ib + 1 < n ? e[ib + 1] : 0.0;
            asm volatile("" ::: "memory");
            if (ia < n) de[ia] = ql_d2{da, ea};
            if (ib < n) de[ib] = ql_d2{db, eb};
            asm volatile("" ::: "memory");
        }
#define QD(i) de[i].x
#define QE(i) de[i].y
        double f = 0.0, tst1 = 0.0;
        for (int l = 0; l < n && !failed; ++l) {
            const ql_d2 del = de[l];
            const double t0 = __builtin_fabs(del.x) + __builtin_fabs(del.y);
            if (tst1 < t0) tst1 = t0;
            int m = l;
            while (m < n - 1 && tst1 + __builtin_fabs(QE(m)) != tst1) ++m;
            double dlf = del.x;                                  // d[l] as the iterations leave it
            if (m > l) {
                int it = 0;
                double el;
                do {
                    if (++it > QL_MAXIT) { failed = 1; break; }
                    ++iters;
                    const ql_d2 pl = de[l], pl1 = de[l + 1];
                    const double g = pl.x, e_l = pl.y;
                    const double p0 = (pl1.x - g) / (2.0 * e_l);
                    const double rr0 = det_sqrt(p0 * p0 + 1.0);
                    const double pr = p0 + (p0 >= 0.0 ? rr0 : -rr0);
                    const double dl = e_l / pr, dl1 = e_l * pr;
                    const double h = g - dl;
                    const double el1 = pl1.y;
                    double p = QD(m);
                    asm volatile("" ::: "memory");
                    if (t == 0) { QD(l) = dl; QD(l + 1) = dl1; }
                    for (int i = l + 2 + t; i < n; i += 64) QD(i) = QD(i) - h;
                    asm volatile("" ::: "memory");
                    f = f + h;
                    if (m == l + 1) p = dl1; else if (m >= l + 2) p = p - h;     // d[m] as the updates above leave it
                    double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                    ql_d2 nx = de[m - 1];                            // the next rotation's inputs are asked for a rotation ahead
                    double zb0 = r0 ? z[k0 * n + m] : 0.0, zb1 = r1 ? z[k1 * n + m] : 0.0;     // column i + 1 of the lane's rows, carried
                    for (int i = m - 1; i >= l; --i) {
                        c3 = c2; c2 = c; s2 = s;
                        const double di = nx.x, ei = nx.y;
                        if (i > l) nx = de[i - 1];
                        const double za0 = r0 ? z[k0 * n + i] : 0.0, za1 = r1 ? z[k1 * n + i] : 0.0;
                        const double gg = c * ei, hh = c * p;
                        double r, ri;
                        ql_root_and_reciprocal(p * p + ei * ei, r, ri);
                        const double e1 = s * r;
                        s = ei * ri;
                        c = p * ri;
                        p = c * di - s * gg;
                        const double d1 = hh + s * (c * gg + s * di);
                        if (t == 0) de[i + 1] = ql_d2{d1, e1};
                        if (r0) z[k0 * n + i + 1] = s * za0 + c * zb0;
                        if (r1) z[k1 * n + i + 1] = s * za1 + c * zb1;
                        zb0 = c * za0 - s * zb0;
                        zb1 = c * za1 - s * zb1;
                    }
                    if (r0) z[k0 * n + l] = zb0;
                    if (r1) z[k1 * n + l] = zb1;
                    p = -s * s2 * c3 * el1 * e_l / dl1;
                    el = s * p;
                    dlf = c * p;
                    asm volatile("" ::: "memory");
                    if (t == 0) de[l] = ql_d2{dlf, el};
                    asm volatile("" ::: "memory");
                } while (tst1 + __builtin_fabs(el) != tst1);
            }
            asm volatile("" ::: "memory");
            if (t == 0) de[l] = ql_d2{dlf + f, 0.0};
            asm volatile("" ::: "memory");
        }
        if (t == 0 && status) {
            if (failed) atomicOr(status, 1);
        }
    }
    __syncthreads();
#ifdef PTMI_QL_PROFILE
    qt3 = __builtin_readcyclecounter();
    if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 3000)) printf("ql block %d: reduce %llu accumulate %llu ql %llu cycles, %d iterations\n", (int)blockIdx.x, qt1 - qt0, qt2 - qt1, qt3 - qt2, iters);
#endif
    // order and signs as eig_jacobi_kernel / orc_eig_ql
    double *Uo = Ut + (size_t)blockIdx.x * ut_stride, *So = S + (size_t)blockIdx.x * s_stride;
    for (int k = t; k < n; k += QL_THREADS) {
        const double mine = __builtin_fabs(QD(k));
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double o = __builtin_fabs(QD(j)); rank += (o > mine) || (o == mine && j < k); }
        int im = 0;
        for (int i = 1; i < n; ++i) if (__builtin_fabs(QZ(i, k)) > __builtin_fabs(QZ(im, k))) im = i;
        const double sg = QZ(im, k) < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) Uo[(size_t)rank * n + i] = sg * QZ(i, k);
        So[rank] = mine;
    }
#undef QZ
#undef QD
#undef QE
}

// ---- the same in three kernels, for MANY matrices (per-walker covariances).  The QL phase is a chain of dependent scalar
// operations (some 45 of them per rotation, ~20 cycles each for a lone wave: 900 cycles per rotation, 7.6 of the 10.3 M cycles a
// matrix takes in eig_ql_kernel) and only two matrices fit a CU's LDS: 64 ms of chain per CU and epoch whatever is done to the rest.
// But the chain needs the tridiagonal matrix alone -- 200 doubles, not the eigenvectors: eig_ql_chain_kernel runs the chains of ALL
// matrices at once (a wave each, four per SIMD) and RECORDS the rotations (c, s) with the (l, m) of every iteration;
// eig_ql_apply_kernel then turns the eigenvector rows with them, a thread per row and no scalar work.  Same operations on the
// same values in the same order as orc_eig_ql: same bits.  A matrix whose rotations do not fit the record (3 n^2; nearly degenerate
// 100 x 100 spectra take 0.8 n^2) is flagged and redone by the apply kernel with the chain and the rows together.
typedef double qls_d2 __attribute__((ext_vector_type(2)));
struct QlScratch {
    double *z;          // [nmat][n][n]  the accumulated transformations, row-major
    qls_d2 *de;         // [nmat][n]     {d[i], e[i]} (subdiagonal shifted: e[i] couples i and i + 1)
    double *ev;         // [nmat][n]     the eigenvalues the chains end with
    qls_d2 *rot;        // [nmat][cap]   the rotations, in the order they are applied
    int32_t *hdr;       // [nmat][2 capit]  l, m of every QL iteration
    int32_t *cnt;       // [nmat][2]     iterations recorded, overflow flag
    int cap, capit;
};
// Reduction and accumulation for the three-kernel form, 256 threads: a dot product is the work of an OCT of lanes -- lane c runs chain
// c of QL_DOT8 (terms k = c, c + 8, ...), the butterfly xor 4, xor 2, xor 1 is the oracle's ((s0 + s4) + (s2 + s6)) + ((s1 + s5) +
// (s3 + s7)) in every lane -- 32 products at a time; the rank-two update and the column updates are 16 x 16 tilings of their
// elements.  (A thread per row with the eight chains side by side left the threads of short rows idle and every wave alone on its
// SIMD: 17 000 cycles per row of the reduction.)
#ifndef PTMI_QLR_THREADS
#define PTMI_QLR_THREADS 512
#endif
constexpr int QLR_THREADS = PTMI_QLR_THREADS, QLR_TY = QLR_THREADS / 16;
__global__ __launch_bounds__(QLR_THREADS) void eig_ql_reduce_kernel(const double *cov, int n, QlScratch q)
{
    extern __shared__ __attribute__((aligned(16))) double qsm[];
    double *z = qsm, *e = qsm + (((size_t)n * n + 1) & ~(size_t)1), *pq = e + n;
    const int t = (int)threadIdx.x;
    const int oct = t >> 3, c8 = t & 7, ty = t >> 4, tx = t & 15;
    const double *A = cov + (size_t)blockIdx.x * n * n;
#define QZ(i, j) z[(i) * n + (j)]
    for (int i = t; i < n * n; i += QLR_THREADS) z[i] = A[i];
    unsigned long long hmask[2] = {0ull, 0ull};
    __syncthreads();
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0;
        if (l > 0) {
            double sc = 0.0;
            for (int k = c8; k <= l; k += 8) { const double v = QZ(i, k); sc = __builtin_fma(v, v, sc); }
            h = jac_oct_sum(sc);
        }
        if (l == 0 || h == 0.0) {                                 // uniform
            if (t == 0) e[i] = QZ(i, l);
            __syncthreads();
            continue;
        }
        const double f0 = QZ(i, l);
        const double g0 = f0 >= 0.0 ? -det_sqrt(h) : det_sqrt(h);
        h = h - f0 * g0;
        __syncthreads();                                          // every thread has read Z(i, l)
        if (t == 0) { e[i] = g0; QZ(i, l) = f0 - g0; }
        __syncthreads();
        for (int j = oct; j <= l; j += QLR_THREADS / 8) {
            double sc = 0.0;
            for (int k = c8; k <= l; k += 8) sc = __builtin_fma(k <= j ? QZ(j, k) : QZ(k, j), QZ(i, k), sc);
            const double g = jac_oct_sum(sc);
            // the two quotients of row j by two lanes of its oct: one division sequence instead of two on the critical path
            const double quo = (c8 == 0 ? g : QZ(i, j)) / h;
            if (c8 == 0) pq[j] = quo;
            else if (c8 == 1) QZ(j, i) = quo;
        }
        __syncthreads();
        double fc = 0.0;
        for (int k = c8; k <= l; k += 8) fc = __builtin_fma(pq[k], QZ(i, k), fc);
        const double f = jac_oct_sum(fc);
        const double hh = f / (h + h);
        __syncthreads();                                          // every thread has its f
        for (int j = t; j <= l; j += QLR_THREADS) pq[j] = pq[j] - hh * QZ(i, j);
        __syncthreads();
        for (int j = ty; j <= l; j += QLR_TY) {
            const double uj = QZ(i, j), qj = pq[j];
            for (int k = tx; k <= j; k += 16) QZ(j, k) = QZ(j, k) - (uj * pq[k] + qj * QZ(i, k));
        }
        hmask[i >> 6] |= 1ull << (i & 63);
        __syncthreads();
    }
    if (t == 0) e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if ((hmask[i >> 6] >> (i & 63)) & 1ull) {                  // uniform
            // the products g_j of the leading block's columns with row i; row i is dead afterwards (zeroed below) and keeps them
            double gj[4] = {0.0, 0.0, 0.0, 0.0};
            int nj = 0;
            for (int j = oct; j <= l; j += QLR_THREADS / 8, ++nj) {
                double sc = 0.0;
                for (int k = c8; k <= l; k += 8) sc = __builtin_fma(QZ(i, k), QZ(k, j), sc);
                gj[nj & 3] = jac_oct_sum(sc);
            }
            __syncthreads();                                      // every product has read row i
            nj = 0;
            for (int j = oct; j <= l; j += QLR_THREADS / 8, ++nj)
                if (c8 == 0) QZ(i, j) = gj[nj & 3];
            __syncthreads();
            for (int k = ty; k <= l; k += QLR_TY) {
                const double zki = QZ(k, i);
                for (int j = tx; j <= l; j += 16) QZ(k, j) = QZ(k, j) - QZ(i, j) * zki;
            }
        }
        __syncthreads();
        if (t == 0) { pq[i] = QZ(i, i); QZ(i, i) = 1.0; }
        for (int j = t; j <= l; j += QLR_THREADS) { QZ(j, i) = 0.0; QZ(i, j) = 0.0; }
        __syncthreads();
    }
#undef QZ
    double *zo = q.z + (size_t)blockIdx.x * n * n;
    for (int i = t; i < n * n; i += QLR_THREADS) zo[i] = z[i];
    qls_d2 *deo = q.de + (size_t)blockIdx.x * n;
    for (int i = t; i < n; i += QLR_THREADS) deo[i] = qls_d2{pq[i], i + 1 < n ? e[i + 1] : 0.0};
}

// the QL iterations on {d, e} pairs in LDS (one wave; every lane runs the scalar recurrence).  ROWS: the lane also turns rows t and
// t + 64 of zt (the eigenvector matrix TRANSPOSED in LDS: column c at zt[c n ...], so that the lanes' rows sit side by side);
// else the rotations and the iterations' (l, m) are recorded.  Returns the iterations (negative: an eigenvalue did not converge).
template <bool ROWS>
__device__ __forceinline__ int ql_iterate(qls_d2 *de, int n, int t, double *zt, qls_d2 *rot, int32_t *hdr, int cap, int capit, int *overflow)
{
    const int k0 = t, k1 = t + 64;
    const bool r0 = ROWS && k0 < n, r1 = ROWS && k1 < n;
    int iters = 0, nrot = 0;
    bool over = false, failed = false;
    double f = 0.0, tst1 = 0.0;
    for (int l = 0; l < n && !failed; ++l) {
        const qls_d2 del = de[l];
        const double t0 = __builtin_fabs(del.x) + __builtin_fabs(del.y);
        if (tst1 < t0) tst1 = t0;
        int m = l;
        while (m < n - 1 && tst1 + __builtin_fabs(de[m].y) != tst1) ++m;
        double dlf = del.x;
        if (m > l) {
            int it = 0;
            double el;
            do {
                if (++it > QL_MAXIT) { failed = true; break; }
                if (!ROWS) {
                    if (iters >= capit || nrot + (m - l) > cap) over = true;
                    if (!over && t == 0) { hdr[2 * iters] = l; hdr[2 * iters + 1] = m; }
                }
                ++iters;
                const qls_d2 pl = de[l], pl1 = de[l + 1];
                const double g = pl.x, e_l = pl.y;
                const double p0 = (pl1.x - g) / (2.0 * e_l);
                const double rr0 = det_sqrt(p0 * p0 + 1.0);
                const double pr = p0 + (p0 >= 0.0 ? rr0 : -rr0);
                const double dl = e_l / pr, dl1 = e_l * pr;
                const double h = g - dl;
                const double el1 = pl1.y;
                double p = de[m].x;
                asm volatile("" ::: "memory");
                if (t == 0) { de[l].x = dl; de[l + 1].x = dl1; }
                for (int i = l + 2 + t; i < n; i += 64) de[i].x = de[i].x - h;
                asm volatile("" ::: "memory");
                f = f + h;
                if (m == l + 1) p = dl1; else if (m >= l + 2) p = p - h;
                double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                qls_d2 nx = de[m - 1];
                double zb0 = r0 ? zt[m * n + k0] : 0.0, zb1 = r1 ? zt[m * n + k1] : 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    const double di = nx.x, ei = nx.y;
                    if (i > l) nx = de[i - 1];
                    double za0 = 0.0, za1 = 0.0;
                    if (ROWS) { za0 = r0 ? zt[i * n + k0] : 0.0; za1 = r1 ? zt[i * n + k1] : 0.0; }
                    const double gg = c * ei, hh = c * p;
                    double r, ri;
                    ql_root_and_reciprocal(p * p + ei * ei, r, ri);
                    const double e1 = s * r;
                    s = ei * ri;
                    c = p * ri;
                    p = c * di - s * gg;
                    const double d1 = hh + s * (c * gg + s * di);
                    if (t == 0) de[i + 1] = qls_d2{d1, e1};
                    if (ROWS) {
                        if (r0) zt[(i + 1) * n + k0] = s * za0 + c * zb0;
                        if (r1) zt[(i + 1) * n + k1] = s * za1 + c * zb1;
                        zb0 = c * za0 - s * zb0;
                        zb1 = c * za1 - s * zb1;
                    } else if (!over && t == 0) {
                        rot[nrot + (m - 1 - i)] = qls_d2{c, s};
                    }
                }
                if (ROWS) {
                    if (r0) zt[l * n + k0] = zb0;
                    if (r1) zt[l * n + k1] = zb1;
                }
                nrot += m - l;
                p = -s * s2 * c3 * el1 * e_l / dl1;
                el = s * p;
                dlf = c * p;
                asm volatile("" ::: "memory");
                if (t == 0) de[l] = qls_d2{dlf, el};
                asm volatile("" ::: "memory");
            } while (tst1 + __builtin_fabs(el) != tst1);
        }
        asm volatile("" ::: "memory");
        if (t == 0) de[l] = qls_d2{dlf + f, 0.0};
        asm volatile("" ::: "memory");
    }
    if (overflow) *overflow = over ? 1 : 0;
    return failed ? -iters - 1 : iters;
}

__global__ __launch_bounds__(64) void eig_ql_chain_kernel(int n, QlScratch q)
{
    extern __shared__ __attribute__((aligned(16))) double qsm[];
    qls_d2 *de = reinterpret_cast<qls_d2 *>(qsm);
    const int t = (int)threadIdx.x;
    const size_t b = blockIdx.x;
    for (int i = t; i < n; i += 64) de[i] = q.de[b * n + i];
    asm volatile("" ::: "memory");
    int over = 0;
    const int iters = ql_iterate<false>(de, n, t, nullptr, q.rot + b * (size_t)q.cap, q.hdr + b * 2 * (size_t)q.capit, q.cap, q.capit, &over);
    asm volatile("" ::: "memory");
    for (int i = t; i < n; i += 64) q.ev[b * n + i] = de[i].x;
    if (t == 0) { q.cnt[2 * b] = iters < 0 ? 0 : iters; q.cnt[2 * b + 1] = (over || iters < 0) ? 1 : 0; }
}

__global__ __launch_bounds__(QL_THREADS) void eig_ql_apply_kernel(double *Ut, double *S, int n, int ut_stride, int s_stride, QlScratch q, int redo_only)
{
    if (redo_only && q.cnt[2 * blockIdx.x + 1] == 0) return;      // eig_ql_apply_reg_kernel has done this matrix

    extern __shared__ __attribute__((aligned(16))) double qsm[];
    double *zt = qsm;                                              // zt[c n + k] = Z(k, c)
    qls_d2 *de = reinterpret_cast<qls_d2 *>(qsm + (((size_t)n * n + 1) & ~(size_t)1));
    const int t = (int)threadIdx.x;
    const size_t b = blockIdx.x;
    const double *zi = q.z + b * n * n;
    for (int i = t; i < n * n; i += QL_THREADS) { const int r = i / n, c = i % n; zt[c * n + r] = zi[i]; }
    const bool redo = q.cnt[2 * b + 1] != 0;                       // the record did not hold this matrix's rotations
    if (redo) {
        for (int i = t; i < n; i += QL_THREADS) de[i] = q.de[b * n + i];
        __syncthreads();
        if (t < 64) ql_iterate<true>(de, n, t, zt, nullptr, nullptr, 0, 0, nullptr);
    } else {
        // A thread per row.  The rotations of ONE iteration (at most n - 1 of them) are staged in LDS -- the pairs' 2 n doubles,
        // free until the eigenvalues go there -- by all threads at once, the next iteration's requested before this one's are
        // applied (a read of the record per rotation sat on every row's chain with its whole memory round trip: 300 cycles per
        // rotation).  Of the two columns a rotation turns, the lower one is the next rotation's upper one and stays in a register.
        const int nit = q.cnt[2 * b];
        const int32_t *hdr = q.hdr + b * 2 * (size_t)q.capit;
        const qls_d2 *rot = q.rot + b * (size_t)q.cap;
        int r = 0;
        int l = nit > 0 ? hdr[0] : 0, m = nit > 0 ? hdr[1] : 0;
        int ln = nit > 1 ? hdr[2] : 0, mn = nit > 1 ? hdr[3] : 0;  // the (l, m) of the iteration after: known two iterations ahead
        qls_d2 mine = (nit > 0 && t < m - l) ? rot[t] : qls_d2{0.0, 0.0};
        for (int itn = 0; itn < nit; ++itn) {
            const int cntr = m - l;
            __syncthreads();                                        // the previous iteration's rotations have been applied
            if (t < cntr) de[t] = mine;
            __syncthreads();
            r += cntr;
            const int lnn = itn + 2 < nit ? hdr[2 * itn + 4] : 0, mnn = itn + 2 < nit ? hdr[2 * itn + 5] : 0;
            if (itn + 1 < nit && t < mn - ln) mine = rot[r + t];
            if (t < n) {
                double zb = zt[m * n + t];
                const double *zp = zt + (size_t)(m - 1) * n + t;    // column i of this thread's row, i descending
                int j = 0;
                for (; j + 4 <= cntr; j += 4, zp -= 4 * n) {       // four rotations a trip: their reads go out together
                    const qls_d2 c0 = de[j], c1 = de[j + 1], c2 = de[j + 2], c3 = de[j + 3];
                    const double a0 = zp[0], a1 = zp[-n], a2 = zp[-2 * n], a3 = zp[-3 * n];
                    const_cast<double *>(zp)[n] = c0.y * a0 + c0.x * zb;
                    zb = c0.x * a0 - c0.y * zb;
                    const_cast<double *>(zp)[0] = c1.y * a1 + c1.x * zb;
                    zb = c1.x * a1 - c1.y * zb;
                    const_cast<double *>(zp)[-n] = c2.y * a2 + c2.x * zb;
                    zb = c2.x * a2 - c2.y * zb;
                    const_cast<double *>(zp)[-2 * n] = c3.y * a3 + c3.x * zb;
                    zb = c3.x * a3 - c3.y * zb;
                }
                for (; j < cntr; ++j, zp -= n) {
                    const qls_d2 cs = de[j];
                    const double za = zp[0];
                    const_cast<double *>(zp)[n] = cs.y * za + cs.x * zb;
                    zb = cs.x * za - cs.y * zb;
                }
                zt[l * n + t] = zb;
            }
            l = ln; m = mn;
            ln = lnn; mn = mnn;
        }
        __syncthreads();
        for (int i = t; i < n; i += QL_THREADS) de[i] = qls_d2{q.ev[b * n + i], 0.0};
    }
    __syncthreads();
    double *Uo = Ut + b * ut_stride, *So = S + b * s_stride;
    for (int k = t; k < n; k += QL_THREADS) {
        const double mine = __builtin_fabs(de[k].x);
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double o = __builtin_fabs(de[j].x); rank += (o > mine) || (o == mine && j < k); }
        const double *col = zt + (size_t)k * n;                    // Z(i, k), i = 0 .. n - 1
        int im = 0;
        for (int i = 1; i < n; ++i) if (__builtin_fabs(col[i]) > __builtin_fabs(col[im])) im = i;
        const double sg = col[im] < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) Uo[(size_t)rank * n + i] = sg * col[i];
        So[rank] = mine;
    }
}

// The apply step with the eigenvector matrix in REGISTERS (n <= 100): thread t holds row t of Z, z[0 .. n - 1], and a rotation of
// columns (i, i + 1) is six instructions on two registers of every thread -- the record's (c, s) and the iterations' (l, m) are the
// same for all rows: uniform branches, (c, s) an LDS broadcast staged by each wave for itself; no barrier until the end, four matrices (eight waves)
// per CU instead of the two that fit with Z in LDS.  Register indices are compile-time: an iteration's sweep i = m - 1 ... l is the
// unrolled sweep 98 ... 0 entered block by block (QLA_BLK = 8 steps, 2.29 ms against 2.41 with 4; a block outside [l, m) is one uniform branch, a block inside it
// runs without tests).  eig_ql_apply_kernel: 4.15 ms per epoch at 4096 x 100 x 100 (a wave per SIMD, an LDS round trip on every
// row's chain per four rotations).  Matrices whose record overflowed are left to that kernel (redo_only).
#ifndef PTMI_QLA_BLK
#define PTMI_QLA_BLK 8
#endif
constexpr int QLA_N = 100, QLA_BLK = PTMI_QLA_BLK;
template <int LO, int HI, bool CHECK>
__device__ __forceinline__ void qla_steps(double (&z)[QLA_N], const qls_d2 *cs_of_step, int l, int m)
{
#pragma unroll
    for (int i = HI; i >= LO; --i) {
        if (!CHECK || (i >= l && i < m)) {
            const qls_d2 cs = cs_of_step[i];                        // {c, s}: an LDS broadcast at a compile-time offset
            const double za = z[i], zb = z[i + 1];
            z[i + 1] = cs.y * za + cs.x * zb;
            z[i] = cs.x * za - cs.y * zb;
        }
    }
}
template <int BLKI>
__device__ __forceinline__ void qla_sweep(double (&z)[QLA_N], const qls_d2 *cs_of_step, int l, int m)
{
    constexpr int LO = QLA_BLK * BLKI, HI = LO + QLA_BLK - 1 < QLA_N - 2 ? LO + QLA_BLK - 1 : QLA_N - 2;
    if (HI >= l && LO < m) {
        if (LO >= l && HI < m) qla_steps<LO, HI, false>(z, cs_of_step, l, m);
        else qla_steps<LO, HI, true>(z, cs_of_step, l, m);
    }
    if constexpr (BLKI > 0) qla_sweep<BLKI - 1>(z, cs_of_step, l, m);
}
__global__ __launch_bounds__(128, 2) void eig_ql_apply_reg_kernel(double *Ut, double *S, int n, int ut_stride, int s_stride,
                                                                  const double *__restrict__ zin, const double *__restrict__ evin,
                                                                  const qls_d2 *__restrict__ rot_all, const int32_t *__restrict__ hdr_all,
                                                                  const int32_t *__restrict__ cnt, int cap, int capit)
{
    __shared__ double wmax[2][QLA_N], wsv[2][QLA_N], sgs[QLA_N];
    __shared__ int rk[QLA_N];
    // an iteration's rotations, staged by each wave for itself (no barrier): slot i = the rotation of step i; the next iteration's
    // are requested before this one's sweep and stored behind it (a scalar load per block of the sweep waited 600 cycles each)
    __shared__ __attribute__((aligned(16))) qls_d2 stg[2][2][QLA_N];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t b = blockIdx.x;
    if (cnt[2 * b + 1] != 0) return;                                // the record did not hold this matrix's rotations
    const bool rowok = t < n;
    double z[QLA_N];
    {
        const double *zi = zin + b * n * n + (size_t)(rowok ? t : 0) * n;
#pragma unroll
        for (int c = 0; c < QLA_N; ++c) z[c] = c < n ? zi[c] : 0.0;
    }
    const int nit = cnt[2 * b];
    const int32_t *hdr = hdr_all + b * 2 * (size_t)capit;
    const qls_d2 *rot = rot_all + b * (size_t)cap;
    int l = nit > 0 ? hdr[0] : 0, m = nit > 0 ? hdr[1] : 0;
    int ln = nit > 1 ? hdr[2] : 0, mn = nit > 1 ? hdr[3] : 0;      // the (l, m) of the iteration after: known two iterations ahead
    int r = 0;
    {
        const int cn = m - l;
        if (lane < cn) stg[wave][0][m - 1 - lane] = rot[lane];
        if (lane + 64 < cn) stg[wave][0][m - 1 - lane - 64] = rot[lane + 64];
    }
    for (int itn = 0; itn < nit; ++itn) {
        r += m - l;
        const int lnn = itn + 2 < nit ? hdr[2 * itn + 4] : 0, mnn = itn + 2 < nit ? hdr[2 * itn + 5] : 0;
        const int cn = itn + 1 < nit ? mn - ln : 0;
        qls_d2 nx0 = qls_d2{0.0, 0.0}, nx1 = qls_d2{0.0, 0.0};
        if (lane < cn) nx0 = rot[r + lane];
        if (lane + 64 < cn) nx1 = rot[r + lane + 64];
        __builtin_amdgcn_wave_barrier();
        qla_sweep<(QLA_N - 2) / QLA_BLK>(z, stg[wave][itn & 1], l, m);
        __builtin_amdgcn_wave_barrier();
        if (lane < cn) stg[wave][(itn + 1) & 1][mn - 1 - lane] = nx0;
        if (lane + 64 < cn) stg[wave][(itn + 1) & 1][mn - 1 - lane - 64] = nx1;
        l = ln; m = mn;
        ln = lnn; mn = mnn;
    }
    // the sign of every eigenvector: its first component of largest magnitude becomes positive (row ascending, strict >)
#pragma unroll
    for (int k = 0; k < QLA_N; ++k) {
        if (k < n) {
            const double a = rowok ? __builtin_fabs(z[k]) : -1.0;
            double mx = a;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { const double ot = __shfl_xor(mx, o, 64); mx = ot > mx ? ot : mx; }
            const unsigned long long eq = __ballot(a == mx);
            const int first = __builtin_ctzll(eq);
            const double sv = __shfl(z[k], first, 64);
            if (lane == 0) { wmax[wave][k] = mx; wsv[wave][k] = sv; }
        }
    }
    __syncthreads();
    if (rowok) {
        const int k = t;
        const double sv = wmax[1][k] > wmax[0][k] ? wsv[1][k] : wsv[0][k];
        sgs[k] = sv < 0.0 ? -1.0 : 1.0;
        const double *ev = evin + b * n;
        const double mine = __builtin_fabs(ev[k]);
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double o = __builtin_fabs(ev[j]); rank += (o > mine) || (o == mine && j < k); }
        rk[k] = rank;
        S[b * s_stride + rank] = mine;
    }
    __syncthreads();
    double *Uo = Ut + b * ut_stride;
#pragma unroll
    for (int k = 0; k < QLA_N; ++k) {
        if (k < n && rowok) Uo[(size_t)rk[k] * n + t] = sgs[k] * z[k];
    }
}

// ------------------------------------------------ launch order of the gradient-jump kernel
// Counting sort of the chains by the NUTS step size of their rank (half-octave classes, smallest first = longest
// trees first; a rank that has no step size yet is in class 0: its first call searches for one), then dealt across the
// waves like cards.  The order only decides which chains share a wave; it is not stable and need not be.
__device__ __forceinline__ int gj_class(const double *gj, size_t r)
{
    const double eps = gj[r * GJ_NSTATE + GJ_EPS];
    if (gj[r * GJ_NSTATE + GJ_HAVE_EPS] == 0.0 || !(eps > 0.0)) return 0;
    const int ex = (int)((__double_as_longlong(eps) >> 52) & 0x7FF) - 1023;          // floor(log2 eps)
    const int half = (int)((__double_as_longlong(eps) >> 51) & 1);                   // upper half of the octave
    const int c = 2 * (ex + 24) + half + 1;                                          // eps = 2^-24 -> class 1
    return c < 1 ? 1 : (c > GJ_BUCKETS - 1 ? GJ_BUCKETS - 1 : c);
}
__global__ void gj_order_count_kernel(const double *gj, const int32_t *temp_of, long long nch, int nt, int32_t *bucket)
{
    __shared__ int32_t local[GJ_BUCKETS];                 // most chains fall into two or three classes: count per block first
    for (int b = (int)threadIdx.x; b < GJ_BUCKETS; b += (int)blockDim.x) local[b] = 0;
    __syncthreads();
    const long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < nch) atomicAdd(&local[gj_class(gj, (size_t)(ch / nt) * nt + temp_of[ch])], 1);
    __syncthreads();
    for (int b = (int)threadIdx.x; b < GJ_BUCKETS; b += (int)blockDim.x)
        if (local[b]) atomicAdd(&bucket[b], local[b]);
}
__global__ void gj_order_scan_kernel(int32_t *bucket)
{
    if (threadIdx.x != 0) return;
    int32_t run = 0;
    for (int b = 0; b < GJ_BUCKETS; ++b) { bucket[GJ_BUCKETS + b] = run; run += bucket[b]; bucket[2 * GJ_BUCKETS + b] = 0; }
}
__global__ void gj_order_fill_kernel(const double *gj, const int32_t *temp_of, long long nch, int nt, int32_t *bucket, int32_t *order, int cpw, int nsolo)
{
    // a block reserves its share of every class with one atomic per class; inside the block the order is by thread
    __shared__ int32_t local[GJ_BUCKETS], base[GJ_BUCKETS];
    for (int b = (int)threadIdx.x; b < GJ_BUCKETS; b += (int)blockDim.x) local[b] = 0;
    __syncthreads();
    const long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = ch < nch ? gj_class(gj, (size_t)(ch / nt) * nt + temp_of[ch]) : 0;
    const int mine = ch < nch ? atomicAdd(&local[b], 1) : 0;
    __syncthreads();
    for (int bb = (int)threadIdx.x; bb < GJ_BUCKETS; bb += (int)blockDim.x)
        base[bb] = local[bb] ? atomicAdd(&bucket[2 * GJ_BUCKETS + bb], local[bb]) : 0;
    __syncthreads();
    if (ch >= nch) return;
    // rank t in the sorted list (longest trees first) -> chain slot: consecutive ranks go to DIFFERENT waves, so the few
    // chains with long trees are dealt one per wave and a launch lasts as long as its slowest chain, not the slowest sum
    const long long t = bucket[GJ_BUCKETS + b] + base[b] + mine;
    // the first nsolo chains of the list (the longest trees) get a wave each -- its first chain slot, the others stay empty (-1: the
    // caller fills the array with it) -- and the rest is dealt over the waves behind them
    if (t < nsolo) { order[t * cpw] = (int32_t)ch; return; }
    const long long tr = t - nsolo, nrest = nch - nsolo;
    const long long nw = nrest / cpw, whole = nw * cpw;
    order[(long long)nsolo * cpw + (tr < whole ? (tr % nw) * cpw + tr / nw : tr)] = (int32_t)ch;
}

// ----------------------------------------------------------------- selftest
__global__ void selftest_math_kernel(int op, const double *in, const double *in2, double *out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double r;
    switch (op) {
    case 0: r = det_log(x); break;
    case 1: r = det_exp(x); break;
    case 2: r = det_cos2pi(x); break;
    case 3: r = det_sqrt(x); break;
    case 4: r = x / in2[i]; break;
    case 5: r = det_normal((u64)__double_as_longlong(x), (u64)__double_as_longlong(in2[i])); break;
    case 10: r = unit_log((u64)__double_as_longlong(x)); break;
    case 11: case 12: {
        u32 j;
        double t, sn, cs;
        unit_angle64((u64)__double_as_longlong(x), j, t);
        unit_sincos(j, t, sn, cs);
        r = op == 11 ? cs : sn;
        break;
    }
    case 13: {
        u32 j;
        double t, sn, cs;
        unit_angle32((u32)(u64)__double_as_longlong(x), j, t);
        unit_sincos(j, t, sn, cs);
        r = det_sqrt(-2.0 * unit_log((u64)__double_as_longlong(in2[i]))) * cs;
        break;
    }
    default: r = group_sum<16>(x); break;
    }
    out[i] = r;
}
__global__ void selftest_philox_kernel(const u32 *ck, u32 *out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 *c = ck + i * 6;
    u64 w0, w1;
    philox_words(((u64)c[5] << 32) | c[4], ((u64)c[1] << 32) | c[0], c[2], c[3], w0, w1);
    out[i * 4 + 0] = (u32)w0; out[i * 4 + 1] = (u32)(w0 >> 32);
    out[i * 4 + 2] = (u32)w1; out[i * 4 + 3] = (u32)(w1 >> 32);
}

// ------------------------------------------------------------------- engine
struct Shape { int G, EPL; };
static bool pick_shape(int d, bool grad, Shape *s)
{
    static const Shape table[] = {
#define PTMI_TABLE_ENTRY(G_, E_) {G_, E_},
        PTMI_SHAPE_LIST(PTMI_TABLE_ENTRY)};
    const int G = grad ? ptmi_lanes_for_grad(d) : ptmi_lanes_for(d);
    for (const Shape &c : table)
        if (c.G == G && c.G * c.EPL >= d && (!grad || c.EPL <= 8) && (!ptmi_shape_exact(c.G, c.EPL) || c.G * c.EPL == d)) { *s = c; return true; }
    return false;
}

static KArgs make_args(ptmi_engine *h)
{
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.box_off = -1;
    a.tab_off = -1;
    const ptmi_config &c = h->cfg;
    const ptmi_buffers &b = h->buf;
    a.X = b.X; a.lnL = b.lnL; a.lp = b.lp; a.temp_of = b.temp_of; a.slot_of = b.slot_of;
    a.Ut = b.Ut; a.S = b.S; a.DE = b.DE; a.AM = c.temp0 == 0 ? b.AM : nullptr; a.AMaux = c.temp0 == 0 ? b.AMaux : nullptr;
#ifdef PTMI_MEASURE      // measurement builds only (-DPTMI_MEASURE; results are wrong): what the AM-row stores of the step kernels cost
    static const bool no_am = getenv("PTMI_MEASURE_NO_AM") != nullptr;
    if (no_am) a.AM = nullptr;
#endif
    a.AMflag = c.temp0 == 0 ? (AmFlag *)b.AMflag : nullptr;
    a.rp_draws = h->rp_draws;
    a.nacc = (u64 *)b.nacc; a.jstat = (u64 *)b.jstat;
    a.temps_mh = h->d_temps; a.beta = h->d_beta; a.logl_par = h->d_loglpar; a.logp_par = h->d_logppar;
    a.gsize = h->d_gsize; a.gmask = h->d_gmask; a.gcn = h->d_gcn; a.gdiv = h->d_gdiv; a.ngroups = c.ngroups > 1 ? c.ngroups : 1;
    a.Q = b.Q; a.qaux = b.qaux; a.Q2 = nullptr; a.sloc = nullptr; a.q_cur = 0; a.q_tgt = 0;
    a.seed = c.seed;
    a.d = c.ndim; a.nt = c.ntemps; a.W = c.nwalkers; a.ntg = c.ntemps_global; a.temp0 = c.temp0; a.walker0 = c.walker0;
    a.w_host = c.w_host; a.w_scam = c.w_scam; a.w_am = c.w_am; a.w_de = c.w_de; a.de_on = h->de_on; a.de_size = c.de_size; a.de_head = h->de_head;
    a.cov_update = c.cov_update; a.tskip = c.tskip; a.per_walker = c.cov_per_walker; a.logp_kind = c.logp_kind;
    a.pick_walker = c.pick_mode == PTMI_PICK_WALKER;
    a.de_ld = h->G == 4 ? 8 * ((h->EPL + 1) / 2) : c.ndim;
    a.lanes = h->G;
    a.am_epl = am_row_epl(h->G, h->EPL);
    a.w_nuts = c.w_nuts; a.w_hmc = c.w_hmc; a.gj_nburn = c.gj_nburn; a.hmc_min = c.hmc_min; a.hmc_max = c.hmc_max;
    a.nuts_maxdepth = c.nuts_maxdepth; a.hmc_eps = c.hmc_eps; a.nuts_delta = c.nuts_delta;
    a.gj_tab = h->d_gj_tab; a.gj_diag = h->gj_diag; a.gj = b.gj; a.gj_scr = h->d_gj_scr; a.gj_scal = h->d_gj_scal;
    return a;
}

static ptmi_shape_fn shape_fn(int G, int EPL, int L)
{
    if (L == PTMI_LOGL_INTERVAL) {
#define PTMI_PICK_SHAPE3(G_, E_) if (G == G_ && EPL == E_) return ptmi_shape_##G_##_##E_##_3;
        PTMI_GJ_SHAPE_LIST(PTMI_PICK_SHAPE3)
        return nullptr;
    }
#define PTMI_PICK_SHAPE(G_, E_)                                                                        \
    if (G == G_ && EPL == E_) return L == 0 ? ptmi_shape_##G_##_##E_##_0 : (L == 1 ? ptmi_shape_##G_##_##E_##_1 : ptmi_shape_##G_##_##E_##_2);
    PTMI_SHAPE_LIST(PTMI_PICK_SHAPE)
    return nullptr;
}
static int run_shape(ptmi_engine *h, int op, KArgs &a, int grid, bool full)
{
    const int L = (op == PTMI_OP_PROPOSE || op == PTMI_OP_ACCEPT) ? 0 : h->cfg.logl_kind;   // the split kernels live in family 0
    ptmi_shape_fn f = shape_fn(h->G, h->EPL, L);
    if (!f) return fail(PTMI_EUNSUPPORTED, "no kernel shape for ndim=%d", h->cfg.ndim);
    return f(op, h, a, grid, full);
}

// am_row0 / swap_last of a launch; a swap iteration may only be the last one of the range
static int set_step_args(const ptmi_engine *h, KArgs *a)
{
    const ptmi_config &c = h->cfg;
    a->am_row0 = (int)(a->iter0 % c.cov_update);
#ifdef PTMI_MEASURE      // measurement builds only (-DPTMI_MEASURE; results are wrong): every AM row of a walker into ONE cache-resident row
    static const bool am_small = getenv("PTMI_MEASURE_AM_SMALL") != nullptr;
    if (am_small) { a->cov_update = 1; a->am_row0 = 0; }
#endif
    if (a->AMflag != nullptr && a->nsteps > 0 && (a->iter0 - 1) / c.cov_update != (a->iter0 + a->nsteps - 2) / c.cov_update && a->iter0 > 0)
        return fail(PTMI_EINVAL, "with AM row flags a launch may not cross a multiple of cov_update (iterations %lld..%lld, cov_update=%d)",
                    a->iter0, a->iter0 + a->nsteps - 1, c.cov_update);
    a->swap_last = 0;
    if (c.tskip > 0 && c.ntemps_global > 1) {
        const long long last = a->iter0 + a->nsteps - 1;
        if (last / c.tskip != (a->iter0 - 1) / c.tskip && last % c.tskip != 0)
            return fail(PTMI_EINVAL, "iterations %lld..%lld contain a swap iteration (Tskip=%d) before their end", a->iter0, last, c.tskip);
        if ((last / c.tskip) - ((a->iter0 - 1) / c.tskip) > 1)
            return fail(PTMI_EINVAL, "iterations %lld..%lld span more than one swap epoch", a->iter0, last);
        a->swap_last = last % c.tskip == 0;
    }
    return PTMI_OK;
}

static int chains_grid(const ptmi_engine *h)
{
    const long long nch = (long long)h->cfg.nwalkers * h->cfg.ntemps;
    const int cpb = 256 / h->G;
    return (int)((nch + cpb - 1) / cpb);
}

// am_gemm_kernel for up to max_events events (blocks beyond the actual count leave at once: no host round trip for it)
template <int G, int MAXT>
static int launch_am_gemm_t(ptmi_engine *h, long long max_events)
{
    const size_t lds = sizeof(double) * 2 * (2 * G) * 64;
    auto kern = am_gemm_kernel<G, MAXT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
    }
    const int d = h->cfg.ndim, ntile = (d + 15) / 16, parts = (ntile + 4 * MAXT - 1) / (4 * MAXT);      // blocks per event tile
    const int ngr = h->cfg.ngroups > 1 ? h->cfg.ngroups : 1;
    // parameter groups: a launch per group (am_gemm_kernel); with per-walker covariances a grid row per walker, its blocks enough for
    // every pick of the walker's chains in the piece
    const int pw = h->cfg.cov_per_walker ? 1 : 0;
    const long long per_key = pw ? (max_events / h->cfg.nwalkers) : max_events;
    // (a grid's z extent ends at 65535: more walkers than that go in several launches)
    const int nz = pw ? h->cfg.nwalkers : 1;
    for (int g = 0; g < ngr; ++g)
        for (int z0 = 0; z0 < nz; z0 += 65535)
            hipLaunchKernelGGL(kern, dim3((unsigned)((per_key + 63) / 64), parts, (unsigned)(nz - z0 < 65535 ? nz - z0 : 65535)), dim3(256), lds, h->stream,
                               (const AmEvent *)h->d_am_ev, (const long long *)h->d_am_base, (long long)h->cfg.nwalkers * h->cfg.ntemps, d,
                               (const double *)h->buf.Ut, (const double *)h->buf.S, h->cfg.seed, h->d_am_inc, ngr > 1 ? h->gsize_host[g] : d,
                               h->d_am_perm ? (const long long *)h->d_am_kbase : nullptr, (const int32_t *)h->d_am_perm, g, ngr, z0);
    return PTMI_OK;
}
static int launch_am_gemm(ptmi_engine *h, long long max_events)
{
    const int ntile = (h->cfg.ndim + 15) / 16;                          // output tiles of an increment
    if (h->G == 4) return launch_am_gemm_t<4, 4>(h, max_events);        // (parameter groups at ndim <= 104: 7 tiles at most)
    if (h->G == 16) return ntile <= 16 ? launch_am_gemm_t<16, 4>(h, max_events) : launch_am_gemm_t<16, 8>(h, max_events);
    if (h->G == 64) return launch_am_gemm_t<64, 8>(h, max_events);
    return fail(PTMI_EINVAL, "AM increments ahead of the launch: unknown shape");
}

extern "C" {

const char *ptmi_last_error(void) { return g_err; }
int ptmi_version(void) { return PTMI_VERSION; }
int ptmi_lanes_for(int ndim) { return ndim <= 104 ? 4 : (ndim <= 416 ? 16 : 64); }
// gradient jumps keep seven chain vectors in registers: at most 8 slots per lane (shapes (4,8), (16,7), (64,8))
int ptmi_lanes_for_grad(int ndim) { return ndim <= 32 ? 4 : (ndim <= 112 ? 16 : (ndim <= 512 ? 64 : 0)); }

// PT:699-720: geometric ladder T_i = Tmin * tstep^i; the spacing is 1 + sqrt(2/ndim) unless Tmax (> 0) or tstep (> 0)
// fixes it.  Host arithmetic in the reference's operation order (libm pow / exp / log, as NumPy's scalar path).
int ptmi_temperature_ladder(int nchain, int ndim, double Tmin, double Tmax, double tstep, double *out)
{
    if (!out || nchain < 1 || ndim < 1) return fail(PTMI_EINVAL, "ladder: bad argument");
    double step = tstep;
    if (!(step > 0.0)) step = Tmax > 0.0 ? exp(log(Tmax / Tmin) / (double)(nchain - 1)) : 1.0 + sqrt(2.0 / (double)ndim);
    for (int i = 0; i < nchain; ++i) out[i] = nchain > 1 ? Tmin * pow(step, (double)i) : 1.0;
    return PTMI_OK;
}

// doubles per row of the DE buffer and the slots per lane of its piece-cyclic format (0 = rows in element order), see de_update_kernel
int ptmi_de_row_stride(int ndim, int grad, int *stride, int *epl)
{
    Shape s;
    if (!stride || !epl) return fail(PTMI_EINVAL, "NULL argument");
    if (!pick_shape(ndim, grad != 0, &s)) return fail(PTMI_EUNSUPPORTED, "ndim=%d not supported", ndim);
    *stride = s.G == 4 ? 8 * ((s.EPL + 1) / 2) : ndim;
    *epl = s.G == 4 ? s.EPL : 0;
    return PTMI_OK;
}

// row format of the AM buffer for a given ndim (grad != 0: with gradient jumps in the cycle): *epl == 0: parameter order; *epl > 0:
// the lanes' order of the exact 4-lane shape (include/ptmi.h)
int ptmi_am_row_format(int ndim, int grad, int *epl)
{
    Shape s;
    if (!epl) return fail(PTMI_EINVAL, "NULL argument");
    if (!pick_shape(ndim, grad != 0, &s)) return fail(PTMI_EUNSUPPORTED, "ndim=%d not supported", ndim);
    *epl = am_row_epl(s.G, s.EPL);
    return PTMI_OK;
}

int ptmi_device_count(int *count)
{
    if (!count) return fail(PTMI_EINVAL, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(PTMI_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return PTMI_OK;
}

static int upload(double **dst, const double *src, long long n)
{
    *dst = nullptr;
    if (n <= 0 || !src) return PTMI_OK;
    HIPCHK(hipMalloc((void **)dst, sizeof(double) * (size_t)n));
    HIPCHK(hipMemcpy(*dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    return PTMI_OK;
}

// AM row flags serve the pooled covariance (the per-walker recurrence of PT:778-794 takes every row in turn) on the GPU that holds
// rank 0
int ptmi_am_flags_ok(const ptmi_config *c)
{
    if (!c) return 0;
    return !c->cov_per_walker && c->temp0 == 0;
}

int ptmi_create(const ptmi_config *cfg, const ptmi_buffers *buf, ptmi_handle *out)
{
    if (!cfg || !buf || !out) return fail(PTMI_EINVAL, "NULL argument");
    *out = nullptr;
    const ptmi_config &c = *cfg;
    if (c.ndim < 1 || c.ntemps < 1 || c.nwalkers < 1) return fail(PTMI_EINVAL, "ndim/ntemps/nwalkers must be >= 1");
    if (c.ntemps_global < c.ntemps || c.temp0 < 0 || c.temp0 + c.ntemps > c.ntemps_global)
        return fail(PTMI_EINVAL, "temperature block [%d,%d) outside ladder of %d", c.temp0, c.temp0 + c.ntemps, c.ntemps_global);
    if (c.w_host < 0 || c.w_scam < 0 || c.w_am < 0 || c.w_de < 0 || c.w_nuts < 0 || c.w_hmc < 0 ||
        c.w_host + c.w_scam + c.w_am + c.w_nuts + c.w_hmc <= 0)
        return fail(PTMI_EINVAL, "No jump proposals specified! (PTMCMCSampler.py:267)");
    if (c.cov_update < 1) return fail(PTMI_EINVAL, "cov_update must be >= 1");
    if (c.w_de > 0 && c.de_size < 2) return fail(PTMI_EINVAL, "de_size must be >= 2 when DE is used");
    if (c.logl_kind < 0 || c.logl_kind > PTMI_LOGL_INTERVAL || c.logp_kind < 0 || c.logp_kind > PTMI_LOGP_BOX)
        return fail(PTMI_EINVAL, "unknown logl/logp kind");
    if (c.logl_kind == PTMI_LOGL_INTERVAL && c.logl_par_len != 3LL * c.ndim)
        return fail(PTMI_EINVAL, "interval logl needs a[d] + w[d] + log w[d] parameters");
    if (c.logl_kind == PTMI_LOGL_INTERVAL && (c.w_host > 0 || c.ngroups > 1))
        return fail(PTMI_EUNSUPPORTED, "the interval logl runs in the fused kernels with one parameter group");
    if (c.logl_kind == PTMI_LOGL_DENSE && c.logl_par_len != (long long)c.ndim * (c.ndim + 1))
        return fail(PTMI_EINVAL, "dense logl needs mu[d] + Pt[d*d] parameters");
    if (c.logl_kind == PTMI_LOGL_CURVED && (c.ndim & 1)) return fail(PTMI_EINVAL, "curved logl needs an even ndim");
    if (c.logp_kind == PTMI_LOGP_BOX && c.logp_par_len != 2LL * c.ndim) return fail(PTMI_EINVAL, "box prior needs lo[d] + hi[d]");
    if (!c.ladder || !c.temps_mh) return fail(PTMI_EINVAL, "ladder / temps_mh missing");
    if (c.ngroups < 0 || c.ngroups > 1024) return fail(PTMI_EINVAL, "ngroups out of range");
    if (c.swap_mode != PTMI_SWAP_SWEEP && c.swap_mode != PTMI_SWAP_ODDEVEN) return fail(PTMI_EINVAL, "unknown swap_mode %d", c.swap_mode);
    if (c.pick_mode != PTMI_PICK_CHAIN && c.pick_mode != PTMI_PICK_WALKER) return fail(PTMI_EINVAL, "unknown pick_mode %d", c.pick_mode);
    if (c.ngroups > 1) {
        if (!c.group_size || !c.group_mask) return fail(PTMI_EINVAL, "group_size / group_mask missing");
        for (int g = 0; g < c.ngroups; ++g)
            if (c.group_size[g] < 1 || c.group_size[g] > c.ndim) return fail(PTMI_EINVAL, "group %d has %d parameters", g, c.group_size[g]);
    }
    const bool gj = c.w_nuts + c.w_hmc > 0;
    const bool gshape = gj || c.logl_kind == PTMI_LOGL_INTERVAL;                     // (the interval family lives in the gradient-jump shapes)
    if (gshape && !gj && c.ndim > 512) return fail(PTMI_EUNSUPPORTED, "the interval logl is built for ndim <= 512 (got %d)", c.ndim);
    if (c.w_nuts < 0 || c.w_hmc < 0) return fail(PTMI_EINVAL, "negative gradient-jump weight");
    if (gj) {
        if (!c.gj_tab) return fail(PTMI_EINVAL, "gradient jumps need the whitening tables (gj_tab)");
        if (!buf->gj) return fail(PTMI_EINVAL, "gradient jumps need the gj buffer");
        if (c.ndim > 512) return fail(PTMI_EUNSUPPORTED, "gradient jumps on the device are built for ndim <= 512 (got %d)", c.ndim);
        if (c.ngroups > 1) return fail(PTMI_EUNSUPPORTED, "gradient jumps with parameter groups are not built");
        if (c.w_host > 0) return fail(PTMI_EUNSUPPORTED, "gradient jumps on the device cannot be mixed with host-served jumps");
        if (c.nuts_maxdepth < 0 || c.nuts_maxdepth > 24) return fail(PTMI_EINVAL, "nuts_maxdepth out of range");
        if (c.w_hmc > 0 && (c.hmc_min < 0 || c.hmc_max <= c.hmc_min)) return fail(PTMI_EINVAL, "HMC needs 0 <= hmc_min < hmc_max");
    }
    if (!buf->X || !buf->lnL || !buf->lp || !buf->temp_of || !buf->slot_of || !buf->Ut || !buf->S || !buf->nacc || !buf->jstat)
        return fail(PTMI_EINVAL, "a required device buffer is NULL");
    if (c.w_de > 0 && !buf->DE) return fail(PTMI_EINVAL, "DE weight > 0 but no DE buffer");
    if (buf->AMflag) {
        if (!ptmi_am_flags_ok(cfg)) return fail(PTMI_EINVAL, "AM row flags (ptmi_buffers.AMflag) serve the pooled covariance on the GPU that holds rank 0 (ptmi_am_flags_ok)");
        if (!buf->AM) return fail(PTMI_EINVAL, "AM row flags need the AM buffer");
        if ((long long)c.nwalkers * c.cov_update > 0x7FFFFFFFLL) return fail(PTMI_EINVAL, "AM row flags index rows with 32 bits: nwalkers * cov_update too large");
    }
    if ((unsigned long long)c.nwalkers * (unsigned)c.ntemps_global > 0xFFFFFFFFull) return fail(PTMI_EINVAL, "too many RNG streams");
    Shape s;
    if (!pick_shape(c.ndim, gshape, &s)) return fail(PTMI_EUNSUPPORTED, "ndim=%d not supported (max 2048)", c.ndim);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(PTMI_ENODEVICE, "no HIP device visible: libptmi has no CPU fallback");
    if (c.device < 0 || c.device >= ndev) return fail(PTMI_EINVAL, "device %d of %d", c.device, ndev);
    HIPCHK(hipSetDevice(c.device));
    ptmi_engine *h = new (std::nothrow) ptmi_engine();
    if (!h) return fail(PTMI_EINVAL, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->cfg = c; h->buf = *buf; h->stream = (hipStream_t)c.stream; h->G = s.G; h->EPL = s.EPL;
    std::vector<double> beta((size_t)c.ntemps);
    for (int t = 0; t < c.ntemps; ++t) beta[(size_t)t] = 1.0 / c.temps_mh[t];   // 1/self.temp, PT:612
    // likelihood parameters on the device.  Dense: mu | Pt | Tl, Tl = the half of the symmetric part of P the VALUE is summed over
    // (Tl[k][i] = Ps[k][i] for k > i, Ps[i][i] / 2 for k == i, 0 for k < i: -r^T P r / 2 = -sum_i r_i sum_{k >= i} Tl[k][i] r_k,
    // half the products of the full form; the oracle's dense_par builds the same table); Pt stays for the gradient -P r
    std::vector<double> loglpar(c.logl_par, c.logl_par + (c.logl_par ? c.logl_par_len : 0));
    if (c.logl_kind == PTMI_LOGL_DENSE) {
        const size_t d = (size_t)c.ndim;
        loglpar.resize(d + 2 * d * d, 0.0);
        const double *Pt = loglpar.data() + d;
        double *Tl = loglpar.data() + d + d * d;
        for (size_t k = 0; k < d; ++k)
            for (size_t i = 0; i <= k; ++i) {
                const double ps = (Pt[k * d + i] + Pt[i * d + k]) * 0.5;
                Tl[k * d + i] = k == i ? ps * 0.5 : ps;
            }
    }
    int rc;
    if ((rc = upload(&h->d_ladder, c.ladder, c.ntemps_global)) || (rc = upload(&h->d_temps, c.temps_mh, c.ntemps)) ||
        (rc = upload(&h->d_beta, beta.data(), c.ntemps)) || (rc = upload(&h->d_loglpar, loglpar.data(), (long long)loglpar.size())) ||
        (rc = upload(&h->d_logppar, c.logp_par, c.logp_par_len))) {
        ptmi_destroy(h);
        return rc;
    }
    {   // per-group constants of the AM / DE scales: 2.4/sqrt(2 n_g) (PT:928) and sqrt(2 n_g) (PT:976)
        const int Ng = c.ngroups > 1 ? c.ngroups : 1;
        std::vector<double> gcn((size_t)Ng), gdiv((size_t)Ng), gmask((size_t)Ng * c.ndim, 1.0);
        std::vector<int32_t> gsize((size_t)Ng, c.ndim);
        for (int g = 0; g < Ng; ++g) {
            if (c.ngroups > 1) gsize[(size_t)g] = c.group_size[g];
            gcn[(size_t)g] = 2.4 / sqrt(2.0 * (double)gsize[(size_t)g]);
            gdiv[(size_t)g] = sqrt(2.0 * (double)gsize[(size_t)g]);
        }
        if (c.ngroups > 1) memcpy(gmask.data(), c.group_mask, sizeof(double) * gmask.size());
        h->gsize_host = (int32_t *)malloc(sizeof(int32_t) * Ng);
        if (h->gsize_host) memcpy(h->gsize_host, gsize.data(), sizeof(int32_t) * Ng);
        hipError_t e2 = h->gsize_host ? hipMalloc((void **)&h->d_gsize, sizeof(int32_t) * Ng) : hipErrorOutOfMemory;
        if (e2 == hipSuccess) e2 = hipMemcpy(h->d_gsize, gsize.data(), sizeof(int32_t) * Ng, hipMemcpyHostToDevice);
        if (e2 != hipSuccess || (rc = upload(&h->d_gcn, gcn.data(), Ng)) || (rc = upload(&h->d_gdiv, gdiv.data(), Ng)) ||
            (rc = upload(&h->d_gmask, gmask.data(), (long long)gmask.size()))) {
            ptmi_destroy(h);
            return e2 != hipSuccess ? fail(PTMI_EHIP, "group tables: %s", hipGetErrorString(e2)) : rc;
        }
    }
    if (gj) {
        const size_t nch = (size_t)c.nwalkers * c.ntemps, lanes = (size_t)s.G * s.EPL;
        const size_t nvec = (size_t)GJV_TOP + (size_t)GJL_VECS * (c.nuts_maxdepth + 1);
        hipError_t e3 = hipMalloc((void **)&h->d_gj_scr, sizeof(double) * nvec * lanes * nch);
        if (e3 == hipSuccess) e3 = hipMalloc((void **)&h->d_gj_scal, sizeof(double) * (size_t)GJS_SCALARS * (c.nuts_maxdepth + 1) * nch);
        static const bool unordered = getenv("PTMI_GJ_UNORDERED") != nullptr;        // measurement switch: same results either way
        // chains with a wave of their own (the longest trees of the launch order): up to 1024, a 32nd of the chains at most (config-5 share
        // with 0 / 128 / 512 / 1024 / 2048 of 65 536: 4.67e8 / 4.92e8 / 4.82e8 / 5.37e8 / 5.18e8 updates/s); PTMI_GJ_SOLO=n overrides (0: none; a
        // measurement / test switch: the order never enters a chain's arithmetic)
        const int cpw = 64 / s.G;
        long long solo = cpw > 1 ? (long long)(nch / 32 < 1024 ? nch / 32 : 1024) : 0;
        if (const char *ev = getenv("PTMI_GJ_SOLO")) solo = cpw > 1 ? atoll(ev) : 0;
        if (solo > (long long)nch) solo = (long long)nch;
        if (solo < 0) solo = 0;
        h->gj_solo = (int)solo;
        if (e3 == hipSuccess && !unordered) e3 = hipMalloc((void **)&h->d_gj_order, sizeof(int32_t) * (nch + (size_t)solo * cpw + cpw));
        if (e3 == hipSuccess && !unordered) e3 = hipMalloc((void **)&h->d_gj_bucket, sizeof(int32_t) * 3 * GJ_BUCKETS);
        if (e3 != hipSuccess || (rc = upload(&h->d_gj_tab, c.gj_tab, 3LL * c.ndim * c.ndim))) {
            ptmi_destroy(h);
            return e3 != hipSuccess ? fail(PTMI_EHIP, "gradient-jump scratch: %s", hipGetErrorString(e3)) : rc;
        }
        // diagonal whitening (cov0 diagonal: the curved-likelihood runs start from the identity): a product is d multiplications
        // (the oracle's tab_vec defines the same rule); PTMI_GJ_NODIAG: the general product (a measurement / test switch)
        h->gj_diag = getenv("PTMI_GJ_NODIAG") == nullptr;
        for (long long i = 0; i < 3LL * c.ndim * c.ndim && h->gj_diag; ++i) {
            const long long r = (i / c.ndim) % c.ndim, col = i % c.ndim;
            if (r != col && c.gj_tab[i] != 0.0) h->gj_diag = 0;
        }
    }
    h->cfg.gj_tab = nullptr;
    h->cfg.ladder = h->cfg.temps_mh = h->cfg.logl_par = h->cfg.logp_par = h->cfg.group_mask = nullptr;  // host copies are not kept
    h->cfg.group_size = nullptr;
    hipError_t e = hipMalloc((void **)&h->d_pre, sizeof(SwapPre) * (size_t)c.nwalkers * c.ntemps_global);
    if (e == hipSuccess && !c.cov_per_walker && c.temp0 == 0) {
        const int SL = pool_slab(c.nwalkers, c.ndim), nslab = (c.nwalkers + SL - 1) / SL;
        e = hipMalloc((void **)&h->d_pool_part, sizeof(double) * (size_t)nslab * c.ndim * (c.ndim + 1));
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_pool_T, sizeof(double) * (size_t)c.ndim * (c.ndim + 1));
        if (buf->AMflag) {                                                // the slabs' lists of stored rows (pool_rle_kernel)
            const size_t nrows = (size_t)c.nwalkers * c.cov_update;
            if (e == hipSuccess) e = hipMalloc((void **)&h->d_rle_ent, sizeof(PoolEnt) * nrows);
            if (e == hipSuccess) e = hipMalloc((void **)&h->d_rle_cnt, sizeof(int32_t) * (size_t)nslab);
        }
    }
    // AM increments ahead of the launch (am_gemm_kernel): the 16- and 64-lane shapes with one pooled table, ndim <= 1024
    // ... and every shape with parameter groups (PT:129-145: a chain's pick has its own group, hence its own table: the step kernels'
    // matrix-core product shares one table between the 16 chains of a wave, and the vector-pipe product they fall back to takes
    // 284 ms per 100 steps of the default mix at 64 x 4096 x 100-d with three groups -- 35 times the one-group kernel)
    // -- with groups also per-walker covariances (an event's table is then its walker's group table: lists per (walker, group)): 505 ms
    // per 100 steps of the default mix at 64 x 4096 x 100-d with three groups before
    // -- and the gradient-jump shapes at 16 / 64 lanes per chain (the interval family, NUTS / HMC cycles at ndim > 32): their step kernels'
    // own AM product is the vector pipe's too
    // -- and per-walker covariances at 16 / 64 lanes per chain without groups (one key per walker)
    // -- and the SPLIT path of every shape (ptmi_propose / ptmi_accept_propose on contiguous rows, csrc/ptmi_split.hip): an AM pick's 2 d^2
    // flop belong on the matrix cores there too; the row kernel reads the increment as it reads a SCAM direction.  For handles the fused
    // kernels do not take this way (4 lanes per chain, one group: their own matrix-core product) the scratch serves the split path alone.
    const bool am_main = c.w_am > 0 && (c.ngroups > 1 || s.G > 4) && c.ndim <= 1024 && c.w_host == 0 && !getenv("PTMI_NO_AM_AHEAD");
    const bool am_split = !am_main && buf->Q != nullptr && c.w_am > 0 && c.ndim <= 1024 && c.w_host == 0 && !getenv("PTMI_NO_SPLIT_AM");
    if (e == hipSuccess && (am_main || am_split)) {
        const long long nch = (long long)c.nwalkers * c.ntemps;
        const char *mb = getenv(am_main ? "PTMI_AM_BUDGET_MB" : "PTMI_SPLIT_AM_BUDGET_MB");     // scratch for the increments of one piece (default 6 GB; 2 GB for the split path alone)
        const double budget = (mb ? atof(mb) : (am_main ? 6144.0 : 2048.0)) * 1048576.0;
        long long piece = (long long)(budget / ((double)c.ndim * 8.0 * (double)nch));
        piece = piece < 1 ? 1 : (piece > 64 ? 64 : piece);
        if ((c.ngroups > 1 || c.cov_per_walker) && nch * piece > 0x7FFFFFFFLL) piece = 0x7FFFFFFFLL / nch;      // (the group lists index the events with 32 bits; nch itself is below 2^32 / ntemps)
        if (piece < 1) piece = 1;
        h->am_piece = am_main ? (int)piece : 0;
        h->split_am_piece = (int)piece;
        h->am_cap = nch * piece;
        if (buf->Q != nullptr) e = hipMalloc((void **)&h->d_am_next, sizeof(long long) * (size_t)(nch + 1));
        if (e == hipSuccess)
        e = hipMalloc((void **)&h->d_am_ev, sizeof(AmEvent) * (size_t)h->am_cap);
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_am_count, sizeof(int32_t) * (size_t)(nch + (nch + 1023) / 1024));      // counts | the scan's block sums
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_am_base, sizeof(long long) * (size_t)(nch + 1));
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_am_inc, sizeof(double) * (size_t)h->am_cap * c.ndim);
        if (e == hipSuccess && (c.ngroups > 1 || c.cov_per_walker)) {    // the events listed key by key: totals | cursors | the scan's block sums; list starts (+ the end)
            const size_t nkeys = (size_t)(c.ngroups > 1 ? c.ngroups : 1) * (c.cov_per_walker ? (size_t)c.nwalkers : 1);
            if (h->am_cap > 0x7FFFFFFFLL) e = hipErrorInvalidValue;
            if (e == hipSuccess) e = hipMalloc((void **)&h->d_am_grp, sizeof(int32_t) * (2 * nkeys + (nkeys + 1023) / 1024 + 1));
            if (e == hipSuccess) e = hipMalloc((void **)&h->d_am_kbase, sizeof(long long) * (nkeys + 1));
            if (e == hipSuccess) e = hipMalloc((void **)&h->d_am_perm, sizeof(int32_t) * (size_t)h->am_cap);
        }
        if (e == hipErrorOutOfMemory) {
            // no room for the scratch: the step kernels compute their own AM products (slower, the same results) instead of failing the
            // handle -- configurations that fitted before this path existed still do
            (void)hipGetLastError();
            (void)hipFree(h->d_am_ev); (void)hipFree(h->d_am_count); (void)hipFree(h->d_am_base); (void)hipFree(h->d_am_inc);
            (void)hipFree(h->d_am_grp); (void)hipFree(h->d_am_perm); (void)hipFree(h->d_am_kbase);
            h->d_am_ev = nullptr; h->d_am_count = nullptr; h->d_am_base = nullptr; h->d_am_inc = nullptr;
            h->d_am_grp = nullptr; h->d_am_perm = nullptr; h->d_am_kbase = nullptr;
            (void)hipFree(h->d_am_next); h->d_am_next = nullptr;
            h->am_piece = 0; h->am_cap = 0; h->split_am_piece = 0;
            e = hipSuccess;
        }
    }
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { ptmi_destroy(h); return fail(PTMI_EHIP, "create: %s", hipGetErrorString(e)); }
    *out = h;
    return PTMI_OK;
}

static void dc_plan_free(ptmi_engine *h);

int ptmi_destroy(ptmi_handle h)
{
    if (!h) return PTMI_OK;
    (void)hipFree(h->d_ladder); (void)hipFree(h->d_temps); (void)hipFree(h->d_beta); (void)hipFree(h->d_loglpar); (void)hipFree(h->d_logppar);
    (void)hipFree(h->d_pre); (void)hipFree(h->d_xint); (void)hipFree(h->d_hop);
    if (h->h_hop) { (void)hipHostFree(h->h_hop); (void)hipEventDestroy(h->hop_ev); }
    (void)hipFree(h->d_gsize); (void)hipFree(h->d_gmask); (void)hipFree(h->d_gcn); (void)hipFree(h->d_gdiv); (void)hipFree(h->d_pool_part); (void)hipFree(h->d_pool_T);
    (void)hipFree(h->d_ql_scr); (void)hipFree(h->d_qlg_scr); (void)hipFree(h->d_sy_scr); (void)hipFree(h->d_utpad);
    free(h->gsize_host);
    dc_plan_free(h);
    if (h->h_sy_info) (void)hipHostFree(h->h_sy_info);
    if (h->sy_lib) {                                                // SyLib: the library's handle, its destructor
        void **sl = (void **)h->sy_lib;
        if (sl[0] && sl[1]) ((int (*)(void *))sl[1])(sl[0]);
        free(h->sy_lib);
    }
    (void)hipFree(h->d_rle_ent); (void)hipFree(h->d_rle_cnt);
    (void)hipFree(h->d_am_ev); (void)hipFree(h->d_am_count); (void)hipFree(h->d_am_base); (void)hipFree(h->d_am_inc);
    (void)hipFree(h->d_am_grp); (void)hipFree(h->d_am_perm); (void)hipFree(h->d_am_kbase); (void)hipFree(h->d_am_next); (void)hipFree(h->d_iter);
    (void)hipFree(h->d_gj_tab); (void)hipFree(h->d_gj_scr); (void)hipFree(h->d_gj_scal); (void)hipFree(h->d_gj_order); (void)hipFree(h->d_gj_bucket);
    if (h->side) { (void)hipStreamDestroy(h->side); (void)hipEventDestroy(h->side_go); (void)hipEventDestroy(h->side_done); }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return PTMI_OK;
}

int ptmi_sync(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    return PTMI_OK;
}

int ptmi_set_de_active(ptmi_handle h, int on)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (on && (h->cfg.w_de <= 0 || !h->buf.DE)) return fail(PTMI_EINVAL, "DE has no weight or no buffer");
    h->de_on = on ? 1 : 0;
    h->split_am_lo = h->split_am_hi = 0;                         // the cycle changed: increments made ahead (ptmi_split_am_prepare) are void
    return PTMI_OK;
}

int ptmi_eval_state(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const KArgs a = make_args(h);
    const int grid = chains_grid(h);
    KArgs aa = a;
    if (int rc = run_shape(h, PTMI_OP_EVAL, aa, grid, false)) return rc;
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

// rows of ld >= d doubles, zero beyond column d: the 16- / 64-lane step kernels read a row with unconditional loads
__global__ __launch_bounds__(256) void ut_pad_kernel(const double *Ut, double *out, int d, int ld, unsigned long long *absmax)
{
    const double *src = Ut + (size_t)blockIdx.x * d;
    double *dst = out + (size_t)blockIdx.x * ld;
    double am = 0.0;
    for (int i = (int)threadIdx.x; i < ld; i += 256) {
        const double v = i < d ? src[i] : 0.0;
        dst[i] = v;
        am = __builtin_fabs(v) > am ? __builtin_fabs(v) : am;
    }
    // max |U| of the table (non-negative doubles order like their bit patterns): the box prior's fast path bounds a SCAM jump by it
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(am, o, 64); am = t > am ? t : am; }
    if ((threadIdx.x & 63) == 0) atomicMax(absmax, (unsigned long long)__double_as_longlong(am));
}
// ONE table for the launch and a wide shape: the padded copy the step kernels read (the caller's Ut may have changed since the
// last launch -- an epoch, put_eig -- so it is made anew every launch: 16 MB of traffic at ndim = 1000 beside a launch of milliseconds)
static int make_ut_pad(ptmi_engine *h, KArgs *a)
{
    const ptmi_config &c = h->cfg;
    a->UtPad = nullptr;
    a->ut_pad_ld = 0;
    a->ut_absmax = nullptr;
    static const bool off = getenv("PTMI_NO_UTPAD") != nullptr;       // measurement / test switch: same results either way
    if (h->G <= 4 || c.cov_per_walker || c.ngroups > 1 || off) return PTMI_OK;
    const int ld = h->G * h->EPL;
    if (!h->d_utpad) HIPCHK(hipMalloc((void **)&h->d_utpad, sizeof(double) * ((size_t)c.ndim * ld + 2)));
    unsigned long long *amax = (unsigned long long *)(h->d_utpad + (size_t)c.ndim * ld);
    HIPCHK(hipMemsetAsync(amax, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(ut_pad_kernel, dim3(c.ndim), dim3(256), 0, h->stream, (const double *)h->buf.Ut, h->d_utpad, c.ndim, ld, amax);
    a->UtPad = h->d_utpad;
    a->ut_pad_ld = ld;
    a->ut_absmax = (const double *)amax;
    return PTMI_OK;
}

// The AM picks of iterations iter0 .. iter0 + ns - 1 listed (am_count / am_scan / am_fill) and their increments computed on the matrix
// cores (am_gemm_kernel) into h->d_am_inc: increment j of a chain's picks in the piece is row h->d_am_base[chain] + j.
static int am_prepare(ptmi_engine *h, long long iter0, int ns)
{
    const ptmi_config &c = h->cfg;
    const long long nch = (long long)c.nwalkers * c.ntemps;
    AmArgs p;
    p.seed = c.seed; p.iter0 = iter0; p.nch = nch; p.nsteps = ns; p.nt = c.ntemps; p.ntg = c.ntemps_global; p.temp0 = c.temp0;
    p.walker0 = c.walker0; p.w_host = c.w_host; p.w_scam = c.w_scam; p.w_am = c.w_am; p.w_de = h->de_on ? c.w_de : 0;
    p.w_gj = c.w_nuts + c.w_hmc;
    p.pick_walker = c.pick_mode == PTMI_PICK_WALKER; p.ngroups = c.ngroups > 1 ? c.ngroups : 1; p.gcn = h->d_gcn;
    p.per_walker = c.cov_per_walker ? 1 : 0;
    p.temp_of = h->buf.temp_of; p.temps_mh = h->d_temps;
    const unsigned gch = (unsigned)((nch + 255) / 256);
    const int ngr = p.ngroups;
    const long long nkeys = (long long)ngr * (c.cov_per_walker ? c.nwalkers : 1);
    int32_t *gtot = h->d_am_perm ? h->d_am_grp : nullptr, *gcur = gtot ? gtot + nkeys : nullptr, *gpart = gtot ? gtot + 2 * nkeys : nullptr;
    if (gtot) HIPCHK(hipMemsetAsync(gtot, 0, sizeof(int32_t) * 2 * (size_t)nkeys, h->stream));          // the keys' totals and the fill's cursors
    hipLaunchKernelGGL(am_count_kernel, dim3(gch), dim3(256), 0, h->stream, p, h->d_am_count, gtot);
    if (gtot) {                                                  // the lists' starts: the chains' scan over the keys' totals
        const unsigned gk = (unsigned)((nkeys + 1023) / 1024);
        hipLaunchKernelGGL(am_scan_sums_kernel, dim3(gk), dim3(1024), 0, h->stream, (const int32_t *)gtot, gpart, nkeys);
        hipLaunchKernelGGL(am_scan_kernel, dim3(gk), dim3(1024), 0, h->stream, (const int32_t *)gtot, (const int32_t *)gpart, h->d_am_kbase, nkeys);
    }
    const unsigned gsc = (unsigned)((nch + 1023) / 1024);
    hipLaunchKernelGGL(am_scan_sums_kernel, dim3(gsc), dim3(1024), 0, h->stream, (const int32_t *)h->d_am_count, h->d_am_count + nch, nch);
    hipLaunchKernelGGL(am_scan_kernel, dim3(gsc), dim3(1024), 0, h->stream, (const int32_t *)h->d_am_count, (const int32_t *)(h->d_am_count + nch),
                       h->d_am_base, nch);
    hipLaunchKernelGGL(am_fill_kernel, dim3(gch), dim3(256), 0, h->stream, p, (const long long *)h->d_am_base, (AmEvent *)h->d_am_ev,
                       (const long long *)h->d_am_kbase, gcur, gtot ? h->d_am_perm : nullptr);
    if (int rc = launch_am_gemm(h, nch * ns)) return rc;
    return PTMI_OK;
}

int ptmi_mh_steps(ptmi_handle h, int64_t iter0, int32_t nsteps)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (nsteps < 0 || iter0 < 0) return fail(PTMI_EINVAL, "iter0/nsteps negative");
    if (nsteps == 0) return PTMI_OK;
    KArgs a = make_args(h);
    a.iter0 = iter0; a.nsteps = nsteps;
    if (int rc = set_step_args(h, &a)) return rc;
    if (h->cfg.w_host > 0) return fail(PTMI_EINVAL, "host-served jumps need the split path (ptmi_propose / ptmi_accept)");
    const bool gjc = h->cfg.w_nuts + h->cfg.w_hmc > 0;       // the fused kernel with the NUTS / HMC branch (csrc/ptmi_gj.inc.h)
    auto launch_gj = [&](KArgs &a) -> int {
        if (h->cfg.w_nuts > 0 && h->d_gj_order) {                                   // chains of similar step size share a wave
            const long long nch = (long long)h->cfg.nwalkers * h->cfg.ntemps;
            const unsigned g = (unsigned)((nch + 255) / 256);
            HIPCHK(hipMemsetAsync(h->d_gj_bucket, 0, sizeof(int32_t) * GJ_BUCKETS, h->stream));
            hipLaunchKernelGGL(gj_order_count_kernel, dim3(g), dim3(256), 0, h->stream, (const double *)h->buf.gj, (const int32_t *)h->buf.temp_of,
                               nch, h->cfg.ntemps, h->d_gj_bucket);
            hipLaunchKernelGGL(gj_order_scan_kernel, dim3(1), dim3(64), 0, h->stream, h->d_gj_bucket);
            const int cpw = 64 / h->G;
            const long long nslots = (long long)h->gj_solo * cpw + ((nch - h->gj_solo + cpw - 1) / cpw) * cpw;
            HIPCHK(hipMemsetAsync(h->d_gj_order, 0xFF, sizeof(int32_t) * (size_t)nslots, h->stream));           // -1: an empty chain slot
            hipLaunchKernelGGL(gj_order_fill_kernel, dim3(g), dim3(256), 0, h->stream, (const double *)h->buf.gj, (const int32_t *)h->buf.temp_of,
                               nch, h->cfg.ntemps, h->d_gj_bucket, h->d_gj_order, cpw, h->gj_solo);
            a.gj_order = h->d_gj_order;
            a.gj_nslots = (int)nslots;
        }
        if (int rc = run_shape(h, PTMI_OP_MH_GJ, a, chains_grid(h), true)) return rc;
        h->last_variant = PTMI_VAR_GRADJUMP | PTMI_VAR_FULL;
        return PTMI_OK;
    };
    if (gjc && h->am_piece <= 0) {
        if (int rc = launch_gj(a)) return rc;
        HIPCHK(hipGetLastError());
        return PTMI_OK;
    }
    const bool full = h->cfg.w_am > 0 || (h->de_on && h->cfg.w_de > 0);
    if (!full && h->cfg.w_scam <= 0 && !gjc) return fail(PTMI_EINVAL, "empty proposal cycle");
    const int grid = chains_grid(h);
    if (!gjc)
        if (int rc = make_ut_pad(h, &a)) return rc;
    if (h->am_piece > 0) {
        // large ndim: the launch goes in pieces, each behind the matrix product that computes its AM increments
        const ptmi_config &c = h->cfg;
        const long long nch = (long long)c.nwalkers * c.ntemps;
        for (int s0 = 0; s0 < nsteps; s0 += h->am_piece) {
            const int ns = nsteps - s0 < h->am_piece ? nsteps - s0 : h->am_piece;
            if (int rc = am_prepare(h, iter0 + s0, ns)) return rc;
            KArgs ap = make_args(h);
            ap.iter0 = iter0 + s0; ap.nsteps = ns;
            if (int rc = set_step_args(h, &ap)) return rc;
            ap.am_inc = h->d_am_inc; ap.am_base = h->d_am_base;
            ap.UtPad = a.UtPad; ap.ut_pad_ld = a.ut_pad_ld; ap.ut_absmax = a.ut_absmax;
            if (gjc) {                                                   // (the gradient-jump kernel reads its AM increments the same way)
                if (int rc = launch_gj(ap)) return rc;
                continue;
            }
            if (int rc = run_shape(h, PTMI_OP_MH, ap, grid, full)) return rc;
        }
        HIPCHK(hipGetLastError());
        return PTMI_OK;
    }
    if (int rc = run_shape(h, PTMI_OP_MH, a, grid, full)) return rc;
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_last_mh_variant(ptmi_handle h, int32_t *variant)
{
    if (!h || !variant) return fail(PTMI_EINVAL, "NULL argument");
    *variant = h->last_variant | ((h->cfg.pick_mode == PTMI_PICK_WALKER && (h->last_variant & PTMI_VAR_FULL)) ? PTMI_VAR_UNIFORM : 0) |
               (h->G << 12) | (h->EPL << 20);
    return PTMI_OK;
}

// ptmi_device_iter: the split calls' `iter` is an offset from the counter in device memory (launches captured in a graph).  The row
// kernels derive the ring row and the swap-iteration test from the counter themselves; what the host cannot know then it cannot check.
static int split_iter_args(ptmi_engine *h, KArgs *a, int mode)
{
    if (!h->dev_iter) return set_step_args(h, a);
    if (!ptmi_split_rows_ok(h) || h->cfg.w_am > 0)
        return fail(PTMI_EUNSUPPORTED, "ptmi_device_iter serves the row kernels' cycles without AM entries (the AM increments are listed on the host's iteration)");
    (void)mode;
    a->iter_dev = h->d_iter;
    a->am_row0 = 0; a->swap_last = 0;
    return PTMI_OK;
}

// The AM increments the split path's row kernel reads for the proposals of iteration `it`: the prepared piece when it covers `it`
// (ptmi_split_am_prepare), else a piece of that one iteration made now (the tables as they are at this call).
static int split_am_args(ptmi_engine *h, KArgs *a, long long it)
{
    if (h->cfg.w_am <= 0) return PTMI_OK;
    if (!(it >= h->split_am_lo && it < h->split_am_hi))
        if (int rc = ptmi_split_am_prepare(h, it, 1)) return rc;
    a->am_inc = h->d_am_inc;
    a->am_next = h->d_am_next;
    return PTMI_OK;
}

int ptmi_propose(ptmi_handle h, int64_t iter)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!h->buf.Q || !h->buf.qaux) return fail(PTMI_EINVAL, "split path needs the Q and qaux buffers");
    KArgs a = make_args(h);
    a.iter0 = iter; a.nsteps = 1;
    if (int rc = split_iter_args(h, &a, 0)) return rc;
    h->q_cur = 0;                                            // the proposals go to Q
    if (ptmi_split_rows_ok(h)) {
        if (h->buf.Q2 && h->buf.sloc) { a.Q2 = h->buf.Q2; a.sloc = h->buf.sloc; }
        if (int rc = split_am_args(h, &a, iter)) return rc;
        if (int rc = ptmi_split_rows(h, a, 0)) return rc;
    } else {
        const int grid = chains_grid(h);
        if (int rc = run_shape(h, PTMI_OP_PROPOSE, a, grid, true)) return rc;
    }
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_accept(ptmi_handle h, int64_t iter, const double *newlnL, const double *newlp)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!h->buf.Q || !h->buf.qaux || !newlnL || !newlp) return fail(PTMI_EINVAL, "split path buffers missing");
    KArgs a = make_args(h);
    a.iter0 = iter; a.nsteps = 1; a.newlnL = newlnL; a.newlp = newlp;
    if (int rc = split_iter_args(h, &a, 1)) return rc;
    if (ptmi_split_rows_ok(h)) {
        if (h->buf.Q2 && h->buf.sloc) { a.Q2 = h->buf.Q2; a.sloc = h->buf.sloc; }
        a.q_cur = h->q_cur;
        if (int rc = ptmi_split_rows(h, a, 1)) return rc;
    } else {
        const int grid = chains_grid(h);
        if (int rc = run_shape(h, PTMI_OP_ACCEPT, a, grid, true)) return rc;
    }
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_accept_propose(ptmi_handle h, int64_t iter, const double *newlnL, const double *newlp)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!h->buf.Q || !h->buf.qaux || !newlnL || !newlp) return fail(PTMI_EINVAL, "split path buffers missing");
    const ptmi_config &c = h->cfg;
    // nothing may sit between the two iterations: a swap (iter a multiple of Tskip), a covariance or DE epoch (the caller's: they
    // change the tables the proposal of iter + 1 reads)
    if (!h->dev_iter && c.tskip > 0 && c.ntemps_global > 1 && iter % c.tskip == 0)
        return fail(PTMI_EINVAL, "ptmi_accept_propose(%lld): a swap iteration (Tskip=%d) is accepted with ptmi_accept, the swap follows", (long long)iter, c.tskip);
    KArgs a = make_args(h);
    a.iter0 = iter; a.nsteps = 1; a.newlnL = newlnL; a.newlp = newlp;
    if (int rc = split_iter_args(h, &a, 2)) return rc;
    if (ptmi_split_rows_ok(h)) {
        a.q_cur = a.q_tgt = h->q_cur;
        if (h->buf.Q2 && h->buf.sloc) { a.Q2 = h->buf.Q2; a.sloc = h->buf.sloc; a.q_tgt = 1 - h->q_cur; }
        if (int rc = split_am_args(h, &a, iter + 1)) return rc;
        if (int rc = ptmi_split_rows(h, a, 2)) return rc;
        h->q_cur = a.q_tgt;
        HIPCHK(hipGetLastError());
        return PTMI_OK;
    }
    // configurations the row kernels do not serve (AM entries in the cycle): the two shape kernels back to back
    const int grid = chains_grid(h);
    if (int rc = run_shape(h, PTMI_OP_ACCEPT, a, grid, true)) return rc;
    KArgs b = make_args(h);
    b.iter0 = iter + 1; b.nsteps = 1;
    if (int rc = set_step_args(h, &b)) return rc;
    if (int rc = run_shape(h, PTMI_OP_PROPOSE, b, grid, true)) return rc;
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_rows_logl(ptmi_handle h, const double *rows, int64_t n, double *out)
{
    if (!h || !rows || !out || n < 0) return fail(PTMI_EINVAL, "bad argument");
    if (h->cfg.logl_kind != PTMI_LOGL_ISO) return fail(PTMI_EUNSUPPORTED, "ptmi_rows_logl serves the isotropic Gaussian (PTMI_LOGL_ISO)");
    if (n == 0) return PTMI_OK;
    if (int rc = ptmi_rows_iso(h, rows, (long long)n, out)) return rc;
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_set_proposals(ptmi_handle h, int32_t which)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    h->q_cur = (which && h->buf.Q2) ? 1 : 0;
    return PTMI_OK;
}

int ptmi_set_stream(ptmi_handle h, void *stream)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    h->stream = (hipStream_t)stream;
    h->cfg.stream = stream;
    return PTMI_OK;
}

int ptmi_device_iter(ptmi_handle h, int32_t on)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (on && !h->d_iter) HIPCHK(hipMalloc((void **)&h->d_iter, sizeof(long long)));
    h->dev_iter = on ? 1 : 0;
    return PTMI_OK;
}

int ptmi_set_device_iter(ptmi_handle h, int64_t value)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!h->d_iter) HIPCHK(hipMalloc((void **)&h->d_iter, sizeof(long long)));
    if (int rc = ptmi_set_iter_device(h, h->d_iter, (long long)value)) return rc;
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_split_am_piece(ptmi_handle h, int32_t *piece)
{
    if (!h || !piece) return fail(PTMI_EINVAL, "NULL argument");
    *piece = (h->cfg.w_am > 0 && ptmi_split_rows_ok(h)) ? h->split_am_piece : 0;
    return PTMI_OK;
}

int ptmi_split_am_prepare(ptmi_handle h, int64_t iter0, int32_t nsteps)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (h->cfg.w_am <= 0) return PTMI_OK;
    if (h->split_am_piece <= 0 || !h->d_am_next) return fail(PTMI_EUNSUPPORTED, "this handle's split path takes its AM proposals from the shape kernels");
    if (nsteps < 1 || nsteps > h->split_am_piece) return fail(PTMI_EINVAL, "ptmi_split_am_prepare: 1 <= nsteps <= %d (ptmi_split_am_piece)", h->split_am_piece);
    if (int rc = am_prepare(h, (long long)iter0, nsteps)) return rc;
    const long long nch = (long long)h->cfg.nwalkers * h->cfg.ntemps;
    HIPCHK(hipMemcpyAsync(h->d_am_next, h->d_am_base, sizeof(long long) * (size_t)(nch + 1), hipMemcpyDeviceToDevice, h->stream));      // every chain's cursor at its first increment
    HIPCHK(hipGetLastError());
    h->split_am_lo = iter0;
    h->split_am_hi = iter0 + nsteps;
    return PTMI_OK;
}

int ptmi_proposals(ptmi_handle h, double **q)
{
    if (!h || !q) return fail(PTMI_EINVAL, "NULL argument");
    *q = (h->q_cur && h->buf.Q2) ? h->buf.Q2 : h->buf.Q;
    return PTMI_OK;
}

int ptmi_swap_write_am(ptmi_handle h, int64_t iter)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (h->cfg.temp0 != 0 || !h->buf.AM) return PTMI_OK;
    hipLaunchKernelGGL(am_write_kernel, dim3(h->cfg.nwalkers), dim3(64), 0, h->stream, (const double *)h->buf.X,
                       (const double *)h->buf.lnL, (const double *)h->buf.lp, (const int32_t *)h->buf.slot_of, h->buf.AM,
                       h->buf.AMaux, h->cfg.nwalkers, h->cfg.ntemps, h->cfg.ndim, h->cfg.cov_update, (long long)iter, am_row_epl(h->G, h->EPL),
                       (AmFlag *)h->buf.AMflag);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

// odd/even mode: swap epoch e = iter / tskip tries the pairs (k, k+1) with k = e (mod 2)
static int swap_parity(const ptmi_config &c, int64_t iter)
{
    return (int)((c.tskip > 0 ? iter / c.tskip : iter) & 1);
}

static int launch_swap_sweep(ptmi_engine *h, int W, int n, const SwapPre *pre, int32_t *slot_of, int32_t *temp_of,
                             int32_t *map, u64 *nswap, int local0, int nlocal, int parity, int32_t *inv, int hop_nt = 0, bool *hop_done = nullptr,
                             const SwapAmRow *amr = nullptr, bool *am_done = nullptr)
{
    if (hop_done) *hop_done = false;
    if (am_done) *am_done = false;
    SwapAmRow none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0, 0, nullptr};
    int wpb = 64;                                                      // 2 tables of wpb x (n + 1) ints must fit the CU's LDS
    while (wpb > 8 && sizeof(int32_t) * (2 * (size_t)wpb * (size_t)(n + 1) + n) > 160 * 1024) wpb /= 2;
    const size_t lds = sizeof(int32_t) * (2 * (size_t)wpb * (size_t)(n + 1) + n);      // forward table, flags, block of a position
    if (lds <= 160 * 1024) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)swap_sweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
        }
        if (hop_nt > 0) {                                              // the multi-hop scan rides on the write-out
            HIPCHK(hipMemsetAsync(h->d_hop, 0, sizeof(int32_t), h->stream));
            if (hop_done) *hop_done = true;
        }
        const bool with_am = amr != nullptr && slot_of != nullptr;
        hipLaunchKernelGGL(swap_sweep_kernel<true>, dim3((unsigned)((W + wpb - 1) / wpb)), dim3(256), lds, h->stream, W, n, h->d_ladder, pre,
                           slot_of, temp_of, map, nswap, local0, nlocal, parity, inv, wpb, hop_nt, h->d_hop, with_am ? *amr : none);
        if (am_done) *am_done = with_am;
    } else {
        hipLaunchKernelGGL(swap_sweep_kernel<false>, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, h->stream, W, n, h->d_ladder, pre,
                           slot_of, temp_of, map, nswap, local0, nlocal, parity, inv, 64, 0, (int32_t *)nullptr, none);
    }
    return PTMI_OK;
}

// The sweep with its records made in the block (swap_fused_kernel) when its tables and the ring fit the LDS; *used says whether it
// was launched (else the caller runs swap_prepare_kernel + swap_sweep_kernel).  PTMI_SWAP_FUSED=0: the two-kernel form (a
// measurement / test switch, same results).
static int launch_swap_fused(ptmi_engine *h, int W, int n, const SwapSrc &src, int32_t *slot_of, int32_t *temp_of, int32_t *map, u64 *nswap,
                             int local0, int nlocal, int parity, int32_t *inv, int hop_nt, bool *hop_done, const SwapAmRow *amr, bool *am_done,
                             bool *used)
{
    *used = false;
    if (hop_done) *hop_done = false;
    if (am_done) *am_done = false;
    const char *sw = getenv("PTMI_SWAP_FUSED");                     // read per call: the tests switch it
    if (sw && atoi(sw) == 0) return PTMI_OK;
    // walkers per block: 16 puts the 4096 walkers of config 2 on every CU (64 per block ran on 64 CUs: 31 -> 24 us per swap epoch
    // at 64 ranks; 8 starve the producers: 38); long ladders are cut further by the LDS their tables need.  PTMI_SWF_WPB: a measurement switch
    int wpb = 64, lg = 6, want = n <= 128 ? 16 : 32;               // 256 ranks: 94 us with 32 or 64, 104 with 16
    if (const char *wv = getenv("PTMI_SWF_WPB")) want = atoi(wv);
    while (wpb > 8 && wpb > want) { wpb /= 2; --lg; }
    while (wpb > 8 && swf_lds_bytes(wpb, n) > 160 * 1024) { wpb /= 2; --lg; }
    const size_t lds = swf_lds_bytes(wpb, n);
    if (lds > 160 * 1024) return PTMI_OK;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)swap_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
    }
    if (hop_nt > 0) {                                              // the multi-hop scan rides on the write-out
        HIPCHK(hipMemsetAsync(h->d_hop, 0, sizeof(int32_t), h->stream));
        if (hop_done) *hop_done = true;
    }
    const SwapAmRow none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0, 0, nullptr};
    const bool with_am = amr != nullptr && slot_of != nullptr;
    hipLaunchKernelGGL(swap_fused_kernel, dim3((unsigned)((W + wpb - 1) / wpb)), dim3(SWF_BLK), lds, h->stream, W, n, src, slot_of, temp_of, map,
                       nswap, local0, nlocal, parity, inv, wpb, lg, hop_nt, h->d_hop, with_am ? *amr : none);
    if (am_done) *am_done = with_am;
    *used = true;
    return PTMI_OK;
}

int ptmi_swap(ptmi_handle h, int64_t iter)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (c.ntemps != c.ntemps_global) return fail(PTMI_EINVAL, "ptmi_swap needs the whole ladder on this GPU; use the three-piece form");
    if (!h->buf.nswap) return fail(PTMI_EINVAL, "nswap buffer missing");
    if (c.ntemps < 2) return PTMI_OK;
    const int W = c.nwalkers;
    if (c.swap_mode == PTMI_SWAP_ODDEVEN) {
        const int parity = swap_parity(c, iter);
        const long long np = (long long)W * ((c.ntemps - parity) / 2);
        if (np > 0)
            hipLaunchKernelGGL(swap_oddeven_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, h->stream, W, c.ntemps,
                               h->d_ladder, (const double *)h->buf.lnL, h->buf.slot_of, h->buf.temp_of, (u64 *)h->buf.nswap,
                               (long long)iter, c.seed, c.walker0, parity);
        HIPCHK(hipGetLastError());
        return ptmi_swap_write_am(h, iter);
    }
    const SwapSrc src = {h->d_ladder, (const double *)nullptr, (const double *)h->buf.lnL, (const int32_t *)h->buf.slot_of, (long long)iter, c.seed,
                         c.walker0, 0, h->rp_swap_u};
    // the sweep's write-out also stores the swap iteration's AM row (one kernel and one launch gap less per swap epoch)
    const SwapAmRow amr = {(const double *)h->buf.X, (const double *)h->buf.lnL, (const double *)h->buf.lp, h->buf.AM, h->buf.AMaux,
                           c.ndim, c.cov_update, am_row_epl(h->G, h->EPL), (long long)iter, (AmFlag *)h->buf.AMflag};
    bool am_done = false, used = false;
    if (int rc = launch_swap_fused(h, W, c.ntemps, src, h->buf.slot_of, h->buf.temp_of, (int32_t *)nullptr, (u64 *)h->buf.nswap, 0, c.ntemps, -1,
                                   (int32_t *)nullptr, 0, nullptr, (c.temp0 == 0 && h->buf.AM) ? &amr : nullptr, &am_done, &used)) return rc;
    if (used) {
        HIPCHK(hipGetLastError());
        return am_done ? PTMI_OK : ptmi_swap_write_am(h, iter);
    }
    hipLaunchKernelGGL(swap_prepare_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)c.ntemps), dim3(256), 0, h->stream, W, c.ntemps, src,
                       (SwapPre *)h->d_pre);
    if (int rc = launch_swap_sweep(h, W, c.ntemps, (const SwapPre *)h->d_pre, h->buf.slot_of, h->buf.temp_of,
                                   (int32_t *)nullptr, (u64 *)h->buf.nswap, 0, c.ntemps, -1, (int32_t *)nullptr, 0, nullptr,
                                   (c.temp0 == 0 && h->buf.AM) ? &amr : nullptr, &am_done)) return rc;
    HIPCHK(hipGetLastError());
    return am_done ? PTMI_OK : ptmi_swap_write_am(h, iter);
}

int ptmi_swap_gather_lnl(ptmi_handle h, double *out)
{
    if (!h || !out) return fail(PTMI_EINVAL, "NULL argument");
    const long long n = (long long)h->cfg.nwalkers * h->cfg.ntemps;
    hipLaunchKernelGGL(gather_lnl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->buf.lnL,
                       h->buf.slot_of, out, n, h->cfg.ntemps);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

// exchange scratch: inv[W][ntg] (written by the sweep), newslot[W][T], arr_slot[nranks][W], lv_slot[2][W], lv_rank[2][W], err[1]
static size_t xint_count(const ptmi_config &c)
{
    const size_t W = (size_t)c.nwalkers, nr = (size_t)((c.ntemps_global + c.ntemps - 1) / c.ntemps);
    return W * c.ntemps_global + W * c.ntemps + nr * W + 4 * W + 1;
}
static int ensure_xint(ptmi_engine *h)
{
    if (h->d_xint) return PTMI_OK;
    HIPCHK(hipMalloc((void **)&h->d_xint, sizeof(int32_t) * xint_count(h->cfg)));
    HIPCHK(hipMemsetAsync(h->d_xint, 0, sizeof(int32_t) * xint_count(h->cfg), h->stream));
    HIPCHK(hipMalloc((void **)&h->d_hop, sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(h->d_hop, 0, sizeof(int32_t), h->stream));
    HIPCHK(hipHostMalloc((void **)&h->h_hop, sizeof(int32_t), hipHostMallocDefault));
    *h->h_hop = 0;
    HIPCHK(hipEventCreateWithFlags(&h->hop_ev, hipEventDisableTiming));
    return PTMI_OK;
}

static int sweep_global(ptmi_handle h, int64_t iter, const double *lnL, int32_t *map, int block_nt)
{
    if (!h || !lnL || !map) return fail(PTMI_EINVAL, "NULL argument");
    if (!h->buf.nswap) return fail(PTMI_EINVAL, "nswap buffer missing");
    const ptmi_config &c = h->cfg;
    const int W = c.nwalkers;
    if (int rc = ensure_xint(h)) return rc;
    const SwapSrc src = {h->d_ladder, lnL, (const double *)nullptr, (const int32_t *)nullptr, (long long)iter, c.seed, c.walker0, block_nt, h->rp_swap_u};
    const int parity = c.swap_mode == PTMI_SWAP_ODDEVEN ? swap_parity(c, iter) : -1;
    bool used = false;
    if (int rc = launch_swap_fused(h, W, c.ntemps_global, src, (int32_t *)nullptr, (int32_t *)nullptr, map, (u64 *)h->buf.nswap, c.temp0, c.ntemps,
                                   parity, h->d_xint /* inv[W][ntemps_global] */, block_nt, &h->hop_from_sweep, nullptr, nullptr, &used)) return rc;
    if (used) {
        HIPCHK(hipGetLastError());
        return PTMI_OK;
    }
    hipLaunchKernelGGL(swap_prepare_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)c.ntemps_global), dim3(256), 0, h->stream, W, c.ntemps_global,
                       src, (SwapPre *)h->d_pre);
    if (int rc = launch_swap_sweep(h, W, c.ntemps_global, (const SwapPre *)h->d_pre, (int32_t *)nullptr,
                                   (int32_t *)nullptr, map, (u64 *)h->buf.nswap, c.temp0, c.ntemps,
                                   c.swap_mode == PTMI_SWAP_ODDEVEN ? swap_parity(c, iter) : -1, h->d_xint /* inv[W][ntemps_global] */,
                                   block_nt, &h->hop_from_sweep)) return rc;
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}
int ptmi_swap_sweep(ptmi_handle h, int64_t iter, const double *lnL_pos_global, int32_t *map)
{
    return sweep_global(h, iter, lnL_pos_global, map, 0);
}
int ptmi_swap_sweep_blocks(ptmi_handle h, int64_t iter, const double *lnL_blocks, int32_t *map)
{
    if (h && h->cfg.ntemps_global % h->cfg.ntemps) return fail(PTMI_EINVAL, "the ladder is not a whole number of blocks");
    return sweep_global(h, iter, lnL_blocks, map, h ? h->cfg.ntemps : 0);
}

// ---- device-side exchange of the rows that cross a block edge --------------------------------------------------
// One wave per walker, one lane per local position.  From the global map and its inverse (both written by the sweep)
// it (1) lists this block's leaving rows (local source, remote destination) and arriving rows (local destination,
// remote source), both in ascending local position -- the k-th arrival takes the slot the k-th departure frees, the
// rule of sharded.py's plan_exchange -- and (2) rewrites slot_of / temp_of.  A hot -> cold sweep moves at most one
// row of a walker down out of a block (the carried state) and at most one up (displaced by one level), hence the
// fixed [2][W] / [nranks][W] tables.  The work is O(local ranks), whatever the length of the whole ladder.
__global__ __launch_bounds__(256) void exchange_plan_kernel(int W, int nt, int ntg, int temp0, int nranks, const int32_t *map,
                                                            int32_t *slot_of, int32_t *temp_of, const int32_t *inv, int32_t *newslot,
                                                            int32_t *arr_slot, int32_t *lv_slot, int32_t *lv_rank, int32_t *err)
{
    const int lane = (int)(threadIdx.x & 63);
    const int w = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= W) return;
    const int me = temp0 / nt;
    const int32_t *m = map + (size_t)w * ntg + temp0, *iv = inv + (size_t)w * ntg + temp0;
    int32_t *so = slot_of + (size_t)w * nt, *to = temp_of + (size_t)w * nt, *ns = newslot + (size_t)w * nt;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));            // lanes below this one
    for (int q = lane; q < nranks; q += 64) arr_slot[(size_t)q * W + w] = -1;
    // departures, ascending local position
    int nlv = 0, freed0 = -1, freed1 = -1, lq0 = -1, lq1 = -1;
    for (int base = 0; base < nt; base += 64) {
        const int p = base + lane;
        const bool valid = p < nt;
        const int q = valid ? iv[p] / nt : me;
        const int slot = valid ? so[p] : -1;
        const bool leaving = valid && q != me;
        u64 mask = __ballot(leaving);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            const int fs = __shfl(slot, l, 64), fq = __shfl(q, l, 64);
            if (nlv == 0) { freed0 = fs; lq0 = fq; }
            else if (nlv == 1) { freed1 = fs; lq1 = fq; }
            ++nlv;
            mask &= mask - 1;
        }
    }
    // arrivals, ascending local position; rows that stay keep their slot
    int narr = 0, aq0 = -1;
    bool bad = nlv > 2 || (nlv == 2 && lq0 == lq1);                       // two destinations on one GPU would collide in send[q][w]
    for (int base = 0; base < nt; base += 64) {
        const int j = base + lane;
        const bool valid = j < nt;
        const int src = valid ? m[j] : temp0;
        const int q = src / nt;
        const bool arriving = valid && q != me;
        const u64 mask = __ballot(arriving);
        const int rank = narr + __builtin_popcountll(mask & lt);
        if (valid) {
            int slot;
            if (!arriving) slot = so[src - temp0];
            else {
                slot = rank == 0 ? freed0 : (rank == 1 ? freed1 : -1);
                if (slot >= 0) arr_slot[(size_t)q * W + w] = slot;
                else { bad = true; slot = 0; }
            }
            ns[j] = slot;
        }
        if (mask) {                                                       // two arrivals from one GPU would collide in recv[q][w]
            const int l0 = __builtin_ctzll(mask);
            const int q0 = __shfl(q, l0, 64);
            if (narr == 0) aq0 = q0;
            else if (q0 == aq0) bad = true;
            const u64 rest = mask & (mask - 1);
            if (rest) {
                const int q1 = __shfl(q, __builtin_ctzll(rest), 64);
                if (q1 == aq0) bad = true;
            }
        }
        narr += __builtin_popcountll(mask);
    }
    if (narr != nlv) bad = true;
    if (lane == 0) {                                                      // -1 = no such departure
        lv_slot[w] = freed0; lv_rank[w] = lq0;
        lv_slot[W + w] = freed1; lv_rank[W + w] = lq1;
    }
    if (__any(bad) && lane == 0) atomicAdd(err, 1);
    for (int base = 0; base < nt; base += 64) {
        const int j = base + lane;
        if (j < nt) { const int sl = ns[j]; so[j] = sl; to[sl] = j; }
    }
}
__global__ void exchange_pack_kernel(int W, int nt, int d, const double *X, const double *lnL, const double *lp,
                                     const int32_t *lv_slot, const int32_t *lv_rank, double *send)
{
    const int w = (int)blockIdx.x, k = (int)blockIdx.y;
    const int slot = lv_slot[(size_t)k * W + w];
    if (slot < 0) return;
    const size_t r = (size_t)w * nt + slot;
    double *dst = send + ((size_t)lv_rank[(size_t)k * W + w] * W + w) * (d + 2);
    for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) dst[i] = X[r * d + i];
    if (threadIdx.x == 0) { dst[d] = lnL[r]; dst[d + 1] = lp[r]; }
}
// a block per walker looks through the source GPUs (at most two of them sent a row): a block per (walker, GPU) was 32 768
// blocks at eight GPUs, nearly all of which found nothing
__global__ void exchange_apply_kernel(int W, int nt, int d, double *X, double *lnL, double *lp, const int32_t *arr_slot,
                                      const double *recv, int nranks)
{
    const int w = (int)blockIdx.x;
    for (int q = 0; q < nranks; ++q) {
        const int slot = arr_slot[(size_t)q * W + w];
        if (slot < 0) continue;
        const size_t r = (size_t)w * nt + slot;
        const double *src = recv + ((size_t)q * W + w) * (d + 2);
        for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) X[r * d + i] = src[i];
        if (threadIdx.x == 0) { lnL[r] = src[d]; lp[r] = src[d + 1]; }
    }
}


// A row travels further than to a neighbouring block when the carried state of the sweep wins every pair of a whole
// block.  Every GPU scans the whole map (identical everywhere), so all of them take the same decision on the transport.
__global__ void exchange_multihop_kernel(int W, int ntg, int nt, const int32_t *map, int32_t *flag)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)W * ntg) return;
    const int j = (int)(idx % ntg);
    const int hop = map[idx] / nt - j / nt;
    if (hop > 1 || hop < -1) atomicOr(flag, 1);
}

int ptmi_exchange_pack(ptmi_handle h, const int32_t *map, double *send)
{
    if (!h || !map || !send) return fail(PTMI_EINVAL, "NULL argument");
    const ptmi_config &c = h->cfg;
    if (c.ntemps_global % c.ntemps) return fail(PTMI_EINVAL, "the ladder is not a whole number of blocks");
    const int W = c.nwalkers, nr = c.ntemps_global / c.ntemps;
    if (int rc = ensure_xint(h)) return rc;
    int32_t *inv = h->d_xint, *newslot = inv + (size_t)W * c.ntemps_global, *arr = newslot + (size_t)W * c.ntemps;
    int32_t *lvs = arr + (size_t)nr * W, *lvr = lvs + 2 * (size_t)W, *err = lvr + 2 * (size_t)W;
    hipLaunchKernelGGL(exchange_plan_kernel, dim3((W + 3) / 4), dim3(256), 0, h->stream, W, c.ntemps, c.ntemps_global, c.temp0, nr,
                       map, h->buf.slot_of, h->buf.temp_of, (const int32_t *)inv, newslot, arr, lvs, lvr, err);
    hipLaunchKernelGGL(exchange_pack_kernel, dim3(W, 2), dim3(64), 0, h->stream, W, c.ntemps, c.ndim, (const double *)h->buf.X,
                       (const double *)h->buf.lnL, (const double *)h->buf.lp, (const int32_t *)lvs, (const int32_t *)lvr, send);
    if (!h->hop_from_sweep) {                                          // the sweep's write-out did not look (tables beyond the LDS)
        HIPCHK(hipMemsetAsync(h->d_hop, 0, sizeof(int32_t), h->stream));
        const long long tot = (long long)W * c.ntemps_global;
        hipLaunchKernelGGL(exchange_multihop_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, W, c.ntemps_global, c.ntemps,
                           map, h->d_hop);
    }
    h->hop_from_sweep = false;
    HIPCHK(hipGetLastError());
    // the flag sets out for the host now, with an event of its own: ptmi_exchange_multihop waits for these four bytes, not for
    // whatever the caller has queued behind the pack step in the meantime (the neighbour exchange)
    HIPCHK(hipMemcpyAsync(h->h_hop, h->d_hop, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->hop_ev, h->stream));
    h->hop_pending = true;
    return PTMI_OK;
}
int ptmi_exchange_multihop(ptmi_handle h, int32_t *flag)
{
    if (!h || !flag) return fail(PTMI_EINVAL, "NULL argument");
    *flag = 0;
    if (!h->d_hop) return PTMI_OK;
    if (h->hop_pending) {
        HIPCHK(hipEventSynchronize(h->hop_ev));
        h->hop_pending = false;
    }
    *flag = *h->h_hop;
    return PTMI_OK;
}
int ptmi_exchange_apply(ptmi_handle h, const double *recv)
{
    if (!h || !recv) return fail(PTMI_EINVAL, "NULL argument");
    if (!h->d_xint) return fail(PTMI_EINVAL, "ptmi_exchange_apply without a preceding ptmi_exchange_pack");
    const ptmi_config &c = h->cfg;
    const int W = c.nwalkers, nr = c.ntemps_global / c.ntemps;
    const int32_t *arr = h->d_xint + (size_t)W * c.ntemps_global + (size_t)W * c.ntemps;
    hipLaunchKernelGGL(exchange_apply_kernel, dim3(W), dim3(64), 0, h->stream, W, c.ntemps, c.ndim, h->buf.X, h->buf.lnL,
                       h->buf.lp, arr, recv, nr);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}
int ptmi_exchange_status(ptmi_handle h, int32_t *violations)
{
    if (!h || !violations) return fail(PTMI_EINVAL, "NULL argument");
    *violations = 0;
    if (!h->d_xint) return PTMI_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(violations, h->d_xint + xint_count(h->cfg) - 1, sizeof(int32_t), hipMemcpyDeviceToHost));
    return PTMI_OK;
}

int ptmi_update_cov(ptmi_handle h, int64_t iter) { return ptmi_update_cov_on(h, iter, nullptr, nullptr, nullptr); }

int ptmi_set_am_buffers(ptmi_handle h, double *AM, double *AMaux, uint64_t *AMflag)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!AM || !h->buf.AM) return fail(PTMI_EINVAL, "the handle was created without an AM buffer, or AM is NULL");
    if ((AMaux != nullptr) != (h->buf.AMaux != nullptr) || (AMflag != nullptr) != (h->buf.AMflag != nullptr))
        return fail(PTMI_EINVAL, "AMaux / AMflag must be given exactly when the handle was created with them");
    h->buf.AM = AM; h->buf.AMaux = AMaux; h->buf.AMflag = AMflag;
    return PTMI_OK;
}

int ptmi_update_cov_on(ptmi_handle hh, int64_t iter, void *stream, const double *AM_in, const uint64_t *AMflag_in)
{
    if (!hh) return fail(PTMI_EINVAL, "NULL handle");
    // a view of the handle with the caller's stream and ring: the statistics below read h->stream / h->buf.AM / h->buf.AMflag only
    // (the scratch, the diagonal tiles' helper stream and mu / M2 / cov are the handle's own: one statistics call at a time)
    ptmi_engine view = *hh;
    if (stream) view.stream = (hipStream_t)stream;
    if (AM_in) { view.buf.AM = const_cast<double *>(AM_in); view.buf.AMflag = const_cast<uint64_t *>(AMflag_in); }
    ptmi_engine *h = &view;
    struct Back { ptmi_engine *to, *from; ~Back() { to->side = from->side; to->side_go = from->side_go; to->side_done = from->side_done; } } back{hh, h};
    const ptmi_config &c = h->cfg;
    if (c.temp0 != 0) return PTMI_OK;   // only the GPU holding rank 0 adapts (PT:545)
    if (!h->buf.AM || !h->buf.mu || !h->buf.M2 || !h->buf.cov) return fail(PTMI_EINVAL, "AM/mu/M2/cov buffers missing");
    if (iter < c.cov_update || iter % c.cov_update) return fail(PTMI_EINVAL, "iter must be a positive multiple of cov_update");
    const int d = c.ndim, nt = (d + WTILE - 1) / WTILE;
    if (c.cov_per_walker) {
        if (d <= 100 && !getenv("PTMI_WELFORD_TILES"))
            hipLaunchKernelGGL((welford_rows_kernel<25, 10, 4, 10>), dim3((c.nwalkers + 1) / 2), dim3(768), 0, h->stream, (const double *)h->buf.AM,
                               h->buf.mu, h->buf.M2, h->buf.cov, d, c.cov_update, (long long)iter, d * d, am_row_epl(h->G, h->EPL), c.nwalkers);
        else
        hipLaunchKernelGGL(welford_kernel<false>, dim3(nt, nt, c.nwalkers), dim3(256), 0, h->stream, (const double *)h->buf.AM,
                           h->buf.mu, h->buf.M2, h->buf.cov, d, c.cov_update, (long long)iter, d * d, am_row_epl(h->G, h->EPL));
        if (nt > 1)
            hipLaunchKernelGGL(welford_mean_kernel, dim3((d + 63) / 64, c.nwalkers), dim3(64), 0, h->stream, (const double *)h->buf.AM,
                               h->buf.mu, d, c.cov_update, (long long)iter, 0, am_row_epl(h->G, h->EPL));
    } else {
        // pooled statistics (orc_pool_update): mu[0 .. d), M2[0 .. d*d) of the buffers are the pooled state
        const int W = c.nwalkers, SL = pool_slab(W, d), nslab = (W + SL - 1) / SL, ng = pool_groups(d);
        const long long nrows = (long long)W * c.cov_update;
        const bool first = iter == c.cov_update;
        const double *shift = first ? (const double *)h->buf.AM : (const double *)h->buf.mu;     // the first epoch: walker 0's row 0
        // Several macro tiles per side: the diagonal ones (nslab x ng blocks, half a round of the chip) go to a side stream and
        // fill the last, partly empty round of the off-diagonal ones instead of a launch of their own behind them
        // (1000-d, 512 walkers: 288 blocks beside 1152 with 512 resident at a time).
        hipStream_t diag_stream = h->stream;
        // AM row flags: the slabs' lists of stored rows and run lengths first, then the sums over them
        const bool rle = h->buf.AMflag != nullptr, pair = d % 2 == 0;
        const PoolRle pr = {(const PoolEnt *)h->d_rle_ent, h->d_rle_cnt};
        const long long rps = (long long)SL * c.cov_update;
        if (rps * d * 8 >= (1ll << 32)) return fail(PTMI_EUNSUPPORTED, "pooled statistics: a slab of %d walkers x %d rows x %d parameters exceeds 4 GB", SL, c.cov_update, d);
        const int aepl = am_row_epl(h->G, h->EPL), sepl = first ? am_row_epl(h->G, h->EPL) : 0;
        if (rle) hipLaunchKernelGGL(pool_rle_kernel, dim3(nslab), dim3(256), 0, h->stream, (const AmFlag *)h->buf.AMflag, nrows, rps, (PoolEnt *)h->d_rle_ent,
                                    h->d_rle_cnt);
        auto syrk = [&](auto diag, dim3 grid, hipStream_t st) {
            constexpr bool DG = decltype(diag)::value;
            const void *fn = rle ? (pair ? (const void *)pool_syrk_kernel<DG, true, true> : (const void *)pool_syrk_kernel<DG, true, false>)
                                 : (pair ? (const void *)pool_syrk_kernel<DG, false, true> : (const void *)pool_syrk_kernel<DG, false, false>);
            const double *rows = (const double *)h->buf.AM;
            long long nr = nrows, rp = rps;
            int dd = d, ae = aepl, se = sepl;
            double *part = h->d_pool_part;
            PoolRle prl = pr;
            void *args[] = {&rows, &nr, &dd, (void *)&shift, &rp, &part, &ae, &se, &prl};
            return hipLaunchKernel(fn, grid, dim3(256), args, 0, st);
        };
        if (ng > 1) {
            if (!h->side) {
                HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->side_go, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&h->side_done, hipEventDisableTiming));
            }
            HIPCHK(hipEventRecord(h->side_go, h->stream));
            HIPCHK(hipStreamWaitEvent(h->side, h->side_go, 0));
            HIPCHK(syrk(std::false_type{}, dim3(ng * (ng - 1) / 2, nslab), h->stream));
            diag_stream = h->side;
        }
        HIPCHK(syrk(std::true_type{}, dim3(ng, nslab), diag_stream));
        if (ng > 1) {
            HIPCHK(hipEventRecord(h->side_done, h->side));
            HIPCHK(hipStreamWaitEvent(h->stream, h->side_done, 0));
        }
        const long long nel = (long long)d * (d + 1);
        hipLaunchKernelGGL(pool_reduce_kernel, dim3((unsigned)((nel + 63) / 64)), dim3(64), 0, h->stream, (const double *)h->d_pool_part, nslab, d,
                           h->d_pool_T);
        const double nb = (double)W * (double)c.cov_update, nprev = (double)W * (double)(iter - c.cov_update);
        hipLaunchKernelGGL(pool_finish_kernel, dim3((unsigned)(((long long)d * d + 255) / 256)), dim3(256), 0, h->stream, (const double *)h->d_pool_T,
                           shift, first ? am_row_epl(h->G, h->EPL) : 0, h->buf.mu, h->buf.M2, h->buf.cov, d, first ? 1 : 0, nb, nprev * nb / (nprev + nb), nb / (nprev + nb),
                           nprev + nb - 1.0);
    }
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_eig_jacobi(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (!h->buf.cov || !h->buf.Ut || !h->buf.S) return fail(PTMI_EINVAL, "cov / Ut / S buffers missing");
    if (c.ngroups > 1) return fail(PTMI_EUNSUPPORTED, "the device eigensolver factorizes the full covariance (no parameter groups)");
    const int d = c.ndim;
    const size_t lds = sizeof(double) * 2 * (size_t)d * d;
    if (lds > 160 * 1024 || d > 101) return fail(PTMI_EUNSUPPORTED, "the device eigensolver keeps two %d x %d tables in LDS: ndim <= 101", d, d);
    if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)eig_jacobi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nmat = c.cov_per_walker ? c.nwalkers : 1;
    hipLaunchKernelGGL(eig_jacobi_kernel, dim3(nmat), dim3(JAC_THREADS), lds, h->stream, (const double *)h->buf.cov, h->buf.Ut, h->buf.S,
                       d, d * d, d);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

// nmat symmetric matrices of order n, packed [nmat][n][n] -> eigenvectors as rows [nmat][n][n], eigenvalues [nmat][n] (see ptmi_eig_ql)
static int eig_ql_run(ptmi_engine *h, int n, int nmat, const double *cov, double *Ut, double *S)
{
    const ptmi_config &c = h->cfg;
    const int d = n, dmax = c.ndim;
    const size_t lds = sizeof(double) * ((((size_t)d * d + 1) & ~(size_t)1) + 2 * (size_t)d);
    if (lds > 160 * 1024 || d > 128) return fail(PTMI_EUNSUPPORTED, "the QL eigensolver keeps the %d x %d matrix in LDS: ndim <= 128", d, d);
    const char *sp = getenv("PTMI_QL_SPLIT");                           // 1 / 0 forces the three-kernel / the one-kernel form (tests, measurements)
    const bool split = sp ? atoi(sp) != 0 : nmat >= 64;
    if (split) {
        // many matrices: reduce -> the scalar chains of all of them at once -> apply (see eig_ql_chain_kernel)
        const int cap = 3 * d * d, capit = 8 * d;
        if (!h->d_ql_scr) {                                             // sized for the full order (a parameter group's matrices are smaller)
            const int capm = 3 * dmax * dmax, capitm = 8 * dmax;
            const size_t bytes = sizeof(double) * (size_t)nmat * ((size_t)dmax * dmax + 2 * (size_t)dmax + (size_t)dmax + 2 * (size_t)capm) +
                                 sizeof(int32_t) * (size_t)nmat * (2 * (size_t)capitm + 2) + 256;
            HIPCHK(hipMalloc((void **)&h->d_ql_scr, bytes));
        }
        QlScratch q;
        char *pb = (char *)h->d_ql_scr;
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        q.z = (double *)pb; pb += up16(sizeof(double) * (size_t)nmat * d * d);
        q.de = (qls_d2 *)pb; pb += sizeof(double) * 2 * (size_t)nmat * d;
        q.rot = (qls_d2 *)pb; pb += sizeof(double) * 2 * (size_t)nmat * cap;
        q.ev = (double *)pb; pb += up16(sizeof(double) * (size_t)nmat * d);
        q.hdr = (int32_t *)pb; pb += sizeof(int32_t) * 2 * (size_t)nmat * capit;
        q.cnt = (int32_t *)pb;
        q.cap = cap; q.capit = capit;
        if (lds > 64 * 1024) {
            HIPCHK(hipFuncSetAttribute((const void *)eig_ql_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(hipFuncSetAttribute((const void *)eig_ql_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL(eig_ql_reduce_kernel, dim3(nmat), dim3(QLR_THREADS), lds, h->stream, cov, d, q);
        hipLaunchKernelGGL(eig_ql_chain_kernel, dim3(nmat), dim3(64), sizeof(double) * 2 * (size_t)d, h->stream, d, q);
        const bool regs = d <= QLA_N && !getenv("PTMI_QL_APPLY_LDS");
        if (regs)
            hipLaunchKernelGGL(eig_ql_apply_reg_kernel, dim3(nmat), dim3(128), 0, h->stream, Ut, S, d, d * d, d, (const double *)q.z,
                               (const double *)q.ev, (const qls_d2 *)q.rot, (const int32_t *)q.hdr, (const int32_t *)q.cnt, cap, capit);
        hipLaunchKernelGGL(eig_ql_apply_kernel, dim3(nmat), dim3(QL_THREADS), lds, h->stream, Ut, S, d, d * d, d, q, regs ? 1 : 0);
        HIPCHK(hipGetLastError());
        return PTMI_OK;
    }
    if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)eig_ql_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(eig_ql_kernel, dim3(nmat), dim3(QL_THREADS), lds, h->stream, cov, Ut, S, d, d * d, d, (int32_t *)nullptr);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

// Parameter groups (PT:129-145, 797-803: one SVD per group's block of the covariance): the group's rows and columns, in ascending
// parameter order, packed into an m x m matrix per walker ...
__global__ __launch_bounds__(256) void group_gather_kernel(const double *cov, const double *gmask, int d, int m, double *sub)
{
    __shared__ int idx[128];
    const double *mk = gmask;                                            // [d] membership of this group
    if (threadIdx.x == 0) {
        int k = 0;
        for (int i = 0; i < d && k < 128; ++i) if (mk[i] != 0.0) idx[k++] = i;
    }
    __syncthreads();
    const double *cw = cov + (size_t)blockIdx.x * d * d;
    double *sw = sub + (size_t)blockIdx.x * m * m;
    for (int t = (int)threadIdx.x; t < m * m; t += 256) sw[t] = cw[(size_t)idx[t / m] * d + idx[t % m]];
}
// ... and its eigenvectors embedded in the full space, one per row of the group's table (zero outside the group, zero rows beyond
// the group's size), the eigenvalues padded with zeros: the layout propose() reads (Ut[Wc][Ng][d][d], S[Wc][Ng][d])
__global__ __launch_bounds__(256) void group_embed_kernel(const double *usub, const double *ssub, const double *gmask, int d, int m, int ng, int gi,
                                                          double *Ut, double *S)
{
    __shared__ int pos[128];                                             // position of parameter i inside the group, or -1
    if (threadIdx.x == 0) {
        int k = 0;
        for (int i = 0; i < d; ++i) pos[i] = gmask[i] != 0.0 ? k++ : -1;
    }
    __syncthreads();
    const double *uw = usub + (size_t)blockIdx.x * m * m, *sw = ssub + (size_t)blockIdx.x * m;
    double *Uo = Ut + ((size_t)blockIdx.x * ng + gi) * d * d, *So = S + ((size_t)blockIdx.x * ng + gi) * d;
    for (int t = (int)threadIdx.x; t < d * d; t += 256) {
        const int k = t / d, i = t % d;
        Uo[t] = (k < m && pos[i] >= 0) ? uw[(size_t)k * m + pos[i]] : 0.0;
    }
    for (int k = (int)threadIdx.x; k < d; k += 256) So[k] = k < m ? sw[k] : 0.0;
}

int ptmi_eig_ql(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (!h->buf.cov || !h->buf.Ut || !h->buf.S) return fail(PTMI_EINVAL, "cov / Ut / S buffers missing");
    const int d = c.ndim, nmat = c.cov_per_walker ? c.nwalkers : 1;
    if (c.ngroups <= 1) return eig_ql_run(h, d, nmat, (const double *)h->buf.cov, h->buf.Ut, h->buf.S);
    if (d > 128) return fail(PTMI_EUNSUPPORTED, "the QL eigensolver keeps a matrix in LDS: ndim <= 128");
    // one factorization per parameter group, as the reference's loop over self.groups (PT:797-803)
    if (!h->d_qlg_scr) HIPCHK(hipMalloc((void **)&h->d_qlg_scr, sizeof(double) * (size_t)nmat * (2 * (size_t)d * d + d)));
    double *sub = (double *)h->d_qlg_scr, *usub = sub + (size_t)nmat * d * d, *ssub = usub + (size_t)nmat * d * d;
    for (int gi = 0; gi < c.ngroups; ++gi) {
        const int m = h->gsize_host[gi];
        const double *mk = h->d_gmask + (size_t)gi * d;
        hipLaunchKernelGGL(group_gather_kernel, dim3(nmat), dim3(256), 0, h->stream, (const double *)h->buf.cov, mk, d, m, sub);
        if (int rc = eig_ql_run(h, m, nmat, (const double *)sub, usub, ssub)) return rc;
        hipLaunchKernelGGL(group_embed_kernel, dim3(nmat), dim3(256), 0, h->stream, (const double *)usub, (const double *)ssub, mk, d, m, c.ngroups, gi,
                           h->buf.Ut, h->buf.S);
    }
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

// ptmi_eig_ql on the caller's stream, from / into the caller's buffers: the engine's eig_lag with per-walker covariances -- the
// factorization of thousands of small matrices (chains of dependent rotations: little of the GPU each) runs BESIDE the step launches of
// the next covariance period instead of between two of them.  One call at a time (the scratch is the handle's).
int ptmi_eig_ql_from(ptmi_handle h, void *stream, const double *cov_in, double *Ut_out, double *S_out)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    const double *cov = cov_in ? cov_in : (const double *)h->buf.cov;
    double *Uo = Ut_out ? Ut_out : h->buf.Ut, *So = S_out ? S_out : h->buf.S;
    if (!cov || !Uo || !So) return fail(PTMI_EINVAL, "cov / Ut / S buffers missing");
    if (c.ngroups > 1) return fail(PTMI_EUNSUPPORTED, "ptmi_eig_ql_from: one parameter group (use ptmi_eig_ql)");
    ptmi_engine view = *h;                                               // eig_ql_run reads the stream, the configuration and the scratch pointer
    if (stream) view.stream = (hipStream_t)stream;
    const int rc = eig_ql_run(&view, c.ndim, c.cov_per_walker ? c.nwalkers : 1, cov, Uo, So);
    h->d_ql_scr = view.d_ql_scr;                                         // (made by the first call)
    return rc;
}

// ---------------------------------------------------------------- one large matrix: tridiagonalization in one kernel
// eig_mode "sytrd" (ptmi_eig_sytrd; ndim <= 1024, one pooled covariance).  The ROCm library's symmetric eigensolver spends two thirds
// of its time reducing the matrix to tridiagonal form in some 7000 launches of one-block kernels (1000 x 1000: 25 of 35 ms of kernel
// time, 3 us each).  Here that step is ONE kernel: the matrix lives in the LDS of its blocks (block b owns the full columns
// b, b + NB, ...: 64 KB of 160 at 1000 x 1000 over 128 blocks), a Householder step is
//   the owner of column k forms v (its own LDS)                                         -> v to all      [grid barrier]
//   every block: p_j = tau (column j . v) for ITS columns (A symmetric: column j is row j)  -> p to all      [grid barrier]
//   every block: w = p - (tau/2 p.v) v, its columns -= v w_j + w v_j
// -- no reduction across blocks, two barriers per column (a few microseconds each on an atomic counter).  Output in LAPACK's
// dsytrd format (uplo = lower: d, e, tau, the reflectors below the subdiagonal), so that the library's divide-and-conquer solver
// for the tridiagonal matrix (rocsolver_dstedc) and its back-transformation (rocsolver_dormtr) take it from there.
struct SytrdArgs {
    double *A;             // [n][n] column-major = row-major (symmetric in); out: the reflectors
    double *D, *E, *tau;   // [n], [n - 1], [n - 1]
    double *vbuf;          // [2][n + 2]: column m as its owner holds it before the update (by parity of m)
    double *pbuf;          // [2][n]: the products p_j (by parity of m)
    unsigned *bar;
    int n;
};
#ifndef PTMI_SY_THREADS
#define PTMI_SY_THREADS 256
#endif
constexpr int SY_THREADS = PTMI_SY_THREADS, SY_NW = SY_THREADS / 64, SY_CMAX = 16, SY_PT = 1024 / SY_THREADS;       // SY_PT: elements of a vector per thread (n <= 1024)
// exchanged data goes through agent-scope relaxed atomics (write-through stores, loads past the caches of the other XCDs): no
// cache write-back / invalidation beside the barrier's own counter
__device__ __forceinline__ void sy_grid_sync(unsigned *bar, unsigned &target, unsigned nb)
{
    // every thread's exchanged stores must be ACKNOWLEDGED before the counter moves.  __syncthreads alone does not wait for them (a
    // workgroup-scope release needs no vmcnt wait on this part: the CU's L1 is the block's own), and the counter's increment is
    // relaxed: beside an idle GPU the stores happened to land first; beside step launches that saturate the L2 / MALL path (the wide
    // kernels of round 5) another block could pass the barrier and read a vector's old contents -- whole runs differed from
    // repeat to repeat (tools/repeat_check.py).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef PTMI_SY_NOBAR
    return;
#endif
    if (threadIdx.x == 0) {
        target += nb;
        // RELAXED: a release / acquire at agent scope writes back / invalidates the XCD's whole L2 at every barrier -- under the step
        // kernel running beside this one (its table rows live there): launches of 3.2 ms instead of 2.5.  The exchanged vectors
        // need neither: they are written and read with agent-scope atomics themselves, and __syncthreads has waited for the stores.
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef PTMI_SY_SLEEP
#define PTMI_SY_SLEEP 8
#endif
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(PTMI_SY_SLEEP);
    }
    __syncthreads();
}
__device__ __forceinline__ double sy_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sy_store(double *p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// elements lo .. n - 1 of an exchanged vector into LDS: every load of a thread in flight at once
__device__ __forceinline__ void sy_fetch(const double *src, double *dst, int lo, int n)
{
    double tmp[SY_PT];
#pragma unroll
    for (int u = 0; u < SY_PT; ++u) { const int i = lo + (int)threadIdx.x + u * SY_THREADS; tmp[u] = i < n ? sy_load(src + i) : 0.0; }
#pragma unroll
    for (int u = 0; u < SY_PT; ++u) { const int i = lo + (int)threadIdx.x + u * SY_THREADS; if (i < n) dst[i] = tmp[u]; }
}
__device__ __forceinline__ double sy_block_sum(double x, double *red)      // red: [SY_NW] doubles of LDS
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < SY_THREADS / 64; ++w) s += red[w];
    return s;
}
// 256 threads and at most 64 registers: a wave per SIMD that fits beside FOUR waves of the config-4 step kernel (112 registers each);
// with 512 threads of 72 registers the step kernel lost a wave per SIMD on every CU that holds a block of this one (launches 3.1 ms
// against 2.5)
__global__ __launch_bounds__(SY_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void sytrd_lds_kernel(SytrdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sy[];
    const int n = a.n, nb = (int)gridDim.x, b = (int)blockIdx.x, t = (int)threadIdx.x;
    const int ncol = b < n ? (n - 1 - b) / nb + 1 : 0;               // owned columns b, b + nb, ...
    double *col = sy;                                                // [ncol][n]
    double *v = sy + (size_t)SY_CMAX * n, *w = v + n, *xr = w + n, *red = xr + n;    // [n] each, [1 + SY_CMAX][SY_NW]
    unsigned target = 0;
    for (int c = 0; c < ncol; ++c)
        for (int i = t; i < n; i += SY_THREADS) col[(size_t)c * n + i] = a.A[(size_t)(b + c * nb) * n + i];
    for (int i = t; i < n; i += SY_THREADS) { v[i] = 0.0; w[i] = 0.0; }
    double tau = 0.0;                                                // of the reflector in v (m - 1)
    __syncthreads();
    // Iteration m: the update by reflector m - 1 (in v, known to every block) and the generation of reflector m, with ONE grid
    // barrier: beside its p_j every block would need column m after the update to form the next reflector -- the column's owner
    // sends it as it is BEFORE the update, and every block applies the update to its copy and forms v_m for itself (the same
    // operations on the same values in every block).
    for (int m = 0; m + 1 < n; ++m) {
        const int par = m & 1;
        double *pb = a.pbuf + (size_t)par * n, *rb = a.vbuf + (size_t)par * (n + 2);
        const int c0 = m <= b ? 0 : (m - b + nb - 1) / nb;             // this block's columns j >= m: those from c0 on
        if (m >= 1 && tau != 0.0) {
            // a wave per column (columns wave, wave + SY_NW, ...: SY_CMAX / SY_NW accumulators per lane, reduced inside the wave,
            // no barrier): with every thread on every column the 16 accumulators' 96 shuffle steps, an LDS exchange between the
            // waves and a barrier made this the longest part of a step (9 of 12.5 us)
            constexpr int CW = SY_CMAX / SY_NW;
            const int wv = t >> 6, ln = t & 63;
            double acc[CW];
#pragma unroll
            for (int q = 0; q < CW; ++q) acc[q] = 0.0;
            for (int i = m + ln; i < n; i += 64) {
                const double vi = v[i];
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    const int c = wv + q * SY_NW;
                    if (c >= c0 && c < ncol) acc[q] = __builtin_fma(col[(size_t)c * n + i], vi, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < CW; ++q) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) acc[q] += __shfl_xor(acc[q], o, 64);
                const int c = wv + q * SY_NW;
                if (ln == 0 && c >= c0 && c < ncol) sy_store(pb + b + c * nb, tau * acc[q]);
            }
        }
        if (b == m % nb) {
            const double *x = col + (size_t)(m / nb) * n;
            for (int i = m + 1 + t; i < n; i += SY_THREADS) sy_store(rb + i, x[i]);
        }
        sy_grid_sync(a.bar, target, (unsigned)nb);
        if (m >= 1 && tau != 0.0) {
            // both vectors' loads in flight at once (a round trip to memory each)
            double tx[SY_PT], tp[SY_PT];
#pragma unroll
            for (int u = 0; u < SY_PT; ++u) {
                const int i = m + t + u * SY_THREADS;
                tx[u] = (i > m && i < n) ? sy_load(rb + i) : 0.0;
                tp[u] = i < n ? sy_load(pb + i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < SY_PT; ++u) {
                const int i = m + t + u * SY_THREADS;
                if (i > m && i < n) xr[i] = tx[u];
                if (i < n) w[i] = tp[u];
            }
            __syncthreads();
            double pv = 0.0;
            for (int i = m + t; i < n; i += SY_THREADS) pv = __builtin_fma(w[i], v[i], pv);
            const double ptv = sy_block_sum(pv, red);
            const double al = -0.5 * tau * ptv;
            for (int i = m + t; i < n; i += SY_THREADS) w[i] = __builtin_fma(al, v[i], w[i]);
            __syncthreads();
            for (int c = c0; c < ncol; ++c) {
                const int j = b + c * nb;
                const double vj = v[j], wj = w[j];
                double *cj = col + (size_t)c * n;
                for (int i = m + t; i < n; i += SY_THREADS) cj[i] -= v[i] * wj + w[i] * vj;
            }
            const double vm = v[m], wm = w[m];
            for (int i = m + 1 + t; i < n; i += SY_THREADS) xr[i] -= v[i] * wm + w[i] * vm;       // column m as its owner now has it
        } else {
            sy_fetch(rb, xr, m + 1, n);
        }
        __syncthreads();
        // reflector m from xr[m + 1 .. n - 1] (dlarfg)
        double ss = 0.0;
        for (int i = m + 2 + t; i < n; i += SY_THREADS) ss = __builtin_fma(xr[i], xr[i], ss);
        const double xn2 = sy_block_sum(ss, red);
        const double alpha = xr[m + 1];
        double beta = alpha, scal = 0.0;
        tau = 0.0;
        if (xn2 != 0.0) {
            const double nrm = det_sqrt(alpha * alpha + xn2);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scal = 1.0 / (alpha - beta);
        }
        __syncthreads();                                              // every thread has read alpha
        for (int i = m + 2 + t; i < n; i += SY_THREADS) v[i] = xr[i] * scal;
        if (t == 0) v[m + 1] = 1.0;
        if (b == m % nb) {
            for (int i = m + 2 + t; i < n; i += SY_THREADS) a.A[(size_t)m * n + i] = xr[i] * scal;      // LAPACK's storage of reflector m
            if (t == 0) {
                a.D[m] = col[(size_t)(m / nb) * n + m];
                a.E[m] = beta;
                a.tau[m] = tau;
            }
        }
        __syncthreads();
    }
    if ((n - 1) % nb == b && t == 0) a.D[n - 1] = col[(size_t)((n - 1) / nb) * n + (n - 1)];
}
// eigenvalues ascending (the library's order) -> by decreasing size in absolute value, the eigenvectors (rows of C) along
__global__ __launch_bounds__(256) void eig_sort_rows_kernel(const double *D, const double *Cm, int n, double *Ut, double *S)
{
    __shared__ int rank_s;
    const int k = (int)blockIdx.x;
    if (threadIdx.x == 0) {
        const double mine = __builtin_fabs(D[k]);
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double o = __builtin_fabs(D[j]); rank += (o > mine) || (o == mine && j > k); }
        rank_s = rank;
        S[rank] = mine;
    }
    __syncthreads();
    const int rank = rank_s;
    for (int i = (int)threadIdx.x; i < n; i += 256) Ut[(size_t)rank * n + i] = Cm[(size_t)k * n + i];
}
#include "ptmi_dc.inc.h"

// host side of the divide-and-conquer solver: the tree of a matrix order (all leaves at one depth, so that every level merges every
// block and the two vector buffers alternate), built once per engine
struct DcPlan {
    int n = 0, nlevels = 0, nleaves = 0, nnodes = 0;
    std::vector<int> lvl_off, lvl_cnt, lvl_nmax;       // per level (bottom-up): first node, nodes, largest node
    dc::Node *d_nodes = nullptr;                       // all merges, level by level
    dc::Leaf *d_leaves = nullptr;
    char *scr = nullptr;                               // two vector buffers, U, the per-row arrays
};
static int dc_plan_get(ptmi_engine *h, int n, DcPlan **out)
{
    if (h->dc_plan) { *out = (DcPlan *)h->dc_plan; return PTMI_OK; }
    DcPlan *P = new (std::nothrow) DcPlan();
    if (!P) return fail(PTMI_EHIP, "out of memory");
    int depth = 0;
    while (((n + (1 << depth) - 1) >> depth) > dc::LEAF) ++depth;
    std::vector<std::vector<dc::Node>> by_depth(depth);
    std::vector<dc::Leaf> leaves;
    struct Rec { static void go(int off, int nn, int dep, int depth, std::vector<std::vector<dc::Node>> &bd, std::vector<dc::Leaf> &lv) {
        if (dep == depth) { lv.push_back({off, nn}); return; }
        const int n1 = nn / 2;
        bd[dep].push_back({off, nn, n1});
        go(off, n1, dep + 1, depth, bd, lv);
        go(off + n1, nn - n1, dep + 1, depth, bd, lv);
    } };
    Rec::go(0, n, 0, depth, by_depth, leaves);
    std::vector<dc::Node> all;
    for (int dep = depth - 1; dep >= 0; --dep) {       // bottom-up
        P->lvl_off.push_back((int)all.size());
        P->lvl_cnt.push_back((int)by_depth[dep].size());
        int mx = 0;
        for (const dc::Node &nd : by_depth[dep]) { all.push_back(nd); mx = nd.n > mx ? nd.n : mx; }
        P->lvl_nmax.push_back(mx);
    }
    P->n = n; P->nlevels = depth; P->nleaves = (int)leaves.size(); P->nnodes = (int)all.size();
    const size_t nn = (size_t)n * n;
    const size_t bytes = sizeof(double) * (3 * nn + 12 * (size_t)n + 2 * all.size() + 64) + sizeof(int) * (6 * (size_t)n + 2 * all.size() + 64);
    hipError_t e = hipMalloc((void **)&P->scr, bytes);
    if (e == hipSuccess && !all.empty()) e = hipMalloc((void **)&P->d_nodes, sizeof(dc::Node) * all.size());
    if (e == hipSuccess) e = hipMalloc((void **)&P->d_leaves, sizeof(dc::Leaf) * leaves.size());
    if (e == hipSuccess && !all.empty()) e = hipMemcpy(P->d_nodes, all.data(), sizeof(dc::Node) * all.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(P->d_leaves, leaves.data(), sizeof(dc::Leaf) * leaves.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(P->scr); (void)hipFree(P->d_nodes); (void)hipFree(P->d_leaves);
        delete P;
        return fail(PTMI_EHIP, "divide-and-conquer scratch: %s", hipGetErrorString(e));
    }
    h->dc_plan = P;
    *out = P;
    return PTMI_OK;
}
static void dc_plan_free(ptmi_engine *h)
{
    DcPlan *P = (DcPlan *)h->dc_plan;
    if (!P) return;
    (void)hipFree(P->scr); (void)hipFree(P->d_nodes); (void)hipFree(P->d_leaves);
    delete P;
    h->dc_plan = nullptr;
}
// eigenvalues (ascending, *Dres) and eigenvectors (vector-major, *Zres) of the tridiagonal matrix (D, E), back-transformed through the
// reflectors (A, tau) of the reduction; everything queued on st
static int dc_solve(ptmi_engine *h, hipStream_t st, int n, const double *D, const double *E, const double *A, const double *tau,
                    const double **Dres, const double **Zres, int *info /* device: zeroed by the caller; a leaf that did not converge sets it */)
{
    DcPlan *P = nullptr;
    if (int rc = dc_plan_get(h, n, &P)) return rc;
    const size_t nn = (size_t)n * n;
    double *p = (double *)P->scr;
    double *Qa = p; p += nn;
    double *Qb = p; p += nn;
    dc::Args a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.U = p; p += nn;
    a.d = p; p += n;
    a.e = p; p += n;
    double *Da = p; p += n;
    double *Db = p; p += n;
    a.dk = p; p += n; a.zk = p; p += n; a.Ddefl = p; p += n; a.mu = p; p += n; a.lam = p; p += n; a.zh = p; p += n;
    a.rho = p; p += P->nnodes + 8;
    int *q = (int *)p;
    a.keepv = q; q += n; a.deflv = q; q += n; a.org = q; q += n; a.rankk = q; q += n; a.rankd = q; q += n;
    a.cnt = q;
    HIPCHK(hipMemcpyAsync(a.d, D, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(a.e, E, sizeof(double) * (n - 1), hipMemcpyDeviceToDevice, st));
    if (P->nnodes) hipLaunchKernelGGL(dc::split_kernel, dim3((P->nnodes + 63) / 64), dim3(64), 0, st, (const dc::Node *)P->d_nodes, P->nnodes, a.d, (const double *)a.e);
    HIPCHK(hipMemsetAsync(Qa, 0, sizeof(double) * nn, st));
    hipLaunchKernelGGL(dc::leaf_kernel, dim3(P->nleaves), dim3(64), 0, st, (const dc::Leaf *)P->d_leaves, n, (const double *)a.d, (const double *)a.e, Da, Qa, info);
    double *Qin = Qa, *Qout = Qb, *Din = Da, *Dout = Db;
    for (int lv = 0; lv < P->nlevels; ++lv) {
        const int cnt = P->lvl_cnt[lv], nmax = P->lvl_nmax[lv];
        a.nodes = P->d_nodes + P->lvl_off[lv];
        a.Qin = Qin; a.Qout = Qout; a.Din = Din; a.Dout = Dout;
        HIPCHK(hipMemsetAsync(Qout, 0, sizeof(double) * nn, st));
        hipLaunchKernelGGL(dc::prep_kernel, dim3(cnt), dim3(dc::PREP_THREADS), 0, st, a);
        hipLaunchKernelGGL(dc::secular_kernel, dim3((nmax + 3) / 4, cnt), dim3(256), 0, st, a);
        hipLaunchKernelGGL(dc::zhat_kernel, dim3((nmax + 3) / 4, cnt), dim3(256), 0, st, a);
        hipLaunchKernelGGL(dc::vectors_kernel, dim3((nmax + 3) / 4, cnt), dim3(256), 0, st, a);
        hipLaunchKernelGGL(dc::rank_kernel, dim3(cnt), dim3(1024), 0, st, a);
        hipLaunchKernelGGL(dc::gemm_kernel, dim3((nmax + 63) / 64, (nmax + 63) / 64, cnt), dim3(256), 0, st, a);
        hipLaunchKernelGGL(dc::copy_deflated_kernel, dim3(nmax, cnt), dim3(256), 0, st, a);
        double *tq = Qin; Qin = Qout; Qout = tq;
        double *td = Din; Din = Dout; Dout = td;
    }
    hipLaunchKernelGGL(dc::backtransform_kernel, dim3((n + 4 * dc::VPW - 1) / (4 * dc::VPW)), dim3(256), 0, st, A, tau, n, Qin);
    HIPCHK(hipGetLastError());
    *Dres = Din;
    *Zres = Qin;
    return PTMI_OK;
}

// the library's entry points, looked up in the copies the process has loaded already (torch brings its own librocsolver / librocblas;
// a second copy from /opt/rocm beside them is not wanted)
struct SyLib {
    void *blas_handle;                     // first two members: ptmi_destroy releases the handle through them
    int (*destroy_handle)(void *);
    int (*create_handle)(void **);
    int (*set_stream)(void *, hipStream_t);
    int (*dstedc)(void *, int, int, double *, double *, double *, int, int *);
    int (*dormtr)(void *, int, int, int, int, int, double *, int, double *, double *, int);
};
static int sy_lib_get(ptmi_engine *h, SyLib **out)
{
    if (h->sy_lib) { *out = (SyLib *)h->sy_lib; return PTMI_OK; }
    void *sol = nullptr, *bla = nullptr;
    for (const char *nm : {"librocsolver.so.0", "librocsolver.so"}) if (!sol) sol = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : {"librocblas.so.5", "librocblas.so.4", "librocblas.so"}) if (!bla) bla = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (!sol) sol = dlopen("librocsolver.so.0", RTLD_NOW);
    if (!bla) bla = dlopen("librocblas.so.5", RTLD_NOW);
    if (!sol || !bla) return fail(PTMI_EUNSUPPORTED, "ptmi_eig_sytrd needs the ROCm libraries librocsolver / librocblas in the process (import torch first): %s", dlerror());
    SyLib *L = (SyLib *)calloc(1, sizeof(SyLib));
    if (!L) return fail(PTMI_EHIP, "out of memory");
    L->destroy_handle = (int (*)(void *))dlsym(bla, "rocblas_destroy_handle");
    L->create_handle = (int (*)(void **))dlsym(bla, "rocblas_create_handle");
    L->set_stream = (int (*)(void *, hipStream_t))dlsym(bla, "rocblas_set_stream");
    L->dstedc = (int (*)(void *, int, int, double *, double *, double *, int, int *))dlsym(sol, "rocsolver_dstedc");
    L->dormtr = (int (*)(void *, int, int, int, int, int, double *, int, double *, double *, int))dlsym(sol, "rocsolver_dormtr");
    if (!L->create_handle || !L->set_stream || !L->dstedc || !L->dormtr) { free(L); return fail(PTMI_EUNSUPPORTED, "rocsolver_dstedc / rocsolver_dormtr not found"); }
    if (L->create_handle(&L->blas_handle) != 0) { free(L); return fail(PTMI_EHIP, "rocblas_create_handle failed"); }
    h->sy_lib = L;
    *out = L;
    return PTMI_OK;
}

int ptmi_eig_sytrd(ptmi_handle h, void *stream, double *Ut_out, double *S_out) { return ptmi_eig_sytrd_from(h, stream, nullptr, Ut_out, S_out); }

int ptmi_eig_sytrd_from(ptmi_handle h, void *stream, const double *cov_in, double *Ut_out, double *S_out)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (!cov_in) cov_in = h->buf.cov;
    if (!cov_in) return fail(PTMI_EINVAL, "cov buffer missing");
    if (c.cov_per_walker || c.ngroups > 1) return fail(PTMI_EUNSUPPORTED, "ptmi_eig_sytrd factorizes ONE pooled covariance (no parameter groups)");
    const int n = c.ndim;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    double *Uo = Ut_out ? Ut_out : h->buf.Ut, *So = S_out ? S_out : h->buf.S;
    if (!Uo || !So) return fail(PTMI_EINVAL, "Ut / S buffers missing");
    int dev = 0, ncu = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    int nb = ncu / 4 > 0 ? ncu / 4 : 1;                             // 64 blocks: the barrier's cost grows with them (21.1 ms at 64, 23.1 at 128, 27.8 at 256)
    while ((n + nb - 1) / nb > SY_CMAX && nb < ncu) nb *= 2;
    if (const char *e = getenv("PTMI_SYTRD_BLOCKS")) nb = atoi(e);
    if (nb > ncu) nb = ncu;
    if (nb > n) nb = n;
    if (n < 3 || (n + nb - 1) / nb > SY_CMAX) return fail(PTMI_EUNSUPPORTED, "ptmi_eig_sytrd: 3 <= ndim <= %d on this device", SY_CMAX * nb);
    int cpb = (n + nb - 1) / nb;                                    // columns per block
    const size_t lds = sizeof(double) * ((size_t)(SY_CMAX + 3) * n + (size_t)(1 + SY_CMAX) * SY_NW);
    if (lds > 160 * 1024 || n > 1024) return fail(PTMI_EUNSUPPORTED, "ptmi_eig_sytrd: ndim = %d does not fit the LDS", n);
    (void)cpb;
    const size_t nn = (size_t)n * n;
    if (!h->d_sy_scr) HIPCHK(hipMalloc(&h->d_sy_scr, sizeof(double) * (2 * nn + 8 * (size_t)n + 64) + 256));
    double *A = (double *)h->d_sy_scr, *Cm = A + nn, *D = Cm + nn, *E = D + n, *tau = E + n, *vbuf = tau + n, *pbuf = vbuf + 2 * (n + 2);
    unsigned *bar = (unsigned *)(pbuf + 2 * n + 2);
    int *info = (int *)(bar + 4);
    HIPCHK(hipMemcpyAsync(A, cov_in, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(bar, 0, 32, st));
    HIPCHK(hipFuncSetAttribute((const void *)sytrd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SytrdArgs sa = {A, D, E, tau, vbuf, pbuf, bar, n};
    // The kernel's grid barrier needs all nb blocks resident at once.  One block per CU always fits an otherwise free CU (checked
    // here against the occupancy the runtime computes); beside persistent step kernels the blocks take the CUs' remaining LDS as it
    // is (the step kernel leaves 78 KB, a block needs up to 160: such a block starts when its CU's step block ends, and every step
    // block ends).  What could deadlock is a SECOND factorization of another engine on the same device holding part of the CUs with
    // blocks that spin: factorizations of one device are therefore serialized by an event chain across engines and streams.
    int occ = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)sytrd_lds_kernel, SY_THREADS, lds));
    if (occ < 1 || (long long)occ * ncu < nb)
        return fail(PTMI_EUNSUPPORTED, "ptmi_eig_sytrd: %d blocks of %zu B of LDS cannot be resident at once on %d CUs", nb, lds, ncu);
    {
        static std::mutex mu;
        static hipEvent_t last[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        const int di = dev & 63;
        if (last[di]) HIPCHK(hipStreamWaitEvent(st, last[di], 0));
        else HIPCHK(hipEventCreateWithFlags(&last[di], hipEventDisableTiming));
        hipLaunchKernelGGL(sytrd_lds_kernel, dim3(nb), dim3(SY_THREADS), lds, st, sa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(last[di], st));
    }
    static const bool use_lib = getenv("PTMI_SYTRD_LIB") != nullptr;   // measurement switch: round 4's path, the library's divide-and-conquer solver and back-transformation
    if (!use_lib) {
        // the tridiagonal matrix's eigenvectors by the engine's own divide-and-conquer kernels, back-transformed through the reflectors
        const double *Dres = nullptr, *Zres = nullptr;
        if (int rc = dc_solve(h, st, n, D, E, A, tau, &Dres, &Zres, info)) return rc;
        hipLaunchKernelGGL(eig_sort_rows_kernel, dim3(n), dim3(256), 0, st, Dres, Zres, n, Uo, So);
        HIPCHK(hipGetLastError());
        // the convergence word (a leaf's QL iteration: dc::leaf_kernel) follows the result to the host on the same stream
        if (!h->h_sy_info) {
            HIPCHK(hipHostMalloc((void **)&h->h_sy_info, 2 * sizeof(int32_t)));
            h->h_sy_info[0] = 0; h->h_sy_info[1] = 0;
        }
        HIPCHK(hipMemcpyAsync(h->h_sy_info, info, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        return PTMI_OK;
    }
    SyLib *L = nullptr;
    if (int rc = sy_lib_get(h, &L)) return rc;
    if (L->set_stream(L->blas_handle, st) != 0) return fail(PTMI_EHIP, "rocblas_set_stream failed");
    // eigenvectors of the tridiagonal matrix (columns of C), then C := Q C with the reflectors of the reduction
    int rs = L->dstedc(L->blas_handle, (int)rocblas_evect_tridiagonal, n, D, E, Cm, n, info);
    if (rs != 0) return fail(PTMI_EHIP, "rocsolver_dstedc: status %d", rs);
    // the solver's convergence word follows the result to the host on the same stream (ptmi_eig_sytrd_info reads the last one that arrived)
    if (!h->h_sy_info) {
        HIPCHK(hipHostMalloc((void **)&h->h_sy_info, 2 * sizeof(int32_t)));
        h->h_sy_info[0] = 0; h->h_sy_info[1] = 0;
    }
    HIPCHK(hipMemcpyAsync(h->h_sy_info, info, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    rs = L->dormtr(L->blas_handle, (int)rocblas_side_left, (int)rocblas_fill_lower, (int)rocblas_operation_none, n, n, A, n, tau, Cm, n);
    if (rs != 0) return fail(PTMI_EHIP, "rocsolver_dormtr: status %d", rs);
    hipLaunchKernelGGL(eig_sort_rows_kernel, dim3(n), dim3(256), 0, st, (const double *)D, (const double *)Cm, n, Uo, So);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_eig_sytrd_info(ptmi_handle h, int32_t *info)
{
    if (!h || !info) return fail(PTMI_EINVAL, "NULL argument");
    *info = h->h_sy_info ? h->h_sy_info[0] : 0;
    return PTMI_OK;
}

int ptmi_am_expand(ptmi_handle h, int32_t w0, int32_t nw, int64_t iter_lo, int64_t iter_hi)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (!h->buf.AMflag || !h->buf.AM) return PTMI_OK;                    // every row is stored already
    if (w0 < 0 || nw < 0 || w0 + nw > c.nwalkers) return fail(PTMI_EINVAL, "walkers [%d, %d) of %d", w0, w0 + nw, c.nwalkers);
    const long long base = iter_hi > 0 ? ((long long)(iter_hi - 1) / c.cov_update) * c.cov_update : 0;
    if (iter_lo < base || iter_hi < iter_lo || iter_hi - iter_lo >= c.cov_update)
        return fail(PTMI_EINVAL, "iterations %lld..%lld are not inside the covariance period that starts at %lld (with AM row flags the ring keeps "
                                 "the rows of the current period only)", (long long)iter_lo, (long long)iter_hi, base);
    if (nw == 0) return PTMI_OK;
    hipLaunchKernelGGL(am_expand_kernel, dim3((unsigned)nw), dim3(128), 0, h->stream, h->buf.AM, (const AmFlag *)h->buf.AMflag, c.ndim, c.cov_update,
                       (int)w0, (long long)iter_lo, (long long)iter_hi, base);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_test_replay(ptmi_handle h, const double *swap_uniforms, const uint64_t *draws)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    h->rp_swap_u = swap_uniforms;
    h->rp_draws = (const u64 *)draws;
    return PTMI_OK;
}

int ptmi_update_de(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (!h->buf.DE) return PTMI_OK;
    if (!h->buf.AM) return fail(PTMI_EINVAL, "DE update needs the AM buffer on this GPU");
    const int wc = c.cov_per_walker ? c.nwalkers : 1;
    hipLaunchKernelGGL(de_update_kernel, dim3(c.cov_update, wc), dim3(64), 0, h->stream, h->buf.DE, (const double *)h->buf.AM,
                       c.ndim, c.de_size, c.cov_update, h->de_head, c.nwalkers, c.cov_per_walker ? 0 : 1,
                       h->G == 4 ? 8 * ((h->EPL + 1) / 2) : c.ndim, h->G == 4 ? h->EPL : 0, am_row_epl(h->G, h->EPL));
    HIPCHK(hipGetLastError());
    const int adv = c.cov_update < c.de_size ? c.cov_update : c.de_size;
    h->de_head = (h->de_head + adv) % c.de_size;
    return PTMI_OK;
}

int ptmi_set_de_head(ptmi_handle h, int32_t head)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (head < 0 || head >= h->cfg.de_size) return fail(PTMI_EINVAL, "head %d outside the ring of %d rows", head, h->cfg.de_size);
    h->de_head = head;
    return PTMI_OK;
}

int ptmi_selftest_math(int device, int op, const double *in, const double *in2, double *out, int64_t n)
{
    if (!in || !out || n < 0) return fail(PTMI_EINVAL, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PTMI_ENODEVICE, "no HIP device visible");
    HIPCHK(hipSetDevice(device));
    double *di = nullptr, *di2 = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **)&di, sizeof(double) * (size_t)n));
    HIPCHK(hipMalloc((void **)&di2, sizeof(double) * (size_t)n));
    HIPCHK(hipMalloc((void **)&dout, sizeof(double) * (size_t)n));
    HIPCHK(hipMemcpy(di, in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(di2, in2 ? in2 : in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, (const double *)di,
                       (const double *)di2, dout, (long long)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(di); (void)hipFree(di2); (void)hipFree(dout);
    return PTMI_OK;
}

int ptmi_selftest_philox(int device, const uint32_t *ck, uint32_t *out, int64_t n)
{
    if (!ck || !out || n < 0) return fail(PTMI_EINVAL, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PTMI_ENODEVICE, "no HIP device visible");
    HIPCHK(hipSetDevice(device));
    u32 *dc = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **)&dc, sizeof(u32) * 6 * (size_t)n));
    HIPCHK(hipMalloc((void **)&dout, sizeof(u32) * 4 * (size_t)n));
    HIPCHK(hipMemcpy(dc, ck, sizeof(u32) * 6 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const u32 *)dc, dout, (long long)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, sizeof(u32) * 4 * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(dc); (void)hipFree(dout);
    return PTMI_OK;
}

int ptmi_malloc(void **p, size_t bytes)
{
    if (!p) return fail(PTMI_EINVAL, "NULL argument");
    HIPCHK(hipMalloc(p, bytes));
    return PTMI_OK;
}
int ptmi_free(void *p) { HIPCHK(hipFree(p)); return PTMI_OK; }
int ptmi_memcpy_h2d(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return PTMI_OK; }
int ptmi_memcpy_d2h(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return PTMI_OK; }
int ptmi_memset(void *dst, int value, size_t bytes) { HIPCHK(hipMemset(dst, value, bytes)); return PTMI_OK; }

int ptmi_timer_start(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    return PTMI_OK;
}
int ptmi_timer_stop_ms(ptmi_handle h, double *ms)
{
    if (!h || !ms) return fail(PTMI_EINVAL, "NULL argument");
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, h->ev0, h->ev1));
    *ms = (double)f;
    return PTMI_OK;
}

}  // extern "C"
