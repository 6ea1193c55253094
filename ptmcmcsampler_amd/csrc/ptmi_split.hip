// ptmi_split.hip -- the split path (ptmi_propose / ptmi_accept / ptmi_accept_propose) for likelihoods that live OUTSIDE the
// library: a batched callback on the device tensor of proposals (PT:605-611 through the reference's _function_wrapper,
// PT:1072-1086).  PT:<lines> = PTMCMCSampler/PTMCMCSampler.py of nanograv/PTMCMCSampler.
//
// This is the one path of the library that is bound by HBM: every iteration the state X [chains][ndim] comes in and the proposals Q
// go out (and back in for the callback).  The shape kernels' layout (4 / 16 / 64 lanes per chain, lane l holding elements l + G e:
// made for K fused steps on registers) reads a row as 25 separate 8-byte pieces per lane, 800 bytes apart across a wave -- 0.107
// of the HBM roofline (round 5).  Here the rows are contiguous:
//
//   * a block takes a TILE of 64 consecutive chains = one contiguous span of X and of Q;
//   * two threads per chain do everything that is a scalar of the chain -- the accept test of iteration `it` (PT:615-622) from the
//     callback's values, then the draws of the next proposal: cycle pick, scale branch, parameter group, SCAM direction and
//     amplitude, DE's two history rows and scale, an AM pick's cursor (PT:1048-1067, 820-876, 936-985) -- the same operations on the same Philox
//     words as propose() of ptmi_mh.inc.h (bit-identical; the oracle is mh_one of oracle/ptmcmc_oracle.c), and leave a 56-byte record in LDS;
//   * all 256 threads then walk the tile's span in 16-byte pieces (dwordx4 loads and stores, a wave instruction = 1 KB of
//     consecutive addresses): new state = the accepted proposal or the old row (written to X only where the rule below says), next
//     proposal = state + increment written to the proposal buffer, the loads of four pieces in flight before the first is used.
//
// ACC and PROP in ONE launch (ptmi_accept_propose: accept of iteration it, proposal of it + 1) moves a row in once and out once per
// iteration where accept + propose as two launches moved it twice: per update 8 d (state or proposal in) + 8 d (proposal out)
// + 8 d x acceptance (state out) + 64 B of scalars, and the callback's own read of Q.
//
// TWO proposal buffers (ptmi_buffers.Q2 + sloc) take the "state out" away as well: an accepted proposal already IS the chain's new
// state, so between ptmi_propose and ptmi_accept a chain's state stays where it was written -- X, or the buffer its accepted
// proposal sits in (sloc[chain] says which) -- the next proposals go to the OTHER buffer, and a row is copied to X only when the
// buffer it lives in is about to be overwritten (the chain was accepted one iteration ago and refused now: acceptance x
// (1 - acceptance) of the rows instead of acceptance) or when the segment ends (ptmi_accept: every state back in X).  At the 91 %
// acceptance of the first thousand iterations from p0 = 0 that is 8 % of the rows instead of 91 %.  With one buffer (Q2 NULL) the
// same rule makes every accepted row go to X at once: the single-buffer case is the one where the target IS the current buffer.
//
// AM picks (PT:879-933, 2 d^2 flop each) get their increments from the matrix cores AHEAD of the launch (am_gemm_kernel of ptmi_abi.hip:
// ptmi_split_am_prepare lists the picks of a piece of iterations and multiplies; an increment depends on the chain's stream, the
// iteration and the scale branch, not on its state) and the row kernel adds the chain's next one as it adds a SCAM direction.
// Not here: host-served cycle entries together with AM entries, and handles without room for the increments -- the shape kernels'
// propose_kernel / accept_kernel keep serving those; ptmi_split_rows_ok says whether a handle's configuration runs here.
#include "ptmi_mh.inc.h"

namespace {

#ifndef PTMI_SPLIT_UNR
#define PTMI_SPLIT_UNR 4             // (measured: 2 and 4 the same, 8 costs 5 %: registers; tiles of 32 / 128 chains 3 % behind 64)
#endif
#ifndef PTMI_SPLIT_NT
#define PTMI_SPLIT_NT 1             // the state rows are read with non-temporal loads: this is their last use, and what they would push out of the
#endif                              // memory-side cache are the proposals just written, which the callback reads next (1.65e9 -> 1.85e9 updates/s)
#ifndef PTMI_SPLIT_NTX
#define PTMI_SPLIT_NTX 1            // rows copied to X are stored non-temporally: nothing reads them before the segment ends (1.86e9 -> 1.89e9)
#endif
#ifndef PTMI_SPLIT_NTQ
#define PTMI_SPLIT_NTQ 0            // 1: the proposals are stored non-temporally (measured: no difference, 1.855e9 either way)
#endif
#ifndef PTMI_SPLIT_TILE
#define PTMI_SPLIT_TILE 64
#endif
constexpr int TILE = PTMI_SPLIT_TILE;   // chains per block (at most 128: two threads of the block's 256 per chain)
constexpr int UNR = PTMI_SPLIT_UNR; // pieces in flight per thread
template <class T>
__device__ __forceinline__ T row_load(const T *p)
{
#if PTMI_SPLIT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct Rec {                        // what the row pass needs of a chain
    int flags;                      // bit 0: the state's row goes to X; bit 1: the chain's AM row is to be written (cold rank, not a swap iteration)
    int src;                        // where the chain's state (after the accept test) is: 0 = X, 1 + b = proposal buffer b
    int jt;                         // the next proposal's type (PTMI_J_*; anything else: the state is handed back unchanged)
    long long urow;                 // SCAM: element offset of the direction's row in Ut
    long long rm, rn;               // DE: element offsets of the two history rows in DE
    long long gm;                   // DE with parameter groups: element offset of the group's mask in gmask, else -1
    double amp;                     // SCAM: z cd sqrt(S_k) (PT:873); DE: the scale (PT:969-976)
};

template <int VEC> struct Piece;
template <> struct Piece<2> { typedef ptmi_d2 T; };
template <> struct Piece<1> { typedef double T; };

// position of parameter i inside a row of the DE history (ptmi_de_row_stride)
__device__ __forceinline__ int de_pos(int i, int lanes)
{
    if (lanes != 4) return i;
    const int e = i >> 2, ln = i & 3;
    return 8 * (e >> 1) + 2 * ln + (e & 1);
}

template <bool ACC, bool PROP, int VEC>
__global__ __launch_bounds__(256) void split_rows_kernel(const KArgs a)
{
    __shared__ Rec rec[TILE];
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    const long long c0 = (long long)blockIdx.x * TILE;
    const int ntile = (int)(nch - c0 < TILE ? nch - c0 : TILE);
    const int tid = (int)threadIdx.x;
    // the iteration: the launch's argument, or (launches captured in a hipGraph, replayed for other iterations: ptmi_device_iter) an
    // offset from a counter in device memory -- then the ring row and the swap-iteration test follow from it here, as set_step_args
    // derives them on the host
    const long long iter0 = a.iter_dev ? *a.iter_dev + a.iter0 : a.iter0;
    const int am_row0 = a.iter_dev ? (int)(iter0 % a.cov_update) : a.am_row0;
    const bool swap_last = a.iter_dev ? (a.tskip > 0 && a.ntg > 1 && iter0 % a.tskip == 0) : a.swap_last != 0;

    // ---------------------------------------------------------------- the chains' scalars: threads 2c (slot 0) and 2c + 1 (slot 1)
    if (tid < 2 * TILE) {
        const int cl = tid >> 1, slot = tid & 1;
        const bool live = cl < ntile;
        const long long ch = live ? c0 + cl : c0;
        const int w = (int)(ch / nt);
        const int t = a.temp_of[ch];
        const double beta = a.beta[t];
        int flags = 0, loc = 0;
        if (slot == 0 && live && a.sloc != nullptr) loc = a.sloc[ch];
        if (ACC && slot == 0 && live) {
            // PT:605-622 with the callback's values (accept_kernel of ptmi_mh.inc.h)
            const double nlp = a.newlp[ch], nlnL = a.newlnL[ch];
            const double lnL0 = a.lnL[ch], lp0 = a.lp[ch];
            const double nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
            const double lnprob0 = beta * lnL0 + lp0;
            const ptmi_d2 q01 = *reinterpret_cast<const ptmi_d2 *>(a.qaux + ch * 4), q23 = *reinterpret_cast<const ptmi_d2 *>(a.qaux + ch * 4 + 2);
            const double diff = nlnprob - lnprob0 + q01.x;
            const int jt = (int)q01.y;
            const bool acc = diff > q23.y;
            const bool cold = a.temp0 + t == 0 && a.AM != nullptr;
            const bool am = cold && !swap_last;
            const size_t r = (size_t)w * nt + t;
            if (jt >= 0 && jt < PTMI_J_NTYPES) a.jstat[(r * PTMI_J_NTYPES + jt) * 2 + 0] += 1;
            if (am && a.AMflag) a.AMflag[(size_t)w * a.cov_update + (size_t)am_row0] = AMROW_KEY | (acc ? AMROW_NEW : 0ull);   // the split path stores every row
            if (am && a.AMaux) {
                double *ax = a.AMaux + ((size_t)w * a.cov_update + (size_t)am_row0) * 2;
                ax[0] = acc ? nlnL : lnL0;
                ax[1] = acc ? nlp : lp0;
            }
            if (acc) {
                a.lnL[ch] = nlnL;
                a.lp[ch] = nlp;
                a.nacc[r] += 1;
                if (jt >= 0 && jt < PTMI_J_NTYPES) a.jstat[(r * PTMI_J_NTYPES + jt) * 2 + 1] += 1;
            }
            if (!PROP) a.qaux[ch * 4 + 2] = acc ? 1.0 : 0.0;     // the decision, for the host's per-name jump statistics
            if (acc) loc = 1 + a.q_cur;                          // the accepted proposal IS the new state, where it sits
            flags = am ? 2 : 0;
        }
        if (slot == 0 && live) {
            // the state's row goes to X when the buffer it lives in is the one the next proposals overwrite, or when no proposal follows
            const int src = loc;
            if (PROP ? loc == 1 + a.q_tgt : loc != 0) { flags |= 1; loc = 0; }
            if (a.sloc != nullptr) a.sloc[ch] = loc;
            rec[cl].src = src;
        }
        if (PROP) {
            const long long it = iter0 + (ACC ? 1 : 0);
            const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
            const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);
            const u32 sid = sid0 + (u32)(a.temp0 + t);
            // this thread's Philox call: slot 0 = (P0 | accept uniform), slot 1 = (Q0, Q1)
            u64 w0, w1;
            philox_words(a.seed, (u64)it, sid, (u32)slot, w0, w1);
            const double lg = unit_log<0>(slot ? w0 : w1);       // slot 0: log of the accept uniform; slot 1: Box-Muller radius
            u32 aj;
            double at, sn, cs;
            unit_angle32((u32)w1, aj, at);
            unit_sincos<0>(aj, at, sn, cs);
            double z = det_sqrt(-2.0 * lg) * cs;                 // the SCAM normal (slot 1)
            u64 P0 = (u64)__shfl((long long)w0, (tid & 63) & ~1, 64);     // the pair's slot-0 word
            u64 Q0 = w0, Q1 = w1;
            u32 pickw = (u32)(P0 >> 32);
            if (a.pick_walker) {
                u64 p0, p1;
                philox_words(a.seed, (u64)it, sid0, 0u, p0, p1);
                pickw = (u32)(p0 >> 32);
            }
            double rp_val = 0.0;
            const bool replay = a.rp_draws != nullptr;
            if (replay) {                                        // TEST HOOK (ptmi_test_replay): the proposal's draws as recorded from the reference
                const u64 *r = a.rp_draws + (size_t)ch * 4;
                P0 = r[0]; Q0 = r[1]; Q1 = r[2];
                rp_val = __longlong_as_double((long long)r[3]);
                z = rp_val;
                pickw = (u32)(P0 >> 32);
            }
            // cycle pick (PT:1058) and scale branch (PT:846-858), as propose()
            const int w_de = a.de_on ? a.w_de : 0;
            const int L = a.w_host + a.w_scam + a.w_am + w_de;
            const int pick = (int)h2index(pickw, (u32)L);
            const int ind = pick - a.w_host;
            int jt = ind < a.w_scam ? PTMI_J_SCAM : (ind < a.w_scam + a.w_am ? PTMI_J_AM : PTMI_J_DE);
            if (ind < 0) jt = PTMI_J_NTYPES + pick;              // a host-served cycle entry: the state is handed back unchanged
            const u32 plo = (u32)P0;
            constexpr u32 T97 = (u32)(0.97 * 4294967296.0), T90 = (u32)(0.9 * 4294967296.0), T50 = 0x80000000u;
            const int br = plo > T97 ? 0 : (plo > T90 ? 1 : 2);
            int g = 0, ng = d;
            if (a.ngroups > 1) {                                 // PT:839,897,955: its own Philox call
                u64 g0, g1;
                philox_words(a.seed, (u64)it, sid, 2u, g0, g1);
                g = (int)h2index((u32)(g0 >> 32), (u32)a.ngroups);
                ng = a.gsize[g];
            }
            if (slot == 1 && live) {
                const size_t wc = a.per_walker ? (size_t)w : 0;
                Rec r;
                r.jt = jt;
                r.urow = 0; r.rm = 0; r.rn = 0; r.gm = -1; r.amp = 0.0;
                if (jt == PTMI_J_SCAM) {
                    const int k = (int)h2index((u32)(Q1 >> 32), (u32)ng);
                    const double *S = a.S + (wc * a.ngroups + (size_t)g) * d;
                    r.amp = z * cc.cd_scam(br) * det_sqrt(S[k]);                       // PT:873
                    r.urow = (long long)(((wc * a.ngroups + (size_t)g) * d + (size_t)k) * d);
                } else if (jt == PTMI_J_AM) {
                    // PT:879-933: the increment U (cd sqrt(S) z) was made on the matrix cores ahead of this launch (am_gemm_kernel, the
                    // same k-ascending fma chain as the step kernels' own product); this pick takes the chain's next one
                    const long long at = a.am_next[ch];
                    a.am_next[ch] = at + 1;
                    r.urow = at * d;
                } else if (jt == PTMI_J_DE) {
                    const u32 Bn = (u32)a.de_size;
                    const u32 mm = h2index((u32)(Q0 >> 32), Bn);
                    const u32 nn = (mm + 1u + h2index((u32)Q0, Bn - 1u)) % Bn;
                    double scale;
                    if (plo > T50) scale = 1.0;
                    else scale = (replay ? rp_val : w2uniform(Q1)) * 2.4 / a.gdiv[g] * cc.de_mul;      // PT:976
                    r.amp = scale;
                    const long long base = (long long)(wc * (size_t)a.de_size * a.de_ld);
                    r.rm = base + (long long)((mm + (u32)a.de_head) % Bn) * a.de_ld;
                    r.rn = base + (long long)((nn + (u32)a.de_head) % Bn) * a.de_ld;
                    if (a.ngroups > 1) r.gm = (long long)g * d;
                }
                rec[cl].jt = r.jt; rec[cl].urow = r.urow; rec[cl].rm = r.rm; rec[cl].rn = r.rn; rec[cl].gm = r.gm; rec[cl].amp = r.amp;
            }
            if (slot == 0 && live) {
                double *qa = a.qaux + ch * 4;
                *reinterpret_cast<ptmi_d2 *>(qa) = ptmi_d2{0.0 /* qxy of the built-in jumps (PT:836,894,952) */, (double)jt};
                *reinterpret_cast<ptmi_d2 *>(qa + 2) = ptmi_d2{w2uniform_open(w1), lg};
            }
        }
        if (slot == 0 && live) rec[cl].flags = flags;
    }
    __syncthreads();

    // ---------------------------------------------------------------------------------- the rows, VEC doubles per piece
    typedef typename Piece<VEC>::T PT;
    const int P = d / VEC;                                       // pieces per row
    const int total = ntile * P;
    const PT *Xp = reinterpret_cast<const PT *>(a.X + (size_t)c0 * d);
    PT *Xw = reinterpret_cast<PT *>(a.X + (size_t)c0 * d);
    const PT *Q0p = reinterpret_cast<const PT *>(a.Q + (size_t)c0 * d);                            // proposal buffer 0, 1 (the same without Q2)
    const PT *Q1p = reinterpret_cast<const PT *>((a.Q2 ? a.Q2 : a.Q) + (size_t)c0 * d);
    PT *Qn = reinterpret_cast<PT *>(((a.q_tgt && a.Q2) ? a.Q2 : a.Q) + (size_t)c0 * d);           // where the next proposals go
    // piece p = tid + 256 j of the tile: chain p / P, piece p % P of its row -- kept current by increments
    const int dc = 256 / P, dp = 256 % P;
    int cl = tid / P, ip = tid % P;
    for (int p0 = tid; p0 < total; p0 += 256 * UNR) {
        PT v[UNR], u[UNR], m[UNR];                               // state, direction / DE difference, DE's group mask
        int cls[UNR], ips[UNR];
        bool on[UNR];
        // all loads of the batch first
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int p = p0 + 256 * j;
            on[j] = p < total;
            cls[j] = cl; ips[j] = ip;
            if (!PROP && on[j] && !(rec[cl].flags & 3)) on[j] = false;       // accept only: a row that goes nowhere is not read
            if (on[j]) {
                const Rec &r = rec[cl];
                const int src = r.src;
                if (src == 0) v[j] = row_load(Xp + p);
                else if (src == 1) v[j] = row_load(Q0p + p);
                else v[j] = row_load(Q1p + p);
                if (PROP) {
                    if (r.jt == PTMI_J_SCAM) u[j] = *reinterpret_cast<const PT *>(a.Ut + r.urow + (size_t)ip * VEC);
                    else if (r.jt == PTMI_J_AM) u[j] = row_load(reinterpret_cast<const PT *>(a.am_inc + r.urow + (size_t)ip * VEC));
                    else if (r.jt == PTMI_J_DE) {
                        const double *rm = a.DE + r.rm, *rn = a.DE + r.rn;
                        if constexpr (VEC == 2) {
                            const int i = 2 * ip, q0 = de_pos(i, a.lanes), q1 = de_pos(i + 1, a.lanes);
                            u[j] = ptmi_d2{rm[q0] - rn[q0], rm[q1] - rn[q1]};
                            if (r.gm >= 0) m[j] = *reinterpret_cast<const PT *>(a.gmask + r.gm + i);
                        } else {
                            const int q0 = de_pos(ip, a.lanes);
                            u[j] = rm[q0] - rn[q0];
                            if (r.gm >= 0) m[j] = a.gmask[r.gm + ip];
                        }
                    }
                }
            }
            cl += dc; ip += dp;
            if (ip >= P) { ip -= P; cl += 1; }
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (!on[j]) continue;
            const int p = p0 + 256 * j;
            const Rec &r = rec[cls[j]];
#if PTMI_SPLIT_NTX
            if (r.flags & 1) __builtin_nontemporal_store(v[j], Xw + p);
#else
            if (r.flags & 1) Xw[p] = v[j];
#endif
            if (ACC) {
                if (r.flags & 2) {                               // PT:327-328: the rank-0 chain's row, in the buffer's row format
                    const long long ch = c0 + cls[j];
                    double *am = a.AM + ((size_t)(ch / nt) * a.cov_update + (size_t)am_row0) * d;
                    if constexpr (VEC == 2) {
                        am[am_pos(2 * ips[j], a.am_epl)] = v[j].x;
                        am[am_pos(2 * ips[j] + 1, a.am_epl)] = v[j].y;
                    } else am[am_pos(ips[j], a.am_epl)] = v[j];
                }
            }
            if (PROP) {
                PT dq;
                if (r.jt == PTMI_J_SCAM) dq = r.amp * u[j];
                else if (r.jt == PTMI_J_AM) dq = u[j];
                else if (r.jt == PTMI_J_DE) {
                    dq = r.amp * u[j];
                    if (r.gm >= 0) {                             // only the group's parameters move (PT:978-983)
                        if constexpr (VEC == 2) {
                            if (m[j].x == 0.0) dq.x = 0.0;
                            if (m[j].y == 0.0) dq.y = 0.0;
                        } else if (m[j] == 0.0) dq = 0.0;
                    }
                } else {
                    if constexpr (VEC == 2) dq = ptmi_d2{0.0, 0.0};
                    else dq = 0.0;
                }
#if PTMI_SPLIT_NTQ
                __builtin_nontemporal_store(v[j] + dq, Qn + p);
#else
                Qn[p] = v[j] + dq;
#endif
            }
        }
    }
}

// The isotropic Gaussian of n rows [n][d] with the bits of the fused kernels (eval_logl<G, EPL, PTMI_LOGL_ISO>: lane gl of a row's G
// lanes sums elements gl, gl + G, ... by fma, the lanes' sums meet in the xor butterfly G/2 .. 1): a block stages 256 / G rows in
// LDS with 16-byte loads of consecutive addresses, the lanes then read their elements from there.
template <int G>
__global__ __launch_bounds__(256) void rows_iso_kernel(const double *rows, long long n, int d, double *out)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];
    constexpr int R = 256 / G;
    const long long r0 = (long long)blockIdx.x * R;
    const int nr = (int)(n - r0 < R ? n - r0 : R), tid = (int)threadIdx.x;
    const double *src = rows + (size_t)r0 * d;
    const int total = nr * d;
    if ((d & 1) == 0) {
        // global -> LDS without a register round trip (global_load_lds_dwordx4: a wave's 64 pieces land at consecutive LDS addresses,
        // which is the tile's own order); all of a thread's pieces are in flight at once
        const ptmi_d2 *s2 = reinterpret_cast<const ptmi_d2 *>(src);
        ptmi_d2 *t2 = reinterpret_cast<ptmi_d2 *>(tile);
        for (int p = tid; p < total / 2; p += 256)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s2 + p), (__attribute__((address_space(3))) void *)(t2 + p), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
    } else {
        for (int p = tid; p < total; p += 256) tile[p] = src[p];
    }
    __syncthreads();
    const int rl = tid / G, gl = tid % G;
    double p = 0.0;
    if (rl < nr) {
        const double *x = tile + (size_t)rl * d;
        for (int i = gl; i < d; i += G) p = __builtin_fma(x[i], x[i], p);
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) p = p + __shfl_xor(p, m, 64);
    if (rl < nr && gl == 0) out[r0 + rl] = -0.5 * p;
}

template <int VEC>
int launch_rows(ptmi_engine *h, const KArgs &a, int mode)
{
    const long long nch = (long long)a.W * a.nt;
    const unsigned grid = (unsigned)((nch + TILE - 1) / TILE);
    if (mode == 0) hipLaunchKernelGGL((split_rows_kernel<false, true, VEC>), dim3(grid), dim3(256), 0, h->stream, a);
    else if (mode == 1) hipLaunchKernelGGL((split_rows_kernel<true, false, VEC>), dim3(grid), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((split_rows_kernel<true, true, VEC>), dim3(grid), dim3(256), 0, h->stream, a);
    return PTMI_OK;
}

}  // namespace

// Does this handle's split path run on the row kernels?  (No AM entries in the cycle; PTMI_SPLIT_ROWS=0: the shape kernels, an A/B
// and test switch -- same results.)
bool ptmi_split_rows_ok(const ptmi_engine *h)
{
    const char *e = getenv("PTMI_SPLIT_ROWS");                   // read per call: the tests switch it
    if (e && atoi(e) == 0) return false;
    return (h->cfg.w_am == 0 || h->split_am_piece > 0) && h->cfg.ndim >= 1;
}

// mode 0: propose(iter0); 1: accept(iter0); 2: accept(iter0) + propose(iter0 + 1)
int ptmi_split_rows(ptmi_engine *h, const KArgs &a, int mode)
{
    if (a.d % 2 == 0) return launch_rows<2>(h, a, mode);
    return launch_rows<1>(h, a, mode);
}

// ptmi_rows_logl (include/ptmi.h)
int ptmi_rows_iso(ptmi_engine *h, const double *rows, long long n, double *out)
{
    const int d = h->cfg.ndim, G = h->G;
    const size_t lds = sizeof(double) * (size_t)(256 / G) * d;
    if (lds > 160 * 1024) return ptmi_fail(PTMI_EUNSUPPORTED, "ptmi_rows_logl: ndim=%d does not fit the staging tile", d);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return ptmi_fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
        }
        const int R = 256 / G;
        hipLaunchKernelGGL(kern, dim3((unsigned)((n + R - 1) / R)), dim3(256), lds, h->stream, rows, n, d, out);
        return PTMI_OK;
    };
    if (G == 4) return go(rows_iso_kernel<4>);
    if (G == 16) return go(rows_iso_kernel<16>);
    return go(rows_iso_kernel<64>);
}

__global__ void set_iter_kernel(long long *p, long long v) { *p = v; }
int ptmi_set_iter_device(ptmi_engine *h, long long *p, long long v)
{
    hipLaunchKernelGGL(set_iter_kernel, dim3(1), dim3(1), 0, h->stream, p, v);
    return PTMI_OK;
}
