// ptmi_mh.inc.h -- the per-chain kernel templates (proposals, likelihoods, fused MH steps, split path).
// Included by ptmi_shape.hip once per shape.  Reference behaviour cited as PT:<lines> =
// PTMCMCSampler/PTMCMCSampler.py of nanograv/PTMCMCSampler.
#pragma once
#include "ptmi_common.h"

// ----------------------------------------------------------- log-likelihoods
// All G lanes of a group hold q[e] = element (gl + G*e); pad elements are 0.
template <int G, int EPL, int LOGL>
__device__ __forceinline__ double eval_logl(const KArgs &a, const double (&q)[EPL], int gl, const double *Pt)
{
    const int d = a.d;
    if (LOGL == PTMI_LOGL_ISO) {
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p = __builtin_fma(q[e], q[e], p);
        return -0.5 * group_sum<G>(p);
    } else if (LOGL == PTMI_LOGL_DENSE) {
        const double *mu = a.logl_par;
        double r[EPL], v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            r[e] = i < d ? q[e] - mu[i] : 0.0;
            v[e] = 0.0;
        }
#pragma unroll
        for (int e2 = 0; e2 < EPL; ++e2) {
#pragma unroll 1
            for (int src = 0; src < G; ++src) {
                const int j = src + G * e2;
                if (j >= d) break;
                const double rj = group_bcast_lane<G>(r[e2], src);
                const double *row = Pt + (size_t)j * d;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int i = gl + G * e;
                    if (i < d) v[e] = __builtin_fma(row[i], rj, v[e]);
                }
            }
        }
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p = __builtin_fma(r[e], v[e], p);
        return -0.5 * group_sum<G>(p);
    } else {  // PTMI_LOGL_CURVED: pairs (2m, 2m+1); G is even so a pair lives in lanes (gl, gl+1) of one slot
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            const double y = dppf64<0xB1>(q[e]);  // partner lane (xor 1)
            double t = 0.0;
            if (!(gl & 1) && i + 1 < d) {
                const double x = q[e], x2 = x * x;
                const double g = 9.0 + 4.0 * x2 + 9.0 * y;
                const double l0 = -x2 - g * g;
                const double ym = y - 2.0;
                const double l1 = -8.0 * x2 - 8.0 * (ym * ym);
                t = det_log(det_exp(l0) + 0.5 * det_exp(l1));
            }
            p = __builtin_fma(t, 1.0, p);
        }
        return group_sum<G>(p);
    }
}

template <int G, int EPL>
__device__ __forceinline__ double eval_logp(const KArgs &a, const double (&q)[EPL], int gl)
{
    if (a.logp_kind == PTMI_LOGP_BOX) {
        const double *lo = a.logp_par, *hi = a.logp_par + a.d;
        bool ok = true;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            if (i < a.d) ok = ok && (lo[i] <= q[e]) && (hi[i] >= q[e]);
        }
        return group_all<G>(ok) ? 0.0 : -__builtin_inf();
    }
    return 0.0;
}

// ---------------------------------------------------------------- proposals
// Per-chain constants of the jump scales, hoisted out of the step loop.  Same operation
// order as the reference: scale in {10, 0.2, 1.0}; scale *= sqrt(temp) if temp <= 100
// (PT:846-862); cd = 2.4 / sqrt(2 neff) * scale (PT:870, 928).
struct ChainConst {
    double cd_scam[3], cd_am[3];   // by scale branch: prob > 0.97, prob > 0.9, else
    double de_div, de_mul;         // DE: rr * 2.4 / de_div * de_mul  (PT:976)
};
__device__ __forceinline__ ChainConst chain_const(double temp, double beta, int d)
{
    ChainConst c;
    const double sT = temp <= 100.0 ? det_sqrt(temp) : 1.0;
    const double base[3] = {10.0, 0.2, 1.0};
    const double c1 = 2.4 / det_sqrt(2.0 * 1.0), cn = 2.4 / det_sqrt(2.0 * (double)d);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double sc = temp <= 100.0 ? base[j] * sT : base[j];
        c.cd_scam[j] = c1 * sc;
        c.cd_am[j] = cn * sc;
    }
    c.de_div = det_sqrt(2.0 * (double)d);
    c.de_mul = det_sqrt(1.0 / beta);
    return c;
}

// Kernel shapes: (lanes per chain G, register slots per lane EPL); a shape serves
// G*EPL_prev < ndim <= G*EPL, so slots e < safe_slots(G, EPL) hold a valid element on every
// lane for every ndim the shape serves and need no bounds check.
constexpr int safe_slots(int G, int EPL)
{
    return G == 4 ? (EPL == 26 ? 20 : EPL == 20 ? 14 : EPL == 14 ? 8 : EPL == 8 ? 5 : EPL == 5 ? 2 : 0)
         : G == 16 ? (EPL == 26 ? 13 : EPL == 13 ? 7 : EPL == 7 ? 6 : 0)
         : (EPL == 32 ? 16 : EPL == 16 ? 8 : EPL == 8 ? 6 : 0);
}
// element i = gl + G*e of a table row
#define PTMI_ROW_LOAD(dst, row, e)                                         \
    do {                                                                   \
        if ((e) < safe_slots(G, EPL)) dst = (row)[gl + G * (e)];           \
        else dst = (gl + G * (e)) < d ? (row)[gl + G * (e)] : 0.0;         \
    } while (0)

// One proposal for the caller's chain (PT:1048-1067, 820-985): writes the increment dq
// (q = x + dq) and returns the jump type.  log_u = log(accept uniform), evaluated in the
// same instruction stream as the Box-Muller log, on another lane of each quad.
template <int G, int EPL, bool FULL>
__device__ __forceinline__ int propose(const KArgs &a, long long it, u32 sid, int gl, const ChainConst &cc,
                                       const double *Ut, const double *S, const double *DE,
                                       double (&dq)[EPL], double &log_u, double &u_acc)
{
    const int d = a.d;
    // the four lanes of a quad evaluate slots A..D of this chain in one pass
    u64 w0, w1;
    philox_words(a.seed, (u64)it, sid, (u32)(gl & 3), w0, w1);
    const u64 A0 = quad_bcast<0>(w0), A1 = quad_bcast<0>(w1);
    const u64 B0 = quad_bcast<1>(w0), B1 = quad_bcast<1>(w1);
    // one log stream: lane B -> log(accept uniform), lane D -> log(u1) of the SCAM normal
    const double larg = (gl & 3) == 1 ? w2uniform(w0) : w2uniform_open(w0);
    const double lg = det_log(larg);
    log_u = quad_bcastf<1>(lg);
    u_acc = w2uniform(B0);

    int jt = PTMI_J_SCAM;
    if (FULL) {
        const int L = a.w_host + a.w_scam + a.w_am + (a.de_on ? a.w_de : 0);
        const int pick = (int)w2index(A0, (u64)L);
        const int ind = pick - a.w_host;
        jt = ind < a.w_scam ? PTMI_J_SCAM : (ind < a.w_scam + a.w_am ? PTMI_J_AM : PTMI_J_DE);
        if (ind < 0) {                          // a host-served cycle entry: hand the state back unchanged
#pragma unroll
            for (int e = 0; e < EPL; ++e) dq[e] = 0.0;
            return PTMI_J_NTYPES + pick;
        }
    }
    const double prob = w2uniform(A1);
    const int br = prob > 0.97 ? 0 : (prob > 0.9 ? 1 : 2);

    if (jt == PTMI_J_SCAM) {
        const int k = (int)w2index(B1, (u64)d);
        const double *col = Ut + (size_t)k * d;
        // the direction lands in dq (issued before the normal is computed, so its latency is covered) and
        // is scaled in place
#pragma unroll
        for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(dq[e], col, e);
        const double sk = S[k];
        const u64 D1 = quad_bcast<3>(w1);
        const double ln1 = quad_bcastf<3>(lg);
        const double z = det_sqrt(-2.0 * ln1) * det_cos2pi(w2uniform(D1));
        const double cd = br == 0 ? cc.cd_scam[0] : (br == 1 ? cc.cd_scam[1] : cc.cd_scam[2]);
        const double amp = z * cd * det_sqrt(sk);             // PT:873
#pragma unroll
        for (int e = 0; e < EPL; ++e) dq[e] = amp * dq[e];
    } else if (FULL && jt == PTMI_J_AM) {
        const double cd = br == 0 ? cc.cd_am[0] : (br == 1 ? cc.cd_am[1] : cc.cd_am[2]);
        double wk[EPL];
        // directions k = gl + G*e and k + G (slots e even / odd) are the cos and sin branches of ONE Box-Muller
#pragma unroll
        for (int e = 0; e < EPL; e += 2) {
            const int k = gl + G * e;
            dq[e] = 0.0;
            wk[e] = 0.0;
            if (e + 1 < EPL) { dq[e + 1] = 0.0; wk[e + 1] = 0.0; }
            if (k < d) {
                u64 e0, e1;
                philox_words(a.seed, (u64)it, sid, SLOT_AM + (u32)k, e0, e1);
                const double r = det_sqrt(-2.0 * det_log(w2uniform_open(e0)));
                double sn, cs;
                det_sincos2pi(w2uniform(e1), sn, cs);
                wk[e] = (r * cs) * cd * det_sqrt(S[k]);                        // PT:930
                if (e + 1 < EPL && k + G < d) wk[e + 1] = (r * sn) * cd * det_sqrt(S[k + G]);
            }
        }
#pragma unroll
        for (int e2 = 0; e2 < EPL; ++e2) {
#pragma unroll 1
            for (int src = 0; src < G; ++src) {
                const int k = src + G * e2;
                if (k >= d) break;
                const double wv = group_bcast_lane<G>(wk[e2], src);
                const double *row = Ut + (size_t)k * d;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    double r;
                    PTMI_ROW_LOAD(r, row, e);
                    dq[e] = __builtin_fma(r, wv, dq[e]);
                }
            }
        }
    } else if (FULL) {
        const int Bn = a.de_size;
        const u64 C0 = quad_bcast<2>(w0), C1 = quad_bcast<2>(w1);
        const int mm = (int)w2index(B1, (u64)Bn);
        const int nn = (int)(((u64)mm + 1ull + w2index(C0, (u64)(Bn - 1))) % (u64)Bn);
        double scale;
        if (prob > 0.5) scale = 1.0;
        else scale = w2uniform(C1) * 2.4 / cc.de_div * cc.de_mul;  // PT:976
        const double *rm = DE + (size_t)((mm + a.de_head) % Bn) * d;
        const double *rn = DE + (size_t)((nn + a.de_head) % Bn) * d;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            double vm, vn;
            PTMI_ROW_LOAD(vm, rm, e);
            PTMI_ROW_LOAD(vn, rn, e);
            dq[e] = scale * (vm - vn);
        }
    }
    return jt;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; make consecutive
// logical blocks (chains of one walker, sharing its Ut) land on one XCD's L2.
__device__ __forceinline__ int logical_block()
{
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// ------------------------------------------------------------ fused MH steps
template <int G, int EPL, int LOGL, bool FULL, bool STAGE>
__global__ __launch_bounds__(256) void mh_steps_kernel(const KArgs a)
{
    constexpr int CPB = 256 / G;
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    long long ch = (long long)logical_block() * CPB + (int)(threadIdx.x / G);
    const bool live = ch < nch;
    if (!live) ch = nch - 1;
    const int gl = (int)(threadIdx.x % G);
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const int tg = a.temp0 + t;
    const double beta = a.beta[t];
    const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
    const u32 sid = (u32)((u64)(a.walker0 + w) * (u32)a.ntg + (u32)tg);
    const size_t wc = a.per_walker ? (size_t)w : 0;
    const double *Ut = a.Ut + wc * d * d, *S = a.S + wc * d;
    const double *DE = (FULL && a.DE) ? a.DE + wc * (size_t)a.de_size * d : nullptr;
    double *xrow = a.X + (size_t)ch * d;

    // STAGE: the tables every chain of the block walks row by row (the block's Ut for AM / SCAM, the precision
    // matrix of the dense likelihood) are copied to LDS once per launch.  The host picks STAGE when they fit in
    // 160 KiB and all chains of a block share them.
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const double *PtG = LOGL == PTMI_LOGL_DENSE ? a.logl_par + d : nullptr;
    const double *Pt = PtG, *Utab = Ut;
    if (STAGE) {
        const int n = d * d;
        double *Pl = smem, *Ul = smem + (LOGL == PTMI_LOGL_DENSE ? (size_t)n : 0);
        if (LOGL == PTMI_LOGL_DENSE) {
            for (int i = (int)threadIdx.x; i < n; i += 256) Pl[i] = PtG[i];
            Pt = Pl;
        }
        if (FULL) {
            // all chains of the block belong to one walker (or the table is pooled): take the first chain's
            const long long ch0 = (long long)logical_block() * CPB;
            const size_t w0 = a.per_walker ? (size_t)((ch0 < nch ? ch0 : nch - 1) / nt) : 0;
            const double *src = a.Ut + w0 * d * d;
            for (int i = (int)threadIdx.x; i < n; i += 256) Ul[i] = src[i];
            Utab = Ul;
        }
        __syncthreads();
    }

    double x[EPL], dq[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(x[e], xrow, e);
    double lnL = a.lnL[ch], lp = a.lp[ch];
    u32 nacc = 0, jp[PTMI_J_NTYPES] = {0, 0, 0}, ja[PTMI_J_NTYPES] = {0, 0, 0};
    const bool cold = live && tg == 0 && a.AM != nullptr;
    int am_row = a.am_row0;

    for (int k = 0; k < a.nsteps; ++k) {
        const long long it = a.iter0 + k;
        double log_u, u_acc;
        const int jt = propose<G, EPL, FULL>(a, it, sid, gl, cc, Utab, S, DE, dq, log_u, u_acc);
        if (FULL) {
#pragma unroll
            for (int j = 0; j < PTMI_J_NTYPES; ++j) jp[j] += (jt == j);
        }
        // PT:605-612
        double nlp, nlnL = 0.0, nlnprob;
        {
            double q[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) q[e] = x[e] + dq[e];
            nlp = eval_logp<G, EPL>(a, q, gl);
            if (nlp == -__builtin_inf()) nlnprob = -__builtin_inf();
            else {
                nlnL = eval_logl<G, EPL, LOGL>(a, q, gl, Pt);
                nlnprob = beta * nlnL + nlp;
            }
        }
        // PT:615-622
        const double lnprob0 = beta * lnL + lp;
        const double diff = nlnprob - lnprob0 + 0.0;
        if (diff > log_u) {
            // x + dq again (bit-identical to q); keeping q alive instead would cost EPL more registers
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                double inc = dq[e];
                asm volatile("" : "+v"(inc));
                x[e] = x[e] + inc;
            }
            lnL = nlnL;
            lp = nlp;
            nacc += 1;
            if (FULL) {
#pragma unroll
                for (int j = 0; j < PTMI_J_NTYPES; ++j) ja[j] += (jt == j);
            }
        }
        // PT:327-328 (the post-swap row of a swap iteration is written by the swap)
        if (cold && !(a.swap_last && k == a.nsteps - 1)) {
            double *am = a.AM + ((size_t)w * a.cov_update + (size_t)am_row) * d;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e;
                if (e < safe_slots(G, EPL) || i < d) am[i] = x[e];
            }
            if (a.AMaux && gl == 0) {
                double *ax = a.AMaux + ((size_t)w * a.cov_update + (size_t)am_row) * 2;
                ax[0] = lnL;
                ax[1] = lp;
            }
        }
        am_row = am_row + 1 == a.cov_update ? 0 : am_row + 1;
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            if (e < safe_slots(G, EPL) || i < d) xrow[i] = x[e];
        }
        if (gl == 0) {
            a.lnL[ch] = lnL;
            a.lp[ch] = lp;
            const size_t r = (size_t)w * nt + t;
            a.nacc[r] += nacc;
            if (!FULL) { jp[PTMI_J_SCAM] = (u32)a.nsteps; ja[PTMI_J_SCAM] = nacc; }
#pragma unroll
            for (int j = 0; j < PTMI_J_NTYPES; ++j) {
                a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 0] += jp[j];
                a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 1] += ja[j];
            }
        }
    }
}

// split path: proposal only / accept only, one iteration (host likelihood callbacks)
template <int G, int EPL>
__global__ __launch_bounds__(256) void propose_kernel(const KArgs a)
{
    constexpr int CPB = 256 / G;
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    long long ch = (long long)logical_block() * CPB + (int)(threadIdx.x / G);
    const bool live = ch < nch;
    if (!live) ch = nch - 1;
    const int gl = (int)(threadIdx.x % G);
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const double beta = a.beta[t];
    const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
    const u32 sid = (u32)((u64)(a.walker0 + w) * (u32)a.ntg + (u32)(a.temp0 + t));
    const size_t wc = a.per_walker ? (size_t)w : 0;
    const double *Ut = a.Ut + wc * d * d, *S = a.S + wc * d;
    const double *DE = a.DE ? a.DE + wc * (size_t)a.de_size * d : nullptr;
    const double *xrow = a.X + (size_t)ch * d;
    double x[EPL], dq[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(x[e], xrow, e);
    double log_u, u_acc;
    const int jt = propose<G, EPL, true>(a, a.iter0, sid, gl, cc, Ut, S, DE, dq, log_u, u_acc);
    if (live) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            if (i < d) a.Q[(size_t)ch * d + i] = x[e] + dq[e];
        }
        if (gl == 0) {
            a.qaux[ch * 4 + 0] = 0.0;  // qxy of the built-in jumps (PT:836,894,952)
            a.qaux[ch * 4 + 1] = (double)jt;
            a.qaux[ch * 4 + 2] = u_acc;
            a.qaux[ch * 4 + 3] = log_u;
        }
    }
}

template <int G, int EPL>
__global__ __launch_bounds__(256) void accept_kernel(const KArgs a)
{
    constexpr int CPB = 256 / G;
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    const long long ch = (long long)logical_block() * CPB + (int)(threadIdx.x / G);
    if (ch >= nch) return;
    const int gl = (int)(threadIdx.x % G);
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const double beta = a.beta[t];
    const double nlp = a.newlp[ch];
    const double nlnL = a.newlnL[ch];
    const double nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
    const double lnprob0 = beta * a.lnL[ch] + a.lp[ch];
    const double diff = nlnprob - lnprob0 + a.qaux[ch * 4 + 0];
    const int jt = (int)a.qaux[ch * 4 + 1];
    const bool acc = diff > a.qaux[ch * 4 + 3];
    const bool cold = a.temp0 + t == 0 && a.AM != nullptr;
    double *am = cold && !a.swap_last ? a.AM + ((size_t)w * a.cov_update + (size_t)a.am_row0) * d : nullptr;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = gl + G * e;
        if (i < d) {
            const double v = acc ? a.Q[(size_t)ch * d + i] : a.X[(size_t)ch * d + i];
            if (acc) a.X[(size_t)ch * d + i] = v;
            if (am) am[i] = v;
        }
    }
    if (gl == 0) {
        const size_t r = (size_t)w * nt + t;
        if (jt >= 0 && jt < PTMI_J_NTYPES) a.jstat[(r * PTMI_J_NTYPES + jt) * 2 + 0] += 1;
        if (am && a.AMaux) {
            double *ax = a.AMaux + ((size_t)w * a.cov_update + (size_t)a.am_row0) * 2;
            ax[0] = acc ? nlnL : a.lnL[ch];
            ax[1] = acc ? nlp : a.lp[ch];
        }
        if (acc) {
            a.lnL[ch] = nlnL;
            a.lp[ch] = nlp;
            a.nacc[r] += 1;
            if (jt >= 0 && jt < PTMI_J_NTYPES) a.jstat[(r * PTMI_J_NTYPES + jt) * 2 + 1] += 1;
        }
        a.qaux[ch * 4 + 2] = acc ? 1.0 : 0.0;   // decision, for the host's per-name jump statistics
    }
}

// initial lnL / lp (PT:479-487)
template <int G, int EPL, int LOGL>
__global__ __launch_bounds__(256) void eval_state_kernel(const KArgs a)
{
    constexpr int CPB = 256 / G;
    const int d = a.d;
    const long long nch = (long long)a.W * a.nt;
    long long ch = (long long)blockIdx.x * CPB + (int)(threadIdx.x / G);
    const bool live = ch < nch;
    if (!live) ch = nch - 1;
    const int gl = (int)(threadIdx.x % G);
    double x[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = gl + G * e;
        x[e] = i < d ? a.X[(size_t)ch * d + i] : 0.0;
    }
    const double lp = eval_logp<G, EPL>(a, x, gl);
    double lnL = -__builtin_inf();
    if (lp != -__builtin_inf()) lnL = eval_logl<G, EPL, LOGL>(a, x, gl, LOGL == PTMI_LOGL_DENSE ? a.logl_par + d : nullptr);
    if (live && gl == 0) {
        a.lp[ch] = lp;
        a.lnL[ch] = lnL;
    }
}

template <int G, int EPL, int LOGL, bool FULL>
static int launch_mh_k(ptmi_engine *h, KArgs &a, int grid)
{
    const ptmi_config &c = h->cfg;
    constexpr bool WANTS = G == 4 && (FULL || LOGL == PTMI_LOGL_DENSE);   // the tables fit only for the small-ndim shapes
    if (WANTS) {
        const size_t tab = sizeof(double) * (size_t)c.ndim * c.ndim;
        const size_t lds = tab * ((FULL ? 1 : 0) + (LOGL == PTMI_LOGL_DENSE ? 1 : 0));
        const bool one_table_per_block = !FULL || !c.cov_per_walker || c.ntemps % (256 / G) == 0;
        if (lds <= 160 * 1024 && one_table_per_block) {
            auto kern = mh_steps_kernel<G, EPL, LOGL, FULL, WANTS>;
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, h->stream, a);
            return PTMI_OK;
        }
    }
    hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, FULL, false>), dim3(grid), dim3(256), 0, h->stream, a);
    return PTMI_OK;
}
template <int G, int EPL, int LOGL>
static int launch_mh_l(ptmi_engine *h, KArgs &a, int grid, bool full)
{
    return full ? launch_mh_k<G, EPL, LOGL, true>(h, a, grid) : launch_mh_k<G, EPL, LOGL, false>(h, a, grid);
}
