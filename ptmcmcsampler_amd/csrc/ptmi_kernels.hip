// ptmi_kernels.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of libptmi.so.
//
// Hot path of a PTSampler-compatible parallel-tempering sampler, batched over
// (walkers x temperatures) chains per GPU.  Reference behaviour cited as PT:<lines>
// = PTMCMCSampler/PTMCMCSampler.py of nanograv/PTMCMCSampler.  See include/ptmi.h
// for the boundary and DESIGN.md for layouts and the RNG schedule.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/ptmi.h"
#include "ptmi_device.h"

using namespace ptmi;

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(PTMI_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------- kernel args
struct KArgs {
    // state
    double *X, *lnL, *lp;
    int32_t *temp_of, *slot_of;
    const double *Ut, *S, *DE;
    double *AM, *AMaux;
    u64 *nacc, *jstat;
    // small device tables owned by the engine
    const double *temps_mh, *beta, *logl_par, *logp_par;
    // split path
    double *Q, *qaux;
    const double *newlnL, *newlp;
    // scalars
    u64 seed;
    long long iter0;
    int nsteps;
    int d, nt, W, ntg, temp0, walker0;
    int w_host, w_scam, w_am, w_de, de_on, de_size, de_head;
    int cov_update, tskip, per_walker, logp_kind;
    int am_row0, swap_last;      // iter0 % cov_update; the last step of the launch is a swap iteration
};

template <int G>
__device__ __forceinline__ double group_bcast_lane(double v, int src)
{
    // lane `src` (0..G-1) of the caller's group
    const int lane = (int)(threadIdx.x & 63);
    return __shfl(v, (lane & ~(G - 1)) + src, 64);
}

// ----------------------------------------------------------- log-likelihoods
// All G lanes of a group hold q[e] = element (gl + G*e); pad elements are 0.
template <int G, int EPL, int LOGL>
__device__ __forceinline__ double eval_logl(const KArgs &a, const double (&q)[EPL], int gl)
{
    const int d = a.d;
    if (LOGL == PTMI_LOGL_ISO) {
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p = __builtin_fma(q[e], q[e], p);
        return -0.5 * group_sum<G>(p);
    } else if (LOGL == PTMI_LOGL_DENSE) {
        const double *mu = a.logl_par, *Pt = a.logl_par + d;
        double r[EPL], v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            r[e] = i < d ? q[e] - mu[i] : 0.0;
            v[e] = 0.0;
        }
#pragma unroll
        for (int e2 = 0; e2 < EPL; ++e2) {
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
    return G == 4 ? (EPL == 26 ? 20 : EPL == 20 ? 13 : EPL == 13 ? 8 : EPL == 8 ? 5 : EPL == 5 ? 2 : 0)
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
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int k = gl + G * e;
            dq[e] = 0.0;
            wk[e] = 0.0;
            if (k < d) {
                u64 e0, e1;
                philox_words(a.seed, (u64)it, sid, SLOT_AM + (u32)k, e0, e1);
                wk[e] = det_normal(e0, e1) * cd * det_sqrt(S[k]);  // PT:930
            }
        }
#pragma unroll
        for (int e2 = 0; e2 < EPL; ++e2) {
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
template <int G, int EPL, int LOGL, bool FULL>
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
        const int jt = propose<G, EPL, FULL>(a, it, sid, gl, cc, Ut, S, DE, dq, log_u, u_acc);
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
                nlnL = eval_logl<G, EPL, LOGL>(a, q, gl);
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
    if (lp != -__builtin_inf()) lnL = eval_logl<G, EPL, LOGL>(a, x, gl);
    if (live && gl == 0) {
        a.lp[ch] = lp;
        a.lnL[ch] = lnL;
    }
}

// --------------------------------------------------------------------- swap
__global__ void gather_lnl_kernel(const double *lnL, const int32_t *slot_of, double *out, long long n, int nt)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long w = i / nt;
    out[i] = lnL[w * nt + slot_of[i]];
}

// PT:666-686, one lane per walker, hot -> cold with the carried map.  When the whole
// ladder is local (slot_of != nullptr) the slot tables are rewritten in place: position
// k+1 becomes final at step k and positions <= k are still untouched.
__global__ void swap_sweep_kernel(int W, int n, const double *ladder, const double *lnL_pos, const double *lnL_rows,
                                  int32_t *slot_of, int32_t *temp_of, int32_t *map, u64 *nswap, int local0, int nlocal,
                                  long long iter, u64 seed, int walker0)
{
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= W) return;
    const u32 sid = (u32)((u64)(walker0 + w) * (u32)n + 0u);
    const bool fused = slot_of != nullptr;
    int32_t *so = fused ? slot_of + (size_t)w * n : nullptr;
    int32_t *to = fused ? temp_of + (size_t)w * n : nullptr;
    const double *Lp = fused ? lnL_rows + (size_t)w * n : lnL_pos + (size_t)w * n;
    int c = n - 1;                         // position whose state is carried at k+1
    int crow = fused ? so[n - 1] : 0;
    double Lc = fused ? Lp[crow] : Lp[n - 1];
    for (int k = n - 2; k >= 0; --k) {
        u64 w0, w1;
        philox_words(seed, (u64)iter, sid, SLOT_SWAP + (u32)k, w0, w1);
        const double u = w2uniform(w0);
        const int krow = fused ? so[k] : 0;
        const double La = fused ? Lp[krow] : Lp[k];
        const double Tk = ladder[k], Tk1 = ladder[k + 1];
        double la = -La / Tk;
        la += -Lc / Tk1;
        la += Lc / Tk;
        la += La / Tk1;
        const bool acc = u <= det_exp(la);
        // position k+1 is final: it keeps the carried state, or takes position k's
        const int fin = acc ? k : c;
        if (map) map[(size_t)w * n + k + 1] = fin;
        if (fused) {
            const int frow = acc ? krow : crow;
            so[k + 1] = frow;
            to[frow] = k + 1;
            if (!acc) crow = krow;
        }
        if (acc) {
            if (k >= local0 && k < local0 + nlocal) nswap[(size_t)w * n + k] += 1;
        } else {
            c = k;
            Lc = La;
        }
    }
    if (map) map[(size_t)w * n] = c;
    if (fused) {
        so[0] = crow;
        to[crow] = 0;
    }
}

// AM-buffer row of a swap iteration: the state that now sits at rank 0 (PT:624-627, 327-328)
__global__ void am_write_kernel(const double *X, const double *lnL, const double *lp, const int32_t *slot_of, double *AM,
                                double *AMaux, int W, int nt, int d, int cov_update, long long iter)
{
    const int w = (int)blockIdx.x;
    const size_t r = (size_t)w * nt + slot_of[(size_t)w * nt];
    const double *row = X + r * d;
    double *am = AM + ((size_t)w * cov_update + (size_t)(iter % cov_update)) * d;
    for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) am[i] = row[i];
    if (AMaux && threadIdx.x == 0) {
        double *ax = AMaux + ((size_t)w * cov_update + (size_t)(iter % cov_update)) * 2;
        ax[0] = lnL[r];
        ax[1] = lp[r];
    }
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
                                                     long long iter, int cov_stride_per_walker)
{
    __shared__ double sh[2][2][WTILE];  // [buf][diff|e][WTILE]
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
    constexpr int PF = 8;   // rows fetched ahead by the carrier threads (one HBM latency per PF rows)
    for (int ii0 = 0; ii0 < mem; ii0 += PF) {
        double vpre[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) vpre[u] = (carrier && ii0 + u < mem) ? am[(size_t)(ii0 + u) * d + rel] : 0.0;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int ii = ii0 + u;
            if (ii >= mem) break;
            it += 1;
            const int bsel = ii & 1;
            if (role < 2) {
                double df = 0.0, ev = 0.0;
                if (carrier) {
                    const double v = vpre[u];
                    df = v - m;
                    m += df / (double)it;
                    ev = v - m;
                }
                if (role == 0) sh[bsel][0][ridx] = df;
                else sh[bsel][1][ridx] = ev;
            }
            __syncthreads();
            double dv[WT], evv[WT];
#pragma unroll
            for (int p = 0; p < WT; ++p) dv[p] = sh[bsel][0][ty + 16 * p];
#pragma unroll
            for (int r = 0; r < WT; ++r) evv[r] = sh[bsel][1][tx + 16 * r];
#pragma unroll
            for (int p = 0; p < WT; ++p)
#pragma unroll
                for (int r = 0; r < WT; ++r) {
                    if (FUSED) acc[p][r] = __builtin_fma(dv[p], evv[r], acc[p][r]);   // pooled mode: not a reference replica
                    else acc[p][r] += dv[p] * evv[r];                                 // PT:792, one product and one sum
                }
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

__global__ void welford_mean_kernel(const double *AM, double *mu, int d, int mem, long long iter)
{
    const int w = (int)blockIdx.y;
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= d) return;
    const double *am = AM + (size_t)w * mem * d;
    long long it = iter - mem;
    double m = it == 0 ? 0.0 : mu[(size_t)w * d + j];
    for (int ii = 0; ii < mem; ++ii) {
        it += 1;
        const double df = am[(size_t)ii * d + j] - m;
        m += df / (double)it;
    }
    mu[(size_t)w * d + j] = m;
}

// Pooled covariance: Chan et al. combination of partial statistics (n_k, mu_k, M2_k), inputs ascending, one thread
// per (i,j).  Level 1 (blockIdx.y = group of 64 walkers) writes group partials, level 2 combines the groups; the
// final pass divides by N-1.
constexpr int POOL_GS = 64;
__global__ void pool_combine_kernel(const double *mu, const double *M2, double *mu_out, double *M2_out, int d, int nin_total,
                                    int gs, double nb, double nb_last_input, double den)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)d * d) return;
    const int g = (int)blockIdx.y;
    const int k0 = g * gs, cnt = k0 + gs <= nin_total ? gs : nin_total - k0;
    const int i = (int)(idx / d), j = (int)(idx % d);
    const double *mw = mu + (size_t)k0 * d, *Mw = M2 + (size_t)k0 * d * d;
    double mi = 0.0, mj = 0.0, M = 0.0, na = 0.0;
    for (int k = 0; k < cnt; ++k) {
        const double nk = (k0 + k == nin_total - 1) ? nb_last_input : nb, nn = na + nk;
        const double f = na * nk / nn, gg = nk / nn;
        const double wi = mw[(size_t)k * d + i], wj = mw[(size_t)k * d + j];
        const double di = wi - mi, dj = wj - mj;
        M = (M + Mw[(size_t)k * d * d + idx]) + (di * dj) * f;
        mi = mi + di * gg;
        mj = mj + dj * gg;
        na = nn;
    }
    M2_out[(size_t)g * d * d + idx] = den > 0.0 ? M / den : M;
    if (mu_out && j == 0) mu_out[(size_t)g * d + i] = mi;
}

// DE history ring: rows [head, head+mem) are the oldest; overwrite them with the AM buffer
__global__ void de_update_kernel(double *DE, const double *AM, int d, int de_size, int mem, int head, int W, int pooled)
{
    const int r = (int)blockIdx.x;   // new row index 0..mem-1 (or the tail when mem > de_size)
    const int wc = (int)blockIdx.y;
    const int skip = mem > de_size ? mem - de_size : 0;
    if (r < skip) return;
    const int phys = (head + (r - skip)) % de_size;
    const int src_w = pooled ? r % W : wc;
    const double *src = AM + ((size_t)src_w * mem + r) * d;
    double *dst = DE + ((size_t)wc * de_size + phys) * d;
    for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) dst[i] = src[i];
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
struct ptmi_engine {
    ptmi_config cfg;
    ptmi_buffers buf;
    hipStream_t stream;
    double *d_ladder, *d_temps, *d_beta, *d_loglpar, *d_logppar;
    double *d_lnlpos;   // [W][ntg] scratch for the fused swap
    double *d_pool_mu, *d_pool_M2;   // [ngroups][d], [ngroups][d*d] partial statistics of the pooled covariance
    int G, EPL;
    int de_on, de_head;
    hipEvent_t ev0, ev1;
};

struct Shape { int G, EPL; };
static bool pick_shape(int d, Shape *s)
{
    static const Shape table[] = {{4, 2}, {4, 5}, {4, 8}, {4, 13}, {4, 20}, {4, 26}, {16, 7}, {16, 13}, {16, 26}, {64, 8}, {64, 16}, {64, 32}};
    const int G = ptmi_lanes_for(d);
    for (const Shape &c : table)
        if (c.G == G && c.G * c.EPL >= d) { *s = c; return true; }
    return false;
}

static KArgs make_args(ptmi_engine *h)
{
    KArgs a;
    memset(&a, 0, sizeof(a));
    const ptmi_config &c = h->cfg;
    const ptmi_buffers &b = h->buf;
    a.X = b.X; a.lnL = b.lnL; a.lp = b.lp; a.temp_of = b.temp_of; a.slot_of = b.slot_of;
    a.Ut = b.Ut; a.S = b.S; a.DE = b.DE; a.AM = c.temp0 == 0 ? b.AM : nullptr; a.AMaux = c.temp0 == 0 ? b.AMaux : nullptr;
    a.nacc = (u64 *)b.nacc; a.jstat = (u64 *)b.jstat;
    a.temps_mh = h->d_temps; a.beta = h->d_beta; a.logl_par = h->d_loglpar; a.logp_par = h->d_logppar;
    a.Q = b.Q; a.qaux = b.qaux;
    a.seed = c.seed;
    a.d = c.ndim; a.nt = c.ntemps; a.W = c.nwalkers; a.ntg = c.ntemps_global; a.temp0 = c.temp0; a.walker0 = c.walker0;
    a.w_host = c.w_host; a.w_scam = c.w_scam; a.w_am = c.w_am; a.w_de = c.w_de; a.de_on = h->de_on; a.de_size = c.de_size; a.de_head = h->de_head;
    a.cov_update = c.cov_update; a.tskip = c.tskip; a.per_walker = c.cov_per_walker; a.logp_kind = c.logp_kind;
    return a;
}

template <int G, int EPL, int LOGL>
static void launch_mh_l(ptmi_engine *h, const KArgs &a, int grid, bool full)
{
    if (full) hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, true>), dim3(grid), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, false>), dim3(grid), dim3(256), 0, h->stream, a);
}
template <int G, int EPL>
static void launch_mh(ptmi_engine *h, const KArgs &a, int grid, bool full)
{
    switch (h->cfg.logl_kind) {
    case PTMI_LOGL_ISO: launch_mh_l<G, EPL, PTMI_LOGL_ISO>(h, a, grid, full); break;
    case PTMI_LOGL_DENSE: launch_mh_l<G, EPL, PTMI_LOGL_DENSE>(h, a, grid, full); break;
    default: launch_mh_l<G, EPL, PTMI_LOGL_CURVED>(h, a, grid, full); break;
    }
}
template <int G, int EPL>
static void launch_eval(ptmi_engine *h, const KArgs &a, int grid)
{
    switch (h->cfg.logl_kind) {
    case PTMI_LOGL_ISO: hipLaunchKernelGGL((eval_state_kernel<G, EPL, PTMI_LOGL_ISO>), dim3(grid), dim3(256), 0, h->stream, a); break;
    case PTMI_LOGL_DENSE: hipLaunchKernelGGL((eval_state_kernel<G, EPL, PTMI_LOGL_DENSE>), dim3(grid), dim3(256), 0, h->stream, a); break;
    default: hipLaunchKernelGGL((eval_state_kernel<G, EPL, PTMI_LOGL_CURVED>), dim3(grid), dim3(256), 0, h->stream, a); break;
    }
}

#define FOR_SHAPE(G_, E_, CALL)                       \
    if (h->G == G_ && h->EPL == E_) { CALL(G_, E_); } else

#ifdef PTMI_ONLY_BENCH_SHAPE   /* developer switch: compile the d=100 shape only (fast asm inspection builds) */
#ifndef PTMI_DEV_G
#define PTMI_DEV_G 4
#define PTMI_DEV_E 26
#endif
#define DISPATCH_SHAPE(CALL) FOR_SHAPE(PTMI_DEV_G, PTMI_DEV_E, CALL) { return fail(PTMI_EUNSUPPORTED, "shape not compiled in"); }
#else
#define DISPATCH_SHAPE(CALL)                                                                        \
    FOR_SHAPE(4, 2, CALL) FOR_SHAPE(4, 5, CALL) FOR_SHAPE(4, 8, CALL) FOR_SHAPE(4, 13, CALL)        \
    FOR_SHAPE(4, 20, CALL) FOR_SHAPE(4, 26, CALL) FOR_SHAPE(16, 7, CALL) FOR_SHAPE(16, 13, CALL)    \
    FOR_SHAPE(16, 26, CALL) FOR_SHAPE(64, 8, CALL) FOR_SHAPE(64, 16, CALL) FOR_SHAPE(64, 32, CALL) { return fail(PTMI_EUNSUPPORTED, "no kernel shape for ndim=%d", h->cfg.ndim); }
#endif

// am_row0 / swap_last of a launch; a swap iteration may only be the last one of the range
static int set_step_args(const ptmi_engine *h, KArgs *a)
{
    const ptmi_config &c = h->cfg;
    a->am_row0 = (int)(a->iter0 % c.cov_update);
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

extern "C" {

const char *ptmi_last_error(void) { return g_err; }
int ptmi_version(void) { return PTMI_VERSION; }
int ptmi_lanes_for(int ndim) { return ndim <= 104 ? 4 : (ndim <= 416 ? 16 : 64); }

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

int ptmi_create(const ptmi_config *cfg, const ptmi_buffers *buf, ptmi_handle *out)
{
    if (!cfg || !buf || !out) return fail(PTMI_EINVAL, "NULL argument");
    *out = nullptr;
    const ptmi_config &c = *cfg;
    if (c.ndim < 1 || c.ntemps < 1 || c.nwalkers < 1) return fail(PTMI_EINVAL, "ndim/ntemps/nwalkers must be >= 1");
    if (c.ntemps_global < c.ntemps || c.temp0 < 0 || c.temp0 + c.ntemps > c.ntemps_global)
        return fail(PTMI_EINVAL, "temperature block [%d,%d) outside ladder of %d", c.temp0, c.temp0 + c.ntemps, c.ntemps_global);
    if (c.w_host < 0 || c.w_scam < 0 || c.w_am < 0 || c.w_de < 0 || c.w_host + c.w_scam + c.w_am <= 0)
        return fail(PTMI_EINVAL, "No jump proposals specified! (PTMCMCSampler.py:267)");
    if (c.cov_update < 1) return fail(PTMI_EINVAL, "cov_update must be >= 1");
    if (c.w_de > 0 && c.de_size < 2) return fail(PTMI_EINVAL, "de_size must be >= 2 when DE is used");
    if (c.logl_kind < 0 || c.logl_kind > PTMI_LOGL_CURVED || c.logp_kind < 0 || c.logp_kind > PTMI_LOGP_BOX)
        return fail(PTMI_EINVAL, "unknown logl/logp kind");
    if (c.logl_kind == PTMI_LOGL_DENSE && c.logl_par_len != (long long)c.ndim * (c.ndim + 1))
        return fail(PTMI_EINVAL, "dense logl needs mu[d] + Pt[d*d] parameters");
    if (c.logl_kind == PTMI_LOGL_CURVED && (c.ndim & 1)) return fail(PTMI_EINVAL, "curved logl needs an even ndim");
    if (c.logp_kind == PTMI_LOGP_BOX && c.logp_par_len != 2LL * c.ndim) return fail(PTMI_EINVAL, "box prior needs lo[d] + hi[d]");
    if (!c.ladder || !c.temps_mh) return fail(PTMI_EINVAL, "ladder / temps_mh missing");
    if (!buf->X || !buf->lnL || !buf->lp || !buf->temp_of || !buf->slot_of || !buf->Ut || !buf->S || !buf->nacc || !buf->jstat)
        return fail(PTMI_EINVAL, "a required device buffer is NULL");
    if (c.w_de > 0 && !buf->DE) return fail(PTMI_EINVAL, "DE weight > 0 but no DE buffer");
    if ((unsigned long long)c.nwalkers * (unsigned)c.ntemps_global > 0xFFFFFFFFull) return fail(PTMI_EINVAL, "too many RNG streams");
    Shape s;
    if (!pick_shape(c.ndim, &s)) return fail(PTMI_EUNSUPPORTED, "ndim=%d not supported (max 2048)", c.ndim);
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
    int rc;
    if ((rc = upload(&h->d_ladder, c.ladder, c.ntemps_global)) || (rc = upload(&h->d_temps, c.temps_mh, c.ntemps)) ||
        (rc = upload(&h->d_beta, beta.data(), c.ntemps)) || (rc = upload(&h->d_loglpar, c.logl_par, c.logl_par_len)) ||
        (rc = upload(&h->d_logppar, c.logp_par, c.logp_par_len))) {
        ptmi_destroy(h);
        return rc;
    }
    h->cfg.ladder = h->cfg.temps_mh = h->cfg.logl_par = h->cfg.logp_par = nullptr;  // host copies are not kept
    hipError_t e = hipMalloc((void **)&h->d_lnlpos, sizeof(double) * (size_t)c.nwalkers * c.ntemps_global);
    if (e == hipSuccess && !c.cov_per_walker && c.temp0 == 0) {
        const size_t ng = (size_t)(c.nwalkers + POOL_GS - 1) / POOL_GS;
        e = hipMalloc((void **)&h->d_pool_mu, sizeof(double) * ng * c.ndim);
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_pool_M2, sizeof(double) * ng * c.ndim * c.ndim);
    }
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { ptmi_destroy(h); return fail(PTMI_EHIP, "create: %s", hipGetErrorString(e)); }
    *out = h;
    return PTMI_OK;
}

int ptmi_destroy(ptmi_handle h)
{
    if (!h) return PTMI_OK;
    (void)hipFree(h->d_ladder); (void)hipFree(h->d_temps); (void)hipFree(h->d_beta); (void)hipFree(h->d_loglpar); (void)hipFree(h->d_logppar);
    (void)hipFree(h->d_lnlpos); (void)hipFree(h->d_pool_mu); (void)hipFree(h->d_pool_M2);
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
    return PTMI_OK;
}

int ptmi_eval_state(ptmi_handle h)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const KArgs a = make_args(h);
    const int grid = chains_grid(h);
#define CALL_EVAL(G_, E_) launch_eval<G_, E_>(h, a, grid)
    DISPATCH_SHAPE(CALL_EVAL)
    HIPCHK(hipGetLastError());
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
    const bool full = h->cfg.w_am > 0 || (h->de_on && h->cfg.w_de > 0);
    if (!full && h->cfg.w_scam <= 0) return fail(PTMI_EINVAL, "empty proposal cycle");
    const int grid = chains_grid(h);
#define CALL_MH(G_, E_) launch_mh<G_, E_>(h, a, grid, full)
    DISPATCH_SHAPE(CALL_MH)
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_propose(ptmi_handle h, int64_t iter)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!h->buf.Q || !h->buf.qaux) return fail(PTMI_EINVAL, "split path needs the Q and qaux buffers");
    KArgs a = make_args(h);
    a.iter0 = iter; a.nsteps = 1;
    if (int rc = set_step_args(h, &a)) return rc;
    const int grid = chains_grid(h);
#define CALL_PROP(G_, E_) hipLaunchKernelGGL((propose_kernel<G_, E_>), dim3(grid), dim3(256), 0, h->stream, a)
    DISPATCH_SHAPE(CALL_PROP)
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_accept(ptmi_handle h, int64_t iter, const double *newlnL, const double *newlp)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (!h->buf.Q || !h->buf.qaux || !newlnL || !newlp) return fail(PTMI_EINVAL, "split path buffers missing");
    KArgs a = make_args(h);
    a.iter0 = iter; a.nsteps = 1; a.newlnL = newlnL; a.newlp = newlp;
    if (int rc = set_step_args(h, &a)) return rc;
    const int grid = chains_grid(h);
#define CALL_ACC(G_, E_) hipLaunchKernelGGL((accept_kernel<G_, E_>), dim3(grid), dim3(256), 0, h->stream, a)
    DISPATCH_SHAPE(CALL_ACC)
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_swap_write_am(ptmi_handle h, int64_t iter)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    if (h->cfg.temp0 != 0 || !h->buf.AM) return PTMI_OK;
    hipLaunchKernelGGL(am_write_kernel, dim3(h->cfg.nwalkers), dim3(64), 0, h->stream, (const double *)h->buf.X,
                       (const double *)h->buf.lnL, (const double *)h->buf.lp, (const int32_t *)h->buf.slot_of, h->buf.AM,
                       h->buf.AMaux, h->cfg.nwalkers, h->cfg.ntemps, h->cfg.ndim, h->cfg.cov_update, (long long)iter);
    HIPCHK(hipGetLastError());
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
    hipLaunchKernelGGL(swap_sweep_kernel, dim3((W + 63) / 64), dim3(64), 0, h->stream, W, c.ntemps, h->d_ladder,
                       (const double *)nullptr, (const double *)h->buf.lnL, h->buf.slot_of, h->buf.temp_of, (int32_t *)nullptr,
                       (u64 *)h->buf.nswap, 0, c.ntemps, (long long)iter, c.seed, c.walker0);
    HIPCHK(hipGetLastError());
    return ptmi_swap_write_am(h, iter);
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

int ptmi_swap_sweep(ptmi_handle h, int64_t iter, const double *lnL_pos_global, int32_t *map)
{
    if (!h || !lnL_pos_global || !map) return fail(PTMI_EINVAL, "NULL argument");
    if (!h->buf.nswap) return fail(PTMI_EINVAL, "nswap buffer missing");
    const ptmi_config &c = h->cfg;
    const int W = c.nwalkers;
    hipLaunchKernelGGL(swap_sweep_kernel, dim3((W + 63) / 64), dim3(64), 0, h->stream, W, c.ntemps_global, h->d_ladder,
                       lnL_pos_global, (const double *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, map, (u64 *)h->buf.nswap,
                       c.temp0, c.ntemps, (long long)iter, c.seed, c.walker0);
    HIPCHK(hipGetLastError());
    return PTMI_OK;
}

int ptmi_update_cov(ptmi_handle h, int64_t iter)
{
    if (!h) return fail(PTMI_EINVAL, "NULL handle");
    const ptmi_config &c = h->cfg;
    if (c.temp0 != 0) return PTMI_OK;   // only the GPU holding rank 0 adapts (PT:545)
    if (!h->buf.AM || !h->buf.mu || !h->buf.M2 || !h->buf.cov) return fail(PTMI_EINVAL, "AM/mu/M2/cov buffers missing");
    if (iter < c.cov_update || iter % c.cov_update) return fail(PTMI_EINVAL, "iter must be a positive multiple of cov_update");
    const int d = c.ndim, nt = (d + WTILE - 1) / WTILE;
    const int per = c.cov_per_walker;
    if (per)
        hipLaunchKernelGGL(welford_kernel<false>, dim3(nt, nt, c.nwalkers), dim3(256), 0, h->stream, (const double *)h->buf.AM,
                           h->buf.mu, h->buf.M2, h->buf.cov, d, c.cov_update, (long long)iter, d * d);
    else
        hipLaunchKernelGGL(welford_kernel<true>, dim3(nt, nt, c.nwalkers), dim3(256), 0, h->stream, (const double *)h->buf.AM,
                           h->buf.mu, h->buf.M2, (double *)nullptr, d, c.cov_update, (long long)iter, 0);
    if (nt > 1)
        hipLaunchKernelGGL(welford_mean_kernel, dim3((d + 63) / 64, c.nwalkers), dim3(64), 0, h->stream, (const double *)h->buf.AM,
                           h->buf.mu, d, c.cov_update, (long long)iter);
    if (!per) {
        const int W = c.nwalkers, ng = (W + POOL_GS - 1) / POOL_GS;
        const unsigned gx = (unsigned)(((long long)d * d + 255) / 256);
        const double n_per = (double)iter;
        hipLaunchKernelGGL(pool_combine_kernel, dim3(gx, ng), dim3(256), 0, h->stream, (const double *)h->buf.mu,
                           (const double *)h->buf.M2, h->d_pool_mu, h->d_pool_M2, d, W, POOL_GS, n_per, n_per, 0.0);
        hipLaunchKernelGGL(pool_combine_kernel, dim3(gx, 1), dim3(256), 0, h->stream, (const double *)h->d_pool_mu,
                           (const double *)h->d_pool_M2, (double *)nullptr, h->buf.cov, d, ng, ng, (double)POOL_GS * n_per,
                           (double)(W - (ng - 1) * POOL_GS) * n_per, (double)W * n_per - 1.0);
    }
    HIPCHK(hipGetLastError());
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
                       c.ndim, c.de_size, c.cov_update, h->de_head, c.nwalkers, c.cov_per_walker ? 0 : 1);
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
