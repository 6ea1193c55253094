// ptmi_mh.inc.h -- the per-chain kernel templates (proposals, likelihoods, fused MH steps, split path).
// Included by ptmi_shape.hip once per shape.  Reference behaviour cited as PT:<lines> =
// PTMCMCSampler/PTMCMCSampler.py of nanograv/PTMCMCSampler.
#pragma once
#include <stdlib.h>

#include "ptmi_common.h"

// ------------------------------------------------------------- lane groups
// Contiguous layout (STR = false): the G lanes of a chain are adjacent (a quad for G = 4).
// Strided layout (STR = true, G = 4 only): chain c of a wave owns lanes {c, c+16, c+32, c+48}, gl = lane >> 4.
// That is the C/D layout of v_mfma_f64_16x16x4_f64 (col = lane & 15, row = (lane >> 4) + 4 reg), so a
// table-times-vector product computed on the matrix cores lands directly in the chain's registers (below).
// Pairing orders of the reductions are the same in both layouts, so results are bit-identical.
template <bool STR, int J>
__device__ __forceinline__ u64 grp_bcast(u64 v)
{
    if (STR) return (u64)__shfl((long long)v, (int)(threadIdx.x & 15) + 16 * J, 64);
    return quad_bcast<J>(v);
}
template <bool STR, int J>
__device__ __forceinline__ double grp_bcastf(double v)
{
    if (STR) return __shfl(v, (int)(threadIdx.x & 15) + 16 * J, 64);
    return quad_bcastf<J>(v);
}
template <int G, bool STR>
__device__ __forceinline__ double grp_sum(double p)
{
    if (STR) {
        p = p + __shfl_xor(p, 32, 64);   // gl ^ 2
        p = p + __shfl_xor(p, 16, 64);   // gl ^ 1
        return p;
    }
    return group_sum<G>(p);
}
template <int G, bool STR>
__device__ __forceinline__ bool grp_all(bool ok)
{
    if (STR) {
        const u64 m = __ballot(ok) >> (threadIdx.x & 15);
        return (m & 0x0001000100010001ull) == 0x0001000100010001ull;
    }
    return group_all<G>(ok);
}
template <bool STR>
__device__ __forceinline__ double grp_xor1(double v)
{
    if (STR) return __shfl_xor(v, 16, 64);
    return dppf64<0xB1>(v);
}

// Kernel shapes: (lanes per chain G, register slots per lane EPL); a shape serves
// G*EPL_prev < ndim <= G*EPL, so slots e < safe_slots(G, EPL) hold a valid element on every
// lane for every ndim the shape serves and need no bounds check.
// Shape (4, 25) is EXACT: it serves ndim = 100 only (ptmi_abi.hip pick_shape), so every slot of every lane is valid.
// The interval family (PTMI_LOGL_INTERVAL, a translation unit of its own: PTMI_L == 3) lives in the gradient-jump shapes, which
// serve every ndim up to G*EPL: no slot is exempt there.  (static: the units disagree about this function on purpose.)
static constexpr int safe_slots(int G, int EPL)
{
#if defined(PTMI_L) && PTMI_L == 3
    return 0;
#endif
    return G == 4 ? (EPL == 26 ? 20 : EPL == 25 ? 25 : EPL == 20 ? 14 : EPL == 14 ? 8 : EPL == 8 ? 5 : EPL == 5 ? 2 : 0)
         : G == 16 ? (EPL == 26 ? 13 : EPL == 13 ? 7 : EPL == 7 ? 6 : 0)
         : (EPL == 32 ? 16 : EPL == 16 ? 8 : EPL == 8 ? 6 : 0);
}

// out[e] = sum_k T[k][i] * vec_k  for the caller's elements i = gl + 4 e, all 16 chains of the wave at once, on
// v_mfma_f64_16x16x4_f64 (strided layout).  The instruction is a k-ascending fma chain (tools/mfma_probe.hip),
// i.e. exactly the order of the scalar definition.  T is row-major with leading dimension ld.  PADDED: T has
// 4*ceil(d/4) rows and 16*NT columns, zero filled (the LDS copy); otherwise bounds are checked per lane.
// Must be called by ALL lanes of the wave (the A operand of a matrix instruction comes from every lane).
typedef double ptmi_d4 __attribute__((ext_vector_type(4)));
typedef double ptmi_d2 __attribute__((ext_vector_type(2)));
template <int EPL>
struct MfmaAcc {
    static constexpr int NT = (4 * EPL + 15) / 16;
    ptmi_d4 t[NT];
    __device__ __forceinline__ double at(int e) const { return t[e >> 2][e & 3]; }   // element gl + 4e (compile-time e)
};
// LOWER: T[k][i] = 0 for k < i (the dense likelihood's half table): the matrix tiles above the diagonal are skipped.
template <int EPL, bool PADDED, bool LOWER = false>
__device__ __forceinline__ void mfma_tab_vec(const double *T, int ld, int d, const double (&vec)[EPL], MfmaAcc<EPL> &acc)
{
    constexpr int NT = MfmaAcc<EPL>::NT;
    const int c = (int)(threadIdx.x & 15), g = (int)((threadIdx.x & 63) >> 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc.t[t] = ptmi_d4{0.0, 0.0, 0.0, 0.0};
    // software pipeline of depth one: the table row block of step e+1 is in flight while step e multiplies; the
    // scheduling barrier keeps the compiler from hoisting all 4*EPL*NT/4 loads to the top (register blow-up)
    double cur[NT], nxt[NT];
    auto fetch = [&](int e, double (&dst)[NT]) {
        const int k = 4 * e + g;
        const double *row = T + (size_t)k * ld + c;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (LOWER && 16 * t > 4 * e + 3) continue;      // rows 4e .. 4e+3 of this tile are all zero
            if (PADDED) dst[t] = row[16 * t];
            else dst[t] = (k < d && 16 * t + c < d) ? row[16 * t] : 0.0;
        }
    };
    fetch(0, cur);
    constexpr bool EXACT = safe_slots(4, EPL) == EPL;      // ndim = 4 EPL: every k-step exists, the product is one straight-line block
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        if (EXACT || 4 * e < d) {                          // wave-uniform
            if (e + 1 < EPL && (EXACT || 4 * (e + 1) < d)) fetch(e + 1, nxt);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!LOWER || 16 * t <= 4 * e + 3) acc.t[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[t], vec[e], acc.t[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// The same product for the padded half table (PADDED, LOWER) with the vector operand FORMED where it is used (vec(e): e.g. the
// residual (x + dq) - mu of a proposal): a caller that holds x and dq does not keep a third row-sized array alive across the product.
#ifndef PTMI_HALF_PF
#define PTMI_HALF_PF 2       // measured (dense default mix, 100 steps): 1: 14.58, 2: 14.45, 3: 14.55, 4: 14.77, 6: 15.1, 10: 15.2 ms; the one-step pipeline before: 15.3
#endif
constexpr int half_tab_pairs(int EPL) { int n = 0; for (int e = 0; e < EPL; ++e) n += (4 * e + 3) / 16 + 1; return n; }
template <int EPL, class VF>
__device__ __forceinline__ void mfma_half_tab_vecf(const double *T, int ld, int d, VF vec, MfmaAcc<EPL> &acc)
{
    constexpr int NT = MfmaAcc<EPL>::NT;
    const int c = (int)(threadIdx.x & 15), g = (int)((threadIdx.x & 63) >> 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc.t[t] = ptmi_d4{0.0, 0.0, 0.0, 0.0};
    constexpr bool EXACT = safe_slots(4, EPL) == EPL;
    if constexpr (EXACT) {
        // ndim = 4 EPL: every k-step exists, the product is ONE straight-line block, and the table operand of matrix instruction
        // i + PF is requested before instruction i is issued (mh_dense_scam_kernel's scheme: with a one-step software pipeline every
        // k-step waited for the LDS round trip of its own operands; the scheduling barriers pin the order)
        constexpr int NP = half_tab_pairs(EPL), PF = PTMI_HALF_PF;
        double av[NP];
        int pe = 0, pt = 0, pi = 0;                            // cursor of the requests (compile-time after unrolling)
        auto request = [&]() {
            av[pi] = T[(size_t)(4 * pe + g) * ld + c + 16 * pt];
            ++pi;
            if (16 * (pt + 1) <= 4 * pe + 3) ++pt;
            else { pt = 0; ++pe; }
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) request();
        __builtin_amdgcn_sched_barrier(0);
        int i = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const double ve = vec(e);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (16 * t > 4 * e + 3) continue;
                if (pi < NP) request();
                acc.t[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], ve, acc.t[t], 0, 0, 0);
                ++i;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    double cur[NT], nxt[NT];
    auto fetch = [&](int e, double (&dst)[NT]) {
        const double *row = T + (size_t)(4 * e + g) * ld + c;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (16 * t <= 4 * e + 3) dst[t] = row[16 * t];
    };
    fetch(0, cur);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        if (4 * e < d) {                                       // wave-uniform
            if (e + 1 < EPL && 4 * (e + 1) < d) fetch(e + 1, nxt);
            const double ve = vec(e);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (16 * t <= 4 * e + 3) acc.t[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[t], ve, acc.t[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
constexpr int mfma_ld(int EPL) { return 16 * ((4 * EPL + 15) / 16); }

// ----------------------------------------------------------- log-likelihoods
// PTMI_LOGL_INTERVAL, one element (oracle: interval_elem; the reference: tests/test_nuts.py:71-140 backward / logjacobian_grad / dxdp /
// lnlikefn_grad around GaussianLikelihood.lnlikefn_grad :22-25, operation for operation): value term and, GRAD, d/dp of it
template <bool GRAD>
__device__ __forceinline__ double interval_elem(double p, double lo, double w, double lw, double &g)
{
    const double E = det_exp(p), onepe = 1.0 + E;
    const double wE = w * E;
    const double x = wE / onepe + lo;                                   // backward (:81-86)
    const double t = (-0.5 * (x * x) - 0x1.d67f1c864beb5p-1) + ((lw + p) - 2.0 * det_log(onepe));     // :19-20 per element + :88-90
    if (GRAD) {
        const double dxdp = wE / (onepe * onepe);                       // :96-101
        g = (-x) * dxdp + (1.0 - E) / onepe;                            // :117-122: ll_grad * dxdp + lj_grad (:92-93)
    }
    return t;
}

// All G lanes of a group hold q[e] = element (gl + G*e); pad elements are 0.
// STR: strided lane layout; with it the dense product runs on the matrix cores from the LDS copy of Pt.
template <int G, int EPL, int LOGL, bool STR>
__device__ __forceinline__ double eval_logl(const KArgs &a, const double (&q)[EPL], int gl, const double *Pt)
{
    const int d = a.d;
    if (LOGL == PTMI_LOGL_ISO) {
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p = __builtin_fma(q[e], q[e], p);
        return -0.5 * grp_sum<G, STR>(p);
    } else if (LOGL == PTMI_LOGL_DENSE) {
        const double *mu = a.logl_par;
        double r[EPL], v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            r[e] = i < d ? q[e] - mu[i] : 0.0;
            v[e] = 0.0;
        }
        // Pt here is the HALF table Tl (k >= i; diagonal halved): -r^T P r / 2 = -sum_i r_i sum_{k >= i} Tl[k][i] r_k
        if (STR) {
            MfmaAcc<EPL> acc;
            mfma_tab_vec<EPL, true, true>(Pt, mfma_ld(EPL), d, r, acc);
            double p = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) p = __builtin_fma(r[e], acc.at(e), p);
            return -grp_sum<G, STR>(p);
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
        return -grp_sum<G, STR>(p);
    } else if (LOGL == PTMI_LOGL_INTERVAL) {
        const double *par = a.logl_par;
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e, ii = i < d ? i : 0;
            double g;
            const double t = interval_elem<false>(q[e], par[ii], par[d + ii], par[2 * d + ii], g);
            p = __builtin_fma(i < d ? t : 0.0, 1.0, p);
        }
        return grp_sum<G, STR>(p);
    } else {  // PTMI_LOGL_CURVED: pairs (2m, 2m+1); G is even so a pair lives in lanes (gl, gl+1) of one slot
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            const double y = grp_xor1<STR>(q[e]);  // partner lane (gl ^ 1)
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
        return grp_sum<G, STR>(p);
    }
}

// Box prior (the reference's usual lnpriorfn: -inf outside [pmin, pmax]): true when every element of the row lies inside.
// The bounds are read from memory in every step (2*EPL registers to keep them are not there), CH slots at a time: the
// requests of a chunk go out together and are compared when they arrive.  The empty asm between chunks pins that shape.
// Left to itself the compiler either turned a short-circuit chain into 2*EPL dependent branches, each waiting for its
// own memory round trip, or hoisted all the loads out of the step loop and spilled them.
template <int G, int EPL, class QF>
__device__ __forceinline__ bool box_inside(const double *lo_, const double *hi_, int d, int gl, QF q)
{
    typedef const __attribute__((address_space(1))) double *gptr;
    constexpr int CH = 7;
    gptr lo = (gptr)lo_, hi = (gptr)hi_;
    int ok = 1;
#pragma unroll
    for (int e0 = 0; e0 < EPL; e0 += CH) {
        asm volatile("" : "+s"(lo), "+s"(hi), "+v"(ok));
        double l[CH], h[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = e0 + c, i = gl + G * e;
            if (e >= EPL) break;
            const int ii = i < d ? i : 0;       // padding slots read element 0 and are ignored (no slot is exempt: gradient-jump
                                               // engines run a shape below its usual ndim range)
            l[c] = lo[ii];
            h[c] = hi[ii];
        }
        bool in = true;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = e0 + c, i = gl + G * e;
            if (e >= EPL) break;
            const double qe = q(e);
            const bool in1 = (l[c] <= qe) & (h[c] >= qe);
            in &= in1 | (i >= d);
        }
        ok &= (int)in;
    }
    return ok != 0;
}

// The same test against the block's LDS copy of the bounds (box_table_fill): {lo, hi} pairs in lane order, one 16-byte
// read per slot, padding slots hold {-inf, +inf}.  A vector-memory read costs the CU's address unit 8+ cycles per wave
// and the test needs 2*EPL of them per step -- more than the rest of a SCAM step; the LDS pipe does it in a quarter.
template <int G, int EPL, class QF>
__device__ __forceinline__ bool box_inside_lds(const double *smem, int off, int gl, QF q)
{
    constexpr int CH = 7;
    int base = off + 2 * EPL * gl, ok = 1;
#pragma unroll
    for (int e0 = 0; e0 < EPL; e0 += CH) {
        asm volatile("" : "+v"(base), "+v"(ok));              // chunk boundary (see box_inside)
        const ptmi_d2 *tab = (const ptmi_d2 *)(smem + base);
        ptmi_d2 b[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (e0 + c < EPL) b[c] = tab[e0 + c];
        bool in = true;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (e0 + c >= EPL) break;
            const double qe = q(e0 + c);
            in &= (b[c].x <= qe) & (b[c].y >= qe);
        }
        ok &= (int)in;
    }
    return ok != 0;
}
// One pass over the block's bounds table for a SCAM step of a box prior's fast path (mh_steps_kernel, BOXFAST): is q = x + dq inside,
// and how far is x from the nearest bound (the lane's elements; padding slots hold {-inf, +inf})?
template <int G, int EPL>
__device__ __forceinline__ void box_test_and_margin(const double *smem, int off, int gl, const double (&x)[EPL], const double (&dq)[EPL],
                                                    bool &inside, double &margin, double &bmax)
{
    const ptmi_d2 *tab = (const ptmi_d2 *)(smem + off + 2 * EPL * gl);
    bool in = true;
    double mg = __builtin_inf(), bm = 0.0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const ptmi_d2 b = tab[e];
        const double qe = x[e] + dq[e];
        in &= (b.x <= qe) & (b.y >= qe);
        const double lo_gap = x[e] - b.x, hi_gap = b.y - x[e];
        const double g2 = lo_gap < hi_gap ? lo_gap : hi_gap;           // (a NaN gap -- x outside every order -- loses every comparison below: margin stays, the test decides)
        mg = g2 < mg ? g2 : mg;
        // the largest finite |bound| of the lane's elements (padding slots hold infinities): what an element inside the box can be at most
        const double al = __builtin_fabs(b.x), ah = __builtin_fabs(b.y);
        bm = (al < __builtin_inf() && al > bm) ? al : bm;
        bm = (ah < __builtin_inf() && ah > bm) ? ah : bm;
    }
    inside = in;
    margin = mg;
    bmax = bm;
}
template <int G, bool STR>
__device__ __forceinline__ double grp_min(double v)
{
    static_assert(!STR, "contiguous layout");
    if (G >= 64) { const double o = __shfl_xor(v, 32, 64); v = o < v ? o : v; }
    if (G >= 32) { const double o = __shfl_xor(v, 16, 64); v = o < v ? o : v; }
    if (G >= 16) { const double o = dppf64<0x128>(v); v = o < v ? o : v; }
    if (G >= 8) { const double o = dppf64<0x124>(v); v = o < v ? o : v; }
    if (G >= 4) { const double o = dppf64<0x4E>(v); v = o < v ? o : v; }
    if (G >= 2) { const double o = dppf64<0xB1>(v); v = o < v ? o : v; }
    return v;
}
constexpr int box_table_doubles(int G, int EPL) { return 2 * G * EPL; }
// Prologue of the step kernels; the caller's __syncthreads follows.
template <int G, int EPL>
__device__ __forceinline__ void box_table_fill(const KArgs &a, double *smem, int nthreads)
{
    if (a.logp_kind != PTMI_LOGP_BOX || a.box_off < 0) return;
    const double *lo = a.logp_par, *hi = a.logp_par + a.d;
    for (int i = (int)threadIdx.x; i < G * EPL; i += nthreads) {
        const int p = i / EPL + G * (i % EPL);
        smem[a.box_off + 2 * i] = p < a.d ? lo[p] : -__builtin_inf();
        smem[a.box_off + 2 * i + 1] = p < a.d ? hi[p] : __builtin_inf();
    }
}

// PT:605-606 for the built-in priors; smem: the block's dynamic LDS (bounds table at a.box_off) or nullptr
template <int G, int EPL, bool STR, class QF>
__device__ __forceinline__ double eval_logp_q(const KArgs &a, const double *smem, int gl, QF q)
{
    if (a.logp_kind == PTMI_LOGP_BOX) {
        bool ok;
        if (smem != nullptr && a.box_off >= 0) ok = box_inside_lds<G, EPL>(smem, a.box_off, gl, q);
        else ok = box_inside<G, EPL>(a.logp_par, a.logp_par + a.d, a.d, gl, q);
        return grp_all<G, STR>(ok) ? 0.0 : -__builtin_inf();
    }
    return 0.0;
}
template <int G, int EPL, bool STR>
__device__ __forceinline__ double eval_logp(const KArgs &a, const double (&q)[EPL], int gl, const double *smem = nullptr)
{
    return eval_logp_q<G, EPL, STR>(a, smem, gl, [&](int e) { return q[e]; });
}

// ---------------------------------------------------------------- proposals
// Per-chain constants of the jump scales, hoisted out of the step loop.  Same operation
// order as the reference: scale in {10, 0.2, 1.0}; scale *= sqrt(temp) if temp <= 100
// (PT:846-862); cd = 2.4 / sqrt(2 neff) * scale (PT:870, 928).
struct ChainConst {
    // by scale branch (prob > 0.97, prob > 0.9, else): SCAM's cd and the bare scale (AM: cd = 2.4/sqrt(2 n_g) * sc).
    // Scalars on purpose: arrays selected by the branch index were lowered to a table in scratch memory.
    double cd0, cd1, cd2, sc0, sc1, sc2;
    double de_mul;                 // DE: rr * 2.4 / sqrt(2 n_g) * de_mul  (PT:976)
    // every member is read before the selects: a conditional read would be folded into a read through a selected address
    static __device__ __forceinline__ double pick3(int br, double v0, double v1, double v2)
    {
        double r = v2;
        r = br == 1 ? v1 : r;
        r = br == 0 ? v0 : r;
        return r;
    }
    __device__ __forceinline__ double cd_scam(int br) const { return pick3(br, cd0, cd1, cd2); }
    __device__ __forceinline__ double sc(int br) const { return pick3(br, sc0, sc1, sc2); }
};
__device__ __forceinline__ ChainConst chain_const(double temp, double beta, int d)
{
    ChainConst c;
    const bool warm = temp <= 100.0;
    const double sT = warm ? det_sqrt(temp) : 1.0;
    const double c1 = 2.4 / det_sqrt(2.0 * 1.0);
    c.sc0 = warm ? 10.0 * sT : 10.0;
    c.sc1 = warm ? 0.2 * sT : 0.2;
    c.sc2 = warm ? 1.0 * sT : 1.0;
    c.cd0 = c1 * c.sc0;
    c.cd1 = c1 * c.sc1;
    c.cd2 = c1 * c.sc2;
    c.de_mul = det_sqrt(1.0 / beta);
    return c;
}

// element i = gl + G*e of a table row
#define PTMI_ROW_LOAD_S(SAFE, dst, row, e)                                 \
    do {                                                                   \
        if ((e) < (SAFE)) dst = (row)[gl + G * (e)];                       \
        else dst = (gl + G * (e)) < d ? (row)[gl + G * (e)] : 0.0;         \
    } while (0)
#define PTMI_ROW_LOAD(dst, row, e) PTMI_ROW_LOAD_S(safe_slots(G, EPL), dst, row, e)

// ------------------------------------------------------------------ draws
// RNG schedule (oracle/ptmcmc_oracle.c "slot numbers"): an iteration of a chain consumes Philox slots
// 0 (P) and 1 (Q) of its rank's stream.  The first four lanes of a chain evaluate them for TWO consecutive iterations in
// one instruction stream: lane j -> slot (j & 1) of iteration it + (j >> 1); the same pass takes the logarithms both
// iterations need (accept uniform on the even lanes, Box-Muller radius on the odd ones) and finishes the SCAM normal on
// the odd lanes.  A step reads its values from lanes 0 / 1; after the first of the two steps the batch is rotated by two
// lanes.  With pick_mode WALKER the word that picks the cycle entry comes from slot 0 of the stream of the walker's
// rank 0, evaluated for four iterations at a time (lane j -> iteration it + j) and rotated by one lane per step.
struct Draws {
    u64 P0, Q0, Q1;       // P0 = [pick | scale-branch uniform]; Q0, Q1: SCAM (u1 | direction, angle) or DE (rows | scale)
    double log_u;         // log(accept uniform)
    double z;             // the SCAM normal (PT:873)
    u32 pickw;            // the word that picks the cycle entry
};
__device__ __forceinline__ u32 h2index(u32 h, u32 n) { return __umulhi(h, n); }
__device__ __forceinline__ double h2uniform(u32 h) { return (double)h * 0x1.0p-32; }
// WIDE draw batches (GW = 16 or 64 lanes per chain, contiguous layout).  The draws depend on the stream alone, so ALL GW lanes of
// a chain take part in a draw pass: lane j -> slot j & 1 of iteration it + (j >> 1), i.e. GW / 2 iterations per pass (32 at 64
// lanes) where the four-lane scheme served two whatever GW was -- at GW = 64 sixteen quads computed the same four Philox / log /
// Box-Muller results sixteen times over, every second step.  Step s of the pass reads its values from lanes 2 s (P word, log u)
// and 2 s + 1 (Q words, normal / amplitude, direction) of the chain: v_readlane at 64 lanes (the chain is the wave, the lane
// index a scalar), ds_bpermute at 16.  Same operations on the same values: bit-identical.
template <int GW>
__device__ __forceinline__ u32 chain_lane32(u32 v, int j)        // lane j (wave-uniform) of the caller's chain
{
    if constexpr (GW == 64) return (u32)__builtin_amdgcn_readlane((int)v, j);
    else return (u32)__builtin_amdgcn_ds_bpermute((int)((((threadIdx.x & 63) & ~(unsigned)(GW - 1)) + (unsigned)j) << 2), (int)v);
}
template <int GW>
__device__ __forceinline__ u64 chain_lane64(u64 v, int j)
{
    return ((u64)chain_lane32<GW>((u32)(v >> 32), j) << 32) | chain_lane32<GW>((u32)v, j);
}
template <int GW>
__device__ __forceinline__ double chain_lanef(double v, int j)
{
    return __longlong_as_double((long long)chain_lane64<GW>((u64)__double_as_longlong(v), j));
}
template <bool STR, int GW = 4 /* > 4: wide batch over all GW lanes of the chain */>
struct DrawBatch {
    static_assert(GW == 4 || !STR, "wide batches serve the contiguous layout");
    u64 w0, w1;           // this lane's Philox words
    double lg, z;         // log of this lane's uniform; odd lanes: the SCAM normal
    u32 pw;               // pick_mode WALKER: this lane's pick word
    template <typename T>
    static __device__ __forceinline__ T rot(T v, int by)   // value of the chain's lane (gl + by) & 3
    {
        if constexpr (STR) return __shfl(v, (int)((threadIdx.x + 16 * by) & 63), 64);
        else if constexpr (sizeof(T) == 8) return (T)(by == 2 ? dpp64<0x4E>((u64)v) : dpp64<0x39>((u64)v));   // quad_perm [2,3,0,1] / [1,2,3,0]
        else return (T)(by == 2 ? dpp32<0x4E>((u32)v) : dpp32<0x39>((u32)v));
    }
    // TM: where the draw tables are read (draw_table); tsm: the block's LDS, tables at a.tab_off
    template <int TM>
    __device__ __forceinline__ void refill(const KArgs &a, long long it, u32 sid, int gl, const double *tsm)
    {
        const int j = GW > 4 ? gl : (gl & 3);
        philox_words(a.seed, (u64)(it + (j >> 1)), sid, (u32)(j & 1), w0, w1);
        u32 aj;
        double at;
        unit_angle32((u32)w1, aj, at);
        if constexpr (TM == 0) {
            // global tables: both reads first, what does not need them behind (a read takes a few hundred cycles;
            // config-2 kernel 1.145 -> 1.105 ms per 100 steps)
            const UnitLogArg g = unit_log_arg((j & 1) ? w0 : w1);   // both uniforms are (0,1] ones
            double sb, cb;
            const ptmi_dev_d2 te = draw_table<0>(nullptr, -1, g.slice), tb = draw_table<0>(nullptr, -1, 32u + aj);
            unit_rotation(at, sb, cb);
            lg = unit_log_finish(g, te);
            z = det_sqrt(-2.0 * lg) * unit_cos_finish(tb, sb, cb);  // meaningful on the odd lanes
        } else {
            double sn, cs;
            lg = unit_log<TM>((j & 1) ? w0 : w1, tsm, a.tab_off);
            unit_sincos<TM>(aj, at, sn, cs, tsm, a.tab_off);
            z = det_sqrt(-2.0 * lg) * cs;
        }
    }
    __device__ __forceinline__ void advance()
    {
        w0 = rot(w0, 2); w1 = rot(w1, 2);
        lg = __longlong_as_double((long long)rot((u64)__double_as_longlong(lg), 2));
        z = __longlong_as_double((long long)rot((u64)__double_as_longlong(z), 2));
    }
    __device__ __forceinline__ void refill_pick(const KArgs &a, long long it, u32 sid0, int gl)
    {
        u64 p0, p1;
        philox_words(a.seed, (u64)(it + (GW > 4 ? gl : (gl & 3))), sid0, 0u, p0, p1);
        pw = (u32)(p0 >> 32);
    }
    __device__ __forceinline__ void advance_pick() { pw = rot(pw, 1); }
    __device__ __forceinline__ u64 P1() const { return grp_bcast<STR, 0>(w1); }   // accept-uniform word (split path)
    __device__ __forceinline__ void take(Draws &d, bool walker) const
    {
        d.P0 = grp_bcast<STR, 0>(w0);
        d.log_u = grp_bcastf<STR, 0>(lg);
        d.Q0 = grp_bcast<STR, 1>(w0);
        d.Q1 = grp_bcast<STR, 1>(w1);
        d.z = grp_bcastf<STR, 1>(z);
        d.pickw = walker ? (u32)grp_bcast<STR, 0>((u64)pw) : (u32)(d.P0 >> 32);
    }
    // wide batch: step s of the pass (lanes 2 s, 2 s + 1), the walker's pick word of step sp of ITS pass (lane sp)
    __device__ __forceinline__ void take_wide(Draws &d, bool walker, int s, int sp) const
    {
        d.P0 = chain_lane64<GW>(w0, 2 * s);
        d.log_u = chain_lanef<GW>(lg, 2 * s);
        d.Q0 = chain_lane64<GW>(w0, 2 * s + 1);
        d.Q1 = chain_lane64<GW>(w1, 2 * s + 1);
        d.z = chain_lanef<GW>(z, 2 * s + 1);
        d.pickw = walker ? chain_lane32<GW>(pw, sp) : (u32)(d.P0 >> 32);
    }
};
// Steps k = 0, 1, ... of a launch that starts at iteration iter0: keep the batch current for step k.
template <bool STR, bool FULL, int TM = 0, int GW = 4>
__device__ __forceinline__ void draws_for_step(DrawBatch<STR, GW> &b, Draws &dr, const KArgs &a, int k, u32 sid, u32 sid0, int gl,
                                               const double *tsm = nullptr)
{
    const bool walker = FULL && a.pick_walker;
    if constexpr (GW > 4) {                            // the caller refills at the head of every pass of GW / 2 steps (mh_steps_kernel)
        const int s = k & (GW / 2 - 1), sp = k & (GW - 1);
        if (walker && sp == 0) b.refill_pick(a, a.iter0 + k, sid0, gl);
        b.take_wide(dr, walker, s, sp);
        return;
    }
    if ((k & 1) == 0) b.template refill<TM>(a, a.iter0 + k, sid, gl, tsm);
    else b.advance();
    if (walker) {
        if ((k & 3) == 0) b.refill_pick(a, a.iter0 + k, sid0, gl);
        else b.advance_pick();
    }
    b.take(dr, walker);
}

// SCAM-only cycles with one parameter group (the config-2 / config-3 bench kernels): everything of the proposal that is a
// scalar of the chain -- scale branch, eigen-direction k, amplitude z cd sqrt(S_k) (PT:843-873) -- is computed IN the draw
// pass, where the four lanes of a chain work on different (iteration, slot) pairs, instead of four times over in the
// step.  A step then takes three values from the batch (log u from the chain's lane 0; amplitude and direction from lane 1)
// with five lane moves, and after the first of the two steps the batch rotates by two lanes.  Same operations in the same
// order as propose(): bit-identical.
struct ScamDraw { double log_u, amp; int k; };
template <bool STR, int GW = 4 /* > 4: wide batch over all GW lanes of the chain */>
struct ScamBatch {
    static_assert(GW == 4 || !STR, "wide batches serve the contiguous layout");
    double lg, amp;       // log of this lane's uniform (even lanes: the accept test's); odd lanes: the jump amplitude
    int kdir;             // odd lanes: the eigen-direction
    static __device__ __forceinline__ double rot2(double v)  // value of the chain's lane (gl + 2) & 3
    {
        if constexpr (STR) return __shfl(v, (int)((threadIdx.x + 32) & 63), 64);
        else return dppf64<0x4E>(v);                         // quad_perm [2,3,0,1]
    }
    static __device__ __forceinline__ int rot2(int v)
    {
        if constexpr (STR) return __shfl(v, (int)((threadIdx.x + 32) & 63), 64);
        else return (int)dpp32<0x4E>((u32)v);
    }
    // root_s(k): sqrt of the k-th eigenvalue of the chain's table; ng: directions to pick from
    // TM / tsm: where the draw tables are read (draw_table): 0 = global memory, 2 = the block's LDS copy when a.tab_off >= 0
    template <int TM, class RS>
    __device__ __forceinline__ void refill(const KArgs &a, long long it, u32 sid, int gl, const ChainConst &cc, int ng, RS root_s, const double *tsm)
    {
        const int j = GW > 4 ? gl : (gl & 3);
        u64 w0, w1;
        philox_words(a.seed, (u64)(it + (j >> 1)), sid, (u32)(j & 1), w0, w1);
        u32 aj;
        double at;
        unit_angle32((u32)w1, aj, at);
        const UnitLogArg g = unit_log_arg((j & 1) ? w0 : w1);       // both uniforms are (0,1] ones
        double sb, cb;
        const ptmi_dev_d2 te = draw_table<TM>(tsm, a.tab_off, g.slice), tb = draw_table<TM>(tsm, a.tab_off, 32u + aj);   // both reads first
        kdir = (int)h2index((u32)(w1 >> 32), (u32)ng);              // PT:868 (odd lanes)
        const double rs = root_s(kdir);
        // the scale branch comes from the iteration's P word, which the chain's lane j - 1 holds (PT:843-858)
        u32 plo;
        if constexpr (STR) plo = (u32)__shfl((int)(u32)w0, (int)((threadIdx.x - 16) & 63), 64);
        else plo = dpp32<0xA0>((u32)w0);                            // quad_perm [0,0,2,2]
        constexpr u32 T97 = (u32)(0.97 * 4294967296.0), T90 = (u32)(0.9 * 4294967296.0);
        const int br = plo > T97 ? 0 : (plo > T90 ? 1 : 2);
        unit_rotation(at, sb, cb);
        lg = unit_log_finish(g, te);
        const double z = det_sqrt(-2.0 * lg) * unit_cos_finish(tb, sb, cb);
        amp = z * cc.cd_scam(br) * rs;                              // PT:873
    }
    __device__ __forceinline__ void advance()
    {
        lg = rot2(lg); amp = rot2(amp); kdir = rot2(kdir);
    }
    __device__ __forceinline__ void take(ScamDraw &d) const
    {
        d.log_u = grp_bcastf<STR, 0>(lg);
        d.amp = grp_bcastf<STR, 1>(amp);
        if constexpr (STR) d.k = __shfl(kdir, (int)(threadIdx.x & 15) + 16, 64);
        else d.k = (int)dpp32<0x55>((u32)kdir);
    }
    __device__ __forceinline__ void take_wide(ScamDraw &d, int s) const      // step s of the pass
    {
        d.log_u = chain_lanef<GW>(lg, 2 * s);
        d.amp = chain_lanef<GW>(amp, 2 * s + 1);
        d.k = (int)chain_lane32<GW>((u32)kdir, 2 * s + 1);
    }
};
template <bool STR, int TM = 0, int GW = 4, class RS>
__device__ __forceinline__ void scam_draws_for_step(ScamBatch<STR, GW> &b, ScamDraw &dr, const KArgs &a, int k, u32 sid, int gl,
                                                    const ChainConst &cc, int ng, RS root_s, const double *tsm = nullptr)
{
    if constexpr (GW > 4) {                            // the caller refills at the head of every pass of GW / 2 steps (mh_steps_kernel)
        b.take_wide(dr, k & (GW / 2 - 1));
        return;
    }
    if ((k & 1) == 0) b.template refill<TM>(a, a.iter0 + k, sid, gl, cc, ng, root_s, tsm);
    else b.advance();
    b.take(dr);
}

// One matrix instruction whose accumulator is pinned to the accumulation registers.  With __builtin_amdgcn_mfma_* in a
// ROLLED loop hipcc keeps the loop-carried accumulators in VGPRs and moves all of them to AGPRs and back on every trip
// (2 x 56 v_accvgpr moves per Box-Muller pair); an asm operand of class "a" stays where it is.  hipcc pads no hazards
// of an asm statement: the chain through C needs no wait state, the first vector read of the result does (mfma_acc_settle).
// GUARD: an operand may have been written by the vector pipe just before (the weight ahead of the first product of a
// k-step; a table value selected against its bounds): two wait states inside the statement.
// ACCV: the accumulators in ordinary vector registers ("+v": gfx950 takes either file as C / D).  mh_pc_kernel asks for it: a
// kernel that pins accumulation registers is given HALF its budget as such (128 + 128 at two waves per SIMD), and its stepper
// waves, which hold three rows and use no accumulator, then shuttle their state through the other half.
template <bool GUARD, bool ACCV = false>
__device__ __forceinline__ void mfma_f64_acc(ptmi_d4 &acc, double ta, double w)
{
    if constexpr (ACCV) {
        if (GUARD) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(ta), "v"(w));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(ta), "v"(w));
    } else {
        if (GUARD) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(ta), "v"(w));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(ta), "v"(w));
    }
}
template <int NT, bool ACCV = false>
__device__ __forceinline__ void mfma_acc_begin(ptmi_d4 (&t)[NT])
{
#pragma unroll
    for (int i = 0; i < NT; ++i) t[i] = ptmi_d4{0.0, 0.0, 0.0, 0.0};
    // the "+a" pins are where the zeros are materialised (v_accvgpr_write); the wait states between such a write and a
    // matrix instruction that reads the register as C go BEHIND them (volatile statements keep their order)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if constexpr (ACCV) asm volatile("" : "+v"(t[i]));
        else asm volatile("" : "+a"(t[i]));
    }
    asm volatile("s_nop 7");
}
template <int NT, bool ACCV = false>
__device__ __forceinline__ void mfma_acc_settle(ptmi_d4 (&t)[NT])
{
    asm volatile("s_nop 15\n\ts_nop 15");              // 16-pass matrix instruction -> vector read of its result (18 states needed)
#pragma unroll
    for (int i = 0; i < NT; ++i) {                      // every read of the result is behind the wait
        if constexpr (ACCV) asm volatile("" : "+v"(t[i]));
        else asm volatile("" : "+a"(t[i]));
    }
}

// The AM increment U (cd sqrt(S) z) (PT:879-933) of the 16 chains in the columns of the wave, on the matrix cores (strided
// layout: lane (c16, g4) draws the weights of directions k = g4 + 4e of column c16's chain).  The weights of two k-steps come
// out of one Box-Muller and go straight into the two accumulation steps -- no weight array is kept.  Nothing overlaps an
// f64 matrix instruction on its SIMD (tools/inst_rates.hip: 64 cycles + the full issue cost of whatever sits between two
// of them, integer or double, one wave or two), so the cost of a pair is the plain sum of generator, draw and 2 NT matrix
// instructions and only fewer instructions help: table-driven draws (ptmi_device.h unit_log / unit_sincos) and no
// accumulator traffic.  The accumulation order (k ascending) is that of mfma_tab_vec.
// active / sid / it / cd are the column's: the chain's own (propose) or those of a queued AM event (mh_steps_kernel).
template <int EPL, bool ACCV = false>
__device__ __forceinline__ void am_mfma_product(const KArgs &a, bool active, u32 sid, long long it, double cd, int ng,
                                                const double *Ut, bool ut_padded, int uld, const double *S, bool s_sqrt, MfmaAcc<EPL> &acc,
                                                const double *tsm = nullptr)
{
    constexpr int G = 4, NT = MfmaAcc<EPL>::NT;
    const int d = a.d;
    const int c16 = (int)(threadIdx.x & 15), g4 = (int)((threadIdx.x & 63) >> 4);
    auto root_s = [&](int k) { const double v = S[k]; return s_sqrt ? v : det_sqrt(v); };
    // directions k = g4 + 4e and k + 4 (slots e even / odd) are the cos and sin branches of ONE Box-Muller; straight-line:
    // lanes without a weight draw all the same and select zero
    auto draw_f64 = [&](int e, u64 e0, u64 e1, double &wa, double &wb) {
        const int k = g4 + G * e;
        const bool on = active && k < ng, on2 = on && e + 1 < EPL && k + G < ng;
        // one function after the other: hoisting both table reads (as DrawBatch::refill does for the global tables) costs
        // this loop registers and time (19.7 -> 20.7 ms per 100 AM steps)
        const double r = det_sqrt(-2.0 * (tsm ? unit_log<2>(e0, tsm, a.tab_off) : unit_log<0>(e0)));
        u32 aj;
        double at, sn, cs;
        unit_angle64(e1, aj, at);
        if (tsm) unit_sincos<2>(aj, at, sn, cs, tsm, a.tab_off);
        else unit_sincos<0>(aj, at, sn, cs);
        const double va = (r * cs) * cd * root_s(on ? k : 0);          // PT:930
        const double vb = (r * sn) * cd * root_s(on2 ? k + G : 0);
        wa = on ? va : 0.0;
        wb = on2 ? vb : 0.0;
    };
    auto rows = [&](int k, double (&dst)[NT]) {            // table row block of one k-step
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = 16 * t + c16;
            if (ut_padded) dst[t] = Ut[(size_t)k * uld + col];
            else dst[t] = (k < d && col < d) ? Ut[(size_t)k * uld + col] : 0.0;
        }
    };
    const int esteps = (d + 3) / 4 < EPL ? (d + 3) / 4 : EPL;     // k-steps that hold a table row
    // the NT products of one k-step; the LDS copy's values come straight from the read, the global table's through a select
    auto kstep = [&](const double (&tt)[NT], double w) {
        if (ut_padded) {
            mfma_f64_acc<true, ACCV>(acc.t[0], tt[0], w);
#pragma unroll
            for (int t = 1; t < NT; ++t) mfma_f64_acc<false, ACCV>(acc.t[t], tt[t], w);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) mfma_f64_acc<true, ACCV>(acc.t[t], tt[t], w);
        }
    };
    // -DPTMI_AM_PROFILE: shader-clock counts of the pieces, printed by the first wave (13 pairs at d = 100; measured:
    // generator 4.3 k cycles, double-precision half of the draws 13.0 k before the tables, matrix instructions 12.9 k)
#ifdef PTMI_AM_PROFILE
    unsigned long long tp[5] = {0, 0, 0, 0, 0}, t0, t1;
#define PTMI_STAMP(i) t1 = __builtin_readcyclecounter(); tp[i] += t1 - t0; t0 = t1;
    t0 = __builtin_readcyclecounter();
#else
#define PTMI_STAMP(i)
#endif
    mfma_acc_begin<NT, ACCV>(acc.t);
#pragma unroll 1
    for (int e = 0; e < esteps; e += 2) {
        double ta[NT], tb[NT], wa, wb;
        u64 e0, e1;
        const bool second = e + 1 < esteps;
        rows(4 * e + g4, ta);                              // both row blocks in flight during the draw (LDS table; the dense
        if (ut_padded) rows(4 * (second ? e + 1 : e) + g4, tb);   // kernel reads the global table and has no registers for a second block)
        philox_words(a.seed, (u64)it, sid, SLOT_AM + (u32)(g4 + G * e), e0, e1);
        PTMI_STAMP(0)
        draw_f64(e, e0, e1, wa, wb);
        PTMI_STAMP(1)
        kstep(ta, wa);
        PTMI_STAMP(2)
        if (second) {
            if (!ut_padded) rows(4 * e + 4 + g4, tb);
            kstep(tb, wb);
        }
        PTMI_STAMP(3)
    }
#ifdef PTMI_AM_PROFILE
    mfma_acc_settle<NT, ACCV>(acc.t);
    PTMI_STAMP(4)
    if (blockIdx.x == 0 && threadIdx.x == 0 && it % 64 == 0)
        printf("am profile it %lld: philox+rows %llu  f64 %llu  mfmaA %llu  mfmaB %llu  settle %llu cycles (13 pairs)\n", it, tp[0], tp[1], tp[2], tp[3], tp[4]);
#endif
    mfma_acc_settle<NT, ACCV>(acc.t);
}

// One proposal for the caller's chain (PT:1048-1067, 820-985) from the iteration's draws: writes the increment dq
// (q = x + dq) and returns the jump type.
// STR: strided lane layout (see "lane groups"); then Ut is the zero-padded LDS copy when ut_padded, and the AM
// product runs on the matrix cores for all 16 chains of the wave at once.
// GRP: parameter groups (compile-time: with one group every bound below is the wave-uniform d).
// GJ: the cycle also holds the gradient jumps: such a pick hands the state back unchanged and the caller
// (mh_steps_gj_kernel, ptmi_gj.inc.h) builds the proposal.
template <int G, int EPL, bool FULL, bool STR, bool GRP, bool GJ = false>
__device__ __forceinline__ int propose(const KArgs &a, long long it, u32 sid, int gl, const ChainConst &cc, const Draws &dr,
                                       const double *Ut, bool ut_padded, const double *S, const double *DE,
                                       double (&dq)[EPL], bool s_sqrt = false, bool am_here = true, const double *tsm = nullptr,
                                       long long *am_next = nullptr /* where the chain's next precomputed AM increment is (a.am_inc) */,
                                       const double *de_u = nullptr /* TEST HOOK (propose_kernel): DE's scale uniform as a double */)
{
    // s_sqrt: S holds sqrt(eigenvalue) already (the block's LDS copy; sqrt is correctly rounded, so the bits are the same)
    auto root_s = [&](int k) { const double v = S[k]; return s_sqrt ? v : det_sqrt(v); };
    const int d = a.d;
    // with gradient jumps a shape serves every ndim up to G*EPL (ptmi_lanes_for_grad), so no slot is exempt from the bounds check
    constexpr int PSAFE = GJ ? 0 : safe_slots(G, EPL);
    const int uld = (STR && ut_padded) ? mfma_ld(EPL) : d;   // leading dimension of the Ut table

    int jt = PTMI_J_SCAM;
    if (FULL) {
        const int w_de = a.de_on ? a.w_de : 0;
        const int L = a.w_host + a.w_scam + a.w_am + w_de + (GJ ? a.w_nuts + a.w_hmc : 0);
        const int pick = (int)h2index(dr.pickw, (u32)L);
        const int ind = pick - a.w_host;
        jt = ind < a.w_scam ? PTMI_J_SCAM : (ind < a.w_scam + a.w_am ? PTMI_J_AM : PTMI_J_DE);
        if (GJ && ind >= a.w_scam + a.w_am + w_de) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) dq[e] = 0.0;
            jt = ind < a.w_scam + a.w_am + w_de + a.w_nuts ? PTMI_J_NUTS : PTMI_J_HMC;
        }
        if (ind < 0) {                          // a host-served cycle entry: hand the state back unchanged
#pragma unroll
            for (int e = 0; e < EPL; ++e) dq[e] = 0.0;
            jt = PTMI_J_NTYPES + pick;
        }
    }
    // scale branch (PT:846-858): prob = lo32 * 2^-32 against 0.97, 0.9 and (DE) 0.5, as exact integer thresholds:
    // lo * 2^-32 > c  <=>  lo > floor(c * 2^32)  (c * 2^32 is no integer for 0.97 and 0.9, and exactly 2^31 for 0.5)
    const u32 plo = (u32)dr.P0;
    constexpr u32 T97 = (u32)(0.97 * 4294967296.0), T90 = (u32)(0.9 * 4294967296.0), T50 = 0x80000000u;
    const int br = plo > T97 ? 0 : (plo > T90 ? 1 : 2);
    // parameter group (PT:839,897,955): its own Philox call, only with more than one group.
    // A group's eigenvectors are embedded in the full space, so the jumps below only change the table they read.
    int g = 0, ng = d;
    if (GRP) {
        u64 g0, g1;
        philox_words(a.seed, (u64)it, sid, 2u, g0, g1);
        g = (int)h2index((u32)(g0 >> 32), (u32)a.ngroups);
        ng = a.gsize[g];
        Ut += (size_t)g * d * d;
        S += (size_t)g * d;
    }

    auto scam_body = [&]() {
        const int k = (int)h2index((u32)(dr.Q1 >> 32), (u32)ng);
        const double *col = Ut + (size_t)k * uld;
        // the direction lands in dq and is scaled in place
        if (G > 4 && !GRP && !GJ && a.UtPad != nullptr) {          // the library's zero-padded copy of the launch's one table
            const double *colp = a.UtPad + (size_t)k * a.ut_pad_ld;
#pragma unroll
            for (int e = 0; e < EPL; ++e) dq[e] = colp[gl + G * e];
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD_S(PSAFE, dq[e], col, e);
        }
        const double amp = dr.z * cc.cd_scam(br) * root_s(k);                // PT:873
#pragma unroll
        for (int e = 0; e < EPL; ++e) dq[e] = amp * dq[e];
    };
    auto de_body = [&]() {
        const u32 Bn = (u32)a.de_size;
        const u32 mm = h2index((u32)(dr.Q0 >> 32), Bn);
        const u32 nn = (mm + 1u + h2index((u32)dr.Q0, Bn - 1u)) % Bn;
        double scale;
        if (plo > T50) scale = 1.0;
        else scale = (de_u ? *de_u : w2uniform(dr.Q1)) * 2.4 / a.gdiv[g] * cc.de_mul;  // PT:976
        const double *rm = DE + (size_t)((mm + (u32)a.de_head) % Bn) * a.de_ld;
        const double *rn = DE + (size_t)((nn + (u32)a.de_head) % Bn) * a.de_ld;
        // G == 4: rows are stored in 16-byte pieces dealt to the four lanes in turn (ptmi_de_row_stride: piece 4 e2 + lane
        // holds the lane's slots 2 e2 and 2 e2 + 1, pads are stored zeros), so one read instruction takes 64 contiguous
        // bytes per chain and the next one the other half of the cache line.  ALL reads of both rows first, then the
        // arithmetic: left to itself hipcc issued them six at a time with a wait behind each batch.
        constexpr int EP2 = (EPL + 1) / 2;
        double vmr[G == 4 ? 2 * EP2 : 1], vnr[G == 4 ? 2 * EP2 : 1];
        if constexpr (G == 4) {
            const ptmi_d2 *pm = reinterpret_cast<const ptmi_d2 *>(rm) + gl, *pn = reinterpret_cast<const ptmi_d2 *>(rn) + gl;
#pragma unroll
            for (int e2 = 0; e2 < EP2; ++e2) { const ptmi_d2 v = pm[4 * e2]; vmr[2 * e2] = v.x; vmr[2 * e2 + 1] = v.y; }
#pragma unroll
            for (int e2 = 0; e2 < EP2; ++e2) { const ptmi_d2 v = pn[4 * e2]; vnr[2 * e2] = v.x; vnr[2 * e2 + 1] = v.y; }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            double vm, vn;
            if constexpr (G == 4) {
                vm = vmr[e];
                vn = vnr[e];
            } else {
                PTMI_ROW_LOAD_S(PSAFE, vm, rm, e);
                PTMI_ROW_LOAD_S(PSAFE, vn, rn, e);
            }
            dq[e] = scale * (vm - vn);
            if (GRP) {                                 // only the group's parameters move (PT:978-983)
                const int i = gl + G * e;
                if (i < d && a.gmask[(size_t)g * d + i] == 0.0) dq[e] = 0.0;
            }
        }
    };
    // AM (PT:879-933): q = x + U (cd sqrt(S) z).  Weights per chain (divergent), product per wave.
    const bool is_am = FULL && jt == PTMI_J_AM;
    auto am_body = [&]() {
        {
            const double cd = a.gcn[g] * cc.sc(br);   // PT:928
            // directions k = gl + G*e and k + G (slots e even / odd) are the cos and sin branches of ONE Box-Muller
            auto weights = [&](int e, double &wa, double &wb) {
                wa = 0.0;
                wb = 0.0;
                const int k = gl + G * e;
                if (is_am && k < ng) {
                    u64 e0, e1;
                    philox_words(a.seed, (u64)it, sid, SLOT_AM + (u32)k, e0, e1);
                    const double r = det_sqrt(-2.0 * unit_log<0>(e0));
                    u32 aj;
                    double at, sn, cs;
                    unit_angle64(e1, aj, at);
                    unit_sincos<0>(aj, at, sn, cs);
                    wa = (r * cs) * cd * root_s(k);                             // PT:930
                    if (e + 1 < EPL && k + G < ng) wb = (r * sn) * cd * root_s(k + G);
                }
            };
            if (STR) {
                if constexpr (G == 4) {
                    MfmaAcc<EPL> acc;
                    am_mfma_product<EPL>(a, is_am, sid, it, cd, ng, Ut, ut_padded, uld, S, s_sqrt, acc, tsm);
                    if (is_am) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) dq[e] = acc.at(e);
                    }
                }
            } else if (a.am_inc != nullptr && am_next != nullptr) {
                // the increment U (cd sqrt(S) z) was computed ahead of the launch on the matrix cores (am_gemm_kernel: the same
                // k-ascending fma chain from the same weights)
                if (is_am) {
                    const double *inc = a.am_inc + (size_t)(*am_next) * d;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD_S(PSAFE, dq[e], inc, e);
                    *am_next += 1;
                }
            } else {
                double wk[EPL];
#pragma unroll
                for (int e = 0; e < EPL; e += 2) {
                    double wa, wb;
                    weights(e, wa, wb);
                    wk[e] = wa;
                    if (e + 1 < EPL) wk[e + 1] = wb;
                    __builtin_amdgcn_sched_barrier(0);     // one Box-Muller at a time: interleaving them only costs registers
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) dq[e] = 0.0;
#pragma unroll
                for (int e2 = 0; e2 < EPL; ++e2) {
#pragma unroll 1
                    for (int src = 0; src < G; ++src) {
                        const int k = src + G * e2;
                        if (k >= ng) break;
                        const double wv = group_bcast_lane<G>(wk[e2], src);
                        const double *row = Ut + (size_t)k * d;
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            double r;
                            PTMI_ROW_LOAD_S(PSAFE, r, row, e);
                            dq[e] = __builtin_fma(r, wv, dq[e]);
                        }
                    }
                }
            }
        }
    };
    if (jt == PTMI_J_SCAM) scam_body();
    else if (FULL && jt == PTMI_J_DE) de_body();
    // !am_here (wave-uniform): the caller takes the AM increments from its queue (mh_steps_kernel, "AM queue")
    if (FULL && am_here && (!STR ? is_am : __any(is_am))) am_body();
    return jt;
}

// The rank-0 chain's row into the AM buffer (PT:327-328), in the buffer's row format (ptmi_common.h am_pos)
// Non-temporal stores: the history is written once per step and read once per covariance epoch, 3.3 GB later -- nothing of it
// is worth a place in the L2 (config-2 kernel 0.785 -> 0.769 ms per 100 steps, the step 0.98 -> 0.96 ms).
template <int G, int EPL>
__device__ __forceinline__ void am_store_row(double *am, const double (&x)[EPL], int gl, int d)
{
    if constexpr (am_row_epl(G, EPL) != 0) {
        ptmi_d2 *ap = reinterpret_cast<ptmi_d2 *>(am) + gl;
#pragma unroll
        for (int e2 = 0; e2 < EPL / 2; ++e2) __builtin_nontemporal_store(ptmi_d2{x[2 * e2], x[2 * e2 + 1]}, &ap[4 * e2]);
        if (EPL & 1) __builtin_nontemporal_store(x[EPL - 1], &am[8 * (EPL / 2) + gl]);
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = gl + G * e;
            if (e < safe_slots(G, EPL) || i < d) __builtin_nontemporal_store(x[e], &am[i]);
        }
    }
}

// The rank-0 chain's row of one step (PT:327-328).  With AM row flags (ptmi_common.h) the row is stored only when the step
// changed it or when it is a KEY row -- first step of a launch, ring rows 0 and 1 -- and its flag word says which.
template <int G, int EPL>
__device__ __forceinline__ void am_store_step(const KArgs &a, int w, int am_row, int k, const double (&x)[EPL], int gl, int d, bool accepted)
{
    const size_t r = (size_t)w * a.cov_update + (size_t)am_row;
    if (a.AMflag != nullptr) {
        const bool key = k == 0 || am_row <= 1;
        if (gl == 0) __builtin_nontemporal_store((key ? AMROW_KEY : 0ull) | (accepted ? AMROW_NEW : 0ull), a.AMflag + r);
        if (!(key || accepted)) return;
    }
    am_store_row<G, EPL>(a.AM + r * d, x, gl, d);
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; make consecutive
// logical blocks (chains of one walker, sharing its Ut) land on one XCD's L2.
__device__ __forceinline__ int logical_block()
{
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// ------------------------------------------------------------ fused MH steps
// STAGE (G = 4 shapes, chosen by the host when all chains of a block share their tables): strided lane layout,
// the dense precision matrix and -- if it still fits -- the block's Ut are copied to LDS zero-padded, and the
// table-times-vector products of the AM proposal and of the dense likelihood run on the matrix cores.
// ULDS (SCAM-only cycles whose block shares one eigenvector table that fits LDS twice per CU): contiguous lane layout,
// the table copied to LDS unpadded, so the one row a step reads comes at LDS latency instead of L2 latency.
// PERS > 0 (ULDS with ONE table for the whole launch, i.e. a pooled covariance): persistent blocks of PERS threads, one per CU,
// over one LDS copy of the table; every wave walks over units of 16 chains on its own (no barrier after the set-up), so the
// occupancy is no longer tied to the number of table copies that fit the LDS,
// the table is staged 256 times per launch instead of 4096 times, and a block's waves do not wait for its cold wave.
// Blocks of 256 threads per CU the register budget is cut for.  The 16- / 64-lane SCAM-only kernels (ndim > 104: the chain's one
// table row per step comes from L2 / MALL, 8 KB at ndim = 1000) hold two row-sized arrays and little else, yet were given the
// 256 registers of two waves per SIMD and used 223 of them (hoisted loads): two chains per SIMD at 64 lanes, every round trip of
// a row exposed.  PTMI_WIDE_MINBLK: an A/B build switch.
#ifndef PTMI_WIDE_MINBLK
#define PTMI_WIDE_MINBLK 0
#endif
constexpr int mh_min_blocks(int G, int EPL, int LOGL, bool FULL, bool STAGE, bool GRP, int PERS, bool UPAD, int PRI)
{
    // the generic kernel of cycles with AM / DE and parameter groups takes its increments, DE rows and group masks from memory: two waves
    // per SIMD (256 registers, 29 spilled at 100-d) against one at 308: 19.2 -> 17.2 ms per 100 steps of the default mix with three groups
    // (three / four waves: 196 / 281 spilled, 24.5 / 29.6 ms)
#ifndef PTMI_GRPFULL_MINBLK
#define PTMI_GRPFULL_MINBLK 2
#endif
    if (FULL && GRP && !STAGE && !PERS && G == 4 && LOGL != PTMI_LOGL_DENSE) return PTMI_GRPFULL_MINBLK;
    if (PERS || STAGE || FULL || LOGL == PTMI_LOGL_DENSE) return 1;
    // the flat-prior instantiation over the padded table: x, dq and the row in flight (the other variants of the wide shapes carry
    // per-slot bounds and the box test and spill under the tighter budget)
    if (UPAD && PRI == PTMI_LOGP_FLAT && LOGL == PTMI_LOGL_ISO) return PTMI_WIDE_MINBLK ? PTMI_WIDE_MINBLK : (EPL <= 16 ? 4 : (EPL <= 26 ? 3 : 2));
    if (UPAD && PRI == PTMI_LOGP_BOX && LOGL == PTMI_LOGL_ISO) return EPL <= 16 ? 3 : 2;       // (+ the one-pass test and margin of the box prior's slow path)
    return 2;
}
template <int G, int EPL, int LOGL, bool FULL, bool STAGE, bool GRP, bool ULDS = false, int PERS = 0, int PRI = -1 /* PERS: the prior kind */,
          bool TLDS = false /* ULDS with a table copy per block: the draw tables are in LDS too (host: a.tab_off >= 0) */,
          bool UPAD = false /* SCAM-only wide shapes: the direction comes from the library's zero-padded table copy (a.UtPad) */>
__global__ __launch_bounds__(PERS ? PERS : 256, mh_min_blocks(G, EPL, LOGL, FULL, STAGE, GRP, PERS, UPAD, PRI)) void mh_steps_kernel(const KArgs a)
{
    static_assert(!UPAD || (G > 4 && !FULL && !STAGE && !GRP && !ULDS), "UPAD is a variant of the wide SCAM-only kernel");
    static_assert(!ULDS || (!STAGE && !FULL && !GRP), "ULDS is the SCAM-only contiguous-layout kernel");
    static_assert(!PERS || (ULDS && G == 4), "persistent blocks are a variant of the ULDS kernel");
    // the draw tables (ptmi_tables.h, 1 KB): the staged full kernels read them from an LDS copy when the host found room
    // (a.tab_off >= 0: 13.0 instead of 14.1 ms per 100 steps of the default mix); the SCAM-only kernels from global memory --
    // measured in the ULDS kernel: 1.14 ms per 100 steps from global, 1.20 from LDS, whose pipe serves the direction rows
    constexpr int TM = (STAGE && FULL) ? 2 : 0;
    constexpr int BLK = PERS ? PERS : 256;
    constexpr int CPB = BLK / G;
    constexpr bool STR = STAGE;
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int cib = STR ? wave * 16 + (lane & 15) : (int)(threadIdx.x / G);       // chain in block
    const int gl = STR ? lane >> 4 : (int)(threadIdx.x % G);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // SCAM-only cycle, one parameter group: the chain-scalar half of the proposal moves into the draw pass (ScamBatch)
    constexpr bool SCAMFAST = !FULL && !GRP;
    // Box prior, SCAM cycle, one table for the launch: a SCAM jump moves no element further than |amp| max|U|, so while that REACH
    // stays below the chain's distance to its nearest bound the proposal is inside without looking at a single bound (the test is
    // 2 EPL comparisons and EPL 16-byte LDS reads per step: as much LDS traffic as the direction row, +75 % on the config-2 step).
    // The margin is a lower bound kept per chain: minus the reach after every accepted step, recomputed (one pass with the full
    // test) when a reach no longer fits under it.  Same decisions as the full test in every step: it is only skipped where its
    // result is known (the slack factors cover the roundings of amp * u and of the margin's own arithmetic).
    // Only where a chain is a whole wave (64 lanes): with several chains per wave the slow path runs whenever ANY of them needs it,
    // and a tempered ladder's hot chains live near their bounds -- measured at config 2 with a box prior: 2.06 ms per 100 steps
    // against 1.27 with the plain test in every step (16 chains per wave), 4.36 against 3.95 at ndim = 300 (4 per wave); at
    // ndim = 1000 (one per wave): 2.40 against 5.90.
    constexpr bool BOXFAST = SCAMFAST && PRI == PTMI_LOGP_BOX && UPAD && G == 64;
    // ULDS with the exact shape (4, 25) (ndim = 100): the table rows are stored in the lanes' order, 16-byte pieces dealt to
    // the four lanes in turn (position 8 (e / 2) + 2 lane + e % 2 holds element lane + 4 e; the odd last slot at 96 + lane), so
    // that a step reads its direction with 12 ds_read_b128 + 1 ds_read_b64 instead of 10 ds_read2_b64 + 6 ds_read_b64:
    // 61 % of the LDS cycles of that kernel were bank conflicts of the 8-byte reads (profiles/r03_scam_lds.txt)
    constexpr bool PAIRED = ULDS && G == 4 && EPL == 25;
    constexpr int LD = mfma_ld(EPL);
    const int tab_n = 4 * ((d + 3) / 4) * LD;                      // doubles of one zero-padded LDS table
    const double *PtG = LOGL == PTMI_LOGL_DENSE ? a.logl_par + d + (size_t)d * d : nullptr;       // the half table Tl
    // LDS pointers are derived from smem at their use so that they stay LDS (ds_read) accesses
#define PTMI_PL (smem)
#define PTMI_UL (smem + (LOGL == PTMI_LOGL_DENSE ? (size_t)tab_n : 0))
    // FULL staged kernels: sqrt of the block's eigenvalues, behind the tables (d doubles)
#define PTMI_SQ (smem + (size_t)tab_n * ((LOGL == PTMI_LOGL_DENSE ? 1 : 0) + ((UT_ALWAYS_LDS || a.lds_u) ? 1 : 0)))
    // with the dense likelihood both tables may not fit: then Ut stays in global memory (host decides, a.lds_u)
    constexpr bool UT_ALWAYS_LDS = LOGL != PTMI_LOGL_DENSE;
    const double *const tsm = TM ? smem : nullptr;
    // ---- set-up of the block: its tables into LDS.  The block's table: the launch's one table (PERS, pooled covariance), else
    // the table of the walker its first chain belongs to (the host checks that all its chains share it, launch_mh_k)
    size_t w0 = 0;
    if (!PERS && a.per_walker) {
        const long long ch0 = (long long)logical_block() * CPB;
        w0 = (size_t)((ch0 < nch ? ch0 : nch - 1) / nt);
    }
    const double *const UtBlk = a.Ut + w0 * d * d;
    if (TM) draw_table_fill(smem, a.tab_off, BLK);
    if (STAGE) {
        if (LOGL == PTMI_LOGL_DENSE) {
            for (int i = (int)threadIdx.x; i < tab_n; i += BLK) {
                const int r = i / LD, c = i % LD;
                PTMI_PL[i] = (r < d && c < d) ? PtG[(size_t)r * d + c] : 0.0;
            }
        }
        // FULL: the block's table serves all its chains.  SCAM-only cycles read one row per step from the chain's OWN table.
        if (FULL) {
            const double *Sb = a.S + w0 * d;                                               // the eigenvalues that go with UtBlk
            for (int i = (int)threadIdx.x; i < d; i += BLK) PTMI_SQ[i] = det_sqrt(Sb[i]);
        }
        if (FULL && (UT_ALWAYS_LDS || a.lds_u)) {
            for (int i = (int)threadIdx.x; i < tab_n; i += BLK) {
                const int r = i / LD, c = i % LD;
                PTMI_UL[i] = (r < d && c < d) ? UtBlk[(size_t)r * d + c] : 0.0;
            }
        }
        __syncthreads();
    }
    // ULDS with a box prior and a table copy per block: the bounds table takes the place of sqrt(S) (both do not fit twice per
    // CU at d = 100); a persistent block (one copy per CU) has room for both
    const bool ulds_box = ULDS && !PERS && a.logp_kind == PTMI_LOGP_BOX && a.box_off >= 0 && a.box_off < d * d + d;
    if (ULDS) {
        const double *src = UtBlk, *srcS = a.S + w0 * d;
        if constexpr (PAIRED) {
            auto perm = [](int c) { const int ln = c & 3, e = c >> 2; return e < 2 * (EPL / 2) ? 8 * (e >> 1) + 2 * ln + (e & 1) : 8 * (EPL / 2) + ln; };
            if (d == 4 * EPL) {
                // 16-byte loads, ten of a thread in flight at once: one 8-byte load per trip of the loop below waited for its own
                // round trip forty times per block -- with a table per walker (memory, not L2) a tenth of the block's life
                // (launches of 0.96 ms against 0.85 with one table for all blocks)
                constexpr int NPAIR = 2 * EPL * 4 * EPL, NB = 10;
                const ptmi_d2 *src2 = reinterpret_cast<const ptmi_d2 *>(src);
                for (int base = 0; base < NPAIR; base += NB * BLK) {
                    ptmi_d2 tmp[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int idx = base + (int)threadIdx.x + u * BLK;
                        tmp[u] = src2[idx < NPAIR ? idx : NPAIR - 1];
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int idx = base + (int)threadIdx.x + u * BLK;
                        if (idx < NPAIR) {
                            const int r = (2 * idx) / (4 * EPL), c = (2 * idx) % (4 * EPL);
                            smem[r * (4 * EPL) + perm(c)] = tmp[u].x;
                            smem[r * (4 * EPL) + perm(c + 1)] = tmp[u].y;
                        }
                    }
                }
            } else {
                for (int i = (int)threadIdx.x; i < d * d; i += BLK) {
                    const int r = i / d, c = i % d;
                    smem[r * d + perm(c)] = src[i];
                }
            }
        } else {
            for (int i = (int)threadIdx.x; i < d * d; i += BLK) smem[i] = src[i];
        }
        if (!ulds_box)
            for (int i = (int)threadIdx.x; i < d; i += BLK) smem[d * d + i] = det_sqrt(srcS[i]);   // sqrt(eigenvalues) too: no vector-memory read is left in the step loop
    }
    if (PRI != PTMI_LOGP_FLAT) box_table_fill<G, EPL>(a, smem, BLK);
    if (ULDS) draw_table_fill(smem, a.tab_off, BLK);
    if constexpr (BOXFAST && PERS != 0) {
        if (threadIdx.x == 0) reinterpret_cast<unsigned long long *>(smem)[a.umax_off] = 0ull;
    }
    if (ULDS || (a.logp_kind == PTMI_LOGP_BOX && a.box_off >= 0)) __syncthreads();
    double box_umax = 0.0;                       // BOXFAST: max |U| over the launch's table
    if constexpr (BOXFAST) {
        if constexpr (PERS != 0) {
            double am = 0.0;
            for (int i = (int)threadIdx.x; i < d * d; i += BLK) { const double v = __builtin_fabs(smem[i]); am = v > am ? v : am; }
            // non-negative doubles order like their bit patterns
            __hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(smem) + a.umax_off, (unsigned long long)__double_as_longlong(am), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            box_umax = smem[a.umax_off];
        } else {
            box_umax = *a.ut_absmax;             // ut_pad_kernel's
        }
        box_umax *= 1.0 + 0x1.0p-30;            // covers the rounding of amp * u
    }

    // ---- the chains.  A block of 256 threads serves its 64 chains; a persistent block's waves walk over units of 16 chains on
    // their own (no barrier from here on)
    const long long nunits = (nch + 15) / 16, ustride = PERS ? (long long)gridDim.x * (BLK / 64) : nunits;
    // The unit that holds a walker's rank-0 chain stores an AM row per step and takes ~13 % longer; which of the walker's units
    // that is changes with every swap.  The waves therefore walk over the units in an order that lists the W cold units first
    // (position p < W: walker p's cold unit), then the others walker by walker: a wave's positions p, p + stride, ... hold the same
    // number of cold units for every wave (two of eight at 64 x 4096), where the plain order gave it a binomial draw.
    const int upw = nt / 16, upw1 = upw > 1 ? upw - 1 : 1;               // units per walker
    const bool cold_first = PERS && a.AM != nullptr && a.temp0 == 0 && nt % 16 == 0;
    for (long long upos = PERS ? (long long)blockIdx.x * (BLK / 64) + wave : 0; upos < nunits; upos += ustride) {
    long long unit = upos;
    if (cold_first) {
        const long long r = upos - a.W;
        const int wq = upos < a.W ? (int)upos : (int)(r / upw1);
        const int cq = __builtin_amdgcn_readfirstlane(a.slot_of[(size_t)wq * nt]) >> 4;        // the unit of the walker's rank 0
        unit = (long long)wq * upw + (upos < a.W ? cq : (cq + 1 + (int)(r % upw1)) % upw);
    }
    long long ch = PERS ? unit * 16 + (STR ? (lane & 15) : (lane >> 2)) : (long long)logical_block() * CPB + cib;
    const bool live = ch < nch;
    if (!live) ch = nch - 1;
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const int tg = a.temp0 + t;
    const double beta = a.beta[t];
    const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
    const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);                 // stream of the walker's rank 0
    const u32 sid = sid0 + (u32)tg;
    const size_t wc = a.per_walker ? (size_t)w : 0;
    const double *Ut = a.Ut + wc * a.ngroups * d * d, *S = a.S + wc * a.ngroups * d;
    const double *DE = (FULL && a.DE) ? a.DE + wc * (size_t)a.de_size * a.de_ld : nullptr;
    double *xrow = a.X + (size_t)ch * d;
    constexpr int GW = (G > 4 && !STR && !ULDS) ? G : 4;   // 16- / 64-lane shapes: wide draw batches (all lanes of the chain draw)
    DrawBatch<STR, GW> batch;
    ScamBatch<STR, GW> sbatch;
    const double *UtBlock = (STAGE && FULL) ? UtBlk : Ut;

    // ---- AM queue (staged full kernels).  An AM increment U (cd sqrt(S) z) does not depend on the chain's state, only on
    // its stream, the iteration and the scale branch -- all known from the draws -- and the matrix instruction computes 16
    // columns whether 16 chains of the wave picked AM or one.  With every chain picking its own proposal (the reference's
    // _jump) a third of the columns were used.  So the wave looks ahead in blocks of four steps, one block in advance: lane
    // (c16, g4) evaluates the pick of chain c16 for step g4 of the block.  The AM events of the wave, in step order, get
    // consecutive ranks, and a matrix pass computes 16 of them at a time into a ring of 16 increments in LDS, just in time
    // for the step that needs them: ranks [done, min(known, consumed + 16)) -- what a pass overwrites has been consumed,
    // and with 4 to 7 steps known ahead a pass is almost always full.  The arithmetic of an increment is unchanged.
    // A pass through the queue costs 1.28 x a pass in place (LDS round trip of the increments, per-lane counters in the
    // generator): the launch uses the queue when 1.28 x the expected events per step / 16 is below the chance that a step
    // has an event at all, i.e. for per-chain picks unless the cycle is nearly all AM; with one pick per walker every pass
    // in place is full anyway.
    constexpr bool AMQ = STAGE && FULL && G == 4;
    constexpr int AMQ_LD = 4 * EPL + 2;              // doubles of one queued increment (lane-major like a DE row; + 2: the 16 slots start 20 banks apart)
#define PTMI_AMQ(slot) (smem + a.amq_off + ((size_t)wave * 16 + (size_t)(slot)) * AMQ_LD)
#define PTMI_AMQ_IDX ((int *)(smem + a.amq_off + (size_t)4 * 16 * AMQ_LD) + wave * 128)
    const bool amq_on = AMQ && a.amq_on;
    u64 mask_c = 0, mask_n = 0;                      // AM events of the current / the next block of four steps (bit 16 step + chain)
    int rank_c = 0, rank_n = 0, base_c = 0, base_n = 0, q_done = 0;
    double cd_c = 0.0, cd_n = 0.0;

    double x[EPL], dq[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(x[e], xrow, e);
    double lnL = a.lnL[ch], lp = a.lp[ch];
    u32 nacc = 0, jp[PTMI_J_FUSED] = {0, 0, 0}, ja[PTMI_J_FUSED] = {0, 0, 0};
    long long am_next = (FULL && a.am_base != nullptr) ? a.am_base[ch] : 0;      // the chain's next precomputed AM increment
    const bool cold = live && tg == 0 && a.AM != nullptr;
    int am_row = a.am_row0;
    double box_margin = -1.0;                    // BOXFAST: a lower bound of the chain's distance to its nearest bound (negative: unknown)
    double box_guard = 0.0;                      // ... and what the rounding of an accepted x + dq can add to an element: 2^-52 x the largest |bound|

    // Wide draw batches: a pass serves GW / 2 steps, so the steps run as an inner loop under a loop over the passes -- with the
    // refill as a rarely taken branch of ONE loop the compiler hoisted the pass's invariants (Philox key schedule, polynomial
    // constants, per-slot bounds) over the steps and spilled the steps' own values to make room (73 registers at a budget of 128)
    constexpr int KPASS = GW > 4 ? GW / 2 : (1 << 30);
    for (int k0 = 0; k0 < a.nsteps; k0 += KPASS) {
    const int kend = (GW > 4 && a.nsteps - k0 > KPASS) ? k0 + KPASS : a.nsteps;
    if constexpr (GW > 4) {
        if constexpr (SCAMFAST) sbatch.template refill<0>(a, a.iter0 + k0, sid, gl, cc, d, [&](int kk) { return det_sqrt(S[kk]); }, nullptr);
        else batch.template refill<TM>(a, a.iter0 + k0, sid, gl, tsm);
    }
    for (int k = k0; k < kend; ++k) {
        const long long it = a.iter0 + k;
        if constexpr (AMQ) {
            if (amq_on) {
                const int s4 = k & 3, c16 = lane & 15;
                if (s4 == 0) {
                    // the picks of the block of four steps that starts at step kb; its events take ranks from `base` on
                    auto look = [&](int kb, u64 &mask, int &rank, double &cdv, int base) {
                        bool ev = false;
                        cdv = 0.0;
                        if (kb + gl < a.nsteps) {
                            u64 p0, p1;
                            philox_words(a.seed, (u64)(a.iter0 + kb + gl), sid, 0u, p0, p1);
                            const int w_de = a.de_on ? a.w_de : 0;
                            const int ind = (int)h2index((u32)(p0 >> 32), (u32)(a.w_host + a.w_scam + a.w_am + w_de)) - a.w_host;   // as propose()
                            ev = live && ind >= a.w_scam && ind < a.w_scam + a.w_am;
                            constexpr u32 T97 = (u32)(0.97 * 4294967296.0), T90 = (u32)(0.9 * 4294967296.0);
                            const u32 plo = (u32)p0;
                            cdv = a.gcn[0] * cc.sc(plo > T97 ? 0 : (plo > T90 ? 1 : 2));                                    // PT:928
                        }
                        mask = __ballot(ev);
                        rank = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
                        if (ev) PTMI_AMQ_IDX[rank & 127] = lane | (((kb >> 2) & 1) << 6);
                    };
                    if (k == 0) look(0, mask_n, rank_n, cd_n, 0);
                    mask_c = mask_n; rank_c = rank_n; cd_c = cd_n; base_c = base_n;
                    base_n = base_c + (int)__popcll(mask_c);
                    look(k + 4, mask_n, rank_n, cd_n, base_n);
                    asm volatile("" ::: "memory");                       // LDS serves a wave in order; this orders the compiler
                }
                const int cons = base_c + (int)__popcll(mask_c & ((1ull << (16 * s4)) - 1ull));                             // events of the steps before this one
                const int need = base_c + (int)__popcll(mask_c & (s4 == 3 ? ~0ull : ((1ull << (16 * s4 + 16)) - 1ull)));    // ... up to and including it
                if (q_done < need) {                                     // wave-uniform: a matrix pass for ranks [q_done, hi)
                    const int known = base_n + (int)__popcll(mask_n);
                    const int hi = known < cons + 16 ? known : cons + 16;
                    const int r = q_done + c16;
                    const bool valid = r < hi;
                    const int entry = PTMI_AMQ_IDX[(valid ? r : q_done) & 127];
                    const int owner = entry & 63;
                    const bool of_cur = (entry >> 6) == ((k >> 2) & 1);
                    const u32 sid_ev = (u32)__shfl((int)sid, owner, 64);
                    const double cdc = __shfl(cd_c, owner, 64), cdn = __shfl(cd_n, owner, 64);
                    const double cd_ev = of_cur ? cdc : cdn;
                    const long long it_ev = a.iter0 + (k - s4) + (of_cur ? 0 : 4) + (owner >> 4);
                    MfmaAcc<EPL> acc;
                    if (UT_ALWAYS_LDS || a.lds_u) am_mfma_product<EPL>(a, valid, sid_ev, it_ev, cd_ev, d, PTMI_UL, true, mfma_ld(EPL), PTMI_SQ, true, acc, tsm);
                    else am_mfma_product<EPL>(a, valid, sid_ev, it_ev, cd_ev, d, UtBlock, false, d, PTMI_SQ, true, acc, tsm);
                    if (valid) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) PTMI_AMQ(r & 15)[gl * EPL + e] = acc.at(e);
                    }
                    q_done = hi;
                    asm volatile("" ::: "memory");
                }
            }
        }
        double log_u;
        int jt = PTMI_J_SCAM;
        double scam_amp = 0.0, box_reach = 0.0;
        if constexpr (SCAMFAST) {
            ScamDraw sd;
            // sqrt(S_k): from the block's LDS copy where it has one, else from the chain's table (sqrt is correctly rounded: same bits)
            // ULDS: the draw tables too come from the block's LDS when the host found room (a.tab_off >= 0).  Global reads share
            // the in-order vector-memory counter with the cold chain's AM-row stores: in the block's one cold wave every draw
            // pass waited for 25 stores to retire first
            // PERS: the host always places the tables (launch_mh_k), so the read is an LDS read at COMPILE time: with the run-time
            // choice (TM = 2) the two paths merged in an s_waitcnt vmcnt(0) -- every draw pass of a cold wave waited for its AM-row
            // stores to retire, although it never took the global path
            scam_draws_for_step<STR, (PERS || TLDS) ? 1 : (ULDS ? 2 : 0), GW>(sbatch, sd, a, k, sid, gl, cc, d, [&](int kk) {
                if (ULDS && !ulds_box) return smem[d * d + kk];
                return det_sqrt(S[kk]);
            }, smem);
            log_u = sd.log_u;
            if constexpr (PAIRED) {
                const double *row = smem + (size_t)sd.k * d;
                const ptmi_d2 *rp = reinterpret_cast<const ptmi_d2 *>(row) + gl;
#pragma unroll
                for (int e2 = 0; e2 < EPL / 2; ++e2) { const ptmi_d2 v = rp[4 * e2]; dq[2 * e2] = v.x; dq[2 * e2 + 1] = v.y; }
                if (EPL & 1) dq[EPL - 1] = row[8 * (EPL / 2) + gl];
            } else {
                if constexpr (ULDS) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(dq[e], smem + (size_t)sd.k * d, e);
                } else {
                    if constexpr (UPAD) {                          // the library's zero-padded copy: no bounds to check
                        // a base per 4 KB of the row (wave-uniform at 64 lanes: scalar adds), so that every load is base + the
                        // lane's offset + an immediate; beyond the 13-bit immediate the compiler kept one offset register per slot
                        const double *col = a.UtPad + (size_t)sd.k * (G * EPL) + gl;
                        constexpr int SPB = 512 / G;                // slots per 4 KB
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const double *base = col + (e / SPB) * 512;
                            dq[e] = base[G * (e % SPB)];
                        }
                        // every request of the row goes out before the first product: left alone the scheduler, short of registers,
                        // issued ten of the sixteen loads one at a time, each behind a wait for the one before
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        const double *col = UtBlock + (size_t)sd.k * d;
#pragma unroll
                        for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(dq[e], col, e);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) dq[e] = sd.amp * dq[e];
            scam_amp = sd.amp;
        } else {
        Draws dr;
        draws_for_step<STR, FULL, TM, GW>(batch, dr, a, k, sid, sid0, gl, tsm);
        log_u = dr.log_u;
        if (ULDS && ulds_box) jt = propose<G, EPL, FULL, STR, GRP>(a, it, sid, gl, cc, dr, smem, false, S, DE, dq, false);
        else if (ULDS) jt = propose<G, EPL, FULL, STR, GRP>(a, it, sid, gl, cc, dr, smem, false, smem + d * d, DE, dq, true);
        else if (STAGE && FULL && (UT_ALWAYS_LDS || a.lds_u)) jt = propose<G, EPL, FULL, STR, GRP>(a, it, sid, gl, cc, dr, PTMI_UL, true, PTMI_SQ, DE, dq, true, !amq_on, tsm);
        else if (STAGE && FULL) jt = propose<G, EPL, FULL, STR, GRP>(a, it, sid, gl, cc, dr, UtBlock, false, PTMI_SQ, DE, dq, true, !amq_on, tsm);
        else jt = propose<G, EPL, FULL, STR, GRP>(a, it, sid, gl, cc, dr, UtBlock, false, S, DE, dq, false, true, nullptr, &am_next);
        }
        if constexpr (AMQ) {
            // the rank of this chain's event of this step is held by its lane of row (k & 3)
            const int rk = __shfl(rank_c, 16 * (k & 3) + (lane & 15), 64);
            if (amq_on && jt == PTMI_J_AM) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) dq[e] = PTMI_AMQ(rk & 15)[gl * EPL + e];
            }
        }
        if (FULL) {
#pragma unroll
            for (int j = 0; j < PTMI_J_FUSED; ++j) jp[j] += (jt == j);
        }
        // PT:605-612
        double nlp, nlnL = 0.0, nlnprob;
        {
            double q[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) q[e] = x[e] + dq[e];
            if constexpr (PRI == PTMI_LOGP_FLAT) nlp = 0.0;
            else if constexpr (BOXFAST) {
                box_reach = __builtin_fabs(scam_amp) * box_umax;
                bool inside = box_reach < box_margin;
                if (!inside) {                   // (divergent between the wave's chains; rare in a box wider than the jumps)
                    bool in1;
                    double mg, bm;
                    box_test_and_margin<G, EPL>(smem, a.box_off, gl, x, dq, in1, mg, bm);
                    inside = grp_all<G, STR>(in1);
                    box_margin = grp_min<G, STR>(mg) * (1.0 - 0x1.0p-40);
                    box_guard = -grp_min<G, STR>(-bm) * 0x1.0p-52;
                }
                nlp = inside ? 0.0 : -__builtin_inf();
            }
            else if constexpr (PERS != 0 && PRI == PTMI_LOGP_BOX)
                nlp = grp_all<G, STR>(box_inside_lds<G, EPL>(smem, a.box_off, gl, [&](int e) { return q[e]; })) ? 0.0 : -__builtin_inf();
            else nlp = eval_logp<G, EPL, STR>(a, q, gl, smem);
            // the reference skips logl when the prior is -inf (PT:607-608); the value is unused then, and the
            // matrix-core path needs every lane, so it is evaluated unconditionally
            if (STAGE) nlnL = eval_logl<G, EPL, LOGL, STR>(a, q, gl, PTMI_PL);
            else nlnL = eval_logl<G, EPL, LOGL, STR>(a, q, gl, PtG);
            nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
        }
        // PT:615-622
        const double lnprob0 = beta * lnL + lp;
        const double diff = nlnprob - lnprob0 + 0.0;
        const bool accepted = diff > log_u;
        if (accepted) {
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
                for (int j = 0; j < PTMI_J_FUSED; ++j) ja[j] += (jt == j);
            }
            // no element moved further than the reach plus the rounding of its sum (relative to the ELEMENT, which a bound limits: the
            // slack factors alone are relative to the margin, and a narrow box far from the origin has a margin far below |x| 2^-13)
            if constexpr (BOXFAST) box_margin = (box_margin - box_reach - box_guard) * (1.0 - 0x1.0p-40);
        }
        // PT:327-328 (the post-swap row of a swap iteration is written by the swap).  These 25 stores of four active lanes, in ONE
        // wave of every block, are 15 % of the config-2 kernel (0.90 -> 0.77 ms without them, PTMI_MEASURE_NO_AM): the wave is its
        // block's straggler.  Sending the row through LDS and out as two coalesced stores of the whole wave was built twice -- stored
        // in the same step, and one step late so that no wait sits on the critical path -- and measured slower both times (1.00 ms).
        // Round 3 (persistent blocks, cold-first walk): 0.783 ms with the stores, 0.751 with every row of a walker sent to ONE
        // cache-resident row (PTMI_MEASURE_AM_SMALL), 0.697 without them.  Units of 16 rank-0 chains of 16 different walkers -- the
        // same rows as 13 stores of a FULL wave in one unit of 64 instead of 13 four-lane stores in one unit of four -- measured
        // 0.780 against 0.778: the cost is the bytes through the CU's store path (1.28 MB per CU and launch) and the scattered
        // 64-byte writes behind it, not the issue slots of the instructions.
        if (cold && !(a.swap_last && k == a.nsteps - 1)) {
            am_store_step<G, EPL>(a, w, am_row, k, x, gl, d, accepted);
            if (a.AMaux && gl == 0) {
                double *ax = a.AMaux + ((size_t)w * a.cov_update + (size_t)am_row) * 2;
                ax[0] = lnL;
                ax[1] = lp;
            }
        }
        am_row = am_row + 1 == a.cov_update ? 0 : am_row + 1;
    }
    }       // passes of a wide draw batch (one trip otherwise)
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
            for (int j = 0; j < PTMI_J_FUSED; ++j) {
                a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 0] += jp[j];
                a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 1] += ja[j];
            }
        }
    }
    }       // units of a persistent block (one trip otherwise)
}

// Cycles with AM entries, per-chain picks (the reference's _jump), 4-lane shapes, one table for the block: the staged full
// kernel as a PRODUCER / CONSUMER pair per SIMD.  mh_steps_kernel<..., FULL, STAGE> runs the AM queue and the steps in ONE wave
// per SIMD (395 registers): every latency of the step -- the DE gather's memory round trip above all -- and every dependency of
// the matrix pass is exposed.  Here a block is eight waves over the same LDS tables: waves 0-3 step their 16 chains each
// (state, proposal bodies, likelihood, accept: under 256 registers without the accumulators and the table operands), waves
// 4-7 -- wave 4 + j on the SIMD of wave j -- compute the AM increments of wave j's chains: they list the AM picks of the
// launch themselves (the picks depend on the streams alone), run the matrix pass for 16 events at a time
// (am_mfma_product: the arithmetic of an increment is unchanged) and hand the increments over through the ring of 16 slots
// in LDS that the queue of mh_steps_kernel uses.  Two words in LDS per pair order them: `produced` (events whose increments
// are in the ring; written by the producer behind its ring writes) and `consumed` (events the stepper has read; written by
// the stepper behind its ring reads): the producer computes a pass ahead and waits for `consumed` before it overwrites slots,
// the stepper waits for `produced` at an AM step, and publishes partial progress inside a step (else a step that straddles
// two passes would wait for a pass that waits for it).  Events are numbered (step, chain) ascending by both sides.
// Matrix and vector instructions of a SIMD still exclude each other: what the split buys is that either wave's waits are the
// other's issue slots.
#ifndef PTMI_PC_PRIO
#define PTMI_PC_PRIO 0       // wave priority of the producers (measured: 1 and 3 cost 9 % against 0) ...
#endif
#ifndef PTMI_PC_SPRIO
#define PTMI_PC_SPRIO 1      // ... and of the steppers (1 or 3: 0.8 % better than 0)
#endif
#ifndef PTMI_PC_ACCV
#define PTMI_PC_ACCV 1       // the producers' accumulators in ordinary vector registers (mfma_f64_acc)
#endif
#ifndef PTMI_PC_SLEEP
#define PTMI_PC_SLEEP 2
#endif
template <int EPL, int LOGL, int PRI, bool PERS>
__global__ __launch_bounds__(512) void mh_pc_kernel(const KArgs a)
{
    constexpr int G = 4, CPB = 64;
    constexpr bool STR = true;
    constexpr int LD = mfma_ld(EPL), AMQ_LD = 4 * EPL + 2;
    // dense likelihood: its half table takes the eigenvector table's place in LDS (both do not fit beside the ring); the
    // eigenvectors are then read from global memory (L2) by the producers' matrix passes and the steppers' SCAM steps
    constexpr bool DENSE = LOGL == PTMI_LOGL_DENSE;
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    const int lane = (int)(threadIdx.x & 63), wave8 = (int)(threadIdx.x >> 6);
    const int pair = wave8 & 3;
    const bool producer = wave8 >= 4;
    const int c16 = lane & 15, gl = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tab_n = 4 * ((d + 3) / 4) * LD;
#define PTMI_PC_UL (smem)                         // the eigenvector table (iso / curved) or the dense likelihood's half table
#define PTMI_PC_SQ (smem + (size_t)tab_n)
#define PTMI_PC_RING(slot) (smem + a.amq_off + ((size_t)pair * 16 + (size_t)(slot)) * AMQ_LD)
#define PTMI_PC_CD (smem + a.amq_off + (size_t)4 * 16 * AMQ_LD + (size_t)pair * 128)
#define PTMI_PC_IDX ((int *)(smem + a.amq_off + (size_t)4 * 16 * AMQ_LD + (size_t)4 * 128) + pair * 128)
#define PTMI_PC_FLG ((int *)(smem + a.amq_off + (size_t)4 * 16 * AMQ_LD + (size_t)4 * 128) + 4 * 128 + pair * 2)
#define PTMI_PC_MU (smem + a.amq_off + (size_t)4 * 16 * AMQ_LD + (size_t)4 * 128 + 260)      // DENSE: the mean, 4 EPL doubles behind the counters
    // ---- the block's tables
    size_t w0 = 0;
    if (!PERS && a.per_walker) {
        const long long ch0 = (long long)logical_block() * CPB;
        w0 = (size_t)((ch0 < nch ? ch0 : nch - 1) / nt);
    }
    {
        const double *UtBlk = a.Ut + w0 * d * d, *Sb = a.S + w0 * d;
        const double *src = DENSE ? a.logl_par + d + (size_t)d * d /* the half table Tl */ : UtBlk;
        draw_table_fill(smem, a.tab_off, 512);
        for (int i = (int)threadIdx.x; i < d; i += 512) PTMI_PC_SQ[i] = det_sqrt(Sb[i]);
        for (int i = (int)threadIdx.x; i < tab_n; i += 512) {
            const int r = i / LD, c = i % LD;
            PTMI_PC_UL[i] = (r < d && c < d) ? src[(size_t)r * d + c] : 0.0;
        }
        box_table_fill<G, EPL>(a, smem, 512);
        if (threadIdx.x < 8) ((int *)(smem + a.amq_off + (size_t)4 * 16 * AMQ_LD + (size_t)4 * 128) + 4 * 128)[threadIdx.x] = 0;
        if constexpr (DENSE)
            for (int i = (int)threadIdx.x; i < 4 * EPL; i += 512) PTMI_PC_MU[i] = i < d ? a.logl_par[i] : 0.0;
        __syncthreads();
    }
    // ---- units of 16 chains.  A block serves four units at a time (one per pair); with ONE table for the launch (pooled
    // covariance: PERS) the blocks are persistent -- one per CU, the tables staged once -- and every pair walks over its own
    // units without waiting for the block's other pairs (the AM picks of a unit vary by a few percent: a block of four ended
    // with its slowest pair).  The event numbers and the two counters run on from unit to unit.
    const long long nunits = (nch + 15) / 16;
    const long long ustride = PERS ? (long long)gridDim.x * 4 : nunits;
    int *const flg = PTMI_PC_FLG;
    const double *const UtG = a.Ut + w0 * d * d;             // DENSE: the block's eigenvector table in global memory
    if (producer) {
        if (PTMI_PC_PRIO) __builtin_amdgcn_s_setprio(PTMI_PC_PRIO);
        const int w_de = a.de_on ? a.w_de : 0;
        const u32 L = (u32)(a.w_host + a.w_scam + a.w_am + w_de);
        int known = 0, q_done = 0;
        int *const idx = PTMI_PC_IDX;
        double *const cdl = PTMI_PC_CD;
        for (long long unit = (long long)(PERS ? blockIdx.x : logical_block()) * 4 + pair; unit < nunits; unit += ustride) {
            // the chain of this lane: the producer's lanes mirror their stepper's
            long long ch = unit * 16 + c16;
            const bool live = ch < nch;
            if (!live) ch = nch - 1;
            const int w = (int)(ch / nt);
            const int t = a.temp_of[ch];
            const ChainConst cc = chain_const(a.temps_mh[t], a.beta[t], d);
            const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);
            const u32 sid = sid0 + (u32)(a.temp0 + t);
            int kb = 0;
            for (;;) {
                // list the AM picks of further blocks of four steps: lane (c16, gl) evaluates chain c16 at step kb + gl (as propose())
                while (known - q_done < 16 && kb < a.nsteps) {
                    bool ev = false;
                    double cdv = 0.0;
                    if (kb + gl < a.nsteps) {
                        u64 p0, p1;
                        philox_words(a.seed, (u64)(a.iter0 + kb + gl), sid, 0u, p0, p1);
                        u32 pickw = (u32)(p0 >> 32);
                        if (a.pick_walker) {                 // pick_mode WALKER: the word of the walker's rank 0 picks the type for all its ranks
                            u64 q0, q1;
                            philox_words(a.seed, (u64)(a.iter0 + kb + gl), sid0, 0u, q0, q1);
                            pickw = (u32)(q0 >> 32);
                        }
                        const int ind = (int)h2index(pickw, L) - a.w_host;
                        ev = live && ind >= a.w_scam && ind < a.w_scam + a.w_am;
                        constexpr u32 T97 = (u32)(0.97 * 4294967296.0), T90 = (u32)(0.9 * 4294967296.0);
                        const u32 plo = (u32)p0;
                        cdv = a.gcn[0] * cc.sc(plo > T97 ? 0 : (plo > T90 ? 1 : 2));               // PT:928
                    }
                    const u64 mask = __ballot(ev);
                    const int rank = known + (int)__popcll(mask & ((1ull << lane) - 1ull));
                    if (ev) {
                        idx[rank & 127] = c16 | ((kb + gl) << 4);
                        cdl[rank & 127] = cdv;
                    }
                    known += (int)__popcll(mask);
                    kb += 4;
                }
                if (known == q_done) break;                  // every AM pick of the unit is served (a pass does not straddle units)
                asm volatile("" ::: "memory");               // LDS serves a wave in order; this orders the compiler
                const int hi = known < q_done + 16 ? known : q_done + 16;
                const int r = q_done + c16;
                const bool valid = r < hi;
                const int entry = idx[(valid ? r : q_done) & 127];
                const double cd_ev = cdl[(valid ? r : q_done) & 127];
                const int owner = entry & 15;
                const u32 sid_ev = (u32)__shfl((int)sid, owner, 64);
                const long long it_ev = a.iter0 + (entry >> 4);
                MfmaAcc<EPL> acc;
                if constexpr (DENSE) am_mfma_product<EPL, PTMI_PC_ACCV != 0>(a, valid, sid_ev, it_ev, cd_ev, d, UtG, false, d, PTMI_PC_SQ, true, acc, smem);
                else am_mfma_product<EPL, PTMI_PC_ACCV != 0>(a, valid, sid_ev, it_ev, cd_ev, d, PTMI_PC_UL, true, LD, PTMI_PC_SQ, true, acc, smem);
                // the slots of ranks q_done .. hi - 1 held ranks 16 below: wait until the stepper has read those
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&flg[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < hi - 16)
                    __builtin_amdgcn_s_sleep(PTMI_PC_SLEEP);
                if (valid) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) PTMI_PC_RING(r & 15)[gl * EPL + e] = acc.at(e);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&flg[0], hi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                q_done = hi;
            }
        }
        return;
    }
    // ---- the stepper
    if (PTMI_PC_SPRIO) __builtin_amdgcn_s_setprio(PTMI_PC_SPRIO);
    int base = 0;                                            // AM events of this wave's chains before the current step
    for (long long unit = (long long)(PERS ? blockIdx.x : logical_block()) * 4 + pair; unit < nunits; unit += ustride) {
    long long ch = unit * 16 + c16;
    const bool live = ch < nch;
    if (!live) ch = nch - 1;
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const int tg = a.temp0 + t;
    const double beta = a.beta[t];
    const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
    const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);
    const u32 sid = sid0 + (u32)tg;
    const double *DE = a.DE ? a.DE + (a.per_walker ? (size_t)w : 0) * (size_t)a.de_size * a.de_ld : nullptr;
    double *xrow = a.X + (size_t)ch * d;
    DrawBatch<STR> batch;
    double x[EPL], dq[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(x[e], xrow, e);
    double lnL = a.lnL[ch], lp = a.lp[ch];
    u32 nacc = 0, jp[PTMI_J_FUSED] = {0, 0, 0}, ja[PTMI_J_FUSED] = {0, 0, 0};
    const bool cold = live && tg == 0 && a.AM != nullptr;
    int am_row = a.am_row0;
    for (int k = 0; k < a.nsteps; ++k) {
        const long long it = a.iter0 + k;
        Draws dr;
        draws_for_step<STR, true, 1>(batch, dr, a, k, sid, sid0, gl, smem);
        const double log_u = dr.log_u;
        int jt;
        if constexpr (DENSE) jt = propose<G, EPL, true, STR, false>(a, it, sid, gl, cc, dr, UtG, false, PTMI_PC_SQ, DE, dq, true, false, smem);
        else jt = propose<G, EPL, true, STR, false>(a, it, sid, gl, cc, dr, PTMI_PC_UL, true, PTMI_PC_SQ, DE, dq, true, false, smem);
        {
            const bool is_am = live && jt == PTMI_J_AM;
            const u32 m16 = (u32)(__ballot(is_am) & 0xFFFFull);           // the chains' first lanes
            if (m16) {                                                    // wave-uniform
                const int my = base + (int)__popc(m16 & ((1u << c16) - 1u));
                const int need = base + (int)__popc(m16);
                int got = base;
                bool pending = is_am;
                while (got < need) {
                    const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flg[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const int avail = p < need ? p : need;
                    if (avail > got) {
                        if (pending && my < avail) {
#pragma unroll
                            for (int e = 0; e < EPL; ++e) dq[e] = PTMI_PC_RING(my & 15)[gl * EPL + e];
                            pending = false;
                        }
                        // the values are in registers before the slots are given back
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        got = avail;
                        if (lane == 0) __hip_atomic_store(&flg[1], got, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        __builtin_amdgcn_s_sleep(PTMI_PC_SLEEP);
                    }
                }
                base = need;
            }
        }
#pragma unroll
        for (int j = 0; j < PTMI_J_FUSED; ++j) jp[j] += (jt == j);
        // PT:605-612
        double nlp, nlnL = 0.0, nlnprob;
        if constexpr (DENSE) {
            // -1/2 r^T P r over the half table (eval_logl), the residual r = (x + dq) - mu formed where it is used: nothing of row
            // size lives beside x, dq and the accumulators
            if constexpr (PRI == PTMI_LOGP_FLAT) nlp = 0.0;
            else nlp = eval_logp_q<G, EPL, STR>(a, smem, gl, [&](int e) { return x[e] + dq[e]; });
            // unconditional: the LDS copy of the mean is zero beyond ndim, and so are x and dq there (a read under `i < d` was a branch
            // and an LDS round trip of its own for every slot, twice per step: fifty of them)
            auto rf = [&](int e) { return (x[e] + dq[e]) - PTMI_PC_MU[gl + G * e]; };
            MfmaAcc<EPL> pacc;
            mfma_half_tab_vecf<EPL>(PTMI_PC_UL, LD, d, rf, pacc);
            double pq = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) pq = __builtin_fma(rf(e), pacc.at(e), pq);
            nlnL = -grp_sum<G, STR>(pq);
            nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
        } else {
            double q[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) q[e] = x[e] + dq[e];
            if constexpr (PRI == PTMI_LOGP_FLAT) nlp = 0.0;
            else nlp = eval_logp<G, EPL, STR>(a, q, gl, smem);
            nlnL = eval_logl<G, EPL, LOGL, STR>(a, q, gl, nullptr);
            nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
        }
        // PT:615-622
        const double lnprob0 = beta * lnL + lp;
        const double diff = nlnprob - lnprob0 + 0.0;
        const bool accepted = diff > log_u;
        if (accepted) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                double inc = dq[e];
                asm volatile("" : "+v"(inc));
                x[e] = x[e] + inc;
            }
            lnL = nlnL;
            lp = nlp;
            nacc += 1;
#pragma unroll
            for (int j = 0; j < PTMI_J_FUSED; ++j) ja[j] += (jt == j);
        }
        if (cold && !(a.swap_last && k == a.nsteps - 1)) {          // PT:327-328
            am_store_step<G, EPL>(a, w, am_row, k, x, gl, d, accepted);
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
#pragma unroll
            for (int j = 0; j < PTMI_J_FUSED; ++j) {
                a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 0] += jp[j];
                a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 1] += ja[j];
            }
        }
    }
    }       // units
}

// Dense Gaussian likelihood, SCAM-only cycle, one eigenvector table for the whole block (pooled covariance, or the ranks of
// one walker filling the block): BASELINE configs[2] as bench.py --logl dense runs it.  Everything a step reads sits in
// LDS -- the precision matrix and the eigenvector table UNPADDED side by side, the mean and sqrt(eigenvalues) behind them
// (161.6 KB at d = 100: the whole CU) -- so one block of 512 threads puts TWO waves on every SIMD over one copy of the
// tables, and the matrix pipe works on one wave's P.r while the other wave draws, proposes and accepts.  To fit two waves
// into the register file nothing of row size is kept besides x, dq and the accumulators: r = (x + dq) - mu is formed where
// it is used (bit-identical both times).  An unpadded table makes the matrix instruction read past a row's end into the
// next row; those columns only feed the padding outputs i >= d, which are masked.  Strided lane layout as STAGE.
// doubles from the first even row to the first odd row of the dense kernel's LDS copy of P: past the even rows, = 16 (mod 32)
__host__ __device__ constexpr int dense_podd(int d) { return ((d + 1) / 2) * d + ((16 - (((d + 1) / 2) * d) % 32) + 32) % 32; }
#ifndef PTMI_DENSE_PF
#define PTMI_DENSE_PF 6
#endif
// matrix instructions of the half-table product (k-steps ascending, tiles 0 .. (4e+3)/16 within one)
constexpr int dense_pairs(int EPL) { int n = 0; for (int e = 0; e < EPL; ++e) n += (4 * e + 3) / 16 + 1; return n; }
template <int EPL, int BLK, bool BOX>
__global__ __launch_bounds__(BLK, BLK / 256) void mh_dense_scam_kernel(const KArgs a)
{
    constexpr int G = 4, CPB = BLK / G, NT = MfmaAcc<EPL>::NT;
    // the exact shape serves ndim = 4 EPL only: a compile-time d turns every table address of the step into one base register
    // plus an immediate offset (with a run-time d the 25 row addresses were loop invariants, hoisted and spilled: a scratch read
    // in front of the LDS read in front of the matrix instruction)
    constexpr bool EXACT = safe_slots(G, EPL) == EPL;
    const int d = EXACT ? 4 * EPL : a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int cib = wave * 16 + (lane & 15), gl = lane >> 4;
    const int c16 = lane & 15, g4 = lane >> 4;
    long long ch = (long long)logical_block() * CPB + cib;
    const bool live = ch < nch;
    if (!live) ch = nch - 1;
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const int tg = a.temp0 + t;
    const double beta = a.beta[t];
    const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
    const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);
    const u32 sid = sid0 + (u32)tg;
    double *xrow = a.X + (size_t)ch * d;
    ScamBatch<true> sbatch;

    extern __shared__ __attribute__((aligned(16))) double smem[];
    // P is split by row parity: even rows first, odd rows dense_podd(d) doubles further on -- an offset of 16 (mod 32) doubles,
    // so that the two rows a 32-lane group of a ds_read_b64 takes (k-step rows 4e + {0,1} and 4e + {2,3}) fall into
    // disjoint bank halves.  Unsplit (consecutive rows 100 doubles = 8 banks apart) every read of the matrix operand was a
    // two- to three-way bank conflict: 2.6 conflict cycles per LDS instruction (profiles/r02_dense_sq.txt).
    const int podd = dense_podd(d), pall = podd + (d / 2) * d;
#define PTMI_D_P (smem)
#define PTMI_D_U (smem + (size_t)pall)
#define PTMI_D_MU (smem + (size_t)pall + (size_t)d * d)
#define PTMI_D_SQ (smem + (size_t)pall + (size_t)d * d + 4 * EPL)
    {
        const long long ch0 = (long long)logical_block() * CPB;
        const size_t w0 = a.per_walker ? (size_t)((ch0 < nch ? ch0 : nch - 1) / nt) : 0;
        const double *Pg = a.logl_par + d + (size_t)d * d /* the half table Tl */, *Ug = a.Ut + w0 * d * d, *Sg = a.S + w0 * d, *mug = a.logl_par;
        for (int i = (int)threadIdx.x; i < d * d; i += BLK) {
            const int r = i / d, c = i % d;
            PTMI_D_P[((r & 1) ? podd : 0) + (r >> 1) * d + c] = Pg[i];
            PTMI_D_U[i] = Ug[i];
        }
        for (int i = ((d + 1) / 2) * d + (int)threadIdx.x; i < podd; i += BLK) PTMI_D_P[i] = 0.0;      // the gap is read past the last even row
        for (int i = (int)threadIdx.x; i < 4 * EPL; i += BLK) PTMI_D_MU[i] = i < d ? mug[i] : 0.0;
        for (int i = (int)threadIdx.x; i < d; i += BLK) PTMI_D_SQ[i] = det_sqrt(Sg[i]);
        box_table_fill<G, EPL>(a, smem, BLK);
        __syncthreads();
    }

    double x[EPL], dq[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(x[e], xrow, e);
    double lnL = a.lnL[ch], lp = a.lp[ch];
    u32 nacc = 0;
    const bool cold = live && tg == 0 && a.AM != nullptr;
    int am_row = a.am_row0;
    const int esteps = (d + 3) / 4 < EPL ? (d + 3) / 4 : EPL;
    // the exact shape (ndim = 4 EPL): every k-step and every output exists, so the product is ONE straight-line block -- with a
    // wave-uniform test around each k-step the software pipeline's row blocks were copied from "next" to "current" in every one
    // of them (7 v_mov_b64 per k-step: 160 of the step's 617 vector instructions)

    for (int k = 0; k < a.nsteps; ++k) {
        ScamDraw sd;
        scam_draws_for_step<true>(sbatch, sd, a, k, sid, gl, cc, d, [&](int kk) { return PTMI_D_SQ[kk]; });
        const double log_u = sd.log_u;
#pragma unroll
        for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(dq[e], PTMI_D_U + (size_t)sd.k * d, e);       // PT:868-873
#pragma unroll
        for (int e = 0; e < EPL; ++e) dq[e] = sd.amp * dq[e];
        // PT:605-612: prior on q = x + dq (BOX: the kernel is instantiated per prior kind -- the flat prior's dead test cost 7 %),
        // then -1/2 r^T P r with r = q - mu, formed ONCE (bit-identical wherever it is used)
        double nlp = 0.0;
        if (BOX) nlp = eval_logp_q<G, EPL, true>(a, smem, gl, [&](int e) { return x[e] + dq[e]; });
        MfmaAcc<EPL> acc;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc.t[tt] = ptmi_d4{0.0, 0.0, 0.0, 0.0};
        // the table is the half Tl (k >= i): in k-step e the tiles with 16 tt > 4 e + 3 hold zeros only and are skipped --
        // 91 matrix instructions per likelihood instead of 175
        auto tile = [&](int e, int tt) -> double {
            return PTMI_D_P[((g4 & 1) ? podd : 0) + (2 * e + (g4 >> 1)) * d + c16 + 16 * tt];
        };
        double rr[EPL];
        if constexpr (EXACT) {
            // the table operand of matrix instruction i + PF is requested before instruction i is issued (PF instructions =
            // 300+ cycles of LDS latency covered from the first k-step on, where a k-step is ONE instruction); the scheduling
            // barriers pin that order -- left alone the compiler put every read directly in front of its consumer
            constexpr int NP = dense_pairs(EPL), PF = PTMI_DENSE_PF;
            double av[NP];
            int pe = 0, pt = 0, pi = 0;                        // cursor of the requests (all compile-time after unrolling)
            auto request = [&]() {
                av[pi] = tile(pe, pt);
                ++pi;
                if (16 * (pt + 1) <= 4 * pe + 3) ++pt;
                else { pt = 0; ++pe; }
            };
#pragma unroll
            for (int i = 0; i < PF; ++i) request();
#pragma unroll
            for (int e = 0; e < EPL; ++e) rr[e] = (x[e] + dq[e]) - PTMI_D_MU[gl + G * e];
            __builtin_amdgcn_sched_barrier(0);
            int i = 0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    if (16 * tt > 4 * e + 3) continue;
                    if (pi < NP) request();
                    acc.t[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], rr[e], acc.t[tt], 0, 0, 0);
                    ++i;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // software pipeline of depth one (the row block of step e + 1 is in flight while step e multiplies); the scheduling
            // barrier keeps the compiler from hoisting all 7 * 26 table reads to the top
            double cur[NT], nxt[NT];
            auto fetch = [&](int e, double (&dst)[NT]) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    if (16 * tt <= 4 * e + 3) dst[tt] = tile(e, tt);
            };
#pragma unroll
            for (int e = 0; e < EPL; ++e) rr[e] = (x[e] + dq[e]) - PTMI_D_MU[gl + G * e];
            fetch(0, cur);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e < esteps) {                                     // wave-uniform
                    if (e + 1 < EPL && e + 1 < esteps) fetch(e + 1, nxt);
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        if (16 * tt <= 4 * e + 3) acc.t[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[tt], rr[e], acc.t[tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) cur[tt] = nxt[tt];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const double re = rr[e];
            const double ve = (EXACT || (gl + G * e) < d) ? acc.at(e) : 0.0;   // outputs past the row are padding
            p = __builtin_fma(re, ve, p);
        }
        const double nlnL = -grp_sum<G, true>(p);
        const double nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
        // PT:615-622
        const double lnprob0 = beta * lnL + lp;
        const double diff = nlnprob - lnprob0 + 0.0;
        const bool accepted = diff > log_u;
        if (accepted) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                double inc = dq[e];
                asm volatile("" : "+v"(inc));
                x[e] = x[e] + inc;
            }
            lnL = nlnL;
            lp = nlp;
            nacc += 1;
        }
        // PT:327-328 (the post-swap row of a swap iteration is written by the swap)
        if (cold && !(a.swap_last && k == a.nsteps - 1)) {
            am_store_step<G, EPL>(a, w, am_row, k, x, gl, d, accepted);
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
            a.jstat[(r * PTMI_J_NTYPES + PTMI_J_SCAM) * 2 + 0] += (u32)a.nsteps;
            a.jstat[(r * PTMI_J_NTYPES + PTMI_J_SCAM) * 2 + 1] += nacc;
        }
    }
}

// split path: proposal only / accept only, one iteration (host likelihood callbacks)
template <int G, int EPL, bool GRP>
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
    const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);
    const u32 sid = sid0 + (u32)(a.temp0 + t);
    const size_t wc = a.per_walker ? (size_t)w : 0;
    const double *Ut = a.Ut + wc * a.ngroups * d * d, *S = a.S + wc * a.ngroups * d;
    const double *DE = a.DE ? a.DE + wc * (size_t)a.de_size * a.de_ld : nullptr;
    const double *xrow = a.X + (size_t)ch * d;
    double x[EPL], dq[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) PTMI_ROW_LOAD(x[e], xrow, e);
    DrawBatch<false> batch;
    Draws dr;
    draws_for_step<false, true>(batch, dr, a, 0, sid, sid0, gl);
    double rp_val = 0.0;
    if (a.rp_draws != nullptr) {               // TEST HOOK (ptmi_test_replay): the proposal's draws as recorded from the reference
        const u64 *r = a.rp_draws + (size_t)ch * 4;
        dr.P0 = r[0]; dr.Q0 = r[1]; dr.Q1 = r[2];
        rp_val = __longlong_as_double((long long)r[3]);    // the SCAM normal (PT:873), or DE's scale uniform (PT:976)
        dr.z = rp_val;
        dr.pickw = (u32)(dr.P0 >> 32);
    }
    const double log_u = dr.log_u, u_acc = w2uniform_open(batch.P1());
    const int jt = propose<G, EPL, true, false, GRP>(a, a.iter0, sid, gl, cc, dr, Ut, false, S, DE, dq, false, true, nullptr, nullptr,
                                                     a.rp_draws != nullptr ? &rp_val : nullptr);
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
            if (am) am[am_pos(i, a.am_epl)] = v;
        }
    }
    if (gl == 0) {
        const size_t r = (size_t)w * nt + t;
        if (jt >= 0 && jt < PTMI_J_NTYPES) a.jstat[(r * PTMI_J_NTYPES + jt) * 2 + 0] += 1;
        if (am && a.AMflag) a.AMflag[(size_t)w * a.cov_update + (size_t)a.am_row0] = AMROW_KEY | (acc ? AMROW_NEW : 0ull);   // the split path stores every row
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
    const double lp = eval_logp<G, EPL, false>(a, x, gl);
    double lnL = -__builtin_inf();
    if (lp != -__builtin_inf()) lnL = eval_logl<G, EPL, LOGL, false>(a, x, gl, LOGL == PTMI_LOGL_DENSE ? a.logl_par + a.d + (size_t)a.d * a.d : nullptr);
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
    a.lds_u = 0;
    a.box_off = -1;
    a.tab_off = -1;
    constexpr size_t DRAWT = sizeof(double) * 128;           // the draw tables (ptmi_tables.h), copied behind the other tables where they fit
    // box prior: the bounds table goes behind the variant's other LDS tables when it still fits (else global reads)
    const size_t box_bytes = c.logp_kind == PTMI_LOGP_BOX ? sizeof(double) * box_table_doubles(G, EPL) : 0;
    auto even = [](size_t doubles) { return (doubles + 1) & ~(size_t)1; };
    if constexpr (G == 4 && !FULL && LOGL == PTMI_LOGL_DENSE) {
        // dense likelihood + SCAM-only + one table for the block: all tables unpadded in LDS, 512-thread blocks
        size_t lds2 = sizeof(double) * ((size_t)dense_podd(c.ndim) + (size_t)(c.ndim / 2) * c.ndim + (size_t)c.ndim * c.ndim + 4 * EPL + c.ndim);
        if (box_bytes && sizeof(double) * even(lds2 / sizeof(double)) + box_bytes <= 160 * 1024) {
            a.box_off = (int)even(lds2 / sizeof(double));
            lds2 = sizeof(double) * (size_t)a.box_off + box_bytes;
        }
        static const char *blk = getenv("PTMI_DENSE_BLK");              // measurement switch: 256, 512 (default) or 0 = the older kernel
        const int want = blk ? atoi(blk) : 512;
        const bool shared = c.ngroups <= 1 && (!c.cov_per_walker || c.ntemps % (want / G) == 0);
        if (want && shared && lds2 <= 160 * 1024) {
            const long long nch = (long long)c.nwalkers * c.ntemps;
            auto launch = [&](auto kern, int BLKv) -> int {
                if (lds2 > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                    if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds2, hipGetErrorString(e));
                }
                const int cpb = BLKv / G;
                hipLaunchKernelGGL(kern, dim3((unsigned)((nch + cpb - 1) / cpb)), dim3(BLKv), lds2, h->stream, a);
                h->last_variant = PTMI_VAR_STAGED | PTMI_VAR_LDS_UT | PTMI_VAR_DENSE_SCAM | (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0);
                return PTMI_OK;
            };
            if (c.logp_kind == PTMI_LOGP_BOX)
                return want == 256 ? launch(mh_dense_scam_kernel<EPL, 256, true>, 256) : launch(mh_dense_scam_kernel<EPL, 512, true>, 512);
            return want == 256 ? launch(mh_dense_scam_kernel<EPL, 256, false>, 256) : launch(mh_dense_scam_kernel<EPL, 512, false>, 512);
        }
    }
    if constexpr (WANTS) {
        const size_t tab = sizeof(double) * (size_t)(4 * ((c.ndim + 3) / 4)) * mfma_ld(EPL);   // zero-padded copy
        // SCAM-only cycles read one row of the chain's own table per step; with AM the block needs ONE table
        const bool one_table_per_block = c.ngroups <= 1 && (!FULL || !c.cov_per_walker || c.ntemps % (256 / G) == 0);
        size_t lds = LOGL == PTMI_LOGL_DENSE ? tab : 0;
        // (dense likelihood with AM in the cycle: the producer / consumer kernel below keeps the likelihood's table in LDS and reads the
        // eigenvectors from global memory at every ndim -- with both tables in LDS the smaller shapes fell back to the one-wave kernel:
        // 11.8 / 18.8 ms per 100 steps at 50 / 80-d against 14.2 at 100-d, now 7.5 / 11.6; up to 32-d the one-wave kernel stays ahead)
        const bool dense_pc = FULL && LOGL == PTMI_LOGL_DENSE && EPL >= 14 && c.w_am > 0 && one_table_per_block && getenv("PTMI_NO_PC") == nullptr;
        if (FULL && !dense_pc && lds + tab + sizeof(double) * c.ndim <= 160 * 1024) { a.lds_u = 1; lds += tab; }      // else Ut is read from global (L2)
        if (FULL) lds += sizeof(double) * c.ndim;                               // sqrt(eigenvalues)
        // AM queue of the block's four waves: 16 increments of 4 EPL + 2 doubles and 128 event entries each
        const size_t amq = FULL ? sizeof(double) * 4 * 16 * (4 * EPL + 2) + sizeof(int) * 4 * 128 : 0;
        if (sizeof(double) * even(lds / sizeof(double)) + amq <= 160 * 1024 && one_table_per_block) {
            if (FULL) {
                a.amq_off = (int)even(lds / sizeof(double));
                lds = sizeof(double) * (size_t)a.amq_off + amq;
                // through the queue when 1.28 x (expected AM events per step) / 16 is below the chance that a step of a wave
                // has an AM event at all; with one pick per walker every pass in place is full anyway
                const int Lw = c.w_host + c.w_scam + c.w_am + (h->de_on ? c.w_de : 0);
                const double f = Lw > 0 ? (double)c.w_am / (double)Lw : 0.0;
                a.amq_on = c.pick_mode != PTMI_PICK_WALKER && c.w_am > 0 && 1.28 * f < 1.0 - pow(1.0 - f, 16.0);
            }
            if (box_bytes && sizeof(double) * even(lds / sizeof(double)) + box_bytes <= 160 * 1024) {
                a.box_off = (int)even(lds / sizeof(double));
                lds = sizeof(double) * (size_t)a.box_off + box_bytes;
            }
            if (FULL && sizeof(double) * even(lds / sizeof(double)) + DRAWT <= 160 * 1024) {
                a.tab_off = (int)even(lds / sizeof(double));
                lds = sizeof(double) * (size_t)a.tab_off + DRAWT;
            }
            // AM in the cycle: the producer / consumer form (mh_pc_kernel) when its lists fit beside the ring;
            // PTMI_NO_PC=1 keeps the one-wave kernel (a measurement / test switch, same results)
            if constexpr (FULL) {
                const bool no_pc = getenv("PTMI_NO_PC") != nullptr;                      // read per launch: the tests switch it
                // cd of the listed events, the pairs' two counters; dense: the likelihood's mean behind them
                const size_t lists = sizeof(double) * 4 * 128 + sizeof(int) * 8 + (LOGL == PTMI_LOGL_DENSE ? sizeof(double) * 4 * EPL : 0);
                // iso / curved: the eigenvector table in LDS; dense: the likelihood's table there, the eigenvectors read from global memory
                const bool tables_ok = LOGL == PTMI_LOGL_DENSE ? !a.lds_u : a.lds_u != 0;
                if (c.w_am > 0 && tables_ok && a.tab_off >= 0 && !no_pc && lds + lists <= 160 * 1024 &&
                    (c.logp_kind == PTMI_LOGP_FLAT || (c.logp_kind == PTMI_LOGP_BOX && a.box_off >= 0))) {
                    // the lists sit directly behind the ring: everything placed behind the queue moves up by their size
                    const int shift = (int)(lists / sizeof(double));
                    if (a.box_off >= 0) a.box_off += shift;
                    a.tab_off += shift;
                    const size_t ldp = lds + lists;
                    // ONE table for the launch (pooled covariance): persistent blocks, one per CU (PTMI_PC_PERS=0: a block per 64 chains)
                    static int ncu = 0;
                    if (!ncu && (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess || ncu < 1)) ncu = 256;
                    const char *pe = getenv("PTMI_PC_PERS");
                    const bool pers = !c.cov_per_walker && !(pe && atoi(pe) == 0);
                    const int gridp = pers ? (grid < ncu ? grid : ncu) : grid;
                    auto launch_pc = [&](auto kp) -> int {
                        if (ldp > 64 * 1024) {
                            hipError_t e = hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldp);
                            if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", ldp, hipGetErrorString(e));
                        }
                        hipLaunchKernelGGL(kp, dim3(gridp), dim3(512), ldp, h->stream, a);
                        h->last_variant = PTMI_VAR_STAGED | PTMI_VAR_FULL | (a.lds_u ? PTMI_VAR_LDS_UT : 0) | PTMI_VAR_PC |
                                          (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0) | PTMI_VAR_LDS_DRAWT | (pers ? PTMI_VAR_PERSISTENT : 0);
                        return PTMI_OK;
                    };
                    if (pers) {
                        if (c.logp_kind == PTMI_LOGP_BOX) return launch_pc(mh_pc_kernel<EPL, LOGL, PTMI_LOGP_BOX, true>);
                        return launch_pc(mh_pc_kernel<EPL, LOGL, PTMI_LOGP_FLAT, true>);
                    }
                    if (c.logp_kind == PTMI_LOGP_BOX) return launch_pc(mh_pc_kernel<EPL, LOGL, PTMI_LOGP_BOX, false>);
                    return launch_pc(mh_pc_kernel<EPL, LOGL, PTMI_LOGP_FLAT, false>);
                }
            }
            auto kern = mh_steps_kernel<G, EPL, LOGL, FULL, true, false>;
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, h->stream, a);
            h->last_variant = PTMI_VAR_STAGED | (FULL ? PTMI_VAR_FULL : 0) | (a.lds_u ? PTMI_VAR_LDS_UT : 0) | (a.amq_on ? PTMI_VAR_AMQ : 0) |
                              (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0) | (a.tab_off >= 0 ? PTMI_VAR_LDS_DRAWT : 0);
            return PTMI_OK;
        }
    }
    // SCAM-only cycle, one eigenvector table per block, two blocks' tables fit the CU's LDS: read the direction from LDS
    if constexpr (!FULL && LOGL != PTMI_LOGL_DENSE) {
        size_t tab = sizeof(double) * ((size_t)c.ndim * c.ndim + c.ndim);
        const bool one_table = c.ngroups <= 1 && (!c.cov_per_walker || c.ntemps % (256 / G) == 0);
        static const bool off = getenv("PTMI_NO_ULDS") != nullptr;      // measurement switch: same results either way
        // ONE table for the whole launch (pooled covariance): persistent blocks, one per CU over one LDS copy of the table
        // (PTMI_ULDS_PERS = 0: the kernel with a table copy per block, a measurement / test switch; 768 threads -- three waves per
        // SIMD inside 168 registers -- measured 0.849 against 0.824 ms per 100 steps: the kernel is issue-bound)
        if constexpr (G == 4) {
            const char *pe = getenv("PTMI_ULDS_PERS");       // read per launch: the tests switch it
            const int pers = pe ? atoi(pe) : 512;
            if (c.ngroups <= 1 && !c.cov_per_walker && pers && !off && (c.logp_kind == PTMI_LOGP_FLAT || c.logp_kind == PTMI_LOGP_BOX)) {
                static int ncu = 0;
                if (!ncu) {
                    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess || ncu < 1) ncu = 256;
                }
                // table | sqrt(S) | bounds | draw tables
                size_t tp = sizeof(double) * even((size_t)c.ndim * c.ndim + c.ndim);
                a.box_off = -1;
                if (box_bytes) { a.box_off = (int)(tp / sizeof(double)); tp += box_bytes; }
                a.tab_off = (int)(tp / sizeof(double));
                tp += DRAWT;
                a.umax_off = (int)(tp / sizeof(double));              // box prior's fast path: max |U| of the table (one double)
                tp += 16;
                auto launch_p = [&](auto kernp, int BLKv) -> int {
                    if (tp > 64 * 1024) {
                        hipError_t e = hipFuncSetAttribute((const void *)kernp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp);
                        if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", tp, hipGetErrorString(e));
                    }
                    const long long nunits = ((long long)c.nwalkers * c.ntemps + 15) / 16, wpb = BLKv / 64;
                    const long long nb = (nunits + wpb - 1) / wpb;
                    hipLaunchKernelGGL(kernp, dim3((unsigned)(nb < ncu ? nb : ncu)), dim3(BLKv), tp, h->stream, a);
                    h->last_variant = PTMI_VAR_LDS_UT | PTMI_VAR_PERSISTENT | (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0) | PTMI_VAR_LDS_DRAWT;
                    return PTMI_OK;
                };
                if (tp <= 160 * 1024) {
                    if (c.logp_kind == PTMI_LOGP_BOX) return launch_p(mh_steps_kernel<G, EPL, LOGL, false, false, false, true, 512, PTMI_LOGP_BOX>, 512);
                    return launch_p(mh_steps_kernel<G, EPL, LOGL, false, false, false, true, 512, PTMI_LOGP_FLAT>, 512);
                }
                a.box_off = -1;
                a.tab_off = -1;
            }
        }
        if (box_bytes) {
            // the bounds table instead of sqrt(S) when both do not fit twice per CU (the kernel then takes the root per step)
            const size_t ut = sizeof(double) * even((size_t)c.ndim * c.ndim);
            if (2 * (sizeof(double) * even(tab / sizeof(double)) + box_bytes) <= 160 * 1024) {
                a.box_off = (int)even(tab / sizeof(double));
                tab = sizeof(double) * (size_t)a.box_off + box_bytes;
            } else if (2 * (ut + box_bytes) <= 160 * 1024) {
                a.box_off = (int)(ut / sizeof(double));
                tab = ut + box_bytes;
            }
        }
        if (one_table && 2 * tab <= 160 * 1024 && !off) {
            static const bool no_ldst = getenv("PTMI_NO_ULDS_DRAWT") != nullptr;     // measurement switch: same results either way
            if (!no_ldst && 2 * (sizeof(double) * even(tab / sizeof(double)) + DRAWT) <= 160 * 1024) {
                a.tab_off = (int)even(tab / sizeof(double));
                tab = sizeof(double) * (size_t)a.tab_off + DRAWT;
            }
            // with the draw tables placed the kernel reads them from LDS at compile time (no vector-memory wait in the step loop: see
            // the persistent kernel); without room for them the run-time choice stays
            auto launch_u = [&](auto kern) -> int {
                if (tab > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab);
                    if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", tab, hipGetErrorString(e));
                }
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), tab, h->stream, a);
                return PTMI_OK;
            };
            if (int rc = a.tab_off >= 0 ? launch_u(mh_steps_kernel<G, EPL, LOGL, false, false, false, true, 0, -1, true>)
                                        : launch_u(mh_steps_kernel<G, EPL, LOGL, false, false, false, true>)) return rc;
            h->last_variant = PTMI_VAR_LDS_UT | (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0) | (a.tab_off >= 0 ? PTMI_VAR_LDS_DRAWT : 0);
            return PTMI_OK;
        }
    }
    a.box_off = box_bytes ? 0 : -1;               // no other table in LDS
    if constexpr (G > 4 && !FULL) {
        if (a.UtPad != nullptr && a.ut_pad_ld == G * EPL && c.ngroups <= 1) {
            // the prior kind as a template parameter: the flat prior's instantiation holds no box-test code (nor the proposal q as an array)
            if (c.logp_kind == PTMI_LOGP_FLAT)
                hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, false, false, false, false, 0, PTMI_LOGP_FLAT, false, true>), dim3(grid), dim3(256), 0, h->stream, a);
            else if (G == 64 && c.logp_kind == PTMI_LOGP_BOX && a.ut_absmax != nullptr)
                hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, false, false, false, false, 0, PTMI_LOGP_BOX, false, true>), dim3(grid), dim3(256), box_bytes, h->stream, a);
            else
                hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, false, false, false, false, 0, -1, false, true>), dim3(grid), dim3(256), box_bytes, h->stream, a);
            h->last_variant = PTMI_VAR_UTPAD | (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0);
            return PTMI_OK;
        }
    }
    if (c.ngroups > 1) hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, FULL, false, true>), dim3(grid), dim3(256), box_bytes, h->stream, a);
    else hipLaunchKernelGGL((mh_steps_kernel<G, EPL, LOGL, FULL, false, false>), dim3(grid), dim3(256), box_bytes, h->stream, a);
    h->last_variant = (FULL ? PTMI_VAR_FULL : 0) | (c.ngroups > 1 ? PTMI_VAR_GROUPS : 0) | (a.box_off >= 0 ? PTMI_VAR_LDS_BOX : 0);
    return PTMI_OK;
}
