// ptmi_gj.inc.h -- gradient jumps (HMC, NUTS) on the device for the built-in likelihoods.
// Behaviour: PTMCMCSampler/nutsjump.py of the reference (NJ:<lines>): GradientJump whitening NJ:51-54,71-90,
// leapfrog NJ:149-169, HMCJump NJ:238-291, NUTSJump NJ:379-840 (find_reasonable_epsilon NJ:435-463, stop_criterion
// NJ:465-493, build_tree NJ:495-652, __call__ + dual averaging NJ:654-840).  The reference takes the gradients from
// the user's Python callbacks; here they are analytic for the built-in families, and the recursion of build_tree is
// unrolled into a loop with an explicit stack of pending left subtrees in global scratch (same merges, same order of
// draws).  Checked bit for bit against oracle/ptmcmc_oracle.c (nuts_call / hmc_call), which replays the reference.
#pragma once
#include "ptmi_common.h"


// the three whitening tables of the block, staged once per launch ([3][d][d], 9.6 KB at d = 20)
extern __shared__ __attribute__((aligned(16))) double gj_lds[];
enum { GJT_BACKWARD = 0, GJT_FORWARD = 1, GJT_GRADIENT = 2 };

template <int G, int EPL, int LOGL>
struct GradJump {
    const KArgs &a;
    const int gl, d;
    const long long ch, nch;
    const double beta;
    const long long it;
    const u32 sid;
    u32 nm = 0, ns = 0;          // momenta / scalar draws used so far in this call
    u32 nleap = 0;               // leapfrogs of this call (statistics: gj[..][GJ_NLEAP])

    __device__ __forceinline__ GradJump(const KArgs &a_, int gl_, long long ch_, double beta_, long long it_, u32 sid_)
        : a(a_), gl(gl_), d(a_.d), ch(ch_), nch((long long)a_.W * a_.nt), beta(beta_), it(it_), sid(sid_) {}

    // ---- scratch: [slot][e][chain][lane] so that a wave's access is one contiguous run
    __device__ __forceinline__ void vload(int slot, double (&v)[EPL]) const
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[e] = a.gj_scr[((size_t)(slot * EPL + e) * nch + ch) * G + gl];
    }
    __device__ __forceinline__ void vstore(int slot, const double (&v)[EPL]) const
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e) a.gj_scr[((size_t)(slot * EPL + e) * nch + ch) * G + gl] = v[e];
    }
    __device__ __forceinline__ double &scal(int level, int k) const { return a.gj_scal[((size_t)(level * GJS_SCALARS + k)) * nch + ch]; }

    // ---- draws (every lane of the chain evaluates the same counters)
    __device__ __forceinline__ void momenta(double (&r)[EPL])           // NJ:92-94
    {
        const u32 block = nm++;
#pragma unroll
        for (int e = 0; e < EPL; ++e) r[e] = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; e += 2) {                               // directions k and k + G share one Box-Muller
            const int k = gl + G * e;
            if (k < d) {
                u64 e0, e1;
                philox_words(a.seed, (u64)it, sid, SLOT_GJ + 4096u * block + (u32)k, e0, e1);
                const double rr = det_sqrt(-2.0 * det_log(w2uniform_open(e0)));
                double sn, cs;
                det_sincos2pi(w2uniform(e1), sn, cs);
                r[e] = rr * cs;
                if (e + 1 < EPL && k + G < d) r[e + 1] = rr * sn;
            }
        }
    }
    __device__ __forceinline__ u64 scalar_word()
    {
        u64 w0, w1;
        philox_words(a.seed, (u64)it, sid, SLOT_GJS + ns++, w0, w1);
        return w0;
    }
    __device__ __forceinline__ double uniform() { return w2uniform(scalar_word()); }
    __device__ __forceinline__ double exponential() { return -det_log(w2uniform_open(scalar_word())); }
    __device__ __forceinline__ int randint(int lo, int hi) { return lo + (int)w2index(scalar_word(), (u64)(hi - lo)); }

    // ---- linear algebra in the chain's lane layout (element i = gl + G e; pads are zero)
    __device__ __forceinline__ double dot(const double (&x)[EPL], const double (&y)[EPL]) const
    {
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p = __builtin_fma(x[e], y[e], p);
        return group_sum<G>(p);
    }
    // out[i] = sum_k T[k][i] v[k], k ascending, one fma per term.  WHICH >= 0: a whitening table in LDS (the pointer
    // is formed from the LDS symbol here so that the reads are ds_read, not flat); WHICH < 0: the global table Tg.
    template <int WHICH>
    __device__ __forceinline__ void tab_vec(const double *Tg, const double (&v)[EPL], double (&out)[EPL]) const
    {
        // quads (ndim <= 32) find the tables in LDS; the wider layouts read them through L2
        const double *T = WHICH < 0 ? Tg : (G == 4 ? gj_lds + (size_t)WHICH * d * d : a.gj_tab + (size_t)WHICH * d * d);
        double acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.0;
#pragma unroll
        for (int e2 = 0; e2 < EPL; ++e2) {
#pragma unroll 1
            for (int src = 0; src < G; ++src) {
                const int k = src + G * e2;
                if (k >= d) break;
                const double vk = group_bcast_lane<G>(v[e2], src);
                const double *row = T + (size_t)k * d;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int i = gl + G * e;
                    if (i < d) acc[e] = __builtin_fma(row[i], vk, acc[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) out[e] = acc[e];
    }

    // logl and its gradient (the value is eval_logl's, operation for operation)
    __device__ __forceinline__ double logl_grad(const double (&x)[EPL], double (&g)[EPL]) const
    {
        if (LOGL == PTMI_LOGL_ISO) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) g[e] = -x[e];
            return -0.5 * dot(x, x);
        } else if (LOGL == PTMI_LOGL_DENSE) {
            const double *mu = a.logl_par, *Pt = a.logl_par + d;
            double r[EPL], v[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e;
                r[e] = i < d ? x[e] - mu[i] : 0.0;
            }
            tab_vec<-1>(Pt, r, v);
#pragma unroll
            for (int e = 0; e < EPL; ++e) g[e] = -v[e];
            return -0.5 * dot(r, v);
        } else {
            double p = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e;
                const double other = dppf64<0xB1>(x[e]);                 // partner lane (gl ^ 1)
                const bool even = !(gl & 1);
                const bool pair = even ? i + 1 < d : i < d;              // both members of the pair exist
                const double xx = even ? x[e] : other, y = even ? other : x[e];
                double t = 0.0, gv = 0.0;
                if (pair) {
                    const double x2 = xx * xx;
                    const double gg = 9.0 + 4.0 * x2 + 9.0 * y;
                    const double l0 = -x2 - gg * gg;
                    const double ym = y - 2.0;
                    const double l1 = -8.0 * x2 - 8.0 * (ym * ym);
                    const double e0 = det_exp(l0), e1 = 0.5 * det_exp(l1);
                    const double sum = e0 + e1;
                    if (even) {
                        t = det_log(sum);
                        const double d0x = -2.0 * xx - 16.0 * gg * xx, d1x = -16.0 * xx;
                        gv = (e0 * d0x + e1 * d1x) / sum;
                    } else {
                        const double d0y = -18.0 * gg, d1y = -16.0 * ym;
                        gv = (e0 * d0y + e1 * d1y) / sum;
                    }
                }
                g[e] = gv;
                p = __builtin_fma(t, 1.0, p);
            }
            return group_sum<G>(p);
        }
    }
    __device__ __forceinline__ double logp(const double (&x)[EPL]) const
    {
        if (a.logp_kind == PTMI_LOGP_BOX) {
            const double *lo = a.logp_par, *hi = a.logp_par + d;
            bool ok = true;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e;
                if (i < d) ok = ok && (lo[i] <= x[e]) && (hi[i] >= x[e]);
            }
            return group_all<G>(ok) ? 0.0 : -__builtin_inf();
        }
        return 0.0;
    }
    // beta*logl + logp and its gradient in the whitened coordinates (NJ:71-90)
    __device__ __forceinline__ double func_grad_white(const double (&q)[EPL], double (&gradw)[EPL]) const
    {
        double x[EPL], g[EPL];
        tab_vec<GJT_BACKWARD>(nullptr, q, x);                                         // backward: x = L^T q
        const double ll = logl_grad(x, g);
        const double lp = logp(x);                                       // the built-in priors have zero gradient
#pragma unroll
        for (int e = 0; e < EPL; ++e) g[e] = beta * g[e] + 0.0;
        tab_vec<GJT_GRADIENT>(nullptr, g, gradw);
        return beta * ll + lp;
    }
    __device__ __forceinline__ double joint_of(double logl, const double (&r)[EPL]) const { return logl - 0.5 * dot(r, r); }

    // NJ:149-169; outputs may alias the inputs
    __device__ __forceinline__ double leapfrog(const double (&theta)[EPL], const double (&r)[EPL], const double (&grad)[EPL], double eps,
                                               double (&to)[EPL], double (&ro)[EPL], double (&go)[EPL])
    {
        nleap += 1;
        const double he = 0.5 * eps;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const double rh = r[e] + he * grad[e];
            ro[e] = rh;
            to[e] = theta[e] + eps * rh;
        }
        const double lpp = func_grad_white(to, go);
#pragma unroll
        for (int e = 0; e < EPL; ++e) ro[e] = ro[e] + he * go[e];
        return lpp;
    }
    __device__ __forceinline__ bool keep_going(const double (&tm)[EPL], const double (&tp)[EPL], const double (&rm)[EPL],
                                               const double (&rp)[EPL]) const                  // NJ:465-493
    {
        double dt[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) dt[e] = tp[e] - tm[e];
        const double x = dot(dt, rm), y = dot(dt, rp);
        return (x >= 0.0) & (y >= 0.0);
    }
    __device__ __forceinline__ bool any_inf(const double (&v)[EPL]) const
    {
        bool fin = true;
#pragma unroll
        for (int e = 0; e < EPL; ++e) fin = fin && !__builtin_isinf(v[e]);
        return !group_all<G>(fin);
    }

    // ------------------------------------------------------------------ HMC (NJ:238-291)
    __device__ __forceinline__ double hmc(double *st, const double (&x)[EPL], double (&qout)[EPL])
    {
        double q[EPL], p[EPL], grad[EPL];
        st[GJ_HITER] += 1.0;
        tab_vec<GJT_FORWARD>(nullptr, x, q);                         // forward
        const double logp0 = func_grad_white(q, grad);
        momenta(p);
        const double joint0 = joint_of(logp0, p);
        const int nsteps = randint(a.hmc_min, a.hmc_max);
        double joint1 = joint0;
        for (int k = 0; k < nsteps; ++k) {
            const double logp1 = leapfrog(q, p, grad, a.hmc_eps, q, p, grad);
            joint1 = joint_of(logp1, p);
            if (joint1 - 1000.0 < joint0) break;                         // NJ:284-286
        }
        tab_vec<GJT_BACKWARD>(nullptr, q, qout);
        return joint1 - joint0;
    }

    // ------------------------------------------------------------------ NUTS
    // NJ:435-463; both loops bounded at 100 turns (the reference's are not)
    __device__ __forceinline__ double find_reasonable_epsilon(const double (&theta0)[EPL], const double (&grad0)[EPL], double logp0)
    {
        double r0[EPL], tp[EPL], rp[EPL], gp[EPL];
        double eps = 1.0;
        momenta(r0);
        double logpp = leapfrog(theta0, r0, grad0, eps, tp, rp, gp);
        const bool ginf = any_inf(gp);                                   // not refreshed in the loop (NJ:449-452)
        double k = 1.0;
        for (int n = 0; n < 100 && (__builtin_isinf(logpp) || ginf); ++n) {
            k *= 0.5;
            logpp = leapfrog(theta0, r0, grad0, eps * k, tp, rp, gp);
        }
        eps = 0.5 * k * eps;
        double ap = det_exp(joint_of(logpp, rp) - joint_of(logp0, r0));
        const bool up = ap > 0.5;
        for (int n = 0; n < 100 && ((up ? ap : 1.0 / ap) > (up ? 0.5 : 2.0)); ++n) {
            eps = eps * (up ? 2.0 : 0.5);
            logpp = leapfrog(theta0, r0, grad0, eps, tp, rp, gp);
            ap = det_exp(joint_of(logpp, rp) - joint_of(logp0, r0));
        }
        return eps;
    }

    struct Tree {                // what the current (sub)tree hands upward; its growth end is the caller's (tg, rg, gg)
        double far_t[EPL], far_r[EPL], cand_t[EPL], cand_g[EPL];
        double logp, alpha;
        long long n, nalpha;
        int s;
    };

    // NJ:495-652 as a loop: leaves are generated left to right in direction v from the growth end; a finished subtree
    // is merged with the pending left sibling of the same height on the stack, or waits there for its right sibling.
    // A left subtree that stopped (s = 0) is handed up unchanged to the height of the next pending sibling (or the root).
    __device__ __forceinline__ void build_tree(double (&tg)[EPL], double (&rg)[EPL], double (&gg)[EPL], double logu, int v, int j,
                                               double eps, double joint0, Tree &cur)
    {
        int sp = 0;
        for (;;) {
            const double logpp = leapfrog(tg, rg, gg, (double)v * eps, tg, rg, gg);
            const double joint = joint_of(logpp, rg);
            cur.n = logu < joint;
            cur.s = (logu - 1000.0) < joint;
#pragma unroll
            for (int e = 0; e < EPL; ++e) { cur.far_t[e] = tg[e]; cur.far_r[e] = rg[e]; cur.cand_t[e] = tg[e]; cur.cand_g[e] = gg[e]; }
            cur.logp = logpp;
            const double ex = det_exp(joint - joint0);
            cur.alpha = ex < 1.0 ? ex : 1.0;                             // Python's min(1.0, e): 1.0 when e is NaN
            cur.nalpha = 1;
            int h = 0;
            for (;;) {
                const int top_h = sp > 0 ? (int)scal(sp - 1, GJS_H) : -1;
                if (sp > 0 && top_h == h) {                              // cur is the right sibling of the stack top
                    --sp;
                    const int base = GJV_TOP + sp * GJL_VECS;
                    const long long tn = (long long)scal(sp, GJS_N);
                    const long long tot = tn + cur.n;
                    const double den = (double)tot > 1.0 ? (double)tot : 1.0;
                    const bool take_u = uniform() < (double)cur.n / den;
                    if (!take_u) {
                        vload(base + GJL_CAND_T, cur.cand_t);
                        vload(base + GJL_CAND_G, cur.cand_g);
                        cur.logp = scal(sp, GJS_LOGP);
                    }
                    vload(base + GJL_FAR_T, cur.far_t);
                    vload(base + GJL_FAR_R, cur.far_r);
                    cur.n = tot;
                    const bool go = v == 1 ? keep_going(cur.far_t, tg, cur.far_r, rg) : keep_going(tg, cur.far_t, rg, cur.far_r);
                    cur.s = cur.s && go;                                 // the popped tree has s = 1
                    cur.alpha = scal(sp, GJS_ALPHA) + cur.alpha;
                    cur.nalpha = (long long)scal(sp, GJS_NALPHA) + cur.nalpha;
                    h += 1;
                    continue;
                }
                if (h == j) return;
                if (cur.s == 0) {
                    if (sp == 0) return;
                    h = top_h;
                    continue;
                }
                const int base = GJV_TOP + sp * GJL_VECS;                // push: wait for the right sibling
                vstore(base + GJL_FAR_T, cur.far_t);
                vstore(base + GJL_FAR_R, cur.far_r);
                vstore(base + GJL_CAND_T, cur.cand_t);
                vstore(base + GJL_CAND_G, cur.cand_g);
                scal(sp, GJS_LOGP) = cur.logp;                           // every lane of the chain writes the same values
                scal(sp, GJS_N) = (double)cur.n;
                scal(sp, GJS_ALPHA) = cur.alpha;
                scal(sp, GJS_NALPHA) = (double)cur.nalpha;
                scal(sp, GJS_H) = (double)h;
                __threadfence_block();
                ++sp;
                break;
            }
        }
    }

    // NUTSJump.__call__ (NJ:654-840), force_trajlen = force_epsilon = None.  Returns qxy.
    __device__ __forceinline__ double nuts(double *st, const double (&x)[EPL], double (&qout)[EPL])
    {
        double q[EPL], grad[EPL], r0[EPL];
        st[GJ_NITER] += 1.0;
        tab_vec<GJT_FORWARD>(nullptr, x, q);
        const double logp0 = func_grad_white(q, grad);
        if (st[GJ_HAVE_EPS] == 0.0) {
            st[GJ_EPS] = find_reasonable_epsilon(q, grad, logp0);
            st[GJ_MU] = det_log(10.0 * st[GJ_EPS]);
            st[GJ_HAVE_EPS] = 1.0;
        }
        momenta(r0);
        const double joint = joint_of(logp0, r0);
        const double logu = joint - exponential();
        double lnprob = logp0;
        vstore(GJV_SAMPLE, q);
        vstore(GJV_TM, q); vstore(GJV_RM, r0); vstore(GJV_GM, grad);
        vstore(GJV_TP, q); vstore(GJV_RP, r0); vstore(GJV_GP, grad);
        int j = 0, s = 1;
        long long n = 1;
        double alpha = 0.0;
        long long nalpha = 1;
        const double eps = st[GJ_EPS];
        while (s == 1) {
            const int dir = 2 * (int)(uniform() < 0.5) - 1;
            double tg[EPL], rg[EPL], gg[EPL];
            const int eb = dir == -1 ? GJV_TM : GJV_TP;
            vload(eb, tg); vload(eb + 1, rg); vload(eb + 2, gg);
            Tree t;
            build_tree(tg, rg, gg, logu, dir, j, eps, joint, t);
            vstore(eb, tg); vstore(eb + 1, rg); vstore(eb + 2, gg);
            if (t.s == 1) {
                const double ratio = (double)t.n / (double)n;
                if (uniform() < (1.0 < ratio ? 1.0 : ratio)) { vstore(GJV_SAMPLE, t.cand_t); lnprob = t.logp; }
            }
            n += t.n;
            double to[EPL], ro[EPL];
            const int ob = dir == -1 ? GJV_TP : GJV_TM;                  // the other end
            vload(ob, to); vload(ob + 1, ro);
            const bool go = dir == -1 ? keep_going(tg, to, rg, ro) : keep_going(to, tg, ro, rg);
            s = t.s && go;
            alpha = t.alpha;
            nalpha = t.nalpha;
            j += 1;
            if (j > a.nuts_maxdepth) s = 0;                              // cap (not in the reference)
        }
        // dual averaging (NJ:805-816): gamma = 0.05, t0 = 10, kappa = 0.75
        const double it_call = st[GJ_NITER];
        double eta = 1.0 / (it_call + 10.0);
        st[GJ_HBAR] = (1.0 - eta) * st[GJ_HBAR] + eta * (a.nuts_delta - alpha / (double)nalpha);
        if (it <= (long long)a.gj_nburn) {
            st[GJ_EPS] = det_exp(st[GJ_MU] - det_sqrt(it_call) / 0.05 * st[GJ_HBAR]);
            eta = det_exp(-0.75 * det_log(it_call));
            st[GJ_EPSBAR] = det_exp((1.0 - eta) * det_log(st[GJ_EPSBAR]) + eta * det_log(st[GJ_EPS]));
        } else {
            st[GJ_EPS] = st[GJ_EPSBAR];
        }
        double sample[EPL];
        vload(GJV_SAMPLE, sample);
        tab_vec<GJT_BACKWARD>(nullptr, sample, qout);
        return logp0 - lnprob;                                           // undoes the outer Hastings ratio (NJ:838)
    }
};

// Fused MH steps with the gradient jumps in the cycle: the non-staged full kernel (ptmi_mh.inc.h) plus the NUTS / HMC
// branch.  Every chain group runs its nsteps iterations on its own, so a long NUTS tree delays only its wave for that
// iteration and the launch costs the longest SUM over iterations, not the sum of the per-iteration maxima.
// One wave per block: there is no block-level cooperation, and single-wave blocks let the dispatcher backfill the
// SIMDs as soon as a wave's chains are through their (very unequal) trees.
constexpr int GJ_BLOCK = 64;
template <int G, int EPL, int LOGL>
__global__ __launch_bounds__(GJ_BLOCK) void mh_steps_gj_kernel(const KArgs a)
{
    constexpr int CPB = GJ_BLOCK / G;
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    if (G == 4) {
        for (int i = (int)threadIdx.x; i < 3 * d * d; i += GJ_BLOCK) gj_lds[i] = a.gj_tab[i];
        __syncthreads();
    }
    const long long cslot = (long long)blockIdx.x * CPB + (int)(threadIdx.x / G);
    if (cslot >= nch) return;                    // no block-wide synchronisation below: whole chain groups may leave
    // The chains of a wave run in lock step: every iteration costs the wave its longest tree, and the launch ends with its
    // slowest wave.  On the curved likelihood a per cent of the ranks keep a small NUTS step size (trees of ~100
    // leapfrogs) while the rest run away to huge ones (one leapfrog): the host deals the chains over the waves by step
    // size (gj_order_*), longest trees first and one per wave, so that no wave has to add up several long trees per
    // iteration (tools/gj_census.py).  Which lanes host a chain does not enter its arithmetic.
    const long long ch = a.gj_order ? (long long)a.gj_order[cslot] : cslot;
    const int gl = (int)(threadIdx.x % G);
    const int w = (int)(ch / nt);
    const int t = a.temp_of[ch];
    const int tg = a.temp0 + t;
    const double beta = a.beta[t];
    const ChainConst cc = chain_const(a.temps_mh[t], beta, d);
    const u32 sid0 = (u32)((u64)(a.walker0 + w) * (u32)a.ntg);
    const u32 sid = sid0 + (u32)tg;
    DrawBatch<false> batch;
    const size_t wc = a.per_walker ? (size_t)w : 0;
    const double *Ut = a.Ut + wc * d * d, *S = a.S + wc * d;
    const double *DE = a.DE ? a.DE + wc * (size_t)a.de_size * a.de_ld : nullptr;
    const double *PtG = LOGL == PTMI_LOGL_DENSE ? a.logl_par + d : nullptr;
    double *xrow = a.X + (size_t)ch * d;
    double *stg = a.gj + ((size_t)w * nt + t) * GJ_NSTATE;

    double x[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = gl + G * e;
        x[e] = i < d ? xrow[i] : 0.0;
    }
    double lnL = a.lnL[ch], lp = a.lp[ch];
    u32 nacc = 0, jp[PTMI_J_NTYPES] = {0, 0, 0, 0, 0}, ja[PTMI_J_NTYPES] = {0, 0, 0, 0, 0};
    const bool cold = tg == 0 && a.AM != nullptr;
    int am_row = a.am_row0;

    for (int k = 0; k < a.nsteps; ++k) {
        const long long it = a.iter0 + k;
        double q[EPL], qxy = 0.0;
        Draws dr;
        draws_for_step<false, true>(batch, dr, a, k, sid, sid0, gl);
        const double log_u = dr.log_u;
        const int jt = propose<G, EPL, true, false, false, true>(a, it, sid, gl, cc, dr, Ut, false, S, DE, q);
        if (jt == PTMI_J_NUTS || jt == PTMI_J_HMC) {
            GradJump<G, EPL, LOGL> gj(a, gl, ch, beta, it, sid);
            double st[GJ_NSTATE];
#pragma unroll
            for (int j = 0; j < GJ_NSTATE; ++j) st[j] = stg[j];
            qxy = jt == PTMI_J_NUTS ? gj.nuts(st, x, q) : gj.hmc(st, x, q);
            st[GJ_NLEAP] += (double)gj.nleap;
            if (gl == 0) {
#pragma unroll
                for (int j = 0; j < GJ_NSTATE; ++j) stg[j] = st[j];
            }
            __threadfence_block();               // the chain's other lanes read the state at its next gradient jump
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) q[e] = x[e] + q[e];            // propose() returned the increment
        }
#pragma unroll
        for (int j = 0; j < PTMI_J_NTYPES; ++j) jp[j] += (jt == j);
        // PT:605-612
        const double nlp = eval_logp<G, EPL, false>(a, q, gl);
        const double nlnL = eval_logl<G, EPL, LOGL, false>(a, q, gl, PtG);
        const double nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
        // PT:615-622
        const double lnprob0 = beta * lnL + lp;
        const double diff = nlnprob - lnprob0 + qxy;
        if (diff > log_u) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) x[e] = q[e];
            lnL = nlnL;
            lp = nlp;
            nacc += 1;
#pragma unroll
            for (int j = 0; j < PTMI_J_NTYPES; ++j) ja[j] += (jt == j);
        }
        // PT:327-328 (the post-swap row of a swap iteration is written by the swap)
        if (cold && !(a.swap_last && k == a.nsteps - 1)) {
            double *am = a.AM + ((size_t)w * a.cov_update + (size_t)am_row) * d;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e;
                if (i < d) am[i] = x[e];
            }
            if (a.AMaux && gl == 0) {
                double *ax = a.AMaux + ((size_t)w * a.cov_update + (size_t)am_row) * 2;
                ax[0] = lnL;
                ax[1] = lp;
            }
        }
        am_row = am_row + 1 == a.cov_update ? 0 : am_row + 1;
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = gl + G * e;
        if (i < d) xrow[i] = x[e];
    }
    if (gl == 0) {
        a.lnL[ch] = lnL;
        a.lp[ch] = lp;
        const size_t r = (size_t)w * nt + t;
        a.nacc[r] += nacc;
#pragma unroll
        for (int j = 0; j < PTMI_J_NTYPES; ++j) {
            a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 0] += jp[j];
            a.jstat[(r * PTMI_J_NTYPES + j) * 2 + 1] += ja[j];
        }
    }
}
