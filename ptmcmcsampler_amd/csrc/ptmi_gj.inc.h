// ptmi_gj.inc.h -- gradient jumps (HMC, NUTS) on the device for the built-in likelihoods.
// Behaviour: PTMCMCSampler/nutsjump.py of the reference (NJ:<lines>): GradientJump whitening NJ:51-54,71-90,
// leapfrog NJ:149-169, HMCJump NJ:238-291, NUTSJump NJ:379-840 (find_reasonable_epsilon NJ:435-463, stop_criterion
// NJ:465-493, build_tree NJ:495-652, __call__ + dual averaging NJ:654-840).  The reference takes the gradients from
// the user's Python callbacks; here they are analytic for the built-in families, and the recursion of build_tree is
// unrolled into a loop with an explicit stack of pending left subtrees in global scratch (same merges, same order of
// draws).  Checked bit for bit against oracle/ptmcmc_oracle.c (nuts_call / hmc_call), which replays the reference.
#pragma once
#include "ptmi_common.h"


// The block's LDS (one wave per block): the three whitening tables, staged once per launch ([3][d][d], 9.6 KB at d = 20,
// 4-lane shapes only); the entries of the lowest heights of the tree stack (KArgs::gj_stack_off, gj_lds_levels); the
// bounds of a box prior (KArgs::box_off).
extern __shared__ __attribute__((aligned(16))) double gj_lds[];
enum { GJT_BACKWARD = 0, GJT_FORWARD = 1, GJT_GRADIENT = 2 };
// doubles of one stack level in LDS: four chain vectors and four scalars for each of the wave's 64 lanes
constexpr int gj_level_doubles(int EPL) { return (GJL_VECS * EPL + 4) * 64; }

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
    // out[i] = sum_k T[k][i] v[k], k ascending, one fma per term.  WHICH >= 0: a whitening table (read through L2: these
    // are the layouts of ndim > 32; the 4-lane shapes run GradJumpWide); WHICH < 0: the global table Tg.
    template <int WHICH>
    __device__ __forceinline__ void tab_vec(const double *Tg, const double (&v)[EPL], double (&out)[EPL]) const
    {
        // every table read is unconditional (a padding slot reads element 0 of the row and its sum is dropped at the end):
        // a read under `if (i < d)` is not speculated, and each term then waited for its own round trip behind a branch
        double acc[EPL];
        int col[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            acc[e] = 0.0;
            col[e] = gl + G * e < d ? gl + G * e : 0;
        }
        const double *T = WHICH < 0 ? Tg : a.gj_tab + (size_t)WHICH * d * d;
        if (WHICH >= 0 && a.gj_diag) {                                   // diagonal whitening table: d multiplications (oracle: tab_vec)
#pragma unroll
            for (int e = 0; e < EPL; ++e) out[e] = gl + G * e < d ? T[(size_t)col[e] * d + col[e]] * v[e] : 0.0;
            return;
        }
#pragma unroll
        for (int e2 = 0; e2 < EPL; ++e2) {
#pragma unroll 1
            for (int src = 0; src < G; ++src) {
                const int k = src + G * e2;
                if (k >= d) break;
                const double vk = group_bcast_lane<G>(v[e2], src);
                const double *row = T + (size_t)k * d;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = __builtin_fma(row[col[e]], vk, acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) out[e] = gl + G * e < d ? acc[e] : 0.0;
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
            tab_vec<-1>(Pt, r, v);                                        // the gradient -P r: the full product
#pragma unroll
            for (int e = 0; e < EPL; ++e) g[e] = -v[e];
            double vh[EPL];
            tab_vec<-1>(Pt + (size_t)d * d, r, vh);                        // the value: eval_logl's half table Tl
            return -dot(r, vh);
        } else if (LOGL == PTMI_LOGL_INTERVAL) {
            const double *par = a.logl_par;
            double p = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e, ii = i < d ? i : 0;
                double gv;
                const double t = interval_elem<true>(x[e], par[ii], par[d + ii], par[2 * d + ii], gv);
                g[e] = i < d ? gv : 0.0;
                p = __builtin_fma(i < d ? t : 0.0, 1.0, p);
            }
            return group_sum<G>(p);
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
        return eval_logp_q<G, EPL, false>(a, gj_lds, gl, [&](int e) { return x[e]; });
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

    // ---- the stack of pending left subtrees.  Heights decrease strictly from the bottom of the stack to its top (a finished
    // subtree merges with a pending sibling of its own height before anything is pushed), so there is at most one entry
    // per height: the entry of height h lives in slot h, and the set of pending heights is a bit mask in a register (the
    // top of the stack is its lowest bit).  Slots below gj_lds_levels are in LDS -- height h is touched once per 2^h
    // leaves, so two levels take three quarters of the traffic -- the rest in the global scratch.
    __device__ __forceinline__ void stack_vec_store(int h, int which, const double (&v)[EPL]) const
    {
        if (h < a.gj_lds_levels) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) gj_lds[a.gj_stack_off + (h * (GJL_VECS * EPL + 4) + which * EPL + e) * 64 + (int)threadIdx.x] = v[e];
        } else {
            vstore(GJV_TOP + h * GJL_VECS + which, v);
        }
    }
    __device__ __forceinline__ void stack_vec_load(int h, int which, double (&v)[EPL]) const
    {
        if (h < a.gj_lds_levels) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = gj_lds[a.gj_stack_off + (h * (GJL_VECS * EPL + 4) + which * EPL + e) * 64 + (int)threadIdx.x];
        } else {
            vload(GJV_TOP + h * GJL_VECS + which, v);
        }
    }
    // scalars of an entry: in LDS every lane keeps its own copy; in global memory the chain's lanes share one (hence the fence)
    __device__ __forceinline__ void stack_scal_store(int h, double logp, double n, double alpha, double nalpha) const
    {
        if (h < a.gj_lds_levels) {
            const int base = a.gj_stack_off + (h * (GJL_VECS * EPL + 4) + GJL_VECS * EPL) * 64 + (int)threadIdx.x;
            gj_lds[base] = logp;
            gj_lds[base + 64] = n;
            gj_lds[base + 128] = alpha;
            gj_lds[base + 192] = nalpha;
        } else {
            scal(h, GJS_LOGP) = logp;                                    // every lane of the chain writes the same values
            scal(h, GJS_N) = n;
            scal(h, GJS_ALPHA) = alpha;
            scal(h, GJS_NALPHA) = nalpha;
            __threadfence_block();
        }
    }
    __device__ __forceinline__ void stack_scal_load(int h, double &logp, double &n, double &alpha, double &nalpha) const
    {
        if (h < a.gj_lds_levels) {
            const int base = a.gj_stack_off + (h * (GJL_VECS * EPL + 4) + GJL_VECS * EPL) * 64 + (int)threadIdx.x;
            logp = gj_lds[base];
            n = gj_lds[base + 64];
            alpha = gj_lds[base + 128];
            nalpha = gj_lds[base + 192];
        } else {
            logp = scal(h, GJS_LOGP);
            n = scal(h, GJS_N);
            alpha = scal(h, GJS_ALPHA);
            nalpha = scal(h, GJS_NALPHA);
        }
    }

    // NJ:495-652 as a loop: leaves are generated left to right in direction v from the growth end; a finished subtree
    // is merged with the pending left sibling of the same height on the stack, or waits there for its right sibling.
    // A left subtree that stopped (s = 0) is handed up unchanged to the height of the next pending sibling (or the root).
    __device__ __forceinline__ void build_tree(double (&tg)[EPL], double (&rg)[EPL], double (&gg)[EPL], double logu, int v, int j,
                                               double eps, double joint0, Tree &cur)
    {
        u32 pend = 0;                                                    // heights with a pending left subtree
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
                const int top_h = pend ? (int)__builtin_ctz(pend) : -1;
                if (top_h == h) {                                        // cur is the right sibling of the stack top
                    pend &= pend - 1u;
                    double t_logp, t_n, t_alpha, t_nalpha;
                    stack_scal_load(h, t_logp, t_n, t_alpha, t_nalpha);
                    const long long tn = (long long)t_n;
                    const long long tot = tn + cur.n;
                    const double den = (double)tot > 1.0 ? (double)tot : 1.0;
                    const bool take_u = uniform() < (double)cur.n / den;
                    if (!take_u) {
                        stack_vec_load(h, GJL_CAND_T, cur.cand_t);
                        stack_vec_load(h, GJL_CAND_G, cur.cand_g);
                        cur.logp = t_logp;
                    }
                    stack_vec_load(h, GJL_FAR_T, cur.far_t);
                    stack_vec_load(h, GJL_FAR_R, cur.far_r);
                    cur.n = tot;
                    const bool go = v == 1 ? keep_going(cur.far_t, tg, cur.far_r, rg) : keep_going(tg, cur.far_t, rg, cur.far_r);
                    cur.s = cur.s && go;                                 // the popped tree has s = 1
                    cur.alpha = t_alpha + cur.alpha;
                    cur.nalpha = (long long)t_nalpha + cur.nalpha;
                    h += 1;
                    continue;
                }
                if (h == j) return;
                if (cur.s == 0) {
                    if (pend == 0) return;
                    h = top_h;
                    continue;
                }
                stack_vec_store(h, GJL_FAR_T, cur.far_t);                // push: wait for the right sibling
                stack_vec_store(h, GJL_FAR_R, cur.far_r);
                stack_vec_store(h, GJL_CAND_T, cur.cand_t);
                stack_vec_store(h, GJL_CAND_G, cur.cand_g);
                stack_scal_store(h, cur.logp, (double)cur.n, cur.alpha, (double)cur.nalpha);
                pend |= 1u << h;
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

// ---------------------------------------------------------------------------------------------------------------------
// The 4-lane shapes (ndim <= 32) run a gradient jump ONE CHAIN AT A TIME ON THE WHOLE WAVE.  A launch ends with its
// slowest chain (a per cent of the ranks of the curved likelihood build trees of ~100 leapfrogs where the rest take one),
// and in the 4-lane layout a leapfrog costs ~7 000 issue slots: every lane walks all EPL slots of every vector, the tree's
// control flow is divergent between the chains of the wave (exec-mask bookkeeping, scalar spills), and the state of the
// recursion does not fit the registers.  Here element i = g + 4 e of a chain vector sits in lane 16 g + e (row g of the
// wave holds what lane g of the chain held, in slot order), a vector is ONE register pair, every scalar and all control
// flow is wave-uniform, the whole tree stack lives in LDS, and a leapfrog is ~700 issue slots.  The arithmetic and its
// order are those of the 4-lane code above (and of the oracle): a sum over a vector is the lane's chain of fmas over its
// slots -- here a scan along the row -- followed by (p0 + p2) + (p1 + p3).
constexpr int gjw_table_doubles(int EPL) { return 3 * (4 * EPL) * (4 * EPL); }          // rows padded to 4 EPL: constant offsets
constexpr int gjw_level_doubles(int EPL) { return GJL_VECS * 4 * EPL + 4; }         // a stack entry: four vectors in element order, four scalars
__device__ __forceinline__ double lane_get(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)b, lane), hi = (u32)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

// -DPTMI_GJ_PROFILE: cycle counts of the pieces of a gradient jump, printed by the first wave (tools/gj_leap_timing.py)
#ifdef PTMI_GJ_PROFILE
#define GJP_T0(var) const unsigned long long var = __builtin_readcyclecounter()
#define GJP_ADD(slot, var) (prof[slot] += __builtin_readcyclecounter() - var)
#else
#define GJP_T0(var)
#define GJP_ADD(slot, var)
#endif
enum { GJP_TABVEC = 0, GJP_LOGL = 1, GJP_DOT = 2, GJP_LEAF = 3, GJP_MERGE = 4, GJP_PUSH = 5, GJP_CALL = 6, GJP_DRAW = 7, GJP_N = 8 };

// the value of lane L ^ 16 (the neighbouring row of 16 lanes): v_permlane16_swap trades the odd rows of its first operand
// for the even rows of its second, so swapping a register with a copy of itself leaves {row0, row0, row2, row2} in the
// first result and {row1, row1, row3, row3} in the second (checked on the hardware); no LDS round trip as with a shuffle
__device__ __forceinline__ double lane_xor16(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((u32)b, (u32)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((u32)(b >> 32), (u32)(b >> 32), false, false);
    const bool odd = (threadIdx.x >> 4) & 1;
    const u32 l = odd ? lo[0] : lo[1], h = odd ? hi[0] : hi[1];
    return __longlong_as_double((long long)(((u64)h << 32) | l));
}

// GC = 16 (round 5; kernel shape (16, 7), ndim <= 64, diagonal whitening): the chain's sixteen lane groups are the wave's sixteen quads,
// element i = g + 16 e in lane 4 g + e -- the 16-lane order of a dot product is then a chain along the quad (quad_perm shifts) and the
// butterfly g ^ 8, 4, 2, 1 = lane ^ 32, 16, 8, 4 in the vector pipe (v_permlane32/16_swap, row rotations); the Box-Muller partners
// (k, k + 16) are quad neighbours.  The template's EPL is then LD / 4 = 16.
template <int EPL, int LOGL, int GC = 4, bool FULLTAB = false /* GC = 16: full whitening tables (else diagonal ones: decided at compile time there) */>
struct GradJumpWide {
    static constexpr int G = GC, NS = GC == 4 ? EPL : 4 /* slots of a lane group */, LD = GC * NS, LEV = gjw_level_doubles(LD / 4);
    static_assert(GC == 4 || (GC == 16 && EPL == 16), "whole-wave layouts: 4 lane groups in rows of 16 lanes, or 16 lane groups in quads");
    const KArgs &a;
    const int d, L, we, wg, wi;
    const bool act;                  // this lane holds an element of the chain's vectors
    const int col;                   // its index (0 on idle lanes: their reads are dropped)
    const long long ch, nch;
    const double beta;
    const long long it;
    const u32 sid;
    const int vb;                    // 64 doubles of the block's LDS: the vector of a table product, in element order
    double blo = 0.0, bhi = 0.0;     // this lane's bounds of a box prior
    double iv_lo = 0.0, iv_w = 0.0, iv_lw = 0.0;     // this lane's parameters of the interval family
    u32 nm = 0, ns = 0, nleap = 0;
#ifdef PTMI_GJ_PROFILE
    mutable unsigned long long prof[GJP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    __device__ __forceinline__ GradJumpWide(const KArgs &a_, long long ch_, double beta_, long long it_, u32 sid_, int vb_)
        : a(a_), d(a_.d), L((int)threadIdx.x), we(GC == 4 ? (L & 15) : (L & 3)), wg(GC == 4 ? (L >> 4) : (L >> 2)), wi(wg + GC * we), act(we < NS && wi < a_.d), col(act ? wi : 0),
          ch(ch_), nch((long long)a_.W * a_.nt), beta(beta_), it(it_), sid(sid_), vb(vb_)
    {
        if (a.logp_kind == PTMI_LOGP_BOX) { blo = a.logp_par[col]; bhi = a.logp_par[d + col]; }
        if (LOGL == PTMI_LOGL_INTERVAL) { iv_lo = a.logl_par[col]; iv_w = a.logl_par[d + col]; iv_lw = a.logl_par[2 * d + col]; }
    }

    // ---- draws
    __device__ __forceinline__ double momenta()                            // NJ:92-94; directions k and k + 4 share one Box-Muller
    {
        const u32 block = nm++;
        double r = 0.0;
        if (act) {
            const int k = wg + GC * (we & ~1);
            u64 e0, e1;
            philox_words(a.seed, (u64)it, sid, SLOT_GJ + 4096u * block + (u32)k, e0, e1);
            const double rr = det_sqrt(-2.0 * det_log(w2uniform_open(e0)));
            double sn, cs;
            det_sincos2pi(w2uniform(e1), sn, cs);
            r = (we & 1) ? rr * sn : rr * cs;
        }
        return r;
    }
    // Scalar draws: slot n of the call is a Philox call of its own.  Lane j evaluates slot 64 b + j, so one pass of the
    // generator (the same instructions a single draw would cost the wave) serves the next 64 draws of the call.
    u64 sw_cache = 0;
    u32 sw_base = 0xFFFFFFFFu;
    __device__ __forceinline__ u64 scalar_word()
    {
        GJP_T0(t0);
        const u32 slot = ns++;
        if ((slot & ~63u) != sw_base) {
            sw_base = slot & ~63u;
            u64 w0, w1;
            philox_words(a.seed, (u64)it, sid, SLOT_GJS + sw_base + (u32)L, w0, w1);
            sw_cache = w0;
        }
        const int src = (int)(slot & 63u);
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)sw_cache, src), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(sw_cache >> 32), src);
        GJP_ADD(GJP_DRAW, t0);
        return ((u64)hi << 32) | lo;
    }
    __device__ __forceinline__ double uniform() { return w2uniform(scalar_word()); }
    __device__ __forceinline__ double exponential() { return -det_log(w2uniform_open(scalar_word())); }
    __device__ __forceinline__ int randint(int lo, int hi) { return lo + (int)w2index(scalar_word(), (u64)(hi - lo)); }

    // ---- linear algebra
    // Sum over the chain in the 4-lane order: per row the chain p = fma(x_e, y_e, p) over its slots.  Every lane recomputes
    // its link from its left neighbour's value in each of the EPL steps: lane e is right from step e + 1 on (lane 0 has no
    // neighbour and starts from 0), so after EPL steps lane EPL - 1 of row g holds the chain of lane group g; then (p0 + p2) + (p1 + p3).
    __device__ __forceinline__ double row_chain(double x, double y) const
    {
        double p = 0.0;
        if constexpr (GC == 4) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) p = __builtin_fma(x, y, dppf64<0x111>(p));      // row_shr:1, 0 into lane 0
        } else {                                                         // along the quad: quad_perm [0, 0, 1, 2], the first lane starts from 0
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                const double left = dppf64<0x90>(p);
                p = __builtin_fma(x, y, we == 0 ? 0.0 : left);
            }
        }
        return p;
    }
    __device__ __forceinline__ double rows_sum(double p) const
    {
        if constexpr (GC == 16) {                                       // group_sum<16>'s butterfly over the lane groups, the chains' ends in lane 3 of the quads
            double s = sum_xor32(p);                                     // g ^ 8
            s = sum_xor16(s);                                            // g ^ 4
            s = s + dppf64<0x128>(s);                                    // g ^ 2: row_ror:8
            s = s + dppf64<0x124>(s);                                    // g ^ 1: row_ror:4 on data of period 8
            return quad_bcastf<3>(s);
        }
        const double p0 = lane_get(p, EPL - 1), p1 = lane_get(p, 16 + EPL - 1), p2 = lane_get(p, 32 + EPL - 1), p3 = lane_get(p, 48 + EPL - 1);
        return (p0 + p2) + (p1 + p3);
    }
    __device__ __forceinline__ double dot(double x, double y) const
    {
        GJP_T0(t0);
        const double r = rows_sum(row_chain(x, y));
        GJP_ADD(GJP_DOT, t0);
        return r;
    }
    // out[i] = sum_k T[k][i] v[k], k ascending, one fma per term; WHICH >= 0: the LDS copy of a whitening table (rows of LD)
    template <int WHICH>
    __device__ __forceinline__ double tab_vec(const double *Tg, double v) const
    {
        GJP_T0(t0);
        double acc = 0.0;
        if (WHICH >= 0 && (GC == 16 ? !FULLTAB : (bool)a.gj_diag)) {     // diagonal whitening table: one multiplication per element (oracle: tab_vec)
            const double r = act ? gj_lds[WHICH * LD + col] * v : 0.0;  // (the block's LDS then holds the three diagonals only)
            GJP_ADD(GJP_TABVEC, t0);
            return r;
        }
        if constexpr (GC == 16) {
            // 16 lane groups (ndim <= 64): the vector through LDS in element order as below, the table's rows from global memory (3 d^2
            // doubles, cache resident: the LDS copies of 64 x 64 tables would leave one wave per CU), sixteen terms at a time
            __syncthreads();
            if (act) gj_lds[vb + wi] = v;
            __syncthreads();
            const double *T = WHICH >= 0 ? a.gj_tab + (size_t)WHICH * d * d : Tg;
#pragma unroll 1
            for (int k0 = 0; k0 < d; k0 += 16) {
                double tk[16], vk[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = k0 + j;
                    tk[j] = T[(size_t)(k < d ? k : 0) * d + col];
                    vk[j] = gj_lds[vb + (k < 64 ? k : 0)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double nxt = __builtin_fma(tk[j], vk[j], acc);
                    acc = k0 + j < d ? nxt : acc;
                }
            }
            GJP_ADD(GJP_TABVEC, t0);
            return act ? acc : 0.0;
        }
        else {
        // The vector goes through LDS in element order and every lane reads all of it back (same address for the whole wave:
        // a broadcast); two readlanes per term instead had each fma wait on a fresh scalar pair.  Straight-line on purpose:
        // with a (wave-uniform) branch around every term each table read waited for its own LDS round trip -- 2 400 cycles
        // per product, more than half of a gradient jump.  Rows k >= d of the LDS tables are zeros; their terms are computed and dropped.
        __syncthreads();                                                 // the previous product's readers are through
        if (act) gj_lds[vb + wi] = v;
        __syncthreads();
        // all reads first (the scheduling barrier keeps them there): left to the scheduler they went out two at a time
        // and the chain of fmas waited for an LDS round trip at every other term
        double tk[4 * EPL], vk[4 * EPL];
#pragma unroll
        for (int k = 0; k < 4 * EPL; ++k) {
            tk[k] = WHICH >= 0 ? gj_lds[(WHICH * LD + k) * LD + col] : Tg[(size_t)(k < d ? k : 0) * d + col];
            vk[k] = gj_lds[vb + k];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (d == 4 * EPL) {
#pragma unroll
            for (int k = 0; k < 4 * EPL; ++k) acc = __builtin_fma(tk[k], vk[k], acc);
        } else {                                                         // (a uniform branch per term instead of the select is 4x slower)
#pragma unroll
            for (int k = 0; k < 4 * EPL; ++k) {
                const double nxt = __builtin_fma(tk[k], vk[k], acc);
                acc = k < d ? nxt : acc;
            }
        }
        GJP_ADD(GJP_TABVEC, t0);
        return act ? acc : 0.0;
        }
    }
    __device__ __forceinline__ double logl_grad(double x, double &g) const
    {
        GJP_T0(t0);
        if (LOGL == PTMI_LOGL_ISO) {
            g = -x;
            const double r = -0.5 * dot(x, x);
            GJP_ADD(GJP_LOGL, t0);
            return r;
        } else if (LOGL == PTMI_LOGL_DENSE) {
            const double r = act ? x - a.logl_par[col] : 0.0;
            const double v = tab_vec<-1>(a.logl_par + d, r);                 // the gradient -P r: the full product
            g = -v;
            const double vh = tab_vec<-1>(a.logl_par + d + (size_t)d * d, r);  // the value: eval_logl's half table Tl
            const double rr = -dot(r, vh);
            GJP_ADD(GJP_LOGL, t0);
            return rr;
        } else if (LOGL == PTMI_LOGL_INTERVAL) {
            double gv;
            const double t = interval_elem<true>(x, iv_lo, iv_w, iv_lw, gv);
            g = act ? gv : 0.0;
            const double r = dot(act ? t : 0.0, 1.0);
            GJP_ADD(GJP_LOGL, t0);
            return r;
        } else {
            auto partner = [&](double v) {                                 // the pair's other member: lane group g ^ 1, same slot
                if constexpr (GC == 4) return lane_xor16(v);
                else return __shfl_xor(v, 4, 64);
            };
            const double other = partner(x);
            const bool even = !(wg & 1);
            const bool pair = we < NS && (even ? wi + 1 < d : wi < d);
            const double xx = even ? x : other, y = even ? other : x;
            const double x2 = xx * xx;
            const double gg = 9.0 + 4.0 * x2 + 9.0 * y;
            const double l0 = -x2 - gg * gg;
            const double ym = y - 2.0;
            const double l1 = -8.0 * x2 - 8.0 * (ym * ym);
            // both lanes of a pair hold the same l0 and l1: the even one takes exp(l0), the odd one exp(l1), and they trade
            const double ex = det_exp(even ? l0 : l1), ox = partner(ex);
            const double e0 = even ? ex : ox, e1 = 0.5 * (even ? ox : ex);
            const double sum = e0 + e1;
            const double tl = det_log(sum);                                // wanted on the even lanes only; same cost for the wave
            const double d0 = even ? -2.0 * xx - 16.0 * gg * xx : -18.0 * gg;
            const double d1 = even ? -16.0 * xx : -16.0 * ym;
            const double gv = (e0 * d0 + e1 * d1) / sum;
            g = pair ? gv : 0.0;
            const double r = dot(pair && even ? tl : 0.0, 1.0);
            GJP_ADD(GJP_LOGL, t0);
            return r;
        }
    }
    __device__ __forceinline__ double logp(double x) const
    {
        if (a.logp_kind == PTMI_LOGP_BOX) {
            const bool ok = !act || ((blo <= x) & (bhi >= x));
            return __ballot(ok) == ~0ull ? 0.0 : -__builtin_inf();
        }
        return 0.0;
    }
    __device__ __forceinline__ double func_grad_white(double q, double &gradw) const       // NJ:71-90
    {
        const double x = tab_vec<GJT_BACKWARD>(nullptr, q);
        double g;
        const double ll = logl_grad(x, g);
        const double lp = logp(x);
        g = beta * g + 0.0;
        gradw = tab_vec<GJT_GRADIENT>(nullptr, g);
        return beta * ll + lp;
    }
    __device__ __forceinline__ double joint_of(double logl, double r) const { return logl - 0.5 * dot(r, r); }
    __device__ __forceinline__ double leapfrog(double theta, double r, double grad, double eps, double &to, double &ro, double &go)   // NJ:149-169
    {
        nleap += 1;
        const double he = 0.5 * eps;
        const double rh = r + he * grad;
        const double tn = theta + eps * rh;
        double gn;
        const double lpp = func_grad_white(tn, gn);
        to = tn;
        go = gn;
        ro = rh + he * gn;
        return lpp;
    }
    __device__ __forceinline__ bool keep_going(double tm, double tp, double rm, double rp) const                // NJ:465-493
    {
        const double dt = tp - tm;
        const double cx = row_chain(dt, rm), cy = row_chain(dt, rp);      // two independent chains: they overlap
        const double x = rows_sum(cx), y = rows_sum(cy);
        return (x >= 0.0) & (y >= 0.0);
    }
    __device__ __forceinline__ bool any_inf(double v) const { return __ballot(act && __builtin_isinf(v)) != 0ull; }

    // ------------------------------------------------------------------ HMC (NJ:238-291)
    __device__ __forceinline__ double hmc(double *st, double x, double &qout)
    {
        st[GJ_HITER] += 1.0;
        double q = tab_vec<GJT_FORWARD>(nullptr, x), grad;
        const double logp0 = func_grad_white(q, grad);
        double p = momenta();
        const double joint0 = joint_of(logp0, p);
        const int nsteps = randint(a.hmc_min, a.hmc_max);
        double joint1 = joint0;
        for (int k = 0; k < nsteps; ++k) {
            const double logp1 = leapfrog(q, p, grad, a.hmc_eps, q, p, grad);
            joint1 = joint_of(logp1, p);
            if (joint1 - 1000.0 < joint0) break;                         // NJ:284-286
        }
        qout = tab_vec<GJT_BACKWARD>(nullptr, q);
        return joint1 - joint0;
    }

    // ------------------------------------------------------------------ NUTS
    __device__ __forceinline__ double find_reasonable_epsilon(double theta0, double grad0, double logp0)   // NJ:435-463, loops bounded
    {
        double tp, rp, gp;
        double eps = 1.0;
        const double r0 = momenta();
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

    struct Tree {
        double far_t, far_r, cand_t, cand_g;
        double logp, alpha;
        long long n, nalpha;
        int s;
    };
    // the tree stack (see GradJump::build_tree): the entry of height h in slot h, pending heights in a mask.  Heights below
    // gj_lds_levels (11: trees of up to 2^11 leapfrogs) are in the block's LDS; the higher ones -- the reference doubles without a
    // cap (NJ:716-802), the ABI allows 24 -- in the wave's slice of the global scratch: reached once in 2^h leapfrogs, if ever,
    // so only their correctness matters (keeping all 25 in LDS cost the config-5 kernel 25 %: five waves per CU instead of eight)
    __device__ __forceinline__ int slot_of(int h) const { return a.gj_stack_off + h * LEV; }
    __device__ __forceinline__ double *glevel(int h) const
    {
        return a.gj_scr + ((size_t)blockIdx.x * (size_t)(a.nuts_maxdepth + 1) + (size_t)h) * LEV;
    }
    // NJ:495-652 as a loop, as GradJump::build_tree
    __device__ __forceinline__ void build_tree(double &tg, double &rg, double &gg, double logu, int v, int j, double eps, double joint0, Tree &cur)
    {
        u32 pend = 0;
        for (;;) {
            const double logpp = leapfrog(tg, rg, gg, (double)v * eps, tg, rg, gg);
            GJP_T0(tl0);
            const double joint = joint_of(logpp, rg);
            cur.n = logu < joint;
            cur.s = (logu - 1000.0) < joint;
            cur.far_t = tg; cur.far_r = rg; cur.cand_t = tg; cur.cand_g = gg;
            cur.logp = logpp;
            const double ex = det_exp(joint - joint0);
            cur.alpha = ex < 1.0 ? ex : 1.0;                             // Python's min(1.0, e): 1.0 when e is NaN
            cur.nalpha = 1;
            GJP_ADD(GJP_LEAF, tl0);
            int h = 0;
            for (;;) {
                const int top_h = pend ? (int)__builtin_ctz(pend) : -1;
                if (top_h == h) {                                        // cur is the right sibling of the stack top
                    GJP_T0(tm0);
                    pend &= pend - 1u;
                    double t_logp, t_n, t_alpha, t_nalpha, e_ct, e_cg, e_ft, e_fr;
                    if (h < a.gj_lds_levels) {                                     // wave-uniform
                        const int b = slot_of(h);
                        t_logp = gj_lds[b + GJL_VECS * LD + GJS_LOGP]; t_n = gj_lds[b + GJL_VECS * LD + GJS_N];
                        t_alpha = gj_lds[b + GJL_VECS * LD + GJS_ALPHA]; t_nalpha = gj_lds[b + GJL_VECS * LD + GJS_NALPHA];
                        e_ct = act ? gj_lds[b + GJL_CAND_T * LD + col] : 0.0; e_cg = act ? gj_lds[b + GJL_CAND_G * LD + col] : 0.0;
                        e_ft = act ? gj_lds[b + GJL_FAR_T * LD + col] : 0.0; e_fr = act ? gj_lds[b + GJL_FAR_R * LD + col] : 0.0;
                    } else {
                        const double *gp = glevel(h);
                        t_logp = gp[GJL_VECS * LD + GJS_LOGP]; t_n = gp[GJL_VECS * LD + GJS_N];
                        t_alpha = gp[GJL_VECS * LD + GJS_ALPHA]; t_nalpha = gp[GJL_VECS * LD + GJS_NALPHA];
                        e_ct = act ? gp[GJL_CAND_T * LD + col] : 0.0; e_cg = act ? gp[GJL_CAND_G * LD + col] : 0.0;
                        e_ft = act ? gp[GJL_FAR_T * LD + col] : 0.0; e_fr = act ? gp[GJL_FAR_R * LD + col] : 0.0;
                    }
                    const long long tot = (long long)t_n + cur.n;
                    const double den = (double)tot > 1.0 ? (double)tot : 1.0;
                    const bool take_u = uniform() < (double)cur.n / den;
                    if (!take_u) {
                        cur.cand_t = e_ct;
                        cur.cand_g = e_cg;
                        cur.logp = t_logp;
                    }
                    cur.far_t = e_ft;
                    cur.far_r = e_fr;
                    cur.n = tot;
                    const bool go = v == 1 ? keep_going(cur.far_t, tg, cur.far_r, rg) : keep_going(tg, cur.far_t, rg, cur.far_r);
                    cur.s = cur.s && go;                                 // the popped tree has s = 1
                    cur.alpha = t_alpha + cur.alpha;
                    cur.nalpha = (long long)t_nalpha + cur.nalpha;
                    h += 1;
                    GJP_ADD(GJP_MERGE, tm0);
                    continue;
                }
                if (h == j) return;
                if (cur.s == 0) {
                    if (pend == 0) return;
                    h = top_h;
                    continue;
                }
                GJP_T0(tp0);
                if (h < a.gj_lds_levels) {                               // push: wait for the right sibling
                    const int b = slot_of(h);
                    if (act) {                                           // in element order: 4 LD doubles per entry instead of 4 x 64
                        gj_lds[b + GJL_FAR_T * LD + wi] = cur.far_t;
                        gj_lds[b + GJL_FAR_R * LD + wi] = cur.far_r;
                        gj_lds[b + GJL_CAND_T * LD + wi] = cur.cand_t;
                        gj_lds[b + GJL_CAND_G * LD + wi] = cur.cand_g;
                    }
                    if (L == 0) {
                        gj_lds[b + GJL_VECS * LD + GJS_LOGP] = cur.logp;
                        gj_lds[b + GJL_VECS * LD + GJS_N] = (double)cur.n;
                        gj_lds[b + GJL_VECS * LD + GJS_ALPHA] = cur.alpha;
                        gj_lds[b + GJL_VECS * LD + GJS_NALPHA] = (double)cur.nalpha;
                    }
                } else {
                    double *gp = glevel(h);
                    if (act) {
                        gp[GJL_FAR_T * LD + wi] = cur.far_t;
                        gp[GJL_FAR_R * LD + wi] = cur.far_r;
                        gp[GJL_CAND_T * LD + wi] = cur.cand_t;
                        gp[GJL_CAND_G * LD + wi] = cur.cand_g;
                    }
                    if (L == 0) {
                        gp[GJL_VECS * LD + GJS_LOGP] = cur.logp;
                        gp[GJL_VECS * LD + GJS_N] = (double)cur.n;
                        gp[GJL_VECS * LD + GJS_ALPHA] = cur.alpha;
                        gp[GJL_VECS * LD + GJS_NALPHA] = (double)cur.nalpha;
                    }
                    __threadfence_block();                               // the wave's own global stores, visible to its other lanes
                }
                __syncthreads();                                         // one wave per block: orders lane 0's writes before the others' reads
                pend |= 1u << h;
                GJP_ADD(GJP_PUSH, tp0);
                break;
            }
        }
    }

    // NUTSJump.__call__ (NJ:654-840), as GradJump::nuts; the two ends of the trajectory and the sample stay in registers
    __device__ __forceinline__ double nuts(double *st, double x, double &qout)
    {
        st[GJ_NITER] += 1.0;
        const double q = tab_vec<GJT_FORWARD>(nullptr, x);
        double grad;
        const double logp0 = func_grad_white(q, grad);
        if (st[GJ_HAVE_EPS] == 0.0) {
            st[GJ_EPS] = find_reasonable_epsilon(q, grad, logp0);
            st[GJ_MU] = det_log(10.0 * st[GJ_EPS]);
            st[GJ_HAVE_EPS] = 1.0;
        }
        const double r0 = momenta();
        const double joint = joint_of(logp0, r0);
        const double logu = joint - exponential();
        double lnprob = logp0;
        double sample = q, tm = q, rm = r0, gm = grad, tp = q, rp = r0, gp = grad;
        int j = 0, s = 1;
        long long n = 1;
        double alpha = 0.0;
        long long nalpha = 1;
        const double eps = st[GJ_EPS];
        while (s == 1) {
            const int dir = 2 * (int)(uniform() < 0.5) - 1;
            double tg = dir == -1 ? tm : tp, rg = dir == -1 ? rm : rp, gg = dir == -1 ? gm : gp;
            Tree t;
            build_tree(tg, rg, gg, logu, dir, j, eps, joint, t);
            if (dir == -1) { tm = tg; rm = rg; gm = gg; } else { tp = tg; rp = rg; gp = gg; }
            if (t.s == 1) {
                const double ratio = (double)t.n / (double)n;
                if (uniform() < (1.0 < ratio ? 1.0 : ratio)) { sample = t.cand_t; lnprob = t.logp; }
            }
            n += t.n;
            const bool go = keep_going(tm, tp, rm, rp);
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
        qout = tab_vec<GJT_BACKWARD>(nullptr, sample);
        return logp0 - lnprob;                                           // undoes the outer Hastings ratio (NJ:838)
    }
};


template <int EPL, int LOGL>
struct GradJumpPair {
    static constexpr int G = 4, LD = 4 * EPL;
    // TWO chains per wave: chain slot hh = lane >> 5 owns a half-wave.  Inside its half a chain's lane group g = 2 r + sub sits in
    // row r = (lane >> 4) & 1 at lanes 8 sub .. 8 sub + EPL - 1 (element i = g + 4 e in lane 16 r + 8 sub + e): row shifts and the
    // xor-8 partner stay inside a row, everything else goes through ds_bpermute inside the half.  Every "scalar" of a call is a
    // per-lane value (it was one already: f64 arithmetic is vector arithmetic), so the two calls run as one instruction stream and
    // part ways only where their control flow does (divergent branches; the halves never read each other's lanes).
    const KArgs &a;
    const int d, L, hh, we, wg, wi;
    const bool act;                  // this lane holds an element of the chain's vectors
    const int col;                   // its index (0 on idle lanes: their reads are dropped)
    const long long ch, nch;
    const double beta;
    const long long it;
    const u32 sid;
    const int vb;                    // 64 doubles of the block's LDS: the vector of a table product, in element order
    double blo = 0.0, bhi = 0.0;     // this lane's bounds of a box prior
    double iv_lo = 0.0, iv_w = 0.0, iv_lw = 0.0;     // this lane's parameters of the interval family
    u32 nm = 0, ns = 0, nleap = 0;
#ifdef PTMI_GJ_PROFILE
    mutable unsigned long long prof[GJP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    __device__ __forceinline__ GradJumpPair(const KArgs &a_, long long ch_, double beta_, long long it_, u32 sid_, int vb_)
        : a(a_), d(a_.d), L((int)threadIdx.x), hh(L >> 5), we(L & 7), wg(2 * ((L >> 4) & 1) + ((L >> 3) & 1)), wi(wg + 4 * we), act(we < EPL && wi < a_.d), col(act ? wi : 0),
          ch(ch_), nch((long long)a_.W * a_.nt), beta(beta_), it(it_), sid(sid_), vb(vb_)
    {
        if (a.logp_kind == PTMI_LOGP_BOX) { blo = a.logp_par[col]; bhi = a.logp_par[d + col]; }
        if (LOGL == PTMI_LOGL_INTERVAL) { iv_lo = a.logl_par[col]; iv_w = a.logl_par[d + col]; iv_lw = a.logl_par[2 * d + col]; }
    }

    __device__ __forceinline__ u32 half_lane32(u32 v, int src) const { return (u32)__builtin_amdgcn_ds_bpermute(((hh << 5) + src) << 2, (int)v); }
    __device__ __forceinline__ double half_lanef(double v, int src) const
    {
        const u64 b = (u64)__double_as_longlong(v);
        return __longlong_as_double((long long)(((u64)half_lane32((u32)(b >> 32), src) << 32) | half_lane32((u32)b, src)));
    }
    __device__ __forceinline__ bool half_all(bool ok) const { return ((__ballot(ok) >> (hh << 5)) & 0xFFFFFFFFull) == 0xFFFFFFFFull; }
    __device__ __forceinline__ bool half_any(bool v) const { return ((__ballot(v) >> (hh << 5)) & 0xFFFFFFFFull) != 0ull; }
    // ---- draws
    __device__ __forceinline__ double momenta()                            // NJ:92-94; directions k and k + 4 share one Box-Muller
    {
        const u32 block = nm++;
        double r = 0.0;
        if (act) {
            const int k = wg + 4 * (we & ~1);
            u64 e0, e1;
            philox_words(a.seed, (u64)it, sid, SLOT_GJ + 4096u * block + (u32)k, e0, e1);
            const double rr = det_sqrt(-2.0 * det_log(w2uniform_open(e0)));
            double sn, cs;
            det_sincos2pi(w2uniform(e1), sn, cs);
            r = (we & 1) ? rr * sn : rr * cs;
        }
        return r;
    }
    // Scalar draws: slot n of the call is a Philox call of its own.  Lane j evaluates slot 64 b + j, so one pass of the
    // generator (the same instructions a single draw would cost the wave) serves the next 64 draws of the call.
    u64 sw_cache = 0;
    u32 sw_base = 0xFFFFFFFFu;
    __device__ __forceinline__ u64 scalar_word()
    {
        GJP_T0(t0);
        const u32 slot = ns++;
        if ((slot & ~31u) != sw_base) {                                  // (per half: lane j of the half evaluates slot 32 b + j of ITS chain's call)
            sw_base = slot & ~31u;
            u64 w0, w1;
            philox_words(a.seed, (u64)it, sid, SLOT_GJS + sw_base + (u32)(L & 31), w0, w1);
            sw_cache = w0;
        }
        const int src = (int)(slot & 31u);
        const u32 lo = half_lane32((u32)sw_cache, src), hi = half_lane32((u32)(sw_cache >> 32), src);
        GJP_ADD(GJP_DRAW, t0);
        return ((u64)hi << 32) | lo;
    }
    __device__ __forceinline__ double uniform() { return w2uniform(scalar_word()); }
    __device__ __forceinline__ double exponential() { return -det_log(w2uniform_open(scalar_word())); }
    __device__ __forceinline__ int randint(int lo, int hi) { return lo + (int)w2index(scalar_word(), (u64)(hi - lo)); }

    // ---- linear algebra
    // Sum over the chain in the 4-lane order: per row the chain p = fma(x_e, y_e, p) over its slots.  Every lane recomputes
    // its link from its left neighbour's value in each of the EPL steps: lane e is right from step e + 1 on (lane 0 has no
    // neighbour and starts from 0), so after EPL steps lane EPL - 1 of row g holds the chain of lane group g; then (p0 + p2) + (p1 + p3).
    __device__ __forceinline__ double row_chain(double x, double y) const
    {
        double p = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const double left = dppf64<0x111>(p);                       // row_shr:1
            p = __builtin_fma(x, y, we == 0 ? 0.0 : left);              // a lane group's chain starts from 0 at its first lane
        }
        return p;
    }
    // (p0 + p2) + (p1 + p3) of the four lane groups' chains (lane EPL - 1 of the half's four rows of eight), on every lane of the
    // half, in the vector pipe: lanes 16 apart through v_permlane16_swap, 8 apart through a row rotation, then the row's lane
    // EPL - 1 to the whole row (row_newbcast).  Four ds_bpermute pairs took an LDS round trip on the critical path of every
    // dot product -- four to a leapfrog of a deep tree, whose chain of dependent instructions is what a launch lasts (§3.5).
    __device__ __forceinline__ double rows_sum(double p) const
    {
        double s = sum_xor16(p);                                        // rows 0 + 1 (2 + 3) of the half: (p0 + p2) at lane EPL - 1, (p1 + p3) at lane 8 + EPL - 1
        s = s + dppf64<0x128>(s);                                       // row_ror:8: (p0 + p2) + (p1 + p3), or the same two terms the other way round
        return dppf64<0x150 + EPL - 1>(s);
    }
    __device__ __forceinline__ double dot(double x, double y) const
    {
        GJP_T0(t0);
        const double r = rows_sum(row_chain(x, y));
        GJP_ADD(GJP_DOT, t0);
        return r;
    }
    // out[i] = sum_k T[k][i] v[k], k ascending, one fma per term; WHICH >= 0: the LDS copy of a whitening table (rows of LD)
    template <int WHICH>
    __device__ __forceinline__ double tab_vec(const double *Tg, double v) const
    {
        GJP_T0(t0);
        double acc = 0.0;
        if (WHICH >= 0 && a.gj_diag) {                                   // diagonal whitening table: one multiplication per element (oracle: tab_vec)
            const double r = act ? gj_lds[WHICH * LD + col] * v : 0.0;  // (the block's LDS then holds the three diagonals only)
            GJP_ADD(GJP_TABVEC, t0);
            return r;
        }
        // (the pair layout runs with diagonal whitening tables and the iso / curved families only: the host picks it then)
        __builtin_trap();
        // The vector goes through LDS in element order and every lane reads all of it back (same address for the whole wave:
        // a broadcast); two readlanes per term instead had each fma wait on a fresh scalar pair.  Straight-line on purpose:
        // with a (wave-uniform) branch around every term each table read waited for its own LDS round trip -- 2 400 cycles
        // per product, more than half of a gradient jump.  Rows k >= d of the LDS tables are zeros; their terms are computed and dropped.
        __syncthreads();                                                 // the previous product's readers are through
        if (act) gj_lds[vb + wi] = v;
        __syncthreads();
        // all reads first (the scheduling barrier keeps them there): left to the scheduler they went out two at a time
        // and the chain of fmas waited for an LDS round trip at every other term
        double tk[4 * EPL], vk[4 * EPL];
#pragma unroll
        for (int k = 0; k < 4 * EPL; ++k) {
            tk[k] = WHICH >= 0 ? gj_lds[(WHICH * LD + k) * LD + col] : Tg[(size_t)(k < d ? k : 0) * d + col];
            vk[k] = gj_lds[vb + k];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (d == 4 * EPL) {
#pragma unroll
            for (int k = 0; k < 4 * EPL; ++k) acc = __builtin_fma(tk[k], vk[k], acc);
        } else {                                                         // (a uniform branch per term instead of the select is 4x slower)
#pragma unroll
            for (int k = 0; k < 4 * EPL; ++k) {
                const double nxt = __builtin_fma(tk[k], vk[k], acc);
                acc = k < d ? nxt : acc;
            }
        }
        GJP_ADD(GJP_TABVEC, t0);
        return act ? acc : 0.0;
    }
    __device__ __forceinline__ double logl_grad(double x, double &g) const
    {
        GJP_T0(t0);
        if (LOGL == PTMI_LOGL_ISO) {
            g = -x;
            const double r = -0.5 * dot(x, x);
            GJP_ADD(GJP_LOGL, t0);
            return r;
        } else if (LOGL == PTMI_LOGL_DENSE) {
            const double r = act ? x - a.logl_par[col] : 0.0;
            const double v = tab_vec<-1>(a.logl_par + d, r);                 // the gradient -P r: the full product
            g = -v;
            const double vh = tab_vec<-1>(a.logl_par + d + (size_t)d * d, r);  // the value: eval_logl's half table Tl
            const double rr = -dot(r, vh);
            GJP_ADD(GJP_LOGL, t0);
            return rr;
        } else if (LOGL == PTMI_LOGL_INTERVAL) {
            double gv;
            const double t = interval_elem<true>(x, iv_lo, iv_w, iv_lw, gv);
            g = act ? gv : 0.0;
            const double r = dot(act ? t : 0.0, 1.0);
            GJP_ADD(GJP_LOGL, t0);
            return r;
        } else {
            const double other = dppf64<0x128>(x);                         // the pair's other member: lane group g ^ 1 = lane ^ 8 (row_ror:8), same slot
            const bool even = !(wg & 1);
            const bool pair = we < EPL && (even ? wi + 1 < d : wi < d);
            const double xx = even ? x : other, y = even ? other : x;
            const double x2 = xx * xx;
            const double gg = 9.0 + 4.0 * x2 + 9.0 * y;
            const double l0 = -x2 - gg * gg;
            const double ym = y - 2.0;
            const double l1 = -8.0 * x2 - 8.0 * (ym * ym);
            // both lanes of a pair hold the same l0 and l1: the even one takes exp(l0), the odd one exp(l1), and they trade
            const double ex = det_exp(even ? l0 : l1), ox = dppf64<0x128>(ex);
            const double e0 = even ? ex : ox, e1 = 0.5 * (even ? ox : ex);
            const double sum = e0 + e1;
            const double tl = det_log(sum);                                // wanted on the even lanes only; same cost for the wave
            const double d0 = even ? -2.0 * xx - 16.0 * gg * xx : -18.0 * gg;
            const double d1 = even ? -16.0 * xx : -16.0 * ym;
            const double gv = (e0 * d0 + e1 * d1) / sum;
            g = pair ? gv : 0.0;
            const double r = dot(pair && even ? tl : 0.0, 1.0);
            GJP_ADD(GJP_LOGL, t0);
            return r;
        }
    }
    __device__ __forceinline__ double logp(double x) const
    {
        if (a.logp_kind == PTMI_LOGP_BOX) {
            const bool ok = !act || ((blo <= x) & (bhi >= x));
            return half_all(ok) ? 0.0 : -__builtin_inf();
        }
        return 0.0;
    }
    __device__ __forceinline__ double func_grad_white(double q, double &gradw) const       // NJ:71-90
    {
        const double x = tab_vec<GJT_BACKWARD>(nullptr, q);
        double g;
        const double ll = logl_grad(x, g);
        const double lp = logp(x);
        g = beta * g + 0.0;
        gradw = tab_vec<GJT_GRADIENT>(nullptr, g);
        return beta * ll + lp;
    }
    __device__ __forceinline__ double joint_of(double logl, double r) const { return logl - 0.5 * dot(r, r); }
    __device__ __forceinline__ double leapfrog(double theta, double r, double grad, double eps, double &to, double &ro, double &go)   // NJ:149-169
    {
        nleap += 1;
        const double he = 0.5 * eps;
        const double rh = r + he * grad;
        const double tn = theta + eps * rh;
        double gn;
        const double lpp = func_grad_white(tn, gn);
        to = tn;
        go = gn;
        ro = rh + he * gn;
        return lpp;
    }
    __device__ __forceinline__ bool keep_going(double tm, double tp, double rm, double rp) const                // NJ:465-493
    {
        const double dt = tp - tm;
        const double cx = row_chain(dt, rm), cy = row_chain(dt, rp);      // two independent chains: they overlap
        const double x = rows_sum(cx), y = rows_sum(cy);
        return (x >= 0.0) & (y >= 0.0);
    }
    __device__ __forceinline__ bool any_inf(double v) const { return half_any(act && __builtin_isinf(v)); }

    // ------------------------------------------------------------------ HMC (NJ:238-291)
    __device__ __forceinline__ double hmc(double *st, double x, double &qout)
    {
        st[GJ_HITER] += 1.0;
        double q = tab_vec<GJT_FORWARD>(nullptr, x), grad;
        const double logp0 = func_grad_white(q, grad);
        double p = momenta();
        const double joint0 = joint_of(logp0, p);
        const int nsteps = randint(a.hmc_min, a.hmc_max);
        double joint1 = joint0;
        for (int k = 0; k < nsteps; ++k) {
            const double logp1 = leapfrog(q, p, grad, a.hmc_eps, q, p, grad);
            joint1 = joint_of(logp1, p);
            if (joint1 - 1000.0 < joint0) break;                         // NJ:284-286
        }
        qout = tab_vec<GJT_BACKWARD>(nullptr, q);
        return joint1 - joint0;
    }

    // ------------------------------------------------------------------ NUTS
    __device__ __forceinline__ double find_reasonable_epsilon(double theta0, double grad0, double logp0)   // NJ:435-463, loops bounded
    {
        double tp, rp, gp;
        double eps = 1.0;
        const double r0 = momenta();
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

    struct Tree {
        double far_t, far_r, cand_t, cand_g;
        double logp, alpha;
        long long n, nalpha;
        int s;
    };
    // the tree stack (see GradJump::build_tree): the entry of height h in slot h, pending heights in a mask.  Heights below
    // gj_lds_levels (11: trees of up to 2^11 leapfrogs) are in the block's LDS; the higher ones -- the reference doubles without a
    // cap (NJ:716-802), the ABI allows 24 -- in the wave's slice of the global scratch: reached once in 2^h leapfrogs, if ever,
    // so only their correctness matters (keeping all 25 in LDS cost the config-5 kernel 25 %: five waves per CU instead of eight)
    __device__ __forceinline__ int slot_of(int h) const { return a.gj_stack_off + (hh * a.gj_lds_levels + h) * gjw_level_doubles(EPL); }
    __device__ __forceinline__ double *glevel(int h) const
    {
        return a.gj_scr + (((size_t)blockIdx.x * 2 + (size_t)hh) * (size_t)(a.nuts_maxdepth + 1) + (size_t)h) * gjw_level_doubles(EPL);
    }
    // NJ:495-652 as a loop, as GradJump::build_tree
    __device__ __forceinline__ void build_tree(double &tg, double &rg, double &gg, double logu, int v, int j, double eps, double joint0, Tree &cur)
    {
        u32 pend = 0;
        for (;;) {
            const double logpp = leapfrog(tg, rg, gg, (double)v * eps, tg, rg, gg);
            GJP_T0(tl0);
            const double joint = joint_of(logpp, rg);
            cur.n = logu < joint;
            cur.s = (logu - 1000.0) < joint;
            cur.far_t = tg; cur.far_r = rg; cur.cand_t = tg; cur.cand_g = gg;
            cur.logp = logpp;
            const double ex = det_exp(joint - joint0);
            cur.alpha = ex < 1.0 ? ex : 1.0;                             // Python's min(1.0, e): 1.0 when e is NaN
            cur.nalpha = 1;
            GJP_ADD(GJP_LEAF, tl0);
            int h = 0;
            for (;;) {
                const int top_h = pend ? (int)__builtin_ctz(pend) : -1;
                if (top_h == h) {                                        // cur is the right sibling of the stack top
                    GJP_T0(tm0);
                    pend &= pend - 1u;
                    double t_logp, t_n, t_alpha, t_nalpha, e_ct, e_cg, e_ft, e_fr;
                    if (h < a.gj_lds_levels) {                                     // wave-uniform
                        const int b = slot_of(h);
                        t_logp = gj_lds[b + GJL_VECS * LD + GJS_LOGP]; t_n = gj_lds[b + GJL_VECS * LD + GJS_N];
                        t_alpha = gj_lds[b + GJL_VECS * LD + GJS_ALPHA]; t_nalpha = gj_lds[b + GJL_VECS * LD + GJS_NALPHA];
                        e_ct = act ? gj_lds[b + GJL_CAND_T * LD + col] : 0.0; e_cg = act ? gj_lds[b + GJL_CAND_G * LD + col] : 0.0;
                        e_ft = act ? gj_lds[b + GJL_FAR_T * LD + col] : 0.0; e_fr = act ? gj_lds[b + GJL_FAR_R * LD + col] : 0.0;
                    } else {
                        const double *gp = glevel(h);
                        t_logp = gp[GJL_VECS * LD + GJS_LOGP]; t_n = gp[GJL_VECS * LD + GJS_N];
                        t_alpha = gp[GJL_VECS * LD + GJS_ALPHA]; t_nalpha = gp[GJL_VECS * LD + GJS_NALPHA];
                        e_ct = act ? gp[GJL_CAND_T * LD + col] : 0.0; e_cg = act ? gp[GJL_CAND_G * LD + col] : 0.0;
                        e_ft = act ? gp[GJL_FAR_T * LD + col] : 0.0; e_fr = act ? gp[GJL_FAR_R * LD + col] : 0.0;
                    }
                    const long long tot = (long long)t_n + cur.n;
                    const double den = (double)tot > 1.0 ? (double)tot : 1.0;
                    const bool take_u = uniform() < (double)cur.n / den;
                    if (!take_u) {
                        cur.cand_t = e_ct;
                        cur.cand_g = e_cg;
                        cur.logp = t_logp;
                    }
                    cur.far_t = e_ft;
                    cur.far_r = e_fr;
                    cur.n = tot;
                    const bool go = v == 1 ? keep_going(cur.far_t, tg, cur.far_r, rg) : keep_going(tg, cur.far_t, rg, cur.far_r);
                    cur.s = cur.s && go;                                 // the popped tree has s = 1
                    cur.alpha = t_alpha + cur.alpha;
                    cur.nalpha = (long long)t_nalpha + cur.nalpha;
                    h += 1;
                    GJP_ADD(GJP_MERGE, tm0);
                    continue;
                }
                if (h == j) return;
                if (cur.s == 0) {
                    if (pend == 0) return;
                    h = top_h;
                    continue;
                }
                GJP_T0(tp0);
                if (h < a.gj_lds_levels) {                               // push: wait for the right sibling
                    const int b = slot_of(h);
                    if (act) {                                           // in element order: 4 LD doubles per entry instead of 4 x 64
                        gj_lds[b + GJL_FAR_T * LD + wi] = cur.far_t;
                        gj_lds[b + GJL_FAR_R * LD + wi] = cur.far_r;
                        gj_lds[b + GJL_CAND_T * LD + wi] = cur.cand_t;
                        gj_lds[b + GJL_CAND_G * LD + wi] = cur.cand_g;
                    }
                    if ((L & 31) == 0) {
                        gj_lds[b + GJL_VECS * LD + GJS_LOGP] = cur.logp;
                        gj_lds[b + GJL_VECS * LD + GJS_N] = (double)cur.n;
                        gj_lds[b + GJL_VECS * LD + GJS_ALPHA] = cur.alpha;
                        gj_lds[b + GJL_VECS * LD + GJS_NALPHA] = (double)cur.nalpha;
                    }
                } else {
                    double *gp = glevel(h);
                    if (act) {
                        gp[GJL_FAR_T * LD + wi] = cur.far_t;
                        gp[GJL_FAR_R * LD + wi] = cur.far_r;
                        gp[GJL_CAND_T * LD + wi] = cur.cand_t;
                        gp[GJL_CAND_G * LD + wi] = cur.cand_g;
                    }
                    if ((L & 31) == 0) {
                        gp[GJL_VECS * LD + GJS_LOGP] = cur.logp;
                        gp[GJL_VECS * LD + GJS_N] = (double)cur.n;
                        gp[GJL_VECS * LD + GJS_ALPHA] = cur.alpha;
                        gp[GJL_VECS * LD + GJS_NALPHA] = (double)cur.nalpha;
                    }
                    __threadfence_block();                               // the wave's own global stores, visible to its other lanes
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    // one wave: its LDS / global accesses are served in order; the fence orders the compiler (no barrier inside divergent code)
                pend |= 1u << h;
                GJP_ADD(GJP_PUSH, tp0);
                break;
            }
        }
    }

    // NUTSJump.__call__ (NJ:654-840), as GradJump::nuts; the two ends of the trajectory and the sample stay in registers
    __device__ __forceinline__ double nuts(double *st, double x, double &qout)
    {
        st[GJ_NITER] += 1.0;
        const double q = tab_vec<GJT_FORWARD>(nullptr, x);
        double grad;
        const double logp0 = func_grad_white(q, grad);
        if (st[GJ_HAVE_EPS] == 0.0) {
            st[GJ_EPS] = find_reasonable_epsilon(q, grad, logp0);
            st[GJ_MU] = det_log(10.0 * st[GJ_EPS]);
            st[GJ_HAVE_EPS] = 1.0;
        }
        const double r0 = momenta();
        const double joint = joint_of(logp0, r0);
        const double logu = joint - exponential();
        double lnprob = logp0;
        double sample = q, tm = q, rm = r0, gm = grad, tp = q, rp = r0, gp = grad;
        int j = 0, s = 1;
        long long n = 1;
        double alpha = 0.0;
        long long nalpha = 1;
        const double eps = st[GJ_EPS];
        while (s == 1) {
            const int dir = 2 * (int)(uniform() < 0.5) - 1;
            double tg = dir == -1 ? tm : tp, rg = dir == -1 ? rm : rp, gg = dir == -1 ? gm : gp;
            Tree t;
            build_tree(tg, rg, gg, logu, dir, j, eps, joint, t);
            if (dir == -1) { tm = tg; rm = rg; gm = gg; } else { tp = tg; rp = rg; gp = gg; }
            if (t.s == 1) {
                const double ratio = (double)t.n / (double)n;
                if (uniform() < (1.0 < ratio ? 1.0 : ratio)) { sample = t.cand_t; lnprob = t.logp; }
            }
            n += t.n;
            const bool go = keep_going(tm, tp, rm, rp);
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
            // seven evaluations of exp / log on values every lane of the half holds alike: the independent ones go side by side on
            // two lanes (log it_call beside log eps_bar, then the two exponentials), four evaluations deep instead of seven; the
            // same functions of the same arguments
            const bool odd = L & 1;
            const double lg = det_log(odd ? st[GJ_EPSBAR] : it_call);
            const double log_it = half_lanef(lg, 0), log_epsbar = half_lanef(lg, 1);
            const double ex = det_exp(odd ? -0.75 * log_it : st[GJ_MU] - det_sqrt(it_call) / 0.05 * st[GJ_HBAR]);
            st[GJ_EPS] = half_lanef(ex, 0);
            eta = half_lanef(ex, 1);
            st[GJ_EPSBAR] = det_exp((1.0 - eta) * log_epsbar + eta * det_log(st[GJ_EPS]));
        } else {
            st[GJ_EPS] = st[GJ_EPSBAR];
        }
        qout = tab_vec<GJT_BACKWARD>(nullptr, sample);
        return logp0 - lnprob;                                           // undoes the outer Hastings ratio (NJ:838)
    }
};


// Fused MH steps with the gradient jumps in the cycle: the non-staged full kernel (ptmi_mh.inc.h) plus the NUTS / HMC
// branch.  Every chain group runs its nsteps iterations on its own, so a long NUTS tree delays only its wave for that
// iteration and the launch costs the longest SUM over iterations, not the sum of the per-iteration maxima.
// One wave per block: there is no block-level cooperation, and single-wave blocks let the dispatcher backfill the
// SIMDs as soon as a wave's chains are through their (very unequal) trees.
constexpr int GJ_BLOCK = 64;
#ifndef PTMI_GJ_WPE
#define PTMI_GJ_WPE 2
#endif
// PAIR (4-lane shapes, diagonal whitening, iso / curved families): two gradient jumps at a time, a half-wave each (GradJumpPair)
// W16 (the 16-lane shape at ndim <= 64): a gradient jump takes the whole wave there too
// (GradJumpWide<16, LOGL, 16>: one element per lane instead of seven slots of every vector in each of the chain's 16 lanes)
template <int G, int EPL, int LOGL, bool PAIR = false, int W16 = 0 /* 1: diagonal whitening tables, 2: full ones (two instantiations: the product's code costs the diagonal kernel 14 %) */>
__global__ __launch_bounds__(GJ_BLOCK, ((G == 4 && EPL <= 5) || W16) ? PTMI_GJ_WPE : 1) void mh_steps_gj_kernel(const KArgs a)
{
    static_assert(!PAIR || (G == 4 && EPL <= 8 && LOGL != PTMI_LOGL_DENSE), "the pair layout serves the 4-lane shapes without table products");
    static_assert(!W16 || (G == 16 && !PAIR), "the 16-group whole-wave layout serves the 16-lane shape");
    constexpr int CPB = GJ_BLOCK / G;
    constexpr bool WIDE = G == 4 || W16;         // a gradient jump takes the whole wave (GradJumpWide)
    constexpr int WEPL = W16 ? 16 : EPL, WNS = W16 ? 4 : EPL;          // GradJumpWide's EPL (LD / 4); slots of a lane that hold elements in its layout
    const int d = a.d, nt = a.nt;
    const long long nch = (long long)a.W * nt;
    if (WIDE && W16 != 2) {                      // (W16 with full whitening tables: they stay in global memory, GradJumpWide::tab_vec)
        constexpr int LD = 4 * WEPL;
        if (a.gj_diag) {                         // diagonal whitening: the three diagonals (3 LD doubles instead of 3 LD^2: 0.5 KB instead of 9.6 KB at d = 20)
            for (int i = (int)threadIdx.x; i < 3 * LD; i += GJ_BLOCK) {
                const int w = i / LD, c = i % LD;
                gj_lds[i] = c < d ? a.gj_tab[((size_t)w * d + c) * d + c] : 0.0;
            }
        } else {
            for (int i = (int)threadIdx.x; i < 3 * LD * LD; i += GJ_BLOCK) {
                const int w = i / (LD * LD), r = (i / LD) % LD, c = i % LD;
                gj_lds[i] = (r < d && c < d) ? a.gj_tab[((size_t)w * d + r) * d + c] : 0.0;
            }
        }
    }
    box_table_fill<G, EPL>(a, gj_lds, GJ_BLOCK);
    __syncthreads();
    const long long cslot0 = (long long)blockIdx.x * CPB + (int)(threadIdx.x / G);
    // a chain group without a chain (past the end, or an empty slot of the launch order) repeats a chain and writes nothing (the whole
    // wave takes part in the gradient jumps)
    const int oslot = a.gj_order ? (cslot0 < (long long)a.gj_nslots ? a.gj_order[cslot0] : -1) : 0;
    const bool live = a.gj_order ? oslot >= 0 : cslot0 < nch;
    if (!WIDE && !live) return;                  // no block-wide synchronisation below for the wider layouts: whole chain groups may leave
    const long long cslot = live ? cslot0 : nch - 1;
    // The chains of a wave run in lock step: every iteration costs the wave its longest tree, and the launch ends with its
    // slowest wave.  On the curved likelihood a per cent of the ranks keep a small NUTS step size (trees of ~100
    // leapfrogs) while the rest run away to huge ones (one leapfrog): the host deals the chains over the waves by step
    // size (gj_order_*), longest trees first and one per wave, so that no wave has to add up several long trees per
    // iteration (tools/gj_census.py).  The chains with the very longest trees have a wave to themselves (the other chain slots of
    // that wave are empty): what a launch lasts is its slowest chain's own serial work, and there it does not wait for fifteen
    // others' jumps and steps.  Which lanes host a chain does not enter its arithmetic.
    const long long ch = a.gj_order ? (live ? (long long)oslot : 0) : cslot;
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
    const double *PtG = LOGL == PTMI_LOGL_DENSE ? a.logl_par + d + (size_t)a.d * a.d : nullptr;   // eval_logl's half table Tl
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
    const bool cold = live && tg == 0 && a.AM != nullptr;
    int am_row = a.am_row0;
    long long am_next = a.am_base != nullptr ? a.am_base[ch] : 0;      // the chain's next AM increment computed ahead of the launch (am_gemm_kernel)
    // PAIR: the step-size states of the wave's 16 chains live in LDS for the launch (behind the exchange area; two doubles per lane
    // in, the same two out at the end): a call began with eight dependent reads from global memory and ended with a store the
    // chain's next call had to see (a fence), some thousand cycles of a one-leapfrog call's twelve
    const int stl = a.gj_stack_off + 2 * a.gj_lds_levels * gjw_level_doubles(EPL) + 72;
    if constexpr (PAIR) {
        static_assert(GJ_NSTATE == 2 * G, "two state words per lane of a chain");
        gj_lds[stl + 2 * (int)threadIdx.x] = stg[2 * gl];
        gj_lds[stl + 2 * (int)threadIdx.x + 1] = stg[2 * gl + 1];
        __syncthreads();
    }

#ifdef PTMI_GJ_PROFILE
    unsigned long long prof_sum[GJP_N + 2] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_k0 = __builtin_readcyclecounter();
#endif
    for (int k = 0; k < a.nsteps; ++k) {
        const long long it = a.iter0 + k;
        double q[EPL], qxy = 0.0;
        Draws dr;
        draws_for_step<false, true>(batch, dr, a, k, sid, sid0, gl);
        const double log_u = dr.log_u;
        const int jt = propose<G, EPL, true, false, false, true>(a, it, sid, gl, cc, dr, Ut, false, S, DE, q, false, true, nullptr, &am_next);
        const bool is_gj = jt == PTMI_J_NUTS || jt == PTMI_J_HMC;
        if constexpr (PAIR) {
            // two chains at a time, a half-wave each: their rows go through LDS into the pair layout (area 32 hh + 16 (g >> 1) +
            // 8 (g & 1) + e of the exchange area), the proposals and the two qxy come back the same way
            u64 todo = __ballot(is_gj && live);
            const int xch = a.gj_stack_off + 2 * a.gj_lds_levels * gjw_level_doubles(EPL);
            const int L = (int)threadIdx.x, hh = L >> 5;
            const int myidx = 16 * (gl >> 1) + 8 * (gl & 1);              // where this lane's slots sit inside a half's area
            while (todo) {
                const int laneA = (int)__builtin_ctzll(todo);
                todo &= ~(0xFull << laneA);
                const bool two = todo != 0;
                const int laneB = two ? (int)__builtin_ctzll(todo) : laneA;
                if (two) todo &= ~(0xFull << laneB);
                const bool mineA = (L & ~3) == laneA, mineB = two && (L & ~3) == laneB;
                if (mineA || mineB) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) gj_lds[xch + (mineB ? 32 : 0) + myidx + e] = x[e];
                }
                __syncthreads();
                const bool on = hh == 0 || two;                             // this half has a chain
                const int lane0 = hh ? laneB : laneA;
                const double xw = (L & 7) < EPL ? gj_lds[xch + L] : 0.0;
                const long long chA = ((long long)__builtin_amdgcn_readlane((int)(ch >> 32), laneA) << 32) | (u32)__builtin_amdgcn_readlane((int)ch, laneA);
                const long long chB = ((long long)__builtin_amdgcn_readlane((int)(ch >> 32), laneB) << 32) | (u32)__builtin_amdgcn_readlane((int)ch, laneB);
                const long long ch_c = hh ? chB : chA;
                const double beta_c = hh ? lane_get(beta, laneB) : lane_get(beta, laneA);
                const u32 sid_c = hh ? (u32)__builtin_amdgcn_readlane((int)sid, laneB) : (u32)__builtin_amdgcn_readlane((int)sid, laneA);
                const int jt_c = hh ? __builtin_amdgcn_readlane(jt, laneB) : __builtin_amdgcn_readlane(jt, laneA);
                double qw = 0.0, qxy_c = 0.0;
                if (on) {
                    double *stc = gj_lds + stl + 2 * lane0;                  // the chain's eight words (its four lanes' pairs)
                    GradJumpPair<EPL, LOGL> gj(a, ch_c, beta_c, it, sid_c, xch);
                    double st[GJ_NSTATE];
#pragma unroll
                    for (int j = 0; j < GJ_NSTATE; ++j) st[j] = stc[j];
                    qxy_c = jt_c == PTMI_J_NUTS ? gj.nuts(st, xw, qw) : gj.hmc(st, xw, qw);
                    st[GJ_NLEAP] += (double)gj.nleap;
                    if ((L & 31) == 0) {
#pragma unroll
                        for (int j = 0; j < GJ_NSTATE; ++j) stc[j] = st[j];
                    }
                }
                gj_lds[xch + L] = qw;
                if ((L & 31) == 0) gj_lds[xch + 64 + hh] = qxy_c;
                __syncthreads();
                if (mineA || mineB) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) q[e] = gj_lds[xch + (mineB ? 32 : 0) + myidx + e];
                    qxy = gj_lds[xch + 64 + (mineB ? 1 : 0)];
                }
                __syncthreads();                 // the exchange area is free for the next pair
            }
            if (!is_gj) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) q[e] = x[e] + q[e];            // propose() returned the increment
            }
        } else if constexpr (WIDE) {
            // one chain after the other, each on all 64 lanes: its row goes through LDS into the whole-wave layout and the
            // proposal comes back the same way; everything in between is wave-uniform
            u64 todo = __ballot(is_gj && live);
            const int xch = a.gj_stack_off + a.gj_lds_levels * gjw_level_doubles(WEPL);
            const int L = (int)threadIdx.x;
            constexpr int XS = W16 ? 4 : 16;                                     // a lane group's elements sit XS apart in the whole-wave layout
            while (todo) {
                const int lane0 = (int)__builtin_ctzll(todo);                    // first lane of the chain
                todo &= ~((G == 4 ? 0xFull : 0xFFFFull) << lane0);
                const bool mine = (L & ~(G - 1)) == lane0;
                if (mine) {
#pragma unroll
                    for (int e = 0; e < WNS; ++e) gj_lds[xch + XS * gl + e] = x[e];
                }
                __syncthreads();
                const double xw = (W16 || (L & 15) < EPL) ? gj_lds[xch + L] : 0.0;
                const long long ch_c = ((long long)__builtin_amdgcn_readlane((int)(ch >> 32), lane0) << 32) | (u32)__builtin_amdgcn_readlane((int)ch, lane0);
                const double beta_c = lane_get(beta, lane0);
                const u32 sid_c = (u32)__builtin_amdgcn_readlane((int)sid, lane0);
                const int jt_c = __builtin_amdgcn_readlane(jt, lane0);
                const int w_c = (int)(ch_c / nt), t_c = __builtin_amdgcn_readlane(t, lane0);
                double *stc = a.gj + ((size_t)w_c * nt + t_c) * GJ_NSTATE;
                double qw;
                GradJumpWide<WEPL, LOGL, W16 ? 16 : 4, W16 == 2> gj(a, ch_c, beta_c, it, sid_c, xch);
                double st[GJ_NSTATE];
#pragma unroll
                for (int j = 0; j < GJ_NSTATE; ++j) st[j] = stc[j];
                GJP_T0(tc0);
                const double qxy_c = jt_c == PTMI_J_NUTS ? gj.nuts(st, xw, qw) : gj.hmc(st, xw, qw);
#ifdef PTMI_GJ_PROFILE
                gj.prof[GJP_CALL] += __builtin_readcyclecounter() - tc0;
                for (int j = 0; j < GJP_N; ++j) prof_sum[j] += gj.prof[j];
                prof_sum[GJP_N] += gj.nleap;
                prof_sum[GJP_N + 1] += 1;
#endif
                st[GJ_NLEAP] += (double)gj.nleap;
                if (L == 0) {
#pragma unroll
                    for (int j = 0; j < GJ_NSTATE; ++j) stc[j] = st[j];
                }
                __threadfence_block();           // the wave reads the state again at the chain's next gradient jump
                gj_lds[xch + L] = qw;
                __syncthreads();
                if (mine) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) q[e] = e < WNS ? gj_lds[xch + XS * gl + e] : 0.0;       // (W16: ndim <= 64, the slots beyond hold no element)
                    qxy = qxy_c;
                }
                __syncthreads();                 // the exchange area is free for the next chain
            }
            if (!is_gj) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) q[e] = x[e] + q[e];            // propose() returned the increment
            }
        } else {
            if (is_gj) {
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
        }
#pragma unroll
        for (int j = 0; j < PTMI_J_NTYPES; ++j) jp[j] += (jt == j);
        // PT:605-612
        const double nlp = eval_logp<G, EPL, false>(a, q, gl, gj_lds);
        const double nlnL = eval_logl<G, EPL, LOGL, false>(a, q, gl, PtG);
        const double nlnprob = nlp == -__builtin_inf() ? -__builtin_inf() : beta * nlnL + nlp;
        // PT:615-622
        const double lnprob0 = beta * lnL + lp;
        const double diff = nlnprob - lnprob0 + qxy;
        const bool accepted = diff > log_u;
        if (accepted) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) x[e] = q[e];
            lnL = nlnL;
            lp = nlp;
            nacc += 1;
#pragma unroll
            for (int j = 0; j < PTMI_J_NTYPES; ++j) ja[j] += (jt == j);
        }
        // PT:327-328 (the post-swap row of a swap iteration is written by the swap); with AM row flags (ptmi_common.h) only a new
        // or a KEY row is stored (am_store_step's rule)
        if (cold && !(a.swap_last && k == a.nsteps - 1)) {
            double *am = a.AM + ((size_t)w * a.cov_update + (size_t)am_row) * d;
            bool store = true;
            if (a.AMflag != nullptr) {
                const bool key = k == 0 || am_row <= 1;
                if (gl == 0) a.AMflag[(size_t)w * a.cov_update + (size_t)am_row] = (key ? AMROW_KEY : 0ull) | (accepted ? AMROW_NEW : 0ull);
                store = key || accepted;
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = gl + G * e;
                if (store && i < d) am[i] = x[e];
            }
            if (a.AMaux && gl == 0) {
                double *ax = a.AMaux + ((size_t)w * a.cov_update + (size_t)am_row) * 2;
                ax[0] = lnL;
                ax[1] = lp;
            }
        }
        am_row = am_row + 1 == a.cov_update ? 0 : am_row + 1;
    }
#ifdef PTMI_GJ_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("gjprof kernel %llu | calls %llu leaps %llu | call %llu tabvec %llu logl %llu dot %llu leaf %llu merge %llu push %llu draw %llu\n",
               __builtin_readcyclecounter() - prof_k0, prof_sum[GJP_N + 1], prof_sum[GJP_N], prof_sum[GJP_CALL], prof_sum[GJP_TABVEC], prof_sum[GJP_LOGL],
               prof_sum[GJP_DOT], prof_sum[GJP_LEAF], prof_sum[GJP_MERGE], prof_sum[GJP_PUSH], prof_sum[GJP_DRAW]);
#endif
    if (!live) return;
    if constexpr (PAIR) {                         // (the last call's LDS stores are behind that call's barriers)
        stg[2 * gl] = gj_lds[stl + 2 * (int)threadIdx.x];
        stg[2 * gl + 1] = gj_lds[stl + 2 * (int)threadIdx.x + 1];
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
