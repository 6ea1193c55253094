/*
 * ptmcmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, one-chain-at-a-time restatement of the Metropolis-Hastings hot
 * path of nanograv/PTMCMCSampler (reference file PTMCMCSampler/PTMCMCSampler.py,
 * cited per function below as PT:<lines>).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product path
 * (ptmcmcsampler_amd/) never does and fails loudly without its HIP library.
 *
 * Two draw sources:
 *   - REPLAY: the reference's own recorded np.random.Generator draws (fixtures
 *     under tests/golden/, made by tests/golden/make_golden.py).  With these the
 *     oracle must reproduce the reference's trajectories -> this is how the
 *     oracle is PINNED to the reference.
 *   - PHILOX: the counter-based Philox4x32-10 schedule the HIP kernels use.
 *     With these the oracle defines the bit-exact expected output of the GPU.
 *
 * Arithmetic is spelled so that it is reproducible bit-for-bit on any IEEE-754
 * machine: only + - * / sqrt and explicit fma(), fixed summation orders
 * (a G-lane strided partial sum followed by an xor-butterfly, G = `lanes`),
 * and its own log/exp/cos(2*pi*u).  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ bits */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ------------------------------------------------------- Philox4x32-10 */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11). */
ORC_API void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* one call = two 64-bit words */
static void philox_words(uint64_t seed, uint64_t iter, uint32_t stream, uint32_t slot, uint64_t w[2])
{
    uint32_t ctr[4] = { (uint32_t)iter, (uint32_t)(iter >> 32), stream, slot };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t o[4];
    orc_philox(ctr, key, o);
    w[0] = ((uint64_t)o[1] << 32) | o[0];
    w[1] = ((uint64_t)o[3] << 32) | o[2];
}

/* slot numbers of the counter-based schedule (DESIGN.md "RNG schedule"):
 *   A: w0 cycle pick, w1 scale-branch uniform     B: w0 accept uniform, w1 SCAM direction / DE row mm
 *   C: w0 DE row nn offset, w1 DE scale uniform    D: (w0,w1) SCAM normal
 *   SWAP+k: w0 uniform of pair (k,k+1), stream of rank 0
 *   AM+k: (w0,w1) Box-Muller pair: cos branch -> eigen-direction k, sin branch -> direction k+lanes, (k/lanes) even */
enum { SLOT_A = 0, SLOT_B = 1, SLOT_C = 2, SLOT_D = 3, SLOT_SWAP = 0x10000, SLOT_AM = 0x1000000 };

static inline double w2uniform(uint64_t w) { return (double)(w >> 11) * 0x1.0p-53; }        /* [0,1) */
static inline double w2uniform_open(uint64_t w) { return (double)((w >> 11) + 1) * 0x1.0p-53; } /* (0,1] */
static inline uint64_t w2index(uint64_t w, uint64_t n) { return (uint64_t)(((unsigned __int128)w * n) >> 64); }

/* ------------------------------------------------- deterministic libm */
ORC_API double orc_log(double x)
{
    if (x != x) return x;
    if (x <= 0.0) return x == 0.0 ? -INFINITY : NAN;
    if (x == INFINITY) return x;
    uint64_t u = d2u(x);
    int k = 0;
    if ((u >> 52) == 0) { x *= 0x1.0p54; u = d2u(x); k = -54; }          /* subnormal */
    k += (int)(u >> 52) - 1023;
    uint64_t man = u & 0x000FFFFFFFFFFFFFull;
    /* m in [sqrt(1/2), sqrt(2)) */
    if (man >= 0x6A09E667F3BCDull) { k += 1; u = man | 0x3FE0000000000000ull; }
    else u = man | 0x3FF0000000000000ull;
    double f = u2d(u) - 1.0;
    double s = f / (2.0 + f);
    double z = s * s, w = z * z;
    double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

ORC_API double orc_exp(double x)
{
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    int k = (int)(1.44269504088896338700e+00 * x + (x < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double hi = x - dk * 6.93147180369123816490e-01;
    double lo = dk * 1.90821492927058770002e-10;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
               t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k == 0) return y;
    if (k == 1024) return y * 2.0 * 0x1.0p1023;
    if (k >= -1021) return u2d(d2u(y) + ((uint64_t)(int64_t)k << 52));
    return u2d(d2u(y) + ((uint64_t)(int64_t)(k + 1000) << 52)) * 0x1.0p-1000;
}

/* cos(2*pi*u), u in [0,1): quarter-turn reduction is exact, then Taylor in r */
ORC_API double orc_cos2pi(double u)
{
    double a = 4.0 * u;
    double q = floor(a + 0.5);
    double r = a - q;               /* [-1/2, 1/2], exact */
    double z = r * r;
    int qi = (int)q & 3;
    if (qi & 1) {                   /* +-sin(pi r/2) */
        double p = -0x1.8a404211f9547p-45;
        p = fma(p, z, 0x1.aaec32af93359p-38);
        p = fma(p, z, -0x1.6fadb9f155744p-31);
        p = fma(p, z, 0x1.e8f434d018d63p-25);
        p = fma(p, z, -0x1.e3074fde8871fp-19);
        p = fma(p, z, 0x1.50783487ee782p-13);
        p = fma(p, z, -0x1.32d2cce62bd86p-8);
        p = fma(p, z, 0x1.466bc6775aae2p-4);
        p = fma(p, z, -0x1.4abbce625be53p-1);
        p = fma(p, z, 0x1.921fb54442d18p+0);
        double s = p * r;
        return qi == 1 ? -s : s;
    } else {                        /* +-cos(pi r/2) */
        double p = 0x1.ef6e308d6d1c4p-49;
        p = fma(p, z, -0x1.2a0c591af8314p-41);
        p = fma(p, z, 0x1.20c62c2f2d7f5p-34);
        p = fma(p, z, -0x1.b6e24f44b128fp-28);
        p = fma(p, z, 0x1.f9d38a3763cc3p-22);
        p = fma(p, z, -0x1.a6d1f2a204a8cp-16);
        p = fma(p, z, 0x1.e1f506891babbp-11);
        p = fma(p, z, -0x1.55d3c7e3cbffap-6);
        p = fma(p, z, 0x1.03c1f081b5ac4p-2);
        p = fma(p, z, -0x1.3bd3cc9be45dep+0);
        p = fma(p, z, 1.0);
        return qi == 0 ? p : -p;
    }
}

/* sin(2*pi*u), same reduction and polynomials */
ORC_API double orc_sin2pi(double u)
{
    double a = 4.0 * u;
    double q = floor(a + 0.5);
    double r = a - q;
    double z = r * r;
    int qi = (int)q & 3;
    if (qi & 1) {                   /* +-cos(pi r/2) */
        double p = 0x1.ef6e308d6d1c4p-49;
        p = fma(p, z, -0x1.2a0c591af8314p-41);
        p = fma(p, z, 0x1.20c62c2f2d7f5p-34);
        p = fma(p, z, -0x1.b6e24f44b128fp-28);
        p = fma(p, z, 0x1.f9d38a3763cc3p-22);
        p = fma(p, z, -0x1.a6d1f2a204a8cp-16);
        p = fma(p, z, 0x1.e1f506891babbp-11);
        p = fma(p, z, -0x1.55d3c7e3cbffap-6);
        p = fma(p, z, 0x1.03c1f081b5ac4p-2);
        p = fma(p, z, -0x1.3bd3cc9be45dep+0);
        p = fma(p, z, 1.0);
        return qi == 1 ? p : -p;
    } else {                        /* +-sin(pi r/2) */
        double p = -0x1.8a404211f9547p-45;
        p = fma(p, z, 0x1.aaec32af93359p-38);
        p = fma(p, z, -0x1.6fadb9f155744p-31);
        p = fma(p, z, 0x1.e8f434d018d63p-25);
        p = fma(p, z, -0x1.e3074fde8871fp-19);
        p = fma(p, z, 0x1.50783487ee782p-13);
        p = fma(p, z, -0x1.32d2cce62bd86p-8);
        p = fma(p, z, 0x1.466bc6775aae2p-4);
        p = fma(p, z, -0x1.4abbce625be53p-1);
        p = fma(p, z, 0x1.921fb54442d18p+0);
        double s = p * r;
        return qi == 0 ? s : -s;
    }
}

/* Box-Muller (cos branch): one normal from two words */
ORC_API double orc_normal(uint64_t w0, uint64_t w1)
{
    double r = sqrt(-2.0 * orc_log(w2uniform_open(w0)));
    return r * orc_cos2pi(w2uniform(w1));
}
/* Box-Muller, sin branch: the second normal of the same two words */
ORC_API double orc_normal_sin(uint64_t w0, uint64_t w1)
{
    double r = sqrt(-2.0 * orc_log(w2uniform_open(w0)));
    return r * orc_sin2pi(w2uniform(w1));
}
ORC_API double orc_uniform(uint64_t w) { return w2uniform(w); }
ORC_API uint64_t orc_index(uint64_t w, uint64_t n) { return w2index(w, n); }

/* ------------------------------------------------------------- config */
enum { LOGL_ISO = 0, LOGL_DENSE = 1, LOGL_CURVED = 2 };
enum { LOGP_FLAT = 0, LOGP_BOX = 1 };
enum { J_SCAM = 0, J_AM = 1, J_DE = 2, J_NTYPES = 3 };
enum { K_INT = 0, K_UNI = 1, K_NRM = 2, K_SHUF = 3 };

typedef struct {
    int32_t ndim, ntemps, nwalkers, lanes;
    int32_t logl_kind, logp_kind;
    int32_t w_scam, w_am, w_de;       /* cycle = [SCAM]*w_scam + [AM]*w_am + [DE]*w_de  (PT:261-264, 579-585) */
    int32_t de_on, de_size;           /* DE in the cycle; rows in a DE buffer (= burn, PT:221) */
    int32_t cov_update, tskip;        /* AM-buffer length (PT:220); swap period (PT:624), 0 = never */
    int32_t cov_per_walker;           /* 1: Ut/S/DE per walker, 0: one shared set */
    int32_t ntemps_global, temp0, walker0, ngroups;   /* ngroups: parameter groups (PT:129-145); 0 or 1 = one full group */
    uint64_t seed;
    const double *logl_par;           /* DENSE: mu[d], Pt[d*d] (Pt[j*d+i] = P[i][j]) */
    const double *logp_par;           /* BOX: lo[d], hi[d] */
    const double *temps_mh;           /* [ntemps] temperature of each local rank as the MH step sees it (PT:278-282) */
    const double *beta;               /* [ntemps] 1/temps_mh */
    const int32_t *gsize;             /* [ngroups] parameters per group (NULL with one full group) */
    const double *gmask;              /* [ngroups][d] 1 where the parameter belongs to the group */
} orc_cfg;

typedef struct {
    double *X;          /* [W][ntemps][d]   rows by SLOT */
    double *lnL, *lp;   /* [W][ntemps]      by SLOT */
    int32_t *temp_of;   /* [W][ntemps]      local rank held by the row in a slot */
    int32_t *slot_of;   /* [W][ntemps]      inverse */
    double *Ut, *S;     /* [Wc][ngroups][d][d] eigvec-major, embedded in the full space (Ut[k][i] = U[i][k], zero
                         * outside the group and for k >= group size); [Wc][ngroups][d] */
    double *DE;         /* [Wc][de_size][d] */
    double *AM;         /* [W][cov_update][d] */
    uint64_t *nacc;     /* [W][ntemps]      by RANK */
    uint64_t *jstat;    /* [W][ntemps][J_NTYPES][2]  (proposed, accepted) by RANK */
} orc_state;

typedef struct {
    const uint8_t *kinds; const double *vals; const int64_t *bounds;
    int64_t n, pos, err;
} orc_replay;

static double rp_next(orc_replay *rp, int kind, int64_t bound)
{
    while (rp->pos < rp->n && rp->kinds[rp->pos] == K_SHUF) rp->pos++;   /* dead randomizedPropCycle draw, PT:1042 */
    if (rp->pos >= rp->n) { rp->err |= 1; return 0.0; }
    if (rp->kinds[rp->pos] != kind) rp->err |= 2;
    if (kind == K_INT && rp->bounds[rp->pos] != bound) rp->err |= 4;
    return rp->vals[rp->pos++];
}

/* G-lane strided partial dot + xor-butterfly; this is the order the kernels use */
static double lane_dot(const double *a, const double *b, int d, int G)
{
    double p[64];
    for (int l = 0; l < G; ++l) {
        double acc = 0.0;
        for (int e = l; e < d; e += G) acc = fma(a[e], b[e], acc);
        p[l] = acc;
    }
    for (int m = G >> 1; m >= 1; m >>= 1) {
        double t[64];
        for (int l = 0; l < G; ++l) t[l] = p[l] + p[l ^ m];
        memcpy(p, t, sizeof(double) * G);
    }
    return p[0];
}

static double eval_logp(const orc_cfg *c, const double *q)
{
    if (c->logp_kind == LOGP_BOX) {          /* tests/test_simple.py:36-41 of the reference */
        const double *lo = c->logp_par, *hi = c->logp_par + c->ndim;
        for (int i = 0; i < c->ndim; ++i)
            if (!(lo[i] <= q[i]) || !(hi[i] >= q[i])) return -INFINITY;
        return 0.0;
    }
    return 0.0;
}

static double eval_logl(const orc_cfg *c, const double *q, double *tmp /* 2d */)
{
    int d = c->ndim;
    if (c->logl_kind == LOGL_ISO)
        return -0.5 * lane_dot(q, q, d, c->lanes);
    if (c->logl_kind == LOGL_DENSE) {        /* -(x-mu)^T P (x-mu) / 2 */
        const double *mu = c->logl_par, *Pt = c->logl_par + d;
        double *r = tmp, *v = tmp + d;
        for (int i = 0; i < d; ++i) { r[i] = q[i] - mu[i]; v[i] = 0.0; }
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) v[i] = fma(Pt[(size_t)j * d + i], r[j], v[i]);
        return -0.5 * lane_dot(r, v, d, c->lanes);
    }
    if (c->logl_kind == LOGL_CURVED) {
        /* d/2 independent copies of the 2-d curved likelihood of the reference's
         * examples/curved_likelihood.ipynb (cell 2, lnlikefn): log(exp(l0) + 0.5 exp(l1)) */
        double *t = tmp, *o = tmp + d;
        for (int i = 0; i < d; ++i) { t[i] = 0.0; o[i] = 1.0; }
        for (int i = 0; i + 1 < d; i += 2) {
            const double x = q[i], y = q[i + 1], x2 = x * x;
            const double g = 9.0 + 4.0 * x2 + 9.0 * y;
            const double l0 = -x2 - g * g;
            const double ym = y - 2.0;
            const double l1 = -8.0 * x2 - 8.0 * (ym * ym);
            t[i] = orc_log(orc_exp(l0) + 0.5 * orc_exp(l1));
        }
        return lane_dot(t, o, d, c->lanes);
    }
    return NAN;
}

/* ------------------------------------------------------------ MH steps */
/* One Metropolis-Hastings update of one chain: PT:601-622 with _jump PT:1048-1067,
 * SCAM PT:820-876, AM PT:879-933, DE PT:936-985. */
static void mh_one(const orc_cfg *c, orc_state *st, int w, int s, int64_t it, orc_replay *rp, double *buf)
{
    const int d = c->ndim, nt = c->ntemps;
    const size_t ch = (size_t)w * nt + s;
    const int t = st->temp_of[ch];
    double *x = st->X + ch * d;
    double *q = buf, *tmp = buf + d;            /* tmp: 2d for logl + d for AM weights */
    double *wk = buf + 3 * d;
    const double temp = c->temps_mh[t], beta = c->beta[t];
    const size_t wc = c->cov_per_walker ? (size_t)w : 0;
    const int ngr = c->ngroups > 1 ? c->ngroups : 1;
    const uint32_t sid = (uint32_t)((uint64_t)(c->walker0 + w) * (uint32_t)c->ntemps_global + (uint32_t)(c->temp0 + t));
    orc_replay *r = rp ? rp + t : NULL;
    uint64_t A[2], B[2], C[2], D[2];
    if (!r) { philox_words(c->seed, (uint64_t)it, sid, SLOT_A, A); philox_words(c->seed, (uint64_t)it, sid, SLOT_B, B); }

    /* pick from the weighted cycle (PT:1058) */
    const int L = c->w_scam + c->w_am + (c->de_on ? c->w_de : 0);
    const int ind = r ? (int)rp_next(r, K_INT, L) : (int)w2index(A[0], (uint64_t)L);
    const int jt = ind < c->w_scam ? J_SCAM : (ind < c->w_scam + c->w_am ? J_AM : J_DE);

    /* group pick (PT:839,897,955); counter mode: word C0 for SCAM / AM, D0 for DE (the words those jumps leave unused) */
    int g = 0;
    if (r) g = (int)rp_next(r, K_INT, ngr);
    else if (ngr > 1) {
        if (jt == J_DE) { philox_words(c->seed, (uint64_t)it, sid, SLOT_D, D); g = (int)w2index(D[0], (uint64_t)ngr); }
        else { philox_words(c->seed, (uint64_t)it, sid, SLOT_C, C); g = (int)w2index(C[0], (uint64_t)ngr); }
    }
    const int ng = (c->ngroups > 1) ? c->gsize[g] : d;
    const double *Ut = st->Ut + (wc * ngr + g) * (size_t)d * d, *S = st->S + (wc * ngr + g) * (size_t)d;
    const double *gm = (c->ngroups > 1) ? c->gmask + (size_t)g * d : NULL;

    if (jt == J_SCAM || jt == J_AM) {
        const double prob = r ? rp_next(r, K_UNI, 0) : w2uniform(A[1]);
        double scale = prob > 0.97 ? 10.0 : (prob > 0.9 ? 0.2 : 1.0);
        if (temp <= 100.0) scale *= sqrt(temp);                         /* PT:861-862 */
        if (jt == J_SCAM) {
            int k;
            double z;
            if (r) { k = (int)rp_next(r, K_INT, ng); z = rp_next(r, K_NRM, 0); }
            else {
                philox_words(c->seed, (uint64_t)it, sid, SLOT_D, D);
                k = (int)w2index(B[1], (uint64_t)ng);
                z = orc_normal(D[0], D[1]);
            }
            const double cd = 2.4 / sqrt(2.0 * 1.0) * scale;            /* PT:870, neff = 1 */
            const double a = z * cd * sqrt(S[k]);                       /* PT:873 */
            for (int i = 0; i < d; ++i) q[i] = x[i] + a * Ut[(size_t)k * d + i];
        } else {
            const double cd = 2.4 / sqrt(2.0 * (double)ng) * scale;     /* PT:928 */
            for (int k = 0; k < ng; ++k) {
                double z;
                if (r) z = rp_next(r, K_NRM, 0);
                else {
                    /* directions k and k + lanes share one Philox call (cos and sin branches of one Box-Muller) */
                    const int which = (k / c->lanes) & 1, base = which ? k - c->lanes : k;
                    uint64_t E[2];
                    philox_words(c->seed, (uint64_t)it, sid, SLOT_AM + (uint32_t)base, E);
                    z = which ? orc_normal_sin(E[0], E[1]) : orc_normal(E[0], E[1]);
                }
                wk[k] = z * cd * sqrt(S[k]);                            /* PT:930 */
            }
            /* q = x + U (cd sqrt(S) z): PT:923-931 up to rounding (U orthogonal) */
            for (int i = 0; i < d; ++i) tmp[i] = 0.0;
            for (int k = 0; k < ng; ++k)
                for (int i = 0; i < d; ++i) tmp[i] = fma(Ut[(size_t)k * d + i], wk[k], tmp[i]);
            for (int i = 0; i < d; ++i) q[i] = x[i] + tmp[i];
        }
    } else {
        const int Bn = c->de_size;
        int mm, nn;
        double prob, scale;
        if (r) {
            mm = (int)rp_next(r, K_INT, Bn); nn = (int)rp_next(r, K_INT, Bn);
            while (mm == nn) nn = (int)rp_next(r, K_INT, Bn);           /* PT:965-966 */
            prob = rp_next(r, K_UNI, 0);
        } else {
            philox_words(c->seed, (uint64_t)it, sid, SLOT_C, C);
            mm = (int)w2index(B[1], (uint64_t)Bn);
            nn = (int)(((uint64_t)mm + 1 + w2index(C[0], (uint64_t)(Bn - 1))) % (uint64_t)Bn);
            prob = w2uniform(A[1]);
        }
        if (prob > 0.5) scale = 1.0;
        else {
            double rr;
            if (r) rr = rp_next(r, K_UNI, 0);
            else rr = w2uniform(C[1]);
            scale = rr * 2.4 / sqrt(2.0 * (double)ng) * sqrt(1.0 / beta); /* PT:976 */
        }
        const double *DE = st->DE + wc * (size_t)Bn * d;
        for (int i = 0; i < d; ++i)
            q[i] = (!gm || gm[i] != 0.0) ? x[i] + scale * (DE[(size_t)mm * d + i] - DE[(size_t)nn * d + i]) : x[i] + 0.0;
    }
    st->jstat[(((size_t)w * nt + t) * J_NTYPES + jt) * 2 + 0] += 1;

    /* prior, likelihood, tempering (PT:605-612) */
    const double lp = eval_logp(c, q);
    double newlnL = 0.0, newlnprob;
    if (lp == -INFINITY) newlnprob = -INFINITY;
    else { newlnL = eval_logl(c, q, tmp); newlnprob = beta * newlnL + lp; }

    /* Hastings test (PT:615-622); lnprob0 is always 1/T*lnL + logp(x) of the held state */
    const double lnprob0 = beta * st->lnL[ch] + st->lp[ch];
    const double u = r ? rp_next(r, K_UNI, 0) : w2uniform(B[0]);
    const double diff = newlnprob - lnprob0 + 0.0;
    if (diff > orc_log(u)) {
        memcpy(x, q, sizeof(double) * d);
        st->lnL[ch] = newlnL; st->lp[ch] = lp;
        st->nacc[(size_t)w * nt + t] += 1;
        st->jstat[(((size_t)w * nt + t) * J_NTYPES + jt) * 2 + 1] += 1;
    }

    /* AM buffer (PT:327-328): the rank-0 chain, unless a swap follows this iteration
     * (the reference stores the post-swap state, PT:624-627; orc_swap then writes it) */
    if (c->temp0 + t == 0 && st->AM) {
        const int swap_follows = c->tskip > 0 && c->ntemps_global > 1 && it % c->tskip == 0;
        if (!swap_follows)
            memcpy(st->AM + ((size_t)w * c->cov_update + (size_t)(it % c->cov_update)) * d, x, sizeof(double) * d);
    }
}

ORC_API int orc_mh_steps(const orc_cfg *c, orc_state *st, int64_t iter0, int nsteps, orc_replay *rp)
{
    double *buf = (double *)malloc(sizeof(double) * 4 * (size_t)c->ndim);
    for (int k = 0; k < nsteps; ++k)
        for (int w = 0; w < c->nwalkers; ++w)
            for (int s = 0; s < c->ntemps; ++s) mh_one(c, st, w, s, iter0 + k, rp, buf);
    free(buf);
    int64_t err = 0;
    if (rp) for (int t = 0; t < c->ntemps; ++t) err |= rp[t].err;
    return (int)err;
}

/* ---------------------------------------------------------------- swap */
/* PT:631-697.  lnL_pos[w][n]: likelihood by temperature POSITION (all ranks of the
 * ladder); map[w][j] = position whose state moves to position j; acc[w][k] += 1 for an
 * accepted pair (k,k+1), credited to the lower rank (PT:681). Hot -> cold, carried map. */
ORC_API int orc_swap_sweep(int nwalkers, int n, const double *ladder, const double *lnL_pos, int64_t iter,
                           uint64_t seed, int walker0, int32_t *map, uint64_t *acc, orc_replay *rp0)
{
    for (int w = 0; w < nwalkers; ++w) {
        const double *L = lnL_pos + (size_t)w * n;
        int32_t *m = map + (size_t)w * n;
        for (int j = 0; j < n; ++j) m[j] = j;
        const uint32_t sid = (uint32_t)((uint64_t)(walker0 + w) * (uint32_t)n + 0u);
        for (int k = n - 2; k >= 0; --k) {
            double u;
            if (rp0) u = rp_next(rp0, K_UNI, 0);
            else { uint64_t W[2]; philox_words(seed, (uint64_t)iter, sid, SLOT_SWAP + (uint32_t)k, W); u = w2uniform(W[0]); }
            double la = -L[m[k]] / ladder[k];
            la += -L[m[k + 1]] / ladder[k + 1];
            la += L[m[k + 1]] / ladder[k];
            la += L[m[k]] / ladder[k + 1];
            if (u <= orc_exp(la)) {
                int32_t tt = m[k]; m[k] = m[k + 1]; m[k + 1] = tt;
                acc[(size_t)w * n + k] += 1;
            }
        }
    }
    return rp0 ? (int)rp0->err : 0;
}

/* Odd/even swap mode of the engine (include/ptmi.h, PTMI_SWAP_ODDEVEN; not in the reference): only the disjoint
 * pairs (k, k+1), k = parity (mod 2), are tried, each with the pair test of PT:672-679 and the uniform the sweep
 * would have used for pair k.  Same outputs as orc_swap_sweep. */
ORC_API void orc_swap_oddeven(int nwalkers, int n, const double *ladder, const double *lnL_pos, int64_t iter,
                              uint64_t seed, int walker0, int parity, int32_t *map, uint64_t *acc)
{
    for (int w = 0; w < nwalkers; ++w) {
        const double *L = lnL_pos + (size_t)w * n;
        int32_t *m = map + (size_t)w * n;
        for (int j = 0; j < n; ++j) m[j] = j;
        const uint32_t sid = (uint32_t)((uint64_t)(walker0 + w) * (uint32_t)n + 0u);
        for (int k = parity; k + 1 < n; k += 2) {
            uint64_t W[2];
            philox_words(seed, (uint64_t)iter, sid, SLOT_SWAP + (uint32_t)k, W);
            double la = -L[k] / ladder[k];
            la += -L[k + 1] / ladder[k + 1];
            la += L[k + 1] / ladder[k];
            la += L[k] / ladder[k + 1];
            if (w2uniform(W[0]) <= orc_exp(la)) {
                m[k] = k + 1;
                m[k + 1] = k;
                acc[(size_t)w * n + k] += 1;
            }
        }
    }
}

/* single-process application of a sweep to the slot tables (all ranks local) */
ORC_API void orc_swap_apply(const orc_cfg *c, orc_state *st, const int32_t *map, int64_t iter)
{
    const int nt = c->ntemps, d = c->ndim;
    int32_t *ns = (int32_t *)malloc(sizeof(int32_t) * nt);
    for (int w = 0; w < c->nwalkers; ++w) {
        int32_t *so = st->slot_of + (size_t)w * nt, *to = st->temp_of + (size_t)w * nt;
        for (int j = 0; j < nt; ++j) ns[j] = so[map[(size_t)w * nt + j]];
        for (int j = 0; j < nt; ++j) { so[j] = ns[j]; to[ns[j]] = j; }
        if (st->AM && c->temp0 == 0)
            memcpy(st->AM + ((size_t)w * c->cov_update + (size_t)(iter % c->cov_update)) * d,
                   st->X + ((size_t)w * nt + so[0]) * d, sizeof(double) * d);
    }
    free(ns);
}

/* ------------------------------------------------------------- Welford */
/* PT:769-794 for one walker: mem buffered rows in buffer order.  fused = 0 is the
 * reference's arithmetic (one product, one sum); fused = 1 accumulates with one fma per
 * element and advances the mean by diff * (1/it) (the pooled mode of the engine, which is not a
 * replica of a reference run). */
ORC_API void orc_welford2(int d, int mem, int64_t iter, const double *AM, double *mu, double *M2, double *cov, int fused)
{
    int64_t it = iter - mem;
    if (it == 0) { memset(M2, 0, sizeof(double) * d * d); memset(mu, 0, sizeof(double) * d); }
    double *diff = (double *)malloc(sizeof(double) * 2 * d), *e = diff + d;
    for (int ii = 0; ii < mem; ++ii) {
        it += 1;
        const double *row = AM + (size_t)ii * d;
        const double rinv = 1.0 / (double)it;              /* fused variant: one reciprocal per row, then a product */
        for (int j = 0; j < d; ++j) {
            diff[j] = row[j] - mu[j];
            if (fused) mu[j] = mu[j] + diff[j] * rinv;
            else mu[j] += diff[j] / (double)it;
        }
        for (int j = 0; j < d; ++j) e[j] = row[j] - mu[j];
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                if (fused) M2[(size_t)i * d + j] = fma(diff[i], e[j], M2[(size_t)i * d + j]);
                else M2[(size_t)i * d + j] += diff[i] * e[j];
            }
    }
    if (cov) for (int i = 0; i < d * d; ++i) cov[i] = M2[i] / (double)(it - 1);
    free(diff);
}
ORC_API void orc_welford(int d, int mem, int64_t iter, const double *AM, double *mu, double *M2, double *cov)
{
    orc_welford2(d, mem, iter, AM, mu, M2, cov, 0);
}

/* Chan et al. combination of `nin` partial statistics, inputs ascending; input k holds nb
 * samples (the last one nb_last). */
static void chan_combine(int d, int nin, double nb, double nb_last, const double *mu, const double *M2,
                         double *mu_out, double *M2_out)
{
    double *m = (double *)calloc((size_t)d, sizeof(double));
    double *M = (double *)calloc((size_t)d * d, sizeof(double));
    double na = 0.0;
    for (int k = 0; k < nin; ++k) {
        const double *mw = mu + (size_t)k * d, *Mw = M2 + (size_t)k * d * d;
        const double nk = k == nin - 1 ? nb_last : nb, nn = na + nk;
        const double f = na * nk / nn, g = nk / nn;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                const double di = mw[i] - m[i], dj = mw[j] - m[j];
                M[(size_t)i * d + j] = (M[(size_t)i * d + j] + Mw[(size_t)i * d + j]) + (di * dj) * f;
            }
        for (int i = 0; i < d; ++i) m[i] = m[i] + (mw[i] - m[i]) * g;
        na = nn;
    }
    memcpy(mu_out, m, sizeof(double) * d);
    memcpy(M2_out, M, sizeof(double) * d * d);
    free(m); free(M);
}

/* pooled covariance over walkers: groups of 64 walkers combined in walker order, then the
 * groups combined in order (the two-level order of the engine's pool kernels) */
ORC_API void orc_pool_cov(int d, int nwalkers, int64_t n_per, const double *mu, const double *M2,
                          double *mu_out, double *cov_out)
{
    const int GS = 64, ng = (nwalkers + GS - 1) / GS;
    double *gm = (double *)malloc(sizeof(double) * (size_t)ng * d), *gM = (double *)malloc(sizeof(double) * (size_t)ng * d * d);
    for (int g = 0; g < ng; ++g) {
        const int w0 = g * GS, cnt = (w0 + GS <= nwalkers) ? GS : nwalkers - w0;
        chan_combine(d, cnt, (double)n_per, (double)n_per, mu + (size_t)w0 * d, M2 + (size_t)w0 * d * d, gm + (size_t)g * d, gM + (size_t)g * d * d);
    }
    const int last = nwalkers - (ng - 1) * GS;
    double *M = (double *)malloc(sizeof(double) * (size_t)d * d);
    chan_combine(d, ng, (double)GS * (double)n_per, (double)last * (double)n_per, gm, gM, mu_out, M);
    const double den = (double)nwalkers * (double)n_per - 1.0;
    for (int i = 0; i < d * d; ++i) cov_out[i] = M[i] / den;
    free(gm); free(gM); free(M);
}

/* ----------------------------------------------------------- DE buffer */
/* PT:806-817 + shift_array PT:27-37: drop the oldest mem rows, append the AM buffer. */
ORC_API void orc_de_update(int d, int de_size, int mem, double *DE, const double *AM)
{
    if (mem >= de_size) { memcpy(DE, AM + (size_t)(mem - de_size) * d, sizeof(double) * (size_t)de_size * d); return; }
    memmove(DE, DE + (size_t)mem * d, sizeof(double) * (size_t)(de_size - mem) * d);
    memcpy(DE + (size_t)(de_size - mem) * d, AM, sizeof(double) * (size_t)mem * d);
}

/* pooled variant: new row r comes from walker (r mod W)'s AM row r */
ORC_API void orc_de_update_pooled(int d, int de_size, int mem, int nwalkers, double *DE, const double *AM)
{
    double *rows = (double *)malloc(sizeof(double) * (size_t)mem * d);
    for (int r = 0; r < mem; ++r)
        memcpy(rows + (size_t)r * d, AM + ((size_t)(r % nwalkers) * mem + r) * d, sizeof(double) * d);
    orc_de_update(d, de_size, mem, DE, rows);
    free(rows);
}

/* initial lnL / lp of every row (PT:479-487) */
ORC_API void orc_eval_state(const orc_cfg *c, orc_state *st)
{
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)c->ndim);
    for (size_t ch = 0; ch < (size_t)c->nwalkers * c->ntemps; ++ch) {
        const double *x = st->X + ch * c->ndim;
        const double lp = eval_logp(c, x);
        st->lp[ch] = lp;
        st->lnL[ch] = lp == -INFINITY ? -INFINITY : eval_logl(c, x, tmp);
    }
    free(tmp);
}

ORC_API double orc_logl(const orc_cfg *c, const double *x)
{
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)c->ndim);
    double v = eval_logl(c, x, tmp);
    free(tmp);
    return v;
}

ORC_API int orc_sizeof_cfg(void) { return (int)sizeof(orc_cfg); }
