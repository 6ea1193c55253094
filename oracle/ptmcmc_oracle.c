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
#include "orc_tables.h"

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

/* slot numbers of the counter-based schedule (DESIGN.md "RNG schedule").  One iteration of one chain consumes two
 * Philox calls of its rank's stream, every bit of which has a use:
 *   slot 0 (P): w0 = [ hi32: cycle pick | lo32: scale-branch uniform ],  w1 = accept uniform
 *   slot 1 (Q): SCAM: w0 = Box-Muller u1, w1 = [ hi32: eigen-direction | lo32: Box-Muller angle ]
 *               DE:   w0 = [ hi32: row mm | lo32: offset of row nn ],     w1 = DE scale uniform
 *   slot 2 (G): w0 hi32 = parameter group (only drawn with more than one group)
 *   SWAP+k: w0 uniform of pair (k,k+1), stream of rank 0
 *   AM+k: (w0,w1) Box-Muller pair: cos branch -> eigen-direction k, sin branch -> direction k+lanes, (k/lanes) even
 * 64-bit words give 53-bit uniforms, 32-bit halves give 32-bit uniforms h * 2^-32 and indices (h * n) >> 32.
 * pick_mode WALKER takes the cycle pick from slot 0 of the stream of the walker's rank 0 instead of the chain's own. */
enum { SLOT_P = 0, SLOT_Q = 1, SLOT_G = 2, SLOT_SWAP = 0x10000, SLOT_AM = 0x1000000,
       SLOT_GJ = 0x2000000 /* + 4096 * (momenta draw of the call) + direction */, SLOT_GJS = 0x3000000 /* + scalar draw of the call */ };
enum { PICK_CHAIN = 0, PICK_WALKER = 1 };

static inline uint32_t hi32(uint64_t w) { return (uint32_t)(w >> 32); }
static inline uint32_t lo32(uint64_t w) { return (uint32_t)w; }
static inline double h2uniform(uint32_t h) { return (double)h * 0x1.0p-32; }                     /* [0,1), 32 bits */
static inline uint32_t h2index(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * n) >> 32); }
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

/* n logarithms at once (tests/test_swap_logspace.py walks 10^7 near-ties) */
ORC_API void orc_log_n(int64_t n, const double *x, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_log(x[i]);
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

/* ------------------------------------------ the draws of the MH path (table driven; ptmi_device.h unit_log / unit_sincos)
 * ln u of the (0,1] uniform u = ((w >> 11) + 1) 2^-53: x = (double)n = z 2^k, z in [0.6953125, 1.390625) cut into 32 slices by
 * bit pattern, r = z invc - 1, ln = k ln2 + logc + log1p(r) with log1p by its Taylor polynomial to r^9. */
ORC_API double orc_unit_log(uint64_t w)
{
    const uint64_t n = (w >> 11) + 1;
    const double x = (double)n;                                        /* exact: n <= 2^53 */
    const uint64_t xb = d2u(x);
    const uint32_t hi = (uint32_t)(xb >> 32), tmp = hi - 0x3FE64000u;
    const int k = (int)((int32_t)tmp >> 20) - 53;
    const uint32_t i = (tmp >> 15) & 31u;
    const double z = u2d(((uint64_t)(hi - (tmp & 0xFFF00000u)) << 32) | (uint32_t)xb);
    const double invc = ORC_DRAWT[2 * i], logc = ORC_DRAWT[2 * i + 1];
    const double r = fma(z, invc, -1.0);
    double p = 0x1.c71c71c71c71cp-4;
    p = fma(p, r, -0x1.0p-3);
    p = fma(p, r, 0x1.2492492492492p-3);
    p = fma(p, r, -0x1.5555555555555p-3);
    p = fma(p, r, 0x1.999999999999ap-3);
    p = fma(p, r, -0x1.0p-2);
    p = fma(p, r, 0x1.5555555555555p-2);
    p = fma(p, r, -0x1.0p-1);
    const double l1 = fma(r * r, p, r);
    return fma((double)k, 0x1.62e42fefa39efp-1, logc) + l1;
}
/* cos and sin of 2 pi (j + 1/2 + t) / 32: the base angle's pair from the table, rotated by beta = 2 pi t / 32 */
static void unit_sincos(uint32_t j, double t, double *sn, double *cs)
{
    const double bc = ORC_DRAWT[64 + 2 * j], bs = ORC_DRAWT[64 + 2 * j + 1];
    const double be = t * 0x1.921fb54442d18p-3, zz = be * be;
    double ps = 0x1.71de3a556c734p-19;
    ps = fma(ps, zz, -0x1.a01a01a01a01ap-13);
    ps = fma(ps, zz, 0x1.1111111111111p-7);
    ps = fma(ps, zz, -0x1.5555555555555p-3);
    const double sb = fma(be * zz, ps, be);
    double pc = 0x1.a01a01a01a01ap-16;
    pc = fma(pc, zz, -0x1.6c16c16c16c17p-10);
    pc = fma(pc, zz, 0x1.5555555555555p-5);
    pc = fma(pc, zz, -0x1.0p-1);
    const double cb = fma(zz, pc, 1.0);
    *cs = fma(-bs, sb, bc * cb);
    *sn = fma(bc, sb, bs * cb);
}
/* the angle of a 64-bit word (j = its top 5 bits, t in [-1/2, 1/2) from the 52 bits below) ... */
ORC_API void orc_unit_sincos64(uint64_t w, double *sn, double *cs)
{
    const double t = u2d(((w >> 7) & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull) - 1.5;
    unit_sincos((uint32_t)(w >> 59), t, sn, cs);
}
/* ... and of a 32-bit half (j = top 5 bits, t from the 27 bits below) */
ORC_API void orc_unit_sincos32(uint32_t h, double *sn, double *cs)
{
    const uint32_t f = h & 0x07FFFFFFu;
    const double t = u2d(((uint64_t)(0x3FF00000u | (f >> 7)) << 32) | (uint64_t)(uint32_t)(f << 25)) - 1.5;
    unit_sincos(h >> 27, t, sn, cs);
}
/* the Box-Muller pair of two words (AM): cos branch, sin branch */
ORC_API void orc_unit_normals(uint64_t w0, uint64_t w1, double *zc, double *zs)
{
    const double r = sqrt(-2.0 * orc_unit_log(w0));
    double sn, cs;
    orc_unit_sincos64(w1, &sn, &cs);
    *zc = r * cs;
    *zs = r * sn;
}
/* the SCAM normal: radius from a word, 32-bit angle */
ORC_API double orc_unit_normal32(uint64_t w0, uint32_t h)
{
    double sn, cs;
    orc_unit_sincos32(h, &sn, &cs);
    return sqrt(-2.0 * orc_unit_log(w0)) * cs;
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
enum { LOGL_ISO = 0, LOGL_DENSE = 1, LOGL_CURVED = 2, LOGL_INTERVAL = 3 };
enum { LOGP_FLAT = 0, LOGP_BOX = 1 };
enum { J_SCAM = 0, J_AM = 1, J_DE = 2, J_NUTS = 3, J_HMC = 4, J_NTYPES = 5 };
enum { K_INT = 0, K_UNI = 1, K_NRM = 2, K_SHUF = 3, K_EXP = 4 };
/* per-rank state of the gradient jumps (the attributes of a rank's NUTSJump / HMCJump object, NJ:379-433) */
enum { GJ_EPS = 0, GJ_MU = 1, GJ_HBAR = 2, GJ_EPSBAR = 3, GJ_NITER = 4, GJ_HITER = 5, GJ_HAVE_EPS = 6, GJ_NLEAP = 7 /* leapfrogs so far */, GJ_NSTATE = 8 };

typedef struct {
    int32_t ndim, ntemps, nwalkers, lanes;
    int32_t logl_kind, logp_kind;
    int32_t w_scam, w_am, w_de;       /* cycle = [SCAM]*w_scam + [AM]*w_am + [DE]*w_de  (PT:261-264, 579-585) */
    int32_t de_on, de_size;           /* DE in the cycle; rows in a DE buffer (= burn, PT:221) */
    int32_t cov_update, tskip;        /* AM-buffer length (PT:220); swap period (PT:624), 0 = never */
    int32_t cov_per_walker;           /* 1: Ut/S/DE per walker, 0: one shared set */
    int32_t ntemps_global, temp0, walker0, ngroups;   /* ngroups: parameter groups (PT:129-145); 0 or 1 = one full group */
    uint64_t seed;
    const double *logl_par;           /* DENSE: mu[d], Pt[d*d] (Pt[j*d+i] = P[i][j]), Tl[d*d] (the half of the symmetric P the value is summed over:
                                       * Tl[k*d+i] = P[k][i] for k > i, P[i][i] / 2 for k == i, 0 for k < i) */
    const double *logp_par;           /* BOX: lo[d], hi[d] */
    const double *temps_mh;           /* [ntemps] temperature of each local rank as the MH step sees it (PT:278-282) */
    const double *beta;               /* [ntemps] 1/temps_mh */
    const int32_t *gsize;             /* [ngroups] parameters per group (NULL with one full group) */
    const double *gmask;              /* [ngroups][d] 1 where the parameter belongs to the group */
    /* gradient jumps on the built-in likelihoods (PT:225-258): cycle += [NUTS]*w_nuts + [HMC]*w_hmc */
    int32_t w_nuts, w_hmc;
    int32_t gj_nburn;                 /* nburn of the jump objects (= burn, PT:227,238,251) */
    int32_t hmc_min, hmc_max;         /* HMC: randint(hmc_min, hmc_max) leapfrogs (PT:240-241: 2, HMCsteps) */
    int32_t nuts_maxdepth;            /* tree heights built per call are 0..nuts_maxdepth (the reference has no cap) */
    int32_t pick_mode;                /* PICK_CHAIN (the reference: every rank draws its own cycle entry) or PICK_WALKER */
    double hmc_eps;                   /* HMCstepsize */
    double nuts_delta;                /* target acceptance of the dual averaging (0.6, PT:256) */
    const double *gj_tab;             /* [3][d][d] whitening tables from L = cholesky(cov0) (NJ:53-54), each used as
                                       * out[i] = sum_k T[k][i] v[k]:  backward T[k][i] = L[k][i] (x = L^T q),
                                       * forward T[k][i] = Linv[k][i] (q = Linv^T x), gradient T[k][i] = L[i][k] */
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
    double *gj;         /* [W][ntemps][GJ_NSTATE]    gradient-jump state by RANK (NULL without gradient jumps) */
    uint64_t *AMflag;   /* [W][cov_update]  the engine's AM row flags (include/ptmi.h AMflag), or NULL: bit 0 NEW = the step was
                         * accepted, bit 1 KEY = first step of a launch, ring rows 0 and 1, the swap's row.  The oracle stores every
                         * row whatever the flags say; they only weight the pooled statistics (orc_pool_update_rle) */
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

/* LOGL_INTERVAL, one element: the reference's own NUTS workload, tests/test_nuts.py -- GaussianLikelihood.lnlikefn_grad (:22-25:
 * -x^2/2 - log(2 pi)/2 per element, gradient -x) seen through intervalTransform (:50-140): backward (:81-86) x = (b - a) e^p / (1 + e^p) + a,
 * logjacobian_grad (:88-94) log(b - a) + p - 2 log(1 + e^p) with gradient (1 - e^p) / (1 + e^p), dxdp (:96-101) (b - a) e^p / (1 + e^p)^2,
 * lnlikefn_grad (:117-122) ll + lj, ll_grad * dxdp + lj_grad -- in the reference's operation order.  w = b - a and lw = log w are
 * parameters (par = a | w | lw).  Returns the value term; *g (if not NULL) its derivative. */
static double interval_elem(double p, double lo, double w, double lw, double *g)
{
    const double E = orc_exp(p), onepe = 1.0 + E;
    const double wE = w * E;
    const double x = wE / onepe + lo;
    const double t = (-0.5 * (x * x) - 0x1.d67f1c864beb5p-1) + ((lw + p) - 2.0 * orc_log(onepe));
    if (g) {
        const double dxdp = wE / (onepe * onepe);
        *g = (-x) * dxdp + (1.0 - E) / onepe;
    }
    return t;
}

static double eval_logl(const orc_cfg *c, const double *q, double *tmp /* 2d */)
{
    int d = c->ndim;
    if (c->logl_kind == LOGL_ISO)
        return -0.5 * lane_dot(q, q, d, c->lanes);
    if (c->logl_kind == LOGL_DENSE) {
        /* -(x-mu)^T P (x-mu) / 2 with P symmetric = -sum_i r_i (P_ii r_i / 2 + sum_{k > i} P_ki r_k): half the products of the
         * full form.  v_i is the k-ascending fma chain over column i of Tl (its zeros leave the sum untouched). */
        const double *mu = c->logl_par, *Tl = c->logl_par + d + (size_t)d * d;
        double *r = tmp, *v = tmp + d;
        for (int i = 0; i < d; ++i) { r[i] = q[i] - mu[i]; v[i] = 0.0; }
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) v[i] = fma(Tl[(size_t)j * d + i], r[j], v[i]);
        return -lane_dot(r, v, d, c->lanes);
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
    if (c->logl_kind == LOGL_INTERVAL) {
        const double *par = c->logl_par;
        double *t = tmp, *o = tmp + d;
        for (int i = 0; i < d; ++i) { t[i] = interval_elem(q[i], par[i], par[d + i], par[2 * d + i], NULL); o[i] = 1.0; }
        return lane_dot(t, o, d, c->lanes);
    }
    return NAN;
}


/* ------------------------------------------------------- gradient jumps */
/* nutsjump.py of the reference on the built-in likelihoods (their gradients are analytic here; the reference takes
 * them from the user's logl_grad / logp_grad callbacks).  NJ:<lines> = PTMCMCSampler/nutsjump.py. */
typedef struct {
    orc_replay *r;          /* replay of the reference's global np.random draws, or NULL: counter mode */
    uint64_t seed, it;
    uint32_t sid, nm, ns;   /* stream; momenta draws and scalar draws used so far in this call */
    int lanes;
} gj_rng;

static void gj_momenta(gj_rng *g, int d, double *r)             /* NJ:92-94  np.random.randn(ndim) */
{
    if (g->r) { for (int i = 0; i < d; ++i) r[i] = rp_next(g->r, K_NRM, 0); return; }
    const uint32_t block = g->nm++;
    for (int k = 0; k < d; ++k) {                               /* paired like the AM normals */
        const int which = (k / g->lanes) & 1, base = which ? k - g->lanes : k;
        uint64_t E[2];
        philox_words(g->seed, g->it, g->sid, SLOT_GJ + 4096u * block + (uint32_t)base, E);
        r[k] = which ? orc_normal_sin(E[0], E[1]) : orc_normal(E[0], E[1]);
    }
}
static double gj_uniform(gj_rng *g)
{
    if (g->r) return rp_next(g->r, K_UNI, 0);
    uint64_t W[2];
    philox_words(g->seed, g->it, g->sid, SLOT_GJS + g->ns++, W);
    return w2uniform(W[0]);
}
static double gj_exponential(gj_rng *g)
{
    if (g->r) return rp_next(g->r, K_EXP, 0);
    uint64_t W[2];
    philox_words(g->seed, g->it, g->sid, SLOT_GJS + g->ns++, W);
    return -orc_log(w2uniform_open(W[0]));
}
static int gj_randint(gj_rng *g, int lo, int hi)                /* np.random.randint(lo, hi) */
{
    if (g->r) return (int)rp_next(g->r, K_INT, hi);
    uint64_t W[2];
    philox_words(g->seed, g->it, g->sid, SLOT_GJS + g->ns++, W);
    return lo + (int)w2index(W[0], (uint64_t)(hi - lo));
}

/* A whitening table that is DIAGONAL (the jump objects were built from a diagonal covariance: L = cholesky(cov), NJ:53-54 --
 * the reference's np.dot then multiplies by zeros) is applied as d products, out[i] = T[i][i] v[i]: the engine's definition for
 * such tables (libptmi decides at ptmi_create; a whitening product is a quarter of a NUTS call on the device).  Equal to the full
 * sum unless a zero meets an infinity or the result is a negative zero. */
static int tab_is_diag(const double *T, int d)
{
    for (int k = 0; k < d; ++k)
        for (int i = 0; i < d; ++i)
            if (i != k && T[(size_t)k * d + i] != 0.0) return 0;
    return 1;
}
static void tab_vec(const double *T, const double *v, double *out, int d)
{
    if (tab_is_diag(T, d)) {
        for (int i = 0; i < d; ++i) out[i] = T[(size_t)i * d + i] * v[i];
        return;
    }
    for (int i = 0; i < d; ++i) out[i] = 0.0;
    for (int k = 0; k < d; ++k)
        for (int i = 0; i < d; ++i) out[i] = fma(T[(size_t)k * d + i], v[k], out[i]);
}

/* logl and its gradient for the built-in families (same value as eval_logl) */
static double eval_logl_grad(const orc_cfg *c, const double *q, double *tmp /* 2d */, double *g)
{
    const int d = c->ndim;
    const double ll = eval_logl(c, q, tmp);
    if (c->logl_kind == LOGL_ISO) {
        for (int i = 0; i < d; ++i) g[i] = -q[i];
    } else if (c->logl_kind == LOGL_DENSE) {                   /* symmetric P: grad = -P (x - mu), the full product (tmp holds r) */
        const double *Pt = c->logl_par + d;
        for (int i = 0; i < d; ++i) tmp[d + i] = 0.0;
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) tmp[d + i] = fma(Pt[(size_t)j * d + i], tmp[j], tmp[d + i]);
        for (int i = 0; i < d; ++i) g[i] = -tmp[d + i];
    } else if (c->logl_kind == LOGL_INTERVAL) {
        const double *par = c->logl_par;
        for (int i = 0; i < d; ++i) (void)interval_elem(q[i], par[i], par[d + i], par[2 * d + i], &g[i]);
    } else {
        for (int i = 0; i < d; ++i) g[i] = 0.0;
        for (int i = 0; i + 1 < d; i += 2) {
            const double x = q[i], y = q[i + 1], x2 = x * x;
            const double gg = 9.0 + 4.0 * x2 + 9.0 * y;
            const double l0 = -x2 - gg * gg;
            const double ym = y - 2.0;
            const double l1 = -8.0 * x2 - 8.0 * (ym * ym);
            const double e0 = orc_exp(l0), e1 = 0.5 * orc_exp(l1);
            const double sum = e0 + e1;
            const double d0x = -2.0 * x - 16.0 * gg * x, d0y = -18.0 * gg;
            const double d1x = -16.0 * x, d1y = -16.0 * ym;
            g[i] = (e0 * d0x + e1 * d1x) / sum;
            g[i + 1] = (e0 * d0y + e1 * d1y) / sum;
        }
    }
    return ll;
}

static int64_t g_leapfrogs;     /* statistics only: leapfrogs taken since the library was loaded */
ORC_API int64_t orc_leapfrog_count(void) { return g_leapfrogs; }

typedef struct {
    const orc_cfg *c;
    double beta;
    double *w;              /* workspace: 5 d-vectors */
    int64_t nleap;          /* leapfrogs taken (statistics) */
} gj_ctx;

/* beta*logl + logp and its gradient in the whitened coordinates (NJ:71-90) */
static double func_grad_white(gj_ctx *G, const double *q, double *gradw)
{
    const orc_cfg *c = G->c;
    const int d = c->ndim;
    double *x = G->w, *g = G->w + d, *tmp = G->w + 2 * d;     /* tmp: 2d */
    tab_vec(c->gj_tab, q, x, d);                                /* backward: x = L^T q */
    const double ll = eval_logl_grad(c, x, tmp, g);
    const double lp = eval_logp(c, x);                          /* gradient of the built-in priors is zero */
    for (int i = 0; i < d; ++i) g[i] = G->beta * g[i] + 0.0;
    tab_vec(c->gj_tab + 2 * (size_t)d * d, g, gradw, d);
    return G->beta * ll + lp;
}

static double loghamiltonian(const orc_cfg *c, double logl, const double *r)     /* NJ:133-147 */
{
    return logl - 0.5 * lane_dot(r, r, c->ndim, c->lanes);
}

/* NJ:149-169: half kick, drift, gradient, half kick; outputs may not alias inputs */
static double leapfrog(gj_ctx *G, const double *theta, const double *r, const double *grad, double eps,
                       double *thetap, double *rp, double *gradp)
{
    const int d = G->c->ndim;
    const double he = 0.5 * eps;
    for (int i = 0; i < d; ++i) { rp[i] = r[i] + he * grad[i]; thetap[i] = theta[i] + eps * rp[i]; }
    const double logpp = func_grad_white(G, thetap, gradp);
    for (int i = 0; i < d; ++i) rp[i] = rp[i] + he * gradp[i];
    G->nleap++;
    g_leapfrogs++;
    return logpp;
}

static double orc_pow_neg(double x, double e) { return orc_exp(-e * orc_log(x)); }   /* x ** -e, x > 0 */

/* HMCJump.__call__ (NJ:238-291) */
static void hmc_call(gj_ctx *G, gj_rng *rng, double *st, const double *x, double *qout, double *qxy)
{
    const orc_cfg *c = G->c;
    const int d = c->ndim;
    double *v = (double *)malloc(sizeof(double) * 7 * (size_t)d);
    double *q = v, *p = v + d, *grad = v + 2 * d, *q1 = v + 3 * d, *p1 = v + 4 * d, *g1 = v + 5 * d;
    st[GJ_HITER] += 1.0;
    tab_vec(c->gj_tab + (size_t)d * d, x, q, d);                 /* forward */
    const double logp0 = func_grad_white(G, q, grad);
    gj_momenta(rng, d, p);
    const double joint0 = loghamiltonian(c, logp0, p);
    const int nsteps = gj_randint(rng, c->hmc_min, c->hmc_max);
    double joint1 = joint0;
    for (int k = 0; k < nsteps; ++k) {
        const double logp1 = leapfrog(G, q, p, grad, c->hmc_eps, q1, p1, g1);
        memcpy(q, q1, sizeof(double) * d); memcpy(p, p1, sizeof(double) * d); memcpy(grad, g1, sizeof(double) * d);
        joint1 = loghamiltonian(c, logp1, p);
        if (joint1 - 1000.0 < joint0) break;                    /* NJ:284-286 */
    }
    tab_vec(c->gj_tab, q, qout, d);
    *qxy = joint1 - joint0;
    free(v);
}

typedef struct {
    double *v;              /* 8 d-vectors: tm rm gm tp rp gp theta grad */
    double logp, alpha;
    int64_t n, nalpha, ip, im;
    int s;
} gj_tree;
#define T_TM(t) ((t)->v)
#define T_RM(t) ((t)->v + d)
#define T_GM(t) ((t)->v + 2 * d)
#define T_TP(t) ((t)->v + 3 * d)
#define T_RP(t) ((t)->v + 4 * d)
#define T_GP(t) ((t)->v + 5 * d)
#define T_TH(t) ((t)->v + 6 * d)
#define T_GR(t) ((t)->v + 7 * d)

static int stop_criterion(const orc_cfg *c, const double *tm, const double *tp, const double *rm, const double *rp, double *tmp)
{                                                               /* NJ:465-493 (force_trajlen is None) */
    const int d = c->ndim;
    for (int i = 0; i < d; ++i) tmp[i] = tp[i] - tm[i];
    const double a = lane_dot(tmp, rm, d, c->lanes), b = lane_dot(tmp, rp, d, c->lanes);
    return (a >= 0.0) & (b >= 0.0);
}

/* NJ:495-652 */
static void build_tree(gj_ctx *G, gj_rng *rng, const double *theta, const double *r, const double *grad, double logu,
                       int v, int j, double eps, double joint0, int64_t ind, gj_tree *t, double *tmp)
{
    const orc_cfg *c = G->c;
    const int d = c->ndim;
    t->v = (double *)malloc(sizeof(double) * 8 * (size_t)d);
    if (j == 0) {
        const double logpp = leapfrog(G, theta, r, grad, (double)v * eps, T_TH(t), T_RM(t), T_GR(t));
        const double joint = loghamiltonian(c, logpp, T_RM(t));
        t->n = logu < joint;
        t->s = (logu - 1000.0) < joint;
        memcpy(T_TM(t), T_TH(t), sizeof(double) * d); memcpy(T_TP(t), T_TH(t), sizeof(double) * d);
        memcpy(T_RP(t), T_RM(t), sizeof(double) * d);
        memcpy(T_GM(t), T_GR(t), sizeof(double) * d); memcpy(T_GP(t), T_GR(t), sizeof(double) * d);
        t->logp = logpp;
        const double e = orc_exp(joint - joint0);
        t->alpha = e < 1.0 ? e : 1.0;                          /* Python's min(1.0, e): 1.0 when e is NaN */
        t->nalpha = 1;
        if (v == 1) { t->ip = ind + 1; t->im = ind; } else { t->ip = ind; t->im = ind + 1; }
        return;
    }
    build_tree(G, rng, theta, r, grad, logu, v, j - 1, eps, joint0, ind, t, tmp);
    if (t->s == 1) {
        gj_tree u;
        if (v == -1) {
            build_tree(G, rng, T_TM(t), T_RM(t), T_GM(t), logu, v, j - 1, eps, joint0, t->im, &u, tmp);
            memcpy(T_TM(t), T_TM(&u), sizeof(double) * 3 * d);  /* tm, rm, gm are adjacent */
        } else {
            build_tree(G, rng, T_TP(t), T_RP(t), T_GP(t), logu, v, j - 1, eps, joint0, t->ip, &u, tmp);
            memcpy(T_TP(t), T_TP(&u), sizeof(double) * 3 * d);
        }
        t->ip = u.ip; t->im = u.im;
        const double den = (double)(t->n + u.n) > 1.0 ? (double)(t->n + u.n) : 1.0;
        if (gj_uniform(rng) < (double)u.n / den) {
            memcpy(T_TH(t), T_TH(&u), sizeof(double) * 2 * d);  /* theta, grad */
            t->logp = u.logp;
        }
        t->n += u.n;
        t->s = t->s && u.s && stop_criterion(c, T_TM(t), T_TP(t), T_RM(t), T_RP(t), tmp);
        t->alpha += u.alpha; t->nalpha += u.nalpha;
        free(u.v);
    }
}

static double accept_ratio(const orc_cfg *c, double logpp, const double *rp, double logp0, const double *r0)
{
    return orc_exp(loghamiltonian(c, logpp, rp) - loghamiltonian(c, logp0, r0));
}

/* NJ:435-463; the two loops are bounded at 100 turns here (the reference's are not) */
static double find_reasonable_epsilon(gj_ctx *G, gj_rng *rng, const double *theta0, const double *grad0, double logp0, double *v /* 4d */)
{
    const orc_cfg *c = G->c;
    const int d = c->ndim;
    double *r0 = v, *tp = v + d, *rp = v + 2 * d, *gp = v + 3 * d;
    double eps = 1.0;
    gj_momenta(rng, d, r0);
    double logpp = leapfrog(G, theta0, r0, grad0, eps, tp, rp, gp);
    int ginf = 0;
    for (int i = 0; i < d; ++i) ginf |= isinf(gp[i]) != 0;
    double k = 1.0;
    for (int n = 0; n < 100 && (isinf(logpp) || ginf); ++n) {  /* gradprime is not refreshed (NJ:449-452) */
        k *= 0.5;
        double *g2 = G->w + 4 * d;
        logpp = leapfrog(G, theta0, r0, grad0, eps * k, tp, rp, g2);
    }
    eps = 0.5 * k * eps;
    double ap = accept_ratio(c, logpp, rp, logp0, r0);
    const double a = 2.0 * (double)(ap > 0.5) - 1.0;
    for (int n = 0; n < 100 && ((a > 0.0 ? ap : 1.0 / ap) > (a > 0.0 ? 0.5 : 2.0)); ++n) {
        eps = eps * (a > 0.0 ? 2.0 : 0.5);
        logpp = leapfrog(G, theta0, r0, grad0, eps, tp, rp, gp);
        ap = accept_ratio(c, logpp, rp, logp0, r0);
    }
    return eps;
}

/* NUTSJump.__call__ (NJ:654-840) with force_trajlen = force_epsilon = None */
static void nuts_call(gj_ctx *G, gj_rng *rng, double *st, const double *x, int64_t iter, double *qout, double *qxy)
{
    const orc_cfg *c = G->c;
    const int d = c->ndim;
    double *v = (double *)malloc(sizeof(double) * 14 * (size_t)d);
    double *q = v, *grad = v + d, *r0 = v + 2 * d, *sample = v + 3 * d, *tmp = v + 4 * d;
    double *ends = v + 5 * d;                                   /* tm rm gm tp rp gp */
    double *fre = v + 11 * d;                                   /* 3d for find_reasonable_epsilon (+ r0 slot) */
    st[GJ_NITER] += 1.0;
    tab_vec(c->gj_tab + (size_t)d * d, x, q, d);
    const double logp = func_grad_white(G, q, grad);
    if (st[GJ_HAVE_EPS] == 0.0) {
        double *w4 = (double *)malloc(sizeof(double) * 4 * (size_t)d);
        st[GJ_EPS] = find_reasonable_epsilon(G, rng, q, grad, logp, w4);
        free(w4);
        st[GJ_MU] = orc_log(10.0 * st[GJ_EPS]);
        st[GJ_HAVE_EPS] = 1.0;
    }
    (void)fre;
    gj_momenta(rng, d, r0);
    const double joint = loghamiltonian(c, logp, r0);
    const double logu = joint - gj_exponential(rng);
    memcpy(sample, q, sizeof(double) * d);
    double lnprob = logp;
    double *tm = ends, *rm = ends + d, *gm = ends + 2 * d, *tp = ends + 3 * d, *rp = ends + 4 * d, *gp = ends + 5 * d;
    memcpy(tm, q, sizeof(double) * d); memcpy(tp, q, sizeof(double) * d);
    memcpy(rm, r0, sizeof(double) * d); memcpy(rp, r0, sizeof(double) * d);
    memcpy(gm, grad, sizeof(double) * d); memcpy(gp, grad, sizeof(double) * d);
    int j = 0, s = 1;
    int64_t n = 1, ip = 0, im = 0;
    double alpha = 0.0; int64_t nalpha = 1;
    while (s == 1) {
        const int dir = 2 * (gj_uniform(rng) < 0.5) - 1;
        gj_tree t;
        if (dir == -1) {
            build_tree(G, rng, tm, rm, gm, logu, dir, j, st[GJ_EPS], joint, im, &t, tmp);
            memcpy(tm, T_TM(&t), sizeof(double) * 3 * d);
        } else {
            build_tree(G, rng, tp, rp, gp, logu, dir, j, st[GJ_EPS], joint, ip, &t, tmp);
            memcpy(tp, T_TP(&t), sizeof(double) * 3 * d);
        }
        ip = t.ip; im = t.im;
        if (t.s == 1) {
            const double ratio = (double)t.n / (double)n;
            if (gj_uniform(rng) < (1.0 < ratio ? 1.0 : ratio)) { memcpy(sample, T_TH(&t), sizeof(double) * d); lnprob = t.logp; }
        }
        n += t.n;
        s = t.s && stop_criterion(c, tm, tp, rm, rp, tmp);
        alpha = t.alpha; nalpha = t.nalpha;
        free(t.v);
        j += 1;
        if (j > c->nuts_maxdepth) s = 0;                       /* cap (not in the reference) */
    }
    /* dual averaging (NJ:805-816) */
    const double it_call = st[GJ_NITER];
    double eta = 1.0 / (it_call + 10.0);
    st[GJ_HBAR] = (1.0 - eta) * st[GJ_HBAR] + eta * (c->nuts_delta - alpha / (double)nalpha);
    if (iter <= c->gj_nburn) {
        st[GJ_EPS] = orc_exp(st[GJ_MU] - sqrt(it_call) / 0.05 * st[GJ_HBAR]);
        eta = orc_pow_neg(it_call, 0.75);
        st[GJ_EPSBAR] = orc_exp((1.0 - eta) * orc_log(st[GJ_EPSBAR]) + eta * orc_log(st[GJ_EPS]));
    } else {
        st[GJ_EPS] = st[GJ_EPSBAR];
    }
    tab_vec(c->gj_tab, sample, qout, d);
    *qxy = logp - lnprob;                                       /* undoes the outer Hastings ratio (NJ:838) */
    free(v);
}

/* one jump call on a single point: the unit the reference fixture (tests/golden/gradjump.npz) pins.
 * kind: J_NUTS or J_HMC; state: GJ_NSTATE doubles (EPSBAR starts at 1.0). */
ORC_API int orc_gradjump(const orc_cfg *c, int kind, const double *x, int64_t iter, double beta, double *state,
                         uint64_t sid, orc_replay *rp, double *q, double *qxy, int64_t *nleap)
{
    gj_ctx G = { c, beta, (double *)malloc(sizeof(double) * 5 * (size_t)c->ndim), 0 };
    gj_rng rng = { rp, c->seed, (uint64_t)iter, (uint32_t)sid, 0, 0, c->lanes };
    if (kind == J_NUTS) nuts_call(&G, &rng, state, x, iter, q, qxy);
    else hmc_call(&G, &rng, state, x, q, qxy);
    if (nleap) *nleap = G.nleap;
    free(G.w);
    return rp ? (int)rp->err : 0;
}

/* ------------------------------------------------------------ MH steps */
/* One Metropolis-Hastings update of one chain: PT:601-622 with _jump PT:1048-1067,
 * SCAM PT:820-876, AM PT:879-933, DE PT:936-985. */
static void mh_one(const orc_cfg *c, orc_state *st, int w, int s, int64_t it, orc_replay *rp, double *buf, int k /* step of the launch */)
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
    uint64_t P[2] = {0, 0}, Q[2] = {0, 0};
    if (!r) { philox_words(c->seed, (uint64_t)it, sid, SLOT_P, P); philox_words(c->seed, (uint64_t)it, sid, SLOT_Q, Q); }

    /* pick from the weighted cycle (PT:1058); pick_mode WALKER: the draw of the walker's rank 0 serves all its ranks */
    const int w_de = c->de_on ? c->w_de : 0;
    const int L = c->w_scam + c->w_am + w_de + c->w_nuts + c->w_hmc;
    uint32_t pickw = hi32(P[0]);
    if (!r && c->pick_mode == PICK_WALKER) {
        uint64_t P0[2];
        philox_words(c->seed, (uint64_t)it, (uint32_t)((uint64_t)(c->walker0 + w) * (uint32_t)c->ntemps_global), SLOT_P, P0);
        pickw = hi32(P0[0]);
    }
    const int ind = r ? (int)rp_next(r, K_INT, L) : (int)h2index(pickw, (uint32_t)L);
    const int jt = ind < c->w_scam ? J_SCAM : (ind < c->w_scam + c->w_am ? J_AM : (ind < c->w_scam + c->w_am + w_de ? J_DE :
                   (ind < c->w_scam + c->w_am + w_de + c->w_nuts ? J_NUTS : J_HMC)));
    double qxy = 0.0;

    /* group pick (PT:839,897,955); counter mode: its own Philox call, drawn only when there is a choice */
    int g = 0;
    if (jt >= J_NUTS) g = 0;                                        /* the gradient jumps move all parameters */
    else if (r) g = (int)rp_next(r, K_INT, ngr);
    else if (ngr > 1) {
        uint64_t Gw[2];
        philox_words(c->seed, (uint64_t)it, sid, SLOT_G, Gw);
        g = (int)h2index(hi32(Gw[0]), (uint32_t)ngr);
    }
    const int ng = (c->ngroups > 1) ? c->gsize[g] : d;
    const double *Ut = st->Ut + (wc * ngr + g) * (size_t)d * d, *S = st->S + (wc * ngr + g) * (size_t)d;
    const double *gm = (c->ngroups > 1) ? c->gmask + (size_t)g * d : NULL;

    if (jt >= J_NUTS) {
        gj_ctx G = { c, beta, (double *)malloc(sizeof(double) * 5 * (size_t)d), 0 };
        gj_rng rng = { r, c->seed, (uint64_t)it, sid, 0, 0, c->lanes };
        double *gst = st->gj + ((size_t)w * nt + t) * GJ_NSTATE;
        if (jt == J_NUTS) nuts_call(&G, &rng, gst, x, it, q, &qxy);
        else hmc_call(&G, &rng, gst, x, q, &qxy);
        gst[GJ_NLEAP] += (double)G.nleap;
        free(G.w);
    } else if (jt == J_SCAM || jt == J_AM) {
        const double prob = r ? rp_next(r, K_UNI, 0) : h2uniform(lo32(P[0]));
        double scale = prob > 0.97 ? 10.0 : (prob > 0.9 ? 0.2 : 1.0);
        if (temp <= 100.0) scale *= sqrt(temp);                         /* PT:861-862 */
        if (jt == J_SCAM) {
            int k;
            double z;
            if (r) { k = (int)rp_next(r, K_INT, ng); z = rp_next(r, K_NRM, 0); }
            else {
                k = (int)h2index(hi32(Q[1]), (uint32_t)ng);
                z = orc_unit_normal32(Q[0], lo32(Q[1]));                 /* Box-Muller, 32-bit angle */
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
                    double zc, zs;
                    orc_unit_normals(E[0], E[1], &zc, &zs);
                    z = which ? zs : zc;
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
            mm = (int)h2index(hi32(Q[0]), (uint32_t)Bn);
            nn = (int)(((uint32_t)mm + 1u + h2index(lo32(Q[0]), (uint32_t)(Bn - 1))) % (uint32_t)Bn);
            prob = h2uniform(lo32(P[0]));
        }
        if (prob > 0.5) scale = 1.0;
        else {
            double rr;
            if (r) rr = rp_next(r, K_UNI, 0);
            else rr = w2uniform(Q[1]);
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
    /* native schedule: the accept uniform is the (0,1] one of word P[1], its log by orc_unit_log */
    const double log_u = r ? orc_log(rp_next(r, K_UNI, 0)) : orc_unit_log(P[1]);
    const double diff = newlnprob - lnprob0 + qxy;                  /* qxy = 0 for SCAM / AM / DE */
    const int accepted = diff > log_u;
    if (accepted) {
        memcpy(x, q, sizeof(double) * d);
        st->lnL[ch] = newlnL; st->lp[ch] = lp;
        st->nacc[(size_t)w * nt + t] += 1;
        st->jstat[(((size_t)w * nt + t) * J_NTYPES + jt) * 2 + 1] += 1;
    }

    /* AM buffer (PT:327-328): the rank-0 chain, unless a swap follows this iteration
     * (the reference stores the post-swap state, PT:624-627; orc_swap then writes it) */
    if (c->temp0 + t == 0 && st->AM) {
        const int swap_follows = c->tskip > 0 && c->ntemps_global > 1 && it % c->tskip == 0;
        if (!swap_follows) {
            const int64_t ring = it % c->cov_update;
            memcpy(st->AM + ((size_t)w * c->cov_update + (size_t)ring) * d, x, sizeof(double) * d);
            if (st->AMflag)       /* the step kernels' rule (am_store_step): KEY = first step of a launch, ring rows 0 and 1 */
                st->AMflag[(size_t)w * c->cov_update + (size_t)ring] = ((k == 0 || ring <= 1) ? 2u : 0u) | (accepted ? 1u : 0u);
        }
    }
}

ORC_API int orc_mh_steps(const orc_cfg *c, orc_state *st, int64_t iter0, int nsteps, orc_replay *rp)
{
    double *buf = (double *)malloc(sizeof(double) * 4 * (size_t)c->ndim);
    for (int k = 0; k < nsteps; ++k)
        for (int w = 0; w < c->nwalkers; ++w)
            for (int s = 0; s < c->ntemps; ++s) mh_one(c, st, w, s, iter0 + k, rp, buf, k);
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
            /* PT:679 accepts iff u <= exp(sum); stated in log space, log(u) <= sum: the same decision (log is monotone; u = 0
             * accepts always, a NaN sum never) up to the last-ulp wiggles of either function, and the form the device's sweep
             * can keep out of its pair-to-pair recurrence (the logarithm depends on the uniform alone) */
            if (orc_log(u) <= la) {
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
            if (orc_log(w2uniform(W[0])) <= la) {
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
        if (st->AM && c->temp0 == 0) {
            memcpy(st->AM + ((size_t)w * c->cov_update + (size_t)(iter % c->cov_update)) * d,
                   st->X + ((size_t)w * nt + so[0]) * d, sizeof(double) * d);
            if (st->AMflag) st->AMflag[(size_t)w * c->cov_update + (size_t)(iter % c->cov_update)] = 2u;   /* the swap's row is a KEY row */
        }
    }
    free(ns);
}

/* ------------------------------------------------------------- Welford */
/* PT:769-794 for one walker: mem buffered rows in buffer order.  fused = 0 is the
 * reference's arithmetic (one product, one sum); fused = 1 accumulates with one fma per
 * element, the upper triangle only (the lower one is its mirror image), and advances the mean by
 * diff * (1/it) (the pooled mode of the engine, which is not a replica of a reference run). */
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
            for (int j = fused ? i : 0; j < d; ++j) {       /* fused: the upper triangle only ... */
                if (fused) M2[(size_t)i * d + j] = fma(diff[i], e[j], M2[(size_t)i * d + j]);
                else M2[(size_t)i * d + j] += diff[i] * e[j];
            }
    }
    if (fused)                                              /* ... mirrored: M2 is symmetric by construction */
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < i; ++j) M2[(size_t)i * d + j] = M2[(size_t)j * d + i];
    if (cov) for (int i = 0; i < d * d; ++i) cov[i] = M2[i] / (double)(it - 1);
    free(diff);
}
ORC_API void orc_welford(int d, int mem, int64_t iter, const double *AM, double *mu, double *M2, double *cov)
{
    orc_welford2(d, mem, iter, AM, mu, M2, cov, 0);
}


/* Pooled covariance (cov_mode "pooled": one covariance for all walkers, adapted from ALL their rank-0 samples; an engine
 * mode, not a replica of a reference run -- the reference has one chain per covariance, PT:769-794).  The definition the
 * HIP kernels are held to bit for bit:
 *   - the epoch's chunk = the W * mem buffered rows, AM[w][r][:] in memory order, cut into slabs of `slab` walkers;
 *   - shift c = the running pooled mean (the first epoch: walker 0's row 0), dx = x - c;
 *   - per slab, rows ascending: T_s[i][j] = fma(dx_i, dx_j, T_s[i][j]) and t_s[i] = fma(dx_i, 1, t_s[i]) -- the order in which
 *     v_mfma_f64_16x16x4 accumulates; the slabs are then summed in order (plain sums);
 *   - chunk statistics about its own mean: M2_b = T - t t^T / n_b; combined with the running (n, mu, M2) by Chan's
 *     formula with delta = t / n_b (the shift IS the running mean); cov = M2 / (n - 1).
 * mu[d], M2[d][d] are the pooled state (updated in place); iter = the multiple of mem just completed. */
ORC_API void orc_pool_update(int d, int nwalkers, int mem, int64_t iter, int slab, const double *AM, double *mu, double *M2, double *cov)
{
    const int first = iter == mem;
    const double nb = (double)nwalkers * (double)mem, nprev = (double)nwalkers * (double)(iter - mem);
    double *c = (double *)malloc(sizeof(double) * (size_t)d), *dx = (double *)malloc(sizeof(double) * (size_t)d);
    double *T = (double *)calloc((size_t)d * d, sizeof(double)), *Ts = (double *)malloc(sizeof(double) * (size_t)d * d);
    double *t = (double *)calloc((size_t)d, sizeof(double)), *ts = (double *)malloc(sizeof(double) * (size_t)d);
    for (int i = 0; i < d; ++i) c[i] = first ? AM[i] : mu[i];
    for (int w0 = 0; w0 < nwalkers; w0 += slab) {
        const int w1 = w0 + slab < nwalkers ? w0 + slab : nwalkers;
        memset(Ts, 0, sizeof(double) * (size_t)d * d);
        memset(ts, 0, sizeof(double) * (size_t)d);
        for (size_t r = (size_t)w0 * mem; r < (size_t)w1 * mem; ++r) {
            const double *x = AM + r * d;
            for (int i = 0; i < d; ++i) dx[i] = x[i] - c[i];
            for (int i = 0; i < d; ++i) {
                ts[i] = fma(dx[i], 1.0, ts[i]);
                for (int j = i; j < d; ++j) Ts[(size_t)i * d + j] = fma(dx[i], dx[j], Ts[(size_t)i * d + j]);
            }
        }
        for (int i = 0; i < d; ++i) {
            t[i] += ts[i];
            for (int j = i; j < d; ++j) T[(size_t)i * d + j] += Ts[(size_t)i * d + j];
        }
    }
    const double f = nprev * nb / (nprev + nb), g = nb / (nprev + nb), den = nprev + nb - 1.0;
    for (int i = 0; i < d; ++i)
        for (int j = i; j < d; ++j) {
            const double M2b = T[(size_t)i * d + j] - (t[i] * t[j]) / nb;
            double m;
            if (first) m = M2b;
            else m = (M2[(size_t)i * d + j] + M2b) + ((t[i] / nb) * (t[j] / nb)) * f;
            M2[(size_t)i * d + j] = M2[(size_t)j * d + i] = m;
            cov[(size_t)i * d + j] = cov[(size_t)j * d + i] = m / den;
        }
    for (int i = 0; i < d; ++i) mu[i] = first ? c[i] + t[i] / nb : mu[i] + (t[i] / nb) * g;
    free(c); free(dx); free(T); free(Ts); free(t); free(ts);
}

/* The same over run-length-compacted rows (the engine's am_mode "rle", include/ptmi.h AMflag).  A rejected proposal leaves the rank-0
 * chain where it was, so its row repeats the row before it; the flags say which rows are STORED rows (NEW or KEY).  Within a slab every
 * stored row r enters ONCE, scaled by s = sqrt(n_r), n_r = the length of its run (to the next stored row, or to the slab's end; ring row
 * 0 of every walker is a KEY row, so a run never leaves its walker's ring): a = s dx, T_ij += a_i a_j, t_i += a_i s -- the sums of
 * n_r dx dx^T and n_r dx, i.e. the same statistics as orc_pool_update with every repeat collapsed into one weighted row (s * s is n_r
 * up to one rounding).  Everything else as above. */
ORC_API void orc_pool_update_rle(int d, int nwalkers, int mem, int64_t iter, int slab, const double *AM, const uint64_t *flag, double *mu,
                                 double *M2, double *cov)
{
    const int first = iter == mem;
    const double nb = (double)nwalkers * (double)mem, nprev = (double)nwalkers * (double)(iter - mem);
    double *c = (double *)malloc(sizeof(double) * (size_t)d), *a = (double *)malloc(sizeof(double) * (size_t)d);
    double *T = (double *)calloc((size_t)d * d, sizeof(double)), *Ts = (double *)malloc(sizeof(double) * (size_t)d * d);
    double *t = (double *)calloc((size_t)d, sizeof(double)), *ts = (double *)malloc(sizeof(double) * (size_t)d);
    for (int i = 0; i < d; ++i) c[i] = first ? AM[i] : mu[i];
    for (int w0 = 0; w0 < nwalkers; w0 += slab) {
        const int w1 = w0 + slab < nwalkers ? w0 + slab : nwalkers;
        const size_t beg = (size_t)w0 * mem, end = (size_t)w1 * mem;
        memset(Ts, 0, sizeof(double) * (size_t)d * d);
        memset(ts, 0, sizeof(double) * (size_t)d);
        for (size_t r = beg; r < end; ++r) {
            if (!(flag[r] & 3u)) continue;
            size_t nx = r + 1;
            while (nx < end && !(flag[nx] & 3u)) ++nx;
            const double s = sqrt((double)(nx - r));
            const double *x = AM + r * d;
            for (int i = 0; i < d; ++i) a[i] = (x[i] - c[i]) * s;
            for (int i = 0; i < d; ++i) {
                ts[i] = fma(a[i], s, ts[i]);
                for (int j = i; j < d; ++j) Ts[(size_t)i * d + j] = fma(a[i], a[j], Ts[(size_t)i * d + j]);
            }
        }
        for (int i = 0; i < d; ++i) {
            t[i] += ts[i];
            for (int j = i; j < d; ++j) T[(size_t)i * d + j] += Ts[(size_t)i * d + j];
        }
    }
    const double f = nprev * nb / (nprev + nb), g = nb / (nprev + nb), den = nprev + nb - 1.0;
    for (int i = 0; i < d; ++i)
        for (int j = i; j < d; ++j) {
            const double M2b = T[(size_t)i * d + j] - (t[i] * t[j]) / nb;
            double m;
            if (first) m = M2b;
            else m = (M2[(size_t)i * d + j] + M2b) + ((t[i] / nb) * (t[j] / nb)) * f;
            M2[(size_t)i * d + j] = M2[(size_t)j * d + i] = m;
            cov[(size_t)i * d + j] = cov[(size_t)j * d + i] = m / den;
        }
    for (int i = 0; i < d; ++i) mu[i] = first ? c[i] + t[i] / nb : mu[i] + (t[i] / nb) * g;
    free(c); free(a); free(T); free(Ts); free(t); free(ts);
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

/* value and gradient of a built-in likelihood (what the gradient jumps see) */
ORC_API double orc_logl_grad(const orc_cfg *c, const double *x, double *g)
{
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)c->ndim);
    double v = eval_logl_grad(c, x, tmp, g);
    free(tmp);
    return v;
}


/* ------------------------------------------------ batched Jacobi eigensolver */
/* The engine's device eigensolver (eig_mode "jacobi", csrc/ptmi_abi.hip eig_jacobi_kernel), restated operation for
 * operation: it replaces np.linalg.svd(cov) of _updateRecursive (PT:797-803) where thousands of per-walker covariances
 * have to be factorized per epoch.  One-sided (Hestenes) Jacobi on the rows of W = V^T A, A symmetric positive
 * semi-definite: rotations of row pairs (p, q) in the round-robin order of the circle method, applied to W and to the
 * accumulated V^T alike, until the rows of W are mutually orthogonal; then W W^T = V^T A^2 V is diagonal, row k of V^T is
 * an eigenvector and ||row k of W|| its eigenvalue.  Each row pair is worked on by eight lanes: lane l sums the
 * elements l, l + 8, ... with fma in ascending order and the lanes combine as ((s0+s4)+(s2+s6)) + ((s1+s5)+(s3+s7)).
 * Output: eigenvalues descending (ties by ascending row), eigenvectors as ROWS of Ut, each with its largest-magnitude
 * component (first of equals) made positive.  Returns the number of sweeps run. */
static double jac_dot(const double *a, const double *b, int d)
{
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int l = 0; l < 8; ++l)
        for (int i = l; i < d; i += 8) s[l] = fma(a[i], b[i], s[l]);
    return ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
}
ORC_API int orc_eig_jacobi(int d, const double *cov, double *Ut, double *S, int max_sweeps)
{
    const int n = d + (d & 1), P = n / 2, rounds = n - 1;
    double *W = (double *)malloc(sizeof(double) * 2 * (size_t)d * d), *V = W + (size_t)d * d;
    memcpy(W, cov, sizeof(double) * (size_t)d * d);
    for (int i = 0; i < d * d; ++i) V[i] = 0.0;
    for (int i = 0; i < d; ++i) V[(size_t)i * d + i] = 1.0;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < rounds; ++r) {
            for (int k = 0; k < P; ++k) {            /* the pairs of a round are disjoint: any order */
                int a = k == 0 ? n - 1 : (r + k) % (n - 1);
                int b = k == 0 ? r : (r - k + (n - 1)) % (n - 1);
                const int p = a < b ? a : b, q = a < b ? b : a;
                if (q >= d) continue;                /* the bye of an odd dimension */
                double *wp = W + (size_t)p * d, *wq = W + (size_t)q * d, *vp = V + (size_t)p * d, *vq = V + (size_t)q * d;
                const double al = jac_dot(wp, wp, d), be = jac_dot(wq, wq, d), ga = jac_dot(wp, wq, d);
                if (!(fabs(ga) > 0x1.0p-50 * sqrt(al * be))) continue;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < d; ++i) {
                    const double x = wp[i], y = wq[i];
                    wp[i] = c * x - sn * y;
                    wq[i] = sn * x + c * y;
                    const double u = vp[i], v = vq[i];
                    vp[i] = c * u - sn * v;
                    vq[i] = sn * u + c * v;
                }
                rotated = 1;
            }
        }
        if (!rotated) break;
    }
    double *nrm = (double *)malloc(sizeof(double) * (size_t)d);
    for (int k = 0; k < d; ++k) nrm[k] = sqrt(jac_dot(W + (size_t)k * d, W + (size_t)k * d, d));
    for (int k = 0; k < d; ++k) {
        int rank = 0;
        for (int j = 0; j < d; ++j) rank += (nrm[j] > nrm[k]) || (nrm[j] == nrm[k] && j < k);
        const double *vk = V + (size_t)k * d;
        int im = 0;
        for (int i = 1; i < d; ++i) if (fabs(vk[i]) > fabs(vk[im])) im = i;
        const double sg = vk[im] < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < d; ++i) Ut[(size_t)rank * d + i] = sg * vk[i];
        S[rank] = nrm[k];
    }
    free(nrm); free(W);
    return sweep;
}

/* ---------------------------------------------- tridiagonal QL eigensolver */
/* The engine's eig_mode "ql" (ptmi_eig_ql, csrc/ptmi_abi.hip eig_ql_kernel): the eigendecomposition of a symmetric matrix by
 * Householder tridiagonalization with the transformations accumulated, then implicit QL iterations with shifts on the tridiagonal
 * matrix (the classical tred2 / tql2 pair of the EISPACK literature, restated without the row scaling: covariances are of moderate
 * size) -- a few passes of O(n) dependent scalar work per eigenvalue, where the Jacobi sweeps of eig_mode "jacobi" take nine sweeps
 * of n^2 / 2 rotations on the nearly degenerate spectra an isotropic target adapts to.  Every dot product is eight interleaved fma chains (QL_DOT8), every other sum one product and one
 * sum, every quotient a correctly rounded division, the only other function sqrt: the kernel does the same
 * operations in the same order, so both give the same bits.  Output as orc_eig_jacobi: eigenvalues in absolute value, descending
 * (ties by ascending column), eigenvectors as ROWS of Ut, each with its largest-magnitude component (first of equals) made
 * positive.  Returns the number of QL iterations (negative: an eigenvalue did not converge within ORC_QL_MAXIT). */
#define ORC_QL_MAXIT 60
/* the dot products of the reduction and accumulation phases: eight interleaved fma chains (term k into chain k mod 8), combined as
 * ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7)) -- a thread of the kernel runs the eight chains side by side */
#define QL_DOT8(n_, term_a, term_b, out)                                                                  \
    do {                                                                                                  \
        double s8_[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};                                         \
        for (int k = 0; k < (n_); ++k) s8_[k & 7] = fma((term_a), (term_b), s8_[k & 7]);                  \
        (out) = ((s8_[0] + s8_[4]) + (s8_[2] + s8_[6])) + ((s8_[1] + s8_[5]) + (s8_[3] + s8_[7]));        \
    } while (0)
ORC_API int orc_eig_ql(int n, const double *cov, double *Ut, double *S)
{
    double *z = (double *)malloc(sizeof(double) * ((size_t)n * n + 2 * (size_t)n)), *d = z + (size_t)n * n, *e = d + n;
    memcpy(z, cov, sizeof(double) * (size_t)n * n);
#define Z(i, j) z[(size_t)(i) * n + (j)]
    /* Householder reduction of the lower triangle, rows n-1 .. 1; row i of z keeps the vector u, column i keeps u / h */
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0;
        if (l > 0) QL_DOT8(l + 1, Z(i, k), Z(i, k), h);
        if (l == 0 || h == 0.0) {
            e[i] = Z(i, l);
            d[i] = 0.0;
            continue;
        }
        const double f0 = Z(i, l);
        const double g0 = f0 >= 0.0 ? -sqrt(h) : sqrt(h);
        e[i] = g0;
        h = h - f0 * g0;
        Z(i, l) = f0 - g0;
        for (int j = 0; j <= l; ++j) {
            Z(j, i) = Z(i, j) / h;
            double g;
            QL_DOT8(l + 1, (k <= j ? Z(j, k) : Z(k, j)), Z(i, k), g);
            e[j] = g / h;
        }
        double f;
        QL_DOT8(l + 1, e[k], Z(i, k), f);
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) e[j] = e[j] - hh * Z(i, j);
        for (int j = 0; j <= l; ++j)
            for (int k = 0; k <= j; ++k) Z(j, k) = Z(j, k) - (Z(i, j) * e[k] + e[j] * Z(i, k));
        d[i] = h;
    }
    d[0] = 0.0;
    e[0] = 0.0;
    /* accumulation of the transformations (column j of the leading block only needs its own product g) */
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g;
                QL_DOT8(l + 1, Z(i, k), Z(k, j), g);
                for (int k = 0; k <= l; ++k) Z(k, j) = Z(k, j) - g * Z(k, i);
            }
        }
        d[i] = Z(i, i);
        Z(i, i) = 1.0;
        for (int j = 0; j <= l; ++j) { Z(j, i) = 0.0; Z(i, j) = 0.0; }
    }
    /* implicit QL on (d, e); the rotations go into the columns of z */
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    int iters = 0, failed = 0;
    for (int l = 0; l < n; ++l) {
        const double t0 = fabs(d[l]) + fabs(e[l]);
        if (tst1 < t0) tst1 = t0;
        int m = l;
        while (m < n - 1 && tst1 + fabs(e[m]) != tst1) ++m;       /* e[n-1] = 0 ends the search at the latest */
        if (m > l) {
            int it = 0;
            do {
                if (++it > ORC_QL_MAXIT) { failed = 1; break; }
                ++iters;
                const double g = d[l];
                const double p0 = (d[l + 1] - g) / (2.0 * e[l]);
                const double r0 = sqrt(p0 * p0 + 1.0);
                const double pr = p0 + (p0 >= 0.0 ? r0 : -r0);
                d[l] = e[l] / pr;
                d[l + 1] = e[l] * pr;
                const double dl1 = d[l + 1];
                const double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] = d[i] - h;
                f = f + h;
                double p = d[m], c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    const double gg = c * e[i], hh = c * p;
                    const double r = sqrt(p * p + e[i] * e[i]);
                    const double ri = 1.0 / r;                       /* ONE division on the rotations' dependent chain */
                    e[i + 1] = s * r;
                    s = e[i] * ri;
                    c = p * ri;
                    p = c * d[i] - s * gg;
                    d[i + 1] = hh + s * (c * gg + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        const double zk = Z(k, i + 1);
                        Z(k, i + 1) = s * Z(k, i) + c * zk;
                        Z(k, i) = c * Z(k, i) - s * zk;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (tst1 + fabs(e[l]) != tst1);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    /* order and signs as orc_eig_jacobi */
    for (int k = 0; k < n; ++k) {
        const double mine = fabs(d[k]);
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (fabs(d[j]) > mine) || (fabs(d[j]) == mine && j < k);
        int im = 0;
        for (int i = 1; i < n; ++i) if (fabs(Z(i, k)) > fabs(Z(im, k))) im = i;
        const double sg = Z(im, k) < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) Ut[(size_t)rank * n + i] = sg * Z(i, k);
        S[rank] = mine;
    }
#undef Z
    free(z);
    return failed ? -iters - 1 : iters;
}

ORC_API int orc_sizeof_cfg(void) { return (int)sizeof(orc_cfg); }

/* The running mean's division as welford_rows_kernel does it (ptmi_abi.hip div_by_count; PT:787 mean += diff / n): q = a r with
   r = 1 / n, corrected twice through the exact remainder.  Not a definition of its own -- it must equal a / n bit for bit
   (tests/test_welford_div.py checks that, the oracle's Welford recurrence divides plainly). */
ORC_API void orc_div_by_count(long len, const double *a, const double *n, double *out)
{
    for (long i = 0; i < len; ++i) {
        const double r = 1.0 / n[i];
        if (fabs(a[i]) < 0x1p-900) { out[i] = a[i] / n[i]; continue; }
        double q = a[i] * r;
        double e = fma(-n[i], q, a[i]);
        q = fma(e, r, q);
        e = fma(-n[i], q, a[i]);
        out[i] = fma(e, r, q);
    }
}
