/*
 * ptmi.h -- C ABI of libptmi.so, the MI355X (gfx950) engine behind the
 * Metropolis-Hastings inner loop of a PTSampler-compatible parallel-tempering
 * sampler.
 *
 * The reference (nanograv/PTMCMCSampler) has no FFI: its hot path is the Python
 * method PTSampler.PTMCMCOneStep (PTMCMCSampler/PTMCMCSampler.py:530-629) plus
 * PTswap (:631-697), _updateRecursive (:769-803) and _updateDEbuffer (:806-817),
 * one chain per MPI rank.  Each entry point below names the reference code it
 * replaces.  The binding a maintainer would add is a ctypes stub: see
 * INTEGRATION.md and ptmcmcsampler_amd/_lib.py.
 *
 * Conventions
 *   - every call returns 0 on success, a negative PTMI_E* code otherwise; the
 *     message is available from ptmi_last_error() (thread-local);
 *   - no C++ exception crosses this boundary;
 *   - every pointer in ptmi_buffers is DEVICE memory owned by the caller (e.g. a
 *     torch-ROCm tensor's data_ptr(), or ptmi_malloc); pointers in ptmi_config
 *     are HOST memory, copied at create time;
 *   - all arrays are C-contiguous; state rows are float64;
 *   - kernels run asynchronously on the stream given at create time; a handle is
 *     not thread-safe, different handles are independent.
 *
 * Layout ("slots and ranks"): a walker is an independent replica of the whole
 * reference run; it has `ntemps` chains.  State rows never move inside a GPU:
 * row (w, s) -- walker w, SLOT s -- holds whichever state currently sits at
 * temperature RANK temp_of[w][s]; slot_of is the inverse table.  A PT swap only
 * rewrites the two tables.  Per-rank quantities (RNG stream, counters) are
 * indexed by rank, exactly as they belong to an MPI rank in the reference.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_VERSION 1

enum { PTMI_OK = 0, PTMI_EINVAL = -1, PTMI_EHIP = -2, PTMI_EUNSUPPORTED = -3, PTMI_ENODEVICE = -4 };

/* built-in device likelihoods / priors (a Python callback cannot run in a kernel;
 * user callbacks go through ptmi_propose / ptmi_accept instead) */
enum { PTMI_LOGL_ISO = 0,      /* -1/2 sum x^2 */
       PTMI_LOGL_DENSE = 1,    /* -(x-mu)^T P (x-mu) / 2 ; par = mu[d], Pt[d*d] (Pt[j*d+i] = P[i][j]).  P is a precision matrix:
                                * the value is summed over the lower half of its symmetric part (half the products) */
       PTMI_LOGL_CURVED = 2,   /* d/2 copies of examples/curved_likelihood.ipynb's 2-d likelihood */
       PTMI_LOGL_INTERVAL = 3 }; /* the reference's own NUTS workload (tests/test_nuts.py:13-47 GaussianLikelihood inside :50-140 intervalTransform):
                                * N(0, I) on the box (a, b) in the coordinates p of all real numbers, x = (b - a) e^p / (1 + e^p) + a, the
                                * log-Jacobian included: sum_i -x_i^2/2 - log(2 pi)/2 + log(b_i - a_i) + p_i - 2 log(1 + e^p_i);
                                * par = a[d], w[d] = b - a, lw[d] = log w (the host's logarithm: a parameter, not part of the arithmetic).
                                * Served by the kernel shapes of the gradient jumps (ndim <= 512, ptmi_lanes_for_grad) with or without them. */
enum { PTMI_LOGP_FLAT = 0,     /* 0 everywhere */
       PTMI_LOGP_BOX = 1 };    /* 0 inside [lo,hi], -inf outside ; par = lo[d], hi[d] */

/* proposal types; also the index into jstat */
enum { PTMI_J_SCAM = 0, PTMI_J_AM = 1, PTMI_J_DE = 2, PTMI_J_NUTS = 3, PTMI_J_HMC = 4, PTMI_J_NTYPES = 5 };

/* ptmi_config.swap_mode.  SWEEP is PTswap as the reference runs it: every adjacent pair is tried, hottest first, and
 * a state can travel several ranks in one call.  ODDEVEN tries, at swap epoch e = iter / tskip, only the pairs
 * (k, k+1) with k = e (mod 2): the pairs are disjoint, so all of them are decided at once on the device and a
 * sharded ladder moves at most one row per walker across each block edge.  Same pair test, same uniform (the one the
 * sweep would have used for pair k); it is a different, equally valid, Markov kernel -- not a replica of a
 * reference run. */
enum { PTMI_SWAP_SWEEP = 0, PTMI_SWAP_ODDEVEN = 1 };

/* ptmi_config.pick_mode: who draws the entry of the proposal cycle (_jump, PTMCMCSampler.py:1058).  CHAIN: every chain
 * draws its own from its rank's stream, as every MPI rank of the reference does -- a walker is then a replica of a
 * reference run.  WALKER: one draw per (walker, iteration), taken from the stream of the walker's rank 0, decides the
 * proposal TYPE for all temperature ranks of the walker (everything else -- group, scale branch, direction, normals,
 * accept uniform -- stays per chain).  The choice is independent of the state, so every chain still runs a valid
 * Metropolis-Hastings kernel with the same mixture weights; on the device the proposal type becomes wave-uniform
 * (no divergence, the matrix-core AM product runs only on AM steps).  Not a replica of a reference run. */
enum { PTMI_PICK_CHAIN = 0, PTMI_PICK_WALKER = 1 };

typedef struct ptmi_config {
    int32_t ndim;            /* parameters per chain */
    int32_t ntemps;          /* temperature ranks held by THIS handle (a contiguous block) */
    int32_t nwalkers;        /* walkers held by this handle */
    int32_t ntemps_global;   /* ranks in the whole ladder (== ntemps on one GPU) */
    int32_t temp0;           /* first global rank of this block */
    int32_t walker0;         /* first global walker index (RNG stream ids are global) */
    int32_t logl_kind, logp_kind;
    int32_t w_host;          /* cycle entries served by host callbacks (custom / gradient jumps added with
                              * addProposalToCycle before SCAM/AM, :988-1014); split path only, 0 for the fused kernel */
    int32_t w_scam, w_am, w_de;   /* proposal-cycle weights, PTMCMCSampler.py:261-264, 579-585 */
    int32_t de_size;         /* rows of a DE buffer (= burn, :221) */
    int32_t cov_update;      /* rows of an AM buffer (= covUpdate, :220) */
    int32_t tskip;           /* swap period (:624); 0 = never */
    int32_t cov_per_walker;  /* 1: Ut/S/DE/cov per walker (faithful replicas); 0: one pooled set */
    int32_t device;          /* HIP device ordinal */
    int32_t ngroups;         /* parameter groups (PTMCMCSampler.py:129-145); 0 or 1 = one group of all parameters */
    int32_t swap_mode;       /* PTMI_SWAP_SWEEP (the reference's hot -> cold sweep, :666-686) or PTMI_SWAP_ODDEVEN */
    /* Gradient jumps on the built-in likelihoods (the reference adds them when logl_grad / logp_grad are given,
     * :225-258; nutsjump.py): cycle += [NUTS]*w_nuts + [HMC]*w_hmc.  ndim <= 512; the engine then shares a chain among
     * ptmi_lanes_for_grad(ndim) lanes (at most 8 register slots per lane), which fixes the summation orders. */
    int32_t w_nuts, w_hmc;
    int32_t gj_nburn;        /* nburn of the jump objects (= burn, :227,238,251) */
    int32_t hmc_min, hmc_max;/* HMC takes randint(hmc_min, hmc_max) leapfrogs (:240-241: 2, HMCsteps) */
    int32_t nuts_maxdepth;   /* tree heights 0..nuts_maxdepth per call, <= 24.  The reference doubles without a cap
                              * (nutsjump.py:716-802): 24 (2^24 leapfrogs in one call) is never reached, i.e. its behaviour */
    int32_t pick_mode;       /* PTMI_PICK_CHAIN or PTMI_PICK_WALKER (below) */
    double hmc_eps;          /* HMCstepsize (:239) */
    double nuts_delta;       /* target acceptance of NUTS' dual averaging (0.6, :256) */
    uint64_t seed;
    void *stream;            /* hipStream_t to launch on; NULL = the null stream */
    const double *ladder;    /* host [ntemps_global]  temperatures used by the swap (:658) */
    const double *temps_mh;  /* host [ntemps] temperature each local rank samples at (:278-282; hot chain = 1e80) */
    const double *logl_par;  /* host, per logl_kind */
    int64_t logl_par_len;
    const double *logp_par;  /* host, per logp_kind */
    int64_t logp_par_len;
    const int32_t *group_size;  /* host [ngroups] parameters per group (ngroups > 1 only) */
    const double *group_mask;   /* host [ngroups][ndim] 1.0 where a parameter belongs to the group (ngroups > 1 only) */
    const double *gj_tab;       /* host [3][ndim][ndim] whitening tables of the gradient jumps from L = cholesky(cov)
                                 * (nutsjump.py:53-54), each used as out[i] = sum_k T[k][i] v[k]: backward T = L
                                 * (x = L^T q), forward T = L^-1 (q = L^-T x), gradient T = L^T.  NULL without them */
} ptmi_config;

/* Device buffers, caller-owned.  W = nwalkers, T = ntemps, d = ndim,
 * Wc = cov_per_walker ? W : 1.  Optional ones may be NULL when unused. */
typedef struct ptmi_buffers {
    double *X;          /* [W][T][d]   state rows by slot */
    double *lnL;        /* [W][T]      log-likelihood of the row in a slot */
    double *lp;         /* [W][T]      log-prior of the row in a slot */
    int32_t *temp_of;   /* [W][T]      local rank held by a slot */
    int32_t *slot_of;   /* [W][T]      slot holding a local rank */
    double *Ut;         /* [Wc][Ng][d][d]  eigenvectors, one per ROW (Ut[k][i] = U[i][k] of :145,803); Ng = max(ngroups,1);
                         *              a group's vectors are embedded in the full space (zero outside the group, zero rows
                         *              beyond its size) */
    double *S;          /* [Wc][Ng][d]  eigenvalues (zero beyond the group's size) */
    double *DE;         /* [Wc][de_size][stride]  DE history ring (optional); row format: ptmi_de_row_stride */
    double *AM;         /* [W][cov_update][d] samples of the rank-0 chain (:327-328); only where temp0 == 0; row format: ptmi_am_row_format */
    uint64_t *nacc;     /* [W][T]      accepted MH updates per rank (:621) */
    uint64_t *jstat;    /* [W][T][PTMI_J_NTYPES][2]  proposed, accepted per jump type (:602,622) */
    uint64_t *nswap;    /* [W][ntemps_global]  accepted swaps credited to the lower rank (:681) */
    double *mu;         /* [Wc][d]     running mean of the rank-0 chain (:148); pooled: of all walkers' rank-0 samples */
    double *M2;         /* [Wc][d][d]  running sum of outer products about the mean (:147) */
    double *cov;        /* [Wc][d][d]  published covariance (:794) */
    double *Q;          /* [W][T][d]   proposals, split path only (optional) */
    double *qaux;       /* [W][T][4]   split path: qxy, jump type (>= PTMI_J_NTYPES: host entry index + PTMI_J_NTYPES),
                         *              accept uniform, log(accept uniform) (optional) */
    double *AMaux;      /* [W][cov_update][2]  lnL and lp of the rank-0 chain beside each AM row: the
                         *              _lnlike/_lnprob columns of updateChains (:331-335) (optional) */
    double *gj;         /* [W][T][8]   by RANK: the attributes of a rank's NUTSJump / HMCJump object (nutsjump.py:379-433):
                         *              epsilon, mu, Hbar, epsilonbar (starts at 1), NUTS calls, HMC calls, have-epsilon flag, leapfrogs taken so far
                         *              (needed with w_nuts + w_hmc > 0) */
    uint64_t *AMflag;   /* [W][cov_update] one flag word beside every AM row (optional; ptmi_am_flags_ok): bit 0 NEW = the step was accepted,
                         *              bit 1 KEY = the row is stored whatever the step did (first step of a launch, ring rows 0 and 1, the swap's
                         *              post-swap row, rows the caller writes).  With it the step kernels store the rank-0 chain's row
                         *              (updateChains, :327-328) only when it is NEW or KEY -- a rejected proposal leaves the chain where it
                         *              was, its row repeats the one before -- and ptmi_update_cov takes every stored row once, weighted by
                         *              the length of its run.  Readers that want every row call ptmi_am_expand first.  The caller
                         *              initialises every word to KEY (2) and marks rows it writes itself as KEY */
    double *Q2;         /* [W][T][d]   a SECOND proposal buffer (optional, with sloc): ptmi_accept_propose then writes the next proposals to
                         *              the buffer that does NOT hold the current ones (ptmi_proposals says which is which), an accepted
                         *              proposal stays where it is as the chain's state, and rows are copied to X only when their buffer
                         *              is about to be overwritten or at ptmi_accept (csrc/ptmi_split.hip) */
    int32_t *sloc;      /* [W][T]      with Q2: where a chain's state lives between ptmi_propose and ptmi_accept (0 = X, 1 = Q, 2 = Q2);
                         *              all zero outside such a span -- the caller zeroes it once */
} ptmi_buffers;

typedef struct ptmi_engine *ptmi_handle;

const char *ptmi_last_error(void);
int ptmi_version(void);
int ptmi_device_count(int *count);
/* lanes of a wavefront that share one chain for a given ndim (fixes the summation order) */
int ptmi_lanes_for(int ndim);
/* the same with gradient jumps in the cycle (w_nuts + w_hmc > 0); 0 = not supported */
int ptmi_lanes_for_grad(int ndim);

/* temperatureLadder (PTMCMCSampler.py:699-720), host arithmetic: out[i] = Tmin * tstep^i, i < nchain, with
 * tstep = the argument if > 0, else exp(log(Tmax/Tmin)/(nchain-1)) if Tmax > 0, else 1 + sqrt(2/ndim); a single chain
 * gets {1}.  `out` is HOST memory [nchain]. */
int ptmi_temperature_ladder(int nchain, int ndim, double Tmin, double Tmax, double tstep, double *out);

/* Row format of the DE buffer for a given ndim (grad != 0: with gradient jumps in the cycle): *stride doubles per row;
 * *epl == 0: a row holds the parameters in order (stride == ndim); *epl > 0 (4 lanes per chain, *epl slots per lane): the
 * row is dealt to the lanes in 16-byte pieces, row[8 * (e / 2) + 2 * lane + e % 2] = parameter lane + 4 e (zero past ndim
 * and for e >= *epl), stride == 8 * ((*epl + 1) / 2) -- one read instruction of the kernel takes 64 contiguous bytes per chain. */
int ptmi_de_row_stride(int ndim, int grad, int *stride, int *epl);

/* Row format of the AM buffer for a given ndim (grad != 0: with gradient jumps in the cycle).  *epl == 0: a row holds the parameters
 * in order.  *epl > 0 (the exact 4-lane shape: ndim = 4 * *epl = 100): the lanes' order -- position 8 * (e / 2) + 2 * lane + e % 2
 * holds parameter lane + 4 e, the odd last slot e = *epl - 1 at 8 * (*epl / 2) + lane -- so that the rank-0 chain's four lanes store
 * their row 16 bytes at a time.  Every kernel that reads the buffer goes through the same map. */
int ptmi_am_row_format(int ndim, int grad, int *epl);

int ptmi_create(const ptmi_config *cfg, const ptmi_buffers *buf, ptmi_handle *out);
int ptmi_destroy(ptmi_handle h);
int ptmi_sync(ptmi_handle h);

/* lnL and lp of every row from X: the initial evaluation of sample(), PTMCMCSampler.py:479-487 */
int ptmi_eval_state(ptmi_handle h);

/* DE enters the proposal cycle after burn (PTMCMCSampler.py:574-585) */
int ptmi_set_de_active(ptmi_handle h, int on);

/* `nsteps` fused Metropolis-Hastings updates of every chain for iterations
 * iter0 .. iter0+nsteps-1: _jump + SCAM/AM/DE (:1048-1067, :820-985), logp/logl +
 * tempering (:605-612), the Hastings test (:615-622) and the AM-buffer write of
 * updateChains (:327-328).  The caller splits the run at swap / covariance / DE
 * epochs (none may fall strictly inside the range). */
int ptmi_mh_steps(ptmi_handle h, int64_t iter0, int32_t nsteps);

/* Which instantiation of the fused kernel the most recent ptmi_mh_steps launched (the parity tests assert that they
 * reached the one they mean to test): a combination of the flags below, plus lanes per chain in bits 12-19 and register
 * slots per lane in bits 20-27 (PTMI_VAR_UTPAD sits above them). */
enum { PTMI_VAR_STAGED = 1,    /* strided lane layout, tables in LDS, products on the f64 matrix cores */
       PTMI_VAR_FULL = 2,      /* the cycle holds AM and / or an active DE (else SCAM only) */
       PTMI_VAR_LDS_UT = 4,    /* staged: the eigenvector table is in LDS too */
       PTMI_VAR_GROUPS = 8,    /* parameter groups */
       PTMI_VAR_GRADJUMP = 16, /* the kernel with the NUTS / HMC branch */
       PTMI_VAR_UNIFORM = 32,  /* wave-uniform cycle pick (pick_mode = PTMI_PICK_WALKER) */
       PTMI_VAR_AMQ = 64,      /* staged full kernel: AM increments queued 16 at a time for the matrix cores */
       PTMI_VAR_LDS_BOX = 128, /* box prior: the bounds table is in LDS */
       PTMI_VAR_LDS_DRAWT = 256, /* the tables of the draws (log slices, base angles) are in LDS */
       PTMI_VAR_DENSE_SCAM = 512, /* mh_dense_scam_kernel: dense likelihood + SCAM-only cycle, P and Ut unpadded in LDS, 512-thread
                                  * blocks (two waves per SIMD over one copy of the tables) -- what bench.py --logl dense times */
       PTMI_VAR_PERSISTENT = 1024, /* SCAM-only cycle, one eigenvector table for the launch (pooled covariance): persistent blocks, one
                                    * per CU over one LDS copy of the table, each wave walking over units of 16 chains -- what bench.py times */
       PTMI_VAR_PC = 2048,     /* cycles with AM entries, per-chain picks: stepper and AM-producer waves paired per SIMD (mh_pc_kernel) */
       PTMI_VAR_UTPAD = 268435456 /* = 1 << 28, above the shape bits: 16- / 64-lane shapes, SCAM-only cycle, one table for the launch: wide draw batches (all lanes of a chain
                                 * draw), the direction read from the library's zero-padded copy of the table -- what bench.py --ndim 1000 times */ };
int ptmi_last_mh_variant(ptmi_handle h, int32_t *variant);

/* PTswap (:631-697) for iteration `iter` when the whole ladder is on this GPU. */
int ptmi_swap(ptmi_handle h, int64_t iter);

/* The same in three pieces for a ladder sharded over GPUs (the caller all-gathers
 * lnL between 1 and 2 and exchanges the rows that cross a block edge after 2):
 *  1. lnL by local rank:  lnL_pos[w][t] = lnL[w][slot_of[w][t]]
 *  2. the sweep over the global ladder -> map[w][j] = position whose state moves to j
 *     (every GPU computes the identical map; credits go to nswap for local ranks only)
 *  3. the caller rewrites the tables and exchanges the rows that cross a block edge
 *     (ptmcmcsampler_amd/sharded.py: plan_exchange + one all-to-all), then
 *     ptmi_swap_write_am stores the AM row of the state that now sits at rank 0. */
int ptmi_swap_gather_lnl(ptmi_handle h, double *lnL_pos_local /* dev [W][T] */);
int ptmi_swap_sweep(ptmi_handle h, int64_t iter, const double *lnL_pos_global /* dev [W][ntemps_global] */,
                    int32_t *map /* dev [W][ntemps_global] */);
int ptmi_swap_write_am(ptmi_handle h, int64_t iter);

/* Device-side form of pieces 2 and 3, with no host synchronisation (what ShardedPTEngine uses on GPUs):
 *  - ptmi_swap_sweep_blocks: the sweep reading lnL exactly as an all-gather delivers it, [nranks][W][T];
 *  - ptmi_exchange_pack: rewrites slot_of / temp_of for this block from the map and copies the rows that leave
 *    into send[q][w][0..d+1] (state, lnL, lp), q = destination GPU.  One sweep moves at most one row of a walker
 *    to a colder block and at most one to the next hotter block, so [nranks][W] slots are always enough;
 *  - (caller) all-to-all with equal splits: send[q] goes to GPU q, recv[q] comes from GPU q;
 *  - ptmi_exchange_apply: stores the rows that arrived (the pack step recorded where each one goes).
 * ptmi_exchange_status reports (after a sync) whether a plan ever exceeded those bounds (must stay 0). */
int ptmi_swap_sweep_blocks(ptmi_handle h, int64_t iter, const double *lnL_blocks /* dev [nranks][W][T] */,
                           int32_t *map /* dev [W][ntemps_global] */);
int ptmi_exchange_pack(ptmi_handle h, const int32_t *map, double *send /* dev [nranks][W][d+2] */);
int ptmi_exchange_apply(ptmi_handle h, const double *recv /* dev [nranks][W][d+2] */);
int ptmi_exchange_status(ptmi_handle h, int32_t *violations);
/* After ptmi_exchange_pack: 1 if some row of the sweep moves further than to a neighbouring block (possible only when the
 * carried state wins every pair of a whole block), else 0 -- the same answer on every GPU, computed from the global map.
 * With 0 only send[rank-1] and send[rank+1] hold rows: the caller exchanges those two segments with its neighbours
 * (RCCL send/recv over one xGMI link each way) and skips the all-to-all.  The four bytes set out for (pinned) host memory at
 * the end of ptmi_exchange_pack; this call waits for THEM only (an event recorded behind the copy), not for the stream: work the
 * caller queued behind the pack step -- the neighbour exchange, which every epoch needs -- runs meanwhile. */
int ptmi_exchange_multihop(ptmi_handle h, int32_t *flag);

/* _updateRecursive (:769-794) for every walker at iteration `iter` (= the multiple of
 * cov_update just completed): updates mu, M2 and cov.  With cov_per_walker == 0 ONE set
 * (mu[0], M2[0], cov[0]) is adapted from all walkers' buffered rows: shifted sums of outer products on the matrix
 * cores, slab by slab, combined with the running statistics by Chan's formula (the sample covariance of every
 * rank-0 sample so far; oracle: orc_pool_update; with AMflag: over the stored rows, each scaled by the square root of its run
 * length, oracle: orc_pool_update_rle).  The eigendecomposition (:797-803) is a
 * separate step: on the host (LAPACK, as the reference) or ptmi_eig_jacobi. */
int ptmi_update_cov(ptmi_handle h, int64_t iter);
/* The same on `stream` (NULL: the handle's) from the ring AM / AMflag (AM NULL: the handle's; AMflag NULL with AM given: every row
 * is stored): the statistics of a covariance period that is OVER need nothing the next launches touch once those write another
 * ring, so a caller that keeps two rings (ptmi_set_am_buffers at every epoch) runs them on a side stream BESIDE the launches of
 * the next period (PTEngine stats_async; :545-560 with the table taking effect eig_lag launches late).  mu / M2 / cov and the
 * scratch are the handle's: one statistics call at a time. */
int ptmi_update_cov_on(ptmi_handle h, int64_t iter, void *stream, const double *AM, const uint64_t *AMflag);
/* The ring (updateChains' buffer, :327-328) the step kernels, the swap and ptmi_update_de use from now on: a caller with two rings
 * switches at a covariance epoch.  AMaux / AMflag exactly when the handle was created with them.  Takes effect for the calls that
 * follow (host-side pointers of the handle; nothing is queued). */
int ptmi_set_am_buffers(ptmi_handle h, double *AM, double *AMaux, uint64_t *AMflag);

/* AM row flags (ptmi_buffers.AMflag; updateChains' buffer, :327-328).  ptmi_am_flags_ok: 1 when the configuration can keep them --
 * pooled covariance, rank 0 on this GPU (the per-walker recurrence of :778-794 takes every row in turn).  ptmi_am_expand copies, in
 * the AM buffer itself, every row that was not stored from the row before it, for iterations iter_lo .. iter_hi of walkers
 * w0 .. w0 + nw - 1: what a flag-free run would have stored.  Both iterations lie in the current covariance period
 * [E, E + cov_update], E = the last multiple of cov_update below iter_hi (once the ring wraps, the stored row an older repeat hangs on
 * is overwritten; the statistics read a period when it is complete).  With AMflag a launch of ptmi_mh_steps may not cross a period. */
int ptmi_am_flags_ok(const ptmi_config *cfg);
int ptmi_am_expand(ptmi_handle h, int32_t w0, int32_t nw, int64_t iter_lo, int64_t iter_hi);

/* The eigendecomposition of _updateRecursive (:797-803, np.linalg.svd of the covariance) for every covariance the handle
 * holds (Wc matrices), on the device: Ut and S are overwritten from cov.  One-sided Jacobi, one block per matrix, both
 * tables in LDS (ndim <= 101, one parameter group); eigenvalues descending, every eigenvector (a row of Ut) with its
 * largest component positive.  Same subspaces as LAPACK, but not its column signs: a run adapted this way is not a
 * bit-replica of one adapted through the host.  Asynchronous on the handle's stream. */
int ptmi_eig_jacobi(ptmi_handle h);
/* The same by Householder tridiagonalization + implicit QL iterations (oracle: orc_eig_ql; ndim <= 128): one block of two waves per
 * matrix, two per CU at ndim = 100.  The choice for thousands of per-walker covariances: a 100 x 100 matrix takes a quarter of the
 * Jacobi kernel's time on the nearly degenerate spectra an isotropic target adapts to.  Eigenvalues in absolute value, descending;
 * sign rule as ptmi_eig_jacobi.  Asynchronous on the handle's stream. */
int ptmi_eig_ql(ptmi_handle h);
/* ptmi_eig_ql on `stream` (NULL: the handle's), reading `cov_in` (NULL: the cov buffer) and writing `Ut_out` / `S_out` (NULL: the Ut / S
 * buffers), one parameter group: the engine's eig_lag with per-walker covariances runs the factorization of a covariance epoch
 * (:797-803) on a side stream beside the step launches that follow and puts its tables into force a fixed number of launches later
 * (oracle: OracleEngine(eig_lag=L)).  One call at a time per handle. */
int ptmi_eig_ql_from(ptmi_handle h, void *stream, const double *cov_in, double *Ut_out, double *S_out);
/* The same for ONE large pooled covariance (3 <= ndim <= 1024; the engine's eig_mode "sytrd"), all of it the library's own kernels:
 * Householder tridiagonalization in ONE kernel with the matrix in the LDS of its blocks (csrc/ptmi_abi.hip sytrd_lds_kernel), the
 * tridiagonal matrix's eigenvectors by divide and conquer (csrc/ptmi_dc.inc.h: QL on leaves of 16 rows, rank-one merges with
 * deflation, roots of the secular equation by bisection, Gu / Eisenstat weights), back-transformed through the reflectors.
 * Replaces the np.linalg.svd of :797-803 where the ROCm library's eigensolver (torch.linalg.eigh: 35 ms of small kernels at
 * 1000 x 1000) is the epoch.  On `stream` (NULL: the handle's), results into Ut_out [ndim][ndim] / S_out [ndim] (NULL: the handle's
 * Ut / S): eigenvalues in absolute value, descending; no sign rule.  No oracle restates its last bits: the tests hold every table
 * it makes to U diag(S) U^T = cov and U U^T = I at 1e-12 and step the ORACLE's chains with it (tests/test_device_tables_gpu.py).
 * (PTMI_SYTRD_LIB=1: the tridiagonal solver and the back-transformation by rocsolver_dstedc / rocsolver_dormtr instead, an A/B switch.) */
int ptmi_eig_sytrd(ptmi_handle h, void *stream, double *Ut_out, double *S_out);
/* The same from the covariance `cov_in` [ndim][ndim] (device; NULL: the handle's cov): a caller that runs the factorization BESIDE
 * later work -- the statistics of the next covariance epoch overwrite the handle's cov -- hands in its own copy (PTEngine: eig_lag
 * with the statistics of the next epoch ahead of the pending table, :545-560). */
int ptmi_eig_sytrd_from(ptmi_handle h, void *stream, const double *cov_in, double *Ut_out, double *S_out);
/* The convergence word of the most recent ptmi_eig_sytrd whose result has reached the host: 0 = converged; k > 0 = leaf k - 1 of the
 * divide-and-conquer tree did not split off an eigenvalue within 60 QL sweeps (with PTMI_SYTRD_LIB=1: LAPACK's `info` of dstedc).  It
 * follows the factorization on its stream into pinned memory, so the call never waits; check it once that stream has been waited for
 * (the engine does at its next covariance epoch and in sync()).  Non-zero: Ut / S of that epoch are not to be trusted. */
int ptmi_eig_sytrd_info(ptmi_handle h, int32_t *info);

/* _updateDEbuffer (:806-817): drop the oldest cov_update rows of each DE history and
 * append the AM buffer (pooled mode: row r comes from walker r mod W). */
int ptmi_update_de(ptmi_handle h);

/* Position of the logical row 0 inside the DE ring; a GPU that does not hold rank 0 receives
 * the new rows from the one that does and then advances its ring by hand. */
int ptmi_set_de_head(ptmi_handle h, int32_t head);

/* Split path for host (Python) likelihood callbacks, one iteration per call:
 * propose writes Q and qaux for every chain; the caller evaluates logp/logl on Q;
 * accept applies the Hastings test with the caller's values (NaN-safe, -inf prior). */
int ptmi_propose(ptmi_handle h, int64_t iter);
int ptmi_accept(ptmi_handle h, int64_t iter, const double *newlnL /* dev [W][T] */, const double *newlp /* dev [W][T] */);
/* ptmi_accept(iter) and ptmi_propose(iter + 1) in ONE launch (csrc/ptmi_split.hip): a chain's row comes in once -- the accepted
 * proposal or the old state -- and goes out once as the next proposal (and to X where accepted), where the two calls moved it twice;
 * the inner loop of a run whose likelihood is a batched device callback (PTMCMCSampler.py:601-622 between epochs).  The caller calls
 * it only where NOTHING sits between the two iterations: no swap (iter a multiple of Tskip: refused), no covariance / DE epoch
 * and no DE activation at iter + 1 (the caller's schedule; PTEngine.run_callback).  qaux[.][2] then holds the next iteration's
 * accept uniform, not this one's decision (nacc / jstat count it).  Same results as the two calls, bit for bit. */
int ptmi_accept_propose(ptmi_handle h, int64_t iter, const double *newlnL /* dev [W][T] */, const double *newlp /* dev [W][T] */);
/* Cycles with AM entries on the split path: the increments U (cd sqrt(S) z) of the AM picks (:879-933) of iterations iter0 .. iter0 +
 * nsteps - 1 are made AHEAD on the matrix cores (one matrix product for all picks of the piece) with the tables as they are at this call;
 * ptmi_propose / ptmi_accept_propose then read them.  nsteps <= ptmi_split_am_piece (0 there: this handle's AM proposals come from the
 * shape kernels and no call is needed).  A proposal for an iteration no prepared piece covers prepares that one iteration itself, so
 * the call is an optimisation -- the caller makes it at the head of a span of iterations between which NOTHING changes the tables or
 * the cycle (PTEngine.callback_segment). */
int ptmi_split_am_piece(ptmi_handle h, int32_t *piece);
int ptmi_split_am_prepare(ptmi_handle h, int64_t iter0, int32_t nsteps);
/* The split path's inner loop in a hipGraph (launch-bound at small batches: two launches of a few microseconds per iteration).
 * ptmi_set_stream retargets the handle's stream (to the stream a graph is being captured on, and back).  With ptmi_device_iter(h, 1)
 * the `iter` argument of ptmi_propose / ptmi_accept / ptmi_accept_propose is an OFFSET from an iteration counter in device memory that
 * ptmi_set_device_iter writes on the stream -- so a captured span [ptmi_propose(0), callback, ptmi_accept_propose(0), ..., ptmi_accept(L - 1)]
 * replays for any first iteration: set the counter, launch the graph (PTEngine.run_callback(graph=True)).  The kernels derive the
 * ring row and the swap-iteration test from the counter; the host-side checks of those calls are off in this mode; cycles with AM
 * entries are refused (their increments are listed on the host's iteration).  Whatever the captured launches bake in -- the DE
 * ring's head, whether DE is in the cycle -- must hold at replay: the caller captures again when it changes. */
int ptmi_set_stream(ptmi_handle h, void *stream);
/* (a replayed graph does not move the host's record of which buffer holds the current proposals: the caller restores what the capture
 * ended with -- 0 = Q, 1 = Q2 -- for ptmi_proposals' sake) */
int ptmi_set_proposals(ptmi_handle h, int32_t which);
int ptmi_device_iter(ptmi_handle h, int32_t on);
int ptmi_set_device_iter(ptmi_handle h, int64_t value);
/* The buffer that holds the current proposals: Q after ptmi_propose; after ptmi_accept_propose Q or Q2 in turn when the handle has
 * both (ptmi_buffers.Q2: X is then authoritative again only after ptmi_accept), else Q.  The callback reads THIS buffer. */
int ptmi_proposals(ptmi_handle h, double **q);
/* The handle's built-in likelihood of n rows [n][ndim] anywhere on the device (the proposals, say), with the BITS the fused kernels
 * give it (the same lanes, the same summation order): a likelihood "callback" that is a device kernel behind the C ABI -- with it
 * the split path reproduces the fused path bit for bit at any size (tests), and bench.py times the split path with a callback that
 * costs one pass over the proposals.  PTMI_LOGL_ISO only.  On the handle's stream. */
int ptmi_rows_logl(ptmi_handle h, const double *rows /* dev [n][ndim] */, int64_t n, double *out /* dev [n] */);

/* Self-test hooks used by the parity tests: evaluate the device's deterministic math on
 * n inputs (op: 0 log, 1 exp, 2 cos2pi, 3 sqrt, 4 reciprocal-free divide a/b with b=in2). */
int ptmi_selftest_math(int device, int op, const double *in, const double *in2, double *out, int64_t n);
int ptmi_selftest_philox(int device, const uint32_t *ctr_key /* [n][6] */, uint32_t *out /* [n][4] */, int64_t n);
/* Replay hook of the parity tests (tests/test_gpu_replay.py): the reference's RECORDED draws go into the production kernels in
 * place of the Philox ones.  swap_uniforms (dev [W][ntemps_global - 1], index [w][k] = the uniform of pair (k, k+1), :679) feeds
 * ptmi_swap / ptmi_swap_sweep*; draws (dev [W][T][4] 64-bit words per chain: P0 = cycle pick << 32 | scale-branch word, Q0, Q1 as
 * DESIGN.md section 4 lays them out, and the bits of a double: the SCAM normal, :873, or DE's scale uniform, :976) feeds ptmi_propose.
 * NULL switches a hook off. */
int ptmi_test_replay(ptmi_handle h, const double *swap_uniforms, const uint64_t *draws);

/* plain device memory helpers, so that a C caller needs nothing but this library */
int ptmi_malloc(void **p, size_t bytes);
int ptmi_free(void *p);
int ptmi_memcpy_h2d(void *dst, const void *src, size_t bytes);
int ptmi_memcpy_d2h(void *dst, const void *src, size_t bytes);
int ptmi_memset(void *dst, int value, size_t bytes);

/* HIP-event stopwatch on the handle's stream (bench.py's kernel timing) */
int ptmi_timer_start(ptmi_handle h);
int ptmi_timer_stop_ms(ptmi_handle h, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
