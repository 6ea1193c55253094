// ptmi_common.h -- declarations shared by the translation units of libptmi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ptmi.h"
#include "ptmi_device.h"

using namespace ptmi;

// thread-local error message of the C ABI (defined in ptmi_abi.hip)
int ptmi_fail(int code, const char *fmt, ...);
#define fail ptmi_fail
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(PTMI_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Position of parameter i inside a row of the AM buffer.  epl == 0: parameter order.  epl > 0 (the exact 4-lane shape, ndim = 4 epl,
// epl odd): the lanes' order -- 16-byte pieces dealt to the four lanes in turn, position 8 (e / 2) + 2 lane + e % 2 holds
// parameter lane + 4 e, the odd last slot at 8 (epl / 2) + lane -- so that the cold chain's four lanes store their row with
// 16-byte instructions (13 instead of 25: the stores are 15 % of the config-2 kernel).  Every reader goes through this map.
__host__ __device__ inline int am_pos(int i, int epl)
{
    if (epl == 0) return i;
    const int e = i >> 2, ln = i & 3;
    return e < 2 * (epl / 2) ? 8 * (e >> 1) + 2 * ln + (e & 1) : 8 * (epl / 2) + ln;
}
// the parameter held at position p of such a row (the inverse of am_pos)
__host__ __device__ inline int am_inv(int p, int epl)
{
    if (epl == 0) return p;
    const int h = 8 * (epl / 2);
    return p < h ? ((p & 7) >> 1) + 4 * (2 * (p >> 3) + (p & 1)) : (p - h) + 4 * (epl - 1);
}
constexpr int am_row_epl(int G, int EPL) { return (G == 4 && EPL == 25) ? EPL : 0; }      // = ptmi_shape_exact (declared below)

// AM row flags (ptmi_buffers.AMflag, include/ptmi.h): one 8-byte word beside every row of the AM buffer.  A rejected proposal
// leaves the rank-0 chain where it was -- its row repeats the row before it (43 % of the rows at the stationary acceptance of a
// SCAM cycle).  With the flags the step kernels store a row only when it is NEW (the step was accepted) or a KEY row (first step of
// a launch, ring rows 0 and 1, the swap's post-swap row), and the pooled statistics take every stored row once, weighted by the
// length of its run (orc_pool_update_rle).  Readers that want every row call ptmi_am_expand (copy-forward) first.
constexpr unsigned long long AMROW_NEW = 1ull;      // the step was accepted: the row differs from its predecessor and is stored
constexpr unsigned long long AMROW_KEY = 2ull;      // the row is stored whatever the step did
typedef unsigned long long AmFlag;

// ------------------------------------------------------------- kernel args
struct KArgs {
    // state
    double *X, *lnL, *lp;
    int32_t *temp_of, *slot_of;
    const double *Ut, *S, *DE;
    double *AM, *AMaux;
    AmFlag *AMflag;              // flags beside the AM rows (pooled covariance), or nullptr: every step stores its row
    u64 *nacc, *jstat;
    // small device tables owned by the engine
    const double *temps_mh, *beta, *logl_par, *logp_par;
    const int32_t *gsize;                 // [Ng] parameters per group
    const double *gmask, *gcn, *gdiv;     // [Ng][d] membership; [Ng] 2.4/sqrt(2 n_g); [Ng] sqrt(2 n_g)
    // split path
    double *Q, *qaux;
    double *Q2;                  // the second proposal buffer, or nullptr (ptmi_buffers.Q2)
    int32_t *sloc;               // [W][T] where a chain's state lives between ptmi_propose and ptmi_accept: 0 = X, 1 + b = proposal buffer b; or nullptr
    int q_cur, q_tgt;            // the buffer that holds the proposals being accepted; the buffer the next proposals go to
    const double *newlnL, *newlp;
    // scalars
    u64 seed;
    long long iter0;
    const long long *iter_dev;   // split path in a captured graph: the iteration is *iter_dev + iter0 (ptmi_device_iter), else nullptr
    int nsteps;
    int d, nt, W, ntg, temp0, walker0;
    int w_host, w_scam, w_am, w_de, de_on, de_size, de_head;
    int cov_update, tskip, per_walker, logp_kind, ngroups;
    int am_row0, swap_last;      // iter0 % cov_update; the last step of the launch is a swap iteration
    int am_epl;                  // row format of the AM buffer (am_pos): 0 = parameter order, 25 = the lanes' order of the exact shape
    int lanes;                   // lanes per chain of the handle's kernel shape (G): the DE rows' piece order depends on it (ptmi_de_row_stride)
    int de_ld;                   // doubles per row of the DE buffer (ptmi_de_row_stride: 8 * ceil(EPL / 2) with 4 lanes per chain, else ndim)
    int pick_walker;             // pick_mode WALKER: the cycle entry comes from the stream of the walker's rank 0
    int lds_u;                   // staged kernels: the block's Ut is copied to LDS (else read from global)
    int amq_off;                 // staged full kernels: offset (doubles, even) of the AM queue in the block's LDS (mh_steps_kernel)
    int amq_on;                  // ... and whether this launch takes its AM increments from the queue (launch_mh_k)
    int box_off;                 // box prior: offset (doubles, even) of the bounds table in the block's LDS, or -1: bounds read from global
    int tab_off;                 // step kernels: offset (doubles, even) of the block's LDS copy of the draw tables (ptmi_tables.h), or -1: read from global
    // AM increments computed ahead of the launch by am_gemm_kernel (ptmi_abi.hip; the 16- and 64-lane shapes): increment j of the
    // chain's AM picks of this launch is am_inc[(am_base[chain] + j) * d ...]; nullptr: the kernel computes its own
    const double *am_inc;
    const long long *am_base;
    long long *am_next;          // split path (ptmi_split.hip): the chains' cursors into am_inc, advanced by the row kernel with every AM pick
    // 16- / 64-lane shapes, ONE table for the launch: the library's zero-padded copy of Ut with rows of ut_pad_ld = G * EPL doubles
    // (ut_pad_kernel, made ahead of the launch), so that a step reads its direction with unconditional loads; nullptr: none
    const double *UtPad;
    int ut_pad_ld;
    const double *ut_absmax;     // max |U| over that table (ut_pad_kernel), for the box prior's fast path (mh_steps_kernel BOXFAST)
    int umax_off;                // persistent SCAM kernel, box prior: where in the block's LDS (doubles) the same maximum is kept
    // gradient jumps (ptmi_gj.inc.h)
    int w_nuts, w_hmc, gj_nburn, hmc_min, hmc_max, nuts_maxdepth;
    double hmc_eps, nuts_delta;
    const double *gj_tab;        // [3][d][d] backward, forward, gradient tables
    int gj_diag;                 // the whitening tables are diagonal (L = cholesky of a diagonal covariance): a product is d multiplications (oracle: tab_vec)
    double *gj;                  // [W][T][8] per-rank jump state
    double *gj_scr, *gj_scal;    // scratch of the tree build: [slot][e][chain][lane], [level][scalar][chain]
    int gj_stack_off, gj_lds_levels;   // tree stack: offset (doubles) in the block's LDS and how many of the lowest heights live there
    const int32_t *gj_order;     // chain handled by each chain slot of the launch (chains of similar NUTS step size share a wave), -1: none
    int gj_nslots;               // chain slots of the launch when gj_order is set (>= the chains: the longest trees' chains have a wave to themselves)
    const u64 *rp_draws;         // TEST HOOK (ptmi_test_replay): propose_kernel takes P0, Q0, Q1 and the SCAM normal's bits of every chain from here
};

// gradient jumps (ptmi_gj.inc.h): per-rank state, Philox slots, layout of a chain's tree scratch
enum { GJ_EPS = 0, GJ_MU = 1, GJ_HBAR = 2, GJ_EPSBAR = 3, GJ_NITER = 4, GJ_HITER = 5, GJ_HAVE_EPS = 6, GJ_NLEAP = 7 /* leapfrogs so far */, GJ_NSTATE = 8 };
constexpr int GJ_BUCKETS = 128;         // step-size classes of the launch order (ptmi_abi.hip gj_order_*)
constexpr u32 SLOT_GJ = 0x2000000u;    // + 4096 * (momenta draw of the call) + direction
constexpr u32 SLOT_GJS = 0x3000000u;   // + scalar draw of the call
// vector slots of a chain's scratch: the two ends and the sample of the outer loop, then 4 per tree level
enum { GJV_TM = 0, GJV_RM = 1, GJV_GM = 2, GJV_TP = 3, GJV_RP = 4, GJV_GP = 5, GJV_SAMPLE = 6, GJV_TOP = 7 };
enum { GJL_FAR_T = 0, GJL_FAR_R = 1, GJL_CAND_T = 2, GJL_CAND_G = 3, GJL_VECS = 4 };
enum { GJS_LOGP = 0, GJS_N = 1, GJS_ALPHA = 2, GJS_NALPHA = 3, GJS_SCALARS = 4 };

template <int G>
__device__ __forceinline__ double group_bcast_lane(double v, int src)
{
    // lane `src` (0..G-1) of the caller's group
    const int lane = (int)(threadIdx.x & 63);
    return __shfl(v, (lane & ~(G - 1)) + src, 64);
}

// ------------------------------------------------------------------- engine
struct ptmi_engine {
    ptmi_config cfg;
    ptmi_buffers buf;
    hipStream_t stream;
    double *d_ladder, *d_temps, *d_beta, *d_loglpar, *d_logppar, *d_gmask, *d_gcn, *d_gdiv;
    int32_t *d_gsize;
    void *d_pre;        // [ntg][W] records of the swap (log uniform, likelihood, own-likelihood quotients, row: swap_prepare_kernel)
    int32_t *d_hop;     // set when a row of the last sweep travels beyond a neighbouring block (by the sweep's write-out, else by ptmi_exchange_pack)
    bool hop_from_sweep;
    int32_t *h_hop;     // pinned host copy of the flag, requested behind ptmi_exchange_pack; hop_ev marks its arrival
    hipEvent_t hop_ev;
    bool hop_pending;
    int32_t *d_xint;    // exchange scratch: inv[W][ntg], newslot[W][T], arr_slot[nranks][W], lv_slot[2][W], lv_rank[2][W], err[1]
    double *d_gj_tab, *d_gj_scr, *d_gj_scal;   // gradient jumps: whitening tables, tree scratch
    int gj_diag;                               // ... and whether the tables are diagonal (decided at ptmi_create)
    int gj_solo;                 // chains with a wave of their own in the gradient-jump launches (gj_order_fill_kernel)
    int32_t *d_gj_order, *d_gj_bucket;         // launch order of the chains ([nch]) and its counting-sort scratch ([3][GJ_BUCKETS])
    // AM increments ahead of the launch (large ndim): events of a piece of the launch, their increments [am_cap][ndim]
    void *d_am_ev;
    int32_t *d_am_count;
    long long *d_am_base;
    double *d_am_inc;
    int32_t *d_am_grp, *d_am_perm;     // parameter groups: picks per key (group, or walker x group) | cursors | scan scratch; the events' indices key by key
    long long *d_am_kbase;             // where each key's list starts in d_am_perm (+ the end)
    long long am_cap;
    int am_piece;       // steps per piece (0: the path is off for this engine)
    double *d_pool_part, *d_pool_T;  // pooled covariance: the slabs' partial sums [nslab][d][d+1] and their total [d][d+1] (column d: the column sums)
    int G, EPL;
    int de_on, de_head;
    int last_variant;   // PTMI_VAR_* flags of the most recent fused-MH launch (ptmi_last_mh_variant)
    hipEvent_t ev0, ev1;
    hipStream_t side;            // pooled statistics at ndim > 111: the diagonal macro tiles run beside the off-diagonal ones
    hipEvent_t side_go, side_done;
    double *d_utpad;             // zero-padded copy of the pooled eigenvector table for the 16- / 64-lane shapes (KArgs::UtPad)
    void *d_sy_scr;              // ptmi_eig_sytrd: the working matrix, d / e / tau, the eigenvectors, the exchange vectors and the barrier word
    void *dc_plan;               // ... the divide-and-conquer solver's tree and scratch (DcPlan, ptmi_abi.hip)
    void *sy_lib;                // ... and the ROCm library's entry points (SyLib, ptmi_abi.hip; PTMI_SYTRD_LIB=1 only)
    int32_t *h_sy_info;          // pinned: the divide-and-conquer solver's convergence word of the last factorization that has finished
    void *d_qlg_scr;             // ptmi_eig_ql with parameter groups: a group's packed matrices, their eigenvectors and eigenvalues
    int32_t *gsize_host;         // [Ng] parameters per group (host copy of d_gsize)
    void *d_ql_scr;              // ptmi_eig_ql with many matrices: the transformations, tridiagonal matrices and recorded rotations (QlScratch)
    void *d_rle_ent;             // pooled statistics over run-length-compacted rows: the stored rows of each slab, 16 bytes each [nrows]
                                 // (PoolEnt, ptmi_abi.hip: the row inside its slab, the square root of its run length) ...
    int32_t *d_rle_cnt;          // ... and how many each slab has [nslab]
    const double *rp_swap_u;     // TEST HOOK (ptmi_test_replay): the swap's uniforms [W][ntemps_global - 1] instead of the Philox ones
    const u64 *rp_draws;         // TEST HOOK: see KArgs
    long long *d_am_next;        // split path: the chains' cursors into d_am_inc (KArgs::am_next)
    int split_am_piece;          // split path: iterations ptmi_split_am_prepare can cover at once (0: AM cycles go through the shape kernels)
    long long split_am_lo, split_am_hi;   // ... and the iterations [lo, hi) the prepared increments cover
    long long *d_iter;           // split path: the iteration counter in device memory (ptmi_set_device_iter) ...
    int dev_iter;                // ... and whether the split calls' `iter` arguments are offsets from it (ptmi_device_iter)
    int q_cur;                   // split path: the proposal buffer (0 = Q, 1 = Q2) that holds the current proposals (ptmi_proposals)
};


// split path on contiguous rows (ptmi_split.hip): does the handle's configuration run there; mode 0 propose(iter0), 1 accept(iter0),
// 2 accept(iter0) + propose(iter0 + 1) in one launch
bool ptmi_split_rows_ok(const ptmi_engine *h);
int ptmi_split_rows(ptmi_engine *h, const KArgs &a, int mode);
int ptmi_rows_iso(ptmi_engine *h, const double *rows, long long n, double *out);
int ptmi_set_iter_device(ptmi_engine *h, long long *p, long long v);
extern "C" int ptmi_split_am_prepare(ptmi_handle h, int64_t iter0, int32_t nsteps);

// The per-chain kernels are templates over the shape (lanes per chain G, register slots per lane EPL).  Each shape
// and likelihood family is compiled in its own translation unit (ptmi_shape.hip with -DPTMI_G -DPTMI_E -DPTMI_L) so
// the build runs in parallel;
// this is the entry point a shape unit exports.
enum { PTMI_OP_MH = 0, PTMI_OP_EVAL = 1, PTMI_OP_PROPOSE = 2, PTMI_OP_ACCEPT = 3, PTMI_OP_MH_GJ = 4 };
// jump types the SCAM / AM / DE kernels count (the gradient jumps have their own fused kernel, ptmi_gj.inc.h)
enum { PTMI_J_FUSED = 3 };
// exact shapes serve ndim == G * EPL only: no slot of any lane needs a bounds check (and odd EPL: the paired LDS rows)
constexpr bool ptmi_shape_exact(int G, int EPL) { return G == 4 && EPL == 25; }
typedef int (*ptmi_shape_fn)(int op, ptmi_engine *h, KArgs &a, int grid, bool full);
#define PTMI_SHAPE_LIST(X) X(4, 2) X(4, 5) X(4, 8) X(4, 14) X(4, 20) X(4, 25) X(4, 26) X(16, 7) X(16, 13) X(16, 26) X(64, 8) X(64, 16) X(64, 32)
// one unit per (shape, likelihood family); the split-path kernels live in the family-0 unit
#define PTMI_DECLARE_SHAPE(G_, E_)                                                        \
    int ptmi_shape_##G_##_##E_##_0(int op, ptmi_engine *h, KArgs &a, int grid, bool full); \
    int ptmi_shape_##G_##_##E_##_1(int op, ptmi_engine *h, KArgs &a, int grid, bool full); \
    int ptmi_shape_##G_##_##E_##_2(int op, ptmi_engine *h, KArgs &a, int grid, bool full);
PTMI_SHAPE_LIST(PTMI_DECLARE_SHAPE)
// family 3 (PTMI_LOGL_INTERVAL) is built for the gradient-jump shapes only (at most 8 slots per lane; _build.py)
#define PTMI_GJ_SHAPE_LIST(X) X(4, 2) X(4, 5) X(4, 8) X(16, 7) X(64, 8)
#define PTMI_DECLARE_SHAPE3(G_, E_) int ptmi_shape_##G_##_##E_##_3(int op, ptmi_engine *h, KArgs &a, int grid, bool full);
PTMI_GJ_SHAPE_LIST(PTMI_DECLARE_SHAPE3)
