// ptmi_shape.hip -- one kernel shape per translation unit: compile with -DPTMI_G=<lanes> -DPTMI_E=<slots> -DPTMI_L=<family>
// -DPTMI_PART=<0|1>.  Part 1 holds the step kernels of cycles with AM / DE entries (FULL) and nothing else, part 0 everything
// else of the shape: the two parts are built with different scheduler options (_build.py).
#include "ptmi_mh.inc.h"
#include "ptmi_gj.inc.h"

#if !defined(PTMI_G) || !defined(PTMI_E) || !defined(PTMI_L)
#error "compile with -DPTMI_G=<lanes per chain> -DPTMI_E=<register slots per lane> -DPTMI_L=<likelihood family>"
#endif
#ifndef PTMI_PART
#error "compile with -DPTMI_PART=<0|1>"
#endif
#define PTMI_CAT_(a, b, c) ptmi_shape_##a##_##b##_##c
#define PTMI_CAT(a, b, c) PTMI_CAT_(a, b, c)
#define PTMI_CATF_(a, b, c) ptmi_shape_full_##a##_##b##_##c
#define PTMI_CATF(a, b, c) PTMI_CATF_(a, b, c)

#if PTMI_PART == 1
int PTMI_CATF(PTMI_G, PTMI_E, PTMI_L)(ptmi_engine *h, KArgs &a, int grid) { return launch_mh_k<PTMI_G, PTMI_E, PTMI_L, true>(h, a, grid); }
#else
int PTMI_CATF(PTMI_G, PTMI_E, PTMI_L)(ptmi_engine *h, KArgs &a, int grid);

int PTMI_CAT(PTMI_G, PTMI_E, PTMI_L)(int op, ptmi_engine *h, KArgs &a, int grid, bool full)
{
    constexpr int G = PTMI_G, E = PTMI_E, L = PTMI_L;
    switch (op) {
    case PTMI_OP_MH: return full ? PTMI_CATF(PTMI_G, PTMI_E, PTMI_L)(h, a, grid) : launch_mh_k<G, E, L, false>(h, a, grid);
    case PTMI_OP_EVAL: hipLaunchKernelGGL((eval_state_kernel<G, E, L>), dim3(grid), dim3(256), 0, h->stream, a); return PTMI_OK;
#if PTMI_L == 0
    case PTMI_OP_PROPOSE:
        if (h->cfg.ngroups > 1) hipLaunchKernelGGL((propose_kernel<G, E, true>), dim3(grid), dim3(256), 0, h->stream, a);
        else hipLaunchKernelGGL((propose_kernel<G, E, false>), dim3(grid), dim3(256), 0, h->stream, a);
        return PTMI_OK;
    case PTMI_OP_ACCEPT: hipLaunchKernelGGL((accept_kernel<G, E>), dim3(grid), dim3(256), 0, h->stream, a); return PTMI_OK;
#endif
    case PTMI_OP_MH_GJ:
        if constexpr (E <= 8) {                 // the tree build keeps seven chain vectors in registers
            const long long nch = a.gj_order ? (long long)a.gj_nslots : (long long)h->cfg.nwalkers * h->cfg.ntemps;     // chain slots of the launch
            const int cpb = GJ_BLOCK / G;
            // LDS of a block (one wave).  4-lane shapes (GradJumpWide): whitening tables with rows of 4 E | the tree stack,
            // one level per height | the exchange area of the layout change (64 doubles) | box bounds of the 4-lane test.
            // Wider shapes: lowest levels of the tree stack | box bounds.
            size_t off = 0;
            bool pair = false, w16 = false;
            const size_t box = h->cfg.logp_kind == PTMI_LOGP_BOX ? (size_t)box_table_doubles(G, E) : 0;
            static const char *lv = getenv("PTMI_GJ_LDS_LEVELS");       // measurement / test switch: same results for any value
            if constexpr (G == 4) {
                off = a.gj_diag ? (size_t)(3 * 4 * E) : (size_t)gjw_table_doubles(E);    // diagonal whitening: the three diagonals only
                a.gj_stack_off = (int)off;
                static const char *lvd = getenv("PTMI_GJ_LDS_DEFAULT");                  // measurement switch: heights kept in LDS by default
                const int lmax = lvd ? atoi(lvd) : 11;
                int levels = h->cfg.nuts_maxdepth + 1 < lmax ? h->cfg.nuts_maxdepth + 1 : lmax;    // heights 0..10 in LDS, the rest in global scratch
                if (lv) levels = atoi(lv) < levels ? atoi(lv) : levels;
                a.gj_lds_levels = levels;
                // two jumps at a time, a half-wave each (GradJumpPair): diagonal whitening, no dense products; PTMI_GJ_NOPAIR: the
                // one-chain-per-wave layout (a measurement / test switch, same results)
                pair = a.gj_diag && L != PTMI_LOGL_DENSE && getenv("PTMI_GJ_NOPAIR") == nullptr;
                off += (size_t)(pair ? 2 : 1) * a.gj_lds_levels * gjw_level_doubles(E) + (pair ? 72 + 2 * GJ_BLOCK : 64);       // pair: + the 16 chains' step-size states
            } else if (G == 16 && a.d <= 64 && getenv("PTMI_GJ_NOWIDE16") == nullptr) {
                // the 16-lane shape at ndim <= 64: a gradient jump takes the whole wave (GradJumpWide<16, L, 16>, one element per lane);
                // PTMI_GJ_NOWIDE16: the per-chain layout (a measurement / test switch, same results)
                w16 = true;
                off = a.gj_diag ? (size_t)(3 * 64) : 0;                     // the three diagonals (full tables stay in global memory)
                a.gj_stack_off = (int)off;
                int levels = h->cfg.nuts_maxdepth + 1 < 11 ? h->cfg.nuts_maxdepth + 1 : 11;
                if (lv) levels = atoi(lv) < levels ? atoi(lv) : levels;
                a.gj_lds_levels = levels;
                off += (size_t)levels * gjw_level_doubles(16) + 64;
            } else {
                const size_t budget = 40 * 1024 / sizeof(double);           // one wave per SIMD (register count): a quarter of the CU's LDS each
                int levels = box < budget ? (int)((budget - box) / gj_level_doubles(E)) : 0;
                if (levels > h->cfg.nuts_maxdepth + 1) levels = h->cfg.nuts_maxdepth + 1;
                if (lv) levels = atoi(lv) < levels ? atoi(lv) : levels;
                a.gj_stack_off = 0;
                a.gj_lds_levels = levels;
                off = (size_t)levels * gj_level_doubles(E);
            }
            off = (off + 1) & ~(size_t)1;
            a.box_off = box ? (int)off : -1;
            off += box;
            if constexpr (G == 4 && L != PTMI_LOGL_DENSE) {
                if (pair) {
                    if (sizeof(double) * off > 64 * 1024) {
                        hipError_t e = hipFuncSetAttribute((const void *)mh_steps_gj_kernel<G, E, L, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * off));
                        if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", sizeof(double) * off, hipGetErrorString(e));
                    }
                    hipLaunchKernelGGL((mh_steps_gj_kernel<G, E, L, true>), dim3((unsigned)((nch + cpb - 1) / cpb)), dim3(GJ_BLOCK), sizeof(double) * off, h->stream, a);
                    return PTMI_OK;
                }
            }
            if constexpr (G == 16) {
                if (w16) {
                    if (a.gj_diag) hipLaunchKernelGGL((mh_steps_gj_kernel<G, E, L, false, 1>), dim3((unsigned)((nch + cpb - 1) / cpb)), dim3(GJ_BLOCK), sizeof(double) * off, h->stream, a);
                    else hipLaunchKernelGGL((mh_steps_gj_kernel<G, E, L, false, 2>), dim3((unsigned)((nch + cpb - 1) / cpb)), dim3(GJ_BLOCK), sizeof(double) * off, h->stream, a);
                    return PTMI_OK;
                }
            }
            if (sizeof(double) * off > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void *)mh_steps_gj_kernel<G, E, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * off));
                if (e != hipSuccess) return fail(PTMI_EHIP, "hipFuncSetAttribute(%zu B of LDS): %s", sizeof(double) * off, hipGetErrorString(e));
            }
            hipLaunchKernelGGL((mh_steps_gj_kernel<G, E, L>), dim3((unsigned)((nch + cpb - 1) / cpb)), dim3(GJ_BLOCK),
                               sizeof(double) * off, h->stream, a);
            return PTMI_OK;
        } else {
            return fail(PTMI_EUNSUPPORTED, "gradient jumps are built for kernel shapes with at most 8 register slots per lane");
        }
    }
    return fail(PTMI_EINVAL, "unknown shape op %d", op);
}
#endif
