// ptmi_shape.hip -- one kernel shape per translation unit: compile with -DPTMI_G=<lanes> -DPTMI_E=<slots>.
#include "ptmi_mh.inc.h"
#include "ptmi_gj.inc.h"

#if !defined(PTMI_G) || !defined(PTMI_E) || !defined(PTMI_L)
#error "compile with -DPTMI_G=<lanes per chain> -DPTMI_E=<register slots per lane> -DPTMI_L=<likelihood family>"
#endif
#define PTMI_CAT_(a, b, c) ptmi_shape_##a##_##b##_##c
#define PTMI_CAT(a, b, c) PTMI_CAT_(a, b, c)

int PTMI_CAT(PTMI_G, PTMI_E, PTMI_L)(int op, ptmi_engine *h, KArgs &a, int grid, bool full)
{
    constexpr int G = PTMI_G, E = PTMI_E, L = PTMI_L;
    switch (op) {
    case PTMI_OP_MH: return launch_mh_l<G, E, L>(h, a, grid, full);
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
            const long long nch = (long long)h->cfg.nwalkers * h->cfg.ntemps;
            const int cpb = GJ_BLOCK / G;
            hipLaunchKernelGGL((mh_steps_gj_kernel<G, E, L>), dim3((unsigned)((nch + cpb - 1) / cpb)), dim3(GJ_BLOCK),
                               G == 4 ? sizeof(double) * 3 * (size_t)h->cfg.ndim * h->cfg.ndim : 0, h->stream, a);
            return PTMI_OK;
        } else {
            return fail(PTMI_EUNSUPPORTED, "gradient jumps are built for kernel shapes with at most 8 register slots per lane");
        }
    }
    return fail(PTMI_EINVAL, "unknown shape op %d", op);
}
