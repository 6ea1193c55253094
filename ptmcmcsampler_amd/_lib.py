"""ctypes binding of libptmi.so (the C ABI in include/ptmi.h).

There is no CPU fallback: if the library is missing or no HIP device is visible the
engine raises instead of computing anything on the host."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PTMI_LIB: another build of the same ABI (A/B measurements of a kernel change); the default is the in-tree library
SO = os.environ.get("PTMI_LIB") or os.path.join(HERE, "libptmi.so")

LOGL = {"iso": 0, "dense": 1, "curved": 2, "interval": 3}
LOGP = {"flat": 0, "box": 1}
J_SCAM, J_AM, J_DE, J_NUTS, J_HMC, J_NTYPES = 0, 1, 2, 3, 4, 5
GJ_NSTATE, GJ_EPSBAR = 8, 3
GJ_NITER, GJ_HITER, GJ_NLEAP = 4, 5, 7          # per-rank jump state (csrc/ptmi_common.h): NUTS calls, HMC calls, leapfrogs taken so far
JUMP_NAMES = ("covarianceJumpProposalSCAM", "covarianceJumpProposalAM", "DEJump")

_dp = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "ndim", "ntemps", "nwalkers", "ntemps_global", "temp0", "walker0", "logl_kind", "logp_kind",
        "w_host", "w_scam", "w_am", "w_de", "de_size", "cov_update", "tskip", "cov_per_walker", "device", "ngroups", "swap_mode",
        "w_nuts", "w_hmc", "gj_nburn", "hmc_min", "hmc_max", "nuts_maxdepth", "pick_mode")] + [
        ("hmc_eps", C.c_double), ("nuts_delta", C.c_double), ("seed", C.c_uint64), ("stream", C.c_void_p), ("ladder", _dp), ("temps_mh", _dp),
        ("logl_par", _dp), ("logl_par_len", C.c_int64), ("logp_par", _dp), ("logp_par_len", C.c_int64),
        ("group_size", C.POINTER(C.c_int32)), ("group_mask", _dp), ("gj_tab", _dp)]


VAR_STAGED, VAR_FULL, VAR_LDS_UT, VAR_GROUPS, VAR_GRADJUMP, VAR_UNIFORM, VAR_AMQ, VAR_LDS_BOX, VAR_LDS_DRAWT, VAR_DENSE_SCAM, VAR_PERSISTENT, VAR_PC, VAR_UTPAD = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 1 << 28   # ptmi_last_mh_variant
SWAP_MODES = {"sweep": 0, "oddeven": 1}
PICK_MODES = {"chain": 0, "walker": 1}        # PTMI_PICK_CHAIN, PTMI_PICK_WALKER      # PTMI_SWAP_SWEEP, PTMI_SWAP_ODDEVEN

BUFFER_FIELDS = ("X", "lnL", "lp", "temp_of", "slot_of", "Ut", "S", "DE", "AM", "nacc", "jstat", "nswap",
                 "mu", "M2", "cov", "Q", "qaux", "AMaux", "gj", "AMflag", "Q2", "sloc")
AMROW_NEW, AMROW_KEY = 1, 2                    # flag word of an AM row (include/ptmi.h ptmi_buffers.AMflag)


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in BUFFER_FIELDS]


# every symbol include/ptmi.h declares
SYMBOLS = (
    "ptmi_last_error", "ptmi_version", "ptmi_device_count", "ptmi_lanes_for", "ptmi_lanes_for_grad", "ptmi_temperature_ladder", "ptmi_de_row_stride", "ptmi_am_row_format", "ptmi_create", "ptmi_destroy",
    "ptmi_sync", "ptmi_eval_state", "ptmi_set_de_active", "ptmi_mh_steps", "ptmi_last_mh_variant", "ptmi_swap", "ptmi_swap_gather_lnl",
    "ptmi_swap_sweep", "ptmi_swap_sweep_blocks", "ptmi_exchange_pack", "ptmi_exchange_apply", "ptmi_exchange_status", "ptmi_exchange_multihop",
    "ptmi_swap_write_am", "ptmi_update_cov", "ptmi_update_cov_on", "ptmi_set_am_buffers", "ptmi_eig_jacobi", "ptmi_eig_ql", "ptmi_eig_ql_from", "ptmi_eig_sytrd", "ptmi_eig_sytrd_from", "ptmi_eig_sytrd_info", "ptmi_update_de", "ptmi_set_de_head", "ptmi_propose", "ptmi_accept", "ptmi_accept_propose", "ptmi_proposals", "ptmi_rows_logl", "ptmi_split_am_piece", "ptmi_split_am_prepare", "ptmi_set_stream", "ptmi_set_proposals", "ptmi_device_iter", "ptmi_set_device_iter",
    "ptmi_am_flags_ok", "ptmi_am_expand", "ptmi_test_replay",
    "ptmi_selftest_math", "ptmi_selftest_philox", "ptmi_malloc", "ptmi_free", "ptmi_memcpy_h2d", "ptmi_memcpy_d2h",
    "ptmi_memset", "ptmi_timer_start", "ptmi_timer_stop_ms",
)

_lib = None


class PtmiError(RuntimeError):
    pass


def load():
    """Load libptmi.so; raises PtmiError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise PtmiError("libptmi.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `python -m ptmcmcsampler_amd._build`; there is no CPU fallback." % SO)
    # torch bundles its own libamdhip64 (same soname as /opt/rocm's): it must be the first HIP
    # runtime in the process, otherwise two copies get loaded and torch loses the GPU
    try:
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = C.CDLL(SO)
    H = C.c_void_p
    L.ptmi_last_error.restype = C.c_char_p
    L.ptmi_device_count.argtypes = [C.POINTER(C.c_int)]
    L.ptmi_lanes_for.argtypes = [C.c_int]
    L.ptmi_lanes_for_grad.argtypes = [C.c_int]
    L.ptmi_temperature_ladder.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp]
    L.ptmi_de_row_stride.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ptmi_am_row_format.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.ptmi_create.argtypes = [C.POINTER(Config), C.POINTER(Buffers), C.POINTER(H)]
    for n in ("ptmi_destroy", "ptmi_sync", "ptmi_eval_state", "ptmi_update_de", "ptmi_timer_start", "ptmi_eig_jacobi", "ptmi_eig_ql"):
        getattr(L, n).argtypes = [H]
    L.ptmi_eig_sytrd.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptmi_update_cov_on.argtypes = [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptmi_set_am_buffers.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptmi_eig_sytrd_from.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptmi_eig_ql_from.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptmi_eig_sytrd_info.argtypes = [H, C.POINTER(C.c_int32)]
    L.ptmi_set_de_active.argtypes = [H, C.c_int]
    L.ptmi_set_de_head.argtypes = [H, C.c_int32]
    L.ptmi_mh_steps.argtypes = [H, C.c_int64, C.c_int32]
    L.ptmi_last_mh_variant.argtypes = [H, C.POINTER(C.c_int32)]
    L.ptmi_swap.argtypes = [H, C.c_int64]
    L.ptmi_swap_gather_lnl.argtypes = [H, C.c_void_p]
    L.ptmi_swap_sweep.argtypes = [H, C.c_int64, C.c_void_p, C.c_void_p]
    L.ptmi_swap_write_am.argtypes = [H, C.c_int64]
    L.ptmi_swap_sweep_blocks.argtypes = [H, C.c_int64, C.c_void_p, C.c_void_p]
    L.ptmi_exchange_pack.argtypes = [H, C.c_void_p, C.c_void_p]
    L.ptmi_exchange_apply.argtypes = [H, C.c_void_p]
    L.ptmi_exchange_status.argtypes = [H, C.POINTER(C.c_int32)]
    L.ptmi_exchange_multihop.argtypes = [H, C.POINTER(C.c_int32)]
    L.ptmi_update_cov.argtypes = [H, C.c_int64]
    L.ptmi_am_flags_ok.argtypes = [C.POINTER(Config)]
    L.ptmi_am_expand.argtypes = [H, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
    L.ptmi_test_replay.argtypes = [H, C.c_void_p, C.c_void_p]
    L.ptmi_propose.argtypes = [H, C.c_int64]
    L.ptmi_accept.argtypes = [H, C.c_int64, C.c_void_p, C.c_void_p]
    L.ptmi_accept_propose.argtypes = [H, C.c_int64, C.c_void_p, C.c_void_p]
    L.ptmi_proposals.argtypes = [H, C.POINTER(C.c_void_p)]
    L.ptmi_rows_logl.argtypes = [H, C.c_void_p, C.c_int64, C.c_void_p]
    L.ptmi_split_am_piece.argtypes = [H, C.POINTER(C.c_int32)]
    L.ptmi_split_am_prepare.argtypes = [H, C.c_int64, C.c_int32]
    L.ptmi_set_stream.argtypes = [H, C.c_void_p]
    L.ptmi_set_proposals.argtypes = [H, C.c_int32]
    L.ptmi_device_iter.argtypes = [H, C.c_int32]
    L.ptmi_set_device_iter.argtypes = [H, C.c_int64]
    L.ptmi_selftest_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.ptmi_selftest_philox.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    L.ptmi_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.ptmi_free.argtypes = [C.c_void_p]
    L.ptmi_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ptmi_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ptmi_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.ptmi_timer_stop_ms.argtypes = [H, C.POINTER(C.c_double)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PtmiError("libptmi error %d: %s" % (rc, load().ptmi_last_error().decode()))


def device_count():
    n = C.c_int(0)
    rc = load().ptmi_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def lanes_for(ndim, grad=False):
    """Lanes that share one chain (ptmi_lanes_for / ptmi_lanes_for_grad)."""
    if grad:
        return 4 if ndim <= 32 else (16 if ndim <= 112 else (64 if ndim <= 512 else 0))
    return 4 if ndim <= 104 else (16 if ndim <= 416 else 64)
