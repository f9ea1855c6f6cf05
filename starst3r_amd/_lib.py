"""ctypes binding of libst3r_hip.so (include/st3r.h).

There is NO fallback: if the shared library is missing or a call fails this module raises.
The product path never routes through oracle/ or any CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libst3r_hip.so")

_lib = None

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
u32, u64 = C.c_uint32, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/st3r.h
SIGNATURES = {
    "st3r_version": [],
    "st3r_last_error": [],
    "st3r_ctx_create": [i32, C.POINTER(vp)],
    "st3r_ctx_destroy": [vp],
    "st3r_ctx_arena_bytes": [vp],
    "st3r_ctx_peek": [vp, vp, i32, vp, i64],
    "st3r_ctx_set_profiling": [vp, i32],
    "st3r_ctx_set_debug": [vp, i32],
    "st3r_ctx_settle": [vp],
    "st3r_ctx_release_scratch": [vp],
    "st3r_ctx_get_stage_ms": [vp, C.POINTER(f64), C.POINTER(i64)],
    "st3r_stage_name": [i32],
    "st3r_gs_project_sh": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32,
                           vp, vp, vp],
    "st3r_gs_isect_scan": [vp, vp, i64, vp, vp, C.POINTER(i64)],
    "st3r_gs_isect_emit": [vp, vp, i32, i32, vp, vp, i32, i32, i32, i64, vp, vp],
    "st3r_gs_sort": [vp, vp, i64, i32, vp, vp, vp, vp],
    "st3r_radix_sort_pairs": [vp, vp, i32, i64, i32, i32, vp, vp, vp, vp],
    "st3r_gs_offsets": [vp, vp, i64, vp, i32, i32, i32, vp],
    "st3r_gs_blend_fwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp, vp],
    "st3r_gs_blend_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp],
    "st3r_gs_project_sh_bwd": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, f32, vp, vp, f32, f32,
                               f32, vp],
    "st3r_loss_l1_ssim": [vp, vp, i32, i32, i32, vp, vp, f32, f32, vp, vp],
    "st3r_loss_gt_moments": [vp, vp, i32, i32, i32, vp, vp],
    "st3r_ctx_set_gt_moments": [vp, vp, vp, i32, i32, i32],
    "st3r_adam_step": [vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, f64, f64, f64, f64, i32],
    "st3r_adam_step_range": [vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, f64, f64, f64, f64, i32, i64, i64, vp],
    "st3r_params_from_stage": [vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, i64, i64, i64],
    "st3r_gs_train_fwd_bwd": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, f32, f32, f32, vp,
                              vp, C.POINTER(i64)],
    "st3r_align_run": [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp,
                       vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, f32, i32, f32, i32, f32, vp, vp, vp, vp, vp, vp, i64,
                       vp, vp, vp],
    "st3r_align_run_opts": [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp,
                            vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, f32, i32, f32, i32, f32, vp, vp, vp, vp, vp, vp,
                            i64, vp, vp, vp, vp, f32, f32, f32, i32, vp, vp, vp, i64],
    "st3r_nn_dot_argmax": [vp, vp, vp, i32, vp, i32, i32, vp, vp],
    "st3r_mcmc_relocate": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, f32, u64, u32, C.POINTER(i64)],
    "st3r_mcmc_add": [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, f32, u64, u32],
    "st3r_mcmc_noise": [vp, vp, i32, vp, vp, vp, vp, f32, u64, u32],
    "st3r_mcmc_noise_rows": [vp, vp, i32, i64, vp, vp, vp, vp, f32, u64, u32],
    "st3r_comm_unique_id": [C.c_char_p],
    "st3r_comm_init": [vp, i32, i32, C.c_char_p],
    "st3r_comm_attach": [vp, vp, i32, i32],
    "st3r_comm_destroy": [vp],
    "st3r_comm_world": [vp, C.POINTER(i32), C.POINTER(i32)],
    "st3r_comm_set_exchange": [vp, i32],
    "st3r_comm_get_exchange": [vp, C.POINTER(i32)],
    "st3r_comm_allgather_pieces": [vp, vp, vp, i64],
    "st3r_grad_allreduce": [vp, vp, vp, i64],
    "st3r_gs_train_step": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp,
                           vp, f64, f64, f64, f64, i32, vp, C.POINTER(i64)],
    "st3r_recip_nn_seed_count": [i32, i32, i32],
    "st3r_recip_nn": [vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp],
    "st3r_dense_unproject": [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "st3r_dense_clean": [vp, vp, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp],
    "st3r_canon_view": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "st3r_focal_weiszfeld": [vp, vp, i32, i32, vp, f32, f32, f32, f32, vp],
    "st3r_focal_weiszfeld_batch": [vp, vp, i32, i32, i32, vp, f32, f32, f32, f32, vp],
    "st3r_anchor_offsets": [vp, vp, i64, i32, i32, i32, vp, vp, vp, vp],
    "st3r_gs_raster_train": [vp, vp, i32, i32, vp, vp, i32, i32, f32, vp, vp, C.POINTER(i64)],
    "st3r_gs_render": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, C.POINTER(i64)],
}
_RESTYPES = {"st3r_last_error": C.c_char_p, "st3r_stage_name": C.c_char_p, "st3r_ctx_arena_bytes": i64}


class St3rError(RuntimeError):
    pass


def lib():
    """Load the HIP library (once).  Raises if it is not built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise St3rError(
                f"{LIB_PATH} is missing: build it with `python -m starst3r_amd.build` "
                "(there is no CPU fallback for the hot path)")
        # torch bundles its own HIP runtime; it must be the one already resident when our library's
        # libamdhip64 dependency is resolved, otherwise two runtimes end up in one process
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().st3r_last_error()
        msg = msg.decode() if msg else ""
        if rc == -1:
            raise ValueError(f"st3r: invalid argument: {msg}")
        err = St3rError(f"st3r error {rc}: {msg}")
        err.code = rc
        raise err
