"""ctypes binding of include/isdf_hip.h (the C-ABI drop-in boundary).

Loads the in-tree `libisdf_hip.so` built by `isdf_amd/build.py`.  There is NO
fallback: if the library is missing or a call fails, an exception is raised --
the product path never routes around the HIP kernels.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ISDF_HIP_LIB: development override (A/B variants and the instrumented build of tools/build_variants.py)
LIB_PATH = os.environ.get("ISDF_HIP_LIB") or os.path.join(HERE, "libisdf_hip.so")

ABI_VERSION = 7

# every symbol include/isdf_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "isdf_abi_version", "isdf_error_string", "isdf_check_net", "isdf_param_count", "isdf_shadow_bytes",
    "isdf_workspace_bytes", "isdf_reduce_floats", "isdf_reduce_split_floats", "isdf_sample_scan_bytes", "isdf_pack_weights",
    "isdf_sample_rays",
    "isdf_sdf_eval", "isdf_train_step", "isdf_train_step_adamw", "isdf_train_step_finish", "isdf_bounds_pc",
    "isdf_frame_avg", "isdf_adamw", "isdf_estimate_normals", "isdf_render_depth", "isdf_allreduce_sum_f32",
]

LS_SDF, LS_GRAD, LS_EIK, LS_TOTAL, LS_COUNT = 0, 1, 2, 3, 4


class NetCfg(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("blocks", C.c_int32), ("n_freqs", C.c_int32),
                ("has_transform", C.c_int32), ("scale_input", C.c_float),
                ("scale_output", C.c_float), ("bounds_T", C.c_float * 12),
                ("fwd_operand", C.c_int32), ("bwd_operand", C.c_int32), ("spill_operand", C.c_int32)]


class SampleArgs(C.Structure):
    _fields_ = [("depth_batch", C.c_void_p), ("normal_batch", C.c_void_p),
                ("T_WC_batch", C.c_void_p), ("frame_idx", C.c_void_p), ("normal_idx", C.c_void_p),
                ("n_frames", C.c_int32), ("n_rays", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("n_strat", C.c_int32), ("n_surf", C.c_int32), ("min_depth", C.c_float),
                ("dist_behind_surf", C.c_float), ("rng_mode", C.c_int32),
                ("draw_h", C.c_void_p), ("draw_w", C.c_void_p), ("draw_u", C.c_void_p),
                ("draw_n", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("n_inline", C.c_int32), ("frame_idx_inline", C.c_int32 * 8), ("normal_idx_inline", C.c_int32 * 8),
                ("reserved_inline", C.c_int32)]


class SampleOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ["n_valid", "indices_b", "indices_h", "indices_w", "depth_sample", "dirs_C_sample",
                 "norm_sample", "T_WC_sample", "dirs_W_sample", "z_vals", "pc"]]


class LossCfg(C.Structure):
    _fields_ = [("bounds_method", C.c_int32), ("loss_type", C.c_int32),
                ("trunc_weight", C.c_float), ("trunc_distance", C.c_float),
                ("eik_weight", C.c_float), ("eik_apply_dist", C.c_float),
                ("grad_weight", C.c_float), ("orien_loss", C.c_int32)]


class StepArgs(C.Structure):
    _fields_ = [("n_valid", C.c_void_p), ("max_rays", C.c_int32), ("S", C.c_int32),
                ("n_frames", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("pc", C.c_void_p), ("z_vals", C.c_void_p), ("depth_sample", C.c_void_p),
                ("dirs_C_sample", C.c_void_p), ("dirs_W_sample", C.c_void_p),
                ("norm_sample", C.c_void_p), ("indices_b", C.c_void_p), ("indices_h", C.c_void_p),
                ("indices_w", C.c_void_p), ("noise", C.c_void_p), ("noise_std", C.c_float),
                ("reserved0", C.c_uint32), ("noise_seed", C.c_uint64), ("noise_offset", C.c_uint64),
                ("pc_bounds", C.c_void_p),
                ("pc_grad_vec", C.c_void_p), ("extra_floats", C.c_int32), ("extra_slot", C.c_int32),
                ("extra_value", C.c_float), ("reserved1", C.c_int32)]


class StepOut(C.Structure):
    _fields_ = [("reduce_buf", C.c_void_p), ("sdf", C.c_void_p), ("sdf_grad", C.c_void_p),
                ("tot_loss_mat", C.c_void_p), ("prof_events", C.POINTER(C.c_void_p)), ("host_mailbox", C.c_void_p),
                ("split_event", C.c_void_p)]


class OptimArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("shadow", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("grad_scale", C.c_float), ("step", C.c_int32), ("reserved", C.c_int32),
                ("loss_approx", C.c_void_p), ("frame_avg", C.c_void_p), ("frame_avg_index", C.c_void_p),
                ("frame_avg_inline_n", C.c_int32), ("frame_avg_index_inline", C.c_int32 * 8), ("reserved_inline", C.c_int32)]
MAX_INLINE_FRAMES = 8



class IsdfError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IsdfError("%s not found: run `python -m isdf_amd.build` (or __graft_entry__.build())"
                        % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    P, i32, i64, f32, vp = C.POINTER, C.c_int32, C.c_int64, C.c_float, C.c_void_p
    L.isdf_abi_version.restype = C.c_int
    L.isdf_error_string.restype = C.c_char_p
    L.isdf_error_string.argtypes = [C.c_int]
    L.isdf_check_net.restype = C.c_int
    L.isdf_check_net.argtypes = [P(NetCfg)]
    for n in ("isdf_param_count", "isdf_shadow_bytes", "isdf_reduce_split_floats"):
        getattr(L, n).restype = i64
        getattr(L, n).argtypes = [P(NetCfg)]
    L.isdf_workspace_bytes.restype = i64
    L.isdf_workspace_bytes.argtypes = [P(NetCfg), i64, i32]
    L.isdf_reduce_floats.restype = i64
    L.isdf_reduce_floats.argtypes = [P(NetCfg), i32]
    L.isdf_pack_weights.argtypes = [P(NetCfg), vp, vp, vp]
    L.isdf_sample_scan_bytes.restype = i64
    L.isdf_sample_scan_bytes.argtypes = [i64]
    L.isdf_sample_rays.argtypes = [P(SampleArgs), P(SampleOut), vp, i64, vp]
    L.isdf_sdf_eval.argtypes = [P(NetCfg), vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.isdf_train_step.argtypes = [P(NetCfg), P(LossCfg), vp, vp, P(StepArgs), P(StepOut), vp, i64, vp]
    L.isdf_train_step_adamw.argtypes = [P(NetCfg), P(LossCfg), P(StepArgs), P(StepOut), P(OptimArgs), vp, i64, vp]
    L.isdf_train_step_finish.argtypes = [P(NetCfg), P(OptimArgs), vp, i32, i32, vp, vp]
    L.isdf_allreduce_sum_f32.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.isdf_bounds_pc.argtypes = [vp, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp]
    L.isdf_frame_avg.argtypes = [vp, i64, i32, vp, vp, vp, vp]
    L.isdf_adamw.argtypes = [P(NetCfg), vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, i32, vp, vp]
    L.isdf_estimate_normals.argtypes = [vp, i32, i32, f32, f32, f32, f32, vp, vp]
    L.isdf_render_depth.argtypes = [vp, i64, i64, i32, vp, vp, vp, f32, vp, vp, vp]
    for n in SYMBOLS[SYMBOLS.index("isdf_pack_weights"):]:
        getattr(L, n).restype = C.c_int
    if L.isdf_abi_version() != ABI_VERSION:
        raise IsdfError("libisdf_hip.so ABI %d != binding ABI %d" % (L.isdf_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise IsdfError("%s failed: %s (%d)" % (what, lib().isdf_error_string(int(rc)).decode(), rc))


def ptr(t):
    """device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
