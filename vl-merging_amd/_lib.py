"""ctypes binding of libvlm_hip.so (the C ABI in include/vlm_hip.h).  Fails loudly when absent."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLM_LIB_PATH") or os.path.join(_HERE, "lib", "libvlm_hip.so")
_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_u64 = ctypes.c_uint64

MERGE_LERP, MERGE_TASKVEC, MERGE_MEAN = 0, 1, 2
MERGE_MAX_SRC = 4


class MergeJob(ctypes.Structure):
    _fields_ = [
        ("dst", c_void_p),
        ("base", c_void_p),
        ("src", c_void_p * MERGE_MAX_SRC),
        ("ratio", c_float * MERGE_MAX_SRC),
        ("n_src", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("n_elem", c_u64),
    ]


class ScatterSrc(ctypes.Structure):
    """vlm_scatter_src_t of include/vlm_hip.h."""
    _fields_ = [("g", c_void_p), ("g_is_f32", ctypes.c_int32), ("ld", ctypes.c_int32), ("first_row", ctypes.c_int32),
                ("row_step", ctypes.c_int32), ("count", ctypes.c_int32)]


class Epilogue(ctypes.Structure):
    _fields_ = [
        ("bias", c_void_p), ("col_scale", c_void_p), ("row_scale", c_void_p), ("residual", c_void_p),
        ("ld_res", ctypes.c_int64), ("aux", c_void_p), ("ld_aux", ctypes.c_int64), ("act", ctypes.c_int32),
        ("alpha", c_float), ("accumulate", ctypes.c_int32), ("reserved", ctypes.c_int32), ("col_sum", c_void_p),
        ("col_sum_ws", c_void_p), ("splitk_ws", c_void_p), ("splitk_ws_bytes", ctypes.c_uint64),
    ]


GEMM_MAX_GROUPS = 4


class GemmGroup(ctypes.Structure):
    _fields_ = [("row0", ctypes.c_int32), ("rows", ctypes.c_int32), ("B", c_void_p), ("ldb", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("bias", c_void_p), ("col_sum", c_void_p), ("col_sum_ws", c_void_p)]


class WgradGroup(ctypes.Structure):
    _fields_ = [("row0", ctypes.c_int32), ("rows", ctypes.c_int32), ("C", c_void_p), ("accumulate", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class FoldJob(ctypes.Structure):
    _fields_ = [("partials", c_void_p), ("nblocks", ctypes.c_int32), ("D", ctypes.c_int32), ("out0", c_void_p),
                ("out1", c_void_p)]


class LayerScale(ctypes.Structure):  # vlm_layerscale_t
    _fields_ = [("y", c_void_p), ("ldy", ctypes.c_int32), ("gamma", c_void_p), ("row_scale", c_void_p), ("dy", c_void_p),
                ("lddy", ctypes.c_int32), ("dgamma", c_void_p), ("dbias", c_void_p), ("workspace", c_void_p),
                ("workspace_bytes", c_size_t)]


class AttnColsum(ctypes.Structure):
    _fields_ = [("dq", c_void_p * 2), ("dv", c_void_p * 2)]


ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_GELU_DERIV, ACT_MUL_AUX = 0, 1, 2, 3, 4
ATTN_JOINT, ATTN_SEPARATE = 0, 1


class AttnDesc(ctypes.Structure):
    _fields_ = [
        ("qkv", c_void_p), ("ld_qkv", ctypes.c_int32), ("H", ctypes.c_int32), ("total_rows", ctypes.c_int32),
        ("R", ctypes.c_int32), ("bias_t", c_void_p), ("rel_index", c_void_p), ("rel_index_t", c_void_p),
        ("ld_index", ctypes.c_int32), ("index_rows", ctypes.c_int32), ("ld_index_t", ctypes.c_int32),
        ("index_t_rows", ctypes.c_int32), ("head_row0", ctypes.c_int32), ("mode", ctypes.c_int32),
        ("keep0", c_void_p), ("keep1", c_void_p),
        ("B", ctypes.c_int32), ("n0", ctypes.c_int32), ("n1", ctypes.c_int32), ("base0", ctypes.c_int32),
        ("base1", ctypes.c_int32), ("pos1", ctypes.c_int32), ("scale", c_float), ("reserved", ctypes.c_int32),
        ("bias_dense", c_void_p), ("bias_dense_t", c_void_p), ("dense_tiles", ctypes.c_int32), ("reserved2", ctypes.c_int32),
    ]


class LayerScaleJob(ctypes.Structure):
    _fields_ = [("weight", c_void_p), ("gamma", c_void_p), ("bias", c_void_p), ("shadow", c_void_p), ("bias_out", c_void_p),
                ("raw_w", c_void_p), ("raw_b", c_void_p), ("dweight", c_void_p), ("dbias", c_void_p), ("dgamma", c_void_p),
                ("N", ctypes.c_int32), ("K", ctypes.c_int32)]


MAX_LAYERSCALE_JOBS = 32


class VlmError(RuntimeError):
    pass


_ERR = {-1: "VLM_ERR_ARG", -2: "VLM_ERR_LAUNCH", -3: "VLM_ERR_WORKSPACE", -4: "VLM_ERR_UNSUPPORTED"}

# name -> (restype, argtypes); every symbol include/vlm_hip.h declares
SIGNATURES = {
    "vlm_abi_version": (c_int, []),
    "vlm_device_cus": (c_int, []),
    "vlm_set_cu_budget": (c_int, [c_int]),
    "vlm_debug_occupy": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "vlm_layerscale_fold": (c_int, [ctypes.POINTER(LayerScaleJob), c_int, c_void_p]),
    "vlm_layerscale_finish": (c_int, [ctypes.POINTER(LayerScaleJob), c_int, c_void_p]),
    "vlm_merge_plan_bytes": (c_size_t, [c_int, c_u64]),
    "vlm_merge_plan_upload": (c_int, [ctypes.POINTER(MergeJob), c_int, c_void_p, c_size_t, c_void_p]),
    "vlm_merge_run": (c_int, [c_void_p, c_void_p]),
    "vlm_gemm_bf16": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                              c_int, ctypes.POINTER(Epilogue), c_void_p]),
    "vlm_gemm_bf16_grouped": (c_int, [c_int, ctypes.POINTER(GemmGroup), c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                      ctypes.POINTER(Epilogue), c_void_p]),
    "vlm_gemm_wgrad_grouped": (c_int, [c_int, ctypes.POINTER(WgradGroup), c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                       c_void_p, c_u64, c_void_p]),
    "vlm_attention_fwd": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_int, c_void_p, c_void_p]),
    "vlm_attention_bwd_ws_floats": (c_size_t, [ctypes.POINTER(AttnDesc), c_int]),
    "vlm_attention_bwd": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                  c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vlm_layernorm_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int,
                                  c_void_p, c_void_p]),
    "vlm_layernorm_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                  c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                  ctypes.POINTER(c_int), c_void_p]),
    "vlm_layernorm_bwd_scale": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                        c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                        ctypes.POINTER(LayerScale), ctypes.POINTER(c_int), c_void_p]),
    "vlm_layerscale_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "vlm_colreduce_batch": (c_int, [ctypes.POINTER(FoldJob), c_int, c_void_p]),
    "vlm_gemm_set_big_tile_mode": (c_int, [c_int]),
    "vlm_colsum_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_float, c_float, c_float,
                               c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "vlm_accumulate_f32_f64": (c_int, [c_void_p, c_void_p, c_u64, c_void_p]),
    "vlm_gram_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_gemm_f64": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.c_double, c_void_p, c_int, c_int, c_void_p, c_int,
                             ctypes.c_double, c_void_p, c_int, c_void_p]),
    "vlm_scale_gram_f64": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_double, c_int, c_void_p]),
    "vlm_potrf_block_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_trsm_block_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vlm_cholesky_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "vlm_solve_spd_right_f64": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "vlm_gemm_f64_batched": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.c_double, ctypes.POINTER(c_void_p), c_int, c_int,
                                     ctypes.POINTER(c_void_p), c_int, ctypes.c_double, ctypes.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "vlm_cholesky_f64_batched": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]),
    "vlm_solve_spd_right_f64_batched": (c_int, [ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "vlm_cross_entropy_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p]),
    "vlm_cross_entropy_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p]),
    "vlm_l2norm_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vlm_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "vlm_contrastive_ws_floats": (c_size_t, [c_int]),
    "vlm_contrastive": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "vlm_small_cross_entropy": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlm_cross_entropy_reduce": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    "vlm_scale_by_scalar": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_void_p, c_void_p]),
    "vlm_text_rows_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float, c_float,
                                  c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "vlm_text_rows_bwd_ws_floats": (c_size_t, [c_int]),
    "vlm_text_rows_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                  c_int, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlm_image_rows_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "vlm_image_lead_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_image_rows_bwd_ws_floats": (c_size_t, [c_int]),
    "vlm_image_rows_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlm_tanh_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_act_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_colsum_small": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_sample_negatives": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vlm_weighted_sum": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_float), c_int, c_void_p, c_void_p]),
    "vlm_scatter_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "vlm_cast_f32_bf16": (c_int, [c_void_p, c_void_p, c_u64, c_void_p]),
    "vlm_embedding_bwd": (c_int, [c_void_p, c_int, c_void_p, ctypes.c_int64, c_int, ctypes.c_int64, c_void_p, c_int,
                                  ctypes.c_int64, c_void_p]),
    "vlm_bias_dense_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vlm_bias_dense": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                               c_void_p]),
    "vlm_transpose_bf16_tiles": (c_int, [c_void_p, c_int, c_void_p]),
    "vlm_droppath_sites": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p]),
    "vlm_droppath_rows": (c_int, [c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlm_patch_im2col": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
}


def get_lib():
    """Load the HIP library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VlmError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
        import torch  # noqa: F401  -- torch's bundled HIP runtime must be the one libvlm_hip.so binds to
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise VlmError(f"{what} failed: {_ERR.get(rc, rc)}")


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VlmError("vl_merging_amd ops run on the GPU only (tensor on %s); there is no CPU path" % t.device)
