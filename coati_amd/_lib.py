"""ctypes binding of libcoati_hip.so (C ABI: include/coati_hip.h).  There is no CPU fallback: if the library is
missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

from . import build as _build

P, I, L, F = c_void_p, c_int, c_int64, c_float


class CoatiConfig(ctypes.Structure):
    _fields_ = [
        ("n_layer_xformer", c_int32), ("n_layer_e3gnn", c_int32), ("n_hidden_xformer", c_int32),
        ("n_hidden_e3nn", c_int32), ("n_embd_common", c_int32), ("n_head", c_int32), ("n_seq", c_int32),
        ("n_tok", c_int32), ("msg_cutoff", c_float), ("pad_token", c_int32), ("stop_token", c_int32),
        ("unk_token", c_int32), ("use_fp8", c_int32), ("norm_clips", c_int32), ("token_mlp", c_int32),
        ("use_point_encoder", c_int32), ("biases", c_int32), ("norm_embed", c_int32), ("torch_emb", c_int32),
        ("old_architecture", c_int32), ("residual", c_int32),
    ]


_SIGS = {
    "coati_comm_unique_id": [P, I],
    "coati_comm_init": [P, I, I, P],
    "coati_comm_rank": [P],
    "coati_comm_world": [P],
    "coati_comm_destroy": [P],
    "coati_allgather_rows": [P, P, P, L, L, I, P],
    "coati_reducescatter_rows": [P, P, P, L, L, I, P],
    "coati_allreduce_bucket": [P, P, L, I, I, P],
    "coati_gemm_nt": [P, I, L, P, L, I, I, I, P, L, I, P, P, P, L, I, P],
    "coati_gemm_lnbwd": [P, L, P, L, I, I, P, P, P, P, P, P, P, P, POINTER(c_int32), P, P, P],
    "coati_quant_mx8": [P, I, L, P, L, P, I, I, P],
    "coati_gemm_mx8": [P, L, P, P, L, P, I, I, I, P, L, P, P, P, L, I, P],
    "coati_gemm_ce_partial": [P, L, P, L, I, I, I, P, P],
    "coati_ce_finish": [P, I, P, L, P, L, P, P, P, I, I, I, P],
    "coati_gemm_ce_bwd": [P, L, P, L, I, I, I, P, L, I, P, P, P, P],
    "coati_wgrad": [P, I, L, P, L, I, I, I, P, L, P, I, P],
    "coati_wgrad_grouped": [I, P, P, P, P, I, P, P, P, P, P, I, P, L, P],
    "coati_sgemm": [P, L, L, P, L, L, P, L, I, I, I, P, F, I, P],
    "coati_layernorm_fwd": [P, L, P, P, P, L, P, L, P, P, I, I, P],
    "coati_layernorm_bwd": [P, I, L, P, L, I, P, P, P, P, P, P, P, P, P, I, I, P],
    "coati_attn_fwd": [P, P, P, I, I, I, P],
    "coati_gemm_qkv_rope": [P, L, P, L, P, I, I, P, L, P, P, I, P],
    "coati_attn_bwd": [P, P, P, P, P, P, P, P, I, I, I, P],
    "coati_embed_fwd": [P, P, P, I, P, I, I, I, I, P],
    "coati_embed_bwd": [P, P, P, P, I, I, I, I, I, P],
    "coati_find_stop": [P, I, P, P, I, I, P],
    "coati_gather_rows": [P, P, P, I, I, I, P],
    "coati_scatter_rows_add": [P, P, P, I, I, I, P],
    "coati_bad_rows": [P, P, I, I, P],
    "coati_silu": [P, P, L, P],
    "coati_attn_decode": [P, P, P, I, I, I, I, P],
    "coati_attn_decode_hs": [P, P, P, I, I, I, I, I, P],
    "coati_attn_fwd_hs": [P, P, P, I, I, I, I, P],
    "coati_attn_bwd_hs": [P, P, P, P, P, P, P, P, I, I, I, I, P],
    "coati_gemm_qkv_rope_hs": [P, L, P, L, P, I, I, P, L, P, P, I, I, P],
    "coati_topk_sample": [P, L, I, I, I, F, P, P, P, I, I, P],
    "coati_engine_decode_begin": [P, P, L, I, I],
    "coati_engine_decode_step": [P, P, P, P, L, P],
    "coati_engine_decode_pos": [P],
    "coati_engine_decode_graph_build": [P, P],
    "coati_engine_decode_graph_step": [P, P, P, P, P, P],
    "coati_batch_ncols": [P, I, I, P, P],
    "coati_batch_tail": [P, I, I, I, P, P, P, I, P],
    "coati_gnn_embed": [P, P, P, P, P, P, P, L, P, P, I, I, P],
    "coati_gnn_geom": [P, P, F, P, P, I, I, P],
    "coati_gnn_edge_pre": [P, L, P, P, P, P, L, P, P, I, I, P],
    "coati_gnn_edge_reduce": [P, P, P, P, L, I, I, P],
    "coati_gnn_compact": [P, P, P, P, P, P, P, P, P, P, I, I, P],
    "coati_infonce_rows": [P, L, I, I, I, P, P, P, F, P],
    "coati_count_valid": [P, I, P, P, P],
    "coati_colsum2": [P, P, P, P, I, I, P],
    "coati_center_rows": [P, P, P, P, P, I, I, P],
    "coati_standardize": [P, P, P, P, P, P, I, I, P],
    "coati_barlow_dc": [P, P, F, P, I, P],
    "coati_standardize_bwd": [P, P, P, P, P, P, F, P, I, I, P],
    "coati_grad_sqnorm": [P, L, P, I, P, F, P, P],
    "coati_adamw": [P, P, P, P, P, L, F, F, F, F, F, I, P, F, P],
    "coati_engine_create": [POINTER(CoatiConfig), POINTER(c_void_p)],
    "coati_engine_entry": [P, I, c_char_p, I, POINTER(c_int64), POINTER(c_int32), POINTER(c_int32)],
    "coati_engine_bind": [P, P, P, P, P, P, P, P, P, P],
    "coati_engine_refresh_shadows": [P, P],
    "coati_engine_bind_fp8": [P, P, L],
    "coati_engine_forward": [P, P, L, I, I, I, I, P, P, P, P, P, P, P, P, P, P, I, L, L, P],
    "coati_engine_forward_decoder": [P, P],
    "coati_seq_pack": [P, P, I, I, I, I, P, P, P, P, P, P],
    "coati_attn_fwd_varlen": [P, P, P, P, I, I, I, I, P],
    "coati_attn_bwd_varlen": [P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "coati_engine_logits": [P, P, L, P],
    "coati_engine_encode": [P, P, L, I, I, I, P, P, P, P, P, P, P],
    "coati_engine_infonce": [P, P, P, P, P, P, I, I, I, F, P, P, P, P],
    "coati_engine_backward": [P, P, P, I, P],
    "coati_engine_optimizer_step": [P, F, F, F, F, F, F, I, P, P],
    "coati_engine_set_error_word": [P, P, P],
    "coati_engine_reserve": [P, I, I, I, I],
    "coati_engine_prof_select": [P, I],
    "coati_engine_prof_keep_overlap": [P, I],
    "coati_engine_prof_pause": [P, I],
    "coati_engine_prof_add_site": [P, I],
    "coati_engine_prof_collect": [P, POINTER(c_double), POINTER(c_int64), POINTER(c_double)],
    "coati_engine_prof_last_bytes": [P, POINTER(c_double)],
}

# operators of csrc/experimental/ -- only in libcoati_hip_x.so (COATI_AMD_EXPERIMENTAL=1, build.py); bound when the loaded library has them
_EXPERIMENTAL_SIGS = {
    "coati_mlp_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, I, P],
    "coati_attn_groups": [P, I, I, P, P],
    "coati_attn_block_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P],
    "coati_ab_probe_swap": [P, P],
    "coati_ab_trace_read": [P],
}

_lib = None
ABI_VERSION = 5


def has_experimental():
    return hasattr(lib(), "coati_attn_block_fwd")


def lib():
    """Loads (building if the sources are newer) the HIP library.  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    # build() is a no-op unless a source under csrc/ or include/ is newer than the library (build.needs_build); when no
    # compiler is around (a box that only received the prebuilt .so) a stale-looking timestamp must not be fatal
    try:
        _build.build(force=os.environ.get("COATI_AMD_REBUILD") == "1", verbose=False)
    except Exception:
        if not os.path.exists(path):
            raise
    if not os.path.exists(path):
        raise RuntimeError(f"coati_amd: {path} is missing and could not be built; there is no CPU fallback")
    # torch first: its wheel carries its own ROCm runtime; loading this library before torch would bind it to the system
    # libamdhip64 and leave the process with two HIP runtimes (one of which then reports "no ROCm-capable device")
    import torch  # noqa: F401
    l = ctypes.CDLL(path)
    l.coati_last_error.restype = c_char_p
    l.coati_abi_version.restype = c_int
    if l.coati_abi_version() != ABI_VERSION:
        raise RuntimeError(f"coati_amd: {path} has ABI version {l.coati_abi_version()}, this package expects {ABI_VERSION}; "
                           "rebuild with `python -m coati_amd.build --force`")
    missing = [n for n in exported_symbols() if not hasattr(l, n)]
    if missing:
        raise RuntimeError(f"coati_amd: {path} lacks symbols {missing[:6]}; rebuild with `python -m coati_amd.build --force`")
    for name, sig in _SIGS.items():
        fn = getattr(l, name)
        fn.argtypes = sig
        fn.restype = c_int
    for name, sig in _EXPERIMENTAL_SIGS.items():
        if hasattr(l, name):
            getattr(l, name).argtypes = sig
            getattr(l, name).restype = c_int
    l.coati_engine_destroy.argtypes = [P]
    l.coati_engine_destroy.restype = None
    for name in ("coati_engine_param_elems", "coati_engine_shadow_elems", "coati_engine_trainable_elems"):
        getattr(l, name).argtypes = [P]
        getattr(l, name).restype = c_int64
    l.coati_engine_workspace_bytes.argtypes = [P, I, I, I, I, I]
    l.coati_engine_workspace_bytes.restype = c_int64
    l.coati_wgrad_grouped_workspace_bytes.argtypes = [I, P, P, I]
    l.coati_wgrad_grouped_workspace_bytes.restype = c_int64
    l.coati_engine_fp8_bytes.argtypes = [P]
    l.coati_engine_fp8_bytes.restype = c_int64
    l.coati_engine_decode_workspace_bytes.argtypes = [P, I, I]
    l.coati_engine_decode_workspace_bytes.restype = c_int64
    l.coati_tokenizer_create.argtypes = [P, P, I, P, P, I, P]
    l.coati_tokenizer_create.restype = c_int
    l.coati_tokenizer_destroy.argtypes = [P]
    l.coati_tokenizer_destroy.restype = None
    l.coati_tokenizer_encode.argtypes = [P, c_char_p, c_int64, P, I]
    l.coati_tokenizer_encode.restype = c_int64
    l.coati_tokenizer_pieces.argtypes = [P, c_char_p, c_int64, P, P, P, I]
    l.coati_tokenizer_pieces.restype = c_int64
    l.coati_tokenizer_encode_batch.argtypes = [P, P, I, I, P, P, I]
    l.coati_tokenizer_encode_batch.restype = c_int
    l.coati_engine_n_entries.argtypes = [P]
    l.coati_engine_n_entries.restype = c_int
    l.coati_engine_site_count.restype = c_int
    l.coati_engine_site_name.argtypes = [I]
    l.coati_engine_site_name.restype = c_char_p
    _lib = l
    return l


def exported_symbols():
    return sorted(list(_SIGS) + (list(_EXPERIMENTAL_SIGS) if _build.EXPERIMENTAL else []) + ["coati_last_error", "coati_abi_version", "coati_engine_destroy",
                                 "coati_engine_param_elems", "coati_engine_shadow_elems", "coati_engine_trainable_elems",
                                 "coati_engine_workspace_bytes", "coati_engine_decode_workspace_bytes", "coati_engine_n_entries",
                                 "coati_wgrad_grouped_workspace_bytes", "coati_engine_fp8_bytes",
                                 "coati_tokenizer_create", "coati_tokenizer_destroy", "coati_tokenizer_encode",
                                 "coati_tokenizer_pieces", "coati_tokenizer_encode_batch",
                                 "coati_engine_site_count", "coati_engine_site_name"])


def check(rc, what=""):
    if rc != 0:
        msg = lib().coati_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libcoati_hip: {what} failed (code {rc}): {msg}")


def call(name, *args):
    if name in _EXPERIMENTAL_SIGS and not hasattr(lib(), name):
        raise RuntimeError(f"libcoati_hip: {name} is an operator of csrc/experimental/ -- build and load with COATI_AMD_EXPERIMENTAL=1")
    check(getattr(lib(), name)(*args), name)
