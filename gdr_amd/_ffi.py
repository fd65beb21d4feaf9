"""ctypes binding of libgdr_hip.so (include/gdr_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.  PyTorch is used only
for device memory and streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GDR_HIP_LIB") or os.path.join(_HERE, "libgdr_hip.so")   # override: A/B builds in the lab

GDR_OK, GDR_EINVAL, GDR_ENOSPC, GDR_EHIP = 0, -1, -2, -3
ABI_VERSION = 8                  # what this binding was written against (gdr_abi_version(), csrc/common.hip)
RERANK_POSITIONS = 1
SIM_EXHAUSTIVE = 1
SIM_NO_STREAM = 2
EPI_NONE, EPI_RESIDUAL, EPI_RELU, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_RESIDUAL, EPI_BIAS_GELU = range(7)


class GdrError(RuntimeError):
    pass


class GdrT5Dims(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("d_model", C.c_int32), ("d_kv", C.c_int32), ("d_ff", C.c_int32),
                ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("rel_buckets", C.c_int32),
                ("rel_max_distance", C.c_int32), ("eps", C.c_float)]


class GdrT5EncLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_attn", "wqkv", "wo", "ln_ff", "wi", "wo_ff")]


class GdrT5EncoderWeights(C.Structure):
    _fields_ = [("dims", GdrT5Dims), ("embed", C.c_void_p), ("rel_bias", C.c_void_p), ("final_ln", C.c_void_p),
                ("layers", C.POINTER(GdrT5EncLayer))]


class GdrBertLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_w", "ln1_b", "wi", "bi", "wo2", "bo2", "ln2_w",
                                          "ln2_b")]


class GdrBertWeights(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("d_model", C.c_int32), ("num_heads", C.c_int32), ("d_ff", C.c_int32),
                ("num_layers", C.c_int32), ("max_pos", C.c_int32), ("type_vocab", C.c_int32), ("eps", C.c_float),
                ("word_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("type_emb", C.c_void_p), ("emb_ln_w", C.c_void_p),
                ("emb_ln_b", C.c_void_p), ("layers", C.POINTER(GdrBertLayer))]


class GdrTrie(C.Structure):
    _fields_ = [("child", C.c_void_p), ("eos_ok", C.c_void_p), ("n_nodes", C.c_int32), ("V", C.c_int32)]


class GdrPrefixTable(C.Structure):
    _fields_ = [("child", C.c_void_p), ("n_nodes", C.c_int32), ("V", C.c_int32), ("n_table", C.c_int32),
                ("kv", C.c_void_p), ("W", C.c_void_p), ("complete_levels", C.c_int32)]


class GdrClusterIndex(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("key_len", C.c_int32), ("table_size", C.c_int32), ("slots", C.c_void_p),
                ("keys", C.c_void_p), ("key_lens", C.c_void_p), ("offsets", C.c_void_p), ("members", C.c_void_p)]


class GdrT5DecLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_self", "wqkv", "wo", "ln_cross", "wq_c", "wkv_c", "wo_c", "ln_ff", "wi",
                                          "wo_ff")]


class GdrAdaptorLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("in_w", "in_b", "out_w", "out_b", "ln1_w", "ln1_b", "cross_const", "ln2_w",
                                          "ln2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "ln3_w", "ln3_b")]


class GdrT5DecoderWeights(C.Structure):
    _fields_ = [("dims", GdrT5Dims), ("out_vocab", C.c_int32), ("max_out_len", C.c_int32),
                ("adaptor_layers", C.c_int32), ("adaptor_nhead", C.c_int32), ("adaptor_ff", C.c_int32),
                ("adaptor_eps", C.c_float), ("dec_embed", C.c_void_p), ("self_rel_bias", C.c_void_p),
                ("cross_rel_bias", C.c_void_p), ("final_ln", C.c_void_p), ("layers", C.POINTER(GdrT5DecLayer)),
                ("alayers", C.POINTER(GdrAdaptorLayer)), ("head_w", C.c_void_p), ("head_e", C.c_void_p)]


# name -> (restype, argtypes); every symbol declared in include/gdr_hip.h
_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
SIGNATURES = {
    "gdr_last_error": (C.c_char_p, []),
    "gdr_abi_version": (_i, []),
    "gdr_device_fault_pending": (_i, []),
    "gdr_device_fault_clear": (None, []),
    "gdr_device_fault_inject_for_tests": (None, []),
    "gdr_prof_enable": (_i, [_i]),
    "gdr_prof_collect": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gdr_linear_f32": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "gdr_linear_bf16": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "gdr_split_row_elems": (_i, [_i, _i]),
    "gdr_split_f32_f16x2": (_i, [_vp, _vp, _i64, _i, _i64, _vp]),
    "gdr_split_f32_bf16x3": (_i, [_vp, _vp, _i64, _i, _i64, _vp]),
    "gdr_linear_split_bf16": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "gdr_linear_bf16_tile_form": (_i, [_i64, _i, _i, _i]),
    "gdr_linear_f32_splitk": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "gdr_l2_normalize": (_i, [_vp, _vp, _i64, _i, _f, _vp]),
    "gdr_t5_layer_norm": (_i, [_vp, _vp, _vp, _i64, _i, _f, _vp]),
    "gdr_t5_encoder_workspace_bytes": (_sz, [C.POINTER(GdrT5Dims), _i, _i]),
    "gdr_t5_encoder_forward": (_i, [C.POINTER(GdrT5EncoderWeights), _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "gdr_t5_encoder_ragged_workspace_bytes": (_sz, [C.POINTER(GdrT5Dims), _i, _i]),
    "gdr_t5_encoder_forward_ragged": (_i, [C.POINTER(GdrT5EncoderWeights), _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "gdr_t5_encoder_forward_ragged_bf16": (_i, [C.POINTER(GdrT5EncoderWeights), _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "gdr_t5_encoder_split_workspace_bytes": (_sz, [C.POINTER(GdrT5Dims), _i, _i]),
    "gdr_t5_encoder_forward_ragged_split": (_i, [C.POINTER(GdrT5EncoderWeights), _vp, _vp, _i, _i, _vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "gdr_t5_encoder_bf16_workspace_bytes": (_sz, [C.POINTER(GdrT5Dims), _i, _i]),
    "gdr_t5_encoder_forward_bf16": (_i, [C.POINTER(GdrT5EncoderWeights), _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "gdr_sim_topk_workspace_bytes": (_sz, [_i, _i64, _i, _i, _i]),
    "gdr_sim_topk": (_i, [_vp, _i, _vp, _i64, _i, _i, C.c_int32, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "gdr_sim_topk_bf16": (_i, [_vp, _i, _vp, _i64, _i, _i, C.c_int32, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "gdr_launch_count": (C.c_int64, []),
    "gdr_prof_gate": (None, [_i]),
    "gdr_sim_topk_prefilter_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "gdr_sim_topk_prefilter": (_i, [_vp, _i, _vp, _vp, C.c_float, _i64, _i, _i, C.c_int32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gdr_row_norm2_max": (_i, [_vp, _i64, _i, _vp, _vp]),
    "gdr_cast_f32_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "gdr_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "gdr_topk_pack": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "gdr_topk_merge_packed": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gdr_rerank_workspace_bytes": (_sz, [_i, _i]),
    "gdr_rerank_topk": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, C.c_int32, C.c_int32, _i,
                             _vp, _sz, _vp]),
    "gdr_rerank_topk_bf16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, C.c_int32, C.c_int32,
                                  _i, _vp, _sz, _vp]),
    "gdr_rerank_wire_pack": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gdr_rerank_wire_unpack": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gdr_rerank_positions_to_ids": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gdr_cluster_key_hash": (C.c_uint64, [C.POINTER(C.c_int32), _i]),
    "gdr_cluster_candidates": (_i, [C.POINTER(GdrClusterIndex), _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "gdr_t5_relative_bucket_table": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_int32)]),
    "gdr_bert_encoder_workspace_bytes": (_sz, [C.POINTER(GdrBertWeights), _i, _i]),
    "gdr_bert_encoder_forward": (_i, [C.POINTER(GdrBertWeights), _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "gdr_bert_encoder_ragged_workspace_bytes": (_sz, [C.POINTER(GdrBertWeights), _i, _i]),
    "gdr_bert_encoder_forward_ragged": (_i, [C.POINTER(GdrBertWeights), _vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "gdr_bert_encoder_forward_ragged_split": (_i, [C.POINTER(GdrBertWeights), _vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "gdr_bert_encoder_forward_ragged_bf16": (_i, [C.POINTER(GdrBertWeights), _vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "gdr_t5_generate_workspace_bytes": (_sz, [C.POINTER(GdrT5DecoderWeights), _i, _i, _i, _i]),
    "gdr_t5_generate": (_i, [C.POINTER(GdrT5DecoderWeights), _vp, _vp, _i, _i, _i, _i, C.c_double, _i, C.POINTER(GdrTrie),
                             C.POINTER(GdrPrefixTable), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gdr_t5_generate_bf16": (_i, [C.POINTER(GdrT5DecoderWeights), _vp, _vp, _i, _i, _i, _i, C.c_double, _i,
                                  C.POINTER(GdrTrie), C.POINTER(GdrPrefixTable), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gdr_t5_generate_early_exits": (C.c_int64, []),
    "gdr_t5_generate_last_done_step": (_i, []),
    "gdr_t5_prefix_table_build_bf16": (_i, [C.POINTER(GdrT5DecoderWeights), _i, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp,
                                            _vp, _sz, _vp]),
    "gdr_t5_prefix_table_workspace_bytes": (_sz, [C.POINTER(GdrT5DecoderWeights), _i]),
    "gdr_t5_prefix_table_build": (_i, [C.POINTER(GdrT5DecoderWeights), _i, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp, _vp,
                                       _sz, _vp]),
    "gdr_beam_search_table_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "gdr_beam_search_table": (_i, [_vp, _i, _i, _i, _i, C.c_double, _i, C.POINTER(GdrTrie), _vp, _vp, _vp, _vp, _sz,
                                   _vp]),
}

_lib = None


def lib():
    """Load the shared library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GdrError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gdr_amd/csrc`). gdr_amd has no CPU fallback.")
        if not os.environ.get("GDR_FFI_NO_TORCH"):
            # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so): it must be the one already in the process
            # when this library's dependency on libamdhip64 is resolved.  Loaded the other way round (this library first,
            # e.g. build() then smoke() in one process) the process ends up with /opt/rocm's runtime under torch's
            # allocator and streams, and the first hipMemsetAsync on a torch stream fails.  (The host-sanitizer test sets
            # GDR_FFI_NO_TORCH: it never touches a GPU.)
            import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)            # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        got = l.gdr_abi_version()
        if got != ABI_VERSION:               # a stale build (or a GDR_HIP_LIB override) with another argument layout
            raise GdrError(f"{LIB_PATH} reports ABI version {got}, this binding needs {ABI_VERSION}: rebuild it with "
                           "`make -C gdr_amd/csrc` (or drop the GDR_HIP_LIB override)")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().gdr_last_error().decode("utf-8", "replace")
        raise GdrError(f"{what} failed (code {rc}): {msg}")


def check_device_fault(what):
    """Raises if a kernel has reported a device-side fault since the last gdr_device_fault_clear() (today: a stream-K
    hand-off that timed out, include/gdr_hip.h) — called where the host has just synchronised with the device, before the
    result it read back is trusted.  The fault stays pending until the caller clears it."""
    if lib().gdr_device_fault_pending():
        raise GdrError(f"{what}: " + lib().gdr_last_error().decode("utf-8", "replace"))


def clear_device_fault():
    """Acknowledges a reported device fault (gdr_device_fault_clear).  WHO OWNS THE CLEAR: the caller that saw the GdrError of
    check_device_fault() and has discarded (or will redo) the step it belongs to — GDRRetriever.validation_steps / main.inference
    do so after reporting the failed batch; a long-lived process that only logs the error must call this before it goes on,
    otherwise every later stream-K launch of the process keeps failing with GDR_EHIP.  Returns True if a fault was pending.
    In a multi-rank job the decision to go on must be collective (dist.all_ranks_ok): a fault is rank-local."""
    pending = bool(lib().gdr_device_fault_pending())
    lib().gdr_device_fault_clear()
    return pending


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
