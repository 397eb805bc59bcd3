"""ctypes binding of libvila_hip.so (include/vila_hip.h).  No CPU fallback: if the library is missing or a symbol
is absent this module raises, and every op raises if handed a non-GPU tensor."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libvila_hip.so")

c_void_p, c_int, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class VilaVitShape(C.Structure):
    _fields_ = [("hidden", c_int), ("inter", c_int), ("heads", c_int), ("image", c_int), ("patch", c_int),
                ("channels", c_int), ("n_layers_run", c_int), ("ln_eps", c_float)]


class VilaVitLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo",
                                        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VilaVitWeights(C.Structure):
    _fields_ = [("shape", VilaVitShape), ("patch_w", c_void_p), ("patch_b", c_void_p), ("pos_emb", c_void_p),
                ("layers", C.POINTER(VilaVitLayer))]


class VilaVitLayerW8(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("wqkv_q", "wo_q", "fc1_q", "fc2_q", "wqkv_s", "wo_s", "fc1_s", "fc2_s")]


class VilaProjWeights(C.Structure):
    _fields_ = [("kind", c_int), ("in_dim", c_int), ("out_dim", c_int)] + \
               [(n, c_void_p) for n in ("ln1_w", "ln1_b", "fc1_w", "fc1_b", "ln2_w", "ln2_b", "fc2_w", "fc2_b", "fc3_w", "fc3_b")]


class VilaLlmShape(C.Structure):
    _fields_ = [("hidden", c_int), ("inter", c_int), ("n_layers", c_int), ("q_heads", c_int), ("kv_heads", c_int),
                ("head_dim", c_int), ("vocab", c_int), ("rms_eps", c_float), ("rope_theta", c_float)]


class VilaLlmLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "ln2_w", "w_gate", "w_up", "w_down")]


class VilaLlmWeights(C.Structure):
    _fields_ = [("shape", VilaLlmShape), ("embed", c_void_p), ("layers", C.POINTER(VilaLlmLayer)),
                ("norm_w", c_void_p), ("lm_head", c_void_p)]


class VilaLlmLayerW4(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv_q", "qkv_sz", "o_q", "o_sz", "gateup_q", "gateup_sz", "down_q", "down_sz")]


class VilaKvCache(C.Structure):
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("max_ctx", c_int), ("n_slots", c_int)]


class VilaDecodeBatch(C.Structure):
    _fields_ = [("n", c_int), ("pos", c_void_p), ("token", c_void_p), ("out_ids", c_void_p), ("n_out", c_void_p), ("max_out", c_int),
                ("logits", c_void_p)]


class VilaSampling(C.Structure):
    _fields_ = [("temperature", c_float), ("top_k", c_int), ("top_p", c_float), ("seed", C.c_uint64), ("seed_dev", c_void_p)]


class VilaDecodeState(C.Structure):
    _fields_ = [("pos", c_void_p), ("token", c_void_p), ("out_ids", c_void_p), ("n_out", c_void_p),
                ("max_out", c_int), ("logits", c_void_p)]


class VilaSftBatch(C.Structure):
    _fields_ = [("pixels", c_void_p), ("n_images", c_int), ("total_tokens", c_int),
                ("txt_src", c_void_p), ("txt_dst", c_void_p), ("n_txt", c_int),
                ("feat_src", c_void_p), ("feat_dst", c_void_p), ("n_feat", c_int),
                ("nl_src", c_void_p), ("nl_dst", c_void_p), ("n_nl", c_int),
                ("positions", c_void_p), ("cu_seqlens", c_void_p), ("n_seq", c_int), ("max_seqlen", c_int),
                ("target_rows", c_void_p), ("targets", c_void_p), ("n_targets", c_int), ("loss_scale", c_float),
                ("s2_desc", c_void_p), ("s2_tile_desc", c_void_p), ("s2_n_blocks", c_int), ("s2_n_scales", c_int), ("s2_splits", C.c_int32 * 4),
                ("pools", c_void_p), ("n_pools", c_int), ("n_media_rows", c_int)]


GRAD_READY_CB = C.CFUNCTYPE(None, c_void_p, c_int, c_int)
BUCKET_LM_HEAD, BUCKET_FINAL_NORM, BUCKET_LLM_LAYER, BUCKET_EMBED, BUCKET_PROJECTOR, BUCKET_VIT_LAYER, BUCKET_VIT_EMBED = range(7)

# name -> (restype, argtypes); every symbol include/vila_hip.h declares
PROTOTYPES = {
    "vila_last_error": (C.c_char_p, []),
    "vila_abi_version": (c_int, []),
    "vila_vit_workspace_bytes": (c_size_t, [C.POINTER(VilaVitShape), c_int]),
    "vila_vit_forward": (c_int, [C.POINTER(VilaVitWeights), c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vila_vit_w8a8_workspace_bytes": (c_size_t, [C.POINTER(VilaVitShape), c_int]),
    "vila_vit_forward_w8a8": (c_int, [C.POINTER(VilaVitWeights), C.POINTER(VilaVitLayerW8), c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vila_quant_rows_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vila_gemm_w8a8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                               c_int, c_int, c_int, c_int, c_void_p]),
    "vila_proj_workspace_bytes": (c_size_t, [C.POINTER(VilaProjWeights), c_int, c_int]),
    "vila_proj_out_tokens": (c_int, [c_int, c_int]),
    "vila_proj_forward": (c_int, [C.POINTER(VilaProjWeights), c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vila_embed_tokens": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vila_copy_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vila_llm_prefill_workspace_bytes": (c_size_t, [C.POINTER(VilaLlmShape), c_int]),
    "vila_llm_prefill": (c_int, [C.POINTER(VilaLlmWeights), c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                 C.POINTER(VilaKvCache), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "vila_llm_decode_workspace_bytes": (c_size_t, [C.POINTER(VilaLlmShape), c_int]),
    "vila_llm_decode_launches": (c_int, [C.POINTER(VilaLlmShape), c_int]),
    "vila_llm_decode_step": (c_int, [C.POINTER(VilaLlmWeights), C.POINTER(VilaKvCache), C.POINTER(VilaDecodeState),
                                     c_void_p, c_size_t, c_void_p]),
    "vila_sample_workspace_bytes": (c_size_t, []),
    "vila_sample_f32": (c_int, [c_void_p, c_int, C.POINTER(VilaSampling), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vila_llm_decode_step_sample": (c_int, [C.POINTER(VilaLlmWeights), C.POINTER(VilaKvCache), C.POINTER(VilaDecodeState),
                                            c_void_p, c_size_t, C.POINTER(VilaSampling), c_void_p]),
    "vila_llm_decode_batch_workspace_bytes": (c_size_t, [C.POINTER(VilaLlmShape), c_int]),
    "vila_llm_decode_step_batch": (c_int, [C.POINTER(VilaLlmWeights), C.POINTER(VilaKvCache), C.POINTER(VilaDecodeBatch), c_void_p, c_size_t, c_void_p]),
    "vila_graph_begin": (c_int, [c_void_p]),
    "vila_graph_end": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "vila_graph_launch": (c_int, [c_void_p, c_void_p]),
    "vila_graph_destroy": (c_int, [c_void_p]),
    "vila_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                               c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_gemm_bf16_ws": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                  c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vila_gemm_force_tile": (None, [c_int]),
    "vila_gemm_force_sched": (None, [c_int]),
    "vila_gemm_force_hybrid": (None, [c_int]),
    "vila_gemm_force_bm": (None, [c_int]),
    "vila_gemm_force_ex": (None, [c_int]),
    "vila_gemm_force_group": (None, [c_int]),
    "vila_gemm_force_fuse_norm": (None, [c_int]),
    "vila_prefill_force_fusions": (None, [c_int, c_int]),
    "vila_norm_force_lat": (None, [c_int]),
    "vila_decode_force_attn": (None, [c_int]),
    "vila_decode_force_chain": (None, [c_int]),
    "vila_decode_force_persist": (None, [c_int]),
    "vila_decode_persist_trace": (None, [c_void_p, c_int]),
    "vila_llm_decode_chain_error": (c_int, [c_void_p, c_void_p]),
    "vila_gemm_bf16_t": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                 c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vila_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "vila_rmsnorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "vila_space_to_depth_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_attn_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                   c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_float, c_void_p, c_void_p]),
    "vila_gemv_bf16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_int, c_int, c_void_p]),
    "vila_argmax_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    # ---- SFT step operators ----
    "vila_transpose_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p]),
    "vila_act_fwd_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vila_act_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vila_silu_mul_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vila_silu_mul_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vila_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vila_grad_accum_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vila_colsum_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "vila_norm_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vila_ce_loss_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_void_p]),
    "vila_scatter_add_rows_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vila_depth_to_space_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_im2col_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_rope_table_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "vila_rope_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_rope_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_attn_bwd_bf16": (c_int, [c_void_p] * 8 + [C.POINTER(c_int64), C.POINTER(C.c_int32), c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "vila_attn_bwd_bf16_parts": (c_int, [c_void_p] * 8 + [C.POINTER(c_int64), C.POINTER(C.c_int32), c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "vila_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int,
                                c_float, c_void_p]),
    "vila_adamw_step_lean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int,
                                     c_float, c_void_p]),
    "vila_sumsq_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vila_colsum_scratch_floats": (c_size_t, [c_int, c_int]),
    "vila_norm_bwd_scratch_floats": (c_size_t, [c_int, c_int]),
    "vila_gemv_w4_bf16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_void_p]),
    "vila_llm_decode_step_w4": (c_int, [C.POINTER(VilaLlmWeights), C.POINTER(VilaLlmLayerW4), C.POINTER(VilaKvCache),
                                        C.POINTER(VilaDecodeState), c_void_p, c_size_t, c_void_p]),
    "vila_llm_decode_step_w4_sample": (c_int, [C.POINTER(VilaLlmWeights), C.POINTER(VilaLlmLayerW4), C.POINTER(VilaKvCache),
                                               C.POINTER(VilaDecodeState), c_void_p, c_size_t, C.POINTER(VilaSampling), c_void_p]),
    "vila_video_pool_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "vila_video_pool_bwd_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vila_sft_workspace_bytes": (c_size_t, [C.POINTER(VilaVitWeights), C.POINTER(VilaProjWeights), C.POINTER(VilaLlmWeights), C.POINTER(VilaSftBatch)]),
    "vila_sft_fwd_bwd": (c_int, [C.POINTER(VilaVitWeights), C.POINTER(VilaVitWeights), C.POINTER(VilaProjWeights), C.POINTER(VilaProjWeights),
                                 C.POINTER(VilaLlmWeights), C.POINTER(VilaLlmWeights), C.POINTER(VilaSftBatch), c_void_p, c_void_p, c_size_t,
                                 GRAD_READY_CB, c_void_p, c_void_p]),
    "vila_s2_merge_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.POINTER(C.c_int32), c_void_p]),
    "vila_s2_merge_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.POINTER(C.c_int32), c_void_p]),
}

_lib: Optional[C.CDLL] = None


class VilaHipError(RuntimeError):
    pass


def load(build_if_missing: bool = False) -> C.CDLL:
    """Load the library.  Never falls back to anything else: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; ours must bind to THAT runtime instance (same SONAME: first loaded wins),
    # otherwise streams/pointers created by torch are foreign to the runtime our kernels are launched through.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build as _b
            _b.build()
        else:
            raise VilaHipError(f"{LIB_PATH} not found. Build it with `python -m vila_amd.build` (needs hipcc); "
                               "vila_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VilaHipError(f"libvila_hip.so does not export `{name}` (stale build? run python -m vila_amd.build --force)") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = load().vila_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(msg)         # argument / shape errors: the reference raises ValueError for these
    raise VilaHipError(f"{what}: {msg} (status {rc})")
