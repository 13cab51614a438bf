"""ctypes binding of libevi_hip.so (the C-ABI declared in include/evi_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol the header
declares is absent, loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p
from typing import Dict, List, Optional

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libevi_hip.so")
HEADER_PATH = os.path.join(REPO_ROOT, "include", "evi_hip.h")

EVI_OK = 0
EVI_ERR_INVALID = -22
EVI_ERR_NOMEM = -12
EVI_ERR_HIP = -5
EVI_ERR_UNSUPPORTED = -95
EVI_TOPK_MAX_K = 2048

_P = c_void_p



class EviRetrieverWeights(Structure):
    """Mirror of `EviRetrieverWeights` in include/evi_hip.h."""

    _fields_ = [("emb_dim", c_int), ("hidden_dim", c_int), ("num_topics", c_int), ("dde_rounds", c_int),
                ("dde_reverse_rounds", c_int)] + [(name, c_void_p) for name in (
                    "entity_w", "entity_b", "relation_w", "relation_b", "query_w", "query_b", "non_text_emb",
                    "q_gate_w", "q_gate_b", "q_bias_w", "q_bias_b", "struct_w", "struct_b", "struct_ln_w",
                    "struct_ln_b", "struct_gate_w", "struct_gate_b", "state0_w", "state0_b", "state_ln_w",
                    "state_ln_b", "state4_w", "state4_b", "score_w", "score_b", "prepared")]


class EviRetrieverBatch(Structure):
    """Mirror of `EviRetrieverBatch` in include/evi_hip.h."""

    _fields_ = [("num_nodes", c_int64), ("num_edges", c_int64), ("num_graphs", c_int),
                ("edge_index", c_void_p), ("node_ptr", c_void_p), ("edge_ptr", c_void_p), ("edge_batch", c_void_p),
                ("question_emb", c_void_p), ("node_embeddings", c_void_p), ("node_embedding_ids", c_void_p),
                ("edge_embeddings", c_void_p), ("edge_attr", c_void_p), ("num_relations", c_int64),
                ("topic_one_hot", c_void_p), ("topic_stride", c_int), ("edge_bias", c_void_p),
                ("dropout_p", ctypes.c_float), ("dropout_seed", ctypes.c_uint64), ("matmul_precision", ctypes.c_int),
                ("relation_rows", c_void_p)]


class EviRetrieverOutput(Structure):
    """Mirror of `EviRetrieverOutput` in include/evi_hip.h."""

    _fields_ = [("logits", c_void_p), ("logits_fwd", c_void_p), ("logits_bwd", c_void_p),
                ("edge_features", c_void_p), ("node_struct", c_void_p), ("status", c_void_p),
                ("saved", c_void_p), ("saved_bytes", c_size_t)]


# name -> (restype, argtypes)
_SIGNATURES = {
    "evi_version": (c_int, []),
    "evi_last_error": (c_size_t, [c_char_p, c_size_t]),
    "evi_timing_enable": (c_int, [c_int]),
    "evi_timing_read": (c_int, [_P, _P, c_int]),
    "evi_row_inv_norm": (c_int, [_P, c_int64, c_int, c_float, _P, _P]),
    "evi_row_normalize": (c_int, [_P, c_int64, c_int, c_float, _P, _P]),
    "evi_cosine_topk_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "evi_cosine_topk_min_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "evi_cosine_topk": (c_int, [_P, c_int, _P, c_int64, c_int, _P, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "evi_cosine_topk_f16": (c_int, [_P, c_int, _P, c_int64, c_int, _P, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "evi_cosine_topk_fp8": (c_int, [_P, c_int, _P, c_int64, c_int, _P, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "evi_cosine_topk_fp8_mfma": (c_int, [_P, c_int, _P, c_int64, c_int, _P, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "evi_quantize_rows_fp8": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "evi_shortest_path_single": (c_int, [_P] * 6 + [c_int] + [_P] * 9 + [c_int, _P, _P, _P, _P]),
    "evi_first_occurrence_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "evi_first_occurrence": (c_int, [_P, c_int, c_int64, _P, c_int, _P, _P, _P, c_size_t, _P]),
    "evi_first_seen_rank": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P, _P]),
    "evi_segment_sort_rank": (c_int, [_P, c_int64, _P, _P, c_int, _P, _P, _P]),
    "evi_group_max_f32": (c_int, [_P, _P, c_int64, _P, _P]),
    "evi_segment_offsets": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P]),
    "evi_gather_segments": (c_int, [_P, c_int, c_int64, _P, c_int64, _P, c_int, _P, _P, _P, _P]),
    "evi_retriever_loss_workspace_bytes": (c_size_t, [c_int]),
    "evi_retriever_loss": (c_int, [_P, _P, _P, c_int, _P, c_float, c_float, c_float, c_float, c_float, _P, _P, _P, c_size_t, _P]),
    "evi_metric_accumulate": (c_int, [_P] * 9 + [c_int, c_int, _P, _P]),
    "evi_row_norms": (c_int, [_P, c_int64, c_int, _P, _P]),
    "evi_cosine_topk_gemm_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "evi_cosine_topk_gemm": (c_int, [_P, c_int, _P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "evi_index_shadow_bf16": (c_int, [_P, c_int64, c_int, _P, _P]),
    "evi_index_shadow_f16": (c_int, [_P, c_int64, c_int, _P, _P]),
    "evi_cosine_topk_two_stage_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "evi_cosine_topk_two_stage": (c_int, [_P, c_int, _P, _P, c_int64, c_int, c_int, c_int64, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "evi_cosine_topk_gemm_f16": (c_int, [_P, c_int, _P, c_int64, c_int, _P, c_int, c_int64, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "evi_topk_merge": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "evi_topk_packed_bytes": (c_size_t, [c_int, c_int]),
    "evi_topk_merge_packed": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "evi_segment_topk": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P]),
    "evi_retriever_metrics": (c_int, [_P, _P, _P, c_int64, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int]
                              + [_P] * 14),
    "evi_bfs_levels": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "evi_bfs_levels_edges": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int, _P, _P]),
    "evi_shortest_path_pairs": (c_int, [c_int, _P, _P, _P, _P, c_int, _P, _P, _P, c_int64, _P, _P, c_int, _P, _P, _P,
                                        _P, _P, _P]),
    "evi_node_softmax_logit_workspace_bytes": (c_size_t, [c_int64]),
    "evi_node_softmax_logit": (c_int, [_P, _P, c_int64, c_int64, _P, _P, c_size_t, _P]),
    "evi_select_start_edges": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int64, c_float, c_int, c_int, _P,
                                       _P, _P]),
    "evi_seed_onehop_stats": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, _P, _P]),
    "evi_masked_mean_pool": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    "evi_scatter_rows_workspace_bytes": (c_size_t, [c_int64]),
    "evi_scatter_rows": (c_int, [_P, _P, c_int64, c_int, _P, c_int64, _P, c_size_t, _P]),
    "evi_graph_class_stats": (c_int, [_P, _P, _P, c_int, _P, _P]),
    "evi_gather_rows": (c_int, [_P, c_int64, c_int, _P, c_int64, _P, _P, _P]),
    "evi_edge_batch": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P]),
    "evi_qa_edge_mask": (c_int, [_P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P]),
    "evi_graph_csr_workspace_bytes": (c_size_t, [c_int64]),
    "evi_graph_csr": (c_int, [_P, c_int64, _P, _P, c_int, c_int64, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "evi_dde_node_struct": (c_int, [_P, c_int, c_int, c_int64, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "evi_dde_node_struct_graphs": (c_int, [_P, c_int, c_int, c_int64, _P, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "evi_gemm_nt_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int, c_int64, _P, c_int, _P, c_int64, _P]),
    "evi_gemm_nt_bf16x3_workspace_bytes": (c_size_t, [c_int, c_int]),
    "evi_gemm_nt_bf16x3": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int, c_int64, _P, c_int, _P, c_int64, _P, c_size_t, _P]),
    "evi_retriever_prepare_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "evi_retriever_prepare": (c_int, [_P, _P, c_size_t, _P]),
    "evi_retriever_backward_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64]),
    "evi_gemm_tn_bf16x3_workspace_bytes": (c_size_t, [c_int, c_int]),
    "evi_gemm_tn_bf16x3": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, c_int64, _P, c_int, _P, c_size_t, _P]),
    "evi_grad_norm_workspace_bytes": (c_size_t, [c_int64]),
    "evi_grad_norm": (c_int, [_P, c_int64, c_float, _P, _P, c_size_t, _P]),
    "evi_adamw_step": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64, c_float, _P, c_float, _P]),
    "evi_retriever_saved_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "evi_retriever_saved_bytes_full": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int]),
    "evi_retriever_backward": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, c_size_t, _P, c_size_t, _P]),
    "evi_retriever_forward_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64]),
    "evi_retriever_forward": (c_int, [POINTER(EviRetrieverWeights), POINTER(EviRetrieverBatch), c_int,
                                      POINTER(EviRetrieverOutput), _P, c_size_t, _P]),
}

_lib: Optional[ctypes.CDLL] = None


class EviLibraryError(RuntimeError):
    """The HIP extension is missing or incomplete."""


def header_symbols(header_path: str = HEADER_PATH) -> List[str]:
    """Every function name include/evi_hip.h declares."""
    with open(header_path, "r", encoding="utf-8") as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evi_[a-z0-9_]+)\s*\(", text)))


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EviLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C evi_rag_amd/csrc` (there is no CPU fallback)."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise EviLibraryError(f"{LIB_PATH} does not export {name}; rebuild the extension.") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.evi_version() != 1:
        raise EviLibraryError(f"ABI version mismatch: library {lib.evi_version()}, binding 1")
    _lib = lib
    return lib


def last_error() -> str:
    lib = load()
    buf = ctypes.create_string_buffer(2048)
    lib.evi_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(status: int) -> None:
    """Map a status code to the reference's exception conventions."""
    if status == EVI_OK:
        return
    msg = last_error() or f"evi_hip status {status}"
    if status == EVI_ERR_INVALID:
        raise ValueError(msg)
    if status == EVI_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if status == EVI_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def bound_symbols() -> Dict[str, tuple]:
    return dict(_SIGNATURES)
