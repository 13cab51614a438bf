"""Cosine normalisation / scoring oracle (test infrastructure only).

reference: scripts/build_retrieval_pipeline.py:833-874 (_normalize_embeddings,
_group_positive_edges_by_pair, _select_canonical_edge_indices).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from .ranking import topk_desc


def normalize_embeddings(embeddings: np.ndarray, eps: float) -> np.ndarray:
    """x / clamp(||x||_2, min=eps) row-wise in f32; empty arrays pass through.
    reference: _normalize_embeddings, scripts/build_retrieval_pipeline.py:833-837."""
    x = np.asarray(embeddings, dtype=np.float32)
    if x.size == 0:
        return x
    norm = np.sqrt(np.sum(x.astype(np.float32) * x, axis=-1, keepdims=True, dtype=np.float32))
    denom = np.maximum(norm, np.float32(eps)).astype(np.float32)
    return (x / denom).astype(np.float32)


def row_inv_norm(embeddings: np.ndarray, eps: float) -> np.ndarray:
    x = np.asarray(embeddings, dtype=np.float32)
    norm = np.sqrt(np.sum(x * x, axis=-1, dtype=np.float32))
    return (np.float32(1.0) / np.maximum(norm, np.float32(eps))).astype(np.float32)


def e4m3_decode_table() -> np.ndarray:
    """All 256 OCP e4m3 ("e4m3fn") codes as f32: 1 sign, 4 exponent (bias 7), 3 mantissa bits;
    exponent 0 is subnormal (m / 8 * 2^-6); 0x7F / 0xFF are NaN; no infinities; max finite 448.
    (The fp8-storage index is this framework's own extension — BASELINE config 5 — so the codec is
    restated from the OCP 8-bit floating point specification, not from the reference.)"""
    c = np.arange(256)
    e, m = (c >> 3) & 15, c & 7
    mag = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    mag = np.where((c & 0x7F) == 0x7F, np.nan, mag)
    return np.where(c & 0x80, -mag, mag).astype(np.float32)


def quantize_rows_e4m3(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-row symmetric e4m3 quantisation: scale = max|x| / 448 (1 for a zero row), code =
    nearest e4m3 value of x / scale, ties to the even code, saturating at +-448."""
    x = np.asarray(x, dtype=np.float32)
    mx = np.max(np.abs(x), axis=1) if x.shape[1] else np.zeros(x.shape[0], np.float32)
    scale = np.where(mx > 0, mx / np.float32(448.0), np.float32(1.0)).astype(np.float32)
    v = (x / scale[:, None]).astype(np.float32)
    pos = e4m3_decode_table()[:127].astype(np.float64)  # codes 0..126 ascend in value
    a = np.minimum(np.abs(v).astype(np.float64), 448.0)
    hi = np.clip(np.searchsorted(pos, a, side="left"), 1, 126)
    lo = hi - 1
    dl, dh = a - pos[lo], pos[hi] - a
    code = np.where(dl < dh, lo, np.where(dh < dl, hi, np.where(lo % 2 == 0, lo, hi)))
    code = np.where(a == 0, 0, code).astype(np.uint8)
    return (code | np.where(np.signbit(v), 0x80, 0).astype(np.uint8)), scale


def cosine_scores(queries: np.ndarray, index: np.ndarray, eps: float) -> np.ndarray:
    """[Q, N] f32 cosine of every query against every index row: normalise, then matmul — the
    generalisation of `torch.mv(rel_vecs, question_vec)` (build_retrieval_pipeline.py:871)."""
    qn = normalize_embeddings(queries, eps)
    xn = normalize_embeddings(index, eps)
    return (qn @ xn.T).astype(np.float32)


def cosine_topk(queries: np.ndarray, index: np.ndarray, k: int, eps: float = 1e-6,
                row_id_base: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Top-k index rows per query by cosine, ordered (score desc, row id asc); (-inf, -1) padding."""
    queries = np.asarray(queries, dtype=np.float32)
    index = np.asarray(index, dtype=np.float32)
    Q = queries.shape[0]
    out_s = np.full((Q, k), -np.inf, dtype=np.float32)
    out_i = np.full((Q, k), -1, dtype=np.int64)
    if index.shape[0] == 0:
        return out_s, out_i
    scores = cosine_scores(queries, index, eps)
    for q in range(Q):
        vals, pos = topk_desc(scores[q], k)
        out_s[q, : pos.shape[0]] = vals
        out_i[q, : pos.shape[0]] = pos + row_id_base
    return out_s, out_i


def dot_topk_prenormalized(qn: np.ndarray, xn: np.ndarray, k: int, row_scale=None,
                           row_id_base: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Top-k by plain dot product of already-normalised rows (what evi_cosine_topk is handed).
    Also returns the full [Q, N] score matrix for margin-aware set checks."""
    scores = (np.asarray(qn, np.float32) @ np.asarray(xn, np.float32).T).astype(np.float32)
    if row_scale is not None:
        scores = (scores * np.asarray(row_scale, np.float32)[None, :]).astype(np.float32)
    Q = scores.shape[0]
    out_s = np.full((Q, k), -np.inf, dtype=np.float32)
    out_i = np.full((Q, k), -1, dtype=np.int64)
    for q in range(Q):
        vals, pos = topk_desc(scores[q], k)
        out_s[q, : pos.shape[0]] = vals
        out_i[q, : pos.shape[0]] = pos + row_id_base
    return out_s, out_i, scores


# ---- C2-C4: canonical edge selection ----------------------------------------------------------------

def group_positive_edges_by_pair(edge_src: Sequence[int], edge_dst: Sequence[int],
                                 positive_mask: Sequence[bool]) -> Dict[Tuple[int, int], List[int]]:
    """Insertion-ordered groups of positive edge indices keyed by the unordered node pair.
    reference: _group_positive_edges_by_pair, scripts/build_retrieval_pipeline.py:840-853."""
    groups: Dict[Tuple[int, int], List[int]] = {}
    for idx, keep in enumerate(positive_mask):
        if not keep:
            continue
        u, v = int(edge_src[idx]), int(edge_dst[idx])
        key = (u, v) if u <= v else (v, u)
        groups.setdefault(key, []).append(idx)
    return groups


def select_canonical_edge_indices(groups: Dict[Tuple[int, int], List[int]], edge_relation_ids: Sequence[int],
                                  relation_embeddings_norm: np.ndarray, question_embedding_norm: np.ndarray) -> List[int]:
    """Per group keep the edge whose relation is most cosine-similar to the question; ties go to
    the first in (relation_id, idx) order.
    reference: _select_canonical_edge_indices, scripts/build_retrieval_pipeline.py:856-874."""
    qv = np.asarray(question_embedding_norm, np.float32).reshape(-1)
    rel = np.asarray(relation_embeddings_norm, np.float32)
    keep: List[int] = []
    for edge_indices in groups.values():
        if len(edge_indices) == 1:
            keep.append(edge_indices[0])
            continue
        ordered = sorted(edge_indices, key=lambda i: (int(edge_relation_ids[i]), i))
        rel_ids = np.asarray([int(edge_relation_ids[i]) for i in ordered], dtype=np.int64)
        scores = (rel[rel_ids] @ qv).astype(np.float32)
        keep.append(ordered[int(np.argmax(scores))])
    return keep


def canonicalize_positive_mask(edge_src, edge_dst, edge_relation_ids, positive_mask, pair_edge_local_ids,
                               pair_edge_counts, question_embedding_norm, relation_embeddings_norm):
    """reference: _canonicalize_graph_edges + _filter_pair_edges,
    scripts/build_retrieval_pipeline.py:877-932.  Returns (keep_mask, new_pair_ids, new_pair_counts)."""
    n = len(edge_src)
    groups = group_positive_edges_by_pair(edge_src, edge_dst, positive_mask)
    if not groups:
        return list(positive_mask), list(pair_edge_local_ids), list(pair_edge_counts)
    keep = select_canonical_edge_indices(groups, edge_relation_ids, relation_embeddings_norm, question_embedding_norm)
    keep_mask = [False] * n
    for i in keep:
        keep_mask[i] = True
    if not pair_edge_local_ids or not pair_edge_counts:
        return keep_mask, list(pair_edge_local_ids), list(pair_edge_counts)
    new_ids: List[int] = []
    new_counts: List[int] = []
    off = 0
    for c in pair_edge_counts:
        span = pair_edge_local_ids[off: off + c]
        kept = [i for i in span if keep_mask[int(i)]]
        new_ids.extend(kept)
        new_counts.append(len(kept))
        off += c
    if off != len(pair_edge_local_ids):
        raise ValueError("pair_edge_counts do not sum to len(pair_edge_local_ids)")
    return keep_mask, new_ids, new_counts
