"""Ranking order and exact top-k on the CPU (oracle; test infrastructure only).

The reference's only *defined* ranking is `argsort(scores, descending=True, stable=True)`
(src/data/components/g_agent_builder.py:651): score descending, position ascending among equal
scores.  `torch.topk` (src/metrics/reachability.py:147, src/metrics/retriever_metrics.py:145,
src/callbacks/retriever_topk_edge_writer.py:302) leaves tie order unspecified, so the build adopts
the stable order everywhere.  -0.0 ties with +0.0; NaN ranks first (torch's convention).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def _rank_keys(scores: np.ndarray) -> np.ndarray:
    """float32 -> uint32 whose ascending order is the ascending float order (NaN largest)."""
    s = np.ascontiguousarray(scores, dtype=np.float32) + np.float32(0.0)
    u = s.view(np.uint32)
    neg = (u & np.uint32(0x80000000)) != 0
    key = np.where(neg, ~u, u | np.uint32(0x80000000)).astype(np.uint32)
    key[np.isnan(s)] = np.uint32(0xFFFFFFFF)
    return key


def stable_desc_order(scores: np.ndarray) -> np.ndarray:
    """Permutation sorting by (score desc, position asc) — argsort(descending=True, stable=True)."""
    key = _rank_keys(scores).astype(np.int64)
    return np.argsort(-key, kind="stable")


def topk_desc(scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """(values, positions) of the min(k, n) best entries in (score desc, position asc) order."""
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = scores.shape[0]
    m = min(int(k), n)
    if m <= 0:
        return np.empty(0, np.float32), np.empty(0, np.int64)
    key = _rank_keys(scores).astype(np.int64)
    if m < n:
        # partition first so that large n stays O(n)
        kth = np.partition(key, n - m)[n - m]
        above = np.nonzero(key > kth)[0]
        equal = np.nonzero(key == kth)[0][: m - above.shape[0]]
        cand = np.concatenate([above, equal])
    else:
        cand = np.arange(n, dtype=np.int64)
    order = np.lexsort((cand, -key[cand]))
    pos = cand[order].astype(np.int64)
    return scores[pos], pos


def segment_topk(scores: np.ndarray, edge_ptr: np.ndarray, k: int):
    """Per-graph top-k of edge scores as LOCAL positions (-1 / -inf padding) + counts.
    reference: src/metrics/reachability.py:146-147; src/metrics/retriever_metrics.py:141-145;
    src/callbacks/retriever_topk_edge_writer.py:299-302; src/data/components/g_agent_builder.py:640-652."""
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    edge_ptr = np.asarray(edge_ptr, dtype=np.int64).reshape(-1)
    B = edge_ptr.shape[0] - 1
    out_idx = np.full((B, k), -1, dtype=np.int32)
    out_val = np.full((B, k), -np.inf, dtype=np.float32)
    out_cnt = np.zeros(B, dtype=np.int32)
    for g in range(B):
        lo, hi = int(edge_ptr[g]), int(edge_ptr[g + 1])
        if hi <= lo:
            continue
        vals, pos = topk_desc(scores[lo:hi], k)
        m = pos.shape[0]
        out_idx[g, :m] = pos
        out_val[g, :m] = vals
        out_cnt[g] = m
    return out_idx, out_val, out_cnt


def merge_topk(scores: np.ndarray, ids: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge [P, Q, k] per-shard lists into [Q, k] by (score desc, id asc); id < 0 is padding.
    Build-side replacement of dist.all_gather_object at
    src/callbacks/retriever_topk_edge_writer.py:450-462."""
    scores = np.asarray(scores, dtype=np.float32)
    ids = np.asarray(ids, dtype=np.int64)
    P, Q, kk = scores.shape
    out_s = np.full((Q, k), -np.inf, dtype=np.float32)
    out_i = np.full((Q, k), -1, dtype=np.int64)
    for q in range(Q):
        s = scores[:, q, :].reshape(-1)
        i = ids[:, q, :].reshape(-1)
        valid = i >= 0
        s, i = s[valid], i[valid]
        key = _rank_keys(s).astype(np.int64)
        order = np.lexsort((i, -key))[:k]
        out_s[q, : order.shape[0]] = s[order]
        out_i[q, : order.shape[0]] = i[order]
    return out_s, out_i
