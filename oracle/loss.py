"""Retriever loss oracle (test infrastructure only).

reference: RetrieverLoss, src/losses/retriever_loss.py:29-325 (multi-positive InfoNCE over each
graph's edges + optional per-graph BCE, optional near / bridge edge weights), called on the eval path
by RetrieverModule._shared_eval_step (src/models/retriever_module.py:410-437).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

MIN_EDGE_WEIGHT = 1e-6


def _segment(fn, init, edge_batch, values, num_graphs):
    out = np.full(num_graphs, init, dtype=values.dtype)
    fn.at(out, edge_batch, values)
    return out


def retriever_loss(logits: np.ndarray, targets: np.ndarray, edge_batch: np.ndarray, num_graphs: int, *,
                   infonce_temperature: float = 1.0, infonce_weight: float = 1.0, bce_weight: float = 0.0,
                   edge_weight_near: float = 1.0, edge_weight_bridge: float = 1.0,
                   edge_is_near: Optional[np.ndarray] = None) -> Tuple[float, Dict[str, float], Dict[str, float], np.ndarray]:
    """(total loss, components, metrics, d total / d logits), f32 arithmetic as the reference (:72-143,
    :145-180, :228-325)."""
    x = np.asarray(logits, np.float32).reshape(-1)
    t = np.asarray(targets, np.float32).reshape(-1)
    eb = np.asarray(edge_batch, np.int64).reshape(-1)
    E, B = x.size, int(num_graphs)
    w = None
    if edge_weight_near != 1.0 or edge_weight_bridge != 1.0:
        w = np.where(np.asarray(edge_is_near, bool), np.float32(edge_weight_near), np.float32(edge_weight_bridge)).astype(np.float32)
    pos = t > 0.5
    grad = np.zeros(E, np.float64)
    pos_count, neg_count = int(pos.sum()), int((~pos).sum())
    metrics: Dict[str, float] = {}
    infonce = 0.0
    info_metrics = {"infonce_pos_edges": float(pos_count), "infonce_neg_edges": float(neg_count), "infonce_graphs": 0.0}
    if pos_count > 0 and neg_count > 0:
        s = x / np.float32(infonce_temperature)
        if w is not None:
            s = s + np.log(np.maximum(w, np.float32(MIN_EDGE_WEIGHT)))
        s = s.astype(np.float32)
        max_all = _segment(np.maximum, -np.inf, eb, s, B)
        max_pos = _segment(np.maximum, -np.inf, eb, np.where(pos, s, -np.inf).astype(np.float32), B)
        exp_all = np.exp(s - max_all[eb]).astype(np.float32)
        sum_all = _segment(np.add, 0.0, eb, exp_all, B)
        with np.errstate(invalid="ignore"):
            exp_pos = np.where(pos, np.exp(s - max_pos[eb]), 0.0).astype(np.float32)
        sum_pos = _segment(np.add, 0.0, eb, exp_pos, B)
        with np.errstate(invalid="ignore"):
            lse_all = max_all + np.log(np.maximum(sum_all, np.float32(1e-12)))
            lse_pos = max_pos + np.log(np.maximum(sum_pos, np.float32(1e-12)))
        pos_counts = _segment(np.add, 0.0, eb, pos.astype(np.float32), B)
        edge_counts = _segment(np.add, 0.0, eb, np.ones(E, np.float32), B)
        neg_counts = edge_counts - pos_counts
        valid = (pos_counts > 0) & (neg_counts > 0)
        info_metrics["infonce_graphs_no_pos"] = float((pos_counts == 0).sum())
        info_metrics["infonce_graphs_no_neg"] = float((neg_counts == 0).sum())
        if valid.any():
            infonce = float((lse_all - lse_pos)[valid].astype(np.float32).mean())
            info_metrics["infonce_graphs"] = float(valid.sum())
            nv = float(valid.sum())
            p_all = exp_all / sum_all[eb]
            with np.errstate(invalid="ignore", divide="ignore"):
                p_pos = np.where(pos, exp_pos / sum_pos[eb], 0.0)
            g = (p_all.astype(np.float64) - p_pos.astype(np.float64)) / (nv * float(infonce_temperature))
            grad += float(infonce_weight) * np.where(valid[eb], g, 0.0)
    bce = 0.0
    bce_metrics = {"bce_graphs": 0.0, "bce_edges": 0.0}
    if bce_weight > 0.0:
        per_edge = (np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))).astype(np.float32)
        if w is not None:
            per_edge = per_edge * w
        loss_sum = _segment(np.add, 0.0, eb, per_edge, B)
        edge_counts = _segment(np.add, 0.0, eb, np.ones(E, np.float32), B)
        if w is not None:
            weight_sum = _segment(np.add, 0.0, eb, w, B)
            valid = weight_sum > 0
            denom = np.maximum(weight_sum, np.float32(MIN_EDGE_WEIGHT))
        else:
            valid = edge_counts > 0
            denom = edge_counts
        bce_metrics["bce_edges"] = float(E)
        if valid.any():
            bce = float((loss_sum[valid] / denom[valid]).astype(np.float32).mean())
            bce_metrics["bce_graphs"] = float(valid.sum())
            sig = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
            ge = (sig - t) * (w if w is not None else 1.0) / np.where(valid, denom, 1.0)[eb] / float(valid.sum())
            grad += float(bce_weight) * np.where(valid[eb], ge, 0.0)
    probs = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    pos_avg = float(probs[pos].mean()) if pos.any() else 0.0
    neg_avg = float(probs[~pos].mean()) if (~pos).any() else 0.0
    total = float(np.float32(infonce_weight) * np.float32(infonce) + np.float32(bce_weight) * np.float32(bce))
    components = {"infonce": infonce, "infonce_weight": float(infonce_weight), "bce": bce, "bce_weight": float(bce_weight),
                  "path": 0.0, "path_weight": 0.0}
    metrics = {"pos_prob": pos_avg, "neg_prob": neg_avg, "separation": pos_avg - neg_avg, **info_metrics, **bce_metrics,
               "path_graphs": 0.0}
    return total, components, metrics, grad.astype(np.float32)
