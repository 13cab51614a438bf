"""GAgentBuilder sample materialisation oracle (test infrastructure only).

reference: GAgentBuilder.process_batch / _build_and_add_sample,
src/data/components/g_agent_builder.py:158-512.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from .graph import node_softmax_logit, select_start_edges, select_topk_edges


def build_sample(*, heads: np.ndarray, tails: np.ndarray, relations: np.ndarray, labels: np.ndarray, scores: np.ndarray,
                 node_global_ids: np.ndarray, node_embedding_ids: np.ndarray, start_entity_ids: np.ndarray,
                 answer_entity_ids: np.ndarray, edge_top_k: int, start_keep_ratio: float, start_min_edges: int,
                 start_max_edges: Optional[int], allow_empty_answer: bool, node_softmax: bool) -> Optional[Dict[str, np.ndarray]]:
    """One retrieval graph (local head/tail ids, calibrated logits) -> the g_agent sample arrays, or
    None when the reference drops the sample ("retrieval_failed").  start_max_edges None = edge_top_k
    (GAgentSettings.__post_init__, :72-75).  reference :238-512."""
    E = int(heads.shape[0])
    if E <= 0:
        return None
    n = int(node_global_ids.shape[0])
    if start_max_edges is None:
        start_max_edges = int(edge_top_k)
    start_mask = np.isin(node_global_ids, np.asarray(start_entity_ids, np.int64))
    if not start_mask.any():
        raise ValueError("Start entities missing from retrieval graph")
    start_locals = np.nonzero(start_mask)[0]
    norm = (lambda s, h, t, m: node_softmax_logit(s, h, t, m)) if node_softmax else (lambda s, h, t, m: np.asarray(s, np.float32))
    select_scores = norm(scores, heads, tails, n)
    topk = select_topk_edges(select_scores, edge_top_k)
    start = select_start_edges(heads, tails, select_scores, start_locals, n, start_keep_ratio, start_min_edges, start_max_edges)
    if topk.size == 0:
        return None
    env = np.unique(np.concatenate([topk, start]))
    # (h, r, t) dedup over GLOBAL ids in first-seen (ascending edge id) order; score / label = max (:338-354)
    agg: Dict[tuple, List[float]] = {}
    for e in env.tolist():
        key = (int(node_global_ids[heads[e]]), int(relations[e]), int(node_global_ids[tails[e]]))
        s, l = float(scores[e]), float(labels[e])
        if key not in agg:
            agg[key] = [s, l]
        else:
            agg[key][0] = max(agg[key][0], s)
            agg[key][1] = max(agg[key][1], l)
    keys = list(agg.keys())
    hg = np.asarray([k[0] for k in keys], np.int64)
    rel = np.asarray([k[1] for k in keys], np.int64)
    tg = np.asarray([k[2] for k in keys], np.int64)
    e_scores = np.asarray([agg[k][0] for k in keys], np.float32)
    e_labels = np.asarray([agg[k][1] for k in keys], np.float32)
    node_entity_ids = np.unique(np.concatenate([hg, tg]))  # sorted
    lookup = {int(g): int(e) for g, e in zip(node_global_ids.tolist(), node_embedding_ids.tolist())}
    node_emb = np.asarray([lookup[int(g)] for g in node_entity_ids], np.int64)
    h_loc = np.searchsorted(node_entity_ids, hg).astype(np.int64)
    t_loc = np.searchsorted(node_entity_ids, tg).astype(np.int64)
    e_scores = norm(e_scores, h_loc, t_loc, int(node_entity_ids.size))
    node_map = {int(g): i for i, g in enumerate(node_entity_ids.tolist())}
    start_list = [node_map[int(g)] for g in np.asarray(start_entity_ids).tolist() if int(g) in node_map]
    if not start_list:
        return None
    start_node_locals = np.asarray(list(dict.fromkeys(start_list)), np.int64)
    answers = np.asarray(list(dict.fromkeys(int(a) for a in np.asarray(answer_entity_ids).tolist())), np.int64)
    answer_node_locals = np.asarray([node_map[int(a)] for a in answers.tolist() if int(a) in node_map], np.int64)
    dummy = False
    if answer_node_locals.size == 0:
        if not allow_empty_answer:
            return None
        e_labels = np.zeros(rel.size, np.float32)
        dummy = True
    return {"edge_relations": rel, "edge_scores": e_scores.astype(np.float32), "edge_labels": e_labels,
            "edge_head_locals": h_loc, "edge_tail_locals": t_loc, "node_entity_ids": node_entity_ids,
            "node_embedding_ids": node_emb, "start_entity_ids": np.asarray(start_entity_ids, np.int64),
            "answer_entity_ids": answers, "start_node_locals": start_node_locals, "answer_node_locals": answer_node_locals,
            "flags": np.asarray([False, (not dummy) and answer_node_locals.size > 0, dummy], bool)}
