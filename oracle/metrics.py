"""Metric oracle (T1, T2, T4, T5): per-graph top-k metrics restated in numpy (test infrastructure
only).  Top-k order is (score desc, position asc) — see oracle/ranking.py.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from . import graph as ograph
from .ranking import stable_desc_order, topk_desc


def normalize_k_values(raw: Sequence[int]) -> List[int]:
    """reference: normalize_k_values, src/utils/metrics.py:25-40."""
    out, seen = [], set()
    for item in raw or []:
        try:
            k = int(item)
        except (TypeError, ValueError):
            continue
        if k <= 0 or k in seen:
            continue
        out.append(k)
        seen.add(k)
    return sorted(out)


# ---- T1 ---------------------------------------------------------------------------------------------------
def edge_recall_at_k(scores: np.ndarray, target: np.ndarray, edge_ptr: np.ndarray, k_values: Sequence[int]):
    """Returns (recall_sum per k, graph_count): per graph with >= 1 edge, recall@k =
    cumhits[min(k, E_g) - 1] / max(pos, 1).
    reference: EdgeRecallAtK, src/metrics/retriever_metrics.py:83-166."""
    ks = normalize_k_values(k_values)
    sums = {k: 0.0 for k in ks}
    count = 0.0
    if not ks:
        return sums, count
    max_k = max(ks)
    scores = np.asarray(scores, np.float32).reshape(-1)
    target = np.asarray(target).reshape(-1).astype(bool)
    for g in range(len(edge_ptr) - 1):
        lo, hi = int(edge_ptr[g]), int(edge_ptr[g + 1])
        if hi <= lo:
            continue
        lab = target[lo:hi]
        _, pos = topk_desc(scores[lo:hi], max_k)
        cum = np.cumsum(lab[pos].astype(np.float32), dtype=np.float32)
        denom = np.float32(max(float(lab.sum()), 1.0))
        for k in ks:
            k_eff = min(k, pos.shape[0])
            sums[k] += float(np.float32(cum[k_eff - 1]) / denom) if k_eff > 0 else 0.0
        count += 1.0
    return sums, count


def edge_recall_compute(sums, count) -> Dict[str, float]:
    return {f"edge/recall@{k}": v / max(count, 1.0) for k, v in sums.items()}


# ---- T2 ---------------------------------------------------------------------------------------------------
class _UnionFind:
    """reference: AnswerReachability._uf_*, src/metrics/reachability.py:296-328."""

    def __init__(self, n: int):
        self.parent = list(range(n))
        self.rank = [0] * n

    def find(self, x: int) -> int:
        p = self.parent
        while p[x] != x:
            p[x] = p[p[x]]
            x = p[x]
        return x

    def union(self, a: int, b: int) -> None:
        pa, pb = self.find(a), self.find(b)
        if pa == pb:
            return
        if self.rank[pa] < self.rank[pb]:
            self.parent[pa] = pb
        elif self.rank[pa] > self.rank[pb]:
            self.parent[pb] = pa
        else:
            self.parent[pb] = pa
            self.rank[pa] += 1


def reachability_at_k(edge_index_local: np.ndarray, top_idx: np.ndarray, start_nodes, answer_nodes, num_nodes: int,
                      k_values: Sequence[int]) -> Dict[int, bool]:
    """Undirected union-find over the ranked edges with a checkpoint at each min(k, k_top).
    reference: AnswerReachability._compute_reachability_at_k, src/metrics/reachability.py:330-381."""
    ks = [int(k) for k in k_values if int(k) > 0]
    if num_nodes <= 0 or not ks:
        return {}
    start_nodes = [int(s) for s in np.asarray(start_nodes).reshape(-1)]
    answer_nodes = [int(a) for a in np.asarray(answer_nodes).reshape(-1)]
    if not start_nodes or not answer_nodes:
        return {k: False for k in ks}
    k_top = min(int(len(top_idx)), max(ks))
    if k_top <= 0:
        return {k: False for k in ks}
    uf = _UnionFind(num_nodes)

    def reachable() -> bool:
        roots = {uf.find(s) for s in start_nodes}
        return any(uf.find(a) in roots for a in answer_nodes)

    k_check = sorted({min(k, k_top) for k in ks})
    reach: Dict[int, bool] = {}
    nxt = 0
    for i in range(k_top):
        u, v = int(edge_index_local[0, top_idx[i]]), int(edge_index_local[1, top_idx[i]])
        if 0 <= u < num_nodes and 0 <= v < num_nodes:
            uf.union(u, v)
        while nxt < len(k_check) and i + 1 >= k_check[nxt]:
            reach[k_check[nxt]] = reachable()
            nxt += 1
    while nxt < len(k_check):
        reach[k_check[nxt]] = reachable()
        nxt += 1
    return {k: reach[min(k, k_top)] for k in ks}


def answer_reachability(scores: np.ndarray, batch, k_values: Sequence[int]):
    """Returns (hits per k, valid graph count).  A graph counts only if it has edges, nodes, and at
    least one in-range seed AND answer.
    reference: AnswerReachability.update/_accumulate_hits/_compute_graph_reachability,
    src/metrics/reachability.py:129-296."""
    ks = normalize_k_values(k_values)
    hits = {k: 0.0 for k in ks}
    valid = 0.0
    if not ks:
        return hits, valid
    max_k = max(ks)
    scores = np.asarray(scores, np.float32).reshape(-1)
    edge_index = np.asarray(batch.edge_index, np.int64)
    for g in range(batch.num_graphs):
        lo, hi = int(batch.edge_ptr[g]), int(batch.edge_ptr[g + 1])
        if hi <= lo:
            continue
        q = np.asarray(batch.q_local_indices[int(batch.q_ptr[g]): int(batch.q_ptr[g + 1])], np.int64)
        a = np.asarray(batch.a_local_indices[int(batch.a_ptr[g]): int(batch.a_ptr[g + 1])], np.int64)
        if q.size == 0 or a.size == 0:
            continue
        n0, n1 = int(batch.ptr[g]), int(batch.ptr[g + 1])
        if n1 - n0 <= 0:
            continue
        ql = q[(q >= n0) & (q < n1)] - n0
        al = a[(a >= n0) & (a < n1)] - n0
        if ql.size == 0 or al.size == 0:
            continue
        _, top = topk_desc(scores[lo:hi], max_k)
        reach = reachability_at_k(edge_index[:, lo:hi] - n0, top, ql, al, n1 - n0, ks)
        if not reach:
            continue
        valid += 1.0
        for k, r in reach.items():
            if r:
                hits[k] += 1.0
    return hits, valid


def answer_reachability_compute(hits, valid) -> Dict[str, float]:
    return {f"answer/reachability@{k}": v / max(valid, 1.0) for k, v in hits.items()}


# ---- T4 ---------------------------------------------------------------------------------------------------
def oracle_metrics_for_sample(head_entity_ids, tail_entity_ids, answer_entity_ids, k_values: Sequence[int]) -> Dict[str, float]:
    """Hits@k / answer-recall@k over a RANKED edge list (rank 1 first).
    reference: _oracle_metrics_for_sample, src/models/reasoner_module.py:17-68."""
    answers = np.unique(np.asarray(answer_entity_ids, np.int64).reshape(-1))
    heads = np.asarray(head_entity_ids, np.int64).reshape(-1)
    tails = np.asarray(tail_entity_ids, np.int64).reshape(-1)
    ks = list(k_values)
    out: Dict[str, float] = {}
    if answers.size == 0 or heads.size == 0:
        for k in ks:
            out[f"answer_hit@{k}"] = 0.0
            out[f"answer_recall@{k}"] = 0.0
        return out
    ans = set(answers.tolist())
    found = set()
    max_scan = min(heads.size, max(int(k) for k in ks) if ks else 0)
    kp = 0
    for rank in range(1, max_scan + 1):
        e = rank - 1
        if int(heads[e]) in ans:
            found.add(int(heads[e]))
        if int(tails[e]) in ans:
            found.add(int(tails[e]))
        while kp < len(ks) and rank == int(ks[kp]):
            out[f"answer_hit@{int(ks[kp])}"] = 1.0 if found else 0.0
            out[f"answer_recall@{int(ks[kp])}"] = float(len(found) / answers.size)
            kp += 1
    while kp < len(ks):
        out[f"answer_hit@{int(ks[kp])}"] = 1.0 if found else 0.0
        out[f"answer_recall@{int(ks[kp])}"] = float(len(found) / answers.size)
        kp += 1
    return out


def answer_hit_recall_batch(scores: np.ndarray, batch, k_values: Sequence[int]):
    """Mean answer_hit@k and answer_recall@k over the graphs of a batch that have answers, ranking
    each graph's edges by (score desc, position asc) and mapping endpoints to global entity ids.
    reference: compute_answer_hit (src/utils/metrics.py:206-238), compute_answer_recall (:167-203)."""
    ks = normalize_k_values(k_values)
    hit = {k: [] for k in ks}
    rec = {k: [] for k in ks}
    if not ks:
        return {}, {}
    max_k = max(ks)
    scores = np.asarray(scores, np.float32).reshape(-1)
    gid = np.asarray(batch.node_global_ids, np.int64)
    ei = np.asarray(batch.edge_index, np.int64)
    for g in range(batch.num_graphs):
        ans = np.asarray(batch.answer_entity_ids[int(batch.answer_ptr[g]): int(batch.answer_ptr[g + 1])], np.int64)
        if ans.size == 0:
            continue
        lo, hi = int(batch.edge_ptr[g]), int(batch.edge_ptr[g + 1])
        _, top = topk_desc(scores[lo:hi], max_k)
        heads = gid[ei[0, lo:hi]][top]
        tails = gid[ei[1, lo:hi]][top]
        aset = set(ans.tolist())
        found = set()
        kp = 0
        for rank in range(1, top.shape[0] + 1):
            for x in (int(heads[rank - 1]), int(tails[rank - 1])):
                if x in aset:
                    found.add(x)
            while kp < len(ks) and rank == ks[kp]:
                hit[ks[kp]].append(1.0 if found else 0.0)
                rec[ks[kp]].append(len(found) / len(aset))
                kp += 1
        while kp < len(ks):
            hit[ks[kp]].append(1.0 if found else 0.0)
            rec[ks[kp]].append(len(found) / len(aset))
            kp += 1
    hits = {f"answer_hit@{k}": (sum(v) / len(v) if v else 0.0) for k, v in hit.items()}
    recs = {f"answer_recall@{k}": (sum(v) / len(v) if v else 0.0) for k, v in rec.items()}
    return hits, recs


# ---- T5 ---------------------------------------------------------------------------------------------------
def _graph_slices(edge_ptr):
    for g in range(len(edge_ptr) - 1):
        yield int(edge_ptr[g]), int(edge_ptr[g + 1])


def score_margin(scores, target, edge_ptr) -> Dict[str, float]:
    """mean over graphs having both classes of (min positive score - max negative score).
    reference: ScoreMargin, src/metrics/retriever_metrics.py:330-397."""
    scores = np.asarray(scores, np.float32).reshape(-1)
    target = np.asarray(target).reshape(-1).astype(bool)
    total, cnt = np.float32(0.0), 0.0
    for lo, hi in _graph_slices(edge_ptr):
        lab = target[lo:hi]
        if hi <= lo or not lab.any() or lab.all():
            continue
        s = scores[lo:hi]
        total = np.float32(total + (s[lab].min() - s[~lab].max()))
        cnt += 1.0
    return {"edge/score_margin": float(total) / max(cnt, 1.0)}


def bridge_metrics(scores, target, batch, k_values: Sequence[int]) -> Dict[str, float]:
    """bridge = edges touching neither a seed nor an answer node.
    reference: BridgeEdgeRecallAtK (:169-267), BridgePositiveCoverage (:270-327), BridgeProbQuality
    (:400-476), _compute_bridge_mask (:66-80), src/metrics/retriever_metrics.py."""
    ks = normalize_k_values(k_values)
    scores = np.asarray(scores, np.float32).reshape(-1)
    target = np.asarray(target).reshape(-1).astype(bool)
    near = ograph.compute_qa_edge_mask(batch.edge_index, batch.num_nodes, batch.q_local_indices, batch.a_local_indices)
    bridge = ~near
    eb, _ = ograph.compute_edge_batch(batch.edge_index, batch.ptr, batch.num_graphs)
    out: Dict[str, float] = {}
    # recall@k restricted to bridge edges, graphs with >= 1 bridge positive
    sums = {k: 0.0 for k in ks}
    cnt = 0.0
    pos_sum = neg_sum = sep_sum = np.float32(0.0)
    q_cnt = 0.0
    for g in range(batch.num_graphs):
        sel = bridge & (eb == g)
        if not sel.any():
            continue
        s, lab = scores[sel], target[sel]
        if lab.any() and ks:
            _, top = topk_desc(s, max(ks))
            cum = np.cumsum(lab[top].astype(np.float32), dtype=np.float32)
            denom = np.float32(max(float(lab.sum()), 1.0))
            for k in ks:
                sums[k] += float(np.float32(cum[min(k, top.shape[0]) - 1]) / denom)
            cnt += 1.0
        if lab.any() and not lab.all():
            p = (1.0 / (1.0 + np.exp(-s.astype(np.float64)))).astype(np.float32)
            pm, nm = p[lab].mean(dtype=np.float32), p[~lab].mean(dtype=np.float32)
            pos_sum = np.float32(pos_sum + pm)
            neg_sum = np.float32(neg_sum + nm)
            sep_sum = np.float32(sep_sum + (pm - nm))
            q_cnt += 1.0
    for k in ks:
        out[f"bridge/recall@{k}"] = sums[k] / max(cnt, 1.0)
    out["bridge/pos_prob"] = float(pos_sum) / max(q_cnt, 1.0)
    out["bridge/neg_prob"] = float(neg_sum) / max(q_cnt, 1.0)
    out["bridge/separation"] = float(sep_sum) / max(q_cnt, 1.0)
    total_pos = float(target.sum())
    bridge_pos = float((target & bridge).sum())
    gp = np.bincount(eb[target], minlength=batch.num_graphs) > 0
    gb = np.bincount(eb[target & bridge], minlength=batch.num_graphs) > 0
    out["bridge/pos_edge_frac"] = bridge_pos / max(total_pos, 1.0)
    out["bridge/pos_graph_frac"] = float((gp & gb).sum()) / max(float(gp.sum()), 1.0)
    return out


# ---- T4b: list-of-samples ranking statistics ------------------------------------------------------------------------
def _ndcg(ranked_labels: np.ndarray, k: int) -> float:
    """reference: _ndcg, src/utils/metrics.py:156-170 (f32 tensors, python-float ratio)."""
    trunc = ranked_labels[:k]
    if trunc.size == 0:
        return 0.0
    positions = np.arange(1, trunc.size + 1, dtype=np.float32)
    discounts = (np.float32(1.0) / np.log2(positions + np.float32(1.0))).astype(np.float32)
    dcg = float((trunc * discounts).sum(dtype=np.float32))
    ideal = np.sort(ranked_labels)[::-1][:k]
    ideal_dcg = float((ideal * discounts[: ideal.size]).sum(dtype=np.float32))
    return 0.0 if ideal_dcg <= 0 else dcg / ideal_dcg


def ranking_metrics(samples, k_values) -> Dict[str, object]:
    """Precision / recall / F1 / nDCG @k and MRR, averaged over the samples that have positives.
    reference: compute_ranking_metrics, src/utils/metrics.py:112-153.  `samples`: (scores, labels) pairs.  Hits and DCG sum
    label VALUES; the recall denominator is the label sum truncated to an int (:121, :136); a sample whose truncated label sum
    is <= 0 is skipped.  Order: (score desc, position asc) — the reference's argsort is not stable, its fixtures avoid ties."""
    ks = normalize_k_values(k_values) or [1]
    tot = {k: {"precision": 0.0, "recall": 0.0, "f1": 0.0, "ndcg": 0.0, "count": 0.0} for k in ks}
    mrr_sum, mrr_count = 0.0, 0
    for scores, labels in samples:
        scores = np.asarray(scores, np.float32).reshape(-1)
        labels = np.asarray(labels, np.float32).reshape(-1)
        positives = int(labels.sum(dtype=np.float32))
        if positives <= 0:
            continue
        ranked = labels[stable_desc_order(scores)]
        pos = np.nonzero(ranked > 0.5)[0]
        if pos.size:
            mrr_sum += 1.0 / float(pos[0] + 1)
            mrr_count += 1
        for k in ks:
            hits = float(ranked[:k].sum(dtype=np.float32))
            precision, recall = hits / float(k), hits / float(positives)
            st = tot[k]
            st["precision"] += precision
            st["recall"] += recall
            st["f1"] += 0.0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)
            st["ndcg"] += _ndcg(ranked, k)
            st["count"] += 1.0
    out = {name: {k: tot[k][name] / (tot[k]["count"] or 1.0) for k in ks} for name in ("precision", "recall", "f1", "ndcg")}
    out["mrr"] = mrr_sum / mrr_count if mrr_count else 0.0
    return out
