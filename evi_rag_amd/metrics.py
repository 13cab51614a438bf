"""Host-side mirrors of the reference's retriever metrics, backed by `evi_retriever_metrics`.

Same class names, constructor kwargs, `update(...)` keyword contract, `compute()` keys and
`reset()` as src/metrics/reachability.py and src/metrics/retriever_metrics.py (torchmetrics-style;
`RetrieverModule._update_metrics` calls `update(preds=, target=, indexes=, batch=, query_ids=,
num_graphs=, features=)`, src/models/retriever_module.py:146-176).  States are f64 scalars summed
in graph order; `sync()` all-reduces them (the reference's `dist_reduce_fx="sum"`).

One fused kernel pass per batch ranks every graph once; the metric objects of one
`RetrieverMetricCollection` share that pass.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import _lib, ops


def normalize_k_values(raw_values: Any, default: Optional[Sequence[int]] = None) -> List[int]:
    """reference: normalize_k_values, src/utils/metrics.py:25-40."""
    if raw_values is None:
        items: list = []
    elif isinstance(raw_values, (list, tuple, set, range)):
        items = list(raw_values)
    elif isinstance(raw_values, str):
        items = [raw_values]
    else:
        try:
            items = list(raw_values)
        except TypeError:
            items = [raw_values]
    out, seen = [], set()
    for item in items:
        try:
            k = int(item)
        except (TypeError, ValueError):
            continue
        if k <= 0 or k in seen:
            continue
        out.append(k)
        seen.add(k)
    if not out and default is not None:
        return normalize_k_values(default)
    return sorted(out)


def _attr(obj: Any, name: str) -> Any:
    return obj.get(name) if isinstance(obj, dict) else getattr(obj, name, None)


class RankedBatch:
    """Per-graph results of one `evi_retriever_metrics` pass (device tensors)."""

    def __init__(self, k_values, **tensors):
        self.k_values = list(k_values)
        self.__dict__.update(tensors)


def rank_batch(preds: torch.Tensor, target: Optional[torch.Tensor], batch: Any, k_values: Sequence[int], *,
               num_graphs: Optional[int] = None, indexes: Optional[torch.Tensor] = None,
               want_topk: bool = False) -> RankedBatch:
    """Run the fused ranking-metrics kernel on one batch."""
    ks = normalize_k_values(k_values)
    if not ks:
        raise ValueError("k_values must contain at least one positive integer")
    scores = preds.detach().reshape(-1)
    dev = ops._require_gpu(scores)
    scores = ops._f32c(scores, "preds")
    edge_index = _attr(batch, "edge_index")
    node_ptr = _attr(batch, "ptr")
    if edge_index is None:
        raise ValueError("Batch missing edge_index required for reachability metrics.")
    if node_ptr is None:
        raise ValueError("Batch missing ptr required for reachability metrics.")
    i64 = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.long).contiguous().view(-1)  # noqa: E731
    edge_index = torch.as_tensor(edge_index).to(device=dev, dtype=torch.long).contiguous()
    node_ptr = i64(node_ptr)
    B = int(num_graphs) if num_graphs is not None else int(node_ptr.numel() - 1)
    E = int(scores.numel())
    if edge_index.size(1) != E:
        raise ValueError(f"preds/edge_index mismatch: {E} scores vs {edge_index.size(1)} edges")
    slice_dict = _attr(batch, "_slice_dict")
    slice_dict = slice_dict if isinstance(slice_dict, dict) else {}
    edge_ptr = _attr(batch, "edge_ptr")
    if edge_ptr is None:
        edge_ptr = slice_dict.get("edge_index")
    if edge_ptr is None:
        ids = indexes if indexes is not None else _attr(batch, "edge_batch")
        if ids is None:
            raise ValueError("query_ids required for reachability metrics when edge ptr is unavailable.")
        ids = i64(ids)
        if ids.numel() != E:
            raise ValueError(f"query_ids/scores mismatch: {tuple(ids.shape)} vs {tuple(scores.shape)}")
        edge_ptr = ops.ids_to_ptr(ids, B)  # no device-to-host read (torch.bincount reads the maximum back)
    edge_ptr = i64(edge_ptr)
    if edge_ptr.numel() != B + 1:
        raise ValueError(f"edge ptr length mismatch: {edge_ptr.numel()} vs expected {B + 1}")

    def ptr_of(name):
        p = _attr(batch, name + "_ptr")
        if p is None:
            p = slice_dict.get(name)
        return p

    q_idx, a_idx = _attr(batch, "q_local_indices"), _attr(batch, "a_local_indices")
    q_ptr, a_ptr = ptr_of("q_local_indices"), ptr_of("a_local_indices")
    if q_idx is None or a_idx is None or q_ptr is None or a_ptr is None:
        raise ValueError("Batch missing q_local_indices/a_local_indices required for reachability metrics.")
    q_idx, a_idx, q_ptr, a_ptr = i64(q_idx), i64(a_idx), i64(q_ptr), i64(a_ptr)
    if q_ptr.numel() != B + 1:
        raise ValueError(f"q_local_indices_ptr length mismatch: {q_ptr.numel()} vs expected {B + 1}")
    if a_ptr.numel() != B + 1:
        raise ValueError(f"a_local_indices_ptr length mismatch: {a_ptr.numel()} vs expected {B + 1}")
    gids, ans, ans_ptr = _attr(batch, "node_global_ids"), _attr(batch, "answer_entity_ids"), ptr_of("answer_entity_ids")
    have_answers = gids is not None and ans is not None and ans_ptr is not None
    if have_answers:
        gids, ans, ans_ptr = i64(gids), i64(ans), i64(ans_ptr)
    tgt = None
    if target is not None:
        tgt = target.detach().reshape(-1).to(device=dev)
        tgt = (tgt > 0.5) if tgt.dtype != torch.bool else tgt
        tgt = tgt.to(torch.uint8).contiguous()
        if tgt.numel() != E:
            raise ValueError(f"preds/target/indexes mismatch: {tuple(scores.shape)} vs {tuple(tgt.shape)}")

    nk, k_max = len(ks), ks[-1]
    N = _attr(batch, "num_nodes")
    N = int(N) if N is not None else (int(node_ptr[-1].item()) if node_ptr.numel() else 0)
    f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
    u8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)  # noqa: E731
    res = dict(edge_recall=f32(B, nk), recall_valid=u8(B), reach=u8(B, nk), reach_valid=u8(B), answer_hit=u8(B, nk),
               answer_recall=f32(B, nk), answer_valid=u8(B), score_margin=f32(B), margin_valid=u8(B))
    if want_topk:
        res.update(topk_index=torch.empty((B, k_max), dtype=torch.int32, device=dev),
                   topk_score=torch.empty((B, k_max), dtype=torch.float32, device=dev),
                   topk_count=torch.empty((B,), dtype=torch.int32, device=dev))
    uf_ws = ops._workspace(dev, "metrics_uf", 8 * max(N, 1))
    karr = (ctypes.c_int32 * nk)(*ks)
    p = ops._ptr
    lib = _lib.load()
    _lib.check(lib.evi_retriever_metrics(
        p(scores), p(tgt), p(edge_index), E, p(edge_ptr), p(node_ptr), B, p(q_idx), p(q_ptr), p(a_idx), p(a_ptr),
        p(gids) if have_answers else None, p(ans) if have_answers else None, p(ans_ptr) if have_answers else None,
        karr, nk, p(res["edge_recall"]), p(res["recall_valid"]), p(res["reach"]), p(res["reach_valid"]),
        p(res["answer_hit"]), p(res["answer_recall"]), p(res["answer_valid"]), p(res["score_margin"]),
        p(res["margin_valid"]), p(res.get("topk_index")), p(res.get("topk_score")), p(res.get("topk_count")),
        uf_ws.data_ptr(), ops._stream(dev)))
    # answer_valid == 2 marks a graph with more than 2048 answer entities; metrics raise when their states are read
    too_many = (res["answer_valid"] == 2).any() if have_answers else None
    return RankedBatch(ks, edge_ptr=edge_ptr, have_answers=have_answers, too_many_answers=too_many, **res)


class _SumMetric:
    """Minimal torchmetrics-style base: named f64 sum states, reset(), sync().

    `update` never reads the device: every batch's contributions are added to one f64 device vector
    (`_accumulate`), and the host-side `_states` are brought up to date only when somebody looks
    (`compute`, `sync`, `_flush`) — the reference's `.item()` per metric per batch stalls the stream."""

    def __init__(self, **_: Any) -> None:
        self._states: Dict[str, float] = {}
        self._order: Dict[str, int] = {}
        self._dev_acc: Optional[torch.Tensor] = None
        self._idx_cache: Dict[Any, torch.Tensor] = {}
        self._overflow: Optional[torch.Tensor] = None
        self._shared: Optional[RankedBatch] = None  # set by RetrieverMetricCollection for the current batch

    def _add_state(self, name: str) -> None:
        self._order[name] = len(self._states)
        self._states[name] = 0.0

    def _accumulate(self, names: Sequence[str], values: torch.Tensor) -> None:
        """states[names[i]] += values[i], on the device (values: 1-D, any float dtype)."""
        dev = values.device
        if self._dev_acc is None or self._dev_acc.device != dev:
            self._flush()
            self._dev_acc = torch.zeros(len(self._states), dtype=torch.float64, device=dev)
        key = (tuple(names), dev)
        idx = self._idx_cache.get(key)
        if idx is None:
            idx = self._idx_cache[key] = torch.tensor([self._order[n] for n in names], dtype=torch.long, device=dev)
        self._dev_acc.index_add_(0, idx, values.to(torch.float64).view(-1))

    def _note_overflow(self, flag: Optional[torch.Tensor]) -> None:
        if flag is not None:
            self._overflow = flag if self._overflow is None else (self._overflow | flag)

    def _flush(self) -> None:
        if self._overflow is not None:
            bad = bool(self._overflow.item())
            self._overflow = None
            if bad:
                raise NotImplementedError("a graph has more than 2048 answer entities")
        if self._dev_acc is not None:
            vals = self._dev_acc.tolist()
            self._dev_acc.zero_()
            for n, i in self._order.items():
                self._states[n] += vals[i]

    def reset(self) -> None:
        self._dev_acc = None
        self._overflow = None
        for k in self._states:
            self._states[k] = 0.0

    def sync(self, group=None) -> None:
        """All-reduce (SUM) the states across ranks — the reference's dist_reduce_fx="sum"."""
        import torch.distributed as dist

        self._flush()
        if not (dist.is_available() and dist.is_initialized()):
            return
        names = sorted(self._states)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor([self._states[n] for n in names], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        for n, v in zip(names, t.tolist()):
            self._states[n] = v

    def _ranked(self, preds, target, batch, k_values, num_graphs, indexes) -> RankedBatch:
        sh = self._shared
        if sh is not None and all(k in sh.k_values for k in k_values):
            return sh
        return rank_batch(preds, target, batch, k_values, num_graphs=num_graphs, indexes=indexes)

    @staticmethod
    def _col(rb: RankedBatch, k: int) -> int:
        return rb.k_values.index(k)

    def _cols(self, rb: RankedBatch, dev: torch.device) -> torch.Tensor:
        """Device index of this metric's k columns inside the shared ranking (cached: no per-batch H2D)."""
        key = ("cols", tuple(rb.k_values), dev)
        idx = self._idx_cache.get(key)
        if idx is None:
            idx = self._idx_cache[key] = torch.tensor([self._col(rb, k) for k in self.k_values], dtype=torch.long, device=dev)
        return idx


class EdgeRecallAtK(_SumMetric):
    """reference: EdgeRecallAtK, src/metrics/retriever_metrics.py:83-166."""

    def __init__(self, k_values: Optional[Sequence[int]] = None, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.k_values = normalize_k_values(k_values)
        for k in self.k_values:
            self._add_state(f"recall_sum_at_{k}")
        self._add_state("graph_count")

    def update(self, preds, target, indexes, batch, num_graphs: Optional[int] = None, **_: Any) -> None:
        if not self.k_values or preds.numel() == 0:
            return
        rb = self._ranked(preds, target, batch, self.k_values, num_graphs, indexes)
        valid = rb.recall_valid.bool()
        sums = (rb.edge_recall.double() * valid.unsqueeze(1)).sum(0)
        cols = self._cols(rb, valid.device)
        self._accumulate([f"recall_sum_at_{k}" for k in self.k_values] + ["graph_count"],
                         torch.cat([sums[cols], valid.sum().double().view(1)]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        denom = max(self._states["graph_count"], 1.0)
        return {f"edge/recall@{k}": torch.tensor(self._states[f"recall_sum_at_{k}"] / denom, dtype=torch.float32)
                for k in self.k_values}


class AnswerReachability(_SumMetric):
    """reference: AnswerReachability, src/metrics/reachability.py:9-381."""

    def __init__(self, k_values: Optional[Sequence[int]] = None, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.k_values = normalize_k_values(k_values)
        for k in self.k_values:
            self._add_state(f"hits_at_{k}")
        self._add_state("total")

    def update(self, preds, batch, query_ids=None, num_graphs: Optional[int] = None, target=None, **_: Any) -> None:
        if not self.k_values or preds.numel() == 0:
            return
        rb = self._ranked(preds, target, batch, self.k_values, num_graphs, query_ids)
        valid = rb.reach_valid.bool()
        hits = (rb.reach.double() * valid.unsqueeze(1)).sum(0)
        cols = self._cols(rb, valid.device)
        self._accumulate([f"hits_at_{k}" for k in self.k_values] + ["total"], torch.cat([hits[cols], valid.sum().double().view(1)]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        denom = max(self._states["total"], 1.0)
        return {f"answer/reachability@{k}": torch.tensor(self._states[f"hits_at_{k}"] / denom, dtype=torch.float32)
                for k in self.k_values}


class AnswerHitAtK(_SumMetric):
    """Hits@k / answer-recall@k over entity ids, averaged over graphs that have answers.
    reference: compute_answer_hit / compute_answer_recall, src/utils/metrics.py:167-238;
    _oracle_metrics_for_sample, src/models/reasoner_module.py:17-68, averaged at :190-214."""

    def __init__(self, k_values: Optional[Sequence[int]] = None, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.k_values = normalize_k_values(k_values)
        for k in self.k_values:
            self._add_state(f"hit_sum_at_{k}")
            self._add_state(f"recall_sum_at_{k}")
        self._add_state("sample_count")

    def update(self, preds, batch, target=None, indexes=None, num_graphs: Optional[int] = None, **_: Any) -> None:
        if not self.k_values or preds.numel() == 0:
            return
        rb = self._ranked(preds, target, batch, self.k_values, num_graphs, indexes)
        if not rb.have_answers:
            raise ValueError("Batch missing node_global_ids/answer_entity_ids required for answer hit metrics.")
        self._note_overflow(rb.too_many_answers)
        valid = rb.answer_valid == 1
        hit = (rb.answer_hit.double() * valid.unsqueeze(1)).sum(0)
        rec = (rb.answer_recall.double() * valid.unsqueeze(1)).sum(0)
        cols = self._cols(rb, valid.device)
        self._accumulate([f"hit_sum_at_{k}" for k in self.k_values] + [f"recall_sum_at_{k}" for k in self.k_values] + ["sample_count"],
                         torch.cat([hit[cols], rec[cols], valid.sum().double().view(1)]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        n = self._states["sample_count"]
        out = {}
        for k in self.k_values:
            out[f"answer_hit@{k}"] = torch.tensor(self._states[f"hit_sum_at_{k}"] / n if n else 0.0, dtype=torch.float32)
            out[f"answer_recall@{k}"] = torch.tensor(self._states[f"recall_sum_at_{k}"] / n if n else 0.0,
                                                     dtype=torch.float32)
        return out


class ScoreMargin(_SumMetric):
    """reference: ScoreMargin, src/metrics/retriever_metrics.py:330-397."""

    def __init__(self, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self._add_state("margin_sum")
        self._add_state("graph_count")

    def update(self, preds, target, indexes, batch, num_graphs: Optional[int] = None, **_: Any) -> None:
        if preds.numel() == 0:
            return
        rb = self._ranked(preds, target, batch, [1], num_graphs, indexes)
        valid = rb.margin_valid.bool()
        self._accumulate(["graph_count", "margin_sum"],
                         torch.stack([valid.sum().double(), (rb.score_margin.double() * valid).sum()]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        denom = max(self._states["graph_count"], 1.0)
        return {"edge/score_margin": torch.tensor(self._states["margin_sum"] / denom, dtype=torch.float32)}


def _bridge_sublists(preds, target, batch, num_graphs, indexes):
    """Scores / labels / edge_ptr restricted to bridge edges (touching neither a seed nor an answer),
    per graph, in the original order.  reference: _compute_bridge_mask,
    src/metrics/retriever_metrics.py:66-80."""
    scores = preds.detach().reshape(-1)
    dev = ops._require_gpu(scores)
    edge_index = torch.as_tensor(_attr(batch, "edge_index")).to(device=dev, dtype=torch.long).contiguous()
    num_nodes = _attr(batch, "num_nodes")
    if num_nodes is None:
        raise ValueError("Batch missing num_nodes required for bridge metrics.")
    q, a = _attr(batch, "q_local_indices"), _attr(batch, "a_local_indices")
    if q is None or a is None:
        raise ValueError("Batch missing q_local_indices/a_local_indices required for bridge metrics.")
    near = ops.qa_edge_mask(edge_index, int(num_nodes), torch.as_tensor(q), torch.as_tensor(a))
    bridge = ~near
    if bridge.numel() != scores.numel():
        raise ValueError(f"bridge_mask length mismatch: {bridge.numel()} vs scores {scores.numel()}")
    node_ptr = torch.as_tensor(_attr(batch, "ptr")).to(device=dev, dtype=torch.long)
    B = int(num_graphs) if num_graphs is not None else int(node_ptr.numel() - 1)
    ids = indexes if indexes is not None else _attr(batch, "edge_batch")
    if ids is None:
        ids, _, _ = ops.edge_batch(edge_index, node_ptr)
    ids = torch.as_tensor(ids).to(device=dev, dtype=torch.long).view(-1)
    tgt = target.detach().reshape(-1).to(dev)
    tgt = (tgt > 0.5) if tgt.dtype != torch.bool else tgt
    edge_ptr = ops.ids_to_ptr(ids[bridge], B)
    full_ptr = ops.ids_to_ptr(ids, B)
    return scores[bridge].contiguous(), tgt[bridge].contiguous(), edge_ptr, bridge, tgt, full_ptr, B


def _class_stats(scores, target_u8, edge_ptr, B):
    dev = scores.device
    out = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    if B > 0:
        lib = _lib.load()
        _lib.check(lib.evi_graph_class_stats(ops._ptr(scores), ops._ptr(target_u8), ops._ptr(edge_ptr.contiguous()), B,
                                             ops._ptr(out), ops._stream(dev)))
    return out


class BridgeEdgeRecallAtK(_SumMetric):
    """Edge recall@k over bridge edges only, averaged over graphs with a bridge positive.
    reference: BridgeEdgeRecallAtK, src/metrics/retriever_metrics.py:169-267."""

    def __init__(self, k_values: Optional[Sequence[int]] = None, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.k_values = normalize_k_values(k_values)
        for k in self.k_values:
            self._add_state(f"recall_sum_at_{k}")
        self._add_state("graph_count")

    def update(self, preds, target, indexes, batch, num_graphs: Optional[int] = None, **_: Any) -> None:
        if not self.k_values or preds.numel() == 0:
            return
        s, t, eptr, _, _, _, B = _bridge_sublists(preds, target, batch, num_graphs, indexes)
        if s.numel() == 0:
            return
        # rank the bridge sub-lists; reachability / answer inputs are irrelevant here
        dev = s.device
        zero_ptr = torch.zeros(B + 1, dtype=torch.long, device=dev)
        fake = dict(edge_index=torch.zeros((2, s.numel()), dtype=torch.long, device=dev), ptr=_attr(batch, "ptr"), edge_ptr=eptr,
                    q_local_indices=torch.zeros(0, dtype=torch.long, device=dev), q_local_indices_ptr=zero_ptr,
                    a_local_indices=torch.zeros(0, dtype=torch.long, device=dev), a_local_indices_ptr=zero_ptr)
        rb = rank_batch(s, t, fake, self.k_values, num_graphs=B)
        stats = _class_stats(s, t.to(torch.uint8), eptr, B)
        valid = (stats[:, 0] > 0) & rb.recall_valid.bool()  # graphs without a bridge positive are skipped (:239-241)
        sums = (rb.edge_recall.double() * valid.unsqueeze(1)).sum(0)
        cols = self._cols(rb, valid.device)
        self._accumulate([f"recall_sum_at_{k}" for k in self.k_values] + ["graph_count"],
                         torch.cat([sums[cols], valid.sum().double().view(1)]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        denom = max(self._states["graph_count"], 1.0)
        return {f"bridge/recall@{k}": torch.tensor(self._states[f"recall_sum_at_{k}"] / denom, dtype=torch.float32)
                for k in self.k_values}


class BridgePositiveCoverage(_SumMetric):
    """reference: BridgePositiveCoverage, src/metrics/retriever_metrics.py:270-327."""

    def __init__(self, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        for n in ("bridge_pos_edges", "total_pos_edges", "graphs_with_pos", "graphs_with_bridge_pos"):
            self._add_state(n)

    def update(self, preds, target, indexes, batch, num_graphs: Optional[int] = None, **_: Any) -> None:
        if preds.numel() == 0:
            return
        s, t, eptr, bridge, tgt_all, full_ptr, B = _bridge_sublists(preds, target, batch, num_graphs, indexes)
        all_stats = _class_stats(preds.detach().reshape(-1).float().contiguous(), tgt_all.to(torch.uint8).contiguous(), full_ptr, B)
        br_stats = _class_stats(s, t.to(torch.uint8), eptr, B) if s.numel() else torch.zeros((B, 4), dtype=torch.float64, device=s.device)
        has_pos = all_stats[:, 0] > 0
        self._accumulate(["total_pos_edges", "bridge_pos_edges", "graphs_with_pos", "graphs_with_bridge_pos"],
                         torch.stack([all_stats[:, 0].sum(), br_stats[:, 0].sum(), has_pos.sum().double(),
                                      (has_pos & (br_stats[:, 0] > 0)).sum().double()]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        e = max(self._states["total_pos_edges"], 1.0)
        g = max(self._states["graphs_with_pos"], 1.0)
        return {"bridge/pos_edge_frac": torch.tensor(self._states["bridge_pos_edges"] / e, dtype=torch.float32),
                "bridge/pos_graph_frac": torch.tensor(self._states["graphs_with_bridge_pos"] / g, dtype=torch.float32)}


class BridgeProbQuality(_SumMetric):
    """Mean sigmoid(score) of positive / negative bridge edges per graph having both.
    reference: BridgeProbQuality, src/metrics/retriever_metrics.py:400-476."""

    def __init__(self, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        for n in ("pos_prob_sum", "neg_prob_sum", "sep_sum", "graph_count"):
            self._add_state(n)

    def update(self, preds, target, indexes, batch, num_graphs: Optional[int] = None, **_: Any) -> None:
        if preds.numel() == 0:
            return
        s, t, eptr, _, _, _, B = _bridge_sublists(preds, target, batch, num_graphs, indexes)
        if s.numel() == 0:
            return
        st = _class_stats(s, t.to(torch.uint8), eptr, B)
        valid = (st[:, 0] > 0) & (st[:, 1] > 0)
        pos_mean = (st[:, 2] / st[:, 0].clamp(min=1.0)).float().double()  # per-graph means are f32 in the reference
        neg_mean = (st[:, 3] / st[:, 1].clamp(min=1.0)).float().double()
        self._accumulate(["pos_prob_sum", "neg_prob_sum", "sep_sum", "graph_count"],
                         torch.stack([(pos_mean * valid).sum(), (neg_mean * valid).sum(), ((pos_mean - neg_mean) * valid).sum(),
                                      valid.sum().double()]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        d = max(self._states["graph_count"], 1.0)
        return {"bridge/pos_prob": torch.tensor(self._states["pos_prob_sum"] / d, dtype=torch.float32),
                "bridge/neg_prob": torch.tensor(self._states["neg_prob_sum"] / d, dtype=torch.float32),
                "bridge/separation": torch.tensor(self._states["sep_sum"] / d, dtype=torch.float32)}


class FeatureMonitor(_SumMetric):
    """Mean sigmoid(score) of positive / negative edges over the whole epoch and the mean L2 norm of the
    edge features.  reference: FeatureMonitor, src/metrics/feature_monitor.py:9-58 (enabled by
    evaluation_cfg.feature_metrics, src/models/retriever_module.py:112-113)."""

    def __init__(self, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        for n in ("pos_score_sum", "pos_count", "neg_score_sum", "neg_count", "feat_norm_sum", "feat_count"):
            self._add_state(n)

    def update(self, preds, target, features=None, **_: Any) -> None:
        scores = preds.detach().reshape(-1)
        if scores.numel() == 0:
            return
        dev = ops._require_gpu(scores)
        tgt = target.detach().reshape(-1).to(dev)
        tgt = (tgt > 0.5) if tgt.dtype != torch.bool else tgt
        ptr = torch.tensor([0, scores.numel()], dtype=torch.long, device=dev)
        st = _class_stats(scores.float().contiguous(), tgt.to(torch.uint8).contiguous(), ptr, 1)[0]  # pos, neg, sum p|pos, sum p|neg
        self._accumulate(["pos_count", "neg_count", "pos_score_sum", "neg_score_sum"], st)
        if features is not None and features.numel() > 0:
            norms = ops.row_norms(features.detach().reshape(-1, features.shape[-1]))
            self._accumulate(["feat_norm_sum", "feat_count"],
                             torch.stack([norms.sum().double(), torch.tensor(float(norms.numel()), dtype=torch.float64, device=dev)]))

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        s = self._states
        pos = s["pos_score_sum"] / max(s["pos_count"], 1.0)
        neg = s["neg_score_sum"] / max(s["neg_count"], 1.0)
        return {"features/pos_prob_avg": torch.tensor(pos, dtype=torch.float32),
                "features/neg_prob_avg": torch.tensor(neg, dtype=torch.float32),
                "features/separation_gap": torch.tensor(pos - neg, dtype=torch.float32),
                "features/norm_avg": torch.tensor(s["feat_norm_sum"] / max(s["feat_count"], 1.0), dtype=torch.float32)}


class RetrieverMetricCollection:
    """The metric set RetrieverModule builds (src/models/retriever_module.py:100-131), sharing one
    ranking pass per batch.  `update` takes the module's keyword set; `compute` merges the dicts."""

    def __init__(self, k_values: Sequence[int], *, answer_hit: bool = True, bridge_metrics: bool = False,
                 feature_metrics: bool = False, prefix: str = "") -> None:
        self.k_values = normalize_k_values(k_values)
        self.prefix = prefix
        self.metrics: Dict[str, _SumMetric] = {
            "edge_recall": EdgeRecallAtK(self.k_values),
            "reachability": AnswerReachability(self.k_values),
            "score_margin": ScoreMargin(),
        }
        if answer_hit:
            self.metrics["answer_hit"] = AnswerHitAtK(self.k_values)
        if bridge_metrics:  # evaluation_cfg.bridge_metrics (configs/model/retriever_module.yaml:52-53)
            self.metrics["bridge_recall"] = BridgeEdgeRecallAtK(self.k_values)
            self.metrics["bridge_coverage"] = BridgePositiveCoverage()
            self.metrics["bridge_quality"] = BridgeProbQuality()
        if feature_metrics:  # evaluation_cfg.feature_metrics (src/models/retriever_module.py:112-113)
            self.metrics["features"] = FeatureMonitor()

    # (metric, state name) of every slot of evi_metric_accumulate's vector, in its order
    def _slots(self):
        ks = self.k_values
        er, re, ah, sm = (self.metrics.get(n) for n in ("edge_recall", "reachability", "answer_hit", "score_margin"))
        return ([(er, f"recall_sum_at_{k}") for k in ks] + [(er, "graph_count")] + [(re, f"hits_at_{k}") for k in ks] + [(re, "total")]
                + [(ah, f"hit_sum_at_{k}") for k in ks] + [(ah, f"recall_sum_at_{k}") for k in ks] + [(ah, "sample_count")]
                + [(sm, "margin_sum"), (sm, "graph_count")])

    def _flush(self) -> None:
        """Moves the fused device accumulator into the metrics' host states (one read)."""
        acc = getattr(self, "_acc", None)
        if acc is None:
            return
        vals = acc.tolist()
        acc.zero_()
        if vals[-1] > 0:
            raise NotImplementedError("a graph has more than 2048 answer entities")
        for (m, name), v in zip(self._slots(), vals):
            if m is not None:
                m._states[name] += v

    def update(self, *, preds, target, indexes, batch, query_ids=None, num_graphs=None, features=None, **_: Any) -> None:
        if preds.numel() == 0:
            return
        shared = rank_batch(preds, target, batch, self.k_values, num_graphs=num_graphs, indexes=indexes)
        # the four ranking metrics share the ranking AND one fused accumulation launch
        dev = shared.edge_recall.device
        nk = len(self.k_values)
        acc = getattr(self, "_acc", None)
        if acc is None or acc.device != dev:
            self._flush()
            acc = self._acc = torch.zeros(4 * nk + 6, dtype=torch.float64, device=dev)
        p = ops._ptr
        ans = shared.have_answers and "answer_hit" in self.metrics
        _lib.check(_lib.load().evi_metric_accumulate(
            p(shared.edge_recall), p(shared.recall_valid), p(shared.reach), p(shared.reach_valid),
            p(shared.answer_hit) if ans else None, p(shared.answer_recall) if ans else None, p(shared.answer_valid) if ans else None,
            p(shared.score_margin), p(shared.margin_valid), int(shared.recall_valid.numel()), nk, acc.data_ptr(), ops._stream(dev)))
        if "features" in self.metrics:
            self.metrics["features"].update(preds=preds, target=target, features=features)
        for name, m in self.metrics.items():
            if not name.startswith("bridge"):
                continue  # ranking metrics were accumulated above
            m._shared = None if name.startswith("bridge") else shared
            try:
                m.update(preds=preds, target=target, indexes=indexes, batch=batch, query_ids=query_ids,
                         num_graphs=num_graphs)
            finally:
                m._shared = None

    def compute(self) -> Dict[str, torch.Tensor]:
        self._flush()
        out: Dict[str, torch.Tensor] = {}
        for m in self.metrics.values():
            out.update({self.prefix + k: v for k, v in m.compute().items()})
        return out

    def reset(self) -> None:
        self._acc = None
        for m in self.metrics.values():
            m.reset()

    def sync(self, group=None) -> None:
        self._flush()
        for m in self.metrics.values():
            m.sync(group)


# ---- list-of-samples ranking statistics (src/utils/metrics.py:104-170) --------------------------------------------------------
@dataclass
class RankingStats:
    """reference: RankingStats, src/utils/metrics.py:104-110."""
    precision_at_k: Dict[int, float]
    recall_at_k: Dict[int, float]
    f1_at_k: Dict[int, float]
    ndcg_at_k: Dict[int, float]
    mrr: float


def _descending_keys(values: torch.Tensor) -> torch.Tensor:
    """int64 keys whose ASCENDING order is the values' descending order (f32; -0.0 counted as +0.0)."""
    bits = (values + 0.0).contiguous().view(torch.int32).to(torch.int64)
    return -torch.where(bits < 0, bits ^ 0x7FFFFFFF, bits)


def compute_ranking_metrics(samples, k_values) -> RankingStats:
    """Precision / recall / F1 / nDCG @k and MRR over an iterable of {"scores", "labels"} samples, averaged over the samples
    that have positives.  reference: compute_ranking_metrics + _ndcg, src/utils/metrics.py:112-170 — same quirks: hits and
    DCG sum label VALUES, the recall denominator is the label sum truncated to an int, samples whose truncated label sum is
    <= 0 are skipped, an empty k list means [1].

    All samples are ranked in ONE pass on the device: the flat score list is sorted per sample by the segmented stable sort
    (`evi_segment_sort_rank`, order (score desc, position asc) — the reference's argsort leaves ties unspecified), every
    per-sample sum is a segment reduction, and one small tensor is read back.  Discounts are f32 like the reference's, the
    sums f64 (the reference sums in f32: agreement to ~1e-7)."""
    ks = normalize_k_values(k_values, default=[1])
    zero = {k: 0.0 for k in ks}
    rows = [(torch.as_tensor(s["scores"]).reshape(-1), torch.as_tensor(s["labels"]).reshape(-1)) for s in samples]
    for sc, lb in rows:
        if sc.numel() != lb.numel():
            raise ValueError(f"scores/labels length mismatch: {sc.numel()} vs {lb.numel()}")
    rows = [r for r in rows if r[0].numel() > 0]  # an empty sample has no positives: skipped by the reference too
    if not rows:
        return RankingStats(dict(zero), dict(zero), dict(zero), dict(zero), 0.0)
    dev = next((r[0].device for r in rows if r[0].is_cuda), None)
    if dev is None:
        if not torch.cuda.is_available():
            raise RuntimeError("compute_ranking_metrics ranks on the GPU (evi_segment_sort_rank): no GPU visible, and there is no CPU path")
        dev = torch.device("cuda", torch.cuda.current_device())
    scores = torch.cat([r[0].to(device=dev, dtype=torch.float32) for r in rows])
    labels = torch.cat([r[1].to(device=dev, dtype=torch.float32) for r in rows])
    lens = torch.tensor([r[0].numel() for r in rows], dtype=torch.long)
    ptr = torch.zeros(len(rows) + 1, dtype=torch.long)
    torch.cumsum(lens, 0, out=ptr[1:])
    ptr, lens = ptr.to(dev), lens.to(dev)
    S = len(rows)
    seg = torch.repeat_interleave(torch.arange(S, device=dev), lens, output_size=int(scores.numel()))
    kt = torch.tensor(ks, dtype=torch.int32, device=dev)

    def seg_sum(x):  # [T] or [T, K] f64 -> per-sample sums, in element order (deterministic)
        return torch.segment_reduce(x, "sum", lengths=lens, axis=0, unsafe=True)

    def discounted(rank):  # 1 / log2(position + 1), position = rank + 1, in f32 (:160-161)
        return (1.0 / torch.log2(rank.to(torch.float32) + 2.0)).to(torch.float64)

    rank, _ = ops.segment_sort_rank(_descending_keys(scores), ptr)
    ideal_rank, _ = ops.segment_sort_rank(_descending_keys(labels), ptr)  # torch.sort(ranked_labels, descending=True) (:163)
    lab64 = labels.to(torch.float64)
    positives = torch.trunc(seg_sum(lab64))                                   # int(labels.sum()) (:121)
    valid = positives > 0
    in_k = (rank.unsqueeze(1) < kt.unsqueeze(0)).to(torch.float64)            # [T, K]
    hits = seg_sum(lab64.unsqueeze(1) * in_k)                                 # ranked_labels[:k].sum() (:132-133)
    dcg = seg_sum((lab64 * discounted(rank)).unsqueeze(1) * in_k)
    ideal_in_k = (ideal_rank.unsqueeze(1) < kt.unsqueeze(0)).to(torch.float64)
    idcg = seg_sum((lab64 * discounted(ideal_rank)).unsqueeze(1) * ideal_in_k)
    pos_mask = labels > 0.5
    first = torch.full((S,), 1 << 30, dtype=torch.int32, device=dev)
    first.scatter_reduce_(0, seg[pos_mask], rank[pos_mask], "amin")           # first ranked label > 0.5 (:127-130)
    has_first = (first < (1 << 30)) & valid
    precision = hits / kt.to(torch.float64).unsqueeze(0)
    recall = hits / positives.clamp(min=1.0).unsqueeze(1)
    pr = precision + recall
    f1 = torch.where(pr == 0, torch.zeros_like(pr), 2.0 * precision * recall / pr.clamp(min=1e-300))
    ndcg = torch.where(idcg > 0, dcg / idcg.clamp(min=1e-300), torch.zeros_like(dcg))
    v = valid.to(torch.float64).unsqueeze(1)
    count = valid.sum().to(torch.float64).clamp(min=1.0)
    mrr_n = has_first.sum().to(torch.float64)
    mrr = (has_first.to(torch.float64) / (first.to(torch.float64) + 1.0)).sum() / mrr_n.clamp(min=1.0)
    packed = torch.cat([((precision * v).sum(0) / count), ((recall * v).sum(0) / count), ((f1 * v).sum(0) / count),
                        ((ndcg * v).sum(0) / count), mrr.view(1)]).cpu().tolist()  # the one read-back
    K = len(ks)
    take = lambda j: {k: float(packed[j * K + i]) for i, k in enumerate(ks)}  # noqa: E731
    return RankingStats(take(0), take(1), take(2), take(3), float(packed[4 * K]))


__all__ = ["normalize_k_values", "rank_batch", "RankedBatch", "EdgeRecallAtK", "AnswerReachability", "AnswerHitAtK",
           "ScoreMargin", "BridgeEdgeRecallAtK", "BridgePositiveCoverage", "BridgeProbQuality", "FeatureMonitor",
           "RetrieverMetricCollection", "RankingStats", "compute_ranking_metrics"]
