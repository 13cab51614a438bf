"""g_agent sample materialisation from retriever scores.

Mirror of `GAgentBuilder` (src/data/components/g_agent_builder.py:115-560) and its settings /
sample dataclasses (:30-90, src/data/g_agent_dataset.py:19-52).  The reference slices a batch into
graphs and, per graph, runs node-softmax, top-k, seed expansion, a Python dict de-duplication of
(head, relation, tail) with max aggregation, a sorted re-indexing of the surviving nodes and a second
node-softmax — all on the CPU, one `.item()` per edge.  Here every one of those steps runs once per
BATCH on the device (one workgroup per graph inside each kernel) and a single device→host copy
hands the finished arrays to the per-sample dataclasses.
"""
from __future__ import annotations

import logging
import statistics
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import _lib, labelling, ops

log = logging.getLogger(__name__)

SCORE_MODE_LOGITS = "logits"
SCORE_MODE_NODE_SOFTMAX = "node_softmax"
DEFAULT_START_KEEP_RATIO = 0.25
DEFAULT_START_MIN_EDGES = 1


@dataclass
class GAgentSettings:
    """reference: GAgentSettings, src/data/components/g_agent_builder.py:30-86 (same fields, same checks)."""

    enabled: bool = True
    allow_empty_answer: bool = False
    edge_top_k: int = 50
    max_hops: int = 2
    score_temperature: float = 1.0
    score_bias: float = 0.0
    start_keep_ratio: float = DEFAULT_START_KEEP_RATIO
    start_min_edges: int = DEFAULT_START_MIN_EDGES
    start_max_edges: Optional[int] = None
    score_mode: str = SCORE_MODE_NODE_SOFTMAX
    output_path: Path = Path("g_agent/g_agent_samples.pt")

    def __post_init__(self) -> None:
        self.enabled = bool(self.enabled)
        self.allow_empty_answer = bool(self.allow_empty_answer)
        self.edge_top_k = int(self.edge_top_k)
        self.max_hops = int(self.max_hops)
        self.score_temperature = float(self.score_temperature)
        self.score_bias = float(self.score_bias)
        if self.max_hops < 0:
            raise ValueError(f"max_hops must be >= 0, got {self.max_hops}")
        if self.edge_top_k <= 0:
            raise ValueError(f"edge_top_k must be > 0, got {self.edge_top_k}")
        if not (self.score_temperature > 0.0):
            raise ValueError(f"score_temperature must be positive, got {self.score_temperature}")
        self.start_keep_ratio = float(self.start_keep_ratio)
        if not (0.0 <= self.start_keep_ratio <= 1.0):
            raise ValueError(f"start_keep_ratio must be in [0, 1], got {self.start_keep_ratio}")
        self.start_min_edges = int(self.start_min_edges)
        if self.start_min_edges < 0:
            raise ValueError(f"start_min_edges must be >= 0, got {self.start_min_edges}")
        self.start_max_edges = int(self.edge_top_k) if self.start_max_edges is None else int(self.start_max_edges)
        if self.start_max_edges < 0:
            raise ValueError(f"start_max_edges must be >= 0, got {self.start_max_edges}")
        if self.start_max_edges != 0 and self.start_min_edges > self.start_max_edges:
            raise ValueError(
                f"start_min_edges must be <= start_max_edges, got {self.start_min_edges} > {self.start_max_edges}")
        if self.score_mode not in {SCORE_MODE_LOGITS, SCORE_MODE_NODE_SOFTMAX}:
            raise ValueError(
                f"score_mode must be one of {SCORE_MODE_LOGITS}/{SCORE_MODE_NODE_SOFTMAX}, got {self.score_mode}")
        self.output_path = Path(self.output_path).expanduser()

    def to_metadata(self) -> Dict[str, Any]:
        payload = asdict(self)
        payload["output_path"] = str(payload.get("output_path"))
        return payload


def _empty_long() -> torch.Tensor:
    return torch.empty(0, dtype=torch.long)


@dataclass
class GAgentSample:
    """reference: GAgentSample, src/data/g_agent_dataset.py:19-52."""

    sample_id: str
    question: str
    question_emb: torch.Tensor
    node_entity_ids: torch.Tensor
    node_embedding_ids: torch.Tensor
    edge_head_locals: torch.Tensor
    edge_tail_locals: torch.Tensor
    edge_relations: torch.Tensor
    edge_scores: torch.Tensor
    edge_labels: torch.Tensor
    start_entity_ids: torch.Tensor
    answer_entity_ids: torch.Tensor
    gt_path_edge_local_ids: torch.Tensor
    start_node_locals: torch.Tensor = field(default_factory=_empty_long)
    answer_node_locals: torch.Tensor = field(default_factory=_empty_long)
    pair_start_node_locals: torch.Tensor = field(default_factory=_empty_long)
    pair_answer_node_locals: torch.Tensor = field(default_factory=_empty_long)
    pair_edge_local_ids: torch.Tensor = field(default_factory=_empty_long)
    pair_edge_counts: torch.Tensor = field(default_factory=_empty_long)
    pair_shortest_lengths: torch.Tensor = field(default_factory=_empty_long)
    gt_path_exists: bool = False
    is_answer_reachable: bool = False
    is_dummy_agent: bool = False


class GAgentBuilder:
    """reference: GAgentBuilder, src/data/components/g_agent_builder.py:115-560."""

    def __init__(self, settings: GAgentSettings, embedding_store=None, aux_embedding_store=None) -> None:
        self.cfg = settings
        self.embedding_store = embedding_store
        self.aux_embedding_store = aux_embedding_store
        self.samples: List[GAgentSample] = []
        self.stats = {"num_samples": 0, "path_exists": 0, "retrieval_failed": 0, "edge_counts": [], "path_lengths": []}

    def reset(self):
        self.samples = []
        for k in ["num_samples", "path_exists", "retrieval_failed"]:
            self.stats[k] = 0
        self.stats["edge_counts"] = []
        self.stats["path_lengths"] = []

    @staticmethod
    def _select_aux_field(aux_data, core_data, name: str, default):
        if aux_data is not None and name in aux_data:
            return aux_data[name]
        return core_data.get(name, default)

    # ---- per-sample metadata (host; :250-287) ---------------------------------------------------------
    def _load_meta(self, sample_id: str) -> Optional[Dict[str, Any]]:
        if self.embedding_store is None:
            raise ValueError("EmbeddingStore must be provided; builder cannot proceed without per-sample metadata.")
        try:
            raw = self.embedding_store.load_sample(sample_id)
        except KeyError:
            log.warning(f"Sample {sample_id} not found in LMDB.")
            return None
        aux = self.aux_embedding_store.load_sample(sample_id) if self.aux_embedding_store is not None else None
        question_emb = torch.as_tensor(raw.get("question_emb", []), dtype=torch.float32).detach().clone()
        if question_emb.dim() == 1:
            question_emb = question_emb.unsqueeze(0)
        elif question_emb.dim() != 2:
            raise ValueError(f"question_emb must be 1D or 2D, got shape {tuple(question_emb.shape)} for {sample_id}")
        question = self._select_aux_field(aux, raw, "question", "")
        if not isinstance(question, str):
            raise TypeError(f"question must be string for {sample_id}, got {type(question).__name__}")
        starts = torch.as_tensor(self._select_aux_field(aux, raw, "seed_entity_ids", []), dtype=torch.long).detach().clone()
        if starts.numel() == 0:
            raise ValueError(
                f"Sample {sample_id} missing seed_entity_ids (start_entity_ids). "
                "If using split LMDBs, pass aux_lmdb_path to GAgentMaterializationCallback.")
        answer_raw = raw.get("answer_entity_ids")
        if answer_raw is None:
            answer_raw = self._select_aux_field(aux, raw, "answer_entity_ids", [])
        answers = torch.as_tensor(answer_raw, dtype=torch.long).detach().clone()
        if answers.numel() == 0:
            raise ValueError(f"Sample {sample_id} missing answer_entity_ids.")
        return {"question_emb": question_emb, "question": question, "starts": starts, "answers": answers}

    # ---- the batch (:158-236 + :238-512 for every graph at once) ----------------------------------------
    def process_batch(self, batch, model_output) -> None:
        if not hasattr(batch, "ptr"):
            raise ValueError("Batch must have 'ptr' for slicing.")
        logits = getattr(model_output, "logits", None)
        if logits is None:
            raise ValueError("Retriever output missing logits; g_agent requires logit scores.")
        scores = logits.detach().view(-1)
        if scores.numel() == 0:
            return
        dev = ops._require_gpu(scores)
        scores = scores.to(torch.float32)
        if self.cfg.score_temperature != 1.0 or self.cfg.score_bias != 0.0:
            scores = scores / float(self.cfg.score_temperature) + float(self.cfg.score_bias)
        ptr = batch.ptr.to(device=dev, dtype=torch.int64).view(-1)
        B = int(ptr.numel() - 1)
        edge_index = batch.edge_index.to(device=dev, dtype=torch.int64)
        relations = batch.edge_attr.to(device=dev, dtype=torch.int64).view(-1)
        labels = batch.labels.to(device=dev, dtype=torch.float32).view(-1)
        node_global_ids = batch.node_global_ids.to(device=dev, dtype=torch.int64).view(-1)
        if not hasattr(batch, "node_embedding_ids"):
            raise AttributeError("Batch missing node_embedding_ids; retriever dataset must provide embedding ids per node.")
        node_embedding_ids = batch.node_embedding_ids.to(device=dev, dtype=torch.int64).view(-1)
        query_ids = getattr(model_output, "query_ids", None)
        if query_ids is None:
            raise ValueError("Retriever output missing query_ids; g_agent requires per-edge graph mapping.")
        query_ids = torch.as_tensor(query_ids).to(device=dev, dtype=torch.long).view(-1)
        if query_ids.numel() != scores.numel():
            raise ValueError(f"query_ids/logits shape mismatch: {query_ids.shape} vs {scores.shape}")
        E = int(query_ids.numel())
        in_range = (query_ids >= 0) & (query_ids < B)
        counts = torch.bincount(query_ids[in_range], minlength=B)
        if int(counts.sum().item()) != E:
            raise ValueError("Invalid query_ids: edge assignments exceed batch_size.")
        if E > 1 and bool((query_ids[1:] < query_ids[:-1]).any().item()):
            order = torch.argsort(query_ids * E + torch.arange(E, device=dev))  # stable grouping by graph (:191-193)
            edge_index, relations, labels, scores = edge_index[:, order], relations[order], labels[order], scores[order]
            query_ids = query_ids[order]
        edge_index = edge_index.contiguous()
        edge_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        edge_ptr[1:] = torch.cumsum(counts, 0)
        counts_h = counts.cpu().numpy()
        sample_ids = getattr(batch, "sample_id", [])

        # host metadata first, in sample order, so that the reference's exceptions surface the same way
        metas: List[Optional[Dict[str, Any]]] = [None] * B
        sids: List[str] = [""] * B
        for g in range(B):
            if counts_h[g] == 0:
                continue
            try:
                sids[g] = str(sample_ids[g])
            except IndexError:
                sids[g] = f"unknown_{g}"
            metas[g] = self._load_meta(sids[g])
        active = [g for g in range(B) if metas[g] is not None]
        if not active:
            return
        N = int(ptr[-1].item())
        ptr_h = ptr.cpu().numpy()

        # ---- start nodes: isin(node_global_ids[graph], start ids) for every graph in one pass (:289-293)
        n_start = np.asarray([metas[g]["starts"].numel() if metas[g] is not None else 0 for g in range(B)], np.int64)
        seg3 = np.concatenate([[0], np.cumsum(n_start + np.diff(ptr_h))]).astype(np.int64)
        keys3 = torch.empty(int(seg3[-1]), dtype=torch.int64, device=dev)
        node_graph = torch.repeat_interleave(torch.arange(B, device=dev), ptr[1:] - ptr[:-1])
        seg3_t, n_start_t = torch.from_numpy(seg3).to(dev), torch.from_numpy(n_start).to(dev)
        node_pos3 = seg3_t[:-1][node_graph] + n_start_t[node_graph] + (torch.arange(N, device=dev) - ptr[:-1][node_graph])
        keys3[node_pos3] = node_global_ids
        if int(n_start.sum()):
            spos = np.concatenate([seg3[g] + np.arange(n_start[g]) for g in range(B)]).astype(np.int64)
            sval = torch.cat([metas[g]["starts"].view(-1) for g in active])
            keys3[torch.from_numpy(spos).to(dev)] = sval.to(dev)
        first3 = ops.first_occurrence(keys3, seg3_t)
        start_mask = first3[node_pos3].to(torch.int64) < n_start_t[node_graph]
        has_start = torch.zeros(B, dtype=torch.int64, device=dev).index_add_(0, node_graph, start_mask.to(torch.int64)).cpu().numpy()
        for g in active:
            if has_start[g] == 0:
                raise ValueError(f"Start entities missing from retrieval graph (sample_id={sids[g]}).")
        seeds = torch.nonzero(start_mask, as_tuple=False).view(-1)

        # ---- E_env = global top-k  U  seed-incident edges, per graph (:294-329)
        node_softmax = self.cfg.score_mode == SCORE_MODE_NODE_SOFTMAX
        sel = labelling.node_softmax_logit(edge_scores=scores, edge_head_locals=edge_index[0], edge_tail_locals=edge_index[1],
                                           num_nodes=N) if node_softmax else scores.contiguous()
        k_eff = int(min(self.cfg.edge_top_k, int(counts_h.max())))
        top_idx, _, _ = ops.segment_topk(sel, edge_ptr, k_eff, want_scores=False)
        env_mask = torch.zeros(E, dtype=torch.uint8, device=dev)
        flat = (edge_ptr[:-1].view(-1, 1) + top_idx.to(torch.int64))[top_idx >= 0]
        env_mask[flat] = 1
        csr = ops.graph_csr(edge_index, ptr, edge_ptr)
        start_edges = torch.empty(E, dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        lib = _lib.load()
        _lib.check(lib.evi_select_start_edges(
            ops._ptr(sel), E, ops._ptr(seeds), seeds.numel(), csr.in_ptr.data_ptr(), csr.in_eid.data_ptr(),
            csr.out_ptr.data_ptr(), csr.out_eid.data_ptr(), N, float(self.cfg.start_keep_ratio), int(self.cfg.start_min_edges),
            int(self.cfg.start_max_edges), ops._ptr(start_edges), status.data_ptr(), ops._stream(dev)))
        env_edges = torch.nonzero(env_mask | start_edges, as_tuple=False).view(-1)  # ascending = torch.unique(cat) (:326)
        env_graph = query_ids[env_edges]
        env_cnt = torch.bincount(env_graph, minlength=B)
        env_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        env_ptr[1:] = torch.cumsum(env_cnt, 0)

        # ---- (h, r, t) de-duplication over GLOBAL ids, first-seen order, max of score / label (:331-364)
        e_h, e_t = edge_index[0, env_edges], edge_index[1, env_edges]
        keys = torch.stack([node_global_ids[e_h], relations[env_edges], node_global_ids[e_t]], dim=1).contiguous()
        first = ops.first_occurrence(keys, env_ptr)
        rank, trip_cnt, uniq = ops.first_seen_rank(first, env_ptr)
        trip_cnt = trip_cnt.to(torch.int64)
        trip_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        trip_ptr[1:] = torch.cumsum(trip_cnt, 0)
        T = int(trip_ptr[-1].item())
        group = (trip_ptr[:-1][env_graph] + rank.to(torch.int64)).to(torch.int32)
        agg_score = ops.group_max(scores[env_edges], group, T)
        agg_label = ops.group_max(labels[env_edges], group, T)
        trip_graph = torch.repeat_interleave(torch.arange(B, device=dev), trip_cnt)
        within = torch.arange(T, device=dev) - trip_ptr[:-1][trip_graph]
        rep = env_ptr[:-1][trip_graph] + uniq[env_ptr[:-1][trip_graph] + within].to(torch.int64)  # env position of each triple
        t_hg, t_rel, t_tg = keys[rep, 0], keys[rep, 1], keys[rep, 2]
        t_hnode, t_tnode = e_h[rep], e_t[rep]

        # ---- sorted unique node ids of each environment graph and the local ids of the endpoints (:366-383)
        seg2 = 2 * trip_ptr
        occ_pos_h = seg2[:-1][trip_graph] + within
        occ_pos_t = occ_pos_h + trip_cnt[trip_graph]
        keys2 = torch.empty(2 * T, dtype=torch.int64, device=dev)
        occ_node = torch.empty(2 * T, dtype=torch.int64, device=dev)
        keys2[occ_pos_h], keys2[occ_pos_t] = t_hg, t_tg
        occ_node[occ_pos_h], occ_node[occ_pos_t] = t_hnode, t_tnode
        first2 = ops.first_occurrence(keys2, seg2)
        rank2, node_cnt, uniq2 = ops.first_seen_rank(first2, seg2)
        node_cnt64 = node_cnt.to(torch.int64)
        # distinct ids in first-seen order, packed at the head of each segment, then their ascending rank
        M = int(node_cnt64.sum().item())
        ugraph = torch.repeat_interleave(torch.arange(B, device=dev), node_cnt64)
        node_ptr2 = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        node_ptr2[1:] = torch.cumsum(node_cnt64, 0)
        uwithin = torch.arange(M, device=dev) - node_ptr2[:-1][ugraph]
        upos = seg2[:-1][ugraph] + uniq2[seg2[:-1][ugraph] + uwithin].to(torch.int64)
        packed = torch.zeros(2 * T, dtype=torch.int64, device=dev)
        packed[seg2[:-1][ugraph] + uwithin] = keys2[upos]
        srank, _ = ops.segment_sort_rank(packed, seg2, node_cnt)
        sorted_slot = node_ptr2[:-1][ugraph] + srank[seg2[:-1][ugraph] + uwithin].to(torch.int64)
        node_entity = torch.empty(M, dtype=torch.int64, device=dev)
        node_emb = torch.empty(M, dtype=torch.int64, device=dev)
        node_entity[sorted_slot] = keys2[upos]
        node_emb[sorted_slot] = node_embedding_ids[occ_node[upos]]
        local_of_occ = srank[(seg2[:-1][trip_graph].repeat(2) + rank2[torch.cat([occ_pos_h, occ_pos_t])].to(torch.int64))].to(torch.int64)
        h_loc, t_loc = local_of_occ[:T], local_of_occ[T:]
        if node_softmax:  # second normalisation, on the environment graph (:384-389)
            out_score = labelling.node_softmax_logit(edge_scores=agg_score, edge_head_locals=node_ptr2[:-1][trip_graph] + h_loc,
                                                     edge_tail_locals=node_ptr2[:-1][trip_graph] + t_loc, num_nodes=max(M, 1))
        else:
            out_score = agg_score

        # ---- one copy to the host, then the per-sample dataclasses (:391-512)
        trip_ptr_h, node_ptr2_h = trip_ptr.cpu().numpy(), node_ptr2.cpu().numpy()
        cols = {name: t.cpu() for name, t in (("rel", t_rel), ("score", out_score), ("label", agg_label), ("h", h_loc),
                                                ("t", t_loc), ("ent", node_entity), ("emb", node_emb))}
        for g in active:
            a, b = int(trip_ptr_h[g]), int(trip_ptr_h[g + 1])
            if b == a:
                self.stats["retrieval_failed"] += 1
                continue
            n0, n1 = int(node_ptr2_h[g]), int(node_ptr2_h[g + 1])
            self._add_sample(sids[g], metas[g], edge_relations=cols["rel"][a:b].clone(), edge_scores=cols["score"][a:b].clone(),
                             edge_labels=cols["label"][a:b].clone(), edge_head_locals=cols["h"][a:b].clone(),
                             edge_tail_locals=cols["t"][a:b].clone(), node_entity_ids=cols["ent"][n0:n1].clone(),
                             node_embedding_ids=cols["emb"][n0:n1].clone())

    def _add_sample(self, sample_id: str, meta: Dict[str, Any], *, edge_relations, edge_scores, edge_labels, edge_head_locals,
                    edge_tail_locals, node_entity_ids, node_embedding_ids) -> None:
        ent = node_entity_ids.numpy()

        def locals_of(ids: List[int]) -> List[int]:
            if not ids:
                return []
            q = np.asarray(ids, np.int64)
            pos = np.searchsorted(ent, q)
            ok = (pos < ent.shape[0]) & (ent[np.minimum(pos, ent.shape[0] - 1)] == q)
            return pos[ok].tolist()

        start_entity_ids = meta["starts"]
        start_list = locals_of(start_entity_ids.tolist())
        if not start_list:
            self.stats["retrieval_failed"] += 1
            return
        start_node_locals = torch.tensor(list(dict.fromkeys(start_list)), dtype=torch.long)
        ordered_answers = list(dict.fromkeys(int(a) for a in meta["answers"].tolist()))
        answer_entity_ids = torch.tensor(ordered_answers, dtype=torch.long)
        ans_list = locals_of(ordered_answers)
        answer_node_locals = torch.tensor(ans_list, dtype=torch.long) if ans_list else _empty_long()
        dummy = False
        if answer_node_locals.numel() == 0:
            if not self.cfg.allow_empty_answer:
                self.stats["retrieval_failed"] += 1
                return
            edge_labels = torch.zeros(int(edge_relations.numel()), dtype=torch.float32)
            dummy = True
        self.samples.append(GAgentSample(
            sample_id=sample_id, question=meta["question"], question_emb=meta["question_emb"], edge_relations=edge_relations,
            edge_scores=edge_scores, edge_labels=edge_labels, edge_head_locals=edge_head_locals, edge_tail_locals=edge_tail_locals,
            node_entity_ids=node_entity_ids, node_embedding_ids=node_embedding_ids, start_entity_ids=start_entity_ids,
            answer_entity_ids=answer_entity_ids, start_node_locals=start_node_locals, answer_node_locals=answer_node_locals,
            gt_path_edge_local_ids=_empty_long(), gt_path_exists=False, is_answer_reachable=not dummy, is_dummy_agent=dummy))
        self.stats["num_samples"] += 1
        self.stats["edge_counts"].append(int(edge_relations.numel()))

    # ---- artifact (:514-560) ---------------------------------------------------------------------------
    def save(self, output_path: Path) -> Optional[Dict[str, Any]]:
        if not self.samples:
            log.warning("No samples collected.")
            return None
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        attempted = int(self.stats["num_samples"]) + int(self.stats["retrieval_failed"])
        final_stats = {
            "num_samples": self.stats["num_samples"],
            "path_exists_ratio": self.stats["path_exists"] / max(1, self.stats["num_samples"]),
            "retrieval_failed_ratio": self.stats["retrieval_failed"] / max(1, attempted),
            "avg_edges": statistics.mean(self.stats["edge_counts"]) if self.stats["edge_counts"] else 0,
            "avg_gt_len": statistics.mean(self.stats["path_lengths"]) if self.stats["path_lengths"] else 0,
        }
        payload = {"settings": self.cfg.to_metadata(), "stats": final_stats,
                   "samples": [self._sample_to_record(s) for s in self.samples]}
        torch.save(payload, output_path)
        log.info(f"Saved {len(self.samples)} samples to {output_path}")
        return final_stats

    @staticmethod
    def _sample_to_record(sample: GAgentSample) -> Dict[str, Any]:
        names = ("sample_id", "question", "question_emb", "edge_relations", "edge_scores", "edge_labels", "edge_head_locals",
                 "edge_tail_locals", "node_entity_ids", "node_embedding_ids", "start_entity_ids", "start_node_locals",
                 "answer_entity_ids", "answer_node_locals", "pair_start_node_locals", "pair_answer_node_locals",
                 "pair_edge_local_ids", "pair_edge_counts", "pair_shortest_lengths", "gt_path_edge_local_ids")
        rec = {n: getattr(sample, n) for n in names}
        rec.update(gt_path_exists=bool(sample.gt_path_exists), is_answer_reachable=bool(sample.is_answer_reachable),
                   is_dummy_agent=bool(sample.is_dummy_agent))
        return rec


__all__ = ["GAgentSettings", "GAgentSample", "GAgentBuilder", "SCORE_MODE_LOGITS", "SCORE_MODE_NODE_SOFTMAX"]
