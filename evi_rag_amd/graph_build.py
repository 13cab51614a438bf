"""Per-sample graph construction (G5): `build_graph` of scripts/build_retrieval_pipeline.py:1450-1603.

The reference walks a sample's string triples once, dropping self loops and repeated (h, r, t),
numbering entities in first-seen order, then labels shortest-path edges.  Here the strings are
coded to integers on the host (the only thing a host can do with them), and the integer work —
triple de-duplication, first-seen node numbering, answer-subgraph / seed / answer look-ups, BFS
labelling — runs on the device for a whole chunk of samples at once (one workgroup per sample).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import labelling, ops

PATH_MODES = ("undirected", "qa_directed")


@dataclass
class GraphRecord:
    """reference: GraphRecord, scripts/build_retrieval_pipeline.py:104-120."""
    graph_id: str
    node_entity_ids: List[int]
    node_embedding_ids: List[int]
    node_labels: List[str]
    edge_src: List[int]
    edge_dst: List[int]
    edge_relation_ids: List[int]
    positive_triple_mask: List[bool]
    pair_start_node_locals: List[int]
    pair_answer_node_locals: List[int]
    pair_edge_local_ids: List[int]
    pair_edge_counts: List[int]
    pair_shortest_lengths: List[int]


@dataclass
class CodedGraph:
    """Integer result of the device pass for one sample (codes are the caller's entity / relation codes)."""
    node_codes: np.ndarray      # [N_g] entity code of each local node, first-seen order
    edge_src: np.ndarray        # [E_g] local ids
    edge_dst: np.ndarray
    edge_rel: np.ndarray        # [E_g] relation codes
    kept_raw: np.ndarray        # [E_g] position of each kept edge in the sample's triple list
    q_local: List[int]
    a_local: List[int]
    answer_edges: List[int]     # kept-edge indices named by answer_subgraph, in order, with repeats


def _validate_path_mode(path_mode: str) -> str:
    mode = str(path_mode)
    if mode not in PATH_MODES:
        raise ValueError(f"Unsupported path_mode: {mode}. Expected one of {PATH_MODES}.")
    return mode


def index_graphs_coded(triples: Sequence[np.ndarray], q_codes: Sequence[Sequence[int]], a_codes: Sequence[Sequence[int]],
                       answer_subgraphs: Sequence[np.ndarray], *, dedup_edges: bool = True,
                       remove_self_loops: bool = True) -> List[CodedGraph]:
    """The device pass over a chunk of samples.  triples[s] is [T_s, 3] i64 (head code, relation code,
    tail code) in sample order; answer_subgraphs[s] likewise; q / a are entity codes."""
    S = len(triples)
    if S == 0:
        return []
    dev = labelling._dev()
    tri = [np.asarray(t, np.int64).reshape(-1, 3) for t in triples]
    sub = [np.asarray(t, np.int64).reshape(-1, 3) for t in answer_subgraphs]
    n_tri = np.asarray([t.shape[0] for t in tri], np.int64)
    n_sub = np.asarray([t.shape[0] for t in sub], np.int64)
    seg = np.concatenate([[0], np.cumsum(n_tri + n_sub)]).astype(np.int64)
    keys_h = np.concatenate([np.concatenate([tri[s], sub[s]]) for s in range(S)]) if seg[-1] else np.empty((0, 3), np.int64)
    keys = torch.from_numpy(np.ascontiguousarray(keys_h)).to(dev)
    seg_t = torch.from_numpy(seg).to(dev)
    limit_t = torch.from_numpy(n_tri).to(dev)
    drop = (keys[:, 0] == keys[:, 2]) if remove_self_loops else None
    if dedup_edges:
        first = ops.first_occurrence(keys, seg_t, drop)
    else:
        # every surviving triple is its own edge; look-ups still need the first equal triple
        first = ops.first_occurrence(keys, seg_t, drop)
        first_lookup = first
        own = torch.arange(keys.size(0), dtype=torch.int64, device=dev) - torch.repeat_interleave(
            seg_t[:-1], seg_t[1:] - seg_t[:-1])
        is_edge = own < torch.repeat_interleave(limit_t, seg_t[1:] - seg_t[:-1])
        first = torch.where(is_edge & (first >= 0), own.to(torch.int32), first)
    rank, count, uniq = ops.first_seen_rank(first, seg_t, limit_t)
    K = count.to(torch.int64)                                             # kept edges per sample
    kept_ptr = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    kept_ptr[1:] = torch.cumsum(K, 0)
    total_kept = int(kept_ptr[-1].item())
    # raw position (global row of `keys`) of every kept edge, sample by sample
    sample_of = torch.repeat_interleave(torch.arange(S, device=dev), K)
    within = torch.arange(total_kept, device=dev) - kept_ptr[:-1][sample_of]
    kept_local = uniq[(seg_t[:-1][sample_of] + within)].to(torch.int64)   # position inside the sample's list
    kept_rows = seg_t[:-1][sample_of] + kept_local
    kh, kr, kt = keys[kept_rows, 0], keys[kept_rows, 1], keys[kept_rows, 2]
    # node occurrences h0 t0 h1 t1 ... of the kept edges, then the q / a entities as look-ups
    q_l = [np.asarray(list(q), np.int64).reshape(-1) for q in q_codes]
    a_l = [np.asarray(list(a), np.int64).reshape(-1) for a in a_codes]
    n_q = np.asarray([x.size for x in q_l], np.int64)
    n_a = np.asarray([x.size for x in a_l], np.int64)
    K_h = K.cpu().numpy()
    seg2 = np.concatenate([[0], np.cumsum(2 * K_h + n_q + n_a)]).astype(np.int64)
    keys2 = torch.empty(int(seg2[-1]), dtype=torch.int64, device=dev)
    seg2_t = torch.from_numpy(seg2).to(dev)
    occ_base = seg2_t[:-1][sample_of] + 2 * within
    keys2[occ_base] = kh
    keys2[occ_base + 1] = kt
    qa_pos = np.concatenate([seg2[s] + 2 * K_h[s] + np.arange(n_q[s] + n_a[s]) for s in range(S)]).astype(np.int64) \
        if int((n_q + n_a).sum()) else np.empty(0, np.int64)
    if qa_pos.size:
        qa_val = np.concatenate([np.concatenate([q_l[s], a_l[s]]) for s in range(S)])
        keys2[torch.from_numpy(qa_pos).to(dev)] = torch.from_numpy(qa_val).to(dev)
    limit2_t = (2 * K).contiguous()
    first2 = ops.first_occurrence(keys2, seg2_t)
    rank2, count2, uniq2 = ops.first_seen_rank(first2, seg2_t, limit2_t)
    # ---- one D2H of everything the host needs
    rank_h, rank2_h, uniq2_h = rank.cpu().numpy(), rank2.cpu().numpy(), uniq2.cpu().numpy()
    count2_h = count2.cpu().numpy()
    keys2_h = keys2.cpu().numpy()
    kept_local_h, kr_h = kept_local.cpu().numpy(), kr.cpu().numpy()
    kept_ptr_h = kept_ptr.cpu().numpy()
    first_lookup_h = None if dedup_edges else first_lookup.cpu().numpy()
    out: List[CodedGraph] = []
    for s in range(S):
        k0, k1 = int(kept_ptr_h[s]), int(kept_ptr_h[s + 1])
        Ks = k1 - k0
        b = int(seg2[s])
        occ = rank2_h[b: b + 2 * Ks].astype(np.int64)
        nodes = keys2_h[b + uniq2_h[b: b + int(count2_h[s])]]
        look = rank2_h[b + 2 * Ks: int(seg2[s + 1])].astype(np.int64)
        ql, al = look[: n_q[s]], look[n_q[s]:]
        lo = int(seg[s]) + int(n_tri[s])
        sub_rank = rank_h[lo: int(seg[s + 1])].astype(np.int64)
        if dedup_edges:
            answer_edges = sub_rank[sub_rank >= 0].tolist()
        else:
            # every kept edge with the looked-up key, ascending (edge_key_to_indices lists, :1497)
            f_edges = first_lookup_h[int(seg[s]): lo]
            kept_of_raw = rank_h[int(seg[s]): lo]
            answer_edges = []
            for f in first_lookup_h[lo: int(seg[s + 1])]:
                if 0 <= f < n_tri[s]:
                    answer_edges.extend(kept_of_raw[np.nonzero(f_edges == f)[0]].tolist())
        out.append(CodedGraph(node_codes=nodes, edge_src=occ[0::2], edge_dst=occ[1::2], edge_rel=kr_h[k0:k1].astype(np.int64),
                              kept_raw=kept_local_h[k0:k1], q_local=ql[ql >= 0].tolist(), a_local=al[al >= 0].tolist(),
                              answer_edges=answer_edges))
    return out


def label_graphs(coded: Sequence[CodedGraph], *, path_mode: str = "undirected"):
    """Positive mask + pair lists per sample (:1501-1584): the answer_subgraph edges when they yield at
    least one (seed, answer) pair, otherwise the whole graph.  Two batched device labelling passes."""
    directed = _validate_path_mode(path_mode) == "qa_directed"
    S = len(coded)
    results: List[Optional[tuple]] = [None] * S
    with_sub = [s for s in range(S) if coded[s].answer_edges]
    subs: Dict[int, List[int]] = {}
    if with_sub:
        for s in with_sub:
            subs[s] = list(dict.fromkeys(coded[s].answer_edges))  # order-preserving dedup (:1520-1526)
        gb = labelling.GraphBatch([len(coded[s].node_codes) for s in with_sub],
                                  [coded[s].edge_src[subs[s]] for s in with_sub],
                                  [coded[s].edge_dst[subs[s]] for s in with_sub])
        res = labelling.shortest_path_union_mask_by_pair_batch(gb, [coded[s].q_local for s in with_sub],
                                                               [coded[s].a_local for s in with_sub], directed=directed)
        for s, (mask, ps, pa, pe, pc, pl) in zip(with_sub, res):
            if len(ps) > 0:
                sub = np.asarray(subs[s], np.int64)
                positive = np.zeros(coded[s].edge_src.shape[0], dtype=bool)
                positive[sub[np.nonzero(mask)[0]]] = True
                results[s] = (positive.tolist(), ps, pa, sub[np.asarray(pe, np.int64)].tolist() if pe else [], pc, pl)
    rest = [s for s in range(S) if results[s] is None]
    if rest:
        gb = labelling.GraphBatch([len(coded[s].node_codes) for s in rest], [coded[s].edge_src for s in rest],
                                  [coded[s].edge_dst for s in rest])
        res = labelling.shortest_path_union_mask_by_pair_batch(gb, [coded[s].q_local for s in rest],
                                                               [coded[s].a_local for s in rest], directed=directed)
        for s, (mask, ps, pa, pe, pc, pl) in zip(rest, res):
            results[s] = (np.asarray(mask, bool).tolist(), ps, pa, pe, pc, pl)
    return results


def validate_graph_record(graph: GraphRecord) -> None:
    """reference: _validate_graph_record, scripts/build_retrieval_pipeline.py:533-567."""
    n, e = len(graph.node_entity_ids), len(graph.edge_src)
    if len(graph.edge_dst) != e or len(graph.edge_relation_ids) != e:
        raise ValueError(f"Edge length mismatch for {graph.graph_id}: edges={e}.")
    if len(graph.positive_triple_mask) != e:
        raise ValueError(f"positive_triple_mask length mismatch for {graph.graph_id}: edges={e}.")
    if e > 0:
        if min(graph.edge_src) < 0 or min(graph.edge_dst) < 0:
            raise ValueError(f"Negative edge index detected for {graph.graph_id}.")
        if max(graph.edge_src) >= n or max(graph.edge_dst) >= n:
            raise ValueError(f"Edge index exceeds num_nodes for {graph.graph_id}.")
    if graph.pair_edge_counts and sum(graph.pair_edge_counts) != len(graph.pair_edge_local_ids):
        raise ValueError(f"pair_edge_counts sum mismatch for {graph.graph_id}.")
    if graph.pair_edge_local_ids and (min(graph.pair_edge_local_ids) < 0 or max(graph.pair_edge_local_ids) >= e):
        raise ValueError(f"pair_edge_local_ids out of range for {graph.graph_id}.")
    if graph.pair_start_node_locals and (min(graph.pair_start_node_locals) < 0 or max(graph.pair_start_node_locals) >= n):
        raise ValueError(f"pair_start_node_locals out of range for {graph.graph_id}.")
    if graph.pair_answer_node_locals and (min(graph.pair_answer_node_locals) < 0 or max(graph.pair_answer_node_locals) >= n):
        raise ValueError(f"pair_answer_node_locals out of range for {graph.graph_id}.")


def build_graphs(samples: Sequence, entity_vocab, relation_vocab, graph_ids: Sequence[str], *,
                 path_mode: str = "undirected", dedup_edges: bool = True, validate_graph_edges: bool = True,
                 remove_self_loops: bool = True) -> List[GraphRecord]:
    """A chunk of `Sample`s (attributes graph, q_entity, a_entity, answer_subgraph) -> GraphRecords.
    The vocabularies are consulted exactly as the reference does: entity_id / embedding_id once per new
    node in first-seen order, relation_id once per kept edge in edge order."""
    path_mode = _validate_path_mode(path_mode)
    tri, subs, qs, as_, ent_names = [], [], [], [], []
    rel_names: List[List[str]] = []
    for sample in samples:
        ecode: Dict[str, int] = {}
        rcode: Dict[str, int] = {}
        t = np.asarray([(ecode.setdefault(h, len(ecode)), rcode.setdefault(r, len(rcode)), ecode.setdefault(tl, len(ecode)))
                        for h, r, tl in sample.graph], np.int64).reshape(-1, 3)
        sub_rows = []
        for tr in (sample.answer_subgraph or []):
            if not isinstance(tr, tuple) or len(tr) != 3:
                continue  # the reference looks the WHOLE tuple up (:1505-1507): only 3-tuples can match an edge key
            h, r, tl = tr
            if h in ecode and r in rcode and tl in ecode:  # any other triple cannot match a graph edge
                sub_rows.append((ecode[h], rcode[r], ecode[tl]))
        tri.append(t)
        subs.append(np.asarray(sub_rows, np.int64).reshape(-1, 3))
        qs.append([ecode[e] for e in sample.q_entity if e in ecode])
        as_.append([ecode[e] for e in sample.a_entity if e in ecode])
        ent_names.append(list(ecode.keys()))
        rel_names.append(list(rcode.keys()))
    coded = index_graphs_coded(tri, qs, as_, subs, dedup_edges=bool(dedup_edges), remove_self_loops=bool(remove_self_loops))
    labels = label_graphs(coded, path_mode=path_mode)
    records = []
    for s, (cg, lab) in enumerate(zip(coded, labels)):
        names = [ent_names[s][c] for c in cg.node_codes.tolist()]
        positive, ps, pa, pe, pc, pl = lab
        rec = GraphRecord(
            graph_id=graph_ids[s],
            node_entity_ids=[entity_vocab.entity_id(n) for n in names],
            node_embedding_ids=[entity_vocab.embedding_id(n) for n in names],
            node_labels=names, edge_src=cg.edge_src.tolist(), edge_dst=cg.edge_dst.tolist(),
            edge_relation_ids=[relation_vocab.relation_id(rel_names[s][c]) for c in cg.edge_rel.tolist()],
            positive_triple_mask=list(positive), pair_start_node_locals=list(ps), pair_answer_node_locals=list(pa),
            pair_edge_local_ids=list(pe), pair_edge_counts=list(pc), pair_shortest_lengths=list(pl))
        if validate_graph_edges:
            validate_graph_record(rec)
        records.append(rec)
    return records


def build_graph(sample, entity_vocab, relation_vocab, graph_id: str, *, path_mode: str = "undirected",
                dedup_edges: bool = True, validate_graph_edges: bool = True, remove_self_loops: bool = True) -> GraphRecord:
    """Single-sample form with the reference's signature (build_graph, :1450-1603)."""
    return build_graphs([sample], entity_vocab, relation_vocab, [graph_id], path_mode=path_mode, dedup_edges=dedup_edges,
                        validate_graph_edges=validate_graph_edges, remove_self_loops=remove_self_loops)[0]


__all__ = ["GraphRecord", "CodedGraph", "index_graphs_coded", "label_graphs", "validate_graph_record", "build_graphs", "records_to_samples",
           "build_graph"]


def records_to_samples(records: Sequence[GraphRecord], seed_entity_ids: Sequence[Sequence[int]],
                       answer_entity_ids: Sequence[Sequence[int]], question_emb, *, questions: Optional[Sequence[str]] = None,
                       num_topics: int = 2, split: str = "") -> List[dict]:
    """GraphRecords -> the per-sample dictionaries the reference's materialisation stage serialises into its LMDBs
    (scripts/build_retrieval_pipeline.py:2141-2224: core keys + aux keys), ready for `packed_dataset.write_packed` — the
    LMDB-free hand-over from graph construction to the HBM-resident split.

    question_emb: [len(records), D] (array or tensor).  seed / answer entity ids are the question's GLOBAL entity ids; their
    local indices are the positions in the graph's `node_entity_ids` (ids outside the graph are dropped: `_local_indices`,
    :1742-1744); `topic_one_hot` marks the seeds (`F.one_hot(mask, num_topics)`, :2188-2191)."""
    q = np.asarray(question_emb.detach().cpu().numpy() if hasattr(question_emb, "detach") else question_emb, dtype=np.float32)
    if q.ndim != 2 or q.shape[0] != len(records):
        raise ValueError(f"question_emb must be [{len(records)}, D], got {tuple(q.shape)}")
    if int(num_topics) < 2:
        raise ValueError(f"num_topics must be >= 2, got {num_topics}")
    out = []
    for i, g in enumerate(records):
        num_nodes, num_edges = len(g.node_entity_ids), len(g.edge_src)
        if num_edges <= 0:
            raise ValueError(f"Invalid graph with zero edges for {g.graph_id} (split={split}). "
                             "Fix raw parquet/filters and rebuild; empty edge_index is unsupported.")
        labels = np.asarray(g.positive_triple_mask, dtype=np.float32)
        if labels.shape[0] != num_edges:
            raise ValueError(f"Label length mismatch for {g.graph_id}: labels={labels.shape[0]} vs num_edges={num_edges}. "
                             "Rebuild normalized parquet caches to match the updated schema.")
        q_entities, a_entities = seed_entity_ids[i], answer_entity_ids[i]
        if q_entities is None:
            raise ValueError(f"seed_entity_ids is null for {g.graph_id}")
        if a_entities is None:
            raise ValueError(f"answer_entity_ids is null for {g.graph_id}")
        position = {int(nid): idx for idx, nid in enumerate(g.node_entity_ids)}
        q_local = [position[int(t)] for t in q_entities if int(t) in position]
        a_local = [position[int(t)] for t in a_entities if int(t) in position]
        topic = np.zeros((num_nodes, int(num_topics)), dtype=np.float32)
        topic[:, 0] = 1.0
        if q_local:
            topic[q_local, 0] = 0.0
            topic[q_local, 1] = 1.0
        if g.pair_start_node_locals and len(g.pair_edge_counts) != len(g.pair_start_node_locals):
            raise ValueError(f"pair_edge_counts length {len(g.pair_edge_counts)} != pair_count "
                             f"{len(g.pair_start_node_locals)} for {g.graph_id}")
        sample = {
            "sample_id": g.graph_id,
            "edge_index": np.asarray([g.edge_src, g.edge_dst], dtype=np.int64),
            "edge_attr": np.asarray(g.edge_relation_ids, dtype=np.int64),
            "labels": labels,
            "num_nodes": num_nodes,
            "node_global_ids": np.asarray(g.node_entity_ids, dtype=np.int64),
            "node_embedding_ids": np.asarray(g.node_embedding_ids, dtype=np.int64),
            "question_emb": q[i: i + 1],
            "topic_one_hot": topic,
            "q_local_indices": np.asarray(q_local, dtype=np.int64),
            "a_local_indices": np.asarray(a_local, dtype=np.int64),
            "answer_entity_ids": np.asarray(list(a_entities), dtype=np.int64),
            "answer_entity_ids_len": np.asarray([len(a_entities)], dtype=np.int64),
            # aux keys (:2212-2224)
            "question": str(questions[i]) if questions is not None else "",
            "seed_entity_ids": np.asarray(list(q_entities), dtype=np.int64),
            "pair_start_node_locals": np.asarray(g.pair_start_node_locals, dtype=np.int64),
            "pair_answer_node_locals": np.asarray(g.pair_answer_node_locals, dtype=np.int64),
            "pair_edge_local_ids": np.asarray(g.pair_edge_local_ids, dtype=np.int64),
            "pair_edge_counts": np.asarray(g.pair_edge_counts, dtype=np.int64),
        }
        if g.pair_shortest_lengths is not None and len(g.pair_shortest_lengths) == len(g.pair_start_node_locals):
            sample["pair_shortest_lengths"] = np.asarray(g.pair_shortest_lengths, dtype=np.int64)
        out.append(sample)
    return out
