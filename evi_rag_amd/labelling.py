"""Host-side mirrors of the graph labelling and edge-selection helpers, backed by the CSR kernels.

  bfs_dist / shortest_path_union_mask_by_pair     scripts/build_retrieval_pipeline.py:610-631, 691-830
  shortest_path_single / has_connectivity         scripts/build_retrieval_pipeline.py:453-530, 946-979
  node_softmax_logit / select_topk_edges / select_start_edges
                                                   GAgentBuilder statics, src/data/components/g_agent_builder.py:595-724
  seed_onehop_stats                                scripts/seed_onehop_stats.py:96-117

The batched entry points take many graphs at once (one workgroup per graph / BFS job); the
single-graph functions keep the reference's signatures and return types.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops


def _dev() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("evi_rag_amd labelling runs on the MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class GraphBatch:
    """Several graphs flattened PyG-style on the device, with their CSR."""

    def __init__(self, num_nodes: Sequence[int], edge_src: Sequence[np.ndarray], edge_dst: Sequence[np.ndarray],
                 device: Optional[torch.device] = None):
        dev = device or _dev()
        self.device = dev
        self.B = len(num_nodes)
        self.node_ptr_h = np.concatenate([[0], np.cumsum(np.maximum(np.asarray(num_nodes, np.int64), 0))]).astype(np.int64)
        srcs, dsts, self.valid_ids, self.orig_counts = [], [], [], []
        counts = []
        for g in range(self.B):
            s = np.asarray(edge_src[g], np.int64).reshape(-1)
            d = np.asarray(edge_dst[g], np.int64).reshape(-1)
            n = int(num_nodes[g])
            self.orig_counts.append(int(s.shape[0]))
            ok = np.nonzero((s >= 0) & (d >= 0) & (s < n) & (d < n))[0]  # _valid_edge_indices (:638-647)
            self.valid_ids.append(ok)
            srcs.append(s[ok] + self.node_ptr_h[g])
            dsts.append(d[ok] + self.node_ptr_h[g])
            counts.append(ok.shape[0])
        self.edge_ptr_h = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        ei = np.stack([np.concatenate(srcs) if srcs else np.empty(0, np.int64),
                       np.concatenate(dsts) if dsts else np.empty(0, np.int64)])
        self.edge_index = torch.from_numpy(np.ascontiguousarray(ei)).to(dev)
        self.node_ptr = torch.from_numpy(self.node_ptr_h).to(dev)
        self.edge_ptr = torch.from_numpy(self.edge_ptr_h).to(dev)
        self.N = int(self.node_ptr_h[-1])
        self.E = int(self.edge_ptr_h[-1])
        self.csr = ops.graph_csr(self.edge_index, self.node_ptr, self.edge_ptr) if self.B > 0 else None


def _bfs(gb: GraphBatch, job_graph: np.ndarray, sources: List[np.ndarray], mode: int):
    """Runs len(job_graph) BFS jobs; returns (dist [total] i32 device tensor, dist_off host array)."""
    dev = gb.device
    J = len(job_graph)
    sizes = (gb.node_ptr_h[1:] - gb.node_ptr_h[:-1])[job_graph] if J else np.empty(0, np.int64)
    dist_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    src_ptr = np.concatenate([[0], np.cumsum([len(s) for s in sources])]).astype(np.int64)
    src_idx = np.concatenate(sources).astype(np.int64) if J and src_ptr[-1] > 0 else np.empty(0, np.int64)
    dist = torch.empty(max(int(dist_off[-1]), 1), dtype=torch.int32, device=dev)
    if J == 0:
        return dist, dist_off
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)  # noqa: E731
    jg, sp, si, do = t(job_graph, torch.int32), t(src_ptr, torch.int64), t(src_idx, torch.int64), t(dist_off[:-1], torch.int64)
    lib = _lib.load()
    c = gb.csr
    _lib.check(lib.evi_bfs_levels(ops._ptr(jg), ops._ptr(sp), ops._ptr(si), do.data_ptr() if J else None, J,
                                  ops._ptr(gb.node_ptr), c.in_ptr.data_ptr(), c.in_nbr.data_ptr(), c.out_ptr.data_ptr(),
                                  c.out_nbr.data_ptr(), int(mode), dist.data_ptr(), ops._stream(dev)))
    return dist, dist_off


def bfs_dist_batch(gb: GraphBatch, sources: Sequence[Sequence[int]], *, mode: int = 0) -> List[np.ndarray]:
    """One multi-source BFS per graph; sources are LOCAL node ids (invalid ones ignored)."""
    srcs = []
    for g in range(gb.B):
        s = np.asarray(list(sources[g]), np.int64).reshape(-1)
        n = gb.node_ptr_h[g + 1] - gb.node_ptr_h[g]
        srcs.append(s[(s >= 0) & (s < n)] + gb.node_ptr_h[g])
    dist, off = _bfs(gb, np.arange(gb.B, dtype=np.int32), srcs, mode)
    d = dist.cpu().numpy()
    return [d[off[g]: off[g + 1]].astype(np.int64) for g in range(gb.B)]


def bfs_dist(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int], sources: Sequence[int], *,
             directed: bool = False) -> List[int]:
    """Levels from `sources` (unreachable -1): _bfs_dist over _build_(un)directed_adjacency."""
    if num_nodes <= 0:
        return []
    gb = GraphBatch([num_nodes], [np.asarray(edge_src)], [np.asarray(edge_dst)])
    return bfs_dist_batch(gb, [sources], mode=1 if directed else 0)[0].tolist()


def shortest_path_union_mask_by_pair_batch(gb: GraphBatch, sources: Sequence[Sequence[int]],
                                           targets: Sequence[Sequence[int]], *, directed: bool = False):
    """Per graph: (mask list[bool] over the ORIGINAL edge list, pair_start, pair_answer, pair_edge_ids,
    pair_edge_counts, pair_lengths) — the reference's 6-tuple."""
    dev = gb.device
    starts, answers = [], []
    for g in range(gb.B):
        n = int(gb.node_ptr_h[g + 1] - gb.node_ptr_h[g])
        starts.append(sorted({int(s) for s in sources[g] if 0 <= int(s) < n}))
        answers.append(sorted({int(t) for t in targets[g] if 0 <= int(t) < n}))
    # BFS jobs: all seeds (mode fwd / undirected), then all answers (mode bwd / undirected)
    jobs_g, jobs_src, seed_job, ans_job = [], [], [], []
    for g in range(gb.B):
        seed_job.append([])
        for s in starts[g]:
            seed_job[g].append(len(jobs_g))
            jobs_g.append(g)
            jobs_src.append(np.asarray([s + gb.node_ptr_h[g]], np.int64))
    n_seed_jobs = len(jobs_g)
    for g in range(gb.B):
        ans_job.append([])
        for a in answers[g]:
            ans_job[g].append(len(jobs_g))
            jobs_g.append(g)
            jobs_src.append(np.asarray([a + gb.node_ptr_h[g]], np.int64))
    jobs_g = np.asarray(jobs_g, np.int32)
    if directed:
        d1, off1 = _bfs(gb, jobs_g[:n_seed_jobs], jobs_src[:n_seed_jobs], 1)
        d2, off2 = _bfs(gb, jobs_g[n_seed_jobs:], jobs_src[n_seed_jobs:], 2)
        dist = torch.cat([d1[: int(off1[-1])], d2[: max(int(off2[-1]), 1)]])
        dist_off = np.concatenate([off1[:-1], off2[:-1] + off1[-1]]).astype(np.int64)
    else:
        dist, off = _bfs(gb, jobs_g, jobs_src, 0)
        dist_off = off[:-1].astype(np.int64)
    # dense pair slots in (graph, seed asc, answer asc) order
    pg, ps, pa, pan, pmeta = [], [], [], [], []
    for g in range(gb.B):
        if gb.edge_ptr_h[g + 1] == gb.edge_ptr_h[g]:
            continue  # no valid edges: the reference returns no pairs (:705-706)
        for i, s in enumerate(starts[g]):
            for j, a in enumerate(answers[g]):
                pg.append(g)
                ps.append(seed_job[g][i])
                pa.append(ans_job[g][j])
                pan.append(a + gb.node_ptr_h[g])
                pmeta.append((g, s, a))
    P = len(pg)
    mask = torch.zeros(max(gb.E, 1), dtype=torch.uint8, device=dev)
    if P > 0:
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(device=dev, dtype=dt)  # noqa: E731
        pg_t, ps_t, pa_t, pan_t = t(pg, torch.int32), t(ps, torch.int32), t(pa, torch.int32), t(pan, torch.int64)
        doff_t = t(dist_off, torch.int64)
        plen = torch.empty(P, dtype=torch.int32, device=dev)
        pcnt = torch.empty(P, dtype=torch.int32, device=dev)
        lib = _lib.load()
        args = (ops._ptr(pg_t), ops._ptr(ps_t), ops._ptr(pa_t), ops._ptr(pan_t), P, ops._ptr(doff_t), dist.data_ptr(),
                ops._ptr(gb.edge_index), gb.E, ops._ptr(gb.node_ptr), ops._ptr(gb.edge_ptr), int(bool(directed)))
        _lib.check(lib.evi_shortest_path_pairs(0, *args, plen.data_ptr(), pcnt.data_ptr(), mask.data_ptr(), None, None,
                                               ops._stream(dev)))
        cnt_h = pcnt.cpu().numpy().astype(np.int64)
        len_h = plen.cpu().numpy().astype(np.int64)
        poff = np.concatenate([[0], np.cumsum(cnt_h)]).astype(np.int64)
        pids = torch.empty(max(int(poff[-1]), 1), dtype=torch.int64, device=dev)
        poff_t = t(poff[:-1], torch.int64)
        _lib.check(lib.evi_shortest_path_pairs(1, *args, plen.data_ptr(), pcnt.data_ptr(), mask.data_ptr(),
                                               poff_t.data_ptr(), pids.data_ptr(), ops._stream(dev)))
        pids_h = pids.cpu().numpy()
    mask_h = mask.cpu().numpy().astype(bool)
    per_graph = [dict(ps=[], pa=[], pe=[], pc=[], pl=[]) for _ in range(gb.B)]
    for p in range(P):
        g, s, a = pmeta[p]
        if len_h[p] < 0:
            continue
        r = per_graph[g]
        r["ps"].append(s)
        r["pa"].append(a)
        r["pl"].append(int(len_h[p]))
        r["pc"].append(int(cnt_h[p]))
        local = pids_h[poff[p]: poff[p + 1]] - gb.edge_ptr_h[g]
        r["pe"].extend(gb.valid_ids[g][local].tolist())  # back to positions in the caller's edge list
    results = []
    for g in range(gb.B):
        r = per_graph[g]
        full = np.zeros(gb.orig_counts[g], dtype=bool)
        full[gb.valid_ids[g]] = mask_h[gb.edge_ptr_h[g]: gb.edge_ptr_h[g + 1]]
        results.append((full, r["ps"], r["pa"], r["pe"], r["pc"], r["pl"]))
    return results


def shortest_path_union_mask_by_pair(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int],
                                     sources: Sequence[int], targets: Sequence[int], *, directed: bool = False):
    """Single-graph form with the reference's return value
    (mask, pair_start_nodes, pair_answer_nodes, pair_edge_local_ids, pair_edge_counts, pair_shortest_lengths)."""
    num_edges = len(edge_src)
    if num_nodes <= 0 or num_edges == 0 or len(sources) == 0 or len(targets) == 0:
        return [False] * num_edges, [], [], [], [], []
    gb = GraphBatch([num_nodes], [np.asarray(edge_src)], [np.asarray(edge_dst)])
    mask, ps, pa, pe, pc, pl = shortest_path_union_mask_by_pair_batch(gb, [sources], [targets], directed=directed)[0]
    return mask.tolist(), ps, pa, pe, pc, pl


# ---- G4: deterministic single shortest path, connectivity ----------------------------------------------

def shortest_path_single_batch(gb: GraphBatch, sources: Sequence[Sequence[int]], targets: Sequence[Sequence[int]],
                               *, path_cap: int = 32) -> List[Tuple[List[int], List[int]]]:
    """Per graph (edge ids, node ids) of the reference's single shortest path; sources / targets are
    LOCAL node ids.  ([], []) when there is no path; ([], [node]) when a target is itself a source."""
    dev = gb.device
    B = gb.B
    if B == 0:
        return []

    def flat(lists):
        ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
        idx = np.concatenate(lists).astype(np.int64) if ptr[-1] > 0 else np.empty(0, np.int64)
        return ptr, idx

    def to_global(lists):
        out = []
        for g in range(B):
            v = np.asarray(list(lists[g]), np.int64).reshape(-1)
            n = gb.node_ptr_h[g + 1] - gb.node_ptr_h[g]
            out.append(v[(v >= 0) & (v < n)] + gb.node_ptr_h[g])
        return out

    sp, si = flat(to_global(sources))
    tp, ti = flat(to_global(targets))
    sizes = gb.node_ptr_h[1:] - gb.node_ptr_h[:-1]
    dist_off = np.concatenate([[0], np.cumsum(2 * sizes)]).astype(np.int64)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)  # noqa: E731
    jg = torch.arange(B, dtype=torch.int32, device=dev)
    sp_t, si_t, tp_t, ti_t, do_t = t(sp, torch.int64), t(si, torch.int64), t(tp, torch.int64), t(ti, torch.int64), t(dist_off[:-1], torch.int64)
    ws = torch.empty(max(int(dist_off[-1]), 1), dtype=torch.int32, device=dev)
    lib = _lib.load()
    c = gb.csr
    while True:
        out_len = torch.empty(B, dtype=torch.int32, device=dev)
        out_nodes = torch.empty((B, path_cap + 1), dtype=torch.int64, device=dev)
        out_edges = torch.empty((B, path_cap), dtype=torch.int64, device=dev)
        _lib.check(lib.evi_shortest_path_single(
            ops._ptr(jg), ops._ptr(sp_t), ops._ptr(si_t), ops._ptr(tp_t), ops._ptr(ti_t), ops._ptr(do_t), B,
            ops._ptr(gb.node_ptr), ops._ptr(gb.edge_ptr), c.in_ptr.data_ptr(), c.in_nbr.data_ptr(), c.in_eid.data_ptr(),
            c.out_ptr.data_ptr(), c.out_nbr.data_ptr(), c.out_eid.data_ptr(), ws.data_ptr(), int(path_cap),
            out_len.data_ptr(), out_nodes.data_ptr(), out_edges.data_ptr(), ops._stream(dev)))
        lens = out_len.cpu().numpy()
        if lens.size == 0 or int(lens.max()) <= path_cap:
            break
        path_cap = int(lens.max())  # a longer path than the buffer: rerun once with room for it
    nodes_h, edges_h = out_nodes.cpu().numpy(), out_edges.cpu().numpy()
    results = []
    for g in range(B):
        L = int(lens[g])
        if L < 0:
            results.append(([], []))
            continue
        edges = gb.valid_ids[g][edges_h[g, :L]].tolist()  # back to positions in the caller's edge list
        results.append((edges, nodes_h[g, : L + 1].tolist()))
    return results


def shortest_path_single(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int], sources: Sequence[int],
                         targets: Sequence[int]) -> Tuple[List[int], List[int]]:
    """reference: _shortest_path_single, scripts/build_retrieval_pipeline.py:453-530."""
    if not len(sources) or not len(targets) or num_nodes <= 0:
        return [], []
    gb = GraphBatch([num_nodes], [np.asarray(edge_src)], [np.asarray(edge_dst)])
    return shortest_path_single_batch(gb, [sources], [targets])[0]


def has_connectivity(graph: Sequence[Tuple[str, str, str]], seeds: Sequence[str], answers: Sequence[str], *,
                     path_mode: str = "undirected") -> bool:
    """Whether any answer is reachable from the seeds (local indexing in first-seen order).
    reference: has_connectivity, scripts/build_retrieval_pipeline.py:946-979."""
    if not graph or not seeds or not answers:
        return False
    if path_mode not in ("undirected", "qa_directed"):
        raise ValueError(f"Unsupported path_mode: {path_mode}. Expected one of ('undirected', 'qa_directed').")
    node_index = {}
    src, dst = [], []
    for h, _, t in graph:
        src.append(node_index.setdefault(h, len(node_index)))
        dst.append(node_index.setdefault(t, len(node_index)))
    seed_ids = [node_index[s] for s in seeds if s in node_index]
    answer_ids = [node_index[a] for a in answers if a in node_index]
    if not seed_ids or not answer_ids:
        return False
    dist = bfs_dist(len(node_index), src, dst, seed_ids, directed=path_mode == "qa_directed")
    return any(dist[a] >= 0 for a in answer_ids)


# ---- C2-C4: canonical edge selection --------------------------------------------------------------------

def canonicalize_positive_edges(edge_src: Sequence[int], edge_dst: Sequence[int], edge_relation_ids: Sequence[int],
                                positive_mask: Sequence[bool], pair_edge_local_ids: Sequence[int],
                                pair_edge_counts: Sequence[int], question_embedding_norm: torch.Tensor,
                                relation_embeddings_norm: torch.Tensor):
    """Among parallel positive edges of the same unordered node pair keep the one whose relation is
    most cosine-similar to the question (first maximum in (relation_id, idx) order); returns
    (keep_mask, new_pair_edge_local_ids, new_pair_edge_counts).
    reference: _group_positive_edges_by_pair / _select_canonical_edge_indices / _filter_pair_edges /
    _canonicalize_graph_edges, scripts/build_retrieval_pipeline.py:840-932.  The relation scores are
    one device GEMV over the normalised relation table (`torch.mv` at :871 for every group at once);
    the grouping is the reference's own host logic."""
    n = len(edge_src)
    if n == 0 or not any(positive_mask):
        return list(positive_mask), list(pair_edge_local_ids), list(pair_edge_counts)
    if question_embedding_norm.numel() == 0:
        raise ValueError("question_embedding is empty")
    if relation_embeddings_norm.numel() == 0:
        raise ValueError("relation_embeddings are empty; cannot canonicalize positives.")
    if relation_embeddings_norm.dim() != 2 or question_embedding_norm.dim() != 1:
        raise ValueError("Embeddings must be 2D (relations) and 1D (question) for canonicalization.")
    if int(relation_embeddings_norm.size(1)) != int(question_embedding_norm.numel()):
        raise ValueError("Question embedding dim does not match relation embedding dim.")
    groups = {}
    for idx, keep in enumerate(positive_mask):
        if not keep:
            continue
        u, v = int(edge_src[idx]), int(edge_dst[idx])
        groups.setdefault((u, v) if u <= v else (v, u), []).append(idx)
    rel_scores = ops.linear_act(relation_embeddings_norm, question_embedding_norm.view(1, -1), None).view(-1).cpu()
    keep_mask = [False] * n
    for members in groups.values():
        if len(members) == 1:
            keep_mask[members[0]] = True
            continue
        ordered = sorted(members, key=lambda i: (int(edge_relation_ids[i]), i))
        s = rel_scores[torch.tensor([int(edge_relation_ids[i]) for i in ordered], dtype=torch.long)]
        keep_mask[ordered[int(torch.argmax(s).item())]] = True
    if not pair_edge_local_ids or not pair_edge_counts:
        return keep_mask, list(pair_edge_local_ids), list(pair_edge_counts)
    new_ids, new_counts, off = [], [], 0
    for c in pair_edge_counts:
        kept = [i for i in pair_edge_local_ids[off: off + c] if keep_mask[int(i)]]
        new_ids.extend(kept)
        new_counts.append(len(kept))
        off += c
    if off != len(pair_edge_local_ids):
        raise ValueError("pair_edge_counts do not sum to len(pair_edge_local_ids)")
    return keep_mask, new_ids, new_counts


# ---- GAgentBuilder statics ---------------------------------------------------------------------------

def node_softmax_logit(*, edge_scores: torch.Tensor, edge_head_locals: torch.Tensor, edge_tail_locals: torch.Tensor,
                       num_nodes: int) -> torch.Tensor:
    """reference: GAgentBuilder._node_softmax_logit, src/data/components/g_agent_builder.py:595-626."""
    if edge_scores.numel() == 0:
        return edge_scores
    dev = ops._require_gpu(edge_scores)
    s = ops._f32c(edge_scores.view(-1), "edge_scores")
    ei = torch.stack([edge_head_locals.to(dev, torch.int64).view(-1), edge_tail_locals.to(dev, torch.int64).view(-1)]).contiguous()
    out = torch.empty_like(s)
    lib = _lib.load()
    ws = ops._workspace(dev, "node_softmax", int(lib.evi_node_softmax_logit_workspace_bytes(int(num_nodes))))
    _lib.check(lib.evi_node_softmax_logit(ops._ptr(s), ops._ptr(ei), s.numel(), int(num_nodes), ops._ptr(out),
                                          ws.data_ptr(), ws.numel(), ops._stream(dev)))
    return out


def select_topk_edges(*, edge_scores: torch.Tensor, edge_top_k: int) -> torch.Tensor:
    """reference: GAgentBuilder._select_topk_edges, src/data/components/g_agent_builder.py:640-652."""
    scores = edge_scores.view(-1)
    num_edges = int(scores.numel())
    if num_edges <= 0:
        return torch.empty(0, dtype=torch.long)
    edge_top_k = int(edge_top_k)
    if edge_top_k <= 0:
        raise ValueError(f"edge_top_k must be > 0, got {edge_top_k}")
    if num_edges <= edge_top_k:
        return torch.arange(num_edges, dtype=torch.long, device=scores.device)
    ptr = torch.tensor([0, num_edges], dtype=torch.int64, device=scores.device)
    idx, _, _ = ops.segment_topk(scores, ptr, edge_top_k, want_scores=False)
    return idx[0].to(torch.long)


def select_start_edges(*, heads: torch.Tensor, tails: torch.Tensor, edge_scores: torch.Tensor,
                       start_node_locals: torch.Tensor, num_nodes: int, start_keep_ratio: float, start_min_edges: int,
                       start_max_edges: Optional[int]) -> torch.Tensor:
    """reference: GAgentBuilder._select_start_edges, src/data/components/g_agent_builder.py:655-724.
    Returns the sorted unique ids of the kept seed-incident edges."""
    dev = ops._require_gpu(edge_scores)
    seeds = torch.unique(start_node_locals.to(dev, torch.int64).view(-1))
    E = int(edge_scores.numel())
    if seeds.numel() == 0 or E == 0:
        return torch.empty(0, dtype=torch.long, device=dev)
    s = ops._f32c(edge_scores.view(-1), "edge_scores")
    ei = torch.stack([heads.to(dev, torch.int64).view(-1), tails.to(dev, torch.int64).view(-1)]).contiguous()
    N = int(num_nodes)
    node_ptr = torch.tensor([0, N], dtype=torch.int64, device=dev)
    edge_ptr = torch.tensor([0, E], dtype=torch.int64, device=dev)
    csr = ops.graph_csr(ei, node_ptr, edge_ptr)
    mask = torch.empty(E, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.evi_select_start_edges(ops._ptr(s), E, ops._ptr(seeds), seeds.numel(), csr.in_ptr.data_ptr(),
                                          csr.in_eid.data_ptr(), csr.out_ptr.data_ptr(), csr.out_eid.data_ptr(), N,
                                          float(start_keep_ratio), int(start_min_edges),
                                          -1 if start_max_edges is None else int(start_max_edges), ops._ptr(mask),
                                          status.data_ptr(), ops._stream(dev)))
    st = int(status.item())
    if st & 1:
        raise IndexError("start_node_locals contains a node outside [0, num_nodes)")
    return torch.nonzero(mask, as_tuple=False).view(-1)


def seed_onehop_stats(heads: torch.Tensor, tails: torch.Tensor, labels: torch.Tensor, seeds: torch.Tensor,
                      num_nodes: int) -> List[Tuple[int, int, int]]:
    """[(seed, incident edge count, positive incident count)] for the unique in-range seeds.
    reference: scripts/seed_onehop_stats.py:96-117."""
    dev = ops._require_gpu(heads)
    E = int(heads.numel())
    uniq = torch.unique(seeds.to(dev, torch.int64).view(-1))
    if uniq.numel() == 0:
        return []
    ei = torch.stack([heads.to(dev, torch.int64).view(-1), tails.to(dev, torch.int64).view(-1)]).contiguous()
    N = int(num_nodes)
    csr = ops.graph_csr(ei, torch.tensor([0, N], dtype=torch.int64, device=dev),
                        torch.tensor([0, E], dtype=torch.int64, device=dev))
    pos = (labels.to(dev).view(-1) > 0.5).to(torch.uint8).contiguous()
    if pos.numel() == 0:
        pos = torch.zeros(1, dtype=torch.uint8, device=dev)
    deg = torch.empty(uniq.numel(), dtype=torch.int32, device=dev)
    pdeg = torch.empty(uniq.numel(), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.evi_seed_onehop_stats(ops._ptr(uniq), uniq.numel(), pos.data_ptr(), csr.in_ptr.data_ptr(),
                                         csr.in_eid.data_ptr(), csr.out_ptr.data_ptr(), csr.out_eid.data_ptr(), N,
                                         deg.data_ptr(), pdeg.data_ptr(), ops._stream(dev)))
    return [(int(s), int(d), int(p)) for s, d, p in zip(uniq.tolist(), deg.tolist(), pdeg.tolist()) if d >= 0]


__all__ = ["GraphBatch", "canonicalize_positive_edges", "bfs_dist", "bfs_dist_batch", "shortest_path_union_mask_by_pair",
           "shortest_path_single", "shortest_path_single_batch", "has_connectivity",
           "shortest_path_union_mask_by_pair_batch", "node_softmax_logit", "select_topk_edges", "select_start_edges",
           "seed_onehop_stats"]
