"""Graph-side oracle: adjacency, BFS, shortest-path labelling, DDE, seed expansion, edge batching
(test infrastructure only).  Plain Python / numpy restatement; each function cites its reference.
"""
from __future__ import annotations

import math
from collections import deque
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .ranking import stable_desc_order

DIST_UNREACHABLE = -1


# ---- G1 ---------------------------------------------------------------------------------------------
def build_undirected_adjacency(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int]) -> List[List[int]]:
    """reference: _build_undirected_adjacency, scripts/build_retrieval_pipeline.py:570-586 —
    both directions unless self-loop, out-of-range edges skipped, each list sorted ascending."""
    adj: List[List[int]] = [[] for _ in range(num_nodes)]
    for u_raw, v_raw in zip(edge_src, edge_dst):
        u, v = int(u_raw), int(v_raw)
        if u < 0 or v < 0 or u >= num_nodes or v >= num_nodes:
            continue
        adj[u].append(v)
        if u != v:
            adj[v].append(u)
    for nbrs in adj:
        nbrs.sort()
    return adj


def build_directed_adjacency(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int]) -> List[List[int]]:
    """reference: _build_directed_adjacency, scripts/build_retrieval_pipeline.py:589-603."""
    adj: List[List[int]] = [[] for _ in range(num_nodes)]
    for u_raw, v_raw in zip(edge_src, edge_dst):
        u, v = int(u_raw), int(v_raw)
        if u < 0 or v < 0 or u >= num_nodes or v >= num_nodes:
            continue
        adj[u].append(v)
    for nbrs in adj:
        nbrs.sort()
    return adj


def adjacency_to_csr(adj: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    ptr = np.zeros(len(adj) + 1, dtype=np.int64)
    for i, nbrs in enumerate(adj):
        ptr[i + 1] = ptr[i] + len(nbrs)
    col = np.asarray([v for nbrs in adj for v in nbrs], dtype=np.int64)
    return ptr, col


# ---- G2 ---------------------------------------------------------------------------------------------
def bfs_dist(num_nodes: int, adjacency: Sequence[Sequence[int]], sources: Sequence[int]) -> List[int]:
    """Multi-source BFS levels, unreachable = -1.
    reference: _bfs_dist, scripts/build_retrieval_pipeline.py:610-631."""
    dist = [DIST_UNREACHABLE] * num_nodes
    if num_nodes <= 0:
        return dist
    q: deque = deque()
    for s_raw in sources:
        s = int(s_raw)
        if 0 <= s < num_nodes and dist[s] < 0:
            dist[s] = 0
            q.append(s)
    while q:
        u = q.popleft()
        du = dist[u] + 1
        for v in adjacency[u]:
            if dist[v] >= 0:
                continue
            dist[v] = du
            q.append(v)
    return dist


# ---- G3 ---------------------------------------------------------------------------------------------
def shortest_path_union_mask_by_pair(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int],
                                     sources: Sequence[int], targets: Sequence[int], *, directed: bool = False):
    """Union over (seed, answer) pairs of the edges lying on ANY shortest path, plus CSR pair lists.
    reference: _shortest_path_union_mask_by_pair (:691-752), ..._directed (:755-815),
    _select_shortest_edges_undirected/_directed (:650-688), scripts/build_retrieval_pipeline.py."""
    num_edges = len(edge_src)
    empty = ([False] * num_edges, [], [], [], [], [])
    if num_nodes <= 0 or num_edges == 0 or len(sources) == 0 or len(targets) == 0:
        return empty
    src = np.asarray(edge_src, dtype=np.int64)
    dst = np.asarray(edge_dst, dtype=np.int64)
    valid = np.nonzero((src >= 0) & (dst >= 0) & (src < num_nodes) & (dst < num_nodes))[0]
    if valid.size == 0:
        return empty
    sv, dv = src[valid], dst[valid]
    starts = sorted({int(s) for s in sources if 0 <= int(s) < num_nodes})
    answers = sorted({int(t) for t in targets if 0 <= int(t) < num_nodes})
    if not starts or not answers:
        return empty
    if directed:
        adj = build_directed_adjacency(num_nodes, edge_src, edge_dst)
        radj = build_directed_adjacency(num_nodes, edge_dst, edge_src)
    else:
        adj = radj = build_undirected_adjacency(num_nodes, edge_src, edge_dst)
    dist_from = {s: np.asarray(bfs_dist(num_nodes, adj, [s]), dtype=np.int64) for s in starts}
    dist_to = {a: np.asarray(bfs_dist(num_nodes, radj, [a]), dtype=np.int64) for a in answers}
    mask = np.zeros(num_edges, dtype=bool)
    ps: List[int] = []
    pa: List[int] = []
    pe: List[int] = []
    pc: List[int] = []
    pl: List[int] = []
    for s in starts:
        ds = dist_from[s]
        for a in answers:
            da = dist_to[a]
            dsa = int(ds[a])
            if dsa < 0:
                continue
            ps.append(s)
            pa.append(a)
            pl.append(dsa)
            uv = (ds[sv] >= 0) & (da[dv] >= 0) & (ds[sv] + 1 + da[dv] == dsa)
            if directed:
                keep = uv
            else:
                vu = (ds[dv] >= 0) & (da[sv] >= 0) & (ds[dv] + 1 + da[sv] == dsa)
                keep = uv | vu
            ids = valid[np.nonzero(keep)[0]]
            if ids.size > 0:
                mask[ids] = True
                pe.extend(ids.tolist())
            pc.append(int(ids.size))
    return mask.tolist(), ps, pa, pe, pc, pl


# ---- G4 ---------------------------------------------------------------------------------------------
def shortest_path_single(num_nodes: int, edge_src: Sequence[int], edge_dst: Sequence[int],
                         sources: Sequence[int], targets: Sequence[int]) -> Tuple[List[int], List[int]]:
    """Deterministic single shortest path (neighbour order (node, edge idx); tie -> smallest target).
    reference: _shortest_path_single, scripts/build_retrieval_pipeline.py:453-530."""
    if len(sources) == 0 or len(targets) == 0 or num_nodes <= 0:
        return [], []
    adjacency: List[List[Tuple[int, int]]] = [[] for _ in range(num_nodes)]
    for idx, (u_raw, v_raw) in enumerate(zip(edge_src, edge_dst)):
        u, v = int(u_raw), int(v_raw)
        if 0 <= u < num_nodes and 0 <= v < num_nodes:
            adjacency[u].append((v, idx))
            if u != v:
                adjacency[v].append((u, idx))
    for nbrs in adjacency:
        nbrs.sort()
    srcs = sorted({int(s) for s in sources if 0 <= int(s) < num_nodes})
    tgts = sorted({int(t) for t in targets if 0 <= int(t) < num_nodes})
    if not srcs or not tgts:
        return [], []
    dist = [-1] * num_nodes
    parent = [-1] * num_nodes
    parent_edge = [-1] * num_nodes
    q: deque = deque()
    for s in srcs:
        dist[s] = 0
        q.append(s)
    while q:
        cur = q.popleft()
        nd = dist[cur] + 1
        for nb, e_idx in adjacency[cur]:
            if dist[nb] != -1:
                continue
            dist[nb] = nd
            parent[nb] = cur
            parent_edge[nb] = e_idx
            q.append(nb)
    best, best_d = None, None
    for t in tgts:
        if dist[t] < 0:
            continue
        if best_d is None or dist[t] < best_d or (dist[t] == best_d and t < best):
            best, best_d = t, dist[t]
    if best is None:
        return [], []
    nodes_rev, edges_rev = [best], []
    cur = best
    src_set = set(srcs)
    while cur not in src_set:
        prev, edge = parent[cur], parent_edge[cur]
        if prev < 0 or edge < 0:
            return [], []
        edges_rev.append(edge)
        nodes_rev.append(prev)
        cur = prev
    return list(reversed(edges_rev)), list(reversed(nodes_rev))


def has_connectivity(graph, seeds, answers, *, directed: bool = False) -> bool:
    """reference: has_connectivity, scripts/build_retrieval_pipeline.py:946-979."""
    if not graph or not seeds or not answers:
        return False
    node_index = {}
    edge_src, edge_dst = [], []

    def local(n):
        if n not in node_index:
            node_index[n] = len(node_index)
        return node_index[n]

    for h, _, t in graph:
        edge_src.append(local(h))
        edge_dst.append(local(t))
    seed_ids = [node_index[s] for s in seeds if s in node_index]
    ans_ids = [node_index[a] for a in answers if a in node_index]
    if not seed_ids or not ans_ids:
        return False
    n = len(node_index)
    adj = build_directed_adjacency(n, edge_src, edge_dst) if directed else build_undirected_adjacency(n, edge_src, edge_dst)
    dist = bfs_dist(n, adj, seed_ids)
    return any(dist[a] >= 0 for a in ans_ids)


# ---- G6 / G7: DDE ---------------------------------------------------------------------------------------
def mean_propagate(x: np.ndarray, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """out[v] = mean_{(u->v)} x[u], 0 if v has no in-edge (f32, edges summed in edge order).
    reference: PEConv, src/models/components/graph.py:13-23 — PyG MessagePassing(aggr="mean",
    flow source_to_target).  The aggregation itself lives in un-vendored torch_geometric:
    PARITY UNPINNED at that boundary; restated from PyG's documented semantics."""
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    out = np.zeros_like(x)
    cnt = np.zeros(n, dtype=np.float32)
    for e in range(src.shape[0]):  # sequential f32 adds in edge order == torch index_add_ on CPU
        out[dst[e]] += x[src[e]]
        cnt[dst[e]] += np.float32(1.0)
    return (out / np.maximum(cnt, np.float32(1.0))[:, None]).astype(np.float32)


def mean_propagate_fast(x: np.ndarray, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """Same as mean_propagate but vectorised (np.add.at keeps edge order per destination)."""
    x = np.asarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    np.add.at(out, dst, x[src])
    cnt = np.bincount(dst, minlength=x.shape[0]).astype(np.float32)
    return (out / np.maximum(cnt, np.float32(1.0))[:, None]).astype(np.float32)


def dde(topic_one_hot: np.ndarray, edge_index: np.ndarray, num_rounds: int, num_reverse_rounds: int) -> List[np.ndarray]:
    """[f1..f_rounds, r1..r_rev]: forward rounds on edge_index, reverse rounds on edge_index.flip(0).
    reference: DDE.forward/_apply_rounds, src/models/components/graph.py:41-74."""
    src, dst = np.asarray(edge_index[0], np.int64), np.asarray(edge_index[1], np.int64)
    feats: List[np.ndarray] = []
    h = np.asarray(topic_one_hot, dtype=np.float32)
    for _ in range(num_rounds):
        h = mean_propagate_fast(h, src, dst)
        feats.append(h)
    h = np.asarray(topic_one_hot, dtype=np.float32)
    for _ in range(num_reverse_rounds):
        h = mean_propagate_fast(h, dst, src)
        feats.append(h)
    return feats


def node_structure_features(topic_one_hot: np.ndarray, edge_index: np.ndarray, num_rounds: int,
                            num_reverse_rounds: int, num_topics: int = 2) -> np.ndarray:
    """stack([topic, f.., r..], -1).reshape(N, -1): topic-major layout [N, C*(1+rounds+rev)].
    reference: Retriever._build_node_structure_features, src/models/components/retriever.py:519-553."""
    t = np.asarray(topic_one_hot, dtype=np.float32)
    if t.ndim == 1:
        t = t[:, None]
    t = t[:, :num_topics]
    feats = [t] + dde(t, edge_index, num_rounds, num_reverse_rounds)
    return np.stack(feats, axis=-1).reshape(t.shape[0], -1).astype(np.float32)


# ---- G9: node-softmax logit + global top-k ---------------------------------------------------------------
PROB_EPS = 1e-6


def node_softmax_logit(edge_scores: np.ndarray, heads: np.ndarray, tails: np.ndarray, num_nodes: int) -> np.ndarray:
    """p = (softmax over the head's out-edges + softmax over the tail's in-edges) / 2, clamped to
    [1e-6, 1-1e-6], then logit.  reference: GAgentBuilder._node_softmax_logit,
    src/data/components/g_agent_builder.py:595-626."""
    s = np.asarray(edge_scores, dtype=np.float32)
    if s.size == 0:
        return s

    def side(idx):
        mx = np.full(num_nodes, -np.inf, dtype=np.float32)
        np.maximum.at(mx, idx, s)
        ex = np.exp(s - mx[idx]).astype(np.float32)
        sm = np.zeros(num_nodes, dtype=np.float32)
        np.add.at(sm, idx, ex)
        return (ex / np.maximum(sm[idx], np.float32(PROB_EPS))).astype(np.float32)

    prob = ((side(np.asarray(heads, np.int64)) + side(np.asarray(tails, np.int64))) * np.float32(0.5)).astype(np.float32)
    prob = np.clip(prob, np.float32(PROB_EPS), np.float32(1.0 - PROB_EPS)).astype(np.float32)
    return (np.log(prob) - np.log1p(-prob)).astype(np.float32)


def select_topk_edges(edge_scores: np.ndarray, edge_top_k: int) -> np.ndarray:
    """All edges if E <= k, else the first k of argsort(descending, stable).
    reference: GAgentBuilder._select_topk_edges, src/data/components/g_agent_builder.py:640-652."""
    s = np.asarray(edge_scores, dtype=np.float32).reshape(-1)
    if s.size == 0:
        return np.empty(0, dtype=np.int64)
    if int(edge_top_k) <= 0:
        raise ValueError(f"edge_top_k must be > 0, got {edge_top_k}")
    if s.size <= edge_top_k:
        return np.arange(s.size, dtype=np.int64)
    return stable_desc_order(s)[:edge_top_k].astype(np.int64)


# ---- G8: undirected one-hop seed expansion ------------------------------------------------------------------
def select_start_edges(heads: np.ndarray, tails: np.ndarray, edge_scores: np.ndarray, start_node_locals: np.ndarray,
                       num_nodes: int, start_keep_ratio: float, start_min_edges: int,
                       start_max_edges: Optional[int]) -> np.ndarray:
    """For each unique seed keep its top min(deg, min(max_edges, max(min_edges, ceil(deg*ratio))))
    incident edges (head OR tail) by score; ties by incidence order (heads block, then tails block,
    ascending edge id); returns sorted unique edge ids.
    reference: GAgentBuilder._select_start_edges, src/data/components/g_agent_builder.py:655-724."""
    heads = np.asarray(heads, np.int64).reshape(-1)
    tails = np.asarray(tails, np.int64).reshape(-1)
    scores = np.asarray(edge_scores, np.float32).reshape(-1)
    start_nodes = np.unique(np.asarray(start_node_locals, np.int64).reshape(-1))
    E = scores.shape[0]
    if start_nodes.size == 0 or E == 0:
        return np.empty(0, dtype=np.int64)
    deg = np.bincount(heads, minlength=num_nodes) + np.bincount(tails, minlength=num_nodes)
    deg_s = deg[start_nodes]
    # torch.ceil(deg.float() * ratio): the product is formed in f32
    k_s = np.ceil(deg_s.astype(np.float32) * np.float32(start_keep_ratio)).astype(np.int64)
    if start_min_edges > 0:
        k_s = np.maximum(k_s, int(start_min_edges))
    if start_max_edges is not None:
        k_s = np.zeros_like(k_s) if int(start_max_edges) == 0 else np.minimum(k_s, int(start_max_edges))
    k_s = np.minimum(k_s, deg_s)
    if k_s.size == 0 or int(k_s.max()) == 0:
        return np.empty(0, dtype=np.int64)
    inc_nodes = np.concatenate([heads, tails])
    inc_edges = np.concatenate([np.arange(E), np.arange(E)]).astype(np.int64)
    inc_scores = np.concatenate([scores, scores])
    start_mask = np.zeros(num_nodes, dtype=bool)
    start_mask[start_nodes] = True
    keep_inc = start_mask[inc_nodes]
    if not keep_inc.any():
        return np.empty(0, dtype=np.int64)
    nodes, edges, sc = inc_nodes[keep_inc], inc_edges[keep_inc], inc_scores[keep_inc]
    order_score = stable_desc_order(sc)
    nodes_sorted, edges_sorted = nodes[order_score], edges[order_score]
    order_node = np.argsort(nodes_sorted, kind="stable")
    nodes_g, edges_g = nodes_sorted[order_node], edges_sorted[order_node]
    counts = np.bincount(nodes_g, minlength=num_nodes)
    offsets = np.cumsum(counts) - counts
    pos = np.arange(nodes_g.shape[0]) - offsets[nodes_g]
    k_per_node = np.zeros(num_nodes, dtype=np.int64)
    k_per_node[start_nodes] = k_s
    keep = pos < k_per_node[nodes_g]
    if not keep.any():
        return np.empty(0, dtype=np.int64)
    return np.unique(edges_g[keep]).astype(np.int64)


# ---- G10 ----------------------------------------------------------------------------------------------------
def seed_onehop_stats(heads: np.ndarray, tails: np.ndarray, labels: np.ndarray, seeds: np.ndarray, num_nodes: int):
    """Per unique valid seed: (incident edge count, positive incident count).
    reference: scripts/seed_onehop_stats.py:96-117 (bincount(heads) + bincount(tails))."""
    heads = np.asarray(heads, np.int64)
    tails = np.asarray(tails, np.int64)
    deg = np.bincount(heads, minlength=num_nodes) + np.bincount(tails, minlength=num_nodes)
    pos = np.asarray(labels).reshape(-1) > 0.5
    pdeg = np.bincount(heads[pos], minlength=num_nodes) + np.bincount(tails[pos], minlength=num_nodes)
    out = []
    for s in np.unique(np.asarray(seeds, np.int64)).tolist():
        if 0 <= s < num_nodes:
            out.append((s, int(deg[s]), int(pdeg[s])))
    return out


# ---- G11 ----------------------------------------------------------------------------------------------------
def compute_edge_batch(edge_index: np.ndarray, node_ptr: np.ndarray, num_graphs: int):
    """edge -> graph id via bucketize(head, ptr[1:], right=True) (ptr[g] <= i < ptr[g+1]); raises the
    reference's ValueErrors; returns (edge_batch, edge_ptr).
    reference: compute_edge_batch, src/utils/graph_utils.py:50-104."""
    edge_index = np.asarray(edge_index, np.int64)
    node_ptr = np.asarray(node_ptr, np.int64)
    if edge_index.ndim != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    if node_ptr.size != num_graphs + 1:
        raise ValueError(f"node_ptr length mismatch: got {node_ptr.size} expected {num_graphs + 1}")
    eb = np.searchsorted(node_ptr[1:], edge_index[0], side="right")
    tb = np.searchsorted(node_ptr[1:], edge_index[1], side="right")
    if eb.size > 0 and (eb.min() < 0 or eb.max() >= num_graphs):
        raise ValueError("edge_batch contains out-of-range indices")
    if not np.array_equal(eb, tb):
        raise ValueError("edge_index crosses graph boundaries; head/tail graph assignments differ.")
    if eb.size > 1 and not np.all(eb[:-1] <= eb[1:]):
        raise ValueError("edge_batch is not non-decreasing along the flattened edge list")
    counts = np.bincount(eb, minlength=num_graphs).astype(np.int64)
    edge_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return eb.astype(np.int64), edge_ptr


def compute_qa_edge_mask(edge_index: np.ndarray, num_nodes: int, q_local_indices: np.ndarray,
                         a_local_indices: np.ndarray) -> np.ndarray:
    """near[e] = head in (Q u A) or tail in (Q u A).
    reference: compute_qa_edge_mask, src/utils/graph_utils.py:107-153."""
    edge_index = np.asarray(edge_index, np.int64)
    qa = np.concatenate([np.asarray(q_local_indices, np.int64).reshape(-1), np.asarray(a_local_indices, np.int64).reshape(-1)])
    if qa.size == 0:
        return np.zeros(edge_index.shape[1], dtype=bool)
    if qa.min() < 0 or qa.max() >= num_nodes:
        raise ValueError("q/a local indices out of range")
    node_mask = np.zeros(num_nodes, dtype=bool)
    node_mask[qa] = True
    return node_mask[edge_index[0]] | node_mask[edge_index[1]]


# ---- G5: build_graph (id-coded) ---------------------------------------------------------------------------
def build_graph_ids(triples, q_entities, a_entities, answer_subgraph, ent_struct, ent_emb, *, directed: bool = False,
                    dedup_edges: bool = True, remove_self_loops: bool = True):
    """The integer core of build_graph: `triples` [T, 3] = (entity index, relation id, entity index)
    in sample order; entities are indices into ent_struct / ent_emb (the vocabulary lookups).
    Local node ids are assigned in first-seen order (head, then tail, of each KEPT edge); self loops
    and repeated (h, r, t) triples are dropped before indexing; positives come from the
    answer_subgraph edges when they yield at least one (seed, answer) pair, else from the whole graph.
    reference: build_graph, scripts/build_retrieval_pipeline.py:1450-1603."""
    node_index = {}
    node_entity_ids: List[int] = []
    node_embedding_ids: List[int] = []

    def local_index(ent: int) -> int:
        if ent not in node_index:
            node_index[ent] = len(node_entity_ids)
            node_entity_ids.append(int(ent_struct[ent]))
            node_embedding_ids.append(int(ent_emb[ent]))
        return node_index[ent]

    edge_src: List[int] = []
    edge_dst: List[int] = []
    edge_rel: List[int] = []
    key_to_indices = {}
    for h, r, t in (tuple(int(v) for v in row) for row in np.asarray(triples, np.int64).reshape(-1, 3)):
        if remove_self_loops and h == t:
            continue
        key = (h, r, t)
        if dedup_edges and key in key_to_indices:
            continue
        edge_src.append(local_index(h))
        edge_dst.append(local_index(t))
        edge_rel.append(r)
        key_to_indices.setdefault(key, []).append(len(edge_src) - 1)
    q_local = [node_index[int(e)] for e in q_entities if int(e) in node_index]
    a_local = [node_index[int(e)] for e in a_entities if int(e) in node_index]
    answer_edges: List[int] = []
    for row in np.asarray(answer_subgraph, np.int64).reshape(-1, 3):
        answer_edges.extend(key_to_indices.get(tuple(int(v) for v in row), []))
    n = len(node_entity_ids)
    full = None
    if answer_edges:
        sub = list(dict.fromkeys(answer_edges))  # order-preserving dedup (:1520-1526)
        mask, ps, pa, pe, pc, pl = shortest_path_union_mask_by_pair(
            n, [edge_src[i] for i in sub], [edge_dst[i] for i in sub], q_local, a_local, directed=directed)
        if len(ps) > 0:
            positive = [False] * len(edge_src)
            for j, keep in enumerate(mask):
                if keep:
                    positive[sub[j]] = True
            full = (positive, ps, pa, [sub[j] for j in pe], pc, pl)
    if full is None:
        full = shortest_path_union_mask_by_pair(n, edge_src, edge_dst, q_local, a_local, directed=directed)
    positive, ps, pa, pe, pc, pl = full
    return {"node_entity_ids": node_entity_ids, "node_embedding_ids": node_embedding_ids, "edge_src": edge_src,
            "edge_dst": edge_dst, "edge_rel": edge_rel, "positive": list(positive), "pair_start": ps, "pair_answer": pa,
            "pair_edges": pe, "pair_counts": pc, "pair_len": pl}
