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
    """Several graphs flattened PyG-style on the device, with their CSR.  The per-graph edge lists are concatenated once
    and filtered / offset with whole-batch numpy operations (no per-graph arithmetic)."""

    def __init__(self, num_nodes: Sequence[int], edge_src: Sequence[np.ndarray], edge_dst: Sequence[np.ndarray],
                 device: Optional[torch.device] = None):
        dev = device or _dev()
        self.device = dev
        self.B = B = len(num_nodes)
        nn = np.maximum(np.asarray(num_nodes, np.int64).reshape(-1), 0)
        self.node_ptr_h = np.concatenate([[0], np.cumsum(nn)]).astype(np.int64)
        srcs = [np.asarray(a, np.int64).reshape(-1) for a in edge_src]
        dsts = [np.asarray(a, np.int64).reshape(-1) for a in edge_dst]
        cnt0 = np.asarray([a.shape[0] for a in srcs], np.int64)
        self.orig_counts = cnt0.tolist()
        off0 = np.concatenate([[0], np.cumsum(cnt0)]).astype(np.int64)
        s_all = np.concatenate(srcs) if B else np.empty(0, np.int64)
        d_all = np.concatenate(dsts) if B else np.empty(0, np.int64)
        if s_all.shape[0] != d_all.shape[0]:
            raise ValueError("edge_src and edge_dst differ in length")
        g_of = np.repeat(np.arange(B, dtype=np.int64), cnt0)
        n_of = nn[g_of] if B else np.empty(0, np.int64)
        # _valid_edge_indices (:638-647): 0 <= s, d < n — as unsigned compares (a negative id is a huge unsigned one): two passes
        ok = (s_all.view(np.uint64) < n_of.view(np.uint64)) & (d_all.view(np.uint64) < n_of.view(np.uint64))
        base = self.node_ptr_h[g_of] if B else np.empty(0, np.int64)
        if bool(ok.all()):
            # the common case (graphs out of build_graph): nothing to filter, no index arrays — positions map to themselves
            self.valid_ids = None
            self.edge_ptr_h = off0
            ei = np.empty((2, s_all.shape[0]), np.int64)
            np.add(s_all, base, out=ei[0])
            np.add(d_all, base, out=ei[1])
        else:
            keep = np.nonzero(ok)[0]
            g_keep = g_of[keep]
            counts = np.bincount(g_keep, minlength=B).astype(np.int64) if B else np.empty(0, np.int64)
            self.edge_ptr_h = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            pos = keep - off0[g_keep]  # position of every kept edge in its own graph's original list
            self.valid_ids = np.split(pos, self.edge_ptr_h[1:-1]) if B else []
            ei = np.stack([s_all[keep] + base[keep], d_all[keep] + base[keep]]) if B else np.empty((2, 0), np.int64)
        self.edge_index = torch.from_numpy(np.ascontiguousarray(ei)).to(dev)
        self.node_ptr = torch.from_numpy(self.node_ptr_h).to(dev)
        self.edge_ptr = torch.from_numpy(self.edge_ptr_h).to(dev)
        self.N = int(self.node_ptr_h[-1])
        self.E = int(self.edge_ptr_h[-1])
        self.csr = ops.graph_csr(self.edge_index, self.node_ptr, self.edge_ptr, num_nodes=self.N) if self.B > 0 else None


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
    _lib.check(lib.evi_bfs_levels_edges(ops._ptr(jg), ops._ptr(sp), ops._ptr(si), do.data_ptr() if J else None, J,
                                        ops._ptr(gb.node_ptr), ops._ptr(gb.edge_ptr), ops._ptr(gb.edge_index), gb.E,
                                        c.in_ptr.data_ptr(), c.in_nbr.data_ptr(), c.out_ptr.data_ptr(),
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


def _upload_packed(dev: torch.device, arrays: Sequence[np.ndarray]) -> List[torch.Tensor]:
    """Several small host arrays -> device tensors with ONE host-to-device copy (8-byte aligned segments of one buffer)."""
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 7) // 8 * 8
    host = np.empty(max(total, 8), np.uint8)
    for a, o in zip(arrays, offs):
        host[o: o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    buf = torch.from_numpy(host).to(dev, non_blocking=True)
    out = []
    for a, o in zip(arrays, offs):
        dt = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32}[a.dtype]
        out.append(buf[o: o + a.nbytes].view(dt))
    return out


def _unique_in_graph(ptr: np.ndarray, idx: np.ndarray, node_ptr_h: np.ndarray):
    """Sorted unique batch-global node ids that lie inside their own graph, and the graph of each.  Global ids of
    different graphs are disjoint, so one np.unique orders them by (graph, local id) — `sorted(set(...))` per graph."""
    ptr = np.asarray(ptr, np.int64).reshape(-1)
    idx = np.asarray(idx, np.int64).reshape(-1)
    B = node_ptr_h.shape[0] - 1
    if ptr.shape[0] != B + 1 or int(ptr[-1]) != idx.shape[0]:
        raise ValueError("seed / answer ptr must have B + 1 entries and end at len(idx)")
    g = np.repeat(np.arange(B, dtype=np.int64), np.diff(ptr))
    ok = (idx >= node_ptr_h[g]) & (idx < node_ptr_h[g + 1])
    u = np.unique(idx[ok])
    ug = np.searchsorted(node_ptr_h, u, side="right") - 1
    return u, ug


_PAIR_IDS_BOUND_LIMIT = 1 << 26  # entries (512 MiB of int64): beyond it label_pairs_flat sizes the id buffer from the exact total


class FlatPairLabels:
    """Shortest-path labels of a whole batch as FLAT arrays (no per-graph Python objects).

    Pair SLOTS are dense: every (seed, answer) combination of every graph that has edges, in (graph, seed asc, answer asc)
    order — `pair_ptr_h[g] .. pair_ptr_h[g + 1]` are graph g's slots.  A slot without a path has pair_len = -1 and
    pair_count = 0 (the reference emits no pair for it, scripts/build_retrieval_pipeline.py:722-724).

      device:  mask [E] u8 (positive edges, batch-global edge order) · pair_len [P] i32 · pair_count [P] i32 ·
               pair_edge_off [P + 1] i64 (exclusive cumsum of the counts) · pair_edge_ids [capacity] i64 (batch-global
               edge ids, ascending inside a pair; the first pair_edge_off[P] entries are valid)
      host:    pair_graph_h [P] · pair_start_h / pair_answer_h [P] (graph-LOCAL node ids) · pair_ptr_h [B + 1]

    Nothing has been read back when the object is returned; `per_graph()` does the D2H copies and builds the
    reference's per-graph 6-tuples on demand."""

    def __init__(self, B, E, edge_ptr_h, mask, pair_len, pair_count, pair_edge_off, pair_edge_ids, pair_graph_h,
                 pair_start_h, pair_answer_h, pair_ptr_h):
        self.B, self.E, self.P = int(B), int(E), int(pair_graph_h.shape[0])
        self.edge_ptr_h = edge_ptr_h
        self.mask, self.pair_len, self.pair_count = mask, pair_len, pair_count
        self.pair_edge_off, self.pair_edge_ids = pair_edge_off, pair_edge_ids
        self.pair_graph_h, self.pair_start_h, self.pair_answer_h, self.pair_ptr_h = pair_graph_h, pair_start_h, pair_answer_h, pair_ptr_h

    def per_graph(self, valid_ids: Optional[Sequence[np.ndarray]] = None, orig_counts: Optional[Sequence[int]] = None):
        """[(mask bool [E_g], pair_start, pair_answer, pair_edge_ids, pair_edge_counts, pair_lengths)] — the reference's
        return value per graph (edge ids graph-local).  valid_ids / orig_counts (GraphBatch): map the ids back to the
        positions of the caller's unfiltered edge lists."""
        B, P = self.B, self.P
        mask_h = self.mask[: self.E].cpu().numpy().astype(bool)
        if P > 0:
            len_h = self.pair_len.cpu().numpy().astype(np.int64)
            cnt_h = self.pair_count.cpu().numpy().astype(np.int64)
            off_h = self.pair_edge_off.cpu().numpy()
            ids_h = self.pair_edge_ids[: int(off_h[-1])].cpu().numpy()
        results = []
        for g in range(B):
            e0, e1 = int(self.edge_ptr_h[g]), int(self.edge_ptr_h[g + 1])
            if valid_ids is not None:
                full = np.zeros(int(orig_counts[g]), dtype=bool)
                full[valid_ids[g]] = mask_h[e0:e1]
            else:
                full = mask_h[e0:e1]
            p0, p1 = int(self.pair_ptr_h[g]), int(self.pair_ptr_h[g + 1])
            if p1 == p0:
                results.append((full, [], [], [], [], []))
                continue
            sel = np.nonzero(len_h[p0:p1] >= 0)[0] + p0
            if sel.shape[0] == p1 - p0:      # every slot has a path: the graph's ids are one contiguous run
                local = ids_h[int(off_h[p0]): int(off_h[p1])] - e0
            else:
                local = np.concatenate([ids_h[int(off_h[p]): int(off_h[p + 1])] for p in sel]) - e0 if sel.shape[0] else np.empty(0, np.int64)
            if valid_ids is not None:
                local = valid_ids[g][local]
            results.append((full, self.pair_start_h[sel].tolist(), self.pair_answer_h[sel].tolist(), local.tolist(),
                            cnt_h[sel].tolist(), len_h[sel].tolist()))
        return results


def label_pairs_flat(edge_index: torch.Tensor, node_ptr: torch.Tensor, edge_ptr: torch.Tensor, seed_ptr, seed_idx, answer_ptr,
                     answer_idx, *, directed: bool = False, csr=None, node_ptr_host: Optional[np.ndarray] = None,
                     edge_ptr_host: Optional[np.ndarray] = None) -> FlatPairLabels:
    """Shortest-path (seed, answer) labelling of a whole batch from flat arrays, results as flat device arrays
    (`FlatPairLabels`) — `_shortest_path_union_mask_by_pair(_directed)` for every graph of the batch at once
    (scripts/build_retrieval_pipeline.py:691-815).

      edge_index [2, E] i64 (batch-global node ids, every endpoint inside its own graph: what `build_graph` / the PyG
      collation produce), node_ptr / edge_ptr [B + 1] i64 — device tensors, as they lie in a collated batch;
      seed_ptr / seed_idx, answer_ptr / answer_idx — HOST arrays (a handful of entries per graph): batch-global node ids
      grouped by graph, the form `q_local_indices` / `a_local_indices` have after collation
      (src/data/g_retrieval_dataset.py:29-37).  Duplicates and ids outside their graph are dropped like the reference's
      `sorted({s for s in sources if 0 <= s < num_nodes})`.

    Host work: whole-batch numpy on the seed / answer lists (job and pair tables), ONE packed host-to-device copy.
    Device work: CSR (unless given), one BFS job per unique seed and per unique answer, evi_shortest_path_pairs pass 0
    (lengths, counts, mask), a device cumsum, pass 1 (edge ids).  No device-to-host read-back: the id buffer is sized by
    the host-side bound sum over pair slots of E_g."""
    dev = ops._require_gpu(edge_index, node_ptr, edge_ptr)
    ei = ops._i64c(edge_index, "edge_index")
    nptr = ops._i64c(node_ptr.view(-1), "node_ptr")
    eptr = ops._i64c(edge_ptr.view(-1), "edge_ptr")
    B = nptr.numel() - 1
    node_ptr_h = np.asarray(node_ptr_host, np.int64) if node_ptr_host is not None else nptr.cpu().numpy()
    edge_ptr_h = np.asarray(edge_ptr_host, np.int64) if edge_ptr_host is not None else eptr.cpu().numpy()
    N, E = int(node_ptr_h[-1]), int(edge_ptr_h[-1])
    if ei.size(1) != E:
        raise ValueError(f"edge_ptr ends at {E} but edge_index holds {ei.size(1)} edges")
    seeds_u, sg = _unique_in_graph(seed_ptr, seed_idx, node_ptr_h)
    ans_u, ag = _unique_in_graph(answer_ptr, answer_idx, node_ptr_h)
    S, A = seeds_u.shape[0], ans_u.shape[0]
    s_cnt = np.bincount(sg, minlength=B).astype(np.int64)
    a_cnt = np.bincount(ag, minlength=B).astype(np.int64)
    e_cnt = np.diff(edge_ptr_h)
    ppg = np.where(e_cnt > 0, s_cnt * a_cnt, 0)  # a graph without edges yields no pairs (:705-706)
    pair_ptr_h = np.concatenate([[0], np.cumsum(ppg)]).astype(np.int64)
    P = int(pair_ptr_h[-1])
    mask = torch.zeros(max(E, 1), dtype=torch.uint8, device=dev)
    empty = lambda dt: torch.empty(0, dtype=dt, device=dev)  # noqa: E731
    if P == 0:
        z = np.empty(0, np.int64)
        return FlatPairLabels(B, E, edge_ptr_h, mask, empty(torch.int32), empty(torch.int32), torch.zeros(1, dtype=torch.int64, device=dev),
                              empty(torch.int64), z, z, z, pair_ptr_h)
    s_start = np.concatenate([[0], np.cumsum(s_cnt)])[:-1]
    a_start = np.concatenate([[0], np.cumsum(a_cnt)])[:-1]
    pg = np.repeat(np.arange(B, dtype=np.int64), ppg)
    r = np.arange(P, dtype=np.int64) - pair_ptr_h[pg]
    ac = a_cnt[pg]
    seed_job = s_start[pg] + r // ac
    ans_slot = a_start[pg] + r % ac
    pan = ans_u[ans_slot]                                   # batch-global answer node of the slot
    job_graph = np.concatenate([sg, ag]).astype(np.int32)   # jobs: all seeds, then all answers
    J = S + A
    sizes = np.diff(node_ptr_h)[job_graph]
    dist_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cap = int((ppg * e_cnt).sum())                          # bound on the pair edge ids: every slot could hold all its graph's edges
    d_jg, d_sp, d_si, d_do, d_pg, d_ps, d_pa, d_pan = _upload_packed(dev, [
        job_graph, np.arange(J + 1, dtype=np.int64), np.concatenate([seeds_u, ans_u]).astype(np.int64), dist_off[:-1].copy(),
        pg.astype(np.int32), seed_job.astype(np.int32), (S + ans_slot).astype(np.int32), pan.astype(np.int64)])
    if csr is None:
        csr = ops.graph_csr(ei, nptr, eptr, num_nodes=N)
    lib = _lib.load()
    st = ops._stream(dev)
    dist = ops._workspace(dev, "label_dist", 4 * max(int(dist_off[-1]), 1)).view(torch.int32)

    def bfs(j0, j1, mode):
        if j1 > j0:  # sub-ranges address the same tables: the offsets inside src_ptr / dist_off are absolute
            _lib.check(lib.evi_bfs_levels_edges(d_jg[j0:].data_ptr(), d_sp[j0:].data_ptr(), d_si.data_ptr(), d_do[j0:].data_ptr(), j1 - j0,
                                                ops._ptr(nptr), ops._ptr(eptr), ops._ptr(ei), E, csr.in_ptr.data_ptr(), csr.in_nbr.data_ptr(),
                                                csr.out_ptr.data_ptr(), csr.out_nbr.data_ptr(), int(mode), dist.data_ptr(), st))

    if directed:
        bfs(0, S, 1)   # forward from the seeds
        bfs(S, J, 2)   # backward from the answers
    else:
        bfs(0, J, 0)
    plen = torch.empty(P, dtype=torch.int32, device=dev)
    pcnt = torch.empty(P, dtype=torch.int32, device=dev)
    args = (d_pg.data_ptr(), d_ps.data_ptr(), d_pa.data_ptr(), d_pan.data_ptr(), P, d_do.data_ptr(), dist.data_ptr(),
            ops._ptr(ei), E, ops._ptr(nptr), ops._ptr(eptr), int(bool(directed)))
    _lib.check(lib.evi_shortest_path_pairs(0, *args, plen.data_ptr(), pcnt.data_ptr(), mask.data_ptr(), None, None, st))
    poff = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    torch.cumsum(pcnt, 0, dtype=torch.int64, out=poff[1:])
    if cap > _PAIR_IDS_BOUND_LIMIT:
        # many pairs on large graphs: the no-read-back bound (every slot holding all its graph's edges) would be gigabytes —
        # read the exact total back instead (one synchronisation) and allocate that
        cap = int(poff[-1].item())
    pids = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
    _lib.check(lib.evi_shortest_path_pairs(1, *args, plen.data_ptr(), pcnt.data_ptr(), mask.data_ptr(), poff.data_ptr(),
                                           pids.data_ptr(), st))
    return FlatPairLabels(B, E, edge_ptr_h, mask, plen, pcnt, poff, pids, pg, seeds_u[seed_job] - node_ptr_h[pg],
                          pan - node_ptr_h[pg], pair_ptr_h)


def shortest_path_union_mask_by_pair_batch(gb: GraphBatch, sources: Sequence[Sequence[int]],
                                           targets: Sequence[Sequence[int]], *, directed: bool = False):
    """Per graph: (mask bool over the ORIGINAL edge list, pair_start, pair_answer, pair_edge_ids,
    pair_edge_counts, pair_lengths) — the reference's 6-tuple.  sources / targets: LOCAL node ids per graph.
    A thin wrapper: the work is `label_pairs_flat`; the tuples are built from its flat result."""
    def flat(lists):
        cnt = np.asarray([len(x) for x in lists], np.int64)
        ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        loc = np.asarray([int(v) for x in lists for v in x], np.int64)
        g = np.repeat(np.arange(gb.B, dtype=np.int64), cnt)
        n = np.diff(gb.node_ptr_h)[g]
        # a local id outside [0, n) must not alias a neighbour graph's node once offset: mark it out of range (-1)
        glob = np.where((loc >= 0) & (loc < n), loc + gb.node_ptr_h[g], -1)
        return ptr, glob

    sp, si = flat(sources)
    tp, ti = flat(targets)
    if gb.B == 0:
        return []
    res = label_pairs_flat(gb.edge_index, gb.node_ptr, gb.edge_ptr, sp, si, tp, ti, directed=directed, csr=gb.csr,
                           node_ptr_host=gb.node_ptr_h, edge_ptr_host=gb.edge_ptr_h)
    return res.per_graph(gb.valid_ids, gb.orig_counts)


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
        edges = (edges_h[g, :L] if gb.valid_ids is None else gb.valid_ids[g][edges_h[g, :L]]).tolist()  # back to positions in the caller's edge list
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


__all__ = ["GraphBatch", "FlatPairLabels", "label_pairs_flat", "canonicalize_positive_edges", "bfs_dist", "bfs_dist_batch", "shortest_path_union_mask_by_pair",
           "shortest_path_single", "shortest_path_single_batch", "has_connectivity",
           "shortest_path_union_mask_by_pair_batch", "node_softmax_logit", "select_topk_edges", "select_start_edges",
           "seed_onehop_stats"]
