#!/usr/bin/env python3
"""BASELINE config 3 (CWQ-shaped 2-hop expansion) measured on one MI355X: CSR build, DDE structure
features, multi-source BFS levels, 2-hop frontier from the seeds, seed-incident edge selection, on
batches of CWQ-shaped graphs (N_g ~ 3 000, E_g ~ 10 000, DDE 2 + 2 rounds, ratio 0.25), with the
algorithmic bytes of each kernel turned into GB/s, beside the CPU oracle on a sample.

usage: python tools/bench_graph_kernels.py [--graphs 3531] [--batch 64] [--out profiles/rNN_config3_graph_kernels.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from evi_rag_amd import _lib, ops, synthetic


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=3531)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=3000)
    ap.add_argument("--edges", type=int, default=10000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-graphs", type=int, default=8)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B = args.batch
    sb = synthetic.make_batch(B, nodes_per_graph=args.nodes, edges_per_graph=args.edges, emb_dim=8, num_relations=512, seed=2,
                              attach_embeddings=False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ei, ptr, eptr, topic = t(sb.edge_index), t(sb.ptr), t(sb.edge_ptr), t(sb.topic_one_hot)
    N, E = sb.num_nodes, sb.num_edges
    scores = torch.randn(E, device=dev)
    seeds = t(sb.q_local_indices)
    res = {"workload": f"{B} CWQ-shaped graphs per batch, N={N}, E={E} (N_g~{args.nodes}, E_g~{args.edges}), "
                       f"{args.graphs} graphs per epoch", "kernels": {}}

    def rec(name, ms, nbytes, note):
        res["kernels"][name] = {"ms_per_batch": ms, "algorithmic_bytes": nbytes, "GB_per_s": nbytes / (ms * 1e-3) / 1e9,
                                "graphs_per_s": B / (ms * 1e-3), "note": note}

    csr = ops.graph_csr(ei, ptr, eptr)
    rec("evi_graph_csr", timed(lambda: ops.graph_csr(ei, ptr, eptr), args.iters), E * 16 + 2 * (E * 8 + N * 4),
        "edge_index read (16 B/edge) + both CSR halves written (nbr + eid per edge, ptr per node)")
    rounds = 2
    rec("evi_dde_node_struct", timed(lambda: ops.dde_node_struct(topic, ptr, csr, rounds, rounds), args.iters),
        2 * rounds * (E * 12 + N * 16) + N * 10 * 4, "per round E*(4 nbr + 8 gathered) + N*(8 ptr + 8 out); 2 + 2 rounds")
    # multi-source BFS from the seeds (one job per graph) and the 2-hop frontier
    jg = torch.arange(B, dtype=torch.int32, device=dev)
    sp, doff = t(sb.q_ptr), t(sb.ptr[:-1])
    dist = torch.empty(N, dtype=torch.int32, device=dev)

    def bfs():
        _lib.check(lib.evi_bfs_levels(jg.data_ptr(), sp.data_ptr(), seeds.data_ptr(), doff.data_ptr(), B, ptr.data_ptr(),
                                      csr.in_ptr.data_ptr(), csr.in_nbr.data_ptr(), csr.out_ptr.data_ptr(),
                                      csr.out_nbr.data_ptr(), 0, dist.data_ptr(), ops._stream(dev)))

    ms = timed(bfs, args.iters)
    levels = int(dist.max().item()) + 1
    rec("evi_bfs_levels", ms, levels * N * 4 + 2 * E * 8, f"undirected, {levels} levels: N*4 scanned per level + every CSR row once (nbr + dist probe)")
    two_hop = int(((dist >= 0) & (dist <= 2)).sum().item())
    res["two_hop_frontier_nodes_per_graph"] = two_hop / B
    mask = torch.empty(E, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    def expand():
        _lib.check(lib.evi_select_start_edges(scores.data_ptr(), E, seeds.data_ptr(), seeds.numel(), csr.in_ptr.data_ptr(),
                                              csr.in_eid.data_ptr(), csr.out_ptr.data_ptr(), csr.out_eid.data_ptr(), N, 0.25, 1,
                                              -1, mask.data_ptr(), status.data_ptr(), ops._stream(dev)))

    deg = (csr.in_ptr[seeds + 1] - csr.in_ptr[seeds] + csr.out_ptr[seeds + 1] - csr.out_ptr[seeds]).sum().item()
    rec("evi_select_start_edges", timed(expand, args.iters), int(deg) * 8 + E, "incident (eid, score) of every seed + the E-byte mask")
    total_ms = sum(k["ms_per_batch"] for k in res["kernels"].values())
    res["gpu_graphs_per_s"] = B / (total_ms * 1e-3)
    res["gpu_epoch_seconds"] = args.graphs / res["gpu_graphs_per_s"]

    # CPU oracle on a sample (the reference's own Python / numpy algorithms, restated)
    from oracle import graph as og

    g = min(args.cpu_graphs, B)
    t0 = time.perf_counter()
    for i in range(g):
        n0, n1, e0, e1 = int(sb.ptr[i]), int(sb.ptr[i + 1]), int(sb.edge_ptr[i]), int(sb.edge_ptr[i + 1])
        src, dst = sb.edge_index[0, e0:e1] - n0, sb.edge_index[1, e0:e1] - n0
        adj = og.build_undirected_adjacency(n1 - n0, src.tolist(), dst.tolist())
        q = (sb.q_local_indices[int(sb.q_ptr[i]): int(sb.q_ptr[i + 1])] - n0).tolist()
        og.bfs_dist(n1 - n0, adj, q)
        og.node_structure_features(sb.topic_one_hot[n0:n1], np.stack([src, dst]), rounds, rounds)
        og.select_start_edges(src, dst, np.zeros(e1 - e0, np.float32), np.asarray(q), n1 - n0, 0.25, 1, None)
    cpu_s = (time.perf_counter() - t0) / g
    res["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "graphs/s", "cores": 1, "kind": "port",
                           "sample": f"oracle adjacency + BFS + DDE + seed expansion on {g} of the graphs, {cpu_s * 1e3:.1f} ms/graph"}
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
