"""Randomised parity sweep: many small random cases per kernel family against the oracle.

By default a handful of cases per family (seconds).  EVI_FUZZ_CASES=N runs N cases per family — used for
longer soak runs on the GPU box (shapes, seeds, tie densities and degenerate inputs are all drawn at random,
so a failure prints the case's seed).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cosine as ocos
from oracle import g_agent as og
from oracle import graph as ograph
from oracle import loss as oloss
from tests.helpers import check_topk_against_scores

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("EVI_FUZZ_CASES", "6"))
EPS = 1e-6


def test_fuzz_cosine_topk_paths_agree(dev):
    """scan vs oracle (margin-aware), scan vs many-query GEMM path and vs the two-stage scan (bit-exact), shards vs single pass (bit-exact)."""
    from evi_rag_amd import ops

    for case in range(CASES):
        rng = np.random.default_rng(1000 + case)
        D = int(rng.choice([16, 32, 64, 128, 384, 768]))
        N = int(rng.integers(1, 60000))
        Q = int(rng.integers(1, 70))
        k = int(rng.choice([1, 5, 50, 500, 1000]))
        x = rng.standard_normal((N, D), dtype=np.float32)
        if N > 4:
            x[rng.integers(0, N, N // 20 + 1)] = x[rng.integers(0, N, N // 20 + 1)]  # ties
            x[int(rng.integers(0, N))] = 0.0
        q = rng.standard_normal((Q, D), dtype=np.float32)
        xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
        qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
        s, i = ops.cosine_topk(qn, xn, k, row_id_base=5, method="scan")
        check_topk_against_scores(s.cpu().numpy(), i.cpu().numpy(), ocos.cosine_scores(q, x, EPS), k, id_base=5)
        if k <= 1000 and D % 16 == 0:
            s2, i2 = ops.cosine_topk_gemm(qn, xn, k, row_id_base=5)  # falls back by itself when it cannot prove exactness
            assert torch.equal(i2, i) and torch.equal(s2, s), f"case {case}: gemm path differs"
        if k <= 1000 and D % 32 == 0:
            # two-stage exact scan (f16 shadow selects, f32 rows re-score); falls back by itself when it cannot prove exactness
            s3, i3 = ops.cosine_topk_two_stage(qn, xn, ops.index_shadow_f16(xn), k, row_id_base=5)
            assert torch.equal(i3, i) and torch.equal(s3, s), f"case {case}: two-stage path differs"
        if N >= 3:
            cuts = sorted(set([0, N] + rng.integers(1, N, 2).tolist()))
            parts = [ops.cosine_topk(qn, xn[a:b], k, row_id_base=5 + a) for a, b in zip(cuts[:-1], cuts[1:])]
            ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
            assert torch.equal(mi, i) and torch.equal(ms, s), f"case {case}: shard merge differs"


def test_fuzz_graph_labelling(dev):
    from evi_rag_amd import labelling as L

    for case in range(CASES):
        rng = np.random.default_rng(2000 + case)
        B = int(rng.integers(1, 6))
        ns, srcs, dsts, seeds, answers = [], [], [], [], []
        for _ in range(B):
            n = int(rng.integers(1, 400))
            e = int(rng.integers(0, 5 * n))
            ns.append(n)
            srcs.append(rng.integers(-1, n + 1, e))  # a few out-of-range endpoints: skipped like the reference
            dsts.append(rng.integers(0, n, e))
            seeds.append(rng.integers(0, n, int(rng.integers(0, 4))).tolist())
            answers.append(rng.integers(0, n, int(rng.integers(0, 4))).tolist())
        gb = L.GraphBatch(ns, srcs, dsts)
        for directed in (False, True):
            res = L.shortest_path_union_mask_by_pair_batch(gb, seeds, answers, directed=directed)
            for g in range(B):
                ref = ograph.shortest_path_union_mask_by_pair(ns[g], srcs[g].tolist(), dsts[g].tolist(), seeds[g], answers[g],
                                                              directed=directed)
                assert res[g][0].tolist() == ref[0] and tuple(res[g][1:]) == tuple(ref[1:]), (case, g, directed)
        single = L.shortest_path_single_batch(gb, seeds, answers)
        for g in range(B):
            if not seeds[g] or not answers[g]:
                continue
            ref = ograph.shortest_path_single(ns[g], srcs[g].tolist(), dsts[g].tolist(), seeds[g], answers[g])
            assert single[g] == (list(ref[0]), list(ref[1])), (case, g)


def test_fuzz_build_graph_and_g_agent(dev):
    import types

    from evi_rag_amd import graph_build
    from evi_rag_amd.g_agent import GAgentBuilder, GAgentSettings

    for case in range(CASES):
        rng = np.random.default_rng(3000 + case)
        # build_graph on coded triples
        S = int(rng.integers(1, 5))
        tri, qs, as_, subs = [], [], [], []
        for _ in range(S):
            n_ent, n_trip = int(rng.integers(2, 200)), int(rng.integers(0, 1500))
            t = np.stack([rng.integers(0, n_ent, n_trip), rng.integers(0, 9, n_trip), rng.integers(0, n_ent, n_trip)], 1).astype(np.int64)
            tri.append(t)
            qs.append(rng.integers(0, n_ent, 2).tolist())
            as_.append(rng.integers(0, n_ent, 3).tolist())
            subs.append(t[rng.integers(0, max(n_trip, 1), min(n_trip, 10))] if n_trip else np.empty((0, 3), np.int64))
        dedup, noloop = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        coded = graph_build.index_graphs_coded(tri, qs, as_, subs, dedup_edges=dedup, remove_self_loops=noloop)
        labels = graph_build.label_graphs(coded)
        for s in range(S):
            ref = ograph.build_graph_ids(tri[s], qs[s], as_[s], subs[s], np.arange(200), np.arange(200), dedup_edges=dedup,
                                         remove_self_loops=noloop)
            assert np.array_equal(coded[s].edge_src, ref["edge_src"]) and np.array_equal(coded[s].edge_dst, ref["edge_dst"]), (case, s)
            assert np.array_equal(coded[s].node_codes, ref["node_entity_ids"]), (case, s)
            assert np.array_equal(np.asarray(labels[s][0], bool), np.asarray(ref["positive"], bool)), (case, s)
            assert tuple(labels[s][1:]) == (ref["pair_start"], ref["pair_answer"], ref["pair_edges"], ref["pair_counts"], ref["pair_len"])
        # g_agent builder on a random batch with duplicated triples
        B = int(rng.integers(1, 5))
        ptr, eptr, ei, rel, gids, store = [0], [0], [], [], [], {}
        for g in range(B):
            n, e = int(rng.integers(3, 80)), int(rng.integers(1, 400))
            h, t = rng.integers(0, n, e), rng.integers(0, n, e)
            ei.append(np.stack([h, t]) + ptr[-1])
            rel.append(rng.integers(0, 4, e))
            ids = rng.choice(10_000, n, replace=False) + 1
            gids.append(ids)
            store[f"s{g}"] = {"question_emb": [0.0] * 4, "question": "q", "seed_entity_ids": ids[rng.integers(0, n, 2)].tolist(),
                              "answer_entity_ids": ids[rng.integers(0, n, 2)].tolist() + [77_777]}
            ptr.append(ptr[-1] + n)
            eptr.append(eptr[-1] + e)
        edge_index, edge_attr, node_gids = np.concatenate(ei, 1), np.concatenate(rel), np.concatenate(gids)
        E = edge_index.shape[1]
        logits = rng.standard_normal(E).astype(np.float32)
        labels_e = (rng.random(E) < 0.2).astype(np.float32)
        t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        batch = types.SimpleNamespace(ptr=t_(np.asarray(ptr, np.int64)), edge_index=t_(edge_index), edge_attr=t_(edge_attr), labels=t_(labels_e),
                                      node_global_ids=t_(node_gids), node_embedding_ids=t_(node_gids % 97), sample_id=[f"s{g}" for g in range(B)])
        out = types.SimpleNamespace(logits=t_(logits), query_ids=t_(np.repeat(np.arange(B), np.diff(eptr))))
        top_k, ratio = int(rng.choice([3, 20, 500])), float(rng.choice([0.0, 0.25, 1.0]))
        b = GAgentBuilder(GAgentSettings(edge_top_k=top_k, start_keep_ratio=ratio, allow_empty_answer=True),
                          embedding_store=types.SimpleNamespace(load_sample=lambda sid: store[sid]))
        b.process_batch(batch, out)
        got = {s.sample_id: s for s in b.samples}
        for g in range(B):
            lo, hi, n0, n1 = eptr[g], eptr[g + 1], ptr[g], ptr[g + 1]
            ref = og.build_sample(heads=edge_index[0, lo:hi] - n0, tails=edge_index[1, lo:hi] - n0, relations=edge_attr[lo:hi],
                                  labels=labels_e[lo:hi], scores=logits[lo:hi], node_global_ids=node_gids[n0:n1],
                                  node_embedding_ids=(node_gids % 97)[n0:n1], start_entity_ids=np.asarray(store[f"s{g}"]["seed_entity_ids"]),
                                  answer_entity_ids=np.asarray(store[f"s{g}"]["answer_entity_ids"]), edge_top_k=top_k,
                                  start_keep_ratio=ratio, start_min_edges=1, start_max_edges=None, allow_empty_answer=True, node_softmax=True)
            assert (ref is None) == (f"s{g}" not in got), (case, g)
            if ref is None:
                continue
            smp = got[f"s{g}"]
            for name in ("edge_relations", "edge_head_locals", "edge_tail_locals", "node_entity_ids", "node_embedding_ids",
                         "start_node_locals", "answer_node_locals", "edge_labels"):
                assert np.array_equal(getattr(smp, name).numpy(), ref[name]), (case, g, name)
            np.testing.assert_allclose(smp.edge_scores.numpy(), ref["edge_scores"], rtol=5e-5, atol=5e-5)


def test_fuzz_retriever_loss(dev):
    import types

    from evi_rag_amd.loss import RetrieverLoss

    for case in range(CASES):
        rng = np.random.default_rng(4000 + case)
        B = int(rng.integers(1, 20))
        counts = rng.integers(0, 300, B)
        if counts.sum() == 0:
            counts[0] = 5
        eb = np.repeat(np.arange(B), counts)
        E = eb.size
        logits = (rng.standard_normal(E) * rng.choice([0.1, 1, 10])).astype(np.float32)
        targets = (rng.random(E) < rng.choice([0.0, 0.05, 0.5, 1.0])).astype(np.float32)
        near = rng.random(E) < 0.3
        cfg = dict(infonce_temperature=float(rng.choice([0.5, 1.0, 2.0])), bce_weight=float(rng.choice([0.0, 0.5])),
                   edge_weight_near=float(rng.choice([1.0, 2.0])), edge_weight_bridge=float(rng.choice([1.0, 0.5])))
        lg = torch.from_numpy(logits).to(dev).requires_grad_(True)
        out = RetrieverLoss(**cfg)(types.SimpleNamespace(logits=lg), torch.from_numpy(targets).to(dev), edge_batch=torch.from_numpy(eb).to(dev),
                                   num_graphs=B, edge_is_near=torch.from_numpy(near).to(dev))
        total, comps, mets, grad = oloss.retriever_loss(logits, targets, eb, B, edge_is_near=near, **cfg)
        assert abs(float(out.loss.detach()) - total) < 3e-5 * max(1.0, abs(total)), case
        assert sorted(out.metrics) == sorted(mets), case
        for kk, v in mets.items():
            assert abs(out.metrics[kk] - v) < 3e-5 * max(1.0, abs(v)), (case, kk)
        if out.loss.requires_grad:
            out.loss.backward()
            np.testing.assert_allclose(lg.grad.cpu().numpy(), grad, rtol=3e-4, atol=1e-7, err_msg=str(case))
