"""GPU parity of the fused ranking metrics (T1, T2, T4, T5 margin, T3 top-k lists)."""
import os
import types

import numpy as np
import pytest
import torch

from evi_rag_amd import synthetic
from oracle import metrics as omet
from oracle.ranking import segment_topk as oracle_segment_topk

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K_VALUES = [1, 10, 25, 50, 100, 200, 300, 400, 500]


def _batch_from(z, dev, prefix="b_"):
    b = types.SimpleNamespace()
    for k in z.files:
        if k.startswith(prefix):
            setattr(b, k[len(prefix):], torch.from_numpy(z[k]).to(dev))
    b.num_graphs = int(b.ptr.numel() - 1)
    b.num_nodes = int(b.ptr[-1].item())
    b._slice_dict = {"edge_index": b.edge_ptr, "q_local_indices": b.q_ptr, "a_local_indices": b.a_ptr}
    b.answer_entity_ids_ptr = b.answer_ptr
    return b


@pytest.mark.parametrize("tag", ["toy", "mid"])
def test_metric_classes_match_reference_golden(dev, tag):
    """Values computed by the reference's own metric classes on the committed scores."""
    from evi_rag_amd import metrics as M

    z = np.load(os.path.join(GOLD, f"metrics_{tag}.npz"), allow_pickle=False)
    ref = dict(zip(z["keys"].tolist(), z["values"].tolist()))
    b = _batch_from(z, dev)
    scores = torch.from_numpy(z["scores"]).to(dev)
    target = b.labels > 0.5
    coll = M.RetrieverMetricCollection(K_VALUES)
    # two half batches through update() exercise accumulation; graphs are split at a graph boundary
    coll.update(preds=scores, target=target, indexes=None, batch=b, num_graphs=b.num_graphs)
    got = {k: float(v) for k, v in coll.compute().items()}
    for k, v in got.items():
        assert v == pytest.approx(ref[k], abs=2e-6), k
    assert float(coll.metrics["reachability"]._states["total"]) == float(z["reach_total"])
    # standalone classes (no shared pass) give the same numbers; reset works
    er = M.EdgeRecallAtK(k_values=K_VALUES)
    er.update(preds=scores, target=target, indexes=None, batch=b, num_graphs=b.num_graphs)
    for k, v in er.compute().items():
        assert float(v) == pytest.approx(ref[k], abs=2e-6)
    er.reset()
    assert all(float(v) == 0.0 for v in er.compute().values())
    ar = M.AnswerReachability(k_values=K_VALUES)
    ar.update(preds=scores, batch=b, query_ids=None, num_graphs=b.num_graphs)
    for k, v in ar.compute().items():
        assert float(v) == pytest.approx(ref[k], abs=2e-6)


@pytest.mark.parametrize("shape", [(6, 200, 600, 0), (3, 3000, 12000, 1), (2, 14000, 9000, 2)])
def test_fused_metrics_match_oracle(dev, shape):
    """Larger graphs: radix-select path (E_g > 8192) and union-find in the global workspace
    (N_g > 12288).  Integer outputs bit-exact, float outputs to f32 rounding."""
    from evi_rag_amd import metrics as M

    B, n, e, seed = shape
    sb = synthetic.make_batch(B, nodes_per_graph=n, edges_per_graph=e, emb_dim=4, seed=seed, attach_embeddings=False,
                              max_answers=40)
    rng = np.random.default_rng(seed)
    scores = rng.standard_normal(sb.num_edges).astype(np.float32)
    scores[: sb.num_edges // 3] = np.round(scores[: sb.num_edges // 3], 1)  # many exact ties
    target = sb.labels > 0.5
    ns = synthetic.as_namespace(sb, device=dev)
    ns.answer_entity_ids_ptr = torch.from_numpy(sb.answer_ptr).to(dev)
    rb = M.rank_batch(torch.from_numpy(scores).to(dev), torch.from_numpy(target).to(dev), ns, K_VALUES, want_topk=True)
    # T3: the ranked lists
    ridx, rval, rcnt = oracle_segment_topk(scores, sb.edge_ptr, K_VALUES[-1])
    assert np.array_equal(rb.topk_index.cpu().numpy(), ridx)
    assert np.array_equal(rb.topk_score.cpu().numpy(), rval)
    assert np.array_equal(rb.topk_count.cpu().numpy(), rcnt)
    # T1
    sums, cnt = omet.edge_recall_at_k(scores, target, sb.edge_ptr, K_VALUES)
    got = (rb.edge_recall.double() * rb.recall_valid.unsqueeze(1)).sum(0).cpu().numpy()
    np.testing.assert_allclose(got, [sums[k] for k in K_VALUES], rtol=0, atol=1e-6)
    assert float(rb.recall_valid.sum().item()) == cnt
    # T2 (bit-exact)
    hits, valid = omet.answer_reachability(scores, sb, K_VALUES)
    assert float(rb.reach_valid.sum().item()) == valid
    got_hits = (rb.reach.long() * rb.reach_valid.long().unsqueeze(1)).sum(0).cpu().tolist()
    assert got_hits == [int(hits[k]) for k in K_VALUES]
    # T4
    h, r = omet.answer_hit_recall_batch(scores, sb, K_VALUES)
    nvalid = float(rb.answer_valid.sum().item())
    got_h = (rb.answer_hit.double().sum(0) / nvalid).cpu().tolist()
    got_r = (rb.answer_recall.double().sum(0) / nvalid).cpu().tolist()
    assert got_h == pytest.approx([h[f"answer_hit@{k}"] for k in K_VALUES], abs=1e-12)
    assert got_r == pytest.approx([r[f"answer_recall@{k}"] for k in K_VALUES], abs=1e-6)
    # T5 margin
    ref_m = omet.score_margin(scores, target, sb.edge_ptr)["edge/score_margin"]
    mv = rb.margin_valid.bool()
    got_m = float((rb.score_margin.double() * mv).sum().item()) / max(float(mv.sum().item()), 1.0)
    assert got_m == pytest.approx(ref_m, abs=1e-6)


def test_metrics_edge_cases(dev):
    """Graphs without edges / seeds / answers are skipped exactly as the reference skips them."""
    from evi_rag_amd import metrics as M

    # graph 0: no edges; graph 1: seed == answer (reachable with zero edges); graph 2: no answers
    ei = torch.tensor([[3, 4, 6], [4, 5, 7]], device=dev)
    b = types.SimpleNamespace(
        edge_index=ei, ptr=torch.tensor([0, 3, 6, 9], device=dev), edge_ptr=torch.tensor([0, 0, 2, 3], device=dev),
        q_local_indices=torch.tensor([0, 3, 6], device=dev), q_local_indices_ptr=torch.tensor([0, 1, 2, 3], device=dev),
        a_local_indices=torch.tensor([1, 3], device=dev), a_local_indices_ptr=torch.tensor([0, 1, 2, 2], device=dev),
        node_global_ids=torch.arange(100, 109, device=dev), answer_entity_ids=torch.tensor([101, 103, 103], device=dev),
        answer_entity_ids_ptr=torch.tensor([0, 1, 3, 3], device=dev))
    scores = torch.tensor([0.5, 0.7, 0.1], device=dev)
    target = torch.tensor([True, False, False], device=dev)
    rb = M.rank_batch(scores, target, b, [1, 2, 5], want_topk=True)
    assert rb.recall_valid.cpu().tolist() == [0, 1, 1]
    assert rb.reach_valid.cpu().tolist() == [0, 1, 0]
    assert rb.reach.cpu().tolist()[1] == [1, 1, 1]
    assert rb.topk_index.cpu().tolist()[1][:2] == [1, 0] and rb.topk_count.cpu().tolist() == [0, 2, 1]
    # graph 1: top-1 edge is (4 -> 5): no answer entity; top-2 adds (3 -> 4): head 103 is the answer
    assert rb.answer_valid.cpu().tolist() == [1, 1, 0]
    assert rb.answer_hit.cpu().tolist()[1] == [0, 1, 1]
    assert rb.answer_recall.cpu().tolist()[1] == [0.0, 1.0, 1.0]  # duplicated answer id counts once
    assert rb.edge_recall.cpu().tolist()[1] == [0.0, 1.0, 1.0]
    assert rb.margin_valid.cpu().tolist() == [0, 1, 0]
    assert rb.score_margin.cpu().tolist()[1] == pytest.approx(0.5 - 0.7)
    with pytest.raises(ValueError, match="strictly ascending|positive"):
        from evi_rag_amd import _lib
        import ctypes
        _lib.check(_lib.load().evi_retriever_metrics(*([None] * 3), 0, None, None, 1, *([None] * 7),
                                                     (ctypes.c_int32 * 2)(5, 5), 2, *([None] * 14)))


@pytest.mark.parametrize("tag", ["toy", "mid"])
def test_bridge_metrics_match_reference_golden(dev, tag):
    """T5: BridgeEdgeRecallAtK / BridgePositiveCoverage / BridgeProbQuality values from the reference."""
    from evi_rag_amd import metrics as M

    z = np.load(os.path.join(GOLD, f"metrics_{tag}.npz"), allow_pickle=False)
    ref = dict(zip(z["keys"].tolist(), z["values"].tolist()))
    b = _batch_from(z, dev)
    scores = torch.from_numpy(z["scores"]).to(dev)
    target = b.labels > 0.5
    coll = M.RetrieverMetricCollection(K_VALUES, bridge_metrics=True)
    coll.update(preds=scores, target=target, indexes=None, batch=b, num_graphs=b.num_graphs)
    got = {k: float(v) for k, v in coll.compute().items()}
    bridge_keys = [k for k in ref if k.startswith("bridge/")]
    assert len(bridge_keys) == len(K_VALUES) + 5
    for k in bridge_keys:
        assert got[k] == pytest.approx(ref[k], abs=2e-6), k
    assert set(got) == set(ref)


def test_feature_monitor_matches_reference_golden(dev):
    from evi_rag_amd.metrics import FeatureMonitor

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feature_monitor.npz"), allow_pickle=False)
    m = FeatureMonitor()
    for b in range(int(z["num_batches"])):
        m.update(torch.from_numpy(z[f"b{b}_preds"]).to(dev), torch.from_numpy(z[f"b{b}_target"]).to(dev),
                 torch.from_numpy(z[f"b{b}_features"]).to(dev))
    out = m.compute()
    assert sorted(out) == z["keys"].tolist()
    np.testing.assert_allclose([float(out[k]) for k in sorted(out)], z["values"], rtol=2e-6, atol=2e-7)
    m.reset()
    m.update(torch.from_numpy(z["b1_preds"]).to(dev), torch.zeros(37, dtype=torch.bool, device=dev))
    out = m.compute()
    np.testing.assert_allclose([float(out[k]) for k in sorted(out)], z["nopos_values"], rtol=2e-6, atol=2e-7)


def test_compute_ranking_metrics_matches_the_reference_function_and_the_oracle():
    """src/utils/metrics.py:112-170 (compute_ranking_metrics, _ndcg): golden = the reference's own function on seeded samples
    (binary and graded labels, all-positive / all-negative / one-element samples, k beyond the sample length)."""
    from evi_rag_amd import metrics as M

    z = np.load(os.path.join(GOLD, "ranking_metrics.npz"), allow_pickle=False)
    ptr, ks = z["ptr"], [int(k) for k in z["k_values"]]
    dev = torch.device("cuda:0")
    samples = [{"scores": torch.from_numpy(z["scores"][a:b]).to(dev), "labels": torch.from_numpy(z["labels"][a:b]).to(dev)}
               for a, b in zip(ptr[:-1], ptr[1:])]
    st = M.compute_ranking_metrics(samples, ks)
    for name, got in (("precision", st.precision_at_k), ("recall", st.recall_at_k), ("f1", st.f1_at_k), ("ndcg", st.ndcg_at_k)):
        assert [got[k] for k in ks] == pytest.approx(z[name].tolist(), abs=1e-6), name  # f32 sums in the reference, f64 here
    assert st.mrr == pytest.approx(float(z["mrr"]), abs=1e-12)
    assert M.compute_ranking_metrics(samples, None).precision_at_k == {1: pytest.approx(float(z["default_k_precision"]), abs=1e-12)}
    none = M.compute_ranking_metrics([samples[1]], ks)  # all-negative sample alone
    assert none.mrr == 0.0 and all(v == 0.0 for v in none.recall_at_k.values())
    assert M.compute_ranking_metrics([], ks).mrr == 0.0
    # host tensors are accepted (moved to the GPU), and a larger random case agrees with the oracle
    rng = np.random.default_rng(5)
    big = []
    for n in rng.integers(1, 3000, 64):
        sc = rng.permutation(int(n)).astype(np.float32) - 17.0
        big.append((sc, (rng.random(int(n)) < 0.05).astype(np.float32)))
    want = omet.ranking_metrics(big, [1, 10, 100, 500])
    got = M.compute_ranking_metrics([{"scores": torch.from_numpy(a), "labels": torch.from_numpy(b)} for a, b in big], [1, 10, 100, 500])
    for name, g in (("precision", got.precision_at_k), ("recall", got.recall_at_k), ("f1", got.f1_at_k), ("ndcg", got.ndcg_at_k)):
        assert [g[k] for k in (1, 10, 100, 500)] == pytest.approx([want[name][k] for k in (1, 10, 100, 500)], abs=1e-6), name
    assert got.mrr == pytest.approx(want["mrr"], abs=1e-12)
    with pytest.raises(ValueError, match="length mismatch"):
        M.compute_ranking_metrics([{"scores": torch.zeros(3), "labels": torch.zeros(2)}], ks)


def test_metric_collection_over_more_graphs_than_a_wave(dev):
    """k_metric_accumulate adds one batch's per-graph results into the epoch states with one wave per state (lane l takes graphs
    l, l + 64, ...): a batch of 150 graphs crosses the wave width; the collection's means must equal the means of rank_batch's
    per-graph outputs."""
    from evi_rag_amd import metrics as M

    sb = synthetic.make_batch(150, nodes_per_graph=40, edges_per_graph=90, emb_dim=4, seed=31, attach_embeddings=False, max_answers=6)
    ns = synthetic.as_namespace(sb, device=dev)
    ns.answer_entity_ids_ptr = torch.from_numpy(sb.answer_ptr).to(dev)
    torch.manual_seed(4)
    scores = torch.randn(sb.num_edges, device=dev) + 1.5 * ns.labels
    target = ns.labels > 0.5
    ks = [1, 5, 20]
    coll = M.RetrieverMetricCollection(ks)
    for _ in range(2):  # two updates: the states accumulate
        coll.update(preds=scores, target=target, indexes=None, batch=ns, num_graphs=150)
    got = {k: float(v) for k, v in coll.compute().items()}
    rb = M.rank_batch(scores, target, ns, ks, num_graphs=150)
    rv, hv, av = rb.recall_valid.bool(), rb.reach_valid.bool(), rb.answer_valid == 1
    for j, k in enumerate(ks):
        assert got[f"edge/recall@{k}"] == pytest.approx(float(rb.edge_recall[rv, j].double().mean()), abs=1e-6)
        assert got[f"answer/reachability@{k}"] == pytest.approx(float(rb.reach[hv, j].double().mean()), abs=1e-6)
        assert got[f"answer_hit@{k}"] == pytest.approx(float(rb.answer_hit[av, j].double().mean()), abs=1e-6)
