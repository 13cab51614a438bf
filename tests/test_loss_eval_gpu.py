"""GPU parity: RetrieverLoss (values, metrics, gradient) vs the reference golden / oracle; the eval loop."""
import os

import numpy as np
import pytest
import torch

from oracle import loss as oloss

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases(z):
    for variant, tkey in (("base", "targets"), ("nopos", "nopos_targets")):
        for ci, (T, wi, wb, wn, wbr) in enumerate(z["cfgs"].tolist()):
            yield f"{variant}_c{ci}", z[tkey], dict(infonce_temperature=T, infonce_weight=wi, bce_weight=wb, edge_weight_near=wn,
                                                   edge_weight_bridge=wbr)


@pytest.mark.parametrize("shuffle", [False, True])
def test_retriever_loss_matches_reference_golden(dev, shuffle):
    import types

    from evi_rag_amd.loss import RetrieverLoss

    z = np.load(os.path.join(GOLD, "loss.npz"), allow_pickle=False)
    E = z["logits"].size
    perm = np.random.default_rng(0).permutation(E) if shuffle else np.arange(E)
    for tag, targets, cfg in _cases(z):
        logits = torch.from_numpy(z["logits"][perm]).to(dev).requires_grad_(True)
        out = RetrieverLoss(**cfg)(types.SimpleNamespace(logits=logits), torch.from_numpy(targets[perm]).to(dev),
                                   edge_batch=torch.from_numpy(z["edge_batch"][perm]).to(dev), num_graphs=int(z["num_graphs"]),
                                   edge_is_near=torch.from_numpy(z["edge_is_near"][perm]).to(dev))
        ref = float(z[f"{tag}_loss"])
        assert abs(float(out.loss.detach()) - ref) < 3e-6 * max(1.0, abs(ref)), tag
        assert sorted(out.components) == z[f"{tag}_component_keys"].tolist() and sorted(out.metrics) == z[f"{tag}_metric_keys"].tolist(), tag
        np.testing.assert_allclose([out.components[k] for k in sorted(out.components)], z[f"{tag}_component_vals"], rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose([out.metrics[k] for k in sorted(out.metrics)], z[f"{tag}_metric_vals"], rtol=3e-6, atol=3e-6)
        if out.loss.requires_grad:
            (2.0 * out.loss).backward()
            np.testing.assert_allclose(logits.grad.cpu().numpy(), 2.0 * z[f"{tag}_grad"][perm], rtol=3e-5, atol=3e-7, err_msg=tag)
    with pytest.raises(ValueError, match="infonce_temperature must be positive"):
        RetrieverLoss(infonce_temperature=0.0)
    with pytest.raises(ValueError, match="requires edge_is_near"):
        RetrieverLoss(edge_weight_near=2.0)(types.SimpleNamespace(logits=logits), torch.zeros(E, device=dev),
                                            edge_batch=torch.from_numpy(z["edge_batch"]).to(dev), num_graphs=7)


def test_retriever_loss_webqsp_shape_matches_oracle(dev):
    import types

    from evi_rag_amd.loss import RetrieverLoss

    rng = np.random.default_rng(8)
    counts = rng.integers(2000, 6000, 32)
    eb = np.repeat(np.arange(32), counts)
    E = eb.size
    logits = (rng.standard_normal(E) * 4).astype(np.float32)
    targets = (rng.random(E) < 0.01).astype(np.float32)
    near = rng.random(E) < 0.3
    cfg = dict(infonce_temperature=0.7, bce_weight=0.5, edge_weight_near=1.5, edge_weight_bridge=0.75)
    lg = torch.from_numpy(logits).to(dev).requires_grad_(True)
    out = RetrieverLoss(**cfg)(types.SimpleNamespace(logits=lg), torch.from_numpy(targets).to(dev), edge_batch=torch.from_numpy(eb).to(dev),
                               num_graphs=32, edge_is_near=torch.from_numpy(near).to(dev))
    total, comps, mets, grad = oloss.retriever_loss(logits, targets, eb, 32, edge_is_near=near, **cfg)
    assert abs(float(out.loss.detach()) - total) < 2e-5 * max(1.0, abs(total))
    for k, v in mets.items():
        assert abs(out.metrics[k] - v) < 2e-5 * max(1.0, abs(v)), k
    out.loss.backward()
    np.testing.assert_allclose(lg.grad.cpu().numpy(), grad, rtol=2e-4, atol=1e-8)


def test_evaluator_epoch_matches_single_batch(dev, tmp_path):
    """The eval loop over a packed split: metrics and the epoch loss do not depend on the batching, the
    top-k writer and the g_agent builder run as callbacks, the loss equals the oracle's on the logits."""
    from evi_rag_amd import packed_dataset as pd, synthetic
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore
    from evi_rag_amd.eval_loop import RetrieverEvaluator
    from evi_rag_amd.g_agent import GAgentBuilder, GAgentSettings
    from evi_rag_amd.retriever import Retriever

    base = synthetic.make_batch(10, nodes_per_graph=80, edges_per_graph=300, emb_dim=16, num_relations=9, seed=12)
    pd.write_packed(tmp_path / "split", pd.samples_from_flat_batch(base))
    rng = np.random.default_rng(1)
    store = GlobalEmbeddingStore.from_tensors(torch.from_numpy(rng.standard_normal((int(base.node_embedding_ids.max()) + 1, 16)).astype(np.float32)),
                                              torch.from_numpy(rng.standard_normal((9, 16)).astype(np.float32)), device=dev)
    ds = pd.PackedRetrievalDataset(tmp_path / "split", device=dev, embeddings=store)
    torch.manual_seed(1)
    model = Retriever(emb_dim=16, hidden_dim=16).to(dev).eval()
    builder = GAgentBuilder(GAgentSettings(edge_top_k=30, allow_empty_answer=True), embedding_store=ds)
    ev = RetrieverEvaluator(model, k_values=[1, 5, 20], callbacks=[builder])
    res = ev.run(pd.PackedLoader(ds, batch_size=4))
    assert res["num_graphs"] == 10 and res["queries_per_sec"] > 0
    assert builder.stats["num_samples"] + builder.stats["retrieval_failed"] == 10
    one = RetrieverEvaluator(model, k_values=[1, 5, 20]).run(pd.PackedLoader(ds, batch_size=10))
    for k, v in one["metrics"].items():
        if k.endswith("/loss"):
            continue  # the epoch loss is a graph-weighted mean of per-batch means over VALID graphs: batching-dependent
        assert abs(res["metrics"][k] - v) < 1e-6, k
    # topic ablation + feature metrics: a second metric set under test/ablate_topic/, features/* keys, batch left intact
    # overlap_metrics (default): loss scalars and ranking metrics of a batch run on a side stream under the next batch's forward.
    # Same kernels on the same inputs, per-batch totals added in batch order: the epoch's numbers must be IDENTICAL to the
    # one-stream evaluation, over several passes (buffers recycled by the allocator between batches would show up here)
    for bs in (3, 5):
        serial = RetrieverEvaluator(model, k_values=[1, 5, 20], bridge_metrics=True, overlap_metrics=False).run(pd.PackedLoader(ds, batch_size=bs))
        for _ in range(3):
            over = RetrieverEvaluator(model, k_values=[1, 5, 20], bridge_metrics=True).run(pd.PackedLoader(ds, batch_size=bs))
            assert over["metrics"] == serial["metrics"] and over["num_graphs"] == serial["num_graphs"]
    ab = RetrieverEvaluator(model, k_values=[1, 5], ablate_topic=True, feature_metrics=True).run(pd.PackedLoader(ds, batch_size=5))
    keys = ab["metrics"]
    assert "test/ablate_topic/edge/recall@5" in keys and "test/features/norm_avg" in keys and keys["test/features/norm_avg"] > 0
    assert abs(keys["test/edge/recall@5"] - one["metrics"]["test/edge/recall@5"]) < 1e-6
    assert keys["test/ablate_topic/edge/score_margin"] != keys["test/edge/score_margin"]  # structure features matter
    batch = ds.collate(list(range(10)))
    assert float(batch.topic_one_hot.abs().sum()) > 0
    with torch.no_grad():
        out = model(batch)
    total, _, _, _ = oloss.retriever_loss(out.logits.cpu().numpy(), batch.labels.cpu().numpy(), out.query_ids.cpu().numpy(), 10)
    assert abs(one["metrics"]["test/loss"] - total) < 1e-5 * max(1.0, abs(total))
