"""GPU: the evaluation entry point end to end — a reference-shaped config tree, `experiment=eval_retriever dataset=...
ckpt.retriever=...` on the command line, packed splits + embedding tables + a Lightning-style checkpoint on disk."""
import json

import numpy as np
import pytest
import torch

from tests.config_tree import write_tree

pytestmark = pytest.mark.gpu


def test_eval_entry_point_runs_every_variant_and_split(dev, tmp_path, monkeypatch):
    from evi_rag_amd import eval as ev
    from evi_rag_amd import packed_dataset as pd, synthetic
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore
    from evi_rag_amd.eval_loop import RetrieverEvaluator
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.retriever import Retriever

    data_dir = tmp_path / "data"
    monkeypatch.setenv("EVI_TEST_PROJECT_ROOT", str(tmp_path / "proj"))
    cfg_dir = write_tree(tmp_path, data_dir)
    rng = np.random.default_rng(5)
    ent = torch.from_numpy(rng.standard_normal((2000, 16)).astype(np.float32))
    rel = torch.from_numpy(rng.standard_normal((9, 16)).astype(np.float32))
    sizes = {}
    for variant, seed in (("toyqa", 1), ("toyqa-sub", 2)):
        emb = data_dir / variant / "materialized" / "embeddings"
        emb.mkdir(parents=True)
        torch.save(ent, emb / "entity_embeddings.pt")
        torch.save(rel, emb / "relation_embeddings.pt")
        for split, graphs in (("validation", 6), ("test", 9)):
            base = synthetic.make_batch(graphs, nodes_per_graph=40, edges_per_graph=150, emb_dim=16, num_relations=9, seed=seed * 10 + graphs)
            assert int(base.node_embedding_ids.max()) < 2000
            pd.write_packed(emb / f"{split}.packed", pd.samples_from_flat_batch(base))
            sizes[(variant, split)] = graphs
    torch.manual_seed(3)
    trained = Retriever(emb_dim=16, hidden_dim=16)
    torch.save({"state_dict": {f"model.{k}": v for k, v in trained.state_dict().items()}, "global_step": 1}, tmp_path / "retriever.ckpt")

    results = ev.run(cfg_dir, ["experiment=eval_retriever", "dataset=toyqa", f"ckpt.retriever={tmp_path / 'retriever.ckpt'}"], device=str(dev))
    assert [(v, s) for v, s, _ in results] == [("toyqa", "validation"), ("toyqa", "test"), ("toyqa-sub", "validation"), ("toyqa-sub", "test")]
    out_dir = tmp_path / "proj" / "logs" / "eval_retriever_toyqa" / "runs" / "fixed"
    for variant, split, metrics in results:
        saved = json.loads((out_dir / f"metrics_{variant}_{split}.json").read_text())  # src/eval.py:381-393 naming
        assert saved == metrics and "test/edge/recall@5" in metrics and "test/answer/reachability@20" in metrics and "test/loss" in metrics
        artifact = data_dir / "toyqa" / "artifacts" / variant / "eval_retriever" / f"{split}.pt"
        assert artifact.exists(), artifact
        payload = torch.load(artifact, weights_only=False)
        assert payload["settings"]["split"] == split and len(payload["samples"]) == sizes[(variant, split)]

    # the same numbers as the evaluator driven by hand with the same weights, data and settings
    store = GlobalEmbeddingStore(data_dir / "toyqa" / "materialized" / "embeddings", device=dev)
    ds = pd.PackedRetrievalDataset(data_dir / "toyqa" / "materialized" / "embeddings" / "test.packed", device=dev, embeddings=store)
    model = Retriever(emb_dim=16, hidden_dim=16).to(dev).eval()
    model.load_state_dict(trained.state_dict())
    ref = RetrieverEvaluator(model, loss=RetrieverLoss(infonce_temperature=0.07), k_values=[1, 5, 20], bridge_metrics=True).run(pd.PackedLoader(ds, batch_size=4))
    got = dict(results[1][2])
    assert set(got) == set(ref["metrics"])
    for k, v in ref["metrics"].items():
        assert abs(got[k] - v) < 1e-6, k

    # a single split of a single dataset: no variants, no split loop -> metrics.json
    one = ev.run(cfg_dir, ["experiment=eval_retriever", "dataset=toyqa-sub", f"ckpt.retriever={tmp_path / 'retriever.ckpt'}",
                           "run.dataset_variants=null", "run.require_dual_datasets=false", "run.run_all_splits=false", "run.split=validation"],
                 device=str(dev))
    assert [(v, s) for v, s, _ in one] == [(None, "validation")]
    assert json.loads((tmp_path / "proj" / "logs" / "eval_retriever_toyqa-sub" / "runs" / "fixed" / "metrics.json").read_text()) == one[0][2]
    assert one[0][2] == results[2][2]
    with pytest.raises(ValueError, match="requires `retriever` checkpoint"):
        ev.run(cfg_dir, ["experiment=eval_retriever", "dataset=toyqa"], device=str(dev))
    with pytest.raises(ValueError, match="trainer.devices"):
        ev.run(cfg_dir, ["experiment=eval_retriever", "dataset=toyqa", f"ckpt.retriever={tmp_path / 'retriever.ckpt'}", "trainer.devices=2"], device=str(dev))
