"""T3: the eval_retriever artifact (payload schema + ranked triplets) against the oracle ranking."""
import json

import numpy as np
import pytest
import torch

from evi_rag_amd import synthetic
from oracle.ranking import topk_desc

pytestmark = pytest.mark.gpu


def test_topk_writer_payload_matches_oracle(dev, tmp_path):
    from evi_rag_amd.retriever import RetrieverOutput
    from evi_rag_amd.topk_writer import RetrieverTopKEdgeWriter

    sb = synthetic.make_batch(5, nodes_per_graph=40, edges_per_graph=30, emb_dim=8, seed=3, attach_embeddings=False)
    rng = np.random.default_rng(0)
    logits = rng.standard_normal(sb.num_edges).astype(np.float32) * 3
    lf, lb = logits + 0.5, logits - 0.5
    batch = synthetic.as_namespace(sb, device=dev)
    batch.answer_entity_ids_ptr = torch.from_numpy(sb.answer_ptr)
    batch.question = [f"q{g}?" for g in range(sb.num_graphs)]
    eb = torch.from_numpy(np.repeat(np.arange(sb.num_graphs), np.diff(sb.edge_ptr))).to(dev)
    out = RetrieverOutput(logits=torch.from_numpy(logits).to(dev), query_ids=eb, logits_fwd=torch.from_numpy(lf).to(dev),
                          logits_bwd=torch.from_numpy(lb).to(dev))
    w = RetrieverTopKEdgeWriter(output_dir=tmp_path / "eval_retriever", split="test", topk_values=[1, 5, 10, 50])
    w.on_test_start()
    w.on_test_batch_end(None, None, out, batch, 0)
    w.on_test_batch_end(None, None, None, batch, 1)  # test_step returned None: nothing collected (:127-129)
    w.on_test_end()
    payload = torch.load(tmp_path / "eval_retriever" / "test.pt", weights_only=False)
    manifest = json.loads((tmp_path / "eval_retriever" / "test.manifest.json").read_text())
    assert manifest["artifact"] == "eval_retriever" and manifest["schema_version"] == 1 and manifest["file"] == "test.pt"
    assert manifest["producer"] == "retriever_topk_edge_writer"
    assert payload["settings"] == {"split": "test", "topk_values": [1, 5, 10, 50]}
    assert len(payload["samples"]) == sb.num_graphs
    scores = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    for g, rec in enumerate(payload["samples"]):
        lo, hi = int(sb.edge_ptr[g]), int(sb.edge_ptr[g + 1])
        assert rec["sample_id"] == sb.sample_id[g] and rec["question"] == f"q{g}?"
        assert rec["answer_entity_ids"] == sb.answer_entity_ids[sb.answer_ptr[g]: sb.answer_ptr[g + 1]].tolist()
        _, order = topk_desc(logits[lo:hi], 50)  # sigmoid is monotone: same ranking away from saturation
        assert sorted(rec["triplets_by_k"].keys()) == [1, 5, 10, 50]
        for k, rows in rec["triplets_by_k"].items():
            assert len(rows) == min(k, hi - lo)
            for i, row in enumerate(rows):
                e = lo + int(order[i])
                assert row["rank"] == i + 1
                assert row["head_entity_id"] == int(sb.node_global_ids[sb.edge_index[0, e]])
                assert row["tail_entity_id"] == int(sb.node_global_ids[sb.edge_index[1, e]])
                assert row["relation_id"] == int(sb.edge_attr[e]) and row["label"] == float(sb.labels[e])
                assert row["score"] == pytest.approx(scores[e], abs=1e-6)
                assert row["logit_fwd"] == pytest.approx(float(lf[e])) and row["logit_bwd"] == pytest.approx(float(lb[e]))
                assert row["head_text"] is None and row["relation_text"] is None
    w2 = RetrieverTopKEdgeWriter(output_dir=tmp_path / "eval_retriever", overwrite=False)
    with pytest.raises(FileExistsError):
        w2.on_predict_start()
    with pytest.raises(ValueError, match="non-empty"):
        RetrieverTopKEdgeWriter(output_dir=tmp_path, artifact_name=" ")
