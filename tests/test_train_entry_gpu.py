"""GPU: the training entry point end to end — a reference-shaped config tree, `experiment=train_retriever dataset=...` on the
command line, packed train / validation splits + embedding tables on disk; then the evaluation entry point on the checkpoint it
wrote (the hand-over the reference does between `src/train.py` and `src/eval.py`)."""
import json

import numpy as np
import pytest
import torch

from tests.config_tree import write_tree

pytestmark = pytest.mark.gpu


def test_train_entry_point_then_eval_entry_point(dev, tmp_path, monkeypatch):
    from evi_rag_amd import eval as ev
    from evi_rag_amd import packed_dataset as pd, synthetic, train_entry

    data_dir = tmp_path / "data"
    monkeypatch.setenv("EVI_TEST_PROJECT_ROOT", str(tmp_path / "proj"))
    cfg_dir = write_tree(tmp_path, data_dir)
    rng = np.random.default_rng(5)
    ent = torch.from_numpy(rng.standard_normal((2000, 16)).astype(np.float32))
    rel = torch.from_numpy(rng.standard_normal((9, 16)).astype(np.float32))
    for variant in ("toyqa", "toyqa-sub"):
        emb = data_dir / variant / "materialized" / "embeddings"
        emb.mkdir(parents=True)
        torch.save(ent, emb / "entity_embeddings.pt")
        torch.save(rel, emb / "relation_embeddings.pt")
        for split, graphs, seed in (("train", 24, 1), ("validation", 8, 2), ("test", 8, 3)):
            base = synthetic.make_batch(graphs, nodes_per_graph=40, edges_per_graph=150, emb_dim=16, num_relations=9, seed=seed,
                                        num_entities=2000)
            pd.write_packed(emb / f"{split}.packed", pd.samples_from_flat_batch(base))

    out = train_entry.run(cfg_dir, ["experiment=train_retriever", "dataset=toyqa"], device=str(dev))
    hist = out["history"]
    assert len(hist) == 6 and [h["epoch"] for h in hist] == list(range(6))          # trainer.max_epochs of the experiment
    assert all("val/answer/reachability@20" in h and "train/loss" in h for h in hist)  # check_val_every_n_epoch: 1
    assert hist[-1]["train/loss"] < hist[0]["train/loss"]
    assert abs(hist[0]["lr"] - (1e-6 + (3e-3 - 1e-6) * (1 + np.cos(np.pi / 6)) / 2)) < 1e-9  # model.scheduler_cfg: cosine, t_max 6
    ckpt_dir = tmp_path / "proj" / "logs" / "train_retriever_toyqa" / "runs" / "fixed" / "checkpoints"  # callbacks.model_checkpoint.dirpath
    assert out["checkpoint_dir"] == str(ckpt_dir) and (ckpt_dir / "last.ckpt").exists()                  # save_last: true
    best = out["best_checkpoint"]
    assert best is not None and best.endswith(".ckpt") and "epoch_00" in best                               # filename: epoch_{epoch:03d}
    assert len(list(ckpt_dir.glob("epoch_*.ckpt"))) == 1                                                   # save_top_k: 1
    assert out["best"] == max(h["val/answer/reachability@20"] for h in hist)                               # mode: max
    saved = json.loads((ckpt_dir.parent / "train_history.json").read_text())
    assert len(saved) == 6

    # resume: two more epochs from last.ckpt
    more = train_entry.run(cfg_dir, ["experiment=train_retriever", "dataset=toyqa", f"ckpt_path={ckpt_dir / 'last.ckpt'}",
                                     "trainer.max_epochs=8"], device=str(dev))
    assert [h["epoch"] for h in more["history"]] == [6, 7]
    # the resumed run inherits the selection state (the checkpoint's `callbacks` entry): a post-resume validation that is not
    # better than the best so far neither replaces the best checkpoint nor leaves a second epoch_*.ckpt behind
    post = max(h["val/answer/reachability@20"] for h in more["history"])
    assert more["best"] == max(out["best"], post)
    assert len(list(ckpt_dir.glob("epoch_*.ckpt"))) == 1
    if post <= out["best"]:
        assert more["best_checkpoint"] == best and (ckpt_dir / best.split("/")[-1]).exists()
    blob = torch.load(ckpt_dir / "last.ckpt", map_location="cpu", weights_only=True)
    assert blob["callbacks"]["best"] == more["best"] and blob["callbacks"]["best_path"] == more["best_checkpoint"]
    assert set(blob["callbacks"]) == {"best", "best_path", "es_best", "bad_checks", "stop_latched"}
    best = more["best_checkpoint"]

    # src/eval.py's side of the hand-over: the evaluation entry point loads the best checkpoint strictly and evaluates
    results = ev.run(cfg_dir, ["experiment=eval_retriever", "dataset=toyqa", f"ckpt.retriever={best}"], device=str(dev))
    assert results and all("test/loss" in m for _, _, m in results)
