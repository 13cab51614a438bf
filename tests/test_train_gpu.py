"""GPU: the training path's optimiser step and trainer.

`FlatAdamW` (csrc/optim.hip) is pinned to torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ — what the reference trains
with (src/utils/optimization.py:20-35, configs/model/retriever_module.yaml:37-40, configs/trainer/default.yaml:20) — run on
the same gradients; `RetrieverTrainer.training_step` to the same loop written with torch's optimiser around the
differentiable mirror; the two-rank form (gradients averaged over ranks, configs/trainer/ddp.yaml) to the one-process
average of the two ranks' gradients.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_flat_adamw_equals_torch_adamw_with_gradient_clipping(dev):
    from evi_rag_amd.train import FlatAdamW

    torch.manual_seed(0)
    shapes = [(33, 17), (5,), (1,), (64, 64), (7, 3)]  # element counts that are not multiples of 4: padded slots in the flat buffer
    ours = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    opt = FlatAdamW(ours, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    ref = torch.optim.AdamW(theirs, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05, foreach=False)
    for step in range(8):
        scale = 10.0 if step % 2 == 0 else 0.01  # clipped and unclipped steps
        opt.zero_grad()
        for p, q in zip(ours, theirs):
            g = torch.randn_like(p) * scale
            p.grad.add_(g)
            q.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(theirs, 1.0)
        ref.step()
        opt.step(max_norm=1.0)
        for p, q in zip(ours, theirs):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-7), (step, float((p - q).abs().max()))
    # grad_scale (the 1 / world of a summed all-reduce) is part of the clipped quantity
    opt.zero_grad()
    for p, q in zip(ours, theirs):
        g = torch.randn_like(p)
        p.grad.add_(2.0 * g)
        q.grad = g.clone()
    torch.nn.utils.clip_grad_norm_(theirs, 1.0)
    ref.step()
    opt.step(grad_scale=0.5, max_norm=1.0)
    for p, q in zip(ours, theirs):
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-7)
    v0 = ours[0]._version
    opt.zero_grad()
    opt.step()
    assert ours[0]._version > v0  # the kernel's write is visible to version-keyed caches


def _manual_reference(dev, batches, steps, lr, t_max=None, more_steps=0, more_lr=None):
    """The same training loop with torch.optim.AdamW + clip_grad_norm_ around the differentiable mirror; `batches`: the
    per-rank batches whose gradients are averaged (DDP)."""
    sys.path.insert(0, HERE)
    import train_rank_worker as w
    from evi_rag_amd.loss import RetrieverLoss

    model = w.make_model(dev)
    model.train()
    loss_fn = RetrieverLoss(infonce_temperature=0.5)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4, foreach=False)
    losses = []
    for it in range(steps + more_steps):
        if it == steps and more_lr is not None:
            for gp in opt.param_groups:
                gp["lr"] = more_lr
        opt.zero_grad(set_to_none=True)
        acc = None
        per_rank = []
        for b in batches:
            for p in model.parameters():
                p.grad = None
            out = model(b)
            lo = loss_fn(out, b.labels, edge_batch=out.query_ids, num_graphs=b.num_graphs)
            lo.loss.backward()
            per_rank.append(float(lo.loss.detach()))
            gs = [p.grad.clone() for p in model.parameters()]
            acc = gs if acc is None else [a + g for a, g in zip(acc, gs)]
        for p, g in zip(model.parameters(), acc):
            p.grad = g / len(batches)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(per_rank)
    return model, losses


def test_trainer_step_equals_the_torch_optimizer_loop(dev):
    sys.path.insert(0, HERE)
    import train_rank_worker as w
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.train import RetrieverTrainer

    batch = w.make_batch(0, dev)
    model = w.make_model(dev)
    trainer = RetrieverTrainer(model, loss=RetrieverLoss(infonce_temperature=0.5), optimizer_cfg={"type": "adamw", "lr": 1e-2, "weight_decay": 1e-4},
                               scheduler_cfg={"type": "cosine", "t_max": 4, "eta_min": 1e-6}, gradient_clip_val=1.0)
    losses = [float(trainer.training_step(batch)) for _ in range(w.STEPS)]
    log = trainer.on_train_epoch_end()
    ref_model, ref_losses = _manual_reference(dev, [batch], w.STEPS, 1e-2)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b[0]) <= 2e-5 * max(1.0, abs(b[0])), (losses, ref_losses)
    assert losses[-1] < losses[0]  # it learns
    for (n, p), q in zip(model.named_parameters(), ref_model.parameters()):
        assert _close(n, p.detach().cpu().numpy(), q.detach().cpu().numpy(), w.STEPS, 1e-2), (n, float((p - q).abs().max()))
    assert abs(log["train/loss"] - sum(losses) / len(losses)) < 1e-6
    assert abs(log["lr"] - (1e-6 + (1e-2 - 1e-6) * (1 + math.cos(math.pi / 4)) / 2)) < 1e-12  # CosineAnnealingLR after one epoch
    # evaluation after training sees the new weights (the prepared-weights cache is keyed on version counters)
    model.eval()
    with torch.no_grad():
        a = model(batch).logits
    ref_model.eval()
    with torch.no_grad():
        b = ref_model(batch).logits
    assert torch.allclose(a - a.mean(), b - b.mean(), rtol=1e-3, atol=2e-4)  # up to the common shift of the two noise-only biases
    assert float((a.mean() - b.mean()).abs()) <= 4.0 * w.STEPS * 1e-2
    # checkpoint / resume: one more step from a restored trainer equals one more step of the original
    sd = trainer.state_dict()
    model2 = w.make_model(dev)
    trainer2 = RetrieverTrainer(model2, loss=RetrieverLoss(infonce_temperature=0.5), optimizer_cfg={"type": "adamw", "lr": 1e-2, "weight_decay": 1e-4},
                                scheduler_cfg={"type": "cosine", "t_max": 4, "eta_min": 1e-6}, gradient_clip_val=1.0)
    trainer2.load_state_dict(sd)
    l1, l2 = float(trainer.training_step(batch)), float(trainer2.training_step(batch))
    assert l1 == l2
    for p, q in zip(model.parameters(), model2.parameters()):
        assert torch.equal(p, q)
    with pytest.raises(ValueError, match="Unsupported optimizer type"):
        RetrieverTrainer(w.make_model(dev), optimizer_cfg={"type": "lion"})


def test_bf16_mixed_trainer_learns_like_the_default(dev):
    """`trainer.precision: bf16-mixed` (configs/trainer/default.yaml:13-14) -> single-product bf16 GEMMs in the forward and the
    backward: from the same initial weights on the same batch the loss falls like the default's (step by step within 5 %), and
    the values Lightning's other precisions map to leave the model's setting alone."""
    sys.path.insert(0, HERE)
    import train_rank_worker as w
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.train import RetrieverTrainer

    batch = w.make_batch(0, dev)
    curves = {}
    for precision in ("32-true", "bf16-mixed"):
        model = w.make_model(dev)
        trainer = RetrieverTrainer(model, loss=RetrieverLoss(infonce_temperature=0.5), precision=precision,
                                   optimizer_cfg={"type": "adamw", "lr": 1e-2, "weight_decay": 1e-4}, gradient_clip_val=1.0)
        assert model.matmul_precision == ("bf16" if precision == "bf16-mixed" else "split")
        curves[precision] = [float(trainer.training_step(batch)) for _ in range(6)]
    a, b = curves["32-true"], curves["bf16-mixed"]
    assert b[-1] < b[0] - 0.05, b
    assert a != b  # the mode is on
    for x, y in zip(a, b):
        assert abs(x - y) <= 0.05 * abs(x), (a, b)
    model = w.make_model(dev)
    model.matmul_precision = "bf16"
    RetrieverTrainer(model, precision="16-mixed")
    assert model.matmul_precision == "bf16"
    with pytest.raises(ValueError, match="precision"):
        RetrieverTrainer(w.make_model(dev), precision="fp8")


# InfoNCE's logit gradients sum to zero inside every graph, so the gradients of the two biases behind the logits
# (state_net.4.bias, score_head.bias) are pure rounding noise — which Adam normalises to steps of size lr.  Their values
# are compared only to within the distance such steps can cover.
NOISE_ONLY = ("state_net.4.bias", "score_head.bias")


def _close(name, a, b, steps, lr):
    a, b = np.asarray(a), np.asarray(b)
    if name in NOISE_ONLY:
        return bool(np.abs(a - b).max() <= 2.0 * steps * lr)
    return bool(np.allclose(a, b, rtol=1e-4, atol=2e-5))


sys.path.insert(0, HERE)
from _procs import run_ranks  # noqa: E402


@pytest.mark.timeout(600)
def test_two_rank_training_averages_gradients(dev, tmp_path):
    world = 2
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    run_ranks(lambda r, port: [sys.executable, os.path.join(HERE, "train_rank_worker.py"), str(r), str(world), str(port), str(tmp_path)],
              world, env=env, timeout=400)
    z = [np.load(tmp_path / f"train_rank{r}.npz") for r in range(world)]
    sys.path.insert(0, HERE)
    import train_rank_worker as w

    # the workers ran 3 training steps, closed the epoch (one cosine step of t_max 4), then fit() 2 more steps at the new
    # learning rate; the torch loop below does the same
    ref_model, ref_losses = _manual_reference(dev, [w.make_batch(0, dev), w.make_batch(1, dev)], w.STEPS, 1e-2, more_steps=2,
                                              more_lr=1e-6 + (1e-2 - 1e-6) * (1 + math.cos(math.pi / 4)) / 2)
    for n, q in ref_model.named_parameters():
        assert np.array_equal(z[0][n], z[1][n]), n  # the ranks stay in lock-step, bit for bit
        assert _close(n, z[0][n], q.detach().cpu().numpy(), w.STEPS + 2, 1e-2), n
    for r in range(world):
        assert np.allclose(z[r]["__losses"], [l[r] for l in ref_losses[: w.STEPS]], rtol=2e-5)
    mean = sum(sum(l) for l in ref_losses[: w.STEPS]) / (world * w.STEPS)  # equal graph counts: the weighted mean is the plain mean
    assert abs(float(z[0]["__epoch_loss"]) - mean) < 1e-5 and float(z[0]["__epoch_loss"]) == float(z[1]["__epoch_loss"])


def test_fit_over_a_packed_split_then_evaluate_from_the_checkpoint(dev, tmp_path):
    """The whole loop on the data path the evaluation uses: packed split resident in HBM -> PackedLoader (shuffled) ->
    RetrieverTrainer.fit with the reference's training-mode defaults (dropout, hide-and-seek) -> a Lightning-layout
    checkpoint -> `eval.load_checkpoint_strict` (what `src/eval.py:80-111` does) into a fresh model -> RetrieverEvaluator.
    Training must lower the evaluation loss, and the restored model must evaluate exactly like the trained one."""
    from evi_rag_amd import eval as ev
    from evi_rag_amd import packed_dataset as pd, synthetic
    from evi_rag_amd.eval_loop import RetrieverEvaluator
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.retriever import Retriever
    from evi_rag_amd.train import RetrieverTrainer

    D = 32
    base = synthetic.make_batch(24, nodes_per_graph=50, edges_per_graph=160, emb_dim=D, num_relations=11, seed=9)
    pd.write_packed(tmp_path / "train.packed", pd.samples_from_flat_batch(base))
    ent = torch.from_numpy(np.random.default_rng(1).standard_normal((int(base.node_embedding_ids.max()) + 1, D)).astype(np.float32))
    rel = torch.from_numpy(np.random.default_rng(2).standard_normal((11, D)).astype(np.float32))
    torch.save(ent, tmp_path / "entity_embeddings.pt")
    torch.save(rel, tmp_path / "relation_embeddings.pt")
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore

    store = GlobalEmbeddingStore(tmp_path, device=dev)
    ds = pd.PackedRetrievalDataset(tmp_path / "train.packed", device=dev, embeddings=store)
    hs = {"enabled": True, "p_near": 0.7, "p_far": 0.1, "bias_near": -2.0, "bias_far": -0.5, "apply_in_eval": False}
    torch.manual_seed(4)
    model = Retriever(emb_dim=D, hidden_dim=D, dropout_p=0.1, hide_seek_cfg=hs).to(dev)
    loss = RetrieverLoss(infonce_temperature=0.5)

    def evaluate(m):
        m.eval()
        return RetrieverEvaluator(m, loss=loss, k_values=[1, 5, 20]).run(pd.PackedLoader(ds, batch_size=8))["metrics"]

    before = evaluate(model)
    trainer = RetrieverTrainer(model, loss=loss, optimizer_cfg={"type": "adamw", "lr": 3e-3, "weight_decay": 1e-4},
                               scheduler_cfg={"type": "cosine", "t_max": 6, "eta_min": 1e-6})
    log = trainer.fit(pd.PackedLoader(ds, batch_size=8, shuffle=True, random_seed=0), max_epochs=6)
    assert log["steps"] == 18 and len(log["epochs"]) == 6
    assert log["epochs"][-1]["train/loss"] < log["epochs"][0]["train/loss"]
    after = evaluate(model)
    assert after["test/loss"] < before["test/loss"]
    trainer.save_checkpoint(tmp_path / "last.ckpt")
    fresh = Retriever(emb_dim=D, hidden_dim=D, dropout_p=0.1, hide_seek_cfg=hs).to(dev)
    ev.load_checkpoint_strict(fresh, str(tmp_path / "last.ckpt"))
    restored = evaluate(fresh)
    assert restored == after
    # resume: a trainer restored from the file continues exactly where the first one stands
    t2 = RetrieverTrainer(Retriever(emb_dim=D, hidden_dim=D, dropout_p=0.1, hide_seek_cfg=hs).to(dev), loss=loss,
                          optimizer_cfg={"type": "adamw", "lr": 3e-3, "weight_decay": 1e-4}, scheduler_cfg={"type": "cosine", "t_max": 6, "eta_min": 1e-6})
    t2.load_checkpoint(tmp_path / "last.ckpt")
    assert t2.global_step == 18 and t2.current_epoch == 6 and t2.optimizer.lr == trainer.optimizer.lr
    batch = next(iter(pd.PackedLoader(ds, batch_size=8)))
    torch.manual_seed(1)
    a = float(trainer.training_step(batch))
    torch.manual_seed(1)
    b = float(t2.training_step(batch))
    assert a == b
    for p, q in zip(trainer.model.parameters(), t2.model.parameters()):
        assert torch.equal(p, q)


def test_example_script_runs(dev, tmp_path, capsys):
    """examples/train_and_eval_synthetic.py end to end at a small size: the training loss falls and the evaluation line prints."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("evi_example", os.path.join(os.path.dirname(HERE), "examples", "train_and_eval_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = mod.main(["--graphs", "48", "--nodes", "40", "--edges", "120", "--dim", "32", "--relations", "9", "--batch-size", "16",
                    "--epochs", "10", "--out", str(tmp_path)])
    # three noisy steps per epoch (dropout, hide-and-seek, a fresh shuffle per epoch): compare the ends of the run, not two epochs
    losses = [e["train/loss"] for e in log["epochs"]]
    assert log["steps"] == 30 and min(losses[-3:]) < losses[0], losses
    out = capsys.readouterr().out
    assert "questions_per_s" in out and "test/loss" in out
