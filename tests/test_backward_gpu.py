"""GPU: backward of the edge scorer against the REFERENCE's autograd.

tests/golden/retriever_{toy,mid,fwd,bwd}.npz hold, next to the forward outputs, d(sum_e g_e logit_e)/d(parameter) for every
parameter of the reference Retriever (eval-mode graph, fixed g) as its own autograd computed it
(tests/golden/make_golden.py:gen_retriever).  evi_retriever_backward must reproduce all 25 gradients — for the bidirectional
toy (one chunk, relations de-duplicated), the mid batch (D != H, 3 + 1 DDE rounds), and the single-direction modes —
with the exact-f32 GEMMs to ~1e-5 and with the default split-bf16 GEMMs to ~1e-4 of each gradient's scale.
"""
import os
import types

import numpy as np
import pytest
import torch

from evi_rag_amd import synthetic

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _batch_from(z, dev, prefix="b_"):
    b = types.SimpleNamespace()
    for k in z.files:
        if k.startswith(prefix):
            setattr(b, k[len(prefix):], torch.from_numpy(z[k]).to(dev))
    b.num_graphs = int(b.ptr.numel() - 1)
    b.num_nodes = int(b.ptr[-1].item())
    b._slice_dict = {"edge_index": b.edge_ptr}
    return b


def _model_from(z, dev, **kw):
    from evi_rag_amd.retriever import Retriever

    rounds = z["rounds"].tolist()
    m = Retriever(emb_dim=int(z["D"]), hidden_dim=int(z["H"]), dde_cfg={"num_rounds": rounds[0], "num_reverse_rounds": rounds[1]},
                  direction_mode=str(z["direction"]), **kw)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_")}, strict=True)
    return m.to(dev).eval()


def _check_grads(model, z, rel_tol):
    worst = {}
    for name, p in model.named_parameters():
        want = z["g_" + name]
        assert p.grad is not None, name
        got = p.grad.detach().cpu().numpy()
        assert got.shape == want.shape, name
        scale = max(float(np.abs(want).max()), 1e-6)
        err = float(np.abs(got - want).max()) / scale
        worst[name] = err
        assert err <= rel_tol, (name, err, scale)
    return worst


@pytest.mark.parametrize("name", ["retriever_toy", "retriever_mid", "retriever_fwd", "retriever_bwd"])
@pytest.mark.parametrize("gemm,tol", [("f32", 2e-5), ("default", 2e-4)])
def test_parameter_gradients_match_the_reference_autograd(dev, name, gemm, tol, monkeypatch):
    if gemm == "f32":
        monkeypatch.setenv("EVI_SCORER_GEMM", "f32")
    else:
        monkeypatch.delenv("EVI_SCORER_GEMM", raising=False)
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    model = _model_from(z, dev)
    model.differentiable = True  # eval-mode graph (the reference's gradients were taken in eval mode: no dropout RNG)
    batch = _batch_from(z, dev)
    gvec = torch.from_numpy(z["gvec"]).to(dev)
    out = model(batch)
    assert out.logits.requires_grad
    np.testing.assert_allclose(out.logits.detach().cpu().numpy(), z["logits"], rtol=0, atol=2e-4)
    (out.logits * gvec).sum().backward()
    torch.cuda.synchronize()
    _check_grads(model, z, tol)
    # a second backward pass gives the same bits (fixed reduction orders, no float atomics)
    first = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    (model(batch).logits * gvec).sum().backward()
    for n, p in model.named_parameters():
        assert torch.equal(p.grad, first[n]), n


def test_backward_without_relation_dedupe_and_through_the_loss(dev):
    """(a) dedupe_relations=False (relation rows projected per edge): same gradients as the de-duplicated path;
    (b) the training-shaped use: RetrieverLoss on the differentiable forward, loss.backward(), every parameter gets a finite
    gradient, and train() mode with dropout_p = 0 takes the same path."""
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.retriever import Retriever

    z = np.load(os.path.join(GOLD, "retriever_mid.npz"), allow_pickle=False)
    batch = _batch_from(z, dev)
    gvec = torch.from_numpy(z["gvec"]).to(dev)
    grads = {}
    for dedupe in (True, False):
        m = _model_from(z, dev, dedupe_relations=dedupe)
        m.differentiable = True
        (m(batch).logits * gvec).sum().backward()
        grads[dedupe] = {n: p.grad.clone() for n, p in m.named_parameters()}
        _check_grads(m, z, 2e-4)
    for n in grads[True]:
        scale = float(grads[True][n].abs().max()) + 1e-6
        assert float((grads[True][n] - grads[False][n]).abs().max()) / scale < 2e-4, n

    m = _model_from(z, dev, dropout_p=0.0, hide_seek_cfg={"enabled": False})
    m.train()
    out = m(batch)
    loss = RetrieverLoss(infonce_temperature=0.07)(out, batch.labels, edge_batch=out.query_ids, num_graphs=batch.num_graphs)
    loss.loss.backward()
    for n, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    assert float(m.state_net[0].weight.grad.abs().max()) > 0


def test_backward_multi_chunk_batch_matches_finite_differences(dev, monkeypatch):
    """More than one 65 536-edge chunk (weight gradients accumulate over chunks) at D = H = 32: the directional derivative
    along a random parameter direction equals the central finite difference of the (f32-GEMM) forward."""
    from evi_rag_amd.retriever import Retriever

    monkeypatch.setenv("EVI_SCORER_GEMM", "f32")
    D = H = 32
    sb = synthetic.make_batch(24, nodes_per_graph=900, edges_per_graph=3000, emb_dim=D, num_relations=50, seed=12)
    assert sb.num_edges > 65536
    batch = synthetic.as_namespace(sb, device=dev)
    torch.manual_seed(5)
    model = Retriever(emb_dim=D, hidden_dim=H).to(dev).eval()
    model.differentiable = True
    g = torch.randn(sb.num_edges, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / sb.num_edges ** 0.5
    (model(batch).logits * g).sum().backward()
    direction = {n: torch.randn_like(p) for n, p in model.named_parameters()}
    analytic = sum(float((p.grad.double() * direction[n].double()).sum()) for n, p in model.named_parameters())
    model.differentiable = False
    eps = 1e-3

    def f(sign):
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.add_(sign * eps * direction[n])
            val = float((model(batch).logits.double() * g.double()).sum())
            for n, p in model.named_parameters():
                p.sub_(sign * eps * direction[n])
        return val

    numeric = (f(+1) - f(-1)) / (2 * eps)
    assert abs(analytic - numeric) <= 2e-3 * max(1.0, abs(numeric)), (analytic, numeric)


@pytest.mark.parametrize("direction_mode", ["bidirectional", "forward"])
def test_replayed_forward_intermediates_give_the_recomputing_backwards_bits(dev, direction_mode):
    """keep_forward_intermediates (the forward's per-edge rows kept in `saved`, the backward replays them) against the
    backward that recomputes the forward: same kernels on the same values, so every gradient is equal bit for bit — over
    two chunks and in a single-direction mode (a different `saved` layout)."""
    from evi_rag_amd.retriever import Retriever

    D = H = 64
    sb = synthetic.make_batch(24, nodes_per_graph=900, edges_per_graph=3000, emb_dim=D, num_relations=50, seed=3)
    assert sb.num_edges > 65536
    batch = synthetic.as_namespace(sb, device=dev)
    batch.num_relations = 50
    torch.manual_seed(2)
    model = Retriever(emb_dim=D, hidden_dim=H, direction_mode=direction_mode).to(dev).eval()
    model.differentiable = True
    g = torch.randn(sb.num_edges, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    grads = {}
    for keep in (True, False):
        model.keep_forward_intermediates = keep
        model.zero_grad(set_to_none=True)
        (model(batch).logits * g).sum().backward()
        grads[keep] = {n: p.grad.clone() for n, p in model.named_parameters()}
    for n in grads[True]:
        assert torch.equal(grads[True][n], grads[False][n]), n
    assert float(grads[True]["state_net.0.weight"].abs().max()) > 0


def _dropout_multipliers(seed, E, H, p):
    """The kernel's dropout mask restated with numpy integers (csrc/scorer.hip: dropout_mul4): [2, E, H] multipliers."""
    thr = min(int(round(p * 65536.0)), 65535)
    scale = np.float32(65536.0 / (65536 - thr))
    rows = np.arange(2 * E, dtype=np.uint64)[:, None]
    quads = np.arange(H // 4, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (rows * np.uint64(H // 4) + quads + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = np.stack([(z >> np.uint64(16 * c)) & np.uint64(0xFFFF) for c in range(4)], axis=-1).reshape(2 * E, H)
    return np.where(u >= thr, scale, np.float32(0)).astype(np.float32).reshape(2, E, H)


def test_training_mode_dropout_and_hide_and_seek(dev, monkeypatch):
    """train() with the reference's defaults (dropout_p = 0.1, hide-and-seek on, retriever.py:105-183): (a) the logits equal
    the oracle's forward given the SAME dropout mask (the kernel's counter-based mask restated in numpy) — the mask is not
    torch's Philox stream, parity with the reference is in distribution; (b) the kept fraction is 1 - p; (c) hide-and-seek
    with p_near = p_far = 1 shifts every logit by exactly bias_near / bias_far; (d) the gradients under dropout agree with
    central finite differences of the forward run with the same seed."""
    from oracle import scorer as oscorer

    monkeypatch.setenv("EVI_SCORER_GEMM", "f32")
    z = np.load(os.path.join(GOLD, "retriever_toy.npz"), allow_pickle=False)
    batch = _batch_from(z, dev)
    w = {k[2:]: z[k] for k in z.files if k.startswith("w_")}
    rounds = z["rounds"].tolist()
    nb = types.SimpleNamespace(**{k[2:]: z[k] for k in z.files if k.startswith("b_")})
    E, H = int(batch.edge_index.size(1)), int(z["H"])
    p = 0.25
    m = _model_from(z, dev, dropout_p=p, hide_seek_cfg={"enabled": False})
    m.train()
    torch.manual_seed(11)
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    torch.manual_seed(11)
    with torch.no_grad():
        got = m(batch)
    mul = _dropout_multipliers(seed, E, H, p)
    assert abs(float((mul > 0).mean()) - (1 - p)) < 0.02
    want = oscorer.retriever_forward(w, nb, num_rounds=rounds[0], num_reverse_rounds=rounds[1], direction_mode=str(z["direction"]),
                                     dropout_mul=mul)
    np.testing.assert_allclose(got.logits.cpu().numpy(), want["logits"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(got.edge_embeddings.cpu().numpy(), want["edge_embeddings"], rtol=2e-4, atol=2e-4)
    with torch.no_grad():
        other = m(batch)  # the next draw: another mask
    assert float((other.logits - got.logits).abs().max()) > 1e-3

    # (c) hide-and-seek, every edge hidden
    hs = {"enabled": True, "p_near": 1.0, "p_far": 1.0, "bias_near": -2.0, "bias_far": -0.5, "apply_in_eval": False}
    mh = _model_from(z, dev, dropout_p=0.0, hide_seek_cfg=hs)
    with torch.no_grad():
        plain = mh(batch).logits
        mh.train()
        hidden = mh(batch).logits
    from evi_rag_amd import ops
    near = ops.qa_edge_mask(batch.edge_index, int(batch.num_nodes), batch.q_local_indices, batch.a_local_indices)
    shift = torch.where(near, torch.tensor(-2.0, device=dev), torch.tensor(-0.5, device=dev))
    assert float((hidden - plain - shift).abs().max()) < 1e-5

    # (d) gradients under dropout vs finite differences with the mask held fixed
    g = torch.from_numpy(z["gvec"]).to(dev)
    m.zero_grad(set_to_none=True)
    torch.manual_seed(11)
    (m(batch).logits * g).sum().backward()
    direction = {n: torch.randn_like(q) for n, q in m.named_parameters()}
    analytic = sum(float((q.grad.double() * direction[n].double()).sum()) for n, q in m.named_parameters())
    eps = 1e-3

    def f(sign):
        with torch.no_grad():
            for n, q in m.named_parameters():
                q.add_(sign * eps * direction[n])
            torch.manual_seed(11)
            val = float((m(batch).logits.double() * g.double()).sum())
            for n, q in m.named_parameters():
                q.sub_(sign * eps * direction[n])
        return val

    numeric = (f(+1) - f(-1)) / (2 * eps)
    assert abs(analytic - numeric) <= 3e-3 * max(1.0, abs(numeric)), (analytic, numeric)


@pytest.mark.parametrize("D,H", [(320, 256), (768, 768), (1024, 1024), (1280, 1280), (256, 1100)])
def test_backward_at_wide_dims_matches_finite_differences(dev, monkeypatch, D, H):
    """The per-edge backward kernels are templated on ceil(width / 256) (1..5 sixteen-byte column groups per lane); the golden
    batches are narrow (one group).  At the reference's widths — 768 (bge-base), 1024 (the default), 1280 (the maximum) and
    D != H — the directional derivative along a random parameter direction must equal the central finite difference of the
    forward (exact-f32 GEMMs), with dropout on, and the replaying and the recomputing backward must agree bit for bit."""
    from evi_rag_amd.retriever import Retriever

    monkeypatch.setenv("EVI_SCORER_GEMM", "f32")
    sb = synthetic.make_batch(3, nodes_per_graph=40, edges_per_graph=120, emb_dim=D, num_relations=9, seed=D + H)
    batch = synthetic.as_namespace(sb, device=dev)
    batch.num_relations = 9
    torch.manual_seed(D)
    model = Retriever(emb_dim=D, hidden_dim=H, dropout_p=0.2, hide_seek_cfg={"enabled": False}).to(dev).train()
    g = torch.randn(sb.num_edges, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / sb.num_edges ** 0.5
    grads = {}
    for keep in (True, False):
        model.keep_forward_intermediates = keep
        model.zero_grad(set_to_none=True)
        torch.manual_seed(77)  # the dropout seed is drawn from torch's CPU generator
        (model(batch).logits * g).sum().backward()
        grads[keep] = {n: p.grad.clone() for n, p in model.named_parameters()}
    for n in grads[True]:
        assert torch.equal(grads[True][n], grads[False][n]), n
        assert bool(torch.isfinite(grads[True][n]).all()), n
    direction = {n: torch.randn_like(p) / p.numel() ** 0.5 for n, p in model.named_parameters()}
    analytic = sum(float((grads[True][n].double() * direction[n].double()).sum()) for n in direction)
    eps = 2e-3

    def f(sign):
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.add_(sign * eps * direction[n])
            torch.manual_seed(77)
            val = float((model(batch).logits.double() * g.double()).sum())
            for n, p in model.named_parameters():
                p.sub_(sign * eps * direction[n])
        return val

    numeric = (f(+1) - f(-1)) / (2 * eps)
    assert abs(analytic - numeric) <= 5e-3 * max(0.05, abs(numeric)), (analytic, numeric)


def test_hide_and_seek_index_check_is_deferred_not_dropped(dev):
    """The range check of the seed / answer indices behind the hide-and-seek mask no longer costs a host-device
    synchronisation per forward: a bad index sets a bit in the module's sticky status word and `check_deferred()` raises
    the reference's ValueError (src/utils/graph_utils.py:107-153) — once per epoch in RetrieverTrainer / RetrieverEvaluator."""
    from evi_rag_amd.retriever import Retriever

    sb = synthetic.make_batch(3, nodes_per_graph=30, edges_per_graph=90, emb_dim=16, num_relations=7, seed=5)
    batch = synthetic.as_namespace(sb, device=dev)
    hs = {"enabled": True, "p_near": 0.7, "p_far": 0.1, "bias_near": -2.0, "bias_far": -0.5}
    m = Retriever(emb_dim=16, hidden_dim=16, dropout_p=0.0, hide_seek_cfg=hs).to(dev).train()
    with torch.no_grad():
        m(batch)
    m.check_deferred()  # clean batch: nothing pending
    batch.q_local_indices = batch.q_local_indices.clone()
    batch.q_local_indices[0] = int(batch.num_nodes) + 5
    if hasattr(batch, "edge_is_near"):
        del batch.edge_is_near
    with torch.no_grad():
        m(batch)  # scored (the bad index is ignored by the mask kernel), flagged
    with pytest.raises(ValueError, match="q/a local indices exceed num_nodes"):
        m.check_deferred()
    m.check_deferred()  # the flag was consumed


def test_backward_narrow_model_many_edges(dev, monkeypatch):
    """D = H = 16 with F = 20 structure features (F > max(D, H)) and enough edges for the full slice count of the
    weight-gradient products: the [S][M][N] partial buffer is sized by max(D, H, F) (it used to be max(D, H): the
    struct_proj.0 product overran it by a quarter).  Split-bf16 and exact-f32 gradients must agree."""
    from evi_rag_amd.retriever import Retriever

    sb = synthetic.make_batch(6, nodes_per_graph=700, edges_per_graph=2200, emb_dim=16, num_relations=9, seed=21)
    assert sb.num_edges > 8192
    batch = synthetic.as_namespace(sb, device=dev)
    batch.num_relations = 9
    g = torch.randn(sb.num_edges, device=dev, generator=torch.Generator(device=dev).manual_seed(2)) / sb.num_edges ** 0.5
    grads = {}
    for mode in ("bf16", "f32"):
        if mode == "f32":
            monkeypatch.setenv("EVI_SCORER_GEMM", "f32")
        torch.manual_seed(8)
        m = Retriever(emb_dim=16, hidden_dim=16).to(dev).eval()
        m.differentiable = True
        (m(batch).logits * g).sum().backward()
        grads[mode] = {n: p.grad.clone() for n, p in m.named_parameters()}
    for n in grads["f32"]:
        scale = float(grads["f32"][n].abs().max()) + 1e-6
        assert float((grads["bf16"][n] - grads["f32"][n]).abs().max()) / scale < 3e-4, n


def test_bf16_matmul_precision_tracks_the_split_products(dev):
    """`matmul_precision = "bf16"` (EviRetrieverBatch.matmul_precision = 1; Lightning's `precision: bf16-mixed`,
    configs/trainer/default.yaml:13-14): the large products multiply one bf16 product instead of three, forward and backward.
    Against the default on the same weights, batch and dropout seed: logits within bf16's rounding of the default's (and not equal:
    the mode is really on), every large gradient within 2 % of its norm and at cosine >= 0.999; an unknown value is refused."""
    from evi_rag_amd.retriever import Retriever

    D, H = 128, 160
    sb = synthetic.make_batch(8, nodes_per_graph=300, edges_per_graph=2000, emb_dim=D, num_relations=40, seed=5)
    batch = synthetic.as_namespace(sb, device=dev)
    batch.num_relations = 40
    torch.manual_seed(11)
    model = Retriever(emb_dim=D, hidden_dim=H, dropout_p=0.1, hide_seek_cfg={"enabled": False}).to(dev).train()
    g = torch.randn(sb.num_edges, device=dev, generator=torch.Generator(device=dev).manual_seed(2)) / sb.num_edges ** 0.5
    runs = {}
    for mode in ("split", "bf16"):
        model.matmul_precision = mode
        model.zero_grad(set_to_none=True)
        torch.manual_seed(99)
        out = model(batch)
        (out.logits * g).sum().backward()
        runs[mode] = (out.logits.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()})
    ls, lb = runs["split"][0], runs["bf16"][0]
    scale = float(ls.std())
    err = float((ls - lb).abs().max())
    assert 1e-6 * scale < err <= 0.05 * scale, (err, scale)
    for n, gs in runs["split"][1].items():
        gb = runs["bf16"][1][n]
        assert bool(torch.isfinite(gb).all()), n
        norm = float(gs.norm())
        if gs.numel() < 1024 or norm < 1e-6:
            continue  # biases / LayerNorm vectors: sums of many rounded rows, checked through the matrices they feed
        rel = float((gs - gb).norm()) / norm
        cos = float((gs * gb).sum()) / (norm * float(gb.norm()))
        assert rel <= 0.02 and cos >= 0.999, (n, rel, cos)
    # evaluation (weights prepared once and cached: the hi planes are the single product's operands) honours it too
    model.eval()
    with torch.no_grad():
        ev = {}
        for mode in ("split", "bf16"):
            model.matmul_precision = mode
            ev[mode] = model(batch).logits.clone()
    err = float((ev["split"] - ev["bf16"]).abs().max())
    assert 1e-6 * scale < err <= 0.05 * scale, (err, scale)
    model.matmul_precision = "fp8"
    with pytest.raises(ValueError, match="matmul_precision"):
        model(batch)
    with pytest.raises(ValueError, match="matmul_precision"):
        Retriever(emb_dim=D, hidden_dim=H, matmul_precision="fp8")
