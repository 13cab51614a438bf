#!/usr/bin/env python3
"""Randomised cross-check of the scorer (forward vs the oracle, backward vs central finite differences) over odd shapes:
widths that are not multiples of 256 or 64, D != H, a single graph, a single edge, every direction mode, DDE round counts,
relation de-duplication on / off / without the num_relations hint, dropout on.  Not part of the test suite (minutes of
oracle time); run it after touching scorer.hip / scorer_bwd.hpp:   python tools/fuzz_scorer.py [cases] [seed]"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EVI_SCORER_GEMM", "f32")
from evi_rag_amd import synthetic  # noqa: E402
from evi_rag_amd.retriever import Retriever  # noqa: E402
from oracle import scorer as oscorer  # noqa: E402


def one_case(rng, dev, idx):
    D = int(rng.choice([4, 8, 12, 36, 100, 260, 516]))
    H = int(rng.choice([4, 8, 20, 64, 132, 300]))
    B = int(rng.choice([1, 2, 5]))
    nodes = int(rng.choice([2, 7, 40]))
    edges = int(rng.choice([1, 3, 60, 200]))
    R = int(rng.choice([1, 3, 17]))
    rounds, rev = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    mode = str(rng.choice(["bidirectional", "forward", "backward"]))
    dedupe = bool(rng.integers(0, 2))
    hint = bool(rng.integers(0, 2))
    sb = synthetic.make_batch(B, nodes_per_graph=nodes, edges_per_graph=edges, emb_dim=D, num_relations=R, seed=int(rng.integers(1 << 30)),
                              size_jitter=0.0)
    if sb.num_edges == 0:
        return "skipped (no edges)"
    batch = synthetic.as_namespace(sb, device=dev)
    if hint:
        batch.num_relations = R
    torch.manual_seed(idx)
    m = Retriever(emb_dim=D, hidden_dim=H, dde_cfg={"num_rounds": rounds, "num_reverse_rounds": rev}, direction_mode=mode,
                  dedupe_relations=dedupe, dropout_p=0.0, hide_seek_cfg={"enabled": False}).to(dev).eval()
    with torch.no_grad():
        out = m(batch)
    w = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = oscorer.retriever_forward(w, sb, num_rounds=rounds, num_reverse_rounds=rev, direction_mode=mode)
    err = float(np.abs(out.logits.cpu().numpy() - want["logits"]).max())
    ferr = float(np.abs(out.edge_embeddings.cpu().numpy() - want["edge_embeddings"]).max())
    scale = max(1.0, float(np.abs(want["logits"]).max()))
    assert err <= 3e-4 * scale and ferr <= 3e-4 * max(1.0, float(np.abs(want["edge_embeddings"]).max())), ("forward", err, ferr)
    # backward vs finite differences (dropout on: the mask is a function of the seed drawn from torch's CPU generator)
    m.train()
    m.state_net[3].p = 0.3
    g = torch.randn(sb.num_edges, device=dev) / max(sb.num_edges, 1) ** 0.5
    torch.manual_seed(99)
    (m(batch).logits * g).sum().backward()
    direction = {n: torch.randn_like(p) / p.numel() ** 0.5 for n, p in m.named_parameters()}
    analytic = sum(float((p.grad.double() * direction[n].double()).sum()) for n, p in m.named_parameters())
    eps = 2e-3

    def f(sign):
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.add_(sign * eps * direction[n])
            torch.manual_seed(99)
            val = float((m(batch).logits.double() * g.double()).sum())
            for n, p in m.named_parameters():
                p.sub_(sign * eps * direction[n])
        return val

    numeric = (f(+1) - f(-1)) / (2 * eps)
    assert abs(analytic - numeric) <= 1e-2 * max(0.05, abs(numeric)), ("backward", analytic, numeric)
    return f"D={D} H={H} B={B} E={sb.num_edges} N={sb.num_nodes} R={R} rounds={rounds}+{rev} {mode} dedupe={dedupe} hint={hint}: fwd {err:.1e} bwd {abs(analytic - numeric):.1e}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    for i in range(cases):
        print(i, one_case(rng, dev, i), flush=True)
    print("fuzz ok")


if __name__ == "__main__":
    main()
