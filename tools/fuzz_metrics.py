#!/usr/bin/env python3
"""Randomised cross-check of the fused ranking metrics (per-graph top-k, edge recall@k, answer reachability@k by union-find,
answer hit / recall@k, score margin) against the oracle over odd batches: one-node graphs, graphs with a single edge, many
exact score ties, k beyond the edge count, up to 40 answers.  python tools/fuzz_metrics.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evi_rag_amd import metrics as M, synthetic  # noqa: E402
from oracle import metrics as omet  # noqa: E402
from oracle.ranking import segment_topk as oracle_segment_topk  # noqa: E402

K_VALUES = (1, 5, 10, 25, 50, 100, 200, 300, 400, 500)


def one_case(rng, dev):
    B = int(rng.choice([1, 2, 7, 33]))
    n = int(rng.choice([1, 2, 9, 150, 2500]))
    e = int(rng.choice([1, 2, 30, 700, 9000]))
    sb = synthetic.make_batch(B, nodes_per_graph=n, edges_per_graph=e, emb_dim=4, seed=int(rng.integers(1 << 30)), attach_embeddings=False,
                              max_answers=int(rng.choice([1, 5, 40])), size_jitter=float(rng.choice([0.0, 0.5])))
    if sb.num_edges == 0:
        return "skipped"
    scores = rng.standard_normal(sb.num_edges).astype(np.float32)
    if rng.random() < 0.5:
        scores = np.round(scores, int(rng.integers(0, 2)))  # heavy ties
    target = sb.labels > 0.5
    ns = synthetic.as_namespace(sb, device=dev)
    ns.answer_entity_ids_ptr = torch.from_numpy(sb.answer_ptr).to(dev)
    rb = M.rank_batch(torch.from_numpy(scores).to(dev), torch.from_numpy(target).to(dev), ns, K_VALUES, want_topk=True)
    ridx, rval, rcnt = oracle_segment_topk(scores, sb.edge_ptr, K_VALUES[-1])
    assert np.array_equal(rb.topk_index.cpu().numpy(), ridx) and np.array_equal(rb.topk_score.cpu().numpy(), rval)
    assert np.array_equal(rb.topk_count.cpu().numpy(), rcnt)
    sums, cnt = omet.edge_recall_at_k(scores, target, sb.edge_ptr, K_VALUES)
    got = (rb.edge_recall.double() * rb.recall_valid.unsqueeze(1)).sum(0).cpu().numpy()
    np.testing.assert_allclose(got, [sums[k] for k in K_VALUES], rtol=0, atol=1e-6)
    assert float(rb.recall_valid.sum().item()) == cnt
    hits, valid = omet.answer_reachability(scores, sb, K_VALUES)
    assert float(rb.reach_valid.sum().item()) == valid
    assert (rb.reach.long() * rb.reach_valid.long().unsqueeze(1)).sum(0).cpu().tolist() == [int(hits[k]) for k in K_VALUES]
    h, r = omet.answer_hit_recall_batch(scores, sb, K_VALUES)
    nvalid = max(float(rb.answer_valid.sum().item()), 1.0)
    np.testing.assert_allclose((rb.answer_hit.double().sum(0) / nvalid).cpu().numpy(), [h[f"answer_hit@{k}"] for k in K_VALUES], atol=1e-9)
    np.testing.assert_allclose((rb.answer_recall.double().sum(0) / nvalid).cpu().numpy(), [r[f"answer_recall@{k}"] for k in K_VALUES], atol=1e-6)
    return f"B={B} n~{n} e~{e} E={sb.num_edges}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    for i in range(cases):
        print(i, one_case(rng, dev), flush=True)
    print("fuzz ok")


if __name__ == "__main__":
    main()
