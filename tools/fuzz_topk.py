#!/usr/bin/env python3
"""Randomised cross-check of the cosine top-k paths against the oracle's full score matrix: the f32 scan, the two-stage
exact scan (must equal the scan bit for bit), the many-query GEMM-shaped pass (must equal the scan bit for bit), the f16 index
(overlap with the f32 result), shard merge (3 uneven shards == one pass) — over odd shapes: one row, k beyond the row count,
duplicate and zero rows, Q not a multiple of 32, D not a multiple of 64 (D % 16 == 0 is the scan's documented limit).   python tools/fuzz_topk.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evi_rag_amd import ops  # noqa: E402
from oracle import cosine as ocos  # noqa: E402
from tests.helpers import check_topk_against_scores  # noqa: E402


def one_case(rng, dev):
    N = int(rng.choice([1, 5, 97, 4096, 70001, 300000]))
    D = int(rng.choice([16, 48, 144, 384, 768]))  # the scan takes D % 16 == 0 (EVI_ERR_UNSUPPORTED otherwise)
    Q = int(rng.choice([1, 3, 32, 33, 100]))
    k = int(rng.choice([1, 7, 100, 500]))
    if N * D > 60_000_000:
        N = 70001
    x = rng.standard_normal((N, D), dtype=np.float32)
    if N > 3:
        x[0] = 0.0                       # a zero row (eps clamp)
        x[N - 1] = x[N // 2]             # an exact duplicate: the lower id must come first
    q = rng.standard_normal((Q, D), dtype=np.float32)
    if N > 3:
        q[0] = x[N // 2]
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), 1e-6)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), 1e-6)
    s, i = ops.cosine_topk(qn, xn, k, method="scan")
    check_topk_against_scores(s.cpu().numpy(), i.cpu().numpy(), ocos.cosine_scores(q, x, 1e-6), k)
    notes = []
    if N >= 64:  # two-stage: needs unit rows (row 0 is zero -> its proof guard may fail: device fallback repairs it)
        try:
            s2, i2 = ops.cosine_topk_two_stage(qn, xn, ops.index_shadow_f16(xn), k)
            assert torch.equal(i2, i) and torch.equal(s2, s), "two-stage differs from the scan"
            notes.append("ts")
        except NotImplementedError:  # a stated limit (D % 32), raised — never a silent difference
            notes.append("ts:unsupported")
    if Q >= 32 and N >= 4096 and k <= 500:
        try:
            s3, i3 = ops.cosine_topk_gemm(qn, xn, k, check_norms=False)
            assert torch.equal(i3, i) and torch.equal(s3, s), "GEMM-shaped pass differs from the scan"
            notes.append("gemm")
        except NotImplementedError:
            notes.append("gemm:unsupported")
    if N >= 97:  # three uneven shards merged == one pass
        cuts = [0, N // 5, N // 5 + (N // 2), N]
        parts = [ops.cosine_topk(qn, xn[cuts[j]: cuts[j + 1]].contiguous(), k, row_id_base=cuts[j]) for j in range(3)]
        ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        assert torch.equal(mi, i) and torch.equal(ms, s), "shard merge differs from the single pass"
        notes.append("shards")
    return f"N={N} D={D} Q={Q} k={k} " + "+".join(notes)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    for c in range(cases):
        print(c, one_case(rng, dev), flush=True)
    print("fuzz ok")


if __name__ == "__main__":
    main()
