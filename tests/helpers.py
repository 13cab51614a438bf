"""Shared checkers for the parity tests."""
from __future__ import annotations

import numpy as np


def check_topk_against_scores(hip_scores, hip_ids, ref_scores_full, k, *, id_base=0, score_tol=1e-5,
                              tie_eps=2e-6):
    """Margin-aware exact-set check of a top-k result against the oracle's full score matrix.

    hip_scores/hip_ids: [Q, k] from the HIP path; ref_scores_full: [Q, N] oracle scores.
      1. every returned score is within score_tol of the oracle's score for that row id;
      2. the returned ids are exactly the oracle's top-k set, except that rows whose oracle score
         is within tie_eps of the oracle's k-th score may swap (fp reassociation near-ties);
      3. the list is ordered (HIP score desc, id asc) and ids are unique; padding is (-inf, -1).
    """
    hip_scores = np.asarray(hip_scores)
    hip_ids = np.asarray(hip_ids)
    Q, N = ref_scores_full.shape
    m = min(k, N)
    for q in range(Q):
        ids = hip_ids[q, :m] - id_base
        sc = hip_scores[q, :m]
        assert np.all(hip_ids[q, m:] == -1), f"q={q}: padding ids"
        assert np.all(np.isneginf(hip_scores[q, m:])), f"q={q}: padding scores"
        assert ids.min(initial=0) >= 0 and ids.max(initial=0) < N, f"q={q}: id out of range"
        assert np.unique(ids).shape[0] == m, f"q={q}: duplicate ids"
        ref_at = ref_scores_full[q, ids]
        assert np.max(np.abs(ref_at - sc), initial=0.0) <= score_tol, (
            f"q={q}: score mismatch {np.max(np.abs(ref_at - sc))}")
        # order: (score desc, id asc)
        ds = np.diff(sc)
        assert np.all(ds <= 0), f"q={q}: scores not descending"
        same = ds == 0
        assert np.all(np.diff(ids)[same] > 0), f"q={q}: equal scores not in ascending id order"
        # exact set up to near-ties at the boundary
        order = np.lexsort((np.arange(N), -ref_scores_full[q].astype(np.float64)))
        ref_set = set(order[:m].tolist())
        got_set = set(ids.tolist())
        if ref_set != got_set:
            kth = ref_scores_full[q, order[m - 1]]
            for r in ref_set ^ got_set:
                assert abs(float(ref_scores_full[q, r]) - float(kth)) <= tie_eps, (
                    f"q={q}: row {r} (score {ref_scores_full[q, r]}) differs from oracle set, k-th={kth}")


def report(name: str, **values) -> None:
    """Measured figures a parity test wants on record (counts of near-tie flips, worst errors): appended as one JSON line to
    gpurun_out/test_report.jsonl when that directory exists (the GPU box's scratch output, merged back by gpurun); a no-op
    elsewhere.  Never part of a verdict — the asserts are."""
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "test_report.jsonl"), "a") as fh:
            fh.write(json.dumps({"test": name, **values}) + "\n")
