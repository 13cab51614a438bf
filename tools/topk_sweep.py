#!/usr/bin/env python3
"""Top-k scan rates over a grid of shapes and storage types, one process, same index per (N, D): ms per batch of Q
queries, algorithmic TB/s of the scan kernel (hipExt kernel events), and whether every method returns what it should
(two_stage == f32 scan bit for bit).  GPU only; writes one JSON document to stdout.

  python tools/topk_sweep.py > profiles/rNN_topk_shape_sweep.json
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import _lib, ops


def timed(lib, dev, fn, iters):
    fn()
    torch.cuda.synchronize(dev)
    lib.evi_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / iters * 1e3
    lib.evi_timing_enable(0)
    ms = (ctypes.c_double * 2)()
    ln = (ctypes.c_int32 * 2)()
    lib.evi_timing_read(ms, ln, 2)
    return wall, ms[0] / iters, ms[1] / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="384,768,1024")
    ap.add_argument("--rows", default="1048576,4194304,8388608")
    ap.add_argument("--ks", default="100,500")
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    rows_out = []
    for D in [int(v) for v in a.dims.split(",")]:
        for N in [int(v) for v in a.rows.split(",")]:
            x = torch.empty((N, D), device=dev)
            for r0 in range(0, N, 1 << 18):
                x[r0:r0 + (1 << 18)] = torch.randn((min(1 << 18, N - r0), D), generator=g, device=dev)
            ops.normalize_embeddings(x, 1e-6, out=x)
            q = ops.normalize_embeddings(torch.randn((a.queries, D), generator=g, device=dev), 1e-6)
            x16 = ops.index_shadow_f16(x)
            x8, s8 = ops.quantize_rows_fp8(x)
            for k in [int(v) for v in a.ks.split(",")]:
                ref = None
                for method in ("f32 scan", "two_stage", "f16 index", "fp8 index"):
                    if method == "f32 scan":
                        fn, nbytes = (lambda: ops.cosine_topk(q, x, k)), N * D * 4
                    elif method == "two_stage":
                        fn, nbytes = (lambda: ops.cosine_topk_two_stage(q, x, x16, k, fallback=False)), N * D * 2
                    elif method == "f16 index":
                        fn, nbytes = (lambda: ops.cosine_topk(q, x16, k)), N * D * 2
                    else:
                        fn, nbytes = (lambda: ops.cosine_topk(q, x8, k, row_scale=s8)), N * D + N * 4
                    wall, scan_ms, sel_ms, out = timed(lib, dev, fn, a.iters)
                    rec = {"D": D, "N": N, "k": k, "Q": a.queries, "method": method, "ms_per_batch": round(wall, 4),
                           "scan_kernel_ms": round(scan_ms, 4), "select_ms": round(sel_ms, 4),
                           "scan_TBps": round(nbytes / (scan_ms * 1e-3) / 1e12, 3), "queries_per_s": round(a.queries / wall * 1e3, 1)}
                    if method == "f32 scan":
                        ref = out
                    elif method == "two_stage":
                        rec["identical_to_f32_scan"] = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
                    else:
                        inter = (out[1].unsqueeze(2) == ref[1].unsqueeze(1)).any(dim=2).float().mean().item()
                        rec["overlap_at_k_vs_f32"] = round(inter, 4)
                    rows_out.append(rec)
                    print(json.dumps(rec), file=sys.stderr, flush=True)
            del x, x16, x8, s8
            torch.cuda.empty_cache()
    print(json.dumps({"what": "cosine top-k sweep, one MI355X, Q queries per batch, exact top-k", "rows": rows_out}, indent=1))


if __name__ == "__main__":
    main()
