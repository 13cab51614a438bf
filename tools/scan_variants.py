#!/usr/bin/env python3
"""A/B the scan-kernel tunables (EVI_SCAN_THREADS / EVI_SCAN_NT / EVI_SCAN_U) in ONE process,
interleaved rounds, same index (cdna_hip_programming.md §5.4 rule 24).  GPU only."""
import argparse
import ctypes
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import _lib, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 22)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--k", type=int, default=500)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--threads", default="512,1024")
    ap.add_argument("--nt", default="0,1")
    ap.add_argument("--u", default="8,4")
    ap.add_argument("--f16", action="store_true", help="store the index as f16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    x = torch.empty((args.rows, args.dim), device=dev)
    step = 1 << 18
    for r0 in range(0, args.rows, step):
        x[r0:r0 + step] = torch.randn((min(step, args.rows - r0), args.dim), generator=g, device=dev)
    ops.normalize_embeddings(x, 1e-6, out=x)
    if args.f16:
        x = x.to(torch.float16)
    q = ops.normalize_embeddings(torch.randn((args.queries, args.dim), generator=g, device=dev), 1e-6)
    variants = list(itertools.product([int(v) for v in args.threads.split(",")],
                                      [int(v) for v in args.nt.split(",")],
                                      [int(v) for v in args.u.split(",")]))
    res = {v: [] for v in variants}
    ref = None
    nbytes = args.rows * args.dim * (2 if args.f16 else 4)
    for rnd in range(args.rounds + 1):
        for v in variants:
            os.environ["EVI_SCAN_THREADS"], os.environ["EVI_SCAN_NT"], os.environ["EVI_SCAN_U"] = map(str, v)
            lib.evi_timing_enable(1)
            s, i = ops.cosine_topk(q, x, args.k)
            torch.cuda.synchronize()
            lib.evi_timing_enable(0)
            ms = (ctypes.c_double * 2)()
            ln = (ctypes.c_int32 * 2)()
            lib.evi_timing_read(ms, ln, 2)
            if ref is None:
                ref = (s.clone(), i.clone())
            else:
                assert torch.equal(i, ref[1]) and torch.equal(s, ref[0]), f"variant {v} changed the result"
            if rnd > 0:
                res[v].append((ms[0], ms[1]))
    print(f"rows={args.rows} dim={args.dim} Q={args.queries} k={args.k}  ({nbytes / 1e9:.2f} GB/scan)")
    for v in variants:
        sc = sorted(m[0] for m in res[v])
        se = sorted(m[1] for m in res[v])
        med = sc[len(sc) // 2]
        print(f"threads={v[0]:5d} nt={v[1]} U={v[2]}: scan median {med:.3f} ms min {sc[0]:.3f} ms "
              f"-> {nbytes / med / 1e6:.0f} GB/s (best {nbytes / sc[0] / 1e6:.0f}); select median {se[len(se) // 2]:.3f} ms",
              flush=True)


if __name__ == "__main__":
    main()
