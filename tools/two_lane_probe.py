#!/usr/bin/env python3
"""Does running two batch chains on two hardware queues hide the selections?  Batches alternate between two streams (one
normal-, one high-priority: separate HIP hardware queues), each with its own workspace; EVI_SCAN_DYNAMIC switches the scan
between the fixed tile stride per wave and the dynamic hand-out.  Prints ms per batch for one lane / two lanes x static /
dynamic, for the f32 scan and the two-stage scan; every variant's results are compared with the one-lane static run.
GPU only."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import _lib, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="8388608,1048576")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batches", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    Q, k, D = 32, 500, a.dim
    for N in [int(v) for v in a.rows.split(",")]:
        x = torch.empty((N, D), device=dev)
        for r0 in range(0, N, 1 << 18):
            x[r0:r0 + (1 << 18)] = torch.randn((min(1 << 18, N - r0), D), generator=g, device=dev)
        ops.normalize_embeddings(x, 1e-6, out=x)
        shadow = ops.index_shadow_f16(x)
        qs = [ops.normalize_embeddings(torch.randn((Q, D), generator=g, device=dev), 1e-6) for _ in range(a.batches)]
        lanes = [torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)]
        ws = [torch.empty(ops.cosine_topk_workspace_bytes(Q, N, D, k), dtype=torch.uint8, device=dev) for _ in range(2)]
        ws2 = [torch.empty(int(lib.evi_cosine_topk_two_stage_workspace_bytes(Q, N, D, k)), dtype=torch.uint8, device=dev) for _ in range(2)]
        flags = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
        outs = [[(torch.empty((Q, k), device=dev), torch.empty((Q, k), dtype=torch.int64, device=dev)) for _ in range(a.batches)]
                for _ in range(2)]
        ref = {}
        for method in ("f32 scan", "two_stage"):
            for nlanes in (1, 2):
                for dyn in ("0", "1"):
                    os.environ["EVI_SCAN_DYNAMIC"] = dyn
                    best = float("inf")
                    for rep in range(3):
                        torch.cuda.synchronize(dev)
                        for s in lanes:
                            s.wait_stream(torch.cuda.current_stream(dev))
                        t0 = time.perf_counter()
                        for b in range(a.batches):
                            lane = b % nlanes
                            with torch.cuda.stream(lanes[lane]):
                                if method == "f32 scan":
                                    ops.cosine_topk(qs[b], x, k, workspace=ws[lane], out=outs[0][b])
                                else:
                                    ops.cosine_topk_two_stage(qs[b], x, shadow, k, status=flags[lane], workspace=ws2[lane], out=outs[1][b])
                        torch.cuda.synchronize(dev)
                        best = min(best, (time.perf_counter() - t0) / a.batches * 1e3)
                    res = outs[0] if method == "f32 scan" else outs[1]
                    key = method
                    if key not in ref:
                        ref[key] = [(s.clone(), i.clone()) for s, i in res]
                        same = True
                    else:
                        same = all(torch.equal(s, rs) and torch.equal(i, ri) for (s, i), (rs, ri) in zip(res, ref[key]))
                    print(f"N={N} {method:9s} lanes={nlanes} dynamic={dyn}: {best:.4f} ms per batch ({Q / best * 1e3:.0f} queries/s) "
                          f"same_as_reference={same} flags={[int(f.item()) for f in flags]}", flush=True)
        del x, shadow, ws, ws2, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
