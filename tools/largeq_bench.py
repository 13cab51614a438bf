#!/usr/bin/env python3
"""Many-query top-k: the 32-queries-per-pass scan vs the one-pass GEMM-shaped path, same index, same results.
usage: python tools/largeq_bench.py [rows] [Q] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import ops


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    D = 768
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.empty((N, D), dtype=torch.float32, device=dev)
    for lo in range(0, N, 1 << 20):
        hi = min(N, lo + (1 << 20))
        x[lo:hi] = ops.normalize_embeddings(torch.randn(hi - lo, D, device=dev, generator=g))
    q = ops.normalize_embeddings(torch.randn(Q, D, device=dev, generator=g))

    def timed(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters, out

    t_scan, (s0, i0) = timed(lambda: ops.cosine_topk(q, x, k))
    t_gemm, (s1, i1) = timed(lambda: ops.cosine_topk_gemm(q, x, k, fallback=False, products=3))
    t_g1, (s2, i2) = timed(lambda: ops.cosine_topk_gemm(q, x, k, fallback=False, products=1))
    shadow = ops.index_shadow_bf16(x)
    t_sh, (s3, i3) = timed(lambda: ops.cosine_topk_gemm(q, x, k, fallback=False, products=1, shadow=shadow))
    same = bool(torch.equal(i0, i1) and torch.equal(s0, s1) and torch.equal(i0, i2) and torch.equal(s0, s2)
                and torch.equal(i0, i3) and torch.equal(s0, s3))
    flops = 2.0 * N * Q * D
    print(f"N={N} Q={Q} k={k} D={D}: scan {t_scan * 1e3:.1f} ms ({Q / t_scan:.0f} q/s, {flops / t_scan / 1e12:.0f} TF/s f32-MFMA), "
          f"gemm {t_gemm * 1e3:.1f} ms ({Q / t_gemm:.0f} q/s, {flops / t_gemm / 1e12:.0f} TF/s algorithmic), "
          f"speedup {t_scan / t_gemm:.2f}x; plain-bf16 selection {t_g1 * 1e3:.1f} ms ({Q / t_g1:.0f} q/s), speedup {t_scan / t_g1:.2f}x; "
          f"with a bf16 shadow index {t_sh * 1e3:.1f} ms ({Q / t_sh:.0f} q/s), speedup {t_scan / t_sh:.2f}x; "
          f"identical results: {same}", flush=True)


if __name__ == "__main__":
    main()
