#!/usr/bin/env python3
"""TN (weight-gradient) split-bf16 GEMM against the NT kernel on the same flops: C [M, N] = A^T B, A [K, M], B [K, N].
EVI_TN_SLICES=<n> overrides the slice count."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evi_rag_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    out = []
    for K, M, N in [(65536, 768, 768), (131072, 768, 768), (131072, 768, 20), (50787, 768, 768), (65536, 1024, 1024)]:
        a = torch.randn(K, M, device=dev)
        b = torch.randn(K, N, device=dev)
        t_tn = timeit(lambda: ops.gemm_tn(a, b))
        # the NT kernel on the transposed problem size: [K, M] x [N, M]^T has the same flops when N == M
        w = torch.randn(N, M, device=dev)
        t_nt = timeit(lambda: ops.linear_act(a, w, None, mode="bf16x3"))
        flops = 3 * 2.0 * K * M * N
        out.append({"K": K, "M": M, "N": N, "tn_ms": t_tn * 1e3, "tn_TFs_executed": flops / t_tn / 1e12,
                    "tn_frac_of_2500": flops / t_tn / 2.5e15, "nt_same_flops_ms": t_nt * 1e3,
                    "operand_GBps": (K * (M + N) * 4) / t_tn / 1e9})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
