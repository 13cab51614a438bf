#!/usr/bin/env python3
"""What a library bf16 GEMM reaches on THIS box on random data (hipBLASLt through torch.matmul) — the calibration for the
scorer GEMM's fraction of the nominal 2.5 PFLOP/s: the chip lowers its clock under dense MFMA load
(MI355X_MICROARCH.md, DVFS give-back), so the nominal peak is not reachable by any kernel on random operands."""
import json
import time

import torch


def run(M, N, K, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        c = a @ b.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"M": M, "N": N, "K": K, "ms": dt * 1e3, "TFLOPs": 2.0 * M * N * K / dt / 1e12, "frac_of_2500": 2.0 * M * N * K / dt / 2.5e15}


if __name__ == "__main__":
    out = [run(8192, 8192, 8192), run(4096, 4096, 4096), run(131072, 768, 768), run(65536, 768, 768), run(131072, 1024, 1024)]
    # sustained: 2 s of the big one
    t0 = time.perf_counter()
    n = 0
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    while time.perf_counter() - t0 < 2.0:
        for _ in range(10):
            c = a @ b.t()
        torch.cuda.synchronize()
        n += 10
    dt = (time.perf_counter() - t0) / n
    out.append({"sustained_8192^3_TFLOPs": 2.0 * 8192 ** 3 / dt / 1e12})
    print(json.dumps(out, indent=1))
