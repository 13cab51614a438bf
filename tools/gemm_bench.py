#!/usr/bin/env python3
"""Times the scorer GEMM shapes (f32-exact and split-bf16 paths) on one GPU, interleaved rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import ops


MODES = ("bf16x3", "f32")


def main():
    dev = torch.device("cuda:0")
    shapes = [(131072, 2308, 768, "state_net.0 (one 65536-edge chunk, both directions)"),
              (131072, 768, 768, "state_net.4 / Wb s (both directions)"), (65536, 768, 768, "Wa p / Wc r_ctx (one chunk)"),
              (50787, 768, 768, "entity_proj / Wc node_repr")]
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    for M, K, N, name in shapes:
        x = torch.randn((M, K), generator=g, device=dev)
        w = torch.randn((N, K), generator=g, device=dev) / K ** 0.5
        b = torch.randn((N,), generator=g, device=dev)
        res = {}
        for mode in MODES:
            ops.linear_act(x, w, b, None, mode=mode)
        torch.cuda.synchronize()
        for rnd in range(5):
            for mode in MODES:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.linear_act(x, w, b, None, mode=mode)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(mode, []).append(e0.elapsed_time(e1))
        fl = 2.0 * M * N * K
        for mode, v in res.items():
            v.sort()
            med = v[len(v) // 2]
            mult = 3.0 if mode.startswith("bf16x3") else 1.0
            print(f"{name:55s} M={M} K={K} N={N} {mode:7s} median {med:.3f} ms  algorithmic {fl / med / 1e9:.0f} TF/s  "
                  f"executed {mult * fl / med / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
