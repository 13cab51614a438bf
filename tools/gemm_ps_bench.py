#!/usr/bin/env python3
"""A/B of the two split-bf16 GEMM kernels on the scorer's shapes, interleaved rounds in one process:
gemm_bf16x3.hip (f32 A split while staged through registers) vs gemm_ps.hip (A and W pre-split, LDS-DMA staging).
The pre-split planes are made once outside the timed region (in the scorer the producer kernels write them)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import _lib, ops


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    shapes = [(131072, 768, 768, "Wb s / state_net.4 (both directions of a 65536-edge chunk)"),
              (65536, 768, 768, "Wa p / Wc r_ctx (one chunk)"), (131072, 1024, 1024, "D = H = 1024"),
              (50787, 768, 768, "entity_proj / Wc node_repr")]
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    s = ops._stream(dev)
    for M, K, N, name in shapes:
        x = torch.randn((M, K), generator=g, device=dev)
        w = torch.randn((N, K), generator=g, device=dev) / K ** 0.5
        b = torch.randn((N,), generator=g, device=dev)
        out = torch.empty((M, N), device=dev)
        out2 = torch.empty((M, N), device=dev)
        Kp = (K + 31) // 32 * 32
        pl = torch.empty((2, M + N, Kp), dtype=torch.bfloat16, device=dev)
        _lib.check(lib.evi_split_rows_bf16(x.data_ptr(), M, K, K, Kp, pl[0, :M].data_ptr(), pl[1, :M].data_ptr(), s))
        _lib.check(lib.evi_split_rows_bf16(w.data_ptr(), N, K, K, Kp, pl[0, M:].data_ptr(), pl[1, M:].data_ptr(), s))
        ws = torch.empty(int(lib.evi_gemm_nt_bf16x3_workspace_bytes(N, K)), dtype=torch.uint8, device=dev)

        def old():
            _lib.check(lib.evi_gemm_nt_bf16x3(x.data_ptr(), M, K, K, w.data_ptr(), N, K, b.data_ptr(), 0, out.data_ptr(), N,
                                              ws.data_ptr(), ws.numel(), s))

        def new():
            _lib.check(lib.evi_gemm_nt_bf16x3_presplit(pl[0, :M].data_ptr(), pl[1, :M].data_ptr(), M, Kp, pl[0, M:].data_ptr(),
                                                       pl[1, M:].data_ptr(), N, b.data_ptr(), 0, out2.data_ptr(), N, s))

        res = {"staged": [], "presplit": []}
        for fn in (old, new):
            fn()
        torch.cuda.synchronize()
        same = bool(torch.equal(out, out2))
        for rnd in range(7):
            for key, fn in (("staged", old), ("presplit", new)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[key].append(e0.elapsed_time(e1) / 4)
        fl = 2.0 * M * N * K
        for key, v in res.items():
            v.sort()
            med, best = v[len(v) // 2], v[0]
            print(f"{name:62s} M={M} K={K} N={N} {key:9s} median {med:.3f} ms (min {best:.3f})  executed {3 * fl / med / 1e9:.0f} TF/s"
                  f"  frac of 2.5 PF {3 * fl / med / 1e9 / 2500:.3f}  identical={same}", flush=True)


if __name__ == "__main__":
    main()
