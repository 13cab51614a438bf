#!/usr/bin/env python3
"""Where does a block selection spend its time?  ops.segment_topk (one workgroup per list, the same block_topk as the
scan's candidate selection) over 32 lists of n keys, for several n and score distributions.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import ops


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for dist_name in ("cosine-like N(0, 0.036)", "uniform [0, 1)", "uniform bit patterns"):
        for n in (1024, 4096, 8192, 16384, 65536, 262144):
            B = 32
            if dist_name.startswith("cosine"):
                s = torch.randn(B * n, generator=g, device=dev) * 0.036
            elif dist_name.startswith("uniform ["):
                s = torch.rand(B * n, generator=g, device=dev)
            else:  # every byte of the key spread out: no hot radix digit in any pass
                bits = torch.randint(0, 2 ** 31 - 1, (B * n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
                s = (bits & 0x7F7FFFFF).view(torch.float32)
                s = torch.nan_to_num(s, nan=0.0, posinf=1.0, neginf=-1.0)
            ptr = (torch.arange(B + 1, device=dev, dtype=torch.int64) * n)
            for k in (500,):
                ops.segment_topk(s, ptr, k)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.segment_topk(s, ptr, k)
                e1.record()
                torch.cuda.synchronize()
                print(f"{dist_name:26s} n={n:7d} k={k}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per call (32 lists)", flush=True)


if __name__ == "__main__":
    main()
