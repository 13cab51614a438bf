#!/usr/bin/env python3
"""Runs one scorer GEMM shape repeatedly (for rocprofv3 counter passes).  usage: gemm_one.py [M K N] [mode] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import ops

M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (131072, 2308, 768)
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, K), generator=g, device=dev)
w = torch.randn((N, K), generator=g, device=dev) / K ** 0.5
b = torch.randn((N,), generator=g, device=dev)
for _ in range(iters):
    ops.linear_act(x, w, b, None, mode=mode)
torch.cuda.synchronize()
print("done")
