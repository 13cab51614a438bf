#!/usr/bin/env python3
"""One many-query top-k configuration run a few times (for rocprofv3).  usage: largeq_one.py [products] [shadow 0|1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import ops

products = int(sys.argv[1]) if len(sys.argv) > 1 else 1
use_shadow = len(sys.argv) > 2 and sys.argv[2] == "1"
N, Q, D, k = 1 << 23, 512, 768, 500
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.empty((N, D), dtype=torch.float32, device=dev)
for lo in range(0, N, 1 << 20):
    x[lo:lo + (1 << 20)] = ops.normalize_embeddings(torch.randn(1 << 20, D, device=dev, generator=g))
q = ops.normalize_embeddings(torch.randn(Q, D, device=dev, generator=g))
shadow = ops.index_shadow_bf16(x) if use_shadow else None
for _ in range(4):
    ops.cosine_topk_gemm(q, x, k, fallback=False, products=products, shadow=shadow)
torch.cuda.synchronize()
print("done")
