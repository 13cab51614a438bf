"""The C-ABI contract says calls never synchronise, allocate or copy to the host, so a step can be
captured into a hipGraph (include/evi_hip.h "Conventions").  Capture and replay prove it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cosine_topk_and_merge_replay_in_a_hip_graph(dev):
    from evi_rag_amd import _lib, ops

    N, D, Q, k = 150000, 256, 32, 200
    g = torch.Generator(device=dev).manual_seed(0)
    xn = ops.normalize_embeddings(torch.randn((N, D), generator=g, device=dev), 1e-6)
    q1 = ops.normalize_embeddings(torch.randn((Q, D), generator=g, device=dev), 1e-6)
    q2 = ops.normalize_embeddings(torch.randn((Q, D), generator=g, device=dev), 1e-6)
    ws = torch.empty(ops.cosine_topk_workspace_bytes(Q, N, D, k), dtype=torch.uint8, device=dev)
    ref1 = ops.cosine_topk(q1, xn, k, workspace=ws)
    ref2 = ops.cosine_topk(q2, xn, k, workspace=ws)
    q_static = q1.clone()
    out = (torch.empty((Q, k), dtype=torch.float32, device=dev), torch.empty((Q, k), dtype=torch.int64, device=dev))
    rec = int(_lib.load().evi_topk_packed_bytes(Q, k))
    packed = torch.empty(2 * rec, dtype=torch.uint8, device=dev)
    merged = None
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(stream):
        ops.cosine_topk(q_static, xn, k, workspace=ws, out=out)  # warm-up on the side stream (sets kernel attributes)
    torch.cuda.current_stream(dev).wait_stream(stream)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        half = N // 2
        v0 = ops.topk_packed_views(packed[:rec], Q, k)
        v1 = ops.topk_packed_views(packed[rec:], Q, k)
        ops.cosine_topk(q_static, xn[:half], k, workspace=ws, out=v0)
        ops.cosine_topk(q_static, xn[half:], k, row_id_base=half, workspace=ws, out=v1)
        merged = ops.topk_merge_packed(packed, 2, Q, k)
    for qs, ref in ((q1, ref1), (q2, ref2), (q1, ref1)):
        q_static.copy_(qs)
        graph.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(merged[1], ref[1]) and torch.equal(merged[0], ref[0])
