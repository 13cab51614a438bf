"""GPU: the multi-rank exchange path (packed all-gather over RCCL + merge, pipelined on a side stream),
rehearsed with a ONE-rank process group on the single GPU of the test box."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_pipelined_exchange_one_rank_group(dev, monkeypatch):
    from evi_rag_amd import ops
    from evi_rag_amd.dist import ShardedIndex

    N, D, Q, k, steps = 300000, 128, 32, 100, 9
    g = torch.Generator(device=dev).manual_seed(2)
    xn = ops.normalize_embeddings(torch.randn(N, D, device=dev, generator=g))
    qs = [ops.normalize_embeddings(torch.randn(Q, D, device=dev, generator=g)) for _ in range(steps)]
    want = [ops.cosine_topk(q, xn, k) for q in qs]
    monkeypatch.setenv("EVI_FORCE_EXCHANGE", "1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        idx = ShardedIndex(xn, N)
        assert idx._exchange
        # synchronous form: scan -> all-gather -> merge on the caller's stream
        s, i = idx.topk(qs[0], k)
        assert torch.equal(i, want[0][1]) and torch.equal(s, want[0][0])
        # pipelined form, read back step by step
        for b in range(steps):
            s, i, ev = idx.topk_async(qs[b], k)
            ev.synchronize()
            assert torch.equal(i, want[b][1]) and torch.equal(s, want[b][0]), b
        # pipelined form, free-running: only the last two results are still live
        outs = [idx.topk_async(qs[b], k) for b in range(steps)]
        torch.cuda.synchronize(dev)
        for b in (steps - 2, steps - 1):
            assert torch.equal(outs[b][1], want[b][1]) and torch.equal(outs[b][0], want[b][0]), b
        assert outs[steps - 1][0].data_ptr() == outs[steps - 3][0].data_ptr()  # two alternating slots
        # the same with one main stream + a side stream for the exchange instead of two pipeline lanes
        assert idx.two_lanes
        idx.two_lanes = False
        outs = [idx.topk_async(qs[b], k) for b in range(steps)]
        torch.cuda.synchronize(dev)
        for b in (steps - 2, steps - 1):
            assert torch.equal(outs[b][1], want[b][1]) and torch.equal(outs[b][0], want[b][0]), b
        idx.two_lanes = True
        # the same through the two-stage exact scan (f16 shadow selects, f32 rows re-score): identical results, the
        # proof flag is sticky across the stream of batches and read once
        idx2 = ShardedIndex(xn, N, method="two_stage", shadow=ops.index_shadow_f16(xn))
        s, i = idx2.topk(qs[1], k)
        assert torch.equal(i, want[1][1]) and torch.equal(s, want[1][0])
        outs = [idx2.topk_async(qs[b], k) for b in range(steps)]
        torch.cuda.synchronize(dev)
        assert not idx2.two_stage_failed()
        for b in (steps - 2, steps - 1):
            assert torch.equal(outs[b][1], want[b][1]) and torch.equal(outs[b][0], want[b][0]), b
    finally:
        dist.destroy_process_group()
