"""world_size-2 `gloo` tests of the multi-GPU orchestration on CPU.

The HIP kernels cannot run here, so the per-shard top-k and the merge are the ORACLE's (the checker
standing in for the kernels); what is under test is the sharding, the fixed-shape all-gather, the
invariance of the merged result to the number of shards, and the metric-state all-reduce.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N, D, Q, K = 3000, 32, 5, 40


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _index():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, D)).astype(np.float32)
    x[100] = x[2500]  # an exact tie across shards: the lower global id must win
    q = rng.standard_normal((Q, D)).astype(np.float32)
    q[0] = x[2500]
    return x, q


def _oracle_local_topk(queries, shard, k, row_id_base):
    from oracle import cosine as ocos

    s, i, _ = ocos.dot_topk_prenormalized(queries.numpy(), shard.numpy(), k, row_id_base=row_id_base)
    return torch.from_numpy(s), torch.from_numpy(i)


def _oracle_merge(scores, ids):
    from oracle import ranking

    s, i = ranking.merge_topk(scores.numpy(), ids.numpy(), scores.shape[-1])
    return torch.from_numpy(s), torch.from_numpy(i)


def _worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from evi_rag_amd import dist as edist
        from evi_rag_amd.metrics import EdgeRecallAtK
        from oracle import cosine as ocos

        x, q = _index()
        xn, qn = ocos.normalize_embeddings(x, 1e-6), ocos.normalize_embeddings(q, 1e-6)
        b = edist.shard_bounds(N, world)
        idx = edist.ShardedIndex(torch.from_numpy(xn[b[rank]: b[rank + 1]]), N, local_topk=_oracle_local_topk,
                                 merge=_oracle_merge)
        assert (idx.row_begin, idx.row_end) == (b[rank], b[rank + 1])
        # one communicator per pipeline lane (a communicator must not be driven from two streams at once): lane 1 got a
        # group of its own over the same ranks, and a record exchanged on either reaches every rank in rank order
        assert idx._lane_groups[0] is None and idx._lane_groups[1] is not None
        assert dist.get_process_group_ranks(idx._lane_groups[1]) == list(range(world))
        for lane in (0, 1):
            mine = torch.full((16,), rank * 10 + lane, dtype=torch.uint8)
            everyone = torch.empty(world * 16, dtype=torch.uint8)
            idx._rccl_all_gather(everyone, mine, lane=lane)
            assert everyone.view(world, 16)[:, 0].tolist() == [r * 10 + lane for r in range(world)]
        s, i = idx.topk(torch.from_numpy(qn), K)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.numpy(), i=i.numpy())
        # pipeline-form agreement (a collective): both ranks got their lane-1 communicator -> two lanes stay
        assert idx.lane_fallback is None and idx.agree_on_lanes() is True and idx.two_lanes
        # ONE rank could not (here: rank 1 says so) -> EVERY rank drops to the one-communicator side-stream form, in-process
        if rank == 1:
            idx.two_lanes, idx.lane_fallback = False, "lane-1 communicator: RuntimeError: simulated on rank 1"
        assert idx.agree_on_lanes() is False and not idx.two_lanes and idx.lane_fallback
        # the communicator refuses on every rank (EVI_INJECT_LANE_FAILURE=group): no second group, fallback recorded
        os.environ["EVI_INJECT_LANE_FAILURE"] = "group"
        try:
            idx2 = edist.ShardedIndex(torch.from_numpy(xn[b[rank]: b[rank + 1]]), N, local_topk=_oracle_local_topk, merge=_oracle_merge)
        finally:
            del os.environ["EVI_INJECT_LANE_FAILURE"]
        assert not idx2.two_lanes and "injected" in idx2.lane_fallback and idx2._lane_groups[1] is None
        assert idx2.agree_on_lanes() is False
        s2, i2 = idx2.topk(torch.from_numpy(qn), K)
        assert torch.equal(i2, i) and torch.equal(s2, s)
        # metric states: each rank saw a different subset of graphs
        m = EdgeRecallAtK(k_values=[1, 10])
        graphs = edist.shard_graphs(7, rank, world)
        m._states["graph_count"] = float(len(graphs))
        m._states["recall_sum_at_1"] = float(sum(graphs))
        m.sync()
        assert m._states["graph_count"] == 7.0 and m._states["recall_sum_at_1"] == float(sum(range(7)))
        assert edist.all_reduce_sum_([1.0, rank]) == [float(world), float(sum(range(world)))]
        # bench.py's loop decisions: a warm-up loop whose steps are collectives must run the SAME number of steps on every rank —
        # `Ctx.agree` is the OR over the ranks.  Rank-local clocks that disagree (rank 0 wants 3 more rounds, rank 1 wants 5)
        # must yield the same count, the larger one, on both ranks.
        import importlib.util

        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        bench.REHEARSAL_BACKEND = "gloo"
        ctx = bench.Ctx(torch.device("cpu"), world, rank, None)
        want_local = 3 if rank == 0 else 5
        rounds = 0
        while ctx.agree(rounds < want_local):
            dist.all_reduce(torch.zeros(1))  # the "collective step" of the loop body
            rounds += 1
        assert rounds == 5
        assert ctx.agree(False) is False and ctx.agree(rank == 1) is True
        # ragged gather used by the top-k artifact writer: rank r contributes r + 2 rows
        from evi_rag_amd.topk_writer import gather_padded

        parts = gather_padded(torch.full((rank + 2, 3), float(rank)))
        assert [tuple(p.shape) for p in parts] == [(r + 2, 3) for r in range(world)]
        assert all(bool((p == r).all()) for r, p in enumerate(parts))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_topk_and_metric_sync_world2(tmp_path):
    from oracle import cosine as ocos

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    x, q = _index()
    xn, qn = ocos.normalize_embeddings(x, 1e-6), ocos.normalize_embeddings(q, 1e-6)
    ref_s, ref_i, _ = ocos.dot_topk_prenormalized(qn, xn, K)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["i"], ref_i), f"rank {r}: merged ids differ from the single-shard result"
        assert np.array_equal(z["s"], ref_s)
    assert ref_i[0, 0] == 100 and ref_i[0, 1] == 2500  # tie across shards resolved by global row id


def test_shard_helpers():
    from evi_rag_amd import dist as edist

    assert edist.shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert edist.shard_bounds(1 << 23, 8)[-1] == 1 << 23
    got = sorted(g for r in range(3) for g in edist.shard_graphs(10, r, 3))
    assert got == list(range(10))
    idx = edist.ShardedIndex(torch.zeros(4, 8), 4, local_topk=_oracle_local_topk, merge=_oracle_merge)
    assert idx.world == 1 and (idx.row_begin, idx.row_end) == (0, 4)
    with pytest.raises(ValueError, match="must hold rows"):
        edist.ShardedIndex(torch.zeros(3, 8), 4)


def test_sharded_index_two_stage_needs_its_shadow():
    """Host-side contract of method="two_stage" (no GPU needed to refuse bad arguments)."""
    import pytest
    import torch

    from evi_rag_amd.dist import ShardedIndex

    rows = torch.zeros((8, 32))
    with pytest.raises(ValueError, match="float16 shadow"):
        ShardedIndex(rows, 8, method="two_stage")
    with pytest.raises(ValueError, match="float16 shadow"):
        ShardedIndex(rows, 8, method="two_stage", shadow=rows.to(torch.bfloat16))
    with pytest.raises(ValueError, match="float16 shadow"):
        ShardedIndex(rows, 8, method="two_stage", shadow=rows.to(torch.float16), row_scale=torch.ones(8))
    idx = ShardedIndex(rows, 8, method="two_stage", shadow=rows.to(torch.float16))
    assert idx.two_stage_status is not None and idx.two_stage_status.dtype == torch.int32 and not idx.two_stage_failed()
