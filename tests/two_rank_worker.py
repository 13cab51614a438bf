"""One rank of the two-ranks-on-one-GPU test of the sharded top-k (started by tests/test_dist_two_rank_gpu.py).

Each rank owns half the rows of the index and runs the REAL kernels on its half (evi_cosine_topk /
evi_cosine_topk_two_stage into the packed record, evi_topk_merge_packed after the exchange).  RCCL refuses two
ranks on one device, so the [Q, k] record exchange — and only that — is injected: a `gloo` all-gather of host-staged
buffers (`ShardedIndex(exchange=...)`).  Everything else is the code path `bench.py --gpus N` runs.

usage: two_rank_worker.py RANK WORLD PORT OUT_DIR
"""
import datetime
import os
import sys

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO_ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N, D, Q, K, STEPS = 300_001, 256, 32, 100, 7  # an odd row count: uneven shards


def build_index(dev, clustered_tail: bool):
    """The same index on every rank (seeded on the device).  Exact duplicates ACROSS the shard boundary exercise the
    global-id tie-break of the merge; `clustered_tail` plants 6 000 nearly identical rows in the upper shard so that
    the two-stage proof fails there and only there."""
    from evi_rag_amd import ops

    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(N, D, device=dev, generator=g)
    x[0] = 0.0
    x[N - 5] = x[17]            # ties across shards: rank 0's row must come first
    x[N // 2 + 3] = x[N // 2 - 3]
    if clustered_tail:
        base = torch.randn(D, device=dev, generator=g)
        x[N - 7000: N - 1000] = base + 1e-5 * torch.randn(6000, D, device=dev, generator=g)
    qs = torch.randn(STEPS, Q, D, device=dev, generator=g)
    qs[0, 0] = x[17]
    qs[1, 3] = x[N // 2 - 3]
    if clustered_tail:
        qs[:, 5] = x[N - 3000]  # every batch has a query that lands in the cluster
    xn = ops.normalize_embeddings(x)
    qn = ops.normalize_embeddings(qs.view(-1, D)).view(STEPS, Q, D)
    return xn, qn


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from evi_rag_amd import ops
    from evi_rag_amd.dist import ShardedIndex, shard_bounds

    calls = {"n": 0}

    def host_exchange(all_records, local_record):
        # ordered on the current stream: wait for the scan, gather through host memory, copy back on the same stream
        torch.cuda.current_stream(dev).synchronize()
        host = local_record.cpu()
        gathered = torch.empty(world * host.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(gathered, host)
        all_records.copy_(gathered)
        calls["n"] += 1

    res = {}
    try:
        b = shard_bounds(N, world)
        for tag, clustered in (("plain", False), ("clustered", True)):
            xn, qn = build_index(dev, clustered)
            shard = xn[b[rank]: b[rank + 1]].contiguous()
            want = [ops.cosine_topk(qn[s], xn, K) for s in range(STEPS)]  # the one-rank result, same kernels
            res[f"{tag}_want_s"] = torch.stack([w[0] for w in want]).cpu().numpy()
            res[f"{tag}_want_i"] = torch.stack([w[1] for w in want]).cpu().numpy()

            def run(idx, name, lanes):
                idx.two_lanes = lanes
                n0 = calls["n"]
                s, i = idx.topk(qn[0], K)
                res[f"{tag}_{name}_sync_s"], res[f"{tag}_{name}_sync_i"] = s.cpu().numpy(), i.cpu().numpy()
                out_s, out_i = [], []
                pending = []
                for st in range(STEPS):  # free-running pipeline: results stay valid for two further calls
                    pending.append(idx.topk_async(qn[st], K))
                    if len(pending) == 2:
                        ps, pi, ev = pending.pop(0)
                        ev.synchronize()
                        out_s.append(ps.clone())
                        out_i.append(pi.clone())
                for ps, pi, ev in pending:
                    ev.synchronize()
                    out_s.append(ps.clone())
                    out_i.append(pi.clone())
                torch.cuda.synchronize(dev)
                res[f"{tag}_{name}_async_s"] = torch.stack(out_s).cpu().numpy()
                res[f"{tag}_{name}_async_i"] = torch.stack(out_i).cpu().numpy()
                assert calls["n"] - n0 == 1 + STEPS, "one exchange per batch"

            scan = ShardedIndex(shard, N, exchange=host_exchange)
            assert scan.world == world and scan._exchange and (scan.row_begin, scan.row_end) == (b[rank], b[rank + 1])
            run(scan, "scan_lanes", True)
            run(scan, "scan_side", False)
            ts = ShardedIndex(shard, N, method="two_stage", shadow=ops.index_shadow_f16(shard), exchange=host_exchange)
            run(ts, "ts_lanes", True)
            local_flag = int(ts.two_stage_status.item())  # this rank's own view, before the collective
            res[f"{tag}_ts_local_flag"] = np.int64(local_flag)
            res[f"{tag}_ts_failed"] = np.int64(ts.two_stage_failed())  # collective: max over ranks
            run(ts, "ts_side", False)
            ts.two_stage_failed()
            del xn, qn, shard, scan, ts
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
