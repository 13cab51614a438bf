"""GPU: the sharded top-k with TWO ranks and the real kernels (replaces the gather at
src/callbacks/retriever_topk_edge_writer.py:450-462).

Two processes share the one GPU of the test box, each owning half the rows: per-shard evi_cosine_topk (or the two-stage
scan) into the packed record, ONE exchange of the [Q, k] records per batch, evi_topk_merge_packed on every rank.  RCCL
refuses two ranks per device, so the exchange is injected as a `gloo` all-gather of host-staged buffers
(tests/two_rank_worker.py); the rest is the code path of `bench.py --gpus N`.  Required: merged ids AND scores equal the
one-rank result bit for bit, on both ranks, for `topk`, the two-lane and the side-stream form of `topk_async`, the scan and
the two-stage method — also when one rank's two-stage proof fails (its shard holds a cluster of near-identical rows): the
device-side fallback repairs that rank's record before the exchange and `two_stage_failed()` reports it on EVERY rank.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


sys.path.insert(0, HERE)
from _procs import free_port as _free_port, run_ranks  # noqa: E402


@pytest.mark.timeout(900)
def test_two_ranks_real_kernels_equal_one_rank_bit_for_bit(dev, tmp_path):
    world = 2
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    run_ranks(lambda r, port: [sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), str(world), str(port), str(tmp_path)],
              world, env=env, timeout=600)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for tag in ("plain", "clustered"):
        want_s, want_i = z[0][f"{tag}_want_s"], z[0][f"{tag}_want_i"]
        assert np.array_equal(want_i, z[1][f"{tag}_want_i"]) and np.array_equal(want_s, z[1][f"{tag}_want_s"])
        for r in range(world):
            for name in ("scan_lanes", "scan_side", "ts_lanes", "ts_side"):
                assert np.array_equal(z[r][f"{tag}_{name}_sync_i"], want_i[0]), (tag, name, r)
                assert np.array_equal(z[r][f"{tag}_{name}_sync_s"], want_s[0]), (tag, name, r)
                assert np.array_equal(z[r][f"{tag}_{name}_async_i"], want_i), (tag, name, r)
                assert np.array_equal(z[r][f"{tag}_{name}_async_s"], want_s), (tag, name, r)
    # the tie across the shard boundary resolves by global row id (row 17 on rank 0 before its copy on rank 1)
    wi = z[0]["plain_want_i"]
    assert wi[0, 0, 0] == 17 and wi[0, 0, 1] == 300_001 - 5
    # proof flags: nothing failed on the plain index; on the clustered one only the upper shard's proof fails, and the
    # collective read reports it on both ranks
    assert [int(z[r]["plain_ts_failed"]) for r in range(world)] == [0, 0]
    assert [int(z[r]["clustered_ts_local_flag"]) for r in range(world)] == [0, 1]
    assert [int(z[r]["clustered_ts_failed"]) for r in range(world)] == [1, 1]


@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks(dev):
    """`python bench.py --gpus 1` through the launcher path and `--gpus 2` without one: the second must start its own
    ranks (the parent stays GPU-free) — on a one-GPU box the two ranks cannot both get a device, so what is checked there
    is that the parent relays a rank failure as a non-zero exit instead of dying at argument parsing."""
    import json

    import torch

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    root = os.path.dirname(HERE)
    small = ["--steps", "3", "--warmup", "1", "--rows", "200000", "--dim", "128", "--k", "50", "--no-cpu-baseline",
             "--no-graph-eval", "--no-encode", "--no-extra-legs"]
    # one rank under torch.distributed.run: world 1, no exchange — the same path as the plain run
    r1 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1"] + small,
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    line = json.loads([ln for ln in r1.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["hits_at_k"]["hits@50"] == 1.0 and line["sorted_ok"]
    if torch.cuda.device_count() >= 2:
        r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + small, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r2.returncode == 0, r2.stderr.decode()[-3000:]
        line = json.loads([ln for ln in r2.stdout.decode().splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["hits_at_k"]["hits@50"] == 1.0
    else:
        r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + small, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r2.returncode != 0  # rank 1 has no device: reported, not hung
        assert b"launch multi-GPU runs with torch.distributed.run" not in r2.stderr


@pytest.mark.timeout(900)
def test_bench_two_rank_control_flow_on_one_gpu(dev):
    """The whole `bench.py --gpus 2` flow under torch.distributed.run with both ranks on the one GPU (EVI_BENCH_BACKEND=gloo:
    gloo process group, host-staged record exchange — RCCL refuses two ranks per device): row sharding of the index, the
    query batch assembled by an all-reduce, `topk_async` lanes with the exchange, barriers, max-over-ranks timing, the
    two-stage leg, ONE JSON line from rank 0 — and the merged result must still find every planted row."""
    import json

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["EVI_BENCH_BACKEND"] = "gloo"
    root = os.path.dirname(HERE)
    small = ["--steps", "4", "--warmup", "2", "--rows", "300001", "--dim", "128", "--k", "50", "--no-cpu-baseline",
             "--no-graph-eval", "--no-encode", "--no-extra-legs"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2"] + small,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "rows/2" and line["scaling"] == "strong"
    assert line["hits_at_k"]["hits@1"] == 1.0 and line["hits_at_k"]["hits@50"] == 1.0 and line["sorted_ok"]
    assert line["value"] > 0 and line["two_stage"]["identical_to_f32_scan"] and not line["two_stage"]["proof_failed"]
    assert line["ranks_seen"] == 2 and [d["rank"] for d in line["devices"]] == [0, 1]
    assert "two lanes" in line["config"]["pipeline"]


def _rehearse(extra_env, args):
    import json

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["EVI_BENCH_BACKEND"] = "gloo"
    env.update(extra_env)
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2"] + args,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_two_rank_config4_and_config5_sharded_legs(dev):
    """An N > 1 run must measure the BASELINE scaling target, not only the config-2 index: the `config4_sharded` leg
    (configs[3]: f16 index row-sharded over the ranks, Q = 32 and Q = 512, ONE all-gather of packed records + merge) and the
    `config5_sharded` leg (configs[4]: e4m3 index, native fp8 MFMA) — here with a miniature row count, two ranks on the one
    GPU, gloo + host-staged exchange (replaces src/callbacks/retriever_topk_edge_writer.py:450-462)."""
    line = _rehearse({}, ["--steps", "4", "--warmup", "2", "--rows", "200001", "--dim", "128", "--k", "50", "--no-cpu-baseline",
                          "--no-graph-eval", "--no-encode", "--config4-rows", "300001"])
    for key, D, dt in (("config4_sharded", 768, "f16"), ("config5_sharded", 1024, "fp8")):
        leg = line[key]
        assert leg["ranks_seen"] == 2 and leg["n_gpus"] == 2 and leg["config"]["sharding"] == "rows/2"
        assert leg["config"]["index_rows"] == 300001 and leg["config"]["dim"] == D and leg["config"]["index_dtype"] == dt
        assert [d["rows"] for d in leg["devices"]] == [[0, 150000], [150000, 300001]]
        assert leg["hits_at_k"]["hits@1"] == 1.0 and leg["hits_at_k"]["hits@50"] == 1.0 and leg["sorted_ok"]
        assert leg["value"] > 0 and leg["roofline"]["bound"] == "hbm" and leg["roofline"]["achieved"] > 0
        assert "two lanes" in leg["config"]["pipeline"]
    many = line["config4_sharded"]["many_query"]
    assert many["queries_per_step"] == 512 and many["value"] > 0 and many["planted_row_in_top10"] == 1.0 and many["sorted_ok"]
    assert "native fp8 MFMA" in line["config5_sharded"]["dtype"]


@pytest.mark.timeout(900)
def test_bench_two_rank_graph_eval_sharded_leg(dev):
    """N > 1 also measures the per-question stage sharded by question (graphs never span ranks, counters summed
    with one all-reduce): every rank evaluates its own split, the line carries the aggregate."""
    line = _rehearse({}, ["--steps", "3", "--warmup", "1", "--rows", "100001", "--dim", "64", "--k", "20", "--no-cpu-baseline",
                          "--no-encode", "--config4-rows", "100001", "--eval-shard-questions", "64"])
    ge = line["graph_eval_sharded"]
    assert "skipped" not in ge, ge
    assert ge["n_gpus"] == 2 and ge["questions"] == 128 and ge["scaling"] == "weak" and ge["value"] > 0
    assert ge["value"] <= 2.0 * ge["rank0_queries_per_s"] * 1.5  # an aggregate of two ranks, not a unit error
    assert 0.0 <= ge["reachability@100_all_ranks"] <= 1.0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("where", ["group", "step"])
def test_bench_two_rank_lane_failure_falls_back_in_process(dev, where):
    """If the lane-1 communicator cannot be created, or the first two-lane step raises, every rank drops IN-PROCESS to the
    one-communicator side-stream pipeline (ShardedIndex.agree_on_lanes), says so in `config.pipeline`, and the results
    stay exact — the run neither hangs nor dies."""
    line = _rehearse({"EVI_INJECT_LANE_FAILURE": where},
                     ["--steps", "4", "--warmup", "2", "--rows", "200001", "--dim", "128", "--k", "50", "--no-cpu-baseline",
                      "--no-graph-eval", "--no-encode", "--no-extra-legs"])
    pipe = line["config"]["pipeline"]
    assert "FALLBACK from two lanes" in pipe and "injected" in pipe and "one communicator" in pipe
    assert line["ranks_seen"] == 2 and line["hits_at_k"]["hits@1"] == 1.0 and line["hits_at_k"]["hits@50"] == 1.0 and line["sorted_ok"]
    assert line["two_stage"]["identical_to_f32_scan"]
