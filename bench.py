#!/usr/bin/env python3
"""bench.py — retriever hot-path benchmark on MI355X (contract: see DESIGN.md "Measurement").

Workload (BASELINE.json configs[1], "WebQSP full index, bge-base-en-v1.5, 1xMI355X brute-force
top-k", concrete shapes from SURVEY.md §8(d) config 2): a synthetic L2-normalised index of
N = 2^23 rows x D = 768 f32 (25.8 GB) resident in HBM; one step = one batch of Q = 32 question
embeddings -> exact cosine top-500 row ids + scores.  With --gpus P > 1 the SAME index is
row-sharded over the ranks (strong scaling): per-shard top-k, one RCCL all-gather of the packed
[Q, k] lists, merge on every rank, batches alternating between two pipeline lanes.

Prints ONE JSON line on rank 0: `value` / `roofline` / `cpu_baseline` for the f32 scan, plus extra objects that are
not part of `value`: `two_stage` (the same batches through the f16-shadow + f32 re-scoring scan, checked bit for bit
against the f32 scan), `graph_eval` (scorer + metrics + evaluation loop), `encode` (text-encoding stage).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
K_WINDOW = [1, 10, 25, 50, 100, 200, 300, 400, 500]  # configs/window/default.yaml:8
CHUNK_ROWS = 1 << 16
EPS = 1e-6


# EVI_BENCH_BACKEND=gloo: rehearse the multi-rank control flow on ONE GPU — every rank uses the same device, the process
# group is gloo and the [Q, k] record exchange is staged through host memory (RCCL refuses two ranks per device).  Sharding,
# barriers, max-over-ranks timing and the rank-0 JSON line are exactly the N > 1 path; the numbers mean nothing.
REHEARSAL_BACKEND = os.environ.get("EVI_BENCH_BACKEND", "nccl").lower()


def all_reduce_(t, op):
    if REHEARSAL_BACKEND == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def host_staged_exchange(dev, world):
    """exchange callable for ShardedIndex under the gloo rehearsal, or None (RCCL all-gather) otherwise."""
    if REHEARSAL_BACKEND != "gloo" or world <= 1:
        return None

    def exchange(all_records, local_record):
        torch.cuda.current_stream(dev).synchronize()
        host = local_record.cpu()
        gathered = torch.empty(world * host.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(gathered, host)
        all_records.copy_(gathered)

    return exchange


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1 << 23, help="total index rows (all ranks)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--k", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-rows", type=int, default=1 << 19, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--index-dtype", choices=["f32", "f16", "fp8"], default="f32",
                    help="storage dtype of the resident index (f16 / fp8 e4m3 + per-row scale: BASELINE configs 4 / 5; "
                         "the headline is f32)")
    ap.add_argument("--fp8-mfma", action="store_true",
                    help="with --index-dtype fp8: the native fp8 matrix instruction (two e4m3 query pieces) instead of widening "
                         "the index bytes to f16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-stage", action="store_true",
                    help="skip the extra leg that times the two-stage exact scan (f16 shadow selection + f32 re-scoring)")
    ap.add_argument("--topk-method", choices=["scan", "gemm", "auto", "two_stage"], default="scan",
                    help="local top-k kernel: the 32-queries-per-pass scan (default), the many-query GEMM-shaped pass "
                         "(same results; pays off for --queries >= 96), or auto")
    ap.add_argument("--graph-kernels", action="store_true",
                    help="run ONLY the BASELINE config 3 leg (CWQ-shaped CSR / DDE / BFS / seed expansion kernels with their "
                         "CPU-oracle baseline) and print its JSON object")
    ap.add_argument("--graph-batch", type=int, default=64, help="graphs per batch of the --graph-kernels leg")
    ap.add_argument("--no-labelling", action="store_true",
                    help="with --graph-kernels: skip the shortest-path labelling leg (counter passes: only CSR / DDE / BFS / expansion run)")
    ap.add_argument("--no-graph-eval", action="store_true")
    ap.add_argument("--eval-shard-questions", type=int, default=256, help="N > 1: questions per rank of the graph_eval_sharded leg")
    ap.add_argument("--no-encode", action="store_true", help="skip the text-encoding leg (random-init BERT + pooling kernel)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the sustained leg and the BASELINE config 3 / 4 / 5 legs of the default line")
    ap.add_argument("--config4-rows", type=int, default=100_000_000,
                    help="rows of the BASELINE configs[3] / configs[4] index (100 M): row-sharded over the ranks in the "
                         "`config4_sharded` / `config5_sharded` legs of an N > 1 run, whole on the one GPU in `config4_full` at N = 1")
    ap.add_argument("--no-config4-full", action="store_true",
                    help="N = 1: skip the one-GPU comparator of the 1 -> 8 scaling target (the whole 100 M x 768 f16 index, 154 GB)")
    ap.add_argument("--dist-timeout", type=float, default=600.0,
                    help="seconds a collective (or the rendezvous) may take before the process group aborts the rank")
    ap.add_argument("--deadline", type=float, default=2700.0,
                    help="seconds after which a rank that is still running gives up with exit code 124 (a hang must not outlive the run)")
    return ap.parse_args()


def build_shard(dev, row_begin, row_end, D, seed, dtype=torch.float32):
    """Rows [row_begin, row_end) of the global synthetic index, generated chunk-wise on the GPU so
    that the content of a global row does not depend on the number of ranks.  N(0,1) entries,
    global row 0 all-zero (eps clamp), 1 % of each chunk's rows duplicated (exact ties).  Every chunk is
    normalised (C1) and stored in `dtype` as it is made, so a 100 M-row f16 index (154 GB, BASELINE config 4) is
    built on one 288 GB GPU without ever holding its f32 form.  dtype "fp8": every chunk is quantised as it is made
    (OCP e4m3 bytes + one f32 scale per row; the codec is row-wise, so chunking changes nothing) and
    (bytes [n, D] uint8, scale [n] f32) is returned."""
    from evi_rag_amd import ops

    n = row_end - row_begin
    fp8 = dtype == "fp8"
    shard = torch.empty((n, D), dtype=torch.uint8 if fp8 else dtype, device=dev)
    scale = torch.empty(n, dtype=torch.float32, device=dev) if fp8 else None
    c0 = row_begin // CHUNK_ROWS
    c1 = (row_end + CHUNK_ROWS - 1) // CHUNK_ROWS
    gen = torch.Generator(device=dev)
    for c in range(c0, c1):
        gen.manual_seed(seed * 1_000_003 + c)
        chunk = torch.randn((CHUNK_ROWS, D), generator=gen, device=dev, dtype=torch.float32)
        ndup = CHUNK_ROWS // 100
        src = torch.randint(1, CHUNK_ROWS, (ndup,), generator=gen, device=dev)
        dst = torch.randint(1, CHUNK_ROWS, (ndup,), generator=gen, device=dev)
        chunk[dst] = chunk[src]
        if c == 0:
            chunk[0] = 0.0
        ops.normalize_embeddings(chunk, EPS, out=chunk)  # C1, in place: the resident index is normalised
        lo = max(row_begin, c * CHUNK_ROWS)
        hi = min(row_end, (c + 1) * CHUNK_ROWS)
        if fp8:
            b8, s8 = ops.quantize_rows_fp8(chunk[lo - c * CHUNK_ROWS: hi - c * CHUNK_ROWS])
            shard[lo - row_begin: hi - row_begin] = b8
            scale[lo - row_begin: hi - row_begin] = s8
        else:
            shard[lo - row_begin: hi - row_begin] = chunk[lo - c * CHUNK_ROWS: hi - c * CHUNK_ROWS]
        del chunk
    return (shard, scale) if fp8 else shard


def gold_row(step, qi, n_total):
    # global row the query (step, qi) is a noisy copy of; a fixed hash so every rank agrees
    return (1 + (step * 7919 + qi * 104729) * 2654435761) % n_total


def build_queries(dev, shard, row_begin, row_end, n_total, n_batches, Q, D, seed, world, row_scale=None):
    """Each query = normalise(gold index row + 0.5 * noise): Hits@k of the gold row is the
    size-independent correctness signal at full scale.  The rank that owns the gold row
    contributes it; an all-reduce SUM assembles the batch on every rank.  An e4m3 shard (uint8 + row_scale)
    contributes the DEQUANTISED gold rows."""
    from evi_rag_amd import ops

    gold = torch.tensor([[gold_row(s, i, n_total) for i in range(Q)] for s in range(n_batches)], dtype=torch.int64)
    base = torch.zeros((n_batches, Q, D), dtype=torch.float32, device=dev)
    mine = (gold >= row_begin) & (gold < row_end)
    if bool(mine.any()):
        local = (gold[mine] - row_begin).to(dev)
        rows = shard.index_select(0, local)
        if rows.dtype == torch.uint8:  # OCP e4m3 = torch.float8_e4m3fn
            rows = rows.view(torch.float8_e4m3fn).float() * row_scale.index_select(0, local).view(-1, 1)
        base[mine.to(dev)] = rows.float()
    if world > 1:
        all_reduce_(base, dist.ReduceOp.SUM)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 7 + 12345)
    noise = torch.randn((n_batches, Q, D), generator=gen, device=dev, dtype=torch.float32)
    q = base + 0.5 * noise / (D ** 0.5)
    q = ops.normalize_embeddings(q.view(-1, D), EPS).view(n_batches, Q, D)
    return q, gold


def cpu_baseline(shard, queries, k, n_total, cpu_rows, budget_s):
    """The oracle's torch-CPU path (normalised matmul + top-k, all host threads) timed on a bounded
    row sample of the same index, extrapolated linearly in N (the scan is linear in rows)."""
    from oracle import cpu_baseline as ob

    rows = int(min(cpu_rows, shard.shape[0]))
    x = shard[:rows].float().cpu()
    q = queries[0].cpu()
    return ob.time_cosine_topk(q, x, k, n_total=n_total, budget_s=budget_s)


def pmc_traffic(N, D, Q, k, world, kernel_ms_per_step, method="scan"):
    """`roofline.traffic` + where it comes from.  Hardware counters cannot be read inside the bench, so the HBM bytes of
    the scan kernel are those of the COMMITTED PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), as GB/s at this run's kernel time; `traffic_source` names the file
    so that nobody reads the figure as live.  Null when no profile exists for this configuration."""
    none = {"traffic": None, "traffic_source": None}
    if world != 1 or kernel_ms_per_step <= 0:
        return none
    best, best_name = None, None
    prof_dir = os.path.join(REPO_ROOT, "profiles")
    for name in sorted(os.listdir(prof_dir)) if os.path.isdir(prof_dir) else []:
        if name.endswith("_pmc_traffic.json"):
            with open(os.path.join(prof_dir, name)) as fh:
                p = json.load(fh)
            c = p.get("config", {})
            if (c.get("index_rows"), c.get("dim"), c.get("queries_per_step"), c.get("k"), c.get("method", "scan")) == (N, D, Q, k, method):
                best, best_name = p, name  # the latest round's file wins
    if best is None:
        return none
    return {"traffic": best["hbm_bytes_per_step"] / (kernel_ms_per_step * 1e-3) / 1e9,
            "traffic_source": f"profiles/{best_name}: committed rocprofv3 --pmc passes ({best['hbm_bytes_per_step']} HBM bytes per step), "
                              "rescaled by this run's kernel time; not read live"}


def exact_env():
    return os.environ.get("EVI_SCORER_GEMM", "")[:1] == "f"


def bench_graph_eval(dev, D, iters=8, warmup=2, graphs=32, nodes=1500, edges=4096, relations=4096, cpu_seconds=8.0, full=True):
    """Secondary leg (not part of `value`): the per-question subgraph scoring stage of the same
    evaluation — Retriever forward (DDE + edge scorer) and the fused ranking metrics on one
    WebQSP-shaped batch (32 graphs, N_g ~ 1500, E_g ~ 4096, D = H)."""
    import ctypes

    from evi_rag_amd import _lib, metrics as M, synthetic
    from evi_rag_amd.retriever import Retriever

    lib = _lib.load()
    sb = synthetic.make_batch(graphs, nodes_per_graph=nodes, edges_per_graph=edges, emb_dim=D, num_relations=relations,
                              num_entities=1 << 17, seed=1)
    batch = synthetic.as_namespace(sb, device=dev)
    batch.answer_entity_ids_ptr = torch.from_numpy(sb.answer_ptr).to(dev)
    batch.num_relations = relations
    torch.manual_seed(0)
    model = Retriever(emb_dim=D, hidden_dim=D).to(dev).eval()
    coll = M.RetrieverMetricCollection(K_WINDOW)
    target = batch.labels > 0.5
    E, N, H = sb.num_edges, sb.num_nodes, D
    # projections + Wc node_repr + per edge (Wa p, Wc r_ctx: E rows; Wb s: 2E rows; state_net.4: E rows — it runs once on the
    # softmax-combined row, the head is folded) — DESIGN.md §4
    # ... of which the relation-context block (Wc r_ctx) is multiplied once per distinct (relation, graph) PAIR by a forward-only
    # call, not once per edge (scorer.hip, k_pair_*): the executed flops count the pair rows
    eb = torch.repeat_interleave(torch.arange(graphs, device=dev), torch.from_numpy(np.diff(sb.edge_ptr)).to(dev))
    pairs = int(torch.unique(eb * relations + batch.edge_attr.to(dev).view(-1)).numel()) if not os.environ.get("EVI_SCORER_PAIRS", "1").startswith("0") else E
    gemm_flops = 2.0 * D * D * (N + 1 + 3 * graphs + relations) + 2.0 * N * D * H + (6.0 * E + 2.0 * pairs) * D * H + 2.0 * E * H * H
    gemm_flops_per_edge_form = gemm_flops + 2.0 * (E - pairs) * D * H

    def one():
        out = model(batch)
        coll.update(preds=out.logits, target=target, indexes=out.query_ids, batch=batch, num_graphs=graphs)
        return out

    for _ in range(warmup):
        one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        out = model(batch)
    torch.cuda.synchronize(dev)
    t_fwd = (time.perf_counter() - t0) / iters
    # the per-kernel-class split in a pass of its own: the library's timing events put ~5 us of gap between kernels, which the
    # wall clock above should not carry
    lib.evi_timing_enable(1)
    for _ in range(iters):
        out = model(batch)
    torch.cuda.synchronize(dev)
    lib.evi_timing_enable(0)
    ms = (ctypes.c_double * 4)()
    ln = (ctypes.c_int32 * 4)()
    _lib.check(lib.evi_timing_read(ms, ln, 4))
    t0 = time.perf_counter()
    for _ in range(iters):
        coll.update(preds=out.logits, target=target, indexes=out.query_ids, batch=batch, num_graphs=graphs)
    torch.cuda.synchronize(dev)
    t_met = (time.perf_counter() - t0) / iters
    # logits-only forward (score_head folded into state_net.4; what predict_step keeps)
    model.emit_edge_embeddings = False
    model(batch)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        model(batch)
    torch.cuda.synchronize(dev)
    t_lite = (time.perf_counter() - t0) / iters
    lite_logits = model(batch).logits.clone()
    # opt-in matmul_precision="f16x2": two f16 products (activations hi + lo in f16, weights rounded once to f16) instead of
    # three bf16 products — the same logits-only forward, its distance to the default's logits, and how many (graph, k)
    # top-k SETS of the evaluation window differ (near-ties flipped by the coarser weights)
    f16x2 = None
    if not exact_env():
        model.matmul_precision = "f16x2"
        lo2 = model(batch).logits
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            model(batch)
        torch.cuda.synchronize(dev)
        t_f16 = (time.perf_counter() - t0) / iters
        model.matmul_precision = "split"
        ra = M.rank_batch(lite_logits, target, batch, K_WINDOW, want_topk=True)
        rb2 = M.rank_batch(lo2, target, batch, K_WINDOW, want_topk=True)
        ia, ib = ra.topk_index.cpu().numpy(), rb2.topk_index.cpu().numpy()
        cnt = ra.topk_count.cpu().numpy()
        changed = sum(1 for g in range(graphs) for kk in K_WINDOW
                      if set(ia[g, :min(kk, int(cnt[g]))].tolist()) != set(ib[g, :min(kk, int(cnt[g]))].tolist()))
        f16x2 = {"forward_logits_only_ms_per_batch": t_f16 * 1e3, "speedup_over_default": t_lite / t_f16,
                 "max_abs_dlogit_vs_default": float((lo2 - lite_logits).abs().max().item()),
                 "topk_sets_changed_vs_default": changed, "of_graph_k_boundaries": graphs * len(K_WINDOW)}
    # How much the per-pair rows save depends on how often a graph repeats its relations.  The batch above draws every edge's
    # relation uniformly from all 4 096 (~2 600 distinct per 4 096-edge graph: the hard end); the same batch with 300 distinct
    # relations per graph — synthetic too, a sensitivity figure, not a dataset claim — shows the other end.
    pair_rows = None
    if pairs < E:
        gen = torch.Generator(device=dev).manual_seed(11)
        per_graph = min(300, relations)
        sets = torch.randint(0, relations, (graphs, per_graph), device=dev, generator=gen)
        attr0, emb0 = batch.edge_attr, batch.edge_embeddings
        attr1 = sets[eb, torch.randint(0, per_graph, (E,), device=dev, generator=gen)]
        table = torch.randn(relations, D, device=dev, generator=gen)
        batch.edge_attr, batch.edge_embeddings = attr1, table[attr1]
        try:
            model(batch)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                model(batch)
            torch.cuda.synchronize(dev)
            t_skew = (time.perf_counter() - t0) / iters
            pairs1 = int(torch.unique(eb * relations + attr1).numel())
        finally:
            batch.edge_attr, batch.edge_embeddings = attr0, emb0
        pair_rows = {"what": "state_net.0's relation-context block multiplied once per distinct (relation, graph) pair (forward-only calls)",
                     "pairs_per_graph_this_batch": pairs / graphs, "edges_per_graph": E / graphs,
                     "with_300_relations_per_graph": {"pairs_per_graph": pairs1 / graphs,
                                                      "forward_logits_only_ms_per_batch": t_skew * 1e3}}
    model.emit_edge_embeddings = True
    gemm_ms = ms[2] / iters
    tf = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    exact = os.environ.get("EVI_SCORER_GEMM", "")[:1] == "f"
    # split-bf16: three bf16 MFMAs per algorithmic product -> executed flops = 3 x algorithmic
    executed_tf, peak, kname = (tf, 157.3, "k_gemm_nt (f32 MFMA 32x32x2)") if exact else \
        (3.0 * tf, 2500.0, "k_gemm_nt_bf16x3 (bf16 MFMA 32x32x16, 3 products per f32 product)")
    roof = {"bound": "mfma", "achieved": executed_tf, "peak": peak, "unit": "TFLOP/s", "frac": executed_tf / peak,
            "kernel": kname, "algorithmic_tflops": tf, "gemm_ms_per_batch": gemm_ms,
            "gemm_launches_per_batch": ln[2] / iters, "algorithmic_flops_per_batch": gemm_flops,
            "relation_graph_pairs": pairs, "edges": E, "algorithmic_flops_per_batch_with_one_relation_row_per_edge": gemm_flops_per_edge_form,
            "edge_feature_ms_per_batch": ms[3] / iters, "traffic": None, "traffic_source": None}
    pm = pmc_leg_traffic("scorer", f"D{D}", "gemm")
    if pm is not None and gemm_ms > 0:
        # HBM bytes the GEMM kernels of ONE forward (edge features on, like the timed pass above) moved (counter passes of tools/scorer_forward_profile.py full on
        # the same batch shape), as GB/s at this run's GEMM time; the per-edge kernels' bytes beside it
        legs = {leg: pmc_leg_traffic("scorer", f"D{D}", leg) for leg in ("edge_features", "state_combine", "pair_rows")}
        roof.update(traffic=pm[1]["hbm_bytes_per_batch"] / (gemm_ms * 1e-3) / 1e9, traffic_unit="GB/s (HBM bytes of the GEMM launches)",
                    hbm_bytes_per_forward={"gemm": pm[1]["hbm_bytes_per_batch"],
                                           **{leg: v[1]["hbm_bytes_per_batch"] for leg, v in legs.items() if v is not None}},
                    traffic_source=f"profiles/{pm[0]}: committed rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE passes of the features-on "
                                   "forward on the same batch shape; not read live")
    if not full:
        return {"workload": f"{graphs} graphs, N={N}, E={E}, D=H={D}, DDE 2+2, bidirectional",
                "forward_ms_per_batch": t_fwd * 1e3, "forward_logits_only_ms_per_batch": t_lite * 1e3,
                "metrics_ms_per_batch": t_met * 1e3, "queries_per_s": graphs / (t_lite + t_met), "edges_per_s": E / t_lite,
                "f16x2": f16x2, "pair_rows": pair_rows, "roofline": roof}
    # training-shaped step (§8f-4): differentiable forward (per-edge intermediates kept) -> RetrieverLoss -> backward
    # (evi_retriever_backward replays them); eval-mode graph (no dropout), gradients of all 25 parameters
    from evi_rag_amd.loss import RetrieverLoss

    loss_fn = RetrieverLoss(infonce_temperature=0.07)
    model.differentiable = True
    model.emit_edge_embeddings = False

    def train_step():
        model.zero_grad(set_to_none=True)
        o = model(batch)
        loss_fn(o, batch.labels, edge_batch=o.query_ids, num_graphs=graphs).loss.backward()

    train_step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(4):
        train_step()
    torch.cuda.synchronize(dev)
    t_train = (time.perf_counter() - t0) / 4
    model.differentiable = None
    model.emit_edge_embeddings = True
    model.zero_grad(set_to_none=True)
    # the whole optimiser step the reference's trainer does per batch (train.RetrieverTrainer): train() mode with the reference's
    # defaults (dropout_p 0.1, hide-and-seek on: configs/model/retriever_module.yaml:8-25), loss, backward, gradient-norm
    # clipping at 1.0, AdamW over the flat parameter buffer — on a model of its own (the trainer re-points its parameters)
    from evi_rag_amd.train import RetrieverTrainer

    def trainer_leg(precision):
        torch.manual_seed(0)
        tmodel = Retriever(emb_dim=D, hidden_dim=D, dropout_p=0.1,
                           hide_seek_cfg={"enabled": True, "p_near": 0.7, "p_far": 0.1, "bias_near": -2.0, "bias_far": -0.5,
                                          "apply_in_eval": False}).to(dev)
        tmodel.emit_edge_embeddings = False
        trainer = RetrieverTrainer(tmodel, loss=RetrieverLoss(infonce_temperature=0.07), precision=precision,
                                   optimizer_cfg={"type": "adamw", "lr": 1e-3, "weight_decay": 1e-4}, gradient_clip_val=1.0)
        first = trainer.training_step(batch)
        trainer.training_step(batch)
        torch.cuda.synchronize(dev)
        # steady state: steps issued back to back, ONE synchronisation at the end — a training loop does not wait for the device
        # between steps (training_step returns the loss as a device scalar, nothing is read back)
        n_steps = 6
        t0 = time.perf_counter()
        for _ in range(n_steps):
            last = trainer.training_step(batch)
        torch.cuda.synchronize(dev)
        t_opt = (time.perf_counter() - t0) / n_steps
        # and with a device synchronisation after every step (a loop that logs the loss each step): the host's per-step set-up is
        # then exposed
        step_ms = []
        for _ in range(4):
            t0 = time.perf_counter()
            last = trainer.training_step(batch)
            torch.cuda.synchronize(dev)
            step_ms.append((time.perf_counter() - t0) * 1e3)
        return {"precision": precision, "matmul_precision": tmodel.matmul_precision, "ms_per_step": t_opt * 1e3,
                "questions_per_s": graphs / t_opt, "loss_first_step": float(first), "loss_last_step": float(last),
                "ms_per_step_synchronised_every_step": sum(step_ms) / len(step_ms), "step_ms_synchronised": step_ms,
                "parameters": int(trainer.optimizer.numel)}

    train_obj = dict(trainer_leg("32-true"),
                     what="RetrieverTrainer.training_step: train() forward (dropout 0.1, hide-and-seek) -> InfoNCE loss -> backward -> "
                          "clip_grad_norm 1.0 -> AdamW (flat buffers); same batch 12 times (2 warm-up, 6 back to back = ms_per_step, 4 synchronised); "
                          "split-bf16 (f32-grade) products")
    # opt-in: trainer.precision = bf16-mixed (configs/trainer/default.yaml:13-14) -> one bf16 product per GEMM, forward and backward
    train_obj["bf16_mixed"] = trainer_leg("bf16-mixed")
    metrics = {k: float(v) for k, v in coll.compute().items()}
    pipeline = bench_eval_pipeline(dev, D, model, nodes=nodes, edges=edges, relations=relations)
    cpu = None
    if cpu_seconds > 0:
        from oracle import cpu_baseline as ob

        small = synthetic.make_batch(2, nodes_per_graph=nodes, edges_per_graph=edges, emb_dim=D, num_relations=relations,
                                     num_entities=1 << 15, seed=3)
        weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        cpu = ob.time_graph_eval(weights, small, K_WINDOW, budget_s=cpu_seconds)
    return {
        "workload": f"{graphs} graphs, N={N}, E={E}, D=H={D}, DDE 2+2, bidirectional, k window {K_WINDOW[0]}..{K_WINDOW[-1]}",
        "forward_ms_per_batch": t_fwd * 1e3,
        "forward_logits_only_ms_per_batch": t_lite * 1e3,
        "metrics_ms_per_batch": t_met * 1e3,
        "f16x2": f16x2,
        "pair_rows": pair_rows,
        "train_step_ms_per_batch": t_train * 1e3,
        "train": train_obj,
        "queries_per_s": graphs / (t_fwd + t_met),
        "edges_per_s": E / t_fwd,
        "roofline": roof,
        "reachability@100": metrics.get("answer/reachability@100"),
        "edge_recall@100": metrics.get("edge/recall@100"),
        "eval_pipeline": pipeline,
        "cpu_baseline": cpu,
    }


def bench_eval_pipeline(dev, D, model, *, graphs_total=512, batch_size=32, nodes=1500, edges=4096, relations=4096, passes=2, seed=2):
    """End-to-end evaluation epoch over an HBM-resident packed split: device collation, embedding gather,
    Retriever forward, loss, ranking metrics (RetrieverEvaluator.run) — queries/s of the whole per-question
    stage, everything the reference does between its DataLoader and `test/...` metrics.  512 questions = 16 batches per pass
    (WebQSP's test split is 1 628 questions = 51 batches): with 4 batches per pass the first batch's collation and the last
    batch's metrics, which nothing overlaps, weighed 8 % (8 650 against 9 300 questions/s)."""
    import shutil
    import tempfile

    from evi_rag_amd import packed_dataset as pd, synthetic
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore
    from evi_rag_amd.eval_loop import RetrieverEvaluator

    num_entities = 1 << 17
    sb = synthetic.make_batch(graphs_total, nodes_per_graph=nodes, edges_per_graph=edges, emb_dim=D, num_relations=relations,
                              num_entities=num_entities, seed=seed, attach_embeddings=False)
    tmp = tempfile.mkdtemp(prefix="evi_packed_")
    try:
        pd.write_packed(tmp, pd.samples_from_flat_batch(sb))
        gen = torch.Generator(device=dev).manual_seed(5)
        store = GlobalEmbeddingStore.from_tensors(torch.randn(num_entities, D, device=dev, generator=gen),
                                                  torch.randn(relations, D, device=dev, generator=gen), device=dev)
        ds = pd.PackedRetrievalDataset(tmp, device=dev, embeddings=store)
        ev = RetrieverEvaluator(model, k_values=K_WINDOW)
        ev.run(pd.PackedLoader(ds, batch_size=batch_size))  # warm-up pass
        best = None
        for _ in range(passes):
            res = ev.run(pd.PackedLoader(ds, batch_size=batch_size))
            if best is None or res["seconds"] < best["seconds"]:
                best = res
        return {"workload": f"{graphs_total} questions in batches of {batch_size}, E={sb.num_edges}, D=H={D}; split resident in HBM "
                            f"({ds.nbytes() / 1e6:.0f} MB)",
                "queries_per_s": best["queries_per_sec"], "ms_per_batch": best["seconds"] / (graphs_total / batch_size) * 1e3,
                "loss": best["metrics"]["test/loss"], "reachability@100": best["metrics"].get("test/answer/reachability@100")}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_eval_sharded(ctx, D, *, graphs_per_rank=256, seed_base=20):
    """N > 1: the per-question stage sharded by QUESTION (second bullet): graphs are independent, so every rank
    evaluates its own share (here: its own synthetic split of `graphs_per_rank` questions, embedding tables replicated in its
    HBM) with no exchange inside the epoch, and the epoch's counters are summed with ONE small all-reduce (the reference's
    dist_reduce_fx="sum").  Weak scaling by construction: value = all ranks' questions / the slowest rank's seconds.
    Every rank returns; rank 0's result goes into the JSON line.  A rank whose local epoch fails says so and ALL ranks skip the
    collectives together (ctx.agree) — the leg never leaves a rank waiting."""
    from evi_rag_amd.retriever import Retriever

    dev = ctx.dev
    local, err = None, None
    try:
        torch.manual_seed(0)
        model = Retriever(emb_dim=D, hidden_dim=D).to(dev).eval()
        local = bench_eval_pipeline(dev, D, model, graphs_total=graphs_per_rank, passes=2, seed=seed_base + ctx.rank)
    except Exception as exc:  # noqa: BLE001 - reported in the line, the other legs stand
        err = f"rank {ctx.rank}: {type(exc).__name__}: {exc}"
    if ctx.agree(err is not None):
        return {"skipped": err or "another rank's local epoch failed"}
    secs = graphs_per_rank / local["queries_per_s"]
    t = torch.tensor([secs, float(graphs_per_rank), (local["reachability@100"] or 0.0) * graphs_per_rank], dtype=torch.float64, device=dev)
    tmax = t[:1].clone()
    all_reduce_(tmax, dist.ReduceOp.MAX)
    all_reduce_(t, dist.ReduceOp.SUM)  # the metric counters' all-reduce: questions and reachability hits over all ranks
    total_q, slowest = float(t[1].item()), float(tmax[0].item())
    return {"workload": f"{ctx.world} ranks x {graphs_per_rank} questions (own split per rank, batches of 32, D=H={D}): device collation + "
                        "embedding gather + Retriever forward + loss + ranking metrics, no exchange inside the epoch; counters summed by one all-reduce",
            "metric": "queries/sec", "value": total_q / slowest, "unit": "queries/s", "n_gpus": ctx.world, "scaling": "weak",
            "questions": int(total_q), "slowest_rank_seconds": slowest, "rank0_queries_per_s": local["queries_per_s"],
            "reachability@100_all_ranks": float(t[2].item()) / max(total_q, 1.0)}


def _random_bert(dev, D):
    from transformers import BertConfig, BertModel

    heads = {384: 12, 768: 12, 1024: 16}.get(D, 12)
    layers = {384: 6, 768: 12, 1024: 24}.get(D, 12)
    torch.manual_seed(0)
    model = BertModel(BertConfig(vocab_size=30522, hidden_size=D, num_hidden_layers=layers, num_attention_heads=heads,
                                 intermediate_size=4 * D, max_position_embeddings=512), add_pooling_layer=False).to(dev).eval()
    return model, layers


class _Fp8Linear(torch.nn.Module):
    """nn.Linear on PyTorch-ROCm's fp8 GEMM (torch._scaled_mm -> hipBLASLt): OCP e4m3 operands with ONE scale per tensor — the
    weight quantised once, the activation per call from its own amax (a device scalar: nothing is read back) — bf16 result,
    bias added after.  BASELINE configs[4] "fp8 encode": the encoder forward stays PyTorch's (north_star), so this is a probe of
    what the framework offers on gfx950, not a kernel of ours."""

    def __init__(self, lin: torch.nn.Linear):
        super().__init__()
        w = lin.weight.detach().float()
        self.scale_w = (w.abs().amax() / 448.0).clamp(min=1e-12)
        self.w8 = (w / self.scale_w).to(torch.float8_e4m3fn)  # [out, in] row-major: its transpose is the column-major B operand
        self.bias = None if lin.bias is None else lin.bias.detach()

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        sx = (x2.abs().amax().float() / 448.0).clamp(min=1e-12)
        x8 = (x2.float() / sx).to(torch.float8_e4m3fn)
        y = torch._scaled_mm(x8, self.w8.t(), scale_a=sx, scale_b=self.scale_w, out_dtype=torch.bfloat16)
        if self.bias is not None:
            y = y + self.bias.to(y.dtype)
        return y.reshape(*shp[:-1], y.shape[-1])


def _swap_linears_fp8(module):
    n = 0
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.Linear) and child.in_features % 16 == 0 and child.out_features % 16 == 0:
            setattr(module, name, _Fp8Linear(child))
            n += 1
        else:
            n += _swap_linears_fp8(child)
    return n


class _LengthTokenizer:
    """Stand-in tokenizer (no vocabulary offline): text "i" has lengths[i] random token ids; a batch is padded to its longest
    text like the reference's `padding=True` (scripts/text_encode_utils.py:52-57)."""

    def __init__(self, lengths):
        self.lengths = lengths

    def __call__(self, batch, padding=True, truncation=True, return_tensors="pt"):
        lens = [self.lengths[int(t)] for t in batch]
        L = max(lens)
        g = torch.Generator().manual_seed(int(batch[0]))
        ids = torch.randint(1000, 30000, (len(batch), L), generator=g)
        mask = (torch.arange(L).view(1, L) < torch.tensor(lens).view(-1, 1)).to(torch.int64)
        return {"input_ids": ids * mask, "attention_mask": mask}


def bench_encode(dev, D, *, texts=4096, batch_size=64, iters=3, autocast=None, fp8_table=False):
    """E1-E4 stage: `TextEncoder.encode_to_device` over synthetic WebQSP-sized texts (8..32 tokens, batches of 64 as
    configs/build_retrieval_pipeline.yaml) with a RANDOM-INIT BERT of the bge shape for D (no weights exist offline).  The
    transformer forward is PyTorch-ROCm, as north_star states; the pooling tail is evi_masked_mean_pool.
    autocast="bf16" + fp8_table=True is the BASELINE config 5 form: the forward under bf16 autocast, the resulting table
    stored as OCP e4m3 + per-row scale, and overlap@k of (bf16 encode, fp8 table) against (f32 encode, f32 table) for the
    same texts and questions — a random-init encoder's embeddings are nearly collinear, so that overlap is a LOWER bound
    on what a trained encoder gives."""
    from evi_rag_amd import ops
    from evi_rag_amd.text_encode import TextEncoder

    model, layers = _random_bert(dev, D)
    rng = np.random.default_rng(0)
    lengths = rng.integers(8, 33, texts)
    enc = TextEncoder.from_components(_LengthTokenizer(lengths), model, str(dev), fp16=False)
    if autocast is not None:
        enc.autocast = {"bf16": torch.bfloat16, "f16": torch.float16}[autocast]
    names = [str(i) for i in range(texts)]
    tokens = int(sum(max(lengths[b: b + batch_size]) * len(lengths[b: b + batch_size]) for b in range(0, texts, batch_size)))
    params = sum(p.numel() for n, p in model.named_parameters() if "embeddings" not in n)
    flops = 2.0 * params * tokens

    def best_of(n):
        best, out = float("inf"), None
        for _ in range(n):
            t0 = time.perf_counter()
            out = enc.encode_to_device(names, batch_size)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return best, out

    # eager launches (TextEncoder.use_graphs = False): what the reference's wrapper does
    enc.use_graphs = False
    enc.encode_to_device(names[: 4 * batch_size], batch_size)
    torch.cuda.synchronize(dev)
    best, out = best_of(iters)
    # the encoder's DEFAULT: the forward + pooling of each (batch, padded length) shape replayed as one hipGraph
    enc.use_graphs = True
    enc.encode_to_device(names, batch_size)  # every shape of the pass was met in the eager passes: all captured here, untimed
    torch.cuda.synchronize(dev)
    best_g, out_g = best_of(iters)
    res = {"workload": f"{texts} texts of 8..32 tokens in batches of {batch_size}; random-init BERT {layers}L/{D}H, "
                       f"{'f32' if autocast is None else autocast + ' autocast'}; PyTorch-ROCm forward + evi_masked_mean_pool; "
                       "the TextEncoder default: one hipGraph replay per (batch, padded length) shape",
           "texts_per_s": texts / best_g, "ms_per_batch": best_g / (texts / batch_size) * 1e3, "padded_tokens": tokens,
           "encoder_tflops": flops / best_g / 1e12, "out_shape": list(out_g.shape), "graphs_captured": len(enc._graphs),
           "max_abs_diff_to_eager": float((out_g - out).abs().max().item()),
           "eager": {"what": "use_graphs = False: every kernel launched from Python", "texts_per_s": texts / best,
                     "ms_per_batch": best / (texts / batch_size) * 1e3, "encoder_tflops": flops / best / 1e12}}
    enc.use_graphs = False
    if fp8_table:
        k = 100
        enc.autocast = None
        ref = ops.normalize_embeddings(enc.encode_to_device(names, batch_size))      # f32 pipeline: f32 encode, f32 table
        low = ops.normalize_embeddings(out)                                          # the reduced-precision encode above
        table8, scale8 = ops.quantize_rows_fp8(low)
        q_ref, q_low = ref[:32].contiguous(), low[:32].contiguous()                  # the first 32 texts double as questions
        _, i_ref = ops.cosine_topk(q_ref, ref, k)
        _, i_low = ops.cosine_topk(q_low, table8, k, row_scale=scale8)
        _, i_mid = ops.cosine_topk(q_low, low, k)
        ov = lambda a, b: float(((a.unsqueeze(2) == b.unsqueeze(1)).any(dim=2).float().sum(dim=1) / k).mean().item())  # noqa: E731
        res["fp8_table"] = {"rows": texts, "k": k, "table_bytes": int(table8.numel() + 4 * scale8.numel()),
                            "overlap_at_k_vs_f32_pipeline": ov(i_low, i_ref),
                            "overlap_at_k_encoder_only (bf16 encode, f32 table)": ov(i_mid, i_ref),
                            "max_abs_embedding_diff": float((low - ref).abs().max().item()),
                            "note": "random-init encoder: embeddings nearly collinear, overlap is a lower bound"}
        # configs[4] "fp8 encode": every nn.Linear of the encoder layers on torch._scaled_mm (e4m3 x e4m3, per-tensor scales,
        # bf16 out), the rest under bf16 autocast as above.  Eager launches (the per-call activation amax + quantisation are
        # extra kernels).  If this PyTorch build has no usable fp8 GEMM on gfx950 the error is recorded and J2 stays bf16.
        try:
            import copy as _copy

            m8 = _copy.deepcopy(model)
            swapped = _swap_linears_fp8(m8.encoder)
            enc8 = TextEncoder.from_components(_LengthTokenizer(lengths), m8, str(dev), fp16=False)
            enc8.autocast, enc8.use_graphs = torch.bfloat16, False
            out8 = enc8.encode_to_device(names[: 4 * batch_size], batch_size)
            torch.cuda.synchronize(dev)
            best8 = float("inf")
            for _ in range(iters):
                t0 = time.perf_counter()
                out8 = enc8.encode_to_device(names, batch_size)
                torch.cuda.synchronize(dev)
                best8 = min(best8, time.perf_counter() - t0)
            low8 = ops.normalize_embeddings(out8)
            _, i_8 = ops.cosine_topk(low8[:32].contiguous(), low8, k)
            enc8.use_graphs = True
            try:
                enc8.encode_to_device(names, batch_size)
                torch.cuda.synchronize(dev)
                best8g = float("inf")
                for _ in range(iters):
                    t0 = time.perf_counter()
                    enc8.encode_to_device(names, batch_size)
                    torch.cuda.synchronize(dev)
                    best8g = min(best8g, time.perf_counter() - t0)
                graphed = {"texts_per_s": texts / best8g, "graphs_captured": len(enc8.__dict__.get("_graphs", {})), "use_graphs_after": bool(enc8.use_graphs)}
            except Exception as exc:  # noqa: BLE001
                graphed = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            res["fp8_linear"] = {"what": "encoder nn.Linear layers on torch._scaled_mm (OCP e4m3 operands, per-tensor scales, bf16 result), "
                                         "attention / LayerNorm / GELU under bf16 autocast",
                                 "linears_swapped": swapped, "texts_per_s": texts / best8, "ms_per_batch": best8 / (texts / batch_size) * 1e3,
                                 "speedup_over_bf16_autocast_eager": best / best8, "graph_replay": graphed,
                                 "overlap_at_k_encoder_only (fp8 linears, f32 table)": ov(i_8, i_ref),
                                 "max_abs_embedding_diff_vs_f32": float((low8 - ref).abs().max().item())}
            del m8, enc8
        except Exception as exc:  # noqa: BLE001 - the probe's outcome IS the error when fp8 GEMMs are not available
            res["fp8_linear"] = {"error": f"{type(exc).__name__}: {exc}"[:600]}
        torch.cuda.empty_cache()
    return res


def _lib_load():
    from evi_rag_amd import _lib

    return _lib.load()


def bench_end_to_end(dev, D, *, rows, k, seed, questions=32, iters=12, warmup=3, nodes=1500, edges=4096, relations=4096):
    """ONE query = one question (SURVEY.md §8d restated metric): a batch of 32 question texts -> `TextEncoder.encode_to_device`
    (random-init bge-base-shaped BERT + pooling kernel) -> L2 normalise -> exact cosine top-500 over the resident index ->
    `Retriever.forward` on the batch's WebQSP-shaped subgraphs (with the encoded questions as `question_emb`) -> fused
    ranking metrics.  The index candidates do not choose the subgraphs (the reference has no such link either: its graphs
    come from the dataset), so the stages are chained in time, not in data, except for the question embeddings.

    `queries_per_s` (top level) is the PIPELINE with the library's defaults: the three stages on three HIP streams with two
    buffer slots, the encoder replayed as hipGraphs (TextEncoder.use_graphs, default), the top-k through
    `ops.cosine_topk` (method "auto": the two-stage exact scan, because the index's f16 shadow is resident).  Sub-objects:
    `serial` = the same defaults issued on ONE stream (per-stage event times; the slowest stage is named), and
    `serial_eager_f32_scan` = round 2's settings (eager encoder launches, f32 scan) for continuity.  Every variant's last
    top-k must equal the f32 scan's bit for bit."""
    import copy as _copy

    from evi_rag_amd import metrics as M, ops, synthetic
    from evi_rag_amd.retriever import Retriever
    from evi_rag_amd.text_encode import TextEncoder

    index = build_shard(dev, 0, rows, D, seed)
    shadow = ops.index_shadow_f16(index)  # attaches itself to `index`: cosine_topk(method="auto") finds it
    out_topk = (torch.empty((questions, k), dtype=torch.float32, device=dev), torch.empty((questions, k), dtype=torch.int64, device=dev))
    model, layers = _random_bert(dev, D)
    rng = np.random.default_rng(1)
    lengths = rng.integers(8, 33, questions * (iters + warmup))
    enc = TextEncoder.from_components(_LengthTokenizer(lengths), model, str(dev), fp16=False)
    sb = synthetic.make_batch(questions, nodes_per_graph=nodes, edges_per_graph=edges, emb_dim=D, num_relations=relations,
                              num_entities=1 << 17, seed=1)
    batch = synthetic.as_namespace(sb, device=dev)
    batch.answer_entity_ids_ptr = torch.from_numpy(sb.answer_ptr).to(dev)
    batch.num_relations = relations
    torch.manual_seed(0)
    scorer = Retriever(emb_dim=D, hidden_dim=D).to(dev).eval()
    scorer.emit_edge_embeddings = False  # what the evaluation keeps: logits
    target = batch.labels > 0.5
    stages = ["encode", "topk", "scorer", "metrics"]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(iters)]
    name_list = lambda it: [str(it * questions + j) for j in range(questions)]  # noqa: E731

    def run_serial(method):
        coll = M.RetrieverMetricCollection(K_WINDOW)

        def one(it, marks=None):
            if marks:
                marks[0].record()
            q = ops.normalize_embeddings(enc.encode_to_device(name_list(it), questions), EPS)
            if marks:
                marks[1].record()
            ops.cosine_topk(q, index, k, out=out_topk, method=method)
            if marks:
                marks[2].record()
            batch.question_emb = q
            out = scorer(batch)
            if marks:
                marks[3].record()
            coll.update(preds=out.logits, target=target, indexes=out.query_ids, batch=batch, num_graphs=questions)
            if marks:
                marks[4].record()

        for it in range(warmup):
            one(it)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for it in range(iters):
            one(warmup + it, ev[it])
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        per = {s: sum(ev[it][j].elapsed_time(ev[it][j + 1]) for it in range(iters)) / iters for j, s in enumerate(stages)}
        metrics = {kk: float(v) for kk, v in coll.compute().items()}
        return {"queries_per_s": questions * iters / wall, "ms_per_batch": wall / iters * 1e3, "stage_ms_per_batch": per,
                "slowest_stage": max(per, key=per.get), "host_gap_ms_per_batch": wall / iters * 1e3 - sum(per.values()),
                "topk_method": ops.cosine_topk.last_method, "reachability@100": metrics.get("answer/reachability@100")}

    # ---- round 2's settings: eager encoder launches, f32 scan; its last top-k is the reference for the others
    enc.use_graphs = False
    legacy = run_serial("scan")
    scan_topk = (out_topk[0].clone(), out_topk[1].clone())
    # ---- the defaults on one stream.  Every (batch, length) shape of the run was met in the eager loop: captured in this
    # untimed priming pass (a capture costs tens of ms, once per shape per process)
    enc.use_graphs = True
    for it in range(warmup + iters):
        enc.encode_to_device(name_list(it), questions)
    torch.cuda.synchronize(dev)
    serial = run_serial("auto")
    serial["last_topk_identical_to_f32_scan"] = bool(torch.equal(out_topk[1], scan_topk[1]) and torch.equal(out_topk[0], scan_topk[0]))
    serial["what"] = ("one stream, library defaults: encoder forward + pooling replayed as one hipGraph per shape; ops.cosine_topk "
                      "method 'auto' = the two-stage exact scan over the resident f16 shadow (device-side repair of a failed proof)")
    legacy["what"] = "one stream, round 2's settings: eager encoder launches, f32 scan (method='scan')"

    # ---- The same work as a three-stream pipeline: encode on one HIP stream, the index top-k (HBM-bound) on a second, scorer +
    # metrics (MFMA-bound) on a third.  Batch i's top-k and scorer both wait for its encode; nothing else orders them, so the
    # HBM-bound scan of one batch runs under the matrix-bound encode / scorer of its neighbours.  Two slots of buffers.
    ts_ws = torch.empty(int(_lib_load().evi_cosine_topk_two_stage_workspace_bytes(questions, rows, D, k)), dtype=torch.uint8, device=dev)
    s_enc, s_topk, s_sc = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
    slots = 2
    topk_slot = [(torch.empty((questions, k), dtype=torch.float32, device=dev), torch.empty((questions, k), dtype=torch.int64, device=dev))
                 for _ in range(slots)]
    batches = [_copy.copy(batch) for _ in range(slots)]
    enc_done = [torch.cuda.Event() for _ in range(slots)]
    topk_done = [torch.cuda.Event() for _ in range(slots)]
    sc_done = [torch.cuda.Event() for _ in range(slots)]
    coll2 = M.RetrieverMetricCollection(K_WINDOW)
    keep = []

    def pipelined(it):
        sl = it % slots
        with torch.cuda.stream(s_enc):
            s_enc.wait_event(topk_done[sl])  # the slot's previous consumers are done with its query buffer
            s_enc.wait_event(sc_done[sl])
            q = ops.normalize_embeddings(enc.encode_to_device(name_list(it), questions), EPS)
            keep.append(q)
            enc_done[sl].record(s_enc)
        with torch.cuda.stream(s_topk):
            s_topk.wait_event(enc_done[sl])
            ops.cosine_topk_two_stage(q, index, shadow, k, workspace=ts_ws, out=topk_slot[sl])  # what method="auto" calls, own workspace
            topk_done[sl].record(s_topk)
        with torch.cuda.stream(s_sc):
            s_sc.wait_event(enc_done[sl])
            bt = batches[sl]
            bt.question_emb = q
            o = scorer(bt)
            coll2.update(preds=o.logits, target=target, indexes=o.query_ids, batch=bt, num_graphs=questions)
            keep.append(o)
            sc_done[sl].record(s_sc)

    def run_pipelined():
        torch.cuda.synchronize(dev)
        # a full untimed pass first: every stream's allocator pool holds the blocks of a whole pass before the clock starts (the
        # `keep` list holds each batch's outputs, so a first pass allocates them fresh — a hipMalloc per tensor, which stalls
        # the other streams too)
        for it in range(warmup + iters):
            pipelined(it)
        torch.cuda.synchronize(dev)
        keep.clear()
        for it in range(warmup):
            pipelined(it)
        torch.cuda.synchronize(dev)
        keep.clear()
        t0 = time.perf_counter()
        for it in range(iters):
            pipelined(warmup + it)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        last = topk_slot[(warmup + iters - 1) % slots]
        return wall, t_issue, bool(torch.equal(last[1], scan_topk[1]) and torch.equal(last[0], scan_topk[0]))

    wall_p, issue_p, same_p = run_pipelined()
    enc.use_graphs = False
    wall_pe, issue_pe, same_pe = run_pipelined()
    enc.use_graphs = True
    # (The three stages do not overlap by themselves: a scan workgroup — 1 024 threads x 128 registers — owns a CU's whole register
    # file, a scorer GEMM workgroup 128 KiB of its LDS, so the encoder's small kernels queue behind whole workgroups.  CU-masked
    # streams (hipExtStreamCreateWithCUMask: 16 CUs for the encoder stream, 240 for the other two) were tried in round 3 to let it
    # run beside them: 25.2 ms per batch against 8.5 ms — masked queues cost far more than the overlap gives.  Not kept.)
    res = {"workload": f"{questions} questions per batch: encode (random-init BERT {layers}L/{D}H, f32) -> top-{k} over {rows} x {D} f32 "
                       f"-> scorer on {questions} WebQSP-shaped graphs (E={sb.num_edges}, D=H={D}, logits only) -> fused metrics",
           "what": "encode | exact top-k | scorer + metrics on three HIP streams, two buffer slots, library defaults (graph-replayed "
                   "encoder, two-stage exact scan): the HBM-bound scan runs under the MFMA-bound encoder and scorer of the "
                   "neighbouring batches",
           "queries_per_s": questions * iters / wall_p, "ms_per_batch": wall_p / iters * 1e3,
           "host_issue_ms_per_batch": issue_p / iters * 1e3,
           "eager_encoder": {"what": "the same pipeline with TextEncoder.use_graphs = False", "queries_per_s": questions * iters / wall_pe,
                             "ms_per_batch": wall_pe / iters * 1e3, "host_issue_ms_per_batch": issue_pe / iters * 1e3,
                             "last_topk_identical_to_f32_scan": same_pe},
           "last_topk_identical_to_f32_scan": same_p,
           "last_topk_identical_to_serial_loop": same_p and serial["last_topk_identical_to_f32_scan"],
           "speedup_over_serial_loop": serial["ms_per_batch"] / (wall_p / iters * 1e3),
           "graphs_captured": len(enc._graphs),
           "serial": serial, "serial_eager_f32_scan": legacy,
           "best_score_mean": float(scan_topk[0][:, 0].mean().item())}
    ops.drop_shadow_f16(index)
    del shadow, ts_ws, index
    torch.cuda.empty_cache()
    return res


GRAPH_LEG_KERNELS = {  # C-ABI entry point -> the kernels it launches (names as rocprofv3 prints them)
    "evi_graph_csr": ("k_csr_part_count", "k_csr_part_scan", "k_csr_part_fill", "k_graph_csr"),
    "evi_dde_node_struct": ("k_dde_round", "k_dde_graph"),
    "evi_bfs_levels": ("k_bfs_levels",),  # also matches k_bfs_levels_edges
    "evi_select_start_edges": ("k_select_start_edges", "k_zero_mask"),
}


def pmc_leg_traffic(kind, key, leg):
    """HBM bytes per batch of one leg from the latest committed counter file `profiles/r*_pmc_<kind>.json` (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, tools/collect_pmc_legs.sh), or None.  Not read live: counters cannot be
    collected inside the bench; the file names its command."""
    prof_dir = os.path.join(REPO_ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(prof_dir)) if os.path.isdir(prof_dir) else []:
        if name.endswith(f"_pmc_{kind}.json"):
            with open(os.path.join(prof_dir, name)) as fh:
                p = json.load(fh)
            ent = (p.get(key) or {}).get("legs", {}).get(leg)
            if ent is not None:
                best = (name, ent)
    return best


def bench_graph_kernels(dev, batch, *, graphs=3531, nodes=3000, edges=10000, iters=20, cpu_graphs=8, cpu=True, labelling_leg=True):
    """BASELINE config 3 (CWQ-shaped 2-hop expansion): CSR build, DDE structure features, multi-source BFS levels,
    seed-incident edge selection on batches of CWQ-shaped graphs (N_g ~ 3 000, E_g ~ 10 000, DDE 2 + 2 rounds,
    ratio 0.25), the algorithmic bytes of each kernel as GB/s, beside the CPU oracle on a sample."""
    from evi_rag_amd import _lib, ops, synthetic

    lib = _lib.load()
    B = batch

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    sb = synthetic.make_batch(B, nodes_per_graph=nodes, edges_per_graph=edges, emb_dim=8, num_relations=512, seed=2,
                              attach_embeddings=False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ei, ptr, eptr, topic = t(sb.edge_index), t(sb.ptr), t(sb.edge_ptr), t(sb.topic_one_hot)
    N, E = sb.num_nodes, sb.num_edges
    scores = torch.randn(E, device=dev)
    seeds = t(sb.q_local_indices)
    res = {"workload": f"{B} CWQ-shaped graphs per batch, N={N}, E={E} (N_g~{nodes}, E_g~{edges}), {graphs} graphs per epoch",
           "kernels": {}}

    def rec(name, ms, nbytes, note):
        ent = {"ms_per_batch": ms, "algorithmic_bytes": nbytes, "GB_per_s": nbytes / (ms * 1e-3) / 1e9,
               "graphs_per_s": B / (ms * 1e-3), "note": note, "traffic": None, "traffic_source": None}
        pm = pmc_leg_traffic("graph", f"batch_{B}", name)
        if pm is not None and pm[1].get("workload_edges") == E:
            hb = pm[1]["hbm_bytes_per_batch"]
            ent.update(traffic=hb / (ms * 1e-3) / 1e9, hbm_bytes_per_batch=hb, traffic_over_algorithmic=hb / nbytes,
                       traffic_source=f"profiles/{pm[0]}: committed rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE passes of the same "
                                      "batch, rescaled by this run's kernel time; not read live")
        res["kernels"][name] = ent

    csr = ops.graph_csr(ei, ptr, eptr, num_nodes=N)
    csr_ws = torch.empty(int(lib.evi_graph_csr_workspace_bytes(N)), dtype=torch.uint8, device=dev)
    rec("evi_graph_csr", timed(lambda: ops.graph_csr(ei, ptr, eptr, num_nodes=N, out=csr, workspace=csr_ws)),
        E * 16 + 2 * (E * 8 + N * 4), "edge_index read (16 B/edge) + both CSR halves written (nbr + eid per edge, ptr per node)")
    rounds = 2
    S = 1 + 2 * rounds
    ns = torch.empty((N, 2 * S), dtype=torch.float32, device=dev)

    def dde():
        _lib.check(lib.evi_dde_node_struct_graphs(topic.data_ptr(), topic.size(1), 2, N, ptr.data_ptr(), B, csr.in_ptr.data_ptr(),
                                                  csr.in_nbr.data_ptr(), csr.out_ptr.data_ptr(), csr.out_nbr.data_ptr(), rounds, rounds,
                                                  ns.data_ptr(), ops._stream(dev)))

    rec("evi_dde_node_struct", timed(dde),
        2 * rounds * (E * 12 + N * 16) + N * 10 * 4, "per round E*(4 nbr + 8 gathered) + N*(8 ptr + 8 out); 2 + 2 rounds")
    jg = torch.arange(B, dtype=torch.int32, device=dev)
    sp, doff = t(sb.q_ptr), t(sb.ptr[:-1])
    dist_lv = torch.empty(N, dtype=torch.int32, device=dev)

    def bfs():
        _lib.check(lib.evi_bfs_levels_edges(jg.data_ptr(), sp.data_ptr(), seeds.data_ptr(), doff.data_ptr(), B, ptr.data_ptr(),
                                            eptr.data_ptr(), ei.data_ptr(), E, csr.in_ptr.data_ptr(), csr.in_nbr.data_ptr(),
                                            csr.out_ptr.data_ptr(), csr.out_nbr.data_ptr(), 0, dist_lv.data_ptr(), ops._stream(dev)))

    ms = timed(bfs)
    levels = int(dist_lv.max().item()) + 1
    rec("evi_bfs_levels", ms, N * 4 + E * 16,
        f"undirected, {levels} levels, edge-parallel in LDS (evi_bfs_levels_edges): the edge list once (16 B/edge), the levels written once")
    res["two_hop_frontier_nodes_per_graph"] = int(((dist_lv >= 0) & (dist_lv <= 2)).sum().item()) / B
    mask = torch.empty(E, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    def expand():
        _lib.check(lib.evi_select_start_edges(scores.data_ptr(), E, seeds.data_ptr(), seeds.numel(), csr.in_ptr.data_ptr(),
                                              csr.in_eid.data_ptr(), csr.out_ptr.data_ptr(), csr.out_eid.data_ptr(), N, 0.25, 1,
                                              -1, mask.data_ptr(), status.data_ptr(), ops._stream(dev)))

    deg = (csr.in_ptr[seeds + 1] - csr.in_ptr[seeds] + csr.out_ptr[seeds + 1] - csr.out_ptr[seeds]).sum().item()
    rec("evi_select_start_edges", timed(expand), int(deg) * 8 + E, "incident (eid, score) of every seed + the E-byte mask")

    # the four steps of a labelling batch as ONE hipGraph (what a steady loop replays: no host launch cost between kernels)
    def sequence():
        ops.graph_csr(ei, ptr, eptr, num_nodes=N, out=csr, workspace=csr_ws)
        dde()
        bfs()
        expand()

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        sequence()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sequence()
    seq_ms = timed(graph.replay)
    res["pipeline"] = {"what": "CSR -> DDE 2+2 -> multi-source BFS -> seed expansion captured as one hipGraph and replayed",
                       "ms_per_batch": seq_ms, "graphs_per_s": B / (seq_ms * 1e-3),
                       "sum_of_eager_kernel_legs_ms": sum(k["ms_per_batch"] for k in res["kernels"].values())}
    total_ms = sum(k["ms_per_batch"] for k in res["kernels"].values())
    dom = max(res["kernels"], key=lambda n: res["kernels"][n]["ms_per_batch"])
    dk = res["kernels"][dom]
    res["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": dk["GB_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": dk["GB_per_s"] / HBM_PEAK_GBS, "traffic": dk.get("traffic"), "traffic_source": dk.get("traffic_source"),
                       "traffic_over_algorithmic": dk.get("traffic_over_algorithmic"),
                       "algorithmic_bytes_per_batch": dk["algorithmic_bytes"], "kernel_ms_per_batch": dk["ms_per_batch"],
                       "all_four_kernels": {"algorithmic_bytes_per_batch": sum(k["algorithmic_bytes"] for k in res["kernels"].values()),
                                            "ms_per_batch": total_ms,
                                            "GB_per_s": sum(k["algorithmic_bytes"] for k in res["kernels"].values()) / (total_ms * 1e-3) / 1e9}}
    res["gpu_graphs_per_s"] = B / (seq_ms * 1e-3)  # the replayed pipeline; the eager per-kernel legs above include host launch gaps
    res["gpu_epoch_seconds"] = graphs / res["gpu_graphs_per_s"]
    if not labelling_leg:  # counter passes: only the four kernels above, so that dispatch counts map to batches
        return res
    # shortest-path labelling of the same batch: the (seed, answer) pairs' shortest-path DAG
    # edges for every graph — one BFS job per seed and per answer, evi_shortest_path_pairs in two passes — through the FLAT
    # entry point `labelling.label_pairs_flat`: the collated batch's device arrays in (edge_index, ptr, edge_ptr), the seed /
    # answer lists as host CSR, flat device arrays out, nothing read back.  The reference's per-graph 6-tuples are built
    # from the flat result only when somebody asks (`per_graph()`, timed separately, compared with the oracle below).
    from evi_rag_amd import labelling

    q_ptr_h, a_ptr_h = np.asarray(sb.q_ptr, np.int64), np.asarray(sb.a_ptr, np.int64)
    q_idx_h, a_idx_h = np.asarray(sb.q_local_indices, np.int64), np.asarray(sb.a_local_indices, np.int64)
    nptr_h, eptr_h = np.asarray(sb.ptr, np.int64), np.asarray(sb.edge_ptr, np.int64)

    def label():
        return labelling.label_pairs_flat(ei, ptr, eptr, q_ptr_h, q_idx_h, a_ptr_h, a_idx_h, node_ptr_host=nptr_h, edge_ptr_host=eptr_h)

    for _ in range(3):
        label()
    torch.cuda.synchronize(dev)
    reps = 20 if B <= 64 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        flat = label()
    t_issue = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize(dev)
    t_lab = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    lab = flat.per_graph()
    t_tuples = time.perf_counter() - t0
    # the same through the list-of-arrays mirror (GraphBatch flattening + CSR + the flat path + the tuples): what a caller
    # that holds per-graph Python lists pays
    per_graph = []
    for i in range(B):
        n0, n1, e0, e1 = int(sb.ptr[i]), int(sb.ptr[i + 1]), int(sb.edge_ptr[i]), int(sb.edge_ptr[i + 1])
        per_graph.append((n1 - n0, sb.edge_index[0, e0:e1] - n0, sb.edge_index[1, e0:e1] - n0,
                          (sb.q_local_indices[int(sb.q_ptr[i]): int(sb.q_ptr[i + 1])] - n0).tolist(),
                          (sb.a_local_indices[int(sb.a_ptr[i]): int(sb.a_ptr[i + 1])] - n0).tolist()))

    def label_lists():
        gb = labelling.GraphBatch([g[0] for g in per_graph], [g[1] for g in per_graph], [g[2] for g in per_graph], device=dev)
        return labelling.shortest_path_union_mask_by_pair_batch(gb, [g[3] for g in per_graph], [g[4] for g in per_graph])

    label_lists()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    lab2 = label_lists()
    t_lists = time.perf_counter() - t0
    same_lists = all(np.array_equal(a[0], b[0]) and list(a[3]) == list(b[3]) and list(a[5]) == list(b[5]) for a, b in zip(lab, lab2))
    res["labelling"] = {"what": "labelling.label_pairs_flat: all (seed, answer) pairs of the batch, undirected — device edge arrays + host "
                                "seed / answer CSR in, flat device arrays out (mask, pair lengths / counts / offsets / edge ids), CSR build "
                                "included, no read-back",
                        "ms_per_batch": t_lab * 1e3, "host_issue_ms_per_batch": t_issue * 1e3, "graphs_per_s": B / t_lab,
                        "pair_slots": int(flat.P), "pairs": int(sum(len(r[1]) for r in lab)),
                        "positive_edges": int(sum(int(r[0].sum()) for r in lab)),
                        "per_graph_tuples_ms": t_tuples * 1e3,
                        "list_of_arrays_mirror": {"what": "GraphBatch (flatten per-graph lists, H2D, CSR) + label_pairs_flat + per-graph 6-tuples on "
                                                          "the host (shortest_path_union_mask_by_pair_batch)",
                                                  "ms_per_batch": t_lists * 1e3, "graphs_per_s": B / t_lists, "equal_to_flat": bool(same_lists)}}
    if cpu:  # the reference's own Python / numpy algorithms, restated (oracle), on a sample of the same graphs
        from oracle import graph as og

        gl = min(4, B)
        t0 = time.perf_counter()
        for i in range(gl):
            n, src, dst, q, a = per_graph[i]
            want = og.shortest_path_union_mask_by_pair(n, src.tolist(), dst.tolist(), q, a)
            assert np.array_equal(np.asarray(want[0], bool), lab[i][0]) and all(list(want[c]) == list(lab[i][c]) for c in range(1, 6)), \
                "labelling differs from the oracle"
        cpu_lab = (time.perf_counter() - t0) / gl
        res["labelling"]["cpu_baseline"] = {"value": 1.0 / cpu_lab, "unit": "graphs/s", "cores": 1, "kind": "port",
                                            "sample": f"oracle _shortest_path_union_mask_by_pair on {gl} of the graphs, {cpu_lab * 1e3:.1f} ms/graph; results equal"}

        g = min(cpu_graphs, B)
        t0 = time.perf_counter()
        for i in range(g):
            n0, n1, e0, e1 = int(sb.ptr[i]), int(sb.ptr[i + 1]), int(sb.edge_ptr[i]), int(sb.edge_ptr[i + 1])
            src, dst = sb.edge_index[0, e0:e1] - n0, sb.edge_index[1, e0:e1] - n0
            adj = og.build_undirected_adjacency(n1 - n0, src.tolist(), dst.tolist())
            q = (sb.q_local_indices[int(sb.q_ptr[i]): int(sb.q_ptr[i + 1])] - n0).tolist()
            og.bfs_dist(n1 - n0, adj, q)
            og.node_structure_features(sb.topic_one_hot[n0:n1], np.stack([src, dst]), rounds, rounds)
            og.select_start_edges(src, dst, np.zeros(e1 - e0, np.float32), np.asarray(q), n1 - n0, 0.25, 1, None)
        cpu_s = (time.perf_counter() - t0) / g
        res["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "graphs/s", "cores": 1, "kind": "port",
                               "sample": f"oracle adjacency + BFS + DDE + seed expansion on {g} of the graphs, {cpu_s * 1e3:.1f} ms/graph"}
    return res


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU)
    BEFORE this process makes any GPU call, relay rank 0's JSON line, exit non-zero if a rank fails.  (A process that has
    initialised the GPU must never be re-exec'ed on this pool, so the parent stays GPU-free and only waits.)"""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    # a session of its own: if the ranks outlive the deadline the whole group (launcher + ranks) is killed — fresh
    # children only, this process is never replaced
    import signal

    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    limit = float(args.deadline) + 120.0  # the ranks' own deadline fires first; this one covers a launcher that hangs
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=15)
                break
            except subprocess.TimeoutExpired:
                continue
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run gave no result within {limit:.0f} s: its process group was killed\n")
        raise SystemExit(124)
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit code {proc.returncode}, {len(lines)} result lines)\n")
        raise SystemExit(proc.returncode or 1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def arm_deadline(seconds, rank):
    """A rank that is still running after `seconds` gives up loudly (exit code 124) instead of sitting in a collective
    or a rendezvous that will never complete: a daemon timer thread, `os._exit` so that no atexit / destructor can block."""
    import threading

    def fire():
        sys.stderr.write(f"bench.py: rank {rank} still running after {seconds:.0f} s (--deadline): giving up, exit code 124\n")
        sys.stderr.flush()
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def barrier(dev):
    """dist.barrier on THIS rank's GPU (RCCL: the device is named, not guessed from the rank)."""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[dev.index])
    else:
        dist.barrier()


class Ctx:
    """What every leg needs: the device, the rank layout and the loaded library."""

    def __init__(self, dev, world, rank, lib):
        self.dev, self.world, self.rank, self.lib = dev, world, rank, lib

    def agree(self, flag):
        """`flag` on one rank, its OR over the ranks on several: every decision that sets how many COLLECTIVE steps a rank runs
        (a time-based warm-up loop, say) goes through here — ranks that count differently pair their all-gathers with the
        wrong partners' and hang."""
        if self.world <= 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=self.dev)
        all_reduce_(t, dist.ReduceOp.MAX)
        return bool(t.item() > 0.0)

    def fence(self):
        torch.cuda.synchronize(self.dev)
        if self.world > 1:
            barrier(self.dev)
            torch.cuda.synchronize(self.dev)


def run_index_leg(ctx, *, N, D, Q, k, index_dtype, method, steps, warmup, seed, want_two_stage=False, sustained_s=0.0,
                  cpu_rows=0, cpu_seconds=0.0, workload=None, fp8_mfma=False, fp8_ab=False, many_query=0):
    """One index configuration end to end: build the (sharded) resident index in its storage type, W untimed + K timed
    steps through ShardedIndex.topk_async, live kernel times (hipExtLaunchKernelGGL events inside the library), planted-row
    Hits@k.  Returns the JSON object of the leg (complete on rank 0)."""
    import ctypes

    from evi_rag_amd import _lib, ops
    from evi_rag_amd.dist import ShardedIndex, _local_scan

    dev, world, rank, lib = ctx.dev, ctx.world, ctx.rank, ctx.lib
    row_begin = N * rank // world
    row_end = N * (rank + 1) // world
    t_build = time.perf_counter()
    row_scale = None
    f32_sample = None
    if index_dtype == "fp8":
        # quantised chunk by chunk as it is generated: the f32 form of the shard never exists (a 50 M x 1024 shard of a
        # 2-rank run would be 205 GB of f32); the first rows are generated once more in f32 as the overlap@k / CPU reference
        shard, row_scale = build_shard(dev, row_begin, row_end, D, seed, "fp8")
        if world == 1:
            f32_sample = build_shard(dev, row_begin, min(row_end, row_begin + max(cpu_rows, 1 << 20)), D, seed)
        elem_bytes = 1
    else:
        shard = build_shard(dev, row_begin, row_end, D, seed, torch.float16 if index_dtype == "f16" else torch.float32)
        elem_bytes = 2 if index_dtype == "f16" else 4
    n_batches = warmup + steps
    queries, gold = build_queries(dev, shard, row_begin, row_end, N, n_batches, Q, D, seed, world, row_scale=row_scale)
    torch.cuda.synchronize(dev)
    t_build = time.perf_counter() - t_build
    ws = torch.empty(ops.cosine_topk_workspace_bytes(Q, row_end - row_begin, D, k), dtype=torch.uint8, device=dev)
    if method == "two_stage" and index_dtype != "f32":
        raise SystemExit("--topk-method two_stage goes with the f32 index (its f16 shadow is built here)")
    method = method if index_dtype in ("f32", "f16") else "scan"
    shadow = ops.index_shadow_f16(shard) if method == "two_stage" else None
    xchg = host_staged_exchange(dev, world)
    index = ShardedIndex(shard, N, row_scale=row_scale, method=method, shadow=shadow, fp8_mfma=fp8_mfma and index_dtype == "fp8",
                         exchange=xchg)
    index.workspace = ws
    # every rank takes the same pipeline form: if the lane-1 communicator or the first two-lane step fails on ANY rank,
    # all ranks drop IN-PROCESS to one communicator driven from one side stream (ShardedIndex.agree_on_lanes)
    if world > 1:
        index.agree_on_lanes(queries[0], k)
    lane_fallback = index.lane_fallback

    def timed_run(index, steps=steps, warmup=warmup, queries=queries):
        """W untimed + K timed steps of `index`; returns (seconds — max over ranks, kernel ms per class, launches, last result)."""
        n_batches = queries.shape[0]

        def step(b):
            # per-shard exact top-k; for world > 1 ONE all-gather of the packed [Q, k] (score, id) records + merge,
            # pipelined with the next batch's scan (every result is complete at the closing fence)
            s, i, _ = index.topk_async(queries[b % n_batches], k)
            return s, i

        # With an exchange the timed region runs two pipelined lanes (ShardedIndex._topk_async_lanes); kernel event times
        # taken there would include the time a kernel waits for CUs behind the other lane, so the kernels are timed in
        # warm-up steps that run the shard's scan and selections alone on ONE stream, and scaled to the step count.
        lanes = bool(getattr(index, "_exchange", False) and getattr(index, "two_lanes", False))
        ms = (ctypes.c_double * 4)()
        launches = (ctypes.c_int32 * 4)()
        if lanes:
            calib = max(warmup, 1)
            _local_scan(index, queries[0], k, None, index.workspace)  # cold start (first touch of the workspace) untimed
            ctx.fence()
            t_ramp = time.perf_counter()  # ~0.1 s of sustained load first: right after start-up the same kernels run ~10 % slower
            while time.perf_counter() - t_ramp < 0.1:
                for b in range(4):
                    _local_scan(index, queries[b % n_batches], k, None, index.workspace)
                torch.cuda.synchronize(dev)
            lib.evi_timing_enable(1)
            for b in range(calib):
                _local_scan(index, queries[b % n_batches], k, None, index.workspace)  # the shard's scan + selections, one stream
            ctx.fence()
            lib.evi_timing_enable(0)
            _lib.check(lib.evi_timing_read(ms, launches, 4))
            for c in range(4):
                ms[c] = ms[c] / calib * steps
                launches[c] = int(round(launches[c] / calib * steps))
            # untimed pipelined steps: buffers and workspaces of both lanes exist before the clock starts, and the pipeline is
            # in its steady state (the first ~20 ms of two-lane steps after an idle period run ~10 % slower)
            t_ramp = time.perf_counter()
            n_pre = 0
            # (each step is a collective: how long to go on is decided by ALL ranks together — ctx.agree — never by a rank's own clock)
            while ctx.agree(n_pre < max(2, warmup) or time.perf_counter() - t_ramp < 0.05):
                for b in range(4):
                    step(n_pre + b)
                n_pre += 4
                torch.cuda.synchronize(dev)
        else:
            t_ramp = time.perf_counter()
            for b in range(warmup):
                step(b)
            torch.cuda.synchronize(dev)
            # keep warming (untimed) until ~50 ms of sustained load have passed: in the first tens of milliseconds after an
            # idle period the same kernels run up to 10 % slower, which a 3-step warm-up of a small shard does not cover
            while ctx.agree(warmup > 0 and time.perf_counter() - t_ramp < 0.05):  # collective steps: the ranks decide together
                for b in range(min(4, n_batches)):
                    step(b)
                torch.cuda.synchronize(dev)
        ctx.fence()
        if not lanes:
            lib.evi_timing_enable(1)
        t0 = time.perf_counter()
        out = None
        for b in range(warmup, warmup + steps):
            out = step(b)
        ctx.fence()
        elapsed = time.perf_counter() - t0
        if not lanes:
            lib.evi_timing_enable(0)
            _lib.check(lib.evi_timing_read(ms, launches, 4))
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            all_reduce_(t, dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, list(ms), list(launches), out

    def two_stage_bytes(shard_rows):
        kk = k + max(256, k // 2)
        # what the timed kernel (the shadow scan) moves: the f16 rows once, the queries, the stage-1 lists.  The kk
        # re-scored f32 rows per query (Q * kk * D * 4 = 74 MB at the defaults) belong to the re-scoring kernel
        return shard_rows * D * 2 + Q * D * 4 + Q * kk * 12

    elapsed, ms, launches, out = timed_run(index)
    # two_stage: a batch whose proof fails is re-done by the gated f32 scan on the device, so the results are exact
    # either way; the flag (max over ranks, a collective) only says that such repairs happened inside the timed region
    two_stage_fallback = index.two_stage_failed() if method == "two_stage" else False

    # sustained leg: >= sustained_s seconds of back-to-back steps in chunks, so that a clock / thermal droop that a
    # 0.1 s timed region cannot show becomes visible (queries/s per chunk, first vs last)
    sustained = None
    if sustained_s > 0:
        per_chunk = max(steps, int(0.25 / max(elapsed / steps, 1e-6)))
        chunks = []
        ctx.fence()
        t_all = time.perf_counter()
        lib.evi_timing_enable(1)
        n_done = 0
        while ctx.agree(time.perf_counter() - t_all < sustained_s or len(chunks) < 4):
            t0 = time.perf_counter()
            for b in range(per_chunk):
                index.topk_async(queries[(n_done + b) % n_batches], k)
            ctx.fence()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                all_reduce_(t, dist.ReduceOp.MAX)
                dt = float(t.item())
            chunks.append(Q * per_chunk / dt)
            n_done += per_chunk
            if len(chunks) >= 64:
                break
        total_s = time.perf_counter() - t_all
        lib.evi_timing_enable(0)
        sms = (ctypes.c_double * 4)()
        sln = (ctypes.c_int32 * 4)()
        _lib.check(lib.evi_timing_read(sms, sln, 4))
        sustained = {"seconds": total_s, "steps": n_done, "queries_per_s": Q * n_done / total_s,
                     "ms_per_step": total_s / n_done * 1e3, "chunk_steps": per_chunk,
                     "queries_per_s_first_chunk": chunks[0], "queries_per_s_last_chunk": chunks[-1],
                     "queries_per_s_min_chunk": min(chunks), "queries_per_s_max_chunk": max(chunks),
                     "kernel_ms_per_step": (sms[0] / n_done) if not (index._exchange and index.two_lanes) else None}

    # extra leg (f32 headline run only): the same batches through the two-stage exact scan; its last result must equal
    # the f32 scan's bit for bit
    two_stage = None
    if method == "scan" and index_dtype == "f32" and want_two_stage:
        shadow = ops.index_shadow_f16(shard)
        idx2 = ShardedIndex(shard, N, method="two_stage", shadow=shadow, exchange=xchg)
        idx2.workspace = ws
        if world > 1:
            idx2.agree_on_lanes(queries[0], k)
        e2, ms2, l2, out2 = timed_run(idx2)
        failed = idx2.two_stage_failed()
        same = bool(torch.equal(out2[0], out[0]) and torch.equal(out2[1], out[1]))
        b2 = two_stage_bytes(row_end - row_begin)
        sc2 = ms2[0] / steps
        two_stage = {
            "what": "f16 shadow of the index scanned for k + max(256, k/2) candidates per query, candidates re-scored from the "
                    "f32 rows with the scan's own MFMA chain; a per-batch gap test proves the result equals the f32 scan's, and a "
                    "batch whose proof fails is re-done on the device by the gated f32 scan (no read-back)",
            "value": Q * steps / e2, "unit": "queries/s", "ms_per_step": e2 / steps * 1e3,
            "identical_to_f32_scan": same, "proof_failed": failed, "extra_index_memory_bytes": int(shadow.numel()) * 2,
            "roofline": {"bound": "hbm", "achieved": b2 / (sc2 * 1e-3) / 1e9 if sc2 > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (b2 / (sc2 * 1e-3) / 1e9 if sc2 > 0 else 0.0) / HBM_PEAK_GBS,
                         "kernel": "k_cosine_score<F16=1> over the shadow", "algorithmic_bytes_per_step": b2,
                         "launches_per_step": l2[0] / steps, "kernel_ms_per_step": sc2,
                         "select_and_rescore_ms_per_step": ms2[1] / steps},
        }
        two_stage["roofline"].update(pmc_traffic(N, D, Q, k, world, sc2, "two_stage"))
        del shadow, idx2, out2
        torch.cuda.empty_cache()

    # A/B in this process (fp8 index): the same batches through the OTHER form of the e4m3 scan — top level = what `fp8_mfma` asks
    # for (the library default is the native fp8 matrix instruction with two e4m3 query pieces; the widening form converts the
    # rows to f16 in registers)
    fp8_native = None
    if index_dtype == "fp8" and fp8_ab:
        idx3 = ShardedIndex(shard, N, row_scale=row_scale, method="scan", fp8_mfma=not fp8_mfma, exchange=xchg)
        idx3.workspace = ws
        if world > 1:
            idx3.agree_on_lanes(queries[0], k)
        e3, ms3, l3, out3 = timed_run(idx3)
        b3 = (row_end - row_begin) * D + Q * D * 4 + Q * k * 12 + (row_end - row_begin) * 4
        sc3 = ms3[0] / steps
        inter = (out3[1].unsqueeze(2) == out[1].unsqueeze(1)).any(dim=2).float().sum(dim=1)
        fp8_native = {"what": ("e4m3 rows widened to f16 in registers, f16 MFMA against the f32 query's f16 pieces (the form the native one replaced "
                               "as the default)" if fp8_mfma else
                               "e4m3 index bytes fed to v_mfma_f32_16x16x32_fp8_fp8 as they are; the f32 query as two e4m3 pieces with "
                               "power-of-two scales (8 significant bits), one accumulator per piece — no widening work on the stream"),
                      "value": Q * steps / e3, "unit": "queries/s", "ms_per_step": e3 / steps * 1e3,
                      "overlap_at_k_vs_top_level_variant": float((inter / k).mean().item()),
                      "roofline": {"bound": "hbm", "achieved": b3 / (sc3 * 1e-3) / 1e9 if sc3 > 0 else 0.0, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": (b3 / (sc3 * 1e-3) / 1e9 if sc3 > 0 else 0.0) / HBM_PEAK_GBS,
                                   "kernel": "k_cosine_score<F16=2> (widening)" if fp8_mfma else "k_cosine_score<F16=3>",
                                   "algorithmic_bytes_per_step": b3, "kernel_ms_per_step": sc3,
                                   "traffic": None, "traffic_source": None}}
        del idx3, out3

    # BASELINE configs[3] also names Q = 512: the same resident shard, 512 queries per step, through the scan (16 passes of 32
    # queries) and through the GEMM-shaped pass (evi_cosine_topk_gemm: split-bf16 selection + exact re-scoring, same result)
    many = None
    if many_query and world == 1 and index_dtype in ("f32", "f16"):
        gq = torch.Generator(device=dev).manual_seed(seed + 77)
        rows = torch.randint(0, row_end - row_begin, (many_query,), device=dev, generator=gq)
        mq = ops.normalize_embeddings(shard[rows].float() + 0.05 * torch.randn(many_query, D, device=dev, generator=gq))

        def wall(fn, iters=4):
            fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                r = fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / iters, r

        t_scan, r_scan = wall(lambda: ops.cosine_topk(mq, shard, k, row_scale=row_scale, method="scan"))
        t_gemm, r_gemm = wall(lambda: ops.cosine_topk_gemm(mq, shard, k, row_scale=row_scale))
        found = float((r_gemm[1][:, :10] == rows.view(-1, 1)).any(dim=1).float().mean().item())
        many = {"queries_per_step": many_query,
                "scan": {"ms_per_step": t_scan * 1e3, "queries_per_s": many_query / t_scan,
                         "what": f"{(many_query + 31) // 32} passes of 32 queries over the shard"},
                "gemm": {"ms_per_step": t_gemm * 1e3, "queries_per_s": many_query / t_gemm,
                         "what": "evi_cosine_topk_gemm: one GEMM-shaped pass selects candidates (split-bf16), the scan's arithmetic "
                                 "re-scores them; proof flag read back once per call",
                         "executed_TFLOPs_lower_bound": 2.0 * many_query * (row_end - row_begin) * D / t_gemm / 1e12},
                "selection_products": ops.cosine_topk_gemm.last_products,
                "identical_ids": bool(torch.equal(r_scan[1], r_gemm[1])), "identical_scores": bool(torch.equal(r_scan[0], r_gemm[0])),
                "planted_row_in_top10": found}
        del mq, r_scan, r_gemm
    elif many_query and world > 1 and index_dtype in ("f32", "f16"):
        # N > 1: Q = 512 per step over the row-sharded index.  Every rank runs the GEMM-shaped pass (ops.cosine_topk
        # method "auto": split-bf16 selection + exact re-scoring, the scan's result bit for bit) over ITS rows for ALL the
        # queries, then the same single all-gather of the packed [Q, k] records + merge.  That pass reads its proof flag
        # back once per call (a stream synchronisation), so this leg runs the one-communicator side-stream form.
        m_steps, m_warm = max(2, steps // 4), 1
        mq, mgold = build_queries(dev, shard, row_begin, row_end, N, m_steps + m_warm, many_query, D, seed + 77, world,
                                  row_scale=row_scale)
        idxm = ShardedIndex(shard, N, row_scale=row_scale, method="auto", exchange=xchg)
        idxm.two_lanes = False
        em, msm, lm, outm = timed_run(idxm, steps=m_steps, warmup=m_warm, queries=mq)
        gm = mgold[(m_warm + m_steps - 1) % (m_steps + m_warm)].to(dev).view(many_query, 1)
        found = float((outm[1][:, :10] == gm).any(dim=1).float().mean().item())
        many = {"queries_per_step": many_query, "value": many_query * m_steps / em, "unit": "queries/s", "steps": m_steps,
                "ms_per_step": em / m_steps * 1e3, "method": "auto (GEMM-shaped pass per shard; the scan if its proof fails)",
                "pipeline": "one communicator, side stream (the pass synchronises its stream once per call)",
                "selection_products": ops.cosine_topk_gemm.last_products,
                "executed_TFLOPs_per_rank_lower_bound": 2.0 * many_query * (row_end - row_begin) * D / (em / m_steps) / 1e12,
                "planted_row_in_top10": found, "sorted_ok": bool((outm[0][:, 1:] <= outm[0][:, :-1]).all().item())}
        del mq, idxm, outm

    # Hits@k of the planted gold rows on the last timed batch (identical on every rank)
    s_last, i_last = out
    g = gold[(warmup + steps - 1) % n_batches].to(dev).view(Q, 1)
    # rank = number of returned rows that score STRICTLY higher than the planted row: 1 % of the index rows are exact duplicates,
    # and a planted row whose twin has the lower id is (correctly, by the (score desc, id asc) order) listed second
    match = i_last == g
    gold_score = s_last.gather(1, match.float().argmax(dim=1).view(Q, 1))
    rank_of_gold = torch.where(match.any(dim=1), (s_last > gold_score).sum(dim=1), torch.full((Q,), 10 ** 9, device=dev))
    hits = {f"hits@{kk}": float((rank_of_gold < kk).float().mean().item()) for kk in K_WINDOW if kk <= k}
    sorted_ok = bool((s_last[:, 1:] <= s_last[:, :-1]).all().item())

    # fp8: overlap@k of the e4m3 result with the f32 result on the same rows (the f32 rows kept aside above)
    overlap = None
    if index_dtype == "fp8" and world == 1:
        nr = f32_sample.shape[0]
        s8, i8 = ops.cosine_topk(queries[0], shard[:nr], k, row_scale=row_scale[:nr])
        s32, i32 = ops.cosine_topk(queries[0], f32_sample, k)
        inter = (i8.unsqueeze(2) == i32.unsqueeze(1)).any(dim=2).float().sum(dim=1)
        overlap = {"rows": int(nr), "overlap_at_k": float((inter / k).mean().item()), "k": k,
                   "max_abs_score_diff_on_common_rows": float((s8[:, 0] - s32[:, 0]).abs().max().item())}

    # who took part: every rank reports its device (a collective; rank 0 prints it)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": dev.index, "name": props.name, "rows": [row_begin, row_end],
          "hbm_GiB": round(props.total_memory / 2 ** 30, 1), "pid": os.getpid()}
    if hasattr(props, "uuid"):
        me["uuid"] = str(props.uuid)
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
    else:
        everyone = [me]

    result = None
    if rank == 0:
        shard_rows = row_end - row_begin
        bytes_per_step = shard_rows * D * elem_bytes + Q * D * 4 + Q * k * 12 + (shard_rows * 4 if row_scale is not None else 0)
        if method == "two_stage":
            bytes_per_step = two_stage_bytes(shard_rows)
        score_ms_per_step = ms[0] / steps
        achieved = bytes_per_step / (score_ms_per_step * 1e-3) / 1e9 if score_ms_per_step > 0 else 0.0
        result = {
            "metric": "queries/sec",
            "value": Q * steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f16 index x f32 queries (f16 MFMA, f32 accumulate)",
                      "fp8": ("e4m3 index + f32 row scale x two-piece e4m3 queries (native fp8 MFMA, f32 accumulate)" if fp8_mfma else
                              "e4m3 index + f32 row scale x f32 queries (e4m3 widened to f16 in registers, f16 MFMA, f32 accumulate)")}[index_dtype],
            "data": "synthetic",
            "config": {
                "workload": workload or ("configs[1]: WebQSP-shaped full index, bge-base dim, brute-force cosine top-k" if N < 50_000_000 else
                                         "configs[3] shape: 100 M-triple index, brute-force cosine top-k" + (" on ONE GPU" if world == 1 else "")),
                "index_rows": N,
                "dim": D,
                "queries_per_step": Q,
                "k": k,
                "index_dtype": index_dtype,
                "topk_method": method,
                "sharding": f"rows/{world}" if world > 1 else "none",
            },
            "ranks_seen": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
            "devices": everyone,
            "index_build_seconds": t_build,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "kernel": "k_cosine_score",
                "algorithmic_bytes_per_step": bytes_per_step,
                "launches_per_step": launches[0] / steps,
                "kernel_ms_per_step": score_ms_per_step,
                "select_ms_per_step": ms[1] / steps,
            },
            "hits_at_k": hits,
            "sorted_ok": sorted_ok,
        }
        result["roofline"].update(pmc_traffic(N, D, Q, k, world, score_ms_per_step, method) if index_dtype == "f32"
                                  else {"traffic": None, "traffic_source": None})
        if bool(getattr(index, "_exchange", False)) and not index.two_lanes:
            result["config"]["pipeline"] = ("one communicator: scan on the main stream, all-gather + merge of the previous batch on a "
                                            "high-priority side stream" +
                                            (f" (FALLBACK from two lanes: {lane_fallback})" if lane_fallback else ""))
        if bool(getattr(index, "_exchange", False) and getattr(index, "two_lanes", False)):
            result["config"]["pipeline"] = "two lanes (two hardware queues): scan, selections, all-gather and merge of a batch on its lane"
            result["roofline"]["timed_in"] = ("warm-up steps that run the shard's scan and selections alone on one stream: kernel event "
                                              "times taken while two lanes run include the time a kernel waits for CUs")
        if sustained is not None:
            result["sustained"] = sustained
        if two_stage is not None:
            result["two_stage"] = two_stage
        if method == "two_stage":
            result["two_stage_proof_failed"] = two_stage_fallback
        if overlap is not None:
            result["overlap_vs_f32"] = overlap
        if fp8_native is not None:
            result["fp8_widening" if fp8_mfma else "fp8_mfma"] = fp8_native
        if many is not None:
            result["many_query"] = many
        if ms[2] > ms[0]:
            # the many-query path did the work: the dominant kernel is the split-bf16 GEMM (MFMA-bound), priced by the
            # flops it executes (3 bf16 products per f32 product) against the dense bf16 peak
            gemm_ms = ms[2] / steps
            products = ops.cosine_topk_gemm.last_products or 3  # 1: plain bf16 selection, 3: split-bf16
            executed = products * 2.0 * Q * shard_rows * D / (gemm_ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "achieved": executed, "peak": 2500.0, "unit": "TFLOP/s", "frac": executed / 2500.0,
                                  "traffic": None, "traffic_source": None, "products_per_f32_product": products,
                                  "kernel": "k_gemm_nt_bf16x3 (threshold-filter epilogue)",
                                  "algorithmic_flops_per_step": 2.0 * Q * shard_rows * D, "launches_per_step": launches[2] / steps,
                                  "kernel_ms_per_step": gemm_ms, "select_ms_per_step": ms[1] / steps}
        if world == 1 and cpu_seconds > 0 and cpu_rows > 0:
            result["cpu_baseline"] = cpu_baseline(shard if f32_sample is None else f32_sample, queries, k, N, cpu_rows, cpu_seconds)
    del shard, ws, index, queries
    torch.cuda.empty_cache()
    return result


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    # stdout carries exactly ONE JSON line: native libraries that write to fd 1 (RCCL prints a version banner at
    # communicator creation) are sent to stderr for the life of the process, and the result goes to the saved fd
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # under a launcher WORLD_SIZE is authoritative
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    if REHEARSAL_BACKEND == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsal: the ranks share the device(s)
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible "
                         "(one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EVI_FORCE_EXCHANGE=1 under torch.distributed.run with ONE rank rehearses the multi-rank path (all-gather + merge
    # on the side stream) on a single GPU
    rehearse = world == 1 and os.environ.get("EVI_FORCE_EXCHANGE", "") == "1" and "MASTER_ADDR" in os.environ
    arm_deadline(float(args.deadline), rank)
    if world > 1 or rehearse:
        from datetime import timedelta

        # bounded rendezvous and collectives: a rank that never arrives makes the others fail after --dist-timeout
        # (RCCL's watchdog aborts the communicator and the process) instead of waiting for ever
        tmo = timedelta(seconds=float(args.dist_timeout))
        if REHEARSAL_BACKEND == "gloo":
            dist.init_process_group("gloo", timeout=tmo)
        else:
            # lazy communicator creation (the first collective of a group builds it: ncclCommInitRank over the store) — the
            # path every torchrun job on ROCm takes.  EVI_NCCL_EAGER_INIT=1 binds the device at init instead (device_id: eager
            # init, sub-groups by ncclCommSplit).
            if os.environ.get("EVI_NCCL_EAGER_INIT", "") == "1":
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            else:
                dist.init_process_group("nccl", timeout=tmo)

    from evi_rag_amd import _lib

    if args.graph_kernels:  # config 3 leg only
        if rank == 0:
            res = bench_graph_kernels(dev, args.graph_batch, cpu=not args.no_cpu_baseline, labelling_leg=not args.no_labelling)
            os.write(result_fd, (json.dumps(res) + "\n").encode())
        return
    ctx = Ctx(dev, world, rank, _lib.load())
    cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_seconds
    extra = world == 1 and not args.no_extra_legs
    result = run_index_leg(ctx, N=args.rows, D=args.dim, Q=args.queries, k=args.k, index_dtype=args.index_dtype,
                           method=args.topk_method, steps=args.steps, warmup=args.warmup, seed=args.seed,
                           want_two_stage=not args.no_two_stage, sustained_s=2.0 if extra else 0.0,
                           cpu_rows=args.cpu_rows, cpu_seconds=cpu_s, fp8_mfma=args.fp8_mfma)
    # ---- N > 1: the BASELINE scaling target is quoted on the 100 M-triple index (configs[3]) — measure THAT, row-sharded
    # over the ranks of this run, and configs[4] (e4m3, D = 1024, native fp8 MFMA) the same way.  Every rank takes part
    # (the legs are collective); rank 0 holds the objects.  N = 1 runs the same index whole on the one GPU (`config4_full`
    # below), so the 1 -> N ratio has both ends in the driver's own lines.
    if world > 1 and not args.no_extra_legs:
        torch.cuda.empty_cache()
        c4 = run_index_leg(
            ctx, N=args.config4_rows, D=768, Q=32, k=args.k, index_dtype="f16", method="scan", steps=args.steps,
            warmup=args.warmup, seed=3, many_query=512,
            workload=f"configs[3]: {args.config4_rows} x 768 index, f16 storage, row-sharded over {world} MI355X, per-shard exact "
                     "top-k + ONE RCCL all-gather of the packed [Q, k] records + merge on every rank; Q = 32 (and Q = 512 in many_query)")
        torch.cuda.empty_cache()
        c5 = run_index_leg(
            ctx, N=args.config4_rows, D=1024, Q=32, k=args.k, index_dtype="fp8", method="scan", steps=args.steps,
            warmup=args.warmup, seed=4, fp8_mfma=True,
            workload=f"configs[4]: {args.config4_rows} x 1024 index, OCP e4m3 storage + f32 row scale (bge-large dim), native fp8 "
                     f"MFMA scoring, row-sharded over {world} MI355X + RCCL all-gather merge")
        torch.cuda.empty_cache()
        ge = bench_eval_sharded(ctx, args.dim, graphs_per_rank=args.eval_shard_questions) if not args.no_graph_eval else None
        if rank == 0:
            result["config4_sharded"] = c4
            result["config5_sharded"] = c5
            if ge is not None:
                result["graph_eval_sharded"] = ge
    if rank == 0:
        D = args.dim
        if extra and not args.no_config4_full:
            # the one-GPU end of the 1 -> 8 target: the WHOLE configs[3] index (100 M x 768 f16 = 154 GB of the 288 GB) on this GPU
            free_b, _total = torch.cuda.mem_get_info(dev)
            need_b = args.config4_rows * 768 * 2 + (12 << 30)
            if free_b >= need_b:
                result["config4_full"] = run_index_leg(
                    ctx, N=args.config4_rows, D=768, Q=32, k=args.k, index_dtype="f16", method="scan", steps=min(args.steps, 10),
                    warmup=min(args.warmup, 2), seed=3, many_query=512 if args.config4_rows <= 100_000_000 else 0,
                    workload=f"configs[3] on ONE GPU: the whole {args.config4_rows} x 768 index, f16 storage "
                             f"({args.config4_rows * 768 * 2 / 1e9:.1f} GB resident), brute-force cosine top-k — the N = 1 end of the "
                             "1 -> 8 scaling target (`config4_sharded` in the N > 1 lines is the other)")
            else:
                result["config4_full"] = {"skipped": f"needs {need_b / 1e9:.0f} GB of free HBM, {free_b / 1e9:.0f} GB free"}
            torch.cuda.empty_cache()
        if extra:
            # BASELINE configs 4 and 5 as far as ONE GPU can show them: the shard one of 8 ranks holds of the 100 M-row index
            # (12.5 M rows), in the storage type the config names, through the same scan — each with its own roofline
            result["config4_shard"] = run_index_leg(
                ctx, N=12_500_000, D=768, Q=args.queries, k=args.k, index_dtype="f16", method="scan", steps=args.steps,
                warmup=args.warmup, seed=3, cpu_rows=1 << 18, cpu_seconds=min(cpu_s, 5.0), many_query=512,
                workload="configs[3] per-rank shard: 1/8 of the 100 M x 768 index, f16 storage (what each of 8 MI355X scans per batch)")
            result["config5_shard"] = run_index_leg(
                ctx, N=12_500_000, D=1024, Q=args.queries, k=args.k, index_dtype="fp8", method="scan", steps=args.steps,
                warmup=args.warmup, seed=4, cpu_rows=1 << 18, cpu_seconds=min(cpu_s, 5.0), fp8_mfma=True, fp8_ab=True,
                workload="configs[4] per-rank shard: 1/8 of the 100 M x 1024 index, OCP e4m3 storage + f32 row scale "
                         "(bge-large dim); overlap@k against the f32 index reported instead of bit-exactness")
            result["config3_graph_kernels"] = {
                f"batch_{b}": bench_graph_kernels(dev, b, cpu=(cpu_s > 0 and b == 32)) for b in (32, 512)}
        if world == 1 and not args.no_graph_eval:
            result["graph_eval"] = bench_graph_eval(dev, D, cpu_seconds=0.0 if args.no_cpu_baseline else 8.0)
            if extra and D != 1024:
                # the reference's DEFAULT scorer width (configs/model/retriever_module.yaml:10-17: emb_dim = hidden_dim = 1024)
                torch.cuda.empty_cache()
                result["graph_eval_1024"] = bench_graph_eval(dev, 1024, cpu_seconds=0.0, full=False)
        if world == 1 and not args.no_encode:
            torch.cuda.empty_cache()
            try:
                result["encode"] = bench_encode(dev, D)
                if extra:
                    result["config5_encode"] = bench_encode(dev, 1024, autocast="bf16", fp8_table=True)
            except ImportError as exc:  # transformers missing: the leg is informational
                result["encode"] = {"skipped": str(exc)}
        if extra and not args.no_graph_eval and not args.no_encode:
            torch.cuda.empty_cache()
            try:
                result["end_to_end"] = bench_end_to_end(dev, D, rows=args.rows, k=args.k, seed=args.seed)
            except ImportError as exc:
                result["end_to_end"] = {"skipped": str(exc)}
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if world > 1 or rehearse:
        barrier(dev)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
