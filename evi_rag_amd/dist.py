"""Multi-GPU orchestration of the retrieval path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

  ShardedIndex      row-sharded embedding index: per-shard exact top-k, ONE fixed-shape all-gather
                    of the packed [Q, k] (score f32, id i64) lists, merge on every rank
                    (SURVEY.md §8e).  `topk_async` pipelines a stream of batches: the exchange of batch b
                    runs on a side HIP stream under the scan of batch b + 1.  Replaces the pickled `dist.all_gather_object` of per-sample
                    lists at src/callbacks/retriever_topk_edge_writer.py:450-462.  The message is
                    Q*k*12 bytes per rank (192 KB at Q = 32, k = 500): latency-bound, so it is a
                    single all-gather, not a ring reduction.
  shard_graphs      round-robin assignment of question graphs to ranks (graphs never span ranks, so
                    the scorer needs no exchange); metric counters are summed with one small
                    all-reduce (`RetrieverMetricCollection.sync`, the reference's dist_reduce_fx="sum").

Because a row's score is one fixed-order f32 FMA chain and ties break on the global row id, the
merged result is bit-identical for every number of shards.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_rows: int, world_size: int) -> List[int]:
    """Row r of the global index lives on the rank whose [bounds[rank], bounds[rank+1]) contains it."""
    return [num_rows * r // world_size for r in range(world_size + 1)]


def shard_graphs(num_graphs: int, rank: int, world_size: int) -> List[int]:
    """Graph ids evaluated by `rank` (round-robin, like a DistributedSampler without padding)."""
    return list(range(rank, num_graphs, world_size))


def _default_local_topk(queries, shard, k, row_id_base, row_scale=None, method="scan", fp8_mfma=False):
    from . import ops

    return ops.cosine_topk(queries, shard, k, row_id_base=row_id_base, row_scale=row_scale, method=method, fp8_mfma=fp8_mfma)


def _local_scan(self: "ShardedIndex", queries, k, out, workspace, lane: int = 0):
    """The shard's exact top-k into `out` on the current stream, without a read-back.  method "two_stage": the f16
    shadow is scanned and the f32 rows re-score (ops.cosine_topk_two_stage) with the DEVICE-SIDE fallback: a batch whose
    exactness proof fails is re-done by the gated f32 scan before the record leaves the device, so what a rank
    contributes to the exchange is always exact and the ranks need no agreement.  self.two_stage_status only records
    that a fallback ran somewhere in the stream of batches (`two_stage_failed`, informational)."""
    from . import ops

    if self.method == "two_stage":
        return ops.cosine_topk_two_stage(queries, self.shard, self.shadow, k, row_id_base=self.row_begin, out=out,
                                         status=self.two_stage_status, fallback="device",
                                         workspace=self._two_stage_workspace(queries, k, lane))
    return ops.cosine_topk(queries, self.shard, k, row_id_base=self.row_begin, row_scale=self.row_scale,
                           workspace=workspace, out=out, method=self.method, fp8_mfma=self.fp8_mfma)


def _default_merge(scores, ids):
    from . import ops

    return ops.topk_merge(scores, ids)


_LANE1_GROUPS: dict = {}


class ShardedIndex:
    """The local shard of a row-sharded, L2-normalised index plus the cross-rank top-k merge."""

    def __init__(self, local_rows: torch.Tensor, num_rows_total: int, *, group=None,
                 local_topk: Optional[Callable] = None, merge: Optional[Callable] = None,
                 row_scale: Optional[torch.Tensor] = None, method: str = "scan",
                 shadow: Optional[torch.Tensor] = None, exchange: Optional[Callable] = None, fp8_mfma: bool = False,
                 lane_groups: Optional[Sequence] = None) -> None:
        """exchange(all_records, local_record): fills `all_records` (world x record bytes, uint8, rank order) from every
        rank's `local_record`, ordered on the CURRENT stream.  Default: one `dist.all_gather_into_tensor` over `group`
        (RCCL).  Injected by the tests that run two ranks on ONE GPU, where RCCL refuses two ranks per device.
        lane_groups: two process groups over the same ranks, one per pipeline lane of `topk_async`.  A communicator must
        not be driven from two streams at once, so each lane owns one: the collectives of a lane are ordered among
        themselves (every rank alternates lanes identically) and never queue behind the other lane's.  Default: `group`
        for lane 0 and a `dist.new_group` over the same ranks for lane 1 — a collective call, so with the default every
        process of the job constructs its ShardedIndex at the same point (as `bench.py` does)."""
        self.group = group
        self.fp8_mfma = bool(fp8_mfma)  # e4m3 shard: native fp8 matrix instruction instead of widening to f16 (ops.cosine_topk)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bounds = shard_bounds(int(num_rows_total), self.world)
        self.row_begin, self.row_end = self.bounds[self.rank], self.bounds[self.rank + 1]
        if local_rows.size(0) != self.row_end - self.row_begin:
            raise ValueError(f"rank {self.rank} must hold rows [{self.row_begin}, {self.row_end}): "
                             f"{self.row_end - self.row_begin} rows, got {local_rows.size(0)}")
        self.shard = local_rows
        # local top-k: ops.cosine_topk's "scan" | "gemm" | "auto", or "two_stage" (f16 shadow of the f32 shard selects,
        # the f32 rows re-score: same result at half the bytes per batch; needs `shadow` = ops.index_shadow_f16(shard))
        self.method = method
        self.shadow = shadow
        self.two_stage_status: Optional[torch.Tensor] = None
        self._ts_ws: List[Optional[torch.Tensor]] = [None, None]  # one per pipeline lane
        if method == "two_stage":
            if row_scale is not None or shadow is None or shadow.shape != local_rows.shape or shadow.dtype != torch.float16:
                raise ValueError("method 'two_stage' needs an L2-normalised f32 shard (no row_scale) and its float16 shadow")
            self.two_stage_status = torch.zeros(1, dtype=torch.int32, device=local_rows.device)
            if local_rows.is_cuda:
                # the proof's error bound assumes rows of norm <= 1 (as cosine_topk_gemm's does): verified once per shard (one pass
                # + one read-back, cached on the tensor), whoever built the shadow
                from . import ops

                if not ops.rows_are_unit_norm(local_rows):
                    raise ValueError("method 'two_stage' needs L2-normalised rows (norm <= 1): run ops.normalize_embeddings on the shard first")
        # per-row scale of the local shard: the fused normalisation of a raw index, or the
        # dequantisation scale of an fp8 index (ops.quantize_rows_fp8)
        self.row_scale = row_scale
        if row_scale is not None and row_scale.numel() != local_rows.size(0):
            raise ValueError(f"row_scale must have one entry per local row ({local_rows.size(0)}), got {row_scale.numel()}")
        self._local_topk = local_topk or _default_local_topk
        self._merge = merge or _default_merge
        self._exchange_fn = exchange or self._rccl_all_gather
        self._gather_s: Optional[torch.Tensor] = None
        self._gather_i: Optional[torch.Tensor] = None
        # device fast path: the kernel writes scores and ids into ONE packed record, so a step costs a
        # single all-gather (192 KB per rank at Q = 32, k = 500) followed by evi_topk_merge_packed
        self._packed_ok = local_topk is None and merge is None
        self._packed_local: Optional[torch.Tensor] = None
        self._packed_all: Optional[torch.Tensor] = None
        self.workspace: Optional[torch.Tensor] = None
        self._pipe = None
        self._lanes = None
        # topk_async with an exchange: two pipeline lanes (see _topk_async) or, when False, one main stream + a side stream
        self.two_lanes = os.environ.get("EVI_TWO_LANES", "1") != "0"
        # EVI_FORCE_EXCHANGE=1 runs the exchange even with one rank (rehearses the multi-rank path on one GPU)
        self._exchange = self.world > 1 or (os.environ.get("EVI_FORCE_EXCHANGE", "") == "1" and dist.is_initialized())
        self._lane_groups = [group, group]
        # why this index runs the one-communicator side-stream pipeline although two lanes were asked for (None: it does not)
        self.lane_fallback: Optional[str] = None
        inject = os.environ.get("EVI_INJECT_LANE_FAILURE", "")  # tests: "group" | "step" | "step:<rank>"
        if lane_groups is not None:
            if len(lane_groups) != 2:
                raise ValueError("lane_groups must hold two process groups (one per pipeline lane)")
            self._lane_groups = list(lane_groups)
        elif self._exchange and self.two_lanes and dist.is_initialized() and (exchange is None or inject == "group"):
            # The second communicator is an optimisation, not a requirement: if it cannot be created (RCCL out of
            # channels / memory, a launcher that forbids a second communicator), this rank falls back to ONE communicator
            # driven from ONE side stream; `agree_on_lanes` then makes every rank take the same form.
            try:
                if inject == "group":
                    raise RuntimeError("injected: the lane-1 communicator could not be created (EVI_INJECT_LANE_FAILURE=group)")
                ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
                key = (tuple(ranks), dist.get_backend(group))
                if key not in _LANE1_GROUPS:  # one extra communicator per set of ranks for the life of the process
                    _LANE1_GROUPS[key] = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
                self._lane_groups[1] = _LANE1_GROUPS[key]
            except Exception as exc:  # noqa: BLE001 - whatever the backend raises
                self.two_lanes = False
                self.lane_fallback = f"lane-1 communicator: {type(exc).__name__}: {exc}"
        self._inject_step_failure = inject.startswith("step") and (":" not in inject or int(inject.split(":")[1]) == self.rank)

    def _rccl_all_gather(self, all_records: torch.Tensor, local_record: torch.Tensor, lane: int = 0) -> None:
        dist.all_gather_into_tensor(all_records, local_record, group=self._lane_groups[lane] if hasattr(self, "_lane_groups") else self.group)

    def _any_rank(self, flag: bool) -> bool:
        """MAX of a boolean over the ranks of `group` (a collective on the MAIN communicator; host-staged unless RCCL)."""
        if not (self.world > 1 and dist.is_initialized()):
            return bool(flag)
        if dist.get_backend(self.group) == "nccl":
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.shard.device)
        else:
            t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def agree_on_lanes(self, queries: Optional[torch.Tensor] = None, k: Optional[int] = None) -> bool:
        """Make every rank run the SAME pipeline form (a collective: every rank calls it at the same point).

        A rank whose lane-1 communicator could not be created has already dropped to the one-communicator side-stream
        form; if `queries` is given, one batch per lane is pushed through the two-lane path first (the first collective
        on a communicator is where RCCL builds its channels — and where it fails if it is going to).  If ANY rank failed
        either step, all ranks drop to the side-stream form, in-process: the lanes' buffers are released, nothing is
        re-launched.  Returns True when two lanes are in use afterwards."""
        if not self._exchange:
            return False
        failed = self.lane_fallback
        if failed is None and self.two_lanes and queries is not None and queries.is_cuda:
            try:
                for _ in range(2):  # one batch per lane
                    self.topk_async(queries, int(k))
                torch.cuda.synchronize(queries.device)
            except Exception as exc:  # noqa: BLE001
                failed = f"first two-lane step: {type(exc).__name__}: {exc}"
        anyone = self._any_rank(failed is not None or not self.two_lanes)
        if anyone:
            self.two_lanes = False
            self._lanes = None
            if self.lane_fallback is None:
                self.lane_fallback = failed or "another rank could not run two lanes"
        return bool(self.two_lanes)

    def _two_stage_workspace(self, queries: torch.Tensor, k: int, lane: int = 0) -> torch.Tensor:
        from . import _lib

        need = int(_lib.load().evi_cosine_topk_two_stage_workspace_bytes(queries.size(0), self.shard.size(0), self.shard.size(1), int(k)))
        if need == 0:
            raise ValueError(f"k + max(256, k // 2) must not exceed 2048, got k = {k}")
        if self._ts_ws[lane] is None or self._ts_ws[lane].numel() < need:
            self._ts_ws[lane] = torch.empty(need, dtype=torch.uint8, device=self.shard.device)
        return self._ts_ws[lane]

    def two_stage_failed(self) -> bool:
        """True when, on ANY rank, some batch since the last call could not be proven exact and was re-done by the gated
        f32 scan (one read-back; resets the flag).  The results are exact either way (device-side fallback); a True here
        says the two-stage scan is the wrong method for this index (clustered rows, heavy ties): every failed batch
        cost a full f32 scan on top.  With more than one rank this is a COLLECTIVE (max over ranks): every rank must
        call it, and every rank gets the same answer."""
        if self.two_stage_status is None:
            return False
        flag = self.two_stage_status
        if self.world > 1 and dist.is_initialized():
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
                bad = bool(flag.item())
            else:  # host-staged (gloo in the tests)
                host = flag.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
                bad = bool(host.item())
        else:
            bad = bool(flag.item())
        flag.zero_()
        return bad

    def topk(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Global top-k (scores [Q, k], global row ids [Q, k]) — identical on every rank."""
        if self._exchange and self._packed_ok and queries.is_cuda:
            return self._topk_packed(queries, k)
        if self.method == "two_stage" and self._local_topk is _default_local_topk:
            from . import ops

            s, i = ops.cosine_topk_two_stage(queries, self.shard, self.shadow, k, row_id_base=self.row_begin,
                                             status=self.two_stage_status,
                                             workspace=self._two_stage_workspace(queries, k))  # device-side fallback: always exact
        elif self._local_topk is _default_local_topk:
            s, i = self._local_topk(queries, self.shard, k, self.row_begin, self.row_scale, self.method, self.fp8_mfma)
        elif self.row_scale is not None:
            s, i = self._local_topk(queries, self.shard, k, self.row_begin, self.row_scale)
        else:
            s, i = self._local_topk(queries, self.shard, k, self.row_begin)
        if self.world == 1:
            return s, i
        Q = queries.size(0)
        if self._gather_s is None or self._gather_s.shape != (self.world, Q, k) or self._gather_s.device != s.device:
            self._gather_s = torch.empty((self.world, Q, k), dtype=torch.float32, device=s.device)
            self._gather_i = torch.empty((self.world, Q, k), dtype=torch.int64, device=s.device)
        # output = concatenation along dim 0 (the layout both RCCL and gloo accept)
        dist.all_gather_into_tensor(self._gather_s.view(self.world * Q, k), s.contiguous(), group=self.group)
        dist.all_gather_into_tensor(self._gather_i.view(self.world * Q, k), i.contiguous(), group=self.group)
        return self._merge(self._gather_s, self._gather_i)


def _topk_packed(self: "ShardedIndex", queries: torch.Tensor, k: int):
    from . import _lib, ops

    Q = queries.size(0)
    rec = int(_lib.load().evi_topk_packed_bytes(Q, k))
    if self._packed_local is None or self._packed_local.numel() != rec:
        self._packed_local = torch.empty(rec, dtype=torch.uint8, device=queries.device)
        self._packed_all = torch.empty(self.world * rec, dtype=torch.uint8, device=queries.device)
    s, i = ops.topk_packed_views(self._packed_local, Q, k)
    _local_scan(self, queries, k, (s, i), self.workspace)  # two_stage: exact on the device, whatever the proof said
    self._exchange_fn(self._packed_all, self._packed_local)
    return ops.topk_merge_packed(self._packed_all, self.world, Q, k)


def _topk_async(self: "ShardedIndex", queries: torch.Tensor, k: int):
    """Pipelined form of `topk` for a stream of query batches: (scores, ids, event).

    The shard scan of this batch runs on the caller's stream; the all-gather and the merge run on a side
    stream, so they overlap the scan of the NEXT batch (the exchange is latency-bound and a batch's scan
    does not depend on the previous batch's merge).  Two slots of buffers alternate: the returned tensors
    are valid once `event` has completed and until two further calls.  With one rank (no exchange) it is
    `topk` plus an already-recorded event."""
    from . import _lib, ops

    dev = queries.device
    main = torch.cuda.current_stream(dev)
    if not (self._exchange and self._packed_ok and queries.is_cuda):
        if self.method == "two_stage" and queries.is_cuda and self._local_topk is _default_local_topk:
            s, i = _local_scan(self, queries, k, None, None)  # no read-back; exact either way (device-side fallback)
        else:
            s, i = self.topk(queries, k)
        ev = torch.cuda.Event()
        ev.record(main)
        return s, i, ev
    Q = queries.size(0)
    if self.two_lanes:
        return _topk_async_lanes(self, queries, k, main)
    p = self._pipe
    # The side stream is a HIGH-PRIORITY stream: that gives it a hardware queue of its own.  As a normal-priority stream it
    # shared the main stream's queue (rocprofv3 kernel trace: same Queue_Id, the next batch's first kernel starting only
    # after the merge), i.e. the exchange ran serialised between two batches' scans: +30 us on a 0.69 ms step.
    if p is None or p["shape"] != (Q, k) or p["dev"] != dev:
        rec = int(_lib.load().evi_topk_packed_bytes(Q, k))
        mk = lambda n, dt: [torch.empty(n, dtype=dt, device=dev) for _ in range(2)]  # noqa: E731
        p = self._pipe = {"shape": (Q, k), "dev": dev, "slot": 0, "used": [False, False], "side": torch.cuda.Stream(dev, priority=-1),
                          "local": mk(rec, torch.uint8), "all": mk(self.world * rec, torch.uint8),
                          "out_s": mk((Q, k), torch.float32), "out_i": mk((Q, k), torch.int64),
                          "scan_done": [torch.cuda.Event() for _ in range(2)], "xchg_done": [torch.cuda.Event() for _ in range(2)]}
    slot = p["slot"]
    p["slot"] = slot ^ 1
    if p["used"][slot]:
        main.wait_event(p["xchg_done"][slot])  # the all-gather two batches ago has consumed this slot's record
    sv, iv = ops.topk_packed_views(p["local"][slot], Q, k)
    _local_scan(self, queries, k, (sv, iv), self.workspace)
    p["scan_done"][slot].record(main)
    side = p["side"]
    with torch.cuda.stream(side):
        side.wait_event(p["scan_done"][slot])
        self._exchange_fn(p["all"][slot], p["local"][slot])
        ops.topk_merge_packed(p["all"][slot], self.world, Q, k, out=(p["out_s"][slot], p["out_i"][slot]))
        p["xchg_done"][slot].record(side)
    p["used"][slot] = True
    return p["out_s"][slot], p["out_i"][slot], p["xchg_done"][slot]


def _topk_async_lanes(self: "ShardedIndex", queries: torch.Tensor, k: int, main):
    """Two pipeline lanes: batches alternate between two streams on two hardware queues (one normal-, one high-priority
    stream), each lane with its own scan workspace, packed record and outputs, and each batch runs ENTIRELY on its lane —
    scan, selections, all-gather, merge.  While one lane is in a selection (32 of 256 CUs busy) or in the exchange, the
    other lane's scan has the machine: measured on a 1/8 shard (2^20 rows), 0.61 ms per batch against 0.66 ms on one stream
    (0.37 against 0.43 ms for the two-stage scan); results are unchanged.  Per-kernel event times taken while two lanes
    run include the time a kernel waits for CUs, so the bench times its kernels in single-lane warm-up steps."""
    from . import _lib, ops

    dev = queries.device
    Q = queries.size(0)
    p = self._lanes
    if p is None or p["shape"] != (Q, k) or p["dev"] != dev:
        rec = int(_lib.load().evi_topk_packed_bytes(Q, k))
        mk = lambda n, dt: [torch.empty(n, dtype=dt, device=dev) for _ in range(2)]  # noqa: E731
        need = ops.cosine_topk_workspace_bytes(Q, self.shard.size(0), self.shard.size(1), k)
        ws0 = self.workspace if (self.workspace is not None and self.workspace.numel() >= need) else torch.empty(need, dtype=torch.uint8, device=dev)
        p = self._lanes = {"shape": (Q, k), "dev": dev, "slot": 0,
                           "streams": [torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)],
                           "ws": [ws0, torch.empty(need, dtype=torch.uint8, device=dev)],
                           "local": mk(rec, torch.uint8), "all": mk(self.world * rec, torch.uint8),
                           "out_s": mk((Q, k), torch.float32), "out_i": mk((Q, k), torch.int64),
                           "done": [torch.cuda.Event() for _ in range(2)]}
    slot = p["slot"]
    p["slot"] = slot ^ 1
    if slot == 1 and self._inject_step_failure:
        self._inject_step_failure = False
        raise RuntimeError("injected: the first lane-1 step failed (EVI_INJECT_LANE_FAILURE=step)")
    lane = p["streams"][slot]
    lane.wait_stream(main)  # the queries (and whatever else the caller enqueued) are ready
    queries.record_stream(lane)
    with torch.cuda.stream(lane):
        sv, iv = ops.topk_packed_views(p["local"][slot], Q, k)
        _local_scan(self, queries, k, (sv, iv), p["ws"][slot], lane=slot)
        if self._exchange_fn == self._rccl_all_gather:
            self._rccl_all_gather(p["all"][slot], p["local"][slot], lane=slot)  # the lane's own communicator
        else:
            self._exchange_fn(p["all"][slot], p["local"][slot])
        ops.topk_merge_packed(p["all"][slot], self.world, Q, k, out=(p["out_s"][slot], p["out_i"][slot]))
        p["done"][slot].record(lane)
    return p["out_s"][slot], p["out_i"][slot], p["done"][slot]


ShardedIndex._topk_packed = _topk_packed
ShardedIndex.topk_async = _topk_async


def all_reduce_sum_(values: Sequence[float], *, device: Optional[torch.device] = None, group=None) -> List[float]:
    """One small all-reduce of f64 counters (metric states)."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(values)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()


__all__ = ["ShardedIndex", "shard_bounds", "shard_graphs", "all_reduce_sum_"]
