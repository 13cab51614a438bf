"""The retriever evaluation step and epoch, without Lightning.

Mirror of `RetrieverModule.forward / _shared_eval_step / _update_metrics / _compute_loss_output /
test_step / on_test_epoch_end` (src/models/retriever_module.py:135-176, 251-275, 390-451): per batch
forward -> loss -> metric update (`preds=logits, target=labels>0.5, indexes=query_ids`) -> callbacks'
`on_test_batch_end`; per epoch the batch-size-weighted mean loss Lightning logs as `{split}/loss` and
the metric dict with the `{split}/` prefix.  Everything between the loader and the final dict stays
on the device; the host reads one block of loss scalars per batch and the metric states at the end.
"""
from __future__ import annotations

import time
from typing import Any, Dict, Iterable, Optional, Sequence

import torch

from . import ops
from .loss import LossOutput, RetrieverLoss
from .metrics import RetrieverMetricCollection

DEFAULT_K_VALUES = (1, 10, 25, 50, 100, 200, 300, 400, 500)  # configs/window/default.yaml:8


class RetrieverEvaluator:
    def __init__(self, model, *, loss: Optional[RetrieverLoss] = None, k_values: Sequence[int] = DEFAULT_K_VALUES,
                 split: str = "test", bridge_metrics: bool = False, feature_metrics: bool = False,
                 ablate_topic: bool = False, callbacks: Sequence[Any] = (), emit_predict_outputs: bool = False,
                 overlap_metrics: bool = True) -> None:
        if split not in ("val", "test"):
            raise ValueError(f"split must be 'val' or 'test', got {split!r}")
        self.model = model
        self.loss = loss if loss is not None else RetrieverLoss()
        self.split = split
        self.metrics = RetrieverMetricCollection(k_values, bridge_metrics=bridge_metrics, feature_metrics=feature_metrics,
                                                 prefix=f"{split}/")
        # evaluation_cfg.ablate_topic: a second forward with topic_one_hot zeroed, its own metric set (:118-120, :438-444)
        self.metrics_ablate = RetrieverMetricCollection(k_values, bridge_metrics=bridge_metrics, feature_metrics=feature_metrics,
                                                        prefix=f"{split}/ablate_topic/") if ablate_topic else None
        self.callbacks = list(callbacks)
        self.emit_predict_outputs = bool(emit_predict_outputs)
        # RetrieverOutput.edge_embeddings ([E, H] features = state_net.4, a third of the scorer's GEMM work) are consumed in
        # the evaluation only by the feature metrics (FeatureMonitor) and by callbacks that say so (`needs_edge_embeddings`);
        # test_step strips them from what it returns anyway (:118).  When nobody reads them the forward runs logits-only:
        # score_head is folded into state_net.4 (csrc/scorer.hip), same logits, no features formed.
        self.need_edge_embeddings = bool(feature_metrics) or any(getattr(cb, "needs_edge_embeddings", False) for cb in self.callbacks)
        # The loss scalars and the ranking metrics of a batch are a handful of launches that fill an eighth of the GPU (one
        # workgroup per graph: top-k, the incremental union-find of reachability@k — 0.4-0.5 ms per batch of 32) and nothing
        # but the epoch totals depends on them: with overlap_metrics they run on a side stream, UNDER the next batch's forward.
        # The batch and its output are held (self._inflight) until their side work has finished — the loader's tensors are
        # fresh allocations per batch, so nothing the side stream reads is rewritten or recycled before that — and
        # epoch_end joins the streams.  Same kernels, same per-batch results; sums of f64 batch totals in batch order.
        self.overlap_metrics = bool(overlap_metrics) and not ablate_topic
        self._side = None
        self._inflight: list = []
        self.reset()

    def reset(self) -> None:
        self.metrics.reset()
        if self.metrics_ablate is not None:
            self.metrics_ablate.reset()
        self._drain()
        self._loss_sum = 0.0          # host part (batches whose loss was read eagerly)
        self._loss_dev = None         # device accumulator: sum over batches of loss * num_graphs
        self._graphs = 0
        self._batches = 0

    def _drain(self) -> None:
        """Wait for the side-stream work of every held batch and let the caller's stream see it."""
        inflight = getattr(self, "_inflight", None)
        if not inflight:
            return
        for _batch, _output, done in inflight:
            done.synchronize()
        inflight.clear()

    def _side_stream(self, dev: torch.device):
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(dev)
        return self._side

    @staticmethod
    def _require_num_graphs(batch: Any) -> int:
        n = getattr(batch, "num_graphs", None)
        if n is None:
            ptr = getattr(batch, "ptr", None)
            if ptr is None:
                raise ValueError("Batch missing num_graphs/ptr; cannot infer the number of graphs.")
            n = int(ptr.numel() - 1)
        n = int(n)
        if n <= 0:
            raise ValueError(f"num_graphs must be positive, got {n}")
        return n

    def _compute_loss_output(self, batch: Any, output, num_graphs: int) -> LossOutput:
        targets = getattr(batch, "labels", None)
        if targets is None:
            raise ValueError("Batch missing labels required for retriever loss.")
        edge_is_near = None
        if self.loss.requires_edge_is_near:
            edge_is_near = getattr(batch, "edge_is_near", None)
            if edge_is_near is None:
                edge_is_near = ops.qa_edge_mask(batch.edge_index, int(batch.ptr[-1].item()) if hasattr(batch, "ptr") else batch.num_nodes,
                                                batch.q_local_indices, batch.a_local_indices)
                batch.edge_is_near = edge_is_near
        return self.loss(output, targets, edge_batch=output.query_ids, num_graphs=num_graphs, edge_is_near=edge_is_near)

    @torch.no_grad()
    def step(self, batch: Any, batch_idx: int = 0):
        """`_shared_eval_step` (:410-451)."""
        num_graphs = self._require_num_graphs(batch)
        output = self._forward(batch)
        overlap = self.overlap_metrics and output.logits.is_cuda and output.logits.numel() > 0
        if overlap:
            dev = output.logits.device
            main, side = torch.cuda.current_stream(dev), self._side_stream(dev)
            side.wait_stream(main)  # the forward (and the batch's collation) are in flight on the caller's stream
            with torch.cuda.stream(side):
                self._loss_and_metrics(batch, output, num_graphs)
                done = torch.cuda.Event()
                done.record(side)
            self._inflight.append((batch, output, done))
            while len(self._inflight) > 2:  # two batches may be held; the third-last has long finished
                self._inflight.pop(0)[2].synchronize()
        else:
            self._loss_and_metrics(batch, output, num_graphs)
        self._graphs += num_graphs
        self._batches += 1
        if self.metrics_ablate is not None:
            topic = getattr(batch, "topic_one_hot", None)
            if topic is None:
                raise ValueError("topic_one_hot is required for ablation metrics.")
            batch.topic_one_hot = torch.zeros_like(torch.as_tensor(topic))
            try:
                ablated = self._forward(batch)
            finally:
                batch.topic_one_hot = topic
            self._update_metrics(self.metrics_ablate, batch, ablated, num_graphs)
        for cb in self.callbacks:
            if hasattr(cb, "on_test_batch_end"):
                cb.on_test_batch_end(None, self, output, batch, batch_idx, 0)
            elif hasattr(cb, "process_batch"):  # GAgentBuilder
                cb.process_batch(batch, output)
        if self.emit_predict_outputs:
            pred = output.detach()
            pred.edge_embeddings = None
            return pred
        return None

    def _loss_and_metrics(self, batch: Any, output, num_graphs: int) -> None:
        """The per-batch loss scalars and metric updates (current stream)."""
        edge_ptr = getattr(batch, "edge_ptr", None)
        if edge_ptr is not None and isinstance(self.loss, RetrieverLoss):
            # edges grouped by graph (compute_edge_batch validated it): loss scalars stay on the device
            near = None
            if self.loss.requires_edge_is_near:
                near = getattr(batch, "edge_is_near", None)
                if near is None:
                    near = batch.edge_is_near = ops.qa_edge_mask(batch.edge_index, int(batch.num_nodes), batch.q_local_indices,
                                                                 batch.a_local_indices)
            sc = self.loss.device_scalars(output.logits, batch.labels, edge_ptr, near)
            term = sc[2] * float(num_graphs)
            self._loss_dev = term if self._loss_dev is None else self._loss_dev + term
        else:
            loss_out = self._compute_loss_output(batch, output, num_graphs)
            self._loss_sum += float(loss_out.components["infonce"] * self.loss.infonce_weight
                                    + loss_out.components["bce"] * self.loss.bce_weight) * num_graphs
        self._update_metrics(self.metrics, batch, output, num_graphs)

    def _forward(self, batch: Any):
        lite = not self.need_edge_embeddings and hasattr(self.model, "emit_edge_embeddings")
        if not lite:
            return self.model(batch)
        keep = self.model.emit_edge_embeddings
        self.model.emit_edge_embeddings = False
        try:
            return self.model(batch)
        finally:
            self.model.emit_edge_embeddings = keep

    @staticmethod
    def _update_metrics(metrics: RetrieverMetricCollection, batch: Any, output, num_graphs: int) -> None:
        """`_update_metrics` (:147-176)."""
        scores = output.logits.detach().view(-1)
        if scores.numel() == 0:
            return
        labels = batch.labels.detach().view(-1).to(dtype=torch.float32)
        if scores.numel() != labels.numel():
            raise ValueError(f"scores/labels shape mismatch: {scores.shape} vs {labels.shape}")
        query_ids = output.query_ids.detach().view(-1).to(dtype=torch.long)
        if query_ids.numel() != scores.numel():
            raise ValueError(f"query_ids/scores mismatch: {query_ids.shape} vs {scores.shape}")
        metrics.update(preds=scores, target=labels > 0.5, indexes=query_ids, batch=batch, query_ids=query_ids,
                       num_graphs=num_graphs, features=output.edge_embeddings)

    def epoch_end(self, *, sync: bool = False) -> Dict[str, float]:
        """`on_test_epoch_end` (:401-403): metric dict + the epoch loss; sync=True sums the metric states
        and the loss accumulators over ranks first (dist_reduce_fx="sum", sync_dist=True)."""
        self._drain()  # the side stream's last batches; their results are visible to this thread's reads below
        loss_sum, graphs = self._loss_sum, float(self._graphs)
        if self._loss_dev is not None:
            loss_sum += float(self._loss_dev.item())
            self._loss_dev = None
        if sync:
            from .dist import all_reduce_sum_

            self.metrics.sync()
            if self.metrics_ablate is not None:
                self.metrics_ablate.sync()
            loss_sum, graphs = all_reduce_sum_([loss_sum, graphs])
        out = {k: float(v) for k, v in self.metrics.compute().items()}
        if self.metrics_ablate is not None:
            out.update({k: float(v) for k, v in self.metrics_ablate.compute().items()})
        out[f"{self.split}/loss"] = loss_sum / max(graphs, 1.0)
        return out

    def run(self, loader: Iterable[Any], *, sync: bool = False) -> Dict[str, Any]:
        """One pass over the loader.  Returns {"metrics", "num_graphs", "seconds", "queries_per_sec"}; the
        clock brackets the loop with device synchronisation on both sides."""
        self.reset()
        for cb in self.callbacks:
            if hasattr(cb, "on_predict_start"):
                cb.on_predict_start(None, self)
        try:  # the device the model lives on, not whatever device happens to be current
            sync_dev = next(self.model.parameters()).device
        except (StopIteration, AttributeError):
            sync_dev = None
        if sync_dev is not None and sync_dev.type != "cuda":
            sync_dev = None
        torch.cuda.synchronize(sync_dev)
        t0 = time.perf_counter()
        for i, batch in enumerate(loader):
            self.step(batch, i)
        check = getattr(getattr(loader, "dataset", None), "check_deferred", None)
        if check is not None:
            check()  # deferred embedding-id range checks of the collated batches
        check = getattr(self.model, "check_deferred", None)
        if check is not None:
            check()  # relation ids outside batch.num_relations seen by a forward (scored clamped, reported here)
        metrics = self.epoch_end(sync=sync)
        torch.cuda.synchronize(sync_dev)
        seconds = time.perf_counter() - t0
        for cb in self.callbacks:
            if hasattr(cb, "on_predict_end"):
                cb.on_predict_end(None, self)
        return {"metrics": metrics, "num_graphs": self._graphs, "seconds": seconds,
                "queries_per_sec": self._graphs / seconds if seconds > 0 else 0.0}


__all__ = ["RetrieverEvaluator", "DEFAULT_K_VALUES"]
