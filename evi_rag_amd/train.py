"""The retriever training step and epoch, without Lightning.

Mirror of `RetrieverModule.training_step / configure_optimizers` (src/models/retriever_module.py:290-336, 336-372) with the
pieces of the Lightning trainer that change the numbers: `gradient_clip_val: 1.0` (configs/trainer/default.yaml:20 ->
`clip_grad_norm_` over all parameters), `strategy: ddp` (configs/trainer/ddp.yaml:4 -> gradients averaged over ranks) and the
per-epoch cosine schedule (configs/model/retriever_module.yaml:42-47).

MI355X-first layout instead of a per-parameter optimiser loop: all parameters live in ONE flat f32 buffer (the module's
tensors are views into it), their gradients in a second one, AdamW's moments in two more.  A step is then
    forward (evi_retriever_forward, intermediates kept) -> loss + dL/dlogits (evi_retriever_loss) -> backward
    (evi_retriever_backward) -> ONE all-reduce of the flat gradient over RCCL -> evi_grad_norm -> evi_adamw_step,
with nothing read back to the host: the loss scalars stay in a device accumulator that is read once per epoch, the clip
coefficient is formed on the device from the norm.  The backward is one library call, so there is no per-layer bucket to
overlap with it; at 9.4 M parameters the all-reduce is a single 38 MB ring pass (latency-bound over xGMI), issued on the
training stream right behind the backward.

Precision: `trainer.precision: bf16-mixed` -> ONE bf16 product per GEMM with f32 accumulation, forward and backward (the
single-product instantiations of the NT / TN kernels; no loss scaling needed, bf16 has f32's exponent range).  `32-true` and
`16-mixed` -> split-bf16 products with f32 accumulation (~1e-5 relative): at least the reference's precision.  Parity of the
mixed modes with the reference's torch.autocast numerics is UNPINNED (no reference fixture covers autocast runs).
"""
from __future__ import annotations

import math
import time
from typing import Any, Dict, Iterable, Mapping, Optional

import torch
import torch.distributed as dist

from . import _lib, ops
from .loss import RetrieverLoss


class FlatAdamW:
    """torch.optim.AdamW's update over one flat buffer (csrc/optim.hip).  `params` are re-pointed into the buffer; their
    `.grad`s are views of `self.grad` (autograd accumulates into them in place)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], *, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2) -> None:
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on the MI355X only (there is no CPU fallback)")
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("FlatAdamW needs f32 parameters on one device")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("Invalid AdamW hyper-parameter")
        self.lr, self.initial_lr, self.betas, self.eps, self.weight_decay = float(lr), float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        # every parameter starts on a 16-byte boundary (the kernels use 16-byte accesses on the flat buffers)
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        self._offsets = offs

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self._offsets):  # a `.grad = None` by someone else would detach the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def step(self, *, grad_scale: float = 1.0, max_norm: Optional[float] = None) -> None:
        """One update from `self.grad * grad_scale`, clipped to `max_norm` (total L2 norm) when given."""
        lib = _lib.load()
        dev = self.flat.device
        s = ops._stream(dev)
        norm_ptr = None
        if max_norm is not None:
            ws = ops._workspace(dev, "grad_norm", int(lib.evi_grad_norm_workspace_bytes(self.numel)))
            _lib.check(lib.evi_grad_norm(self.grad.data_ptr(), self.numel, float(grad_scale), self.grad_norm.data_ptr(),
                                         ws.data_ptr(), ws.numel(), s))
            norm_ptr = self.grad_norm.data_ptr()
        self.step_count += 1
        _lib.check(lib.evi_adamw_step(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                      self.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                      self.step_count, float(grad_scale), norm_ptr, float(max_norm or 0.0), s))
        # the kernel wrote the parameters behind torch's back: bump their version counters (no launch) so that anything keyed
        # on them — Retriever's prepared-weights cache, autograd's saved-tensor checks — sees the change
        torch.autograd.graph.increment_version(self.params)

    def state_dict(self) -> Dict[str, Any]:
        return {"step": self.step_count, "lr": self.lr, "initial_lr": self.initial_lr, "exp_avg": self.exp_avg.clone(),
                "exp_avg_sq": self.exp_avg_sq.clone()}

    def load_state_dict(self, sd: Mapping[str, Any]) -> None:
        self.step_count, self.lr, self.initial_lr = int(sd["step"]), float(sd["lr"]), float(sd.get("initial_lr", sd["lr"]))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def setup_optimizer(module: torch.nn.Module, optimizer_cfg: Optional[Mapping[str, Any]]) -> FlatAdamW:
    """`setup_optimizer` (src/utils/optimization.py:20-35) for the optimiser the retriever trains with (`type: adamw`)."""
    if module is None:
        raise ValueError("setup_optimizer requires a valid nn.Module.")
    cfg = dict(optimizer_cfg or {})
    opt_type = str(cfg.pop("type", cfg.pop("name", "adamw"))).lower()
    if cfg.pop("param_groups", None):
        raise NotImplementedError("param_groups are not supported by the flat optimiser (one group: all parameters)")
    if opt_type != "adamw":
        raise ValueError(f"Unsupported optimizer type '{opt_type}' (this backend implements the retriever's default, adamw).")
    return FlatAdamW(module.parameters(), **cfg)


class CosineSchedule:
    """torch.optim.lr_scheduler.CosineAnnealingLR in closed form, stepped per epoch (retriever_module.py:341-354)."""

    def __init__(self, optimizer: FlatAdamW, t_max: int = 10, eta_min: float = 0.0) -> None:
        self.opt, self.t_max, self.eta_min, self.last_epoch = optimizer, int(t_max), float(eta_min), 0

    def step(self) -> None:
        self.last_epoch += 1
        base = self.opt.initial_lr
        self.opt.lr = self.eta_min + (base - self.eta_min) * (1.0 + math.cos(math.pi * self.last_epoch / self.t_max)) / 2.0


def matmul_precision_for(precision: Any) -> str:
    """Lightning's `trainer.precision` (configs/trainer/default.yaml:13-14) -> Retriever.matmul_precision.

    `bf16-mixed` (what the reference's comment recommends where the GPU has bf16) -> "bf16": the large products of the forward
    and the backward multiply one bf16 product with f32 accumulation; results, LayerNorm, GELU, the loss and the optimiser stay
    f32 (bf16 autocast rounds the Linear results to bf16 as well, so this is at least its arithmetic).  Everything else —
    `32-true`, and the reference's default `16-mixed` — -> "split" (three bf16 products, f32-grade): f16 autocast is not
    mirrored (it needs Lightning's dynamic loss scaling to keep gradients inside f16's range; the split products are more
    precise than f16 and need none)."""
    p = str(precision).strip().lower()
    if p in ("bf16-mixed", "bf16", "bf16-true"):
        return "bf16"
    if p in ("32", "32-true", "16", "16-mixed", "16-true", "64", "64-true", "none"):
        return "split"
    raise ValueError(f"Unsupported trainer.precision {precision!r}")


class RetrieverTrainer:
    def __init__(self, model, *, loss: Optional[RetrieverLoss] = None, optimizer_cfg: Optional[Mapping[str, Any]] = None,
                 scheduler_cfg: Optional[Mapping[str, Any]] = None, gradient_clip_val: Optional[float] = 1.0,
                 process_group=None, precision: Any = None) -> None:
        self.model = model
        if precision is not None and matmul_precision_for(precision) == "bf16":
            model.matmul_precision = "bf16"  # other values leave the model's own setting (`model.retriever.matmul_precision`) alone
        self.loss = loss if loss is not None else RetrieverLoss()
        self.optimizer = setup_optimizer(model, optimizer_cfg if optimizer_cfg is not None else
                                         {"type": "adamw", "lr": 1e-3, "weight_decay": 1e-4})
        sched = dict(scheduler_cfg or {})
        stype = str(sched.get("type", "") or "").lower()
        if stype in ("", "none"):
            self.scheduler = None
        elif stype == "cosine":
            self.scheduler = CosineSchedule(self.optimizer, t_max=int(sched.get("t_max", 10)), eta_min=float(sched.get("eta_min", 0.0)))
        else:
            raise ValueError(f"Unsupported scheduler type '{stype}' (cosine or none).")
        self.gradient_clip_val = None if not gradient_clip_val else float(gradient_clip_val)
        self.group = process_group
        self.global_step = 0
        self.current_epoch = 0
        self._loss_dev = None   # sum over steps of loss * num_graphs (device, f64)
        self._graphs = 0
        self._ungrouped = None  # OR of "edges were not grouped by graph" (checked every `check_every` steps and at epoch end)
        # the sticky device flags (ungrouped edges; relation ids / seed / answer indices out of range, which the forward clamps)
        # are read back every `check_every` optimiser steps: a malformed loader is reported after at most that many updates, at
        # the price of one host-device synchronisation per `check_every` steps (0: at epoch end only)
        self.check_every = 64

    # -- distributed ----------------------------------------------------------------------------------------
    def _world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _all_reduce_grads(self) -> None:
        g = self.optimizer.grad
        if dist.get_backend(self.group) == "nccl":  # RCCL: on the device, on the training stream
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        else:  # gloo (tests): staged through the host
            h = g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            g.copy_(h)

    # -- one step -------------------------------------------------------------------------------------------
    def training_step(self, batch: Any, batch_idx: int = 0) -> torch.Tensor:
        """`training_step` (:290-336) + backward + gradient sync + clip + optimiser step.  Returns the loss as a device
        scalar (nothing is read back)."""
        model = self.model
        model.train()
        num_graphs = int(getattr(batch, "num_graphs", 0) or (batch.ptr.numel() - 1))
        if num_graphs <= 0:
            raise ValueError(f"num_graphs must be positive, got {num_graphs}")
        if int(batch.edge_index.size(1)) == 0:
            raise ValueError("training batch without edges: nothing to score, no gradient to take (filter edge-less samples out of the split)")
        targets = getattr(batch, "labels", None)
        if targets is None:
            raise ValueError("Batch missing labels required for retriever loss.")
        self.optimizer.zero_grad()
        # the loss reads the logits only: the [E, H] edge features (state_net.4 on the combined rows, a GEMM per chunk) are
        # not formed — score_head is folded into state_net.4, same logits and gradients
        keep = getattr(model, "emit_edge_embeddings", None)
        if keep is not None:
            model.emit_edge_embeddings = False
        try:
            output = model(batch)
        finally:
            if keep is not None:
                model.emit_edge_embeddings = keep
        logits = output.logits
        near = None
        if self.loss.requires_edge_is_near:
            near = getattr(batch, "edge_is_near", None)
            if near is None:
                near = batch.edge_is_near = ops.qa_edge_mask(batch.edge_index, int(batch.num_nodes), batch.q_local_indices,
                                                             batch.a_local_indices)
            near = near.to(device=logits.device, dtype=torch.bool).view(-1)
        x = logits.detach().to(torch.float32).contiguous().view(-1)
        scalars, grad, _ = self.loss._launch(x, targets.to(x.device).view(-1).float(), output.query_ids.view(-1).to(torch.long),
                                             num_graphs, near, True)
        bad = scalars[15] == 0.0  # edges not grouped by graph: the loss kernel's groups would be wrong (loader contract)
        self._ungrouped = bad if self._ungrouped is None else (self._ungrouped | bad)
        # one backward per zero_grad and `.grad`s that are views of the flat buffer cleared above: for THIS backward the kernels
        # write the gradients in place (overwrite semantics — so the switch does not outlive the call)
        in_place = hasattr(model, "_launch_backward")
        if in_place:
            model.grads_in_place = True
        try:
            logits.backward(grad.view_as(logits))
        finally:
            if in_place:
                model.grads_in_place = False
        world = self._world()
        if world > 1:
            self._all_reduce_grads()
        self.optimizer.step(grad_scale=1.0 / world, max_norm=self.gradient_clip_val)
        term = scalars[2] * float(num_graphs)
        self._loss_dev = term if self._loss_dev is None else self._loss_dev + term
        self._graphs += num_graphs
        self.global_step += 1
        if self.check_every and self.global_step % int(self.check_every) == 0:
            self._check_sticky_flags()
        return scalars[2]

    def _check_sticky_flags(self) -> None:
        if self._ungrouped is not None and bool(self._ungrouped.item()):
            raise ValueError("edge_batch is not sorted by graph in a training batch; the loader must group edges by graph.")
        check = getattr(self.model, "check_deferred", None)
        if check is not None:
            check()  # deferred range checks of the forwards so far (relation ids, seed / answer indices)

    def _common_steps(self, loader) -> Optional[int]:
        """With more than one rank: the smallest number of batches any rank's loader holds this epoch (one small MIN
        all-reduce), None otherwise or when the loader has no length."""
        if self._world() <= 1 or not hasattr(loader, "__len__"):
            return None
        n = torch.tensor([len(loader)], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            n = n.to(self.optimizer.flat.device)
        dist.all_reduce(n, op=dist.ReduceOp.MIN, group=self.group)
        return int(n.item())

    def on_train_epoch_end(self) -> Dict[str, float]:
        """The epoch's `train/loss` (batch-size-weighted mean, summed over ranks like sync_dist=True) + scheduler step."""
        loss_sum = float(self._loss_dev.item()) if self._loss_dev is not None else 0.0
        self._check_sticky_flags()
        graphs = float(self._graphs)
        if self._world() > 1:
            from .dist import all_reduce_sum_

            loss_sum, graphs = all_reduce_sum_([loss_sum, graphs], group=self.group)
        self._loss_dev, self._graphs, self._ungrouped = None, 0, None
        if self.scheduler is not None:
            self.scheduler.step()
        self.current_epoch += 1
        return {"train/loss": loss_sum / max(graphs, 1.0), "lr": self.optimizer.lr}

    def fit(self, loader: Iterable[Any], max_epochs: int = 1) -> Dict[str, Any]:
        """`max_epochs` passes over the loader.  Returns the per-epoch logs and the wall clock (device-synchronised)."""
        logs = []
        dev = self.optimizer.flat.device
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        steps = 0
        for _ in range(int(max_epochs)):
            if hasattr(loader, "set_epoch"):
                loader.set_epoch(self.current_epoch)
            limit = self._common_steps(loader)
            for i, batch in enumerate(loader):
                if limit is not None and i >= limit:
                    break  # another rank's share of the split is one batch shorter: every rank stops there (the gradient
                           # all-reduce is a collective — an extra step on one rank would wait forever)
                self.training_step(batch, i)
                steps += 1
            logs.append(self.on_train_epoch_end())
        torch.cuda.synchronize(dev)
        return {"epochs": logs, "steps": steps, "seconds": time.perf_counter() - t0}

    # -- checkpoint / resume ----------------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, Any]:
        return {"model": {k: v.clone() for k, v in self.model.state_dict().items()}, "optimizer": self.optimizer.state_dict(),
                "global_step": self.global_step, "current_epoch": self.current_epoch,
                "scheduler_last_epoch": self.scheduler.last_epoch if self.scheduler is not None else 0}

    def load_state_dict(self, sd: Mapping[str, Any]) -> None:
        self.model.load_state_dict(sd["model"], strict=True)  # copies into the flat views
        self.optimizer.load_state_dict(sd["optimizer"])
        self.global_step, self.current_epoch = int(sd["global_step"]), int(sd["current_epoch"])
        if self.scheduler is not None:
            self.scheduler.last_epoch = int(sd.get("scheduler_last_epoch", 0))


def _save_checkpoint(self, path, callbacks: Optional[Mapping[str, Any]] = None) -> None:
    """A checkpoint in Lightning's layout as far as the reference reads it: `state_dict` with the retriever under the
    `model.` prefix of `RetrieverModule` (retriever_module.py:59) — what `src/eval.py:_load_checkpoint_strict` (:80-111) and
    `evi_rag_amd.eval` load — plus `epoch`, `global_step` and this trainer's optimiser / schedule state for resuming.
    Tensors only: loadable with `torch.load(weights_only=True)`."""
    sd = self.state_dict()
    blob = {"state_dict": {f"model.{k}": v.detach().cpu() for k, v in sd["model"].items()}, "epoch": self.current_epoch,
            "global_step": self.global_step,
            "optimizer_states": [{k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd["optimizer"].items()}],
            "lr_schedulers": [{"last_epoch": sd["scheduler_last_epoch"]}]}
    if callbacks:
        # the run's selection state (Lightning keeps it under the same key): best score / path of model_checkpoint, best score /
        # wait count / latched stop of early_stopping — plain scalars and strings, so weights_only=True still loads the file
        blob["callbacks"] = {str(k): (v if isinstance(v, (int, float, str, bool)) or v is None else str(v)) for k, v in callbacks.items()}
    torch.save(blob, str(path))


def _load_checkpoint(self, path) -> Dict[str, Any]:
    """Restores weights, optimiser, schedule, epoch and step; returns the `callbacks` entry of the file ({} if none)."""
    blob = torch.load(str(path), map_location="cpu", weights_only=True)
    self.load_state_dict({"model": {k[len("model."):]: v for k, v in blob["state_dict"].items() if k.startswith("model.")},
                          "optimizer": blob["optimizer_states"][0], "global_step": blob["global_step"], "current_epoch": blob["epoch"],
                          "scheduler_last_epoch": blob["lr_schedulers"][0]["last_epoch"]})
    return dict(blob.get("callbacks") or {})


RetrieverTrainer.save_checkpoint = _save_checkpoint
RetrieverTrainer.load_checkpoint = _load_checkpoint

__all__ = ["RetrieverTrainer", "FlatAdamW", "CosineSchedule", "setup_optimizer"]
