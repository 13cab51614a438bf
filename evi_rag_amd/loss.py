"""Mirror of `RetrieverLoss` (src/losses/retriever_loss.py:22-325): same constructor checks, same
`forward` keywords, same `LossOutput` (loss tensor + components + metrics dictionaries).  The whole
loss — per-graph maxima, log-sum-exps, BCE, separation metrics — is one device pass per batch
(`evi_retriever_loss`) and ONE device→host read of 15 scalars, where the reference issues a dozen
`.item()` synchronisations.  `loss` carries the gradient w.r.t. the logits (computed by the same
call), so it can drive an optimiser step through any differentiable producer of the logits.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib, ops

_LABEL_POSITIVE_THRESHOLD = 0.5


@dataclass
class LossOutput:
    loss: torch.Tensor
    components: Dict[str, float]
    metrics: Dict[str, float]


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, total, grad):
        ctx.save_for_backward(grad)
        return total.clone()

    @staticmethod
    def backward(ctx, upstream):
        (grad,) = ctx.saved_tensors
        return grad * upstream, None, None


class RetrieverLoss(torch.nn.Module):
    """Composite loss: multi-positive InfoNCE (+ optional BCE), triple-only supervision."""

    def __init__(self, *, path_weight: float = 0.0, path_warmup_steps: int = 0, infonce_temperature: float = 1.0,
                 infonce_weight: float = 1.0, bce_weight: float = 0.0, edge_weight_near: float = 1.0,
                 edge_weight_bridge: float = 1.0) -> None:
        super().__init__()
        self.path_weight = float(path_weight)
        if self.path_weight != 0.0:
            raise ValueError("RetrieverLoss forbids path supervision; set path_weight=0 and keep path_edge_indices unset.")
        self.path_warmup_steps = int(path_warmup_steps)
        if self.path_warmup_steps != 0:
            raise ValueError("RetrieverLoss forbids path warmup; set path_warmup_steps=0.")
        self.infonce_temperature = float(infonce_temperature)
        if self.infonce_temperature <= 0.0:
            raise ValueError(f"infonce_temperature must be positive, got {self.infonce_temperature}")
        self.infonce_weight = float(infonce_weight)
        self.bce_weight = float(bce_weight)
        if self.infonce_weight < 0.0 or self.bce_weight < 0.0:
            raise ValueError("infonce_weight and bce_weight must be non-negative.")
        if self.infonce_weight == 0.0 and self.bce_weight == 0.0:
            raise ValueError("RetrieverLoss requires at least one non-zero loss weight.")
        self.edge_weight_near = float(edge_weight_near)
        self.edge_weight_bridge = float(edge_weight_bridge)
        if self.edge_weight_near <= 0.0 or self.edge_weight_bridge <= 0.0:
            raise ValueError("edge_weight_near and edge_weight_bridge must be positive.")

    @property
    def requires_edge_is_near(self) -> bool:
        return self.edge_weight_near != 1.0 or self.edge_weight_bridge != 1.0

    def forward(self, output, targets: torch.Tensor, training_step: int = 0, *, edge_batch: Optional[torch.Tensor] = None,
                num_graphs: Optional[int] = None, edge_is_near: Optional[torch.Tensor] = None,
                path_edge_indices: Optional[torch.Tensor] = None) -> LossOutput:
        logits_in = output.logits
        if logits_in is None:
            raise ValueError("RetrieverLoss requires output.logits.")
        if edge_batch is None:
            raise ValueError("RetrieverLoss requires edge_batch to define per-graph groups.")
        if path_edge_indices is not None:
            raise ValueError("RetrieverLoss forbids path_edge_indices; retriever is triple-only.")
        dev = ops._require_gpu(logits_in)
        logits = logits_in.view(-1)
        targets = targets.to(dev).view(-1).float()
        edge_batch = edge_batch.to(dev).view(-1).to(dtype=torch.long)
        if logits.numel() == 0:
            raise ValueError("RetrieverLoss received empty logits/targets; check dataset filtering.")
        if logits.numel() != targets.numel() or logits.numel() != edge_batch.numel():
            raise ValueError(f"logits/targets/edge_batch shape mismatch: {logits.shape} vs {targets.shape} vs {edge_batch.shape}")
        if num_graphs is None:
            num_graphs = int(edge_batch.max().item()) + 1
        num_graphs = int(num_graphs)
        if num_graphs <= 0:
            raise ValueError(f"num_graphs must be positive, got {num_graphs}")
        near = None
        if self.requires_edge_is_near:
            if edge_is_near is None:
                raise ValueError("RetrieverLoss requires edge_is_near when edge weights are enabled.")
            near = edge_is_near.to(device=dev, dtype=torch.bool).view(-1)
            if near.numel() != logits.numel():
                raise ValueError(f"edge_is_near length mismatch: {near.numel()} vs logits {logits.numel()}")
        x = logits.detach().to(torch.float32).contiguous()
        want_grad = bool(logits_in.requires_grad and torch.is_grad_enabled())
        scalars, grad, order = self._launch(x, targets, edge_batch, num_graphs, near, want_grad)
        v = scalars.cpu().tolist()  # the one host read
        if v[15] == 0.0:  # edges were not grouped by graph: group them (stable) and run again
            order = torch.argsort(edge_batch, stable=True)
            scalars, grad, _ = self._launch(x[order], targets[order], edge_batch[order], num_graphs,
                                            None if near is None else near[order], want_grad)
            v = scalars.cpu().tolist()
            if grad is not None:
                grad = torch.empty_like(grad).index_copy_(0, order, grad)
        total = scalars[2].to(torch.float32)
        loss = _LossFn.apply(logits, total, grad) if want_grad else total
        infonce_metrics = {"infonce_pos_edges": v[3], "infonce_neg_edges": v[4], "infonce_graphs": v[5]}
        if v[3] > 0 and v[4] > 0:  # the early return at :92-97 carries no per-graph counters
            infonce_metrics.update(infonce_graphs_no_pos=v[6], infonce_graphs_no_neg=v[7])
        return LossOutput(
            loss=loss,
            components={"infonce": v[0], "infonce_weight": self.infonce_weight, "bce": v[1], "bce_weight": self.bce_weight,
                        "path": 0.0, "path_weight": 0.0},
            metrics={"pos_prob": v[10], "neg_prob": v[11], "separation": v[12], **infonce_metrics, "bce_graphs": v[8],
                     "bce_edges": v[9], "path_graphs": 0.0})

    def device_scalars(self, logits: torch.Tensor, targets: torch.Tensor, edge_ptr: torch.Tensor,
                       edge_is_near: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The 15 loss scalars of `evi_retriever_loss` as a DEVICE tensor (index 2 = the weighted total),
        for callers that accumulate over batches and read once per epoch.  Edges must be grouped by
        graph (edge_ptr [B+1]); nothing is copied to the host."""
        dev = ops._require_gpu(logits)
        x = logits.detach().view(-1).to(torch.float32).contiguous()
        t = targets.to(dev).view(-1).float().contiguous()
        ptr = edge_ptr.to(device=dev, dtype=torch.int64).contiguous().view(-1)
        B = int(ptr.numel() - 1)
        near_u8 = None
        if self.requires_edge_is_near:
            if edge_is_near is None:
                raise ValueError("RetrieverLoss requires edge_is_near when edge weights are enabled.")
            near_u8 = edge_is_near.to(device=dev).view(-1).to(torch.uint8).contiguous()
        scalars = torch.zeros(16, dtype=torch.float64, device=dev)
        lib = _lib.load()
        ws = ops._workspace(dev, "retriever_loss", int(lib.evi_retriever_loss_workspace_bytes(B)))
        _lib.check(lib.evi_retriever_loss(
            ops._ptr(x), ops._ptr(t), ops._ptr(ptr), B, ops._ptr(near_u8), self.infonce_temperature, self.infonce_weight,
            self.bce_weight, self.edge_weight_near, self.edge_weight_bridge, scalars.data_ptr(), None, ws.data_ptr(), ws.numel(),
            ops._stream(dev)))
        return scalars

    def _launch(self, x, targets, edge_batch, num_graphs, near, want_grad):
        dev = x.device
        edge_ptr = ops.ids_to_ptr(edge_batch.clamp(0, num_graphs - 1), num_graphs)
        scalars = torch.zeros(16, dtype=torch.float64, device=dev)
        grouped = (edge_batch[1:] >= edge_batch[:-1]).all() if edge_batch.numel() > 1 else torch.ones((), dtype=torch.bool, device=dev)
        scalars[15] = grouped.to(torch.float64)
        grad = torch.empty_like(x) if want_grad else None
        near_u8 = None if near is None else near.to(torch.uint8).contiguous()
        lib = _lib.load()
        ws = ops._workspace(dev, "retriever_loss", int(lib.evi_retriever_loss_workspace_bytes(num_graphs)))
        _lib.check(lib.evi_retriever_loss(
            ops._ptr(x), ops._ptr(targets.contiguous()), ops._ptr(edge_ptr), num_graphs, ops._ptr(near_u8),
            self.infonce_temperature, self.infonce_weight, self.bce_weight, self.edge_weight_near, self.edge_weight_bridge,
            scalars.data_ptr(), ops._ptr(grad), ws.data_ptr(), ws.numel(), ops._stream(dev)))
        return scalars, grad, None


__all__ = ["RetrieverLoss", "LossOutput"]
