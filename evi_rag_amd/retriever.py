"""Host-side mirror of `src.models.components.retriever.Retriever` (the reference's `_target_` at
configs/model/retriever_module.yaml:8-25) backed by libevi_hip.so.

Same constructor kwargs, same `state_dict` keys and shapes (so `load_state_dict(strict=True)` of a
reference checkpoint works, src/eval.py:80-111), same `forward(batch) -> RetrieverOutput` and
`extract_edge_tokens(batch)` contracts, same error types and messages.  The arithmetic runs in
`evi_retriever_forward`; this module only owns the parameters and marshals pointers.

Scope: the evaluation path (eval mode, no autograd).  Training (dropout, hide-and-seek bias,
backward) is out of scope for this build and raises.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
from torch import nn

from . import _lib, ops

_NUM_TOPICS_BINARY = 2
_MAX_DDE_ROUNDS = 4
_DIRECTION_CODE = {"bidirectional": 0, "forward": 1, "backward": 2}
_MATMUL_PRECISION_CODE = {"split": 0, "bf16": 1, "f16x2": 2}  # EviRetrieverBatch.matmul_precision


@dataclass
class RetrieverOutput:
    """reference: RetrieverOutput, src/models/components/retriever.py:80-99."""

    logits: torch.Tensor
    query_ids: torch.Tensor
    relation_ids: Optional[torch.Tensor] = None
    logits_fwd: Optional[torch.Tensor] = None
    logits_bwd: Optional[torch.Tensor] = None
    edge_embeddings: Optional[torch.Tensor] = None

    def detach(self) -> "RetrieverOutput":
        d = lambda t: t.detach() if t is not None else None  # noqa: E731
        return RetrieverOutput(d(self.logits), d(self.query_ids), d(self.relation_ids), d(self.logits_fwd),
                               d(self.logits_bwd), d(self.edge_embeddings))


class EmbeddingProjector(nn.Module):
    """Linear + Tanh; parameters live under `.network.0` as in
    src/models/components/projections.py:9-40.  forward() runs on the MFMA f32 GEMM."""

    def __init__(self, output_dim: int, *, input_dim: Optional[int] = None, finetune: bool = False) -> None:
        super().__init__()
        if input_dim is None or input_dim <= 0:
            raise ValueError("EmbeddingProjector requires a positive input_dim; lazy layers are disallowed.")
        self.network = nn.Sequential(nn.Linear(input_dim, output_dim), nn.Tanh())
        if not finetune:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        lin = self.network[0]
        return ops.linear_act(tensor, lin.weight.detach(), lin.bias.detach(), "tanh")


class DDE(nn.Module):
    """Parameter-free holder of the DDE round counts (src/models/components/graph.py:26-40)."""

    def __init__(self, num_rounds: int = 2, num_reverse_rounds: int = 2) -> None:
        super().__init__()
        self.num_rounds = int(max(0, num_rounds))
        self.num_reverse_rounds = int(max(0, num_reverse_rounds))
        if self.num_rounds > _MAX_DDE_ROUNDS or self.num_reverse_rounds > _MAX_DDE_ROUNDS:
            raise ValueError(
                f"DDE supports at most {_MAX_DDE_ROUNDS} rounds per direction; "
                f"got num_rounds={self.num_rounds}, num_reverse_rounds={self.num_reverse_rounds}."
            )


def compute_edge_batch(edge_index: torch.Tensor, *, node_ptr: torch.Tensor, num_graphs: int,
                       device: Optional[torch.device] = None, debug_batch: object = None):
    """(edge_batch, edge_ptr) with the reference's validation errors.
    reference: compute_edge_batch, src/utils/graph_utils.py:50-104."""
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    if node_ptr.numel() != num_graphs + 1:
        raise ValueError(f"node_ptr length mismatch: got {node_ptr.numel()} expected {num_graphs + 1}")
    eb, eptr, status = ops.edge_batch(edge_index, node_ptr)
    st = int(status.item())
    if st & 1:
        raise ValueError(f"edge_batch contains out-of-range indices; num_graphs={num_graphs}.")
    if st & 2:
        raise ValueError("edge_index crosses graph boundaries; head/tail graph assignments differ. ")
    if st & 4:
        raise ValueError(
            "edge_batch is not non-decreasing along the flattened edge list, which breaks per-graph slicing; "
            "Ensure edges are concatenated per-graph (PyG Batch)."
        )
    return eb, eptr


def compute_qa_edge_mask(edge_index: torch.Tensor, *, num_nodes: int, q_local_indices: torch.Tensor,
                         a_local_indices: torch.Tensor, deferred_status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference: compute_qa_edge_mask, src/utils/graph_utils.py:107-153."""
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    if int(num_nodes) <= 0:
        raise ValueError(f"num_nodes must be positive, got {num_nodes}")
    q = torch.as_tensor(q_local_indices, dtype=torch.long)
    a = torch.as_tensor(a_local_indices, dtype=torch.long)
    return ops.qa_edge_mask(edge_index, int(num_nodes), q, a, deferred_status=deferred_status)


class Retriever(nn.Module):
    """Drop-in for src.models.components.retriever.Retriever: evaluation, and — in train() mode — an autograd node whose backward
    is evi_retriever_backward (the reference's dropout and hide-and-seek included)."""

    def __init__(
        self,
        emb_dim: int,
        hidden_dim: int,
        topic_pe: bool = True,
        num_topics: int = 2,
        dde_cfg: Optional[Dict[str, int]] = None,
        dropout_p: float = 0.1,
        core_mode: str = "geometry",
        direction_mode: str = "bidirectional",
        hide_seek_cfg: Optional[Dict[str, Any]] = None,
        dedupe_relations: bool = True,
        emit_edge_embeddings: bool = True,
        matmul_precision: str = "split",
        **_: Any,
    ) -> None:
        super().__init__()
        # False: forward() returns edge_embeddings=None and the library folds score_head into state_net.4
        # (logits-only evaluation, e.g. predict_step drops edge_embeddings anyway: retriever_module.py:277-285)
        self.emit_edge_embeddings = bool(emit_edge_embeddings)
        self.cache_prepared_weights = True  # eval mode: keep the weight-derived pieces of the forward across calls
        # None: the forward is a differentiable autograd node in train() mode only (evaluation never builds a graph, whatever
        # torch.is_grad_enabled() says); True / False force it — True gives eval-mode gradients (the parity tests use that)
        self.differentiable: Optional[bool] = None
        # training: keep the per-edge intermediates of the forward for the backward (memory for time: ≈ 16 (D + H) bytes per edge)
        self.keep_forward_intermediates = True
        # "split" (default): every large product as three bf16 MFMA products of the split f32 operands (f32-grade results);
        # "bf16": one bf16 product with f32 accumulation and results, forward and backward — the arithmetic class of
        # Lightning's `precision: bf16-mixed` (configs/trainer/default.yaml:13-14), opt-in for training runs;
        # "f16x2" (evaluation only): two f16 products — activations as f16 hi + lo, weights rounded once to f16 (2^-12 relative,
        # finer than the TF32 the reference's CUDA run uses): two thirds of the matrix work of "split", logits within ~1e-4
        self.matmul_precision = str(matmul_precision)
        if self.matmul_precision not in _MATMUL_PRECISION_CODE:
            raise ValueError(f"matmul_precision must be one of {sorted(_MATMUL_PRECISION_CODE)}, got {matmul_precision!r}")
        self.emb_dim = int(emb_dim)
        self.hidden_dim = int(hidden_dim)
        self.use_topic_pe = bool(topic_pe)
        if not self.use_topic_pe:
            raise ValueError("topic_pe must be enabled; retriever requires topic_one_hot + DDE.")
        self.num_topics = int(num_topics)
        if self.num_topics != _NUM_TOPICS_BINARY:
            raise ValueError(f"num_topics must be {_NUM_TOPICS_BINARY} (seed vs non-seed), got {self.num_topics}")
        if str(core_mode or "").strip().lower() not in {"geometry", "structured", "full"}:
            raise ValueError(f"core_mode must be 'geometry' for DDE-based retriever, got {core_mode!r}")
        mode = str(direction_mode or "").strip().lower()
        if mode not in _DIRECTION_CODE:
            raise ValueError(
                "direction_mode must be one of {'bidirectional', 'forward', 'backward'}, " f"got {direction_mode!r}."
            )
        self.direction_mode = mode
        self.dedupe_relations = bool(dedupe_relations)

        # construction order == the reference's, so the same torch seed gives the same init
        D, H = self.emb_dim, self.hidden_dim
        self.entity_proj = EmbeddingProjector(D, input_dim=D, finetune=True)
        self.relation_proj = EmbeddingProjector(D, input_dim=D, finetune=True)
        self.query_proj = EmbeddingProjector(D, input_dim=D, finetune=True)
        self.non_text_entity_emb = nn.Embedding(1, D)
        self.dde = DDE(**(dde_cfg or {}))
        rounds, rev = self.dde.num_rounds, self.dde.num_reverse_rounds
        self._topic_struct_dim = self.num_topics * (1 + rounds + rev)
        self.register_buffer("parity_meta", torch.tensor([1, self.num_topics, rounds, rev], dtype=torch.long))
        self.q_gate = nn.Sequential(nn.Linear(D, D), nn.Sigmoid())
        self.q_bias = nn.Sequential(nn.Linear(D, D), nn.Tanh())
        self.struct_proj = nn.Sequential(nn.Linear(2 * self._topic_struct_dim, D), nn.LayerNorm(D), nn.GELU())
        self.struct_gate_net = nn.Sequential(nn.Linear(D, 1), nn.Sigmoid())
        self.state_net = nn.Sequential(nn.Linear(3 * D + 1, H), nn.LayerNorm(H), nn.GELU(), nn.Dropout(dropout_p),
                                       nn.Linear(H, H))
        self.score_head = nn.Linear(H, 1)
        cfg = hide_seek_cfg or {}
        self.hide_seek_enabled = bool(cfg.get("enabled", False))
        self.hide_seek_apply_in_eval = bool(cfg.get("apply_in_eval", False))
        for name in ("p_near", "p_far"):
            prob = float(cfg.get(name, 0.0))
            if prob < 0.0 or prob > 1.0:
                raise ValueError(f"hide_seek_cfg.{name} must be in [0, 1], got {prob}")
            setattr(self, f"hide_seek_{name}", prob)
        for name in ("bias_near", "bias_far"):
            bias = float(cfg.get(name, 0.0))
            if bias > 0.0:
                raise ValueError(f"hide_seek_cfg.{name} must be <= 0 (penalty), got {bias}")
            setattr(self, f"hide_seek_{name}", bias)

    # ---- public API ----------------------------------------------------------------------------------
    def forward(self, batch: Any) -> RetrieverOutput:
        output, _ = self._forward_impl(batch, return_features=False)
        return output

    def extract_edge_tokens(self, batch: Any) -> torch.Tensor:
        _, features = self._forward_impl(batch, return_features=True)
        if features is None:
            raise RuntimeError("extract_edge_tokens expected non-empty features but got None.")
        return features

    # ---- internals -----------------------------------------------------------------------------------
    def _weights_struct(self) -> _lib.EviRetrieverWeights:
        p = lambda t: t.detach().data_ptr()  # noqa: E731
        w = _lib.EviRetrieverWeights()
        w.emb_dim, w.hidden_dim, w.num_topics = self.emb_dim, self.hidden_dim, self.num_topics
        w.dde_rounds, w.dde_reverse_rounds = self.dde.num_rounds, self.dde.num_reverse_rounds
        for name, mod in (("entity", self.entity_proj), ("relation", self.relation_proj), ("query", self.query_proj)):
            setattr(w, f"{name}_w", p(mod.network[0].weight))
            setattr(w, f"{name}_b", p(mod.network[0].bias))
        w.non_text_emb = p(self.non_text_entity_emb.weight)
        w.q_gate_w, w.q_gate_b = p(self.q_gate[0].weight), p(self.q_gate[0].bias)
        w.q_bias_w, w.q_bias_b = p(self.q_bias[0].weight), p(self.q_bias[0].bias)
        w.struct_w, w.struct_b = p(self.struct_proj[0].weight), p(self.struct_proj[0].bias)
        w.struct_ln_w, w.struct_ln_b = p(self.struct_proj[1].weight), p(self.struct_proj[1].bias)
        w.struct_gate_w, w.struct_gate_b = p(self.struct_gate_net[0].weight), p(self.struct_gate_net[0].bias)
        w.state0_w, w.state0_b = p(self.state_net[0].weight), p(self.state_net[0].bias)
        w.state_ln_w, w.state_ln_b = p(self.state_net[1].weight), p(self.state_net[1].bias)
        w.state4_w, w.state4_b = p(self.state_net[4].weight), p(self.state_net[4].bias)
        w.score_w, w.score_b = p(self.score_head.weight), p(self.score_head.bias)
        w.prepared = None
        return w

    def _prepared_weights(self, w: "_lib.EviRetrieverWeights", dev: torch.device) -> Optional[torch.Tensor]:
        """In eval mode the weight-derived pieces of the forward (column blocks of state_net.0, folded head, bf16 planes of the
        GEMM weights) are made once and kept until a parameter changes (tensor version counters / storage / device), instead
        of ~25 small launches per forward.  Training mode (weights change every step) never caches."""
        if self.training or not self.cache_prepared_weights:
            return None
        params = list(self.parameters())
        key = (str(dev),) + tuple((t.data_ptr(), t._version) for t in params)
        cached = getattr(self, "_prep_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        lib = _lib.load()
        need = int(lib.evi_retriever_prepare_bytes(self.emb_dim, self.hidden_dim, self.dde.num_rounds, self.dde.num_reverse_rounds))
        buf = torch.empty(need, dtype=torch.uint8, device=dev)
        _lib.check(lib.evi_retriever_prepare(ctypes.byref(w), buf.data_ptr(), buf.numel(), torch.cuda.current_stream(dev).cuda_stream))
        self._prep_cache = (key, buf)
        return buf

    def _param_fields(self):
        """(EviRetrieverWeights field, parameter) in the struct's order — also the order of the autograd inputs."""
        out = []
        for name, mod in (("entity", self.entity_proj), ("relation", self.relation_proj), ("query", self.query_proj)):
            out += [(f"{name}_w", mod.network[0].weight), (f"{name}_b", mod.network[0].bias)]
        out += [("non_text_emb", self.non_text_entity_emb.weight),
                ("q_gate_w", self.q_gate[0].weight), ("q_gate_b", self.q_gate[0].bias),
                ("q_bias_w", self.q_bias[0].weight), ("q_bias_b", self.q_bias[0].bias),
                ("struct_w", self.struct_proj[0].weight), ("struct_b", self.struct_proj[0].bias),
                ("struct_ln_w", self.struct_proj[1].weight), ("struct_ln_b", self.struct_proj[1].bias),
                ("struct_gate_w", self.struct_gate_net[0].weight), ("struct_gate_b", self.struct_gate_net[0].bias),
                ("state0_w", self.state_net[0].weight), ("state0_b", self.state_net[0].bias),
                ("state_ln_w", self.state_net[1].weight), ("state_ln_b", self.state_net[1].bias),
                ("state4_w", self.state_net[4].weight), ("state4_b", self.state_net[4].bias),
                ("score_w", self.score_head.weight), ("score_b", self.score_head.bias)]
        return out

    def _batch_struct(self, pack) -> "_lib.EviRetrieverBatch":
        b = _lib.EviRetrieverBatch()
        b.num_nodes, b.num_edges, b.num_graphs = pack["N"], pack["E"], pack["B"]
        b.edge_index, b.node_ptr = pack["edge_index"].data_ptr(), pack["node_ptr"].data_ptr()
        b.edge_ptr, b.edge_batch = pack["edge_ptr"].data_ptr(), pack["edge_batch"].data_ptr()
        b.question_emb, b.node_embeddings = pack["question_emb"].data_ptr(), pack["node_embeddings"].data_ptr()
        b.node_embedding_ids = pack["node_embedding_ids"].data_ptr()
        b.edge_embeddings = pack["edge_embeddings"].data_ptr() if pack["edge_embeddings"] is not None else None
        b.relation_rows = pack["relation_rows"].data_ptr() if pack.get("relation_rows") is not None else None
        b.edge_attr, b.num_relations = pack["edge_attr"].data_ptr(), pack["num_relations"]
        b.topic_one_hot, b.topic_stride = pack["topic_one_hot"].data_ptr(), int(pack["topic_one_hot"].size(1))
        b.edge_bias = pack["edge_bias"].data_ptr() if pack["edge_bias"] is not None else None
        b.dropout_p, b.dropout_seed = float(pack.get("dropout_p", 0.0)), int(pack.get("dropout_seed", 0))
        b.matmul_precision = int(pack.get("matmul_precision", 0))
        return b

    def _status_word(self, dev: torch.device) -> torch.Tensor:
        """The sticky device word behind check_deferred(): bit 0 = a relation id outside num_relations, bit 1 = seed / answer
        indices outside the batch's nodes (hide-and-seek mask)."""
        st = getattr(self, "_deferred_status", None)
        if st is None or st.device != dev:
            st = self._deferred_status = torch.zeros(1, dtype=torch.int32, device=dev)
        return st

    def _launch_forward(self, pack, keep_for_backward: bool = False):
        """One evi_retriever_forward call: (logits, logits_fwd, logits_bwd, features-or-None), device tensors.
        keep_for_backward: the per-edge intermediates are written to `pack["saved"]` (evi_retriever_saved_bytes) so that the
        backward replays them instead of recomputing the per-edge forward."""
        lib = _lib.load()
        dev, E, N, B = pack["dev"], pack["E"], pack["N"], pack["B"]
        D, H = self.emb_dim, self.hidden_dim
        logits = torch.empty(E, dtype=torch.float32, device=dev)
        both = self.direction_mode == "bidirectional"
        logits_fwd = torch.empty(E, dtype=torch.float32, device=dev) if self.direction_mode != "backward" else None
        logits_bwd = torch.empty(E, dtype=torch.float32, device=dev) if self.direction_mode != "forward" else None
        features = torch.empty((E, H), dtype=torch.float32, device=dev) if pack["want_features"] else None
        w = self._weights_struct()
        prep = self._prepared_weights(w, dev)
        w.prepared = prep.data_ptr() if prep is not None else None
        b = self._batch_struct(pack)
        o = _lib.EviRetrieverOutput()
        o.logits = logits.data_ptr()
        o.logits_fwd = logits_fwd.data_ptr() if logits_fwd is not None else None
        o.logits_bwd = logits_bwd.data_ptr() if logits_bwd is not None else None
        o.edge_features = features.data_ptr() if features is not None else None
        o.node_struct = None
        # sticky flag for relation ids outside the stated num_relations (no read-back here: check_deferred())
        o.status = self._status_word(dev).data_ptr()
        o.saved, o.saved_bytes = None, 0
        if keep_for_backward and E > 0:
            # per-edge rows + the node-level results (projections, structure features, CSR): the backward replays both
            nbytes = int(lib.evi_retriever_saved_bytes_full(N, E, B, D, H, self.dde.num_rounds, self.dde.num_reverse_rounds,
                                                            pack["num_relations"], _DIRECTION_CODE[self.direction_mode]))
            pack["saved"] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            o.saved, o.saved_bytes = pack["saved"].data_ptr(), nbytes
        need = int(lib.evi_retriever_forward_workspace_bytes(N, E, B, D, H, self.dde.num_rounds,
                                                             self.dde.num_reverse_rounds, pack["num_relations"]))
        ws = ops._workspace(dev, "retriever_forward", need)
        _lib.check(lib.evi_retriever_forward(ctypes.byref(w), ctypes.byref(b), _DIRECTION_CODE[self.direction_mode],
                                             ctypes.byref(o), ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream(dev).cuda_stream))
        if not both:
            pass
        return logits, logits_fwd, logits_bwd, features

    def _launch_backward(self, pack, dlogits: torch.Tensor):
        """evi_retriever_backward: the gradient tensors of every parameter, in `_param_fields()` order."""
        lib = _lib.load()
        dev, E, N, B = pack["dev"], pack["E"], pack["N"], pack["B"]
        D, H = self.emb_dim, self.hidden_dim
        fields = self._param_fields()
        # `grads_in_place` (set by RetrieverTrainer, whose `.grad`s are views of ONE flat buffer it has just cleared): the kernels
        # write every gradient straight into `p.grad` and autograd gets None — no 25 temporaries, no 25 accumulate-adds per step.
        # OVERWRITE semantics (the C backward clears what it writes): only for callers that run one backward per zero_grad.
        in_place = bool(getattr(self, "grads_in_place", False)) and all(
            p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.device == p.device
            and p.grad.shape == p.shape for _, p in fields)
        grads = [p.grad if in_place else torch.empty_like(p, memory_format=torch.contiguous_format) for _, p in fields]
        w = self._weights_struct()
        g = _lib.EviRetrieverWeights()
        g.emb_dim, g.hidden_dim, g.num_topics = w.emb_dim, w.hidden_dim, w.num_topics
        g.dde_rounds, g.dde_reverse_rounds = w.dde_rounds, w.dde_reverse_rounds
        for (name, _), t in zip(fields, grads):
            setattr(g, name, t.data_ptr())
        g.prepared = None
        R = pack["num_relations"]
        perm = ptr = None
        if R > 0:  # edges grouped by relation id (stable), for the ordered segment sums of the relation rows' gradients
            attr = pack["edge_attr"].clamp(0, R - 1)
            perm = torch.argsort(attr, stable=True)
            ptr = ops.ids_to_ptr(attr, R)
        need = int(lib.evi_retriever_backward_workspace_bytes(N, E, B, D, H, self.dde.num_rounds, self.dde.num_reverse_rounds, R))
        ws = ops._workspace(dev, "retriever_backward", need)
        dl = dlogits.detach().to(device=dev, dtype=torch.float32).contiguous().view(-1)
        b = self._batch_struct(pack)
        saved = pack.pop("saved", None)  # written by this batch's forward (keep_forward_intermediates); None: recompute
        _lib.check(lib.evi_retriever_backward(ctypes.byref(w), ctypes.byref(b), _DIRECTION_CODE[self.direction_mode], dl.data_ptr(),
                                              ctypes.byref(g), perm.data_ptr() if perm is not None else None,
                                              ptr.data_ptr() if ptr is not None else None, ws.data_ptr(), ws.numel(),
                                              saved.data_ptr() if saved is not None else None,
                                              saved.numel() if saved is not None else 0,
                                              torch.cuda.current_stream(dev).cuda_stream))
        return [None] * len(grads) if in_place else grads

    def _empty_output(self, head_idx: torch.Tensor, return_features: bool):
        dev = head_idx.device
        empty = torch.empty(0, device=dev, dtype=torch.float32)
        feats = torch.empty((0, self.hidden_dim), device=dev, dtype=torch.float32)
        out = RetrieverOutput(logits=empty, query_ids=head_idx.new_empty(0), relation_ids=None, logits_fwd=empty,
                              logits_bwd=empty, edge_embeddings=feats)
        return out, (feats if return_features else None)

    def _compute_hide_seek_bias(self, batch: Any, *, edge_index: torch.Tensor) -> Optional[torch.Tensor]:
        """The hide-and-seek logit penalty (always in training, in evaluation with hide_seek_cfg.apply_in_eval): each edge is
        "hidden" with probability p_near / p_far by whether it touches a seed or answer node, and a hidden edge gets
        bias_near / bias_far added to both directional logits.  reference: _should_apply_hide_seek /
        _compute_hide_seek_bias, src/models/components/retriever.py:307-367 (the draw is torch.rand on the device, as there;
        it carries no gradient)."""
        if not (self.hide_seek_enabled and (self.training or self.hide_seek_apply_in_eval)):
            return None
        dev = edge_index.device
        edge_is_near = getattr(batch, "edge_is_near", None)
        if edge_is_near is None:
            q, a = getattr(batch, "q_local_indices", None), getattr(batch, "a_local_indices", None)
            if q is None or a is None:
                raise ValueError("Batch missing q_local_indices/a_local_indices required for hide-and-seek.")
            num_nodes = getattr(batch, "num_nodes", None)
            if num_nodes is None:
                raise ValueError("Batch missing num_nodes required for hide-and-seek.")
            # the range check of the seed / answer indices is deferred (bit 1 of the sticky status word: check_deferred())
            near = compute_qa_edge_mask(edge_index, num_nodes=int(num_nodes), q_local_indices=q, a_local_indices=a,
                                        deferred_status=self._status_word(dev))
        else:
            near = torch.as_tensor(edge_is_near).to(device=dev, dtype=torch.bool).view(-1)
            if near.numel() != edge_index.size(1):
                raise ValueError(f"edge_is_near length mismatch: {near.numel()} vs edges {edge_index.size(1)}")
        if near.numel() == 0:
            return None
        if self.hide_seek_p_near <= 0.0 and self.hide_seek_p_far <= 0.0:
            return None
        if self.hide_seek_bias_near == 0.0 and self.hide_seek_bias_far == 0.0:
            return None
        # python scalars, not torch.tensor(x, device=...): that is a synchronous host-to-device copy, i.e. a wait for the stream
        near_f = near.to(torch.float32)
        drop_prob = near_f * self.hide_seek_p_near + (1.0 - near_f) * self.hide_seek_p_far
        drop = torch.rand_like(drop_prob) < drop_prob
        bias = near_f * self.hide_seek_bias_near + (1.0 - near_f) * self.hide_seek_bias_far
        return torch.where(drop, bias, torch.zeros_like(bias)).contiguous()

    def _forward_impl(self, batch: Any, *, return_features: bool):
        param = self.score_head.weight
        dev = param.device
        if dev.type != "cuda":
            raise RuntimeError("evi_rag_amd.Retriever runs on the MI355X only: move the module to a GPU "
                               "(there is no CPU fallback).")
        if param.dtype != torch.float32:
            raise ValueError("evi_rag_amd.Retriever computes in float32 (trainer precision 32-true)")
        edge_index = getattr(batch, "edge_index", None)
        if edge_index is None:
            raise ValueError("Batch missing edge_index required for scoring.")
        edge_index = edge_index.to(device=dev, dtype=torch.long).contiguous()
        head_idx = edge_index[0]
        if head_idx.numel() == 0:
            return self._empty_output(head_idx, return_features)

        question_emb = getattr(batch, "question_emb", None)
        node_embedding_ids = getattr(batch, "node_embedding_ids", None)
        edge_attr = getattr(batch, "edge_attr", None)
        if question_emb is None or node_embedding_ids is None or edge_attr is None:
            raise ValueError("Batch must provide question_emb, node_embedding_ids, and edge_attr.")
        node_embeddings = getattr(batch, "node_embeddings", None)
        # A loader that knows the relation TABLE (GlobalEmbeddingStore.attach) hands it over as `relation_embedding_table` [R, D] and
        # leaves `edge_embeddings` (= table[edge_attr], [E, D]) to be gathered only if somebody asks for it: with relation
        # de-duplication on, the forward wants one row per relation, which the table already is.
        relation_rows = None
        table = getattr(batch, "relation_embedding_table", None) if self.dedupe_relations else None
        if table is not None and getattr(batch, "num_relations", None) is not None:
            t = torch.as_tensor(table)
            R_hint = int(batch.num_relations)
            if t.dim() == 2 and t.size(0) == R_hint and t.size(1) == self.emb_dim and 0 < R_hint <= int(edge_index.size(1)):
                relation_rows = t
        edge_embeddings = None if relation_rows is not None else getattr(batch, "edge_embeddings", None)
        if node_embeddings is None or (edge_embeddings is None and relation_rows is None):
            raise ValueError("Batch must provide node_embeddings and edge_embeddings.")
        topic_one_hot = getattr(batch, "topic_one_hot", None)
        if topic_one_hot is None:
            raise ValueError("topic_one_hot is required for DDE-based structure features.")
        if getattr(batch, "reverse_edge_index", None) is not None:
            raise NotImplementedError("a custom reverse_edge_index is not supported; the reverse rounds use edge_index.flip(0)")
        node_ptr = getattr(batch, "ptr", None)
        if node_ptr is None:
            raise ValueError("Batch missing ptr required for edge_batch computation.")

        f32 = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()  # noqa: E731
        i64 = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.long, non_blocking=True).contiguous()  # noqa: E731
        question_emb, node_embeddings = f32(question_emb), f32(node_embeddings)
        edge_embeddings = f32(edge_embeddings) if edge_embeddings is not None else None
        relation_rows = f32(relation_rows) if relation_rows is not None else None
        node_embedding_ids, edge_attr, node_ptr = i64(node_embedding_ids).view(-1), i64(edge_attr).view(-1), i64(node_ptr).view(-1)
        topic_one_hot = f32(topic_one_hot)
        if topic_one_hot.dim() == 1:
            topic_one_hot = topic_one_hot.unsqueeze(-1)
        if topic_one_hot.dim() != 2:
            raise ValueError(f"topic_one_hot must be 2D (N, C), got shape {tuple(topic_one_hot.shape)}")
        E = int(edge_index.size(1))
        N = int(node_embeddings.size(0))
        B = int(node_ptr.numel() - 1)
        D, H = self.emb_dim, self.hidden_dim
        if B <= 0:
            raise ValueError(f"num_graphs must be positive, got {B}")
        if topic_one_hot.size(0) != N:
            raise ValueError(f"topic_one_hot first dim {topic_one_hot.size(0)} != num_nodes {N}")
        if topic_one_hot.size(-1) < self.num_topics:
            raise ValueError(
                f"topic_one_hot feature dim {topic_one_hot.size(-1)} < num_topics={self.num_topics}; "
                "rebuild g_retrieval caches or update configs/build_retrieval_pipeline.yaml."
            )
        for name, t, rows in (("question_emb", question_emb, B), ("node_embeddings", node_embeddings, N),
                              ("edge_embeddings", edge_embeddings, E)):
            if t is None:
                continue
            if t.dim() != 2 or t.size(0) != rows or t.size(1) != D:
                raise ValueError(f"{name} must have shape [{rows}, {D}], got {tuple(t.shape)}")
        if node_embedding_ids.numel() != N or edge_attr.numel() != E:
            raise ValueError("node_embedding_ids / edge_attr length mismatch with node_embeddings / edge_index")

        # query ids (edge -> graph), validated as the reference validates them (:572-620)
        edge_batch = getattr(batch, "edge_batch", None)
        edge_ptr = getattr(batch, "edge_ptr", None)
        if edge_batch is None or edge_ptr is None:
            total_nodes = int(node_ptr[-1].item())
            if total_nodes <= 0:
                raise ValueError(f"total_nodes must be positive, got {total_nodes}")
            mn, mx = int(edge_index.min().item()), int(edge_index.max().item())
            if mn < 0 or mx >= total_nodes:
                raise ValueError(f"Invalid edge_index range; edge_index out of range: min={mn} max={mx} "
                                 f"total_nodes={total_nodes} num_graphs={B}.")
            edge_batch, edge_ptr = compute_edge_batch(edge_index, node_ptr=node_ptr, num_graphs=B, device=dev)
            try:
                batch.edge_batch = edge_batch  # the reference caches it on the batch too (:612)
                batch.edge_ptr = edge_ptr      # (the loader attaches both: loader.py:98-99)
            except Exception:  # pragma: no cover - read-only batch objects
                pass
        edge_batch, edge_ptr = i64(edge_batch).view(-1), i64(edge_ptr).view(-1)
        if edge_batch.numel() != E:
            raise ValueError(f"edge_batch length mismatch: {edge_batch.numel()} vs edges {E}")

        num_relations = 0
        if self.dedupe_relations:
            hint = getattr(batch, "num_relations", None)
            if hint is not None:
                # stated by whoever gathered edge_embeddings by these ids (the gather range-checks them): no device read
                num_relations = int(hint) if 0 < int(hint) <= E else 0
            else:
                num_relations = int(edge_attr.max().item()) + 1
                if num_relations > E or int(edge_attr.min().item()) < 0:
                    num_relations = 0

        pack = dict(N=N, E=E, B=B, dev=dev, edge_index=edge_index, node_ptr=node_ptr, edge_ptr=edge_ptr, edge_batch=edge_batch,
                    question_emb=question_emb, node_embeddings=node_embeddings, node_embedding_ids=node_embedding_ids,
                    edge_embeddings=edge_embeddings, relation_rows=relation_rows, edge_attr=edge_attr, num_relations=num_relations,
                    topic_one_hot=topic_one_hot,
                    edge_bias=self._compute_hide_seek_bias(batch, edge_index=edge_index),  # None unless apply_in_eval
                    want_features=return_features or self.emit_edge_embeddings)
        if self.matmul_precision not in _MATMUL_PRECISION_CODE:
            raise ValueError(f"matmul_precision must be one of {sorted(_MATMUL_PRECISION_CODE)}, got {self.matmul_precision!r}")
        pack["matmul_precision"] = _MATMUL_PRECISION_CODE[self.matmul_precision]  # the backward of this call uses the same
        drop_p = float(self.state_net[3].p) if self.training else 0.0
        if drop_p > 0.0:
            # nn.Dropout between state_net's GELU and state_net.4 (:179): the mask is a counter-based hash of a per-call seed
            # drawn from torch's CPU generator (torch.manual_seed makes a run reproducible); the backward regenerates it
            if drop_p >= 1.0:
                raise ValueError(f"dropout probability has to be below 1 for training, but got {drop_p}")
            pack["dropout_p"] = drop_p
            pack["dropout_seed"] = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        both = self.direction_mode == "bidirectional"
        params = [p for _, p in self._param_fields()]
        differentiable = self.training if self.differentiable is None else bool(self.differentiable)
        if differentiable and self.matmul_precision == "f16x2" and torch.is_grad_enabled():
            raise ValueError("matmul_precision='f16x2' is an evaluation-time option (the backward runs with 'split' or 'bf16')")
        if differentiable and torch.is_grad_enabled() and any(p.requires_grad for p in params):
            # the differentiable form: gradients with respect to the parameters come from
            # evi_retriever_backward; logits_fwd / logits_bwd / edge_embeddings are returned detached
            logits, logits_fwd, logits_bwd, features = _RetrieverFunction.apply(self, pack, *params)
        else:
            logits, logits_fwd, logits_bwd, features = self._launch_forward(pack)
        if not both:  # single-direction modes: logits IS the directional logit (:267-276)
            logits_fwd = logits if self.direction_mode == "forward" else None
            logits_bwd = logits if self.direction_mode == "backward" else None
        output = RetrieverOutput(logits=logits, query_ids=edge_batch, relation_ids=getattr(batch, "edge_attr", None),
                                 logits_fwd=logits_fwd, logits_bwd=logits_bwd, edge_embeddings=features)
        return output, (features if return_features else None)


class _RetrieverFunction(torch.autograd.Function):
    """Retriever.forward as one autograd node: forward = evi_retriever_forward, backward = evi_retriever_backward.  With
    `module.keep_forward_intermediates` (default) the forward keeps its per-edge operand / product rows
    (≈ (4 D + 4 H) floats per edge) and the backward replays them; otherwise the backward recomputes the forward chunk by
    chunk and nothing but the batch is kept between the two (same gradients either way, bit for bit)."""

    @staticmethod
    def forward(ctx, module, pack, *params):
        logits, logits_fwd, logits_bwd, features = module._launch_forward(
            pack, keep_for_backward=bool(getattr(module, "keep_forward_intermediates", True)))
        ctx.module, ctx.pack = module, pack
        extra = [t for t in (logits_fwd, logits_bwd, features) if t is not None]
        if extra:
            ctx.mark_non_differentiable(*extra)
        return logits, logits_fwd, logits_bwd, features

    @staticmethod
    def backward(ctx, dlogits, *_unused):
        if dlogits is None:
            return (None, None) + tuple(None for _ in ctx.module._param_fields())
        grads = ctx.module._launch_backward(ctx.pack, dlogits)
        return (None, None) + tuple(grads)


def _check_deferred(self) -> None:
    """Raises the IndexError the reference raises at its embedding gather (embedding_store.py:139-150) when a forward
    since the last call met an `edge_attr` outside [0, batch.num_relations) on the relation-dedupe path.  Such edges were
    scored with a clamped relation row (never an out-of-bounds read).  One read-back: call once per epoch."""
    st = getattr(self, "_deferred_status", None)
    code = int(st.item()) if st is not None else 0
    if code != 0:
        st.zero_()
        if code & 2:
            raise ValueError("q/a local indices exceed num_nodes; batch collation is invalid.")
        raise IndexError("edge_attr out of range: a relation id exceeds batch.num_relations")


Retriever.check_deferred = _check_deferred


__all__ = ["Retriever", "RetrieverOutput", "EmbeddingProjector", "DDE", "compute_edge_batch", "compute_qa_edge_mask"]
