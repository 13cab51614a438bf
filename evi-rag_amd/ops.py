"""Thin host wrappers over the C-ABI: torch tensors in, torch tensors out.

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic happens in
libevi_hip.so.  Every wrapper requires its tensors to live on a HIP device and raises otherwise —
there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_WORKSPACES: dict = {}


def _require_gpu(*tensors: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "evi_rag_amd ops run on the MI355X only: got a tensor on "
                f"{t.device}. There is no CPU fallback (the CPU oracle lives under oracle/ for tests)."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise ValueError("no tensors given")
    return dev


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() if t.numel() > 0 else None


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _workspace(dev: torch.device, tag: str, nbytes: int) -> torch.Tensor:
    """Grow-only per-(device, tag) scratch buffer (mirrors the reference's grow-only pinned
    buffers, src/data/components/embedding_store.py:101-150, but in HBM)."""
    key = (dev.index, tag)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        _WORKSPACES[key] = buf
    return buf


def release_workspaces() -> None:
    _WORKSPACES.clear()


# ---- C1 -----------------------------------------------------------------------------------------

def row_inv_norm(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """1 / clamp(||x_i||, min=eps).  reference: scripts/build_retrieval_pipeline.py:836."""
    dev = _require_gpu(x)
    if x.dim() != 2:
        raise ValueError(f"x must be 2D [n, D], got shape {tuple(x.shape)}")
    x = _f32c(x, "x")
    n, D = x.shape
    out = torch.empty(n, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    lib = _lib.load()
    _lib.check(lib.evi_row_inv_norm(_ptr(x), n, D, float(eps), _ptr(out), _stream(dev)))
    return out


def normalize_embeddings(embeddings: torch.Tensor, eps: float = 1e-6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x / clamp(||x||, min=eps) row-wise; empty tensors pass through.
    reference: _normalize_embeddings, scripts/build_retrieval_pipeline.py:833-837."""
    if embeddings.numel() == 0:
        return embeddings
    dev = _require_gpu(embeddings, out)
    if embeddings.dim() == 1:
        return normalize_embeddings(embeddings.unsqueeze(0), eps).squeeze(0)
    if embeddings.dim() != 2:
        raise ValueError(f"embeddings must be 1D or 2D, got shape {tuple(embeddings.shape)}")
    x = _f32c(embeddings, "embeddings")
    n, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of the input's shape")
    lib = _lib.load()
    _lib.check(lib.evi_row_normalize(_ptr(x), n, D, float(eps), _ptr(out), _stream(dev)))
    return out


# ---- cosine top-k -------------------------------------------------------------------------------

def cosine_topk_workspace_bytes(Q: int, N: int, D: int, k: int) -> int:
    return int(_lib.load().evi_cosine_topk_workspace_bytes(int(Q), int(N), int(D), int(k)))


def cosine_topk(
    queries: torch.Tensor,
    index: torch.Tensor,
    k: int,
    *,
    row_scale: Optional[torch.Tensor] = None,
    row_id_base: int = 0,
    workspace: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k rows of `index` per query by dot product (cosine when both are L2-normalised, or when
    `index` is raw and row_scale = row_inv_norm(index)).  Returns (scores [Q,k] f32, ids [Q,k] i64),
    ordered (score desc, id asc); slots past min(k, N) hold (-inf, -1)."""
    dev = _require_gpu(queries, index, row_scale, workspace)
    if queries.dim() != 2 or index.dim() != 2:
        raise ValueError("queries and index must be 2D")
    q = _f32c(queries, "queries")
    x = _f32c(index, "index")
    Q, D = q.shape
    N, D2 = x.shape
    if D != D2:
        raise ValueError(f"query dim {D} != index dim {D2}")
    if row_scale is not None:
        row_scale = _f32c(row_scale, "row_scale").view(-1)
        if row_scale.numel() != N:
            raise ValueError(f"row_scale length {row_scale.numel()} != N {N}")
    if Q == 0:
        return (torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int64, device=dev))
    lib = _lib.load()
    need = int(lib.evi_cosine_topk_workspace_bytes(Q, N, D, int(k)))
    if workspace is None:
        workspace = _workspace(dev, "cosine_topk", need)
    out_score = torch.empty((Q, k), dtype=torch.float32, device=dev)
    out_index = torch.empty((Q, k), dtype=torch.int64, device=dev)
    _lib.check(
        lib.evi_cosine_topk(
            _ptr(q), Q, _ptr(x), N, D, _ptr(row_scale), int(k), int(row_id_base),
            _ptr(out_score), _ptr(out_index), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
            _stream(dev),
        )
    )
    return out_score, out_index


def topk_merge(scores: torch.Tensor, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge per-shard top-k lists [P, Q, k] (shards in ascending row-id order) into [Q, k]."""
    dev = _require_gpu(scores, ids)
    if scores.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("scores and ids must both be [P, Q, k]")
    if ids.dtype != torch.int64:
        raise ValueError("ids must be int64")
    s = _f32c(scores, "scores")
    ids = ids.contiguous()
    P, Q, k = s.shape
    out_score = torch.empty((Q, k), dtype=torch.float32, device=dev)
    out_index = torch.empty((Q, k), dtype=torch.int64, device=dev)
    if Q == 0:
        return out_score, out_index
    lib = _lib.load()
    _lib.check(lib.evi_topk_merge(_ptr(s), _ptr(ids), P, Q, k, _ptr(out_score), _ptr(out_index), _stream(dev)))
    return out_score, out_index


# ---- segmented top-k ----------------------------------------------------------------------------

def segment_topk(scores: torch.Tensor, edge_ptr: torch.Tensor, k: int, *, want_scores: bool = True):
    """Per-graph top-k of edge scores.  Returns (local_index [B,k] i32, score [B,k] f32 | None,
    count [B] i32)."""
    dev = _require_gpu(scores, edge_ptr)
    s = _f32c(scores.view(-1), "scores")
    if edge_ptr.dtype != torch.int64:
        raise ValueError("edge_ptr must be int64")
    ptr = edge_ptr.contiguous().view(-1)
    B = ptr.numel() - 1
    if B < 0:
        raise ValueError("edge_ptr must have at least one entry")
    out_index = torch.empty((B, k), dtype=torch.int32, device=dev)
    out_score = torch.empty((B, k), dtype=torch.float32, device=dev) if want_scores else None
    out_count = torch.empty((B,), dtype=torch.int32, device=dev)
    if B == 0:
        return out_index, out_score, out_count
    lib = _lib.load()
    _lib.check(
        lib.evi_segment_topk(_ptr(s), _ptr(ptr), B, int(k), _ptr(out_index), _ptr(out_score), _ptr(out_count), _stream(dev))
    )
    return out_index, out_score, out_count
