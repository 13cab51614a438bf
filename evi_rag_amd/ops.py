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


def row_norms(x: torch.Tensor) -> torch.Tensor:
    """||x_i||_2 per row (f32).  reference: torch.norm(features, p=2, dim=-1), src/metrics/feature_monitor.py:43."""
    dev = _require_gpu(x)
    if x.dim() != 2:
        raise ValueError(f"x must be 2D [n, D], got shape {tuple(x.shape)}")
    x = _f32c(x, "x")
    n, D = x.shape
    out = torch.empty(n, dtype=torch.float32, device=dev)
    if n:
        _lib.check(_lib.load().evi_row_norms(_ptr(x), n, D, _ptr(out), _stream(dev)))
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


def quantize_rows_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(e4m3 bytes [n, D] uint8, scale [n] f32): x ~ e4m3 * scale, scale = max|x| / 448 per row."""
    dev = _require_gpu(x)
    if x.dim() != 2:
        raise ValueError(f"x must be 2D [n, D], got shape {tuple(x.shape)}")
    x = _f32c(x, "x")
    n, D = x.shape
    out = torch.empty((n, D), dtype=torch.uint8, device=dev)
    scale = torch.empty(n, dtype=torch.float32, device=dev)
    if n:
        lib = _lib.load()
        _lib.check(lib.evi_quantize_rows_fp8(_ptr(x), n, D, _ptr(out), _ptr(scale), _stream(dev)))
    return out, scale


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
    out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
    method: str = "auto",
    fp8_mfma: Optional[bool] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k rows of `index` per query by dot product (cosine when both are L2-normalised, or when
    `index` is raw and row_scale = row_inv_norm(index)).  Returns (scores [Q,k] f32, ids [Q,k] i64),
    ordered (score desc, id asc); slots past min(k, N) hold (-inf, -1).
    method — every choice returns the SAME ids and scores, bit for bit (tested):
      "auto" (default): the fastest exact path the inputs allow —
          * Q >= 96 on an f32 / f16 index of unit rows: the GEMM-shaped pass (cosine_topk_gemm; one synchronisation);
          * an f32 index whose f16 shadow is resident (built by `index_shadow_f16(index)`, which attaches it to the
            tensor): the two-stage scan — half the HBM bytes per batch; a batch whose exactness proof fails is re-done
            by the gated f32 scan ON THE DEVICE, nothing is read back (2.36 vs 4.57 ms per batch at 2^23 x 768);
          * else the scan.
      "scan": 32 queries per pass over the index, never synchronises.
      "gemm": force the GEMM-shaped pass (see cosine_topk_gemm).
    fp8_mfma (e4m3 index only; default None = True): feed the index bytes to the native fp8 matrix instruction with the
    query as two e4m3 pieces (evi_cosine_topk_fp8_mfma) — no conversion work on the stream (5.86 vs 5.43 TB/s); scores
    carry 16 significant query bits, ~20x inside the index's own quantisation noise (overlap@500 with the widening
    variant 0.999).  False: widen the index bytes to f16 in registers and keep the f32 query exact."""
    if method not in ("scan", "gemm", "auto"):
        raise ValueError(f"method must be 'scan', 'gemm' or 'auto', got {method!r}")
    if method != "scan" and index.dtype in (torch.float32, torch.float16) and queries.dim() == 2 and index.dim() == 2:
        eligible = index.size(1) % (32 if index.dtype == torch.float16 else 16) == 0 and k + max(256, k // 2) <= 2048 and index.size(0) >= 1 and queries.size(0) >= 1
        if method == "gemm" or (eligible and queries.size(0) >= 96):
            cosine_topk.last_method = "gemm"
            return cosine_topk_gemm(queries, index, k, row_scale=row_scale, row_id_base=row_id_base, out=out)
        shadow = resident_shadow_f16(index) if (method == "auto" and row_scale is None and index.dtype == torch.float32) else None
        if shadow is not None and queries.size(0) >= 1 and index.size(0) >= 1 and k + max(256, k // 2) <= 2048 and index.size(1) % 32 == 0 \
                and queries.dtype == torch.float32 and queries.size(1) == index.size(1):
            cosine_topk.last_method = "two_stage"
            return cosine_topk_two_stage(queries, index, shadow, k, row_id_base=row_id_base, out=out, fallback="device")
    cosine_topk.last_method = "scan"
    dev = _require_gpu(queries, index, row_scale, workspace)
    if queries.dim() != 2 or index.dim() != 2:
        raise ValueError("queries and index must be 2D")
    q = _f32c(queries, "queries")
    if index.dtype == torch.float16:
        x = index.contiguous()  # f16-storage index: evi_cosine_topk_f16
    elif index.dtype == torch.uint8:
        x = index.contiguous()  # e4m3 bytes from quantize_rows_fp8: evi_cosine_topk_fp8 (row_scale required)
        if row_scale is None:
            raise ValueError("an fp8 index needs row_scale (the per-row scale returned by quantize_rows_fp8)")
    else:
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
    if out is not None:
        out_score, out_index = out
        if (out_score.shape != (Q, k) or out_index.shape != (Q, k) or out_score.dtype != torch.float32
                or out_index.dtype != torch.int64 or not out_score.is_contiguous() or not out_index.is_contiguous()):
            raise ValueError("out must be contiguous (float32 [Q, k], int64 [Q, k]) tensors")
    else:
        out_score = torch.empty((Q, k), dtype=torch.float32, device=dev)
        out_index = torch.empty((Q, k), dtype=torch.int64, device=dev)
    if fp8_mfma and x.dtype != torch.uint8:
        raise ValueError("fp8_mfma goes with an e4m3 (uint8) index")
    if fp8_mfma is None:
        fp8_mfma = x.dtype == torch.uint8
    fn = {torch.float16: lib.evi_cosine_topk_f16,
          torch.uint8: lib.evi_cosine_topk_fp8_mfma if fp8_mfma else lib.evi_cosine_topk_fp8}.get(x.dtype, lib.evi_cosine_topk)
    _lib.check(
        fn(
            _ptr(q), Q, _ptr(x), N, D, _ptr(row_scale), int(k), int(row_id_base),
            _ptr(out_score), _ptr(out_index), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
            _stream(dev),
        )
    )
    return out_score, out_index


cosine_topk.last_method = None  # which path the last call took: "scan" | "gemm" | "two_stage" (introspection / tests)


def index_shadow_bf16(index: torch.Tensor) -> torch.Tensor:
    """bf16 copy of an f32 index (+ 50 % memory) for cosine_topk_gemm(shadow=...): the plain-bf16 candidate selection
    then streams half the bytes and converts nothing.  Same results as without it."""
    dev = _require_gpu(index)
    x = _f32c(index, "index")
    if x.dim() != 2:
        raise ValueError("index must be 2D")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
    if x.numel():
        _lib.check(_lib.load().evi_index_shadow_bf16(_ptr(x), x.size(0), x.size(1), _ptr(out), _stream(dev)))
    return out


def _tensor_cache(t: torch.Tensor) -> dict:
    """Per-tensor-OBJECT cache (an attribute of the tensor): it dies with the tensor, so a later tensor that the caching
    allocator places at the same address never inherits a verdict; entries carry the tensor's `_version` so in-place edits
    are seen."""
    c = t.__dict__.get("_evi_cache")
    if c is None:
        c = {}
        t.__dict__["_evi_cache"] = c
    return c


def resident_shadow_f16(index: torch.Tensor) -> Optional[torch.Tensor]:
    """The f16 shadow `index_shadow_f16(index)` attached to this index tensor, if it is still valid (same version, same
    storage), else None."""
    ent = index.__dict__.get("_evi_cache", {}).get("shadow_f16") if isinstance(index, torch.Tensor) else None
    if ent is None:
        return None
    version, ptr, shadow = ent
    if version != index._version or ptr != index.data_ptr() or shadow.shape != index.shape:
        return None
    return shadow


def rows_are_unit_norm(index: torch.Tensor, row_scale: Optional[torch.Tensor] = None) -> bool:
    """True when every (scaled) row of `index` has norm <= 1 + 1e-4 (1 + 1e-3 for an f16-stored index) — what the exactness proofs of the GEMM-shaped and
    two-stage paths assume (their error bounds are kEps * |q| * |x| with |x| <= 1).  One pass over the index and one
    read-back, cached ON the index tensor per (version, row_scale object + version): in-place edits are seen, and the
    verdict cannot outlive the tensor (a new tensor at a recycled address starts unverified)."""
    import weakref

    cache = _tensor_cache(index)
    ent = cache.get("unit_rows")
    hit = None
    if ent is not None:
        version, ptr, rs_ref, rs_version, verdict = ent
        same_rs = (rs_ref is None and row_scale is None) or (rs_ref is not None and row_scale is not None and rs_ref() is row_scale
                                                             and rs_version == row_scale._version)
        if version == index._version and ptr == index.data_ptr() and same_rs:
            hit = verdict
    if hit is None:
        if index.numel() == 0:
            hit = True
        else:
            x = index if index.dtype == torch.float32 else None
            if x is not None:
                norms = row_norms(x)
            else:  # f16 storage: chunked, in f32
                norms = torch.cat([index[i: i + (1 << 20)].float().norm(dim=1) for i in range(0, index.size(0), 1 << 20)])
            if row_scale is not None:
                norms = norms * row_scale.view(-1).abs()
            # f16 storage: rounding a unit row element by element moves its norm by up to 2^-11 (5e-4)
            hit = bool((norms.max() <= 1.0 + (1e-4 if index.dtype == torch.float32 else 1e-3)).item())
        cache["unit_rows"] = (index._version, index.data_ptr(), None if row_scale is None else weakref.ref(row_scale),
                              None if row_scale is None else row_scale._version, hit)
    return hit


def cosine_topk_gemm(queries: torch.Tensor, index: torch.Tensor, k: int, *, row_scale: Optional[torch.Tensor] = None,
                     row_id_base: int = 0, fallback: bool = True,
                     out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                     products: Optional[int] = None, shadow: Optional[torch.Tensor] = None,
                     check_norms: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """cosine_topk for many queries at once (see evi_cosine_topk_gemm): one split-bf16 GEMM pass over the index
    selects candidates, the scan's arithmetic re-scores them; the result equals cosine_topk bit for bit.  Reads the
    proof flag back (one synchronisation); when the proof fails (heavy score ties, adversarial row order) the scan
    runs instead (fallback=True) or RuntimeError is raised.  products: 3 (split-bf16 selection), 1 (plain bf16 selection:
    three times fewer MFMAs, proof fails earlier) or None = try 1 when k <= 1024, then 3, then the scan."""
    if products not in (None, 1, 3):
        raise ValueError(f"products must be None, 1 or 3, got {products}")
    if shadow is not None:
        if index.dtype != torch.float32 or shadow.dtype != torch.bfloat16 or shadow.shape != index.shape or not shadow.is_contiguous():
            raise ValueError("shadow must be the contiguous bfloat16 copy of an f32 index (index_shadow_bf16)")
    dev = _require_gpu(queries, index, row_scale)
    if queries.dim() != 2 or index.dim() != 2:
        raise ValueError("queries and index must be 2D")
    q = _f32c(queries, "queries")
    x = index.contiguous() if index.dtype == torch.float16 else _f32c(index, "index")  # f16 storage: evi_cosine_topk_gemm_f16
    Q, D = q.shape
    N = x.shape[0]
    if x.shape[1] != D:
        raise ValueError(f"query dim {D} != index dim {x.shape[1]}")
    if row_scale is not None:
        row_scale = _f32c(row_scale, "row_scale").view(-1)
        if row_scale.numel() != N:
            raise ValueError(f"row_scale length {row_scale.numel()} != N {N}")
    if Q == 0 or N == 0:
        return cosine_topk(q, x, k, row_scale=row_scale, row_id_base=row_id_base, out=out, method="scan")
    # the proof's error bound is kApproxEps * |q| * |row|, with |row| <= 1 built in: an index of longer rows would pass
    # the gap test with true top-k rows discarded.  Verified once per index (cached); such an index goes to the scan.
    if check_norms and not rows_are_unit_norm(x, row_scale):
        if not fallback:
            raise ValueError("cosine_topk_gemm needs rows of norm <= 1 (normalize_embeddings, or row_scale = row_inv_norm): "
                             "its exactness proof assumes them")
        cosine_topk_gemm.last_products = 0
        return cosine_topk(q, x, k, row_scale=row_scale, row_id_base=row_id_base, out=out, method="scan")
    lib = _lib.load()
    ws = _workspace(dev, "cosine_topk_gemm", int(lib.evi_cosine_topk_gemm_workspace_bytes(Q, N, D, int(k))))
    if out is not None:
        out_score, out_index = out
        if (out_score.shape != (Q, k) or out_index.shape != (Q, k) or out_score.dtype != torch.float32
                or out_index.dtype != torch.int64 or not out_score.is_contiguous() or not out_index.is_contiguous()):
            raise ValueError("out must be contiguous (float32 [Q, k], int64 [Q, k]) tensors")
    else:
        out_score = torch.empty((Q, k), dtype=torch.float32, device=dev)
        out_index = torch.empty((Q, k), dtype=torch.int64, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    fn = lib.evi_cosine_topk_gemm_f16 if x.dtype == torch.float16 else lib.evi_cosine_topk_gemm
    # The single-product (plain bf16) selection keeps max(1024, k) reserve rows per query and needs the k-th and the last kept
    # score 2 x 4.2e-3 |q| apart: the rows inside that band grow with N (about 390 of the 1024 at 2^23 rows of an exchangeable
    # index, i.e. more than the reserve beyond ~2 x 10^7 rows) — there the attempt only costs a wasted pass (measured at 10^8
    # rows: 319 ms with the failed attempt in front, against the three-product pass alone), so it is skipped
    plan = [products] if products is not None else ([1, 3] if (k <= 1024 and N <= (1 << 24)) else [3])
    st = -1
    for prod in plan:
        if x.dtype == torch.float16:
            args = (int(prod),)
        else:
            args = (int(prod), _ptr(shadow) if (shadow is not None and prod == 1) else None)
        _lib.check(fn(_ptr(q), Q, _ptr(x), N, D, _ptr(row_scale), int(k), int(row_id_base), *args, _ptr(out_score),
                      _ptr(out_index), status.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
        st = int(status.item())
        if st == 0:
            cosine_topk_gemm.last_products = prod  # which selection precision proved the result (introspection / bench)
            break
    if st != 0:
        cosine_topk_gemm.last_products = 0  # the scan produced the result
        if not fallback:
            raise RuntimeError(f"evi_cosine_topk_gemm could not prove exactness (status {st}): run cosine_topk")
        return cosine_topk(q, x, k, row_scale=row_scale, row_id_base=row_id_base, out=out, method="scan")
    return out_score, out_index


cosine_topk_gemm.last_products = None


def index_shadow_f16(index: torch.Tensor, *, check_norms: bool = True, attach: bool = True) -> torch.Tensor:
    """f16 copy (round to nearest) of an f32 index of unit rows (+ 50 % memory) for cosine_topk_two_stage.
    The exactness proof of the two-stage scan bounds the f16 rounding of a row by 2^-11 of its norm and assumes
    norm <= 1 (normalize_embeddings output): check_norms verifies that once here (one pass + one read-back).
    attach (default): the shadow is remembered ON the index tensor, and `cosine_topk(queries, index, k)` (method "auto")
    then takes the two-stage scan by itself for as long as the index is not edited in place."""
    dev = _require_gpu(index)
    x = _f32c(index, "index")
    if x.dim() != 2:
        raise ValueError("index must be 2D")
    if check_norms and x.numel():
        worst = float(row_norms(x).max().item())
        if not worst <= 1.0 + 1e-4:
            raise ValueError(f"index_shadow_f16 needs L2-normalised rows (largest row norm {worst:.6g}): "
                             "run normalize_embeddings first")
    out = torch.empty(x.shape, dtype=torch.float16, device=dev)
    if x.numel():
        _lib.check(_lib.load().evi_index_shadow_f16(_ptr(x), x.size(0), x.size(1), _ptr(out), _stream(dev)))
    if attach and x is index and check_norms:
        _tensor_cache(index)["shadow_f16"] = (index._version, index.data_ptr(), out)
    return out


def drop_shadow_f16(index: torch.Tensor) -> None:
    """Forget (and so free) the shadow attached to `index`."""
    index.__dict__.get("_evi_cache", {}).pop("shadow_f16", None)


def cosine_topk_two_stage(queries: torch.Tensor, index: torch.Tensor, shadow: torch.Tensor, k: int, *, row_id_base: int = 0,
                          fallback="device", out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                          status: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, check_norms: bool = True):
    """cosine_topk over an L2-normalised f32 index at half the HBM bytes (see evi_cosine_topk_two_stage): the f16
    `shadow` (index_shadow_f16(index)) is scanned to select k + max(256, k/2) candidates per query, which are re-scored
    from the f32 rows with the scan's arithmetic; ids and scores equal cosine_topk(queries, index, k) bit for bit.
    What happens when the per-batch exactness proof fails (heavy ties, clustered rows, NaN):
      fallback="device" (default; True means the same): the f32 scan of the batch is enqueued behind the two-stage scan,
        gated on the proof flag ON THE DEVICE — the outputs are always cosine_topk's, nothing is read back, ranks of a
        sharded index need no agreement.  `status` (optional int32 [1] device tensor, zeroed by the caller, sticky) only
        records that a fallback ran.
      fallback="host": the flag is read back (one synchronisation) and the scan is run from the host if needed.
      fallback=False: with status=None the flag is read back and RuntimeError raised on failure; with a `status` tensor
        nothing is read back and the CALLER must check it before using the outputs (the raw C-ABI contract)."""
    if fallback is True:
        fallback = "device"
    if fallback not in ("device", "host", False):
        raise ValueError(f"fallback must be 'device', 'host' or False, got {fallback!r}")
    if fallback == "host" and status is not None:
        raise ValueError("fallback='host' reads the flag back itself: do not pass status")
    dev = _require_gpu(queries, index, shadow, status, workspace)
    if queries.dim() != 2 or index.dim() != 2:
        raise ValueError("queries and index must be 2D")
    if index.dtype != torch.float32 or shadow.dtype != torch.float16 or shadow.shape != index.shape or not shadow.is_contiguous():
        raise ValueError("shadow must be the contiguous float16 copy of an f32 index (index_shadow_f16)")
    q = _f32c(queries, "queries")
    x = _f32c(index, "index")
    Q, D = q.shape
    N = x.shape[0]
    if x.shape[1] != D:
        raise ValueError(f"query dim {D} != index dim {x.shape[1]}")
    # the exactness proof bounds the f16 rounding of a row by 2^-11 of its norm and assumes norm <= 1, like cosine_topk_gemm's:
    # the same guard (one pass over the index + one read-back the FIRST time this tensor is seen, cached on the tensor)
    if check_norms and N and not rows_are_unit_norm(x):
        raise ValueError("cosine_topk_two_stage needs an index of rows with norm <= 1 (normalize_embeddings): its exactness proof "
                         "assumes them; use cosine_topk for a raw index")
    if status is not None and (status.dtype != torch.int32 or status.numel() != 1):
        raise ValueError("status must be an int32 tensor with one element")
    if Q == 0 or N == 0:
        return cosine_topk(q, x, k, row_id_base=row_id_base, out=out, method="scan")
    lib = _lib.load()
    need = int(lib.evi_cosine_topk_two_stage_workspace_bytes(Q, N, D, int(k)))
    if need == 0:
        raise ValueError(f"k + max(256, k // 2) must not exceed 2048, got k = {k}")
    ws = workspace if workspace is not None else _workspace(dev, "cosine_topk_two_stage", need)
    if out is not None:
        out_score, out_index = out
        if (out_score.shape != (Q, k) or out_index.shape != (Q, k) or out_score.dtype != torch.float32
                or out_index.dtype != torch.int64 or not out_score.is_contiguous() or not out_index.is_contiguous()):
            raise ValueError("out must be contiguous (float32 [Q, k], int64 [Q, k]) tensors")
    else:
        out_score = torch.empty((Q, k), dtype=torch.float32, device=dev)
        out_index = torch.empty((Q, k), dtype=torch.int64, device=dev)
    on_device = fallback == "device"
    flag = status
    if flag is None and not on_device:
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.evi_cosine_topk_two_stage(_ptr(q), Q, _ptr(x), _ptr(shadow), N, D, int(k), int(row_id_base), _ptr(out_score),
                                             _ptr(out_index), _ptr(flag), 1 if on_device else 0, ws.data_ptr(),
                                             ws.numel() * ws.element_size(), _stream(dev)))
    if on_device or status is not None:
        return out_score, out_index
    st = int(flag.item())
    cosine_topk_two_stage.last_status = st
    if st != 0:
        if fallback is False:
            raise RuntimeError(f"evi_cosine_topk_two_stage could not prove exactness (status {st}): run cosine_topk")
        return cosine_topk(q, x, k, row_id_base=row_id_base, out=out, method="scan")
    return out_score, out_index


cosine_topk_two_stage.last_status = None


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


def topk_packed_views(packed: torch.Tensor, Q: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(scores [Q, k] f32, ids [Q, k] i64) views into one rank's packed exchange record."""
    n = Q * k
    ids_off = (n * 4 + 7) // 8 * 8
    rec = packed.view(torch.uint8).view(-1)
    return rec[: n * 4].view(torch.float32).view(Q, k), rec[ids_off: ids_off + n * 8].view(torch.int64).view(Q, k)


def topk_merge_packed(packed: torch.Tensor, P: int, Q: int, k: int,
                      out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge P packed per-shard records (the output of ONE all-gather) into the global top-k."""
    dev = _require_gpu(packed)
    lib = _lib.load()
    rec = int(lib.evi_topk_packed_bytes(Q, k))
    if packed.numel() * packed.element_size() < P * rec:
        raise ValueError(f"packed buffer holds {packed.numel() * packed.element_size()} B, need {P * rec} B")
    if out is not None:
        out_score, out_index = out
        if out_score.shape != (Q, k) or out_index.shape != (Q, k) or out_score.dtype != torch.float32 or \
                out_index.dtype != torch.int64 or not out_score.is_contiguous() or not out_index.is_contiguous():
            raise ValueError("out must be contiguous (f32 [Q, k], i64 [Q, k]) tensors")
    else:
        out_score = torch.empty((Q, k), dtype=torch.float32, device=dev)
        out_index = torch.empty((Q, k), dtype=torch.int64, device=dev)
    _lib.check(lib.evi_topk_merge_packed(packed.data_ptr(), P, Q, k, _ptr(out_score), _ptr(out_index), _stream(dev)))
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


# ---- graph structure -----------------------------------------------------------------------------

def _i64c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.int64:
        raise ValueError(f"{name} must be int64, got {t.dtype}")
    return t.contiguous()


def edge_batch(edge_index: torch.Tensor, node_ptr: torch.Tensor):
    """(edge_batch [E] i64, edge_ptr [B+1] i64, status int) — see evi_edge_batch.  Reading `status`
    synchronises the stream, as the reference's `.item()` checks do (src/utils/graph_utils.py:65-99)."""
    dev = _require_gpu(edge_index, node_ptr)
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    ei = _i64c(edge_index, "edge_index")
    ptr = _i64c(node_ptr.view(-1), "node_ptr")
    E, B = ei.size(1), ptr.numel() - 1
    eb = torch.empty(E, dtype=torch.int64, device=dev)
    eptr = torch.empty(B + 1, dtype=torch.int64, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.evi_edge_batch(_ptr(ei), E, _ptr(ptr), B, _ptr(eb), _ptr(eptr), status.data_ptr(), _stream(dev)))
    return eb, eptr, status


def qa_edge_mask(edge_index: torch.Tensor, num_nodes: int, q_local_indices: torch.Tensor,
                 a_local_indices: torch.Tensor, *, deferred_status: Optional[torch.Tensor] = None, deferred_bit: int = 2) -> torch.Tensor:
    """deferred_status (device int32 [1]): instead of reading the kernel's range-check flag back (one host-device
    synchronisation per call), OR `deferred_bit` into this sticky word; whoever owns the word checks it once per epoch."""
    dev = _require_gpu(edge_index)
    ei = _i64c(edge_index, "edge_index")
    E = ei.size(1)
    q = _i64c(q_local_indices.to(dev).view(-1), "q_local_indices")
    a = _i64c(a_local_indices.to(dev).view(-1), "a_local_indices")
    out = torch.empty(E, dtype=torch.uint8, device=dev)
    ws = torch.empty(int(num_nodes) + 4, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.evi_qa_edge_mask(_ptr(ei), E, int(num_nodes), _ptr(q), q.numel(), _ptr(a), a.numel(), ws.data_ptr(),
                                    _ptr(out), status.data_ptr(), _stream(dev)))
    if deferred_status is not None:
        deferred_status.bitwise_or_((status != 0).to(torch.int32) * int(deferred_bit))
    elif int(status.item()) != 0:
        raise ValueError("q/a local indices exceed num_nodes; batch collation is invalid.")
    return out.bool()


class GraphCSR:
    """In-/out-edge CSR of a batch (device int32 arrays)."""

    def __init__(self, in_ptr, in_nbr, in_eid, out_ptr, out_nbr, out_eid):
        self.in_ptr, self.in_nbr, self.in_eid = in_ptr, in_nbr, in_eid
        self.out_ptr, self.out_nbr, self.out_eid = out_ptr, out_nbr, out_eid


def graph_csr(edge_index: torch.Tensor, node_ptr: torch.Tensor, edge_ptr: torch.Tensor, *, num_nodes: Optional[int] = None,
              out: Optional[GraphCSR] = None, workspace: Optional[torch.Tensor] = None) -> GraphCSR:
    """In-/out-edge CSR of every graph of the batch (evi_graph_csr).  num_nodes: the batch's node count when the caller
    knows it (saves the read-back of node_ptr[-1]); out / workspace: caller-owned buffers (hipGraph capture, steady loops)."""
    dev = _require_gpu(edge_index, node_ptr, edge_ptr)
    ei = _i64c(edge_index, "edge_index")
    ptr = _i64c(node_ptr.view(-1), "node_ptr")
    eptr = _i64c(edge_ptr.view(-1), "edge_ptr")
    E, B = ei.size(1), ptr.numel() - 1
    N = int(num_nodes) if num_nodes is not None else int(ptr[-1].item())
    mk = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)  # noqa: E731
    csr = out if out is not None else GraphCSR(mk(N + 1), mk(E), mk(E), mk(N + 1), mk(E), mk(E))
    if out is not None and (csr.in_ptr.numel() < N + 1 or csr.in_nbr.numel() < max(E, 1) or csr.out_ptr.numel() < N + 1):
        raise ValueError("out: CSR buffers too small for this batch")
    lib = _lib.load()
    need = int(lib.evi_graph_csr_workspace_bytes(N))
    if workspace is not None and workspace.numel() * workspace.element_size() < need:
        raise ValueError(f"workspace holds {workspace.numel() * workspace.element_size()} B, need {need} B")
    ws = workspace if workspace is not None else _workspace(dev, "graph_csr", need)
    _lib.check(lib.evi_graph_csr(_ptr(ei), E, _ptr(ptr), _ptr(eptr), B, N, csr.in_ptr.data_ptr(), csr.in_nbr.data_ptr(),
                                 csr.in_eid.data_ptr(), csr.out_ptr.data_ptr(), csr.out_nbr.data_ptr(),
                                 csr.out_eid.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(), _stream(dev)))
    return csr


def dde_node_struct(topic_one_hot: torch.Tensor, node_ptr: torch.Tensor, csr: GraphCSR, num_rounds: int,
                    num_reverse_rounds: int, num_topics: int = 2) -> torch.Tensor:
    """DDE structure features ns [N, num_topics * (1 + rounds + reverse rounds)] (src/models/components/graph.py:41-74,
    retriever.py:519-553): node-parallel launches per round pair for small batches, one workgroup per graph with the graph's
    feature block in LDS from 112 graphs per batch on (evi_dde_node_struct_graphs).  Same bits either way."""
    dev = _require_gpu(topic_one_hot, node_ptr)
    t = _f32c(topic_one_hot, "topic_one_hot")
    if t.dim() == 1:
        t = t.unsqueeze(-1)
    ptr = _i64c(node_ptr.view(-1), "node_ptr")
    N, B = t.size(0), ptr.numel() - 1
    S = 1 + int(num_rounds) + int(num_reverse_rounds)
    ns = torch.empty((N, num_topics * S), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.evi_dde_node_struct_graphs(_ptr(t), t.size(1), int(num_topics), N, _ptr(ptr), B, csr.in_ptr.data_ptr(),
                                              csr.in_nbr.data_ptr(), csr.out_ptr.data_ptr(), csr.out_nbr.data_ptr(),
                                              int(num_rounds), int(num_reverse_rounds), _ptr(ns), _stream(dev)))
    return ns


# ---- segmented de-duplication / re-indexing (G5, f2) -------------------------------------------------

def first_occurrence(keys: torch.Tensor, seg_ptr: torch.Tensor, drop: Optional[torch.Tensor] = None) -> torch.Tensor:
    """first[p] = segment-local position of the first entry with p's key (keys [T] or [T, W<=3] i64);
    -1 where drop[p].  See evi_first_occurrence."""
    dev = _require_gpu(keys, seg_ptr)
    k = _i64c(keys if keys.dim() == 2 else keys.view(-1, 1), "keys")
    ptr = _i64c(seg_ptr.view(-1), "seg_ptr")
    T, W, S = k.size(0), k.size(1), ptr.numel() - 1
    first = torch.empty(T, dtype=torch.int32, device=dev)
    if T == 0 or S <= 0:
        return first
    d = None
    if drop is not None:
        d = drop.to(device=dev, dtype=torch.uint8).contiguous().view(-1)
        if d.numel() != T:
            raise ValueError(f"drop must have one entry per key ({T}), got {d.numel()}")
    lib = _lib.load()
    ws = _workspace(dev, "first_occurrence", int(lib.evi_first_occurrence_workspace_bytes(T, S)))
    _lib.check(lib.evi_first_occurrence(_ptr(k), W, T, _ptr(ptr), S, _ptr(d), _ptr(first), ws.data_ptr(), ws.numel(),
                                        _stream(dev)))
    return first


def first_seen_rank(first: torch.Tensor, seg_ptr: torch.Tensor, limit: Optional[torch.Tensor] = None):
    """(rank [T] i32, count [S] i32, uniq_pos [T] i32) from first_occurrence's output: ids in first-seen
    order, distinct keys per segment, and the position of each distinct key.  See evi_first_seen_rank."""
    dev = _require_gpu(first, seg_ptr)
    ptr = _i64c(seg_ptr.view(-1), "seg_ptr")
    T, S = first.numel(), ptr.numel() - 1
    rank = torch.empty(T, dtype=torch.int32, device=dev)
    count = torch.zeros(max(S, 0), dtype=torch.int32, device=dev)
    uniq = torch.empty(T, dtype=torch.int32, device=dev)
    if S <= 0:
        return rank, count, uniq
    lim = None if limit is None else _i64c(limit.to(dev).view(-1), "limit")
    lib = _lib.load()
    _lib.check(lib.evi_first_seen_rank(_ptr(first.contiguous()), T, _ptr(ptr), S, _ptr(lim), _ptr(rank), _ptr(count),
                                       _ptr(uniq), _stream(dev)))
    return rank, count, uniq


def segment_sort_rank(keys: torch.Tensor, seg_ptr: torch.Tensor, seg_len: Optional[torch.Tensor] = None):
    """(rank [T] i32, sorted [T] i64): stable ascending position of the first seg_len[s] keys of each
    segment and the sorted keys (entries beyond seg_len are left untouched)."""
    dev = _require_gpu(keys, seg_ptr)
    k = _i64c(keys.view(-1), "keys")
    ptr = _i64c(seg_ptr.view(-1), "seg_ptr")
    T, S = k.numel(), ptr.numel() - 1
    rank = torch.empty(T, dtype=torch.int32, device=dev)
    srt = torch.empty(T, dtype=torch.int64, device=dev)
    if T == 0 or S <= 0:
        return rank, srt
    sl = None if seg_len is None else seg_len.to(device=dev, dtype=torch.int32).contiguous()
    lib = _lib.load()
    _lib.check(lib.evi_segment_sort_rank(_ptr(k), T, _ptr(ptr), _ptr(sl), S, _ptr(rank), _ptr(srt), _stream(dev)))
    return rank, srt


def group_max(values: torch.Tensor, group: torch.Tensor, num_groups: int) -> torch.Tensor:
    """out[g] = max of values[i] with group[i] == g (group < 0 skipped); -inf for empty groups."""
    dev = _require_gpu(values, group)
    v = _f32c(values.view(-1), "values")
    g = group.to(device=dev, dtype=torch.int32).contiguous().view(-1)
    if g.numel() != v.numel():
        raise ValueError("values and group must have the same length")
    out = torch.full((int(num_groups),), float("-inf"), dtype=torch.float32, device=dev)
    if v.numel():
        lib = _lib.load()
        _lib.check(lib.evi_group_max_f32(_ptr(v), _ptr(g), v.numel(), _ptr(out), _stream(dev)))
    return out


# ---- dense building block --------------------------------------------------------------------------

_ACT = {None: 0, "none": 0, "tanh": 1, "sigmoid": 2}


def ids_to_ptr(ids: torch.Tensor, num_segments: int) -> torch.Tensor:
    """[num_segments + 1] int64 offsets of the segments of `ids` (values in [0, num_segments); any order): what
    `cat([0], bincount(ids, minlength=B).cumsum(0))` gives — without torch.bincount's device-to-host read of the maximum."""
    ids = ids.view(-1).to(torch.long)
    ptr = torch.zeros(int(num_segments) + 1, dtype=torch.long, device=ids.device)
    if ids.numel():
        ptr[1:].scatter_add_(0, ids, torch.ones_like(ids))
        torch.cumsum(ptr, 0, out=ptr)
    return ptr


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, *, accumulate: bool = False) -> torch.Tensor:
    """a.T @ b for a [K, M], b [K, N] f32 (row-major, K long): the weight gradient of a Linear layer over K rows
    (`grad_out.t() @ input`).  Split-bf16 arithmetic (~1e-5 relative), split-K with an ordered reduction: two calls give
    the same bits.  accumulate: added to `out`."""
    dev = _require_gpu(a, b)
    a2, b2 = _f32c(a, "a"), _f32c(b, "b")
    if a2.dim() != 2 or b2.dim() != 2 or a2.size(0) != b2.size(0):
        raise ValueError(f"gemm_tn: a [K, M] and b [K, N] must share K, got {tuple(a2.shape)} and {tuple(b2.shape)}")
    K, M = a2.shape
    N = b2.size(1)
    if out is None:
        if accumulate:
            raise ValueError("gemm_tn: accumulate needs an output tensor")
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    elif out.shape != (M, N) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("gemm_tn: out must be a contiguous f32 [M, N] tensor")
    lib = _lib.load()
    ws = _workspace(dev, "gemm_tn", int(lib.evi_gemm_tn_bf16x3_workspace_bytes(M, N)))
    _lib.check(lib.evi_gemm_tn_bf16x3(_ptr(a2), M, M, _ptr(b2), N, N, K, _ptr(out), 1 if accumulate else 0, ws.data_ptr(),
                                      ws.numel(), _stream(dev)))
    return out


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: Optional[str] = None, *,
               mode: str = "f32") -> torch.Tensor:
    """act(x @ weight.T + bias).  mode "f32": exact f32-input MFMA GEMM; "bf16x3": split-bf16 GEMM
    (three bf16 MFMAs per product, ~1e-5 relative, ~5x faster)."""
    dev = _require_gpu(x, weight, bias)
    x2 = _f32c(x.reshape(-1, x.shape[-1]), "x")
    w = _f32c(weight, "weight")
    M, K = x2.shape
    N = w.size(0)
    if w.size(1) != K:
        raise ValueError(f"weight in_features {w.size(1)} != input dim {K}")
    if K % 4 != 0:  # zero-pad K to a 16-byte multiple
        pad = 4 - K % 4
        x2 = torch.nn.functional.pad(x2, (0, pad))
        w = torch.nn.functional.pad(w, (0, pad))
        K += pad
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    lib = _lib.load()
    b = _ptr(_f32c(bias, "bias")) if bias is not None else None
    if mode == "f32":
        _lib.check(lib.evi_gemm_nt_f32(_ptr(x2), M, K, K, _ptr(w), N, K, b, _ACT[act], _ptr(out), N, _stream(dev)))
    elif mode == "bf16x3":
        ws = _workspace(dev, "gemm_bf16x3", int(lib.evi_gemm_nt_bf16x3_workspace_bytes(N, K)))
        _lib.check(lib.evi_gemm_nt_bf16x3(_ptr(x2), M, K, K, _ptr(w), N, K, b, _ACT[act], _ptr(out), N, ws.data_ptr(),
                                          ws.numel(), _stream(dev)))
    else:
        raise ValueError(f"mode must be 'f32' or 'bf16x3', got {mode!r}")
    return out.reshape(*x.shape[:-1], N)
