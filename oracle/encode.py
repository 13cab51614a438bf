"""Text-encoding tail oracle (E2/E3): masked mean pooling and the id-addressed embedding table
(test infrastructure only).  The transformer forward itself is third-party (`transformers`) and
stays in PyTorch on both sides — PARITY UNPINNED for real-model embeddings; what
is pinned here is pooling, batching, dtype and scatter semantics.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

ENCODER_EPS = 1e-6  # scripts/text_encode_utils.py:10


def masked_mean_pool(hidden: np.ndarray, attention_mask: np.ndarray, *, fp16: bool = False) -> np.ndarray:
    """(hidden * mask).sum(1) / clamp(mask.sum(1), min=1e-6), computed in the pooling dtype
    (f16 when fp16 else f32), returned as f32.
    reference: TextEncoder.encode, scripts/text_encode_utils.py:60-65."""
    dt = np.float16 if fp16 else np.float32
    hid = np.asarray(hidden).astype(dt)
    mask = np.asarray(attention_mask).astype(dt)[..., None]
    if fp16:
        # torch's f16 sum on CPU accumulates in f32 and rounds the result once
        summed = (hid * mask).astype(np.float16).astype(np.float32).sum(axis=1).astype(np.float16)
        denom = np.maximum(mask.astype(np.float32).sum(axis=1).astype(np.float16), np.float16(ENCODER_EPS))
        return (summed / denom).astype(np.float16).astype(np.float32)
    summed = (hid * mask).sum(axis=1, dtype=np.float32)
    denom = np.maximum(mask.sum(axis=1, dtype=np.float32), np.float32(ENCODER_EPS))
    return (summed / denom).astype(np.float32)


def scatter_rows(emb_chunks: Sequence[np.ndarray], id_chunks: Sequence[Sequence[int]], max_embedding_id: int,
                 emb_dim: int) -> np.ndarray:
    """Zero table [(max_id + 1), D]; row emb_id <- embedding, ids outside [0, max_id] skipped, later
    writes win.  reference: encode_to_memmap/_init_memmap/_write_chunk,
    scripts/text_encode_utils.py:70-146."""
    table = np.zeros((max_embedding_id + 1, emb_dim), dtype=np.float32)
    for emb, ids in zip(emb_chunks, id_chunks):
        for row, emb_id in zip(np.asarray(emb, np.float32), ids):
            if not 0 <= int(emb_id) <= max_embedding_id:
                continue
            table[int(emb_id)] = row
    return table


def iter_batches(total: int, batch_size: int, offset: int = 0):
    """reference: _iter_batches, scripts/text_encode_utils.py:115-118."""
    for start in range(offset, total, batch_size):
        yield start, min(start + batch_size, total)
