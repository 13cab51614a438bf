"""Host-side mirror of scripts/text_encode_utils.py (TextEncoder, encode_to_memmap).

The transformer forward stays in PyTorch-ROCm (third-party `transformers`, exactly as the
reference runs it); the tail — masked mean pooling and the id-addressed scatter into the embedding
table — runs in libevi_hip.so (`evi_masked_mean_pool`, `evi_scatter_rows`), and the table is built
in HBM instead of a host memmap with a Python row loop (scripts/text_encode_utils.py:137-146).
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops

ENCODER_EPS = 1e-6  # scripts/text_encode_utils.py:10
_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def masked_mean_pool(hidden: torch.Tensor, attention_mask: torch.Tensor, *, fp16: bool = False,
                     eps: float = ENCODER_EPS) -> torch.Tensor:
    """[b, L, D] x [b, L] -> [b, D] f32 on the device (reference arithmetic: text_encode_utils.py:60-64)."""
    dev = ops._require_gpu(hidden, attention_mask)
    if hidden.dim() != 3 or attention_mask.dim() != 2 or hidden.shape[:2] != attention_mask.shape:
        raise ValueError(f"hidden {tuple(hidden.shape)} / attention_mask {tuple(attention_mask.shape)} mismatch")
    if hidden.dtype not in _DTYPE_CODE:
        raise ValueError(f"unsupported hidden dtype {hidden.dtype}")
    hid = hidden.contiguous()
    mask = attention_mask.to(torch.int64).contiguous()
    b, L, D = hid.shape
    out = torch.empty((b, D), dtype=torch.float32, device=dev)
    if b == 0 or D == 0:
        return out
    lib = _lib.load()
    _lib.check(lib.evi_masked_mean_pool(ops._ptr(hid), _DTYPE_CODE[hid.dtype], ops._ptr(mask), b, L, D, int(bool(fp16)),
                                        float(eps), ops._ptr(out), ops._stream(dev)))
    return out


def scatter_rows(table: torch.Tensor, rows: torch.Tensor, ids: torch.Tensor) -> None:
    """table[ids[i]] = rows[i] for ids inside [0, table.size(0)); later rows win on repeats."""
    dev = ops._require_gpu(table, rows, ids)
    if table.dtype != torch.float32 or not table.is_contiguous():
        raise ValueError("table must be a contiguous float32 tensor")
    rows = ops._f32c(rows, "rows")
    ids = ids.to(torch.int64).contiguous().view(-1)
    if rows.dim() != 2 or rows.size(0) != ids.numel() or rows.size(1) != table.size(1):
        raise ValueError("rows must be [len(ids), table.size(1)]")
    max_id = table.size(0) - 1
    if max_id < 0 or ids.numel() == 0 or table.size(1) == 0:
        return
    lib = _lib.load()
    ws = ops._workspace(dev, "scatter_rows", int(lib.evi_scatter_rows_workspace_bytes(max_id)))
    _lib.check(lib.evi_scatter_rows(ops._ptr(rows), ops._ptr(ids), ids.numel(), table.size(1), ops._ptr(table), max_id,
                                    ws.data_ptr(), ws.numel(), ops._stream(dev)))


def _iter_batches(total: int, batch_size: int, offset: int = 0) -> Iterable[Tuple[int, int]]:
    for start in range(offset, total, batch_size):
        yield start, min(start + batch_size, total)


class TextEncoder:
    """Same constructor and `encode` contract as the reference wrapper
    (scripts/text_encode_utils.py:13-67): returns a CPU float32 [n, D] tensor."""

    def __init__(self, model_name: str, device: str, fp16: bool, progress: bool) -> None:
        try:
            from transformers import AutoModel, AutoTokenizer
        except ImportError as exc:
            raise SystemExit("transformers is required for text encoding. pip install transformers.") from exc
        self.tokenizer = AutoTokenizer.from_pretrained(model_name, trust_remote_code=True)
        self.model = AutoModel.from_pretrained(model_name, trust_remote_code=True)
        self.model.to(device)
        self.model.eval()
        self.device = device
        self.dtype = torch.float16 if fp16 else torch.float32
        self.progress = progress

    @classmethod
    def from_components(cls, tokenizer, model, device: str, fp16: bool = False, progress: bool = False) -> "TextEncoder":
        """Build from an already-loaded tokenizer/model (offline use, tests)."""
        self = object.__new__(cls)
        self.tokenizer, self.model, self.device = tokenizer, model, device
        self.dtype = torch.float16 if fp16 else torch.float32
        self.progress = progress
        return self

    # Reduced-precision forward (BASELINE config 5; not in the reference, whose `fp16` flag only changes the pooling
    # dtype, scripts/text_encode_utils.py:28-30): set to torch.bfloat16 / torch.float16 and the transformer runs under
    # torch.autocast on the device; the pooling kernel takes the half-precision hidden states as they are.
    autocast: Optional[torch.dtype] = None

    # hipGraph replay of the transformer forward + pooling (ON by default).  A batch of questions is a few dozen short rows: the
    # ~200 kernels of a 12-layer forward finish faster than PyTorch can launch them, so the stage runs at the speed of the host
    # (measured: 4.14 -> 2.40 ms per batch of 32 questions, embeddings bit-identical).  The forward of a (batch, padded length)
    # shape is captured the `graph_after`-th time that shape is met (torch.cuda.CUDAGraph over static input buffers, the
    # pooling kernel included) and replayed from then on: one launch per batch, the same kernels on the same shapes — a shape
    # met once (the ragged last batch of a table) never pays a capture.  Shapes are cached up to max_graphs (oldest dropped).
    # A model whose forward cannot be captured (host-side control flow on device values, as some remote-code models have)
    # makes the first capture raise: the encoder then switches itself to the eager path for good and says so once.
    use_graphs: bool = True
    graph_after: int = 2
    max_graphs: int = 64

    def _forward_pooled(self, inputs) -> torch.Tensor:
        if self.autocast is not None:
            with torch.autocast(device_type="cuda", dtype=self.autocast):
                hidden = self.model(**inputs).last_hidden_state
        else:
            hidden = self.model(**inputs).last_hidden_state
        return masked_mean_pool(hidden, inputs["attention_mask"], fp16=self.dtype == torch.float16)

    def _forward_pooled_graphed(self, host_inputs) -> torch.Tensor:
        dev = torch.device(self.device)
        key = self._shape_key(host_inputs)
        cache = self.__dict__.setdefault("_graphs", {})
        entry = cache.get(key)
        if entry is None:
            static = {k: v.to(dev) for k, v in host_inputs.items()}
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up outside the capture: library handles, workspaces, autotuning
                for _ in range(2):
                    self._forward_pooled(static)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward_pooled(static)
            while cache and len(cache) >= max(int(self.max_graphs), 1):
                cache.pop(next(iter(cache)))
            entry = cache[key] = (graph, static, out)
        graph, static, out = entry
        for k, v in host_inputs.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return out.clone()

    @torch.no_grad()
    def encode_to_device(self, texts: Sequence[str], batch_size: int) -> torch.Tensor:
        """As `encode`, but the result stays in HBM: no per-batch device-to-host sync."""
        if not texts:
            return torch.empty((0, 0), dtype=torch.float32, device=self.device)
        pooled: List[torch.Tensor] = []
        for start, end in _iter_batches(len(texts), batch_size):
            inputs = self.tokenizer(list(texts[start:end]), padding=True, truncation=True, return_tensors="pt")
            if self._wants_graph(inputs):
                try:
                    pooled.append(self._forward_pooled_graphed(dict(inputs)))
                    continue
                except Exception as exc:  # noqa: BLE001 - capture refused by the model's forward: eager from now on
                    import warnings

                    self.use_graphs = False
                    self.__dict__.pop("_graphs", None)
                    torch.cuda.synchronize(torch.device(self.device))
                    warnings.warn(f"TextEncoder: hipGraph capture of the encoder forward failed ({type(exc).__name__}: {exc}); "
                                  "running eagerly from now on", RuntimeWarning, stacklevel=2)
            inputs = {k: v.to(self.device) for k, v in inputs.items()}
            pooled.append(self._forward_pooled(inputs))
        return torch.cat(pooled, dim=0)

    def _wants_graph(self, host_inputs) -> bool:
        """Replay (or capture) this batch?  Only on a HIP device, outside someone else's capture, for tensors-only tokenizer
        output, and from the `graph_after`-th meeting of the shape on (counted in eager passes too)."""
        if not self.use_graphs or torch.device(self.device).type != "cuda":
            return False
        try:
            items = dict(host_inputs).items()
            if not all(isinstance(v, torch.Tensor) for _, v in items):
                return False
        except Exception:  # noqa: BLE001
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        key = self._shape_key(dict(host_inputs))
        if key in self.__dict__.get("_graphs", {}):
            return True
        seen = self.__dict__.setdefault("_shape_seen", {})
        if len(seen) > 4096:
            seen.clear()
        seen[key] = seen.get(key, 0) + 1
        return seen[key] >= max(int(self.graph_after), 1)

    def _shape_key(self, host_inputs):
        return (tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in host_inputs.items())), str(self.autocast), str(self.dtype))

    @torch.no_grad()
    def encode(self, texts: Sequence[str], batch_size: int, show_progress: Optional[bool] = None,
               desc: Optional[str] = None) -> torch.Tensor:
        if not texts:
            return torch.empty((0, 0), dtype=torch.float32)
        return self.encode_to_device(texts, batch_size).to("cpu", dtype=torch.float32)


def encode_to_memmap(encoder: TextEncoder, texts: Sequence[str], emb_ids: Sequence[int], batch_size: int,
                     max_embedding_id: int, out_path: Path, desc: Optional[str], show_progress: bool) -> torch.Tensor:
    """Same contract as the reference (scripts/text_encode_utils.py:70-112): a zero-initialised
    [(max_id + 1), D] f32 table with row emb_id = embedding, saved with torch.save to out_path and
    returned (CPU).  The table is assembled in HBM; no host memmap is needed."""
    if max_embedding_id < 0:
        return torch.empty((0, 0), dtype=torch.float32)
    if len(texts) != len(emb_ids):
        raise ValueError("texts and emb_ids must have the same length")
    if not texts:
        tensor = torch.zeros((max_embedding_id + 1, 0), dtype=torch.float32)
        torch.save(tensor, out_path)
        return tensor
    table = None
    for start, end in _iter_batches(len(texts), batch_size):
        emb = encoder.encode_to_device(list(texts[start:end]), batch_size)
        if table is None:
            dim = int(emb.shape[1]) if emb.numel() > 0 else 0
            table = torch.zeros((max_embedding_id + 1, dim), dtype=torch.float32, device=emb.device)
        ids = torch.as_tensor(list(emb_ids[start:end]), dtype=torch.int64, device=emb.device)
        scatter_rows(table, emb, ids)
    tensor = table.to("cpu")
    torch.save(tensor, out_path)
    return tensor


def encode_tables(encoder: TextEncoder, *, entity_embedding_records: Sequence[dict], entity_struct_records: Sequence[dict],
                  relation_records: Sequence[dict], batch_size: int, embeddings_out_dir, precompute_entities: bool = True,
                  precompute_relations: bool = True, show_progress: bool = False):
    """The entity / relation encode sequence of the offline pipeline (scripts/build_retrieval_pipeline.py:1262-1309), with
    the vocabulary records the reference takes them from (`EntityVocab.embedding_records` / `.struct_records`,
    `RelationVocab.records` = the rows of embedding_vocab / entity_vocab / relation_vocab.parquet):

      * entity labels sorted by `embedding_id` (:1270-1275) -> `encode_to_memmap` into a zero table of
        max(struct_records.embedding_id, default 0) + 1 rows (:1276-1287; row 0 = the non-text placeholder stays zero)
        -> `entity_embeddings.pt`;
      * relation labels sorted by `relation_id` (:1290-1300) -> `encoder.encode` -> `relation_embeddings.pt` [R, D].

    Returns (entity_table or None, relation_table or None) as CPU f32 tensors, files written like the reference's."""
    out_dir = Path(embeddings_out_dir)
    if precompute_entities or precompute_relations:
        out_dir.mkdir(parents=True, exist_ok=True)
    entity_table = relation_table = None
    if precompute_entities:
        emb_rows = sorted(((rec["embedding_id"], rec.get("label", "")) for rec in entity_embedding_records), key=lambda x: x[0])
        text_labels = [str(label) for _, label in emb_rows]
        text_ids = [int(eid) for eid, _ in emb_rows]
        max_embedding_id = max((int(rec["embedding_id"]) for rec in entity_struct_records), default=0)
        entity_table = encode_to_memmap(encoder, text_labels, text_ids, batch_size, max_embedding_id,
                                        out_dir / "entity_embeddings.pt", "Entities", show_progress)
    if precompute_relations:
        relation_rows = sorted(((rec["relation_id"], rec.get("label", "")) for rec in relation_records), key=lambda x: x[0])
        relation_labels = [str(label) for _, label in relation_rows]
        relation_table = encoder.encode(relation_labels, batch_size, show_progress=show_progress, desc="Relations")
        torch.save(relation_table, out_dir / "relation_embeddings.pt")
    return entity_table, relation_table


def encode_questions(encoder: TextEncoder, questions: Sequence[str], batch_size: int, chunk_size: int = 2000) -> List[List[float]]:
    """Question embeddings as the pipeline's pass 2 makes them (scripts/build_retrieval_pipeline.py:1318-1334, 1360): the
    samples are processed in chunks of `chunk_size` (parquet_chunk_size), every chunk's question texts go through
    `encoder.encode(texts, batch_size)` on their own — so batch boundaries restart at each chunk — and a sample's
    `question_emb` column is the list of floats of its row.  One D2H copy per chunk instead of one `.tolist()` per row."""
    out: List[List[float]] = []
    for start in range(0, len(questions), max(int(chunk_size), 1)):
        chunk = [str(q) for q in questions[start: start + chunk_size]]
        emb = encoder.encode(chunk, batch_size, show_progress=False, desc="Questions")
        out.extend(emb.tolist())
    return out


__all__ = ["TextEncoder", "encode_to_memmap", "encode_tables", "encode_questions", "masked_mean_pool", "scatter_rows",
           "ENCODER_EPS"]
