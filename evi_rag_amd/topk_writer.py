"""Host-side mirror of `RetrieverTopKEdgeWriter` (src/callbacks/retriever_topk_edge_writer.py):
persists the per-question top-k edges of a retriever evaluation as `eval_retriever/<split>.pt`
plus a manifest, in the reference's schema, so the downstream oracle / reasoner stages
(src/data/reasoner_oracle_datamodule.py:98-145) consume it unchanged.

Differences in mechanism, not in contract:
  * the per-graph `torch.topk` + `.tolist()` loop (:294-320) is one fused ranking pass per batch
    (`evi_retriever_metrics` with the top-k lists requested) and ONE device-to-host copy of fixed-shape
    [B, k_max] arrays;
  * across ranks the pickled `dist.all_gather_object` (:450-462) is replaced by fixed-shape tensor
    all-gathers of those arrays (counts first, then padded payload); rank 0 formats the records.
It is a plain class with the Lightning hook names (no Lightning import is needed to use it).
"""
from __future__ import annotations

import json
import logging
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from collections import namedtuple

from . import ops
from .metrics import normalize_k_values
from .retriever import RetrieverOutput

RankedLists = namedtuple("RankedLists", "topk_score topk_count")

logger = logging.getLogger(__name__)

_DEFAULT_TOPK_VALUES = (1, 5, 10)


def _attr(obj: Any, name: str) -> Any:
    return obj.get(name) if isinstance(obj, dict) else getattr(obj, name, None)


def gather_padded(t: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather tensors whose first dimension differs per rank: lengths first, then rows padded to
    the longest.  Works on CPU (gloo) and GPU (RCCL) tensors; returns the per-rank tensors."""
    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    world = dist.get_world_size(group)
    n = torch.tensor([t.size(0)], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.size(0)] = t
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:s] for o, s in zip(out, sizes)]


class RetrieverTopKEdgeWriter:
    """Same constructor kwargs, hooks and output files as the reference callback (:68-187)."""

    def __init__(self, *, output_dir, split: str = "test", enabled: bool = True, artifact_name: str = "eval_retriever",
                 schema_version: int = 1, topk_values: Optional[Sequence[int]] = None, textualize: bool = False,
                 entity_vocab_path: Optional[str] = None, relation_vocab_path: Optional[str] = None,
                 overwrite: bool = True) -> None:
        self.enabled = bool(enabled)
        self.output_dir = Path(output_dir)
        self.split = str(split)
        self.artifact_name = str(artifact_name).strip()
        if not self.artifact_name:
            raise ValueError("artifact_name must be a non-empty string.")
        self.schema_version = int(schema_version)
        if self.schema_version <= 0:
            raise ValueError("schema_version must be a positive integer.")
        self.topk_values = normalize_k_values(topk_values, default=_DEFAULT_TOPK_VALUES)
        if not self.topk_values:
            raise ValueError("topk_values must be a non-empty list of positive integers.")
        self._max_topk = max(self.topk_values)
        self.textualize = bool(textualize)
        self.entity_vocab_path = entity_vocab_path
        self.relation_vocab_path = relation_vocab_path
        self.overwrite = bool(overwrite)
        self._entity_map: Optional[Dict[int, str]] = None
        self._relation_map: Optional[Dict[int, str]] = None
        self._output_path: Optional[Path] = None
        self._manifest_path: Optional[Path] = None
        self._chunks: List[Dict[str, Any]] = []  # per batch: fixed-shape CPU arrays + per-sample metadata

    # ---- hooks (names and signatures of the Lightning callback) --------------------------------------
    def on_predict_start(self, trainer=None, pl_module=None) -> None:
        if not self.enabled:
            return
        self._chunks = []
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self._output_path = self.output_dir / f"{self.split}.pt"
        self._manifest_path = self.output_dir / f"{self.split}.manifest.json"
        if self._output_path.exists() and not self.overwrite:
            raise FileExistsError(f"Output path already exists: {self._output_path}")
        if self.textualize:
            self._entity_map, self._relation_map = self._resolve_vocab_maps()

    on_test_start = on_predict_start

    def write_on_batch_end(self, trainer, pl_module, prediction, batch_indices, batch, batch_idx, dataloader_idx) -> None:
        self._collect_prediction(prediction=prediction, batch=batch)

    def on_test_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx: int = 0) -> None:
        self._collect_prediction(prediction=outputs, batch=batch)

    def on_predict_end(self, trainer=None, pl_module=None) -> None:
        if not self.enabled:
            return
        records = self._gather_records()
        if not records:
            return
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if rank != 0:
            return
        if self._output_path is None or self._manifest_path is None:
            raise RuntimeError("RetrieverTopKEdgeWriter missing output path; on_predict_start was not called.")
        payload = {"settings": {"split": self.split, "topk_values": [int(k) for k in self.topk_values]},
                   "samples": records}
        torch.save(payload, self._output_path)
        manifest = {"artifact": self.artifact_name, "schema_version": self.schema_version, "file": self._output_path.name,
                    "created_at": datetime.now(timezone.utc).replace(tzinfo=None).isoformat(timespec="seconds") + "Z",
                    "producer": "retriever_topk_edge_writer"}
        self._manifest_path.write_text(json.dumps(manifest, indent=2), encoding="utf-8")

    on_test_end = on_predict_end

    # ---- per batch: one ranking pass, one D2H copy ----------------------------------------------------
    def _collect_prediction(self, *, prediction: Any, batch: Any) -> None:
        if not self.enabled or prediction is None:
            return
        output = self._coerce_prediction(prediction)
        if output is None:
            return
        for name, err in (("edge_index", "Batch missing edge_index required for persistence."),
                          ("edge_attr", "Batch missing edge_attr required for persistence."),
                          ("node_global_ids", "Batch missing node_global_ids required for persistence."),
                          ("answer_entity_ids", "Batch missing answer_entity_ids required for persistence.")):
            if _attr(batch, name) is None:
                raise ValueError(err)
        scores = torch.sigmoid(output.logits.detach().view(-1).float())  # :209
        if scores.numel() == 0:
            return
        labels = torch.as_tensor(_attr(batch, "labels")).detach().view(-1).to(scores.device, torch.float32)
        if scores.numel() != labels.numel():
            raise ValueError(f"scores/labels shape mismatch: {scores.shape} vs {labels.shape}")
        slice_dict = _attr(batch, "_slice_dict")
        slice_dict = slice_dict if isinstance(slice_dict, dict) else {}
        answer_ptr = _attr(batch, "answer_entity_ids_ptr")
        if answer_ptr is None:
            answer_ptr = slice_dict.get("answer_entity_ids")
        if answer_ptr is None:
            raise ValueError("Batch missing answer_entity_ids_ptr required for persistence.")
        answer_ptr = torch.as_tensor(answer_ptr, dtype=torch.long).view(-1).tolist()
        num_graphs = len(answer_ptr) - 1
        dev = scores.device
        query_ids = output.query_ids.detach().view(-1).to(dev)
        # graph boundaries like the reference (:212-215, :415-422): the collate's edge ptr when the batch carries one of the
        # right length, else the edges of graph g are those with query_ids == g, in their stored order (:269-281)
        edge_ptr = None
        cand = slice_dict.get("edge_index")
        if cand is not None:
            cand = torch.as_tensor(cand, dtype=torch.long).view(-1)
            if cand.numel() == num_graphs + 1:
                edge_ptr = cand.to(dev)
        perm = None
        if edge_ptr is None:
            if query_ids.numel() != scores.numel():
                raise ValueError(f"query_ids/scores mismatch: {query_ids.shape} vs {scores.shape}")
            if bool((query_ids[1:] < query_ids[:-1]).any()):  # unsorted edge->graph ids: group them, order kept inside a graph
                perm = torch.argsort(query_ids, stable=True)
                query_ids = query_ids[perm]
            counts = torch.bincount(query_ids.clamp(min=0), minlength=num_graphs)[:num_graphs]
            edge_ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        ranked = scores if perm is None else scores[perm]
        # one segmented top-k launch for the whole batch (evi_segment_topk: (score desc, position asc) — the reference's
        # torch.topk leaves the order of equal scores unspecified), replacing the per-graph torch.topk loop (:294-302)
        topk_index, topk_score, topk_count = ops.segment_topk(ranked, edge_ptr, self._max_topk)
        pos = topk_index.long().clamp(min=0) + edge_ptr[:-1].view(-1, 1)  # positions in the (grouped) edge order [B, kmax]
        pos = pos.clamp(max=max(scores.numel() - 1, 0))  # padding slots (count < k_max) are never read back
        if perm is not None:
            pos = perm[pos]
        rb = RankedLists(topk_score=topk_score, topk_count=topk_count)
        ei = torch.as_tensor(_attr(batch, "edge_index")).to(dev)
        gids = torch.as_tensor(_attr(batch, "node_global_ids")).to(dev).view(-1)
        rel = torch.as_tensor(_attr(batch, "edge_attr")).to(dev).view(-1)
        gather = lambda t: t[pos]  # noqa: E731
        chunk = {"count": rb.topk_count.cpu(), "score": rb.topk_score.cpu(), "head": gather(gids[ei[0]]).cpu(),
                 "tail": gather(gids[ei[1]]).cpu(), "rel": gather(rel).cpu(), "label": gather(labels).cpu()}
        for key, src in (("lf", output.logits_fwd), ("lb", output.logits_bwd)):
            chunk[key] = gather(src.detach().view(-1).float()).cpu() if src is not None else None
        ans = torch.as_tensor(_attr(batch, "answer_entity_ids")).view(-1).cpu().tolist()
        chunk["answers"] = [[int(x) for x in ans[answer_ptr[g]: answer_ptr[g + 1]]] for g in range(num_graphs)]
        chunk["sample_ids"] = self._extract_sample_ids(batch, num_graphs)
        chunk["questions"] = self._extract_questions(batch, num_graphs)
        self._chunks.append(chunk)

    @staticmethod
    def _extract_sample_ids(batch: Any, num_graphs: int) -> List[str]:
        """src/utils/metrics.py:46-55 (`extract_sample_ids`): list -> strings, tensor -> str(item), scalar -> one entry,
        absent -> graph numbers; a graph beyond the list gets its number (:386-389)."""
        raw = _attr(batch, "sample_id")
        if raw is None:
            return [str(i) for i in range(num_graphs)]
        if isinstance(raw, (list, tuple)):
            return [str(s) for s in raw]
        if torch.is_tensor(raw):
            return [str(s.item()) for s in raw]
        return [str(raw)]

    @staticmethod
    def _extract_questions(batch: Any, num_graphs: int) -> List[str]:
        """retriever_topk_edge_writer.py:403-414."""
        raw = _attr(batch, "question")
        if raw is None:
            return ["" for _ in range(num_graphs)]
        if isinstance(raw, (list, tuple)):
            return [str(q) for q in raw]
        if torch.is_tensor(raw):
            if raw.numel() == num_graphs:
                return [str(v.item()) for v in raw]
            return [str(raw.cpu().tolist()) for _ in range(num_graphs)]
        return [str(raw) for _ in range(num_graphs)]

    def _records_from_chunk(self, c: Dict[str, Any]) -> List[Dict[str, Any]]:
        records = []
        for g, m in enumerate(c["count"].tolist()):
            if m <= 0:
                continue  # graphs without edges produce no record (:275-277, :287-289)
            cols = {k: c[k][g, :m].tolist() for k in ("score", "head", "tail", "rel", "label")}
            lf = c["lf"][g, :m].tolist() if c["lf"] is not None else None
            lb = c["lb"][g, :m].tolist() if c["lb"] is not None else None
            triplets_by_k: Dict[int, List[Dict[str, Any]]] = {}
            for k in self.topk_values:
                rows = []
                for i in range(min(int(k), m)):
                    h, r, t = int(cols["head"][i]), int(cols["rel"][i]), int(cols["tail"][i])
                    rec = {"head_entity_id": h, "relation_id": r, "tail_entity_id": t,
                           "head_text": self._lookup(self._entity_map, h), "relation_text": self._lookup(self._relation_map, r),
                           "tail_text": self._lookup(self._entity_map, t), "score": float(cols["score"][i]),
                           "label": float(cols["label"][i]), "rank": i + 1}
                    if lf is not None:
                        rec["logit_fwd"] = float(lf[i])
                    if lb is not None:
                        rec["logit_bwd"] = float(lb[i])
                    rows.append(rec)
                triplets_by_k[int(k)] = rows
            records.append({"sample_id": c["sample_ids"][g] if g < len(c["sample_ids"]) else str(g),
                            "question": c["questions"][g] if g < len(c["questions"]) else "",
                            "triplets_by_k": triplets_by_k, "answer_entity_ids": c["answers"][g]})
        return records

    def _gather_records(self) -> List[Dict[str, Any]]:
        records: List[Dict[str, Any]] = []
        for c in self._chunks:
            records.extend(self._records_from_chunk(c))
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return records
        # fixed-shape exchange: every rank contributes its formatted records as one uint8 tensor
        import io

        buf = io.BytesIO()
        torch.save(records, buf)
        local = torch.frombuffer(bytearray(buf.getvalue()), dtype=torch.uint8)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        parts = gather_padded(local.to(dev))
        if dist.get_rank() != 0:
            return []
        merged: List[Dict[str, Any]] = []
        for p in parts:
            if p.numel():
                merged.extend(torch.load(io.BytesIO(p.cpu().numpy().tobytes()), weights_only=False))
        return merged

    @staticmethod
    def _lookup(mapping: Optional[Dict[int, str]], key: Optional[int]) -> Optional[str]:
        if mapping is None or key is None:
            return None
        return mapping.get(key)

    def _resolve_vocab_maps(self):
        if not self.entity_vocab_path or not self.relation_vocab_path:
            raise ValueError("textualize=true requires both entity_vocab_path and relation_vocab_path.")
        entity_path, relation_path = Path(self.entity_vocab_path), Path(self.relation_vocab_path)
        if not entity_path.exists():
            raise FileNotFoundError(f"entity_vocab_path not found: {entity_path}")
        if not relation_path.exists():
            raise FileNotFoundError(f"relation_vocab_path not found: {relation_path}")
        ent_map: Optional[Dict[int, str]] = None
        rel_map: Optional[Dict[int, str]] = None
        try:  # like the reference (:430-446): an unreadable vocabulary is a warning on rank 0, the texts stay None
            import pyarrow.parquet as pq

            ent = pq.read_table(entity_path, columns=["entity_id", "label"])
            ent_map = {int(i): str(l) for i, l in zip(ent.column("entity_id").to_pylist(), ent.column("label").to_pylist())
                       if i is not None and l is not None}
            rel = pq.read_table(relation_path, columns=["relation_id", "label"])
            rel_map = {int(i): str(l) for i, l in zip(rel.column("relation_id").to_pylist(), rel.column("label").to_pylist())
                       if i is not None and l is not None}
        except Exception as exc:
            if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
                logger.warning("Failed to load vocab for textualize: %s", exc)
        return ent_map, rel_map

    @staticmethod
    def _coerce_prediction(prediction: Any) -> Optional[RetrieverOutput]:
        if isinstance(prediction, RetrieverOutput):
            return prediction
        if isinstance(prediction, dict) and "logits" in prediction and "query_ids" in prediction:
            return RetrieverOutput(logits=prediction["logits"], query_ids=prediction["query_ids"],
                                   relation_ids=prediction.get("relation_ids"), logits_fwd=prediction.get("logits_fwd"),
                                   logits_bwd=prediction.get("logits_bwd"))
        return None


__all__ = ["RetrieverTopKEdgeWriter", "gather_padded"]
