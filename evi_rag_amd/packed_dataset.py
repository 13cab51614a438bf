"""HBM-resident retrieval split and device-side batch collation.

The reference keeps one pickled dict per sample in LMDB (schema written at
scripts/build_retrieval_pipeline.py:2200-2228), unpickles them in 16 DataLoader workers
(`GRetrievalDataset.get/_build_data`, src/data/g_retrieval_dataset.py:99-154), lets PyG's `Collater`
concatenate them (`GRetrievalData.__inc__`, :29-37; `RetrievalCollater`, src/data/components/loader.py:22-99)
and gathers embeddings on the CPU.  A whole WebQSP / CWQ split is a few hundred MB to a few GB of
integers, and an MI355X has 288 GB: here the split is written once as FLAT arrays (every sample's items
concatenated, one pointer array per field family), loaded into HBM, and a batch is B segment copies per
field (`evi_gather_segments`) with the collate increments applied on the fly, then the embedding
gather and edge->graph assignment on the device.  No worker processes, no pickles, no H2D per batch.

    write_packed(dir, samples)         samples: iterable of dicts with the LMDB sample keys
    PackedRetrievalDataset(dir)        the resident split
    PackedLoader(dataset, batch_size)  iterates batches with the attributes of the PyG `Batch` the
                                       reference loader yields (edge_index, ptr, batch, edge_attr, labels, ...)
"""
from __future__ import annotations

import json
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib, ops
from .embedding_store import LazyEdgeEmbeddings

FORMAT_VERSION = 1

# field -> (pointer family, numpy dtype, trailing width | None, collate increment)
#   increment "nodes": += first node of the sample in the batch; "edges": += its first edge (GRetrievalData.__inc__)
_FIELDS = {
    "edge_src": ("edge", np.int64, None, "nodes"),
    "edge_dst": ("edge", np.int64, None, "nodes"),
    "edge_attr": ("edge", np.int64, None, None),
    "labels": ("edge", np.float32, None, None),
    "node_global_ids": ("node", np.int64, None, None),
    "node_embedding_ids": ("node", np.int64, None, None),
    "topic_one_hot": ("node", np.float32, "topics", None),
    "q_local_indices": ("q", np.int64, None, "nodes"),
    "a_local_indices": ("a", np.int64, None, "nodes"),
    "answer_entity_ids": ("answer", np.int64, None, None),
    "seed_entity_ids": ("seed", np.int64, None, None),
    "pair_start_node_locals": ("pair", np.int64, None, "nodes"),
    "pair_answer_node_locals": ("pair", np.int64, None, "nodes"),
    "pair_edge_counts": ("pair", np.int64, None, None),
    "pair_shortest_lengths": ("pair", np.int64, None, None),
    "pair_edge_local_ids": ("pair_edge", np.int64, None, "edges"),
}
_FAMILIES = ("node", "edge", "q", "a", "answer", "seed", "pair", "pair_edge")
_REQUIRED = ("edge_index", "edge_attr", "labels", "num_nodes", "node_global_ids", "node_embedding_ids", "question_emb",
             "topic_one_hot", "q_local_indices", "a_local_indices", "answer_entity_ids")


def _np(x, dtype) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


def write_packed(out_dir: Union[str, Path], samples: Iterable[Dict[str, Any]]) -> Dict[str, Any]:
    """Writes the flat split.  Each sample carries the reference's core keys (and optionally the aux keys
    question, seed_entity_ids, pair_*).  Missing keys raise KeyError like `_validate_raw_sample`."""
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    cols: Dict[str, List[np.ndarray]] = {k: [] for k in _FIELDS}
    counts: Dict[str, List[int]] = {f: [] for f in _FAMILIES}
    q_emb: List[np.ndarray] = []
    sample_ids: List[str] = []
    questions: List[str] = []
    topics = None
    for i, smp in enumerate(samples):
        sid = str(smp.get("sample_id", i))
        for key in _REQUIRED:
            if key not in smp:
                raise KeyError(f"Sample {sid} missing key: {key}")
        ei = _np(smp["edge_index"], np.int64).reshape(2, -1)
        n = int(smp["num_nodes"])
        if ei.size and (ei.min() < 0 or ei.max() >= n):
            raise ValueError(f"Sample {sid}: edge_index out of range for num_nodes={n}")
        toh = _np(smp["topic_one_hot"], np.float32).reshape(n, -1)
        topics = toh.shape[1] if topics is None else topics
        if toh.shape[1] != topics:
            raise ValueError(f"Sample {sid}: topic_one_hot width {toh.shape[1]} != {topics}")
        fam_len = {"node": n, "edge": ei.shape[1]}
        data = {"edge_src": ei[0], "edge_dst": ei[1], "topic_one_hot": toh}
        for key, (fam, dt, _, _) in _FIELDS.items():
            if key in data:
                arr = data[key]
            elif key not in smp and fam_len.get(fam, 0) > 0:
                arr = np.full(fam_len[fam], -1, dt)  # optional aux field absent (e.g. pair_shortest_lengths)
            else:
                arr = _np(smp.get(key, []), dt).reshape(-1)
            if fam in fam_len and arr.shape[0] != fam_len[fam]:
                raise ValueError(f"Sample {sid}: {key} has {arr.shape[0]} rows, expected {fam_len[fam]}")
            fam_len.setdefault(fam, arr.shape[0])
            if arr.shape[0] != fam_len[fam]:
                raise ValueError(f"Sample {sid}: {key} length {arr.shape[0]} disagrees with its family '{fam}' ({fam_len[fam]})")
            cols[key].append(arr)
        for fam in _FAMILIES:
            counts[fam].append(int(fam_len.get(fam, 0)))
        q_emb.append(_np(smp["question_emb"], np.float32).reshape(1, -1))
        sample_ids.append(sid)
        questions.append(str(smp.get("question", "")))
    S = len(sample_ids)
    for fam in _FAMILIES:
        np.save(out / f"ptr_{fam}.npy", np.concatenate([[0], np.cumsum(counts[fam])]).astype(np.int64))
    for key, (_, dt, width, _) in _FIELDS.items():
        if cols[key]:
            arr = np.concatenate(cols[key])
        else:
            arr = np.empty((0, topics or 2) if width else (0,), dt)
        np.save(out / f"{key}.npy", arr)
    qe = np.concatenate(q_emb) if q_emb else np.empty((0, 0), np.float32)
    np.save(out / "question_emb.npy", qe)
    meta = {"format_version": FORMAT_VERSION, "num_samples": S, "emb_dim": int(qe.shape[1]) if S else 0,
            "num_topics": int(topics or 0), "sample_ids": sample_ids, "questions": questions}
    (out / "meta.json").write_text(json.dumps(meta))
    return meta


class PackedRetrievalDataset:
    """A split resident in HBM.  `embeddings` (a GlobalEmbeddingStore) is optional; with it every batch
    carries node_embeddings / edge_embeddings like the reference's collater attaches."""

    def __init__(self, root: Union[str, Path], *, device: Union[str, torch.device, None] = None, embeddings=None) -> None:
        self.root = Path(root)
        meta_path = self.root / "meta.json"
        if not meta_path.exists():
            raise FileNotFoundError(f"Packed split not found at {self.root} (meta.json missing)")
        self.meta = json.loads(meta_path.read_text())
        if self.meta.get("format_version") != FORMAT_VERSION:
            raise ValueError(f"Unsupported packed format version {self.meta.get('format_version')}")
        if not torch.cuda.is_available():
            raise RuntimeError("PackedRetrievalDataset keeps the split in HBM: no GPU visible (no CPU fallback)")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.sample_ids: List[str] = list(self.meta["sample_ids"])
        self.questions: List[str] = list(self.meta["questions"])
        self.global_embeddings = embeddings
        load = lambda name: torch.from_numpy(np.load(self.root / f"{name}.npy", mmap_mode="r")[...].copy()).to(self.device)  # noqa: E731
        self.ptr = {fam: load(f"ptr_{fam}") for fam in _FAMILIES}
        self.ptr_h = {fam: np.load(self.root / f"ptr_{fam}.npy") for fam in _FAMILIES}  # host copies: batch offsets need no device read
        self.cols = {key: load(key) for key in _FIELDS}
        self.question_emb = load("question_emb")
        self.num_topics = int(self.meta["num_topics"])

    def __len__(self) -> int:
        return len(self.sample_ids)

    def len(self) -> int:
        return len(self)

    def nbytes(self) -> int:
        return int(sum(t.numel() * t.element_size() for t in list(self.ptr.values()) + list(self.cols.values()))
                   + self.question_emb.numel() * 4)

    # ---- collation ------------------------------------------------------------------------------------
    def collate(self, indices: Union[Sequence[int], torch.Tensor]) -> SimpleNamespace:
        """The batch the reference's loader yields for these samples, assembled on the device."""
        dev = self.device
        ids_h = np.asarray(indices.cpu() if isinstance(indices, torch.Tensor) else indices, dtype=np.int64).reshape(-1)
        B = int(ids_h.size)
        if B == 0:
            raise ValueError("cannot collate an empty batch")
        S = len(self)
        if ids_h.min() < 0 or ids_h.max() >= S:
            raise IndexError(f"sample index out of range for a split of {S} samples")
        lib = _lib.load()
        st = ops._stream(dev)
        # batch offsets of every field family from the host copies of the pointer arrays: one small H2D copy,
        # no device->host read anywhere in the collation (evi_segment_offsets does the same on the device
        # for callers whose sample ids live there)
        offs = np.zeros((len(_FAMILIES), B + 1), np.int64)
        for f, fam in enumerate(_FAMILIES):
            p = self.ptr_h[fam]
            np.cumsum(p[ids_h + 1] - p[ids_h], out=offs[f, 1:])
        packed = torch.from_numpy(np.concatenate([ids_h, offs.reshape(-1)])).to(dev, non_blocking=True)
        ids = packed[:B]
        out_ptr = {fam: packed[B + f * (B + 1): B + (f + 1) * (B + 1)] for f, fam in enumerate(_FAMILIES)}
        totals = offs[:, -1]
        total = dict(zip(_FAMILIES, (int(t) for t in totals)))
        inc = {"nodes": out_ptr["node"], "edges": out_ptr["edge"], None: None}
        got: Dict[str, torch.Tensor] = {}
        for key, (fam, dt, width, how) in _FIELDS.items():
            src = self.cols[key]
            row = int(src.size(1)) if src.dim() == 2 else 1
            shape = (total[fam], row) if src.dim() == 2 else (total[fam],)
            dst = torch.empty(shape, dtype=src.dtype, device=dev)
            if total[fam] > 0:
                add = inc[how]
                _lib.check(lib.evi_gather_segments(src.data_ptr(), src.element_size(), row, self.ptr[fam].data_ptr(), S,
                                                   ids.data_ptr(), B, out_ptr[fam].data_ptr(),
                                                   add.data_ptr() if add is not None else None, dst.data_ptr(), st))
            got[key] = dst
        b = LazyEdgeEmbeddings()  # a SimpleNamespace whose [E, D] edge_embeddings are gathered only if somebody reads them
        b.edge_index = torch.stack([got.pop("edge_src"), got.pop("edge_dst")]).contiguous()
        for key, val in got.items():
            setattr(b, key, val)
        b.ptr = out_ptr["node"]
        b.edge_ptr = out_ptr["edge"]
        b.num_graphs, b.num_nodes = B, total["node"]
        graph_ids = torch.arange(B, device=dev)  # output sizes are known on the host: no device read
        b.batch = torch.repeat_interleave(graph_ids, b.ptr[1:] - b.ptr[:-1], output_size=total["node"])
        b.edge_batch = torch.repeat_interleave(graph_ids, b.edge_ptr[1:] - b.edge_ptr[:-1], output_size=total["edge"])
        b.question_emb = self.question_emb.index_select(0, ids)
        b.answer_entity_ids_ptr = out_ptr["answer"]
        b._slice_dict = {"edge_index": out_ptr["edge"], "q_local_indices": out_ptr["q"], "a_local_indices": out_ptr["a"],
                         "answer_entity_ids": out_ptr["answer"], "pair_edge_local_ids": out_ptr["pair_edge"],
                         "pair_start_node_locals": out_ptr["pair"], "seed_entity_ids": out_ptr["seed"]}
        b.idx = ids
        b.sample_id = [self.sample_ids[i] for i in ids_h.tolist()]
        b.question = [self.questions[i] for i in ids_h.tolist()]
        if self.global_embeddings is not None:
            self.global_embeddings.attach(b, check=False)  # range check deferred: check_deferred()
        return b

    def check_deferred(self) -> None:
        """Raises if any batch collated since the last call carried an embedding id outside its table."""
        if self.global_embeddings is not None:
            self.global_embeddings.raise_if_failed()

    def load_sample(self, sample_id: str) -> Dict[str, Any]:
        """Per-sample metadata with the LMDB keys `GAgentBuilder` reads (question_emb, question,
        seed_entity_ids, answer_entity_ids): lets the split stand in for its `EmbeddingStore`."""
        if not hasattr(self, "_index_of"):
            self._index_of = {sid: i for i, sid in enumerate(self.sample_ids)}
            self._host = {k: self.cols[k].cpu() for k in ("seed_entity_ids", "answer_entity_ids")}
            self._host_ptr = {k: self.ptr[k].cpu().numpy() for k in ("seed", "answer")}
            self._host_q = self.question_emb.cpu()
        i = self._index_of[sample_id]  # KeyError when absent, like LMDB
        sp, ap = self._host_ptr["seed"], self._host_ptr["answer"]
        return {"question_emb": self._host_q[i: i + 1], "question": self.questions[i],
                "seed_entity_ids": self._host["seed_entity_ids"][sp[i]: sp[i + 1]],
                "answer_entity_ids": self._host["answer_entity_ids"][ap[i]: ap[i + 1]]}


class PackedLoader:
    """Batches of a resident split.  shuffle uses a seeded torch.Generator like the reference's loader
    (`UnifiedDataLoader`, src/data/components/loader.py:108-160); drop_last as in torch's DataLoader."""

    def __init__(self, dataset: PackedRetrievalDataset, batch_size: int = 32, shuffle: bool = False,
                 random_seed: Optional[int] = None, drop_last: bool = False, rank: int = 0, world_size: int = 1) -> None:
        if batch_size <= 0:
            raise ValueError(f"batch_size must be positive, got {batch_size}")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), bool(shuffle), bool(drop_last)
        self.rank, self.world_size = int(rank), int(world_size)
        self._gen = torch.Generator()
        self._seed = None if random_seed is None else int(random_seed)
        if random_seed is not None:
            self._gen.manual_seed(int(random_seed))

    def set_epoch(self, epoch: int) -> None:
        """The permutation of epoch e is a function of (random_seed, e) (DistributedSampler.set_epoch's contract), so a run
        resumed at epoch e shuffles like the uninterrupted run did.  Without a seed the running generator is kept."""
        if self._seed is not None:
            self._gen.manual_seed(self._seed * 1_000_003 + int(epoch))

    def _order(self) -> torch.Tensor:
        n = len(self.dataset)
        order = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        return order[self.rank:: self.world_size]  # graphs never span ranks

    def __len__(self) -> int:
        n = len(range(self.rank, len(self.dataset), self.world_size))
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # (r03: collating batch i + 1 on a side stream under the consumer's work on batch i was built and measured: the evaluation
    # pipeline went from 8 650-9 300 to 8 100 questions/s — tensors allocated on one stream and consumed on another have to be
    # recorded on the consumer's stream, and the caching allocator then cannot recycle them in time.  Collation stays on the
    # consumer's stream.)
    def __iter__(self) -> Iterator[SimpleNamespace]:
        order = self._order()
        for lo in range(0, order.numel(), self.batch_size):
            chunk = order[lo: lo + self.batch_size]
            if self.drop_last and chunk.numel() < self.batch_size:
                return
            yield self.dataset.collate(chunk)


def samples_from_flat_batch(batch) -> List[Dict[str, Any]]:
    """Splits a flat batch (e.g. `synthetic.make_batch`) back into per-sample dicts with the LMDB keys."""
    out = []
    ptr, eptr = np.asarray(batch.ptr), np.asarray(batch.edge_ptr)
    for g in range(int(batch.num_graphs)):
        n0, n1, e0, e1 = int(ptr[g]), int(ptr[g + 1]), int(eptr[g]), int(eptr[g + 1])
        q = np.asarray(batch.q_local_indices[int(batch.q_ptr[g]): int(batch.q_ptr[g + 1])])
        a = np.asarray(batch.a_local_indices[int(batch.a_ptr[g]): int(batch.a_ptr[g + 1])])
        ans = np.asarray(batch.answer_entity_ids[int(batch.answer_ptr[g]): int(batch.answer_ptr[g + 1])])
        out.append({
            "sample_id": batch.sample_id[g] if len(getattr(batch, "sample_id", [])) > g else f"sample_{g}",
            "edge_index": np.asarray(batch.edge_index[:, e0:e1]) - n0, "edge_attr": np.asarray(batch.edge_attr[e0:e1]),
            "labels": np.asarray(batch.labels[e0:e1]), "num_nodes": n1 - n0,
            "node_global_ids": np.asarray(batch.node_global_ids[n0:n1]),
            "node_embedding_ids": np.asarray(batch.node_embedding_ids[n0:n1]),
            "question_emb": np.asarray(batch.question_emb[g: g + 1]), "topic_one_hot": np.asarray(batch.topic_one_hot[n0:n1]),
            "q_local_indices": q - n0, "a_local_indices": a - n0, "answer_entity_ids": ans,
            "seed_entity_ids": np.asarray(batch.node_global_ids)[q], "question": f"question {g}",
        })
    return out


__all__ = ["write_packed", "PackedRetrievalDataset", "PackedLoader", "samples_from_flat_batch", "FORMAT_VERSION"]
