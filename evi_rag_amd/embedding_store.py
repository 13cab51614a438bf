"""Host-side mirror of `GlobalEmbeddingStore` (src/data/components/embedding_store.py:12-158) with
the tables resident in HBM.

The reference keeps `entity_embeddings.pt` / `relation_embeddings.pt` on the CPU, gathers rows with
a CPU `index_select` into a grow-only pinned buffer and copies them to the GPU every batch
(`:101-150`, called from `src/data/components/loader.py:60-66,171-185`) — at D = 1024 that is about
0.5 GB per batch over PCIe.  Here the tables are loaded once into device memory (a WebQSP-scale
entity table is a few GB of 288) and `get_*_embeddings` is a device gather (`evi_gather_rows`).
Same method names, arguments and return shapes; results are always on the table's device.
"""
from __future__ import annotations

import logging
from pathlib import Path
from types import SimpleNamespace
from typing import Optional, Union

import torch

from . import _lib, ops

logger = logging.getLogger(__name__)


def gather_rows(table: torch.Tensor, ids: torch.Tensor, *, status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """table[ids] on the device (IndexError for ids outside the table, like index_select).  With a
    caller-owned `status` (int32 [1], zeroed) the range check is DEFERRED: the kernel ORs a flag into it
    and the caller reads it later (no host synchronisation here)."""
    dev = ops._require_gpu(table)
    table = ops._f32c(table, "table")
    if table.dim() != 2:
        raise ValueError(f"table must be 2D, got shape {tuple(table.shape)}")
    ids = torch.as_tensor(ids).to(device=dev, dtype=torch.int64).contiguous().view(-1)
    n, D = int(ids.numel()), int(table.size(1))
    out = torch.empty((n, D), dtype=torch.float32, device=dev)
    if n == 0 or D == 0:
        return out
    deferred = status is not None
    if not deferred:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.evi_gather_rows(ops._ptr(table), table.size(0), D, ops._ptr(ids), n, ops._ptr(out), status.data_ptr(),
                                   ops._stream(dev)))
    if not deferred and int(status.item()) != 0:
        raise IndexError(f"index out of range in gather_rows: table has {table.size(0)} rows")
    return out


class LazyEdgeEmbeddings(SimpleNamespace):
    """A batch namespace whose `edge_embeddings` ([E, D] = relation_table[edge_attr], what the reference's collater attaches) is
    gathered on first read and then kept: `getattr`, `hasattr` and plain attribute access all see it."""

    def __getattr__(self, name):  # only reached when the attribute is not set
        if name == "edge_embeddings":
            fn = self.__dict__.get("_edge_embeddings_fn")
            if fn is not None:
                value = fn()
                self.__dict__["edge_embeddings"] = value
                return value
        raise AttributeError(name)


class GlobalEmbeddingStore:
    """Read-only entity / relation embedding tables in HBM."""

    def __init__(self, embeddings_dir: Union[str, Path], vocabulary_path: Union[str, Path, None] = None,
                 device: Union[str, torch.device, None] = None) -> None:
        self.embeddings_dir = Path(embeddings_dir)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_total_entities = self._load_vocab_size(Path(vocabulary_path)) if vocabulary_path is not None else 0
        self.entity_embeddings = self._load_tensor(self.embeddings_dir / "entity_embeddings.pt")
        self.relation_embeddings = self._load_tensor(self.embeddings_dir / "relation_embeddings.pt")
        rows = self.entity_embeddings.size(0)
        if rows <= 1:
            logger.warning("Entity embedding table has <=1 row (rows=%d). Non-text fallback uses id=0, "
                           "but textual entities would be missing.", rows)

    @classmethod
    def from_tensors(cls, entity_embeddings: torch.Tensor, relation_embeddings: torch.Tensor,
                     device: Union[str, torch.device, None] = None) -> "GlobalEmbeddingStore":
        self = object.__new__(cls)
        self.embeddings_dir = None
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_total_entities = 0
        self.entity_embeddings = entity_embeddings.to(self.device, torch.float32).contiguous()
        self.relation_embeddings = relation_embeddings.to(self.device, torch.float32).contiguous()
        return self

    def clear_device_cache(self) -> None:
        """Nothing to release: there are no transient pinned buffers (the tables live in HBM)."""

    @staticmethod
    def _load_vocab_size(path: Path) -> int:
        if not path.exists():
            raise FileNotFoundError(f"Vocab LMDB not found at {path}")
        try:
            import pickle

            import lmdb  # optional, exactly as in the reference deployment

            with lmdb.open(str(path), readonly=True, lock=False, max_readers=1) as env:
                with env.begin() as txn:
                    data = txn.get(b"entity_to_id")
                    if data:
                        return len(pickle.loads(data))
            return 0
        except Exception as exc:  # noqa: BLE001 - the reference logs and continues (:65-67)
            logger.warning(f"Failed to read vocab size: {exc}")
            return 0

    def _load_tensor(self, path: Path) -> torch.Tensor:
        if not path.exists():
            raise FileNotFoundError(f"Embedding file missing: {path}")
        logger.info(f"Loading {path}...")
        t = torch.load(path, map_location="cpu", weights_only=True)
        return t.to(self.device, torch.float32).contiguous()

    def get_entity_embeddings(self, entity_ids: torch.Tensor, *, device: Optional[torch.device] = None,
                              status: Optional[torch.Tensor] = None) -> torch.Tensor:
        if entity_ids.numel() == 0:
            return torch.empty((0, int(self.entity_embeddings.size(1))), dtype=self.entity_embeddings.dtype, device=self.device)
        return gather_rows(self.entity_embeddings, entity_ids, status=status)

    def get_relation_embeddings(self, relation_ids: torch.Tensor, *, device: Optional[torch.device] = None,
                                status: Optional[torch.Tensor] = None) -> torch.Tensor:
        if relation_ids.numel() == 0:
            return torch.empty((0, int(self.relation_embeddings.size(1))), dtype=self.relation_embeddings.dtype, device=self.device)
        return gather_rows(self.relation_embeddings, relation_ids, status=status)

    def attach(self, batch, *, check: bool = True) -> None:
        """`RetrievalCollater._attach_embeddings` (src/data/components/loader.py:60-66), on the device.
        check=False defers the id range check to `raise_if_failed()` (one read per epoch, not per batch)."""
        status = None
        if not check:
            if getattr(self, "_deferred_status", None) is None:
                self._deferred_status = torch.zeros(1, dtype=torch.int32, device=self.device)
            status = self._deferred_status
        batch.node_embeddings = self.get_entity_embeddings(torch.as_tensor(batch.node_embedding_ids), status=status)
        # the relation table itself rides along: a consumer that projects each relation once (Retriever with relation
        # de-duplication) reads it instead of the [E, D] per-edge gather ...
        batch.relation_embedding_table = self.relation_embeddings
        batch.num_relations = int(self.relation_embeddings.size(0))
        if isinstance(batch, LazyEdgeEmbeddings):
            # ... which a batch of that type performs only when `edge_embeddings` is actually read (400 MB per WebQSP batch of 32)
            edge_attr = torch.as_tensor(batch.edge_attr)
            batch.__dict__["_edge_embeddings_fn"] = lambda: self.get_relation_embeddings(edge_attr, status=status)
        else:
            batch.edge_embeddings = self.get_relation_embeddings(torch.as_tensor(batch.edge_attr), status=status)

    def raise_if_failed(self) -> None:
        """Reports an out-of-range id seen by any deferred gather since the last call."""
        st = getattr(self, "_deferred_status", None)
        if st is not None and int(st.item()) != 0:
            st.zero_()
            raise IndexError("index out of range in gather_rows: an embedding id exceeds its table")

    @property
    def entity_dim(self) -> int:
        return self.entity_embeddings.size(-1)

    @property
    def relation_dim(self) -> int:
        return self.relation_embeddings.size(-1)


__all__ = ["GlobalEmbeddingStore", "gather_rows"]
