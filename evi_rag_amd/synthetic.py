"""Seeded synthetic WebQSP/CWQ-shaped inputs: no dataset or checkpoint ships
with the reference, so tests and bench.py build retrieval batches here.  Pure numpy, host side;
nothing in the compute path depends on this module.

A batch is the flat form of the PyG `Batch` the reference's loader produces
(src/data/components/loader.py:43-99, src/data/g_retrieval_dataset.py:29-37,113-154; sample
schema written at scripts/build_retrieval_pipeline.py:2200-2224): per-graph arrays concatenated,
node indices offset by the graph's first node, plus the CSR pointers PyG keeps in `ptr` and
`_slice_dict`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class SyntheticBatch:
    num_graphs: int
    num_nodes: int
    edge_index: np.ndarray          # [2, E] i64, global (batch) node indices
    ptr: np.ndarray                 # [B+1] i64 node pointer
    edge_ptr: np.ndarray            # [B+1] i64 edge pointer (= _slice_dict["edge_index"])
    edge_attr: np.ndarray           # [E] i64 relation ids
    labels: np.ndarray              # [E] f32 in {0, 1}
    node_global_ids: np.ndarray     # [N] i64 entity ids
    node_embedding_ids: np.ndarray  # [N] i64, 0 = non-text entity
    topic_one_hot: np.ndarray       # [N, 2] f32 (col 1 = seed)
    question_emb: np.ndarray        # [B, D] f32
    q_local_indices: np.ndarray     # [sum q] i64, batch-offset
    q_ptr: np.ndarray               # [B+1]
    a_local_indices: np.ndarray     # [sum a] i64, batch-offset
    a_ptr: np.ndarray               # [B+1]
    answer_entity_ids: np.ndarray   # [sum a] i64 global entity ids
    answer_ptr: np.ndarray          # [B+1]
    node_embeddings: Optional[np.ndarray] = None   # [N, D] f32 (gathered entity rows)
    edge_embeddings: Optional[np.ndarray] = None   # [E, D] f32 (gathered relation rows)
    sample_id: List[str] = field(default_factory=list)

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def slice_dict(self) -> Dict[str, np.ndarray]:
        return {"edge_index": self.edge_ptr, "q_local_indices": self.q_ptr, "a_local_indices": self.a_ptr,
                "answer_entity_ids": self.answer_ptr}


def _power_law_endpoints(rng: np.random.Generator, n_nodes: int, n_edges: int, alpha: float) -> np.ndarray:
    """Edge endpoints with a power-law degree profile (node weight ~ rank^-1/(alpha-1))."""
    w = np.arange(1, n_nodes + 1, dtype=np.float64) ** (-1.0 / (alpha - 1.0))
    w /= w.sum()
    perm = rng.permutation(n_nodes)
    heads = perm[rng.choice(n_nodes, size=n_edges, p=w)]
    tails = perm[rng.choice(n_nodes, size=n_edges, p=w)]
    loops = heads == tails
    tails[loops] = (tails[loops] + 1 + rng.integers(0, max(n_nodes - 1, 1), size=int(loops.sum()))) % n_nodes
    return np.stack([heads, tails]).astype(np.int64)


def make_batch(
    num_graphs: int,
    *,
    nodes_per_graph: int = 64,
    edges_per_graph: int = 31,
    emb_dim: int = 32,
    num_relations: int = 16,
    num_entities: Optional[int] = None,
    seed: int = 0,
    alpha: float = 2.1,
    size_jitter: float = 0.25,
    max_seeds: int = 2,
    max_answers: int = 5,
    attach_embeddings: bool = True,
    non_text_frac: float = 0.1,
    entity_table: Optional[np.ndarray] = None,
    relation_table: Optional[np.ndarray] = None,
) -> SyntheticBatch:
    rng = np.random.default_rng(seed)
    num_entities = num_entities or max(4 * nodes_per_graph * num_graphs, 16)
    ei, eattr, labels = [], [], []
    ptr, eptr = [0], [0]
    ngid, nemb, topic = [], [], []
    qli, qp, ali, ap, aent = [], [0], [], [0], []
    for g in range(num_graphs):
        jn = 1.0 + size_jitter * (2 * rng.random() - 1)
        je = 1.0 + size_jitter * (2 * rng.random() - 1)
        n = max(2, int(round(nodes_per_graph * jn)))
        e = max(1, int(round(edges_per_graph * je)))
        ends = _power_law_endpoints(rng, n, e, alpha)
        rel = rng.integers(0, num_relations, size=e).astype(np.int64)
        # (h, r, t) dedup in first-seen order, as build_graph does
        # (scripts/build_retrieval_pipeline.py:1450-1603): duplicate triples would score identically
        # and make the reference's torch.topk order (unspecified among ties) unpinnable.
        trip = ends[0] * (n * num_relations) + ends[1] * num_relations + rel
        _, first = np.unique(trip, return_index=True)
        first.sort()
        ends, rel = ends[:, first], rel[first]
        e = int(first.shape[0])
        gids = rng.choice(num_entities - 1, size=n, replace=False).astype(np.int64) + 1
        emb_ids = gids.copy()
        emb_ids[rng.random(n) < non_text_frac] = 0  # non-text entities share embedding row 0
        n_seed = int(rng.integers(1, max_seeds + 1))
        n_ans = int(rng.integers(1, max_answers + 1))
        seeds = np.unique(ends[0, rng.integers(0, e, size=n_seed)])  # seeds touch at least one edge
        answers = np.unique(rng.integers(0, n, size=n_ans))
        t = np.zeros((n, 2), dtype=np.float32)
        t[:, 0] = 1.0
        t[seeds, 0] = 0.0
        t[seeds, 1] = 1.0
        lab = (rng.random(e) < 0.02).astype(np.float32)
        near = np.isin(ends[0], answers) | np.isin(ends[1], answers)
        lab[near & (rng.random(e) < 0.5)] = 1.0
        off = ptr[-1]
        ei.append(ends + off)
        eattr.append(rel)
        labels.append(lab)
        ngid.append(gids)
        nemb.append(emb_ids)
        topic.append(t)
        qli.append(seeds + off)
        qp.append(qp[-1] + seeds.shape[0])
        ali.append(answers + off)
        ap.append(ap[-1] + answers.shape[0])
        aent.append(gids[answers])
        ptr.append(off + n)
        eptr.append(eptr[-1] + e)
    edge_index = np.concatenate(ei, axis=1)
    edge_attr = np.concatenate(eattr)
    node_embedding_ids = np.concatenate(nemb)
    batch = SyntheticBatch(
        num_graphs=num_graphs,
        num_nodes=int(ptr[-1]),
        edge_index=edge_index,
        ptr=np.asarray(ptr, np.int64),
        edge_ptr=np.asarray(eptr, np.int64),
        edge_attr=edge_attr,
        labels=np.concatenate(labels),
        node_global_ids=np.concatenate(ngid),
        node_embedding_ids=node_embedding_ids,
        topic_one_hot=np.concatenate(topic, axis=0),
        question_emb=rng.standard_normal((num_graphs, emb_dim), dtype=np.float32),
        q_local_indices=np.concatenate(qli).astype(np.int64),
        q_ptr=np.asarray(qp, np.int64),
        a_local_indices=np.concatenate(ali).astype(np.int64),
        a_ptr=np.asarray(ap, np.int64),
        answer_entity_ids=np.concatenate(aent).astype(np.int64),
        answer_ptr=np.asarray(ap, np.int64),
        sample_id=[f"synthetic/test/{seed}-{g}" for g in range(num_graphs)],
    )
    if attach_embeddings:
        if entity_table is None:
            entity_table = rng.standard_normal((num_entities, emb_dim), dtype=np.float32)
            entity_table[0] = 0.0  # row 0 = non-text placeholder (scripts/text_encode_utils.py:125-134)
        if relation_table is None:
            relation_table = rng.standard_normal((num_relations, emb_dim), dtype=np.float32)
        batch.node_embeddings = entity_table[node_embedding_ids]
        batch.edge_embeddings = relation_table[edge_attr]
    return batch


def as_namespace(batch: SyntheticBatch, *, device=None):
    """The batch as an attribute bag of torch tensors — the duck type the reference's
    Retriever.forward / metrics read (`getattr(batch, ...)`)."""
    import types

    import torch

    def t(a, dtype=None):
        x = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            x = x.to(dtype)
        return x.to(device) if device is not None else x

    ns = types.SimpleNamespace(
        edge_index=t(batch.edge_index), ptr=t(batch.ptr), edge_attr=t(batch.edge_attr), labels=t(batch.labels),
        node_global_ids=t(batch.node_global_ids), node_embedding_ids=t(batch.node_embedding_ids),
        topic_one_hot=t(batch.topic_one_hot), question_emb=t(batch.question_emb),
        q_local_indices=t(batch.q_local_indices), a_local_indices=t(batch.a_local_indices),
        answer_entity_ids=t(batch.answer_entity_ids), answer_entity_ids_ptr=t(batch.answer_ptr),
        num_nodes=batch.num_nodes, num_graphs=batch.num_graphs, sample_id=list(batch.sample_id),
        _slice_dict={k: t(v) for k, v in batch.slice_dict().items()},
    )
    if batch.node_embeddings is not None:
        ns.node_embeddings = t(batch.node_embeddings)
        ns.edge_embeddings = t(batch.edge_embeddings)
    return ns
