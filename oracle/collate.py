"""Batch collation oracle (test infrastructure only).

Restates what the reference's loader produces for a list of samples: torch_geometric's `Collater`
concatenates every tensor attribute along dim 0 (`edge_index` along dim 1) after adding the
per-sample increment returned by `__inc__` — `num_nodes` for `edge_index` (PyG's default) and for
q/a_local_indices and pair_*_node_locals, `num_edges` for pair_edge_local_ids, 0 for the rest
(GRetrievalData.__inc__, src/data/g_retrieval_dataset.py:29-37) — and records the cumulative item
counts in `ptr` (nodes) and `_slice_dict[...]`.  torch_geometric is absent from this image, so the
concatenation rule itself is restated from its documented behaviour: parity unpinned at that
boundary; the increments are the reference's own code.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence

import numpy as np

NODE_INC = ("q_local_indices", "a_local_indices", "pair_start_node_locals", "pair_answer_node_locals")
EDGE_INC = ("pair_edge_local_ids",)
PLAIN = ("edge_attr", "labels", "node_global_ids", "node_embedding_ids", "topic_one_hot", "answer_entity_ids",
         "seed_entity_ids", "pair_edge_counts", "pair_shortest_lengths")


def collate(samples: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
    out: Dict[str, List[np.ndarray]] = {k: [] for k in ("edge_index",) + NODE_INC + EDGE_INC + PLAIN}
    ptr, eptr = [0], [0]
    slices: Dict[str, List[int]] = {k: [0] for k in NODE_INC + EDGE_INC + ("answer_entity_ids", "seed_entity_ids")}
    for s in samples:
        n, n0, e0 = int(s["num_nodes"]), ptr[-1], eptr[-1]
        ei = np.asarray(s["edge_index"], np.int64).reshape(2, -1)
        out["edge_index"].append(ei + n0)
        for k in NODE_INC:
            v = np.asarray(s.get(k, []), np.int64).reshape(-1)
            out[k].append(v + n0)
            slices[k].append(slices[k][-1] + v.size)
        for k in EDGE_INC:
            v = np.asarray(s.get(k, []), np.int64).reshape(-1)
            out[k].append(v + e0)
            slices[k].append(slices[k][-1] + v.size)
        for k in PLAIN:
            v = np.asarray(s.get(k, []))
            out[k].append(v)
            if k in slices:
                slices[k].append(slices[k][-1] + v.shape[0])
        ptr.append(n0 + n)
        eptr.append(e0 + ei.shape[1])
    res: Dict[str, Any] = {"edge_index": np.concatenate(out["edge_index"], axis=1)}
    for k in NODE_INC + EDGE_INC + PLAIN:
        parts = [p for p in out[k] if p.size or p.ndim > 1]
        res[k] = np.concatenate(parts) if parts else np.empty(0, np.int64)
    res["ptr"] = np.asarray(ptr, np.int64)
    res["edge_ptr"] = np.asarray(eptr, np.int64)
    res["batch"] = np.repeat(np.arange(len(samples)), np.diff(ptr))
    res["question_emb"] = np.concatenate([np.asarray(s["question_emb"], np.float32).reshape(1, -1) for s in samples])
    res["slices"] = {k: np.asarray(v, np.int64) for k, v in slices.items()}
    return res
