#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN FUNCTIONS (build container only).

The reference checkout (/root/reference) never travels to the GPU box; what travels are the
small input/output vectors written here.  Third-party packages the reference imports but this
image lacks (lmdb, torch_geometric, torchmetrics, lightning, hydra, ...) are replaced by inert
stubs: only `torchmetrics.Metric` (a state container) and `torch_geometric.nn.MessagePassing`
(mean aggregation, restated from PyG's documented semantics — "parity unpinned" at that one
boundary) carry behaviour.  Everything else in the fixtures is the reference's
arithmetic, unmodified.

Usage:  python tests/golden/make_golden.py            (writes next to this file)
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("EVI_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

# ---- inert stubs for absent third-party packages -----------------------------------------------
_STUB_ROOTS = ("lmdb", "torch_geometric", "torchmetrics", "lightning", "lightning_utilities", "wandb", "openai", "ollama", "tiktoken", "vllm")


class _StubModule(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (object,), {"__init__": lambda self, *a, **k: None, "__module__": self.__name__})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        return None


sys.meta_path.append(_StubFinder())


class _Metric:
    """torchmetrics.Metric as a plain state container (add_state / reset; no sync)."""

    def __init__(self, **kwargs):
        self._defaults = {}

    def add_state(self, name, default, dist_reduce_fx=None):
        self._defaults[name] = default.clone()
        setattr(self, name, default.clone())

    def reset(self):
        for k, v in self._defaults.items():
            setattr(self, k, v.clone())


class _MessagePassing(torch.nn.Module):
    """PyG MessagePassing(aggr="mean", flow="source_to_target", node_dim=0), restated:
    out[i] = mean over edges (j -> i) of message(x_j); 0 where i has no incoming edge."""

    def __init__(self, aggr="mean", node_dim=0, **kwargs):
        super().__init__()
        assert aggr == "mean" and node_dim == 0

    def propagate(self, edge_index, x):
        src, dst = edge_index[0], edge_index[1]
        msg = self.message(x.index_select(0, src))
        out = torch.zeros((x.size(0),) + tuple(msg.shape[1:]), dtype=msg.dtype)
        out.index_add_(0, dst, msg)
        cnt = torch.zeros(x.size(0), dtype=msg.dtype)
        cnt.index_add_(0, dst, torch.ones(dst.numel(), dtype=msg.dtype))
        return out / cnt.clamp(min=1).view(-1, *([1] * (msg.dim() - 1)))


importlib.import_module("torchmetrics").Metric = _Metric
importlib.import_module("torch_geometric.nn").MessagePassing = _MessagePassing

from evi_rag_amd import synthetic  # noqa: E402  (the build's own input generator)

import scripts.build_retrieval_pipeline as brp  # noqa: E402
import scripts.text_encode_utils as teu  # noqa: E402
from src.data.components.g_agent_builder import GAgentBuilder  # noqa: E402
from src.metrics.reachability import AnswerReachability  # noqa: E402
from src.metrics.retriever_metrics import (  # noqa: E402
    BridgeEdgeRecallAtK, BridgePositiveCoverage, BridgeProbQuality, EdgeRecallAtK, ScoreMargin)
from src.models.components.retriever import Retriever  # noqa: E402
from src.models.reasoner_module import _oracle_metrics_for_sample  # noqa: E402
from src.utils.graph_utils import compute_edge_batch, compute_qa_edge_mask  # noqa: E402
from src.utils.metrics import compute_answer_hit, compute_answer_recall, compute_ranking_metrics  # noqa: E402

K_VALUES = [1, 10, 25, 50, 100, 200, 300, 400, 500]  # configs/window/default.yaml:8


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {os.path.relpath(path, REPO)} ({os.path.getsize(path)} B)")


# ---- T4b: compute_ranking_metrics (src/utils/metrics.py:112-170) ------------------------------------------------
def gen_ranking_metrics():
    """Precision / recall / F1 / nDCG @k and MRR over a list of {"scores", "labels"} samples, computed by the reference.
    Scores are distinct inside a sample (torch.argsort's tie order is not specified); the cases: ordinary binary labels,
    a sample without positives (skipped), one with only positives, samples shorter than the largest k, a single-element
    sample, graded (non-binary) labels — the reference sums label VALUES in the hits and DCG and truncates their sum to an
    int for the recall denominator."""
    rng = np.random.default_rng(77)
    lens = [40, 7, 1, 120, 15, 64, 3, 9]
    scores, labels = [], []
    for i, n in enumerate(lens):
        s = rng.permutation(n).astype(np.float32) * 0.37 - 5.0 + rng.random(1).astype(np.float32)  # distinct
        lab = (rng.random(n) < 0.2).astype(np.float32)
        if i == 1:
            lab[:] = 0.0  # no positives: the sample is skipped
        if i == 2:
            lab[:] = 1.0
        if i == 4:
            lab[:] = 1.0  # every item positive
        if i == 6:
            lab = np.array([0.0, 1.0, 0.0], np.float32)
        if i == 7:
            lab = np.array([0.0, 2.0, 0.5, 0.0, 1.0, 0.25, 0.0, 0.0, 3.0], np.float32)  # graded relevance
        scores.append(s)
        labels.append(lab)
    ks = [1, 3, 5, 10, 50, 100]
    samples = [{"scores": torch.from_numpy(s), "labels": torch.from_numpy(l)} for s, l in zip(scores, labels)]
    st = compute_ranking_metrics(samples, ks)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    save("ranking_metrics", scores=np.concatenate(scores), labels=np.concatenate(labels), ptr=ptr, k_values=np.asarray(ks),
         precision=[st.precision_at_k[k] for k in ks], recall=[st.recall_at_k[k] for k in ks], f1=[st.f1_at_k[k] for k in ks],
         ndcg=[st.ndcg_at_k[k] for k in ks], mrr=st.mrr,
         # the default k list ([1]) when none is given, and an all-negative sample list (every mean 0, MRR 0)
         default_k_precision=compute_ranking_metrics(samples, None).precision_at_k[1],
         empty_mrr=compute_ranking_metrics([samples[1]], ks).mrr,
         empty_recall=[compute_ranking_metrics([samples[1]], ks).recall_at_k[k] for k in ks])


# ---- C1-C4 ---------------------------------------------------------------------------------------
def gen_cosine():
    rng = np.random.default_rng(101)
    x = rng.standard_normal((37, 48)).astype(np.float32)
    x[0] = 0.0
    x[5] *= 1e-9  # norm below eps: the clamp decides
    out = brp._normalize_embeddings(torch.from_numpy(x), 1e-6).numpy()
    # canonical edges: parallel edges between the same unordered pair, some with equal relations
    R, D = 12, 48
    rel = rng.standard_normal((R, D)).astype(np.float32)
    rel[3] = rel[7]  # exact tie between two relations
    q = rng.standard_normal(D).astype(np.float32)
    edge_src = [0, 1, 0, 2, 2, 3, 1, 0, 4, 4, 4, 2]
    edge_dst = [1, 0, 1, 3, 3, 2, 0, 1, 5, 5, 5, 3]
    edge_rel = [3, 7, 1, 2, 9, 4, 7, 3, 5, 5, 6, 9]
    pos = [True, True, True, True, True, False, True, True, True, True, True, True]
    reln = brp._normalize_embeddings(torch.from_numpy(rel), 1e-6)
    qn = brp._normalize_embeddings(torch.from_numpy(q).view(1, -1), 1e-6).view(-1)
    groups = brp._group_positive_edges_by_pair(edge_src, edge_dst, pos)
    keep = brp._select_canonical_edge_indices(groups, edge_rel, reln, qn)
    pair_ids = [0, 1, 2, 6, 7, 3, 4, 11, 8, 9, 10]
    pair_counts = [5, 3, 3]
    keep_mask = [False] * len(edge_src)
    for i in keep:
        keep_mask[i] = True
    new_ids, new_counts = brp._filter_pair_edges(pair_ids, pair_counts, keep_mask)
    save("cosine", x=x, eps=np.float32(1e-6), x_norm=out, rel=rel, q=q, rel_norm=reln.numpy(), q_norm=qn.numpy(),
         edge_src=edge_src, edge_dst=edge_dst, edge_rel=edge_rel, positive=pos,
         group_keys=np.asarray(list(groups.keys()), np.int64),
         group_sizes=np.asarray([len(v) for v in groups.values()], np.int64),
         group_members=np.concatenate([np.asarray(v, np.int64) for v in groups.values()]),
         keep_indices=np.asarray(keep, np.int64), pair_ids=pair_ids, pair_counts=pair_counts,
         new_pair_ids=new_ids, new_pair_counts=new_counts)


# ---- G1-G5 ---------------------------------------------------------------------------------------
def gen_bfs():
    rng = np.random.default_rng(202)
    cases = {}
    for c, (n, e) in enumerate([(12, 14), (40, 55), (64, 31), (150, 400), (9, 0)]):
        src = rng.integers(0, n, size=e).tolist()
        dst = rng.integers(0, n, size=e).tolist()
        if e > 4:
            src[1], dst[1] = 3, 3          # self loop
            src[2], dst[2] = -1, 2         # out-of-range edges are skipped
            src[3], dst[3] = 1, n + 5
        seeds = sorted(set(rng.integers(0, n, size=2).tolist())) + [n + 3]  # one invalid seed
        answers = rng.integers(0, n, size=4).tolist()
        adj = brp._build_undirected_adjacency(n, src, dst)
        dadj = brp._build_directed_adjacency(n, src, dst)
        dist = brp._bfs_dist(n, adj, seeds)
        ddist = brp._bfs_dist(n, dadj, seeds)
        mask, ps, pa, pe, pc, pl = brp._shortest_path_union_mask_by_pair(n, src, dst, seeds, answers)
        dmask, dps, dpa, dpe, dpc, dpl = brp._shortest_path_union_mask_by_pair_directed(n, src, dst, seeds, answers)
        sp_edges, sp_nodes = brp._shortest_path_single(n, src, dst, seeds, answers)
        cases.update({
            f"c{c}_n": n, f"c{c}_src": np.asarray(src, np.int64), f"c{c}_dst": np.asarray(dst, np.int64),
            f"c{c}_seeds": np.asarray(seeds, np.int64), f"c{c}_answers": np.asarray(answers, np.int64),
            f"c{c}_adj_ptr": np.cumsum([0] + [len(a) for a in adj]).astype(np.int64),
            f"c{c}_adj": np.asarray([v for a in adj for v in a], np.int64),
            f"c{c}_dadj_ptr": np.cumsum([0] + [len(a) for a in dadj]).astype(np.int64),
            f"c{c}_dadj": np.asarray([v for a in dadj for v in a], np.int64),
            f"c{c}_dist": np.asarray(dist, np.int64), f"c{c}_ddist": np.asarray(ddist, np.int64),
            f"c{c}_mask": np.asarray(mask, bool), f"c{c}_pair_start": np.asarray(ps, np.int64),
            f"c{c}_pair_answer": np.asarray(pa, np.int64), f"c{c}_pair_edges": np.asarray(pe, np.int64),
            f"c{c}_pair_counts": np.asarray(pc, np.int64), f"c{c}_pair_len": np.asarray(pl, np.int64),
            f"c{c}_dmask": np.asarray(dmask, bool), f"c{c}_dpair_start": np.asarray(dps, np.int64),
            f"c{c}_dpair_answer": np.asarray(dpa, np.int64), f"c{c}_dpair_edges": np.asarray(dpe, np.int64),
            f"c{c}_dpair_counts": np.asarray(dpc, np.int64), f"c{c}_dpair_len": np.asarray(dpl, np.int64),
            f"c{c}_sp_edges": np.asarray(sp_edges, np.int64), f"c{c}_sp_nodes": np.asarray(sp_nodes, np.int64),
        })
    # has_connectivity on string triples
    triples = [("a", "r", "b"), ("b", "r", "c"), ("d", "r", "e"), ("c", "r2", "a")]
    conn = [
        brp.has_connectivity(triples, ["a"], ["c"]), brp.has_connectivity(triples, ["a"], ["e"]),
        brp.has_connectivity(triples, ["c"], ["b"], path_mode="qa_directed"),
        brp.has_connectivity(triples, ["e"], ["d"], path_mode="qa_directed"),
        brp.has_connectivity(triples, ["zz"], ["a"]), brp.has_connectivity([], ["a"], ["b"]),
    ]
    cases["num_cases"] = 5
    cases["has_connectivity"] = np.asarray(conn, bool)
    save("bfs", **cases)


# ---- G11 -----------------------------------------------------------------------------------------
def gen_graph_utils(batch):
    ns = synthetic.as_namespace(batch)
    eb, eptr = compute_edge_batch(ns.edge_index, node_ptr=ns.ptr, num_graphs=batch.num_graphs,
                                  device=torch.device("cpu"))
    near = compute_qa_edge_mask(ns.edge_index, num_nodes=batch.num_nodes,
                                q_local_indices=ns.q_local_indices, a_local_indices=ns.a_local_indices)
    return eb.numpy(), eptr.numpy(), near.numpy()


# ---- S1-S7 + G6/G7 -----------------------------------------------------------------------------------
def batch_arrays(batch, prefix=""):
    out = {}
    for name in ("edge_index", "ptr", "edge_ptr", "edge_attr", "labels", "node_global_ids", "node_embedding_ids",
                 "topic_one_hot", "question_emb", "q_local_indices", "q_ptr", "a_local_indices", "a_ptr",
                 "answer_entity_ids", "answer_ptr", "node_embeddings", "edge_embeddings"):
        out[prefix + name] = getattr(batch, name)
    return out


def gen_retriever(name, D, H, batch, seed, rounds=(2, 2), direction="bidirectional"):
    torch.manual_seed(seed)
    model = Retriever(emb_dim=D, hidden_dim=H, topic_pe=True, num_topics=2,
                      dde_cfg={"num_rounds": rounds[0], "num_reverse_rounds": rounds[1]}, dropout_p=0.1,
                      direction_mode=direction,
                      hide_seek_cfg={"enabled": True, "p_near": 0.7, "p_far": 0.1, "bias_near": -2.0,
                                     "bias_far": -0.5, "apply_in_eval": False})
    model.eval()
    # random-init biases/LayerNorm affine so that every parameter matters
    with torch.no_grad():
        for p_name, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    ns = synthetic.as_namespace(batch)
    with torch.no_grad():
        out = model(ns)
        node_struct = model._build_node_structure_features(ns, edge_index=ns.edge_index, num_nodes=batch.num_nodes)
        feats = model.extract_edge_tokens(ns)
    # §8f-4 pin: the reference's autograd through the same (eval-mode) graph — d(sum_e g_e logit_e)/d(parameter) for a fixed g
    gen = torch.Generator().manual_seed(1000 + seed)
    gvec = torch.randn(out.logits.numel(), generator=gen)
    model.zero_grad()
    (model(ns).logits * gvec).sum().backward()
    arrays = batch_arrays(batch, "b_")
    arrays["gvec"] = gvec.numpy()
    for p_name, p in model.named_parameters():
        arrays["g_" + p_name] = p.grad.detach().numpy().copy()
    sd = model.state_dict()
    for k, v in sd.items():
        arrays["w_" + k] = v.numpy()
    arrays.update(
        D=D, H=H, rounds=np.asarray(rounds), direction=direction,
        logits=out.logits.numpy(), logits_fwd=out.logits_fwd.numpy() if out.logits_fwd is not None else np.zeros(0),
        logits_bwd=out.logits_bwd.numpy() if out.logits_bwd is not None else np.zeros(0),
        edge_embeddings=out.edge_embeddings.numpy(), query_ids=out.query_ids.numpy(),
        relation_ids=out.relation_ids.numpy(), node_struct=node_struct.numpy(), edge_tokens=feats.numpy(),
        state_dict_keys=np.asarray(list(sd.keys())),
    )
    save(name, **arrays)
    return out


# ---- T1-T5 -----------------------------------------------------------------------------------------
def gen_metrics(batch, logits, tag):
    ns = synthetic.as_namespace(batch)
    scores = logits.detach().view(-1)
    target = ns.labels > 0.5
    eb, _ = compute_edge_batch(ns.edge_index, node_ptr=ns.ptr, num_graphs=batch.num_graphs, device=torch.device("cpu"))
    kw = dict(preds=scores, target=target, indexes=eb, batch=ns, num_graphs=batch.num_graphs)
    out = {}
    m = EdgeRecallAtK(k_values=K_VALUES)
    m.update(**kw)
    out.update({k: float(v) for k, v in m.compute().items()})
    m = AnswerReachability(k_values=K_VALUES)
    m.update(preds=scores, batch=ns, query_ids=eb, num_graphs=batch.num_graphs)
    out.update({k: float(v) for k, v in m.compute().items()})
    reach_total = float(m.total)
    for cls in (BridgeEdgeRecallAtK,):
        m = cls(k_values=K_VALUES)
        m.update(**kw)
        out.update({k: float(v) for k, v in m.compute().items()})
    for cls in (BridgePositiveCoverage, ScoreMargin, BridgeProbQuality):
        m = cls()
        m.update(**kw)
        out.update({k: float(v) for k, v in m.compute().items()})
    # T4: answer hit / recall over ranked edge lists
    samples, oracle_rows = [], []
    for g in range(batch.num_graphs):
        lo, hi = int(batch.edge_ptr[g]), int(batch.edge_ptr[g + 1])
        heads = ns.node_global_ids[ns.edge_index[0, lo:hi]]
        tails = ns.node_global_ids[ns.edge_index[1, lo:hi]]
        ans = ns.answer_entity_ids[int(batch.answer_ptr[g]): int(batch.answer_ptr[g + 1])]
        samples.append({"scores": scores[lo:hi], "head_ids": heads, "tail_ids": tails, "answer_ids": ans})
        order = torch.argsort(scores[lo:hi], descending=True, stable=True)[:500]
        row = _oracle_metrics_for_sample(head_entity_ids=heads[order], tail_entity_ids=tails[order],
                                         answer_entity_ids=ans, k_values=K_VALUES)
        oracle_rows.append([row[f"answer_hit@{k}"] for k in K_VALUES] + [row[f"answer_recall@{k}"] for k in K_VALUES])
    out.update(compute_answer_hit(samples, K_VALUES))
    out.update(compute_answer_recall(samples, K_VALUES))
    keys = sorted(out.keys())
    save(f"metrics_{tag}", keys=np.asarray(keys), values=np.asarray([out[k] for k in keys], np.float64),
         reach_total=reach_total, scores=scores.numpy(), oracle_rows=np.asarray(oracle_rows, np.float64),
         k_values=np.asarray(K_VALUES), **batch_arrays(batch, "b_"))


# ---- G8/G9 -------------------------------------------------------------------------------------------
def gen_g_agent(batch, logits):
    ns = synthetic.as_namespace(batch)
    scores = logits.detach().view(-1)
    arrays = {"num_graphs": batch.num_graphs}
    for g in range(batch.num_graphs):
        lo, hi = int(batch.edge_ptr[g]), int(batch.edge_ptr[g + 1])
        n0, n1 = int(batch.ptr[g]), int(batch.ptr[g + 1])
        heads = ns.edge_index[0, lo:hi] - n0
        tails = ns.edge_index[1, lo:hi] - n0
        s = scores[lo:hi]
        seeds = ns.q_local_indices[int(batch.q_ptr[g]): int(batch.q_ptr[g + 1])] - n0
        logit = GAgentBuilder._node_softmax_logit(edge_scores=s, edge_head_locals=heads, edge_tail_locals=tails,
                                                  num_nodes=n1 - n0)
        for tk in (5, 500):
            arrays[f"g{g}_topk{tk}"] = GAgentBuilder._select_topk_edges(edge_scores=logit, edge_top_k=tk).numpy()
        for ri, (ratio, mn, mx) in enumerate([(0.25, 1, None), (0.5, 2, 3), (0.0, 0, None), (1.0, 1, 0)]):
            arrays[f"g{g}_start{ri}"] = GAgentBuilder._select_start_edges(
                heads=heads, tails=tails, edge_scores=logit, start_node_locals=seeds, num_nodes=n1 - n0,
                start_keep_ratio=ratio, start_min_edges=mn, start_max_edges=mx).numpy()
        arrays[f"g{g}_logit"] = logit.numpy()
    arrays["start_params"] = np.asarray([[0.25, 1, -1], [0.5, 2, 3], [0.0, 0, -1], [1.0, 1, 0]], np.float64)
    save("g_agent_select", scores=scores.numpy(), **arrays, **batch_arrays(batch, "b_"))


# ---- G5: build_graph ----------------------------------------------------------------------------------
def gen_build_graph():
    """Runs the reference's build_graph on string samples; stores the id-coded inputs and the record."""
    rng = np.random.default_rng(404)
    n_ent, n_rel = 60, 7
    ents = [f"m.{i:03d}" for i in range(n_ent)]
    rels = [f"rel.{i}" for i in range(n_rel)]
    struct = {e: 1000 + 3 * i for i, e in enumerate(ents)}            # structural entity ids (arbitrary, unique)
    emb = {e: (i + 1 if i % 3 else 0) for i, e in enumerate(ents)}    # 0 = non-text entity
    ent_lookup = brp.EntityLookup(entity_to_struct=struct, text_kg_id_to_embed_id={e: v for e, v in emb.items() if v})
    rel_lookup = brp.RelationLookup(rel_to_id={r: 5 + i for i, r in enumerate(rels)})
    arrays = {"ent_struct": np.asarray([struct[e] for e in ents], np.int64),
              "ent_emb": np.asarray([emb[e] for e in ents], np.int64)}
    cases = []
    for c, (n_trip, n_pool, mode, dedup, noloop, sub) in enumerate([
            (40, 14, "undirected", True, True, "none"), (80, 20, "undirected", True, True, "some"),
            (80, 20, "qa_directed", True, True, "some"), (30, 10, "undirected", False, False, "some"),
            (50, 25, "undirected", True, False, "absent"), (0, 5, "undirected", True, True, "none"),
            (120, 30, "qa_directed", False, True, "none"), (60, 12, "undirected", True, True, "nopath")]):
        pool = rng.choice(n_ent, size=n_pool, replace=False)
        h = pool[rng.integers(0, n_pool, size=n_trip)]
        t = pool[rng.integers(0, n_pool, size=n_trip)]
        r = rng.integers(0, n_rel, size=n_trip)
        if n_trip > 10:
            t[3] = h[3]                                   # self loop
            h[7], r[7], t[7] = h[2], r[2], t[2]           # duplicate triple
            h[9], r[9], t[9] = h[2], r[2], t[2]
        graph = [(ents[a], rels[b], ents[d]) for a, b, d in zip(h, r, t)]
        q_ent = [ents[i] for i in rng.choice(pool, size=min(2, n_pool), replace=False)] + [ents[(int(pool[0]) + 1) % n_ent]]
        a_ent = [ents[i] for i in rng.choice(pool, size=min(3, n_pool), replace=False)]
        if sub == "some" and n_trip:
            pick = rng.choice(n_trip, size=n_trip // 2, replace=False)
            asub = [graph[i] for i in pick] + [(ents[0], rels[0], ents[1])]
        elif sub == "absent":
            asub = [(ents[0], rels[0], ents[0])]
        elif sub == "nopath" and n_trip:
            asub = [graph[0]]
        else:
            asub = []
        sample = brp.Sample(dataset="syn", split="train", question_id=f"q{c}", kb="fb", question="?", graph=graph,
                            q_entity=q_ent, a_entity=a_ent, answer_texts=[], answer_subgraph=asub)
        rec = brp.build_graph(sample, ent_lookup, rel_lookup, f"syn/train/q{c}", path_mode=mode, dedup_edges=dedup,
                              validate_graph_edges=True, remove_self_loops=noloop)
        e2i = {e: i for i, e in enumerate(ents)}
        arrays.update({
            f"c{c}_triples": np.asarray([[e2i[a], rel_lookup.rel_to_id[b], e2i[d]] for a, b, d in graph], np.int64).reshape(-1, 3),
            f"c{c}_q": np.asarray([e2i[e] for e in q_ent], np.int64), f"c{c}_a": np.asarray([e2i[e] for e in a_ent], np.int64),
            f"c{c}_asub": np.asarray([[e2i[a], rel_lookup.rel_to_id[b], e2i[d]] for a, b, d in asub], np.int64).reshape(-1, 3),
            f"c{c}_directed": mode == "qa_directed", f"c{c}_dedup": dedup, f"c{c}_noloop": noloop,
            f"c{c}_node_entity_ids": np.asarray(rec.node_entity_ids, np.int64),
            f"c{c}_node_embedding_ids": np.asarray(rec.node_embedding_ids, np.int64),
            f"c{c}_edge_src": np.asarray(rec.edge_src, np.int64), f"c{c}_edge_dst": np.asarray(rec.edge_dst, np.int64),
            f"c{c}_edge_rel": np.asarray(rec.edge_relation_ids, np.int64),
            f"c{c}_positive": np.asarray(rec.positive_triple_mask, bool),
            f"c{c}_pair_start": np.asarray(rec.pair_start_node_locals, np.int64),
            f"c{c}_pair_answer": np.asarray(rec.pair_answer_node_locals, np.int64),
            f"c{c}_pair_edges": np.asarray(rec.pair_edge_local_ids, np.int64),
            f"c{c}_pair_counts": np.asarray(rec.pair_edge_counts, np.int64),
            f"c{c}_pair_len": np.asarray(rec.pair_shortest_lengths, np.int64),
        })
        cases.append(c)
    # triples are stored with ENTITY INDEX (0..n_ent) in columns 0/2; ent_struct / ent_emb map it to ids
    save("build_graph", num_cases=len(cases), **arrays)


# ---- f2: GAgentBuilder.process_batch / _build_and_add_sample -------------------------------------------
class _FakeStore:
    def __init__(self, samples):
        self.samples = samples

    def load_sample(self, sample_id):
        return self.samples[sample_id]


def gen_g_agent_build():
    from src.data.components.g_agent_builder import GAgentSettings

    rng = np.random.default_rng(505)
    base = synthetic.make_batch(5, nodes_per_graph=50, edges_per_graph=140, emb_dim=8, num_relations=6, seed=11)
    # re-assemble the batch with a few duplicated (h, r, t) edges per graph (different scores / labels)
    ei, ea, lab, eptr = [], [], [], [0]
    for g in range(base.num_graphs):
        lo, hi = int(base.edge_ptr[g]), int(base.edge_ptr[g + 1])
        idx = np.arange(lo, hi)
        dup = rng.choice(idx, size=12, replace=False)
        order = rng.permutation(np.concatenate([idx, dup]))
        ei.append(base.edge_index[:, order])
        ea.append(base.edge_attr[order])
        lab.append((rng.random(order.size) < 0.2).astype(np.float32))
        eptr.append(eptr[-1] + order.size)
    edge_index = np.concatenate(ei, axis=1)
    edge_attr, labels, edge_ptr = np.concatenate(ea), np.concatenate(lab), np.asarray(eptr, np.int64)
    E = edge_index.shape[1]
    logits = rng.standard_normal(E).astype(np.float32)
    query_ids = np.repeat(np.arange(base.num_graphs), np.diff(edge_ptr))
    sample_ids = [f"s{g}" for g in range(base.num_graphs)]
    store = {}
    seeds_all, answers_all = [], []
    for g in range(base.num_graphs):
        q = base.q_local_indices[int(base.q_ptr[g]): int(base.q_ptr[g + 1])]
        seeds = base.node_global_ids[q].tolist() + [987654321]                       # one seed outside the graph
        ans = base.answer_entity_ids[int(base.answer_ptr[g]): int(base.answer_ptr[g + 1])].tolist()
        ans = ans + ans[:1] + [123456789]                                            # duplicate + unknown answer
        if g == 3:
            ans = [123456789]                                                        # no answer inside the graph
        store[sample_ids[g]] = {"question_emb": base.question_emb[g].tolist(), "question": f"question {g}",
                                "seed_entity_ids": seeds, "answer_entity_ids": ans}
        seeds_all.append(np.asarray(seeds, np.int64))
        answers_all.append(np.asarray(ans, np.int64))
    batch = types.SimpleNamespace(
        ptr=torch.from_numpy(base.ptr), edge_index=torch.from_numpy(edge_index), edge_attr=torch.from_numpy(edge_attr),
        labels=torch.from_numpy(labels), node_global_ids=torch.from_numpy(base.node_global_ids),
        node_embedding_ids=torch.from_numpy(base.node_embedding_ids), sample_id=sample_ids)
    out = types.SimpleNamespace(logits=torch.from_numpy(logits), query_ids=torch.from_numpy(query_ids))
    arrays = {"ptr": base.ptr, "edge_index": edge_index, "edge_attr": edge_attr, "labels": labels, "edge_ptr": edge_ptr,
              "node_global_ids": base.node_global_ids, "node_embedding_ids": base.node_embedding_ids, "logits": logits,
              "query_ids": query_ids, "question_emb": base.question_emb, "num_graphs": base.num_graphs,
              "seed_ptr": np.cumsum([0] + [len(s) for s in seeds_all]), "seeds": np.concatenate(seeds_all),
              "ans_ptr": np.cumsum([0] + [len(a) for a in answers_all]), "answers": np.concatenate(answers_all)}
    configs = [dict(edge_top_k=20, start_keep_ratio=0.25, start_min_edges=1, allow_empty_answer=False),
               dict(edge_top_k=500, start_keep_ratio=0.5, start_min_edges=2, start_max_edges=4, allow_empty_answer=True,
                    score_mode="logits", score_temperature=2.0, score_bias=0.5)]
    for ci, cfg in enumerate(configs):
        b = GAgentBuilder(GAgentSettings(**cfg), embedding_store=_FakeStore(store))
        b.process_batch(batch, out)
        arrays[f"cfg{ci}_num_samples"] = b.stats["num_samples"]
        arrays[f"cfg{ci}_retrieval_failed"] = b.stats["retrieval_failed"]
        arrays[f"cfg{ci}_edge_counts"] = np.asarray(b.stats["edge_counts"], np.int64)
        arrays[f"cfg{ci}_sample_ids"] = np.asarray([smp.sample_id for smp in b.samples])
        for si, smp in enumerate(b.samples):
            for name in ("question_emb", "edge_relations", "edge_scores", "edge_labels", "edge_head_locals", "edge_tail_locals",
                         "node_entity_ids", "node_embedding_ids", "start_entity_ids", "answer_entity_ids", "start_node_locals",
                         "answer_node_locals"):
                arrays[f"cfg{ci}_s{si}_{name}"] = getattr(smp, name).numpy()
            arrays[f"cfg{ci}_s{si}_flags"] = np.asarray([smp.gt_path_exists, smp.is_answer_reachable, smp.is_dummy_agent], bool)
    save("g_agent_build", **arrays)


# ---- loss on the eval path (S7: RetrieverModule._shared_eval_step logs {split}/loss) ----------------------
def gen_loss():
    from src.losses.retriever_loss import RetrieverLoss

    rng = np.random.default_rng(606)
    B = 7
    counts = [40, 1, 25, 60, 3, 0, 30]                     # one single-edge graph, one empty graph
    eb = np.repeat(np.arange(B), counts)
    E = eb.size
    logits = (rng.standard_normal(E) * 3).astype(np.float32)
    targets = (rng.random(E) < 0.15).astype(np.float32)
    targets[eb == 2] = 0.0                                  # a graph without positives
    targets[eb == 4] = 1.0                                  # a graph without negatives
    near = rng.random(E) < 0.4
    arrays = {"logits": logits, "targets": targets, "edge_batch": eb, "edge_is_near": near, "num_graphs": B}
    cfgs = [dict(), dict(infonce_temperature=0.5, bce_weight=0.3), dict(edge_weight_near=2.0, edge_weight_bridge=0.5, bce_weight=1.0),
            dict(infonce_weight=0.0, bce_weight=1.0)]
    arrays["cfgs"] = np.asarray([[c.get("infonce_temperature", 1.0), c.get("infonce_weight", 1.0), c.get("bce_weight", 0.0),
                                  c.get("edge_weight_near", 1.0), c.get("edge_weight_bridge", 1.0)] for c in cfgs], np.float64)
    variants = {"base": targets, "nopos": np.zeros_like(targets)}
    for vi, (vname, tg) in enumerate(variants.items()):
        for ci, cfg in enumerate(cfgs):
            loss_fn = RetrieverLoss(**cfg)
            lg = torch.from_numpy(logits).clone().requires_grad_(True)
            out = loss_fn(types.SimpleNamespace(logits=lg), torch.from_numpy(tg), edge_batch=torch.from_numpy(eb), num_graphs=B,
                          edge_is_near=torch.from_numpy(near))
            grad = torch.autograd.grad(out.loss, lg, allow_unused=True)[0] if out.loss.requires_grad else None
            tag = f"{vname}_c{ci}"
            arrays[f"{tag}_loss"] = np.float64(out.loss.item())
            arrays[f"{tag}_grad"] = grad.numpy() if grad is not None else np.zeros(E, np.float32)
            arrays[f"{tag}_component_keys"] = np.asarray(sorted(out.components))
            arrays[f"{tag}_component_vals"] = np.asarray([out.components[k] for k in sorted(out.components)], np.float64)
            arrays[f"{tag}_metric_keys"] = np.asarray(sorted(out.metrics))
            arrays[f"{tag}_metric_vals"] = np.asarray([out.metrics[k] for k in sorted(out.metrics)], np.float64)
    arrays["nopos_targets"] = variants["nopos"]
    save("loss", **arrays)


# ---- FeatureMonitor (evaluation_cfg.feature_metrics) ------------------------------------------------------
def gen_feature_monitor():
    from src.metrics.feature_monitor import FeatureMonitor

    rng = np.random.default_rng(707)
    arrays = {}
    m = FeatureMonitor()
    for b, (n, h) in enumerate([(500, 48), (37, 48), (1, 48)]):
        preds = (rng.standard_normal(n) * 2).astype(np.float32)
        target = rng.random(n) < 0.2
        feats = rng.standard_normal((n, h)).astype(np.float32)
        feats[0] = 0.0
        m.update(torch.from_numpy(preds), torch.from_numpy(target), torch.from_numpy(feats))
        arrays.update({f"b{b}_preds": preds, f"b{b}_target": target, f"b{b}_features": feats})
    out = m.compute()
    arrays["num_batches"] = 3
    arrays["keys"] = np.asarray(sorted(out))
    arrays["values"] = np.asarray([float(out[k]) for k in sorted(out)], np.float64)
    m2 = FeatureMonitor()
    m2.update(torch.from_numpy(arrays["b1_preds"]), torch.zeros(37, dtype=torch.bool))  # no positives, no features
    out2 = m2.compute()
    arrays["nopos_values"] = np.asarray([float(out2[k]) for k in sorted(out2)], np.float64)
    save("feature_monitor", **arrays)


# ---- E2/E3 -------------------------------------------------------------------------------------------
class _FakeTokenizer:
    """Whitespace tokenizer with padding=True semantics (pad id 0, mask 0 on pads)."""

    def __call__(self, texts, padding=True, truncation=True, return_tensors="pt"):
        ids = [[(sum(map(ord, w)) % 97) + 1 for w in t.split()][:8] for t in texts]
        L = max(1, max(len(i) for i in ids))
        input_ids = torch.zeros((len(ids), L), dtype=torch.long)
        mask = torch.zeros((len(ids), L), dtype=torch.long)
        for r, row in enumerate(ids):
            input_ids[r, : len(row)] = torch.tensor(row, dtype=torch.long)
            mask[r, : len(row)] = 1
        return {"input_ids": input_ids, "attention_mask": mask}


class _FakeModel(torch.nn.Module):
    """last_hidden_state = table[input_ids] + position ramp: pins pooling, not a real encoder."""

    def __init__(self, table):
        super().__init__()
        self.table = table

    def forward(self, input_ids, attention_mask):
        hid = self.table[input_ids] + 0.01 * torch.arange(input_ids.size(1), dtype=torch.float32).view(1, -1, 1)
        return types.SimpleNamespace(last_hidden_state=hid)


def gen_encode(tmpdir):
    torch.manual_seed(7)
    D = 24
    table = torch.randn(98, D)
    texts = ["alpha beta", "gamma", "delta epsilon zeta eta", "", "theta iota", "kappa lambda mu", "nu",
             "xi omicron pi rho sigma tau upsilon phi chi psi omega", "alpha", "beta beta beta"]
    arrays = {"table": table.numpy(), "D": D}
    tok = _FakeTokenizer()
    for fp16 in (False, True):
        enc = object.__new__(teu.TextEncoder)  # bypass __init__ (needs a network fetch)
        enc.tokenizer, enc.model, enc.device = tok, _FakeModel(table), "cpu"
        enc.dtype = torch.float16 if fp16 else torch.float32
        enc.progress = False
        pooled = enc.encode(texts, batch_size=4)
        arrays[f"pooled_fp16_{int(fp16)}"] = pooled.numpy()
        if not fp16:
            emb_ids = [3, 1, 7, 12, 2, 9, 40, 5, 11, 6]  # 40 > max_embedding_id: skipped
            path = os.path.join(tmpdir, "entity_embeddings.pt")
            from pathlib import Path

            tensor = teu.encode_to_memmap(enc, texts, emb_ids, batch_size=4, max_embedding_id=12,
                                          out_path=Path(path), desc=None, show_progress=False)
            arrays["memmap_table"] = tensor.numpy().copy()
            arrays["emb_ids"] = np.asarray(emb_ids, np.int64)
            arrays["max_embedding_id"] = 12
            assert torch.equal(torch.load(path), tensor)
    # per-batch token ids / masks so the build's pooling kernel sees the same inputs
    for bi, (s, e) in enumerate(teu._iter_batches(len(texts), 4)):
        t = tok(texts[s:e])
        arrays[f"ids_{bi}"] = t["input_ids"].numpy()
        arrays[f"mask_{bi}"] = t["attention_mask"].numpy()
        arrays[f"hidden_{bi}"] = _FakeModel(table)(**t).last_hidden_state.numpy()
    arrays["num_batches"] = bi + 1
    arrays["empty_shape"] = np.asarray(object.__new__(teu.TextEncoder).encode.__wrapped__(enc, [], 4).shape) \
        if hasattr(teu.TextEncoder.encode, "__wrapped__") else np.asarray(enc.encode([], 4).shape)
    save("encode", **arrays)


# ---- E4: the encode call sites of the offline pipeline ----------------------------------------------------
def gen_encode_tables(tmpdir):
    """Runs the reference's `preprocess` (scripts/build_retrieval_pipeline.py:1140-1447) end to end on a synthetic raw
    parquet split with embeddings requested — vocabulary pass, then its encode sequence (:1243-1309: entity labels sorted
    by embedding_id -> encode_to_memmap; relation labels sorted by relation_id -> encode; :1318-1334, :1360: questions per
    chunk -> question_emb list per sample) — with `TextEncoder` replaced by the lookup model of gen_encode (no weights
    offline: E1 stays unpinned).  Stores the vocabulary records it encoded from and the three tables it wrote."""
    from pathlib import Path

    import pyarrow as pa
    import pyarrow.parquet as pq

    torch.manual_seed(17)
    D = 16
    table = torch.randn(98, D)
    tok = _FakeTokenizer()

    class _Enc(teu.TextEncoder):
        def __init__(self, model_name, device, fp16, progress):
            self.tokenizer, self.model, self.device = tok, _FakeModel(table), "cpu"
            self.dtype = torch.float16 if fp16 else torch.float32
            self.progress = progress

    raw = Path(tmpdir) / "raw"
    out = Path(tmpdir) / "normalized"
    emb_dir = Path(tmpdir) / "embeddings"
    raw.mkdir()
    names = ["Barack Obama", "Honolulu", "United States of America", "m.02mjmr", "Michelle Obama", "Chicago", "g.11b6", "Hawaii",
             "Pacific Ocean", "m.0abc", "Illinois", "White House", "Harvard Law School", "lawyer", "1961"]
    rels = ["people.person.place_of_birth", "location.location.containedby", "people.person.spouse_s", "common.topic.alias",
            "people.person.profession", "education.education.institution", "location.location.adjoin_s"]
    rng = np.random.default_rng(55)
    rows = {"train": [], "test": []}
    for split, n in (("train", 5), ("test", 3)):
        for qi in range(n):
            m = int(rng.integers(4, 11))
            graph = [[names[int(rng.integers(len(names)))], rels[int(rng.integers(len(rels)))], names[int(rng.integers(len(names)))]]
                     for _ in range(m)]
            ents = sorted({t[0] for t in graph} | {t[2] for t in graph})
            rows[split].append({"id": f"{split}-{qi}", "question": f"where was {names[qi % 5]} born {qi}",
                                "answer": [ents[-1]], "q_entity": [ents[0]], "a_entity": [ents[-1]], "graph": graph})
    for split, rr in rows.items():
        pq.write_table(pa.Table.from_pylist(rr), raw / f"{split}-00000.parquet")
    keep_all = brp.SplitFilter(skip_no_topic=False, skip_no_ans=False, skip_no_path=False)
    cfg = brp.EmbeddingConfig(encoder="fake", device="cpu", batch_size=4, fp16=False, progress_bar=False, embeddings_out_dir=emb_dir,
                              precompute_entities=True, precompute_relations=True, precompute_questions=True,
                              canonicalize_relations=False, cosine_eps=1e-6)
    import re as _re

    text_cfg = brp.TextEntityConfig(mode="regex", prefixes=(), regex=_re.compile("^(?!m\\.|g\\.).*"))  # configs/dataset/webqsp.yaml:25
    orig = brp.TextEncoder
    brp.TextEncoder = _Enc
    try:
        brp.preprocess(dataset="webqsp", kb="freebase", raw_root=raw, out_dir=out,
                       column_map={"question_id_field": "id", "question_field": "question", "answer_text_field": "answer",
                                   "q_entity_field": "q_entity", "a_entity_field": "a_entity", "graph_field": "graph"},
                       entity_normalization="none", text_cfg=text_cfg, train_filter=keep_all, eval_filter=keep_all,
                       override_filters={}, embedding_cfg=cfg, parquet_chunk_size=max(brp._MIN_CHUNK_SIZE, 3))
    finally:
        brp.TextEncoder = orig
    ent = torch.load(emb_dir / "entity_embeddings.pt")
    rel = torch.load(emb_dir / "relation_embeddings.pt")
    ev = pq.read_table(out / "entity_vocab.parquet").to_pylist()
    ee = pq.read_table(out / "embedding_vocab.parquet").to_pylist() if (out / "embedding_vocab.parquet").exists() else None
    rv = pq.read_table(out / "relation_vocab.parquet").to_pylist()
    qs = pq.read_table(out / "questions.parquet").to_pylist()
    print("normalized files:", sorted(p.name for p in out.iterdir()))
    import json

    fixture = {"D": D, "lookup_table": table.numpy().tolist(), "batch_size": 4,
               "entity_vocab": ev, "embedding_vocab": ee, "relation_vocab": rv,
               "questions": [{"question_uid": q.get("question_uid"), "question": q["question"], "question_emb": q["question_emb"]} for q in qs],
               "entity_embeddings": ent.numpy().tolist(), "relation_embeddings": rel.numpy().tolist()}
    import gzip

    path = os.path.join(HERE, "encode_tables.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(fixture).encode())
    print(f"wrote {os.path.relpath(path, REPO)} ({os.path.getsize(path)} B)")


# ---- T3: eval_retriever/<split>.pt payload of the reference's writer callback -----------------------------
def gen_topk_writer(cases, tmpdir):
    """Drives the reference's RetrieverTopKEdgeWriter (src/callbacks/retriever_topk_edge_writer.py: on_test_start ->
    on_test_batch_end -> on_test_end, i.e. _build_context / _slice_graph / _select_topk_edges / _build_record) on the
    committed batches with the reference Retriever's own outputs and stores the payload it saved.  Lightning's
    BasePredictionWriter is an inert stub; everything else is the reference's code."""
    import json
    import types as _types

    # src/utils/logging_utils.py imports omegaconf at module level; stubbed only from here on (the earlier imports keep
    # seeing it as absent, which is what scripts/build_retrieval_pipeline.py's own guards expect)
    global _STUB_ROOTS
    _STUB_ROOTS = _STUB_ROOTS + ("omegaconf", "hydra", "rich")
    from src.callbacks.retriever_topk_edge_writer import RetrieverTopKEdgeWriter
    from src.models.components.retriever import RetrieverOutput

    out_cases = []
    for case in cases:
        ns = case["batch"]
        wdir = os.path.join(tmpdir, case["name"])
        w = RetrieverTopKEdgeWriter(output_dir=wdir, split=case.get("split", "test"), topk_values=case["topk_values"])
        trainer = _types.SimpleNamespace(global_rank=0)
        w.on_test_start(trainer, None)
        w.on_test_batch_end(trainer, None, case["output"], ns, 0)
        w.on_test_end(trainer, None)
        path = os.path.join(wdir, f"{case.get('split', 'test')}.pt")
        payload = torch.load(path, weights_only=False) if os.path.exists(path) else None  # our own file, written above
        manifest = json.load(open(os.path.join(wdir, f"{case.get('split', 'test')}.manifest.json"))) if payload is not None else None
        if manifest is not None:
            manifest.pop("created_at")
        if payload is not None:  # JSON keys are strings: triplets_by_k's int keys are restored by the test
            for smp in payload["samples"]:
                smp["triplets_by_k"] = {str(k): v for k, v in smp["triplets_by_k"].items()}
        out_cases.append({"name": case["name"], "source": case["source"], "topk_values": list(case["topk_values"]) if case["topk_values"] is not None else None,
                          "split": case.get("split", "test"), "batch_extras": case["extras"], "inline": case.get("inline"),
                          "payload": payload, "manifest": manifest})
    import gzip

    path = os.path.join(HERE, "topk_writer.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:  # mtime=0: byte-reproducible
        fh.write(json.dumps({"cases": out_cases}).encode())
    print(f"wrote {os.path.relpath(path, REPO)} ({os.path.getsize(path)} B)")


def _writer_cases(toy, out_toy, mid, out_mid):
    import types as _types

    cases = []
    # (1) toy batch: edge ptr from _slice_dict, sample ids and questions as lists, k window beyond E_g (31 edges)
    ns = synthetic.as_namespace(toy)
    ns.num_graphs = toy.num_graphs
    ns._slice_dict = {"edge_index": torch.from_numpy(toy.edge_ptr)}
    ns.answer_entity_ids_ptr = torch.from_numpy(toy.answer_ptr)
    ns.sample_id = [f"WebQTest-{g}" for g in range(toy.num_graphs)]
    ns.question = [f"question number {g}?" for g in range(toy.num_graphs)]
    cases.append({"name": "toy_ptr", "source": "retriever_toy", "batch": ns, "output": out_toy, "topk_values": [1, 5, 10, 100],
                  "extras": {"slice_dict": True, "sample_id": "list", "question": "list"}})
    # (2) mid batch: no _slice_dict -> per-graph boolean masks over query_ids; sample ids as a tensor; no questions;
    #     answer ptr from _slice_dict["answer_entity_ids"]; split name "validation"
    ns = synthetic.as_namespace(mid)
    ns.num_graphs = mid.num_graphs
    ns._slice_dict = {"answer_entity_ids": torch.from_numpy(mid.answer_ptr)}
    ns.sample_id = torch.arange(100, 100 + mid.num_graphs)
    cases.append({"name": "mid_mask", "source": "retriever_mid", "batch": ns, "output": out_mid, "topk_values": [3, 50, 500],
                  "split": "validation", "extras": {"slice_dict": False, "sample_id": "tensor", "question": None}})
    # (3) hand-made: graph 1 has no edges (no record), forward-only output (no logit_bwd), default topk_values
    g = torch.Generator().manual_seed(9)
    E = 9
    ei = torch.tensor([[0, 1, 2, 0, 3, 5, 6, 5, 6], [1, 2, 0, 2, 1, 6, 5, 5, 6]])
    logits = torch.randn(E, generator=g)
    inline = {"edge_index": ei.tolist(), "edge_ptr": [0, 5, 5, 9], "edge_attr": [4, 2, 7, 7, 1, 3, 3, 0, 9],
              "labels": [1, 0, 0, 1, 0, 0, 1, 0, 0], "node_global_ids": [10, 11, 12, 13, 20, 30, 31],
              "answer_entity_ids": [12, 99, 31], "answer_ptr": [0, 2, 2, 3], "logits": logits.tolist(),
              "logits_fwd": (logits + 0.25).tolist(), "query_ids": [0, 0, 0, 0, 0, 2, 2, 2, 2]}
    ns = _types.SimpleNamespace(
        num_graphs=3, edge_index=ei, edge_attr=torch.tensor(inline["edge_attr"]), labels=torch.tensor(inline["labels"]).float(),
        node_global_ids=torch.tensor(inline["node_global_ids"]), answer_entity_ids=torch.tensor(inline["answer_entity_ids"]),
        answer_entity_ids_ptr=torch.tensor(inline["answer_ptr"]), _slice_dict={"edge_index": torch.tensor(inline["edge_ptr"])})
    from src.models.components.retriever import RetrieverOutput

    out = RetrieverOutput(logits=logits, query_ids=torch.tensor(inline["query_ids"]), relation_ids=ns.edge_attr,
                          logits_fwd=logits + 0.25, logits_bwd=None)
    cases.append({"name": "empty_graph_forward_only", "source": None, "batch": ns, "output": out, "topk_values": None,
                  "extras": {"slice_dict": True, "sample_id": None, "question": None}, "inline": inline})
    return cases


def main():
    import tempfile

    gen_cosine()
    gen_bfs()
    gen_build_graph()
    gen_g_agent_build()
    gen_loss()
    gen_feature_monitor()
    gen_ranking_metrics()
    # toy batch (BASELINE config 1 graph shape: 32 graphs, N_g = 64, E_g ~ 31), D = H = 32
    toy = synthetic.make_batch(32, nodes_per_graph=64, edges_per_graph=31, emb_dim=32, num_relations=16, seed=0)
    eb, eptr, near = gen_graph_utils(toy)
    out = gen_retriever("retriever_toy", 32, 32, toy, seed=0)
    save("graph_utils_toy", edge_batch=eb, edge_ptr=eptr, near_mask=near, **batch_arrays(toy, "b_"))
    gen_metrics(toy, out.logits, "toy")
    gen_g_agent(toy, out.logits)
    # denser graphs, D != H, more DDE rounds, k window smaller than E_g
    mid = synthetic.make_batch(6, nodes_per_graph=300, edges_per_graph=900, emb_dim=64, num_relations=40, seed=3)
    out_mid = gen_retriever("retriever_mid", 64, 48, mid, seed=1, rounds=(3, 1))
    gen_metrics(mid, out_mid.logits, "mid")
    fwd = synthetic.make_batch(4, nodes_per_graph=40, edges_per_graph=80, emb_dim=16, num_relations=8, seed=5)
    gen_retriever("retriever_fwd", 16, 16, fwd, seed=2, rounds=(2, 2), direction="forward")
    gen_retriever("retriever_bwd", 16, 16, fwd, seed=2, rounds=(0, 4), direction="backward")
    with tempfile.TemporaryDirectory() as tmp:
        gen_encode(tmp)
    with tempfile.TemporaryDirectory() as tmp:
        gen_encode_tables(tmp)
    with tempfile.TemporaryDirectory() as tmp:
        gen_topk_writer(_writer_cases(toy, out, mid, out_mid), tmp)


if __name__ == "__main__":
    main()
