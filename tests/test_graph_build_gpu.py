"""GPU parity: segmented de-duplication primitives, build_graph (G5), single shortest path (G4)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import graph as ograph

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def _first_occurrence_ref(keys, seg, drop):
    first = np.full(keys.shape[0], -1, np.int64)
    for s in range(len(seg) - 1):
        seen = {}
        for p in range(seg[s], seg[s + 1]):
            if drop is not None and drop[p]:
                continue
            k = tuple(keys[p].tolist())
            first[p] = seen.setdefault(k, p - seg[s])
    return first


@pytest.mark.parametrize("W,lens,vocab", [(1, [0, 5, 3000, 1], 50), (3, [700, 0, 20000], 12), (2, [1500], 40), (1, [70000], 3000)])
def test_first_occurrence_and_rank(dev, W, lens, vocab):
    from evi_rag_amd import ops

    rng = np.random.default_rng(sum(lens) + W)
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    keys = rng.integers(-vocab, vocab, size=(int(seg[-1]), W)).astype(np.int64)
    keys[::7] *= 1 << 33  # wide values: the whole 64-bit word takes part
    drop = rng.random(int(seg[-1])) < 0.1
    limit = np.asarray([int(n * 0.8) for n in lens], np.int64)
    seg_t = torch.from_numpy(seg).to(dev)
    first = ops.first_occurrence(torch.from_numpy(keys).to(dev), seg_t, torch.from_numpy(drop).to(dev))
    ref = _first_occurrence_ref(keys, seg, drop)
    assert np.array_equal(first.cpu().numpy().astype(np.int64), ref)
    rank, count, uniq = ops.first_seen_rank(first, seg_t, torch.from_numpy(limit).to(dev))
    rank, count, uniq = rank.cpu().numpy(), count.cpu().numpy(), uniq.cpu().numpy()
    for s, n in enumerate(lens):
        f = ref[seg[s]: seg[s + 1]]
        firsts = [p for p in range(n) if p < limit[s] and f[p] == p]
        assert count[s] == len(firsts)
        assert uniq[seg[s]: seg[s] + len(firsts)].tolist() == firsts
        want = np.full(n, -1, np.int64)
        order = {p: r for r, p in enumerate(firsts)}
        for p in range(n):
            if f[p] >= 0 and f[p] in order:
                want[p] = order[f[p]]
        assert np.array_equal(rank[seg[s]: seg[s + 1]].astype(np.int64), want)


@pytest.mark.parametrize("lens", [[0, 1, 17, 1024, 1025], [5000, 3]])
def test_segment_sort_rank_and_group_max(dev, lens):
    from evi_rag_amd import ops

    rng = np.random.default_rng(len(lens))
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    keys = rng.integers(-1000, 1000, size=int(seg[-1])).astype(np.int64)  # with repeats: the rank is stable
    use = np.asarray([n - n // 5 for n in lens], np.int32)
    rank, srt = ops.segment_sort_rank(torch.from_numpy(keys).to(dev), torch.from_numpy(seg).to(dev), torch.from_numpy(use).to(dev))
    rank, srt = rank.cpu().numpy(), srt.cpu().numpy()
    for s, n in enumerate(lens):
        k = keys[seg[s]: seg[s] + use[s]]
        order = np.argsort(k, kind="stable")
        want = np.empty(use[s], np.int64)
        want[order] = np.arange(use[s])
        assert np.array_equal(rank[seg[s]: seg[s] + use[s]].astype(np.int64), want)
        assert np.array_equal(srt[seg[s]: seg[s] + use[s]], k[order])
    vals = rng.standard_normal(5000).astype(np.float32)
    vals[::11] = 0.0
    grp = rng.integers(-1, 40, size=5000).astype(np.int32)
    got = ops.group_max(torch.from_numpy(vals).to(dev), torch.from_numpy(grp).to(dev), 41).cpu().numpy()
    want = np.full(41, -np.inf, np.float32)
    np.maximum.at(want, grp[grp >= 0], vals[grp >= 0])
    assert np.array_equal(got, want)


def _golden_samples():
    z = load("build_graph")
    ent = lambda i: f"m.{int(i):03d}"  # noqa: E731
    rel = lambda r: f"rel.{int(r)}"    # noqa: E731
    struct = {ent(i): int(v) for i, v in enumerate(z["ent_struct"])}
    emb = {ent(i): int(v) for i, v in enumerate(z["ent_emb"])}
    ent_vocab = types.SimpleNamespace(entity_id=lambda e: struct[e], embedding_id=lambda e: emb[e])
    rel_vocab = types.SimpleNamespace(relation_id=lambda r: int(r.split(".")[1]))
    cases = []
    for c in range(int(z["num_cases"])):
        graph = [(ent(h), rel(r), ent(t)) for h, r, t in z[f"c{c}_triples"].tolist()]
        asub = [(ent(h), rel(r), ent(t)) for h, r, t in z[f"c{c}_asub"].tolist()]
        sample = types.SimpleNamespace(graph=graph, q_entity=[ent(i) for i in z[f"c{c}_q"]], a_entity=[ent(i) for i in z[f"c{c}_a"]],
                                       answer_subgraph=asub)
        cases.append((c, sample))
    return z, ent_vocab, rel_vocab, cases


def _check_record(rec, z, c):
    for name, got in (("node_entity_ids", rec.node_entity_ids), ("node_embedding_ids", rec.node_embedding_ids),
                      ("edge_src", rec.edge_src), ("edge_dst", rec.edge_dst), ("edge_rel", rec.edge_relation_ids),
                      ("positive", rec.positive_triple_mask), ("pair_start", rec.pair_start_node_locals),
                      ("pair_answer", rec.pair_answer_node_locals), ("pair_edges", rec.pair_edge_local_ids),
                      ("pair_counts", rec.pair_edge_counts), ("pair_len", rec.pair_shortest_lengths)):
        assert np.array_equal(np.asarray(got), z[f"c{c}_{name}"]), (c, name)


def test_build_graph_matches_reference_golden(dev):
    """Every golden case (outputs of the reference's build_graph), one at a time and grouped in chunks
    that share path_mode / dedup / self-loop settings."""
    from evi_rag_amd import graph_build

    z, ent_vocab, rel_vocab, cases = _golden_samples()
    by_cfg = {}
    for c, sample in cases:
        cfg = ("qa_directed" if bool(z[f"c{c}_directed"]) else "undirected", bool(z[f"c{c}_dedup"]), bool(z[f"c{c}_noloop"]))
        rec = graph_build.build_graph(sample, ent_vocab, rel_vocab, f"g{c}", path_mode=cfg[0], dedup_edges=cfg[1],
                                      remove_self_loops=cfg[2])
        _check_record(rec, z, c)
        by_cfg.setdefault(cfg, []).append((c, sample))
    for cfg, group in by_cfg.items():
        recs = graph_build.build_graphs([s for _, s in group], ent_vocab, rel_vocab, [f"g{c}" for c, _ in group],
                                        path_mode=cfg[0], dedup_edges=cfg[1], remove_self_loops=cfg[2])
        for (c, _), rec in zip(group, recs):
            _check_record(rec, z, c)
    with pytest.raises(ValueError, match="Unsupported path_mode"):
        graph_build.build_graph(cases[0][1], ent_vocab, rel_vocab, "g", path_mode="sideways")


def test_index_graphs_coded_matches_oracle_large(dev):
    """CWQ-sized samples (10^4 triples, heavy duplication) against the oracle restatement."""
    from evi_rag_amd import graph_build

    rng = np.random.default_rng(77)
    tri, qs, as_, subs = [], [], [], []
    for n_trip, n_ent in [(12000, 3000), (30000, 900), (1, 2), (4096, 1500)]:
        t = np.stack([rng.integers(0, n_ent, n_trip), rng.integers(0, 30, n_trip), rng.integers(0, n_ent, n_trip)], 1)
        tri.append(t.astype(np.int64))
        qs.append(rng.integers(0, n_ent, 2).tolist())
        as_.append(rng.integers(0, n_ent, 3).tolist())
        subs.append(t[rng.integers(0, n_trip, 40)])
    for dedup, noloop in [(True, True), (False, True), (True, False)]:
        coded = graph_build.index_graphs_coded(tri, qs, as_, subs, dedup_edges=dedup, remove_self_loops=noloop)
        labels = graph_build.label_graphs(coded)
        for s, (cg, lab) in enumerate(zip(coded, labels)):
            n_ent = int(tri[s][:, [0, 2]].max()) + 1
            ref = ograph.build_graph_ids(tri[s], qs[s], as_[s], subs[s], np.arange(n_ent), np.arange(n_ent), dedup_edges=dedup,
                                         remove_self_loops=noloop)
            assert np.array_equal(cg.node_codes, ref["node_entity_ids"])
            assert np.array_equal(cg.edge_src, ref["edge_src"]) and np.array_equal(cg.edge_dst, ref["edge_dst"])
            assert np.array_equal(cg.edge_rel, ref["edge_rel"])
            positive, ps, pa, pe, pc, pl = lab
            assert np.array_equal(np.asarray(positive, bool), np.asarray(ref["positive"], bool))
            assert (ps, pa, pe, pc, pl) == (ref["pair_start"], ref["pair_answer"], ref["pair_edges"], ref["pair_counts"], ref["pair_len"])


def test_shortest_path_single_golden_and_random(dev):
    from evi_rag_amd import labelling

    z = load("bfs")
    for c in range(int(z["num_cases"])):
        n = int(z[f"c{c}_n"])
        e, nodes = labelling.shortest_path_single(n, z[f"c{c}_src"], z[f"c{c}_dst"], z[f"c{c}_seeds"].tolist(), z[f"c{c}_answers"].tolist())
        assert e == z[f"c{c}_sp_edges"].tolist() and nodes == z[f"c{c}_sp_nodes"].tolist(), c
    assert labelling.shortest_path_single(5, [0], [1], [], [1]) == ([], [])
    # random multigraphs, batched: many parallel edges and equal-length alternatives stress the tie rules
    rng = np.random.default_rng(9)
    ns, srcs, dsts, seeds, answers = [], [], [], [], []
    for g in range(40):
        n = int(rng.integers(2, 60))
        e = int(rng.integers(0, 4 * n))
        ns.append(n)
        srcs.append(rng.integers(0, n, e))
        dsts.append(rng.integers(0, n, e))
        seeds.append(rng.integers(0, n, int(rng.integers(1, 4))).tolist())
        answers.append(rng.integers(0, n, int(rng.integers(1, 4))).tolist())
    ns.append(400)  # a long chain: path longer than the default buffer
    srcs.append(np.arange(399))
    dsts.append(np.arange(1, 400))
    seeds.append([0])
    answers.append([399])
    gb = labelling.GraphBatch(ns, srcs, dsts)
    got = labelling.shortest_path_single_batch(gb, seeds, answers)
    for g in range(len(ns)):
        ref = ograph.shortest_path_single(ns[g], srcs[g].tolist(), dsts[g].tolist(), seeds[g], answers[g])
        assert got[g] == (list(ref[0]), list(ref[1])), g


def test_has_connectivity_golden(dev):
    from evi_rag_amd import labelling

    z = load("bfs")
    t = [("a", "r", "b"), ("b", "r", "c"), ("d", "r", "e"), ("c", "r2", "a")]
    got = [labelling.has_connectivity(t, ["a"], ["c"]), labelling.has_connectivity(t, ["a"], ["e"]),
           labelling.has_connectivity(t, ["c"], ["b"], path_mode="qa_directed"),
           labelling.has_connectivity(t, ["e"], ["d"], path_mode="qa_directed"),
           labelling.has_connectivity(t, ["zz"], ["a"]), labelling.has_connectivity([], ["a"], ["b"])]
    assert got == z["has_connectivity"].tolist()


def test_build_graphs_to_packed_split_to_forward(dev, tmp_path):
    """The offline chain without an LMDB: raw samples -> build_graphs (device) -> records_to_samples -> write_packed ->
    PackedRetrievalDataset -> Retriever.forward + metrics.  The golden cases with at least one kept edge go through; what
    the packed batch holds equals the GraphRecords it was made from."""
    from evi_rag_amd import graph_build, packed_dataset as pd
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore
    from evi_rag_amd.metrics import RetrieverMetricCollection
    from evi_rag_amd.retriever import Retriever

    z, ent_vocab, rel_vocab, cases = _golden_samples()
    group = [(c, s) for c, s in cases if not bool(z[f"c{c}_directed"]) and bool(z[f"c{c}_dedup"]) and bool(z[f"c{c}_noloop"])]
    recs = graph_build.build_graphs([s for _, s in group], ent_vocab, rel_vocab, [f"g{c}" for c, _ in group], path_mode="undirected",
                                    dedup_edges=True, remove_self_loops=True)
    keep = [(g, r) for g, r in zip(group, recs) if len(r.edge_src) > 0]
    assert len(keep) >= 2
    recs = [r for _, r in keep]
    seeds = [[ent_vocab.entity_id(e) for e in s.q_entity] for (_, s), _ in keep]
    answers = [[ent_vocab.entity_id(e) for e in s.a_entity] for (_, s), _ in keep]
    D = 16
    rng = np.random.default_rng(0)
    samples = graph_build.records_to_samples(recs, seeds, answers, rng.standard_normal((len(recs), D)).astype(np.float32),
                                             questions=[f"question {i}" for i in range(len(recs))])
    pd.write_packed(tmp_path / "s.packed", samples)
    n_emb = max(max(r.node_embedding_ids) for r in recs) + 1
    n_rel = max(max(r.edge_relation_ids) for r in recs) + 1
    store = GlobalEmbeddingStore.from_tensors(torch.from_numpy(rng.standard_normal((n_emb, D)).astype(np.float32)).to(dev),
                                              torch.from_numpy(rng.standard_normal((n_rel, D)).astype(np.float32)).to(dev), device=dev)
    ds = pd.PackedRetrievalDataset(tmp_path / "s.packed", device=dev, embeddings=store)
    batch = next(iter(pd.PackedLoader(ds, batch_size=len(recs))))
    ptr = batch.ptr.cpu().tolist()
    for i, r in enumerate(recs):
        e0, e1 = int(batch.edge_ptr[i]), int(batch.edge_ptr[i + 1])
        assert (batch.edge_index[0, e0:e1] - ptr[i]).cpu().tolist() == r.edge_src
        assert batch.edge_attr[e0:e1].cpu().tolist() == r.edge_relation_ids
        assert (batch.labels[e0:e1] > 0.5).cpu().tolist() == list(r.positive_triple_mask)
        assert batch.node_embedding_ids[ptr[i]: ptr[i + 1]].cpu().tolist() == r.node_embedding_ids
    torch.manual_seed(0)
    model = Retriever(emb_dim=D, hidden_dim=D).to(dev).eval()
    with torch.no_grad():
        out = model(batch)
    assert out.logits.numel() == int(batch.edge_ptr[-1]) and bool(torch.isfinite(out.logits).all())
    coll = RetrieverMetricCollection([1, 5])
    coll.update(preds=out.logits, target=batch.labels > 0.5, indexes=out.query_ids, batch=batch, num_graphs=len(recs))
    assert all(np.isfinite(float(v)) for v in coll.compute().values())
