"""Pins the CPU oracle (oracle/) to outputs of the reference's own functions.

tests/golden/*.npz were produced by tests/golden/make_golden.py, which imports the reference from
/root/reference in the build container and runs its functions on seeded inputs.  Integer outputs
must match bit-for-bit; float outputs within the stated tolerances (f32 reassociation only).
"""
import os
import types

import numpy as np
import pytest

from oracle import cosine as ocos
from oracle import encode as oenc
from oracle import graph as ograph
from oracle import metrics as omet
from oracle import scorer as oscorer
from oracle.ranking import stable_desc_order

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def batch_from(z, prefix="b_"):
    b = types.SimpleNamespace(**{k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)})
    b.num_graphs = int(b.ptr.shape[0] - 1)
    b.num_nodes = int(b.ptr[-1])
    return b


# ---- C1-C4 -------------------------------------------------------------------------------------------
def test_normalize_embeddings():
    z = load("cosine")
    got = ocos.normalize_embeddings(z["x"], float(z["eps"]))
    np.testing.assert_allclose(got, z["x_norm"], rtol=0, atol=1.5e-7)
    assert np.all(got[0] == 0)
    # the clamp is on the norm: a row with ||x|| < eps is divided by eps, not normalised to 1
    assert np.linalg.norm(got[5]) < 1e-2
    assert ocos.normalize_embeddings(np.empty((0, 4), np.float32), 1e-6).shape == (0, 4)


def test_canonical_edge_selection():
    z = load("cosine")
    groups = ocos.group_positive_edges_by_pair(z["edge_src"], z["edge_dst"], z["positive"])
    assert np.array_equal(np.asarray(list(groups.keys())), z["group_keys"])
    assert [len(v) for v in groups.values()] == z["group_sizes"].tolist()
    assert [i for v in groups.values() for i in v] == z["group_members"].tolist()
    keep = ocos.select_canonical_edge_indices(groups, z["edge_rel"], z["rel_norm"], z["q_norm"])
    assert keep == z["keep_indices"].tolist()
    mask, ids, counts = ocos.canonicalize_positive_mask(
        z["edge_src"], z["edge_dst"], z["edge_rel"], z["positive"], z["pair_ids"].tolist(), z["pair_counts"].tolist(),
        z["q_norm"], z["rel_norm"])
    assert ids == z["new_pair_ids"].tolist() and counts == z["new_pair_counts"].tolist()
    assert sorted(np.nonzero(mask)[0].tolist()) == sorted(z["keep_indices"].tolist())


# ---- G1-G5 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", range(5))
def test_bfs_and_shortest_path_labelling(c):
    z = load("bfs")
    n = int(z[f"c{c}_n"])
    src, dst = z[f"c{c}_src"].tolist(), z[f"c{c}_dst"].tolist()
    seeds, answers = z[f"c{c}_seeds"].tolist(), z[f"c{c}_answers"].tolist()
    adj = ograph.build_undirected_adjacency(n, src, dst)
    ptr, col = ograph.adjacency_to_csr(adj)
    assert np.array_equal(ptr, z[f"c{c}_adj_ptr"]) and np.array_equal(col, z[f"c{c}_adj"])
    dadj = ograph.build_directed_adjacency(n, src, dst)
    ptr, col = ograph.adjacency_to_csr(dadj)
    assert np.array_equal(ptr, z[f"c{c}_dadj_ptr"]) and np.array_equal(col, z[f"c{c}_dadj"])
    assert ograph.bfs_dist(n, adj, seeds) == z[f"c{c}_dist"].tolist()
    assert ograph.bfs_dist(n, dadj, seeds) == z[f"c{c}_ddist"].tolist()
    for directed, p in ((False, ""), (True, "d")):
        mask, ps, pa, pe, pc, pl = ograph.shortest_path_union_mask_by_pair(n, src, dst, seeds, answers, directed=directed)
        assert mask == z[f"c{c}_{p}mask"].tolist()
        assert ps == z[f"c{c}_{p}pair_start"].tolist() and pa == z[f"c{c}_{p}pair_answer"].tolist()
        assert pe == z[f"c{c}_{p}pair_edges"].tolist() and pc == z[f"c{c}_{p}pair_counts"].tolist()
        assert pl == z[f"c{c}_{p}pair_len"].tolist()
    e, nodes = ograph.shortest_path_single(n, src, dst, seeds, answers)
    assert e == z[f"c{c}_sp_edges"].tolist() and nodes == z[f"c{c}_sp_nodes"].tolist()


def test_has_connectivity():
    z = load("bfs")
    t = [("a", "r", "b"), ("b", "r", "c"), ("d", "r", "e"), ("c", "r2", "a")]
    got = [
        ograph.has_connectivity(t, ["a"], ["c"]), ograph.has_connectivity(t, ["a"], ["e"]),
        ograph.has_connectivity(t, ["c"], ["b"], directed=True), ograph.has_connectivity(t, ["e"], ["d"], directed=True),
        ograph.has_connectivity(t, ["zz"], ["a"]), ograph.has_connectivity([], ["a"], ["b"]),
    ]
    assert got == z["has_connectivity"].tolist()


# ---- G11 ---------------------------------------------------------------------------------------------
def test_edge_batch_and_qa_mask():
    z = load("graph_utils_toy")
    b = batch_from(z)
    eb, eptr = ograph.compute_edge_batch(b.edge_index, b.ptr, b.num_graphs)
    assert np.array_equal(eb, z["edge_batch"]) and np.array_equal(eptr, z["edge_ptr"])
    assert np.array_equal(eptr, b.edge_ptr)
    near = ograph.compute_qa_edge_mask(b.edge_index, b.num_nodes, b.q_local_indices, b.a_local_indices)
    assert np.array_equal(near, z["near_mask"])
    bad = b.edge_index.copy()
    bad[1, 0] = b.ptr[-1] - 1  # tail in the last graph, head in the first
    with pytest.raises(ValueError, match="crosses graph boundaries"):
        ograph.compute_edge_batch(bad, b.ptr, b.num_graphs)
    with pytest.raises(ValueError, match="non-decreasing"):
        ograph.compute_edge_batch(b.edge_index[:, ::-1], b.ptr, b.num_graphs)


# ---- S1-S6, G6-G7 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["retriever_toy", "retriever_mid", "retriever_fwd", "retriever_bwd"])
def test_retriever_forward(name):
    z = load(name)
    b = batch_from(z)
    w = {k[2:]: z[k] for k in z.files if k.startswith("w_")}
    rounds = z["rounds"].tolist()
    out = oscorer.retriever_forward(w, b, num_rounds=rounds[0], num_reverse_rounds=rounds[1],
                                    direction_mode=str(z["direction"]))
    assert np.array_equal(out["query_ids"], z["query_ids"])
    assert np.array_equal(out["relation_ids"], z["relation_ids"])
    np.testing.assert_allclose(out["node_struct"], z["node_struct"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["logits"], z["logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["edge_embeddings"], z["edge_embeddings"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["edge_embeddings"], z["edge_tokens"], rtol=0, atol=2e-5)
    if str(z["direction"]) == "bidirectional":
        np.testing.assert_allclose(out["logits_fwd"], z["logits_fwd"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out["logits_bwd"], z["logits_bwd"], rtol=0, atol=2e-5)
    # the ranking the metrics consume is identical
    for g in range(b.num_graphs):
        lo, hi = int(b.edge_ptr[g]), int(b.edge_ptr[g + 1])
        got, ref = stable_desc_order(out["logits"][lo:hi]), stable_desc_order(z["logits"][lo:hi])
        if not np.array_equal(got, ref):  # only near-ties may swap
            diff = np.nonzero(got != ref)[0]
            assert np.max(np.abs(z["logits"][lo:hi][got[diff]] - z["logits"][lo:hi][ref[diff]])) < 1e-4


def test_state_dict_layout():
    z = load("retriever_toy")
    D, H = int(z["D"]), int(z["H"])
    shapes = {k[2:]: z[k].shape for k in z.files if k.startswith("w_")}
    expect = {
        "entity_proj.network.0.weight": (D, D), "entity_proj.network.0.bias": (D,),
        "relation_proj.network.0.weight": (D, D), "relation_proj.network.0.bias": (D,),
        "query_proj.network.0.weight": (D, D), "query_proj.network.0.bias": (D,),
        "non_text_entity_emb.weight": (1, D), "q_gate.0.weight": (D, D), "q_gate.0.bias": (D,),
        "q_bias.0.weight": (D, D), "q_bias.0.bias": (D,), "struct_proj.0.weight": (D, 20),
        "struct_proj.0.bias": (D,), "struct_proj.1.weight": (D,), "struct_proj.1.bias": (D,),
        "struct_gate_net.0.weight": (1, D), "struct_gate_net.0.bias": (1,),
        "state_net.0.weight": (H, 3 * D + 1), "state_net.0.bias": (H,), "state_net.1.weight": (H,),
        "state_net.1.bias": (H,), "state_net.4.weight": (H, H), "state_net.4.bias": (H,),
        "score_head.weight": (1, H), "score_head.bias": (1,), "parity_meta": (4,),
    }
    assert shapes == expect
    assert z["w_parity_meta"].tolist() == [1, 2, 2, 2]


# ---- T1, T2, T4, T5 -----------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["toy", "mid"])
def test_metrics(tag):
    z = load(f"metrics_{tag}")
    b = batch_from(z)
    ks = z["k_values"].tolist()
    ref = dict(zip(z["keys"].tolist(), z["values"].tolist()))
    scores = z["scores"]
    target = b.labels > 0.5
    got = {}
    sums, cnt = omet.edge_recall_at_k(scores, target, b.edge_ptr, ks)
    got.update(omet.edge_recall_compute(sums, cnt))
    hits, valid = omet.answer_reachability(scores, b, ks)
    assert valid == float(z["reach_total"])
    got.update(omet.answer_reachability_compute(hits, valid))
    got.update(omet.score_margin(scores, target, b.edge_ptr))
    got.update(omet.bridge_metrics(scores, target, b, ks))
    h, r = omet.answer_hit_recall_batch(scores, b, ks)
    got.update(h)
    got.update(r)
    assert set(got) == set(ref), set(got) ^ set(ref)
    for k in sorted(ref):
        assert got[k] == pytest.approx(ref[k], abs=2e-6), k
    # per-sample Hits@k rows (T4) over the ranked lists
    gid = b.node_global_ids
    for g in range(b.num_graphs):
        lo, hi = int(b.edge_ptr[g]), int(b.edge_ptr[g + 1])
        order = stable_desc_order(scores[lo:hi])[:500]
        row = omet.oracle_metrics_for_sample(gid[b.edge_index[0, lo:hi]][order], gid[b.edge_index[1, lo:hi]][order],
                                             b.answer_entity_ids[int(b.answer_ptr[g]): int(b.answer_ptr[g + 1])], ks)
        flat = [row[f"answer_hit@{k}"] for k in ks] + [row[f"answer_recall@{k}"] for k in ks]
        assert flat == pytest.approx(z["oracle_rows"][g].tolist(), abs=1e-12)


# ---- G8, G9 -------------------------------------------------------------------------------------------
def test_g_agent_edge_selection():
    z = load("g_agent_select")
    b = batch_from(z)
    scores = z["scores"]
    params = z["start_params"]
    for g in range(int(z["num_graphs"])):
        lo, hi = int(b.edge_ptr[g]), int(b.edge_ptr[g + 1])
        n0, n1 = int(b.ptr[g]), int(b.ptr[g + 1])
        heads, tails = b.edge_index[0, lo:hi] - n0, b.edge_index[1, lo:hi] - n0
        logit = ograph.node_softmax_logit(scores[lo:hi], heads, tails, n1 - n0)
        np.testing.assert_allclose(logit, z[f"g{g}_logit"], rtol=0, atol=3e-5)
        # selections are checked on the REFERENCE's logits so that integer outputs are exact
        ref_logit = z[f"g{g}_logit"]
        for tk in (5, 500):
            assert np.array_equal(ograph.select_topk_edges(ref_logit, tk), z[f"g{g}_topk{tk}"])
        seeds = b.q_local_indices[int(b.q_ptr[g]): int(b.q_ptr[g + 1])] - n0
        for ri, (ratio, mn, mx) in enumerate(params.tolist()):
            got = ograph.select_start_edges(heads, tails, ref_logit, seeds, n1 - n0, ratio, int(mn),
                                            None if mx < 0 else int(mx))
            assert np.array_equal(got, z[f"g{g}_start{ri}"]), (g, ri)


# ---- E2, E3 -------------------------------------------------------------------------------------------
def test_masked_mean_pool_and_table_scatter():
    z = load("encode")
    nb = int(z["num_batches"])
    for fp16 in (False, True):
        pooled = np.concatenate([oenc.masked_mean_pool(z[f"hidden_{i}"], z[f"mask_{i}"], fp16=fp16) for i in range(nb)])
        tol = 2e-3 if fp16 else 1e-6
        np.testing.assert_allclose(pooled, z[f"pooled_fp16_{int(fp16)}"], rtol=0, atol=tol)
    # an empty text pools to 0/eps = 0 (mask all zero), and encode([]) returns shape (0, 0)
    assert np.all(z["pooled_fp16_0"][3] == 0)
    assert z["empty_shape"].tolist() == [0, 0]
    chunks = [z["pooled_fp16_0"][s:e] for s, e in oenc.iter_batches(10, 4)]
    ids = [z["emb_ids"][s:e].tolist() for s, e in oenc.iter_batches(10, 4)]
    table = oenc.scatter_rows(chunks, ids, int(z["max_embedding_id"]), int(z["D"]))
    assert np.array_equal(table, z["memmap_table"])
    assert np.all(table[0] == 0)  # row 0 = non-text placeholder stays zero


def test_build_graph_ids_matches_reference():
    from oracle import graph as ograph

    z = load("build_graph")
    for c in range(int(z["num_cases"])):
        got = ograph.build_graph_ids(z[f"c{c}_triples"], z[f"c{c}_q"], z[f"c{c}_a"], z[f"c{c}_asub"], z["ent_struct"],
                                     z["ent_emb"], directed=bool(z[f"c{c}_directed"]), dedup_edges=bool(z[f"c{c}_dedup"]),
                                     remove_self_loops=bool(z[f"c{c}_noloop"]))
        for name in ("node_entity_ids", "node_embedding_ids", "edge_src", "edge_dst", "edge_rel", "positive", "pair_start",
                     "pair_answer", "pair_edges", "pair_counts", "pair_len"):
            assert np.array_equal(np.asarray(got[name]), z[f"c{c}_{name}"]), (c, name)


def _g_agent_cases(z):
    cfgs = [dict(edge_top_k=20, start_keep_ratio=0.25, start_min_edges=1, start_max_edges=None, allow_empty_answer=False,
                 node_softmax=True, temperature=1.0, bias=0.0),
            dict(edge_top_k=500, start_keep_ratio=0.5, start_min_edges=2, start_max_edges=4, allow_empty_answer=True,
                 node_softmax=False, temperature=2.0, bias=0.5)]
    return cfgs


def test_g_agent_build_sample_matches_reference():
    from oracle import g_agent as og

    z = load("g_agent_build")
    for ci, cfg in enumerate(_g_agent_cases(z)):
        temperature, bias = cfg.pop("temperature"), cfg.pop("bias")
        scores_all = z["logits"].astype(np.float32)
        if temperature != 1.0 or bias != 0.0:
            scores_all = (scores_all / np.float32(temperature) + np.float32(bias)).astype(np.float32)
        kept = []
        for g in range(int(z["num_graphs"])):
            lo, hi = int(z["edge_ptr"][g]), int(z["edge_ptr"][g + 1])
            n0, n1 = int(z["ptr"][g]), int(z["ptr"][g + 1])
            out = og.build_sample(
                heads=z["edge_index"][0, lo:hi] - n0, tails=z["edge_index"][1, lo:hi] - n0, relations=z["edge_attr"][lo:hi],
                labels=z["labels"][lo:hi], scores=scores_all[lo:hi], node_global_ids=z["node_global_ids"][n0:n1],
                node_embedding_ids=z["node_embedding_ids"][n0:n1],
                start_entity_ids=z["seeds"][int(z["seed_ptr"][g]): int(z["seed_ptr"][g + 1])],
                answer_entity_ids=z["answers"][int(z["ans_ptr"][g]): int(z["ans_ptr"][g + 1])], **cfg)
            if out is not None:
                kept.append((f"s{g}", out))
        assert [k for k, _ in kept] == z[f"cfg{ci}_sample_ids"].tolist()
        assert len(kept) == int(z[f"cfg{ci}_num_samples"])
        for si, (_, out) in enumerate(kept):
            for name, val in out.items():
                ref = z[f"cfg{ci}_s{si}_{name}"]
                if val.dtype == np.float32:
                    np.testing.assert_allclose(val, ref, rtol=1e-6, atol=1e-6, err_msg=f"{ci}/{si}/{name}")
                else:
                    assert np.array_equal(val, ref), (ci, si, name)


def _loss_cases(z):
    for variant, tkey in (("base", "targets"), ("nopos", "nopos_targets")):
        for ci, (T, wi, wb, wn, wbr) in enumerate(z["cfgs"].tolist()):
            yield f"{variant}_c{ci}", z[tkey], dict(infonce_temperature=T, infonce_weight=wi, bce_weight=wb, edge_weight_near=wn,
                                                   edge_weight_bridge=wbr)


def test_retriever_loss_matches_reference():
    from oracle import loss as oloss

    z = load("loss")
    for tag, targets, cfg in _loss_cases(z):
        total, comps, mets, grad = oloss.retriever_loss(z["logits"], targets, z["edge_batch"], int(z["num_graphs"]),
                                                        edge_is_near=z["edge_is_near"], **cfg)
        assert abs(total - float(z[f"{tag}_loss"])) < 2e-6 * max(1.0, abs(total)), tag
        np.testing.assert_allclose(grad, z[f"{tag}_grad"], rtol=2e-5, atol=2e-7, err_msg=tag)
        assert sorted(comps) == z[f"{tag}_component_keys"].tolist() and sorted(mets) == z[f"{tag}_metric_keys"].tolist(), tag
        np.testing.assert_allclose([comps[k] for k in sorted(comps)], z[f"{tag}_component_vals"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose([mets[k] for k in sorted(mets)], z[f"{tag}_metric_vals"], rtol=2e-6, atol=2e-6)


# ---- T4b -------------------------------------------------------------------------------------------------------------
def test_ranking_metrics_match_the_reference_function():
    z = load("ranking_metrics")
    ptr, ks = z["ptr"], [int(k) for k in z["k_values"]]
    samples = [(z["scores"][a:b], z["labels"][a:b]) for a, b in zip(ptr[:-1], ptr[1:])]
    got = omet.ranking_metrics(samples, ks)
    for name in ("precision", "recall", "f1", "ndcg"):
        np.testing.assert_allclose([got[name][k] for k in ks], z[name], rtol=0, atol=1e-7)  # f32 sums: reassociation only
    assert got["mrr"] == pytest.approx(float(z["mrr"]), abs=1e-15)
    assert omet.ranking_metrics(samples, None)["precision"][1] == pytest.approx(float(z["default_k_precision"]), abs=1e-15)
    none = omet.ranking_metrics([samples[1]], ks)  # the all-negative sample alone: every mean 0, MRR 0
    assert none["mrr"] == float(z["empty_mrr"]) and [none["recall"][k] for k in ks] == z["empty_recall"].tolist()
