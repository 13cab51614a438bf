"""GPU parity of BFS / shortest-path labelling (G1-G4), seed expansion (G8-G10) and the encoding
tail (E2/E3) against the reference-generated golden vectors and the oracle."""
import os
import types

import numpy as np
import pytest
import torch

from evi_rag_amd import synthetic
from oracle import encode as oenc
from oracle import graph as ograph

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("c", range(5))
def test_bfs_and_shortest_path_match_reference_golden(dev, c):
    from evi_rag_amd import labelling as L

    z = _load("bfs")
    n = int(z[f"c{c}_n"])
    src, dst = z[f"c{c}_src"], z[f"c{c}_dst"]
    seeds, answers = z[f"c{c}_seeds"].tolist(), z[f"c{c}_answers"].tolist()
    assert L.bfs_dist(n, src, dst, seeds) == z[f"c{c}_dist"].tolist()
    assert L.bfs_dist(n, src, dst, seeds, directed=True) == z[f"c{c}_ddist"].tolist()
    for directed, p in ((False, ""), (True, "d")):
        mask, ps, pa, pe, pc, pl = L.shortest_path_union_mask_by_pair(n, src, dst, seeds, answers, directed=directed)
        assert mask == z[f"c{c}_{p}mask"].tolist()
        assert ps == z[f"c{c}_{p}pair_start"].tolist() and pa == z[f"c{c}_{p}pair_answer"].tolist()
        assert pe == z[f"c{c}_{p}pair_edges"].tolist() and pc == z[f"c{c}_{p}pair_counts"].tolist()
        assert pl == z[f"c{c}_{p}pair_len"].tolist()


def test_batched_labelling_matches_oracle(dev):
    """CWQ-shaped graphs (N_g ~ 3000, E_g ~ 10000: BASELINE config 3), all graphs in one launch."""
    from evi_rag_amd import labelling as L

    sb = synthetic.make_batch(6, nodes_per_graph=3000, edges_per_graph=10000, emb_dim=4, seed=2,
                              attach_embeddings=False, max_seeds=3, max_answers=6)
    nn, es, ed, seeds, answers = [], [], [], [], []
    for g in range(sb.num_graphs):
        n0, n1, e0, e1 = sb.ptr[g], sb.ptr[g + 1], sb.edge_ptr[g], sb.edge_ptr[g + 1]
        nn.append(int(n1 - n0))
        es.append(sb.edge_index[0, e0:e1] - n0)
        ed.append(sb.edge_index[1, e0:e1] - n0)
        seeds.append((sb.q_local_indices[sb.q_ptr[g]: sb.q_ptr[g + 1]] - n0).tolist())
        answers.append((sb.a_local_indices[sb.a_ptr[g]: sb.a_ptr[g + 1]] - n0).tolist())
    gb = L.GraphBatch(nn, es, ed)
    for mode, directed in ((0, False), (1, True)):
        dists = L.bfs_dist_batch(gb, seeds, mode=mode)
        for g in range(sb.num_graphs):
            adj = (ograph.build_directed_adjacency if directed else ograph.build_undirected_adjacency)(nn[g], es[g], ed[g])
            assert dists[g].tolist() == ograph.bfs_dist(nn[g], adj, seeds[g])
    for directed in (False, True):
        res = L.shortest_path_union_mask_by_pair_batch(gb, seeds, answers, directed=directed)
        for g in range(sb.num_graphs):
            ref = ograph.shortest_path_union_mask_by_pair(nn[g], es[g], ed[g], seeds[g], answers[g], directed=directed)
            assert res[g][0].tolist() == ref[0]
            assert tuple(res[g][1:]) == tuple(ref[1:]), (g, directed)


def test_flat_labelling_entry_point_matches_oracle(dev):
    """`labelling.label_pairs_flat` (row G3 / §8f-3, scripts/build_retrieval_pipeline.py:691-815): the collated batch's
    device arrays + host seed / answer CSR in, flat device arrays out — dense pair slots with -1 for "no path", counts,
    offsets and ascending edge ids; the tuples built from them equal the oracle's for every graph, both path modes, with
    duplicate / out-of-graph seeds, an answer that is a seed, a graph without edges, one without seeds and an empty graph."""
    from evi_rag_amd import labelling as L

    rng = np.random.default_rng(5)
    nn = [40, 7, 0, 25, 12, 300]
    edges = []
    for g, n in enumerate(nn):
        m = [120, 0, 0, 30, 14, 900][g]
        if n == 0 or m == 0:
            edges.append((np.empty(0, np.int64), np.empty(0, np.int64)))
            continue
        src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
        if g == 3:  # two components: some pairs have no path
            src, dst = src % 12, dst % 12
        edges.append((src.astype(np.int64), dst.astype(np.int64)))
    seeds = [[3, 3, 1, 999, -2], [0, 1], [], [0, 20], [], [5, 17, 250]]
    answers = [[1, 8, 39], [2], [], [3, 24, 24], [4], [17, 0, 299, 123]]
    node_ptr = np.concatenate([[0], np.cumsum(nn)]).astype(np.int64)
    edge_ptr = np.concatenate([[0], np.cumsum([e[0].shape[0] for e in edges])]).astype(np.int64)
    ei = np.stack([np.concatenate([e[0] + node_ptr[g] for g, e in enumerate(edges)]),
                   np.concatenate([e[1] + node_ptr[g] for g, e in enumerate(edges)])])

    def csr(lists):  # batch-global ids; an id outside its graph stays outside after the offset (or lands in ANOTHER graph)
        ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
        idx = np.asarray([v + node_ptr[g] if 0 <= v < nn[g] else -7 for g, x in enumerate(lists) for v in x], np.int64)
        return ptr, idx

    sp, si = csr(seeds)
    ap, ai = csr(answers)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for directed in (False, True):
        flat = L.label_pairs_flat(t(ei), t(node_ptr), t(edge_ptr), sp, si, ap, ai, directed=directed)
        # flat contract
        assert flat.B == len(nn) and flat.E == ei.shape[1] and flat.mask.dtype == torch.uint8
        assert flat.pair_ptr_h.tolist()[-1] == flat.P == flat.pair_len.numel() == flat.pair_count.numel()
        off = flat.pair_edge_off.cpu().numpy()
        cnt = flat.pair_count.cpu().numpy()
        ln = flat.pair_len.cpu().numpy()
        assert off[0] == 0 and np.array_equal(np.diff(off), cnt) and (cnt[ln < 0] == 0).all()
        ids = flat.pair_edge_ids[: off[-1]].cpu().numpy()
        for p in range(flat.P):
            seg = ids[off[p]: off[p + 1]]
            g = flat.pair_graph_h[p]
            assert (np.diff(seg) > 0).all() and (seg >= edge_ptr[g]).all() and (seg < edge_ptr[g + 1]).all()
        # graph 1 has no edges, graph 2 is empty, graph 4 has no seeds: no pair slots
        assert [int(flat.pair_ptr_h[g + 1] - flat.pair_ptr_h[g]) for g in range(len(nn))] == [2 * 3, 0, 0, 2 * 2, 0, 3 * 4]
        res = flat.per_graph()
        for g in range(len(nn)):
            ref = ograph.shortest_path_union_mask_by_pair(nn[g], edges[g][0], edges[g][1], seeds[g], answers[g], directed=directed)
            assert res[g][0].tolist() == list(ref[0]), (g, directed)
            assert tuple(list(x) for x in res[g][1:]) == tuple(list(x) for x in ref[1:]), (g, directed)
        # the list-of-arrays mirror is a wrapper over the same entry point
        gb = L.GraphBatch(nn, [e[0] for e in edges], [e[1] for e in edges])
        res2 = L.shortest_path_union_mask_by_pair_batch(gb, seeds, answers, directed=directed)
        for a, b in zip(res, res2):
            assert a[0].tolist() == b[0].tolist() and tuple(a[1:]) == tuple(b[1:])
    # the id buffer: sized by the host-side bound without a read-back, or — when that bound is too large — from the exact total
    ref_flat = L.label_pairs_flat(t(ei), t(node_ptr), t(edge_ptr), sp, si, ap, ai)
    old_limit, L._PAIR_IDS_BOUND_LIMIT = L._PAIR_IDS_BOUND_LIMIT, 10
    try:
        tight = L.label_pairs_flat(t(ei), t(node_ptr), t(edge_ptr), sp, si, ap, ai)
    finally:
        L._PAIR_IDS_BOUND_LIMIT = old_limit
    total = int(ref_flat.pair_edge_off[-1].item())
    assert tight.pair_edge_ids.numel() == max(total, 1) and ref_flat.pair_edge_ids.numel() >= total
    assert torch.equal(tight.pair_edge_ids[:total], ref_flat.pair_edge_ids[:total]) and torch.equal(tight.mask, ref_flat.mask)
    # nothing to pair at all
    none = L.label_pairs_flat(t(ei), t(node_ptr), t(edge_ptr), np.zeros(7, np.int64), np.empty(0, np.int64), ap, ai)
    assert none.P == 0 and int(none.mask.sum()) == 0 and all(r[1:] == ([], [], [], [], []) for r in none.per_graph())
    with pytest.raises(ValueError, match="ptr must have"):
        L.label_pairs_flat(t(ei), t(node_ptr), t(edge_ptr), np.zeros(3, np.int64), np.empty(0, np.int64), ap, ai)


def test_g_agent_selection_matches_reference_golden(dev):
    from evi_rag_amd import labelling as L

    z = _load("g_agent_select")
    b = types.SimpleNamespace(**{k[2:]: z[k] for k in z.files if k.startswith("b_")})
    scores = z["scores"]
    params = z["start_params"].tolist()
    for g in range(int(z["num_graphs"])):
        lo, hi = int(b.edge_ptr[g]), int(b.edge_ptr[g + 1])
        n0, n1 = int(b.ptr[g]), int(b.ptr[g + 1])
        heads = torch.from_numpy(b.edge_index[0, lo:hi] - n0).to(dev)
        tails = torch.from_numpy(b.edge_index[1, lo:hi] - n0).to(dev)
        logit = L.node_softmax_logit(edge_scores=torch.from_numpy(scores[lo:hi]).to(dev), edge_head_locals=heads,
                                     edge_tail_locals=tails, num_nodes=n1 - n0)
        np.testing.assert_allclose(logit.cpu().numpy(), z[f"g{g}_logit"], rtol=0, atol=3e-5)
        ref_logit = torch.from_numpy(z[f"g{g}_logit"]).to(dev)  # integer outputs checked on the reference's logits
        for tk in (5, 500):
            assert np.array_equal(L.select_topk_edges(edge_scores=ref_logit, edge_top_k=tk).cpu().numpy(), z[f"g{g}_topk{tk}"])
        seeds = torch.from_numpy(b.q_local_indices[int(b.q_ptr[g]): int(b.q_ptr[g + 1])] - n0).to(dev)
        for ri, (ratio, mn, mx) in enumerate(params):
            got = L.select_start_edges(heads=heads, tails=tails, edge_scores=ref_logit, start_node_locals=seeds,
                                       num_nodes=n1 - n0, start_keep_ratio=ratio, start_min_edges=int(mn),
                                       start_max_edges=None if mx < 0 else int(mx))
            assert np.array_equal(got.cpu().numpy(), z[f"g{g}_start{ri}"]), (g, ri)


def test_seed_expansion_hub_matches_oracle(dev):
    """A hub seed with thousands of incident edges, exact score ties and a self loop."""
    from evi_rag_amd import labelling as L

    rng = np.random.default_rng(0)
    n, e = 500, 20000
    heads = rng.integers(0, n, size=e)
    tails = rng.integers(0, n, size=e)
    heads[: e // 2] = 7  # hub as head
    tails[e // 2: e // 2 + 3000] = 7  # and as tail
    heads[5], tails[5] = 7, 7  # self loop
    scores = np.round(rng.standard_normal(e), 1).astype(np.float32)  # many ties
    labels = (rng.random(e) < 0.1).astype(np.float32)
    seeds = np.array([7, 3, 7, 11])
    for ratio, mn, mx in [(0.25, 1, None), (0.01, 1, 100), (1.0, 1, None), (0.5, 0, 2000)]:
        ref = ograph.select_start_edges(heads, tails, scores, seeds, n, ratio, mn, mx)
        got = L.select_start_edges(heads=torch.from_numpy(heads).to(dev), tails=torch.from_numpy(tails).to(dev),
                                   edge_scores=torch.from_numpy(scores).to(dev), start_node_locals=torch.from_numpy(seeds),
                                   num_nodes=n, start_keep_ratio=ratio, start_min_edges=mn, start_max_edges=mx)
        assert np.array_equal(got.cpu().numpy(), ref), (ratio, mn, mx)
    stats = L.seed_onehop_stats(torch.from_numpy(heads).to(dev), torch.from_numpy(tails).to(dev),
                                torch.from_numpy(labels).to(dev), torch.tensor([7, 3, 7, 11, 9999, -1]), n)
    assert stats == ograph.seed_onehop_stats(heads, tails, labels, np.array([7, 3, 7, 11, 9999, -1]), n)


def test_masked_mean_pool_and_scatter_match_reference_golden(dev):
    from evi_rag_amd import text_encode as T

    z = _load("encode")
    nb = int(z["num_batches"])
    for fp16 in (False, True):
        pooled = torch.cat([T.masked_mean_pool(torch.from_numpy(z[f"hidden_{i}"]).to(dev),
                                               torch.from_numpy(z[f"mask_{i}"]).to(dev), fp16=fp16) for i in range(nb)])
        np.testing.assert_allclose(pooled.cpu().numpy(), z[f"pooled_fp16_{int(fp16)}"], rtol=0, atol=2e-3 if fp16 else 1e-6)
    # f16 / bf16 hidden states (a half-precision encoder) pool like their f32 upcast
    h = torch.from_numpy(z["hidden_1"]).to(dev)
    m = torch.from_numpy(z["mask_1"]).to(dev)
    for dt in (torch.float16, torch.bfloat16):
        got = T.masked_mean_pool(h.to(dt), m).cpu().numpy()
        np.testing.assert_allclose(got, oenc.masked_mean_pool(h.to(dt).float().cpu().numpy(), z["mask_1"]), rtol=0, atol=1e-6)
    table = torch.zeros((int(z["max_embedding_id"]) + 1, int(z["D"])), device=dev)
    rows = torch.from_numpy(z["pooled_fp16_0"]).to(dev)
    ids = torch.from_numpy(z["emb_ids"]).to(dev)
    for s, e in oenc.iter_batches(10, 4):
        T.scatter_rows(table, rows[s:e], ids[s:e])
    assert np.array_equal(table.cpu().numpy(), z["memmap_table"])
    # repeated id: the later row wins, like the reference's sequential loop
    T.scatter_rows(table, rows[:3], torch.tensor([4, 4, 4], device=dev))
    assert np.array_equal(table[4].cpu().numpy(), z["pooled_fp16_0"][2])


def test_text_encoder_mirror_end_to_end(dev, tmp_path):
    """TextEncoder.encode / encode_to_memmap with the same fake tokenizer + lookup model the golden
    generator drove the reference with (tests/golden/make_golden.py:gen_encode)."""
    from evi_rag_amd import text_encode as T

    z = _load("encode")
    table = torch.from_numpy(z["table"]).to(dev)

    class Tok:
        def __call__(self, texts, padding=True, truncation=True, return_tensors="pt"):
            ids = [[(sum(map(ord, w)) % 97) + 1 for w in t.split()][:8] for t in texts]
            L = max(1, max(len(i) for i in ids))
            input_ids = torch.zeros((len(ids), L), dtype=torch.long)
            mask = torch.zeros((len(ids), L), dtype=torch.long)
            for r, row in enumerate(ids):
                input_ids[r, : len(row)] = torch.tensor(row, dtype=torch.long)
                mask[r, : len(row)] = 1
            return {"input_ids": input_ids, "attention_mask": mask}

    class Model(torch.nn.Module):
        def forward(self, input_ids, attention_mask):
            hid = table[input_ids] + 0.01 * torch.arange(input_ids.size(1), dtype=torch.float32, device=dev).view(1, -1, 1)
            return types.SimpleNamespace(last_hidden_state=hid)

    texts = ["alpha beta", "gamma", "delta epsilon zeta eta", "", "theta iota", "kappa lambda mu", "nu",
             "xi omicron pi rho sigma tau upsilon phi chi psi omega", "alpha", "beta beta beta"]
    enc = T.TextEncoder.from_components(Tok(), Model(), str(dev), fp16=False)
    pooled = enc.encode(texts, batch_size=4)
    assert pooled.device.type == "cpu" and pooled.dtype == torch.float32
    np.testing.assert_allclose(pooled.numpy(), z["pooled_fp16_0"], rtol=0, atol=1e-6)
    assert tuple(enc.encode([], 4).shape) == (0, 0)
    out = tmp_path / "entity_embeddings.pt"
    tensor = T.encode_to_memmap(enc, texts, z["emb_ids"].tolist(), 4, int(z["max_embedding_id"]), out, None, False)
    np.testing.assert_allclose(tensor.numpy(), z["memmap_table"], rtol=0, atol=1e-6)
    assert torch.equal(torch.load(out), tensor) and torch.all(tensor[0] == 0)
    with pytest.raises(ValueError, match="same length"):
        T.encode_to_memmap(enc, texts, [1], 4, 12, out, None, False)


def test_text_encoder_graph_replay_equals_eager(dev):
    """`TextEncoder.use_graphs`: the transformer forward + pooling of each (batch, padded length) shape is captured once in a
    hipGraph and replayed — the same kernels on the same shapes, so the embeddings equal the eager path's (a random-init
    4-layer BERT stands in for the checkpoint that does not exist offline), for f32 and under bf16 autocast, for shapes met
    again (replay) and for new ones (capture), with the cache bounded."""
    transformers = pytest.importorskip("transformers")
    from evi_rag_amd import text_encode as T

    torch.manual_seed(0)
    model = transformers.BertModel(transformers.BertConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=4, num_attention_heads=4,
                                                           intermediate_size=256, max_position_embeddings=64),
                                   add_pooling_layer=False).to(dev).eval()

    class Tok:
        def __call__(self, texts, padding=True, truncation=True, return_tensors="pt"):
            lens = [3 + (len(t) * 7) % 20 for t in texts]
            L = max(lens)
            g = torch.Generator().manual_seed(sum(map(len, texts)))
            mask = (torch.arange(L).view(1, L) < torch.tensor(lens).view(-1, 1)).to(torch.int64)
            ids = torch.randint(1, 1000, (len(texts), L), generator=g) * mask
            return {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}

    texts = ["q" * (1 + (i * 5) % 23) for i in range(50)]
    enc = T.TextEncoder.from_components(Tok(), model, str(dev), fp16=False)
    for autocast in (None, torch.bfloat16):
        enc.autocast = autocast
        enc.use_graphs = False
        eager = enc.encode_to_device(texts, 8)
        enc.use_graphs = True
        assert T.TextEncoder.use_graphs is True and T.TextEncoder.graph_after == 2  # replay is the default; a shape is captured
        first = enc.encode_to_device(texts, 8)    # captures (every shape was met once in the eager pass above)    # the 2nd time it is met
        again = enc.encode_to_device(texts, 8)    # replays
        assert len(enc._graphs) > 0
        assert torch.equal(first, again)
        assert torch.allclose(first, eager, rtol=0, atol=1e-6 if autocast is None else 1e-3), float((first - eager).abs().max())
        other = enc.encode_to_device(texts[::-1], 7)  # new shapes on a warm cache
        enc.use_graphs = False
        assert torch.allclose(other, enc.encode_to_device(texts[::-1], 7), rtol=0, atol=1e-6 if autocast is None else 1e-3)
    enc.use_graphs, enc.max_graphs, enc.graph_after = True, 2, 1
    enc.encode_to_device(texts, 5)  # shapes not met before: every capture evicts down to the bound
    assert len(enc._graphs) <= 2
    # a shape met once is not captured under the default graph_after = 2; a forward that cannot be captured (a host
    # read-back of a device value inside the model) switches the encoder to the eager path with a warning, results intact
    fresh = T.TextEncoder.from_components(Tok(), model, str(dev), fp16=False)
    one = fresh.encode_to_device(texts[:8], 8)
    assert not fresh.__dict__.get("_graphs")
    assert torch.allclose(one, eager_ref(T, Tok(), model, dev, texts[:8]), rtol=0, atol=1e-6)

    class Syncing(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, **kw):
            out = self.inner(**kw)
            float(out.last_hidden_state.sum().item())  # device -> host inside the forward: illegal during capture
            return out

    bad = T.TextEncoder.from_components(Tok(), Syncing(model), str(dev), fp16=False)
    bad.graph_after = 1
    with pytest.warns(RuntimeWarning, match="capture of the encoder forward failed"):
        got = bad.encode_to_device(texts[:16], 8)
    assert bad.use_graphs is False
    assert torch.allclose(got, eager_ref(T, Tok(), model, dev, texts[:16]), rtol=0, atol=1e-6)


def eager_ref(T, tok, model, dev, texts):
    enc = T.TextEncoder.from_components(tok, model, str(dev), fp16=False)
    enc.use_graphs = False
    return enc.encode_to_device(texts, 8)


def test_canonical_edge_selection_matches_reference_golden(dev):
    """C2-C4: keep indices / filtered pair lists produced by the reference's own functions."""
    from evi_rag_amd import labelling as L
    from evi_rag_amd import ops

    z = _load("cosine")
    reln = ops.normalize_embeddings(torch.from_numpy(z["rel"]).to(dev), 1e-6)
    qn = ops.normalize_embeddings(torch.from_numpy(z["q"]).to(dev), 1e-6)
    mask, ids, counts = L.canonicalize_positive_edges(z["edge_src"], z["edge_dst"], z["edge_rel"], z["positive"].tolist(),
                                                      z["pair_ids"].tolist(), z["pair_counts"].tolist(), qn, reln)
    assert sorted(np.nonzero(mask)[0].tolist()) == sorted(z["keep_indices"].tolist())
    assert ids == z["new_pair_ids"].tolist() and counts == z["new_pair_counts"].tolist()


def test_embedding_store_gather(dev):
    """D1: device gather == index_select; out-of-range ids raise like index_select."""
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore

    g = torch.Generator().manual_seed(0)
    ent, rel = torch.randn((1000, 96), generator=g), torch.randn((37, 96), generator=g)
    store = GlobalEmbeddingStore.from_tensors(ent, rel, device=dev)
    ids = torch.randint(0, 1000, (5000,), generator=g)
    assert torch.equal(store.get_entity_embeddings(ids).cpu(), ent.index_select(0, ids))
    rids = torch.randint(0, 37, (77,), generator=g)
    assert torch.equal(store.get_relation_embeddings(rids.to(dev)).cpu(), rel.index_select(0, rids))
    assert store.get_entity_embeddings(torch.empty(0, dtype=torch.long)).shape == (0, 96)
    assert (store.entity_dim, store.relation_dim) == (96, 96)
    with pytest.raises(IndexError):
        store.get_relation_embeddings(torch.tensor([0, 37]))
    b = types.SimpleNamespace(node_embedding_ids=ids[:10], edge_attr=rids[:20])
    store.attach(b)
    assert b.node_embeddings.shape == (10, 96) and b.edge_embeddings.shape == (20, 96) and b.num_relations == 37


def test_large_graph_takes_the_global_memory_paths(dev):
    """A 20 000-node graph exceeds the LDS capacity of the CSR counters (6 144 nodes) and of the BFS levels
    (12 288 nodes): both kernels must fall back to their global-memory paths with identical results; a hub
    of degree 5 000 exercises the wave-cooperative frontier expansion on that path too."""
    from evi_rag_amd import labelling as L

    rng = np.random.default_rng(31)
    n, e = 20000, 60000
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, n, size=e)
    src[:5000] = 17  # hub
    small_n, small_e = 300, 900
    s2, d2 = rng.integers(0, small_n, size=small_e), rng.integers(0, small_n, size=small_e)
    gb = L.GraphBatch([n, small_n], [src, s2], [dst, d2])  # one graph on each path, same launch
    seeds = [[5, 17, 19999], [0, 7]]
    for mode, directed in ((0, False), (1, True)):
        got = L.bfs_dist_batch(gb, seeds, mode=mode)
        for g, (nn, a, b) in enumerate(((n, src, dst), (small_n, s2, d2))):
            adj = (ograph.build_directed_adjacency if directed else ograph.build_undirected_adjacency)(nn, a.tolist(), b.tolist())
            assert got[g].tolist() == ograph.bfs_dist(nn, adj, seeds[g]), (g, mode)
    # CSR rows (as sets: row order is unspecified) against the oracle adjacency
    csr = gb.csr
    out_ptr, out_nbr = csr.out_ptr.cpu().numpy(), csr.out_nbr.cpu().numpy()
    dadj = ograph.build_directed_adjacency(n, src.tolist(), dst.tolist())
    for v in (0, 17, 4242, n - 1):
        assert sorted(out_nbr[out_ptr[v]: out_ptr[v + 1]].tolist()) == dadj[v]
    res = L.shortest_path_single_batch(gb, [[5], [0]], [[19999, 123], [250]])
    for g, (nn, a, b, s, t) in enumerate(((n, src, dst, [5], [19999, 123]), (small_n, s2, d2, [0], [250]))):
        ref = ograph.shortest_path_single(nn, a.tolist(), b.tolist(), s, t)
        assert res[g] == (list(ref[0]), list(ref[1])), g


def test_encode_tables_sequence_matches_the_reference_pipeline(dev, tmp_path):
    """E4: the tables the REFERENCE's `preprocess` wrote (entity_embeddings.pt, relation_embeddings.pt, the question_emb
    column) for a synthetic raw split, from the vocabulary records it wrote — tests/golden/make_golden.py:gen_encode_tables
    ran it with the lookup encoder — against `encode_tables` / `encode_questions` driven by the same records here."""
    import gzip
    import json

    from evi_rag_amd import text_encode as T

    with gzip.open(os.path.join(GOLD, "encode_tables.json.gz"), "rb") as fh:
        z = json.loads(fh.read().decode())
    table = torch.tensor(z["lookup_table"], device=dev)

    class Tok:
        def __call__(self, texts, padding=True, truncation=True, return_tensors="pt"):
            ids = [[(sum(map(ord, w)) % 97) + 1 for w in t.split()][:8] for t in texts]
            L = max(1, max(len(i) for i in ids))
            input_ids = torch.zeros((len(ids), L), dtype=torch.long)
            mask = torch.zeros((len(ids), L), dtype=torch.long)
            for r, row in enumerate(ids):
                input_ids[r, : len(row)] = torch.tensor(row, dtype=torch.long)
                mask[r, : len(row)] = 1
            return {"input_ids": input_ids, "attention_mask": mask}

    class Model(torch.nn.Module):
        def forward(self, input_ids, attention_mask):
            hid = table[input_ids] + 0.01 * torch.arange(input_ids.size(1), dtype=torch.float32, device=dev).view(1, -1, 1)
            return types.SimpleNamespace(last_hidden_state=hid)

    enc = T.TextEncoder.from_components(Tok(), Model(), str(dev), fp16=False)
    # the records arrive in vocabulary (first-seen) order, not sorted by id: the sort is part of what is pinned
    assert [r["embedding_id"] for r in z["embedding_vocab"]] != sorted(r["embedding_id"] for r in z["embedding_vocab"])
    ent, rel = T.encode_tables(enc, entity_embedding_records=z["embedding_vocab"], entity_struct_records=z["entity_vocab"],
                               relation_records=z["relation_vocab"], batch_size=z["batch_size"], embeddings_out_dir=tmp_path / "emb")
    want_ent, want_rel = np.asarray(z["entity_embeddings"], np.float32), np.asarray(z["relation_embeddings"], np.float32)
    assert ent.shape == want_ent.shape and rel.shape == want_rel.shape and ent.dtype == torch.float32
    np.testing.assert_allclose(ent.numpy(), want_ent, rtol=0, atol=1e-6)
    np.testing.assert_allclose(rel.numpy(), want_rel, rtol=0, atol=1e-6)
    zero_rows = np.nonzero(np.abs(want_ent).sum(1) == 0)[0]
    assert 0 in zero_rows and np.all(ent.numpy()[zero_rows] == 0)  # the non-text placeholder row stays exactly zero
    assert torch.equal(torch.load(tmp_path / "emb" / "entity_embeddings.pt"), ent)
    assert torch.equal(torch.load(tmp_path / "emb" / "relation_embeddings.pt"), rel)
    # questions: per chunk of parquet_chunk_size samples (the golden run used the reference's minimum chunk size)
    qs = z["questions"]
    got = T.encode_questions(enc, [q["question"] for q in qs], z["batch_size"], chunk_size=3)
    assert len(got) == len(qs) and all(isinstance(v, float) for v in got[0])
    np.testing.assert_allclose(np.asarray(got, np.float32), np.asarray([q["question_emb"] for q in qs], np.float32), rtol=0, atol=1e-6)
