"""GPU parity: the HBM-resident split and its device collation vs the collate oracle."""
import numpy as np
import pytest
import torch

from oracle import collate as ocollate

pytestmark = pytest.mark.gpu


def _split(tmp_path, graphs=24, seed=4):
    from evi_rag_amd import packed_dataset as pd, synthetic

    base = synthetic.make_batch(graphs, nodes_per_graph=90, edges_per_graph=260, emb_dim=16, num_relations=12, seed=seed)
    samples = pd.samples_from_flat_batch(base)
    rng = np.random.default_rng(seed)
    for i, s in enumerate(samples):  # ragged pair supervision on some samples, none on others
        if i % 3:
            npair = int(rng.integers(1, 4))
            cnt = rng.integers(0, 5, npair)
            s.update(pair_start_node_locals=rng.integers(0, s["num_nodes"], npair), pair_answer_node_locals=rng.integers(0, s["num_nodes"], npair),
                     pair_edge_counts=cnt, pair_shortest_lengths=rng.integers(1, 4, npair),
                     pair_edge_local_ids=rng.integers(0, s["edge_index"].shape[1], int(cnt.sum())))
    meta = pd.write_packed(tmp_path / "split", samples)
    return base, samples, meta


def _check(batch, ref):
    for key in ("edge_index", "edge_attr", "labels", "node_global_ids", "node_embedding_ids", "topic_one_hot", "q_local_indices",
                "a_local_indices", "answer_entity_ids", "seed_entity_ids", "pair_start_node_locals", "pair_answer_node_locals",
                "pair_edge_counts", "pair_shortest_lengths", "pair_edge_local_ids", "ptr", "edge_ptr", "batch", "question_emb"):
        got = getattr(batch, key).cpu().numpy()
        want = np.asarray(ref[key])
        assert got.shape == want.reshape(got.shape).shape and np.array_equal(got, want.reshape(got.shape)), key
    assert np.array_equal(batch.edge_batch.cpu().numpy(), np.repeat(np.arange(len(ref["ptr"]) - 1), np.diff(ref["edge_ptr"])))
    sd = batch._slice_dict
    assert np.array_equal(sd["q_local_indices"].cpu().numpy(), ref["slices"]["q_local_indices"])
    assert np.array_equal(sd["answer_entity_ids"].cpu().numpy(), ref["slices"]["answer_entity_ids"])
    assert np.array_equal(batch.answer_entity_ids_ptr.cpu().numpy(), ref["slices"]["answer_entity_ids"])
    assert np.array_equal(sd["pair_edge_local_ids"].cpu().numpy(), ref["slices"]["pair_edge_local_ids"])


def test_collate_matches_oracle_and_flat_batch(dev, tmp_path):
    from evi_rag_amd import packed_dataset as pd

    base, samples, meta = _split(tmp_path)
    ds = pd.PackedRetrievalDataset(tmp_path / "split", device=dev)
    assert len(ds) == 24 and meta["emb_dim"] == 16 and ds.nbytes() > 0
    # in-order, the whole split: the flat batch it was cut from
    full = ds.collate(list(range(24)))
    assert np.array_equal(full.edge_index.cpu().numpy(), base.edge_index) and np.array_equal(full.ptr.cpu().numpy(), base.ptr)
    assert np.array_equal(full.q_local_indices.cpu().numpy(), base.q_local_indices)
    _check(full, ocollate.collate(samples))
    # arbitrary order with repeats
    for ids in ([5], [23, 0, 7, 7, 12], list(np.random.default_rng(1).permutation(24)[:17])):
        _check(ds.collate(ids), ocollate.collate([samples[i] for i in ids]))
    with pytest.raises(IndexError):
        ds.collate([0, 24])
    with pytest.raises(ValueError):
        ds.collate([])
    meta1 = ds.load_sample("sample_3") if "sample_3" in ds.sample_ids else ds.load_sample(ds.sample_ids[3])
    assert torch.equal(meta1["seed_entity_ids"], torch.from_numpy(np.asarray(samples[3]["seed_entity_ids"])))
    with pytest.raises(KeyError):
        ds.load_sample("nope")


def test_loader_feeds_retriever_and_metrics(dev, tmp_path):
    """Packed split -> loader -> embedding attach -> Retriever.forward -> metrics == the same on the flat batch."""
    from evi_rag_amd import metrics as M, packed_dataset as pd, synthetic
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore
    from evi_rag_amd.retriever import Retriever

    base, samples, _ = _split(tmp_path, graphs=12, seed=9)
    rng = np.random.default_rng(0)
    ent = torch.from_numpy(rng.standard_normal((int(base.node_embedding_ids.max()) + 1, 16)).astype(np.float32))
    rel = torch.from_numpy(rng.standard_normal((12, 16)).astype(np.float32))
    store = GlobalEmbeddingStore.from_tensors(ent, rel, device=dev)
    ds = pd.PackedRetrievalDataset(tmp_path / "split", device=dev, embeddings=store)
    loader = pd.PackedLoader(ds, batch_size=5)
    assert len(loader) == 3
    torch.manual_seed(0)
    model = Retriever(emb_dim=16, hidden_dim=16).to(dev).eval()
    coll = M.RetrieverMetricCollection(k_values=[1, 5, 10])
    seen = 0
    for batch in loader:
        out = model(batch)
        assert out.logits.shape[0] == batch.edge_index.shape[1]
        coll.update(preds=out.logits, target=batch.labels > 0.5, indexes=out.query_ids, batch=batch, query_ids=out.query_ids,
                    num_graphs=batch.num_graphs)
        seen += batch.num_graphs
    assert seen == 12
    got = {k: float(v) for k, v in coll.compute().items()}
    # the same 12 graphs as ONE batch
    one = ds.collate(list(range(12)))
    out = model(one)
    ref = M.RetrieverMetricCollection(k_values=[1, 5, 10])
    ref.update(preds=out.logits, target=one.labels > 0.5, indexes=out.query_ids, batch=one, query_ids=out.query_ids, num_graphs=12)
    for k, v in ref.compute().items():
        assert abs(got[k] - float(v)) < 1e-6, k
    # shuffled, sharded loaders partition the split
    parts = [pd.PackedLoader(ds, batch_size=4, shuffle=True, random_seed=3, rank=r, world_size=2) for r in range(2)]
    ids = sorted(i for p in parts for b in p for i in b.idx.cpu().tolist())
    assert ids == list(range(12))


def test_segment_offsets_kernel_matches_numpy(dev):
    """evi_segment_offsets (device-side batch offsets for sample ids that live on the device)."""
    from evi_rag_amd import _lib, ops

    rng = np.random.default_rng(3)
    lens = rng.integers(0, 50, 400)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ptr_d = torch.from_numpy(ptr).to(dev)
    for B in (1, 7, 1024, 3000):
        ids = rng.integers(0, 400, B).astype(np.int64)
        ids_d = torch.from_numpy(ids).to(dev)
        out = torch.empty(B + 1, dtype=torch.int64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.load().evi_segment_offsets(ptr_d.data_ptr(), 400, ids_d.data_ptr(), B, out.data_ptr(), status.data_ptr(),
                                                   ops._stream(dev)))
        assert np.array_equal(out.cpu().numpy(), np.concatenate([[0], np.cumsum(lens[ids])])), B
        assert int(status.item()) == 0
    bad = torch.tensor([3, 400], dtype=torch.int64, device=dev)
    out = torch.empty(3, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().evi_segment_offsets(ptr_d.data_ptr(), 400, bad.data_ptr(), 2, out.data_ptr(), status.data_ptr(),
                                               ops._stream(dev)))
    assert int(status.item()) == 1


def test_deferred_embedding_id_check(dev, tmp_path):
    from evi_rag_amd import packed_dataset as pd
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore

    base, samples, _ = _split(tmp_path, graphs=4, seed=2)
    small = GlobalEmbeddingStore.from_tensors(torch.zeros(3, 16), torch.zeros(12, 16), device=dev)  # entity table too small
    ds = pd.PackedRetrievalDataset(tmp_path / "split", device=dev, embeddings=small)
    ds.collate([0, 1])  # does not raise here ...
    with pytest.raises(IndexError):
        ds.check_deferred()  # ... but the epoch-end check does
    ds.check_deferred()  # flag cleared


def test_relation_table_rides_along_and_edge_embeddings_are_lazy(dev, tmp_path):
    """`GlobalEmbeddingStore.attach` hands the relation TABLE to the batch; the Retriever (relation de-duplication on) reads one
    row per relation from it (EviRetrieverBatch.relation_rows) and the [E, D] `edge_embeddings` the reference's collater attaches
    are gathered only when somebody reads them.  Same outputs as the gathered path; a relation id outside the table is still
    reported (by the forward's range check instead of the gather's)."""
    from evi_rag_amd import packed_dataset as pd
    from evi_rag_amd.embedding_store import GlobalEmbeddingStore
    from evi_rag_amd.retriever import Retriever

    base, samples, _ = _split(tmp_path, graphs=6, seed=9)
    rng = np.random.default_rng(3)
    ent = torch.from_numpy(rng.standard_normal((int(base.node_embedding_ids.max()) + 1, 16)).astype(np.float32))
    rel = torch.from_numpy(rng.standard_normal((12, 16)).astype(np.float32))
    store = GlobalEmbeddingStore.from_tensors(ent, rel, device=dev)
    ds = pd.PackedRetrievalDataset(tmp_path / "split", device=dev, embeddings=store)
    torch.manual_seed(2)
    model = Retriever(emb_dim=16, hidden_dim=24).to(dev).eval()
    batch = ds.collate(list(range(6)))
    assert "edge_embeddings" not in vars(batch) and batch.relation_embedding_table.shape == (12, 16) and batch.num_relations == 12
    out = model(batch)
    assert "edge_embeddings" not in vars(batch)  # the forward did not need them
    gathered = batch.edge_embeddings              # first read: gathered now, then kept
    assert "edge_embeddings" in vars(batch) and torch.equal(gathered, rel.to(dev)[batch.edge_attr])
    # the gathered path (no table on the batch): identical outputs
    plain = ds.collate(list(range(6)))
    _ = plain.edge_embeddings
    del plain.relation_embedding_table
    ref = model(plain)
    assert torch.equal(out.logits, ref.logits) and torch.equal(out.edge_embeddings, ref.edge_embeddings)
    # training: gradients equal too (the relation_proj weight gradient multiplies the table's rows instead of gathered copies)
    grads = []
    for b in (ds.collate(list(range(6))), plain):
        model.differentiable = True
        model.zero_grad(set_to_none=True)
        model(b).logits.sum().backward()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters()})
    model.differentiable = None
    for n, g in grads[0].items():
        assert float((g - grads[1][n]).abs().max()) <= 1e-6 * (float(g.abs().max()) + 1e-9) + 1e-8, n
    # a relation id beyond the table: scored clamped, reported by check_deferred (the reference raises at its gather)
    bad = ds.collate([0, 1])
    bad.edge_attr = bad.edge_attr.clone()
    bad.edge_attr[3] = 12
    model(bad)
    with pytest.raises(IndexError):
        model.check_deferred()
