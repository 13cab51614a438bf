"""CPU-side checks of the host mirror of the reference interface (no compute calls)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_retriever_state_dict_matches_reference_checkpoint_layout():
    """Keys, order and shapes of the reference Retriever's state_dict (fixture written by the
    reference's own module) — eval loads checkpoints with strict=True (src/eval.py:111)."""
    from evi_rag_amd.retriever import Retriever

    z = np.load(os.path.join(GOLD, "retriever_mid.npz"), allow_pickle=False)
    rounds = z["rounds"].tolist()
    m = Retriever(emb_dim=int(z["D"]), hidden_dim=int(z["H"]),
                  dde_cfg={"num_rounds": rounds[0], "num_reverse_rounds": rounds[1]},
                  hide_seek_cfg={"enabled": True, "p_near": 0.7, "p_far": 0.1, "bias_near": -2.0, "bias_far": -0.5,
                                 "apply_in_eval": False}, some_future_kwarg=1)
    sd = m.state_dict()
    assert list(sd.keys()) == z["state_dict_keys"].tolist()
    for k, v in sd.items():
        assert tuple(v.shape) == z["w_" + k].shape, k
    assert sd["parity_meta"].tolist() == z["w_parity_meta"].tolist()
    m.load_state_dict({k: torch.from_numpy(z["w_" + k]) for k in sd}, strict=True)


def test_same_seed_gives_the_reference_initialisation():
    """Construction order mirrors the reference, so torch.manual_seed reproduces its random init."""
    from evi_rag_amd.retriever import Retriever

    z = np.load(os.path.join(GOLD, "retriever_fwd.npz"), allow_pickle=False)
    torch.manual_seed(2)  # tests/golden/make_golden.py: gen_retriever(..., seed=2)
    m = Retriever(emb_dim=16, hidden_dim=16, dde_cfg={"num_rounds": 2, "num_reverse_rounds": 2})
    # 2-D parameters were left at their seeded init by the generator (1-D ones were perturbed)
    for k in ("entity_proj.network.0.weight", "state_net.0.weight", "score_head.weight", "non_text_entity_emb.weight"):
        assert np.array_equal(m.state_dict()[k].numpy(), z["w_" + k]), k


def test_cpu_module_refuses_to_run():
    from evi_rag_amd import synthetic
    from evi_rag_amd.retriever import Retriever

    m = Retriever(emb_dim=16, hidden_dim=16).eval()
    sb = synthetic.make_batch(2, nodes_per_graph=10, edges_per_graph=12, emb_dim=16, seed=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(synthetic.as_namespace(sb))


def test_synthetic_batch_is_a_valid_pyg_style_batch():
    from evi_rag_amd import synthetic
    from oracle import graph as ograph

    sb = synthetic.make_batch(8, nodes_per_graph=50, edges_per_graph=120, emb_dim=8, seed=4)
    eb, eptr = ograph.compute_edge_batch(sb.edge_index, sb.ptr, sb.num_graphs)  # raises if malformed
    assert np.array_equal(eptr, sb.edge_ptr)
    assert sb.topic_one_hot.sum(axis=1).tolist() == [1.0] * sb.num_nodes
    for g in range(sb.num_graphs):  # no duplicate (h, r, t) inside a graph
        lo, hi = sb.edge_ptr[g], sb.edge_ptr[g + 1]
        trip = np.stack([sb.edge_index[0, lo:hi], sb.edge_attr[lo:hi], sb.edge_index[1, lo:hi]], 1)
        assert np.unique(trip, axis=0).shape[0] == hi - lo


def test_write_packed_layout_and_validation(tmp_path):
    """The flat split writer (host side of the HBM-resident dataset): pointer arrays, optional aux
    fields, and the reference's fail-fast KeyError on a missing core key."""
    import json

    from evi_rag_amd import packed_dataset as pd, synthetic

    base = synthetic.make_batch(5, nodes_per_graph=20, edges_per_graph=40, emb_dim=8, num_relations=4, seed=1)
    samples = pd.samples_from_flat_batch(base)
    samples[2].update(pair_start_node_locals=[1, 2], pair_answer_node_locals=[3, 4], pair_edge_counts=[1, 0], pair_edge_local_ids=[7])
    meta = pd.write_packed(tmp_path / "s", samples)
    assert meta["num_samples"] == 5 and meta["emb_dim"] == 8 and meta["num_topics"] == 2
    assert json.loads((tmp_path / "s" / "meta.json").read_text())["sample_ids"] == meta["sample_ids"]
    assert np.array_equal(np.load(tmp_path / "s" / "ptr_node.npy"), base.ptr)
    assert np.array_equal(np.load(tmp_path / "s" / "ptr_edge.npy"), base.edge_ptr)
    assert np.load(tmp_path / "s" / "ptr_pair.npy").tolist() == [0, 0, 0, 2, 2, 2]
    assert np.load(tmp_path / "s" / "pair_shortest_lengths.npy").tolist() == [-1, -1]  # absent optional field
    assert np.array_equal(np.load(tmp_path / "s" / "edge_src.npy") + np.repeat(base.ptr[:-1], np.diff(base.edge_ptr)), base.edge_index[0])
    bad = dict(samples[0])
    del bad["topic_one_hot"]
    with pytest.raises(KeyError, match="missing key: topic_one_hot"):
        pd.write_packed(tmp_path / "bad", [bad])
    bad = dict(samples[0], edge_index=samples[0]["edge_index"] + 1000)
    with pytest.raises(ValueError, match="out of range"):
        pd.write_packed(tmp_path / "bad2", [bad])


def test_cosine_schedule_equals_torch_cosine_annealing():
    """train.CosineSchedule (closed form, stepped per epoch) against torch.optim.lr_scheduler.CosineAnnealingLR — the
    scheduler the reference builds (src/models/retriever_module.py:341-354; t_max 200, eta_min 1e-6 in
    configs/model/retriever_module.yaml:42-47)."""
    import types

    import torch

    from evi_rag_amd.train import CosineSchedule

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, eta_min=1e-6)
    ours_opt = types.SimpleNamespace(lr=1e-3, initial_lr=1e-3)
    ours = CosineSchedule(ours_opt, t_max=20, eta_min=1e-6)
    for _ in range(20):
        opt.step()
        ref.step()
        ours.step()
        assert abs(ours_opt.lr - ref.get_last_lr()[0]) < 1e-12


def test_lmdb_to_packed_reading_loop(tmp_path, monkeypatch):
    """tools/lmdb_to_packed.py against a stand-in for the `lmdb` package (absent from the build image) with the same
    open / begin / cursor interface: samples come out in key order with the key as `sample_id`, metadata records and
    non-sample values are skipped, torch tensors (what the reference pickles) are accepted, `--limit` stops early."""
    import importlib.util
    import pickle
    import sys
    import types

    import torch

    from evi_rag_amd import packed_dataset as pd, synthetic

    base = synthetic.make_batch(4, nodes_per_graph=12, edges_per_graph=30, emb_dim=8, num_relations=4, seed=3)
    samples = pd.samples_from_flat_batch(base)
    records = {}
    for i, s in enumerate(samples):
        s = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in s.items()
             if k not in ("sample_id", "question", "seed_entity_ids")}  # the core record (:2196-2209); the rest lives in the aux LMDB
        records[f"q{i:03d}".encode()] = pickle.dumps(s)
    records[b"__meta__"] = pickle.dumps({"version": 1})
    records[b"zzz_not_a_sample"] = pickle.dumps([1, 2, 3])

    class _Txn:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def cursor(self):
            return iter(sorted(records.items()))

    class _Env:
        def begin(self, write=False):
            return _Txn()

        def close(self):
            pass

    aux_records = {f"q{i:03d}".encode(): pickle.dumps({"question": f"who is {i}?", "seed_entity_ids": torch.tensor([i])}) for i in range(4)}

    class _AuxTxn(_Txn):
        def cursor(self):
            return iter(sorted(aux_records.items()))

    class _AuxEnv(_Env):
        def begin(self, write=False):
            return _AuxTxn()

    (tmp_path / "test.aux.lmdb").mkdir()  # the pipeline writes the aux records beside the core ones (<split>.aux.lmdb)
    fake = types.ModuleType("lmdb")
    fake.open = lambda path, **kw: _AuxEnv() if str(path).endswith(".aux.lmdb") else _Env()
    monkeypatch.setitem(sys.modules, "lmdb", fake)
    spec = importlib.util.spec_from_file_location("lmdb_to_packed", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lmdb_to_packed.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    meta = mod.main(["--lmdb", str(tmp_path / "test.lmdb")])
    assert meta["num_samples"] == 4 and meta["sample_ids"] == ["q000", "q001", "q002", "q003"]
    import json

    assert json.loads((tmp_path / "test.packed" / "meta.json").read_text())["questions"][2] == "who is 2?"  # merged from the aux records
    assert np.array_equal(np.load(tmp_path / "test.packed" / "ptr_node.npy"), base.ptr)
    assert np.array_equal(np.load(tmp_path / "test.packed" / "ptr_edge.npy"), base.edge_ptr)
    assert mod.main(["--lmdb", str(tmp_path / "test.lmdb"), "--out", str(tmp_path / "two"), "--limit", "2"])["num_samples"] == 2
    # the reader's unpickler admits the sample types only: a record that names another callable is refused, not executed
    import pickle as _pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))

    with pytest.raises(_pickle.UnpicklingError, match="refusing to unpickle"):
        mod.loads_sample(_pickle.dumps({"edge_index": Evil()}))
    ok = mod.loads_sample(_pickle.dumps({"a": torch.arange(3), "b": np.arange(4, dtype=np.int32), "c": [1, 2.5, "x", None], "d": np.float32(2)}))
    assert ok["a"].tolist() == [0, 1, 2] and ok["b"].tolist() == [0, 1, 2, 3] and float(ok["d"]) == 2.0


def test_records_to_samples_is_the_lmdb_free_hand_over(tmp_path):
    """graph_build.records_to_samples: GraphRecords -> the sample dictionaries of the reference's materialisation stage
    (scripts/build_retrieval_pipeline.py:2141-2224) -> write_packed, without an LMDB in between: local seed / answer indices
    by position in node_entity_ids (ids outside the graph dropped), topic one-hot on the seeds, labels from the positive mask,
    the reference's fail-fast errors."""
    from evi_rag_amd import packed_dataset as pd
    from evi_rag_amd.graph_build import GraphRecord, records_to_samples

    g0 = GraphRecord(graph_id="g0", node_entity_ids=[50, 7, 19, 3], node_embedding_ids=[5, 0, 2, 9], node_labels=["a", "b", "c", "d"],
                     edge_src=[0, 1, 2], edge_dst=[1, 2, 3], edge_relation_ids=[4, 4, 1], positive_triple_mask=[True, False, True],
                     pair_start_node_locals=[0], pair_answer_node_locals=[3], pair_edge_local_ids=[0, 2], pair_edge_counts=[2],
                     pair_shortest_lengths=[3])
    g1 = GraphRecord(graph_id="g1", node_entity_ids=[8, 9], node_embedding_ids=[1, 1], node_labels=["x", "y"], edge_src=[1], edge_dst=[0],
                     edge_relation_ids=[0], positive_triple_mask=[False], pair_start_node_locals=[], pair_answer_node_locals=[],
                     pair_edge_local_ids=[], pair_edge_counts=[], pair_shortest_lengths=[])
    qe = np.arange(16, dtype=np.float32).reshape(2, 8)
    samples = records_to_samples([g0, g1], [[50, 777], [9]], [[3, 19, 123456], [8]], qe, questions=["q zero", "q one"])
    s0, s1 = samples
    assert s0["q_local_indices"].tolist() == [0] and s0["a_local_indices"].tolist() == [3, 2]  # 777 / 123456 are not in the graph
    assert s0["answer_entity_ids"].tolist() == [3, 19, 123456] and s0["answer_entity_ids_len"].tolist() == [3]
    assert s0["topic_one_hot"].tolist() == [[0, 1], [1, 0], [1, 0], [1, 0]] and s1["topic_one_hot"].tolist() == [[1, 0], [0, 1]]
    assert s0["labels"].tolist() == [1.0, 0.0, 1.0] and s0["edge_index"].tolist() == [[0, 1, 2], [1, 2, 3]]
    assert s0["question_emb"].shape == (1, 8) and s1["question"] == "q one" and s0["seed_entity_ids"].tolist() == [50, 777]
    meta = pd.write_packed(tmp_path / "p", samples)
    assert meta["num_samples"] == 2 and meta["sample_ids"] == ["g0", "g1"] and meta["emb_dim"] == 8
    assert np.load(tmp_path / "p" / "ptr_node.npy").tolist() == [0, 4, 6] and np.load(tmp_path / "p" / "ptr_edge.npy").tolist() == [0, 3, 4]
    assert np.load(tmp_path / "p" / "pair_shortest_lengths.npy").tolist() == [3]
    empty = GraphRecord(graph_id="e", node_entity_ids=[1], node_embedding_ids=[1], node_labels=["z"], edge_src=[], edge_dst=[],
                        edge_relation_ids=[], positive_triple_mask=[], pair_start_node_locals=[], pair_answer_node_locals=[],
                        pair_edge_local_ids=[], pair_edge_counts=[], pair_shortest_lengths=[])
    with pytest.raises(ValueError, match="empty edge_index is unsupported"):
        records_to_samples([empty], [[1]], [[1]], qe[:1])
    with pytest.raises(ValueError, match="seed_entity_ids is null"):
        records_to_samples([g1], [None], [[8]], qe[:1])


def test_trainer_precision_maps_to_matmul_precision():
    """`trainer.precision` of the reference's trainer configs (configs/trainer/default.yaml:14 `16-mixed`, predict.yaml:8 `32-true`)."""
    from evi_rag_amd.train import matmul_precision_for

    assert matmul_precision_for("bf16-mixed") == "bf16"
    assert matmul_precision_for("32-true") == "split" and matmul_precision_for(32) == "split"
    assert matmul_precision_for("16-mixed") == "split"  # f16 autocast is not mirrored; the split products are more precise
    with pytest.raises(ValueError, match="precision"):
        matmul_precision_for("fp8-mixed")


def test_packed_loader_shuffle_is_a_function_of_seed_and_epoch():
    """A run resumed at epoch e must shuffle like the uninterrupted run did (PackedLoader.set_epoch; the trainer calls it with
    its current epoch): the permutation depends on (random_seed, epoch) only, not on how many epochs this process has drawn."""
    from evi_rag_amd.packed_dataset import PackedLoader

    class Ds:
        def __len__(self):
            return 37

    a = PackedLoader(Ds(), batch_size=4, shuffle=True, random_seed=11)
    orders = []
    for e in range(3):
        a.set_epoch(e)
        orders.append(a._order().tolist())
    assert orders[0] != orders[1] != orders[2] and all(sorted(o) == list(range(37)) for o in orders)
    b = PackedLoader(Ds(), batch_size=4, shuffle=True, random_seed=11)  # a fresh process resuming at epoch 2
    b.set_epoch(2)
    assert b._order().tolist() == orders[2]
    c = PackedLoader(Ds(), batch_size=4, shuffle=True, random_seed=12)
    c.set_epoch(2)
    assert c._order().tolist() != orders[2]
    # rank shares of one epoch are disjoint and cover the split
    shares = []
    for r in range(3):
        d = PackedLoader(Ds(), batch_size=4, shuffle=True, random_seed=11, rank=r, world_size=3)
        d.set_epoch(1)
        shares += d._order().tolist()
    assert sorted(shares) == list(range(37))
