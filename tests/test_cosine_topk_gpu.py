"""GPU parity: evi_cosine_topk / evi_row_normalize / evi_topk_merge / evi_segment_topk vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import cosine as ocos
from oracle import ranking as orank
from tests.helpers import check_topk_against_scores

pytestmark = pytest.mark.gpu

EPS = 1e-6


def _make_index(n, d, seed, dup_frac=0.01):
    """N(0,1) rows, row 0 all-zero (eps clamp), 1 % duplicated rows (ties)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    if n > 0:
        x[0] = 0.0
    ndup = int(n * dup_frac)
    if ndup > 0 and n > 2:
        src = rng.integers(1, n, size=ndup)
        dst = rng.integers(1, n, size=ndup)
        x[dst] = x[src]
    return x


@pytest.mark.parametrize("n,d", [(1, 16), (5, 64), (1000, 384), (4097, 768)])
def test_row_normalize_matches_oracle(dev, n, d):
    from evi_rag_amd import ops

    x = _make_index(n, d, seed=n + d)
    got = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS).cpu().numpy()
    ref = ocos.normalize_embeddings(x, EPS)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)
    assert np.all(got[0] == 0.0)  # zero rows stay zero rows (clamp on the norm)
    inv = ops.row_inv_norm(torch.from_numpy(x).to(dev), EPS).cpu().numpy()
    np.testing.assert_allclose(inv, ocos.row_inv_norm(x, EPS), rtol=2e-6)
    assert inv[0] == np.float32(1.0 / EPS)


def test_row_normalize_empty_passthrough(dev):
    from evi_rag_amd import ops

    e = torch.empty((0, 0), device=dev)
    assert ops.normalize_embeddings(e, EPS) is e  # reference :834-835


@pytest.mark.parametrize(
    "Q,N,D,k",
    [
        (32, 1000, 384, 100),    # BASELINE config 1 (toy)
        (1, 1, 16, 1),           # smallest
        (3, 7, 16, 10),          # N < k: padding
        (16, 300, 64, 500),      # one query block, k > N
        (17, 5000, 128, 50),     # two query blocks, ragged
        (33, 70000, 768, 500),   # > 32 queries: two passes; > first dense segment
        (32, 300000, 768, 500),  # dense first segment + filtered segments
        (8, 150000, 1024, 2048), # max k, radix-select path in later segments
        (5, 12345, 1280, 7),     # max D, N not a multiple of 16
        (4, 10000, 48, 64),      # D/16 odd (U = 1 path)
    ],
)
def test_cosine_topk_matches_oracle(dev, Q, N, D, k):
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N * 7 + D)
    rng = np.random.default_rng(Q + 13)
    q = rng.standard_normal((Q, D), dtype=np.float32)
    if Q > 2:
        q[1] = 0.0  # a zero query: all scores tie at 0 -> pure id order
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    sc, ids = ops.cosine_topk(qn, xn, k, row_id_base=1000)
    torch.cuda.synchronize()
    ref_full = ocos.cosine_scores(q, x, EPS)  # reference arithmetic: normalise, then matmul
    check_topk_against_scores(sc.cpu().numpy(), ids.cpu().numpy(), ref_full, k, id_base=1000)
    if Q > 2:
        m = min(k, N)
        assert np.array_equal(ids[1, :m].cpu().numpy(), np.arange(m) + 1000)


@pytest.mark.parametrize("Q,N,D,k", [(32, 200000, 768, 500), (5, 1000, 64, 50), (17, 70001, 1024, 100)])
def test_cosine_topk_f16_index_matches_oracle(dev, Q, N, D, k):
    """f16-storage index (BASELINE config 4): scores == f32 queries . f16-rounded rows, to ~1e-6."""
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N + D)
    q = np.random.default_rng(Q).standard_normal((Q, D), dtype=np.float32)
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    x16 = xn.to(torch.float16)
    sc, ids = ops.cosine_topk(qn, x16, k, row_id_base=7)
    ref_full = (qn.cpu().numpy().astype(np.float64) @ x16.cpu().numpy().astype(np.float64).T).astype(np.float32)
    check_topk_against_scores(sc.cpu().numpy(), ids.cpu().numpy(), ref_full, k, id_base=7, score_tol=2e-6)
    # and the f16 result stays within the f16 rounding of the f32-index result
    s32, i32 = ops.cosine_topk(qn, xn, k, row_id_base=7)
    assert float((sc[:, 0] - s32[:, 0]).abs().max()) < 1e-3
    with pytest.raises(NotImplementedError):
        ops.cosine_topk(qn[:, :48].contiguous(), x16[:, :48].contiguous(), 5)  # D % 32 != 0


def test_quantize_rows_fp8_matches_oracle(dev):
    from evi_rag_amd import ops

    x = _make_index(3000, 192, seed=21)
    x[5, :8] = [1e-9, -1e-9, 3.0, -3.0, 0.0, 1e-3, -1e-3, 2.9999]  # subnormal codes, signed zeros
    codes, scale = ops.quantize_rows_fp8(torch.from_numpy(x).to(dev))
    ref_c, ref_s = ocos.quantize_rows_e4m3(x)
    np.testing.assert_array_equal(scale.cpu().numpy(), ref_s)
    got = codes.cpu().numpy()
    # -0 and +0 are the same value: compare the decoded values, then the codes away from zero
    tab = ocos.e4m3_decode_table()
    np.testing.assert_array_equal(tab[got], tab[ref_c])
    assert not np.isnan(tab[got]).any()
    assert np.abs(tab[got]).max() == 448.0 and scale[0].item() == 1.0


@pytest.mark.parametrize("Q,N,D,k", [(32, 200000, 768, 500), (5, 1000, 64, 50), (17, 70001, 1024, 100)])
def test_cosine_topk_fp8_index_matches_oracle(dev, Q, N, D, k):
    """e4m3-storage index (BASELINE config 5): exact w.r.t. the dequantised rows; overlap@k vs f32."""
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N + D + 1)
    q = np.random.default_rng(Q + 2).standard_normal((Q, D), dtype=np.float32)
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    codes, scale = ops.quantize_rows_fp8(xn)
    sc, ids = ops.cosine_topk(qn, codes, k, row_scale=scale, row_id_base=7, fp8_mfma=False)  # widening variant: f32 query kept exact
    deq = ocos.e4m3_decode_table()[codes.cpu().numpy()].astype(np.float64)
    ref_full = ((qn.cpu().numpy().astype(np.float64) @ deq.T) * scale.cpu().numpy().astype(np.float64)).astype(np.float32)
    check_topk_against_scores(sc.cpu().numpy(), ids.cpu().numpy(), ref_full, k, id_base=7, score_tol=2e-6)
    # quantisation quality against the f32 index: the retrieved sets mostly agree
    s32, i32 = ops.cosine_topk(qn, xn, k, row_id_base=7)
    m = min(k, N)
    a, b = ids[:, :m].cpu().numpy(), i32[:, :m].cpu().numpy()
    overlap = np.mean([len(np.intersect1d(a[r], b[r])) / m for r in range(Q)])
    assert overlap > 0.8, overlap
    assert float((sc[:, 0] - s32[:, 0]).abs().max()) < 2e-2
    with pytest.raises(NotImplementedError):
        ops.cosine_topk(qn[:, :32].contiguous(), codes[:, :32].contiguous(), 5, row_scale=scale)  # D % 64 != 0
    with pytest.raises(ValueError):
        ops.cosine_topk(qn, codes, 5)  # no scale


@pytest.mark.parametrize("Q,N,D,k", [(32, 200000, 768, 500), (5, 3000, 64, 50), (17, 70001, 1024, 100), (32, 40000, 1280, 200)])
def test_cosine_topk_fp8_native_mfma_variant(dev, Q, N, D, k):
    """BASELINE config 5 "fp8 MFMA scoring": the e4m3 index through v_mfma_f32_16x16x32_fp8_fp8 with the f32 query written as
    two e4m3 pieces (power-of-two scales).  Checked against the exact scores of the DEQUANTISED rows (float64 oracle):
      * every returned score is within the two-piece bound of the exact one: |q - s1 p1 - s2 p2|_inf <= 2^-8 max|q| per
        element, so |delta score| <= 2^-8 max|q| * sum|x_row| * row_scale (measured: far smaller);
      * the list is sorted (score desc, id asc) and holds no duplicate or out-of-range id;
      * it agrees with the widening variant (exact for the dequantised rows) on >= 97 % of the ids, and differing ids are
        near-ties: their exact scores lie within the bound of the exact k-th score."""
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N + D + 3)
    q = np.random.default_rng(Q + 5).standard_normal((Q, D), dtype=np.float32)
    q[0] *= 1e-3   # tiny and huge queries: the piece scales are per query
    q[min(1, Q - 1)] *= 1e4
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qd = torch.from_numpy(q).to(dev)
    codes, scale = ops.quantize_rows_fp8(xn)
    sc, ids = ops.cosine_topk(qd, codes, k, row_scale=scale, row_id_base=7, fp8_mfma=True)
    sw, iw = ops.cosine_topk(qd, codes, k, row_scale=scale, row_id_base=7, fp8_mfma=False)
    sd, idd = ops.cosine_topk(qd, codes, k, row_scale=scale, row_id_base=7)  # the default for an e4m3 index IS the native variant
    assert torch.equal(idd, ids) and torch.equal(sd, sc)
    deq = ocos.e4m3_decode_table()[codes.cpu().numpy()].astype(np.float64) * scale.cpu().numpy().astype(np.float64)[:, None]
    exact = q.astype(np.float64) @ deq.T  # [Q, N]
    bound = (2.0 ** -8) * np.abs(q).max(axis=1, keepdims=True) * np.abs(deq).sum(axis=1)[None, :]  # [Q, N]
    m = min(k, N)
    s_np, i_np = sc.cpu().numpy(), ids.cpu().numpy() - 7
    assert (i_np[:, :m] >= 0).all() and (i_np[:, :m] < N).all()
    for r in range(Q):
        row_ids = i_np[r, :m]
        assert np.unique(row_ids).size == m
        err = np.abs(s_np[r, :m].astype(np.float64) - exact[r, row_ids])
        assert (err <= bound[r, row_ids] + 1e-6 * np.abs(exact[r, row_ids]) + 1e-30).all(), (r, err.max())
        ds = np.diff(s_np[r, :m])
        assert (ds <= 0).all() and (np.diff(row_ids)[ds == 0] > 0).all()
        kth = np.sort(exact[r])[::-1][m - 1]
        for rid in np.setdiff1d(row_ids, iw[r, :m].cpu().numpy() - 7):
            assert abs(exact[r, rid] - kth) <= 2 * bound[r, rid] + 1e-30, (r, rid)
    a, b = i_np[:, :m], iw[:, :m].cpu().numpy() - 7
    overlap = np.mean([len(np.intersect1d(a[r], b[r])) / m for r in range(Q)])
    assert overlap >= 0.97, overlap
    with pytest.raises(ValueError, match="e4m3"):
        ops.cosine_topk(qd, xn, k, fp8_mfma=True)


def test_cosine_topk_row_scale_equals_prenormalised(dev):
    """raw index + row_scale (fused normalisation) returns the same ids as a normalised index."""
    from evi_rag_amd import ops

    N, D, Q, k = 30000, 768, 32, 500
    x = _make_index(N, D, seed=5)
    q = np.random.default_rng(6).standard_normal((Q, D), dtype=np.float32)
    xd = torch.from_numpy(x).to(dev)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    sc, ids = ops.cosine_topk(qn, xd, k, row_scale=ops.row_inv_norm(xd, EPS))
    ref_full = ocos.cosine_scores(q, x, EPS)
    check_topk_against_scores(sc.cpu().numpy(), ids.cpu().numpy(), ref_full, k)


def test_cosine_topk_sorted_index_worst_case(dev):
    """Adversarial order: every later row beats every earlier one, so every row passes the filter."""
    from evi_rag_amd import ops

    N, D, k = 200000, 64, 100
    base = np.zeros((N, D), dtype=np.float32)
    base[:, 0] = 1.0
    base[:, 1] = np.linspace(-1.0, 1.0, N, dtype=np.float32)  # cosine with e1 increases with row id
    q = np.zeros((2, D), dtype=np.float32)
    q[0, 1] = 1.0
    q[1, 1] = -1.0
    xn = ops.normalize_embeddings(torch.from_numpy(base).to(dev), EPS)
    qn = torch.from_numpy(q).to(dev)
    sc, ids = ops.cosine_topk(qn, xn, k)
    ids = ids.cpu().numpy()
    assert np.array_equal(ids[0], np.arange(N - 1, N - 1 - k, -1))
    assert np.array_equal(ids[1], np.arange(k))


def test_cosine_topk_more_rows_than_the_largest_segment(dev):
    """N > 2^24 rows: the scan runs several maximum-size segments (the 100 M-row regime).  Checked
    through size-independent properties: planted rows are found at rank 1 and every returned score
    equals a direct recomputation of that row's dot product."""
    from evi_rag_amd import ops

    N, D, Q, k = (1 << 24) + 100_003, 32, 8, 64
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((N, D), generator=g, device=dev)
    ops.normalize_embeddings(x, EPS, out=x)
    gold = torch.tensor([0, 65535, 65536, 1 << 20, (1 << 24) - 1, 1 << 24, N - 2, N - 1], device=dev)
    q = x[gold].clone()  # cosine 1.0 with its own row
    sc, ids = ops.cosine_topk(q, x, k)
    assert torch.equal(ids[:, 0], gold)
    assert float((sc[:, 0] - 1.0).abs().max()) < 1e-5
    direct = (x[ids.view(-1)].view(Q, k, D) * q.view(Q, 1, D)).sum(-1)
    assert float((direct - sc).abs().max()) < 2e-6
    assert bool((sc[:, 1:] <= sc[:, :-1]).all())
    # nothing better was missed: the full score matrix via a library GEMM as an independent check
    ref = torch.topk(q @ x.T, k, dim=1)
    for i in range(Q):
        assert set(ref.indices[i].tolist()) == set(ids[i].tolist()), f"query {i}: rows missing"
    assert float((ref.values - sc).abs().max()) < 2e-6


def test_cosine_topk_small_workspace_same_result(dev):
    """A minimum-size workspace only changes the segment schedule, never the result."""
    from evi_rag_amd import _lib, ops

    N, D, Q, k = 250000, 384, 32, 300
    x = _make_index(N, D, seed=11)
    q = np.random.default_rng(12).standard_normal((Q, D), dtype=np.float32)
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    s1, i1 = ops.cosine_topk(qn, xn, k)
    small = int(_lib.load().evi_cosine_topk_min_workspace_bytes(Q, N, D, k))
    ws = torch.empty(small, dtype=torch.uint8, device=dev)
    s2, i2 = ops.cosine_topk(qn, xn, k, workspace=ws)
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    with pytest.raises(MemoryError):
        ops.cosine_topk(qn, xn, k, workspace=ws[: small // 2])


def test_cosine_topk_rejects_bad_shapes(dev):
    from evi_rag_amd import ops

    q = torch.zeros((2, 24), device=dev)
    x = torch.zeros((4, 24), device=dev)
    with pytest.raises(NotImplementedError):
        ops.cosine_topk(q, x, 2)  # D % 16 != 0
    q = torch.zeros((2, 32), device=dev)
    x = torch.zeros((4, 32), device=dev)
    with pytest.raises(ValueError):
        ops.cosine_topk(q, x, 0)
    with pytest.raises(ValueError):
        ops.cosine_topk(q, x, 5000)


def test_sharded_topk_merge_equals_single_shard(dev):
    """Row-sharding invariance: per-shard top-k + merge == one-shard top-k, bit-exact."""
    from evi_rag_amd import ops

    N, D, Q, k, P = 300000, 768, 32, 500, 4
    x = _make_index(N, D, seed=21)
    q = np.random.default_rng(22).standard_normal((Q, D), dtype=np.float32)
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    s_all, i_all = ops.cosine_topk(qn, xn, k)
    bounds = [N * r // P for r in range(P + 1)]
    parts = [ops.cosine_topk(qn, xn[bounds[r]:bounds[r + 1]], k, row_id_base=bounds[r]) for r in range(P)]
    ss = torch.stack([p[0] for p in parts])
    ii = torch.stack([p[1] for p in parts])
    s_m, i_m = ops.topk_merge(ss, ii)
    assert torch.equal(i_m, i_all)
    assert torch.equal(s_m, s_all)
    # the packed exchange layout (one all-gather per step): kernels write straight into the records
    from evi_rag_amd import _lib

    rec = int(_lib.load().evi_topk_packed_bytes(Q, k))
    packed = torch.empty(P * rec, dtype=torch.uint8, device=dev)
    for r in range(P):
        views = ops.topk_packed_views(packed[r * rec:(r + 1) * rec], Q, k)
        ops.cosine_topk(qn, xn[bounds[r]:bounds[r + 1]], k, row_id_base=bounds[r], out=views)
    s_p, i_p = ops.topk_merge_packed(packed, P, Q, k)
    assert torch.equal(i_p, i_all) and torch.equal(s_p, s_all)
    # and the merge kernel itself against the oracle merge
    rs, ri = orank.merge_topk(ss.cpu().numpy(), ii.cpu().numpy(), k)
    assert np.array_equal(ri, i_m.cpu().numpy())
    assert np.array_equal(rs, s_m.cpu().numpy())


def test_topk_merge_padding(dev):
    from evi_rag_amd import ops

    scores = torch.tensor([[[3.0, 1.0, -np.inf]], [[3.0, 2.0, 0.5]]], device=dev)
    ids = torch.tensor([[[4, 9, -1]], [[10, 11, 12]]], device=dev, dtype=torch.int64)
    s, i = ops.topk_merge(scores, ids)
    assert i.cpu().tolist() == [[4, 10, 11]]
    assert s.cpu().tolist() == [[3.0, 3.0, 2.0]]


@pytest.mark.parametrize("k", [1, 10, 500])
def test_segment_topk_matches_oracle(dev, k):
    from evi_rag_amd import ops

    rng = np.random.default_rng(k)
    sizes = [0, 1, 5, 31, 4096, 0, 10000, 20000, 3]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    scores = rng.standard_normal(int(ptr[-1])).astype(np.float32)
    scores[ptr[4]:ptr[4] + 2000] = 0.25  # a long run of exact ties
    scores[ptr[6] + 5] = np.inf
    idx, val, cnt = ops.segment_topk(torch.from_numpy(scores).to(dev), torch.from_numpy(ptr).to(dev), k)
    ridx, rval, rcnt = orank.segment_topk(scores, ptr, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(val.cpu().numpy(), rval)
    assert np.array_equal(cnt.cpu().numpy(), rcnt)


@pytest.mark.parametrize("storage", ["f16", "fp8"])
def test_reduced_precision_index_at_shard_size(dev, storage):
    """BASELINE configs 4-5 at ONE GPU's share of the 100 M-row index (12.5 M rows x 768): properties that
    do not need a full-size oracle — planted rows come back first, lists are sorted with (score desc, id asc)
    ties, results do not depend on how the shard is split, and a dense re-scoring of the returned ids
    reproduces the returned scores."""
    from evi_rag_amd import ops

    N, D, Q, k = 12_500_000, 768, 32, 500
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.empty((N, D), dtype=torch.float16, device=dev)
    for lo in range(0, N, 1 << 21):  # generate in slabs: no f32 copy of the whole table
        hi = min(N, lo + (1 << 21))
        x[lo:hi] = ops.normalize_embeddings(torch.randn(hi - lo, D, device=dev, generator=g)).to(torch.float16)
    gold = torch.randint(0, N, (Q,), device=dev, generator=g)
    q = ops.normalize_embeddings(x[gold].float() + 0.02 * torch.randn(Q, D, device=dev, generator=g))
    if storage == "fp8":
        codes = torch.empty((N, D), dtype=torch.uint8, device=dev)
        scale = torch.empty(N, dtype=torch.float32, device=dev)
        for lo in range(0, N, 1 << 21):
            hi = min(N, lo + (1 << 21))
            codes[lo:hi], scale[lo:hi] = ops.quantize_rows_fp8(x[lo:hi].float())
        index, row_scale = codes, scale
        del x
    else:
        index, row_scale = x, None
    torch.cuda.empty_cache()
    # fp8: the widening variant (f32 query exact) carries the tight re-scoring check below; the native fp8-MFMA variant —
    # the default for an e4m3 index — is checked after it with its own (two-piece query) tolerance
    kw = {"fp8_mfma": False} if storage == "fp8" else {}
    s, i = ops.cosine_topk(q, index, k, row_scale=row_scale, **kw)
    assert torch.equal(i[:, 0], gold)                                     # Hits@1 of the planted rows
    assert bool((s[:, 1:] <= s[:, :-1]).all())                            # sorted
    tie = s[:, 1:] == s[:, :-1]
    assert bool((i[:, 1:][tie] > i[:, :-1][tie]).all())                   # ties by ascending id
    assert int(i.min()) >= 0 and int(i.max()) < N
    # shard invariance: three uneven shards merged == the single pass, bit for bit
    cuts = [0, 4_000_001, 9_999_984, N]
    parts = [ops.cosine_topk(q, index[a:b], k, row_scale=None if row_scale is None else row_scale[a:b], row_id_base=a, **kw)
             for a, b in zip(cuts[:-1], cuts[1:])]
    ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, i) and torch.equal(ms, s)
    # dense re-scoring of the returned rows
    rows = index[i[:, :50].reshape(-1)]
    if storage == "fp8":
        from oracle import cosine as ocos

        tab = torch.from_numpy(ocos.e4m3_decode_table()).to(dev)
        deq = tab[rows.long()] * row_scale[i[:, :50].reshape(-1)].view(-1, 1)
    else:
        deq = rows.float()
    ref = (deq.view(Q, 50, D).double() * q.double().view(Q, 1, D)).sum(-1).float()
    assert float((ref - s[:, :50]).abs().max()) < 3e-6
    if storage == "fp8":
        sn, i_n = ops.cosine_topk(q, index, k, row_scale=row_scale)       # default = native fp8 MFMA, two e4m3 query pieces
        assert torch.equal(i_n[:, 0], gold) and bool((sn[:, 1:] <= sn[:, :-1]).all())
        tie = sn[:, 1:] == sn[:, :-1]
        assert bool((i_n[:, 1:][tie] > i_n[:, :-1][tie]).all())
        inter = (i_n.unsqueeze(2) == i.unsqueeze(1)).any(dim=2).float().sum(dim=1) / k
        assert float(inter.mean()) > 0.99, float(inter.mean())
        rows_n = index[i_n[:, :50].reshape(-1)]
        deq_n = tab[rows_n.long()] * row_scale[i_n[:, :50].reshape(-1)].view(-1, 1)
        ref_n = (deq_n.view(Q, 50, D).double() * q.double().view(Q, 1, D)).sum(-1).float()
        assert float((ref_n - sn[:, :50]).abs().max()) < 1e-4            # query pieces carry 16 significant bits


@pytest.mark.parametrize("Q,N,D,k,scaled", [(512, 400000, 768, 500, False), (130, 50000, 128, 100, True), (96, 3000, 64, 10, False),
                                             (64, 700, 32, 500, False)])
def test_cosine_topk_gemm_equals_scan_bit_for_bit(dev, Q, N, D, k, scaled):
    """Many-query path: ids AND scores identical to the scan (its candidates are re-scored with the scan's own
    MFMA sequence); duplicated rows (exact ties) and a zero row are in the index."""
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N + 3)
    q = np.random.default_rng(Q).standard_normal((Q, D), dtype=np.float32)
    xd = torch.from_numpy(x).to(dev)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    if scaled:
        idx, scale = xd, ops.row_inv_norm(xd, EPS)
    else:
        idx, scale = ops.normalize_embeddings(xd, EPS), None
    s0, i0 = ops.cosine_topk(qn, idx, k, row_scale=scale, row_id_base=11, method="scan")
    for products in (3, 1, None):  # split-bf16 selection, plain-bf16 selection, the automatic plan
        s1, i1 = ops.cosine_topk_gemm(qn, idx, k, row_scale=scale, row_id_base=11, fallback=False, products=products)
        assert torch.equal(i1, i0), products
        assert torch.equal(s1, s0), products
    shadow = ops.index_shadow_bf16(idx)  # bf16 copy of the rows: same selection arithmetic, no conversion in the GEMM
    assert torch.equal(shadow, idx.to(torch.bfloat16))
    s1, i1 = ops.cosine_topk_gemm(qn, idx, k, row_scale=scale, row_id_base=11, fallback=False, products=1, shadow=shadow)
    assert torch.equal(i1, i0) and torch.equal(s1, s0)


def test_cosine_topk_gemm_refuses_what_it_cannot_prove(dev):
    """All rows identical: every approximate score ties, the gap test fails, fallback=False raises and the
    default falls back to the scan."""
    from evi_rag_amd import ops

    N, D, Q, k = 5000, 64, 40, 50
    row = torch.randn(D, device=dev)
    idx = ops.normalize_embeddings(row.repeat(N, 1))
    qn = ops.normalize_embeddings(torch.randn(Q, D, device=dev))
    with pytest.raises(RuntimeError, match="could not prove"):
        ops.cosine_topk_gemm(qn, idx, k, fallback=False)
    s, i = ops.cosine_topk_gemm(qn, idx, k)
    s0, i0 = ops.cosine_topk(qn, idx, k, method="scan")
    assert torch.equal(i, i0) and torch.equal(s, s0)
    big_q = ops.normalize_embeddings(torch.randn(100, D, device=dev))
    sa, ia = ops.cosine_topk(big_q, idx, k, method="auto")  # 100 queries: routed to the GEMM path, which falls back here
    sb, ib = ops.cosine_topk(big_q, idx, k, method="scan")
    assert torch.equal(ia, ib) and torch.equal(sa, sb)
    with pytest.raises(NotImplementedError):
        ops.cosine_topk_gemm(qn, idx, 1500)  # k + reserve exceeds the selector's capacity
    with pytest.raises(NotImplementedError):
        ops.cosine_topk_gemm(qn, idx, 1100, products=1)  # the coarse selection needs k + 1100 > 2048 slots


@pytest.mark.parametrize("Q,N,D,k", [(256, 300000, 768, 500), (100, 20000, 64, 40)])
def test_cosine_topk_gemm_f16_index_equals_f16_scan(dev, Q, N, D, k):
    """f16-stored index: the many-query path must reproduce evi_cosine_topk_f16 bit for bit."""
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N + 9)
    q = np.random.default_rng(Q + 1).standard_normal((Q, D), dtype=np.float32)
    x16 = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS).to(torch.float16)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    s0, i0 = ops.cosine_topk(qn, x16, k, row_id_base=3, method="scan")
    for products in (3, 1):
        s1, i1 = ops.cosine_topk_gemm(qn, x16, k, row_id_base=3, fallback=False, products=products)
        assert torch.equal(i1, i0) and torch.equal(s1, s0), products
    s2, i2 = ops.cosine_topk(qn, x16, k, row_id_base=3, method="auto")
    assert torch.equal(i2, i0) and torch.equal(s2, s0)


@pytest.mark.parametrize("Q,N,D,k", [(32, 400000, 768, 500), (33, 70000, 768, 500), (7, 3000, 64, 10), (32, 600, 32, 500),
                                     (1, 1, 32, 1), (40, 100000, 384, 100), (16, 1 << 20, 128, 1365)])
def test_cosine_topk_two_stage_equals_scan_bit_for_bit(dev, Q, N, D, k):
    """f16-shadow selection + f32 re-scoring: ids AND scores identical to the f32 scan; the index holds duplicated rows
    (exact ties), a zero row, and (N < k + reserve cases) fewer rows than the candidate list."""
    from evi_rag_amd import ops

    x = _make_index(N, D, seed=N + 5)
    q = np.random.default_rng(Q + 7).standard_normal((Q, D), dtype=np.float32)
    idx = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    shadow = ops.index_shadow_f16(idx)
    assert torch.equal(shadow, idx.to(torch.float16))  # round to nearest even, element by element
    s0, i0 = ops.cosine_topk(qn, idx, k, row_id_base=5, method="scan")
    s1, i1 = ops.cosine_topk_two_stage(qn, idx, shadow, k, row_id_base=5, fallback=False)
    assert torch.equal(i1, i0) and torch.equal(s1, s0)
    # the default method ("auto") finds the shadow that index_shadow_f16 attached to the index tensor and takes the two-stage
    # scan by itself; an in-place edit of the index invalidates the attachment (the scan runs again), as does dropping it
    if D % 32 == 0 and k + max(256, k // 2) <= 2048:
        assert ops.resident_shadow_f16(idx) is shadow
        sa, ia = ops.cosine_topk(qn, idx, k, row_id_base=5)
        assert torch.equal(ia, i0) and torch.equal(sa, s0)
        assert ops.cosine_topk.last_method == "two_stage"
        idx.add_(0.0)
        assert ops.resident_shadow_f16(idx) is None
        sb, ib = ops.cosine_topk(qn, idx, k, row_id_base=5)
        assert torch.equal(ib, i0) and torch.equal(sb, s0) and ops.cosine_topk.last_method == "scan"
        other = idx.clone()
        assert ops.resident_shadow_f16(other) is None  # a verdict / shadow never travels to another tensor object
    # without a read-back: the flag lands in the caller's tensor (default: device-side fallback; raw contract: fallback=False)
    for fb in ("device", False):
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        s2, i2 = ops.cosine_topk_two_stage(qn, idx, shadow, k, row_id_base=5, status=flag, fallback=fb)
        assert int(flag.item()) == 0 and torch.equal(i2, i0) and torch.equal(s2, s0)
    s3, i3 = ops.cosine_topk_two_stage(qn, idx, shadow, k, row_id_base=5)  # default: no status, no read-back
    assert torch.equal(i3, i0) and torch.equal(s3, s0)


def test_cosine_topk_two_stage_error_bound_and_refusal(dev):
    """(a) the shadow scan's scores stay inside the bound the proof uses (7e-4 |q|, measured ~1e-5); (b) a clustered
    index (thousands of rows inside the bound of the k-th) cannot be proven: status 1, fallback=False raises, the
    default falls back to the f32 scan and returns its result."""
    from evi_rag_amd import ops

    N, D, Q, k = 200000, 768, 32, 500
    idx = ops.normalize_embeddings(torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3)))
    qn = ops.normalize_embeddings(torch.randn(Q, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4)))
    shadow = ops.index_shadow_f16(idx)
    s32, i32 = ops.cosine_topk(qn, idx, k, method="scan")
    s16, i16 = ops.cosine_topk(qn, shadow, N if N <= 2048 else 2048)
    # compare row by row where both lists hold the row
    pos = {}
    worst = 0.0
    s16c, i16c, s32c, i32c = s16.cpu().numpy(), i16.cpu().numpy(), s32.cpu().numpy(), i32.cpu().numpy()
    for qi in range(Q):
        lut = dict(zip(i16c[qi].tolist(), s16c[qi].tolist()))
        for r, s in zip(i32c[qi].tolist(), s32c[qi].tolist()):
            assert r in lut  # the true top-500 sits inside the shadow's top-2048
            worst = max(worst, abs(lut[r] - s))
    assert worst < 7e-4 / 10, worst

    base = torch.randn(D, device=dev)
    clustered = ops.normalize_embeddings(base.repeat(6000, 1) + 1e-5 * torch.randn(6000, D, device=dev))
    sh = ops.index_shadow_f16(clustered)
    with pytest.raises(RuntimeError, match="could not prove"):
        ops.cosine_topk_two_stage(qn, clustered, sh, 50, fallback=False)
    s, i = ops.cosine_topk_two_stage(qn, clustered, sh, 50, fallback="host")
    assert ops.cosine_topk_two_stage.last_status == 1
    s0, i0 = ops.cosine_topk(qn, clustered, 50, method="scan")
    assert torch.equal(i, i0) and torch.equal(s, s0)
    sa, ia = ops.cosine_topk(qn, clustered, 50)  # auto: two-stage, proof fails, repaired by the gated f32 scan on the device
    assert torch.equal(ia, i0) and torch.equal(sa, s0)
    # raw contract (fallback=False + status): the unproven result is handed out, the flag says so
    flag = torch.zeros(1, dtype=torch.int32, device=dev)  # sticky: a failure stays visible after a later success
    ops.cosine_topk_two_stage(qn, clustered, sh, 50, status=flag, fallback=False)
    ops.cosine_topk_two_stage(qn, idx, shadow, 50, status=flag, fallback=False)
    assert int(flag.item()) == 1
    # default = device-side fallback: the gated f32 scan repairs the failed batch on the device (no read-back), the
    # outputs are the scan's bit for bit, with and without a status word, also into caller-owned outputs
    sd, idd = ops.cosine_topk_two_stage(qn, clustered, sh, 50)
    assert torch.equal(idd, i0) and torch.equal(sd, s0)
    flag.zero_()
    out = (torch.full((Q, 50), 7.0, device=dev), torch.full((Q, 50), 7, dtype=torch.int64, device=dev))
    ops.cosine_topk_two_stage(qn, clustered, sh, 50, status=flag, out=out, row_id_base=11)
    assert int(flag.item()) == 1 and torch.equal(out[1], i0 + 11) and torch.equal(out[0], s0)
    # ... and a batch whose proof holds right after a repaired one is untouched by the (closed) gate
    s_ok, i_ok = ops.cosine_topk_two_stage(qn, idx, shadow, 50)
    s_ref, i_ref = ops.cosine_topk(qn, idx, 50, method="scan")
    assert torch.equal(i_ok, i_ref) and torch.equal(s_ok, s_ref)
    with pytest.raises(ValueError):
        ops.cosine_topk_two_stage(qn, clustered, sh, 1500)  # k + reserve exceeds the selector's capacity
    with pytest.raises(ValueError):
        ops.cosine_topk_two_stage(qn, clustered, sh.to(torch.bfloat16), 50)
    with pytest.raises(ValueError, match="L2-normalised"):
        ops.index_shadow_f16(2.0 * clustered)  # the proof's bound assumes rows of norm <= 1


@pytest.mark.parametrize("k", [1, 500, 1500])
def test_segment_topk_long_lists_every_score_shape(dev, k):
    """Lists long enough for the selector's linear pre-partition (>= 16 384 keys), with the score shapes that could
    upset it: a narrow cluster plus a far outlier, heavy ties, all-equal scores, +-inf, NaN, denormals, a constant
    list with one larger value.  The result must be the stable descending order, whatever the bucket spread."""
    from evi_rag_amd import ops

    rng = np.random.default_rng(100 + k)
    n = 50000
    lists = []
    lists.append(rng.standard_normal(n).astype(np.float32) * 0.036)                     # cosine-like
    a = rng.standard_normal(n).astype(np.float32) * 1e-3 + 5.0
    a[123] = 1e30                                                                        # far outlier: everything else in one bucket
    lists.append(a)
    lists.append(rng.integers(0, 4, n).astype(np.float32))                               # four distinct values
    lists.append(np.full(n, 0.5, np.float32))                                            # all equal
    b = rng.standard_normal(n).astype(np.float32)
    b[:700] = np.inf
    b[700:1400] = -np.inf
    b[1400:1500] = np.nan
    lists.append(b)
    lists.append((rng.standard_normal(n) * 1e-41).astype(np.float32))                    # denormals
    c = np.zeros(n, np.float32)
    c[n - 1] = 1.0
    lists.append(c)
    lists.append(-np.abs(rng.standard_normal(n).astype(np.float32)) * 1e20)             # huge negatives
    scores = np.concatenate(lists)
    ptr = (np.arange(len(lists) + 1) * n).astype(np.int64)
    idx, val, cnt = ops.segment_topk(torch.from_numpy(scores).to(dev), torch.from_numpy(ptr).to(dev), k)
    ridx, rval, rcnt = orank.segment_topk(scores, ptr, k)
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(val.cpu().numpy(), rval, equal_nan=True)


def test_proof_paths_refuse_rows_and_queries_far_from_unit_norm(dev):
    """The exactness proofs bound the selection error by eps * |q| * |row| with |row| <= 1 and |q| ~ 1 built in.
    (a) an index of rows of norm 10 with near-ties around the k-th score: the GEMM-shaped path must not 'prove' a wrong
    result — it hands the batch to the scan (method gemm / auto) or raises (fallback=False); (b) queries of norm 1e-4 or
    1e4 in the two-stage scan: the f16 hi / lo split loses bits there, the proof is refused and the device-side fallback
    returns the scan's result bit for bit."""
    from evi_rag_amd import ops

    N, D, Q, k = 50000, 256, 128, 100
    g = torch.Generator(device=dev).manual_seed(21)
    x = ops.normalize_embeddings(torch.randn(N, D, device=dev, generator=g))
    q = ops.normalize_embeddings(torch.randn(Q, D, device=dev, generator=g))
    # near-ties: 400 rows whose scores against every query differ by ~1e-5 relative
    x[1000:1400] = ops.normalize_embeddings(x[999].repeat(400, 1) + 2e-5 * torch.randn(400, D, device=dev, generator=g))
    long_rows = 10.0 * x
    assert ops.rows_are_unit_norm(x) and not ops.rows_are_unit_norm(long_rows)
    s0, i0 = ops.cosine_topk(q, long_rows, k)
    for method in ("gemm", "auto"):
        s1, i1 = ops.cosine_topk(q, long_rows, k, method=method)
        assert torch.equal(i1, i0) and torch.equal(s1, s0), method
    assert ops.cosine_topk_gemm.last_products == 0  # the scan produced it
    with pytest.raises(ValueError, match="norm <= 1"):
        ops.cosine_topk_gemm(q, long_rows, k, fallback=False)
    # with the inverse norms as row_scale the rows count as unit again and the GEMM path may run
    inv = ops.row_inv_norm(long_rows)
    s2, i2 = ops.cosine_topk(q, long_rows, k, row_scale=inv)
    s3, i3 = ops.cosine_topk_gemm(q, long_rows, k, row_scale=inv)
    assert torch.equal(i3, i2) and torch.equal(s3, s2)
    # in-place edits are seen (cache keyed on the tensor version)
    long_rows.mul_(0.1)
    assert ops.rows_are_unit_norm(long_rows)

    shadow = ops.index_shadow_f16(x)
    for scale in (1e-4, 1e4):
        qs = q[:32] * scale
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        s4, i4 = ops.cosine_topk_two_stage(qs, x, shadow, k, status=flag)
        s5, i5 = ops.cosine_topk(qs, x, k)
        assert int(flag.item()) == 1  # proof refused ...
        assert torch.equal(i4, i5) and torch.equal(s4, s5)  # ... and repaired on the device


@pytest.mark.parametrize("storage", ["f32", "f16", "fp8", "fp8_mfma"])
def test_cosine_topk_widest_rows_d1280_32_queries(dev, storage):
    """D = 1280 (the widest supported row): the f32 / f16 / widened-fp8 fragments of 32 queries fill the 160 KiB of LDS
    exactly, so the scan takes the queries 16 at a time — same results as two separate 16-query calls and as the oracle."""
    from evi_rag_amd import ops

    N, D, Q, k = 30000, 1280, 32, 64
    x = _make_index(N, D, seed=77)
    q = np.random.default_rng(78).standard_normal((Q, D), dtype=np.float32)
    xn = ops.normalize_embeddings(torch.from_numpy(x).to(dev), EPS)
    qn = ops.normalize_embeddings(torch.from_numpy(q).to(dev), EPS)
    kw = {}
    if storage == "f16":
        idx = xn.to(torch.float16)
    elif storage.startswith("fp8"):
        idx, sc = ops.quantize_rows_fp8(xn)
        kw["row_scale"] = sc
        kw["fp8_mfma"] = storage == "fp8_mfma"
    else:
        idx = xn
    s, i = ops.cosine_topk(qn, idx, k, **kw)
    s_a, i_a = ops.cosine_topk(qn[:16].contiguous(), idx, k, **kw)
    s_b, i_b = ops.cosine_topk(qn[16:].contiguous(), idx, k, **kw)
    assert torch.equal(i, torch.cat([i_a, i_b])) and torch.equal(s, torch.cat([s_a, s_b]))
    if storage == "f32":
        check_topk_against_scores(s.cpu().numpy(), i.cpu().numpy(), ocos.cosine_scores(q, x, EPS), k)
