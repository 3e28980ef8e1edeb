"""GPU parity of the single-launch Hamming search (ucfp_amd/csrc/hamming_direct.hip): 1..8 queries, k <= 32 -- the request
shape of the reference's /v1/query (src/server/handlers.rs:143-159: ONE query per request).  Bit-exact (ids, distances,
counts) against the oracle, which is the spec (the reference has no Hamming search): order (d asc, id asc).

Covered: every (nq, k) corner, corpora from 1 code to 12.5 M (the per-GPU shard of BASELINE config 5), planted
neighbours, copies (every distance ties: only the id order decides) in ascending / descending / random id order, rows
sorted so that every later row is a better match (the list machinery's worst case), extreme words, searches in flight on
two streams, and the same inputs through the staged path (UCFP_HAMMING_NO_DIRECT=1) for the shapes both serve."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(ix, oracle, ids, codes, queries, k, tenant=0):
    g_ids, g_sc, g_d, g_c = ix.search(tenant, queries, k)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
    assert np.array_equal(g_c, o_c)
    assert np.array_equal(g_d, o_d)
    assert np.array_equal(g_ids, o_ids)
    valid = g_d != 0xFFFFFFFF
    assert np.allclose(g_sc[valid], 1.0 - g_d[valid] / 64.0)
    assert (g_sc[~valid] == -1.0).all()


def _corpus(rng, n, nq, planted=5):
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    queries = rng.integers(0, 2**64, nq, dtype=np.uint64)
    for q in range(nq):
        for _ in range(min(planted, n)):
            flips = rng.choice(64, size=int(rng.integers(0, 9)), replace=False)
            mask = np.uint64(0)
            for f in flips:
                mask |= np.uint64(1) << np.uint64(f)
            codes[int(rng.integers(0, n))] = queries[q] ^ mask
    ids = rng.permutation(np.arange(n, dtype=np.uint64) * np.uint64(11) + np.uint64(5))
    return ids, codes, queries


@pytest.mark.parametrize("n", [1, 2, 9, 63, 64, 65, 511, 512, 513, 8191, 8193, 100_003, 1_300_001])
def test_direct_every_query_count_and_k(gpu_ctx, oracle, n):
    from ucfp_amd import index
    rng = np.random.default_rng(n)
    ids, codes, queries = _corpus(rng, n, 8)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    for nq in (1, 2, 3, 5, 8):
        for k in (1, 10, 31, 32):
            _check(ix, oracle, ids, codes, queries[:nq], k)
    ix.close()


@pytest.mark.parametrize("order", ["ascending", "descending", "random"])
def test_direct_corpus_of_copies(gpu_ctx, oracle, order):
    """Every row equal (and a corpus of four distinct values): every distance ties, only the id order decides.  With
    descending ids every later row beats the list's k-th entry -- the tie-limited mode of the wave lists is exercised
    on every trip."""
    from ucfp_amd import index
    rng = np.random.default_rng(17)
    n = 260_000
    ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(1)
    if order == "descending":
        ids = ids[::-1].copy()
    elif order == "random":
        ids = rng.permutation(ids)
    queries = np.array([0xDEADBEEFCAFEF00D, 0, 2**64 - 1, 0xDEADBEEFCAFEF00C], np.uint64)
    for codes in (np.full(n, 0xDEADBEEFCAFEF00D, np.uint64), rng.choice(np.array([1, 3, 7, 2**63], np.uint64), n)):
        ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
        ix.upsert(0, ids, codes)
        for k in (1, 10, 32):
            _check(ix, oracle, ids, codes, queries, k)
            _check(ix, oracle, ids, codes, queries[:1], k)
        ix.close()


def test_direct_every_later_row_is_better(gpu_ctx, oracle):
    """Rows sorted by DEscending distance to the query: each trip's codes all beat everything seen before, so the lists
    fill and prune at the highest possible rate; and the mirror image (best rows first)."""
    from ucfp_amd import index
    rng = np.random.default_rng(23)
    n = 150_000
    q = np.uint64(0x0123456789ABCDEF)
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    d = np.array([bin(int(c ^ q)).count("1") for c in codes])
    ids = rng.permutation(n).astype(np.uint64)
    for srt in (np.argsort(-d, kind="stable"), np.argsort(d, kind="stable")):
        ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
        ix.upsert(0, ids, codes[srt])
        for k in (10, 32):
            _check(ix, oracle, ids, codes[srt], np.array([q, ~q, q ^ np.uint64(0xFF)], np.uint64), k)
        ix.close()


def test_direct_extreme_words_and_fewer_rows_than_k(gpu_ctx, oracle):
    from ucfp_amd import index
    rng = np.random.default_rng(29)
    codes = np.concatenate([np.zeros(40, np.uint64), np.full(40, 2**64 - 1, np.uint64),
                            rng.integers(0, 2**64, 70_000, dtype=np.uint64)])
    ids = rng.permutation(codes.size).astype(np.uint64) + np.uint64(2**63)      # ids above 2^63: compares are unsigned
    queries = np.array([0, 2**64 - 1, 1, 2**63], np.uint64)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    for k in (10, 32):
        _check(ix, oracle, ids, codes, queries, k)
    ix.close()
    small = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    small.upsert(5, ids[:7], codes[:7])
    g_ids, g_sc, g_d, g_c = small.search(5, queries[:2], 10)
    assert list(g_c) == [7, 7] and (g_ids[:, 7:] == np.uint64(2**64 - 1)).all() and (g_d[:, 7:] == 0xFFFFFFFF).all()
    _check(small, oracle, ids[:7], codes[:7], queries, 10, tenant=5)
    g_ids, _, _, g_c = small.search(6, queries[:2], 10)        # unknown tenant: empty shard
    assert list(g_c) == [0, 0] and (g_ids == np.uint64(2**64 - 1)).all()
    small.close()


def test_direct_shard_of_config5_and_two_streams(gpu_ctx, oracle, torch_cuda):
    """12.5 M codes (the per-GPU share of the 100 M corpus on 8 GPUs), 1 / 3 / 8 queries; then two single-query searches
    in flight on two streams (the index alternates between two state blocks), many times over."""
    torch = torch_cuda
    from ucfp_amd import index
    rng = np.random.default_rng(31)
    n = 12_500_000
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    ids = np.arange(n, dtype=np.uint64)
    queries = rng.integers(0, 2**64, 8, dtype=np.uint64)
    for j in range(8):
        codes[(j * 1_562_501 + 7) % n] = queries[j] ^ np.uint64(1 << (5 * j))
    ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
    d_ids = torch.from_numpy(ids.view(np.int64)).cuda()
    d_codes = torch.from_numpy(codes.view(np.int64)).cuda()
    ix.append_dev(0, d_ids.data_ptr(), d_codes.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, 10)
    for nq in (1, 3, 8):
        g_ids, _, g_d, g_c = ix.search(0, queries[:nq], 10)
        assert np.array_equal(g_ids, o_ids[:nq]) and np.array_equal(g_d, o_d[:nq]) and np.array_equal(g_c, o_c[:nq])
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_q = torch.from_numpy(queries.view(np.int64)).cuda()
    outs = [(torch.empty((1, 10), dtype=torch.int64, device="cuda"), torch.empty((1, 10), dtype=torch.int32, device="cuda"),
             torch.empty((1,), dtype=torch.int32, device="cuda")) for _ in range(8)]
    torch.cuda.synchronize()
    for rep in range(6):
        for j in range(8):
            o = outs[j]
            ix.search_dev(0, d_q.data_ptr() + 8 * j, 1, 10, o[0].data_ptr(), 0, o[1].data_ptr(), o[2].data_ptr(),
                          s[j & 1].cuda_stream)
    torch.cuda.synchronize()
    for j in range(8):
        assert np.array_equal(outs[j][0].cpu().numpy().view(np.uint64)[0], o_ids[j]), j
        assert np.array_equal(outs[j][1].cpu().numpy().view(np.uint32)[0], o_d[j]), j
    ix.close()


@pytest.mark.parametrize("ascending", [True, False])
def test_direct_append_only_shard_row_keys(gpu_ctx, oracle, torch_cuda, ascending):
    """An append-only shard keeps on the device whether its ids ascend with the row; if so the single-launch search
    breaks ties by ROW (the same order) and fetches ids only for the k answers.  Both states of the flag, on corpora
    where ties decide everything (copies) and on random codes, appended in several blocks."""
    torch = torch_cuda
    from ucfp_amd import index
    rng = np.random.default_rng(41)
    n = 2_100_000
    ids = np.arange(n, dtype=np.uint64) * np.uint64(5) + np.uint64(9)
    if not ascending:
        ids[n // 2], ids[n // 2 + 1] = ids[n // 2 + 1], ids[n // 2]          # one inversion clears the flag for good
    queries = rng.integers(0, 2**64, 8, dtype=np.uint64)
    for codes in (rng.integers(0, 2**64, n, dtype=np.uint64), np.full(n, 0x0F0F0F0F0F0F0F0F, np.uint64),
                  rng.choice(np.array([0, 1, 3], np.uint64), n)):
        ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
        st = torch.cuda.current_stream().cuda_stream
        for lo in range(0, n, 700_000):
            hi = min(n, lo + 700_000)
            d_i = torch.from_numpy(ids[lo:hi].view(np.int64)).cuda()
            d_c = torch.from_numpy(codes[lo:hi].view(np.int64)).cuda()
            ix.append_dev(0, d_i.data_ptr(), d_c.data_ptr(), hi - lo, st)
            torch.cuda.synchronize()
        for nq, k in ((1, 10), (3, 32), (8, 10)):
            _check(ix, oracle, ids, codes, queries[:nq], k)
        ix.close()


@pytest.mark.parametrize("n,nq,k", [(300_000, 1, 10), (300_000, 8, 32), (3000, 4, 10), (1_000_000, 6, 5)])
def test_staged_path_still_serves_the_same_shapes(gpu_ctx, oracle, monkeypatch, n, nq, k):
    """UCFP_HAMMING_NO_DIRECT=1 routes 1..8 queries through the staged search (sample, lane scan, lists) as before
    round 3: both paths stay exact on the shapes they share."""
    from ucfp_amd import index
    rng = np.random.default_rng(n + nq)
    ids, codes, queries = _corpus(rng, n, nq)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    monkeypatch.setenv("UCFP_HAMMING_NO_DIRECT", "1")
    _check(ix, oracle, ids, codes, queries, k)
    monkeypatch.delenv("UCFP_HAMMING_NO_DIRECT")
    _check(ix, oracle, ids, codes, queries, k)
    ix.close()
