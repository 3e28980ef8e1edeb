"""GPU parity for the kNN index through the C ABI.

Hamming: bit-exact ids + distances vs the oracle (order d asc, id asc).  Cosine: ids equal and
scores within 1e-5 of the oracle's reference-exact arithmetic (tolerance stated by BASELINE
north_star).  Also replays the reference's own index tests (src/index/embedded/mod.rs:522-589):
round trip ordering, tenant isolation, delete, records without embedding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COS_TOL = 1e-5


def _planted_corpus(rng, n, nq, planted_per_q=6):
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    queries = rng.integers(0, 2**64, nq, dtype=np.uint64)
    # plant near neighbours (Hamming 0..8) and exact duplicates (ties on distance)
    for q in range(nq):
        for j in range(planted_per_q):
            pos = rng.integers(0, n)
            flips = rng.choice(64, size=rng.integers(0, 9), replace=False)
            mask = np.uint64(0)
            for f in flips:
                mask |= np.uint64(1) << np.uint64(f)
            codes[pos] = queries[q] ^ mask
    ids = rng.permutation(np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(3))
    return ids, codes, queries


@pytest.mark.parametrize("n,nq,k", [(1000, 1, 10), (5000, 7, 10), (70000, 64, 10), (200000, 130, 10),
                                    (300000, 33, 1), (50000, 20, 37), (20000, 5, 100), (5, 3, 10)])
def test_hamming_matches_oracle(gpu_ctx, oracle, n, nq, k):
    from ucfp_amd import index
    rng = np.random.default_rng(n + nq + k)
    ids, codes, queries = _planted_corpus(rng, n, nq)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(3, ids, codes)
    assert ix.size(3) == n
    g_ids, g_sc, g_d, g_c = ix.search(3, queries, k)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
    assert np.array_equal(g_c, o_c)
    assert np.array_equal(g_d, o_d)
    assert np.array_equal(g_ids, o_ids)
    valid = g_d != 0xFFFFFFFF
    assert np.allclose(g_sc[valid], 1.0 - g_d[valid] / 64.0)
    ix.close()


@pytest.mark.parametrize("n,extra_q", [(40000, 0), (300_000, 0), (300_000, 90)])
def test_hamming_massive_ties_and_duplicates(gpu_ctx, oracle, n, extra_q):
    """All rows identical (every distance ties) and a corpus of few distinct codes: the id
    tie-break alone decides, under the heaviest slow-path load (robust tier, lane scan, matrix-core scan)."""
    from ucfp_amd import index
    rng = np.random.default_rng(9)
    ids = rng.permutation(n).astype(np.uint64)
    for codes in (np.full(n, 0xDEADBEEFCAFEF00D, np.uint64),
                  rng.choice(np.array([1, 3, 7, 2**63], np.uint64), n)):
        ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
        ix.upsert(0, ids, codes)
        q = np.concatenate([np.array([0xDEADBEEFCAFEF00D, 0, 1], np.uint64),
                            rng.integers(0, 2**64, extra_q, dtype=np.uint64)])
        g_ids, _, g_d, g_c = ix.search(0, q, 10)
        o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, q, 10)
        assert np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids) and np.array_equal(g_c, o_c)
        ix.close()


@pytest.mark.parametrize("nq", [8, 100])
def test_hamming_adversarial_order(gpu_ctx, oracle, nq):
    """The sample pre-pass only sees the head of the corpus; put all near neighbours at the tail
    and far codes (distance ~64) at the head so tau0 is useless. Correctness must not depend on it."""
    from ucfp_amd import index
    rng = np.random.default_rng(4)
    n = 300000
    q = rng.integers(0, 2**64, nq, dtype=np.uint64)
    codes = np.empty(n, np.uint64)
    codes[:] = ~q[0]                       # far from q[0] everywhere
    codes[n - 5000:] = q[0] ^ rng.integers(0, 2**12, 5000, dtype=np.uint64)  # near, at the tail
    ids = np.arange(n, dtype=np.uint64)[::-1].copy()
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    g_ids, _, g_d, g_c = ix.search(0, q, 10)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, q, 10)
    assert np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids)
    ix.close()


@pytest.mark.parametrize("n,nq,k", [(1_200_000, 70, 10), (2_500_000, 9, 10), (1_100_000, 3, 64),
                                    (300_000, 130, 10), (777_777, 300, 37), (600_000, 2100, 5),
                                    (400_003, 65, 128), (262_144, 64, 10), (262_145, 97, 1),
                                    # a full LDS image (128 query tiles) and the chunking above it
                                    (300_000, 4096, 4), (270_000, 4097, 3)])
def test_hamming_two_tier_matches_oracle(gpu_ctx, oracle, n, nq, k):
    """n >= 2^18 takes the staged filter path: the matrix-core scan (hamming_scan_mfma + hamming_rescan)
    for more than 64 queries, the lane-per-code scan (hamming_scan_lanes) below; ragged tails, more than
    2048 queries in the LDS image, k up to 128."""
    from ucfp_amd import index
    rng = np.random.default_rng(n + k)
    ids, codes, queries = _planted_corpus(rng, n, nq, planted_per_q=12)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    g_ids, _, g_d, g_c = ix.search(0, queries, k)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
    assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids)
    ix.close()


def test_hamming_matrix_filter_extreme_words(gpu_ctx, oracle):
    """The matrix-core filter packs two sums per result register in fields that hold -64 .. +63: the all-ones query
    (whose sum against an all-ones code is +64) is filtered one bit off and one distance wider (hamming.hip
    filter_query), and all-zero / all-one queries and codes, in both tiles of a pair and next to ordinary near
    neighbours, must come out exact."""
    from ucfp_amd import index
    rng = np.random.default_rng(6464)
    n, nq, k = 400_000, 160, 10
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    q = rng.integers(0, 2**64, nq, dtype=np.uint64)
    ones, zeros = np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)
    q[0], q[1], q[33], q[70] = ones, zeros, ones, zeros
    q[2] = ones ^ np.uint64(1)                      # popc 63: the ordinary path right next to the special one
    for pos in (5, 37, 64 + 9, 100_003, 250_000 + 40, n - 1):            # both halves of a pair, several steps and stages
        codes[pos] = ones
        codes[pos + 1 if pos + 1 < n else pos - 1] = zeros
    for j in range(nq):                                                  # ordinary planted neighbours in the same tiles
        for d in (1, 3, 6):
            pos = int(rng.integers(0, n))
            if codes[pos] in (ones, zeros):
                continue
            flip = np.uint64(0)
            for b in rng.choice(64, d, replace=False):
                flip |= np.uint64(1) << np.uint64(b)
            codes[pos] = q[j] ^ flip
    ids = rng.permutation(n).astype(np.uint64)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    g_ids, _, g_d, g_c = ix.search(0, q, k)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, q, k)
    assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids)
    assert g_d[0, 0] == 0 and g_d[1, 0] == 0      # the all-ones / all-zeros queries found their exact matches
    ix.close()
    # ... and through the append-only (strict thresholds) path with ids ascending
    ix = index.DeviceIndex(index.HAMMING64, flags=index.APPEND_ONLY, ctx=gpu_ctx)
    ids2 = np.arange(n, dtype=np.uint64) * 2 + 7
    ix.upsert(0, ids2, codes)
    g_ids, _, g_d, g_c = ix.search(0, q, k)
    o_ids, o_d, o_c = oracle.hamming_topk(ids2, codes, q, k)
    assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids)
    ix.close()


@pytest.mark.parametrize("nq", [5, 80])
def test_hamming_two_tier_overflow_falls_back(gpu_ctx, oracle, nq):
    """Prefix far from the query (tau = 64) and a tail full of near rows: every candidate list (and, on
    the matrix-core path, the suspect-block log) overflows, the device-side flag routes the batch through
    the robust tier, results stay exact."""
    from ucfp_amd import index
    rng = np.random.default_rng(77)
    n = 1_300_000
    q = rng.integers(0, 2**64, nq, dtype=np.uint64)
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    codes[: n // 4] = ~q[0]                                   # prefix: distance 64 from q[0]
    codes[n // 2:] = q[0] ^ rng.integers(0, 2**10, n - n // 2, dtype=np.uint64)  # tail: d <= 10
    ids = rng.permutation(n).astype(np.uint64)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    g_ids, _, g_d, g_c = ix.search(0, q, 10)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, q, 10)
    assert np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids) and np.array_equal(g_c, o_c)
    ix.close()


def test_upsert_overwrite_delete_tenants(gpu_ctx, oracle):
    from ucfp_amd import index
    rng = np.random.default_rng(1)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ids = np.arange(1000, dtype=np.uint64)
    codes = rng.integers(0, 2**64, 1000, dtype=np.uint64)
    ix.upsert(1, ids, codes)
    ix.upsert(2, ids[:10], codes[:10])
    # overwrite half of tenant 1, including a duplicate id inside the batch (last wins)
    new = rng.integers(0, 2**64, 501, dtype=np.uint64)
    up_ids = np.concatenate([ids[:500], ids[:1]])
    ix.upsert(1, up_ids, new)
    codes[:500] = new[:500]
    codes[0] = new[500]
    assert ix.size(1) == 1000 and ix.size(2) == 10 and ix.size(99) == 0
    assert ix.delete(1, np.array([5, 6, 7, 100000], np.uint64)) == 3
    keep = np.ones(1000, bool)
    keep[[5, 6, 7]] = False
    q = rng.integers(0, 2**64, 9, dtype=np.uint64)
    g_ids, _, g_d, g_c = ix.search(1, q, 10)
    o_ids, o_d, o_c = oracle.hamming_topk(ids[keep], codes[keep], q, 10)
    assert np.array_equal(g_ids, o_ids) and np.array_equal(g_d, o_d)
    # tenant isolation + unknown tenant + k = 0
    g2, _, _, c2 = ix.search(2, q[:1], 10)
    assert c2[0] == 10 and set(g2[0]) == set(range(10))
    _, _, _, c3 = ix.search(42, q[:1], 10)
    assert c3[0] == 0
    _, _, _, c0 = ix.search(1, q[:1], 0)
    assert c0[0] == 0
    ix.close()


@pytest.mark.parametrize("n,dim,nq,k", [(2000, 768, 3, 10), (5000, 384, 17, 10), (3000, 100, 40, 5),
                                        (1500, 3, 4, 10), (800, 1536, 2, 20), (50, 33, 2, 100),
                                        # batches beyond the LDS-resident kernel: the K-sliced GEMM kernel
                                        # (64 / 128 / 256 queries per corpus read, ragged row and query counts)
                                        (3001, 768, 60, 10), (2500, 128, 130, 5), (1777, 64, 300, 3),
                                        (900, 1024, 257, 10), (4000, 96, 70, 7),
                                        # one to four queries over >= 4096 rows: the row-streaming kernel (1-4
                                        # 256-dim blocks per lane, ragged last block, ragged last row group)
                                        (6001, 768, 1, 10), (5003, 384, 2, 10), (4099, 1024, 4, 5), (7000, 100, 3, 10),
                                        (4097, 256, 1, 1), (9000, 260, 4, 20),
                                        # 5 .. 16 queries over wide rows: the 4x4x1 row-stream kernel
                                        (20011, 768, 16, 10), (9001, 512, 5, 1), (12345, 1024, 9, 16), (8200, 640, 12, 17),
                                        (4100, 768, 7, 3), (70001, 768, 16, 10)])
def test_cosine_matches_oracle(gpu_ctx, oracle, n, dim, nq, k):
    from ucfp_amd import index
    rng = np.random.default_rng(n + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[rng.integers(0, n, 5)] = 0.0            # zero-norm rows are skipped
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(10)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    queries[0] = rows[7] * 3.0                   # an exact direction match -> score ~ 1
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    for q in range(nq):
        o_ids, o_sc = oracle.cosine_knn(ids, rows, queries[q], k)
        m = len(o_ids)
        assert g_c[q] == m
        assert np.abs(g_sc[q, :m] - o_sc).max() <= COS_TOL, (q, g_sc[q, :m], o_sc)
        # ids must agree wherever the oracle's neighbouring scores differ by more than the tolerance
        gap_ok = np.ones(m, bool)
        if m > 1:
            close = np.abs(np.diff(o_sc)) <= 2 * COS_TOL
            gap_ok[:-1] &= ~close
            gap_ok[1:] &= ~close
        if m == k and m > 0:
            gap_ok[-1] = False   # the k-th may swap with the (k+1)-th inside the tolerance
        assert np.array_equal(g_ids[q, :m][gap_ok], o_ids[gap_ok])
    ix.close()


def _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k):
    """float64 reference for a whole batch: one matrix product per 64 queries, then per query the rows at or above its k-th best
    score ordered by (score desc, id asc)."""
    r64 = rows.astype(np.float64)
    rn = np.sqrt((r64 * r64).sum(1))
    n = r64.shape[0]
    q64 = queries.astype(np.float64)
    qn = np.sqrt((q64 * q64).sum(1))
    for q0 in range(0, queries.shape[0], 64):
        dots = r64 @ q64[q0:q0 + 64].T                                      # n x (<= 64)
        for j in range(dots.shape[1]):
            q = q0 + j
            sc = dots[:, j] / np.maximum(rn * qn[q], 1e-300)
            sc[rn == 0] = -np.inf
            kk = min(k, n)
            thr = np.partition(sc, n - kk)[n - kk]                          # the k-th best score: everything at or above it competes
            cand = np.nonzero(sc >= thr)[0]
            order = cand[np.lexsort((ids[cand], -sc[cand]))][:k]
            o_sc, o_ids = sc[order], ids[order]
            m = int(np.isfinite(o_sc).sum())
            assert g_c[q] == m, (q, g_c[q], m)
            assert np.abs(g_sc[q, :m] - o_sc[:m]).max() <= COS_TOL, (q, g_sc[q, :m], o_sc[:m])
            gap_ok = np.ones(m, bool)
            if m > 1:
                close = np.abs(np.diff(o_sc[:m])) <= 2 * COS_TOL
                gap_ok[:-1] &= ~close
                gap_ok[1:] &= ~close
            if m == k and m > 0:
                gap_ok[-1] = False
            assert np.array_equal(g_ids[q, :m][gap_ok], o_ids[:m][gap_ok]), q


@pytest.mark.parametrize("nq,k", [(100, 10), (300, 1), (64, 50)])
def test_cosine_filtered_batch_pass(gpu_ctx, nq, k):
    """Batches over a shard of >= 2^18 rows take the thresholded GEMM pass (sample answer -> per-query threshold ->
    candidate lists); 300 queries are a full pass of 256 and a short one that goes the dense way."""
    import os
    from ucfp_amd import index
    n, dim = 300_000, 64
    rng = np.random.default_rng(nq * 31 + k)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[rng.integers(0, n, 50)] = 0.0
    rows[100_000:100_020] = rows[5]              # exact duplicates across the sample boundary: ties by id
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(3)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    queries[1] = rows[5] * 0.5
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    os.environ["UCFP_COSINE_NO_F16"] = "1"       # (dim 64 takes the f16 minima otherwise)
    try:
        g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    finally:
        del os.environ["UCFP_COSINE_NO_F16"]
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    ix.close()


def test_cosine_filtered_pass_overflow_falls_back(gpu_ctx):
    """The best rows sit behind the sample: thousands of rows beat the sample's k-th score for some queries, their
    candidate lists overflow, and the flag-gated dense pass has to produce the answer."""
    from ucfp_amd import index
    n, dim, nq, k = 280_000, 32, 80, 10
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    rows[-4000:] = queries[3] + 0.05 * rng.standard_normal((4000, dim)).astype(np.float32)
    rows[-8000:-4000] = -queries[7] * 2.0 + 0.3 * rng.standard_normal((4000, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    assert (g_ids[3] >= n - 4000).all()
    ix.close()


def test_cosine_filtered_pass_degenerate_sample(gpu_ctx):
    """The sample (the first rows of the shard) holds fewer than k scorable rows -- all-zero vectors -- so no
    threshold exists: every row passes, the lists overflow and the gated dense pass answers.  Also the smallest
    shard that takes the thresholded pass (2^18 rows) with a ragged tail tile."""
    from ucfp_amd import index
    n, dim, nq, k = (1 << 18) + 37, 32, 70, 10
    rng = np.random.default_rng(11)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[:40_000] = 0.0
    rows[5] = 1.0                                   # one scorable row inside the sample
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7)) % np.uint64(1_000_003 * 7)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    ix.close()


def test_cosine_reference_index_tests(gpu_ctx):
    """src/index/embedded/mod.rs:522-589 replayed on GpuIndex with Records."""
    from ucfp_amd import index
    from ucfp_amd.core import Modality, Record

    def rec(tenant, rid, emb):
        return Record(tenant, rid, Modality.Image, 1, "test", 0, b"fp", embedding=emb, model_id="test-model")

    db = index.GpuIndex(ctx=gpu_ctx)
    db.upsert([rec(1, 100, [1.0, 0.0, 0.0]), rec(1, 200, [0.0, 1.0, 0.0]), rec(1, 300, [0.7, 0.7, 0.0])])
    hits = db.knn(1, [0.6, 0.6, 0.0], 2)
    assert len(hits) == 2 and hits[0].record_id == 300 and hits[0].score > hits[1].score
    assert all(h.tenant_id == 1 and h.source == "vector" for h in hits)
    # knn_ignores_other_tenants
    db2 = index.GpuIndex(ctx=gpu_ctx)
    db2.upsert([rec(1, 1, [1.0, 0.0]), rec(2, 1, [1.0, 0.0])])
    hits = db2.knn(1, [1.0, 0.0], 10)
    assert len(hits) == 1 and hits[0].tenant_id == 1
    # delete_removes_records
    db3 = index.GpuIndex(ctx=gpu_ctx)
    db3.upsert([rec(1, 1, [1.0, 0.0]), rec(1, 2, [0.0, 1.0])])
    db3.delete(1, [1])
    hits = db3.knn(1, [1.0, 0.0], 10)
    assert len(hits) == 1 and hits[0].record_id == 2
    # knn_skips_records_with_no_embedding (and dimension mismatch)
    db4 = index.GpuIndex(ctx=gpu_ctx)
    without = rec(1, 9, None)
    db4.upsert([without, rec(1, 10, [1.0, 0.0])])
    hits = db4.knn(1, [1.0, 0.0], 10)
    assert len(hits) == 1 and hits[0].record_id == 10
    # empty query / k = 0 / zero-norm query -> []
    assert db4.knn(1, [], 10) == [] and db4.knn(1, [1.0, 0.0], 0) == [] and db4.knn(1, [0.0, 0.0], 10) == []


def test_image_records_feed_hamming_space(gpu_ctx, oracle):
    """End to end on one GPU: hash frames, upsert the Records, query by pHash."""
    from ucfp_amd import image, index
    from ucfp_amd.core import Modality, Record
    rng = np.random.default_rng(3)
    frames = oracle.image_synth(64, 256, 256, 0)
    frames[40] = frames[10]
    frames[40, :4, :4] ^= 1                  # near-duplicate of frame 10
    recs, _ = image.fingerprint_frames(frames, algo=image.MULTI, ctx=gpu_ctx)
    db = index.GpuIndex(ctx=gpu_ctx)
    db.upsert([Record(0, i, Modality.Image, 1, image.ALGORITHM_MULTIHASH, 0, recs[i].tobytes()) for i in range(64)])
    ph = image.global_hashes(recs[10].tobytes())["phash"]
    hits = db.hamming(0, image.ALGORITHM_PHASH, ph, 3)
    assert hits[0].record_id == 10 and hits[0].distance == 0 and hits[0].score == 1.0
    assert hits[1].record_id == 40


def test_topk_merge_dev_matches_global(gpu_ctx, oracle, torch_cuda):
    """The multi-GPU final step on one device: split a corpus into 4 shards, search each,
    merge the stacked partial lists, compare with the oracle over the whole corpus."""
    torch = torch_cuda
    from ucfp_amd import index
    rng = np.random.default_rng(8)
    n, nq, k, parts = 80000, 50, 10, 4
    ids, codes, queries = _planted_corpus(rng, n, nq)
    shard_ids, shard_keys = [], []
    for p in range(parts):
        sl = slice(p * n // parts, (p + 1) * n // parts)
        ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
        ix.upsert(0, ids[sl], codes[sl])
        gi, _, gd, _ = ix.search(0, queries, k)
        shard_ids.append(gi)
        shard_keys.append(gd)
        ix.close()
    pid = torch.from_numpy(np.stack(shard_ids).astype(np.int64)).cuda()
    pk = torch.from_numpy(np.stack(shard_keys).astype(np.int32)).cuda()
    out_ids = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    out_keys = torch.zeros((nq, k), dtype=torch.int32, device="cuda")
    out_sc = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    out_cnt = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    index.topk_merge_dev(index.HAMMING64, pid.data_ptr(), pk.data_ptr(), parts, nq, k, out_ids.data_ptr(),
                         out_sc.data_ptr(), out_keys.data_ptr(), out_cnt.data_ptr(),
                         torch.cuda.current_stream().cuda_stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
    assert np.array_equal(out_ids.cpu().numpy().astype(np.uint64), o_ids)
    assert np.array_equal(out_keys.cpu().numpy().astype(np.uint32), o_d)


def test_hamming_properties_at_scale(gpu_ctx, torch_cuda):
    """BASELINE-scale shard (12.5 M codes, the per-GPU share of configs[4]) without the oracle:
    planted neighbours come back first with their exact distance, distances are sorted, ids are
    ascending inside a distance, the search is idempotent, and a query equal to a stored code
    returns that code's id at distance 0."""
    torch = torch_cuda
    from ucfp_amd import index
    n, nq, k = 12_500_000, 512, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    codes = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
    ids = torch.arange(n, dtype=torch.int64, device="cuda") * 3 + 1
    q = torch.randint(-2**63, 2**63 - 1, (nq,), dtype=torch.int64, device="cuda", generator=g)
    pos = (torch.arange(nq, device="cuda") * 24_391 + 17) % n
    flips = torch.ones(nq, dtype=torch.int64, device="cuda") << (torch.arange(nq, device="cuda") % 60)
    codes[pos] = q ^ flips                      # one neighbour at distance exactly 1 per query
    q[0] = codes[123]                           # and an exact hit
    ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
    stream = torch.cuda.current_stream().cuda_stream
    ix.append_dev(0, ids.data_ptr(), codes.data_ptr(), n, stream)
    outs = []
    for _ in range(2):
        o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        o_sc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        o_d = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        o_c = torch.empty((nq,), dtype=torch.int32, device="cuda")
        ix.search_dev(0, q.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_d.data_ptr(), o_c.data_ptr(), stream)
        torch.cuda.synchronize()
        outs.append((o_ids.cpu().numpy(), o_d.cpu().numpy(), o_c.cpu().numpy()))
    (i1, d1, c1), (i2, d2, c2) = outs
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2) and (c1 == k).all()
    assert (np.diff(d1, axis=1) >= 0).all()
    same = np.diff(d1, axis=1) == 0
    assert (np.diff(i1, axis=1)[same] > 0).all()
    assert d1[0, 0] == 0 and i1[0, 0] == 123 * 3 + 1
    exp_ids = (pos.cpu().numpy() * 3 + 1)[1:]
    assert (d1[1:, 0] <= 1).all()
    assert (i1[1:, 0] == exp_ids).mean() > 0.99     # a random code at distance <= 1 is essentially impossible
    ix.close()


def test_many_queries_small_corpus(gpu_ctx, oracle):
    """Query batches far larger than the corpus (grid-dimension limits of the select/merge launches)."""
    from ucfp_amd import index
    rng = np.random.default_rng(12)
    n, dim, nq = 300, 16, 70000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    cx = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    cx.upsert(0, ids, rows)
    gi, gs, _, gc = cx.search(0, q, 3)
    assert (gc == 3).all()
    for j in (0, 1, 32767, 32768, 65535, 65536, nq - 1):
        oi, osc = oracle.cosine_knn(ids, rows, q[j], 3)
        assert np.array_equal(gi[j], oi) and np.abs(gs[j] - osc).max() <= COS_TOL
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    hx = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    hx.upsert(0, ids, codes)
    hq = rng.integers(0, 2**64, nq, dtype=np.uint64)
    hi, _, hd, hc = hx.search(0, hq, 4)
    oi, od, _ = oracle.hamming_topk(ids, codes, hq[:64], 4)
    assert np.array_equal(hi[:64], oi) and np.array_equal(hd[:64], od) and (hc == 4).all()
    oi, od, _ = oracle.hamming_topk(ids, codes, hq[-64:], 4)
    assert np.array_equal(hi[-64:], oi) and np.array_equal(hd[-64:], od)


def test_hamming_randomised_configs(gpu_ctx, oracle):
    """Twenty random (n, nq, k, data shape) draws across the robust / lane / matrix-core paths, including
    clustered corpora (many rows near a few centres: fat distance bins, heavy candidate traffic)."""
    from ucfp_amd import index
    rng = np.random.default_rng(20260101)
    for trial in range(20):
        n = int(rng.integers(1, 1_500_000)) if trial % 4 else int(rng.integers(262_144, 400_000))
        nq = int(rng.integers(1, 260))
        k = int(rng.choice([1, 3, 10, 17, 64, 128]))
        style = trial % 3
        if style == 0:
            codes = rng.integers(0, 2**64, n, dtype=np.uint64)
        elif style == 1:   # clustered: centres with 0..6 random bit flips
            centres = rng.integers(0, 2**64, 50, dtype=np.uint64)
            codes = centres[rng.integers(0, 50, n)]
            for _ in range(6):
                flip = rng.random(n) < 0.5
                codes = np.where(flip, codes ^ (np.uint64(1) << rng.integers(0, 64, n).astype(np.uint64)), codes)
        else:              # low entropy: only 12 bits vary
            codes = rng.integers(0, 2**12, n, dtype=np.uint64) * np.uint64(0x0010000100001001)
        queries = codes[rng.integers(0, n, nq)] ^ (np.uint64(1) << rng.integers(0, 64, nq).astype(np.uint64))
        ids = rng.permutation(n).astype(np.uint64) + np.uint64(1000)
        ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
        ix.upsert(0, ids, codes)
        g_ids, _, g_d, g_c = ix.search(0, queries, k)
        o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
        assert np.array_equal(g_c, o_c), (trial, n, nq, k, style)
        assert np.array_equal(g_d, o_d), (trial, n, nq, k, style)
        assert np.array_equal(g_ids, o_ids), (trial, n, nq, k, style)
        ix.close()


def test_snapshot_round_trip(gpu_ctx, oracle, tmp_path):
    """ucfp_index_save / ucfp_index_load (SURVEY 8f N2 sidecar): a fresh index loaded from the file answers
    exactly like the one it was saved from -- several tenants, both kinds, after deletes and overwrites."""
    from ucfp_amd import index
    from ucfp_amd.errors import UcfpError
    rng = np.random.default_rng(5)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    for tenant, n in ((0, 30000), (7, 500), (9, 1)):
        ix.upsert(tenant, np.arange(n, dtype=np.uint64) * 3 + tenant, rng.integers(0, 2**64, n, dtype=np.uint64))
    ix.delete(0, np.arange(0, 3000, 3, dtype=np.uint64) * 3)
    ix.upsert(7, np.array([7 + 3 * 10], np.uint64), np.array([0xABCDEF], np.uint64))       # overwrite
    path = tmp_path / "ham.idx"
    ix.save(path)
    iy = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    iy.load(path)
    q = rng.integers(0, 2**64, 33, dtype=np.uint64)
    for tenant in (0, 7, 9, 4):
        assert ix.size(tenant) == iy.size(tenant)
        a, b = ix.search(tenant, q, 10), iy.search(tenant, q, 10)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    cx = index.DeviceIndex(index.COSINE_F32, 48, ctx=gpu_ctx)
    rows = rng.standard_normal((2000, 48)).astype(np.float32)
    cx.upsert(2, np.arange(2000, dtype=np.uint64), rows)
    cpath = tmp_path / "cos.idx"
    cx.save(cpath)
    cy = index.DeviceIndex(index.COSINE_F32, 48, ctx=gpu_ctx)
    cy.load(cpath)
    qv = rng.standard_normal((5, 48)).astype(np.float32)
    a, b = cx.search(2, qv, 7), cy.search(2, qv, 7)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    with pytest.raises(UcfpError):            # kind / dim mismatch is refused
        index.DeviceIndex(index.COSINE_F32, 32, ctx=gpu_ctx).load(cpath)
    with pytest.raises(UcfpError):
        iy.load(tmp_path / "missing.idx")


def test_record_codes_and_query_dispatch(gpu_ctx, oracle, torch_cuda):
    """Stored image records -> Hamming codes on the device (SURVEY 8f N2 offsets), and POST /v1/query with the
    additive `hash` field routed to the Hamming space (N3)."""
    torch = torch_cuda
    from ucfp_amd import image, index
    from ucfp_amd.core import Modality, QueryRequest, Record
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (40, 512, 512), dtype=np.uint8)
    multi, _ = image.fingerprint_frames(frames, algo=image.MULTI)
    ph, _ = image.fingerprint_frames(frames, algo=image.PHASH)
    d_multi = torch.from_numpy(multi.copy()).cuda()
    codes = torch.empty(40, dtype=torch.int64, device="cuda")
    for which, off in ((image.AHASH, 64), (image.PHASH, 232), (image.DHASH, 400)):
        image.record_codes_dev(d_multi.data_ptr(), 40, codes.data_ptr(), algo=image.MULTI, which=which)
        torch.cuda.synchronize()
        want = np.ascontiguousarray(multi[:, off:off + 8]).view("<u8").reshape(-1)
        assert np.array_equal(codes.cpu().numpy().view(np.uint64), want)
    d_ph = torch.from_numpy(ph.copy()).cuda()
    image.record_codes_dev(d_ph.data_ptr(), 40, codes.data_ptr(), algo=image.PHASH)
    torch.cuda.synchronize()
    assert np.array_equal(codes.cpu().numpy().view(np.uint64), np.ascontiguousarray(ph[:, 32:40]).view("<u8").reshape(-1))
    # the same records through the IndexBackend-shaped facade and the query DTO
    g = index.GpuIndex(gpu_ctx)
    g.upsert([Record(tenant_id=3, record_id=100 + i, modality=Modality.Image, format_version=1,
                     algorithm="imgfprint-multihash-v1", config_hash=0, fingerprint=multi[i].tobytes(),
                     embedding=[float(i), 1.0, 0.5]) for i in range(40)])
    want_hash = int.from_bytes(multi[17, 232:240].tobytes(), "little")
    hits = g.query(QueryRequest.from_json({"tenant_id": 3, "modality": "Image", "k": 3, "hash": want_hash,
                                           "algorithm": "imgfprint-phash-v1"}))
    assert hits[0].record_id == 117 and hits[0].distance == 0 and hits[0].source == "hamming" and hits[0].score == 1.0
    vh = g.query(QueryRequest.from_json({"tenant_id": 3, "modality": "Image", "vector": [5.0, 1.0, 0.5]}))
    assert vh[0].record_id == 105 and vh[0].source == "vector" and vh[0].vector_rank == 1 and len(vh) == 10
    assert g.query(QueryRequest.from_json({"tenant_id": 4, "modality": "Image", "vector": [5.0, 1.0, 0.5]})) == []


def test_sharded_submit_collect_two_in_flight(gpu_ctx, oracle, torch_cuda):
    """ShardedIndex.submit/collect (the pipelined form of search): two batches in flight use separate buffer
    sets and both answers equal the oracle's (single process: world = 1, the exchange is the merge alone)."""
    torch = torch_cuda
    from ucfp_amd import index, sharded
    rng = np.random.default_rng(77)
    n, nq, k = 50_000, 70, 10
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    ids = rng.permutation(n).astype(np.uint64)
    qa = codes[:nq] ^ np.uint64(0b101)
    qb = rng.integers(0, 2**64, nq, dtype=np.uint64)
    six = sharded.ShardedIndex(index.HAMMING64, ctx=gpu_ctx)
    six.append_local(torch.from_numpy(ids.view(np.int64)).cuda(), torch.from_numpy(codes.view(np.int64)).cuda())
    da = torch.from_numpy(qa.view(np.int64)).cuda()
    db = torch.from_numpy(qb.view(np.int64)).cuda()
    ta = six.submit(da, k)
    tb = six.submit(db, k)
    ra = [t.clone() for t in six.collect(ta)]
    rb = [t.clone() for t in six.collect(tb)]
    torch.cuda.synchronize()
    for (g_ids, _, g_keys, g_cnt), q in ((ra, qa), (rb, qb)):
        o_ids, o_d, _ = oracle.hamming_topk(ids, codes, q, k)
        assert np.array_equal(g_ids.cpu().numpy().view(np.uint64), o_ids)
        assert np.array_equal(g_keys.cpu().numpy().view(np.uint32), o_d)
        assert (g_cnt.cpu().numpy() == k).all()


@pytest.mark.parametrize("order", ["ascending", "ascending_blocks", "descending", "shuffled", "one_inversion"])
@pytest.mark.parametrize("nq", [7, 200])
def test_hamming_strict_thresholds_only_when_ids_ascend(gpu_ctx, oracle, order, nq):
    """An APPEND_ONLY shard tracks on the device whether its ids ascend with the row number; if they do, the stages after
    the first filter with d < (k-th distance) instead of <= (a later row cannot win a tie).  Heavy ties make the
    difference visible: codes drawn from 300 values, so whole plateaus of equal distance straddle every threshold.  Ids in
    any other order must keep the non-strict filter -- the answer is the oracle's in every case."""
    from ucfp_amd import index
    rng = np.random.default_rng(len(order) * 31 + nq)
    n, k = 700_000, 10
    palette = rng.integers(0, 2**64, 300, dtype=np.uint64)
    codes = palette[rng.integers(0, 300, n)]
    queries = palette[rng.integers(0, 300, nq)] ^ (np.uint64(1) << rng.integers(0, 64, nq).astype(np.uint64))
    if order in ("ascending", "ascending_blocks"):
        ids = np.cumsum(rng.integers(1, 5, n)).astype(np.uint64)
    elif order == "descending":
        ids = np.arange(n, 0, -1).astype(np.uint64) * 3
    elif order == "shuffled":
        ids = rng.permutation(n).astype(np.uint64) + 5
    else:
        ids = np.cumsum(rng.integers(1, 5, n)).astype(np.uint64)
        ids[n - 1000], ids[n - 999] = ids[n - 999], ids[n - 1000]          # a single swap far into the corpus
    ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
    if order == "ascending_blocks":                                         # several appends: the seam between blocks is checked too
        for a, b in ((0, 1), (1, 300_001), (300_001, n)):
            ix.upsert(0, ids[a:b], codes[a:b])
    else:
        ix.upsert(0, ids, codes)
    g_ids, _, g_d, g_c = ix.search(0, queries, k)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
    assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids)
    ix.close()


def test_hamming_searches_in_flight_on_two_streams(gpu_ctx, oracle, torch_cuda):
    """Searches enqueued on different streams alternate between the index's two workspaces and may run side by side
    (index.hip ws_slot): twelve batches of different queries on two streams, no host synchronisation in between, and a
    mutation behind them -- every batch must get its own exact answer."""
    import torch
    from ucfp_amd import index
    rng = np.random.default_rng(2424)
    n, nq, k = 400_000, 300, 10
    dev = torch.device("cuda", 0)
    ids = rng.permutation(n).astype(np.uint64)
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
    ix.upsert(0, ids, codes)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    qs, outs = [], []
    for b in range(12):
        q = rng.integers(0, 2**64, nq, dtype=np.uint64)
        q[::3] = codes[rng.integers(0, n, len(q[::3]))] ^ np.uint64(1 << (b % 60))     # near neighbours for a third of them
        qs.append(q)
    torch.cuda.synchronize()
    d_q = [torch.from_numpy(q.view(np.int64)).to(dev) for q in qs]
    torch.cuda.synchronize()
    for b in range(12):
        o = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.int32, device=dev),
             torch.empty((nq,), dtype=torch.int32, device=dev))
        outs.append(o)
        ix.search_dev(0, d_q[b].data_ptr(), nq, k, o[0].data_ptr(), 0, o[1].data_ptr(), o[2].data_ptr(),
                      streams[b & 1].cuda_stream)
    # a mutation right behind the searches must wait for both workspaces' users
    extra_ids = np.arange(n, n + 5, dtype=np.uint64) + np.uint64(10_000_000)
    ix.upsert(0, extra_ids, qs[0][:5].copy())
    torch.cuda.synchronize()
    for b in range(12):
        o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, qs[b], k)
        assert np.array_equal(outs[b][0].cpu().numpy().view(np.uint64), o_ids), b
        assert np.array_equal(outs[b][1].cpu().numpy().view(np.uint32), o_d), b
    g_ids, _, g_d, _ = ix.search(0, qs[0][:5], 1)         # the appended rows are exact matches of these queries
    assert np.array_equal(g_d[:, 0], np.zeros(5, np.uint32))
    ix.close()


def test_cosine_duplicate_rows_and_any_id_order(gpu_ctx, oracle):
    """Ties are broken by record id: a corpus where every row exists in many copies (equal keys everywhere, ids in random
    order) must return the k smallest ids of the best rows (8 queries over 768-d rows: the 4x4x1 row-stream kernel)."""
    from ucfp_amd import index
    rng = np.random.default_rng(99)
    base = rng.standard_normal((40, 768)).astype(np.float32)
    n = 30_000
    pick = rng.integers(0, 40, n)
    rows = base[pick].copy()
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(7) + np.uint64(3)
    queries = (base[:8] + 0.01 * rng.standard_normal((8, 768))).astype(np.float32)
    ix = index.DeviceIndex(index.COSINE_F32, 768, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = ix.search(0, queries, 10)
    for q in range(8):
        best = np.sort(ids[pick == q])[:10]        # all copies of base[q] score the same, far above the rest
        assert g_c[q] == 10 and np.array_equal(g_ids[q], best), q
        assert np.all(g_sc[q] > 0.99)
    ix.close()


@pytest.mark.parametrize("nq,dim", [(1, 384), (3, 768), (7, 768), (16, 512)])
def test_cosine_few_queries_massive_ties(gpu_ctx, nq, dim):
    """One to sixteen queries select through chunk minima + a threshold + a gather (topk.hip select_pruned_u32): a corpus
    that is mostly copies of a few rows makes every chunk qualify and the candidate list overflow again and again, and
    the answer must still be the k smallest ids among the best-scoring copies -- plus the plain case around it (distinct
    rows, k larger than the number of chunks, zero rows)."""
    from ucfp_amd import index
    rng = np.random.default_rng(1000 + nq)
    n, k = 150_000, 25
    base = rng.standard_normal((20, dim)).astype(np.float32)
    pick = rng.integers(0, 20, n)
    rows = base[pick].copy()
    loose = rng.integers(0, n, 2000)
    rows[loose] = rng.standard_normal((2000, dim)).astype(np.float32)     # some distinct rows in between
    pick[loose] = -1
    rows[rng.integers(0, n, 7)] = 0.0
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(3) + np.uint64(11)
    queries = (base[:nq] if nq <= 20 else base[np.arange(nq) % 20]).copy()
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    nrm = np.linalg.norm(rows, axis=1)
    for q in range(nq):
        live = (pick == q) & (nrm > 0)
        best = np.sort(ids[live])[:k]            # every copy of base[q] scores 1.0, above everything else
        assert g_c[q] == k and np.array_equal(g_ids[q], best), q
        assert np.all(np.abs(g_sc[q] - 1.0) <= 1e-5)
    ix.close()
    # a corpus smaller than one chunk, k above its size
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids[:17], rows[:17])
    g_ids, g_sc, _, g_c = ix.search(0, queries, k)
    live17 = int((nrm[:17] > 0).sum())
    assert np.all(g_c == live17) and np.all(g_ids[:, live17:] == np.uint64(0xFFFFFFFFFFFFFFFF))
    ix.close()


@pytest.mark.parametrize("nq", [6, 90])
def test_hamming_strict_thresholds_with_overflowing_lists(gpu_ctx, oracle, nq):
    """Ascending ids (strict stage thresholds) AND a clustered corpus whose candidate lists overflow: the fallback tier then
    filters the whole corpus with the SAMPLE's thresholds -- it must not see a strict stage threshold (found by the
    randomised soak: it once did, and dropped the k-th best itself)."""
    from ucfp_amd import index
    rng = np.random.default_rng(5 + nq)
    n = 700_000
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    centres = codes[rng.integers(0, n, 8)]
    near = centres[rng.integers(0, 8, n // 2)] ^ (np.uint64(1) << rng.integers(0, 63, n // 2).astype(np.uint64))
    codes[: n // 2] = near                                   # half of the corpus within one bit of eight centres
    ids = np.cumsum(rng.integers(1, 4, n)).astype(np.uint64)
    queries = codes[rng.integers(0, n, nq)] ^ np.uint64(5)
    ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
    ix.upsert(0, ids, codes)
    for k in (1, 5, 10):
        g_ids, _, g_d, g_c = ix.search(0, queries, k)
        o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
        assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids), k
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [262_144, 262_145, 262_144 + 127, 400_000, 524_288, 2_200_000])
def test_hamming_bound_pass_edges(gpu_ctx, oracle, n):
    """Batches on the matrix-core filter take their first thresholds from hamming_bound_mfma: the k-th smallest of 256
    group minima over the first min(n, 2^18) & ~127 codes.  Edges: corpora that END on the bound range (one stage that
    starts over at row 0), one row and one partial step behind it; k at and above the pass's limit (k <= 64, above it the
    sample histogram); the best neighbours all inside ONE group (the bound then comes from the other groups); the
    all-ones query (filtered one bit off, bound one wider); codes behind the bound range that beat everything in it."""
    from ucfp_amd import index
    rng = np.random.default_rng(n % 9973)
    nq = 96
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    q = rng.integers(0, 2**64, nq, dtype=np.uint64)
    q[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    q[1] = np.uint64(0)
    for j in range(2, 40):                                    # 30 close neighbours of query j in consecutive rows (one group)
        at = int(rng.integers(0, 262_144 - 64))
        for i in range(30):
            codes[at + i] = q[j] ^ (np.uint64(1) << np.uint64(i))
    for j in range(40, 60):                                   # the best ones sit in the LAST rows (behind the bound range if any)
        codes[n - 1 - (j - 40) * 3] = q[j] ^ np.uint64(3)
    codes[n - 1] = q[0]
    ids = rng.permutation(n).astype(np.uint64)
    for flags, the_ids in ((0, ids), (index.APPEND_ONLY, np.arange(n, dtype=np.uint64) * 3 + 1)):
        ix = index.DeviceIndex(index.HAMMING64, 0, flags, gpu_ctx)
        ix.upsert(0, the_ids, codes)
        for k in (1, 10, 64, 65):
            g_ids, _, g_d, g_c = ix.search(0, q, k)
            o_ids, o_d, o_c = oracle.hamming_topk(the_ids, codes, q, k)
            assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids), (n, flags, k)
        ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim,nq,k", [(140_001, 512, 5, 10), (131_072, 768, 16, 1), (150_017, 1024, 11, 64),
                                        (140_001, 256, 6, 10), (133_333, 768, 17, 10), (140_016, 256, 48, 33),
                                        (262_147, 128, 33, 64), (131_073, 512, 16, 65)])
def test_cosine_pruned_pass(gpu_ctx, n, dim, nq, k):
    """5 .. 48 queries over >= 2^17 rows write no key matrix (cosine.hip CosinePrune): chunk minima -> a threshold and ~k
    listed chunks per query -> their keys recomputed -> answer, with the dense pass as the gated fallback.  Both row-stream
    Ragged last chunks, zero rows, duplicated rows whose copies sit in different chunks (ties resolved by id), k up to the
    pass's limit (64) and one above it (dense path), an exact-direction match; and the answer must equal, bit for bit, the one
    of the pass's own fallback (the dense keys of the same kernel + selection; forced here by UCFP_COSINE_PRUNE_FALLBACK):
    the list pass recomputes the same tiles with the same arithmetic."""
    import os
    from ucfp_amd import index
    rng = np.random.default_rng(n + dim + nq)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[rng.integers(0, n, 9)] = 0.0
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    queries[0] = rows[n - 3] * 2.5                      # its best match sits in the ragged last chunk
    for j in range(1, min(nq, 5)):                      # three copies of a near-match, far apart: equal scores, id order decides
        near = (queries[j] + 0.05 * rng.standard_normal(dim)).astype(np.float32)
        for pos in rng.integers(0, n, 3):
            rows[pos] = near
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(5) + np.uint64(1)
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, g_key, g_c = ix.search(0, queries, k)
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    os.environ["UCFP_COSINE_PRUNE_FALLBACK"] = "1"
    try:
        d_ids, d_sc, d_key, d_c = ix.search(0, queries, k)
    finally:
        del os.environ["UCFP_COSINE_PRUNE_FALLBACK"]
    assert np.array_equal(g_c, d_c) and np.array_equal(g_ids, d_ids) and np.array_equal(g_sc, d_sc)
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim,nq,k,cluster", [(140_001, 768, 64, 10, 120), (131_073, 384, 49, 33, 40), (150_000, 128, 57, 64, 30),
                                                (133_000, 1024, 50, 3, 25), (140_016, 256, 48, 33, 120), (131_072, 1152, 30, 10, 30),
                                                (140_000, 192, 130, 10, 30), (262_144, 64, 300, 5, 20)])
def test_cosine_f16_minima(gpu_ctx, n, dim, nq, k, cluster):
    """5 .. 64 queries over rows whose dim is a multiple of 128: the chunk minima come from the f16 matrix pipe
    (cosine_mins_f16), approximate within cosine_mins_eps; the thresholds are widened by it and the listed chunks' exact keys
    decide; batches above 64 queries go in passes.  Near-ties far inside that margin (a cluster of rows whose scores differ
    by 1e-6 .. 1e-3: small clusters fit the chunk lists, 120 rows overflow them and the gated dense pass answers), rows of tiny and of
    huge magnitude, rows with a few large and many f16-subnormal components, zero rows; the answer must equal the f32
    minima's (UCFP_COSINE_NO_F16) bit for bit up to 48 queries (the same exact list pass), and the float64 reference always."""
    import os
    from ucfp_amd import index
    rng = np.random.default_rng(n + dim + nq)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[rng.integers(0, n, 9)] = 0.0
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    for j in range(0, min(nq, 6)):          # a cluster around the query: 120 rows with scores 1 - O(noise^2), noise 1e-3 .. 5e-2
        for t, pos in enumerate(rng.integers(0, n, cluster)):
            rows[pos] = queries[j] + (1e-3 + 4e-4 * t * (120 / cluster)) * rng.standard_normal(dim).astype(np.float32)
    rows[rng.integers(0, n, 50)] *= np.float32(1e-12)      # tiny rows, huge rows: same scores
    rows[rng.integers(0, n, 50)] *= np.float32(1e12)
    spiky = rng.integers(0, n, 200)                          # a few large components, the rest subnormal as f16 after scaling
    rows[spiky] *= np.float32(1e-6)
    rows[spiky, :4] = rng.standard_normal((200, 4)).astype(np.float32)
    queries[6 % nq, 4:] *= np.float32(1e-6)
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(3) + np.uint64(7)
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, g_key, g_c = ix.search(0, queries, k)
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    for env in ("UCFP_COSINE_NO_F16", "UCFP_COSINE_PRUNE_FALLBACK"):
        os.environ[env] = "1"
        try:
            d_ids, d_sc, d_key, d_c = ix.search(0, queries, k)
        finally:
            del os.environ[env]
        if nq <= 48:
            assert np.array_equal(g_c, d_c) and np.array_equal(g_ids, d_ids) and np.array_equal(g_sc, d_sc), env
        else:
            _check_cosine_against_f64(d_ids, d_sc, d_c, ids, rows, queries, k)
    ix.close()


@pytest.mark.gpu
def test_cosine_f16_minima_non_finite_rows(gpu_ctx):
    """Rows with a NaN or an infinite component have no cosine score (the f32 arithmetic drops or zeroes them); the f16 pass
    cannot bound its error on them, raises the fallback flag, and the answer is the f32 path's bit for bit."""
    import os
    from ucfp_amd import index
    rng = np.random.default_rng(12)
    n, dim, nq, k = 135_000, 128, 24, 10
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    rows[17, 3] = np.nan
    rows[90_000, :] = np.inf
    rows[123_456, 5] = -np.inf
    rows[50_000] = queries[2] * np.float32(4.0)              # a finite exact-direction match must still be found
    ids = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g = ix.search(0, queries, k)
    os.environ["UCFP_COSINE_NO_F16"] = "1"
    try:
        d = ix.search(0, queries, k)
    finally:
        del os.environ["UCFP_COSINE_NO_F16"]
    assert np.array_equal(g[3], d[3]) and np.array_equal(g[0], d[0]) and np.array_equal(g[1], d[1], equal_nan=True)
    assert g[0][2, 0] == 50_000 * 2 + 1 and abs(g[1][2, 0] - 1.0) <= COS_TOL
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [640, 896, 576, 832])
def test_cosine_small_shard_dims_between_kernel_instances(gpu_ctx, dim):
    """5-16 queries over a shard below 2^17 rows take the 4x4x1 row-stream kernel, whose instances walk 2 / 4 / 6 / 8 / 12 / 16
    chunks of 64 floats: a dim whose chunk count lies between two instances (577-640, 833-896) runs an instance wider than its
    rows, and the query image has to be as wide (round 4: it was not -- the surplus chunks' operands of the last queries came
    from behind the image, NaN patterns there made NaN scores and the rows were dropped.  Whether it bites depends on what an
    earlier kernel left in LDS: `python tools/soak_index.py --seed 33` reproduced it within 25 s (profiles/r04/soak_index_seed33.txt);
    this test pins the shapes."""
    from ucfp_amd import index
    rng = np.random.default_rng(dim)
    n, nq, k = 40_000, 16, 50
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    ids = rng.permutation(n).astype(np.uint64)
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    for _ in range(3):          # (what lay behind the image depended on the previous launch)
        g_ids, g_sc, _, g_c = ix.search(0, queries, k)
        _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
        g_ids, g_sc, _, g_c = ix.search(0, queries[:9], 10)
        _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries[:9], 10)
    ix.close()


@pytest.mark.gpu
def test_cosine_f16_minima_out_of_range_norms_fall_back(gpu_ctx):
    """A row or a query whose f32 norm is not a finite number in [1e-30, 1e30] (components around 1e20 and above: the sum of
    squares overflows) cannot be scaled into f16 with a bounded error: the f16 pass raises the fallback flag and the dense pass
    answers -- whatever the f32 arithmetic makes of such rows, with or without the f16 minima."""
    import os
    from ucfp_amd import index
    rng = np.random.default_rng(5)
    n, dim, nq, k = 140_000, 256, 20, 10
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    rows[99_999] = queries[5] * np.float32(3e-15)           # fine: norm ~ 5e-14
    ids = np.arange(n, dtype=np.uint64) + np.uint64(10)

    def both(ix, q):
        g = ix.search(0, q, k)
        os.environ["UCFP_COSINE_NO_F16"] = "1"
        try:
            d = ix.search(0, q, k)
        finally:
            del os.environ["UCFP_COSINE_NO_F16"]
        assert np.array_equal(g[3], d[3]) and np.array_equal(g[0], d[0]) and np.array_equal(g[1], d[1], equal_nan=True)
        return g

    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = both(ix, queries)
    _check_cosine_against_f64(g_ids, g_sc, g_c, ids, rows, queries, k)
    assert g_ids[5, 0] == 99_999 + 10
    q2 = queries.copy()
    q2[7] *= np.float32(1e25)                               # |q| = inf in f32
    both(ix, q2)
    ix.close()
    rows[1234] = queries[3] * np.float32(1e25)              # |v| = inf in f32
    rows[77] = queries[4] * np.float32(1e18)                # |v| ~ 1.6e19: in range
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    g_ids, g_sc, _, g_c = both(ix, queries)
    assert g_ids[4, 0] == 77 + 10
    ix.close()


def test_hamming_few_tiles_pipeline_drain(gpu_ctx, oracle):
    """Batches of up to 256 queries run the matrix-core filter as ONE pipeline across code steps (hamming_scan_mfma, stream
    path) whose last fold happens in a drain step per wave.  Tie-heavy data makes nearly every step a suspect, so every
    wave's drain step logs a record: a record damaged there (round 4: its ballots were read back from registers the
    matrix core was still writing) loses a boundary candidate in the last query tile.  Repeated, because the damage
    depended on timing."""
    from ucfp_amd import index
    rng = np.random.default_rng(77)
    n = 752_253
    codes = rng.integers(0, 2**12, n, dtype=np.uint64) * np.uint64(0x0010000100001001)
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(1000)
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, ids, codes)
    for nq, k in ((216, 64), (30, 10), (64, 17), (256, 3)):
        queries = codes[rng.integers(0, n, nq)] ^ (np.uint64(1) << rng.integers(0, 64, nq).astype(np.uint64))
        o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
        for _ in range(6):
            g_ids, _, g_d, g_c = ix.search(0, queries, k)
            assert np.array_equal(g_c, o_c) and np.array_equal(g_d, o_d) and np.array_equal(g_ids, o_ids), (nq, k)
    ix.close()
