"""GPU parity: banded MinHash LSH (band keys, build, query -- through the C ABI) vs the numpy checker
in oracle/ (spec: DESIGN.md "LSH").  Bit-exact ids, scores and counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(slots):
    """uint64 [n, 128] -> 1032-byte MinHash records (header: u16 schema = 1, six zero pad bytes)."""
    n = slots.shape[0]
    rec = np.zeros((n, 1032), np.uint8)
    rec[:, 0] = 1
    rec[:, 8:] = np.ascontiguousarray(slots, dtype="<u8").view(np.uint8).reshape(n, 1024)
    return rec


def _planted(rng, n_bases, variants, n_noise, keep):
    """Clusters of near-duplicates: each variant keeps a slot of its base with probability `keep`."""
    bases = rng.integers(0, 1 << 63, size=(n_bases, 128), dtype=np.uint64)
    rows = [rng.integers(0, 1 << 63, size=(n_noise, 128), dtype=np.uint64)]
    for v in range(variants):
        mask = rng.random((n_bases, 128)) < keep
        rows.append(np.where(mask, bases, rng.integers(0, 1 << 63, size=(n_bases, 128), dtype=np.uint64)))
    corpus = np.concatenate(rows)
    perm = rng.permutation(corpus.shape[0])
    return bases, corpus[perm]


@pytest.mark.parametrize("bands,rows", [(16, 8), (32, 4), (8, 16), (20, 6), (1, 64), (2, 64), (128, 1), (5, 3)])
def test_band_keys(gpu_ctx, oracle, bands, rows):
    from ucfp_amd import text
    rng = np.random.default_rng(bands * 100 + rows)
    rec = _records(rng.integers(0, 1 << 64, size=(777, 128), dtype=np.uint64))
    got = text.lsh_band_keys(rec, bands, rows)
    assert got.shape == (777, bands) and np.array_equal(got, oracle.lsh_band_keys(rec, bands, rows))


def test_band_keys_of_real_signatures(gpu_ctx, oracle):
    from ucfp_amd import text
    docs = ["the quick brown fox jumps over the lazy dog number %d and then some more words" % i for i in range(50)]
    rec, st = text.minhash_batch(docs)
    assert not st.any()
    assert np.array_equal(text.lsh_band_keys(rec), oracle.lsh_band_keys(rec))


@pytest.mark.parametrize("bands,rows,k,keep", [(16, 8, 10, 0.9), (32, 4, 5, 0.7), (8, 16, 128, 0.95), (20, 6, 1, 0.8)])
def test_query_matches_checker(gpu_ctx, oracle, bands, rows, k, keep):
    from ucfp_amd import text
    rng = np.random.default_rng(1000 + bands)
    bases, corpus = _planted(rng, n_bases=150, variants=6, n_noise=20000, keep=keep)
    ids = rng.permutation(np.arange(10_000, 10_000 + corpus.shape[0], dtype=np.uint64))
    crec, qrec = _records(corpus), _records(bases)
    idx = text.LshIndex(bands, rows)
    idx.build(ids, crec)
    g_ids, g_sc, g_ct = idx.query(qrec, k)
    o_ids, o_sc, o_ct = oracle.lsh_query(ids, crec, qrec, k, bands, rows)
    assert np.array_equal(g_ct, o_ct)
    assert np.array_equal(g_ids, o_ids)
    assert np.array_equal(g_sc, o_sc)
    assert g_ct.max() == min(k, 6) or g_ct.max() > 1   # the planted clusters are found
    idx.close()


@pytest.mark.parametrize("cand_per_band", [1, 64, 100, 5000])
def test_heavy_buckets_and_candidate_cap(gpu_ctx, oracle, cand_per_band):
    """3000 identical rows + a few near copies: exercises the per-band cap, the 1024-candidate cap
    and de-duplication across bands."""
    from ucfp_amd import text
    rng = np.random.default_rng(7)
    base = rng.integers(0, 1 << 63, size=(1, 128), dtype=np.uint64)
    same = np.repeat(base, 3000, axis=0)
    near = np.repeat(base, 40, axis=0)
    near[np.arange(40), rng.integers(0, 128, 40)] ^= np.uint64(1)
    noise = rng.integers(0, 1 << 63, size=(500, 128), dtype=np.uint64)
    corpus = np.concatenate([noise, near[:20], same, near[20:]])
    ids = rng.permutation(np.arange(corpus.shape[0], dtype=np.uint64)) + np.uint64(1 << 40)
    crec, qrec = _records(corpus), _records(np.concatenate([base, near[:3], noise[:2]]))
    idx = text.LshIndex(16, 8, cand_per_band)
    idx.build(ids, crec)
    for k in (1, 17, 128):
        g = idx.query(qrec, k)
        o = oracle.lsh_query(ids, crec, qrec, k, 16, 8, cand_per_band)
        for a, b in zip(g, o):
            assert np.array_equal(a, b)


def test_rebuild_empty_and_edge_sizes(gpu_ctx, oracle):
    from ucfp_amd import text
    from ucfp_amd.errors import InvalidArgument
    rng = np.random.default_rng(3)
    q = _records(rng.integers(0, 1 << 63, size=(5, 128), dtype=np.uint64))
    idx = text.LshIndex()
    ids, sc, ct = idx.query(q, 4)                       # never built
    assert not ct.any() and (ids == np.uint64(2**64 - 1)).all() and (sc == -1).all()
    idx.build(np.arange(5, dtype=np.uint64), q)          # query == corpus: every record finds itself
    ids, sc, ct = idx.query(q, 4)
    assert (ct == 1).all() and np.array_equal(ids[:, 0], np.arange(5, dtype=np.uint64)) and (sc[:, 0] == 1).all()
    idx.build(np.zeros(0, np.uint64), np.zeros((0, 1032), np.uint8))   # rebuild to empty
    assert not idx.query(q, 4)[2].any()
    assert idx.query(np.zeros((0, 1032), np.uint8), 4)[0].shape == (0, 4)
    with pytest.raises(InvalidArgument):
        idx.query(q, 129)
    with pytest.raises(InvalidArgument):
        text.LshIndex(32, 8)                              # 256 slots > 128
    with pytest.raises(InvalidArgument):
        text.LshIndex(1, 65)


def test_near_duplicate_documents_found(gpu_ctx):
    """End to end on text: an edited copy retrieves its original first."""
    import random
    from ucfp_amd import text
    rng = random.Random(11)
    vocab = ["w%03d" % i for i in range(400)]
    docs = [" ".join(rng.choice(vocab) for _ in range(120)) for _ in range(2000)]
    rec, st = text.minhash_batch(docs)
    assert not st.any()
    idx = text.LshIndex(32, 4)
    idx.build(np.arange(len(docs), dtype=np.uint64), rec)
    picks = list(range(0, 2000, 40))
    edited = []
    for p in picks:
        w = docs[p].split(" ")
        for _ in range(3):
            w[rng.randrange(len(w))] = "edit"
        edited.append(" ".join(w))
    qrec, st = text.minhash_batch(edited)
    ids, sc, ct = idx.query(qrec, 3)
    assert (ct >= 1).all() and np.array_equal(ids[:, 0], np.array(picks, np.uint64))
    assert (sc[:, 0] > 0.6).all()
