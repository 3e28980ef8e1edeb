"""Every value the reference's own tests / UI decoders pin at the SDK boundary,
checked against the CPU oracle (and, on the GPU box, against the HIP path in test_golden.py)."""
import numpy as np
import pytest


def _synthetic_png(w, h):   # src/server/tests.rs:227-235
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([xx % 256, yy % 256, np.full_like(xx, 128)], axis=-1).astype(np.uint8)


def test_multihash_bundle_is_536_bytes(oracle):
    rec, st = oracle.image_hash_batch(_synthetic_png(64, 64)[None], 7, pixfmt=1)   # tests.rs:1170,1206-1207
    assert rec.shape == (1, 536) and st[0] == 0 and len(rec[0].tobytes().hex()) == 1072


def test_single_image_fingerprint_is_168_bytes_and_matches_bundle_slots(oracle):
    img = _synthetic_png(64, 64)[None]
    multi, _ = oracle.image_hash_batch(img, 7, pixfmt=1)
    # MultiHashFingerprint = exact | ahash | phash | dhash  (AlgorithmView.svelte:30-37)
    for algo, off in ((1, 32), (2, 200), (4, 368)):
        one, _ = oracle.image_hash_batch(img, algo, pixfmt=1)
        assert one.shape == (1, 168)                                   # algorithmView.ts:11-17
        assert np.array_equal(one[0], multi[0, off:off + 168])
    # ImageFingerprint = exact[32] | global u64 | 16 block u64  (ImageHashView.svelte:2-5,26-30)
    assert 32 + 8 + 16 * 8 == 168


def test_exact_prefix_is_carried_into_every_slot(oracle):
    ex = np.arange(32, dtype=np.uint8)[None]
    rec, _ = oracle.image_hash_batch(_synthetic_png(64, 64)[None], 7, pixfmt=1, exact=ex)
    for off in (0, 32, 200, 368):
        assert np.array_equal(rec[0, off:off + 32], ex[0])


def test_ahash_mean_is_integer_mean_of_gray8(oracle):
    """inspect_image's `ahash_mean = sum / 64` (src/modality/image.rs:317-318)."""
    rng = np.random.default_rng(1)
    fr = rng.integers(0, 256, (256, 256), dtype=np.uint8)
    norm = oracle.image_normalize(fr)
    g8 = norm.reshape(8, 32, 8, 32).astype(np.int64).sum(axis=(1, 3))
    g8 = (2 * g8 + 1024) // 2048
    mean = int(g8.sum()) // 64
    bits = 0
    for i, v in enumerate(g8.reshape(-1)):
        if v > mean:
            bits |= 1 << i
    assert int(oracle.image_hashes17(norm, 1)[0]) == bits


def test_minhash_layout_and_length(oracle):
    rec, st = oracle.text_minhash_batch([b"Hello world, this is a test of the pipeline inspector."])
    assert rec.shape == (1, 1032) and st[0] == 0                       # tests.rs:1114-1118,1162
    assert bytes(rec[0, :8]) == b"\x01\x00\x00\x00\x00\x00\x00\x00"    # schema u16 = 1 + 6 pad


@pytest.mark.xfail(strict=True, reason="EXTERNAL PIN, UNMET: txtfp 0.2.0's slot derivation is not recoverable "
                   "offline (SURVEY 8c probe); our MinHash matches layout/primitive, not txtfp's bits")
def test_minhash_golden_prefix_of_reference(oracle):
    rec, _ = oracle.text_minhash_batch([b"the quick brown fox jumps over the lazy dog"])
    assert rec[0, :16].tobytes().hex() == "0100000000000000a26accc88c8a8106"   # tests.rs:1153-1157


def test_default_config_hash_is_a_carried_constant_not_a_pin():
    """tests.rs:1158-1161 shows txtfp::config_hash for ONE configuration; the function itself is unavailable and
    unrecovered, so the value is carried verbatim -- this test only guards the constant and that
    nothing is invented for any other configuration."""
    from ucfp_amd import text
    from ucfp_amd.errors import UnsupportedError
    opts = text.TextOpts()
    assert opts.tokenizer_tag() == "shingle-k=5/word-uax29"            # text.rs:152-159
    assert text.config_hash(opts.canonicalizer, opts.tokenizer_tag(), text.ALGORITHM_MINHASH_128) == \
        2_212_816_233_060_047_056
    for canon, tag, alg in ((text.Canonicalizer(normalization="nfc"), opts.tokenizer_tag(), text.ALGORITHM_MINHASH_128),
                            (opts.canonicalizer, "shingle-k=3/word-uax29", text.ALGORITHM_MINHASH_128),
                            (opts.canonicalizer, "word-uax29", text.ALGORITHM_SIMHASH_TF)):
        with pytest.raises(UnsupportedError):
            text.config_hash(canon, tag, alg)


def test_simhash_is_8_bytes(oracle):
    rec, st = oracle.text_simhash_batch([b"the quick brown fox"])
    assert rec.shape == (1, 8) and st[0] == 0                          # algorithmView.ts:18


def test_wang_hash_layout(oracle):
    """8 bytes per hash; u32 LE f_a(9)|f_b(9)|dt(14) with the anchor in bits 31..23, then u32 LE
    t_anchor (LandmarkScatter.svelte:4,31-37); record length a multiple of 8 (algorithmView.ts:22)."""
    sr = 8000
    t = np.arange(4 * sr) / sr
    rng = np.random.default_rng(0)
    x = (0.3 * np.sin(2 * np.pi * (300 + 200 * t) * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    h = oracle.wang(x)
    assert h.dtype == np.uint32 and h.shape[1] == 2 and h.shape[0] > 0 and len(h.tobytes()) % 8 == 0
    fa, fb, dt = h[:, 0] >> 23, (h[:, 0] >> 14) & 0x1FF, h[:, 0] & 0x3FFF
    assert fa.max() < 512 and fb.max() < 512 and dt.min() >= 1 and dt.max() <= 63      # target_zone_t default
    assert (np.abs(fa.astype(int) - fb.astype(int)) <= 64).all()                       # target_zone_f default
    assert h[:, 1].max() < 4 * 62.5                                                    # 62.5 frames / s
    # at most fan_out hashes per anchor (audio.rs:996-999)
    _, counts = np.unique((fa.astype(np.uint64) << np.uint64(32)) | h[:, 1].astype(np.uint64), return_counts=True)
    assert counts.max() <= 10


def test_audio_inspect_fixture_has_peaks(oracle):
    """440 Hz, 1 s @ 8 kHz must yield peaks (tests.rs:1215-1261 `total_peaks > 0`)."""
    sr = 8000
    t = np.arange(sr, dtype=np.float32) / sr
    x = (np.sin(2 * np.pi * 440.0 * t) * 0.5).astype(np.float32)
    P = oracle.stft_power(x, 1024, 128)
    pt, pk, pp = oracle.wang_peaks(P)
    assert len(pt) > 0
    assert 56 in set(pk.tolist()) or 57 in set(pk.tolist())            # 440 Hz / 7.8125 Hz per bin


def test_haitsma_rate_is_312_bytes_per_second(oracle):
    x = np.zeros(5000 * 10, np.float32)
    fr = oracle.haitsma(x, 5000)
    assert abs(fr.size * 4 / 10 - 312.5) < 15                          # algorithms_manifest.rs:654


def test_cosine_knn_reference_toy(oracle):
    """src/index/embedded/mod.rs:522-544: [0.6,0.6,0] -> 300 first, strictly better than the rest."""
    ids = np.array([100, 200, 300], np.uint64)
    rows = np.array([[1, 0, 0], [0, 1, 0], [0.7, 0.7, 0]], np.float32)
    q = np.array([0.6, 0.6, 0.0], np.float32)
    for fold in (False, True):
        got, sc = oracle.cosine_knn(ids, rows, q, 2, ref_fold=fold)
        assert len(got) == 2 and got[0] == 300 and sc[0] > sc[1]
    # empty query / k = 0 / zero-norm query -> no hits (:275-286)
    assert len(oracle.cosine_knn(ids, rows, np.zeros(3, np.float32), 2)[0]) == 0
    assert len(oracle.cosine_knn(ids, rows, q, 0)[0]) == 0


def test_cosine_total_order_equals_reference_fold_without_ties(oracle):
    rng = np.random.default_rng(2)
    rows = rng.standard_normal((400, 37)).astype(np.float32)
    rows[5] = 0
    ids = rng.permutation(400).astype(np.uint64)
    for _ in range(5):
        q = rng.standard_normal(37).astype(np.float32)
        a = oracle.cosine_knn(ids, rows, q, 10)
        b = oracle.cosine_knn(ids, rows, q, 10, ref_fold=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_dot_product_accumulation_order_is_the_references(oracle):
    """8 independent lanes over chunks of 8, lanes summed in order, then the remainder
    (src/index/embedded/mod.rs:454-472) -- restated in numpy f32 and compared bit for bit."""
    rng = np.random.default_rng(3)
    for dim in (3, 8, 19, 768):
        a = rng.standard_normal(dim).astype(np.float32)
        b = rng.standard_normal(dim).astype(np.float32)
        accs = np.zeros(8, np.float32)
        for c in range(dim // 8):
            accs += a[8 * c:8 * c + 8] * b[8 * c:8 * c + 8]
        s = np.float32(0)
        for j in range(8):
            s = np.float32(s + accs[j])
        for i in range(8 * (dim // 8), dim):
            s = np.float32(s + np.float32(a[i] * b[i]))
        na = np.float32(0)
        # cos(a, a*2) via the oracle exposes dot and both norms: score = dot/(|a||2a|)
        ids = np.array([1], np.uint64)
        got_ids, got_sc = oracle.cosine_knn(ids, b[None, :], a, 1)
        accs_a = np.zeros(8, np.float32)
        accs_b = np.zeros(8, np.float32)
        for c in range(dim // 8):
            accs_a += a[8 * c:8 * c + 8] * a[8 * c:8 * c + 8]
            accs_b += b[8 * c:8 * c + 8] * b[8 * c:8 * c + 8]
        sa, sb = np.float32(0), np.float32(0)
        for j in range(8):
            sa, sb = np.float32(sa + accs_a[j]), np.float32(sb + accs_b[j])
        for i in range(8 * (dim // 8), dim):
            sa = np.float32(sa + np.float32(a[i] * a[i]))
            sb = np.float32(sb + np.float32(b[i] * b[i]))
        expect = np.float32(s / np.float32(np.sqrt(sa) * np.sqrt(sb)))
        assert got_sc[0] == expect, (dim, got_sc[0], expect)
        del na
