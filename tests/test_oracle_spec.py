"""Independent numpy / scipy / regex / xxhash cross-checks of the CPU oracle, so the oracle is not
only self-consistent: each stage of the spec is re-derived a second way."""
import random

import numpy as np


def test_image_normalize_is_exact_box_mean(oracle):
    rng = np.random.default_rng(0)
    for s in (1, 2, 3, 4):
        fr = rng.integers(0, 256, (256 * s, 256 * s), dtype=np.uint8)
        box = fr.reshape(256, s, 256, s).astype(np.int64).sum(axis=(1, 3))
        assert np.array_equal(oracle.image_normalize(fr), (2 * box + s * s) // (2 * s * s))


def test_image_normalize_general_geometry_against_float_area(oracle):
    """Arbitrary geometry: compare with an independent float64 area-overlap resample (+-1 LSB only
    where the exact value sits within 1e-9 of a rounding boundary)."""
    rng = np.random.default_rng(1)
    for (w, h) in ((300, 200), (33, 47), (640, 480), (100, 1000)):
        fr = rng.integers(0, 256, (h, w), dtype=np.uint8)
        wy = np.zeros((256, h))
        wx = np.zeros((256, w))
        for j in range(256):
            for y in range(h):
                wy[j, y] = max(0.0, min((y + 1) * 256, (j + 1) * h) - max(y * 256, j * h))
        for i in range(256):
            for x in range(w):
                wx[i, x] = max(0.0, min((x + 1) * 256, (i + 1) * w) - max(x * 256, i * w))
        exact = wy @ fr.astype(np.float64) @ wx.T / (w * h)
        got = oracle.image_normalize(fr).astype(np.float64)
        assert np.abs(got - exact).max() <= 0.5 + 1e-9


def test_rgb_luma_is_bt601_integer(oracle):
    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    r64 = rgb.astype(np.int64)
    y = (77 * r64[..., 0] + 150 * r64[..., 1] + 29 * r64[..., 2] + 128) >> 8
    assert np.array_equal(oracle.image_normalize(rgb, pixfmt=1), y)


def test_phash_dct_matches_scipy(oracle):
    from scipy.fft import dctn
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    co = oracle.image_phash_coefs(g)
    ref = dctn(g.astype(np.float64), norm="ortho")[:8, :8]
    assert np.abs(co - ref).max() < 2e-3
    # hash = coefficient > median of the 63 AC terms; re-derived from the oracle's own coefficients
    norm = rng.integers(0, 256, (256, 256), dtype=np.uint8)
    g32 = oracle.image_region_gray32(norm, 0)
    assert np.array_equal(g32, (norm.reshape(32, 8, 32, 8).astype(np.int64).sum(axis=(1, 3)) + 32) // 64)
    c = oracle.image_phash_coefs(g32).reshape(-1)
    med = np.sort(c[1:])[31]
    bits = sum(1 << i for i in range(64) if c[i] > med)
    assert int(oracle.image_hashes17(norm, 2)[0]) == bits


def test_dhash_uses_exact_9x8_area_resample(oracle):
    rng = np.random.default_rng(4)
    norm = rng.integers(0, 256, (256, 256), dtype=np.uint8)
    rows = norm.reshape(8, 32, 256).astype(np.int64).sum(axis=1)          # 8 x 256 column sums
    px = np.zeros((8, 9), np.int64)
    for c in range(9):
        for x in range(256):
            ov = max(0, min(9 * x + 9, 256 * (c + 1)) - max(9 * x, 256 * c))
            px[:, c] += ov * rows[:, x]
    px = (px * 8 * 2 + 65536) // (2 * 65536)
    bits = sum(1 << (r * 8 + c) for r in range(8) for c in range(8) if px[r, c] > px[r, c + 1])
    assert int(oracle.image_hashes17(norm, 4)[0]) == bits


def test_xxh3_matches_xxhash_module_for_every_length_class(oracle):
    import xxhash
    rnd = random.Random(1)
    for n in list(range(0, 260)) + [511, 512, 513, 1023, 1024, 1025, 2048, 4097, 10000]:
        d = bytes(rnd.getrandbits(8) for _ in range(n))
        assert oracle.xxh3_64(d) == xxhash.xxh3_64_intdigest(d), n


def test_ascii_tokenizer_is_uax29(oracle):
    """Against the `regex` module's UAX#29 word boundaries. '_' is treated as a letter by our ASCII
    path and the module keeps a pre-Unicode-11 apostrophe rule, so both are kept out
    of the random alphabet; the hand-written cases cover them explicitly."""
    import regex
    rnd = random.Random(3)

    def ref(s):
        return [t.lower() for t in regex.split(r"(?w)\b", s, flags=regex.V1) if regex.search(r"[A-Za-z0-9]", t)]
    alpha = "abcXYZ019 .,;:\"-!?()\n\t/@#"
    for _ in range(4000):
        s = "".join(rnd.choice(alpha) for _ in range(rnd.randint(0, 40)))
        c, nt = oracle.text_canon(s.encode())
        mine = c.decode().split(" ") if c else []
        assert mine == ref(s) and nt == len(mine), repr(s)
    c, nt = oracle.text_canon(b"don't stop 3.14 1,000,000 a.b.c x:y e.g. U.S.A. foo_bar 12:30 it's 'quoted'")
    assert c.decode().split(" ") == ["don't", "stop", "3.14", "1,000,000", "a.b.c", "x:y", "e.g", "u.s.a",
                                    "foo_bar", "12", "30", "it's", "quoted"]
    assert oracle.text_canon("café".encode())[1] == -1      # non-ASCII: host path


def test_minhash_definition_recomputed_in_python(oracle):
    import xxhash
    doc = b"One two, three four five six seven. Eight nine ten!"
    toks = oracle.text_canon(doc)[0].decode().split(" ")
    M = (1 << 64) - 1
    slots = [M] * 128
    for s in range(len(toks) - 4):
        h1 = xxhash.xxh3_64_intdigest(" ".join(toks[s:s + 5]).encode())
        z = (h1 + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        h2 = (z ^ (z >> 31)) | 1
        for i in range(128):
            slots[i] = min(slots[i], (h1 + i * h2) & M)
    rec, _ = oracle.text_minhash_batch([doc])
    assert rec[0, 8:].copy().view(np.uint64).tolist() == slots
    # SimHash: bit b set iff more than half of the tokens have bit b set
    ones = [0] * 64
    for t in toks:
        h = xxhash.xxh3_64_intdigest(t.encode())
        for b in range(64):
            ones[b] += (h >> b) & 1
    v = sum(1 << b for b in range(64) if 2 * ones[b] > len(toks))
    sh, _ = oracle.text_simhash_batch([doc])
    assert int(sh[0].copy().view(np.uint64)[0]) == v


def test_stft_power_matches_numpy_fft(oracle):
    rng = np.random.default_rng(5)
    x = (0.3 * rng.standard_normal(8000)).astype(np.float32)
    for n_fft, hop in ((1024, 128), (2048, 64)):
        P = oracle.stft_power(x, n_fft, hop)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
        fr = np.stack([x[i * hop:i * hop + n_fft] for i in range(P.shape[0])]).astype(np.float64) * w
        ref = np.abs(np.fft.rfft(fr, axis=1))[:, :n_fft // 2] ** 2
        assert P.shape == ref.shape and np.abs(P - ref).max() <= 2e-5 * ref.max()


def test_wang_pairing_rule_restated_in_python(oracle):
    """src/modality/audio.rs:973-1001, re-run on the oracle's own peaks."""
    rng = np.random.default_rng(6)
    t = np.arange(5 * 8000) / 8000.0
    x = (0.3 * np.sin(2 * np.pi * (200 + 300 * t) * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    P = oracle.stft_power(x, 1024, 128)
    pt, pk, pp = oracle.wang_peaks(P)
    floor = np.float32(65536.0 * 10 ** (-50.0 / 10))
    out = []
    for i in range(len(pt)):
        if not pp[i] >= floor:
            continue
        taken = 0
        for j in range(i + 1, len(pt)):
            dt = int(pt[j]) - int(pt[i])
            if dt <= 0:
                continue
            if dt > 63:
                break
            if abs(int(pk[j]) - int(pk[i])) > 64:
                continue
            out.append(((int(pk[i]) << 23) | (int(pk[j]) << 14) | dt, int(pt[i])))
            taken += 1
            if taken >= 10:
                break
    assert oracle.wang(x).tolist() == [list(p) for p in out]
    # peaks: time-sorted, <= 30 per second, each a strict neighbourhood maximum
    assert (np.diff(pt.astype(np.int64)) >= 0).all()
    secs = (pt.astype(np.int64) * 128) // 8000
    assert np.bincount(secs).max() <= 30
    for i in range(0, len(pt), 7):
        tt, kk = int(pt[i]), int(pk[i])
        win = P[max(0, tt - 7):tt + 8, max(0, kk - 15):kk + 16]
        assert pp[i] == win.max()


def test_resample_linear_matches_numpy_interp(oracle):
    rng = np.random.default_rng(8)
    x = rng.standard_normal(4410).astype(np.float32)
    y = oracle.resample_linear(x, 44100, 8000)
    pos = np.arange(y.size) * 44100 / 8000
    ref = np.interp(pos, np.arange(x.size), x.astype(np.float64))
    assert y.size == 4410 * 8000 // 44100 and np.abs(y - ref).max() < 1e-6


def test_lsh_checker_against_pure_python(oracle):
    """The numpy LSH checker agrees with a loop-level restatement."""
    M = (1 << 64) - 1
    rng = np.random.default_rng(9)
    slots = rng.integers(0, 1 << 64, size=(60, 128), dtype=np.uint64)
    slots[30:40] = slots[0]                       # a bucket of identical signatures
    slots[40:50] = slots[1]
    slots[40:50, :32] ^= np.uint64(7)             # near copies: bands 0-3 differ, 96 of 128 slots survive
    rec = np.zeros((60, 1032), np.uint8)
    rec[:, 8:] = slots.view(np.uint8).reshape(60, 1024)

    def key(row, b, rows):
        h = 0xCBF29CE484222325
        for r in range(rows):
            h = ((h ^ int(slots[row, b * rows + r])) * 0x100000001B3) & M
        h ^= h >> 30
        h = (h * 0xBF58476D1CE4E5B9) & M
        h ^= h >> 27
        h = (h * 0x94D049BB133111EB) & M
        return h ^ (h >> 31)

    for bands, rows in ((16, 8), (20, 6), (2, 64)):
        k = oracle.lsh_band_keys(rec, bands, rows)
        assert all(int(k[i, b]) == key(i, b, rows) for i in (0, 17, 59) for b in range(bands))
    ids = np.arange(100, 160, dtype=np.uint64)
    o_ids, o_sc, o_ct = oracle.lsh_query(ids, rec, rec[:2], 5, 16, 8)
    assert o_ct.tolist() == [5, 5]
    assert o_ids[0].tolist() == [100, 130, 131, 132, 133] and (o_sc[0] == 1.0).all()   # ties: id ascending
    assert o_ids[1, 0] == 101 and o_sc[1, 0] == 1.0 and (o_sc[1, 1:] == 0.75).all()
    # the per-band cap keeps the first rows of a run
    o_ids, _, o_ct = oracle.lsh_query(ids, rec, rec[:1], 128, 16, 8, cand_per_band=3)
    assert o_ct[0] == 3 and o_ids[0, :3].tolist() == [100, 130, 131]


def test_resample_weight_reciprocal_form_is_exact():
    """The Wang stream kernel computes the A1 interpolation weight as (float)((double)rem * (1.0 / sr_out)) instead of
    the oracle's (float)((double)rem / (double)sr_out): identical floats for every remainder (sr_out = 8000, 5000)."""
    for sr in (8000, 5000):
        rem = np.arange(sr, dtype=np.float64)
        a = (rem / np.float64(sr)).astype(np.float32)
        b = (rem * (np.float64(1.0) / np.float64(sr))).astype(np.float32)
        assert np.array_equal(a, b)


def test_timed_hamming_scan_equals_the_plain_one(oracle):
    """The CPU baseline's OpenMP scan (AVX-512 VPOPCNTDQ tile where the CPU has it, scalar tile otherwise) returns exactly
    what the plain oracle returns: ties, copies, slices that are not multiples of the 32-code vector trip."""
    rng = np.random.default_rng(77)
    for n, nq, k in ((1, 2, 10), (31, 3, 4), (1000, 3, 10), (65_537, 5, 1), (200_003, 70, 10), (150_000, 9, 37)):
        codes = rng.integers(0, 2**64, n, dtype=np.uint64)
        if n == 65_537:
            codes[:] = 7
        ids = rng.permutation(n).astype(np.uint64)
        q = rng.integers(0, 2**64, nq, dtype=np.uint64)
        want = oracle.hamming_topk(ids, codes, q, k)
        for scalar in (False, True):
            got = oracle.hamming_topk_omp(ids, codes, q, k, force_scalar=scalar)
            assert all(np.array_equal(a, b) for a, b in zip(want, got)), (n, nq, k, scalar)
