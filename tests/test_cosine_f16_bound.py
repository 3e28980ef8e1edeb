"""The error bound the f16 cosine minima rest on (ucfp_amd/csrc/cosine.hip, cosine_mins_f16 / cosine_mins_eps), restated in numpy
and checked on the CPU: the kernel's arithmetic -- rows and queries scaled by the reciprocal of their f32 norm, rounded to f16
(round to nearest even, subnormals kept), multiplied exactly, accumulated in f32 -- stays within eps(dim) of the exact score
for random, clustered, spiky, tiny / huge and worst-case-rounding inputs.  (The GPU side: tests/test_index_gpu.py
test_cosine_f16_minima*, tools/probe_mfma_f16_denorm.hip for the matrix pipe's treatment of subnormals.)"""
import numpy as np
import pytest


def eps(dim):
    return 9.9e-4 + 1.25e-7 * dim          # cosine.hip cosine_mins_eps


def f32_norm(x):
    # cosine_norms: 64 lanes, fma chains over i = lane, lane + 64, ..., then a butterfly sum; any f32 order is within dim 2^-24
    return np.sqrt(np.float32(np.sum(x.astype(np.float32) ** 2, dtype=np.float32)))


def approx_scores(q, rows):
    """score~ of one query against rows, as cosine_mins_f16 computes it"""
    qn, rn = f32_norm(q), np.array([f32_norm(r) for r in rows], np.float32)
    qh = (q.astype(np.float32) * (np.float32(1) / qn)).astype(np.float16)
    rh = (rows.astype(np.float32) * (np.float32(1) / rn)[:, None]).astype(np.float16)
    # f16 x f16 products are exact in f32; the sum in f32 (the order is the matrix pipe's: any order is within dim 2^-24)
    return (rh.astype(np.float32) * qh.astype(np.float32)[None, :]).sum(axis=1, dtype=np.float32)


def exact_scores(q, rows):
    q64, r64 = q.astype(np.float64), rows.astype(np.float64)
    return (r64 @ q64) / (np.linalg.norm(r64, axis=1) * np.linalg.norm(q64))


@pytest.mark.parametrize("dim", [64, 128, 384, 768, 1024, 1536])
def test_f16_minima_error_bound_random_and_structured(dim):
    rng = np.random.default_rng(dim)
    q = rng.standard_normal(dim).astype(np.float32)
    rows = [rng.standard_normal((200, dim)).astype(np.float32)]
    rows.append((q[None, :] + 1e-3 * rng.standard_normal((50, dim))).astype(np.float32))            # near-matches: score ~ 1
    rows.append((-q[None, :] + 1e-2 * rng.standard_normal((20, dim))).astype(np.float32))           # score ~ -1
    spiky = (1e-6 * rng.standard_normal((50, dim))).astype(np.float32)                              # f16 subnormals after scaling
    spiky[:, :4] = rng.standard_normal((50, 4))
    rows.append(spiky)
    rows.append((rows[0][:40] * np.float32(1e-12)).astype(np.float32))                              # tiny and huge rows
    rows.append((rows[0][:40] * np.float32(1e12)).astype(np.float32))
    rows.append(np.abs(rng.standard_normal((40, dim))).astype(np.float32))                          # one sign: nothing cancels
    rows = np.concatenate(rows)
    for qq in (q, np.abs(q), spiky[0]):
        err = np.abs(approx_scores(qq, rows).astype(np.float64) - exact_scores(qq, rows))
        assert err.max() <= eps(dim), (dim, err.max(), eps(dim))


def test_f16_minima_error_bound_worst_case_rounding():
    """q = v = constant vectors: every component is 1 / sqrt(dim) after scaling, every product carries the same two f16
    roundings in the same direction and nothing cancels -- the case the 2^-10 term of the bound is for.  Over all the dims
    the kernel takes (multiples of 64 up to 2048) the error stays below eps, and it comes close for the worst dim."""
    worst = 0.0
    for dim in range(64, 2049, 64):
        v = np.ones((1, dim), np.float32)
        err = abs(float(approx_scores(v[0], v)[0]) - 1.0)
        assert err <= eps(dim), (dim, err)
        worst = max(worst, err / eps(dim))
    assert worst > 0.3          # (the bound is not slack by an order of magnitude: 0.71 of eps at dim 960, error 7.9e-4)


def test_f16_minima_error_bound_adversarial_search():
    """A crude search for bad cases: components drawn from the f16 rounding midpoints' neighbourhood, all of one sign."""
    rng = np.random.default_rng(3)
    dim = 768
    worst = 0.0
    for _ in range(40):
        base = rng.uniform(0.5, 1.0)
        # values (1 + (2 m + 1) 2^-11) 2^e: halfway between two f16 numbers before the scaling moves them
        m = rng.integers(0, 1024, dim)
        v = ((1.0 + (2 * m + 1) * 2.0 ** -11) * base).astype(np.float32)
        q = v * rng.uniform(0.9, 1.1)
        err = abs(float(approx_scores(q.astype(np.float32), v[None, :])[0]) - float(exact_scores(q, v[None, :])[0]))
        worst = max(worst, err)
    assert worst <= eps(dim), worst


# ---- the thresholds' widening (ucfp_amd/csrc/topk.hip relax_key), restated -------------------------------------------------
def score_to_key(s):
    u = np.asarray(s, np.float32).view(np.uint32)
    u = np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000))
    return ~u


def key_to_score(k):
    u = ~np.asarray(k, np.uint32)
    u = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u)
    return u.view(np.float32)


def relax_key(key, d):
    key = np.asarray(key, np.uint32)
    r = score_to_key(key_to_score(key) - np.float32(d))
    r = np.where(r < key, key, r)
    return np.where(key == np.uint32(0xFFFFFFFF), key, r)


def test_keys_are_the_inverted_order_image_of_the_score_and_relaxing_only_widens():
    rng = np.random.default_rng(9)
    s = np.concatenate([rng.uniform(-1, 1, 4000), [0.0, 1.0, -1.0, 1e-30, -1e-30, 0.177, 0.9999999]]).astype(np.float32)
    k = score_to_key(s)                 # (-0.0 sorts one key behind +0.0: the order image of the bits, harmless)
    assert np.array_equal(key_to_score(k).view(np.uint32), s.view(np.uint32))                       # a bijection on the bits
    o = np.argsort(s, kind="stable")
    assert np.all(np.diff(k[o].astype(np.int64)) <= 0)                                              # larger score, smaller key
    for d in (0.0, 1.1e-3, 2.2e-3):
        r = relax_key(k, d)
        assert np.all(r >= k)                                                                       # never tighter
        # everything within d of the threshold's score passes the widened threshold: key(s') <= relax(key(s), d) for s' >= s - d
        sp = (s.astype(np.float64) - d * rng.uniform(0, 1, s.size)).astype(np.float32)
        sp = np.maximum(sp, (s - np.float32(d)).astype(np.float32))
        assert np.all(score_to_key(sp) <= r)
    assert relax_key(np.uint32(0xFFFFFFFF), 1e-3) == np.uint32(0xFFFFFFFF)                          # "no threshold" stays
