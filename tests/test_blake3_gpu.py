"""BLAKE3 on the device (ucfp_blake3_batch_dev) against the host statement of the same function (ucfp_blake3, itself
checked against the official test vectors in tests/test_abi.py), and the PNG front end filling the records' `exact`
field from it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_batch_matches_host_blake3_on_every_length_class(gpu_ctx):
    from ucfp_amd.blake3 import blake3_batch, blake3_digest
    rng = np.random.default_rng(3)
    # block and chunk boundaries, 2 / 3 / 5 / 6 / 7 chunks (every tree shape up to three levels), 64 + 1 and 127 chunks (a
    # lane takes two), a long input, and the official test-vector pattern (byte i = i mod 251)
    lens = [0, 1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 3073, 4096, 5000,
            5 * 1024 + 1, 6 * 1024, 7 * 1024 - 1, 8 * 1024, 31744, 65 * 1024, 65 * 1024 + 7, 127 * 1024 + 513, 300_001,
            1_048_577]
    items = [bytes((np.arange(n) % 251).astype(np.uint8)) for n in lens]
    items += [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in rng.integers(0, 200_000, 40)]
    got = blake3_batch(items, ctx=gpu_ctx)
    for i, b in enumerate(items):
        assert got[i].tobytes() == blake3_digest(b), (i, len(b))
    # the same inputs at other alignments inside the blob (a 1- and a 3-byte item in front)
    got = blake3_batch([b"x"] + items[:20] + [b"abc"] + items[20:], ctx=gpu_ctx)
    assert got[0].tobytes() == blake3_digest(b"x")
    for i, b in enumerate(items[:20]):
        assert got[1 + i].tobytes() == blake3_digest(b), (i, len(b))
    for i, b in enumerate(items[20:]):
        assert got[22 + i].tobytes() == blake3_digest(b), (i, len(b))
    # official vectors (BLAKE3 test_vectors.json, input byte i = i % 251): first 8 digest bytes
    known = {0: "af1349b9f5f9a1a6", 1: "2d3adedff11b61f1", 1023: "10108970eeda3eb9", 1024: "42214739f095a406",
             1025: "d00278ae47eb27b3", 2048: "e776b6028c7cd22a", 31744: "62b6960e1a44bcc1"}
    for n, hx in known.items():
        assert blake3_digest(bytes((np.arange(n) % 251).astype(np.uint8))).hex().startswith(hx), n


def test_png_front_end_fills_exact_from_the_device(gpu_ctx, oracle):
    pytest.importorskip("PIL.Image")
    from test_oracle_png import config1_png
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    pngs, imgs = zip(*[config1_png(i, side=64) for i in range(12)])
    rec, st = image.fingerprint_pngs(list(pngs), 64, 64, image.PIX_RGB8, algo=image.MULTI, ctx=gpu_ctx)   # exact=None
    ex = np.stack([np.frombuffer(blake3_digest(p), np.uint8) for p in pngs])
    ref, _ = oracle.image_hash_batch(np.stack(imgs), 7, pixfmt=1, exact=ex)
    assert not st.any() and np.array_equal(rec, ref)
