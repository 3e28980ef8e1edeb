#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz from the CPU oracle.

The reference cannot run here (no Rust toolchain; imgfprint/audiofp/txtfp un-vendored), so these
vectors are OUR specification's known answers on the reference's own synthetic inputs
(src/server/tests.rs:227-235 synthetic_png, :322-331 synthetic_audio_bytes, the pangram of :1141)
plus seeded random inputs.  They freeze the oracle (CPU test) and the HIP path (GPU test) against
silent drift.  Values the reference itself pins (sizes, the MinHash prefix, the cosine toy test) live
in tests/test_reference_pins.py, not here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def synthetic_png_pixels(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([xx % 256, yy % 256, np.full_like(xx, 128)], axis=-1).astype(np.uint8)


def sine(seconds, sr, f=440.0):
    t = np.arange(int(seconds * sr), dtype=np.float32) / np.float32(sr)
    return (np.sin(np.float32(2.0 * np.pi) * np.float32(f) * t) * np.float32(0.5)).astype(np.float32)


TEXTS = [b"the quick brown fox jumps over the lazy dog",
         b"the quick brown fox jumps over the lazy dog. " * 128,
         b"Hello world, this is a test of the pipeline inspector.",
         b"It's 3.14 o'clock in the U.S.A., isn't it? 1,000,000 x:y foo_bar"]


def main():
    oracle.build()
    g = {}
    for side in (64, 256):
        rec, st = oracle.image_hash_batch(synthetic_png_pixels(side, side)[None], 7, pixfmt=1)
        g[f"image_synthpng{side}_multi"] = rec[0]
    rng = np.random.default_rng(20261004)
    frames = rng.integers(0, 256, (4, 512, 512), dtype=np.uint8)
    g["image_rand512_seed"] = np.array([20261004])
    g["image_rand512_multi"] = oracle.image_hash_batch(frames, 7)[0]
    g["image_synth512_first8_multi"] = oracle.image_hash_batch(oracle.image_synth(8, 512, 512, 0), 7)[0]
    mh, _ = oracle.text_minhash_batch(TEXTS)
    sh, _ = oracle.text_simhash_batch(TEXTS)
    g["text_minhash"] = mh
    g["text_simhash"] = sh
    for secs in (1, 4):
        x = sine(secs, 8000)
        g[f"audio_sine440_{secs}s_wang"] = oracle.wang(x)
        g[f"audio_sine440_{secs}s_haitsma"] = oracle.haitsma(x, 8000)
    rng = np.random.default_rng(7)
    t = np.arange(6 * 8000) / 8000.0
    y = (0.3 * np.sin(2 * np.pi * (300 + 200 * t) * t) + 0.2 * np.sin(2 * np.pi * 1500 * t * (1 + 0.1 * t))
         + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    g["audio_chirp_pcm"] = y
    g["audio_chirp_wang"] = oracle.wang(y)
    g["audio_chirp_haitsma"] = oracle.haitsma(y, 8000)
    codes = rng.integers(0, 2**64, 4096, dtype=np.uint64)
    ids = rng.permutation(4096).astype(np.uint64)
    q = codes[:6] ^ np.uint64(0x8421)
    hi, hd, hc = oracle.hamming_topk(ids, codes, q, 10)
    g.update(hamming_codes=codes, hamming_ids=ids, hamming_q=q, hamming_top_ids=hi, hamming_top_d=hd)
    rows = rng.standard_normal((512, 48)).astype(np.float32)
    cq = rng.standard_normal(48).astype(np.float32)
    ci, cs = oracle.cosine_knn(np.arange(512, dtype=np.uint64), rows, cq, 10)
    g.update(cosine_rows=rows, cosine_q=cq, cosine_top_ids=ci, cosine_top_scores=cs)
    out = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
