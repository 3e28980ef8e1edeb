#!/usr/bin/env python3
"""Randomised differential soak of the image and text fingerprints against the CPU oracle: random frame
geometries / pixel formats / algorithms, random ASCII documents (word lengths, punctuation, case, k).
    python tools/soak_image_text.py --seconds 90 --seed 1"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from ucfp_amd import _lib, image, text  # noqa: E402


def random_doc(rng):
    nwords = int(rng.choice([0, 1, 3, 4, 5, 6, 40, 300, 900]))
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789'_", np.uint8)
    seps = [b" ", b"  ", b", ", b". ", b"\n", b"\t", b" - ", b"; ", b"! "]
    parts = []
    for _ in range(nwords):
        ln = int(rng.choice([1, 2, 3, 5, 8, 13, 40, 200])) if rng.random() < 0.9 else int(rng.integers(1, 1200))
        parts.append(bytes(alphabet[rng.integers(0, alphabet.size, ln)]))
        parts.append(seps[int(rng.integers(len(seps)))])
    return b"".join(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = _lib.default_context(0)
    oracle.build()
    rng = np.random.default_rng(a.seed)
    t0, rounds, frames_done, docs_done = time.time(), 0, 0, 0
    while time.time() - t0 < a.seconds:
        rounds += 1
        if rounds % 2:
            fmt = int(rng.choice([image.PIX_GRAY8, image.PIX_RGB8, image.PIX_RGBA8]))
            bpp = {image.PIX_GRAY8: 1, image.PIX_RGB8: 3, image.PIX_RGBA8: 4}[fmt]
            if rng.random() < 0.3:
                w = h = int(rng.choice([256, 512, 1024]))
            else:
                w, h = int(rng.integers(32, 1500)), int(rng.integers(32, 1100))
                if rng.random() < 0.6:
                    w &= ~3
            n = int(rng.integers(1, 7))
            shape = (n, h, w) if bpp == 1 else (n, h, w, bpp)
            fr = rng.integers(0, 256, shape, dtype=np.uint8)
            if rng.random() < 0.3:
                fr[:] = (fr // 64) * 64          # flat regions: ties in the medians and comparisons
            algo = int(rng.choice([image.MULTI, image.PHASH, image.DHASH, image.AHASH]))
            g, st = image.fingerprint_frames(fr, algo=algo, pixfmt=fmt, ctx=ctx)
            o, ost = oracle.image_hash_batch(fr, algo, fmt)
            assert np.array_equal(st, ost) and np.array_equal(g, o), ("image", w, h, bpp, algo)
            frames_done += n
        else:
            docs = [random_doc(rng) for _ in range(int(rng.integers(1, 40)))]
            k = int(rng.choice([1, 2, 5, 9]))
            texts = [d.decode("ascii") for d in docs]
            opts = text.TextOpts(k=k)
            g, st = text.minhash_batch(texts, opts, ctx=ctx)
            o, ost = oracle.text_minhash_batch(docs, 0, k)
            sup = st != -2          # -2: a token longer than the LDS batch (documented limit), everything else must agree
            assert np.array_equal(st[sup], ost[sup]), ("text status", st, ost)
            both = (st == 0) & (ost == 0)
            assert np.array_equal(g[both], o[both]), ("minhash", k)
            gs, sst = text.simhash_batch(texts, text.TextOpts(), ctx=ctx)
            os_, osst = oracle.text_simhash_batch(docs, 0)
            both = (sst == 0) & (osst == 0)
            assert np.array_equal(gs[both], os_[both]), "simhash"
            docs_done += len(docs)
    print(f"soak ok: {rounds} rounds, {frames_done} frames, {docs_done} documents in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
