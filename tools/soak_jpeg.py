#!/usr/bin/env python3
"""Randomised differential soak of the GPU JPEG front end against libjpeg (Pillow draft "L") and the oracle: random
geometries (1 .. 700 px a side, a few up to 3000 wide), contents (noise, ramps, flat, sparse, periodic), qualities 1 .. 100,
subsamplings, optimised tables, restart intervals, greyscale files, batches of one geometry; luma planes must equal
libjpeg's; damaged copies of the files must get the oracle's status (and its pixels where both decode).
    python tools/soak_jpeg.py --seconds 120 --seed 1"""
import argparse
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from PIL import Image, ImageFile  # noqa: E402
import oracle  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402

ImageFile.MAXBLOCK = 1 << 24


def content(rng, h, w, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:
        base = np.add.outer(np.arange(h) * int(rng.integers(1, 7)), np.arange(w) * int(rng.integers(1, 7)))
        return ((base[..., None] + np.arange(3) * 37) & 255).astype(np.uint8) ^ rng.integers(0, 1 << int(rng.integers(0, 6)), (h, w, 3), dtype=np.uint8)
    if kind == 2:
        return np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
    if kind == 3:
        t = rng.integers(0, 256, (int(rng.integers(1, 17)), w, 3), dtype=np.uint8)
        return np.tile(t, (h // t.shape[0] + 1, 1, 1))[:h]
    a = np.zeros((h, w, 3), np.uint8)
    m = rng.random((h, w, 3)) < 0.02
    a[m] = 255
    return a


def luma(j):
    im = Image.open(io.BytesIO(j))
    im.draft("L", im.size)
    return np.asarray(im)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    oracle.build()
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    t0, rounds, files, damaged, both = time.time(), 0, 0, 0, 0
    while time.time() - t0 < a.seconds:
        rounds += 1
        h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        if rng.random() < 0.08:
            w = int(rng.integers(700, 3000))
        n = int(rng.choice([1, 5, 24, 24, 300]))
        if n > 100:
            h, w = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        jpgs = []
        for _ in range(n):
            img = content(rng, h, w, int(rng.integers(0, 5)))
            kw = {"quality": int(rng.choice([1, 10, 30, 50, 75, 85, 95, 100]))}
            grey = rng.random() < 0.15
            if not grey:
                kw["subsampling"] = int(rng.integers(0, 3))
            r = rng.random()
            if r < 0.2:
                kw["optimize"] = True
            elif r < 0.4:
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            elif r < 0.5:
                kw["restart_marker_blocks"] = int(rng.integers(1, 40))
            b = io.BytesIO()
            src = Image.fromarray(img, "RGB")
            (src.convert("L") if grey else src).save(b, "JPEG", **kw)
            jpgs.append(b.getvalue())
        fr, st = image.decode_jpegs(jpgs, w, h, ctx=ctx)
        assert not st.any(), (rounds, h, w, st)
        for i, j in enumerate(jpgs):
            assert np.array_equal(fr[i], luma(j)), (rounds, h, w, i)
        files += n
        # damaged copies: the device and the oracle must agree file by file
        bad = []
        for j in jpgs[: min(n, 40)]:
            b = bytearray(j)
            sos = bytes(b).index(b"\xff\xda")
            kind = int(rng.integers(0, 4))
            if kind == 0 and len(b) > sos + 20:
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(sos + 14, len(b) - 2))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b = b[: int(rng.integers(min(sos + 14, len(b) - 1), len(b)))]
            elif kind == 2 and len(b) > sos + 40:
                cut = int(rng.integers(sos + 14, len(b) - 20))
                b[cut:cut + 12] = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
            else:
                b[int(rng.integers(2, sos))] ^= 1 << int(rng.integers(0, 8))
            bad.append(bytes(b))
        fr, st = image.decode_jpegs(bad, w, h, ctx=ctx)
        for i, j in enumerate(bad):
            rc, px = oracle.jpeg_decode_luma(j)
            if rc == 0 and oracle.jpeg_probe(j)[1:] != (w, h):
                rc = 1
            assert st[i] == rc, (rounds, i, st[i], rc)
            if rc == 0:
                assert np.array_equal(fr[i], px), (rounds, i)
                both += 1
        damaged += len(bad)
    print(f"soak_jpeg ok: {rounds} rounds, {files} files equal to libjpeg's luma, {damaged} damaged files with the oracle's status "
          f"({both} of them decoded by both, same pixels), seed {a.seed}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
