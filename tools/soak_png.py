#!/usr/bin/env python3
"""Randomised differential soak of the GPU PNG front end against Pillow: random geometries (1 .. 400 px a side), colour
types, contents (noise, ramps, flat, periodic, sparse), compression levels and encoder strategies, batches of one
geometry; decoded pixels must equal Pillow's, and the device BLAKE3 the host's.
    python tools/soak_png.py --seconds 120 --seed 1"""
import argparse
import io
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402
from ucfp_amd.blake3 import blake3_batch, blake3_digest  # noqa: E402


def content(rng, h, w, c, kind):
    shape = (h, w) if c == 1 else (h, w, c)
    if kind == 0:
        return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == 1:
        base = np.add.outer(np.arange(h) * int(rng.integers(1, 7)), np.arange(w) * int(rng.integers(1, 7)))
        a = base if c == 1 else base[..., None] + np.arange(c) * 37
        return (a & 255).astype(np.uint8)
    if kind == 2:
        return np.full(shape, int(rng.integers(0, 256)), np.uint8)
    if kind == 3:
        t = rng.integers(0, 256, (int(rng.integers(1, 9)), w) if c == 1 else (int(rng.integers(1, 9)), w, c), dtype=np.uint8)
        return np.tile(t, (h // t.shape[0] + 1,) + (1,) * (t.ndim - 1))[:h]
    a = np.zeros(shape, np.uint8)
    m = rng.random(shape) < 0.02
    a[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
    return a ^ (rng.integers(0, 4, shape, dtype=np.uint8) if rng.random() < 0.5 else np.uint8(0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    t0, rounds, files = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        rounds += 1
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        if rng.random() < 0.1:
            w = int(rng.integers(400, 3000))
        c = int(rng.choice([1, 3, 4]))
        mode, fmt = {1: ("L", 0), 3: ("RGB", 1), 4: ("RGBA", 2)}[c]
        n = int(rng.choice([1, 7, 30, 30, 600, 1100]))       # 600 / 1100 files: the 256-bit and 128-bit round shapes of the inflate kernel
        if n > 100:
            h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        pngs, want = [], []
        for _ in range(n):
            arr = content(rng, h, w, c, int(rng.integers(0, 5)))
            b = io.BytesIO()
            kw = {"compress_level": int(rng.integers(0, 10))}
            if rng.random() < 0.3:
                kw["optimize"] = True
            Image.fromarray(arr, mode).save(b, "PNG", **kw)
            pngs.append(b.getvalue())
            want.append(arr)
        fr, st = image.decode_pngs(pngs, w, h, fmt, ctx=ctx)
        assert not st.any(), ("status", h, w, c, st)
        for i in range(n):
            assert np.array_equal(fr[i], want[i]), ("pixels", h, w, c, i)
        if rounds % 5 == 0:
            dg = blake3_batch(pngs, ctx=ctx)
            for i in range(n):
                assert dg[i].tobytes() == blake3_digest(pngs[i]), ("blake3", len(pngs[i]))
        files += n
    print(f"soak ok: {rounds} rounds, {files} PNG files in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
