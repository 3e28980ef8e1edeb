#!/usr/bin/env python3
"""Randomised differential soak of round 4's any-size paths: (a) ragged batches of decoded frames
(ucfp_image_hash_ragged[_dev]) against the oracle's records, geometry by geometry; (b) mixed batches of encoded uploads
(ucfp_image_upload_hash_batch_dev, probe on the host or on the device) against the oracle's records of Pillow's / libjpeg's
pixels, with the kinds the device hands back, empty and damaged files in between.
    python tools/soak_uploads.py --seconds 120 --seed 1"""
import argparse
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402
import oracle  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402
from ucfp_amd.blake3 import blake3_digest  # noqa: E402
from test_oracle_jpeg import jpeg_of, libjpeg_luma, picture  # noqa: E402


def png_of(arr, mode=None, **kw):
    b = io.BytesIO()
    Image.fromarray(arr, mode).save(b, "PNG", **kw)
    return b.getvalue()


def fmt_of(fr):
    return 0 if fr.ndim == 2 else (1 if fr.shape[2] == 3 else 2)


def random_frame(rng):
    """A decoded frame of a random geometry and pixel format; small sides are common, a few rows are long."""
    r = rng.random()
    if r < 0.6:
        h, w = int(rng.integers(32, 300)), int(rng.integers(32, 300))
    elif r < 0.9:
        h, w = int(rng.integers(32, 700)), int(rng.integers(32, 1100))
    else:
        h, w = int(rng.integers(32, 120)), int(rng.integers(1100, 2049))
    ch = int(rng.choice([1, 3, 4]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    elif kind == 1:
        a = ((np.add.outer(np.arange(h) * int(rng.integers(1, 9)), np.arange(w) * int(rng.integers(1, 9)))[..., None]
              + np.arange(ch) * 41) & 255).astype(np.uint8)
    elif kind == 2:
        a = np.full((h, w, ch), int(rng.integers(0, 256)), np.uint8)
    else:
        a = np.zeros((h, w, ch), np.uint8)
        a[rng.random((h, w, ch)) < 0.03] = 255
    return a[..., 0] if ch == 1 else a


def soak_ragged(rng, ctx, stats):
    n = int(rng.integers(1, 40))
    frames = [random_frame(rng) for _ in range(n)]
    # guards in between: a frame below min_dimension gets its own status
    if rng.random() < 0.3:
        frames[int(rng.integers(0, n))] = np.zeros((int(rng.integers(1, 31)), 64), np.uint8)
    algo = int(rng.choice([1, 2, 4, 7]))
    ex = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    rec, st = image.fingerprint_frames_ragged(frames, [fmt_of(f) for f in frames], algo=algo, exact=ex, ctx=ctx)
    for i, fr in enumerate(frames):
        ref, rst = oracle.image_hash_batch(fr[None], algo, pixfmt=fmt_of(fr), exact=ex[i][None])
        assert st[i] == rst[0], ("ragged status", i, fr.shape, st[i], rst[0])
        assert np.array_equal(rec[i], ref[0]), ("ragged record", i, fr.shape, algo)
    stats["ragged_frames"] += n


def random_upload(rng):
    """-> (file bytes, expected frame or None, expected status: 0, 1 (needs host), -1 (rejected), None (any non-zero))"""
    h, w = int(rng.integers(32, 360)), int(rng.integers(32, 480))
    if rng.random() < 0.08:
        w, h = int(rng.integers(500, 1400)), int(rng.integers(32, 160))
    img = picture(h, w, seed=int(rng.integers(1 << 30)))
    k = int(rng.integers(0, 16))
    if k == 0:
        g = np.asarray(Image.fromarray(img, "RGB").convert("L"))
        return png_of(g, "L", compress_level=int(rng.integers(0, 10))), g, 0
    if k == 1:
        return png_of(img, "RGB", compress_level=int(rng.integers(0, 10))), img, 0
    if k == 2:
        a = np.dstack([img, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)])
        return png_of(a, "RGBA"), a, 0
    if k == 3:
        p = Image.fromarray(img, "RGB").quantize(int(rng.choice([17, 64, 256])))
        b = io.BytesIO()
        p.save(b, "PNG", **({"transparency": 1} if rng.random() < 0.5 else {}))
        f = b.getvalue()
        if f[24] != 8:
            return f, None, 1                                   # Pillow packed it below 8 bits: the host's
        return f, np.asarray(p.convert("RGB")), 0
    if k == 4:
        g = np.asarray(Image.fromarray(img, "RGB").convert("L"))
        return png_of(np.dstack([g, rng.integers(0, 256, (h, w), dtype=np.uint8)]), "LA"), g, 0
    if k in (5, 6, 7, 8):
        extra = [{}, {"optimize": True}, {"restart_marker_rows": int(rng.integers(1, 4))}, {}][k - 5]
        f = jpeg_of(img, quality=int(rng.integers(5, 99)), subsampling=int(rng.integers(0, 3)), **extra)
        return f, libjpeg_luma(f), 0
    if k == 9:
        f = jpeg_of(np.asarray(Image.fromarray(img, "RGB").convert("L")), "L", quality=int(rng.integers(20, 96)))
        return f, libjpeg_luma(f), 0
    if k == 10:
        return jpeg_of(img, progressive=True), None, 1
    if k == 11:
        b = io.BytesIO()
        Image.fromarray(img, "RGB").save(b, str(rng.choice(["BMP", "GIF"])))
        return b.getvalue(), None, 1
    if k == 12:
        return png_of((img[..., 0].astype(np.uint16) * 257)), None, 1          # 16-bit
    if k == 13:
        return b"", None, -1                                                  # empty upload
    if k == 14:
        f = png_of(img, "RGB") if rng.random() < 0.5 else jpeg_of(img, quality=80)
        cut = int(rng.integers(8, max(9, int(0.9 * len(f)))))
        return f[:cut], None, None                                            # truncated: any non-zero status
    return png_of(picture(int(rng.integers(1, 31)), 200)), None, -1           # below min_dimension


def soak_uploads(rng, ctx, stats):
    n = int(rng.integers(1, 48))
    items = [random_upload(rng) for _ in range(n)]
    files = [f for f, _, _ in items]
    rec, st = image.fingerprint_uploads(files, probe_on_device=bool(rng.random() < 0.5), ctx=ctx)
    for i, (f, fr, want) in enumerate(items):
        if want == 0:
            ex = np.frombuffer(blake3_digest(f), np.uint8)
            ref, rst = oracle.image_hash_batch(fr[None], 7, pixfmt=fmt_of(fr), exact=ex[None])
            assert st[i] == 0 and rst[0] == 0, ("upload status", i, fr.shape, st[i])
            assert np.array_equal(rec[i], ref[0]), ("upload record", i, fr.shape, f[:4])
            stats["uploads_decoded"] += 1
        else:
            assert (st[i] != 0) if want is None else (st[i] == want), ("upload hand-back", i, want, st[i], f[:8])
            assert not rec[i].any()
            stats["uploads_handed_back"] += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    oracle.build()
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    stats = {"ragged_frames": 0, "uploads_decoded": 0, "uploads_handed_back": 0}
    t0, rounds = time.time(), 0
    while time.time() - t0 < a.seconds:
        soak_ragged(rng, ctx, stats)
        soak_uploads(rng, ctx, stats)
        rounds += 1
        if rounds % 10 == 0:
            print(f"[{time.time() - t0:6.0f} s] rounds {rounds} {stats}", flush=True)
    print(f"soak ok: {rounds} rounds in {time.time() - t0:.0f} s: {stats}")


if __name__ == "__main__":
    main()
