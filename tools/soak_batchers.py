#!/usr/bin/env python3
"""Stress of the host micro-batchers: many request threads hammer each batcher with random items for a while; every answer
is compared with the direct batch call's answer for that item (itself checked against the oracle by the test suite).
Small max_batch / byte budgets force every path of the coalescing core: full sets, sets closed on the byte budget,
lone requests, submitters blocked on room.
    python tools/soak_batchers.py --seconds 60 --threads 64"""
import argparse
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from ucfp_amd import _lib, audio, image, text  # noqa: E402

WORDS = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "don't", "U.S.A.", "3.14", "x", "HELLO", "ab12cd"]


def hammer(name, submit, items, refs, seconds, threads):
    stop = time.time() + seconds
    bad, done = [], [0]
    lock = threading.Lock()

    def work(seed):
        rng = random.Random(seed)
        n = 0
        while time.time() < stop and not bad:
            i = rng.randrange(len(items))
            got = submit(items[i])
            if got != refs[i]:
                bad.append((name, i))
                return
            n += 1
            if rng.random() < 0.01:
                time.sleep(rng.random() * 0.002)      # leave the batcher idle now and then: lone requests
        with lock:
            done[0] += n
    ts = [threading.Thread(target=work, args=(s,)) for s in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad[:3]
    return done[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--threads", type=int, default=64)
    a = ap.parse_args()
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(1)
    prng = random.Random(2)
    each = a.seconds / 6
    # text
    docs = [" ".join(prng.choice(WORDS) for _ in range(prng.choice([3, 40, 300, 1500]))) for _ in range(400)]
    docs += ["Café naïve — " + d for d in docs[:40]]
    recs, st = text.minhash_batch(docs)
    refs = [(recs[i].tobytes() if st[i] == 0 else bytes(1032), int(st[i])) for i in range(len(docs))]
    b = text.TextBatcher("minhash", max_batch=48, max_bytes=20_000, max_delay_us=0, ctx=ctx)
    n1 = hammer("text", lambda d: (lambda r: (r[0] if r[1] == 0 else bytes(1032), r[1]))(b.submit(d)), docs, refs, each, a.threads)
    b.close()
    # image
    frames = [rng.integers(0, 256, (128, 128), dtype=np.uint8) for _ in range(200)]
    recs, st = image.fingerprint_frames(np.stack(frames), algo=image.MULTI, ctx=ctx)
    refs = [(recs[i].tobytes(), int(st[i])) for i in range(len(frames))]
    b = image.ImageBatcher(128, 128, max_batch=40, max_delay_us=100, ctx=ctx)
    n2 = hammer("image", b.submit, frames, refs, each, a.threads)
    b.close()
    # audio
    clips = [(0.3 * rng.standard_normal(int(n))).astype(np.float32) for n in rng.integers(0, 40_000, 120)]
    refs = [h.tobytes() for h in audio.wang_hashes_batch(clips, 8000, ctx=ctx)]
    b = audio.WangBatcher(8000, max_batch=16, max_samples=200_000, max_delay_us=0, ctx=ctx)
    n3 = hammer("audio", lambda c: b.submit(c).tobytes(), clips, refs, each, a.threads)
    b.close()
    # png uploads
    import io
    from PIL import Image
    pngs = []
    for i in range(120):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256 >> (i % 5), (48, 48, 3), dtype=np.uint8), "RGB").save(buf, "PNG", compress_level=i % 10)
        pngs.append(buf.getvalue())
    recs, st = image.fingerprint_pngs(pngs, 48, 48, image.PIX_RGB8, algo=image.MULTI, ctx=ctx)
    refs = [(recs[i].tobytes(), int(st[i])) for i in range(len(pngs))]
    b = image.PngBatcher(48, 48, image.PIX_RGB8, max_batch=32, max_bytes=100_000, max_delay_us=0, ctx=ctx)
    n4 = hammer("png", b.submit, pngs, refs, each, a.threads)
    b.close()
    # uploads of any kind and size through ONE batcher (round 4): PNG + JPEG of mixed geometries, host kinds in between
    ups = list(pngs[:30])
    for i in range(90):
        h, w = int(rng.integers(32, 260)), int(rng.integers(32, 330))
        arr = rng.integers(0, 256 >> (i % 4), (h, w, 3), dtype=np.uint8)
        buf = io.BytesIO()
        if i % 3 == 0:
            Image.fromarray(arr, "RGB").save(buf, "PNG", compress_level=i % 10)
        elif i % 3 == 1:
            Image.fromarray(arr, "RGB").save(buf, "JPEG", quality=40 + i % 55, subsampling=i % 3)
        else:
            Image.fromarray(arr, "RGB").save(buf, ["BMP", "GIF", "JPEG"][i % 9 // 3], **({"progressive": True} if i % 9 // 3 == 2 else {}))
        ups.append(buf.getvalue())
    recs, st = image.fingerprint_uploads(ups, ctx=ctx)
    refs = [(recs[i].tobytes(), int(st[i])) for i in range(len(ups))]
    b = image.UploadBatcher(max_batch=24, max_bytes=400_000, max_delay_us=0, ctx=ctx)
    n5 = hammer("upload", b.submit, ups, refs, each, a.threads)
    b.close()
    # the query route (round 4): one query and its own k per request against a resident Hamming shard
    from ucfp_amd import index
    codes = rng.integers(0, 2**63, 600_000, dtype=np.uint64)
    ids = np.arange(600_000, dtype=np.uint64) + np.uint64(7)
    ix = index.DeviceIndex(index.HAMMING64, ctx=ctx)
    ix.upsert(0, ids, codes)
    qs = [(int(codes[int(rng.integers(0, 600_000))]) ^ (1 << int(rng.integers(0, 63))), int(rng.choice([1, 3, 10, 17, 32]))) for _ in range(300)]
    refs = []
    for q, k in qs:
        gi, _, gd, _ = ix.search(0, np.array([q], dtype=np.uint64), k)
        refs.append((gi[0].tobytes(), gd[0].tobytes()))
    sb = index.SearchBatcher(ix, 0, max_batch=40, max_delay_us=0)
    n6 = hammer("search", lambda qk: (lambda r: (np.ascontiguousarray(np.pad(r[0], (0, qk[1] - len(r[0])), constant_values=np.uint64(2**64 - 1))).tobytes(),
                                                  np.ascontiguousarray(np.pad(r[2], (0, qk[1] - len(r[2])), constant_values=np.uint32(2**32 - 1))).tobytes()))(sb.submit(*qk)),
                qs, refs, each, a.threads)
    sb.close()
    ix.close()
    print(f"soak ok: {n1} documents, {n2} frames, {n3} clips, {n4} PNG uploads, {n5} mixed uploads, {n6} searches through the batchers "
          f"from {a.threads} threads")


if __name__ == "__main__":
    main()
