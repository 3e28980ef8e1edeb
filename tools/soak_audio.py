#!/usr/bin/env python3
"""Randomised differential soak of the audio fingerprints against the CPU oracle: random lengths (one frame to
minutes), signal kinds and Wang configurations; Wang hashes and Haitsma frames must be bit-identical.
    python tools/soak_audio.py --seconds 90 --seed 1"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from ucfp_amd import _lib, audio  # noqa: E402


def signal(rng, n, sr):
    kind = rng.integers(5)
    t = np.arange(n, dtype=np.float64) / sr
    if kind == 0:
        x = 0.2 * rng.standard_normal(n)
    elif kind == 1:
        x = 0.5 * np.sin(2 * np.pi * rng.uniform(50, 3500) * t) + 0.01 * rng.standard_normal(n)
    elif kind == 2:      # sparse clicks: many exact ties and empty rows
        x = np.zeros(n)
        x[:: int(rng.integers(50, 5000))] = rng.uniform(0.1, 0.9)
    elif kind == 3:      # hop-periodic staircase: runs of bit-identical frames
        base = 0.3 * rng.standard_normal(128)
        x = np.tile(base, n // 128 + 1)[:n] * np.repeat(rng.choice([1.0, 0.5, 0.25], n // 2560 + 1), 2560)[:n]
    else:
        x = np.zeros(n)
    return x.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = _lib.default_context(0)
    oracle.build()
    rng = np.random.default_rng(a.seed)
    t0, rounds = time.time(), 0
    while time.time() - t0 < a.seconds:
        rounds += 1
        secs = float(rng.choice([0.128, 0.2, 1.0, 3.3, 7.68, 30.0, 61.5, 200.0]) * rng.uniform(0.9, 1.1))
        n = max(1024, int(secs * 8000))
        x = signal(rng, n, 8000)
        cfg = None
        ocfg = None
        if rng.random() < 0.3:
            kw = dict(fan_out=int(rng.integers(1, 20)), target_zone_t=int(rng.integers(5, 120)),
                      target_zone_f=int(rng.integers(5, 200)), peaks_per_sec=int(rng.integers(5, 60)),
                      min_anchor_mag_db=float(rng.uniform(-80, -10)))
            cfg = audio.WangConfig(**kw)
            ocfg = oracle.WangCfg(kw["fan_out"], kw["target_zone_t"], kw["target_zone_f"], kw["peaks_per_sec"],
                                  kw["min_anchor_mag_db"])
        g = audio.wang_hashes(x, 8000, cfg, ctx=ctx) if cfg else audio.wang_hashes(x, 8000, ctx=ctx)
        o = oracle.wang(x, ocfg, cap=max(64, g.shape[0] + 1000)) if ocfg else oracle.wang(x)
        assert g.shape == o.shape and np.array_equal(g, o), ("wang", n, rounds)
        if rounds % 4 == 1:
            # the ragged-batch entries: a mix of clip lengths (empty, under one frame, seconds, a minute) at a random source
            # rate, Wang (resampler fused into the stream kernel) and Haitsma (batched 5 kHz resample + frame map)
            sr = int(rng.choice([8000, 11025, 16000, 22050, 44100, 48000, 7999, 96000]))
            lens = [int(v * sr) for v in rng.choice([0.0, 0.01, 0.1285, 0.3, 1.0, 2.5, 4.0, 9.7, 60.0], size=int(rng.integers(1, 24)))]
            if sum(lens) > 90 * sr:
                lens = lens[:3]
            clips = [signal(rng, max(n_, 0), sr) if n_ else np.zeros(0, np.float32) for n_ in lens]
            got = audio.wang_hashes_batch(clips, sr, cfg, ctx=ctx) if cfg else audio.wang_hashes_batch(clips, sr, ctx=ctx)
            for c_, g_ in zip(clips, got):
                r8 = c_ if sr == 8000 else oracle.resample_linear(c_, sr, 8000)
                o_ = oracle.wang(r8, ocfg, cap=max(64, g_.shape[0] + 1000)) if ocfg else oracle.wang(r8)
                assert g_.shape == o_.shape and np.array_equal(g_, o_), ("wang batch", sr, c_.size, rounds)
            goth = audio.haitsma_frames_batch(clips, sr, ctx=ctx)
            for c_, g_ in zip(clips, goth):
                o_ = oracle.haitsma(c_, sr)
                assert g_.shape == o_.shape and np.array_equal(g_, o_), ("haitsma batch", sr, c_.size, rounds)
        if rounds % 3 == 0:
            sr = int(rng.choice([5000, 8000, 16000, 44100]))
            y = signal(rng, max(4096, int(min(secs, 30.0) * sr)), sr)
            gh = audio.haitsma_frames(y, sr, ctx=ctx)
            oh = oracle.haitsma(y, sr)
            assert gh.shape == oh.shape and np.array_equal(gh, oh), ("haitsma", y.size, sr, rounds)
    print(f"soak ok: {rounds} random audio inputs in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
