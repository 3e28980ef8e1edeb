#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer image entry point: frames start in
pageable host memory, records end in host memory. Never the headline `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402


def main():
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(0)
    for n in (1, 64, 2048):
        fr = rng.integers(0, 256, (n, 512, 512), dtype=np.uint8)
        image.fingerprint_frames(fr, ctx=ctx)
        reps = 50 if n == 1 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            image.fingerprint_frames(fr, ctx=ctx)
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"frames_per_call": n, "ms_per_call": dt * 1e3, "frames_per_s": n / dt,
                          "host_GBs": n * 512 * 512 / dt / 1e9}), flush=True)


def batcher():
    """Per-request shape through the micro-batcher: T threads, one frame per call."""
    from concurrent.futures import ThreadPoolExecutor
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(1)
    fr = rng.integers(0, 256, (256, 512, 512), dtype=np.uint8)
    for threads in (16, 64):
        b = image.ImageBatcher(512, 512, max_batch=256, max_delay_us=300, ctx=ctx)
        total = 8192
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda i: b.submit(fr[i % 256]), range(256)))
            t0 = time.perf_counter()
            list(pool.map(lambda i: b.submit(fr[i % 256]), range(total)))
            dt = time.perf_counter() - t0
        nb, ni = b.stats()
        print(json.dumps({"micro_batcher_threads": threads, "frames_per_s": total / dt,
                          "avg_batch": ni / max(nb, 1)}), flush=True)
        b.close()


if __name__ == "__main__":
    main()
    batcher()
