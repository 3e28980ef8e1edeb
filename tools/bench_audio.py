"""Audio leg on its own: Wang over 44.1 kHz PCM (BASELINE config 3) -- fused (the stream kernel resamples) vs two-pass
(ucfp_audio_resample_linear_dev + ucfp_audio_wang_dev) -- the 8 kHz stream alone, and a batch of 4-second clips
(benches/end_to_end.rs:55-75) in clips/s.  One JSON line per case on stdout."""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucfp_amd import _lib  # noqa: E402


def synth(n, sr, dev, seed=0xA0D10):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.arange(n, dtype=torch.float32, device=dev) / sr
    x = torch.zeros(n, dtype=torch.float32, device=dev)
    for i in range(8):
        f0 = 110.0 * (1.6 ** i)
        x += 0.06 * torch.sin(2 * np.pi * (f0 * t + 3.0 * torch.sin(0.05 * (i + 1) * t)))
    del t
    x += 0.0158 * torch.randn(n, dtype=torch.float32, device=dev, generator=g)
    return x.clamp_(-0.5, 0.5)


def timeit(fn, steps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=36000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--clips", type=int, default=8192)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = _lib.default_context(0)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    sr, secs = 44100, a.seconds
    n = sr * secs
    x = synth(n, sr, dev)
    one = torch.tensor([0, n], dtype=torch.int64, device=dev)
    cap = int(lib.ucfp_audio_wang_batch_max_hashes(n, 1, sr, None))
    out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    oo = torch.zeros(2, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def fused():
        _lib.check(lib.ucfp_audio_wang_batch_dev(ctx.handle, x.data_ptr(), one.data_ptr(), n, 1, sr, None, out.data_ptr(), cap,
                                                 oo.data_ptr(), st))
    if a.only in ("", "fused"):
        ms = timeit(fused, a.steps)
        nh = int(oo[1].item())
        print(json.dumps({"case": "wang 44.1k fused resample", "seconds": secs, "ms": ms, "x_real_time": secs / ms * 1e3,
                          "hashes": nh, "algorithmic_GBs": (n * 4 + nh * 8) / ms / 1e6}), flush=True)
    m = int(lib.ucfp_audio_resample_len(n, sr, 8000))
    x8 = torch.empty(m, dtype=torch.float32, device=dev)

    def two_pass():
        _lib.check(lib.ucfp_audio_resample_linear_dev(ctx.handle, x.data_ptr(), n, sr, 8000, x8.data_ptr(), m, st))
        _lib.check(lib.ucfp_audio_wang_dev(ctx.handle, x8.data_ptr(), m, 8000, None, out.data_ptr(), cap, cnt.data_ptr(), st))
    if a.only in ("", "two"):
        ms2 = timeit(two_pass, a.steps)
        nh2 = int(cnt.item())
        print(json.dumps({"case": "wang 44.1k resample pass + 8k stream", "seconds": secs, "ms": ms2, "hashes": nh2}), flush=True)

    def only8k():
        _lib.check(lib.ucfp_audio_wang_dev(ctx.handle, x8.data_ptr(), m, 8000, None, out.data_ptr(), cap, cnt.data_ptr(), st))
    if a.only in ("", "8k"):
        two_pass()
        ms3 = timeit(only8k, a.steps)
        print(json.dumps({"case": "wang 8k stream alone", "seconds": secs, "ms": ms3}), flush=True)
    del x8
    if a.only in ("", "clips"):
        # batch of 4 s clips at 8 kHz (the reference's bench clip) cut from the stream
        clip_n = 4 * 8000
        nc = a.clips
        xc = synth(nc * clip_n, 8000, dev, seed=7)
        offs = (torch.arange(nc + 1, dtype=torch.int64, device=dev) * clip_n).contiguous()
        capc = int(lib.ucfp_audio_wang_batch_max_hashes(nc * clip_n, nc, 8000, None))
        outc = torch.empty((capc, 2), dtype=torch.int32, device=dev)
        ooc = torch.zeros(nc + 1, dtype=torch.int64, device=dev)

        def clips():
            _lib.check(lib.ucfp_audio_wang_batch_dev(ctx.handle, xc.data_ptr(), offs.data_ptr(), nc * clip_n, nc, 8000, None,
                                                     outc.data_ptr(), capc, ooc.data_ptr(), st))
        msc = timeit(clips, a.steps)
        print(json.dumps({"case": "wang batch of 4 s clips @ 8 kHz", "clips": nc, "ms": msc, "clips_per_s": nc / msc * 1e3,
                          "hashes": int(ooc[-1].item())}), flush=True)

        def single():
            _lib.check(lib.ucfp_audio_wang_dev(ctx.handle, xc.data_ptr(), clip_n, 8000, None, outc.data_ptr(), capc,
                                               cnt.data_ptr(), st))
        mss = timeit(single, 50)
        print(json.dumps({"case": "wang ONE 4 s clip per call", "ms": mss, "clips_per_s": 1e3 / mss}), flush=True)


if __name__ == "__main__":
    main()
