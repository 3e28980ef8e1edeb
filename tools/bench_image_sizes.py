#!/usr/bin/env python3
"""Image multi-hash throughput across frame geometries (fused and generic paths), one GPU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    stream = torch.cuda.current_stream().cuda_stream
    cases = [(512, 512, image.PIX_GRAY8), (512, 512, image.PIX_RGB8), (256, 256, image.PIX_RGB8),
             (1024, 1024, image.PIX_GRAY8), (640, 480, image.PIX_RGB8), (1280, 720, image.PIX_RGB8),
             (1920, 1080, image.PIX_RGB8), (1920, 1080, image.PIX_GRAY8), (300, 200, image.PIX_RGB8),
             (3840, 2160, image.PIX_RGB8), (1000, 1000, image.PIX_RGBA8),
             # round 3: widths that are not multiples of 4 (byte-granular strips; the gather kernel before), small frames
             (301, 200, image.PIX_RGB8), (641, 481, image.PIX_GRAY8), (1023, 767, image.PIX_RGB8), (1366, 768, image.PIX_RGB8),
             (300, 200, image.PIX_GRAY8), (640, 480, image.PIX_GRAY8)]
    bpp = {image.PIX_GRAY8: 1, image.PIX_RGB8: 3, image.PIX_RGBA8: 4}
    if len(sys.argv) >= 4:            # one case: w h bytes-per-pixel [frames]  (profiling runs)
        fmt = {1: image.PIX_GRAY8, 3: image.PIX_RGB8, 4: image.PIX_RGBA8}[int(sys.argv[3])]
        cases = [(int(sys.argv[1]), int(sys.argv[2]), fmt)]
    for w, h, fmt in cases:
        fb = w * h * bpp[fmt]
        n = max(8, min(20000, int(4e9 // fb)))
        if len(sys.argv) >= 5:
            n = int(sys.argv[4])
        frames = torch.randint(0, 256, (n, fb), dtype=torch.uint8, device=dev)
        out = torch.empty((n, 536), dtype=torch.uint8, device=dev)

        def go():
            image.fingerprint_frames_dev(frames.data_ptr(), n, w, h, algo=image.MULTI, pixfmt=fmt,
                                         out_ptr=out.data_ptr(), stream=stream, ctx=ctx)
        go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"w": w, "h": h, "bpp": bpp[fmt], "frames": n, "ms": ms, "frames_per_s": n / ms * 1e3,
                          "GBs": n * (fb + 536) / ms / 1e6}), flush=True)
        del frames, out


if __name__ == "__main__":
    main()
