#!/usr/bin/env python3
"""JPEG front end alone: N config-1 images as JPEG (quality / subsampling / restart interval selectable) through
ucfp_image_jpeg_hash_batch_dev, images/s with the encoded bytes resident.  One JSON line.
usage: python3 tools/bench_jpeg.py [--n 1000] [--quality 85] [--sub 2] [--restart-rows 0] [--side 256]"""
import argparse
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--sub", type=int, default=2)
    ap.add_argument("--restart-rows", type=int, default=0)
    ap.add_argument("--side", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import numpy as np
    import torch
    from PIL import Image
    from ucfp_amd import _lib, image
    side = a.side
    yy, xx = np.mgrid[0:side, 0:side]
    rng = np.random.default_rng(0xC0F1)
    uniq = min(a.n, 256)
    jpgs = []
    for i in range(uniq):
        base = np.stack([(xx + i) & 255, (yy + 2 * i) & 255, (xx + yy) & 255], -1).astype(np.uint8)
        img = base ^ rng.integers(0, 8, (side, side, 3), dtype=np.uint8)
        b = io.BytesIO()
        kw = {"restart_marker_rows": a.restart_rows} if a.restart_rows else {}
        Image.fromarray(img, "RGB").save(b, "JPEG", quality=a.quality, subsampling=a.sub, **kw)
        jpgs.append(b.getvalue())
    jpgs = [jpgs[i % uniq] for i in range(a.n)]
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    offs = np.zeros(a.n + 1, np.int64)
    np.cumsum([len(j) for j in jpgs], out=offs[1:])
    jb = int(offs[-1])
    d_blob = torch.from_numpy(np.frombuffer(b"".join(jpgs) + b"\0" * 16, np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    d_out = torch.zeros((a.n, 168), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(a.n, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def go():
        image.fingerprint_jpegs_dev(d_blob.data_ptr(), d_off.data_ptr(), a.n, jb, side, side, algo=image.PHASH,
                                    out_ptr=d_out.data_ptr(), status_ptr=d_st.data_ptr(), stream=st, ctx=ctx)
    go()
    torch.cuda.synchronize()
    assert not d_st.any().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(json.dumps({"n": a.n, "side": side, "quality": a.quality, "subsampling": a.sub, "restart_rows": a.restart_rows,
                      "bytes_per_file": jb / a.n, "ms_per_batch": ms, "images_per_s": a.n / ms * 1e3}), flush=True)


if __name__ == "__main__":
    main()
