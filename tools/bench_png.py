#!/usr/bin/env python3
"""BASELINE config 1 through the GPU PNG front end: n synthetic 256x256 RGB PNGs (bench.py's generator), encoded bytes
resident in HBM -> records; plus the end-to-end rate including the H2D copy of the encoded bytes from pinned memory.
  python tools/bench_png.py [n] [--level L]"""
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402


def make(n, level, side=256):
    yy, xx = np.mgrid[0:side, 0:side]
    rng = np.random.default_rng(0xC0F1)
    pngs, imgs = [], []
    for i in range(n):
        base = np.stack([(xx + i) & 255, (yy + 2 * i) & 255, (xx + yy) & 255], -1).astype(np.uint8)
        img = base ^ rng.integers(0, 8, (side, side, 3), dtype=np.uint8)
        b = io.BytesIO()
        Image.fromarray(img, "RGB").save(b, "PNG", compress_level=level)
        pngs.append(b.getvalue())
        imgs.append(img)
    return pngs, imgs


def main():
    if "--dump" in sys.argv:              # write 64 config-1 files for tools/bench_batcher.cpp --png=DIR
        d = sys.argv[sys.argv.index("--dump") + 1]
        os.makedirs(d, exist_ok=True)
        for i, p in enumerate(make(64, 1)[0]):
            open(os.path.join(d, f"{i}.png"), "wb").write(p)
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
    level = int(sys.argv[sys.argv.index("--level") + 1]) if "--level" in sys.argv else 1
    ctx = _lib.current_context()
    dev = f"cuda:{ctx.device}"
    pngs, imgs = make(n, level)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum([len(p) for p in pngs], out=offs[1:])
    total = int(offs[-1])
    h_blob = torch.from_numpy(np.frombuffer(b"".join(pngs) + b"\0" * 16, np.uint8).copy()).pin_memory()
    d_blob = h_blob.to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    d_out = torch.zeros((n, 168), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def go():
        image.fingerprint_pngs_dev(d_blob.data_ptr(), d_off.data_ptr(), n, total, 256, 256, image.PIX_RGB8, algo=image.PHASH,
                                   out_ptr=d_out.data_ptr(), status_ptr=d_st.data_ptr(), stream=stream, ctx=ctx)
    go()
    torch.cuda.synchronize()
    ok = not bool(d_st.any().item())
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    t_res = e0.elapsed_time(e1) / reps / 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        d_blob.copy_(h_blob, non_blocking=True)
        go()
    torch.cuda.synchronize()
    t_e2e = (time.perf_counter() - t0) / reps
    # the same with the copy of batch i + 1 under the decode of batch i (two device blobs, a copy stream and a compute stream)
    blobs = [d_blob, torch.empty_like(d_blob)]
    s_copy, s_comp = torch.cuda.Stream(), torch.cuda.Stream()
    done = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2 * reps):
        b = blobs[i & 1]
        with torch.cuda.stream(s_copy):
            if done[i & 1] is not None:
                s_copy.wait_event(done[i & 1])
            b.copy_(h_blob, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s_copy)
        s_comp.wait_event(ev)
        image.fingerprint_pngs_dev(b.data_ptr(), d_off.data_ptr(), n, total, 256, 256, image.PIX_RGB8, algo=image.PHASH,
                                   out_ptr=d_out.data_ptr(), status_ptr=d_st.data_ptr(), stream=s_comp.cuda_stream, ctx=ctx)
        done[i & 1] = torch.cuda.Event()
        done[i & 1].record(s_comp)
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / (2 * reps)
    # two batches in flight: a second context (its own decode workspace) on a second stream, batches alternate -- a batch of
    # 1000 files is one wave per SIMD, so two of them share the chip (encoded bytes resident)
    ctx2 = _lib.Context(ctx.device)
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [(torch.zeros_like(d_out), torch.zeros_like(d_st)) for _ in range(2)]
    cs = [ctx, ctx2]

    def go2(i):
        image.fingerprint_pngs_dev(d_blob.data_ptr(), d_off.data_ptr(), n, total, 256, 256, image.PIX_RGB8, algo=image.PHASH,
                                   out_ptr=outs[i & 1][0].data_ptr(), status_ptr=outs[i & 1][1].data_ptr(),
                                   stream=ss[i & 1].cuda_stream, ctx=cs[i & 1])
    for i in range(2):
        go2(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2 * reps):
        go2(i)
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / (2 * reps)
    same2 = bool(torch.equal(outs[0][0], d_out) and torch.equal(outs[1][0], d_out))
    # CPU: Pillow decode of the same files, one thread
    m = min(n, 200)
    t0 = time.perf_counter()
    for p in pngs[:m]:
        np.asarray(Image.open(io.BytesIO(p)).convert("RGB"))
    t_cpu = (time.perf_counter() - t0) / m
    print(json.dumps({"workload": f"{n} synthetic 256x256 RGB PNGs (compress_level {level}), ?algorithm=phash",
                      "png_bytes_per_image": total / n, "status_all_ok": ok,
                      "gpu_images_per_s_encoded_bytes_resident": n / t_res, "ms_per_batch": t_res * 1e3,
                      "gpu_images_per_s_incl_h2d_of_encoded_bytes": n / t_e2e,
                      "gpu_images_per_s_incl_h2d_copy_of_next_batch_under_decode": n / t_pipe,
                      "gpu_images_per_s_two_batches_in_flight_resident": n / t_two, "two_in_flight_records_equal": same2,
                      "cpu_pillow_decode_images_per_s_1_thread": 1 / t_cpu}), flush=True)


if __name__ == "__main__":
    main()
