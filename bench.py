#!/usr/bin/env python3
"""bench.py -- headline measurement of the UCFP hot path on MI355X.

Workload (BASELINE.json configs[1]): batched image multi-hash (pHash + dHash + aHash, global +
16 block hashes each = one 536-byte imgfprint bundle per frame) over 100 000 synthetic 512x512
GRAY8 frames resident in HBM.  One "step" = one pass of the hot path over the whole batch =
ONE kernel launch reading 26.2 GB.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

N > 1 is launched by torch.distributed.run, one rank per GPU.  The path shards by frame with
no data-path collective, so every rank hashes its own F frames (weak scaling) and
only the timing is reduced (MAX over ranks).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_SIDE = 512
REC_BYTES = 536
ALGO_BYTES_PER_FRAME = FRAME_SIDE * FRAME_SIDE + REC_BYTES  # 262 680 B: SURVEY 8(d), DESIGN.md
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=100_000, help="frames per GPU (BASELINE: 100k)")
    ap.add_argument("--cpu-sample", type=int, default=8192,
                    help="frames of the same workload timed through the CPU oracle (0 = skip)")
    ap.add_argument("--ann-corpus", type=int, default=100_000_000,
                    help="TOTAL 64-bit fingerprints in the sharded Hamming corpus (BASELINE configs[4]: "
                         "100M; split evenly over the ranks = strong scaling). 0 skips the ANN leg")
    ap.add_argument("--ann-queries", type=int, default=4096)
    ap.add_argument("--ann-steps", type=int, default=20)
    ap.add_argument("--no-forced-rccl", action="store_true",
                    help="N = 1 only: skip the extra ANN measurement through a one-rank RCCL communicator")
    ap.add_argument("--rgb-frames", type=int, default=30_000, help="RGB8 512x512 frames for the RGB variant; 0 skips")
    ap.add_argument("--cosine-rows", type=int, default=1_000_000, help="768-d f32 rows per GPU for the cosine leg; 0 skips")
    ap.add_argument("--text-docs", type=int, default=1_000_000, help="4 KiB docs per GPU (BASELINE configs[3]); 0 skips")
    ap.add_argument("--audio-seconds", type=int, default=36_000, help="seconds of 44.1 kHz audio per GPU "
                    "(BASELINE configs[2]: 10 h); 0 skips")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/)")
    return ap.parse_args()


def traffic_from_profiles(frames):
    """HBM bytes per launch measured by rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes,
    gfx950 correction applied) and committed under profiles/; only valid for the profiled batch."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for rd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, rd, "image_multi_pmc_summary.json")
        if os.path.exists(f):
            d = json.load(open(f))["corrected_bytes_per_launch"]
            if d.get("algorithmic") == ALGO_BYTES_PER_FRAME * frames:
                best = d["total"]
    return best


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share() -> dict:
    """CPUs this PROCESS may actually use: the scheduler affinity and the cgroup CPU quota (a one-GPU lease of the pool
    gets a share of the host, not the host: 128 OpenMP threads on a 16-CPU quota measured 9 % "parallel efficiency" in
    round 2 -- that was the quota, not the code)."""
    import math
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())   # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            pass
    usable = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    return {"affinity_cpus": aff, "cgroup_cpu_quota": quota, "usable_cpus": usable}


def host_cpu() -> dict:
    """Physical cores and hardware threads of this host (report both), the CPU model, and what share of the
    host this process is allowed to use."""
    phys = logical = None
    try:
        import psutil
        phys, logical = psutil.cpu_count(logical=False), psutil.cpu_count(logical=True)
    except Exception:  # noqa: BLE001
        logical = os.cpu_count()
    return {"cpu_model": cpu_model(), "cores": phys, "hw_threads": logical, **cpu_share()}


_ORACLE_THREADS = None


def timed_oracle():
    """The CPU restatement for the TIMED legs: the same sources built `-O3 -march=native -ffp-contract=off` on this
    machine (oracle.use_native); results are bit-identical to the portable build the tests use.  OpenMP runs one thread
    per CPU this process may use (cpu_share), not per core of the host."""
    global _ORACLE_THREADS
    import oracle
    oracle.use_native()
    if _ORACLE_THREADS is None:
        _ORACLE_THREADS = max(1, min(oracle.num_threads(), cpu_share()["usable_cpus"]))
    oracle.set_threads(_ORACLE_THREADS)
    return oracle


def best_of(fn, repeats=3):
    """(best wall seconds, all wall seconds, last result) of `repeats` calls: CPU baselines are quoted on their best run."""
    times, res = [], None
    for _ in range(repeats):
        t0 = time.perf_counter()
        res = fn()
        times.append(time.perf_counter() - t0)
    return min(times), times, res


def bench_config1_phash_png(dev, ctx, n_img=1000):
    """BASELINE configs[0]: ?algorithm=phash on 1 k 256x256 PNGs -- the reference's own per-item plumbing
    (decode -> hash -> 168-B record), CPU side = Pillow decode + the C restatement, split decode vs hash;
    GPU side = the same decoded frames hashed by the HIP path (device-resident, and through the
    host-pointer ABI incl. PCIe).  N = 1, rank 0 only; a bounded ~10 s of CPU work."""
    import io
    import numpy as np
    import torch
    from PIL import Image
    oracle = timed_oracle()
    from ucfp_amd import _lib, image
    side = 256
    yy, xx = np.mgrid[0:side, 0:side]
    rng = np.random.default_rng(0xC0F1)
    pngs = []
    for i in range(n_img):   # colour ramp of benches/end_to_end.rs:77-85 xor per-image noise
        base = np.stack([(xx + i) & 255, (yy + 2 * i) & 255, (xx + yy) & 255], -1).astype(np.uint8)
        img = base ^ rng.integers(0, 8, (side, side, 3), dtype=np.uint8)
        b = io.BytesIO()
        Image.fromarray(img, "RGB").save(b, "PNG", compress_level=1)
        pngs.append(b.getvalue())
    def decode(p):
        return np.asarray(Image.open(io.BytesIO(p)).convert("RGB"))
    t0 = time.perf_counter()
    frames = np.stack([decode(p) for p in pngs])
    t_dec = time.perf_counter() - t0
    cores = oracle.num_threads()
    # ... and on every CPU this process may use (1 and N threads): Pillow's decoders release the GIL,
    # so a thread pool scales; worker PROCESSES are avoided on purpose (this process holds the GPU)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=cores) as pool:
        list(pool.map(decode, pngs[:cores * 2]))                        # threads started, code paths warm
        t_decn, _, frames_n = best_of(lambda: list(pool.map(decode, pngs, chunksize=max(1, n_img // (cores * 8)))))
    assert all(np.array_equal(a, b) for a, b in zip(frames_n[:8], frames[:8]))
    oracle.set_threads(1)
    t0 = time.perf_counter()
    ref, _ = oracle.image_hash_batch(frames, 2, pixfmt=1)
    t_h1 = time.perf_counter() - t0
    oracle.set_threads(cores)
    t_hn, _, _ = best_of(lambda: oracle.image_hash_batch(frames, 2, pixfmt=1))
    # GPU: host-pointer ABI (pageable host memory in, records out), then device-resident frames
    t0 = time.perf_counter()
    got, st = image.fingerprint_frames(frames, algo=image.PHASH, pixfmt=image.PIX_RGB8, ctx=ctx)
    t_gh = time.perf_counter() - t0
    d_frames = torch.from_numpy(frames).to(dev)
    d_out = torch.empty((n_img, 168), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def go():
        image.fingerprint_frames_dev(d_frames.data_ptr(), n_img, side, side, algo=image.PHASH, pixfmt=image.PIX_RGB8,
                                     out_ptr=d_out.data_ptr(), stream=stream, ctx=ctx)
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        go()
    e1.record()
    torch.cuda.synchronize()
    t_gd = e0.elapsed_time(e1) / 20 / 1e3
    # GPU PNG front end: the ENCODED files are the input -- chunk walk, inflate, unfilter and hash on the
    # device; first with the encoded bytes already in HBM, then including their H2D copy from pinned host memory
    offs = np.zeros(n_img + 1, np.int64)
    np.cumsum([len(p) for p in pngs], out=offs[1:])
    png_bytes = int(offs[-1])
    h_blob = torch.from_numpy(np.frombuffer(b"".join(pngs) + b"\0" * 16, np.uint8).copy()).pin_memory()
    d_blob, d_off = h_blob.to(dev), torch.from_numpy(offs).to(dev)
    d_out2 = torch.zeros((n_img, 168), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n_img, dtype=torch.int32, device=dev)

    def go_png():
        image.fingerprint_pngs_dev(d_blob.data_ptr(), d_off.data_ptr(), n_img, png_bytes, side, side, image.PIX_RGB8,
                                   algo=image.PHASH, out_ptr=d_out2.data_ptr(), status_ptr=d_st.data_ptr(), stream=stream,
                                   ctx=ctx)
    go_png()
    torch.cuda.synchronize()
    # no `exact` is handed in, so the front end hashes the files itself (BLAKE3 on the device): the expected records are
    # the oracle's with the host's BLAKE3 of each file in their first 32 bytes
    from ucfp_amd.blake3 import blake3_digest
    ref_png = ref.copy()
    ref_png[:, :32] = np.stack([np.frombuffer(blake3_digest(p), np.uint8) for p in pngs])
    png_ok = bool(not d_st.any().item() and np.array_equal(d_out2.cpu().numpy(), ref_png))
    e0.record()
    for _ in range(10):
        go_png()
    e1.record()
    torch.cuda.synchronize()
    t_png = e0.elapsed_time(e1) / 10 / 1e3
    t0 = time.perf_counter()
    for _ in range(10):
        d_blob.copy_(h_blob, non_blocking=True)
        go_png()
    torch.cuda.synchronize()
    t_png_h2d = (time.perf_counter() - t0) / 10
    # ... and with the copy of batch i + 1 under the decode of batch i (two device blobs, a copy and a compute stream)
    blobs = [d_blob, torch.empty_like(d_blob)]
    s_copy, s_comp = torch.cuda.Stream(), torch.cuda.Stream()
    done = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        b = blobs[i & 1]
        with torch.cuda.stream(s_copy):
            if done[i & 1] is not None:
                s_copy.wait_event(done[i & 1])
            b.copy_(h_blob, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s_copy)
        s_comp.wait_event(ev)
        image.fingerprint_pngs_dev(b.data_ptr(), d_off.data_ptr(), n_img, png_bytes, side, side, image.PIX_RGB8,
                                   algo=image.PHASH, out_ptr=d_out2.data_ptr(), status_ptr=d_st.data_ptr(),
                                   stream=s_comp.cuda_stream, ctx=ctx)
        done[i & 1] = torch.cuda.Event()
        done[i & 1].record(s_comp)
    torch.cuda.synchronize()
    t_png_pipe = (time.perf_counter() - t0) / 20
    # ... and two batches in flight (bytes resident): a second context = a second decode workspace, on a second stream --
    # a batch of 1000 files is one wave per SIMD, two of them share the chip
    ctx2 = _lib.Context(ctx.device)
    two_s = [torch.cuda.Stream(), torch.cuda.Stream()]
    two_o = [(torch.zeros_like(d_out2), torch.zeros_like(d_st)) for _ in range(2)]

    def go_two(i):
        image.fingerprint_pngs_dev(d_blob.data_ptr(), d_off.data_ptr(), n_img, png_bytes, side, side, image.PIX_RGB8,
                                   algo=image.PHASH, out_ptr=two_o[i & 1][0].data_ptr(),
                                   status_ptr=two_o[i & 1][1].data_ptr(), stream=two_s[i & 1].cuda_stream,
                                   ctx=(ctx, ctx2)[i & 1])
    go_two(0), go_two(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        go_two(i)
    torch.cuda.synchronize()
    t_png_two = (time.perf_counter() - t0) / 20
    png_two_ok = bool(np.array_equal(two_o[0][0].cpu().numpy(), ref_png) and np.array_equal(two_o[1][0].cpu().numpy(), ref_png))
    ctx2.close()
    jpeg = bench_config1_jpeg(dev, ctx, frames, oracle, cores)
    return {
        "jpeg_front_end": jpeg,
        "workload": f"{n_img} synthetic 256x256 RGB PNGs, ?algorithm=phash (168-B records)",
        "cpu": {"kind": "port", **host_cpu(), "threads": cores,
                "png_decode_s": t_dec, "decode_images_per_s_1_thread": n_img / t_dec,
                "decode_images_per_s_n_threads": n_img / t_decn,
                "hash_images_per_s_1_thread": n_img / t_h1, "hash_images_per_s_all_cores": n_img / t_hn,
                "decode_plus_hash_images_per_s_1_thread": n_img / (t_dec + t_h1),
                "decode_plus_hash_images_per_s_n_threads": n_img / (t_decn + t_hn),
                "decode_parallel_efficiency": (n_img / t_decn) / (cores * n_img / t_dec)},
        "gpu": {"hash_images_per_s_device_resident": n_img / t_gd,
                "hash_images_per_s_host_pointer_abi_incl_pcie": n_img / t_gh,
                "matches_oracle": bool(np.array_equal(got, ref) and not st.any()
                                       and np.array_equal(d_out.cpu().numpy(), ref)),
                "png_front_end": {"images_per_s_encoded_bytes_resident": n_img / t_png,
                                  "images_per_s_incl_h2d_of_encoded_bytes": n_img / t_png_h2d,
                                  "images_per_s_incl_h2d_next_batch_copied_under_decode": n_img / t_png_pipe,
                                  "images_per_s_two_batches_in_flight_resident": n_img / t_png_two,
                                  "two_in_flight_records_match_oracle": png_two_ok,
                                  "png_bytes_per_image": png_bytes / n_img, "records_match_oracle": png_ok}},
        "gpu_hash_over_cpu_hash_all_cores": (n_img / t_gd) / (n_img / t_hn),
        "gpu_png_front_end_over_cpu_decode_plus_hash_1_thread": (n_img / t_png_h2d) / (n_img / (t_dec + t_h1)),
        "gpu_png_front_end_over_cpu_decode_plus_hash_n_threads": (n_img / t_png_h2d) / (n_img / (t_decn + t_hn)),
        "note": "cpu: Pillow decode + the C restatement's hash, one thread (the reference path is decode-bound here). "
                "gpu.png_front_end: encoded files in, records out, decode on the device (ucfp_image_png_hash_batch_dev; "
                "SURVEY 8f N4), including the BLAKE3 of every file for the records' exact field",
    }


def bench_config1_jpeg(dev, ctx, frames, oracle, cores):
    """The config-1 images as JPEG uploads (quality 85, 4:2:0 -- what cameras and browsers send): CPU = libjpeg (Pillow) decoding
    the luma plane + the C restatement's hash, 1 and N threads; GPU = the encoded files through the JPEG front end (decode,
    BLAKE3 of every file, hash).  Records are checked against the oracle's records of libjpeg's luma planes."""
    import io
    import numpy as np
    import torch
    from PIL import Image
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    n_img, side = frames.shape[0], frames.shape[1]
    jpgs = []
    for i in range(n_img):
        b = io.BytesIO()
        Image.fromarray(frames[i], "RGB").save(b, "JPEG", quality=85, subsampling=2)
        jpgs.append(b.getvalue())

    def decode(j):
        im = Image.open(io.BytesIO(j))
        im.draft("L", im.size)
        return np.asarray(im)
    t0 = time.perf_counter()
    planes = np.stack([decode(j) for j in jpgs])
    t_dec = time.perf_counter() - t0
    with ThreadPoolExecutor(max_workers=cores) as pool:
        list(pool.map(decode, jpgs[:cores * 2]))
        t_decn, _, _ = best_of(lambda: list(pool.map(decode, jpgs, chunksize=max(1, n_img // (cores * 8)))))
    oracle.set_threads(1)
    t0 = time.perf_counter()
    ref, _ = oracle.image_hash_batch(planes, 2, pixfmt=0)
    t_h1 = time.perf_counter() - t0
    oracle.set_threads(cores)
    t_hn, _, _ = best_of(lambda: oracle.image_hash_batch(planes, 2, pixfmt=0))
    ref[:, :32] = np.stack([np.frombuffer(blake3_digest(j), np.uint8) for j in jpgs])
    offs = np.zeros(n_img + 1, np.int64)
    np.cumsum([len(j) for j in jpgs], out=offs[1:])
    jb = int(offs[-1])
    h_blob = torch.from_numpy(np.frombuffer(b"".join(jpgs) + b"\0" * 16, np.uint8).copy()).pin_memory()
    d_blob, d_off = h_blob.to(dev), torch.from_numpy(offs).to(dev)
    d_out = torch.zeros((n_img, 168), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n_img, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def go():
        image.fingerprint_jpegs_dev(d_blob.data_ptr(), d_off.data_ptr(), n_img, jb, side, side, algo=image.PHASH,
                                    out_ptr=d_out.data_ptr(), status_ptr=d_st.data_ptr(), stream=stream, ctx=ctx)
    go()
    torch.cuda.synchronize()
    ok = bool(not d_st.any().item() and np.array_equal(d_out.cpu().numpy(), ref))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    t_g = e0.elapsed_time(e1) / 10 / 1e3
    t0 = time.perf_counter()
    for _ in range(10):
        d_blob.copy_(h_blob, non_blocking=True)
        go()
    torch.cuda.synchronize()
    t_gh = (time.perf_counter() - t0) / 10
    return {"workload": f"the {n_img} config-1 images as JPEG (quality 85, 4:2:0; {jb / n_img:.0f} B per file), ?algorithm=phash",
            "cpu": {"decoder": "libjpeg-turbo through Pillow, luma plane only (draft L)", "threads": cores,
                    "decode_images_per_s_1_thread": n_img / t_dec, "decode_images_per_s_n_threads": n_img / t_decn,
                    "decode_plus_hash_images_per_s_1_thread": n_img / (t_dec + t_h1),
                    "decode_plus_hash_images_per_s_n_threads": n_img / (t_decn + t_hn)},
            "gpu": {"images_per_s_encoded_bytes_resident": n_img / t_g, "images_per_s_incl_h2d_of_encoded_bytes": n_img / t_gh,
                    "records_match_oracle_records_of_libjpeg_luma": ok},
            "gpu_over_cpu_decode_plus_hash_1_thread": (n_img / t_gh) / (n_img / (t_dec + t_h1)),
            "gpu_over_cpu_decode_plus_hash_n_threads": (n_img / t_gh) / (n_img / (t_decn + t_hn))}


def upload_mix_files(n_img=400, seed=0xA11, lo=64, hi=2048):
    """A stated mix of uploads: sides log-uniform in [lo, hi] px (width and height drawn independently),
    half PNG (RGB, compress level 1-6) and half baseline JPEG (quality 70-95, 4:2:0 / 4:4:4), content = smooth gradients + a
    little noise; on top, every 16th upload is of a kind the device hands back (progressive JPEG, 16-bit PNG, BMP in turn).
    -> (files, decoded pixel bytes of the PNG / JPEG ones as the device produces them)."""
    import io
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    files, px_bytes = [], 0
    for i in range(n_img):
        w = int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))
        h = int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))
        yy, xx = np.mgrid[0:h, 0:w]
        f = rng.uniform(1, 9, 3)
        img = np.stack([128 + 90 * np.sin(xx / w * f[0] + i), 128 + 90 * np.cos(yy / h * f[1]), (xx * 255 // w + yy * 255 // h) // 2 +
                        40 * np.sin((xx + yy) / (w + h) * f[2] * 3)], -1)
        img = np.clip(img + rng.normal(0, 3, img.shape), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        if i % 16 == 15:
            kind = (i // 16) % 3
            if kind == 0:
                Image.fromarray(img, "RGB").save(b, "JPEG", quality=85, progressive=True)
            elif kind == 1:
                Image.fromarray((img[..., 0].astype(np.uint16) << 8)).save(b, "PNG")
            else:
                Image.fromarray(img, "RGB").save(b, "BMP")
        elif i % 2 == 0:
            Image.fromarray(img, "RGB").save(b, "PNG", compress_level=int(rng.integers(1, 7)))
            px_bytes += w * h * 3
        else:
            Image.fromarray(img, "RGB").save(b, "JPEG", quality=int(rng.integers(70, 96)), subsampling=int(rng.integers(0, 2)) * 2)
            px_bytes += w * h
        files.append(b.getvalue())
    return files, px_bytes


def bench_upload_mix(dev, ctx, n_img=300, threads=64):
    """Uploads of any size and kind through the any-upload entry (ucfp_image_upload_hash_batch_dev) with the files resident in
    HBM, and through ONE micro-batcher (ucfp_upload_batcher_*) fed by request threads -- the reference's route
    (src/server/handlers.rs:232-302) with the decode on the device.  Records are checked against the per-request host path
    on a sample."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import _lib, image
    files, px_bytes = upload_mix_files(n_img)
    n = len(files)
    enc_bytes = sum(len(f) for f in files)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum([len(f) for f in files], out=offs[1:])
    d_blob = torch.from_numpy(np.frombuffer(b"".join(files) + bytes(64), np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    d_out = torch.zeros((n, 536), dtype=torch.uint8, device=dev)
    d_st = torch.zeros((n,), dtype=torch.int32, device=dev)
    info = image._probe_all(files)
    stream = torch.cuda.current_stream().cuda_stream

    def go():
        image.fingerprint_uploads_dev(d_blob.data_ptr(), d_off.data_ptr(), n, int(offs[-1]), info, out_ptr=d_out.data_ptr(),
                                      status_ptr=d_st.data_ptr(), stream=stream, ctx=ctx)
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    st = d_st.cpu().numpy()
    rec = d_out.cpu().numpy()
    needs_host = int((st == 1).sum())
    # the per-request host path (Pillow decode + the same hash kernels) must give the same records
    same = True
    for i in [k for k in range(n) if st[k] == 0][:12]:
        same = same and bytes(image.fingerprint(files[i], 0, i).fingerprint) == rec[i].tobytes()
    # one batcher, many request threads
    bt = image.UploadBatcher(max_batch=512, max_bytes=128 << 20, max_delay_us=200, ctx=ctx)

    def timed(f):
        t = time.perf_counter()
        r = bt.submit(f)
        return r, time.perf_counter() - t
    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(bt.submit, files[:threads]))
        t0 = time.perf_counter()
        res_t = list(pool.map(timed, files * 2))
        dt = time.perf_counter() - t0
    res = [r for r, _ in res_t]
    batches, items = bt.stats()
    bt.close()
    same_b = all(r[0] == rec[i % n].tobytes() and r[1] == st[i % n] for i, r in enumerate(res))
    # request latency by kind (device-decoded uploads only): PNG and JPEG coalesce in lanes of their own inside the batcher
    lat = {"png": [], "jpeg": []}
    for i, (r, t) in enumerate(res_t):
        if r[1] == 0:
            lat["png" if files[i % n][:4] == b"\x89PNG" else "jpeg"].append(t)
    lat_ms = {k: {"median": float(np.median(v)) * 1e3, "p90": float(np.percentile(v, 90)) * 1e3} for k, v in lat.items() if v}
    return {"what": f"{n} uploads, sides log-uniform 64-2048 px, 47 % PNG (RGB) / 47 % baseline JPEG / 6 % kinds the device hands "
                    "back (progressive JPEG, 16-bit PNG, BMP); records = 536-B bundles",
            "files": n, "encoded_MB": enc_bytes / 1e6, "decoded_MB": px_bytes / 1e6,
            "resident": {"ms": ms, "images_per_s": n / ms * 1e3, "encoded_GBs": enc_bytes / ms / 1e6,
                         "decoded_pixel_GBs": px_bytes / ms / 1e6},
            "needs_host": needs_host, "needs_host_share": needs_host / n, "rejected": int((st < 0).sum()),
            "batcher": {"request_threads": threads, "images_per_s": 2 * n / dt, "batches": batches, "items": items,
                        "request_latency_ms": lat_ms,
                        "note": "Python request threads (ctypes releases the GIL inside submit); host memory in and out"},
            "records_equal_per_request_host_path": bool(same), "batcher_records_equal_resident_call": bool(same_b)}


def bench_search_batcher():
    """/v1/query's request shape -- ONE query per request thread (src/server/handlers.rs:143-187), 64 / 256 threads -- through the
    search micro-batcher over a 12.5 M-code Hamming shard, from NATIVE threads (tools/bench_search_batcher.cpp, built by
    __graft_entry__.build(); a Python driver would measure the interpreter lock).  A child process with its own context."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "bench_search_batcher.bin")
    if not os.path.exists(exe):
        return {"error": "tools/bench_search_batcher.bin is not built (__graft_entry__.build() makes it)"}
    try:
        r = subprocess.run([exe, "--n=12500000", "--k=10", "64", "256"], capture_output=True, text=True, timeout=240)
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        return {"what": "12.5 M codes, k = 10, one query per request thread, host memory in and out; every 170th request's answer "
                        "checked against a brute-force scan on the host", "runs": rows, "rc": r.returncode}
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_baseline(sample: int, gpu_records_head):
    """Time the CPU oracle (OpenMP over frames) on the first frames of the same synthetic workload -- at least 64 frames
    per thread, best of 3 -- and use the occasion to check the GPU records of those frames."""
    import numpy as np
    oracle = timed_oracle()
    cores = oracle.num_threads()
    sample = max(sample, 64 * cores)
    if gpu_records_head is not None:
        sample = min(sample, gpu_records_head.shape[0])
    frames = oracle.image_synth(sample, FRAME_SIDE, FRAME_SIDE, 0)
    oracle.image_hash_batch(frames[:cores * 2], 7)        # thread team up, code paths warm
    dt, dts, (recs, _) = best_of(lambda: oracle.image_hash_batch(frames, 7))
    parity = None
    if gpu_records_head is not None:
        parity = bool(np.array_equal(recs, gpu_records_head[:sample]))
    # the same restatement on ONE thread (SURVEY 8d asks for both), on a smaller slice
    ns = max(1, min(sample, 256))
    oracle.set_threads(1)
    dt1, _, _ = best_of(lambda: oracle.image_hash_batch(frames[:ns], 7))
    oracle.set_threads(cores)
    return {
        "value": sample / dt, "unit": "fingerprints/s", **host_cpu(), "threads": cores, "kind": "port",
        "single_thread_value": ns / dt1, "parallel_efficiency": (sample / dt) / (cores * ns / dt1),
        "build": "gcc -O3 -march=native -ffp-contract=off -fopenmp",
        "sample": f"first {sample} frames of the same synthetic batch ({sample // cores} per thread; best of 3: "
                  f"{', '.join('%.2f' % x for x in dts)} s wall); C restatement oracle/ucfp_oracle_image.c, "
                  "not the reference Rust binary (no Rust toolchain, SDK crates un-vendored)",
        "gpu_matches_oracle_on_sample": parity,
    }


def bench_ann(args, rank, world, dev, ctx, corpus_total=None):
    """Secondary leg (BASELINE configs[4]): /v1/query Hamming k=10 over a corpus sharded across
    the ranks, one all-gather of per-shard top-k, merge on every rank. Returns a dict (rank 0)."""
    import torch
    import torch.distributed as dist
    from ucfp_amd import index, sharded

    k, nq = 10, args.ann_queries
    corpus_total = corpus_total or args.ann_corpus
    start, end = sharded.shard_range(corpus_total, rank, world)
    n_local = end - start

    def agree(err, where):
        """All ranks leave the leg TOGETHER when any of them failed: a rank that raised alone would leave the others
        blocked in the next collective (ncclCommInitRank, ncclAllGather, the barrier).  Every rank calls this at the
        same points with its own error (or None); one MIN all-reduce over the job's process group decides."""
        if world > 1:
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32,
                              device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and err is None:
                raise RuntimeError(f"another rank failed in the ANN leg ({where}); rank {rank} leaves with it")
        if err is not None:
            raise err

    try:
        setup_err = None
        codes, ids, queries = _ann_inputs(torch, dev, rank, world, n_local, start, end, nq)
        _ann_plant(torch, codes, queries, start, end, corpus_total)
    except Exception as e:   # noqa: BLE001
        setup_err = e
    agree(setup_err, "corpus generation")
    six = sharded.ShardedIndex(index.HAMMING64, ctx=ctx)      # collective at N > 1 (ncclCommInitRank)
    try:
        six.append_local(ids, codes)
        torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001
        setup_err = e
    agree(setup_err, "shard upload")
    cpu_sample = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:      # keep a bounded corpus sample for the CPU leg below
        m = min(n_local, 8_000_000)
        cpu_sample = (ids[:m].cpu().numpy().view("uint64"), codes[:m].cpu().numpy().view("uint64"))
    del codes, ids
    return _bench_ann_run(args, rank, world, dev, ctx, six, queries, cpu_sample, corpus_total, n_local, agree)


def ann_corpus_codes(torch, dev, start, end):
    """code of GLOBAL row i = one xorshift64* step from the state (0x5EED, i) -- a function of the
    global index alone, so the union of the shards is the same corpus at every world size.  int64 arithmetic wraps like
    uint64; logical right shifts are arithmetic shifts with the sign extension masked off."""
    out = torch.empty((end - start,), dtype=torch.int64, device=dev)
    step = 1 << 24
    for lo in range(start, end, step):
        hi = min(end, lo + step)
        x = torch.arange(lo + 1, hi + 1, dtype=torch.int64, device=dev) * -7046029254386353131     # 0x9E3779B97F4A7C15
        x ^= 0x5EED
        x ^= (x >> 12) & ((1 << 52) - 1)
        x ^= x << 25
        x ^= (x >> 27) & ((1 << 37) - 1)
        out[lo - start:hi - start] = x * 2685821657736338717                                         # 0x2545F4914F6CDD1D
    return out


def ann_planted(nq, corpus_total):
    """(query j, global row) of the planted neighbours: every second query, rows spread over the whole corpus by a
    multiplicative hash of j -- global positions, so the same rows at every world size."""
    return [(j, (j * 2654435761 + 12345) % corpus_total) for j in range(0, nq, 2)]


def _ann_inputs(torch, dev, rank, world, n_local, start, end, nq):
    codes = ann_corpus_codes(torch, dev, start, end)
    ids = torch.arange(start, end, dtype=torch.int64, device=dev)
    gq = torch.Generator(device=dev)
    gq.manual_seed(0xC0FFEE)
    queries = torch.randint(-2**63, 2**63 - 1, (nq,), dtype=torch.int64, device=dev, generator=gq)
    # plant a true neighbour (two bit flips) for every second query, at GLOBAL rows: the owner of the row writes it
    return codes, ids, queries


def _ann_plant(torch, codes, queries, start, end, corpus_total):
    pl = [(j, pos) for j, pos in ann_planted(queries.numel(), corpus_total) if start <= pos < end]
    if pl:
        jj = torch.tensor([j for j, _ in pl], dtype=torch.int64, device=codes.device)
        pp = torch.tensor([pos - start for _, pos in pl], dtype=torch.int64, device=codes.device)
        flips = torch.tensor([(1 << (j % 61)) ^ (1 << ((j * 3) % 59)) for j, _ in pl], dtype=torch.int64, device=codes.device)
        codes[pp] = queries[jj] ^ flips


def _bench_ann_run(args, rank, world, dev, ctx, six, queries, cpu_sample, corpus_total, n_local, agree):
    import torch
    import torch.distributed as dist
    from ucfp_amd import index, sharded
    k, nq = 10, args.ann_queries

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_pipeline():
        """ann_steps batches, pipelined two deep: the library runs the shard scans of the two batches in flight on two
        streams (one batch's staging kernels fill the gaps of the other's matrix-core scan) and the exchange of a
        batch on a third.  Returns (seconds, MAX over ranks; the last batch's outputs)."""
        # A rank whose own submit fails has still joined that batch's all-gather inside the library (shard.hip), so it
        # keeps submitting -- the collectives stay matched -- and the error surfaces at the agreement point below.
        err, res = None, None

        def submit():
            nonlocal err
            try:
                return six.submit(queries, k)
            except Exception as e:   # noqa: BLE001
                err = err or e
                return None

        def collect(t):
            nonlocal err, res
            if t is not None:
                try:
                    res = six.collect(t)
                except Exception as e:   # noqa: BLE001
                    err = err or e

        for _ in range(2):
            collect(submit())
        barrier()
        t0 = time.perf_counter()
        ticket = submit()
        for _ in range(args.ann_steps - 1):
            nxt = submit()
            collect(ticket)
            ticket = nxt
        collect(ticket)
        barrier()
        el = time.perf_counter() - t0
        agree(err, "pipelined search")
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res

    dt, out = timed_pipeline()
    out_ids, _, out_keys, out_cnt = out
    planted_found = int((out_keys[0::2, 0] <= 3).sum().item())
    # every rank holds the full answer after the merge: a CRC of the last batch's ids || keys, equal on every rank and --
    # the corpus being a function of the global row -- equal at every world size for the same --ann-corpus
    import zlib
    answers_crc = zlib.crc32(out_keys.cpu().numpy().tobytes(), zlib.crc32(out_ids.cpu().numpy().tobytes())) & 0xffffffff
    print(f"[bench ann] rank {rank}/{world} corpus {corpus_total} shard {n_local} answers_crc {answers_crc:08x} "
          f"planted {planted_found}/{(nq + 1) // 2} {dt / args.ann_steps * 1e3:.3f} ms/batch", file=sys.stderr, flush=True)
    if world > 1:
        crcs = torch.tensor([answers_crc], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        lo, hi = crcs.clone(), crcs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        crc_same = bool(int(lo.item()) == int(hi.item()))
    else:
        crc_same = True
    # the exchange step alone ("all-gather time separately"): the same batches searched one at a time
    # (submit + collect back to back, so the all-gather + merge is NOT hidden under the next scan) minus the bare
    # shard scan through ucfp_index_search_dev
    exch_ms = None
    reps = max(4, args.ann_steps // 2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        six.search(queries, k)          # submit + collect back to back: one batch at a time
    barrier()
    t_seq = (time.perf_counter() - t0) / reps
    if world > 1:
        b = six._buffers(nq, k, dev, 0)
        cur = torch.cuda.current_stream().cuda_stream
        t0 = time.perf_counter()
        for _ in range(reps):
            six.local.search_dev(0, queries.data_ptr(), nq, k, b["out_ids"].data_ptr(), 0, b["out_keys"].data_ptr(),
                                 b["out_cnt"].data_ptr(), cur)
        barrier()
        t_loc = (time.perf_counter() - t0) / reps
        exch_ms = max(0.0, (t_seq - t_loc) * 1e3)
    rccl_ranks = world if six.rccl else 0
    exchanges = six.comm.exchanges()
    # the request shape of the reference's /v1/query (ONE query per request, src/server/handlers.rs:143-159): this rank's
    # shard through the single-launch search (hamming_direct.hip), 1 and 8 queries; a pure HBM stream of the codes
    single = {}
    b0 = six._buffers(8, k, dev, 0)
    cur = torch.cuda.current_stream().cuda_stream
    for nq1 in (1, 8):
        def one():
            six.local.search_dev(0, queries.data_ptr(), nq1, k, b0["out_ids"].data_ptr(), b0["out_scores"].data_ptr(),
                                 b0["out_keys"].data_ptr(), b0["out_cnt"].data_ptr(), cur)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps1 = 50
        e0.record()
        for _ in range(reps1):
            one()
        e1.record()
        torch.cuda.synchronize()
        ms1 = e0.elapsed_time(e1) / reps1
        single[f"batch{nq1}"] = {"ms": ms1, "qps_per_gpu": nq1 / ms1 * 1e3, "codes": n_local,
                                 "roofline": {"bound": "hbm", "kernel": "hamming_direct_kernel",
                                              "achieved": n_local * 8 / (ms1 / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": n_local * 8 / (ms1 / 1e3) / 1e9 / HBM_PEAK_GBS,
                                              "algorithmic_bytes": "8 B per code per pass (SURVEY 8d); launches back to "
                                                                   "back from Python: host-launch-bound below ~35 us"}}
    # One GPU: the same batches once more through a ONE-RANK RCCL communicator (UCFP_SHARD_FORCE_RCCL) -- the code a
    # multi-GPU job runs (ncclCommInitRank, one ncclAllGather per batch on the exchange stream, merge over the gathered
    # buffer) executed and timed on the hardware there is; results must equal the local short cut's.
    forced = None
    if world == 1 and not args.no_forced_rccl:
        try:
            base = [t.clone() for t in out]
            plain = six.comm
            six.comm = sharded.ShardComm(ctx, None, force_rccl=True)
            f_dt, f_out = timed_pipeline()
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                six.search(queries, k)
            barrier()
            f_seq = (time.perf_counter() - t0) / reps
            same = all(bool(torch.equal(a, b)) for a, b in zip(base, f_out))
            forced = {"rccl_ranks": 1, "uses_rccl": six.comm.uses_rccl, "rccl_all_gathers": six.comm.exchanges(),
                      "ms_per_batch": f_dt / args.ann_steps * 1e3, "ms_per_batch_one_at_a_time": f_seq * 1e3,
                      "exchange_ms_per_batch": max(0.0, (f_seq - t_seq) * 1e3),
                      "value": nq * args.ann_steps / f_dt, "unit": "queries/s",
                      "equals_local_path": same}
            six.comm.close()
            six.comm = plain
        except Exception as e:   # RCCL missing on the host: say so, the leg is extra
            forced = {"error": f"{type(e).__name__}: {e}"}
    cpu = None
    if cpu_sample is not None:
        # CPU leg (the reference has NO Hamming search, F3: this is the C statement of the same scan,
        # popcount(q ^ x) + k-best, OpenMP over corpus slices x queries), on a bounded sample of the same corpus;
        # a brute-force scan is linear in the corpus, so queries/s at the full corpus = pairs/s / corpus
        import numpy as np
        oracle = timed_oracle()
        c_ids, c_codes = cpu_sample
        qh = queries.cpu().numpy().view("uint64")
        oracle.hamming_topk_omp(c_ids[:1 << 20], c_codes[:1 << 20], qh[:64], k)      # thread team up, pages touched
        dt_b, dts_b, (b_ids, b_d, _) = best_of(lambda: oracle.hamming_topk_omp(c_ids, c_codes, qh, k))
        reps1 = 8
        dts_1 = []
        for j in range(reps1):
            t0 = time.perf_counter()
            oracle.hamming_topk_omp(c_ids, c_codes, qh[j:j + 1], k)
            dts_1.append(time.perf_counter() - t0)
        dt_1 = sorted(dts_1)[len(dts_1) // 2]          # median: a single-query scan is a few ms, one outlier is 100x
        # the GPU on the same sample must give the same lists
        chk = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, ctx)
        d_i = torch.from_numpy(c_ids.view("int64")).to(dev)
        d_c = torch.from_numpy(c_codes.view("int64")).to(dev)
        chk.append_dev(0, d_i.data_ptr(), d_c.data_ptr(), c_ids.size, torch.cuda.current_stream().cuda_stream)
        g_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        g_d = torch.empty((nq, k), dtype=torch.int32, device=dev)
        g_c = torch.empty((nq,), dtype=torch.int32, device=dev)
        chk.search_dev(0, queries.data_ptr(), nq, k, g_ids.data_ptr(), 0, g_d.data_ptr(), g_c.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        same = bool(np.array_equal(g_ids.cpu().numpy().view("uint64"), b_ids) and
                    np.array_equal(g_d.cpu().numpy().view("uint32"), b_d))
        chk.close()
        del d_i, d_c
        cpu = {"kind": "port", **host_cpu(), "threads": oracle.num_threads(),
               "build": "gcc -O3 -march=native -fopenmp (oracle/ucfp_oracle_index.c ucfp_oracle_hamming_topk_omp2)",
               "inner_loop": "AVX-512 VPOPCNTDQ, 32 codes per trip, scalar insert for passers" if oracle.hamming_simd()
                             else "scalar popcount (this CPU has no AVX-512 VPOPCNTDQ)",
               "sample": f"first {c_ids.size} codes of the corpus, all {nq} queries (best of 3: "
                         f"{', '.join('%.2f' % x for x in dts_b)} s) and 1 query x {reps1} (median)",
               "pairs_per_s_batch": c_ids.size * nq / dt_b, "pairs_per_s_single_query": c_ids.size / dt_1,
               "value": c_ids.size * nq / dt_b / corpus_total, "unit": "queries/s at the full corpus (batch of %d)" % nq,
               "single_query_value": c_ids.size / dt_1 / corpus_total,
               "gpu_matches_oracle_on_sample": same}
    if rank != 0:
        return None
    qps = nq * args.ann_steps / dt
    pairs_per_s = qps * corpus_total
    # The filter is a +-1 x 0/1 contraction with FP4 (e2m1) operands on v_mfma_f32_32x32x64_f8f6f4: 64 MACs =
    # 128 ops per code-query pair, exact in f32.  Dense FP4 peak = 4 x the bf16 rate = 10 PFLOP/s
    # (MI355X_MICROARCH.md, matrix cores); the int8 form of rounds 1-2 (5 POP/s peak) is kept as the second
    # yardstick.  Two code tiles share an accumulator (the second MFMA block-scaled by 2^16) and v_pk_maximum3_f16
    # folds four sums per instruction, so the matrix pipe bounds the loop: tools/ubench_mfma_i8.hip mode 23 measures
    # 63-64 T pairs/s (8.1 PFLOP/s) for that bare stream on random operands (the chip holds ~1.95 GHz under toggling
    # data, not 2.4), 72 T without any fold.  PMC on the 100 M search's largest dispatch
    # (profiles/r02/hamming_scan_pmc.txt): matrix pipes busy 80 % of the kernel's cycles at 1.86 GHz.
    fp4_peak = 10.0e15
    i8_peak = 5.0e15
    ops_per_pair = 128.0
    return {
        "metric": "ANN queries/sec (Hamming k=10, brute force, exact)", "value": qps, "unit": "queries/s",
        "corpus_total": corpus_total, "corpus_per_gpu": n_local, "queries_per_batch": nq, "k": k,
        "ms_per_batch": dt / args.ann_steps * 1e3, "batches_in_flight": 2,
        "ms_per_batch_one_at_a_time": t_seq * 1e3, "scaling": "strong",
        "exchange": "ONE ncclAllGather (RCCL, called by libucfp_hip.so itself) of nq*k*16 B per rank + merge on every "
                    "rank, on the library's side stream under the next batch's shard scan (exchange_ms_per_batch is "
                    "the step alone, unoverlapped)" if world > 1 else "none",
        "rccl_ranks": rccl_ranks, "rccl_all_gathers": exchanges, "rccl_forced": forced, "single_query": single,
        "pairs_per_s": pairs_per_s, "exchange_ms_per_batch": exch_ms, "cpu_baseline": cpu,
        "roofline": {"bound": "mfma", "kernel": "hamming_scan_mfma",
                     "achieved": pairs_per_s * ops_per_pair / world / 1e12, "peak": fp4_peak / 1e12,
                     "unit": "TFLOP/s per GPU (FP4 MFMA, f32 accumulate, 128 ops per code-query pair; whole search "
                             "incl. staging, rescan and selection)",
                     "frac": pairs_per_s * ops_per_pair / world / fp4_peak,
                     # the same with ONE search in flight (ms_per_batch_one_at_a_time): what a lone batch sees
                     "frac_one_at_a_time": nq * n_local * ops_per_pair / t_seq / fp4_peak,
                     "frac_of_int8_peak": pairs_per_s * ops_per_pair / world / i8_peak,
                     "matrix_pipe_busy_pmc": 0.80, "clock_GHz_pmc": 1.86,
                     "T_pairs_per_s_per_gpu": pairs_per_s / world / 1e12,
                     "hbm_GBs_per_gpu": ((nq + 4095) // 4096) * n_local * 8 / (dt / args.ann_steps) / 1e9},
        "planted_neighbours_found": f"{planted_found}/{(nq + 1) // 2}",
        # CRC-32 of the last batch's out_ids || out_keys; the corpus is a function of the GLOBAL row (ann_corpus_codes), so
        # this value is the same at N = 1, 2, 4, 8 for the same --ann-corpus / --ann-queries: an N > 1 run verifies itself
        # against the N = 1 line
        "answers_crc": f"{answers_crc:08x}", "answers_crc_equal_on_all_ranks": crc_same,
        "corpus": "code(i) = xorshift64*((i + 1) * 0x9E3779B97F4A7C15 ^ 0x5EED), i = global row; shard g = rows "
                  "[g n / G, (g + 1) n / G); a neighbour two bit flips away planted for every second query at a global row",
    }


def synth_docs_dev(n_docs, doc_len, dev, seed):
    """SURVEY 8(d) config 4, built on the device: `n_docs` ASCII documents of exactly `doc_len` bytes, words drawn from a
    50 k-word vocabulary (word i = 1 + i % 7 base-26 letters) with Zipf(1.1) frequencies, single spaces; every tenth
    document (index 10 m + 1) is a NEAR-DUPLICATE of its predecessor: the same words with 5 % of them redrawn.
    Returns the [n_docs, doc_len] uint8 blob."""
    import torch
    V, W = 50000, 1200
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    i = torch.arange(V, device=dev)
    wlen = (1 + i % 7).to(torch.int64)
    letters = (97 + (i[:, None] // (26 ** torch.arange(7, device=dev))[None, :]) % 26).to(torch.uint8)      # [V, 7]
    w = 1.0 / torch.arange(1, V + 1, device=dev, dtype=torch.float64) ** 1.1
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    blob = torch.full((n_docs, doc_len), 32, dtype=torch.uint8, device=dev)
    C = 20000                                   # documents per chunk (a multiple of 10: pairs never straddle chunks)
    for c0 in range(0, n_docs, C):
        c1 = min(n_docs, c0 + C)
        m = c1 - c0
        ranks = torch.searchsorted(cdf, torch.rand((m, W), device=dev, generator=g, dtype=torch.float64)).clamp_(max=V - 1)
        dup = torch.arange(c0, c1, device=dev) % 10 == 1
        dup[0] = False
        src = torch.nonzero(dup).squeeze(1)
        edited = ranks[src - 1].clone()
        redraw = torch.rand((src.numel(), W), device=dev, generator=g) < 0.05
        edited[redraw] = ranks[src][redraw]
        ranks[src] = edited
        ln = wlen[ranks]                                         # [m, W]
        start = torch.cumsum(ln + 1, 1) - (ln + 1)               # byte offset of every word (one space after each)
        row = torch.arange(m, device=dev)[:, None].expand(m, W) + c0
        for bpos in range(7):
            pos = start + bpos
            ok = (ln > bpos) & (pos < doc_len)
            blob[row[ok], pos[ok]] = letters[ranks[ok], bpos]
        del ranks, ln, start, row
    return blob


def bench_text(args, rank, world, dev, ctx):
    """Secondary leg (BASELINE configs[3]): MinHash-128 over the synthetic 4 KiB ASCII documents of SURVEY 8(d)
    (distinct documents, 10 % near-duplicates), tokenisation + shingling + hashing all on the GPU, then the banded LSH
    index built FROM those signatures and queried with the near-duplicates.  Documents shard by index, no collective."""
    import numpy as np
    import torch
    from ucfp_amd import _lib, text
    n_docs, doc_len = args.text_docs, 4096
    blob = synth_docs_dev(n_docs, doc_len, dev, 0xD0C5 + rank)
    offs = (torch.arange(n_docs + 1, dtype=torch.int64, device=dev) * doc_len).contiguous()
    out = torch.empty((n_docs, 1032), dtype=torch.uint8, device=dev)
    status = torch.empty((n_docs,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    lib = _lib.load()

    def step():
        _lib.check(lib.ucfp_text_minhash_batch_dev(ctx.handle, blob.data_ptr(), offs.data_ptr(), n_docs, 0, 5,
                                                   out.data_ptr(), status.data_ptr(), stream))
    step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    steps = 3
    ev[0].record()
    for _ in range(steps):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    assert int(status.abs().sum().item()) == 0
    # SimHash-64 over the same documents (reference row a6)
    sout = torch.empty((n_docs, 8), dtype=torch.uint8, device=dev)
    _lib.check(lib.ucfp_text_simhash_batch_dev(ctx.handle, blob.data_ptr(), offs.data_ptr(), n_docs, 0, sout.data_ptr(),
                                               status.data_ptr(), stream))
    torch.cuda.synchronize()
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev2[0].record()
    for _ in range(steps):
        _lib.check(lib.ucfp_text_simhash_batch_dev(ctx.handle, blob.data_ptr(), offs.data_ptr(), n_docs, 0,
                                                   sout.data_ptr(), status.data_ptr(), stream))
    ev2[1].record()
    torch.cuda.synchronize()
    sms = ev2[0].elapsed_time(ev2[1]) / steps
    # Roofline of the MinHash kernel: it is bound by integer VALU issue, not bytes (HBM sees ~2 %).  Numerator = VALU
    # wave-instructions per document from the PMC pass tracked under profiles/ (SQ_INSTS_VALU / documents) x 64 lanes;
    # peak = what tools/ubench_valu.hip measures for a dependent integer chain on this chip (lane-ops/s), the nominal
    # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T beside it.
    pmc = text_pmc_summary()
    docs_per_s_gpu = n_docs / (ms / 1e3)
    roof = None
    if pmc:
        ach = pmc["valu_wave_instr_per_doc"] * 64 * docs_per_s_gpu
        roof = {"bound": "valu", "kernel": "text_hash_kernel<false>", "achieved": ach / 1e12, "peak": pmc["valu_peak_lane_ops_per_s"] / 1e12,
                "unit": "T lane-op/s (integer VALU)", "frac": ach / pmc["valu_peak_lane_ops_per_s"],
                "valu_wave_instr_per_doc": pmc["valu_wave_instr_per_doc"], "source": pmc["source"],
                "nominal_peak": 39.3216, "hbm_GBs": n_docs * (doc_len + 1032) / (ms / 1e3) / 1e9, "hbm_frac": n_docs * (doc_len + 1032) / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    res = {"metric": "documents/s (MinHash-128, k=5 word shingles, tokenised on GPU)",
           "simhash": {"ms_per_pass": sms, "docs_per_s": n_docs / (sms / 1e3) * world},
           "value": docs_per_s_gpu * world, "unit": "docs/s", "docs_per_gpu": n_docs, "doc_bytes": doc_len,
           "workload": "SURVEY 8(d) config 4: distinct 4 KiB ASCII documents, 50 k-word Zipf(1.1) vocabulary, every tenth "
                       "document a near-duplicate (5 % of the words redrawn) of its predecessor; generated on the device",
           "ms_per_pass": ms, "algorithmic_GBs": n_docs * (doc_len + 1032) / (ms / 1e3) / 1e9, "roofline": roof}
    # LSH chained to the signatures just computed: index = all records of this rank, queries = the near-duplicates
    # (documents 10 m + 1); a pair is found when document 10 m is among the k best by slot agreement
    nq, k = min(4096, n_docs // 10), 10
    if nq:
        qdocs = torch.arange(nq, device=dev) * 10 + 1
        qrec = out[qdocs].contiguous()
        ids = torch.arange(n_docs, dtype=torch.int64, device=dev)
        o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        o_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        o_ct = torch.empty((nq,), dtype=torch.int32, device=dev)
        chained = {}
        for bands, rows in ((16, 8), (32, 4)):
            idx = text.LshIndex(bands, rows, ctx=ctx)
            idx.build_dev(ids.data_ptr(), out.data_ptr(), n_docs, stream)
            idx.query_dev(qrec.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_ct.data_ptr(), stream)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            idx.build_dev(ids.data_ptr(), out.data_ptr(), n_docs, stream)
            e[1].record()
            for _ in range(5):
                idx.query_dev(qrec.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_ct.data_ptr(), stream)
            e[2].record()
            torch.cuda.synchronize()
            found = float(((o_ids == (qdocs - 1)[:, None]).any(1)).float().mean().item())
            self_top = float((o_ids[:, 0] == qdocs).float().mean().item())
            # the estimate of the pair's Jaccard similarity the signatures give (equal slots / 128), for the record
            a8 = out[qdocs][:, 8:].view(nq, 128, 8)
            b8 = out[qdocs - 1][:, 8:].view(nq, 128, 8)
            jac = float((a8 == b8).all(2).float().mean().item())
            chained[f"{bands}x{rows}"] = {"build_ms": e[0].elapsed_time(e[1]), "query_ms": e[1].elapsed_time(e[2]) / 5,
                                          "qps": nq / (e[1].elapsed_time(e[2]) / 5) * 1e3 * world,
                                          "near_duplicate_found": found, "query_is_its_own_top1": self_top,
                                          "mean_pair_jaccard_estimate": jac}
            idx.close()
        res["lsh_chained"] = {"records": n_docs, "queries": nq, "k": k, **chained,
                              "note": "index built from THIS leg's MinHash records; queries = the near-duplicate documents; "
                                      "found = the original is among the k best.  With ~5 % of the words redrawn a pair shares "
                                      "~0.6 of its shingles: 16 bands x 8 rows (0.6^8 per band) is too selective for that, 32 x 4 "
                                      "is the setting for this similarity range"}
    res["lsh"] = bench_lsh(n_docs, rank, world, dev, ctx)
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        oracle = timed_oracle()
        ns = min(n_docs, max(32768, 2048 * oracle.num_threads()))      # ~0.1 s of CPU work and up: not a 4 ms blip
        docs = [bytes(r) for r in blob[:ns].cpu().numpy()]
        dt, dts, (o, _) = best_of(lambda: oracle.text_minhash_batch(docs))
        res["cpu_baseline"] = {"value": ns / dt, "unit": "docs/s", **host_cpu(), "threads": oracle.num_threads(), "kind": "port",
                               "sample": f"the first {ns} of the 4 KiB documents (best of 3: {', '.join('%.2f' % x for x in dts)} s)",
                               "gpu_matches_oracle_on_sample": bool(np.array_equal(o, out[:ns].cpu().numpy()))}
    return res


def audio_issue_roofline():
    """How far the Wang stream kernel is from ITS roof: the share of cycles the SIMDs' VALU and LDS issue ports are busy
    (PMC, tracked summary profiles/r0x/audio_pmc_summary.json made by tools/pmc_summary.sh)."""
    for rnd in ("r03",):
        f = os.path.join(ROOT, "profiles", rnd, "audio_pmc_summary.json")
        if os.path.exists(f):
            try:
                d = json.load(open(f))
                v, l = d.get("valu_busy_frac"), d.get("lds_busy_frac")
                return {"bound": "valu+lds issue", "kernel": "wang_stream_kernel<true>", "valu_busy_frac": v, "lds_issue_busy_frac": l,
                        "frac": (v or 0) + (l or 0), "lds_pipe_busy_frac": d.get("lds_pipe_busy_frac"),
                        "lds_bank_conflict_share": d.get("lds_bank_conflict_share"),
                        "valu_wave_instr_per_frame": d.get("valu_wave_instr_per_unit"),
                        "lds_wave_instr_per_frame": d.get("lds_wave_instr_per_unit"),
                        "unit": "share of the kernel's SIMD cycles in which the port issues (one wave-instruction per SIMD at a time)",
                        "source": f"profiles/{rnd}/audio_pmc_summary.json"}
            except Exception:  # noqa: BLE001
                pass
    return None


def text_pmc_summary():
    """VALU instruction count per document and the measured VALU peak, from the tracked PMC summary of this round (or the
    last one that has it); None when no summary is tracked (the roofline object is then omitted, not guessed)."""
    for rnd in ("r03", "r02"):
        f = os.path.join(ROOT, "profiles", rnd, "text_pmc_summary.json")
        if os.path.exists(f):
            try:
                d = json.load(open(f))
                return {"valu_wave_instr_per_doc": float(d["valu_wave_instr_per_doc"]),
                        "valu_peak_lane_ops_per_s": float(d["valu_peak_lane_ops_per_s"]), "source": f"profiles/{rnd}/text_pmc_summary.json"}
            except Exception:  # noqa: BLE001
                pass
    return None


def bench_lsh(n, rank, world, dev, ctx, nq=4096, k=10):
    """Banded LSH (16 x 8) over n synthetic MinHash-128 records per GPU: build = band keys + 16 radix
    sorts; query = 4096 near-duplicates (10 % of slots changed) -> top-10 by slot agreement."""
    import torch
    from ucfp_amd import text
    g = torch.Generator(device=dev)
    g.manual_seed(0x15A + rank)
    rec = torch.zeros((n, 1032), dtype=torch.uint8, device=dev)
    rec[:, 8:] = torch.randint(0, 256, (n, 1024), dtype=torch.uint8, device=dev, generator=g)
    ids = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64, device=dev)
    q = rec[:nq].clone()
    flip = torch.rand((nq, 128), device=dev, generator=g) < 0.10
    q[:, 8:].view(nq, 128, 8)[flip] ^= 0x5A
    o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_ct = torch.empty((nq,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    idx = text.LshIndex(16, 8, ctx=ctx)
    idx.build_dev(ids.data_ptr(), rec.data_ptr(), n, stream)
    idx.query_dev(q.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_ct.data_ptr(), stream)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    idx.build_dev(ids.data_ptr(), rec.data_ptr(), n, stream)
    e[1].record()
    for _ in range(5):
        idx.query_dev(q.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_ct.data_ptr(), stream)
    e[2].record()
    torch.cuda.synchronize()
    recall = float((o_ids[:, 0] == ids[:nq]).float().mean().item())
    qms = e[1].elapsed_time(e[2]) / 5
    idx.close()
    return {"bands": 16, "rows": 8, "records_per_gpu": n, "build_ms": e[0].elapsed_time(e[1]),
            "build_records_per_s": n / e[0].elapsed_time(e[1]) * 1e3 * world, "queries": nq, "k": k,
            "query_ms": qms, "qps": nq / qms * 1e3 * world, "top1_is_source": recall}


def bench_cosine(args, rank, world, dev, ctx):
    """Secondary leg: IndexBackend::knn as the reference ships it (cosine over f32 embeddings,
    src/index/embedded/mod.rs:268-360), 1 M x 768-d per GPU, k = 10; every answer checked against torch, and against the reference's own arithmetic on a row sample."""
    import torch
    from ucfp_amd import index
    n, dim, k = args.cosine_rows, 768, 10
    g = torch.Generator(device=dev)
    g.manual_seed(0xC05 + rank)
    rows = torch.randn((n, dim), dtype=torch.float32, device=dev, generator=g)
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ix = index.DeviceIndex(index.COSINE_F32, dim, index.APPEND_ONLY, ctx)
    ix.append_dev(0, ids.data_ptr(), rows.data_ptr(), n, stream)
    out = {}
    for nq in (1, 16, 32, 48, 64, 256):
        q = torch.randn((nq, dim), dtype=torch.float32, device=dev, generator=g)
        o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        o_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        o_key = torch.empty((nq, k), dtype=torch.int32, device=dev)
        o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)

        def step():
            ix.search_dev(0, q.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_key.data_ptr(),
                          o_cnt.data_ptr(), stream)
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        # every answer of the batch against torch (f32 matmul of normalised rows, top-k): ids equal wherever the
        # reference scores are not tied within the tolerance, scores within 1e-5 (north_star)
        rn = torch.nn.functional.normalize(rows, dim=1)
        ok_all, worst = True, 0.0
        for q0 in range(0, nq, 64):
            ref = torch.nn.functional.normalize(q[q0:q0 + 64], dim=1) @ rn.T
            top = torch.topk(ref, k, dim=1)
            worst = max(worst, float((top.values - o_sc[q0:q0 + 64]).abs().max()))
            mism = top.indices != o_ids[q0:q0 + 64]
            if bool(mism.any()):     # an id may differ only where the two candidates' scores agree within the tolerance
                alt = torch.gather(ref, 1, o_ids[q0:q0 + 64])
                ok_all = ok_all and bool(((alt - top.values).abs()[mism] < 1e-5).all())
            del ref
        ok_all = ok_all and worst < 1e-5
        del rn
        # every batch reads the rows once per pass of at most 64 queries (round 4: the chunk minima of 2 .. 64 queries come
        # from the f16 matrix pipe, cosine.hip cosine_mins_f16, and only the listed chunks are rescored in f32): HBM-bound at
        # every batch size; whole search = minima + thresholds + exact keys of the listed chunks + selection
        passes = (nq + 63) // 64
        roof = {"bound": "hbm", "achieved": passes * n * dim * 4 / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": passes * n * dim * 4 / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "passes": passes,
                "algorithmic_bytes": "4 x dim x rows per pass of <= 64 queries (SURVEY 8d)"}
        out[f"batch{nq}"] = {"ms": ms, "qps": nq / ms * 1e3, "all_answers_match_torch_within_1e-5": ok_all,
                             "max_abs_score_diff": worst, "roofline": roof}
        if nq == 256:
            last_q, last_ids, last_sc = q, o_ids, o_sc
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        # CPU leg: the reference's own kernel (dot_product / l2_norm / insert_topk restated line for line, rayon's
        # fold/reduce as OpenMP over row chunks: src/index/embedded/mod.rs:324-340,454-495) on the first 262 144 rows
        import numpy as np
        oracle = timed_oracle()
        m = min(n, 262_144)
        h_rows = rows[:m].cpu().numpy()
        h_ids = np.arange(m, dtype=np.uint64)
        h_q = last_q.cpu().numpy()
        t0 = time.perf_counter()
        c_ids, c_sc, _ = oracle.cosine_knn_batch_omp(h_ids, h_rows, h_q, k)
        dt_b = time.perf_counter() - t0
        t0 = time.perf_counter()
        for j in range(4):
            oracle.cosine_knn_batch_omp(h_ids, h_rows, h_q[j:j + 1], k)
        dt_1 = (time.perf_counter() - t0) / 4
        sub = index.DeviceIndex(index.COSINE_F32, dim, index.APPEND_ONLY, ctx)
        sub.append_dev(0, ids.data_ptr(), rows.data_ptr(), m, stream)
        s_ids = torch.empty((256, k), dtype=torch.int64, device=dev)
        s_sc = torch.empty((256, k), dtype=torch.float32, device=dev)
        s_key = torch.empty((256, k), dtype=torch.int32, device=dev)
        s_cnt = torch.empty((256,), dtype=torch.int32, device=dev)
        sub.search_dev(0, last_q.data_ptr(), 256, k, s_ids.data_ptr(), s_sc.data_ptr(), s_key.data_ptr(), s_cnt.data_ptr(),
                       stream)
        torch.cuda.synchronize()
        g_i, g_s = s_ids.cpu().numpy().view("uint64"), s_sc.cpu().numpy()
        diff = float(np.abs(g_s - c_sc).max())
        ids_ok = bool((g_i == c_ids).mean() > 0.999)      # ids may swap only between scores tied within the tolerance
        sub.close()
        cpu = {"kind": "port", **host_cpu(), "threads": oracle.num_threads(),
               "build": "gcc -O3 -march=native -ffp-contract=off -fopenmp (reference arithmetic, "
                        "oracle/ucfp_oracle_index.c ucfp_oracle_cosine_knn_batch_omp)",
               "sample": f"first {m} of the {n} rows x {dim} d, 256 queries ({dt_b:.2f} s) and 1 query x 4",
               "rows_per_s_batch256": m * 256 / dt_b, "rows_per_s_single_query": m / dt_1,
               "value": m * 256 / dt_b / n, "unit": "queries/s at %d rows (batch of 256)" % n,
               "single_query_value": m / dt_1 / n,
               "gpu_vs_reference_arithmetic_max_abs_score_diff": diff, "gpu_ids_equal_reference": ids_ok}
    ix.close()
    return {"metric": "cosine kNN queries/s (exact, k=10)", "rows_per_gpu": n, "dim": dim,
            "value": out["batch256"]["qps"] * world, "unit": "queries/s", **out, "cpu_baseline": cpu,
            "single_query_row_GBs": n * dim * 4 / out["batch1"]["ms"] / 1e6,
            "reference_claim": "~8 ms per query at 1M x 768 on 16 cores (REPORT.md:1233, unmeasured)"}


def bench_audio(args, rank, world, dev, ctx):
    """Secondary leg (BASELINE configs[2]): Wang landmarks over synthetic 44.1 kHz mono PCM through the ragged-batch
    entry point (ucfp_audio_wang_batch_dev): the stream kernel resamples to 8 kHz itself, cuts the STFT frames from an
    LDS ring, picks peaks; pairs follow -- one stream per GPU, HBM sees the samples once."""
    import numpy as np
    import torch
    from ucfp_amd import _lib
    sr, secs = 44100, args.audio_seconds
    n = sr * secs
    g = torch.Generator(device=dev)
    g.manual_seed(0xA0D10 + rank)
    t = torch.arange(n, dtype=torch.float32, device=dev) / sr
    x = torch.zeros(n, dtype=torch.float32, device=dev)
    for i in range(8):
        f0 = 110.0 * (1.6 ** i)
        x += 0.06 * torch.sin(2 * np.pi * (f0 * t + 3.0 * torch.sin(0.05 * (i + 1) * t)))
    del t
    x += 0.0158 * torch.randn(n, dtype=torch.float32, device=dev, generator=g)
    x.clamp_(-0.5, 0.5)
    lib = _lib.load()
    offs = torch.tensor([0, n], dtype=torch.int64, device=dev)
    cap = int(lib.ucfp_audio_wang_batch_max_hashes(n, 1, sr, None))
    out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    oo = torch.zeros(2, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        _lib.check(lib.ucfp_audio_wang_batch_dev(ctx.handle, x.data_ptr(), offs.data_ptr(), n, 1, sr, None, out.data_ptr(),
                                                 cap, oo.data_ptr(), stream))
    step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    steps = 5
    ev[0].record()
    for _ in range(steps):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    nh = int(oo[1].item())
    # the reference's request shape: a corpus of 4-second clips at 8 kHz (benches/end_to_end.rs:55-75), one call
    clip_n = 4 * 8000
    n_clips = min(8192, n // clip_n)         # a reduced --audio-seconds holds fewer than 8192 clips' worth of samples
    xc = x[:n_clips * clip_n]
    coffs = (torch.arange(n_clips + 1, dtype=torch.int64, device=dev) * clip_n).contiguous()
    ccap = int(lib.ucfp_audio_wang_batch_max_hashes(n_clips * clip_n, n_clips, 8000, None))
    cout = torch.empty((ccap, 2), dtype=torch.int32, device=dev)
    coo = torch.zeros(n_clips + 1, dtype=torch.int64, device=dev)

    def cstep():
        _lib.check(lib.ucfp_audio_wang_batch_dev(ctx.handle, xc.data_ptr(), coffs.data_ptr(), n_clips * clip_n, n_clips, 8000,
                                                 None, cout.data_ptr(), ccap, coo.data_ptr(), stream))
    cstep()
    torch.cuda.synchronize()
    cev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cev[0].record()
    for _ in range(steps):
        cstep()
    cev[1].record()
    torch.cuda.synchronize()
    cms = cev[0].elapsed_time(cev[1]) / steps
    del cout, coo
    # Haitsma-Kalker over the first hour (reference row a4): resample to 5 kHz, 2048-point STFT, 33 bands
    hsecs = min(secs, 3600)
    hn = sr * hsecs
    hm = int(lib.ucfp_audio_resample_len(hn, sr, 5000))
    x5 = torch.empty(hm, dtype=torch.float32, device=dev)
    hframes = int(lib.ucfp_audio_haitsma_frames(hm, 5000))
    hout = torch.empty((max(hframes, 1),), dtype=torch.int32, device=dev)

    def hstep():
        _lib.check(lib.ucfp_audio_resample_linear_dev(ctx.handle, x.data_ptr(), hn, sr, 5000, x5.data_ptr(), hm, stream))
        _lib.check(lib.ucfp_audio_haitsma_dev(ctx.handle, x5.data_ptr(), hm, None, hout.data_ptr(), hframes, stream))
    hstep()
    torch.cuda.synchronize()
    hev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    hev[0].record()
    for _ in range(steps):
        hstep()
    hev[1].record()
    torch.cuda.synchronize()
    hms = hev[0].elapsed_time(hev[1]) / steps
    algo_bytes = n * 4 + nh * 8            # 176 400 B read per audio-second + 8 B per hash written
    res = {"haitsma": {"seconds": hsecs, "ms_per_pass": hms, "x_real_time": hsecs / (hms / 1e3) * world,
                       "frames": hframes},
           "metric": "audio-seconds/s (Wang landmarks incl. 44.1k->8k linear resample, fused)",
           "value": secs / (ms / 1e3) * world, "unit": "x real time", "seconds_per_gpu": secs, "ms_per_pass": ms,
           "hashes": nh, "algorithmic_GBs": algo_bytes / (ms / 1e3) / 1e9,
           "roofline": {"bound": "hbm", "achieved": algo_bytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                        "note": "the SURVEY 8(d) yardstick; the stream kernel itself is bound by VALU + LDS issue "
                                "(roofline_issue): 446 frames x (real 1024-point FFT + peak picking) per ms per CU"},
           "roofline_issue": audio_issue_roofline(),
           "clips_4s_8k": {"clips": n_clips, "ms_per_batch": cms, "clips_per_s": n_clips / (cms / 1e3) * world},
           "note": "one launch sequence per batch; the resampler runs inside the STFT kernel (LDS sample ring); peaks are "
                   "picked in the same kernel (LDS ring of row maxima); no spectrogram spill"}
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        oracle = timed_oracle()
        s = 60 * sr
        xs = x[:s].cpu().numpy()
        t0 = time.perf_counter()
        o = oracle.wang(oracle.resample_linear(xs, sr, 8000))
        dt = time.perf_counter() - t0
        so = torch.tensor([0, s], dtype=torch.int64, device=dev)
        _lib.check(lib.ucfp_audio_wang_batch_dev(ctx.handle, x.data_ptr(), so.data_ptr(), s, 1, sr, None, out.data_ptr(), cap,
                                                 oo.data_ptr(), stream))
        torch.cuda.synchronize()
        gh = out[:int(oo[1].item())].cpu().numpy().view(np.uint32)
        res["cpu_baseline"] = {"value": 60.0 / dt, "unit": "x real time", **host_cpu(), "threads": oracle.num_threads(), "kind": "port",
                               "sample": "first 60 s (resample + Wang, C restatement under OpenMP)",
                               "gpu_matches_oracle_on_sample": bool(np.array_equal(gh, o))}
    return res


_LINE_FD = None


def emit_line(res):
    """The ONE JSON line goes to the process's original stdout; everything else written to fd 1 while the bench runs
    (RCCL prints a version banner there when a communicator is created) has been sent to stderr by main()."""
    data = (json.dumps(res) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)


def main():
    global _LINE_FD
    args = parse()
    # keep stdout to exactly one line: native libraries (RCCL's init banner) write to fd 1 directly
    sys.stdout.flush()
    _LINE_FD = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched via torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    from ucfp_amd import _lib, image

    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    # UCFP_BENCH_REHEARSAL=1: every rank uses GPU 0 and the collectives run over gloo -- lets the
    # N > 1 code path be exercised on a one-GPU box. Never used for reported numbers.
    rehearsal = os.environ.get("UCFP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = _lib.Context(local_rank)
    lib = _lib.load()
    n = args.frames
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- resident synthetic batch (generated on device; rank r holds frames [r*n, (r+1)*n)) ----
    frames = torch.empty((n, FRAME_SIDE, FRAME_SIDE), dtype=torch.uint8, device=dev)
    _lib.check(lib.ucfp_image_synth_dev(ctx.handle, frames.data_ptr(), n, FRAME_SIDE, FRAME_SIDE,
                                        rank * n, stream))
    out = torch.zeros((n, REC_BYTES), dtype=torch.uint8, device=dev)
    status = torch.zeros((n,), dtype=torch.int32, device=dev)

    def step():
        image.fingerprint_frames_dev(frames.data_ptr(), n, FRAME_SIDE, FRAME_SIDE, algo=image.MULTI,
                                     pixfmt=image.PIX_GRAY8, out_ptr=out.data_ptr(),
                                     status_ptr=status.data_ptr(), stream=stream, ctx=ctx)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events on the stream the kernel is launched on (torch's current stream == `stream`)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(status.abs().sum().item()) == 0, "some frames were rejected"

    # ---- RGB8 variant of the same workload ("also report RGB8"): 786 432 B/frame ----
    rgb = None
    if args.rgb_frames > 0:
        nr = args.rgb_frames
        rgbf = torch.empty((nr, FRAME_SIDE, FRAME_SIDE * 3), dtype=torch.uint8, device=dev)
        _lib.check(lib.ucfp_image_synth_dev(ctx.handle, rgbf.data_ptr(), nr, FRAME_SIDE * 3, FRAME_SIDE, 0, stream))
        rout = torch.empty((nr, REC_BYTES), dtype=torch.uint8, device=dev)

        def rstep():
            image.fingerprint_frames_dev(rgbf.data_ptr(), nr, FRAME_SIDE, FRAME_SIDE, algo=image.MULTI,
                                         pixfmt=image.PIX_RGB8, out_ptr=rout.data_ptr(), stream=stream, ctx=ctx)
        rstep()
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(5):
            rstep()
        r1.record()
        torch.cuda.synchronize()
        rms = r0.elapsed_time(r1) / 5
        rgb = {"frames": nr, "ms_per_launch": rms, "fingerprints_per_s": nr / rms * 1e3,
               "achieved_GBs": nr * (FRAME_SIDE * FRAME_SIDE * 3 + REC_BYTES) / rms / 1e6,
               "frac_of_hbm_peak": nr * (FRAME_SIDE * FRAME_SIDE * 3 + REC_BYTES) / rms / 1e6 / HBM_PEAK_GBS}
        del rgbf, rout
    # ---- the same launch followed by what an ingest does with the records: the three 64-bit global hashes of every
    # bundle go into per-algorithm Hamming shards, all on the device (hash -> ucfp_image_record_codes_dev ->
    # ucfp_index_append_dev); the main line above stays the bare fingerprint rate BASELINE names ----
    from ucfp_amd import index as _index
    shards = [_index.DeviceIndex(_index.HAMMING64, 0, _index.APPEND_ONLY, ctx) for _ in range(3)]
    codes = torch.empty((3, n), dtype=torch.int64, device=dev)
    rec_ids = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64, device=dev)

    def ingest_step():
        step()
        for a, which in enumerate((image.AHASH, image.PHASH, image.DHASH)):
            image.record_codes_dev(out.data_ptr(), n, codes[a].data_ptr(), algo=image.MULTI, which=which, stream=stream,
                                   ctx=ctx)
            shards[a].append_dev(0, rec_ids.data_ptr(), codes[a].data_ptr(), n, stream)
    ingest_step()
    torch.cuda.synchronize()
    i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    i0.record()
    for _ in range(5):
        ingest_step()
    i1.record()
    torch.cuda.synchronize()
    ingest_ms = i0.elapsed_time(i1) / 5
    ingest = {"ms_per_batch": ingest_ms, "fingerprints_per_s": n / ingest_ms * 1e3 * world,
              "indexed_codes_per_gpu": int(shards[0].size(0)), "what": "hash + 3 x (record -> code, append to a Hamming shard)"}
    for sh in shards:
        sh.close()
    del shards, codes, rec_ids
    if rank == 0:
        total_frames = n * world * args.steps
        value = total_frames / elapsed
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        achieved = ALGO_BYTES_PER_FRAME * n / avg_kernel_s / 1e9
        res = {
            "metric": "fingerprints/sec (batched ingest)",
            "value": value,
            "unit": "fingerprints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: ranks share one GPU, not a result)",
            "config": {
                "workload": f"image multi (pHash+dHash+aHash bundle, 536 B/frame), {n} synthetic "
                            f"{FRAME_SIDE}x{FRAME_SIDE} GRAY8 frames per GPU resident in HBM "
                            "(BASELINE.json configs[1])",
                "frames_per_gpu": n, "width": FRAME_SIDE, "height": FRAME_SIDE,
                "pixfmt": "gray8", "algorithm": "multi", "sharding": f"frames/{world} (no collective)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "image_hash_gray_kernel<2>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * n,
                "avg_launch_ms": avg_kernel_s * 1e3,
                "traffic": traffic_from_profiles(n) if args.traffic_bytes is None else args.traffic_bytes,
            },
        }
        res["rgb8_variant"] = rgb
        res["ingest_to_index"] = ingest
        res["ann"] = None
        if args.cpu_sample > 0 and world == 1:
            want = min(n, max(args.cpu_sample, 64 * cpu_share()["usable_cpus"]))     # >= 64 frames per thread
            head = out[:want].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(want, head)
            res["config1_phash_png"] = bench_config1_phash_png(dev, ctx)
            res["search_batcher"] = bench_search_batcher()
            try:
                res["upload_mix"] = bench_upload_mix(dev, ctx)
            except Exception as e:   # noqa: BLE001  (a secondary leg: reported, never substituted)
                res["upload_mix"] = {"error": f"{type(e).__name__}: {e}"}
        elif args.cpu_sample > 0:
            res["cpu_baseline"] = None  # measured at N=1 only (see BENCH at n_gpus=1)
    else:
        res = None
    # ---- secondary leg: sharded Hamming ANN (frees the image batch first) ----
    del frames, out, status
    torch.cuda.empty_cache()
    # At N > 1 the secondary legs below contain the only collectives of the run that go through the library's own RCCL
    # communicator.  A rank that fails alone would leave the others waiting in a collective for ever and the driver
    # without any line: a watchdog prints what has been measured (the headline above) and ends the rank instead.
    watchdog = None
    if world > 1:
        import threading

        def _fire():
            if rank == 0 and res is not None:
                res["watchdog"] = ("a secondary leg did not finish within 600 s at N > 1; the line carries what was "
                                   "measured before it")
                emit_line(res)
            os._exit(3)      # non-zero: the driver must record the hang, not a success
        watchdog = threading.Timer(600.0, _fire)
        watchdog.daemon = True
        watchdog.start()
    if args.ann_corpus > 0:
        # a failure of this secondary leg (e.g. the RCCL communicator cannot be created on this node) is REPORTED in
        # the line, it does not take the headline measurement above down with it; nothing is substituted for it
        try:
            ann = bench_ann(args, rank, world, dev, ctx)
            torch.cuda.empty_cache()
            # the size BASELINE.json's metric string quotes ("ANN queries/sec @10M corpus")
            ann10 = bench_ann(args, rank, world, dev, ctx, corpus_total=10_000_000)
            # one GPU: the shard a rank holds when the 100 M corpus is spread over 8 GPUs (what the 8-GPU point runs per rank)
            ann12 = bench_ann(args, rank, world, dev, ctx, corpus_total=12_500_000) if world == 1 else None
            # ... and the shard a rank holds at "@10M corpus" on 8 GPUs (where the fixed stages of a search weigh most)
            ann1m = bench_ann(args, rank, world, dev, ctx, corpus_total=1_250_000) if world == 1 else None
        except Exception as e:   # noqa: BLE001
            ann = ann10 = ann12 = ann1m = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            res["ann"] = ann
            res["ann_10m"] = ann10
            if world == 1:
                res["ann_shard_12m5"] = ann12
                res["ann_shard_1m25"] = ann1m
    if args.cosine_rows > 0:
        r = bench_cosine(args, rank, world, dev, ctx)
        if rank == 0:
            res["cosine"] = r
        torch.cuda.empty_cache()
    if args.text_docs > 0:
        r = bench_text(args, rank, world, dev, ctx)
        if rank == 0:
            res["text"] = r
        torch.cuda.empty_cache()
    if args.audio_seconds > 0:
        r = bench_audio(args, rank, world, dev, ctx)
        if rank == 0:
            res["audio"] = r
        torch.cuda.empty_cache()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        emit_line(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
