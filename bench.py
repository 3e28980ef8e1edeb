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
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/)")
    return ap.parse_args()


def cpu_baseline(sample: int, gpu_records_head):
    """Time the CPU oracle (OpenMP over frames) on the first `sample` frames of the same
    synthetic workload, and use the occasion to check the GPU records of those frames."""
    import numpy as np
    import oracle
    oracle.build()
    cores = oracle.num_threads()
    frames = oracle.image_synth(sample, FRAME_SIDE, FRAME_SIDE, 0)
    t0 = time.perf_counter()
    recs, _ = oracle.image_hash_batch(frames, 7)
    dt = time.perf_counter() - t0
    parity = None
    if gpu_records_head is not None:
        m = min(sample, gpu_records_head.shape[0])
        parity = bool(np.array_equal(recs[:m], gpu_records_head[:m]))
    return {
        "value": sample / dt, "unit": "fingerprints/s", "cores": cores, "kind": "port",
        "sample": f"first {sample} frames of the same synthetic batch ({dt:.2f} s wall, "
                  f"{dt * cores:.1f} core-s); C restatement oracle/ucfp_oracle_image.c, "
                  "not the reference Rust binary (no Rust toolchain, SDK crates un-vendored)",
        "gpu_matches_oracle_on_sample": parity,
    }


def main():
    args = parse()
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(status.abs().sum().item()) == 0, "some frames were rejected"

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
            "data": "synthetic",
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
                "traffic": args.traffic_bytes,
            },
        }
        if args.cpu_sample > 0 and world == 1:
            head = out[:min(args.cpu_sample, n)].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), head)
        elif args.cpu_sample > 0:
            res["cpu_baseline"] = None  # measured at N=1 only (see BENCH at n_gpus=1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
