#!/usr/bin/env python3
"""Hamming top-k timing probe (one GPU): python tools/bench_hamming.py --n 100000000 --nq 4096
Prints one JSON line per (n, nq) with ms per search and pairs/s; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[100_000_000])
    ap.add_argument("--nq", type=int, nargs="+", default=[4096])
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--streams", type=int, default=1, help="2: batches alternate between two streams (two searches in flight)")
    ap.add_argument("--clustered", type=int, default=0, help="centres: half of the corpus sits within 6 bits of one of them, queries are perturbed corpus rows")
    a = ap.parse_args()
    import torch
    from ucfp_amd import _lib, index
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    for n in a.n:
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        codes = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
        if a.clustered:
            centres = torch.randint(-2**63, 2**63 - 1, (a.clustered,), dtype=torch.int64, device=dev, generator=g)
            near = centres[torch.randint(0, a.clustered, (n // 2,), device=dev, generator=g)]
            for _ in range(6):
                bit = torch.randint(0, 63, (n // 2,), device=dev, generator=g)
                flip = torch.rand((n // 2,), device=dev, generator=g) < 0.5
                near = torch.where(flip, near ^ (torch.ones_like(near) << bit), near)
            codes[: n // 2] = near
            codes = codes[torch.randperm(n, device=dev, generator=g)]
        corpus_sample = codes[torch.randint(0, n, (max(a.nq),), device=dev, generator=g)].clone()
        ids = torch.arange(n, dtype=torch.int64, device=dev)
        ix = index.DeviceIndex(index.HAMMING64, flags=index.APPEND_ONLY, ctx=ctx)
        ix.append_dev(0, ids.data_ptr(), codes.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        del codes, ids
        for nq in a.nq:
            q = torch.randint(-2**63, 2**63 - 1, (nq,), dtype=torch.int64, device=dev, generator=g)
            if a.clustered:
                q = corpus_sample[:nq] ^ (torch.ones((nq,), dtype=torch.int64, device=dev) << (torch.arange(nq, device=dev) % 61))
            o_ids = torch.empty((nq, a.k), dtype=torch.int64, device=dev)
            o_sc = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
            o_d = torch.empty((nq, a.k), dtype=torch.int32, device=dev)
            o_ct = torch.empty((nq,), dtype=torch.int32, device=dev)
            st = torch.cuda.current_stream().cuda_stream

            def go():
                ix.search_dev(0, q.data_ptr(), nq, a.k, o_ids.data_ptr(), o_sc.data_ptr(), o_d.data_ptr(),
                              o_ct.data_ptr(), st)
            go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                go()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            row = {"n": n, "nq": nq, "k": a.k, "ms": ms, "qps": nq / ms * 1e3, "T_pairs_per_s": n * nq / ms / 1e9}
            if a.streams == 2:
                import time
                ss = [torch.cuda.Stream(), torch.cuda.Stream()]
                outs = [(torch.empty_like(o_ids), torch.empty_like(o_sc), torch.empty_like(o_d), torch.empty_like(o_ct))
                        for _ in range(2)]

                def go2(i):
                    oi, os_, od, oc = outs[i & 1]
                    ix.search_dev(0, q.data_ptr(), nq, a.k, oi.data_ptr(), os_.data_ptr(), od.data_ptr(), oc.data_ptr(),
                                  ss[i & 1].cuda_stream)
                for i in range(4):
                    go2(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps2 = a.reps * 4
                for i in range(reps2):
                    go2(i)
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t0) / reps2 * 1e3
                assert torch.equal(outs[0][0], o_ids) and torch.equal(outs[1][2], o_d)
                row["ms_two_streams"] = ms2
                row["qps_two_streams"] = nq / ms2 * 1e3
            print(json.dumps(row), flush=True)
        ix.close()


if __name__ == "__main__":
    main()
