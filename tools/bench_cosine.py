#!/usr/bin/env python3
"""Micro-benchmark of the cosine kNN path (IndexBackend::knn) on one GPU; prints one JSON line per case."""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ucfp_amd import _lib, index  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, nargs="+", default=[1, 4, 8, 16, 32, 48, 64, 256])
    ap.add_argument("--shapes", type=str, nargs="+", default=["1000000x768", "4000000x384"])
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ctx = _lib.default_context(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for n, dim in [tuple(int(v) for v in sh.split("x")) for sh in a.shapes]:
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        rows = torch.randn((n, dim), dtype=torch.float32, device=dev, generator=g)
        ids = torch.arange(n, dtype=torch.int64, device=dev)
        ix = index.DeviceIndex(index.COSINE_F32, dim, index.APPEND_ONLY, ctx)
        ix.append_dev(0, ids.data_ptr(), rows.data_ptr(), n, stream)
        torch.cuda.synchronize()
        for nq in a.nq:
            q = torch.randn((nq, dim), dtype=torch.float32, device=dev, generator=g)
            k = 10
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
            reps = 5
            e0.record()
            for _ in range(reps):
                step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            per = 48 if dim <= 784 else 16
            passes = (nq + per - 1) // per
            # check against torch on the first query
            ref = torch.nn.functional.normalize(rows, dim=1) @ torch.nn.functional.normalize(q[0], dim=0)
            top = torch.topk(ref, k)
            ok = bool(torch.equal(top.indices.cpu(), o_ids[0].cpu())) and \
                float((top.values.cpu() - o_sc[0].cpu()).abs().max()) < 1e-5
            print(json.dumps({"n": n, "dim": dim, "nq": nq, "ms": ms, "qps": nq / ms * 1e3,
                              "corpus_passes": passes, "row_GBs": passes * n * dim * 4 / ms / 1e6,
                              "matches_torch_top10": ok}), flush=True)
        ix.close()
        del rows


if __name__ == "__main__":
    main()
