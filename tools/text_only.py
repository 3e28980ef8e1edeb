#!/usr/bin/env python3
"""The text leg alone (for profilers): N synthetic 4 KiB documents of bench.py's workload through MinHash-128 once.
usage: python3 tools/text_only.py [n_docs = 200000]; prints {"n_docs": N, "ms": t}"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from ucfp_amd import _lib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    blob = bench.synth_docs_dev(n, 4096, dev, 0xD0C5)
    offs = (torch.arange(n + 1, dtype=torch.int64, device=dev) * 4096).contiguous()
    out = torch.empty((n, 1032), dtype=torch.uint8, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.ucfp_text_minhash_batch_dev(ctx.handle, blob.data_ptr(), offs.data_ptr(), n, 0, 5, out.data_ptr(),
                                               st.data_ptr(), s))
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"n_docs": n, "ms": e0.elapsed_time(e1)}), flush=True)


if __name__ == "__main__":
    main()
