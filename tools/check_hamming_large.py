#!/usr/bin/env python3
"""One-off differential check of the Hamming search at sizes the soak does not reach: n codes (default 100 M), batches of
several sizes, against an exhaustive torch evaluation on the same GPU ((d, id) order, bit-exact).
    python tools/check_hamming_large.py --n 100000000 --nq 1 8 9 32 64 256 300"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ucfp_amd import _lib, index  # noqa: E402


def popcount64(x):
    x = x - ((x >> 1) & 0x5555555555555555)
    x = (x & 0x3333333333333333) + ((x >> 2) & 0x3333333333333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F0F0F0F0F
    return ((x * 0x0101010101010101) >> 56) & 0xFF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--nq", type=int, nargs="+", default=[1, 8, 9, 32, 64, 256, 300])
    ap.add_argument("--k", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n = a.n
    codes = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    ix = index.DeviceIndex(index.HAMMING64, flags=index.APPEND_ONLY, ctx=ctx)
    ix.append_dev(0, ids.data_ptr(), codes.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for nq in a.nq:
        rows = torch.randint(0, n, (nq,), device=dev, generator=g)
        q = codes[rows].clone()
        for b in range(3):      # neighbours up to three flips away, in every region of the corpus
            flip = torch.randint(0, 63, (nq,), device=dev, generator=g)
            q ^= (torch.ones_like(q) << flip) * (torch.arange(nq, device=dev) % (b + 2) == 0)
        o_ids = torch.empty((nq, a.k), dtype=torch.int64, device=dev)
        o_sc = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
        o_d = torch.empty((nq, a.k), dtype=torch.int32, device=dev)
        o_ct = torch.empty((nq,), dtype=torch.int32, device=dev)
        ix.search_dev(0, q.data_ptr(), nq, a.k, o_ids.data_ptr(), o_sc.data_ptr(), o_d.data_ptr(), o_ct.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        bad = 0
        for i in range(nq):
            # exhaustive: key = d * 2^40 + row (ids ascend with the row), smallest k
            best = None
            for c0 in range(0, n, 1 << 25):
                c = codes[c0:c0 + (1 << 25)]
                key = popcount64(c ^ q[i]) * (1 << 40) + (torch.arange(c.numel(), device=dev) + c0)
                top = torch.topk(key, min(a.k, key.numel()), largest=False).values
                best = top if best is None else torch.topk(torch.cat([best, top]), a.k, largest=False).values
            want_d = (best >> 40).to(torch.int32)
            want_id = best & ((1 << 40) - 1)
            if not (torch.equal(want_d, o_d[i]) and torch.equal(want_id, o_ids[i]) and int(o_ct[i]) == a.k):
                bad += 1
        print(f"n {n} nq {nq} k {a.k}: {'ok' if bad == 0 else f'{bad} queries WRONG'}", flush=True)
        assert bad == 0


if __name__ == "__main__":
    main()
