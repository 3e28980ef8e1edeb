#!/usr/bin/env python3
"""Randomised differential soak of the index searches against plain torch on the same GPU (exhaustive distances +
torch.topk): Hamming (d, id) bit-exact, cosine ids wherever the score gaps exceed the tolerance.
    python tools/soak_index.py --seconds 120 --seed 1"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ucfp_amd import _lib, index  # noqa: E402


def popcount64(x):
    x = x - ((x >> 1) & 0x5555555555555555)
    x = (x & 0x3333333333333333) + ((x >> 2) & 0x3333333333333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F0F0F0F0F
    return ((x * 0x0101010101010101) >> 56) & 0xFF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    two = [torch.cuda.Stream(), torch.cuda.Stream()]
    t0, rounds = time.time(), 0
    while time.time() - t0 < a.seconds:
        rounds += 1
        if rounds % 2:
            n = int(rng.choice([300, 5000, 70_000, 300_000, 1_200_000, 3_000_000]))
            # (1 .. 8 queries with k <= 32 take the single-launch search of round 3, hamming_direct.hip)
            nq = int(rng.choice([1, 1, 2, 3, 5, 8, 8, 17, 64, 65, 200, 700, 2049, 4096]))
            k = int(rng.choice([1, 5, 10, 16, 17, 32, 50]))
            g = torch.Generator(device=dev)
            g.manual_seed(int(rng.integers(1 << 30)))
            codes = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
            if rng.random() < 0.5:     # clustered: half of the corpus near a few centres
                c = codes[torch.randint(0, n, (8,), device=dev, generator=g)]
                near = c[torch.randint(0, 8, (n // 2,), device=dev, generator=g)]
                near = near ^ (torch.ones_like(near) << torch.randint(0, 63, (n // 2,), device=dev, generator=g))
                codes[: n // 2] = near
            if rng.random() < 0.15:    # a corpus of a few distinct codes: every distance ties, the id order decides
                codes = codes[torch.randint(0, n, (4,), device=dev, generator=g)][torch.randint(0, 4, (n,), device=dev, generator=g)]
            ids = torch.randperm(n, device=dev, generator=g).to(torch.int64)
            if rng.random() < 0.5:     # ids ascending with the row: the stages after the first filter strictly (hamming_list_tau)
                ids = torch.sort(ids * 3 + 1).values
                if rng.random() < 0.3 and n > 10:     # ... except for one inversion far into the corpus
                    ids[n - 5], ids[n - 4] = ids[n - 4].clone(), ids[n - 5].clone()
            if rng.random() < 0.3:     # extreme words: the packed fields of the matrix filter hold sums -64 .. +63
                pos = torch.randint(0, n, (6,), device=dev, generator=g)
                codes[pos[:3]] = -1
                codes[pos[3:]] = 0
            q = codes[torch.randint(0, n, (nq,), device=dev, generator=g)] ^ 5
            if rng.random() < 0.3:
                q[0] = -1
                q[nq // 2] = 0
            ix = index.DeviceIndex(index.HAMMING64, flags=index.APPEND_ONLY, ctx=ctx)
            ix.append_dev(0, ids.data_ptr(), codes.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
            o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
            o_d = torch.empty((nq, k), dtype=torch.int32, device=dev)
            o_c = torch.empty((nq,), dtype=torch.int32, device=dev)
            # two searches in flight on two streams (different queries): the index alternates its two workspaces
            q2 = q ^ (q << 7) ^ 0x5DEECE66D
            o2_ids, o2_d, o2_c = torch.empty_like(o_ids), torch.empty_like(o_d), torch.empty_like(o_c)
            torch.cuda.synchronize()
            ix.search_dev(0, q.data_ptr(), nq, k, o_ids.data_ptr(), 0, o_d.data_ptr(), o_c.data_ptr(), two[0].cuda_stream)
            ix.search_dev(0, q2.data_ptr(), nq, k, o2_ids.data_ptr(), 0, o2_d.data_ptr(), o2_c.data_ptr(), two[1].cuda_stream)
            torch.cuda.synchronize()
            kk = min(k, n)
            for qq, oi, od in ((q, o_ids, o_d), (q2, o2_ids, o2_d)):
                for j in range(0, nq, 64):
                    d = popcount64(qq[j:j + 64, None] ^ codes[None, :])
                    key = d * (1 << 32) + ids[None, :]                 # (d, id) ascending
                    ref = torch.topk(key, kk, dim=1, largest=False).values
                    assert torch.equal(od[j:j + 64, :kk].to(torch.int64), ref >> 32), ("hamming d", n, nq, k)
                    assert torch.equal(oi[j:j + 64, :kk], ref & 0xFFFFFFFF), ("hamming id", n, nq, k)
            ix.close()
        else:
            n = int(rng.choice([900, 5000, 40_000, 131_072, 140_001, 270_000, 600_000]))
            dim = int(rng.choice([32, 64, 96, 100, 128, 160, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1152, 1536, 2048]))
            nq = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 33, 40, 48, 49, 64, 65, 100, 130, 300]))
            k = int(rng.choice([1, 10, 20, 50, 64]))
            if n * dim > 250_000_000:
                n = 250_000_000 // dim
            g = torch.Generator(device=dev)
            g.manual_seed(int(rng.integers(1 << 30)))
            rows = torch.randn((n, dim), dtype=torch.float32, device=dev, generator=g)
            ids = torch.randperm(n, device=dev, generator=g).to(torch.int64)
            q = torch.randn((nq, dim), dtype=torch.float32, device=dev, generator=g)
            planted = []
            if rng.random() < 0.5:      # clusters of near-matches around some queries (scores 1 - O(noise^2)): inside the f16 minima's margin
                for j in rng.integers(0, nq, int(rng.integers(1, 5))):
                    m = int(rng.choice([5, 30, 200]))
                    pos = torch.from_numpy(rng.integers(0, n, m)).to(dev)
                    noise = float(rng.choice([1e-3, 1e-2, 5e-2]))
                    rows[pos] = q[int(j)][None, :] + noise * torch.randn((m, dim), dtype=torch.float32, device=dev, generator=g)
                    planted.append(("cluster", int(j), m, noise))
            if rng.random() < 0.3:      # rows of very different magnitudes: the scores do not change
                pos = torch.from_numpy(rng.integers(0, n, 64)).to(dev)
                scale = float(rng.choice([1e-9, 1e-4, 1e6, 1e11]))
                rows[pos] *= scale
                planted.append(("scale", scale))
            ix = index.DeviceIndex(index.COSINE_F32, dim, index.APPEND_ONLY, ctx)
            ix.append_dev(0, ids.data_ptr(), rows.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
            o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
            o_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
            o_c = torch.empty((nq,), dtype=torch.int32, device=dev)
            ix.search_dev(0, q.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), 0, o_c.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            sc = (q.double() @ rows.double().T) / (q.double().norm(dim=1)[:, None] * rows.double().norm(dim=1)[None, :])
            kk = min(k, n)
            ref_sc, ref_ix = torch.topk(sc, min(kk + 1, n), dim=1)
            diff = (o_sc[:, :kk].double() - ref_sc[:, :kk]).abs()
            if float(diff.max()) > 1e-5:       # what was planted, which query and rank, what came back
                qi = int(diff.max(dim=1).values.argmax())
                print("cosine score mismatch:", (n, dim, nq, k), "planted", planted, "query", qi, "counts", o_c.tolist(),
                      "\n gpu", o_sc[qi, :kk].tolist(), "\n ref", ref_sc[qi, :kk].tolist(), flush=True)
            assert float(diff.max()) <= 1e-5, ("cosine score", n, dim, nq, k)
            gap = (ref_sc[:, :-1] - ref_sc[:, 1:]) > 2e-5 if ref_sc.shape[1] > 1 else None
            ref_ids = ids[ref_ix[:, :kk]]
            same = o_ids[:, :kk] == ref_ids
            if gap is not None:
                ok = torch.ones_like(same)
                ok[:, :gap.shape[1]][:, :kk] &= gap[:, :kk]          # gap to the next
                ok[:, 1:] &= gap[:, :kk - 1] if kk > 1 else ok[:, 1:]  # gap to the previous
                assert bool((same | ~ok).all()), ("cosine ids", n, dim, nq, k)
            ix.close()
            del rows, sc
    print(f"soak ok: {rounds} random configurations in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
