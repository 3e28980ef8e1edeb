import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
oracle.build()
from ucfp_amd import index, _lib
ctx = _lib.default_context(0)
rng = np.random.default_rng(20260101)
want = int(sys.argv[1]) if len(sys.argv) > 1 else 11
for trial in range(20):
    n = int(rng.integers(1, 1_500_000)) if trial % 4 else int(rng.integers(262_144, 400_000))
    nq = int(rng.integers(1, 260))
    k = int(rng.choice([1, 3, 10, 17, 64, 128]))
    style = trial % 3
    if style == 0:
        codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    elif style == 1:
        centres = rng.integers(0, 2**64, 50, dtype=np.uint64)
        codes = centres[rng.integers(0, 50, n)]
        for _ in range(6):
            flip = rng.random(n) < 0.5
            codes = np.where(flip, codes ^ (np.uint64(1) << rng.integers(0, 64, n).astype(np.uint64)), codes)
    else:
        codes = rng.integers(0, 2**12, n, dtype=np.uint64) * np.uint64(0x0010000100001001)
    queries = codes[rng.integers(0, n, nq)] ^ (np.uint64(1) << rng.integers(0, 64, nq).astype(np.uint64))
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(1000)
    if trial not in (4, 5):
        continue
    print("trial", trial, n, nq, k, style)
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, queries, k)
    row_of = {int(i): r for r, i in enumerate(ids)}
    ix = index.DeviceIndex(index.HAMMING64, ctx=ctx)
    ix.upsert(0, ids, codes)
    for rep in range(12):
        g_ids, _, g_d, g_c = ix.search(0, queries, k)
        bad = np.argwhere(g_ids != o_ids)
        print("rep", rep, "mismatches", len(bad), "dist mismatches", int((g_d != o_d).sum()))
        for q, r in bad[:6]:
            oi, gi = int(o_ids[q, r]), int(g_ids[q, r])
            print("  q", q, "tile", q // 32, "rank", r, "want id", oi, "row", row_of[oi], "step", row_of[oi] // 128, "in-step", row_of[oi] % 128, "d", int(o_d[q, r]), "got id", gi, "row", row_of.get(gi), "d", int(g_d[q, r]),
                  "want-in-got", oi in set(map(int, g_ids[q])))
    ix.close()
