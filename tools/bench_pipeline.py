#!/usr/bin/env python3
"""Two Hamming searches in flight through ShardedIndex submit / collect: one at a time (host-synchronised), pipelined the way
bench.py does it, and with the caller alternating its own stream too.  python tools/exp_pipeline.py [codes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucfp_amd import _lib, index, sharded

n, nq, k, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000, 4096, 10, 16
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
g = torch.Generator(device=dev); g.manual_seed(1)
codes = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
ids = torch.arange(n, dtype=torch.int64, device=dev)
q = torch.randint(-2**63, 2**63 - 1, (nq,), dtype=torch.int64, device=dev, generator=g)
six = sharded.ShardedIndex(index.HAMMING64, ctx=ctx, group=None)
six.append_local(ids, codes)
torch.cuda.synchronize()

def run(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3

def serial():
    for _ in range(steps):
        six.collect(six.submit(q, k)); torch.cuda.synchronize()
def piped():
    t = six.submit(q, k)
    for _ in range(steps - 1):
        nx = six.submit(q, k); six.collect(t); t = nx
    six.collect(t)
two = [torch.cuda.Stream(), torch.cuda.Stream()]
def piped2():   # the caller alternates its own stream too
    ts = []
    for i in range(steps):
        with torch.cuda.stream(two[i & 1]):
            if len(ts) >= 2: six.collect(ts[-2])
            ts.append(six.submit(q, k))
    for i, t in enumerate(ts[-2:]):
        with torch.cuda.stream(two[(steps - 2 + i) & 1]): six.collect(t)
for name, fn in (("serial", serial), ("pipelined (bench.py)", piped), ("pipelined, caller alternates streams", piped2)):
    fn()
    print(name, round(run(fn), 4), "ms per batch", flush=True)
t0 = time.perf_counter(); tk = six.submit(q, k); t1 = time.perf_counter(); six.collect(tk); torch.cuda.synchronize()
print("host time of one submit:", round((t1 - t0) * 1e6), "us")
