"""N > 1 path on CPU: two gloo ranks exercise the sharding rule and the all-gather plumbing of
ucfp_amd/sharded.py.  The gathered [parts][nq][k] tensors are merged here by a numpy checker with
the specified order (key asc, id asc) and compared with the oracle over the whole corpus; on the
GPU box the same tensors feed ucfp_topk_merge_dev (tests/test_index_gpu.py covers that kernel)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, nq, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from ucfp_amd import sharded
        rng = np.random.default_rng(123)           # same stream on every rank = replicated inputs
        codes = rng.integers(0, 2**64, n, dtype=np.uint64)
        ids = rng.permutation(n).astype(np.uint64)
        queries = codes[:nq] ^ np.uint64(0b1011)
        s, e = sharded.shard_range(n, rank, world)
        # local top-k of this rank's shard (the oracle stands in for the HIP search on CPU)
        l_ids, l_d, _ = oracle.hamming_topk(ids[s:e], codes[s:e], queries, k)
        g_ids, g_keys = sharded.all_gather_topk(torch.from_numpy(l_ids.view(np.int64)),
                                                torch.from_numpy(l_d.view(np.int32)))
        assert g_ids.shape == (world, nq, k) and g_keys.shape == (world, nq, k)
        gi = g_ids.numpy().view(np.uint64)
        gk = g_keys.numpy().view(np.uint32)
        merged_ids = np.zeros((nq, k), np.uint64)
        merged_d = np.zeros((nq, k), np.uint32)
        for q in range(nq):
            cand = sorted((int(gk[p, q, j]), int(gi[p, q, j])) for p in range(world) for j in range(k)
                          if gk[p, q, j] != 0xFFFFFFFF)[:k]
            merged_d[q, :len(cand)] = [c[0] for c in cand]
            merged_ids[q, :len(cand)] = [c[1] for c in cand]
        o_ids, o_d, _ = oracle.hamming_topk(ids, codes, queries, k)
        ok = np.array_equal(merged_ids, o_ids) and np.array_equal(merged_d, o_d)
        # every rank ends with the same answer
        t = torch.tensor([int(ok)])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(t.item()))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from ucfp_amd import sharded
    for n in (0, 1, 7, 8, 100, 12_500_001):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allgather_merge_matches_global():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5001, 12, 10, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1
