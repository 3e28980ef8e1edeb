"""N > 1 path on CPU: two gloo ranks exercise the sharding rule and the wire format of the sharded search
(ucfp_amd/sharded.py: packed 16-byte entries, ONE all-gather into [parts][nq][k]).  The gathered entries are
merged here by a numpy checker with the specified order (key asc, id asc) and compared with the oracle over the
whole corpus; on the GPU box the same entries feed ucfp_topk_merge_packed_dev, and the RCCL path inside the library
(ucfp_index_search_sharded_*) is exercised at world = 1 (tests/test_sharded_gpu.py)."""
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
        ent = sharded.pack_entries(l_ids, l_d)
        assert ent.shape == (nq, k, 2) and ent.dtype == np.int64 and ent.nbytes == nq * k * sharded.ENTRY_BYTES
        g = sharded.all_gather_entries(torch.from_numpy(ent))            # ONE collective
        assert g.shape == (world, nq, k, 2)
        gi, gk = sharded.unpack_entries(g.numpy())
        assert np.array_equal(gi[rank], l_ids) and np.array_equal(gk[rank], l_d)      # own slot = own list
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
    import ctypes as C
    from ucfp_amd import _lib, sharded
    lib = _lib.load()
    for n in (0, 1, 7, 8, 100, 12_500_001):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, world) for r in range(world)]
            for r in range(world):       # the C rule (ucfp_shard_range) is the same rule
                a, b = C.c_uint64(0), C.c_uint64(0)
                lib.ucfp_shard_range(n, r, world, C.byref(a), C.byref(b))
                assert (a.value, b.value) == spans[r]
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


def test_wire_format_is_16_little_endian_bytes():
    from ucfp_amd import sharded
    ids = np.array([[0x0102030405060708, 0xFFFFFFFFFFFFFFFF]], np.uint64)
    keys = np.array([[7, 0xFFFFFFFF]], np.uint32)
    raw = sharded.pack_entries(ids, keys).tobytes()
    assert raw[:16] == bytes([8, 7, 6, 5, 4, 3, 2, 1, 7, 0, 0, 0, 0, 0, 0, 0])
    assert raw[16:] == b"\xff" * 12 + b"\0" * 4
    i2, k2 = sharded.unpack_entries(sharded.pack_entries(ids, keys))
    assert np.array_equal(i2, ids) and np.array_equal(k2, keys)


def test_bench_corpus_is_a_function_of_the_global_row():
    """bench.py's config-5 corpus (xorshift64*(0x5EED, index)): the union of the shards is the same corpus at
    every world size, so an N > 1 line's answers_crc can be compared with the N = 1 line's."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from ucfp_amd import sharded
    m = (1 << 64) - 1

    def ref(i):
        x = ((i + 1) * 0x9E3779B97F4A7C15) & m
        x ^= 0x5EED
        x ^= x >> 12
        x ^= (x << 25) & m
        x ^= x >> 27
        return (x * 0x2545F4914F6CDD1D) & m

    n = 5003
    whole = bench.ann_corpus_codes(torch, "cpu", 0, n)
    assert [int(v) & m for v in whole[:50].tolist()] == [ref(i) for i in range(50)]
    for world in (2, 3, 8):
        parts = [bench.ann_corpus_codes(torch, "cpu", *sharded.shard_range(n, r, world)) for r in range(world)]
        assert torch.equal(torch.cat(parts), whole)
    # planted rows are global positions: the shards' plants together are the plants of the whole
    q = torch.arange(64, dtype=torch.int64) * 7919
    a = whole.clone()
    bench._ann_plant(torch, a, q, 0, n, n)
    for world in (2, 8):
        got = []
        for r in range(world):
            s, e = sharded.shard_range(n, r, world)
            c = whole[s:e].clone()
            bench._ann_plant(torch, c, q, s, e, n)
            got.append(c)
        assert torch.equal(torch.cat(got), a)
    assert int((a != whole).sum()) == 32
