"""GPU tests of the round-2 boundary work, all through the C ABI:

  * the sharded search (`ucfp_index_search_sharded_*`, `ucfp_shard_comm_*`) at world = 1 against the oracle, and
    the multi-shard merge (`ucfp_topk_pack_dev` -> concatenated entries = what ONE all-gather delivers ->
    `ucfp_topk_merge_packed_dev`) with the corpus split into 2 / 3 / 8 shard indexes on one device.  The RCCL
    all-gather itself needs one GPU per rank and cannot run on a one-GPU box (RCCL refuses two ranks on a device):
    it is unmeasured here and covered by construction + the 2-rank gloo wire-format test (test_sharded_cpu.py);
  * round 3: the RCCL branch itself at world = 1 -- a ONE-RANK RCCL communicator (legal in RCCL) forced with
    UCFP_SHARD_FORCE_RCCL: dlopen + ncclGetUniqueId + ncclCommInitRank + ONE ncclAllGather per batch on the exchange
    stream + the merge over the gathered buffer, two tickets in flight, vs the oracle (`test_forced_rccl_*`);
  * ordering of the shared normalisation workspace across streams;
  * IndexBackend::upsert overwrite semantics (src/index/embedded/mod.rs:184-191);
  * append on one stream, search on another.
"""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.cuda() if dtype is None else t.to(dtype).cuda()


def test_sharded_search_world1_matches_oracle(gpu_ctx, oracle, torch_cuda):
    torch = torch_cuda
    from ucfp_amd import _lib, index, sharded
    rng = np.random.default_rng(2024)
    n, nq, k = 120_000, 300, 10
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(5)
    q = codes[:nq] ^ np.uint64(0b100101)
    six = sharded.ShardedIndex(index.HAMMING64, ctx=gpu_ctx)
    assert six.world == 1 and six.comm.exchanges() == 0
    six.append_local(_dev(torch, ids.view(np.int64)), _dev(torch, codes.view(np.int64)))
    g_ids, g_sc, g_keys, g_cnt = six.search(_dev(torch, q.view(np.int64)), k)
    torch.cuda.synchronize()
    o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, q, k)
    assert np.array_equal(g_ids.cpu().numpy().view(np.uint64), o_ids)
    assert np.array_equal(g_keys.cpu().numpy().view(np.uint32), o_d)
    assert np.array_equal(g_cnt.cpu().numpy().view(np.uint32), o_c)
    assert np.allclose(g_sc.cpu().numpy(), 1.0 - o_d / 64.0)
    # the one-call form, straight on the C entry point, NULL scores / keys
    lib = _lib.load()
    d_q = _dev(torch, q.view(np.int64))
    o1 = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c1 = torch.empty((nq,), dtype=torch.int32, device="cuda")
    _lib.check(lib.ucfp_index_search_sharded_dev(six.local.handle, six.comm.handle, 0, d_q.data_ptr(), nq, k,
                                                 o1.data_ptr(), None, None, c1.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream or None))
    torch.cuda.synchronize()
    assert np.array_equal(o1.cpu().numpy().view(np.uint64), o_ids)
    # k = 0 and an empty batch are no-ops that still hand out tickets
    t = C.c_uint64(0)
    _lib.check(lib.ucfp_index_search_sharded_submit(six.local.handle, six.comm.handle, 0, d_q.data_ptr(), nq, 0,
                                                    o1.data_ptr(), None, None, c1.data_ptr(), None, C.byref(t)))
    _lib.check(lib.ucfp_index_search_sharded_collect(six.comm.handle, t.value, None))
    torch.cuda.synchronize()
    assert t.value > 0 and int(c1.abs().sum().item()) == 0
    six.close()


def test_forced_rccl_one_rank_two_tickets_in_flight(gpu_ctx, oracle, torch_cuda):
    """The library's RCCL branch executed on the one GPU there is: ucfp_shard_unique_id -> ncclCommInitRank(nranks = 1)
    -> per batch ONE ncclAllGather of the packed entries on the exchange stream -> topk_merge_u32<PACKED> over the
    gathered buffer.  Two batches are submitted before either is collected (both buffer sets, both scan streams)."""
    torch = torch_cuda
    from ucfp_amd import index, sharded
    rng = np.random.default_rng(808)
    n, nq, k = 700_000, 520, 10
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(3)
    qa = codes[:nq] ^ np.uint64(0b1011)
    qb = codes[-nq:] ^ (np.uint64(1) << np.uint64(40))
    six = sharded.ShardedIndex(index.HAMMING64, ctx=gpu_ctx, force_rccl=True)
    assert six.world == 1 and six.comm.uses_rccl and six.comm.exchanges() == 0
    six.append_local(_dev(torch, ids.view(np.int64)), _dev(torch, codes.view(np.int64)))
    d_qa, d_qb = _dev(torch, qa.view(np.int64)), _dev(torch, qb.view(np.int64))
    ta = six.submit(d_qa, k)
    tb = six.submit(d_qb, k)                      # in flight together with ta
    ra = [t.clone() for t in six.collect(ta)]
    rb = [t.clone() for t in six.collect(tb)]
    torch.cuda.synchronize()
    assert six.comm.exchanges() == 2
    for (g_ids, g_sc, g_keys, g_cnt), q in ((ra, qa), (rb, qb)):
        o_ids, o_d, o_c = oracle.hamming_topk(ids, codes, q, k)
        assert np.array_equal(g_ids.cpu().numpy().view(np.uint64), o_ids)
        assert np.array_equal(g_keys.cpu().numpy().view(np.uint32), o_d)
        assert np.array_equal(g_cnt.cpu().numpy().view(np.uint32), o_c)
        assert np.allclose(g_sc.cpu().numpy(), 1.0 - o_d / 64.0)
    # a third and fourth batch reuse the two buffer sets behind their previous exchanges
    for rep in range(2):
        g = six.search(d_qa if rep else d_qb, 5)
        torch.cuda.synchronize()
        o_ids, _, _ = oracle.hamming_topk(ids, codes, qa if rep else qb, 5)
        assert np.array_equal(g[0].cpu().numpy().view(np.uint64), o_ids)
    assert six.comm.exchanges() == 4
    six.close()


def test_forced_rccl_cosine_and_env_switch(gpu_ctx, oracle, torch_cuda, monkeypatch):
    """Same branch for the cosine kind, switched on through the environment (what a deployment would set to rehearse
    the multi-GPU path on one device) with the plain ucfp_shard_comm_create entry."""
    torch = torch_cuda
    from ucfp_amd import _lib, index
    lib = _lib.load()
    rng = np.random.default_rng(909)
    n, nq, k, dim = 30_000, 19, 6, 64
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    ids = rng.permutation(n).astype(np.uint64)
    monkeypatch.setenv("UCFP_SHARD_FORCE_RCCL", "1")
    uid = (C.c_uint8 * 128)()
    _lib.check(lib.ucfp_shard_unique_id(uid))
    assert any(bytes(uid))
    comm = C.c_void_p()
    _lib.check(lib.ucfp_shard_comm_create(gpu_ctx.handle, uid, 0, 1, C.byref(comm)))
    monkeypatch.delenv("UCFP_SHARD_FORCE_RCCL")
    assert lib.ucfp_shard_comm_uses_rccl(comm) == 1
    # without a uid a forced communicator is refused, not silently downgraded
    bad = C.c_void_p()
    assert lib.ucfp_shard_comm_create_ex(gpu_ctx.handle, None, 0, 1, 1, C.byref(bad)) != 0
    ix = index.DeviceIndex(index.COSINE_F32, dim, ctx=gpu_ctx)
    ix.upsert(0, ids, rows)
    d_q = _dev(torch, q)
    o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_sc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    _lib.check(lib.ucfp_index_search_sharded_dev(ix.handle, comm, 0, d_q.data_ptr(), nq, k, o_ids.data_ptr(),
                                                 o_sc.data_ptr(), None, o_cnt.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream or None))
    torch.cuda.synchronize()
    ex = C.c_uint64(0)
    _lib.check(lib.ucfp_shard_comm_info(comm, None, None, C.byref(ex)))
    assert ex.value == 1
    got, sc = o_ids.cpu().numpy().view(np.uint64), o_sc.cpu().numpy()
    for qi in range(nq):
        e_ids, e_sc = oracle.cosine_knn(ids, rows, q[qi], k)
        assert np.array_equal(got[qi], e_ids), qi
        assert np.abs(sc[qi] - e_sc).max() <= 1e-5
    lib.ucfp_shard_comm_destroy(comm)
    ix.close()


@pytest.mark.parametrize("kind_name,parts", [("hamming", 2), ("hamming", 8), ("cosine", 3)])
def test_multi_shard_wire_format_and_merge(gpu_ctx, oracle, torch_cuda, kind_name, parts):
    """The corpus is split with ucfp_shard_range into `parts` shard indexes (all on this device); each shard's answer
    is packed, the packed lists are laid out [parts][nq][k] exactly as the all-gather delivers them, and
    ucfp_topk_merge_packed_dev must give the oracle's answer over the whole corpus."""
    torch = torch_cuda
    from ucfp_amd import _lib, index, sharded
    lib = _lib.load()
    rng = np.random.default_rng(99 + parts)
    st = torch.cuda.current_stream().cuda_stream or None
    if kind_name == "hamming":
        n, nq, k, kind, dim = 40_000, 130, 10, index.HAMMING64, 0
        rows = rng.integers(0, 2**64, n, dtype=np.uint64)
        rows[n // 2:n // 2 + 40] = rows[:40]            # cross-shard ties: the same code in two shards
        q = rows[:nq] ^ np.uint64(0b11)
        d_q = _dev(torch, q.view(np.int64))
    else:
        n, nq, k, kind, dim = 9_000, 37, 7, index.COSINE_F32, 48
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        d_q = _dev(torch, q)
    ids = rng.permutation(n).astype(np.uint64)
    entries = torch.empty((parts, nq, k, 2), dtype=torch.int64, device="cuda")
    shards = []
    for r in range(parts):
        s, e = sharded.shard_range(n, r, parts)
        ix = index.DeviceIndex(kind, dim, ctx=gpu_ctx)
        ix.upsert(0, ids[s:e], rows[s:e])
        l_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        l_keys = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        l_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
        ix.search_dev(0, d_q.data_ptr(), nq, k, l_ids.data_ptr(), 0, l_keys.data_ptr(), l_cnt.data_ptr(), st or 0)
        _lib.check(lib.ucfp_topk_pack_dev(gpu_ctx.handle, l_ids.data_ptr(), l_keys.data_ptr(), nq, k,
                                          entries[r].data_ptr(), st))
        torch.cuda.synchronize()
        # the device packer and the host statement of the wire format agree byte for byte
        host = sharded.pack_entries(l_ids.cpu().numpy().view(np.uint64), l_keys.cpu().numpy().view(np.uint32))
        assert np.array_equal(entries[r].cpu().numpy(), host)
        shards.append(ix)
    o_ids_t = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_keys_t = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    o_sc_t = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    o_cnt_t = torch.empty((nq,), dtype=torch.int32, device="cuda")
    _lib.check(lib.ucfp_topk_merge_packed_dev(gpu_ctx.handle, kind, entries.data_ptr(), parts, nq, k, o_ids_t.data_ptr(),
                                              o_sc_t.data_ptr(), o_keys_t.data_ptr(), o_cnt_t.data_ptr(), st))
    torch.cuda.synchronize()
    got_ids = o_ids_t.cpu().numpy().view(np.uint64)
    if kind_name == "hamming":
        o_ids, o_d, o_c = oracle.hamming_topk(ids, rows, q, k)
        assert np.array_equal(got_ids, o_ids)
        assert np.array_equal(o_keys_t.cpu().numpy().view(np.uint32), o_d)
        assert np.array_equal(o_cnt_t.cpu().numpy().view(np.uint32), o_c)
    else:
        sc = o_sc_t.cpu().numpy()
        for qi in range(nq):
            e_ids, e_sc = oracle.cosine_knn(ids, rows, q[qi], k)
            assert np.array_equal(got_ids[qi], e_ids), qi
            assert np.abs(sc[qi] - e_sc).max() <= 1e-5          # north_star tolerance for float distances
    for ix in shards:
        ix.close()


def test_generic_geometry_batches_on_two_streams_do_not_share_scratch(gpu_ctx, oracle, torch_cuda):
    """Two generic-geometry batches (they normalise into the context's ONE scratch area) enqueued back to back on
    two streams, plus a host-pointer call from a second thread at the same time: all equal the oracle."""
    torch = torch_cuda
    from ucfp_amd import image
    rng = np.random.default_rng(5150)
    a = rng.integers(0, 256, (96, 300, 420), dtype=np.uint8)           # GRAY8 420x300: streaming normaliser
    b = rng.integers(0, 256, (96, 230, 340, 3), dtype=np.uint8)        # RGB8 340x230
    c = rng.integers(0, 256, (64, 200, 301), dtype=np.uint8)           # width % 4 != 0: gather fallback
    ref_a, _ = oracle.image_hash_batch(a, 7, pixfmt=0)
    ref_b, _ = oracle.image_hash_batch(b, 7, pixfmt=1)
    ref_c, _ = oracle.image_hash_batch(c, 7, pixfmt=0)
    d_a, d_b = _dev(torch, a), _dev(torch, b)
    o_a = torch.zeros((a.shape[0], 536), dtype=torch.uint8, device="cuda")
    o_b = torch.zeros((b.shape[0], 536), dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    host_out = {}

    def host_call():
        host_out["c"] = image.fingerprint_frames(c, algo=7, pixfmt=0, ctx=gpu_ctx)[0]
    th = threading.Thread(target=host_call)
    th.start()
    for rep in range(6):      # interleave many launches so that an unordered scratch area WOULD be overwritten
        image.fingerprint_frames_dev(d_a.data_ptr(), a.shape[0], 420, 300, algo=7, pixfmt=0, out_ptr=o_a.data_ptr(),
                                     stream=s1.cuda_stream, ctx=gpu_ctx)
        image.fingerprint_frames_dev(d_b.data_ptr(), b.shape[0], 340, 230, algo=7, pixfmt=1, out_ptr=o_b.data_ptr(),
                                     stream=s2.cuda_stream, ctx=gpu_ctx)
    th.join()
    torch.cuda.synchronize()
    assert np.array_equal(o_a.cpu().numpy(), ref_a)
    assert np.array_equal(o_b.cpu().numpy(), ref_b)
    assert np.array_equal(host_out["c"], ref_c)


def test_upsert_replaces_stale_rows_like_the_reference(gpu_ctx):
    """EmbeddedBackend::upsert keys everything by (tenant, record_id): re-ingesting without an embedding drops the
    stale vector (embedded/mod.rs:184-191); a new dimension / algorithm replaces the old row."""
    from ucfp_amd import index
    from ucfp_amd.core import Modality, Record

    def rec(rid, emb=None, algo="test", fp=b"fp"):
        return Record(1, rid, Modality.Image, 1, algo, 0, fp, embedding=emb, model_id="m")

    db = index.GpuIndex(ctx=gpu_ctx)
    db.upsert([rec(1, [1.0, 0.0, 0.0]), rec(2, [0.0, 1.0, 0.0])])
    assert [h.record_id for h in db.knn(1, [1.0, 0.1, 0.0], 5)] == [1, 2]
    db.upsert([rec(1, None)])                                  # "Drop any stale vector for this key"
    assert [h.record_id for h in db.knn(1, [1.0, 0.1, 0.0], 5)] == [2]
    db.upsert([rec(2, [0.0, 1.0])])                            # new dimension replaces the 3-d row
    assert db.knn(1, [1.0, 0.1, 0.0], 5) == []
    assert [h.record_id for h in db.knn(1, [0.0, 1.0], 5)] == [2]
    # hash spaces: a pHash record re-ingested as aHash leaves the pHash space
    fp = bytearray(168)
    fp[32:40] = (0xABCDEF).to_bytes(8, "little")
    db.upsert([rec(7, None, "imgfprint-phash-v1", bytes(fp))])
    assert [h.record_id for h in db.hamming(1, "imgfprint-phash-v1", 0xABCDEF, 3)] == [7]
    db.upsert([rec(7, None, "imgfprint-ahash-v1", bytes(fp))])
    assert db.hamming(1, "imgfprint-phash-v1", 0xABCDEF, 3) == []
    assert [h.record_id for h in db.hamming(1, "imgfprint-ahash-v1", 0xABCDEF, 3)] == [7]
    # within one batch the last record of a key wins
    db.upsert([rec(9, [1.0, 0.0]), rec(9, None)])
    assert 9 not in [h.record_id for h in db.knn(1, [1.0, 0.0], 5)]


def test_append_on_one_stream_search_on_another(gpu_ctx, oracle, torch_cuda):
    torch = torch_cuda
    from ucfp_amd import index
    rng = np.random.default_rng(31337)
    n, nq, k = 400_000, 64, 10
    codes = rng.integers(0, 2**64, n, dtype=np.uint64)
    ids = np.arange(n, dtype=np.uint64)
    q = codes[-nq:] ^ np.uint64(1)                              # neighbours live in the LAST rows appended
    ix = index.DeviceIndex(index.HAMMING64, 0, index.APPEND_ONLY, gpu_ctx)
    d_ids, d_codes, d_q = _dev(torch, ids.view(np.int64)), _dev(torch, codes.view(np.int64)), _dev(torch, q.view(np.int64))
    o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_keys = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    half = n // 2
    ix.append_dev(0, d_ids.data_ptr(), d_codes.data_ptr(), half, sa.cuda_stream)
    ix.append_dev(0, d_ids.data_ptr() + 8 * half, d_codes.data_ptr() + 8 * half, n - half, sa.cuda_stream)
    ix.search_dev(0, d_q.data_ptr(), nq, k, o_ids.data_ptr(), 0, o_keys.data_ptr(), o_cnt.data_ptr(), sb.cuda_stream)
    torch.cuda.synchronize()
    e_ids, e_d, _ = oracle.hamming_topk(ids, codes, q, k)
    assert np.array_equal(o_ids.cpu().numpy().view(np.uint64), e_ids)
    assert np.array_equal(o_keys.cpu().numpy().view(np.uint32), e_d)
    ix.close()


def test_a_missing_shard_is_marked_on_every_rank(gpu_ctx, oracle, torch_cuda):
    """A rank whose scan failed joins the all-gather with 0xff bytes: an empty list whose pad word says 'missing'.  The
    merge every rank runs reports it as a bit mask (ucfp_topk_merge_packed_ex_dev for hosts with their own transport,
    ucfp_index_search_sharded_missing for the RCCL path), and the answer is the merge of the shards that did answer."""
    torch = torch_cuda
    from ucfp_amd import _lib, errors, index, sharded
    lib = _lib.load()
    rng = np.random.default_rng(4242)
    nq, k, parts = 33, 7, 3
    ids = rng.permutation(3000).astype(np.uint64).reshape(parts, 1000)
    codes = rng.integers(0, 2**64, (parts, 1000), dtype=np.uint64)
    q = rng.integers(0, 2**64, nq, dtype=np.uint64)
    ent = np.zeros((parts, nq, k, 2), np.int64)
    for p in range(parts):
        o_ids, o_d, _ = oracle.hamming_topk(ids[p], codes[p], q, k)
        ent[p] = sharded.pack_entries(o_ids, o_d)
    ent[1] = -1                                                   # shard 1 could not scan: 0xff in every byte
    d_ent = _dev(torch, ent)
    o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_keys = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    mask = torch.zeros((1,), dtype=torch.int64, device="cuda")
    _lib.check(lib.ucfp_topk_merge_packed_ex_dev(gpu_ctx.handle, index.HAMMING64, d_ent.data_ptr(), parts, nq, k,
                                                 o_ids.data_ptr(), None, o_keys.data_ptr(), o_cnt.data_ptr(), mask.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream or None))
    torch.cuda.synchronize()
    assert int(mask.item()) == 0b010
    keep = [0, 2]
    w_ids, w_d, _ = oracle.hamming_topk(ids[keep].reshape(-1), codes[keep].reshape(-1), q, k)
    assert np.array_equal(o_ids.cpu().numpy().view(np.uint64), w_ids)
    assert np.array_equal(o_keys.cpu().numpy().view(np.uint32), w_d)
    # the RCCL path: nothing missing after ordinary searches; a ticket whose buffers moved on is refused
    six = sharded.ShardedIndex(index.HAMMING64, ctx=gpu_ctx, force_rccl=True)
    six.append_local(_dev(torch, ids.reshape(-1).view(np.int64)), _dev(torch, codes.reshape(-1).view(np.int64)))
    d_q = _dev(torch, q.view(np.int64))
    t1 = six.submit(d_q, k)
    assert six.missing_shards(t1) == 0
    six.search(d_q, k, check=True)
    t3 = six.submit(d_q, k)                                        # the buffer set of t1 again
    assert six.missing_shards(t3) == 0
    six._bufs[(nq, k, 0)]["ticket"], keep_t = 1, six._bufs[(nq, k, 0)]["ticket"]      # ticket 1: long overwritten
    with pytest.raises(errors.InvalidArgument):
        six.missing_shards((nq, k, 0))
    six._bufs[(nq, k, 0)]["ticket"] = keep_t
    six.close()
