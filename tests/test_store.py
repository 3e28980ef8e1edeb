"""The sidecar log of the stored tables (ucfp_sidecar_* / ucfp_amd/store.py).  The log calls are pure host
code, so replay semantics are checked here without a GPU; the rebuild of the device shards is the gpu-marked test."""
import json
import os

import numpy as np
import pytest

from ucfp_amd.core import Modality, Record


def _img_record(rng, tenant, rid, algorithm="imgfprint-multihash-v1", emb=None):
    n = 536 if algorithm == "imgfprint-multihash-v1" else 168
    return Record(tenant_id=tenant, record_id=rid, modality=Modality.Image, format_version=1, algorithm=algorithm,
                  config_hash=0, fingerprint=rng.integers(0, 256, n, dtype=np.uint8).tobytes(), embedding=emb)


def test_catalog_json_is_the_references_row():
    """serde_json of CatalogEntry (src/index/embedded/mod.rs:93-116, :193-204): same keys, same order, compact."""
    from ucfp_amd import store
    r = Record(tenant_id=1, record_id=2, modality=Modality.Text, format_version=1, algorithm="minhash-h128", config_hash=7,
               fingerprint=b"x" * 1032, embedding=[0.5, 1.0], model_id="m", metadata=b"abc")
    js = store.catalog_json(r)
    assert js == (b'{"modality":2,"format_version":1,"config_hash":7,"fingerprint_len":1032,"embedding_dim":2,'
                  b'"algorithm":"minhash-h128","model_id":"m","metadata_len":3}')
    r.model_id = None
    assert b'"model_id":null' in store.catalog_json(r)


def test_replay_applies_overwrites_and_deletes_in_order(tmp_path):
    from ucfp_amd import store
    rng = np.random.default_rng(1)
    path = str(tmp_path / "shard.sidecar")
    sc = store.Sidecar(path)
    a = _img_record(rng, 7, 10)
    b = _img_record(rng, 7, 11, "imgfprint-phash-v1", emb=[1.0, 2.0, 3.0])
    c = _img_record(rng, 3, 10)                          # same record id, another tenant
    sc.append([a, b, c])
    a2 = _img_record(rng, 7, 10, "imgfprint-dhash-v1")   # re-ingest under another algorithm: replaces the row
    sc.append([a2])
    sc.delete(7, [11])
    sc.delete(7, [999])                                  # deleting what is not there is not an error (mod.rs:229-266)
    b2 = _img_record(rng, 7, 11, "imgfprint-phash-v1")   # comes back without an embedding: "drop any stale vector"
    sc.append([b2])
    sc.sync()
    sc.close()
    snap = store.Snapshot(path)
    assert (snap.live_rows, snap.log_entries, snap.torn_bytes) == (3, 7, 0)
    rows = list(snap)
    assert [(r.tenant_id, r.record_id) for r in rows] == [(3, 10), (7, 10), (7, 11)]     # redb range-scan order
    assert rows[0].fingerprint == c.fingerprint and rows[0].algorithm == "imgfprint-multihash-v1"
    assert rows[1].fingerprint == a2.fingerprint and rows[1].algorithm == "imgfprint-dhash-v1"
    assert rows[2].fingerprint == b2.fingerprint and rows[2].embedding is None
    t, ids, blobs = snap.gather_fingerprints("imgfprint-multihash-v1", 536)
    assert list(t) == [3] and list(ids) == [10] and blobs[0].tobytes() == c.fingerprint
    assert snap.gather_fingerprints("imgfprint-multihash-v1", 168)[1].size == 0
    assert snap.dims() == []
    snap.close()


def test_torn_tail_is_reported_and_cut_on_reopen(tmp_path):
    from ucfp_amd import store
    rng = np.random.default_rng(2)
    path = str(tmp_path / "shard.sidecar")
    sc = store.Sidecar(path)
    recs = [_img_record(rng, 1, i, emb=rng.standard_normal(8).astype(np.float32).tolist()) for i in range(20)]
    sc.append(recs)
    sc.close()
    whole = os.path.getsize(path)
    with open(path, "r+b") as f:                         # a crash in the middle of the last append
        f.truncate(whole - 100)
    snap = store.Snapshot(path)
    assert snap.live_rows == 19 and snap.torn_bytes > 0
    t, ids, rows = snap.gather_vectors(8)
    assert list(ids) == list(range(19)) and np.array_equal(rows[5], np.float32(recs[5].embedding))
    snap.close()
    with open(path, "r+b") as f:                         # a flipped bit in the middle of the log: everything after is dropped
        f.seek(whole // 2)
        byte = f.read(1)
        f.seek(whole // 2)
        f.write(bytes([byte[0] ^ 0x10]))
    snap = store.Snapshot(path)
    assert 0 < snap.live_rows < 19
    keep = snap.live_rows
    snap.close()
    sc = store.Sidecar(path)                             # reopen for writing: the bad tail is cut, new rows follow the good ones
    sc.append([_img_record(rng, 1, 500)])
    sc.close()
    snap = store.Snapshot(path)
    assert snap.live_rows == keep + 1 and snap.torn_bytes == 0 and list(snap)[-1].record_id == 500
    snap.close()
    with pytest.raises(Exception):
        store.Snapshot(str(tmp_path / "missing"))
    (tmp_path / "junk").write_bytes(b"not a log at all")
    with pytest.raises(Exception):
        store.Sidecar(str(tmp_path / "junk"))


def test_one_writer_per_log_and_snapshots_beside_it(tmp_path):
    """a second opener must not cut the file under a live writer -- the log takes an exclusive (OFD) lock
    for the writer's life; snapshots open beside a live writer and see a prefix of what it has appended."""
    import threading
    from ucfp_amd import store
    rng = np.random.default_rng(3)
    path = str(tmp_path / "w.sidecar")
    sc = store.Sidecar(path)
    sc.append([_img_record(rng, 1, i) for i in range(10)])
    with pytest.raises(Exception, match="open for writing elsewhere"):
        store.Sidecar(path)
    errs = []

    def second():                                        # same process, another thread: OFD locks still exclude it
        try:
            store.Sidecar(path).close()
            errs.append("second writer got in")
        except Exception:
            pass
    th = threading.Thread(target=second)
    th.start()
    th.join()
    assert not errs
    snap = store.Snapshot(path)                          # a reader beside the live writer
    assert snap.live_rows == 10
    sc.append([_img_record(rng, 1, 100 + i) for i in range(5)])
    assert snap.live_rows == 10                          # a snapshot is a snapshot
    snap.close()
    sc.close()
    sc = store.Sidecar(path)                             # the lock went with the descriptor
    sc.append([_img_record(rng, 1, 999)])
    sc.close()
    snap = store.Snapshot(path)
    assert snap.live_rows == 16 and snap.torn_bytes == 0
    snap.close()


@pytest.mark.gpu
def test_rebuild_equals_the_index_that_wrote_the_log(gpu_ctx, tmp_path):
    """Ingest through a GpuIndex with the sidecar attached (overwrites, deletes, several tenants / algorithms /
    dimensions), then rebuild a second index from the log alone: every query answers the same."""
    from ucfp_amd import index, store
    rng = np.random.default_rng(3)
    path = str(tmp_path / "shard.sidecar")
    live = index.GpuIndex(gpu_ctx, sidecar=store.Sidecar(path))
    recs = []
    for i in range(600):
        algo = ("imgfprint-multihash-v1", "imgfprint-phash-v1", "imgfprint-dhash-v1")[i % 3]
        emb = rng.standard_normal(16 if i % 2 else 32).astype(np.float32).tolist() if i % 5 else None
        recs.append(_img_record(rng, 1 + i % 2, 1000 + i, algo, emb))
    recs += [Record(tenant_id=1, record_id=5000 + i, modality=Modality.Text, format_version=1, algorithm="simhash-b64-tf",
                    config_hash=0, fingerprint=rng.integers(0, 256, 8, dtype=np.uint8).tobytes()) for i in range(50)]
    for i in range(0, len(recs), 97):
        live.upsert(recs[i:i + 97])
    live.upsert([_img_record(rng, 1, 1000 + i, "imgfprint-ahash-v1") for i in range(0, 60, 2)])     # replace some
    live.delete(2, [1001 + 2 * i for i in range(40)])
    live.flush()
    again = store.rebuild(path, gpu_ctx)
    assert sorted(again._ham) == sorted(live._ham) and sorted(again._cos) == sorted(live._cos)
    for space in live._ham:
        for tenant in (1, 2):
            assert again._ham[space].size(tenant) == live._ham[space].size(tenant), (space, tenant)
            q = rng.integers(0, 1 << 63, 5, dtype=np.uint64)
            a, b = live._ham[space].search(tenant, q, 10), again._ham[space].search(tenant, q, 10)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (space, tenant)
    for dim in live._cos:
        for tenant in (1, 2):
            q = rng.standard_normal((3, dim)).astype(np.float32)
            a, b = live._cos[dim].search(tenant, q, 10), again._cos[dim].search(tenant, q, 10)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (dim, tenant)
    # keep writing through the rebuilt index: the log stays the record of both.  One writer per log: while `live` holds
    # it a second opener is refused; the hand-over is close, then reopen
    with pytest.raises(Exception, match="open for writing elsewhere"):
        store.rebuild(path, gpu_ctx, sidecar=True)
    live._sidecar.close()
    live.attach_sidecar(None)
    more = store.rebuild(path, gpu_ctx, sidecar=True)
    more.upsert([_img_record(rng, 1, 9999, "imgfprint-phash-v1")])
    more.flush()
    snap = store.Snapshot(path)
    assert any(r.record_id == 9999 for r in snap)
    snap.close()


def test_random_operation_sequences_replay_like_a_dict(tmp_path):
    """3000 random upserts / overwrites / deletes over a few tenants, the writer closed and reopened now and then: the
    snapshot equals a plain dict that applied the same operations, in redb's range-scan order."""
    from ucfp_amd import store
    rng = np.random.default_rng(11)
    path = str(tmp_path / "rand.sidecar")
    model = {}
    sc = store.Sidecar(path)
    algos = ["imgfprint-multihash-v1", "imgfprint-phash-v1", "simhash-b64-tf", "minhash-h128"]
    for step in range(3000):
        tenant, rid = int(rng.integers(0, 4)), int(rng.integers(0, 300))
        op = rng.random()
        if op < 0.7:
            algo = algos[int(rng.integers(0, 4))]
            fp = rng.integers(0, 256, {"imgfprint-multihash-v1": 536, "imgfprint-phash-v1": 168, "simhash-b64-tf": 8,
                                       "minhash-h128": 1032}[algo], dtype=np.uint8).tobytes()
            emb = rng.standard_normal(int(rng.choice([4, 16]))).astype(np.float32).tolist() if rng.random() < 0.5 else None
            r = Record(tenant_id=tenant, record_id=rid, modality=Modality.Image, format_version=1, algorithm=algo,
                       config_hash=int(rng.integers(0, 1 << 62)), fingerprint=fp, embedding=emb)
            sc.append([r])
            model[(tenant, rid)] = r
        else:
            sc.delete(tenant, [rid])
            model.pop((tenant, rid), None)
        if step % 700 == 699:
            sc.close()
            sc = store.Sidecar(path)
    sc.sync()
    sc.close()
    snap = store.Snapshot(path)
    rows = list(snap)
    assert [(r.tenant_id, r.record_id) for r in rows] == sorted(model)
    for r in rows:
        m = model[(r.tenant_id, r.record_id)]
        assert r.fingerprint == m.fingerprint and r.algorithm == m.algorithm and r.config_hash == m.config_hash
        assert (r.embedding is None) == (m.embedding is None)
        if m.embedding is not None:
            assert np.array_equal(np.float32(r.embedding), np.float32(m.embedding))
    for algo, n in (("imgfprint-multihash-v1", 536), ("simhash-b64-tf", 8)):
        t, ids, blobs = snap.gather_fingerprints(algo, n)
        want = sorted(k for k, v in model.items() if v.algorithm == algo)
        assert list(zip(t.tolist(), ids.tolist())) == want
    snap.close()
