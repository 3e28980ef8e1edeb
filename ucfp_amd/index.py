"""kNN index -- host-side mirror of `trait IndexBackend` (src/index/mod.rs:17-78) for the vector
path, backed by the GPU-resident shard behind the C ABI (ucfp_index_*).

    GpuIndex.upsert(records)          IndexBackend::upsert   src/index/mod.rs:20-22
    GpuIndex.delete(tenant, ids)      IndexBackend::delete   :24-27
    GpuIndex.knn(tenant, query, k)    IndexBackend::knn      :29-35  (cosine over Record.embedding,
                                      as EmbeddedBackend::knn src/index/embedded/mod.rs:268-360)
    GpuIndex.hamming(tenant, h, k)    the new Hamming search behind /v1/query
    GpuIndex.flush()                  IndexBackend::flush    :63

The reference keeps redb as the source of truth; this object is the device mirror of one shard
(one process per GPU).  `ShardedIndex` in ucfp_amd/sharded.py spreads a corpus over the ranks of a
node and merges per-shard top-k after an RCCL all-gather.
"""
import ctypes as C
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _lib
from .core import Hit, HitSource, Record
from .errors import InvalidArgument

HAMMING64, COSINE_F32 = 1, 2
APPEND_ONLY = 1
MAX_K = 128
INVALID_ID = 0xFFFFFFFFFFFFFFFF


class SearchBatcher:
    """Host micro-batcher for the query route (ucfp_index_search_batcher_*): request threads `submit` ONE query each (their
    own k); the library coalesces them into one search launch per flush.  /v1/query is one query per request
    (src/server/handlers.rs:143-187), up to 512 in flight (src/bin/ucfp.rs:267)."""

    def __init__(self, index: "DeviceIndex", tenant: int = 0, *, max_batch: int = 256, max_delay_us: int = 0):
        self._lib = _lib.load()
        self.index = index
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_index_search_batcher_create(index.handle, tenant, max_batch, max_delay_us, C.byref(h)))
        self.handle = h

    def submit(self, query, k: int):
        """query: int (64-bit hash) or float32 [dim].  -> (ids u64 [n], scores f32 [n], dist u32 [n]) with n <= k hits."""
        if self.index.kind == HAMMING64:
            q = np.array([query], dtype=np.uint64)
        else:
            q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.index.dim)
        kk = max(int(k), 1)
        ids = np.empty(kk, np.uint64)
        sc = np.empty(kk, np.float32)
        d = np.empty(kk, np.uint32)
        cnt = C.c_uint32(0)
        _lib.check(self._lib.ucfp_index_search_batcher_submit(self.handle, q.ctypes.data, int(k), ids.ctypes.data, sc.ctypes.data,
                                                              d.ctypes.data, C.byref(cnt)))
        n = int(cnt.value)
        return ids[:n], sc[:n], d[:n]

    def stats(self):
        b, i = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.ucfp_index_search_batcher_stats(self.handle, C.byref(b), C.byref(i)))
        return int(b.value), int(i.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_index_search_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceIndex:
    """Thin RAII wrapper over one ucfp_index handle (one kind, one dim)."""

    def __init__(self, kind: int, dim: int = 0, flags: int = 0, ctx=None):
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.kind, self.dim, self.flags = kind, dim, flags
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_index_create(self.ctx.handle, kind, dim, flags, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- mutation ----
    def _rows(self, rows):
        if self.kind == HAMMING64:
            a = np.ascontiguousarray(rows, dtype=np.uint64).reshape(-1)
        else:
            a = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.dim)
        return a

    def upsert(self, tenant: int, ids, rows) -> None:
        ids = np.ascontiguousarray(ids, dtype=np.uint64).reshape(-1)
        rows = self._rows(rows)
        if rows.shape[0] != ids.shape[0]:
            raise InvalidArgument("ids and rows disagree on the number of records")
        _lib.check(self._lib.ucfp_index_upsert(self.handle, tenant, ids.ctypes.data, rows.ctypes.data,
                                               ids.shape[0]))

    def save(self, path: str) -> None:
        """Snapshot every tenant's ids + rows to a flat file (the device mirror's checkpoint)."""
        _lib.check(self._lib.ucfp_index_save(self.handle, str(path).encode()))

    def load(self, path: str) -> None:
        """Upsert a snapshot written by `save` (same kind / dim)."""
        _lib.check(self._lib.ucfp_index_load(self.handle, str(path).encode()))

    def append_dev(self, tenant: int, ids_ptr: int, rows_ptr: int, n: int, stream: int = 0) -> None:
        _lib.check(self._lib.ucfp_index_append_dev(self.handle, tenant, ids_ptr, rows_ptr, n, stream or None))

    def delete(self, tenant: int, ids) -> int:
        ids = np.ascontiguousarray(ids, dtype=np.uint64).reshape(-1)
        removed = C.c_size_t(0)
        _lib.check(self._lib.ucfp_index_delete(self.handle, tenant, ids.ctypes.data, ids.shape[0],
                                               C.byref(removed)))
        return int(removed.value)

    def size(self, tenant: int) -> int:
        out = C.c_size_t(0)
        _lib.check(self._lib.ucfp_index_size(self.handle, tenant, C.byref(out)))
        return int(out.value)

    def flush(self) -> None:
        _lib.check(self._lib.ucfp_index_flush(self.handle))

    # ---- search ----
    def search(self, tenant: int, queries, k: int):
        """Host-memory batch search. Returns (ids [nq,k] u64, scores [nq,k] f32, keys [nq,k] u32,
        counts [nq] u32); keys = Hamming distance, or the inverted order image of the cosine score."""
        q = self._rows(queries)
        nq = q.shape[0]
        kk = max(k, 1)
        ids = np.full((nq, kk), INVALID_ID, np.uint64)
        scores = np.zeros((nq, kk), np.float32)
        keys = np.full((nq, kk), 0xFFFFFFFF, np.uint32)
        counts = np.zeros(nq, np.uint32)
        _lib.check(self._lib.ucfp_index_search(self.handle, tenant, q.ctypes.data, nq, k, ids.ctypes.data,
                                               scores.ctypes.data, keys.ctypes.data, counts.ctypes.data))
        return ids[:, :k], scores[:, :k], keys[:, :k], counts

    def search_dev(self, tenant: int, queries_ptr: int, nq: int, k: int, out_ids_ptr: int,
                   out_scores_ptr: int, out_keys_ptr: int, out_counts_ptr: int, stream: int = 0) -> None:
        _lib.check(self._lib.ucfp_index_search_dev(self.handle, tenant, queries_ptr, nq, k, out_ids_ptr,
                                                   out_scores_ptr or None, out_keys_ptr or None,
                                                   out_counts_ptr, stream or None))


def topk_merge_dev(kind: int, part_ids_ptr: int, part_keys_ptr: int, parts: int, nq: int, k: int,
                   out_ids_ptr: int, out_scores_ptr: int, out_keys_ptr: int, out_counts_ptr: int,
                   stream: int = 0, ctx=None) -> None:
    ctx = ctx or _lib.current_context()
    _lib.check(_lib.load().ucfp_topk_merge_dev(ctx.handle, kind, part_ids_ptr, part_keys_ptr, parts, nq, k,
                                               out_ids_ptr, out_scores_ptr or None, out_keys_ptr,
                                               out_counts_ptr, stream or None))


class GpuIndex:
    """IndexBackend-shaped facade: cosine kNN over `Record.embedding` plus Hamming search over
    64-bit hashes pulled out of `Record.fingerprint`.  Embedding indexes are keyed by dimension,
    like the reference skips rows whose stored length differs from the query's
    (src/index/embedded/mod.rs:307-309)."""

    def __init__(self, ctx=None, sidecar=None):
        self.ctx = ctx or _lib.current_context()
        self._cos = {}        # dim -> DeviceIndex
        self._ham = {}        # hash space name -> DeviceIndex
        self._sidecar = sidecar   # ucfp_amd.store.Sidecar: the stored-table mirror written at upsert

    def attach_sidecar(self, sidecar) -> None:
        self._sidecar = sidecar

    def _cosine(self, dim: int) -> DeviceIndex:
        ix = self._cos.get(dim)
        if ix is None:
            ix = self._cos[dim] = DeviceIndex(COSINE_F32, dim, 0, self.ctx)
        return ix

    def _hamming(self, space: str) -> DeviceIndex:
        ix = self._ham.get(space)
        if ix is None:
            ix = self._ham[space] = DeviceIndex(HAMMING64, 0, 0, self.ctx)
        return ix

    def upsert(self, records: Sequence[Record]) -> None:
        """Embeddings go to the cosine index of their dimension; image records also feed the
        Hamming spaces `<algorithm>` with their 64-bit global hashes (SURVEY 8f N2 offsets).

        Overwrite semantics are the reference's: everything is keyed by (tenant_id, record_id), a re-ingested record
        REPLACES the old one -- "Drop any stale vector for this key" when the new record has no embedding
        (src/index/embedded/mod.rs:184-191), a new dimension or algorithm replaces the old row.  So before inserting,
        the key is removed from every cosine index of another dimension and every hash space the new record does not
        feed.  Within one batch the last record of a key wins, as successive `insert`s in one redb transaction do."""
        if self._sidecar is not None:     # the log first (the host does this right after its redb commit), then the mirror
            self._sidecar.append(records)
        last = {}
        for r in records:
            last[(r.tenant_id, r.record_id)] = r
        by_cos, by_ham, stale_cos, stale_ham = {}, {}, {}, {}
        for r in last.values():
            dim = len(r.embedding) if r.embedding is not None else 0
            if dim > 0:
                by_cos.setdefault((r.tenant_id, dim), []).append(r)
            for d in self._cos:
                if d != dim:
                    stale_cos.setdefault((r.tenant_id, d), []).append(r.record_id)
            fed = set()
            for space, h in _hash_spaces(r):
                by_ham.setdefault((r.tenant_id, space), []).append((r.record_id, h))
                fed.add(space)
            for space in self._ham:
                if space not in fed:
                    stale_ham.setdefault((r.tenant_id, space), []).append(r.record_id)
        for (tenant, d), ids in stale_cos.items():
            self._cos[d].delete(tenant, np.array(ids, np.uint64))
        for (tenant, space), ids in stale_ham.items():
            self._ham[space].delete(tenant, np.array(ids, np.uint64))
        for (tenant, dim), recs in by_cos.items():
            ids = np.array([r.record_id for r in recs], np.uint64)
            rows = np.array([r.embedding for r in recs], np.float32)
            self._cosine(dim).upsert(tenant, ids, rows)
        for (tenant, space), items in by_ham.items():
            ids = np.array([i for i, _ in items], np.uint64)
            rows = np.array([h for _, h in items], np.uint64)
            self._hamming(space).upsert(tenant, ids, rows)

    def delete(self, tenant_id: int, record_ids: Iterable[int]) -> None:
        ids = np.array(list(record_ids), np.uint64)
        if self._sidecar is not None:
            self._sidecar.delete(tenant_id, ids.tolist())
        for ix in list(self._cos.values()) + list(self._ham.values()):
            ix.delete(tenant_id, ids)

    def knn(self, tenant_id: int, query: Sequence[float], k: int, _filter: Optional[bytes] = None) -> List[Hit]:
        """EmbeddedBackend::knn: empty query or k == 0 -> [] (src/index/embedded/mod.rs:275-277)."""
        q = np.asarray(query, np.float32).reshape(-1)
        if q.size == 0 or k == 0:
            return []
        ix = self._cos.get(q.size)
        if ix is None:
            return []
        ids, scores, _, counts = ix.search(tenant_id, q[None, :], min(k, MAX_K))
        return [Hit(tenant_id=tenant_id, record_id=int(ids[0, i]), score=float(scores[0, i]),
                    source=HitSource.Vector) for i in range(int(counts[0]))]

    def hamming(self, tenant_id: int, space: str, query_hash: int, k: int) -> List[Hit]:
        if k == 0 or space not in self._ham:
            return []
        ids, scores, dist, counts = self._ham[space].search(
            tenant_id, np.array([query_hash], np.uint64), min(k, MAX_K))
        return [Hit(tenant_id=tenant_id, record_id=int(ids[0, i]), score=float(scores[0, i]),
                    source=HitSource.Hamming, distance=int(dist[0, i])) for i in range(int(counts[0]))]

    def query(self, req) -> List[Hit]:
        """POST /v1/query (handlers.rs:143-187) with the additive `hash` field: a vector goes to the cosine kNN,
        a hash to the Hamming space `algorithm` (default: the only hash space present)."""
        if req.hash is not None:
            space = req.algorithm
            if space is None:
                if len(self._ham) != 1:
                    raise InvalidArgument("`algorithm` is required when several hash spaces exist")
                space = next(iter(self._ham))
            hits = self.hamming(req.tenant_id, space, req.hash, req.k)
        else:
            hits = self.knn(req.tenant_id, req.vector or [], req.k)
        for rank, h in enumerate(hits):   # Matcher::search fills the rank of the only list it fused (matcher/mod.rs:140-207)
            if h.source == HitSource.Vector:
                h.vector_score, h.vector_rank = h.score, rank + 1
        return hits

    def flush(self) -> None:
        if self._sidecar is not None:
            self._sidecar.sync()
        for ix in list(self._cos.values()) + list(self._ham.values()):
            ix.flush()


def _hash_spaces(r: Record):
    """(space, u64) pairs a record contributes to Hamming search."""
    fp = r.fingerprint
    rd = lambda off: int.from_bytes(fp[off:off + 8], "little")  # noqa: E731
    if r.algorithm == "imgfprint-multihash-v1" and len(fp) == 536:
        return [("imgfprint-ahash-v1", rd(64)), ("imgfprint-phash-v1", rd(232)), ("imgfprint-dhash-v1", rd(400))]
    if r.algorithm in ("imgfprint-ahash-v1", "imgfprint-phash-v1", "imgfprint-dhash-v1") and len(fp) == 168:
        return [(r.algorithm, rd(32))]
    if r.algorithm in ("simhash-b64-tf", "simhash-b64-idf") and len(fp) == 8:
        return [(r.algorithm, rd(0))]
    return []
