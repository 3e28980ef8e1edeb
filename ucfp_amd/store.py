"""Stored tables -> GPU shards -- host-side mirror of what `EmbeddedBackend::open` has to do when a
device mirror sits behind it (src/index/embedded/mod.rs:104-125).

The reference keeps three tables per record in one redb file, all keyed (tenant_id, record_id): the fingerprint blob
(`ucfp/fingerprints/v1`), the embedding (`ucfp/vectors/v1`) and a serde_json catalog row naming modality, algorithm,
format version ... (`ucfp/catalog/v2`; CatalogEntry, mod.rs:93-116).  redb's page format is a third-party crate's, so
the drop-in mirrors exactly those rows into an append-only SIDECAR log (ucfp_sidecar_* in the C ABI, written by the
host right after its redb transaction commits) and rebuilds the shards from it:

    Sidecar(path).append(records) / .delete(tenant, ids) / .sync()      the writer, one call per upsert / delete
    Snapshot(path)                                                      live rows after replay, redb range-scan order
    rebuild(path, ctx) -> GpuIndex                                      start-up: every hash space and cosine index

Hash spaces are keyed by the catalog's `algorithm` tag, so only comparable 64-bit hashes share an index; the 64-bit
global hashes are cut out of the stored 168 / 536-byte records ON THE DEVICE (ucfp_image_record_codes_dev, offsets
32 / 64 / 232 / 400); only ids and 8-byte codes reach the index calls.
"""
import ctypes as C
import json
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np

from . import _lib
from .core import Modality, Record
from .errors import InvalidArgument


def catalog_json(r: Record) -> bytes:
    """serde_json::to_vec(&CatalogEntry) (mod.rs:193-204): field order and spelling of the reference's struct."""
    return json.dumps({
        "modality": int(r.modality), "format_version": int(r.format_version), "config_hash": int(r.config_hash),
        "fingerprint_len": len(r.fingerprint), "embedding_dim": len(r.embedding) if r.embedding is not None else 0,
        "algorithm": r.algorithm, "model_id": r.model_id, "metadata_len": len(r.metadata)},
        separators=(",", ":")).encode()


class Sidecar:
    """Writer handle on the log: the host calls `append` where it commits `IndexBackend::upsert`."""

    def __init__(self, path: str):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_sidecar_open(str(path).encode(), C.byref(h)))
        self.handle, self.path = h, str(path)

    def append(self, records: Sequence[Record]) -> None:
        for r in records:
            emb = np.ascontiguousarray(r.embedding, dtype=np.float32) if r.embedding is not None else None
            dim = int(emb.size) if emb is not None else 0
            js = catalog_json(r)
            fp = bytes(r.fingerprint)
            _lib.check(self._lib.ucfp_sidecar_append_upsert(self.handle, r.tenant_id, r.record_id, fp, len(fp),
                                                            emb.ctypes.data if dim else None, dim, js, len(js)))

    def delete(self, tenant_id: int, record_ids: Iterable[int]) -> None:
        for i in record_ids:
            _lib.check(self._lib.ucfp_sidecar_append_delete(self.handle, tenant_id, int(i)))

    def sync(self) -> None:
        _lib.check(self._lib.ucfp_sidecar_sync(self.handle))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.ucfp_sidecar_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Snapshot:
    """The live rows of a log, ascending (tenant_id, record_id)."""

    def __init__(self, path: str):
        self._lib = _lib.load()
        h = C.c_void_p()
        live, entries, torn = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.ucfp_sidecar_snapshot_open(str(path).encode(), C.byref(h), C.byref(live), C.byref(entries),
                                                        C.byref(torn)))
        self.handle = h
        self.live_rows, self.log_entries, self.torn_bytes = int(live.value), int(entries.value), int(torn.value)

    def __len__(self) -> int:
        return self.live_rows

    def row(self, i: int) -> Record:
        t, rid, fl, dim, jl = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        fp, emb, js = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self._lib.ucfp_sidecar_snapshot_row(self.handle, i, C.byref(t), C.byref(rid), C.byref(fp), C.byref(fl),
                                                       C.byref(emb), C.byref(dim), C.byref(js), C.byref(jl)))
        cat = json.loads(C.string_at(js.value, jl.value)) if jl.value else {}
        e = None
        if dim.value:
            e = np.frombuffer(C.string_at(emb.value, dim.value * 4), np.float32).tolist()
        return Record(tenant_id=t.value, record_id=rid.value, modality=Modality(cat.get("modality", 1)),
                      format_version=cat.get("format_version", 0), algorithm=cat.get("algorithm", ""),
                      config_hash=cat.get("config_hash", 0), fingerprint=C.string_at(fp.value, fl.value) if fl.value else b"",
                      embedding=e, model_id=cat.get("model_id"))

    def __iter__(self) -> Iterator[Record]:
        for i in range(self.live_rows):
            yield self.row(i)

    def gather_fingerprints(self, algorithm: str, fp_len: int):
        """-> (tenants u32 [n], ids u64 [n], blobs u8 [n, fp_len]) of the rows stored under `algorithm`."""
        n = C.c_uint64(0)
        f = self._lib.ucfp_sidecar_snapshot_gather_fingerprints
        _lib.check(f(self.handle, algorithm.encode(), fp_len, None, None, None, 0, C.byref(n)))
        m = int(n.value)
        t, ids, blobs = np.zeros(m, np.uint32), np.zeros(m, np.uint64), np.zeros((m, fp_len), np.uint8)
        if m:
            _lib.check(f(self.handle, algorithm.encode(), fp_len, t.ctypes.data, ids.ctypes.data, blobs.ctypes.data, m,
                         C.byref(n)))
        return t, ids, blobs

    def gather_vectors(self, dim: int):
        n = C.c_uint64(0)
        f = self._lib.ucfp_sidecar_snapshot_gather_vectors
        _lib.check(f(self.handle, dim, None, None, None, 0, C.byref(n)))
        m = int(n.value)
        t, ids, rows = np.zeros(m, np.uint32), np.zeros(m, np.uint64), np.zeros((m, dim), np.float32)
        if m:
            _lib.check(f(self.handle, dim, t.ctypes.data, ids.ctypes.data, rows.ctypes.data, m, C.byref(n)))
        return t, ids, rows

    def dims(self) -> List[int]:
        """Embedding dimensions present (one cosine index each, like knn skips rows of another length, mod.rs:307-309)."""
        seen = set()
        dim = C.c_uint32(0)
        for i in range(self.live_rows):
            _lib.check(self._lib.ucfp_sidecar_snapshot_row(self.handle, i, None, None, None, None, None, C.byref(dim), None,
                                                           None))
            if dim.value:
                seen.add(int(dim.value))
        return sorted(seen)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.ucfp_sidecar_snapshot_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# image records: (stored algorithm, blob bytes) -> [(hash space, record-codes `algo`, `which`)]  (SURVEY 8f N2 offsets)
_IMAGE_SPACES = {
    ("imgfprint-multihash-v1", 536): [("imgfprint-ahash-v1", 7, 1), ("imgfprint-phash-v1", 7, 2), ("imgfprint-dhash-v1", 7, 4)],
    ("imgfprint-ahash-v1", 168): [("imgfprint-ahash-v1", 1, 1)],
    ("imgfprint-phash-v1", 168): [("imgfprint-phash-v1", 2, 2)],
    ("imgfprint-dhash-v1", 168): [("imgfprint-dhash-v1", 4, 4)],
}
_SIMHASH = ("simhash-b64-tf", "simhash-b64-idf")


def rebuild(path: str, ctx=None, sidecar: bool = False):
    """Start-up: replay the sidecar at `path` into a fresh GpuIndex (every hash space, every cosine dimension).
    `sidecar=True` keeps the log attached, so later upserts / deletes are appended to it."""
    import torch
    from .index import GpuIndex
    ctx = ctx or _lib.current_context()
    lib = _lib.load()
    dev = f"cuda:{ctx.device}"
    gi = GpuIndex(ctx)
    snap = Snapshot(path)
    try:
        stream = torch.cuda.current_stream().cuda_stream
        for (algorithm, fp_len), spaces in _IMAGE_SPACES.items():
            tenants, ids, blobs = snap.gather_fingerprints(algorithm, fp_len)
            for tenant in np.unique(tenants):
                sel = tenants == tenant
                d_rec = torch.from_numpy(np.ascontiguousarray(blobs[sel])).to(dev)
                n = int(sel.sum())
                for space, algo, which in spaces:
                    d_codes = torch.empty(n, dtype=torch.int64, device=dev)
                    _lib.check(lib.ucfp_image_record_codes_dev(ctx.handle, d_rec.data_ptr(), n, algo, which,
                                                               d_codes.data_ptr(), stream or None))
                    # the shard of a mutable index keeps an id map on the host: ids + 8-byte codes go back (the 168 / 536-byte
                    # records do not)
                    gi._hamming(space).upsert(int(tenant), ids[sel], d_codes.cpu().numpy().view(np.uint64))
        for algorithm in _SIMHASH:
            tenants, ids, blobs = snap.gather_fingerprints(algorithm, 8)
            for tenant in np.unique(tenants):
                sel = tenants == tenant
                gi._hamming(algorithm).upsert(int(tenant), ids[sel], np.ascontiguousarray(blobs[sel]).view(np.uint64).reshape(-1))
        for dim in snap.dims():
            tenants, ids, rows = snap.gather_vectors(dim)
            for tenant in np.unique(tenants):
                sel = tenants == tenant
                gi._cosine(dim).upsert(int(tenant), ids[sel], rows[sel])
    finally:
        snap.close()
    if sidecar:
        gi.attach_sidecar(Sidecar(path))
    return gi
