"""Sharded brute-force search across the GPUs of one node -- a BINDING of the C entry points
`ucfp_shard_comm_*` / `ucfp_index_search_sharded_*` (ucfp_amd/csrc/shard.hip).

One process per GPU.  The corpus is range-partitioned by insertion index; queries are replicated; every rank
searches its shard and the ONLY data-path exchange is ONE all-gather of the per-shard top-k lists as packed
16-byte entries {u64 id, u32 key, u32 0} (nq x k x 16 B per rank -- a few hundred KB, latency-bound on xGMI), after
which every rank runs the same deterministic merge ((key asc, id asc)) and holds the full answer.  The library calls
RCCL itself (ncclAllGather on its own side stream, under the next batch's shard scan); a Rust host reaches exactly
the same code.

torch is plumbing here: device buffers, the current stream, and `torch.distributed` as the CONTROL channel that
ships the 128-byte RCCL unique id from rank 0 to the other ranks.  With the `gloo` backend (CPU tests; several
ranks rehearsing on one GPU) the exchange goes pack -> gloo all-gather -> merge through `ucfp_topk_pack_dev` /
`ucfp_topk_merge_packed_dev`, the same wire format.
"""
import ctypes as C
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import errors as _errors
from . import index as _index

ENTRY_BYTES = 16
FORCE_RCCL = 1      # UCFP_SHARD_FORCE_RCCL


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) of the global insertion order owned by `rank`; the first
    n_total % world ranks hold one extra row (same rule as ucfp_shard_range)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_entries(ids: np.ndarray, keys: np.ndarray) -> np.ndarray:
    """Host-side statement of the wire format: ids u64 [nq, k], keys u32 [nq, k] -> int64 [nq, k, 2]
    (word 0 = id, word 1 = key in the low 32 bits, zero above) = 16 little-endian bytes per entry."""
    e = np.zeros(ids.shape + (2,), np.uint64)
    e[..., 0] = ids
    e[..., 1] = keys.astype(np.uint64)
    return e.view(np.int64)


def unpack_entries(entries: np.ndarray):
    e = np.ascontiguousarray(entries).view(np.uint64)
    return e[..., 0].copy(), (e[..., 1] & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def all_gather_entries(entries: torch.Tensor, group=None) -> torch.Tensor:
    """ONE collective: entries int64 [nq, k, 2] -> [world, nq, k, 2] in rank order, exactly the [parts][nq][k]
    layout ucfp_topk_merge_packed_dev consumes.  Used by the gloo paths (CPU tensors, or device tensors staged
    through the host); the RCCL path runs inside the library."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return entries.unsqueeze(0).contiguous()
    if entries.is_cuda and dist.get_backend(group) == "gloo":
        return all_gather_entries(entries.cpu(), group).to(entries.device)
    out = torch.empty((world,) + tuple(entries.shape), dtype=entries.dtype, device=entries.device)
    dist.all_gather_into_tensor(out.view(-1), entries.contiguous().view(-1), group=group)
    return out


class ShardComm:
    """RAII wrapper of one ucfp_shard_comm.  Collective: every rank of `group` constructs it."""

    def __init__(self, ctx=None, group=None, force_rccl: bool = False):
        """force_rccl: build a real RCCL communicator even for a single rank (UCFP_SHARD_FORCE_RCCL): the one-GPU
        way to execute the multi-GPU code path (ncclCommInitRank + one ncclAllGather per batch + merge)."""
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        inited = dist.is_initialized()
        self.rank = dist.get_rank(group) if inited else 0
        self.world = dist.get_world_size(group) if inited else 1
        self.backend = dist.get_backend(group) if inited and self.world > 1 else None
        # gloo: the host moves the entries itself; the C communicator is then a world-1 one (local scan + merge only)
        c_world = self.world if self.backend == "nccl" else 1
        uid = (C.c_uint8 * 128)()
        if c_world == 1 and force_rccl:
            _lib.check(self._lib.ucfp_shard_unique_id(uid))
        if c_world > 1:
            if self.rank == 0:
                _lib.check(self._lib.ucfp_shard_unique_id(uid))
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=f"cuda:{self.ctx.device}")
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_shard_comm_create_ex(self.ctx.handle, uid, self.rank if c_world > 1 else 0, c_world,
                                                       FORCE_RCCL if force_rccl else 0, C.byref(h)))
        self.handle = h
        self.uses_rccl = bool(self._lib.ucfp_shard_comm_uses_rccl(h))

    def exchanges(self) -> int:
        n = C.c_uint64(0)
        _lib.check(self._lib.ucfp_shard_comm_info(self.handle, None, None, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_shard_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedIndex:
    """This rank's shard plus the collective search. All tensors live on this rank's GPU."""

    def __init__(self, kind: int, dim: int = 0, ctx=None, group=None, tenant: int = 0, force_rccl: bool = False):
        self.kind, self.dim, self.group, self.tenant = kind, dim, group, tenant
        self.ctx = ctx or _lib.current_context()
        self._lib = _lib.load()
        self.local = _index.DeviceIndex(kind, dim, _index.APPEND_ONLY, self.ctx)
        self.comm = ShardComm(self.ctx, group, force_rccl)
        self.world = self.comm.world
        self.rccl = self.comm.backend == "nccl"
        self._bufs = {}
        self._slot = 1

    def append_local(self, ids: torch.Tensor, rows: torch.Tensor) -> None:
        """Append device-resident rows of THIS rank's range (ids int64 view of u64)."""
        self._check_device(ids)
        stream = torch.cuda.current_stream().cuda_stream
        self.local.append_dev(self.tenant, ids.data_ptr(), rows.data_ptr(), ids.numel(), stream)

    def _check_device(self, t: torch.Tensor) -> None:
        if not t.is_cuda or t.device.index != self.ctx.device:
            raise _index.InvalidArgument(f"tensor on {t.device}, index context on cuda:{self.ctx.device}")

    def _buffers(self, nq: int, k: int, device, slot: int = 0):
        key = (nq, k, slot)
        b = self._bufs.get(key)
        if b is None:
            mk = lambda dt: torch.empty((nq, k), dtype=dt, device=device)  # noqa: E731
            b = self._bufs[key] = dict(
                out_ids=mk(torch.int64), out_keys=mk(torch.int32), out_scores=mk(torch.float32),
                out_cnt=torch.empty((nq,), dtype=torch.int32, device=device), ticket=None, done=None)
            if self.world > 1 and not self.rccl:   # gloo rehearsal: local lists + entries live here
                b.update(ids=mk(torch.int64), keys=mk(torch.int32),
                         cnt=torch.empty((nq,), dtype=torch.int32, device=device),
                         entries=torch.empty((nq, k, 2), dtype=torch.int64, device=device))
        return b

    def submit(self, queries: torch.Tensor, k: int) -> Tuple[int, int, int]:
        """Start one batch: the local search runs on the current stream; the exchange (all-gather of the per-shard
        top-k + merge) runs on the communicator's side stream, so it overlaps the NEXT batch's local search (two
        buffer sets; at most two batches in flight).  The query tensor must stay untouched until the ticket is
        collected.  Returns a ticket for `collect`."""
        self._check_device(queries)
        nq = queries.shape[0]
        slot = self._slot = self._slot ^ 1
        b = self._buffers(nq, k, queries.device, slot)
        cur = torch.cuda.current_stream()
        if self.world == 1 or self.rccl:
            t = C.c_uint64(0)
            _lib.check(self._lib.ucfp_index_search_sharded_submit(
                self.local.handle, self.comm.handle, self.tenant, queries.data_ptr(), nq, k, b["out_ids"].data_ptr(),
                b["out_scores"].data_ptr(), b["out_keys"].data_ptr(), b["out_cnt"].data_ptr(), cur.cuda_stream or None,
                C.byref(t)))
            b["ticket"] = int(t.value)
            return (nq, k, slot)
        # gloo: same stages, the host's transport in the middle
        if b["done"] is not None:
            cur.wait_event(b["done"])
        self.local.search_dev(self.tenant, queries.data_ptr(), nq, k, b["ids"].data_ptr(), 0, b["keys"].data_ptr(),
                              b["cnt"].data_ptr(), cur.cuda_stream)
        _lib.check(self._lib.ucfp_topk_pack_dev(self.ctx.handle, b["ids"].data_ptr(), b["keys"].data_ptr(), nq, k,
                                                b["entries"].data_ptr(), cur.cuda_stream or None))
        gathered = all_gather_entries(b["entries"], self.group)
        _lib.check(self._lib.ucfp_topk_merge_packed_dev(
            self.ctx.handle, self.kind, gathered.data_ptr(), gathered.shape[0], nq, k, b["out_ids"].data_ptr(),
            b["out_scores"].data_ptr(), b["out_keys"].data_ptr(), b["out_cnt"].data_ptr(), cur.cuda_stream or None))
        b["gathered"] = gathered
        done = torch.cuda.Event()
        done.record(cur)
        b["done"] = done
        return (nq, k, slot)

    def collect(self, ticket: Tuple[int, int, int]):
        """Make the current stream wait for a submitted batch; returns (ids, scores, keys, counts)."""
        nq, k, slot = ticket
        b = self._bufs[(nq, k, slot)]
        cur = torch.cuda.current_stream()
        if b["ticket"] is not None:
            _lib.check(self._lib.ucfp_index_search_sharded_collect(self.comm.handle, b["ticket"], cur.cuda_stream or None))
        elif b["done"] is not None:
            cur.wait_event(b["done"])
        return b["out_ids"], b["out_scores"], b["out_keys"], b["out_cnt"]

    def missing_shards(self, ticket: Tuple[int, int, int]) -> int:
        """Bit mask of the ranks whose shard is ABSENT from the ticket's answer (their scan failed; they joined the
        all-gather with the marked empty list) -- the same on every rank.  Synchronises with the batch."""
        nq, k, slot = ticket
        b = self._bufs[(nq, k, slot)]
        if b["ticket"] is None:
            return 0                      # host transport (gloo): a failing rank raises in its own all-gather
        m = C.c_uint64(0)
        _lib.check(self._lib.ucfp_index_search_sharded_missing(self.comm.handle, b["ticket"], C.byref(m)))
        return int(m.value)

    def search(self, queries: torch.Tensor, k: int, check: bool = False):
        """queries: device tensor, [nq] int64 (u64 hashes) or [nq, dim] float32, identical on
        every rank. Returns (ids int64 [nq,k], scores f32 [nq,k], keys int32 [nq,k], counts [nq]).
        The returned tensors belong to the index and are overwritten two searches later.
        check: wait for the batch and raise on EVERY rank if some rank's shard is missing from the answer."""
        t = self.submit(queries, k)
        out = self.collect(t)
        if check:
            m = self.missing_shards(t)
            if m:
                raise _errors.IndexError_(f"partial result: shards {[r for r in range(64) if m >> r & 1]} missing")
        return out

    def close(self):
        self.comm.close()
        self.local.close()
