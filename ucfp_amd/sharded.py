"""Sharded brute-force search across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The corpus is
range-partitioned by insertion index; queries are replicated; every rank searches its shard and
the ONLY data-path exchange is one all-gather of the per-shard top-k lists
(nq x k x (u64 id + u32 key) per rank -- a few hundred KB, latency-bound on xGMI), after which
every rank runs the same deterministic merge ((key asc, id asc)) and holds the full answer.

torch is plumbing here: device buffers, the current stream and the collective.  The search and
merge kernels are the C-ABI ones (ucfp_index_search_dev / ucfp_topk_merge_dev).
"""
from typing import Tuple

import torch
import torch.distributed as dist

from . import index as _index


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) of the global insertion order owned by `rank`; the first
    n_total % world ranks hold one extra row."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_topk(ids: torch.Tensor, keys: torch.Tensor, group=None):
    """ids int64 [nq, k], keys int32 [nq, k] (bit patterns of u64 / u32) -> stacked
    [world, nq, k] tensors in rank order -- exactly the [parts][nq][k] layout
    ucfp_topk_merge_dev consumes.  Works on nccl (device tensors) and gloo (CPU tensors)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return ids.unsqueeze(0).contiguous(), keys.unsqueeze(0).contiguous()
    if ids.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal mode (several ranks sharing one GPU over gloo): stage through the host
        g_ids, g_keys = all_gather_topk(ids.cpu(), keys.cpu(), group)
        return g_ids.to(ids.device), g_keys.to(keys.device)
    nq = ids.shape[0]
    # concatenation along dim 0 in rank order == [world][nq][k] row-major
    g_ids = torch.empty((world * nq,) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    g_keys = torch.empty((world * nq,) + tuple(keys.shape[1:]), dtype=keys.dtype, device=keys.device)
    dist.all_gather_into_tensor(g_ids, ids.contiguous(), group=group)
    dist.all_gather_into_tensor(g_keys, keys.contiguous(), group=group)
    return g_ids.view((world,) + tuple(ids.shape)), g_keys.view((world,) + tuple(keys.shape))


class ShardedIndex:
    """This rank's shard plus the collective search. All tensors live on this rank's GPU."""

    def __init__(self, kind: int, dim: int = 0, ctx=None, group=None, tenant: int = 0):
        self.kind, self.dim, self.group, self.tenant = kind, dim, group, tenant
        self.ctx = ctx
        self.local = _index.DeviceIndex(kind, dim, _index.APPEND_ONLY, ctx)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bufs = {}

    def append_local(self, ids: torch.Tensor, rows: torch.Tensor) -> None:
        """Append device-resident rows of THIS rank's range (ids int64 view of u64)."""
        stream = torch.cuda.current_stream().cuda_stream
        self.local.append_dev(self.tenant, ids.data_ptr(), rows.data_ptr(), ids.numel(), stream)

    def _buffers(self, nq: int, k: int, device, slot: int = 0):
        key = (nq, k, slot)
        b = self._bufs.get(key)
        if b is None:
            mk = lambda dt: torch.empty((nq, k), dtype=dt, device=device)  # noqa: E731
            b = self._bufs[key] = dict(
                ids=mk(torch.int64), keys=mk(torch.int32), cnt=torch.empty((nq,), dtype=torch.int32, device=device),
                out_ids=mk(torch.int64), out_keys=mk(torch.int32), out_scores=mk(torch.float32),
                out_cnt=torch.empty((nq,), dtype=torch.int32, device=device), done=None)
        return b

    def submit(self, queries: torch.Tensor, k: int) -> Tuple[int, int, int]:
        """Start one batch: the local search runs on the current stream; the exchange (all-gather of
        the per-shard top-k + merge) runs on a side stream of this index, so it overlaps the NEXT
        batch's local search (two buffer sets; at most two batches in flight).  The query tensor must
        stay untouched until the ticket is collected.  Returns a ticket for `collect`."""
        nq = queries.shape[0]
        slot = self._slot = getattr(self, "_slot", 1) ^ 1
        b = self._buffers(nq, k, queries.device, slot)
        cur = torch.cuda.current_stream()
        if b["done"] is not None:
            cur.wait_event(b["done"])          # the exchange that last used this buffer set has finished
        self.local.search_dev(self.tenant, queries.data_ptr(), nq, k, b["ids"].data_ptr(), 0,
                              b["keys"].data_ptr(), b["cnt"].data_ptr(), cur.cuda_stream)
        if self.world == 1:
            xs = cur
        else:
            if getattr(self, "_xs", None) is None:
                self._xs = torch.cuda.Stream(device=queries.device)
            xs = self._xs
            searched = torch.cuda.Event()
            searched.record(cur)
            xs.wait_event(searched)
        with torch.cuda.stream(xs):
            g_ids, g_keys = all_gather_topk(b["ids"], b["keys"], self.group)
            _index.topk_merge_dev(self.kind, g_ids.data_ptr(), g_keys.data_ptr(), g_ids.shape[0], nq, k,
                                  b["out_ids"].data_ptr(), b["out_scores"].data_ptr(), b["out_keys"].data_ptr(),
                                  b["out_cnt"].data_ptr(), xs.cuda_stream, ctx=self.ctx)
            b["gathered"] = (g_ids, g_keys)    # keep the gather buffers alive until the merge has run
            done = torch.cuda.Event()
            done.record(xs)
            b["done"] = done
        return (nq, k, slot)

    def collect(self, ticket: Tuple[int, int, int]):
        """Make the current stream wait for a submitted batch; returns (ids, scores, keys, counts)."""
        nq, k, slot = ticket
        b = self._bufs[(nq, k, slot)]
        torch.cuda.current_stream().wait_event(b["done"])
        return b["out_ids"], b["out_scores"], b["out_keys"], b["out_cnt"]

    def search(self, queries: torch.Tensor, k: int):
        """queries: device tensor, [nq] int64 (u64 hashes) or [nq, dim] float32, identical on
        every rank. Returns (ids int64 [nq,k], scores f32 [nq,k], keys int32 [nq,k], counts [nq]).
        The returned tensors belong to the index and are overwritten two searches later."""
        return self.collect(self.submit(queries, k))
