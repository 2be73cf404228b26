"""Multi-GPU PK-FK join: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm), hash-owner partitioning and ONE all-to-all exchange over xGMI.

This is the only step of the path that needs communication, and it sits exactly
where the reference shuffles partitions through host DRAM
(Partitioner::GetOffsets / LoadPartitions, host/partition/partitioner.cc:280-375;
host->DPU reload, host/join/join_dpu.cc:269,293):

  1. every rank holds a slice of the batches of both tables;
  2. each rank groups its tuples by owner = top log2(G) bits of wang_hash(key)
     (the reference's BUCKET_OF with nr_partitions = G, partition.c:45-46) --
     olap_partition_u32, so each destination's tuples are one contiguous region;
  3. counts, then keys and payloads, are exchanged with all_to_all_single
     (per-peer message ~ 1/G of the local rows; xGMI is point to point, each
     message rides its own link);
  4. each rank runs the single-GPU join on what it received, telling it that the
     top log2(G) hash bits are already fixed (olap_join_set_hash_skip_bits).
The result stays sharded by owner (the reference's output is a list of per-
partition batches too, join_dpu.cc:371-399).

`ops` abstracts the local operators so the host logic can be exercised on CPU with
gloo in tests (tests provide an oracle-backed ops object); the product backend is
GpuLocalOps (C ABI on device pointers of torch CUDA tensors)."""
from __future__ import annotations

import os
import time
from typing import Sequence

import torch
import torch.distributed as dist

from . import capi


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.view(torch.int32)


class DeviceBatches(list):
    """A table's batches as (key, payload) tensor pairs that also carry their pointer and length tables,
    built once -- like an Arrow RecordBatchVector, whose batches hold their buffer addresses: a Run() then
    hands the tables over instead of asking thousands of tensors for their data_ptr() again."""

    def __init__(self, pairs):
        super().__init__(pairs)
        self.keys = [p[0] for p in self]
        self.payloads = [p[1] for p in self]
        self.rows = [int(k.numel()) for k in self.keys]
        self.key_table = capi.ptr_table([k.data_ptr() for k in self.keys])
        self.payload_table = capi.ptr_table([p.data_ptr() for p in self.payloads])
        self.rows_table = capi.u64_table(self.rows)


class GpuLocalOps:
    """Local partition / join through libolap_gpu.so on torch CUDA tensors (int32 tensors
    carry the uint32 bit patterns; torch is only the allocator and the RCCL front end)."""

    overlap_right_exchange = True   # fused path: the left table's radix pass runs under the right table's exchange
    left_slices = 2                 # fused path: the left table is split and shipped in this many input slices, so the
                                    # links start working after 1/left_slices of the split instead of all of it
    exchange_groups = 4             # fused path: a table travels as this many sub-bucket groups; group g is joined
                                    # while group g + 1 is on the links

    def __init__(self, ctx: capi.Context, device: torch.device):
        self.ctx = ctx
        self.device = device

    def empty(self, n: int) -> torch.Tensor:
        return torch.empty(max(n, 1), dtype=torch.int32, device=self.device)[:n]

    # Columns that a kernel streams side by side (the {fk, y, x} result, a table's key and payload) should not sit an
    # exact power of two apart: equal-sized columns allocated back to back do (16 GiB each at SF=2048), their row i
    # then maps to the same HBM channel in every column and the streams collide.  column_skew_bytes > 0 carves the
    # columns out of ONE allocation, column c shifted by (2c + 1) x column_skew_bytes + c x 4352 bytes.
    column_skew_bytes = 1 << 20

    def empty_columns(self, n: int, k: int):
        """k int32 columns of n rows each."""
        skew = int(self.column_skew_bytes)
        if not skew or n == 0:
            return tuple(self.empty(n) for _ in range(k))
        off = lambda c: (c * ((n * 4 + 255) // 256 * 256) + ((2 * c + 1) * skew + c * 4352)) // 4
        slab = torch.empty(off(k) + 4, dtype=torch.int32, device=self.device)
        return tuple(slab[off(c):off(c) + n] for c in range(k))

    def partition(self, keys: Sequence[torch.Tensor], payload: Sequence[torch.Tensor], nr_partitions: int):
        rows = [int(k.numel()) for k in keys]
        n = sum(rows)
        ok, op = self.empty(n), self.empty(n)
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        offs = self.ctx.partition_u32([k.data_ptr() for k in keys], [p.data_ptr() for p in payload], rows,
                                      capi.MEM_DEVICE, nr_partitions, ok.data_ptr(), op.data_ptr(), n, capi.MEM_DEVICE)
        return ok, op, offs

    def empty_tuples(self, n: int) -> torch.Tensor:
        return torch.empty(max(n, 2), dtype=torch.int64, device=self.device)[:n]

    def partition_tuples(self, keys: Sequence[torch.Tensor], payload: Sequence[torch.Tensor], nr_partitions: int):
        """Owner split that keeps (key, payload) zipped: int64 tensor key | payload << 32."""
        rows = [int(k.numel()) for k in keys]
        n = sum(rows)
        out = self.empty_tuples(n)
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        offs = self.ctx.partition_tuples_u32([k.data_ptr() for k in keys], [p.data_ptr() for p in payload], rows,
                                             capi.MEM_DEVICE, nr_partitions, out.data_ptr(), n)
        return out, offs

    def split_tuples(self, keys: Sequence[torch.Tensor], payload: Sequence[torch.Tensor], owner_bits: int, group_bits: int,
                     low_bits: int):
        """The multi-GPU split (olap_split_tuples_u32): buckets [owner | group | low], laid out group-major."""
        rows = [int(k.numel()) for k in keys]
        n = sum(rows)
        out = self.empty_tuples(n)
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        offs = self.ctx.split_tuples_u32([k.data_ptr() for k in keys], [p.data_ptr() for p in payload], rows, capi.MEM_DEVICE,
                                         owner_bits, group_bits, low_bits, out.data_ptr(), n)
        return out, offs

    def empty_result(self, n: int):
        """The three result columns of this rank, filled group by group."""
        return self.empty_columns(n, 3)

    def join_tuples(self, lt: torch.Tensor, rt: torch.Tensor, skip_bits: int = 0):
        n = int(lt.numel())
        out = self.empty_columns(n, 3)
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        self.ctx.join_set_hash_skip_bits(skip_bits)
        try:
            got, stats = self.ctx.join_tuples_u32(lt.data_ptr(), n, rt.data_ptr(), int(rt.numel()), out[0].data_ptr(),
                                                  out[1].data_ptr(), out[2].data_ptr(), n)
        finally:
            self.ctx.join_set_hash_skip_bits(0)
        return out[0][:got], out[1][:got], out[2][:got], stats

    def join_presplit(self, lt: torch.Tensor, l_lo, l_hi, rt: torch.Tensor, r_lo, r_hi, n_segments: int, chunks: int,
                      skip_bits: int = 0, right_ready: "torch.cuda.Event | None" = None, out=None):
        """Local join of exchanged tuples that are still grouped (sender, sub-bucket): the first
        radix pass reads the chunks in place (olap_join_presplit_u32).  right_ready: an event
        recorded behind the right table's exchange -- the left table's pass then runs while the
        right table is still arriving."""
        l_ptr, l_span, l_lo, l_hi, n = self._span(lt, l_lo, l_hi)
        r_ptr, r_span, r_lo, r_hi, _ = self._span(rt, r_lo, r_hi)
        out = list(out) if out is not None else list(self.empty_columns(n, 3))   # out: slices of empty_result()
        if right_ready is None:
            torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        else:
            self.ctx.join_set_right_ready_event(right_ready.cuda_event)
        self.ctx.join_set_hash_skip_bits(skip_bits)
        try:
            got, stats = self.ctx.join_presplit_u32(l_ptr, l_span, l_lo, l_hi, r_ptr, r_span, r_lo, r_hi,
                                                    n_segments, chunks, out[0].data_ptr(), out[1].data_ptr(),
                                                    out[2].data_ptr(), n)
        finally:
            self.ctx.join_set_hash_skip_bits(0)
        return out[0][:got], out[1][:got], out[2][:got], stats

    @staticmethod
    def _span(buffers, lo, hi):
        """A table given as one tensor, or as several received one after the other (chunk indices then count
        through their concatenation): -> (base pointer, tuples spanned, chunk lo / hi relative to the base,
        rows).  Separate buffers are NOT copied together: the chunks are re-based on the lowest address, and
        whatever lies between the buffers is never read."""
        rows = sum(h - l for l, h in zip(lo, hi))
        if isinstance(buffers, torch.Tensor):
            return buffers.data_ptr(), int(buffers.numel()), lo, hi, rows
        bufs = [b for b in buffers]
        if len(bufs) == 1:
            return bufs[0].data_ptr(), int(bufs[0].numel()), lo, hi, rows
        base = min(b.data_ptr() for b in bufs)
        starts, v = [], 0                       # virtual start of every buffer in the concatenation
        for b in bufs:
            starts.append(v)
            v += int(b.numel())
        shift = [(b.data_ptr() - base) // 8 - st for b, st in zip(bufs, starts)]
        import bisect
        which = lambda pos: max(0, bisect.bisect_right(starts, pos) - 1)
        nlo, nhi = [], []
        for l, h in zip(lo, hi):
            k = which(l if h > l else max(l - 1, 0))
            nlo.append(l + shift[k])
            nhi.append(h + shift[k])
        span = max((b.data_ptr() - base) // 8 + int(b.numel()) for b in bufs)
        return base, span, nlo, nhi, rows

    def join_batches(self, left: DeviceBatches, right: DeviceBatches, skip_bits: int = 0):
        """join() on batches that carry their tables (DeviceBatches)."""
        if not len(left) or not len(right):           # (an empty ctypes table still has one slot)
            return self.join(left.keys, left.payloads, right.keys, right.payloads, skip_bits)
        n = sum(left.rows)
        out = self.empty_columns(n, 3)
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        self.ctx.join_set_hash_skip_bits(skip_bits)
        try:
            got, stats = self.ctx.join_u32(left.key_table, left.payload_table, left.rows_table, right.key_table, right.payload_table,
                                           right.rows_table, capi.MEM_DEVICE, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                           n, capi.MEM_DEVICE)
        finally:
            self.ctx.join_set_hash_skip_bits(0)
        return out[0][:got], out[1][:got], out[2][:got], stats

    def join(self, lfk, ly, rpk, rx, skip_bits: int = 0):
        lrows = [int(t.numel()) for t in lfk]
        rrows = [int(t.numel()) for t in rpk]
        n = sum(lrows)
        out = self.empty_columns(n, 3)
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        self.ctx.join_set_hash_skip_bits(skip_bits)
        try:
            got, stats = self.ctx.join_u32([t.data_ptr() for t in lfk], [t.data_ptr() for t in ly], lrows,
                                           [t.data_ptr() for t in rpk], [t.data_ptr() for t in rx], rrows,
                                           capi.MEM_DEVICE, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                           n, capi.MEM_DEVICE)
        finally:
            self.ctx.join_set_hash_skip_bits(0)
        return out[0][:got], out[1][:got], out[2][:got], stats


class _Pending:
    """An exchange in flight: .wait() returns the received tensor."""

    def __init__(self, recv, work=None):
        self.recv, self.work = recv, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.recv


def _pairwise_only(group) -> bool:
    """gloo has no all_to_all_single on some builds: only THERE is the pairwise isend/irecv form used.  On RCCL
    ("nccl") a failing collective is an error of the product path and is raised as such -- a silent
    degradation would hide exactly the failure an 8-GPU run has to show."""
    return dist.get_backend(group) == "gloo"


# RCCL (torch 2.10 + ROCm 7) moves a message of more than 1 GiB only in part, SILENTLY (scripts/dbg_a2a.py: a 2 GiB
# all_to_all_single at world size 1 delivers its first GiB).  Messages above this many elements therefore travel in rounds of
# list-form all_to_alls over views of the same buffers; every rank must run the same number of rounds, so the largest
# message of the whole exchange travels with the counts (_split).
MAX_MESSAGE_ELEMS = int(os.environ.get("OLAP_MAX_MESSAGE_ELEMS", 1 << 26))      # int64 tuples: 512 MiB (the override is for tests)


class _PendingMany:
    """Several collectives in flight that fill one receive buffer."""

    def __init__(self, recv, works):
        self.recv, self.works = recv, works

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        return self.recv


def _exchange(send: torch.Tensor, in_splits, out_splits, empty, group, async_op: bool = False, largest: int | None = None):
    """Personalised all-to-all of a 1-D tensor.  RCCL/NCCL: all_to_all_single (optionally
    asynchronous, so it overlaps the partition kernels of the other table); gloo builds
    without it: pairwise isend/irecv.  largest: the largest per-peer message of this exchange over ALL ranks (None: unknown --
    then only a one-rank group, which knows it, splits oversized messages)."""
    recv = empty(int(sum(out_splits)))
    world = dist.get_world_size(group)
    if largest is None and world == 1:
        largest = max([int(v) for v in in_splits] + [0])
    if largest is not None and largest > MAX_MESSAGE_ELEMS and not _pairwise_only(group):
        so, ro = [0], [0]
        for v in in_splits:
            so.append(so[-1] + int(v))
        for v in out_splits:
            ro.append(ro[-1] + int(v))
        works = []
        for r in range((largest + MAX_MESSAGE_ELEMS - 1) // MAX_MESSAGE_ELEMS):
            lo = r * MAX_MESSAGE_ELEMS
            piece = lambda base, n: (base + min(lo, n), base + min(lo + MAX_MESSAGE_ELEMS, n))
            ins = [send[slice(*piece(so[p], int(in_splits[p])))] for p in range(world)]
            outs = [recv[slice(*piece(ro[p], int(out_splits[p])))] for p in range(world)]
            works.append(dist.all_to_all(outs, ins, group=group, async_op=async_op))
        return _PendingMany(recv, works) if async_op else recv
    try:
        work = dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits),
                                      group=group, async_op=async_op)
        return _Pending(recv, work) if async_op else recv
    except (RuntimeError, NotImplementedError):
        if not _pairwise_only(group):
            raise
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    so = [0]
    for s in in_splits:
        so.append(so[-1] + s)
    ro = [0]
    for s in out_splits:
        ro.append(ro[-1] + s)
    recv[ro[rank]:ro[rank + 1]].copy_(send[so[rank]:so[rank + 1]])
    reqs = []
    for peer in range(world):
        if peer == rank:
            continue
        gp = dist.get_global_rank(group, peer) if group is not None else peer
        if in_splits[peer]:
            reqs.append(dist.isend(send[so[peer]:so[peer + 1]].contiguous(), gp, group=group))
        if out_splits[peer]:
            reqs.append(dist.irecv(recv[ro[peer]:ro[peer + 1]], gp, group=group))
    for r in reqs:
        r.wait()
    return _Pending(recv) if async_op else recv


def _exchange_counts(cnt_in: torch.Tensor, world: int, group) -> torch.Tensor:
    cnt_out = torch.empty_like(cnt_in)
    try:
        dist.all_to_all_single(cnt_out, cnt_in, group=group)
        return cnt_out
    except (RuntimeError, NotImplementedError):
        if not _pairwise_only(group):
            raise
    per = cnt_in.numel() // world
    gathered = [torch.empty_like(cnt_in) for _ in range(world)]
    dist.all_gather(gathered, cnt_in, group=group)
    rank = dist.get_rank(group)
    return torch.stack([gt.view(world, per)[rank] for gt in gathered]).view(-1)


FUSED_SPLIT_BITS = 9  # owner bits + sub-bucket bits of the fused split: one 512-bucket pass


def _chunk_tables(cnt: torch.Tensor):
    """cnt[q][j] = tuples sender q holds for sub-bucket j of this rank, in the order they sit in
    the received buffer (sender-major).  Returns segment-major (lo, hi) tables: chunk (j, q)."""
    world, S = cnt.shape
    flat = cnt.reshape(-1)
    end = torch.cumsum(flat, 0)
    lo = (end - flat).view(world, S).t().contiguous().view(-1)
    hi = end.view(world, S).t().contiguous().view(-1)
    return [int(v) for v in lo], [int(v) for v in hi]


class _Shipped:
    """One input slice of one table after the split.  ship(g) starts the all-to-all of sub-bucket group g;
    pend[g].wait() -> what this rank received for group g (sender-major, then low sub-bucket);
    recv[q][g][l] = tuples sender q shipped for (group g, low sub-bucket l); sent[g][p], got[g][q] = message sizes."""

    def __init__(self, ops, t, off, recv, sent, got, world, L, group, largest=None):
        self.ops, self.t, self.off, self.recv, self.sent, self.got = ops, t, off, recv, sent, got
        self.world, self.L, self.group = world, L, group
        self.largest = largest      # the largest message (tuples) any rank sends any peer for any group of this slice
        self.pend = [None] * len(sent)

    def ship(self, g: int):
        run = self.t[self.off[g * self.world * self.L]:self.off[(g + 1) * self.world * self.L]]   # everything group g sends, owner-major
        self.pend[g] = _exchange(run, self.sent[g], self.got[g], self.ops.empty_tuples, self.group, async_op=True, largest=self.largest)

    def ship_all(self):
        for g in range(len(self.pend)):
            self.ship(g)
        return self


def _split(ops, keys, payload, world: int, gbits: int, lbits: int, group) -> _Shipped:
    """Split one slice into world x K x L buckets (owner = top hash bits, then group, then low sub-bucket)
    laid out group-major, and tell every owner how much it will receive (one small all-to-all)."""
    obits, K, L = world.bit_length() - 1, 1 << gbits, 1 << lbits
    t, off = ops.split_tuples(keys, payload, obits, gbits, lbits)          # bucket order (group, owner, low)
    cnt = torch.tensor([off[k + 1] - off[k] for k in range(K * world * L)], dtype=torch.int64).view(K, world, L)
    sent = [[int(v) for v in cnt[g].sum(1)] for g in range(K)]
    # per owner: its K x L counts, and this rank's largest message (so that every rank learns the largest of all: _exchange)
    mine = max(max(row) for row in sent) if sent else 0
    to_peers = torch.cat([cnt.permute(1, 0, 2).contiguous().view(world, K * L), torch.full((world, 1), mine, dtype=torch.int64)], dim=1).view(-1)
    back = _exchange_counts(to_peers.to(t.device), world, group).view(world, K * L + 1).cpu()
    recv = back[:, :K * L].contiguous().view(world, K, L)
    largest = int(back[:, K * L].max())
    got = [[int(v) for v in recv[:, g].sum(1)] for g in range(K)]
    return _Shipped(ops, t, off, recv, sent, got, world, L, group, largest)


def _group_tables(slices, g: int, world: int, L: int, width: int):
    """Chunk tables of group g for one table shipped in len(slices) input slices: segment l has `width`
    chunk places; place (h, q) holds what sender q shipped in slice h, the rest stay empty.  Chunk indices
    count through the concatenation of the slices' receive buffers."""
    lo, hi, v = [0] * (L * width), [0] * (L * width), 0
    for h, sh in enumerate(slices):
        c_lo, c_hi = _chunk_tables(sh.recv[:, g])
        for l in range(L):
            for q in range(world):
                lo[l * width + h * world + q], hi[l * width + h * world + q] = v + c_lo[l * world + q], v + c_hi[l * world + q]
        v += sum(sh.got[g])
    for l in range(L):                                                     # empty places repeat the previous end (lo == hi)
        for k in range(len(slices) * world, width):
            lo[l * width + k] = hi[l * width + k] = hi[l * width + len(slices) * world - 1]
    return lo, hi


def distributed_join(ops, left_batches: Sequence, right_batches: Sequence, group=None, timers: dict | None = None):
    """left_batches / right_batches: this rank's batches as (key, payload) tensor pairs.
    Returns (fk, y, x, info): this rank's shard of the join result (rows whose key it owns)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    # force_exchange: take the split -> all-to-all -> pre-split join route at world size 1 too (a process group of one
    # rank), so that the exchange code runs -- through RCCL -- on a one-GPU box (tests/test_bench_multirank_gpu.py)
    forced = world == 1 and dist.is_initialized() and getattr(ops, "force_exchange", False)
    if forced:
        pass
    elif world == 1 and isinstance(left_batches, DeviceBatches) and isinstance(right_batches, DeviceBatches) and hasattr(ops, "join_batches"):
        fk, y, x, stats = ops.join_batches(left_batches, right_batches, 0)
        return fk, y, x, {"world": 1, "stats": stats, "sent_rows": 0}
    lfk, ly = [b[0] for b in left_batches], [b[1] for b in left_batches]
    rpk, rx = [b[0] for b in right_batches], [b[1] for b in right_batches]
    if world == 1 and not forced:
        fk, y, x, stats = ops.join(lfk, ly, rpk, rx, 0)
        return fk, y, x, {"world": 1, "stats": stats, "sent_rows": 0}
    if world & (world - 1):
        raise ValueError("the owner split uses the top log2(G) hash bits: world size must be a power of two")
    g = world.bit_length() - 1
    rank = dist.get_rank(group)
    if hasattr(ops, "join_presplit") and world <= (1 << (FUSED_SPLIT_BITS - 1)) and getattr(ops, "fused", True):
        # fused path: the owner split IS the first radix pass (world x K x L = 512 buckets).  A table is
        # shipped as K sub-bucket groups (one all-to-all each); the receiver joins group g -- its next radix
        # pass reads the received runs in place -- while group g + 1 is still on the links.  The left table
        # goes out in H input slices, so the links start after 1/H of its split.
        S = (1 << FUSED_SPLIT_BITS) // world
        K = max(1, min(int(getattr(ops, "exchange_groups", 1)), S // 2))
        K = 1 << (K.bit_length() - 1)
        gbits, lbits = K.bit_length() - 1, (S // K).bit_length() - 1
        L = 1 << lbits
        t0 = time.perf_counter()
        # H is the SAME on every rank (a knob, not a function of this rank's batch count: every slice costs one
        # count exchange + K all-to-alls, and all ranks must issue the same collectives); a rank with fewer
        # batches than slices ships empty slices.
        H = max(1, int(getattr(ops, "left_slices", 1)))
        cuts = [len(lfk) * h // H for h in range(H + 1)]
        # Issue order of the 2-table x K-group exchanges (RCCL executes them in issue order):
        #   "table": L slice 0 (all groups), L slice 1, ..., then R (all groups) -- the links start after 1/H of the
        #            left split, but group 0 of R travels behind ALL of L;
        #   "group": split everything first, then L g0, R g0, L g1, R g1, ... -- the join of group 0 can start after
        #            2/K of the bytes have moved, at the price of idle links during the splits.
        order = getattr(ops, "exchange_order", "table")
        left = []
        for h in range(H):
            sh = _split(ops, lfk[cuts[h]:cuts[h + 1]], ly[cuts[h]:cuts[h + 1]], world, gbits, lbits, group)
            left.append(sh.ship_all() if order == "table" else sh)
        t1 = time.perf_counter()   # left table split (its exchanges in flight under "table")
        right = _split(ops, rpk, rx, world, gbits, lbits, group)
        if order == "table":
            right.ship_all()
        else:
            for gi in range(K):
                for sh in left:
                    sh.ship(gi)
                right.ship(gi)
        t2 = time.perf_counter()   # right table split, all exchanges issued
        overlap = getattr(ops, "overlap_right_exchange", False)
        n_left = [sum(sum(sh.got[gi]) for sh in left) for gi in range(K)]          # left rows this rank joins, per group
        out = ops.empty_result(sum(n_left)) if hasattr(ops, "empty_result") else None
        pieces, stats, row0 = [], None, 0
        width = H * world
        for gi in range(K):
            RL = [sh.pend[gi].wait() for sh in left]
            if RL[0].is_cuda:
                torch.cuda.current_stream(RL[0].device).synchronize()            # group gi of the left table has arrived
            RR = right.pend[gi].wait()   # RCCL: orders torch's stream behind the transfer, does not block the host
            ready = None
            if overlap and RR.is_cuda:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(RR.device))
            elif RR.is_cuda:
                torch.cuda.current_stream(RR.device).synchronize()
            l_lo, l_hi = _group_tables(left, gi, world, L, width)
            r_lo, r_hi = _group_tables([right], gi, world, L, width)
            kw = {}
            if ready is not None:
                kw["right_ready"] = ready
            if out is not None:
                kw["out"] = tuple(c[row0:row0 + n_left[gi]] for c in out)
            fk, y, x, st = ops.join_presplit(RL if H > 1 else RL[0], l_lo, l_hi, RR, r_lo, r_hi, L, width, g + gbits, **kw)
            pieces.append((fk, y, x))
            row0 += n_left[gi]
            if stats is None:
                stats = dict(st)
            else:
                for key in ("left_rows", "right_rows", "missing_keys", "duplicate_keys"):
                    if key in st:
                        stats[key] = stats.get(key, 0) + st[key]
                if "max_build_rows" in st:
                    stats["max_build_rows"] = max(stats.get("max_build_rows", 0), st["max_build_rows"])
                if "nr_partitions" in st:
                    stats["nr_partitions"] = stats.get("nr_partitions", 0) + st["nr_partitions"]
        t3 = time.perf_counter()
        if out is not None:
            got = sum(int(p[0].numel()) for p in pieces)
            fk, y, x = (c[:got] for c in out) if got == sum(n_left) else (torch.cat([p[i] for p in pieces]) for i in range(3))
        else:
            fk, y, x = (torch.cat([p[i] for p in pieces]) for i in range(3))
        l_in = [sum(sh.sent[gi][p] for sh in left for gi in range(K)) for p in range(world)]
        r_in = [sum(right.sent[gi][p] for gi in range(K)) for p in range(world)]
        recv_left, recv_right = sum(n_left), sum(sum(right.got[gi]) for gi in range(K))
        sent = sum(l_in) + sum(r_in) - l_in[rank] - r_in[rank]
        wall = {"split_left+counts": (t1 - t0) * 1e3, "split_right+counts": (t2 - t1) * 1e3, "exchanges+group_joins": (t3 - t2) * 1e3}
        if timers is not None:
            for key, v in wall.items():
                timers[key] = timers.get(key, 0.0) + v
        return fk, y, x, {"world": world, "stats": stats, "sent_rows": sent, "recv_left": recv_left, "recv_right": recv_right,
                          "path": "fused", "groups": K, "left_slices": H, "order": order, "wall_ms": wall,
                          "backend": dist.get_backend(group)}
    if hasattr(ops, "partition_tuples"):
        # zipped path: one int64 tensor per table; the left table's exchange overlaps the
        # owner split of the right table (RCCL runs on its own stream)
        lt, loff = ops.partition_tuples(lfk, ly, world)
        l_in = [loff[p + 1] - loff[p] for p in range(world)]
        dev = lt.device
        l_out = [int(v) for v in _exchange_counts(torch.tensor(l_in, dtype=torch.int64, device=dev), world, group).cpu()]
        pend_l = _exchange(lt, l_in, l_out, ops.empty_tuples, group, async_op=True)
        rt, roff = ops.partition_tuples(rpk, rx, world)
        r_in = [roff[p + 1] - roff[p] for p in range(world)]
        r_out = [int(v) for v in _exchange_counts(torch.tensor(r_in, dtype=torch.int64, device=dev), world, group).cpu()]
        RR = _exchange(rt, r_in, r_out, ops.empty_tuples, group)
        RL = pend_l.wait()
        if RL.is_cuda:
            torch.cuda.current_stream(RL.device).synchronize()
        fk, y, x, stats = ops.join_tuples(RL, RR, g)
        sent = sum(l_in) + sum(r_in) - l_in[rank] - r_in[rank]
        return fk, y, x, {"world": world, "stats": stats, "sent_rows": sent, "recv_left": sum(l_out), "recv_right": sum(r_out)}
    # (2) group by owner
    lk, lp, loff = ops.partition(lfk, ly, world)
    rk, rp, roff = ops.partition(rpk, rx, world)
    l_in = [loff[p + 1] - loff[p] for p in range(world)]
    r_in = [roff[p + 1] - roff[p] for p in range(world)]
    # (3a) counts
    dev = lk.device
    cnt_in = torch.tensor(l_in + r_in, dtype=torch.int64, device=dev).view(2, world).t().contiguous().view(-1)
    cnt_out = torch.empty_like(cnt_in)
    try:
        dist.all_to_all_single(cnt_out, cnt_in, group=group)
    except (RuntimeError, NotImplementedError):
        if not _pairwise_only(group):
            raise
        gathered = [torch.empty_like(cnt_in) for _ in range(world)]
        dist.all_gather(gathered, cnt_in, group=group)
        rank = dist.get_rank(group)
        cnt_out = torch.stack([gt.view(world, 2)[rank] for gt in gathered]).view(-1)
    co = cnt_out.view(world, 2).cpu()
    l_out, r_out = [int(v) for v in co[:, 0]], [int(v) for v in co[:, 1]]
    # (3b) payload exchange: four columns
    RLk = _exchange(_i32(lk), l_in, l_out, ops.empty, group)
    RLp = _exchange(_i32(lp), l_in, l_out, ops.empty, group)
    RRk = _exchange(_i32(rk), r_in, r_out, ops.empty, group)
    RRp = _exchange(_i32(rp), r_in, r_out, ops.empty, group)
    # (4) local join below the owner bits
    fk, y, x, stats = ops.join([RLk], [RLp], [RRk], [RRp], g)
    sent = sum(l_in) + sum(r_in) - l_in[rank] - r_in[rank]
    return fk, y, x, {"world": world, "stats": stats, "sent_rows": sent, "recv_left": sum(l_out), "recv_right": sum(r_out)}
