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

import time
from typing import Sequence

import torch
import torch.distributed as dist

from . import capi


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.view(torch.int32)


class GpuLocalOps:
    """Local partition / join through libolap_gpu.so on torch CUDA tensors (int32 tensors
    carry the uint32 bit patterns; torch is only the allocator and the RCCL front end)."""

    overlap_right_exchange = True   # fused path: the left table's radix pass runs under the right table's exchange
    left_slices = 2                 # fused path: the left table is split and shipped in this many input slices, so the
                                    # links start working after 1/left_slices of the split instead of all of it

    def __init__(self, ctx: capi.Context, device: torch.device):
        self.ctx = ctx
        self.device = device

    def empty(self, n: int) -> torch.Tensor:
        return torch.empty(max(n, 1), dtype=torch.int32, device=self.device)[:n]

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

    def join_tuples(self, lt: torch.Tensor, rt: torch.Tensor, skip_bits: int = 0):
        n = int(lt.numel())
        out = [self.empty(n) for _ in range(3)]
        torch.cuda.current_stream(self.device).synchronize()  # inputs come from torch's stream; do not wait for RCCL's
        self.ctx.join_set_hash_skip_bits(skip_bits)
        try:
            got, stats = self.ctx.join_tuples_u32(lt.data_ptr(), n, rt.data_ptr(), int(rt.numel()), out[0].data_ptr(),
                                                  out[1].data_ptr(), out[2].data_ptr(), n)
        finally:
            self.ctx.join_set_hash_skip_bits(0)
        return out[0][:got], out[1][:got], out[2][:got], stats

    def join_presplit(self, lt: torch.Tensor, l_lo, l_hi, rt: torch.Tensor, r_lo, r_hi, n_segments: int, chunks: int,
                      skip_bits: int = 0, right_ready: "torch.cuda.Event | None" = None):
        """Local join of exchanged tuples that are still grouped (sender, sub-bucket): the first
        radix pass reads the chunks in place (olap_join_presplit_u32).  right_ready: an event
        recorded behind the right table's exchange -- the left table's pass then runs while the
        right table is still arriving."""
        l_ptr, l_span, l_lo, l_hi, n = self._span(lt, l_lo, l_hi)
        r_ptr, r_span, r_lo, r_hi, _ = self._span(rt, r_lo, r_hi)
        out = [self.empty(n) for _ in range(3)]
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

    def join(self, lfk, ly, rpk, rx, skip_bits: int = 0):
        lrows = [int(t.numel()) for t in lfk]
        rrows = [int(t.numel()) for t in rpk]
        n = sum(lrows)
        out = [self.empty(n) for _ in range(3)]
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


def _exchange(send: torch.Tensor, in_splits, out_splits, empty, group, async_op: bool = False):
    """Personalised all-to-all of a 1-D tensor.  RCCL/NCCL: all_to_all_single (optionally
    asynchronous, so it overlaps the partition kernels of the other table); backends
    without it (gloo on some builds): pairwise isend/irecv."""
    recv = empty(int(sum(out_splits)))
    try:
        work = dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits),
                                      group=group, async_op=async_op)
        return _Pending(recv, work) if async_op else recv
    except (RuntimeError, NotImplementedError):
        pass
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


def _spread(lo, hi, S: int, world: int, H: int):
    """Re-index a (segment, sender) chunk table as (segment, H x world) with empty chunks in the extra places."""
    if H == 1:
        return lo, hi
    nlo, nhi = [0] * (S * H * world), [0] * (S * H * world)
    for j in range(S):
        for q in range(world):
            nlo[j * H * world + q], nhi[j * H * world + q] = lo[j * world + q], hi[j * world + q]
        for k in range(world, H * world):
            nlo[j * H * world + k] = nhi[j * H * world + k] = hi[j * world + world - 1]
    return nlo, nhi


def _fused_side(ops, keys, payload, world: int, S: int, group, async_op: bool):
    """Split one table into world x S buckets (owner = top bits, sub-bucket = next bits), ship
    every owner its S runs as one message, and return (pending tuples, chunk lo, chunk hi,
    rows sent, rows received)."""
    t, off = ops.partition_tuples(keys, payload, world * S)
    cnt_in = torch.tensor([off[k + 1] - off[k] for k in range(world * S)], dtype=torch.int64, device=t.device)
    cnt_out = _exchange_counts(cnt_in, world, group).view(world, S).cpu()
    in_splits = [int(v) for v in cnt_in.view(world, S).sum(1).cpu()]
    out_splits = [int(v) for v in cnt_out.sum(1)]
    pend = _exchange(t, in_splits, out_splits, ops.empty_tuples, group, async_op=True)
    if not async_op:
        pend = _Pending(pend.wait())
    lo, hi = _chunk_tables(cnt_out)
    return pend, lo, hi, in_splits, out_splits


def distributed_join(ops, left_batches: Sequence, right_batches: Sequence, group=None, timers: dict | None = None):
    """left_batches / right_batches: this rank's batches as (key, payload) tensor pairs.
    Returns (fk, y, x, info): this rank's shard of the join result (rows whose key it owns)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lfk, ly = [b[0] for b in left_batches], [b[1] for b in left_batches]
    rpk, rx = [b[0] for b in right_batches], [b[1] for b in right_batches]
    if world == 1:
        fk, y, x, stats = ops.join(lfk, ly, rpk, rx, 0)
        return fk, y, x, {"world": 1, "stats": stats, "sent_rows": 0}
    if world & (world - 1):
        raise ValueError("the owner split uses the top log2(G) hash bits: world size must be a power of two")
    g = world.bit_length() - 1
    rank = dist.get_rank(group)
    if hasattr(ops, "join_presplit") and world <= (1 << (FUSED_SPLIT_BITS - 1)) and getattr(ops, "fused", True):
        # fused path: the owner split IS the first radix pass (world x S = 512 buckets); each
        # owner receives, per sender, its S sub-bucket runs, and its next pass reads them in
        # place.  The left table's exchange overlaps the split of the right table.
        S = (1 << FUSED_SPLIT_BITS) // world
        t0 = time.perf_counter()
        # the left table goes out in H input slices: slice h's exchange runs while slice h + 1 is being split
        H = max(1, min(int(getattr(ops, "left_slices", 1)), len(lfk)))
        cuts = [len(lfk) * h // H for h in range(H + 1)]
        sides = [_fused_side(ops, lfk[cuts[h]:cuts[h + 1]], ly[cuts[h]:cuts[h + 1]], world, S, group, async_op=True) for h in range(H)]
        l_in = [sum(sd[3][p] for sd in sides) for p in range(world)]
        l_out = [sum(sd[4][p] for sd in sides) for p in range(world)]
        # segment j = chunks (slice h, sender q), numbered through the concatenation of the H receive buffers
        l_lo, l_hi, v = [0] * (S * H * world), [0] * (S * H * world), 0
        for h, sd in enumerate(sides):
            for j in range(S):
                for q in range(world):
                    k = j * (H * world) + h * world + q
                    l_lo[k], l_hi[k] = v + sd[1][j * world + q], v + sd[2][j * world + q]
            v += sum(sd[4])
        t1 = time.perf_counter()   # left split + counts done, left tuples in flight
        overlap = getattr(ops, "overlap_right_exchange", False)
        pend_r, r_lo, r_hi, r_in, r_out = _fused_side(ops, rpk, rx, world, S, group, async_op=overlap)
        RL = [sd[0].wait() for sd in sides]
        if RL[0].is_cuda:
            torch.cuda.current_stream(RL[0].device).synchronize()   # the left table has arrived
        ready = None
        RR = pend_r.wait()          # RCCL: orders torch's stream behind the transfer, does not block the host
        if overlap and RR.is_cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(RR.device))
        t2 = time.perf_counter()   # right split done; exchanges done (or, overlapped, the right one still in flight)
        r_lo, r_hi = _spread(r_lo, r_hi, S, world, H)   # both sides use H x world chunks per segment (the right one: empty extras)
        RLx = RL if H > 1 else RL[0]
        fk, y, x, stats = (ops.join_presplit(RLx, l_lo, l_hi, RR, r_lo, r_hi, S, H * world, g, right_ready=ready) if ready is not None
                           else ops.join_presplit(RLx, l_lo, l_hi, RR, r_lo, r_hi, S, H * world, g))
        t3 = time.perf_counter()
        sent = sum(l_in) + sum(r_in) - l_in[rank] - r_in[rank]
        wall = {"split_left+counts": (t1 - t0) * 1e3, "split_right+exchanges": (t2 - t1) * 1e3, "local_join": (t3 - t2) * 1e3}
        if timers is not None:
            for k, v in wall.items():
                timers[k] = timers.get(k, 0.0) + v
        return fk, y, x, {"world": world, "stats": stats, "sent_rows": sent, "recv_left": sum(l_out),
                          "recv_right": sum(r_out), "path": "fused", "wall_ms": wall}
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
