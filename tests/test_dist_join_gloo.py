"""Multi-rank join host logic (dpu_olap_amd/dist_join.py) on CPU: world_size 2 and 4 with
the gloo backend.  The local operators are oracle-backed here (tests may use the oracle);
on a GPU node the same function runs with GpuLocalOps over RCCL.  Checks: every rank only
ends up with keys it owns (top log2(G) bits of the Wang hash), the union of the shards equals
the Arrow golden join, counts and exchange bookkeeping are consistent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import GOLDEN_DIR, sort_rows


class OracleLocalOps:
    """partition / join on CPU tensors through the oracle (int32 tensors carry uint32 bits)."""

    def empty(self, n):
        return torch.empty(n, dtype=torch.int32)

    @staticmethod
    def _np(t):
        return t.contiguous().numpy().view(np.uint32)

    def partition(self, keys, payload, nr_partitions):
        allk = np.concatenate([self._np(k) for k in keys]) if keys else np.empty(0, np.uint32)
        allp = np.concatenate([self._np(p) for p in payload]) if payload else np.empty(0, np.uint32)
        ok, orow, hist = oracle.partition(allk, nr_partitions)
        offs = [0] + [int(h) for h in hist]
        return (torch.from_numpy(ok.view(np.int32).copy()), torch.from_numpy(allp[orow].view(np.int32).copy()), offs)

    def join(self, lfk, ly, rpk, rx, skip_bits=0):
        fk, y, x = oracle.join([self._np(t) for t in lfk], [self._np(t) for t in ly], [self._np(t) for t in rpk],
                               [self._np(t) for t in rx], nr_partitions=4)
        conv = lambda a: torch.from_numpy(a.view(np.int32).copy())
        return conv(fk), conv(y), conv(x), {"skip_bits": skip_bits}


class OracleTupleOps(OracleLocalOps):
    """Adds the zipped-tuple entry points, so distributed_join takes its tuple path (two
    int64 exchanges, the left one asynchronous) like GpuLocalOps does."""

    def empty_tuples(self, n):
        return torch.empty(n, dtype=torch.int64)

    def partition_tuples(self, keys, payload, nr_partitions):
        ok, op, offs = self.partition(keys, payload, nr_partitions)
        z = self._np(ok).astype(np.uint64) | (self._np(op).astype(np.uint64) << np.uint64(32))
        return torch.from_numpy(z.view(np.int64).copy()), offs

    def join_tuples(self, lt, rt, skip_bits=0):
        unz = lambda t: (torch.from_numpy((t.numpy().view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)),
                         torch.from_numpy((t.numpy().view(np.uint64) >> np.uint64(32)).astype(np.uint32).view(np.int32)))
        lk, lp = unz(lt.contiguous())
        rk, rp = unz(rt.contiguous())
        return self.join([lk], [lp], [rk], [rp], skip_bits)


class OracleFusedOps(OracleTupleOps):
    """Adds the grouped split and the pre-split join, so distributed_join takes its fused path (owner
    split = first radix pass, tables shipped as sub-bucket groups, one join per group).  Checks what it
    is handed: every key of a call shares the leading skip bits (owner = this rank, then the group), and
    chunk (j, c) holds only keys whose next bits are j."""

    left_slices = 2      # like GpuLocalOps: the left table is shipped in two input slices
    exchange_groups = 2  # and every table as sub-bucket groups that are joined one by one

    def split_tuples(self, keys, payload, owner_bits, group_bits, low_bits):
        bits = owner_bits + group_bits + low_bits
        t, off = self.partition_tuples(keys, payload, 1 << bits)             # natural order [owner | group | low]
        z = t.numpy()
        parts, sizes = [], []
        for g in range(1 << group_bits):                                     # emitted as [group | owner | low]
            for o in range(1 << owner_bits):
                for l in range(1 << low_bits):
                    b = (o << (group_bits + low_bits)) | (g << low_bits) | l
                    parts.append(z[off[b]:off[b + 1]])
                    sizes.append(off[b + 1] - off[b])
        out = torch.from_numpy(np.concatenate(parts).copy()) if parts else t
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + sz)
        return out, offs

    def join_presplit(self, lt, l_lo, l_hi, rt, r_lo, r_hi, n_segments, chunks, skip_bits=0, **_kw):
        rank, world = dist.get_rank(), dist.get_world_size()
        obits, sbits = world.bit_length() - 1, n_segments.bit_length() - 1
        lead = set()

        def gather(t, lo, hi):
            if isinstance(t, (list, tuple)):       # several receive buffers: chunk indices count through their concatenation
                t = torch.cat(list(t)) if len(t) else torch.empty(0, dtype=torch.int64)
            z = t.contiguous().numpy().view(np.uint64)
            assert len(lo) == len(hi) == n_segments * chunks
            assert sum(h - l for l, h in zip(lo, hi)) == z.size, "chunks must tile the received buffers"
            parts = []
            for j in range(n_segments):
                for c in range(chunks):
                    ch = z[lo[j * chunks + c]:hi[j * chunks + c]]
                    h = oracle.wang_hash_np((ch & np.uint64(0xFFFFFFFF)).astype(np.uint32))
                    if ch.size:
                        assert ((h >> np.uint32(32 - obits)) == rank).all(), "a rank received keys it does not own"
                        lead.update(np.unique(h >> np.uint32(32 - skip_bits)).tolist())
                        if sbits:
                            sub = (h << np.uint32(skip_bits)) >> np.uint32(32 - sbits)
                            assert (sub == j).all(), "chunk holds keys of another sub-bucket"
                    parts.append(ch)
            return torch.from_numpy(np.concatenate(parts).view(np.int64).copy()) if parts else torch.empty(0, dtype=torch.int64)

        L, R = gather(lt, l_lo, l_hi), gather(rt, r_lo, r_hi)
        assert len(lead) <= 1, "one call = one (owner, group): all keys share the skipped hash bits"
        if L.numel() == 0:
            e = torch.empty(0, dtype=torch.int32)
            return e, e, e, {"skip_bits": skip_bits, "presplit": (n_segments, chunks)}
        fk, y, x, st = self.join_tuples(L, R, skip_bits)
        st["presplit"] = (n_segments, chunks)
        return fk, y, x, st


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, outdir, tuple_path, order="table", uneven=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dpu_olap_amd.dist_join import distributed_join
        g = np.load(os.path.join(GOLDEN_DIR, "arrow_golden.npz"))
        fk, y, pk, x = (g[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))
        # every rank holds an interleaved slice of the rows, cut into two ragged batches
        sl = lambda a: np.ascontiguousarray(a[rank::world])
        cut = lambda a: [a[: a.size // 3], a[a.size // 3:]]
        if uneven:   # the ranks hold DIFFERENT numbers of batches (1, 3, 0, 2, ...): the collectives issued must not depend on that
            nb = [1, 3, 0, 2][rank % 4]
            cut = lambda a: ([a[a.size * i // nb: a.size * (i + 1) // nb] for i in range(nb)] if nb else [])
            if nb == 0:  # this rank's rows go to its neighbour instead
                sl = lambda a: np.empty(0, dtype=a.dtype)
            elif [1, 3, 0, 2][(rank + 1) % 4] == 0 and rank + 1 < world:
                sl = lambda a: np.ascontiguousarray(np.concatenate([a[rank::world], a[rank + 1::world]]))
        t = lambda a: torch.from_numpy(a.view(np.int32).copy())
        left = [(t(k), t(p)) for k, p in zip(cut(sl(fk)), cut(sl(y)))]
        right = [(t(k), t(p)) for k, p in zip(cut(sl(pk)), cut(sl(x)))]
        ops = {"fused": OracleFusedOps, "tuple": OracleTupleOps, "cols": OracleLocalOps}[tuple_path]()
        ops.exchange_order = order
        ofk, oy, ox, info = distributed_join(ops, left, right)
        if tuple_path == "fused":
            assert info["path"] == "fused" and info["groups"] == 2 and info["left_slices"] == 2 and info["order"] == order
            assert info["stats"]["presplit"] == (512 // world // 2, 2 * world)
        u = lambda tt: tt.contiguous().numpy().view(np.uint32)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), fk=u(ofk), y=u(oy), x=u(ox),
                 sent=np.array([info["sent_rows"]]), skip=np.array([info["stats"]["skip_bits"]]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tuple_path", [(2, "fused"), (4, "fused"), (8, "fused"), (2, "tuple"), (2, "cols"), (4, "tuple")])
@pytest.mark.parametrize("case", ["a", "b"])
def test_distributed_join_gloo(tmp_path, world, case, tuple_path):
    if world == 8 and case == "a":
        pytest.skip("the 8-rank run is exercised once (case b)")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path), tuple_path), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN_DIR, "arrow_golden.npz"))
    shards = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    bits = world.bit_length() - 1
    total_sent = 0
    for r, s in enumerate(shards):
        owner = oracle.wang_hash_np(s["fk"]) >> np.uint32(32 - bits)
        assert (owner == r).all(), "a rank received keys it does not own"
        assert int(s["skip"][0]) in (bits, bits + 1)   # fused path: owner bits + one group bit
        total_sent += int(s["sent"][0])
    fk = np.concatenate([s["fk"] for s in shards])
    y = np.concatenate([s["y"] for s in shards])
    x = np.concatenate([s["x"] for s in shards])
    fk, y, x = sort_rows(fk, y, x)
    np.testing.assert_array_equal(fk, g[f"join_{case}_out_fk"])
    np.testing.assert_array_equal(y, g[f"join_{case}_out_y"])
    np.testing.assert_array_equal(x, g[f"join_{case}_out_x"])
    n_all = g[f"join_{case}_fk"].size + g[f"join_{case}_pk"].size
    assert 0 < total_sent < n_all  # roughly (G-1)/G of the tuples change rank
    assert abs(total_sent / n_all - (world - 1) / world) < 0.1


def _check_shards(tmp_path, world, case):
    g = np.load(os.path.join(GOLDEN_DIR, "arrow_golden.npz"))
    shards = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    bits = world.bit_length() - 1
    for r, s in enumerate(shards):
        if bits:
            assert (oracle.wang_hash_np(s["fk"]) >> np.uint32(32 - bits) == r).all(), "a rank received keys it does not own"
    fk, y, x = sort_rows(*(np.concatenate([s[k] for s in shards]) for k in ("fk", "y", "x")))
    np.testing.assert_array_equal(fk, g[f"join_{case}_out_fk"])
    np.testing.assert_array_equal(y, g[f"join_{case}_out_y"])
    np.testing.assert_array_equal(x, g[f"join_{case}_out_x"])


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_join_group_major_issue_order(tmp_path, world):
    """exchange_order = "group": both tables are split first, then the exchanges go out as L g0, R g0, L g1, R g1."""
    mp.spawn(_worker, args=(world, _free_port(), "b", str(tmp_path), "fused", "group"), nprocs=world, join=True)
    _check_shards(tmp_path, world, "b")


@pytest.mark.parametrize("order", ["table", "group"])
def test_distributed_join_uneven_batch_counts(tmp_path, order):
    """Ranks holding 1, 3, 0 and 2 batches: the number of left slices (and with it the number of collectives every
    rank issues) is a knob, not a function of the local batch count -- a rank short of batches ships empty slices."""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), "a", str(tmp_path), "fused", order, True), nprocs=world, join=True)
    _check_shards(tmp_path, world, "a")


def _worker_world1_forced(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from dpu_olap_amd.dist_join import distributed_join
        g = np.load(os.path.join(GOLDEN_DIR, "arrow_golden.npz"))
        t = lambda a: torch.from_numpy(a.view(np.int32).copy())
        ops = OracleFusedOps()
        ops.force_exchange = True
        ofk, oy, ox, info = distributed_join(ops, [(t(g["join_a_fk"]), t(g["join_a_y"]))], [(t(g["join_a_pk"]), t(g["join_a_x"]))])
        assert info["path"] == "fused" and info["world"] == 1 and info["left_slices"] == 2 and info["backend"] == "gloo"
        u = lambda tt: tt.contiguous().numpy().view(np.uint32)
        np.savez(os.path.join(outdir, "rank0.npz"), fk=u(ofk), y=u(oy), x=u(ox))
    finally:
        dist.destroy_process_group()


def test_forced_exchange_at_world_size_one(tmp_path):
    """ops.force_exchange: a process group of ONE rank takes the split -> all-to-all -> pre-split join route (what the
    GPU test runs over RCCL on a one-GPU box)."""
    mp.spawn(_worker_world1_forced, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    _check_shards(tmp_path, 1, "a")


def test_world_size_one_is_the_plain_join():
    from dpu_olap_amd.dist_join import distributed_join
    g = np.load(os.path.join(GOLDEN_DIR, "arrow_golden.npz"))
    t = lambda a: torch.from_numpy(a.view(np.int32).copy())
    ofk, oy, ox, info = distributed_join(OracleLocalOps(), [(t(g["join_a_fk"]), t(g["join_a_y"]))],
                                         [(t(g["join_a_pk"]), t(g["join_a_x"]))])
    u = lambda tt: tt.numpy().view(np.uint32)
    fk, y, x = sort_rows(u(ofk), u(oy), u(ox))
    np.testing.assert_array_equal(x, g["join_a_out_x"])
    assert info["world"] == 1
