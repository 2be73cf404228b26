"""GPU parity of the ring scatter's EXACT mode (csrc/ring_scatter.h k_part_scatter_ring<.., EXACT>: the scatter of the
histogram + scan + scatter pipeline -- build_histogram / prefix_sum / partition_array, dpu/shared/kernels/partition.c:67-341 --
through per-bucket LDS rings): forced at small sizes with ring_exact = 2.  Bit-exact against the oracle's partition, the
write-combining kernel's output and numpy lookups."""
import numpy as np
import pytest

import oracle
from conftest import sort_rows

pytestmark = pytest.mark.gpu
RING = "olap::k_part_scatter_ring<0, false, 512, 16, 1024, 8, 8, 0, true>"


@pytest.fixture(scope="module")
def dev():
    from dpu_olap_amd import Device
    d = Device(0)
    d.set_tuning("ring_exact", 2)
    yield d
    d.set_tuning("ring_exact", 1)
    d.close()


@pytest.mark.parametrize("P", [2, 32, 512])
@pytest.mark.parametrize("space", ["host", "device"])
def test_partition_operator_ring_exact_vs_oracle(dev, P, space):
    """PartitionGpu (one pass of log2 P bits) over ragged batches: per-partition multisets and END offsets equal the oracle's
    (partition.c:296-341, join_dpu.cc:54-66)."""
    from dpu_olap_amd import PartitionGpu
    rng = np.random.default_rng(P)
    sizes = [65536, 1, 0, 40000, 8192, 8191, 300001]
    keys = [rng.integers(0, 1 << 32, size=s, dtype=np.uint32) for s in sizes]
    pay = [rng.integers(0, 1 << 32, size=s, dtype=np.uint32) for s in sizes]
    wrap = (lambda a: a) if space == "host" else dev.to_device
    parts = PartitionGpu(dev, [(wrap(k), wrap(p)) for k, p in zip(keys, pay)], P).Prepare().Run()
    assert dev.phase_kernel("partition-scatter") == RING
    allk, allp = np.concatenate(keys), np.concatenate(pay)
    b = oracle.wang_hash_np(allk) >> np.uint32(32 - int(np.log2(P)))
    assert sum(len(k) for k, _ in parts) == allk.size
    for p in range(P):
        k, v = (np.asarray(a) for a in parts[p])   # (PartitionGpu hands back host arrays)
        ek, ev = allk[b == p], allp[b == p]
        got, exp = sort_rows(k, v), sort_rows(ek, ev)
        np.testing.assert_array_equal(got[0], exp[0])
        np.testing.assert_array_equal(got[1], exp[1])


@pytest.mark.parametrize("obits,gbits,lbits", [(1, 2, 6), (3, 2, 4), (2, 0, 5)])
def test_split_tuples_ring_exact_grouped_order(dev, obits, gbits, lbits):
    """olap_split_tuples_u32 (the multi-GPU split) through the ring kernel: bucket [owner | group | low] lands at
    [group | owner | low]; every run holds exactly the tuples of its bucket."""
    import torch
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rng = np.random.default_rng(obits * 100 + gbits * 10 + lbits)
    sizes = [70001, 0, 12345, 300000, 4096 * 37]
    kk = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]
    pp = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]
    t = lambda a: torch.from_numpy(a.view(np.int32)).to(device)
    grp, goff = ops.split_tuples([t(a) for a in kk], [t(a) for a in pp], obits, gbits, lbits)
    assert dev.phase_kernel("partition-scatter") == RING
    grp = grp.cpu().numpy().view(np.uint64)
    allk, allp = np.concatenate(kk), np.concatenate(pp)
    tup = allk.astype(np.uint64) | (allp.astype(np.uint64) << np.uint64(32))
    bits = obits + gbits + lbits
    bucket = oracle.wang_hash_np(allk) >> np.uint32(32 - bits)
    assert goff[-1] == allk.size
    k = 0
    for g in range(1 << gbits):
        for o in range(1 << obits):
            for l in range(1 << lbits):
                b = (o << (gbits + lbits)) | (g << lbits) | l
                np.testing.assert_array_equal(np.sort(grp[goff[k]:goff[k + 1]]), np.sort(tup[bucket == b]))
                k += 1


@pytest.mark.parametrize("seed", range(10))
def test_join_ring_exact_passes_random_plans(dev, seed):
    """The join with the histogram-free layouts switched off: every pass is histogram + scan + ring scatter (pass 0 from the
    columns, later passes over the segments of the pass before, skewed segments included); rows against numpy."""
    from dpu_olap_amd import JoinGpu
    rng = np.random.default_rng(9100 + seed)
    nr = int(rng.integers(1 << 10, 1 << 19))
    nl = int(rng.integers(1 << 10, 1 << 20))
    bits = int(rng.integers(4, 19))
    maxb = int(rng.choice([5, 7, 9]))
    pk = rng.permutation(1 << 22)[:nr].astype(np.uint32) * np.uint32(1021)
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    if seed % 3 == 0:
        fk[rng.random(nl) < 0.3] = pk[7]     # a hot foreign key: one segment far larger than the others
    y = np.arange(nl, dtype=np.uint32)
    cut = lambda a, n: np.array_split(a, n)
    nbl, nbr = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    dev.join_configure(bits, maxb)
    dev.set_tuning("capacity_mode", 0)
    dev.set_tuning("claim_mode", 0)
    try:
        j = JoinGpu(dev, list(zip(cut(fk, nbl), cut(y, nbl))), list(zip(cut(pk, nbr), cut(x, nbr)))).Prepare()
        out = j.Run()
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("capacity_mode", 1)
        dev.set_tuning("claim_mode", 1)
    assert "k_part_scatter_ring<" in dev.phase_kernel("partition-scatter") and dev.phase_kernel("partition-scatter").endswith("0, true>")
    assert j.stats["missing_keys"] == 0 and out["fk"].size == nl
    np.testing.assert_array_equal(np.sort(out["y"]), y)
    np.testing.assert_array_equal(fk[out["y"]], out["fk"])
    order = np.argsort(pk, kind="stable")
    np.testing.assert_array_equal(x[order][np.searchsorted(pk[order], out["fk"])], out["x"])


@pytest.mark.parametrize("payload", [True, False])
def test_partition_ring_exact_whole_tiles_straight_line_loader(dev, payload):
    """Device-resident batches that are whole 4096-row tiles in 16-byte aligned buffers take the straight-line loader (FULL = 1 with
    a payload column, 2 without: the payload is then the global row index); same contract."""
    import torch
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rng = np.random.default_rng(77)
    sizes = [65536, 4096 * 3, 8192, 4096 * 50]
    kk = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]
    pp = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]
    t = lambda a: torch.from_numpy(a.view(np.int32)).to(device)
    P = 256
    keys = [t(a) for a in kk]
    if payload:
        tup, off = ops.partition_tuples(keys, [t(a) for a in pp], P)
        want = "1, true>"
    else:
        rows = [int(k.numel()) for k in keys]
        n = sum(rows)
        ok, op = ops.empty(n), ops.empty(n)
        torch.cuda.current_stream(device).synchronize()
        off = dev.partition_u32([k.data_ptr() for k in keys], None, rows, 1, P, ok.data_ptr(), op.data_ptr(), n, 1)   # MEM_DEVICE = 1
        tup = (ok.cpu().numpy().view(np.uint32).astype(np.uint64) | (op.cpu().numpy().view(np.uint32).astype(np.uint64) << np.uint64(32)))
        want = "2, true>"
    assert dev.phase_kernel("partition-scatter").endswith(want), dev.phase_kernel("partition-scatter")
    tup = tup.cpu().numpy().view(np.uint64) if payload else tup
    allk = np.concatenate(kk)
    allp = np.concatenate(pp) if payload else np.arange(allk.size, dtype=np.uint32)
    exp = allk.astype(np.uint64) | (allp.astype(np.uint64) << np.uint64(32))
    bucket = oracle.wang_hash_np(allk) >> np.uint32(32 - 8)
    assert off[-1] == allk.size
    for p in range(P):
        np.testing.assert_array_equal(np.sort(tup[off[p]:off[p + 1]]), np.sort(exp[bucket == p]))


@pytest.mark.parametrize("log2_right", [13, 18])
def test_join_large_probe_side_small_build_side_takes_512_partitions(dev, log2_right):
    """The fact x dimension shape under the AUTOMATIC plan: a probe side of >= 2^26 rows with a small build side gets 9 radix bits --
    512 partitions, the exact ring scatter for the probe side, probe slices of 64 Ki rows to fill the chip -- and every left row
    comes back with its right payload (x = gen(fk) since pk = row index; join_benchmark.cc:170-171 shape)."""
    import dpu_olap_amd as o
    from dpu_olap_amd import datagen as dg
    nl, nr, rows = 1 << 26, 1 << log2_right, 2 << 20
    cols = {k: dev.empty(n, "uint32") for k, n in (("fk", nl), ("y", nl), ("exp", nl), ("pk", nr), ("x", nr))}
    dev.gen_fk_u32(cols["fk"].ptr, nl, dg.SEED, dg.STREAM_FK, 0, 1 << 62, nr)     # uniform in [0, nr)
    dev.gen_uniform_u32(cols["y"].ptr, nl, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, nr, 0)
    dev.gen_uniform_u32(cols["x"].ptr, nr, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, nl, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, nl)
    batch = lambda c, b, n: o.DeviceArray(cols[c].ptr + 4 * b * n, n, "uint32")
    j = o.JoinGpu(dev, [(batch("fk", b, rows), batch("y", b, rows)) for b in range(nl // rows)], [(batch("pk", 0, nr), batch("x", 0, nr))]).Prepare()
    out = j.Run()
    assert out["fk"].rows == nl and j.stats["missing_keys"] == 0
    assert j.stats["radix_bits"][:1] == [9] and j.stats["passes"] == 1, j.stats
    assert "k_part_scatter_ring<" in dev.phase_kernel("partition-scatter")
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, nl) == expect
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))
