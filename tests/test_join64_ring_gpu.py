"""GPU parity tests of the histogram-free plan of olap_join_u64 (csrc/ring64.h: two ring-scatter passes on 12-byte
tuples, capacity-layout probe; HT_64BIT_KEYS, dpu/shared/hashtable/hashtable.h:13-18) against the Arrow golden rows, the
oracle's join (oracle/olap_oracle.c oracle_join_u64) and numpy lookups.  Bit-exact (integer work).  The plan is forced
with the tuning knob join64_ring = 2 at sizes the automatic choice would leave to the exact-layout pipeline."""
import numpy as np
import pytest

import oracle
from conftest import sort_rows, split_batches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dpu_olap_amd import Device
    d = Device(0)
    yield d
    d.close()


def _ops():
    import dpu_olap_amd
    return dpu_olap_amd


def _ring(dev, bits, fn):
    dev.join_configure(bits, 0)
    dev.set_tuning("join64_ring", 2)
    try:
        return fn()
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("join64_ring", 1)


def _used_ring(dev):
    return dev.phase_kernel("join-probe") == "olap::k64_probe12" and "k64_ring<true" in dev.phase_kernel("partition-scatter-claim") \
        and "k64_ring<false" in dev.phase_kernel("partition-scatter-capacity")


@pytest.mark.parametrize("case", ["c", "d"])
@pytest.mark.parametrize("bits,space", [(8, "host"), (10, "device"), (13, "host"), (18, "device")])
def test_join_u64_ring_vs_arrow_golden(dev, golden, case, bits, space):
    """The golden uint64 / int64 joins (keys 0, 2^64 - 1, keys that differ only above bit 32, negatives) through two ring
    passes of 4+4 .. 9+9 bits."""
    o = _ops()
    fk, y, pk, x = (golden[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))
    wrap = (lambda a: a) if space == "host" else dev.to_device
    lb = [(wrap(k), wrap(v)) for k, v in zip(split_batches(fk, golden[f"join_{case}_left_rows"]), split_batches(y, golden[f"join_{case}_left_rows"]))]
    rb = [(wrap(k), wrap(v)) for k, v in zip(split_batches(pk, golden[f"join_{case}_right_rows"]), split_batches(x, golden[f"join_{case}_right_rows"]))]

    def run():
        j = o.JoinGpu(dev, lb, rb).Prepare()
        return j, j.Run()
    j, out = _ring(dev, bits, run)
    assert _used_ring(dev) and dev.get_counter("join_attempts") == 1
    get = (lambda a: a) if space == "host" else dev.to_host
    got = sort_rows(get(out["fk"]).view(fk.dtype), get(out["y"]), get(out["x"]))
    assert j.stats["missing_keys"] == 0 and j.stats["left_rows"] == fk.size and j.stats["passes"] == 2 and sum(j.stats["radix_bits"]) == bits
    for g, k in zip(got, ("fk", "y", "x")):
        np.testing.assert_array_equal(g, golden[f"join_{case}_out_{k}"])


@pytest.mark.parametrize("seed", range(12))
def test_join_u64_ring_random_shapes_vs_oracle(dev, seed):
    """Seeded random shapes: ragged / empty / one-row batches, unaligned slices, 8..18 radix bits, payload columns or
    selection vectors, missing keys in inner-join mode, a duplicate build key; rows against the oracle / numpy."""
    o = _ops()
    rng = np.random.default_rng(6400 + seed)
    nr = int(rng.integers(1 << 12, 1 << 20))
    nl = int(nr * rng.uniform(0.25, 2.0))   # (a foreign key repeated ~20 times already overruns regions sized for Poisson(1): that is the fallback test)
    bits = int(rng.integers(8, 19))
    while (nr >> bits) > 1536:   # (a build partition must fit the 4096-slot LDS table at load <= 0.75)
        bits += 1
    pk = (rng.permutation(nr).astype(np.uint64) << np.uint64(40)) ^ rng.integers(0, 1 << 40, size=nr, dtype=np.uint64)  # unique: the top bits are a permutation
    if seed % 3 == 0:   # 0, the LDS table's empty marker and the sign bit; any accidental repeat of them is dropped
        pk[:3] = [0, 0xFFFFFFFFFFFFFFFF, 1 << 63]
        _, first = np.unique(pk, return_index=True)
        pk = pk[np.sort(first)]
        nr = pk.size
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    y = rng.integers(0, 1 << 32, size=nl, dtype=np.uint32)
    n_missing = 0
    if seed % 4 == 1:
        miss = rng.choice(nl, size=37, replace=False)
        fk[miss] = np.uint64(0x0123456789ABCDEF) + np.arange(37, dtype=np.uint64)
        n_missing = int((~np.isin(fk[miss], pk)).sum())

    def cut(a, parts, off):
        edges = np.sort(rng.integers(0, a.size + 1, size=parts - 1))
        sizes = np.diff(np.concatenate([[0], edges, [a.size]]))
        if off:   # unaligned slices: every batch is a view one element into its own buffer
            return [np.concatenate([a[:1], b])[1:] for b in split_batches(a, sizes)], sizes
        return split_batches(a, sizes), sizes
    parts = int(rng.integers(1, 9))
    off = seed % 2 == 1
    st = rng.bit_generator.state
    lfk, _ = cut(fk, parts, off)
    rng.bit_generator.state = st
    ly, _ = cut(y, parts, off)
    st = rng.bit_generator.state
    rpk, _ = cut(pk, parts, False)
    rng.bit_generator.state = st
    rx, _ = cut(x, parts, False)
    select = seed % 5 == 2
    drop = n_missing > 0

    def run():
        if select:
            j = o.JoinGpu(dev, [a for a in lfk], [a for a in rpk], select=True, drop_unmatched=drop).Prepare()
        else:
            j = o.JoinGpu(dev, list(zip(lfk, ly)), list(zip(rpk, rx)), drop_unmatched=drop).Prepare()
        return j, j.Run()
    j, out = _ring(dev, bits, run)
    assert _used_ring(dev), (dev.phase_kernel("join-probe"), j.stats)
    assert j.stats["passes"] == 2 and sum(j.stats["radix_bits"]) == bits and j.stats["missing_keys"] == n_missing
    keep = np.isin(fk, pk) if drop else np.ones(nl, dtype=bool)
    assert out["fk"].size == int(keep.sum())
    if select:
        np.testing.assert_array_equal(pk[out["x"]], out["fk"])
        np.testing.assert_array_equal(fk[out["y"]], out["fk"])
        np.testing.assert_array_equal(np.sort(out["y"]), np.flatnonzero(keep).astype(np.uint32))
    else:
        order = np.argsort(pk, kind="stable")
        pos = np.searchsorted(pk[order], fk[keep])
        exp = sort_rows(fk[keep], y[keep], x[order][pos])
        for g, e in zip(sort_rows(out["fk"], out["y"], out["x"]), exp):
            np.testing.assert_array_equal(g, e)
    if seed == 0:   # the same rows from the oracle's own composition (same hash, same table semantics)
        e2 = sort_rows(*oracle.join_u64(lfk, ly, rpk, rx, nr_partitions=64))
        for g, e in zip(sort_rows(out["fk"], out["y"], out["x"]), e2):
            np.testing.assert_array_equal(g, e)


def test_join_u64_ring_partition_major_and_duplicates(dev):
    """2^22 x 2^22 rows, 6 + 6 bits: the result is partition-major (bucket ids = top bits of wang(hash64) never decrease), a
    duplicate build key overwrites and is counted (hashtable.c:133)."""
    o = _ops()
    rng = np.random.default_rng(66)
    n = 1 << 22
    pk = (rng.permutation(n).astype(np.uint64) << np.uint64(21)) ^ rng.integers(0, 1 << 21, size=n, dtype=np.uint64)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    fk = pk[rng.integers(0, n, size=n)]
    y = np.arange(n, dtype=np.uint32)
    pk2, x2 = np.concatenate([pk, pk[5:7]]), np.concatenate([x, x[5:7]])

    def run():
        j = o.JoinGpu(dev, [(fk, y)], [(pk2, x2)]).Prepare()
        return j, j.Run()
    j, out = _ring(dev, 12, run)
    assert _used_ring(dev) and j.stats["duplicate_keys"] == 2 and j.stats["missing_keys"] == 0 and j.stats["max_build_rows"] <= 3072
    h = ((out["fk"] >> np.uint64(33)) ^ out["fk"] ^ (out["fk"] << np.uint64(11))).astype(np.uint32)
    part = oracle.wang_hash_np(h) >> np.uint32(32 - 12)
    assert (np.diff(part.astype(np.int64)) >= 0).all()
    np.testing.assert_array_equal(np.sort(out["y"]), y)
    np.testing.assert_array_equal(fk[out["y"]], out["fk"])
    order = np.argsort(pk, kind="stable")
    np.testing.assert_array_equal(x[order][np.searchsorted(pk[order], out["fk"])], out["x"])


def test_join_u64_ring_hot_key_falls_back_to_exact(dev):
    """A hot foreign key (30 % of the left rows) overruns its regions: the histogram-free passes report it, the join is redone
    with the exact layout (join_attempts == 2) and the rows are right; a hot key inside one 2048-tuple tile exercises the
    rings' overflow path (direct stores) on the way."""
    o = _ops()
    rng = np.random.default_rng(67)
    nr, nl = 1 << 18, 1 << 21
    pk = (rng.permutation(nr).astype(np.uint64) << np.uint64(30)) | rng.integers(0, 1 << 30, size=nr, dtype=np.uint64)
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    fk[rng.random(nl) < 0.3] = pk[123]
    y = np.arange(nl, dtype=np.uint32)

    def run():
        j = o.JoinGpu(dev, list(zip(np.array_split(fk, 3), np.array_split(y, 3))), [(pk, x)]).Prepare()
        return j, j.Run()
    j, out = _ring(dev, 10, run)
    assert dev.get_counter("join_attempts") == 2 and dev.get_counter("join_region_overflows") > 0
    assert dev.phase_kernel("join-probe") == "olap::k64_probe"
    np.testing.assert_array_equal(np.sort(out["y"]), y)
    np.testing.assert_array_equal(fk[out["y"]], out["fk"])
    order = np.argsort(pk, kind="stable")
    np.testing.assert_array_equal(x[order][np.searchsorted(pk[order], out["fk"])], out["x"])


@pytest.mark.parametrize("block", [64, 256, 1 << 13])
def test_join_u64_ring_bucket_bursts_overflow_the_rings(dev, block):
    """Uniform keys that ARRIVE in bursts: the left rows are sorted by their partition hash and dealt back in blocks of
    `block` rows, so a tile brings one bucket far more than its LDS ring holds (21 tuples at 512 buckets) -- in both passes --
    while the regions fill evenly.  block = 64 / 256: the surplus goes straight to the allotment in use and the next one (two
    ranges); 2^13: whole tiles for one bucket, a block of allotments claimed on the spot.  One attempt, rows right."""
    o = _ops()
    rng = np.random.default_rng(680 + block)
    nr, nl = 1 << 19, 1 << 21
    pk = (rng.permutation(nr).astype(np.uint64) << np.uint64(33)) | rng.integers(0, 1 << 33, size=nr, dtype=np.uint64)
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    h = oracle.wang_hash_np(((fk >> np.uint64(33)) ^ fk ^ (fk << np.uint64(11))).astype(np.uint32))
    fk = fk[np.argsort(h, kind="stable")].reshape(-1, block)[rng.permutation(nl // block)].reshape(-1)
    y = np.arange(nl, dtype=np.uint32)

    def run():
        j = o.JoinGpu(dev, [(fk, y)], [(pk, x)]).Prepare()
        return j, j.Run()
    j, out = _ring(dev, 18, run)
    assert _used_ring(dev) and dev.get_counter("join_attempts") == 1, (dev.get_counter("join_attempts"), dev.get_counter("join_region_overflows"))
    np.testing.assert_array_equal(np.sort(out["y"]), y)
    np.testing.assert_array_equal(fk[out["y"]], out["fk"])
    order = np.argsort(pk, kind="stable")
    np.testing.assert_array_equal(x[order][np.searchsorted(pk[order], out["fk"])], out["x"])


def test_join_u64_ring_automatic_at_2_26_rows(dev):
    """2^26 x 2^26 rows, device resident, default settings: the automatic plan takes the ring passes (8 + 7 bits); checked by
    row count, sum(x) and fk == pk[x] on the selection-vector form."""
    o = _ops()
    n = 1 << 26
    rng = np.random.default_rng(69)
    rows = np.arange(n, dtype=np.uint64)
    pk = rows * np.uint64(0x9E3779B97F4A7C15)     # a bijection on 2^64: unique keys
    pick = rng.integers(0, n, size=n, dtype=np.int64)
    fk = pk[pick]
    br = 1 << 21
    dfk, dpk = dev.to_device(fk), dev.to_device(pk)
    from dpu_olap_amd.operators import DeviceArray
    lb = [DeviceArray(dfk.ptr + 8 * b * br, br, "uint64") for b in range(n // br)]
    rb = [DeviceArray(dpk.ptr + 8 * b * br, br, "uint64") for b in range(n // br)]
    j = o.JoinGpu(dev, lb, rb, select=True).Prepare()
    out = j.Run()
    assert _used_ring(dev) and j.stats["passes"] == 2 and j.stats["missing_keys"] == 0 and dev.get_counter("join_attempts") == 1
    ox, oy, ofk = dev.to_host(out["x"]), dev.to_host(out["y"]), dev.to_host(out["fk"])
    assert ox.size == n
    np.testing.assert_array_equal(pk[ox], ofk)
    np.testing.assert_array_equal(pick[oy].astype(np.uint32), ox)
    np.testing.assert_array_equal(np.sort(oy), np.arange(n, dtype=np.uint32))
