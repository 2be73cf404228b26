"""GPU parity tests of the packed (6-byte) tuple path between the last radix pass and the probe
(dpu_olap_amd/csrc/pack6.h: k_part_scatter_ring6, k_join_probe_direct6), through the C ABI.  The join's rows must be
exactly the rows numpy derives from the inputs -- in particular the foreign keys, which the probe REBUILDS from the
partition's leading hash bits and a tuple's low hash bits with the inverse of the Wang hash (partition.c:20-28)."""
import numpy as np
import pytest

import oracle

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


def owned_keys(n, owner, skip_bits=2, mult=2654435761, add=12345):
    """n distinct keys whose leading `skip_bits` hash bits equal `owner` (what an upstream owner split leaves on a rank)."""
    m = int(n * (1 << skip_bits) * 1.05) + 1024
    c = np.arange(m, dtype=np.uint32) * np.uint32(mult) + np.uint32(add)
    c = c[(oracle.wang_hash_np(c) >> np.uint32(32 - skip_bits)) == owner]
    assert c.size >= n
    return c[:n]


def run_join(dev, fk, y, pk, x, rows, skip_bits, drop=False, knobs=None, radix_bits=16):
    """8 + 8 radix bits below `skip_bits` fixed ones -> 14 table bits: claiming pass 0, capacity-layout last pass, direct table.
    rows(n) -> batch lengths.  Returns (host columns, stats, (last-pass kernel, probe kernel), (attempts, overflows))."""
    o = _ops()
    defaults = {"pack6": 1, "probe6_geometry": 0, "probe6_prefetch": 0, "probe_slices": 1, "ring6_tpt": 8, "pack6_hash": 1, "claim_lean": 1, "contig_columns": 1}
    d, out = {}, {}
    try:
        d = {k: dev.to_device(v) for k, v in (("fk", fk), ("y", y), ("pk", pk), ("x", x))}

        def batches(a, b, n):
            res, at = [], 0
            for r in rows(n):
                res.append((o.DeviceArray(d[a].ptr + 4 * at, r, "uint32"), o.DeviceArray(d[b].ptr + 4 * at, r, "uint32")))
                at += r
            assert at == n
            return res
        dev.join_configure(radix_bits, 8)
        dev.set_tuning("claim_mode", 2)
        dev.join_set_hash_skip_bits(skip_bits)
        for k, v in (knobs or {}).items():
            dev.set_tuning(k, v)
        dev.reset_timers()
        j = o.JoinGpu(dev, batches("fk", "y", fk.size), batches("pk", "x", pk.size), drop_unmatched=drop).Prepare()
        out = j.Run()
        kernels = (dev.phase_kernel("partition-scatter-capacity"), dev.phase_kernel("join-probe"))
        counters = (dev.get_counter("join_attempts"), dev.get_counter("join_region_overflows"))
        res = tuple(dev.to_host(out[c]) for c in ("fk", "y", "x"))
        return res, j.stats, kernels, counters
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("claim_mode", 1)
        dev.join_set_hash_skip_bits(0)
        for k in (knobs or {}):
            dev.set_tuning(k, defaults[k])
        for c in list(d.values()) + list(out.values()):
            dev.release(o.DeviceArray(c.ptr, 0))


def even_batches(nb):
    return lambda n: [n // nb] * (nb - 1) + [n - (n // nb) * (nb - 1)]


def ragged_batches(nb):
    def rows(n):
        r = [n // (nb + 1) - 4321 + (7 * b) % 11 for b in range(nb - 1)]
        return r + [n - sum(r)]
    return rows


@pytest.mark.parametrize("geometry", [0, 1, 2])
@pytest.mark.parametrize("shape", ["uniform", "clustered_left", "clustered_both", "ragged", "misses"])
def test_join_packed_tuples_rows_exact(dev, shape, geometry):
    """2^25 x 2^25 rows whose keys share their two leading hash bits (one owner's share of a 4-way split).  clustered_*: rows
    ordered by hash, so every tile of the last pass falls into a few buckets (ring overflow -> direct range writes in every
    tile); ragged: odd batch lengths; misses: foreign keys without a primary key, dropped (inner join) -- their rows must
    disappear and nothing else.  Both probe geometries, and (geometry 2) the probe whose loads run ahead of the partition's
    phases (probe6_prefetch)."""
    n = 1 << 25
    rng = np.random.default_rng(11)
    pk = owned_keys(n, owner=2)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    idx = rng.integers(0, n, size=n)
    idx[: n // 16] = idx[n // 16: 2 * (n // 16)]          # repeated foreign keys
    fk, exp_x = pk[idx], x[idx]
    keep = np.ones(n, dtype=bool)
    if shape == "misses":
        # keys of the same owner that are NOT primary keys (the candidates beyond the first n)
        extra = owned_keys(n + (n >> 4), owner=2)[n:]
        where = rng.choice(n, size=extra.size, replace=False)
        fk[where] = extra
        keep[where] = False
    if shape.startswith("clustered"):
        order = np.argsort(oracle.wang_hash_np(fk), kind="stable")
        fk, exp_x, keep = fk[order], exp_x[order], keep[order]
        if shape == "clustered_both":
            order = np.argsort(oracle.wang_hash_np(pk), kind="stable")
            pk, x = pk[order], x[order]
    y = np.arange(n, dtype=np.uint32)
    rows = even_batches(32) if shape != "ragged" else ragged_batches(33)
    (ofk, oy, ox), stats, kernels, counters = run_join(dev, fk, y, pk, x, rows, 2, drop=(shape == "misses"), knobs={"probe6_geometry": geometry} if geometry < 2 else {"probe6_prefetch": 1})
    assert "k_part_scatter_ring6" in kernels[0] and "k_join_probe_direct6" in kernels[1], kernels
    assert kernels[1].endswith("true, true>") if geometry < 2 else kernels[1] == "olap::k_join_probe_direct6_pf<512, 256>", kernels
    assert ("<1024, 512," in kernels[1]) == (geometry == 1), kernels
    assert counters == (1, 0), counters                                    # no exact rerun behind the result
    assert stats["radix_bits"][:2] == [8, 8] and stats["table_slots"] == 1 << 14, stats
    assert stats["missing_keys"] == int((~keep).sum()) and ofk.size == int(keep.sum())
    np.testing.assert_array_equal(np.sort(oy), y[keep])                    # every (kept) left row exactly once
    np.testing.assert_array_equal(fk[oy], ofk)                             # its rebuilt foreign key
    np.testing.assert_array_equal(exp_x[oy], ox)                           # and the x of its primary key
    part = oracle.wang_hash_np(ofk) >> np.uint32(14)
    assert (np.diff(part.astype(np.int64)) >= 0).all()                     # partition-major


@pytest.mark.parametrize("side", ["left", "right"])
def test_join_packed_one_side_redone_exactly(dev, side):
    """Skew on ONE side overflows that side's histogram-free regions: it is redone with exact passes (8-byte tuples) while
    the other side keeps its packed partitions -- the mixed instantiations of the probe.  left: one foreign key on 5 % of
    the rows; right: one primary key 300 000 times (the build side's duplicates overwrite each other, hashtable.c:133;
    they carry the same x)."""
    n = 1 << 25
    rng = np.random.default_rng(12)
    pk = owned_keys(n, owner=1)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    idx = rng.integers(0, n, size=n)
    dups = 0
    if side == "left":
        idx[rng.random(n) < 0.05] = 777
    else:
        dups = 300_000
        pk[-dups:] = pk[5]
        x[-dups:] = x[5]
        idx = rng.integers(0, n - dups, size=n)
    fk, exp_x = pk[idx], x[idx]
    y = np.arange(n, dtype=np.uint32)
    (ofk, oy, ox), stats, kernels, counters = run_join(dev, fk, y, pk, x, even_batches(32), 2)
    assert counters[0] == 2 and counters[1] > 0, counters
    assert kernels[1].endswith("true, false>" if side == "left" else "false, true>"), kernels
    assert stats["missing_keys"] == 0 and stats["duplicate_keys"] == dups, stats
    np.testing.assert_array_equal(np.sort(oy), y)
    np.testing.assert_array_equal(fk[oy], ofk)
    np.testing.assert_array_equal(exp_x[oy], ox)


def test_join_packed_refuses_unsplit_inputs(dev):
    """The packed tuples drop the key, so the promise of olap_join_set_hash_skip_bits is checked where the keys are still
    whole (the last pass): keys of all four owners under skip_bits = 2 -> OLAP_ERR_INVALID, not wrong rows."""
    o = _ops()
    from dpu_olap_amd import capi
    n = 1 << 25
    rng = np.random.default_rng(13)
    pk = np.arange(n, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(99)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    fk = pk[rng.integers(0, n, size=n)]
    y = np.arange(n, dtype=np.uint32)
    with pytest.raises(capi.OlapError) as e:
        run_join(dev, fk, y, pk, x, even_batches(32), 2)
    assert e.value.code == capi.ERR_INVALID and "leading hash bits" in str(e.value)


@pytest.mark.parametrize("pack", [1, 0])
def test_join_packed_generator_shape_sf64_digest(dev, pack):
    """BM_Join shape at SF=64 (2^27 rows per side, generated in HBM) with the SF=2048 plan pinned (9 + 9 bits, direct
    table, no skipped bits): row digest against the digest predicted from the inputs, with and without packing."""
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows, nb = 2 << 20, 64
    n = rows * nb
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    dev.join_configure(18, 9)
    dev.set_tuning("claim_mode", 2)
    dev.set_tuning("pack6", pack)
    try:
        j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(nb)], [(batch("pk", b), batch("x", b)) for b in range(nb)]).Prepare()
        out = j.Run()
        probe = dev.phase_kernel("join-probe")
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("claim_mode", 1)
        dev.set_tuning("pack6", 1)
    assert ("k_join_probe_direct6" in probe) == bool(pack), probe
    assert out["fk"].rows == n and j.stats["missing_keys"] == 0 and j.stats["table_slots"] == 1 << 14, j.stats
    assert dev.get_counter("join_attempts") == 1
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


@pytest.mark.parametrize("knobs", [{"ring6_tpt": 4}, {"ring6_tpt": 4, "pack6_hash": 0}, {"ring6_tpt": 8, "pack6_hash": 0}, {"claim_lean": 0}, {"contig_columns": 0}],
                         ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
@pytest.mark.parametrize("shape", ["uniform", "clustered_left"])
def test_join_packed_tuples_kernel_variants(dev, shape, knobs):
    """The other builds of the packed path on 2^25 x 2^25 rows: 2048-tuple tiles with the flush one tile behind the insert
    (ring6_tpt = 4), keys instead of hashes between the passes (pack6_hash = 0), ring_scatter.h's claiming kernel in front
    (claim_lean = 0), the pointer tables instead of the one-buffer addressing (contig_columns = 0)."""
    n = 1 << 25
    rng = np.random.default_rng(21)
    pk = owned_keys(n, owner=3)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    idx = rng.integers(0, n, size=n)
    fk, exp_x = pk[idx], x[idx]
    if shape == "clustered_left":
        order = np.argsort(oracle.wang_hash_np(fk), kind="stable")
        fk, exp_x = fk[order], exp_x[order]
    y = np.arange(n, dtype=np.uint32)
    (ofk, oy, ox), stats, kernels, counters = run_join(dev, fk, y, pk, x, even_batches(32), 2, knobs=knobs)
    assert "k_part_scatter_ring6" in kernels[0] and f", {knobs.get('ring6_tpt', 8)}, true, " in kernels[0], kernels
    assert kernels[1].endswith("true, true>") and counters == (1, 0) and stats["missing_keys"] == 0, (kernels, counters, stats)
    np.testing.assert_array_equal(np.sort(oy), y)
    np.testing.assert_array_equal(fk[oy], ofk)
    np.testing.assert_array_equal(exp_x[oy], ox)


@pytest.mark.parametrize("tpt", [8, 4])
@pytest.mark.parametrize("shape", ["uniform", "clustered_both", "ragged_misses"])
def test_join_packed_tuples_128_buckets(dev, shape, tpt):
    """An 8 + 7 bit plan below 3 skipped hash bits (2^24 x 2^24 rows of one owner of an 8-way split): the last pass has 128
    buckets -- rings of eight lines, eight dummy rings, several lines per bucket and tile in the flusher's rounds."""
    n = 1 << 24
    rng = np.random.default_rng(31)
    pk = owned_keys(n, owner=5, skip_bits=3)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    idx = rng.integers(0, n, size=n)
    fk, exp_x = pk[idx], x[idx]
    keep = np.ones(n, dtype=bool)
    if shape == "ragged_misses":
        extra = owned_keys(n + (n >> 5), owner=5, skip_bits=3)[n:]
        where = rng.choice(n, size=extra.size, replace=False)
        fk[where] = extra
        keep[where] = False
    if shape == "clustered_both":
        order = np.argsort(oracle.wang_hash_np(fk), kind="stable")
        fk, exp_x = fk[order], exp_x[order]
        order = np.argsort(oracle.wang_hash_np(pk), kind="stable")
        pk, x = pk[order], x[order]
    y = np.arange(n, dtype=np.uint32)
    rows = ragged_batches(17) if shape == "ragged_misses" else even_batches(16)
    (ofk, oy, ox), stats, kernels, counters = run_join(dev, fk, y, pk, x, rows, 3, drop=(shape == "ragged_misses"), knobs={"ring6_tpt": tpt}, radix_bits=15)
    assert stats["radix_bits"][:2] == [8, 7] and stats["table_slots"] == 1 << 14, stats
    assert "k_part_scatter_ring6" in kernels[0] and kernels[1].endswith("true, true>") and counters == (1, 0), (kernels, counters)
    assert stats["missing_keys"] == int((~keep).sum()) and ofk.size == int(keep.sum())
    np.testing.assert_array_equal(np.sort(oy), y[keep])
    np.testing.assert_array_equal(fk[oy], ofk)
    np.testing.assert_array_equal(exp_x[oy], ox)
