"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C
ABI (include/olap_gpu.h via dpu_olap_amd.capi), against the CPU oracle, the
reference's known-answer data and the Arrow golden vectors.  Bit-exact for all
integer / index work."""
import os

import numpy as np
import pytest

import oracle
from conftest import sort_rows, split_batches

pytestmark = pytest.mark.gpu

TYPES = {"u32": np.uint32, "i32": np.int32, "u64": np.uint64, "i64": np.int64}


@pytest.fixture(scope="module")
def dev():
    from dpu_olap_amd import Device
    d = Device(0)
    yield d
    d.close()


def _ops():
    import dpu_olap_amd
    return dpu_olap_amd


# ----------------------------------------------------------------- filter
def test_filter_reference_kats(dev, kats, golden):
    o = _ops()
    small = np.array(kats["small_column"], dtype=np.uint32)
    f = o.FilterGpu(dev, [small]).Prepare()
    assert f.Run() == kats["filter_simple_test"]["expected_count"]  # FilterTest.SimpleTest
    frt = kats["filter_result_test"]
    pat = np.arange(frt["rows"], dtype=np.uint32) + np.uint32(1 << 30)
    for i in frt["small_positions"]:
        pat[i] = i
    res = o.FilterGpu(dev, [pat]).Prepare().GetResult()  # FilterTest.ResultTest
    np.testing.assert_array_equal(res[0], np.array(frt["expected"], dtype=np.uint32))
    res = o.FilterGpu(dev, [golden["edge_u32_in"]]).Prepare().GetResult()
    np.testing.assert_array_equal(res[0], golden["edge_u32_out"])  # unsigned compare


@pytest.mark.parametrize("t", list(TYPES))
@pytest.mark.parametrize("space", ["host", "device"])
def test_filter_vs_arrow_golden(dev, golden, t, space):
    o = _ops()
    batches = split_batches(golden[f"rand_{t}_values"], golden[f"rand_{t}_sizes"])
    thr = golden[f"rand_{t}_thr"][0].item()
    ins = batches if space == "host" else [dev.to_device(b) for b in batches]
    res = o.FilterGpu(dev, ins, threshold=thr).Prepare().GetResult()
    if space == "device":
        res = [dev.to_host(r) for r in res]
    np.testing.assert_array_equal(np.array([r.size for r in res], dtype=np.uint64), golden[f"rand_{t}_filter_counts"])
    np.testing.assert_array_equal(np.concatenate(res), golden[f"rand_{t}_filter_out"])


@pytest.mark.parametrize("rows", [1, 4095, 4096, 4097, 65536, 65537, 1 << 20, (1 << 22) + 13])
def test_filter_single_batch_sizes_vs_oracle(dev, rows):
    """FilterTest.LongerTest shape (1 x 64Ki) and the multi-segment path (batch longer
    than one segment -> count + scan + compact)."""
    o = _ops()
    rng = np.random.default_rng(rows)
    v = rng.integers(0, 1 << 32, size=rows, dtype=np.uint32)
    res = o.FilterGpu(dev, [v]).Prepare().GetResult()
    np.testing.assert_array_equal(res[0], oracle.filter_lt(v, 1 << 30))


def test_filter_many_batches_device_and_unaligned(dev):
    o = _ops()
    rng = np.random.default_rng(7)
    nb, rows = 128, 65536
    v = rng.integers(0, 1 << 32, size=nb * rows + 3, dtype=np.uint32)
    d = dev.to_device(v)
    # aligned batches
    ins = [o.DeviceArray(d.ptr + 4 * b * rows, rows, "uint32") for b in range(nb)]
    res = o.FilterGpu(dev, ins).Prepare().GetResult()
    for b in (0, 1, 63, 127):
        np.testing.assert_array_equal(dev.to_host(res[b]), oracle.filter_lt(v[b * rows:(b + 1) * rows], 1 << 30))
    total = sum(r.rows for r in res)
    assert total == oracle.filter_lt(v[: nb * rows], 1 << 30).size
    # 4-byte aligned only (Arrow slices): scalar path
    ins = [o.DeviceArray(d.ptr + 4 * (b * rows + 1), rows, "uint32") for b in range(4)]
    res = o.FilterGpu(dev, ins).Prepare().GetResult()
    for b in range(4):
        np.testing.assert_array_equal(dev.to_host(res[b]), oracle.filter_lt(v[b * rows + 1:(b + 1) * rows + 1], 1 << 30))


@pytest.mark.parametrize("t", ["u32", "i64"])
def test_filter_long_batches_one_workgroup_each(dev, t):
    """Batches longer than a 64 Ki-row segment: with >= 2 of them per CU one workgroup walks a whole batch with its
    running offset (no count pass, timer phase filter-count absent); with the knob at 0, or with few batches, count +
    scan + compact.  Ragged lengths, an empty batch, an unaligned slice; same rows either way (oracle)."""
    o = _ops()
    rng = np.random.default_rng(21)
    dt = TYPES[t]
    nb = 2 * dev.device_info()["compute_units"] + 5
    lens = [65536 + 4096 * (b % 5) + (b % 7) for b in range(nb)]
    lens[3], lens[4] = 0, 300001
    hi = 1 << 32 if dt in (np.uint32, np.uint64) else 1 << 31
    whole = rng.integers(0, hi, size=sum(lens) + 1, dtype=np.int64).astype(dt)
    d = dev.to_device(whole)
    w = np.dtype(dt).itemsize
    starts = np.concatenate([[0], np.cumsum(lens)])
    ins = [o.DeviceArray(d.ptr + w * int(starts[b] + (1 if b % 2 else 0)), lens[b], np.dtype(dt).name) for b in range(nb)]   # odd batches one element off
    host = [whole[int(starts[b] + (1 if b % 2 else 0)):][:lens[b]] for b in range(nb)]
    for knob, few in ((2, False), (0, False), (2, True)):
        dev.set_tuning("filter_whole_batches", knob)
        try:
            f = o.FilterGpu(dev, ins[:9] if few else ins).Prepare()
            res = f.GetResult()
            timers = f.Timers()
        finally:
            dev.set_tuning("filter_whole_batches", 2)
        assert bool(timers.get("filter-count", 0)) == (knob == 0 or few), timers
        for b in range(len(res)):
            if b < 12 or b % 37 == 0:
                np.testing.assert_array_equal(dev.to_host(res[b]), oracle.filter_lt(host[b], 1 << 30))
        assert sum(r.rows for r in res) == sum(oracle.filter_lt(h, 1 << 30).size for h in host[:len(res)])


def test_filter_empty_inputs(dev):
    o = _ops()
    assert o.FilterGpu(dev, []).Prepare().Run() == 0
    res = o.FilterGpu(dev, [np.empty(0, np.uint32), np.array([1], np.uint32)]).Prepare().GetResult()
    assert res[0].size == 0 and res[1].tolist() == [1]


# -------------------------------------------------------------------- sum
def test_sum_reference_kat(dev, kats):
    o = _ops()
    small = np.array(kats["small_column"], dtype=np.uint32)
    assert o.SumGpu(dev, [small]).Prepare().Run() == kats["sum_simple_test"]["expected"]


@pytest.mark.parametrize("t", list(TYPES))
@pytest.mark.parametrize("space", ["host", "device"])
def test_sum_vs_arrow_golden(dev, golden, t, space):
    o = _ops()
    batches = split_batches(golden[f"rand_{t}_values"], golden[f"rand_{t}_sizes"])
    ins = batches if space == "host" else [dev.to_device(b) for b in batches]
    assert o.SumGpu(dev, ins).Prepare().Run() == int(golden[f"rand_{t}_sum"][0])


def test_sum_large_and_wrap(dev):
    """SumTest.LargeTest shape: 128 x 64Ki random u32; plus u64/i64 wrap like Arrow."""
    o = _ops()
    rng = np.random.default_rng(42)
    batches = [rng.integers(0, 1 << 32, size=65536, dtype=np.uint32) for _ in range(128)]
    assert o.SumGpu(dev, batches).Prepare().Run() == sum(oracle.sum_(b) for b in batches)
    assert o.SumGpu(dev, [np.array([(1 << 64) - 1, 5], dtype=np.uint64)]).Prepare().Run() == 4
    assert o.SumGpu(dev, [np.array([(1 << 63) - 1, 1], dtype=np.int64)]).Prepare().Run() == -(1 << 63)
    assert o.SumGpu(dev, [np.full(1 << 20, -7, dtype=np.int32)]).Prepare().Run() == -7 * (1 << 20)
    assert o.SumGpu(dev, []).Prepare().Run() == 0


# ------------------------------------------------------------------- take
def test_take_reference_kat(dev, kats):
    o = _ops()
    small = np.array(kats["small_column"], dtype=np.uint32)
    idx = np.array(kats["take_indices"], dtype=np.uint32)
    res = o.TakeGpu(dev, [small], [idx]).Prepare().Run()
    assert res[0].tolist() == kats["take_simple_test"]["expected"]


@pytest.mark.parametrize("t", list(TYPES))
def test_take_vs_arrow_golden(dev, golden, t):
    o = _ops()
    data = split_batches(golden[f"rand_{t}_values"], golden[f"rand_{t}_sizes"])[-1]
    idx = golden[f"rand_{t}_take_idx"]
    res = o.TakeGpu(dev, [data], [idx]).Prepare().Run()
    np.testing.assert_array_equal(res[0], golden[f"rand_{t}_take_out"])


@pytest.mark.parametrize("space", ["host", "device"])
def test_take_large_test_shape(dev, space):
    """TakeTest.LargeTest: 128 batches x 64Ki rows, 8Ki indices each."""
    o = _ops()
    rng = np.random.default_rng(42)
    nb = 128
    data = [rng.integers(0, 1 << 32, size=65536, dtype=np.uint32) for _ in range(nb)]
    idx = [rng.integers(0, 65536, size=8192, dtype=np.uint32) for _ in range(nb)]
    idx[5] = idx[5][:1000]  # ragged + partial tile
    if space == "device":
        res = o.TakeGpu(dev, [dev.to_device(d) for d in data], [dev.to_device(i) for i in idx]).Prepare().Run()
        res = [dev.to_host(r) for r in res]
    else:
        res = o.TakeGpu(dev, data, idx).Prepare().Run()
    for b in range(nb):
        np.testing.assert_array_equal(res[b], oracle.take(data[b], idx[b]))


# -------------------------------------------------------------- partition
def test_partition_reference_kat(dev, kats):
    o = _ops()
    p = kats["partition_simple_test"]
    parts = o.PartitionGpu(dev, [(np.array(p["pk"], np.uint32), np.array(p["x"], np.uint32))], p["nr_partitions"]).Prepare().Run()
    assert [k.size for k, _ in parts] == p["expected_sizes"]  # PartitionTest.SimpleTest
    assert sum(int(k.astype(np.uint64).sum()) for k, _ in parts) == p["expected_sum_pk"]
    assert sum(int(x.astype(np.uint64).sum()) for _, x in parts) == p["expected_sum_x"]


@pytest.mark.parametrize("P", [2, 32, 1024, 4096])
@pytest.mark.parametrize("space", ["host", "device"])
def test_partition_vs_oracle(dev, P, space):
    o = _ops()
    rng = np.random.default_rng(P)
    sizes = [65536, 1, 0, 40000, 8192, 8191]
    keys = [rng.integers(0, 1 << 32, size=s, dtype=np.uint32) for s in sizes]
    pay = [rng.integers(0, 1 << 32, size=s, dtype=np.uint32) for s in sizes]
    if space == "device":
        batches = [(dev.to_device(k), dev.to_device(v)) for k, v in zip(keys, pay)]
    else:
        batches = list(zip(keys, pay))
    parts = o.PartitionGpu(dev, batches, P).Prepare().Run()
    allk, allp = np.concatenate(keys), np.concatenate(pay)
    bucket = oracle.wang_hash_np(allk) >> np.uint32(32 - int(np.log2(P)))
    assert sum(k.size for k, _ in parts) == allk.size
    for p in (0, 1, P // 2, P - 1):
        k, v = parts[p]
        assert (oracle.wang_hash_np(k) >> np.uint32(32 - int(np.log2(P))) == p).all()
        ek, ev = sort_rows(allk[bucket == p], allp[bucket == p])
        gk, gv = sort_rows(k, v)
        np.testing.assert_array_equal(gk, ek)
        np.testing.assert_array_equal(gv, ev)
    # sizes of every partition == oracle's per-batch histograms summed (join_dpu.cc:54-66)
    counts = np.zeros(P, dtype=np.int64)
    for kb in keys:
        if kb.size:
            _, _, hist = oracle.partition(kb, P)
            counts += np.diff(np.concatenate([[0], hist]))
    assert [k.size for k, _ in parts] == counts.tolist()


@pytest.mark.parametrize("dtype", ["int32", "int64"])
def test_partition_converted_host_batches(dev, dtype):
    """Host batches that are not uint32 already are converted (copied) before the C call; the copies must live
    through it (the pointer tables held addresses of arrays that had been freed)."""
    o = _ops()
    rng = np.random.default_rng(11)
    keys = [rng.integers(0, 1 << 31, size=s).astype(dtype) for s in (50000, 3, 70001)]
    pay = [rng.integers(0, 1 << 31, size=k.size).astype(dtype) for k in keys]
    parts = o.PartitionGpu(dev, list(zip(keys, pay)), 64).Prepare().Run()
    allk, allp = np.concatenate(keys).astype(np.uint32), np.concatenate(pay).astype(np.uint32)
    bucket = oracle.wang_hash_np(allk) >> np.uint32(26)
    for p in range(64):
        ek, ev = sort_rows(allk[bucket == p], allp[bucket == p])
        gk, gv = sort_rows(*parts[p])
        np.testing.assert_array_equal(gk, ek)
        np.testing.assert_array_equal(gv, ev)


def test_partition_keys_only_returns_row_ids(dev):
    o = _ops()
    rng = np.random.default_rng(3)
    keys = [rng.integers(0, 1 << 32, size=10000, dtype=np.uint32) for _ in range(3)]
    parts = o.PartitionGpu(dev, keys, 16, payload_column=None).Prepare().Run()
    allk = np.concatenate(keys)
    for k, rid in parts:
        np.testing.assert_array_equal(allk[rid], k)  # selection_indices_vector semantics


# ------------------------------------------------------------------- join
def _join_sorted(dev, lfk, ly, rpk, rx, device=False, **cfg):
    o = _ops()
    dev.join_configure(cfg.get("bits", -1), cfg.get("max_bits", 0))
    try:
        if device:
            lb = [(dev.to_device(a), dev.to_device(b)) for a, b in zip(lfk, ly)]
            rb = [(dev.to_device(a), dev.to_device(b)) for a, b in zip(rpk, rx)]
        else:
            lb, rb = list(zip(lfk, ly)), list(zip(rpk, rx))
        j = o.JoinGpu(dev, lb, rb).Prepare()
        out = j.Run()
        if device:
            out = {k: dev.to_host(v) for k, v in out.items()}
    finally:
        dev.join_configure(-1, 0)
    return sort_rows(out["fk"], out["y"], out["x"]), j.stats


def test_join_simple_test_kat(dev, kats):
    """JoinTest.SimpleTest: keys shuffled across batches; compare after sort by (v_l, fk)."""
    j = kats["join_simple_test"]
    a = lambda k: [np.array(b, dtype=np.uint32) for b in j[k]]
    (fk, y, x), _ = _join_sorted(dev, a("left_fk"), a("left_v_l"), a("right_pk"), a("right_v_r"))
    y2, fk2, x2 = sort_rows(y, fk, x)
    np.testing.assert_array_equal(fk2, np.array(j["expected_sorted_fk"], dtype=np.uint32))
    np.testing.assert_array_equal(y2, np.array(j["expected_sorted_v_l"], dtype=np.uint32))
    np.testing.assert_array_equal(x2, np.array(j["expected_sorted_v_r"], dtype=np.uint32))


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("cfg", [dict(), dict(bits=1), dict(bits=6, max_bits=2), dict(bits=12, max_bits=4), dict(bits=11), dict(bits=16, max_bits=8)])
@pytest.mark.parametrize("device", [False, True])
def test_join_vs_arrow_golden(dev, golden, case, cfg, device):
    """Arrow (Acero hash join) golden rows; every radix configuration -- 1, 2, 3 and 4
    passes -- must give the same table."""
    lf = split_batches(golden[f"join_{case}_fk"], golden[f"join_{case}_left_rows"])
    ly = split_batches(golden[f"join_{case}_y"], golden[f"join_{case}_left_rows"])
    rp = split_batches(golden[f"join_{case}_pk"], golden[f"join_{case}_right_rows"])
    rx = split_batches(golden[f"join_{case}_x"], golden[f"join_{case}_right_rows"])
    (fk, y, x), stats = _join_sorted(dev, lf, ly, rp, rx, device=device, **cfg)
    np.testing.assert_array_equal(fk, golden[f"join_{case}_out_fk"])
    np.testing.assert_array_equal(y, golden[f"join_{case}_out_y"])
    np.testing.assert_array_equal(x, golden[f"join_{case}_out_x"])
    assert stats["missing_keys"] == 0 and stats["duplicate_keys"] == 0
    if "bits" in cfg:
        assert sum(stats["radix_bits"]) == cfg["bits"]


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("cfg", [dict(), dict(bits=12, max_bits=4), dict(bits=20), dict(bits=18)])
def test_join_scatter_variants_agree(dev, golden, variant, cfg):
    """All scatter kernels (0 = write-combining 64-B lines, 1 = tile-staged, 2 = write-combining
    with 128-B lines for passes of <= 512 buckets) give the golden table, also with the 10+10-bit
    plan of the SF=2048 benchmark and a 9+9-bit plan (tiny partitions here)."""
    dev.set_tuning("scatter_variant", 1 if variant == 1 else 0)
    dev.set_tuning("line16", 1 if variant == 2 else 0)
    try:
        for case in ("a", "b"):
            lf = split_batches(golden[f"join_{case}_fk"], golden[f"join_{case}_left_rows"])
            ly = split_batches(golden[f"join_{case}_y"], golden[f"join_{case}_left_rows"])
            rp = split_batches(golden[f"join_{case}_pk"], golden[f"join_{case}_right_rows"])
            rx = split_batches(golden[f"join_{case}_x"], golden[f"join_{case}_right_rows"])
            (fk, y, x), _ = _join_sorted(dev, lf, ly, rp, rx, device=True, **cfg)
            np.testing.assert_array_equal(fk, golden[f"join_{case}_out_fk"])
            np.testing.assert_array_equal(y, golden[f"join_{case}_out_y"])
            np.testing.assert_array_equal(x, golden[f"join_{case}_out_x"])
    finally:
        dev.set_tuning("scatter_variant", 0)
        dev.set_tuning("line16", 0)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_partition_variants_vs_oracle_multiset(dev, variant):
    o = _ops()
    rng = np.random.default_rng(11)
    keys = [rng.integers(0, 1 << 32, size=s, dtype=np.uint32) for s in (300000, 77777, 5)]
    # heavy skew: one hot key and one hot bucket
    keys[0][:100000] = 12345
    pay = [np.arange(k.size, dtype=np.uint32) for k in keys]
    dev.set_tuning("scatter_variant", 1 if variant == 1 else 0)
    dev.set_tuning("line16", 1 if variant == 2 else 0)
    try:
        parts = o.PartitionGpu(dev, list(zip(keys, pay)), 1024 if variant != 2 else 512 * 512).Prepare().Run()
        if variant == 2:  # 9+9-bit plan: fold the 2^18 partitions back to the top 10 bits
            merged = []
            for i in range(1024):
                grp = parts[i * 256:(i + 1) * 256]
                merged.append((np.concatenate([k for k, _ in grp]), np.concatenate([p for _, p in grp])))
            parts = merged
    finally:
        dev.set_tuning("scatter_variant", 0)
        dev.set_tuning("line16", 0)
    allk, allp = np.concatenate(keys), np.concatenate(pay)
    bucket = oracle.wang_hash_np(allk) >> np.uint32(22)
    gk = np.concatenate([k for k, _ in parts])
    gp = np.concatenate([p for _, p in parts])
    gb = np.concatenate([np.full(k.size, i, dtype=np.uint32) for i, (k, _) in enumerate(parts)])
    np.testing.assert_array_equal(oracle.wang_hash_np(gk) >> np.uint32(22), gb)
    e = sort_rows(bucket, allk, allp)
    g = sort_rows(gb, gk, gp)
    for a, b in zip(e, g):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("bits", [18, 20, 24])
def test_join_direct_address_table_vs_probing_table(dev, golden, direct, bits):
    """With <= 14 hash bits left below the partition bits the build side goes into a
    collision-free direct-address LDS table (distinct keys have distinct Wang hashes); the
    probing table must give the same rows.  Includes the keys 0 and 0xFFFFFFFF (case b)."""
    dev.set_tuning("direct_table", direct)
    try:
        for case in ("a", "b"):
            lf = split_batches(golden[f"join_{case}_fk"], golden[f"join_{case}_left_rows"])
            ly = split_batches(golden[f"join_{case}_y"], golden[f"join_{case}_left_rows"])
            rp = split_batches(golden[f"join_{case}_pk"], golden[f"join_{case}_right_rows"])
            rx = split_batches(golden[f"join_{case}_x"], golden[f"join_{case}_right_rows"])
            (fk, y, x), stats = _join_sorted(dev, lf, ly, rp, rx, device=True, bits=bits)
            np.testing.assert_array_equal(fk, golden[f"join_{case}_out_fk"])
            np.testing.assert_array_equal(y, golden[f"join_{case}_out_y"])
            np.testing.assert_array_equal(x, golden[f"join_{case}_out_x"])
            assert stats["table_slots"] == ((1 << (32 - bits)) if direct else 4096)  # (tiny partitions: the small probing table)
    finally:
        dev.set_tuning("direct_table", 1)


def test_join_large_test_shape_vs_oracle(dev):
    """JoinTest.LargeTest: 128 x 64Ki rows per side, generator data; all columns equal after
    sort by (fk, y); row count == left rows."""
    from dpu_olap_amd import datagen as dg
    nb, L, R = 128, 65536, 65536
    pk = [dg.index_u32(R, b * R) for b in range(nb)]
    x = [dg.uniform_u32(R, dg.SEED, dg.STREAM_X, b * R) for b in range(nb)]
    fk = [dg.fk_u32(L, dg.SEED, dg.STREAM_FK, b * L, L, R) for b in range(nb)]
    y = [dg.uniform_u32(L, dg.SEED, dg.STREAM_Y, b * L) for b in range(nb)]
    (gfk, gy, gx), stats = _join_sorted(dev, fk, y, pk, x, device=True)
    assert gfk.size == nb * L
    ofk, oy, ox = oracle.join(fk, y, pk, x, nr_partitions=128)
    ofk, oy, ox = sort_rows(ofk, oy, ox)
    np.testing.assert_array_equal(gfk, ofk)
    np.testing.assert_array_equal(gy, oy)
    np.testing.assert_array_equal(gx, ox)
    assert stats["passes"] >= 1 and stats["max_build_rows"] <= max(6144, stats["table_slots"])


def test_join_partition_major_output_order(dev):
    o = _ops()
    rng = np.random.default_rng(5)
    pk = rng.permutation(1 << 16).astype(np.uint32)
    x = rng.integers(0, 1 << 32, size=pk.size, dtype=np.uint32)
    fk = pk[rng.integers(0, pk.size, size=100000)]
    y = np.arange(fk.size, dtype=np.uint32)
    j = o.JoinGpu(dev, [(fk, y)], [(pk, x)]).Prepare()
    out = j.Run()
    bits = sum(j.stats["radix_bits"])
    part = oracle.wang_hash_np(out["fk"]) >> np.uint32(32 - bits)
    assert (np.diff(part.astype(np.int64)) >= 0).all()  # join_dpu.cc:371-399
    lut = np.zeros(1 << 16, dtype=np.uint32)
    lut[pk] = x
    np.testing.assert_array_equal(out["x"], lut[out["fk"]])
    np.testing.assert_array_equal(out["fk"], fk[out["y"]])


def test_join_missing_key_is_an_error(dev):
    o = _ops()
    pk = np.arange(1000, dtype=np.uint32)
    x = pk + np.uint32(5)
    fk = np.array([1, 2, 5000], dtype=np.uint32)
    with pytest.raises(o.OlapError) as e:
        o.JoinGpu(dev, [(fk, fk)], [(pk, x)]).Prepare().Run()
    assert e.value.code == -4  # OLAP_ERR_NO_MATCH (reference asserts, hash_probe.c:32-33)
    with pytest.raises(o.OlapError):
        o.JoinGpu(dev, [(fk, fk)], []).Prepare().Run()


def test_join_duplicate_build_keys_overwrite(dev):
    o = _ops()
    pk = np.array([7, 7, 9], dtype=np.uint32)
    x = np.array([1, 1, 3], dtype=np.uint32)
    fk = np.array([9, 7], dtype=np.uint32)
    j = o.JoinGpu(dev, [(fk, fk)], [(pk, x)]).Prepare()
    out = j.Run()
    assert sorted(zip(out["fk"].tolist(), out["x"].tolist())) == [(7, 1), (9, 3)]  # hashtable.c:133
    assert j.stats["duplicate_keys"] == 1


def test_join_empty_left_and_skew(dev):
    o = _ops()
    pk = np.arange(100, dtype=np.uint32)
    assert o.JoinGpu(dev, [], [(pk, pk)]).Prepare().Run()["fk"].size == 0
    # all probes hit the same key: one huge probe partition
    fk = np.full(300000, 42, dtype=np.uint32)
    out = o.JoinGpu(dev, [(fk, np.arange(fk.size, dtype=np.uint32))], [(pk, pk + np.uint32(1))]).Prepare().Run()
    assert out["fk"].size == fk.size and (out["x"] == 43).all()
    assert np.array_equal(np.sort(out["y"]), np.arange(fk.size, dtype=np.uint32))


# -------------------------------------------------- generator + full-size style properties
def test_device_generator_matches_host_mirror(dev):
    from dpu_olap_amd import datagen as dg
    n = 100003
    d = dev.empty(n, "uint32")
    dev.gen_uniform_u32(d.ptr, n, 42, dg.STREAM_V, 12345)
    np.testing.assert_array_equal(dev.to_host(d), dg.uniform_u32(n, 42, dg.STREAM_V, 12345))
    dev.gen_index_u32(d.ptr, n, 77)
    np.testing.assert_array_equal(dev.to_host(d), dg.index_u32(n, 77))
    dev.gen_fk_u32(d.ptr, n, 42, dg.STREAM_FK, 1000, 3000, 2048)
    np.testing.assert_array_equal(dev.to_host(d), dg.fk_u32(n, 42, dg.STREAM_FK, 1000, 3000, 2048))
    rows = dev.to_device(np.arange(5, 5 + n, dtype=np.uint32))
    dev.gen_uniform_u32_at(d.ptr, rows.ptr, n, 42, dg.STREAM_X)
    np.testing.assert_array_equal(dev.to_host(d), dg.uniform_u32(n, 42, dg.STREAM_X, 5))
    d64 = dev.empty(n, "uint64")
    dev.gen_uniform_u64(d64.ptr, n, 42, 9, 3)
    h64 = dg.uniform_u64(n, 42, 9, 3)
    np.testing.assert_array_equal(dev.to_host(d64), h64)
    assert dev.checksum(d64.ptr, n, 8) == dg.checksum(h64)
    h = dev.to_host(d)
    assert dev.checksum(d.ptr, n, 4) == dg.checksum(h)
    assert dev.checksum_rows3_u32(d.ptr, rows.ptr, d.ptr, n) == dg.checksum_rows3(h, np.arange(5, 5 + n, dtype=np.uint32), h)


def test_join_capacity_layout_overflow_falls_back_to_exact(dev):
    """At this size the last radix pass skips its histogram and writes into fixed-capacity
    regions; a skewed probe side (only 1024 distinct foreign keys) overflows them and the
    join must transparently redo the partitioning exactly.  Digest must still match."""
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows, sf = 2 << 20, 64
    n = sf * rows
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, n, 1024)  # fk uniform in [0, 1024)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(sf)],
                  [(batch("pk", b), batch("x", b)) for b in range(sf)]).Prepare()
    out = j.Run()
    assert out["fk"].rows == n
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
    # the overflow is found before the probe and only the LEFT side (the skewed one) is redone: one probe launch, and the
    # histogram ran for one side's passes only
    t = dev.timers()
    assert dev.get_counter("join_attempts") == 2 and dev.get_counter("join_region_overflows") > 0
    assert t["join-probe"]["launches"] == 1 and t["partition-hist"]["launches"] == j.stats["passes"], t
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


@pytest.mark.parametrize("capacity_mode", [1, 0])
def test_join_capacity_and_exact_layouts_agree(dev, capacity_mode):
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows, sf = 2 << 20, 64
    n = sf * rows
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    dev.set_tuning("capacity_mode", capacity_mode)
    try:
        j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(sf)],
                      [(batch("pk", b), batch("x", b)) for b in range(sf)]).Prepare()
        out = j.Run()
        timers = j.Timers()
    finally:
        dev.set_tuning("capacity_mode", 1)
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
    # partition-major order survives the capacity layout: partition ids never decrease
    fk = dev.to_host(out["fk"])
    part = oracle.wang_hash_np(fk) >> np.uint32(32 - sum(j.stats["radix_bits"]))
    assert (np.diff(part.astype(np.int64)) >= 0).all()
    # capacity mode on: the histogram-free pair (claiming pass 0 + capacity-layout pass 1); off: histogram + scan + scatter
    assert ("partition-scatter-capacity" in timers) == bool(capacity_mode) and ("partition-hist" in timers) == (not capacity_mode), timers
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


@pytest.mark.parametrize("sf", [8, 12, 16, 64])
def test_join_generator_scale_digest(dev, sf):
    """Size-independent property at benchmark shape (SF batches x 2Mi rows per side,
    join_benchmark.cc:170-171): pk is the global row index, so the joined x of a row must
    be gen32(STREAM_X, fk); the digest of the output rows {fk, y, x} must equal the digest
    of {fk_in, y_in, gen_at(fk_in)} computed from the inputs, row order ignored."""
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows = 2 << 20
    n = sf * rows
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(sf)],
                  [(batch("pk", b), batch("x", b)) for b in range(sf)]).Prepare()
    out = j.Run()
    assert out["fk"].rows == n
    got = dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n)
    assert got == expect
    assert j.stats["missing_keys"] == 0 and j.stats["max_build_rows"] <= max(6144, j.stats["table_slots"])
    if sf in (12, 16):  # 14 radix bits: pass 0 takes log2(#CU) = 8 of them, so that both passes run without histograms
        assert j.stats["radix_bits"][:2] == [8, 6] and dev.get_counter("join_attempts") == 1, j.stats
    if sf in (8, 12):  # and the oracle agrees on this size (a few seconds of CPU)
        h = {k: dev.to_host(cols[k]) for k in ("fk", "y", "pk", "x")}
        sp = lambda a: [a[b * rows:(b + 1) * rows] for b in range(sf)]
        ofk, oy, ox = oracle.join(sp(h["fk"]), sp(h["y"]), sp(h["pk"]), sp(h["x"]), nr_partitions=64)
        assert dg.checksum_rows3(ofk, oy, ox) == got
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


@pytest.mark.parametrize("world", [2, 8])
def test_multirank_join_loopback_on_one_gpu(dev, world):
    """The multi-GPU join's device steps -- owner split (olap_partition_u32 with
    nr_partitions = G), exchange (emulated by slicing the send regions) and the local join
    below the owner bits (hash skip bits) -- run for G logical ranks on this one GPU with the
    product's GpuLocalOps; the union of the shards must digest like the whole join."""
    import torch
    from dpu_olap_amd import datagen as dg
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rows, nb = 1 << 18, 16          # 16 batches of 256 Ki rows per side, 2 per logical rank at G = 8
    n = rows * nb
    col = lambda: torch.empty(n, dtype=torch.int32, device=device)
    fk, y, pk, x, exp = col(), col(), col(), col(), col()
    dev.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(pk.data_ptr(), n, 0)
    dev.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(exp.data_ptr(), fk.data_ptr(), n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(fk.data_ptr(), y.data_ptr(), exp.data_ptr(), n)
    per = nb // world
    # zipped path first: partition_tuples -> slices -> join_tuples
    zs = []
    for r in range(world):
        sl = lambda t: [t[(r * per + b) * rows:(r * per + b + 1) * rows] for b in range(per)]
        zs.append(ops.partition_tuples(sl(fk), sl(y), world) + ops.partition_tuples(sl(pk), sl(x), world))
    bits = world.bit_length() - 1
    zsum, zxor, zrows = 0, 0, 0
    for r in range(world):
        RL = torch.cat([z[0][z[1][r]:z[1][r + 1]] for z in zs])
        RR = torch.cat([z[2][z[3][r]:z[3][r + 1]] for z in zs])
        ofk, oy, ox, stats = ops.join_tuples(RL, RR, bits)
        assert stats["missing_keys"] == 0
        s, xo = dev.checksum_rows3_u32(ofk.data_ptr(), oy.data_ptr(), ox.data_ptr(), ofk.numel())
        zsum = (zsum + s) & ((1 << 64) - 1)
        zxor ^= xo
        zrows += ofk.numel()
    assert zrows == n and (zsum, zxor) == expect
    sends = []
    for r in range(world):          # every logical rank groups its batches by owner
        sl = lambda t: [t[(r * per + b) * rows:(r * per + b + 1) * rows] for b in range(per)]
        lk, lp, loff = ops.partition(sl(fk), sl(y), world)
        rk, rp, roff = ops.partition(sl(pk), sl(x), world)
        sends.append((lk, lp, loff, rk, rp, roff))
    tot_sum, tot_xor, tot_rows = 0, 0, 0
    for r in range(world):          # rank r receives region r of every sender
        cat = lambda i, o: torch.cat([s[i][s[o][r]:s[o][r + 1]] for s in sends])
        RLk, RLp, RRk, RRp = cat(0, 2), cat(1, 2), cat(3, 5), cat(4, 5)
        owner = oracle.wang_hash_np(RLk.cpu().numpy().view(np.uint32)) >> np.uint32(32 - bits)
        assert (owner == r).all()
        ofk, oy, ox, stats = ops.join([RLk], [RLp], [RRk], [RRp], bits)
        assert stats["missing_keys"] == 0 and stats["max_build_rows"] <= max(6144, stats["table_slots"])
        s, xo = dev.checksum_rows3_u32(ofk.data_ptr(), oy.data_ptr(), ox.data_ptr(), ofk.numel())
        tot_sum = (tot_sum + s) & ((1 << 64) - 1)
        tot_xor ^= xo
        tot_rows += ofk.numel()
    assert tot_rows == n and (tot_sum, tot_xor) == expect


@pytest.mark.parametrize("world", [2, 8])
def test_multirank_fused_split_loopback_on_one_gpu(dev, world):
    """The fused multi-GPU path: every logical rank splits into G x S = 512 buckets
    (olap_partition_tuples_u32), rank r receives per sender its S runs (emulated by
    slicing), and olap_join_presplit_u32 reads those chunks in place.  Union of the shards
    digests like the whole join."""
    import torch
    from dpu_olap_amd import datagen as dg
    from dpu_olap_amd.dist_join import GpuLocalOps, _chunk_tables
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rows, nb = 1 << 18, 16
    n = rows * nb
    col = lambda: torch.empty(n, dtype=torch.int32, device=device)
    fk, y, pk, x, exp = col(), col(), col(), col(), col()
    dev.gen_fk_u32(fk.data_ptr(), n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(y.data_ptr(), n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(pk.data_ptr(), n, 0)
    dev.gen_uniform_u32(x.data_ptr(), n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(exp.data_ptr(), fk.data_ptr(), n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(fk.data_ptr(), y.data_ptr(), exp.data_ptr(), n)
    per, S, bits = nb // world, 512 // world, world.bit_length() - 1
    sends = []
    for r in range(world):
        sl = lambda t: [t[(r * per + b) * rows:(r * per + b + 1) * rows] for b in range(per)]
        sends.append(ops.partition_tuples(sl(fk), sl(y), world * S) + ops.partition_tuples(sl(pk), sl(x), world * S))
    tsum, txor, trows = 0, 0, 0
    for r in range(world):
        def recv(ti, oi):
            buf = torch.cat([z[ti][z[oi][r * S]:z[oi][(r + 1) * S]] for z in sends])
            cnt = torch.tensor([[z[oi][r * S + j + 1] - z[oi][r * S + j] for j in range(S)] for z in sends], dtype=torch.int64)
            return (buf,) + _chunk_tables(cnt)
        RL, l_lo, l_hi = recv(0, 1)
        RR, r_lo, r_hi = recv(2, 3)
        if r % 2:   # the overlapped form: the right table is "still arriving" (a copy in flight on torch's
            # stream) when the join is called; the join's left pass runs meanwhile and its right pass waits for the event
            torch.cuda.current_stream(device).synchronize()
            RR2 = torch.empty_like(RR)
            RR2.copy_(RR, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
            ofk, oy, ox, stats = ops.join_presplit(RL, l_lo, l_hi, RR2, r_lo, r_hi, S, world, bits, right_ready=ready)
        else:
            ofk, oy, ox, stats = ops.join_presplit(RL, l_lo, l_hi, RR, r_lo, r_hi, S, world, bits)
        assert stats["missing_keys"] == 0 and stats["duplicate_keys"] == 0
        owner = oracle.wang_hash_np(ofk[:4096].cpu().numpy().view(np.uint32)) >> np.uint32(32 - bits)
        assert (owner == r).all()
        s_, xo = dev.checksum_rows3_u32(ofk.data_ptr(), oy.data_ptr(), ox.data_ptr(), ofk.numel())
        tsum = (tsum + s_) & ((1 << 64) - 1)
        txor ^= xo
        trows += ofk.numel()
    assert trows == n and (tsum, txor) == expect


def test_join_direct_table_checks_the_skipped_hash_bits(dev):
    """olap_join_set_hash_skip_bits promises that every tuple shares its leading hash bits; the key-less direct-address
    table is only sound under that promise, so it is checked: inputs that were NOT split by owner are refused with
    OLAP_ERR_INVALID instead of joining a foreign key to another key's x."""
    import torch
    from dpu_olap_amd import capi
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rng = np.random.default_rng(21)
    n = 1 << 16
    pk = rng.permutation(1 << 20)[:n].astype(np.uint32)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    fk = pk[rng.integers(0, n, size=n)]
    y = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    zip64 = lambda k, p: torch.from_numpy((k.astype(np.uint64) | (p.astype(np.uint64) << np.uint64(32))).view(np.int64)).to(device)
    owner = lambda k: oracle.wang_hash_np(k) >> np.uint32(30)
    dev.join_configure(16, 0)          # 2 skipped + 16 radix bits -> 14 bits left: the direct-address table
    try:
        with pytest.raises(capi.OlapError) as e:      # all four owners mixed
            ops.join_tuples(zip64(fk, y), zip64(pk, x), 2)
        assert e.value.code == capi.ERR_INVALID and "leading hash bits" in str(e.value)
        lm, rm = owner(fk) == 3, owner(pk) == 3       # split by owner: accepted, rows = the owner's foreign keys
        ofk, oy, ox, st = ops.join_tuples(zip64(fk[lm], y[lm]), zip64(pk[rm], x[rm]), 2)
        assert st["missing_keys"] == 0 and ofk.numel() == int(lm.sum()) and st["table_slots"] == 1 << 14
        got = sort_rows(*(t.cpu().numpy().view(np.uint32) for t in (ofk, oy, ox)))
        lut = dict(zip(pk.tolist(), x.tolist()))
        exp = sort_rows(fk[lm], y[lm], np.array([lut[k] for k in fk[lm].tolist()], dtype=np.uint32))
        for g, ex in zip(got, exp):
            np.testing.assert_array_equal(g, ex)
    finally:
        dev.join_configure(-1, 0)


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("n_seg,chunks,bits", [(4, 3, -1), (16, 2, 9), (1, 5, 3), (64, 1, 8)])
def test_join_presplit_matches_golden(dev, golden, case, n_seg, chunks, bits):
    """olap_join_presplit_u32 on ragged chunks (odd boundaries, an empty sender) against the
    Arrow golden join; rows compared as sorted sets."""
    import torch
    from dpu_olap_amd.dist_join import GpuLocalOps, _chunk_tables
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(device)
    fk, y, pk, x = (golden[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))

    def side(k, p):
        cuts = np.linspace(0, k.size, chunks + 1).astype(np.int64)
        cuts = np.minimum(cuts | 1, k.size) if chunks > 1 else cuts      # odd boundaries
        cuts[0], cuts[-1] = 0, k.size
        if chunks > 2:
            cuts[2] = cuts[1]                                            # sender 1 sends nothing
        parts, cnt = [], []
        for c in range(chunks):
            kk, pp = k[cuts[c]:cuts[c + 1]], p[cuts[c]:cuts[c + 1]]
            if n_seg > 1 and kk.size:
                z, off = ops.partition_tuples([t(kk)], [t(pp)], n_seg)
            else:
                z = (torch.from_numpy((kk.astype(np.uint64) | (pp.astype(np.uint64) << np.uint64(32))).view(np.int64).copy()).to(device))
                off = [0, kk.size] if n_seg == 1 else [0] * (n_seg + 1)
            parts.append(z)
            cnt.append([off[j + 1] - off[j] for j in range(n_seg)])
        buf = torch.cat(parts) if parts else ops.empty_tuples(0)
        pad = ops.empty_tuples(buf.numel() + 2)                          # 16-B aligned base, slack for the tail vector
        pad[:buf.numel()] = buf
        return (pad[:buf.numel()],) + _chunk_tables(torch.tensor(cnt, dtype=torch.int64))

    L, l_lo, l_hi = side(fk, y)
    R, r_lo, r_hi = side(pk, x)
    dev.join_configure(bits, 0)
    try:
        ofk, oy, ox, stats = ops.join_presplit(L, l_lo, l_hi, R, r_lo, r_hi, n_seg, chunks, 0)
    finally:
        dev.join_configure(-1, 0)
    assert stats["nr_partitions"] >= 2 * n_seg and stats["missing_keys"] == 0
    u = lambda tt: tt.cpu().numpy().view(np.uint32)
    rfk, ry, rx = sort_rows(u(ofk), u(oy), u(ox))
    np.testing.assert_array_equal(rfk, golden[f"join_{case}_out_fk"])
    np.testing.assert_array_equal(ry, golden[f"join_{case}_out_y"])
    np.testing.assert_array_equal(rx, golden[f"join_{case}_out_x"])


@pytest.mark.parametrize("t", ["u32", "i64"])
def test_take_lds_window_kernel_many_small_batches(dev, t):
    """>= #CU batches of <= 512 KiB with <= 8192 indices each take the LDS-window kernel
    (data streamed through LDS, indices / results in registers); ragged sizes, partial windows."""
    o = _ops()
    rng = np.random.default_rng(99)
    dt = TYPES[t]
    nb = 300
    rows = [65536] * nb
    rows[3], rows[7], rows[11] = 1, 40000, 65537 if t == "u32" else 16385
    data = [rng.integers(0, 1 << 31, size=r, dtype=np.int64).astype(dt) for r in rows]
    idx = [rng.integers(0, r, size=8192, dtype=np.uint32) for r in rows]
    idx[5] = idx[5][:17]
    idx[6] = idx[6][:0]
    res = o.TakeGpu(dev, [dev.to_device(d) for d in data], [dev.to_device(i) for i in idx]).Prepare().Run()
    for b in list(range(0, 16)) + [nb - 1]:
        np.testing.assert_array_equal(dev.to_host(res[b]), oracle.take(data[b], idx[b]))
    dev.set_tuning("take_variant", 1)  # same through the direct-gather kernel
    try:
        res2 = o.TakeGpu(dev, [dev.to_device(d) for d in data[:16]] + [dev.to_device(d) for d in data[16:]],
                         [dev.to_device(i) for i in idx]).Prepare().Run()
    finally:
        dev.set_tuning("take_variant", 0)
    for b in (0, 3, 5, 7, 11, nb - 1):
        np.testing.assert_array_equal(dev.to_host(res2[b]), oracle.take(data[b], idx[b]))


def test_take_large_batches_small_tile_kernel(dev):
    o = _ops()
    rng = np.random.default_rng(5)
    data = [rng.integers(0, 1 << 32, size=(1 << 20) + 3, dtype=np.uint32) for _ in range(3)]
    idx = [rng.integers(0, d.size, size=100003, dtype=np.uint32) for d in data]
    for variant in (0, 3):
        dev.set_tuning("take_variant", variant)
        try:
            res = o.TakeGpu(dev, [dev.to_device(d) for d in data], [dev.to_device(i) for i in idx]).Prepare().Run()
        finally:
            dev.set_tuning("take_variant", 0)
        for b in range(3):
            np.testing.assert_array_equal(dev.to_host(res[b]), oracle.take(data[b], idx[b]))


@pytest.mark.parametrize("t", ["u32", "i32", "u64", "i64"])
def test_take_window_bucketed_path(dev, t):
    """Batches whose data buffer is far beyond LDS: indices bucketed by 64 KiB window, windows served from LDS
    (take_variant 4 selects the path; it is opt-in because the direct gather is faster on the MI355X).  Ragged data sizes
    (partial last window, one-row buffer, exactly one window, unaligned device slices), ragged and empty index
    batches, duplicate and clustered indices; compared with the oracle, as is the direct gather (variant 0)."""
    o = _ops()
    rng = np.random.default_rng(1234)
    dt = TYPES[t]
    win = (64 << 10) // np.dtype(dt).itemsize
    nb = 144
    rows = [5 * win + 77] * nb
    rows[0], rows[1], rows[2], rows[3], rows[4] = 1, win, win + 1, 37 * win - 5, 2 * win
    data = [rng.integers(0, 1 << 31, size=r, dtype=np.int64).astype(dt) for r in rows]
    idx = [rng.integers(0, r, size=40000 + 13 * (b % 7), dtype=np.uint32) for b, r in enumerate(rows)]
    idx[5] = idx[5][:0]
    idx[6] = idx[6][:1]
    idx[7] = np.full(5000, rows[7] - 1, dtype=np.uint32)                         # one window, one element
    idx[8] = np.sort(idx[8])                                                       # clustered
    idx[9] = (np.arange(30000, dtype=np.uint32) * 7) % np.uint32(rows[9])
    # unaligned device buffers: slices of a larger allocation, off by one element
    keep = []

    def shifted(a, name):
        whole = dev.to_device(np.concatenate([a[:1], a]))
        keep.append(whole)
        return o.DeviceArray(whole.ptr + a.itemsize, a.size, name)

    d_data = [shifted(d, np.dtype(dt).name) if b % 3 == 1 else dev.to_device(d) for b, d in enumerate(data)]
    d_idx = [shifted(i, "uint32") if b % 4 == 2 and i.size else dev.to_device(i) for b, i in enumerate(idx)]
    want = [oracle.take(d, i) for d, i in zip(data, idx)]
    for variant in (4, 0):
        dev.set_tuning("take_variant", variant)
        try:
            res = o.TakeGpu(dev, d_data, d_idx).Prepare().Run()
        finally:
            dev.set_tuning("take_variant", 0)
        for b in range(nb):
            np.testing.assert_array_equal(dev.to_host(res[b]), want[b], err_msg=f"variant {variant} batch {b}")


def test_take_window_bucketed_host_operands(dev):
    """The same path with OLAP_MEM_HOST operands and results (staged through the arena)."""
    o = _ops()
    rng = np.random.default_rng(77)
    data = [rng.integers(0, 1 << 32, size=(1 << 18) + 5 * b, dtype=np.uint32) for b in range(6)]
    idx = [rng.integers(0, d.size, size=50000 + b, dtype=np.uint32) for b, d in enumerate(data)]
    dev.set_tuning("take_variant", 4)
    try:
        res = o.TakeGpu(dev, data, idx).Prepare().Run()
    finally:
        dev.set_tuning("take_variant", 0)
    for b in range(6):
        np.testing.assert_array_equal(np.asarray(res[b]), oracle.take(data[b], idx[b]))


# ------------------------------------------------------- host staging ring
@pytest.mark.parametrize("stage,threads", [(1, 0), (1, 1), (1, 3), (0, 0)])
def test_host_operands_through_the_pinned_ring(dev, stage, threads):
    """OLAP_MEM_HOST operands larger than one 32 MiB pinned slot, ragged batches (so windows
    cut batches and span many of them), host results: filter / sum / take / partition / join
    against the oracle, with the ring (1, 3 or default copy threads) and with the plain
    per-buffer copies (host_stage=0)."""
    o = _ops()
    dev.set_tuning("host_stage", stage)
    dev.set_tuning("host_copy_threads", threads)
    try:
        rng = np.random.default_rng(11)
        sizes = [int(x) for x in rng.integers(1, 400_000, size=120)] + [9_000_001, 0, 5]
        batches = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]     # ~130 MB
        res = o.FilterGpu(dev, batches).Prepare().GetResult()
        for b in (0, 1, 57, 119, 120, 121, 122):
            np.testing.assert_array_equal(res[b], oracle.filter_lt(batches[b], 1 << 30))
        assert sum(r.size for r in res) == sum(oracle.filter_lt(b, 1 << 30).size for b in batches)
        assert o.SumGpu(dev, batches).Prepare().Run() == sum(int(b.sum(dtype=np.uint64)) for b in batches)
        b64 = [rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64) for n in sizes[:40]]
        res = o.FilterGpu(dev, b64, threshold=0).Prepare().GetResult()
        for b in (0, 13, 39):
            np.testing.assert_array_equal(res[b], b64[b][b64[b] < 0])
        data = [b for b in batches if b.size]
        idx = [rng.integers(0, d.size, size=min(d.size, 70_000), dtype=np.uint32) for d in data]
        out = o.TakeGpu(dev, data, idx).Prepare().Run()
        for b in (0, 60, len(data) - 1):
            np.testing.assert_array_equal(out[b], data[b][idx[b]])
        # join: 12 M x 10 M rows in ragged host batches, host result columns
        nr = 10_000_000
        pk = rng.permutation(nr).astype(np.uint32)
        x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
        fk = rng.integers(0, nr, size=12_000_000, dtype=np.uint32)
        y = rng.integers(0, 1 << 32, size=fk.size, dtype=np.uint32)
        cut = lambda a, k: np.array_split(a, k)
        j = o.JoinGpu(dev, list(zip(cut(fk, 7), cut(y, 7))), list(zip(cut(pk, 5), cut(x, 5)))).Prepare()
        r = j.Run()
        ofk, oy, ox = r["fk"], r["y"], r["x"]
        lut = np.empty(nr, np.uint32)
        lut[pk] = x
        assert ofk.size == fk.size
        np.testing.assert_array_equal(ox, lut[ofk])
        a, b_, c = sort_rows(ofk, oy, ox)
        e = sort_rows(fk, y, lut[fk])
        np.testing.assert_array_equal(a, e[0]); np.testing.assert_array_equal(b_, e[1]); np.testing.assert_array_equal(c, e[2])
    finally:
        dev.set_tuning("host_stage", 1)
        dev.set_tuning("host_copy_threads", 0)


@pytest.mark.parametrize("space", ["host", "device"])
def test_join_selection_vectors(dev, golden, space):
    """olap_join_u32 with NULL payload tables: y / x come back as the global left / right row
    index (kernel_hash_probe's selection indices); gathering through them reproduces the
    Arrow golden join."""
    o = _ops()
    fk, y, pk, x = (golden[f"join_b_{k}"] for k in ("fk", "y", "pk", "x"))
    lb = split_batches(fk, golden["join_b_left_rows"])
    rb = split_batches(pk, golden["join_b_right_rows"])
    if space == "device":
        lb, rb = [dev.to_device(b) for b in lb], [dev.to_device(b) for b in rb]
    r = o.JoinGpu(dev, lb, rb, select=True).Prepare().Run()
    get = (lambda a: a) if space == "host" else dev.to_host
    ofk, lrow, rrow = get(r["fk"]), get(r["y"]), get(r["x"])
    np.testing.assert_array_equal(np.sort(lrow), np.arange(fk.size, dtype=np.uint32))   # every left row exactly once
    np.testing.assert_array_equal(fk[lrow], ofk)
    np.testing.assert_array_equal(pk[rrow], ofk)
    a, b_, c = sort_rows(ofk, y[lrow], x[rrow])
    np.testing.assert_array_equal(a, golden["join_b_out_fk"])
    np.testing.assert_array_equal(b_, golden["join_b_out_y"])
    np.testing.assert_array_equal(c, golden["join_b_out_x"])


@pytest.mark.parametrize("space", ["device", "host"])
@pytest.mark.parametrize("nr", [1, 37, 4096, 6144, 6145])
def test_join_small_build_side_broadcast(dev, space, nr):
    """A right table of <= 6144 rows (one LDS table) is joined without partitioning (k_join_broadcast): stats.passes == 0,
    rows in LEFT ROW ORDER; 6145 rows take the partitioned path again.  Ragged
    and empty left batches, an unaligned device slice, the key 0xFFFFFFFF (the table's empty marker) on both sides, a
    duplicate build key (counted once, the later value wins as in hashtable.c:133 -- here both carry the same x), payloads
    and selection vectors, misses as error and dropped (inner join).  Everything against the oracle / numpy."""
    o = _ops()
    rng = np.random.default_rng(1000 + nr)
    pk = rng.choice(1 << 32, size=nr, replace=False).astype(np.uint32)
    if nr >= 37:
        pk[5] = 0xFFFFFFFF
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    sizes = [70001, 0, 4096, 8192 * 5 + 3, 1, 300000]
    nl = sum(sizes)
    fk = pk[rng.integers(0, nr, size=nl)]
    y = rng.integers(0, 1 << 32, size=nl, dtype=np.uint32)
    cut = lambda a, sz: np.split(a, np.cumsum(sz)[:-1])
    keep = []

    def wrap(a):
        if space == "host":
            return a
        if a.size and a.size % 2 == 1:   # an odd-sized batch lives one element into its allocation: 4-byte aligned only
            whole = dev.to_device(np.concatenate([a[:1], a]))
            keep.append(whole)
            return o.DeviceArray(whole.ptr + 4, a.size, "uint32")
        return dev.to_device(a)

    rsz = [nr] if nr < 100 else [nr // 3, nr - nr // 3]
    left = [(wrap(k), wrap(p)) for k, p in zip(cut(fk, sizes), cut(y, sizes))]
    right = [(wrap(k), wrap(p)) for k, p in zip(cut(pk, rsz), cut(x, rsz))]
    get = (lambda a: np.asarray(a)) if space == "host" else dev.to_host
    lut = dict(zip(pk.tolist(), x.tolist()))
    exp_x = np.array([lut[k] for k in fk.tolist()], dtype=np.uint32)
    j = o.JoinGpu(dev, left, right).Prepare()
    r = j.Run()
    ofk, oy, ox = get(r["fk"]), get(r["y"]), get(r["x"])
    if nr <= 6144:
        assert j.stats["passes"] == 0 and j.stats["nr_partitions"] == 1 and dev.phase_kernel("join-probe") == "olap::k_join_broadcast", j.stats
        np.testing.assert_array_equal(ofk, fk)      # left row order
        np.testing.assert_array_equal(oy, y)
        np.testing.assert_array_equal(ox, exp_x)
    else:
        assert j.stats["passes"] >= 1 and j.stats["nr_partitions"] >= 2, j.stats
        for a, b in zip(sort_rows(ofk, oy, ox), sort_rows(fk, y, exp_x)):
            np.testing.assert_array_equal(a, b)
    assert j.stats["missing_keys"] == 0 and j.stats["duplicate_keys"] == 0 and j.stats["max_build_rows"] <= nr
    # the same rows when partitioning is forced
    dev.set_tuning("broadcast_join", 0)
    try:
        r2 = o.JoinGpu(dev, left, right).Prepare().Run()
    finally:
        dev.set_tuning("broadcast_join", 1)
    for a, b in zip(sort_rows(get(r2["fk"]), get(r2["y"]), get(r2["x"])), sort_rows(fk, y, exp_x)):
        np.testing.assert_array_equal(a, b)
    # selection vectors
    js = o.JoinGpu(dev, [k for k, _ in left], [k for k, _ in right], select=True).Prepare()
    rs = js.Run()
    sfk, lrow, rrow = get(rs["fk"]), get(rs["y"]), get(rs["x"])
    np.testing.assert_array_equal(fk[lrow], sfk)
    np.testing.assert_array_equal(pk[rrow], sfk)
    np.testing.assert_array_equal(np.sort(lrow), np.arange(nl, dtype=np.uint32))
    if nr <= 6144:
        np.testing.assert_array_equal(lrow, np.arange(nl, dtype=np.uint32))
    if nr >= 37:
        # a duplicate build key and foreign keys without a match
        pk2, x2 = np.concatenate([pk, pk[7:8]]), np.concatenate([x, x[7:8]])
        fk2 = fk.copy()
        miss = rng.choice(nl, size=1000, replace=False)
        absent = np.setdiff1d(rng.integers(0, 1 << 32, size=4000, dtype=np.uint64).astype(np.uint32), pk)[:1000]
        fk2[miss] = absent
        left2 = [(wrap(k), wrap(p)) for k, p in zip(cut(fk2, sizes), cut(y, sizes))]
        right2 = [(wrap(pk2), wrap(x2))]
        if nr + 1 <= 6144:
            with pytest.raises(o.OlapError):
                o.JoinGpu(dev, left2, right2).Prepare().Run()
            jd = o.JoinGpu(dev, left2, right2, drop_unmatched=True).Prepare()
            rd = jd.Run()
            assert jd.stats["passes"] == 0 and jd.stats["duplicate_keys"] == 1 and jd.stats["missing_keys"] == 1000, jd.stats
            hit = np.ones(nl, dtype=bool)
            hit[miss] = False
            np.testing.assert_array_equal(get(rd["fk"]), fk2[hit])      # inner join, left order preserved
            np.testing.assert_array_equal(get(rd["y"]), y[hit])
            np.testing.assert_array_equal(get(rd["x"]), exp_x[hit])


@pytest.mark.parametrize("plan", ["probing", "direct"])
def test_join_hot_foreign_key_probe_slices(dev, plan):
    """A probe partition far beyond 64 Ki rows (40 % of the left rows carry ONE foreign key, another 10 % a key WITHOUT a
    match): the partition is cut into slices that several workgroups take up (k_probe_slices), each rebuilding the
    partition's table.  Same rows as with probe_slices = 0 and as numpy; misses inside the hot slices are reported /
    dropped correctly; duplicates on the build side are counted once, not once per slice."""
    o = _ops()
    rng = np.random.default_rng(77)
    nr, nl, nb = 1 << 20, 3_000_000, 5
    pk = rng.permutation(nr).astype(np.uint32) * np.uint32(2)              # even keys
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    u = rng.random(nl)
    fk[u < 0.4] = pk[123]                                                  # the hot key
    fk[(u >= 0.4) & (u < 0.5)] = np.uint32(7)                              # odd: not a primary key
    y = np.arange(nl, dtype=np.uint32)
    pk2, x2 = np.concatenate([pk, pk[123:124]]), np.concatenate([x, x[123:124]])   # the hot key twice on the build side
    cut = lambda a: np.array_split(a, nb)
    left = [(dev.to_device(k), dev.to_device(p)) for k, p in zip(cut(fk), cut(y))]
    right = [(dev.to_device(k), dev.to_device(p)) for k, p in zip(cut(pk2), cut(x2))]
    lut = np.zeros(2 * nr, np.uint32)
    lut[pk] = x
    keep = fk % 2 == 0
    results = []
    for knob in (1, 0):
        dev.set_tuning("probe_slices", knob)
        if plan == "direct":
            dev.join_configure(18, 9)       # 14 hash bits remain: direct-address table
        else:
            dev.join_configure(8, 8)        # probing table, 4096 build rows per partition
        try:
            with pytest.raises(o.OlapError):
                o.JoinGpu(dev, left, right).Prepare().Run()
            j = o.JoinGpu(dev, left, right, drop_unmatched=True).Prepare()
            r = j.Run()
        finally:
            dev.set_tuning("probe_slices", 1)
            dev.join_configure(-1, 0)
        assert j.stats["missing_keys"] == int((~keep).sum()) and j.stats["duplicate_keys"] == 1, j.stats
        assert ("direct" in dev.phase_kernel("join-probe")) == (plan == "direct")
        ofk, oy, ox = dev.to_host(r["fk"]), dev.to_host(r["y"]), dev.to_host(r["x"])
        assert ofk.size == int(keep.sum())
        np.testing.assert_array_equal(np.sort(oy), y[keep])
        np.testing.assert_array_equal(fk[oy], ofk)
        np.testing.assert_array_equal(lut[ofk], ox)
        results.append(sort_rows(ofk, oy, ox))
    for a, b in zip(*results):
        np.testing.assert_array_equal(a, b)


def test_join_many_short_batches_scratch_sizing(dev):
    """24 x 50021-row batches: every batch rounds up to whole 4096-row tiles, so pass 0 has far
    more tiles than rows / 8192 -- the histogram matrix must be sized for that."""
    o = _ops()
    rng = np.random.default_rng(5)
    nb, rows = 24, 50021
    pk = [np.arange(b * rows, (b + 1) * rows, dtype=np.uint32) for b in range(nb)]
    x = [rng.integers(0, 1 << 32, size=rows, dtype=np.uint32) for _ in range(nb)]
    fk = [rng.integers(0, nb * rows, size=rows, dtype=np.uint32) for _ in range(nb)]
    y = [rng.integers(0, 1 << 32, size=rows, dtype=np.uint32) for _ in range(nb)]
    r = o.JoinGpu(dev, list(zip(fk, y)), list(zip(pk, x))).Prepare().Run()
    np.testing.assert_array_equal(r["x"], np.concatenate(x)[r["fk"]])
    a, b_ = sort_rows(r["fk"], r["y"])
    e = sort_rows(np.concatenate(fk), np.concatenate(y))
    np.testing.assert_array_equal(a, e[0]); np.testing.assert_array_equal(b_, e[1])


@pytest.mark.parametrize("shape", ["uniform", "skew", "tuples", "ragged"])
def test_join_claiming_first_pass(dev, shape):
    """Pass 0 as a claiming scatter (no histogram; allotments claimed with atomics, fillers
    skipped by the next pass), forced at a size where it would not switch on by itself.
    uniform: digest + partition-major order; skew: regions overflow -> exact rerun; tuples:
    the zipped-tuple entry point; ragged: odd batch lengths (partial tiles)."""
    import torch
    o = _ops()
    from dpu_olap_amd import datagen as dg
    from dpu_olap_amd.dist_join import GpuLocalOps
    rows, sf = (2 << 20, 64) if shape != "ragged" else ((2 << 20) - 12345, 61)
    n = sf * rows
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, *((n, 1024) if shape == "skew" else (rows, rows)))
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    dev.set_tuning("claim_mode", 2)
    try:
        if shape == "tuples":
            device = torch.device("cuda", 0)
            ops = GpuLocalOps(dev, device)
            tl = torch.empty(n, dtype=torch.int64, device=device)
            tr = torch.empty(n, dtype=torch.int64, device=device)
            dev.partition_tuples_u32([batch("fk", b).ptr for b in range(sf)], [batch("y", b).ptr for b in range(sf)], [rows] * sf,
                                     1, 2, tl.data_ptr(), n)
            dev.partition_tuples_u32([batch("pk", b).ptr for b in range(sf)], [batch("x", b).ptr for b in range(sf)], [rows] * sf,
                                     1, 2, tr.data_ptr(), n)
            ofk, oy, ox, stats = ops.join_tuples(tl, tr, 0)
            got = dev.checksum_rows3_u32(ofk.data_ptr(), oy.data_ptr(), ox.data_ptr(), ofk.numel())
            assert ofk.numel() == n and stats["missing_keys"] == 0
        else:
            j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(sf)],
                          [(batch("pk", b), batch("x", b)) for b in range(sf)]).Prepare()
            out = j.Run()
            assert out["fk"].rows == n
            got = dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n)
            if shape == "uniform":
                fk = dev.to_host(out["fk"])
                part = oracle.wang_hash_np(fk) >> np.uint32(32 - sum(j.stats["radix_bits"]))
                assert (np.diff(part.astype(np.int64)) >= 0).all()
                assert "partition-hist" not in j.Timers() or j.Timers()["partition-hist"] == 0
            for c in out.values():
                dev.release(o.DeviceArray(c.ptr, 0))
    finally:
        dev.set_tuning("claim_mode", 1)
    assert got == expect
    for c in cols.values():
        dev.release(o.DeviceArray(c.ptr, 0))


# ------------------------------------------------ Arrow C Data Interface front end
def test_arrow_c_data_interface_front_end(dev, golden):
    """include/olap_arrow.h through pyarrow's C Data Interface export/import: sliced inputs
    (non-zero offsets), results owned by pyarrow after the call, against pyarrow.compute."""
    import pyarrow as pa
    import pyarrow.compute as pc
    from dpu_olap_amd import arrow_cdata as ac
    rng = np.random.default_rng(3)
    chunks = [pa.array(rng.integers(0, 1 << 32, size=n, dtype=np.uint32)) for n in (70000, 1, 0, 123457)]
    chunks[0] = chunks[0].slice(5)                       # offset 5
    col = pa.chunked_array(chunks)
    out, total = ac.filter_lt(dev, col, 1 << 30)
    assert out.num_chunks == col.num_chunks and total == len(out)
    for a, b in zip(out.chunks, col.chunks):
        assert a.equals(pc.filter(b, pc.less(b, pa.scalar(1 << 30, pa.uint32()))))
    assert ac.sum_(dev, col) == pc.sum(col).as_py()
    i64 = pa.chunked_array([pa.array(rng.integers(-(1 << 40), 1 << 40, size=5000, dtype=np.int64)) for _ in range(3)])
    out64, _ = ac.filter_lt(dev, i64, 0)
    assert out64.equals(pc.filter(i64, pc.less(i64, 0))) and ac.sum_(dev, i64) == pc.sum(i64).as_py()
    data = [c for c in col.chunks if len(c)]
    idx = [pa.array(rng.integers(0, len(c), size=1000, dtype=np.uint32)) for c in data]
    for got, d, i in zip(ac.take(dev, data, idx), data, idx):
        assert got.equals(d.take(i))
    # join: golden case b in its ragged batches, plus the selection-vector form
    fk, y, pk, x = (golden[f"join_b_{k}"] for k in ("fk", "y", "pk", "x"))
    cut = lambda a, sizes: pa.chunked_array([pa.array(c) for c in split_batches(a, sizes)])
    lrows, rrows = golden["join_b_left_rows"], golden["join_b_right_rows"]
    tbl, stats = ac.join(dev, cut(fk, lrows), cut(y, lrows), cut(pk, rrows), cut(x, rrows))
    assert stats["missing_keys"] == 0 and tbl.num_rows == fk.size
    a, b_, c = sort_rows(*(tbl[k].to_numpy() for k in ("fk", "y", "x")))
    np.testing.assert_array_equal(a, golden["join_b_out_fk"])
    np.testing.assert_array_equal(b_, golden["join_b_out_y"])
    np.testing.assert_array_equal(c, golden["join_b_out_x"])
    sel, _ = ac.join(dev, cut(fk, lrows), None, cut(pk, rrows), None)
    np.testing.assert_array_equal(pk[sel["x"].to_numpy()], sel["fk"].to_numpy())
    np.testing.assert_array_equal(fk[sel["y"].to_numpy()], sel["fk"].to_numpy())
    # 64-bit keys (olap_arrow_join_u64): golden case d (int64 keys, negatives included); fk keeps its type
    fk8, y8, pk8, x8 = (golden[f"join_d_{k}"] for k in ("fk", "y", "pk", "x"))
    l8, r8 = golden["join_d_left_rows"], golden["join_d_right_rows"]
    t8, st8 = ac.join(dev, cut(fk8, l8), cut(y8, l8), cut(pk8, r8), cut(x8, r8))
    assert st8["missing_keys"] == 0 and t8.num_rows == fk8.size and str(t8["fk"].type) == str(fk8.dtype)
    a8, b8, c8 = sort_rows(*(t8[k].to_numpy() for k in ("fk", "y", "x")))
    np.testing.assert_array_equal(a8, golden["join_d_out_fk"])
    np.testing.assert_array_equal(b8, golden["join_d_out_y"])
    np.testing.assert_array_equal(c8, golden["join_d_out_x"])
    s8, _ = ac.join(dev, cut(fk8, l8), None, cut(pk8, r8), None)
    np.testing.assert_array_equal(pk8[s8["x"].to_numpy()], s8["fk"].to_numpy())
    del tbl, sel, out, out64, t8, s8                      # release callbacks run here
    with pytest.raises(ac.capi.OlapError):               # nulls are refused, not silently read
        ac.sum_(dev, pa.chunked_array([pa.array([1, None, 3], type=pa.uint32())]))
    # null_count == -1 ("not computed", legal in the C Data Interface) with a validity bitmap: the bitmap decides
    import ctypes as C
    def sum_with_unknown_null_count(arr):
        ins, schema, out = ac._Exported([arr]), ac._schema_of(arr.type), C.c_uint64()
        ins.structs[0].null_count = -1
        try:
            return ac._lib().olap_arrow_sum(dev.h, C.byref(schema), ins.table, 1, C.byref(out)), out.value
        finally:
            ins.release()
            ac._release_schema(schema)
    with_nulls = pa.array([1, None, 3, 4], type=pa.uint32())
    assert sum_with_unknown_null_count(with_nulls)[0] == ac.capi.ERR_INVALID
    assert sum_with_unknown_null_count(with_nulls.slice(2)) == (0, 7)     # bitmap present, the slice has no null


@pytest.mark.parametrize("space", ["host", "device"])
@pytest.mark.parametrize("size", ["small", "large"])
def test_join_drop_unmatched_is_an_inner_join(dev, space, size):
    """OLAP_UNMATCHED_DROP: rows whose foreign key has no primary key are left out (Arrow's
    inner hash join, the reference's JoinNative); default mode stays an error.  small: probing
    table; large: direct-address table + capacity layout; misses spread over all partitions."""
    o = _ops()
    rng = np.random.default_rng(9)
    nr, nl, nb = (20_000, 50_000, 3) if size == "small" else (40_000_000, 30_000_000, 6)
    pk = (rng.permutation(nr).astype(np.uint32) * np.uint32(2))            # even keys only
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = rng.integers(0, 2 * nr, size=nl, dtype=np.uint32)                  # about half of them odd -> no match
    fk[:5] = (2 * nr + 7, 0xFFFFFFFF, 1, 3, 5)
    y = np.arange(nl, dtype=np.uint32)
    cut = lambda a: np.array_split(a, nb)
    wrap = (lambda a: a) if space == "host" else dev.to_device
    left = [(wrap(k), wrap(p)) for k, p in zip(cut(fk), cut(y))]
    right = [(wrap(k), wrap(p)) for k, p in zip(cut(pk), cut(x))]
    with pytest.raises(o.OlapError):
        o.JoinGpu(dev, left, right).Prepare().Run()
    j = o.JoinGpu(dev, left, right, drop_unmatched=True).Prepare()
    r = j.Run()
    get = (lambda a: a) if space == "host" else dev.to_host
    ofk, oy, ox = get(r["fk"]), get(r["y"]), get(r["x"])
    keep = (fk % 2 == 0) & (fk < 2 * nr)
    assert j.stats["missing_keys"] == int((~keep).sum()) and ofk.size == int(keep.sum())
    lut = np.zeros(2 * nr, np.uint32)
    lut[pk] = x
    np.testing.assert_array_equal(np.sort(oy), y[keep])                     # exactly the matching left rows, once each
    np.testing.assert_array_equal(fk[oy], ofk)
    np.testing.assert_array_equal(lut[ofk], ox)
    # still fine when nothing is missing, and with an empty build side
    r2 = o.JoinGpu(dev, [(wrap(pk[:1000].copy()), wrap(x[:1000].copy()))], right, drop_unmatched=True).Prepare().Run()
    assert get(r2["fk"]).size == 1000
    r3 = o.JoinGpu(dev, left[:1], [], drop_unmatched=True).Prepare().Run()
    assert get(r3["fk"]).size == 0


@pytest.mark.parametrize("tpt", [8, 4])
@pytest.mark.parametrize("shape", ["uniform", "clustered_left", "clustered_both", "runs", "ragged"])
def test_join_ring_scatter_vs_digest(dev, shape, tpt):
    """The ring scatter (csrc/ring_scatter.h: per-bucket LDS rings, 128-byte lines, producer and
    flusher waves) as claiming pass 0 AND as capacity-layout last pass, forced at 2^25 rows with an
    8 + 8 bit plan.  clustered_*: the rows are ordered by their hash, so every tile falls into ONE
    bucket -- each tile takes the hot path (ring overflow -> direct range writes) while the bucket
    sizes stay balanced, i.e. nothing overflows and the ring kernels' output is what is checked;
    runs: bursts of 64 rows per bucket; ragged: odd batch lengths.  Checked: row digest, that the two
    ring phases ran (and no histogram pass), partition-major output."""
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows, nb = ((1 << 20), 32) if shape != "ragged" else ((1 << 20) - 4321, 33)
    n = rows * nb
    rng = np.random.default_rng(7)
    pk = np.arange(n, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(12345)   # a permutation of distinct keys
    x = dg.uniform_u32(n, dg.SEED, dg.STREAM_X, 0)
    fk = pk[rng.permutation(n)]
    fk[: n // 16] = fk[n // 16 : 2 * (n // 16)]     # some repeated foreign keys
    y = dg.uniform_u32(n, dg.SEED, dg.STREAM_Y, 0)
    if shape.startswith("clustered"):
        order = np.argsort(oracle.wang_hash_np(fk), kind="stable")
        fk, y = fk[order], y[order]
        if shape == "clustered_both":
            order = np.argsort(oracle.wang_hash_np(pk), kind="stable")
            pk, x = pk[order], x[order]
    elif shape == "runs":   # bursts: blocks of 64 consecutive rows share their top 8 hash bits
        order = np.argsort(oracle.wang_hash_np(fk) >> np.uint32(24), kind="stable")
        blocks = order.reshape(-1, 64)
        blocks = blocks[rng.permutation(blocks.shape[0])]
        fk, y = fk[blocks.reshape(-1)], y[blocks.reshape(-1)]
    xs = dg.uniform_u32(n, dg.SEED, dg.STREAM_X, 0)                              # x of key number i, before any reordering
    exp_x = xs[(fk - np.uint32(12345)) * np.uint32(pow(2654435761, -1, 1 << 32))]   # i = (key - 12345) / 2654435761 mod 2^32
    d = {k: dev.to_device(v) for k, v in (("fk", fk), ("y", y), ("pk", pk), ("x", x), ("exp", exp_x))}
    expect = dev.checksum_rows3_u32(d["fk"].ptr, d["y"].ptr, d["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(d[c].ptr + 4 * b * rows, rows, "uint32")
    dev.join_configure(16, 8)
    dev.set_tuning("claim_mode", 2)
    dev.set_tuning("ring_scatter", 3)
    dev.set_tuning("ring_tpt", tpt)
    dev.reset_timers()
    try:
        j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(nb)],
                      [(batch("pk", b), batch("x", b)) for b in range(nb)]).Prepare()
        out = j.Run()
        timers = j.Timers()
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("claim_mode", 1)
        dev.set_tuning("ring_tpt", 8)
    assert out["fk"].rows == n and j.stats["missing_keys"] == 0, j.stats
    assert j.stats["radix_bits"][:2] == [8, 8], j.stats
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
    assert timers.get("partition-scatter-claim", 0) and timers.get("partition-scatter-capacity", 0), timers
    assert dev.get_counter("join_attempts") == 1 and dev.get_counter("join_region_overflows") == 0   # no exact rerun behind the result
    assert not timers.get("partition-hist", 0), timers
    # whole aligned 4096-row tiles are read by the straight-line loader (last template argument 1), anything else by the generic one
    # (ragged batches: two launches, whole tiles through the straight-line loader and the tails through the generic one, which
    # is the last one launched)
    claim_kernel = dev.phase_kernel("partition-scatter-claim")
    assert claim_kernel.endswith(", 1>" if (tpt == 8 and shape != "ragged") else ", 0>"), claim_kernel
    if shape == "ragged" and tpt == 8:   # the same digest with the split off (one launch, generic loader)
        dev.join_configure(16, 8)
        dev.set_tuning("claim_mode", 2)
        dev.set_tuning("ring_split_ragged", 0)
        try:
            out0 = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(nb)], [(batch("pk", b), batch("x", b)) for b in range(nb)]).Prepare().Run()
        finally:
            dev.join_configure(-1, 0)
            dev.set_tuning("claim_mode", 1)
            dev.set_tuning("ring_split_ragged", 1)
        assert dev.checksum_rows3_u32(out0["fk"].ptr, out0["y"].ptr, out0["x"].ptr, n) == expect
        for c in out0.values():
            dev.release(o.DeviceArray(c.ptr, 0))
    ofk = dev.to_host(out["fk"])
    part = oracle.wang_hash_np(ofk) >> np.uint32(16)
    assert (np.diff(part.astype(np.int64)) >= 0).all()
    if shape == "uniform" and tpt == 8:
        # selection vectors (no payload columns: payload = row index, loader variant 2), and the generic loader on the same input
        for knob, suffix in ((1, ", 2>"), (0, ", 0>")):
            dev.join_configure(16, 8)
            dev.set_tuning("claim_mode", 2)
            dev.set_tuning("ring_full_tiles", knob)
            try:
                js = o.JoinGpu(dev, [batch("fk", b) for b in range(nb)], [batch("pk", b) for b in range(nb)], select=True).Prepare()
                sel = js.Run()
            finally:
                dev.join_configure(-1, 0)
                dev.set_tuning("claim_mode", 1)
                dev.set_tuning("ring_full_tiles", 1)
            assert dev.phase_kernel("partition-scatter-claim").endswith(suffix)
            sfk, sy, sx = dev.to_host(sel["fk"]), dev.to_host(sel["y"]), dev.to_host(sel["x"])
            np.testing.assert_array_equal(fk[sy], sfk)
            np.testing.assert_array_equal(pk[sx], sfk)
            assert np.array_equal(np.sort(sy), np.arange(n, dtype=np.uint32))
            for c in sel.values():
                dev.release(o.DeviceArray(c.ptr, 0))
    for c in list(d.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


@pytest.mark.parametrize("ring,weave", [(1, 1), (1, 0), (2, 1), (3, 0)])
def test_join_ring_scatter_mixed_with_write_combining_kernels(dev, ring, weave):
    """ring_scatter = 1: the claiming ring scatter (woven or plain regions) feeds the write-combining capacity-layout kernel;
    2: the write-combining claiming kernel feeds the ring scatter; 3 with claim_weave = 0: both ring passes on plain
    regions.  Same digest in every combination."""
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows, nb = 1 << 20, 32
    n = rows * nb
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    dev.join_configure(16, 8)
    dev.set_tuning("claim_mode", 2)
    dev.set_tuning("ring_scatter", ring)
    dev.set_tuning("claim_weave", weave)
    try:
        j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(nb)],
                      [(batch("pk", b), batch("x", b)) for b in range(nb)]).Prepare()
        out = j.Run()
        timers = j.Timers()
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("claim_mode", 1)
        dev.set_tuning("ring_scatter", 3)
        dev.set_tuning("claim_weave", 1)
    assert out["fk"].rows == n and j.stats["missing_keys"] == 0, j.stats
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
    assert timers.get("partition-scatter-claim", 0) and timers.get("partition-scatter-capacity", 0) and not timers.get("partition-hist", 0), timers
    assert dev.get_counter("join_attempts") == 1
    claim_k, cap_k = dev.phase_kernel("partition-scatter-claim"), dev.phase_kernel("partition-scatter-capacity")
    # (the claiming ring scatter from columns is k_part_claim_ring, csrc/ring_claim.h; k_part_scatter_ring<0, true, ..> with claim_lean = 0)
    assert ("k_part_claim_ring" in claim_k or "k_part_scatter_ring" in claim_k) == bool(ring & 1) and ("k_part_scatter_ring" in cap_k) == bool(ring & 2), (claim_k, cap_k)
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


def test_join_ring_scatter_skew_falls_back_to_exact(dev):
    """Ring scatter with a skewed probe side (1024 distinct foreign keys): the claimed regions / capacity
    regions overflow, the kernels drop what does not fit and raise the overflow counter, and the join is
    redone with exact passes.  The digest must match."""
    o = _ops()
    from dpu_olap_amd import datagen as dg
    rows, nb = 1 << 20, 32
    n = rows * nb
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, n, 1024)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    batch = lambda c, b: o.DeviceArray(cols[c].ptr + 4 * b * rows, rows, "uint32")
    dev.join_configure(16, 8)
    dev.set_tuning("claim_mode", 2)
    try:
        j = o.JoinGpu(dev, [(batch("fk", b), batch("y", b)) for b in range(nb)],
                      [(batch("pk", b), batch("x", b)) for b in range(nb)]).Prepare()
        out = j.Run()
    finally:
        dev.join_configure(-1, 0)
        dev.set_tuning("claim_mode", 1)
    assert out["fk"].rows == n
    assert dev.get_counter("join_attempts") == 2 and dev.get_counter("join_region_overflows") > 0   # the fallback did run
    assert dev.checksum_rows3_u32(out["fk"].ptr, out["y"].ptr, out["x"].ptr, n) == expect
    for c in list(cols.values()) + list(out.values()):
        dev.release(o.DeviceArray(c.ptr, 0))


@pytest.mark.parametrize("ranks", [1, 2, 4, 8])
@pytest.mark.parametrize("case", ["a", "b"])
def test_group_join_loopback_vs_golden(golden, ranks, case):
    """olap_group_*: the join across a device group behind the C ABI, here as `ranks` logical ranks on the ONE device of
    a gpurun box (loopback: the exchange is same-device copies; on a multi-GPU host the same call moves the runs with
    RCCL or peer copies).  Host batches in, host columns out (rank-major); rows must equal the Arrow golden join, and
    every rank must only return keys it owns (top log2 G bits of the Wang hash)."""
    from dpu_olap_amd import capi
    fk, y, pk, x = (golden[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))
    lb, rb = split_batches(fk, golden[f"join_{case}_left_rows"]), split_batches(pk, golden[f"join_{case}_right_rows"])
    yb, xb = split_batches(y, golden[f"join_{case}_left_rows"]), split_batches(x, golden[f"join_{case}_right_rows"])
    g = capi.Group([0] * ranks)
    try:
        assert g.size == ranks
        out = [np.empty(fk.size, dtype=np.uint32) for _ in range(3)]
        ptrs = lambda bs: [b.ctypes.data for b in bs]
        rows, per_rank, st = g.join_u32(ptrs(lb), ptrs(yb), [b.size for b in lb], ptrs(rb), ptrs(xb), [b.size for b in rb],
                                        capi.MEM_HOST, *(o.ctypes.data for o in out), fk.size)
        assert rows == fk.size == sum(per_rank) and st["missing_keys"] == 0 and st["left_rows"] == fk.size
        if ranks > 1:
            assert g.last_transport() == capi.TRANSPORT_COPY and set(g.last_phase_ms()) == {"split", "exchange", "join", "copy-out"}
            owner = oracle.wang_hash_np(out[0]) >> np.uint32(32 - (ranks.bit_length() - 1))
            assert (np.diff(owner.astype(np.int64)) >= 0).all()                       # rank-major
            assert np.bincount(owner, minlength=ranks).tolist() == per_rank            # and every rank holds exactly the keys it owns
        got = sort_rows(*out)
        for a, k in zip(got, ("fk", "y", "x")):
            np.testing.assert_array_equal(a, golden[f"join_{case}_out_{k}"])
        if ranks > 1:   # RCCL needs one device per rank: asked for in loopback it is refused, not silently replaced
            g.set_transport(capi.TRANSPORT_RCCL)
            with pytest.raises(capi.OlapError) as e:
                g.join_u32(ptrs(lb), ptrs(yb), [b.size for b in lb], ptrs(rb), ptrs(xb), [b.size for b in rb],
                           capi.MEM_HOST, *(o.ctypes.data for o in out), fk.size)
            assert e.value.code == capi.ERR_INVALID and "one device per rank" in str(e.value)
    finally:
        g.close()


def test_group_join_device_resident_generator_shape(dev):
    """Group join on device-resident batches at benchmark shape (SF=16, 4 loopback ranks): digest of the rows against the
    digest predicted from the inputs."""
    from dpu_olap_amd import capi, datagen as dg
    rows, sf, ranks = 2 << 20, 16, 4
    n = rows * sf
    o = _ops()
    cols = {k: dev.empty(n, "uint32") for k in ("fk", "y", "pk", "x", "exp")}
    dev.gen_fk_u32(cols["fk"].ptr, n, dg.SEED, dg.STREAM_FK, 0, rows, rows)
    dev.gen_uniform_u32(cols["y"].ptr, n, dg.SEED, dg.STREAM_Y, 0)
    dev.gen_index_u32(cols["pk"].ptr, n, 0)
    dev.gen_uniform_u32(cols["x"].ptr, n, dg.SEED, dg.STREAM_X, 0)
    dev.gen_uniform_u32_at(cols["exp"].ptr, cols["fk"].ptr, n, dg.SEED, dg.STREAM_X)
    expect = dev.checksum_rows3_u32(cols["fk"].ptr, cols["y"].ptr, cols["exp"].ptr, n)
    dev.synchronize()
    g = capi.Group([0] * ranks)
    try:
        out = [np.empty(n, dtype=np.uint32) for _ in range(3)]
        bp = lambda c: [cols[c].ptr + 4 * b * rows for b in range(sf)]
        got_rows, per_rank, st = g.join_u32(bp("fk"), bp("y"), [rows] * sf, bp("pk"), bp("x"), [rows] * sf, capi.MEM_DEVICE,
                                            *(a.ctypes.data for a in out), n)
        assert got_rows == n and st["missing_keys"] == 0 and abs(per_rank[0] / n - 0.25) < 0.01
    finally:
        g.close()
    d = [dev.to_device(a) for a in out]
    assert dev.checksum_rows3_u32(d[0].ptr, d[1].ptr, d[2].ptr, n) == expect
    for c in list(cols.values()) + d:
        dev.release(o.DeviceArray(c.ptr, 0))


# ------------------------------------------------------------ 64-bit join keys
@pytest.mark.parametrize("case", ["c", "d"])
@pytest.mark.parametrize("bits,space", [(-1, "host"), (-1, "device"), (7, "host"), (12, "device"), (20, "host")])
def test_join_u64_vs_arrow_golden(dev, golden, case, bits, space):
    """olap_join_u64 (HT_64BIT_KEYS, hashtable.h:13-18): uint64 keys incl. 0, 2^64 - 1 (the LDS table's empty marker) and
    keys that differ only above bit 32 (case c), int64 keys incl. negatives (case d), through 1-, 2- and 3-pass plans
    of 16-byte tuples; rows must equal Arrow's hash join."""
    o = _ops()
    fk, y, pk, x = (golden[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))
    wrap = (lambda a: a) if space == "host" else dev.to_device
    lb = [(wrap(k), wrap(v)) for k, v in zip(split_batches(fk, golden[f"join_{case}_left_rows"]), split_batches(y, golden[f"join_{case}_left_rows"]))]
    rb = [(wrap(k), wrap(v)) for k, v in zip(split_batches(pk, golden[f"join_{case}_right_rows"]), split_batches(x, golden[f"join_{case}_right_rows"]))]
    dev.join_configure(bits, 0)
    try:
        j = o.JoinGpu(dev, lb, rb).Prepare()
        out = j.Run()
    finally:
        dev.join_configure(-1, 0)
    get = (lambda a: a) if space == "host" else dev.to_host
    got = sort_rows(get(out["fk"]).view(fk.dtype), get(out["y"]), get(out["x"]))
    assert j.stats["missing_keys"] == 0 and j.stats["left_rows"] == fk.size and j.stats["table_slots"] == 4096
    if bits > 0:
        assert sum(j.stats["radix_bits"]) == bits and j.stats["passes"] == (bits + 8) // 9
    for g, k in zip(got, ("fk", "y", "x")):
        np.testing.assert_array_equal(g, golden[f"join_{case}_out_{k}"])


def test_join_u64_hot_foreign_key_probe_slices(dev):
    """64-bit keys with a hot foreign key (40 % of 2 M left rows) and a hot key WITHOUT a match (10 %): the hot partitions'
    probe rows are cut into slices shared by several workgroups (as in the 32-bit kernels).  Rows against numpy; the
    same with probe_slices = 0; a duplicate build key is counted once."""
    o = _ops()
    rng = np.random.default_rng(640)
    nr, nl = 1 << 19, 2_000_000
    pk = ((rng.permutation(nr).astype(np.uint64) << np.uint64(30)) | rng.integers(0, 1 << 29, size=nr, dtype=np.uint64)) << np.uint64(1)   # unique, even
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=nl)]
    u = rng.random(nl)
    fk[u < 0.4] = pk[77]
    fk[(u >= 0.4) & (u < 0.5)] = np.uint64(0x123456789ABCDEF1)                # odd: no such primary key
    y = np.arange(nl, dtype=np.uint32)
    pk2, x2 = np.concatenate([pk, pk[77:78]]), np.concatenate([x, x[77:78]])
    keep = (fk & np.uint64(1)) == 0
    lut = dict(zip(pk.tolist(), x.tolist()))
    exp_x = np.array([lut[k] for k in fk[keep].tolist()], dtype=np.uint32)
    cut = lambda a: np.array_split(a, 3)
    left, right = list(zip(cut(fk), cut(y))), list(zip(cut(pk2), cut(x2)))
    for knob in (1, 0):
        dev.set_tuning("probe_slices", knob)
        try:
            j = o.JoinGpu(dev, left, right, drop_unmatched=True).Prepare()
            r = j.Run()
        finally:
            dev.set_tuning("probe_slices", 1)
        assert j.stats["missing_keys"] == int((~keep).sum()) and j.stats["duplicate_keys"] == 1, j.stats
        np.testing.assert_array_equal(np.sort(r["y"]), y[keep])
        np.testing.assert_array_equal(fk[r["y"]], r["fk"])
        order = np.argsort(r["y"])
        np.testing.assert_array_equal(r["x"][order], exp_x)


def test_join_u64_large_vs_oracle_and_error_modes(dev):
    """2^22 x 2^22 rows of 64-bit keys in ragged batches against the oracle's join (same hash, same table semantics);
    then the contract: a missing key is OLAP_ERR_NO_MATCH (or dropped in inner-join mode, order preserved), a
    duplicate build key overwrites and is counted, NULL payload tables give selection vectors."""
    o = _ops()
    from dpu_olap_amd import capi
    rng = np.random.default_rng(64)
    n = 1 << 22
    pk = (rng.permutation(n).astype(np.uint64) << np.uint64(21)) ^ rng.integers(0, 1 << 21, size=n, dtype=np.uint64)   # unique: the top bits are a permutation
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    fk = pk[rng.integers(0, n, size=n)]
    y = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    sizes = [n // 3, 5, 0, n - n // 3 - 5]
    lb = list(zip(split_batches(fk, sizes), split_batches(y, sizes)))
    rb = list(zip(split_batches(pk, sizes[::-1]), split_batches(x, sizes[::-1])))
    j = o.JoinGpu(dev, lb, rb).Prepare()
    out = j.Run()
    assert j.stats["passes"] == 2 and j.stats["max_build_rows"] <= 3072 and j.stats["missing_keys"] == 0
    exp = sort_rows(*oracle.join_u64([a for a, _ in lb], [b for _, b in lb], [a for a, _ in rb], [b for _, b in rb], nr_partitions=64))
    got = sort_rows(out["fk"], out["y"], out["x"])
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
    # partition-major output: bucket ids (top bits of wang(hash64)) never decrease
    h = ((out["fk"] >> np.uint64(33)) ^ out["fk"] ^ (out["fk"] << np.uint64(11))).astype(np.uint32)
    part = oracle.wang_hash_np(h) >> np.uint32(32 - sum(j.stats["radix_bits"]))
    assert (np.diff(part.astype(np.int64)) >= 0).all()
    # missing key
    small_pk, small_x = pk[:5000].copy(), x[:5000].copy()
    small_fk = small_pk[rng.integers(0, 5000, size=7000)]
    small_fk[[10, 4000]] = np.uint64(0xDEADBEEF00000001)
    small_y = rng.integers(0, 1 << 32, size=7000, dtype=np.uint32)
    with pytest.raises(capi.OlapError) as e:
        o.JoinGpu(dev, [(small_fk, small_y)], [(small_pk, small_x)]).Prepare().Run()
    assert e.value.code == capi.ERR_NO_MATCH
    jd = o.JoinGpu(dev, [(small_fk, small_y)], [(small_pk, small_x)], drop_unmatched=True).Prepare()
    od = jd.Run()
    keep = np.isin(small_fk, small_pk)
    assert jd.stats["missing_keys"] == 2 and od["fk"].size == 6998
    lut = dict(zip(small_pk.tolist(), small_x.tolist()))
    e3 = sort_rows(small_fk[keep], small_y[keep], np.array([lut[k] for k in small_fk[keep].tolist()], dtype=np.uint32))
    for g, ee in zip(sort_rows(od["fk"], od["y"], od["x"]), e3):
        np.testing.assert_array_equal(g, ee)
    # duplicate build key: overwrites (hashtable.c:133), counted
    dup_pk = np.concatenate([small_pk, small_pk[:3]])
    dup_x = np.concatenate([small_x, np.array([1, 2, 3], dtype=np.uint32)])
    jdup = o.JoinGpu(dev, [(small_pk[:100].copy(), small_y[:100].copy())], [(dup_pk, dup_x)]).Prepare()
    jdup.Run()
    assert jdup.stats["duplicate_keys"] == 3
    # selection vectors
    js = o.JoinGpu(dev, [small_fk[keep].copy()], [small_pk], select=True).Prepare()
    sel = js.Run()
    np.testing.assert_array_equal(small_pk[sel["x"]], sel["fk"])
    np.testing.assert_array_equal(small_fk[keep][sel["y"]], sel["fk"])


# ----------------------------------------------------------- randomized shapes
_FUZZ = int(os.environ.get("OLAP_FUZZ_SEEDS", "0"))   # soak runs: OLAP_FUZZ_SEEDS=500 pytest -m gpu -k random


@pytest.mark.parametrize("seed", range(_FUZZ or 48))
def test_join_random_shapes_and_plans_vs_oracle(dev, seed):
    """Seeded random join shapes (empty / one-row / ragged batches, full-range keys incl.
    0xFFFFFFFF and 0) under random radix plans and kernel variants (tile-staged or
    write-combining scatter, exact / capacity / claiming layouts, probing or direct table,
    both workgroup geometries, 64- or 128-byte lines, broadcast join on / off, probe slices on / off, woven or plain
    claimed regions, both column loaders, both probing-table sizes), a hot foreign key in every fifth case, host or device operands -- rows must
    equal the CPU oracle's join as a sorted set."""
    o = _ops()
    rng = np.random.default_rng(1000 + seed)
    big = seed % 6 == 0
    def sizes(k):
        s = [int(x) for x in rng.choice([0, 1, 2, 7, 4095, 4096, 4097, 9000, 20011], size=k)]
        if big and s:
            s[int(rng.integers(len(s)))] = int(rng.integers(200_000, 600_000))
        return s
    rrows, lrows = sizes(int(rng.integers(1, 6))), sizes(int(rng.integers(0, 6)))
    nr = sum(rrows)
    if nr == 0:
        rrows[0] = nr = 5
    pk = rng.choice(1 << 32, size=nr, replace=False).astype(np.uint32) if not big else (rng.permutation(nr).astype(np.uint32) + np.uint32(1)) * np.uint32(4099)
    if seed % 3 == 0:
        pk[0] = 0xFFFFFFFF
    if seed % 4 == 0 and nr > 1:
        pk[1] = 0
    x = rng.integers(0, 1 << 32, size=nr, dtype=np.uint32)
    fk = pk[rng.integers(0, nr, size=sum(lrows))]
    if seed % 5 == 0 and fk.size:                # a hot foreign key: one partition holds a third of the probe side
        fk[rng.random(fk.size) < 0.33] = pk[int(rng.integers(nr))]
    y = rng.integers(0, 1 << 32, size=fk.size, dtype=np.uint32)
    space = "device" if seed % 2 else "host"
    wrap = (lambda a: a) if space == "host" else dev.to_device
    left = [(wrap(k), wrap(p)) for k, p in zip(split_batches(fk, lrows), split_batches(y, lrows))]
    right = [(wrap(k), wrap(p)) for k, p in zip(split_batches(pk, rrows), split_batches(x, rrows))]
    bits = int(rng.choice([-1, -1, 1, 3, 6, 9, 11, 14]))
    if bits >= 0 and nr >> bits > 6000:
        bits = -1                               # keep build partitions inside the probing table
    knobs = {"scatter_variant": int(rng.integers(2)), "capacity_mode": int(rng.integers(2)), "claim_mode": int(rng.choice([0, 2])),
             "direct_table": int(rng.integers(2)), "wc_geometry": int(rng.choice([1, 2])), "line16": int(rng.integers(2)),
             "ring_scatter": int(rng.integers(4)), "ring_tpt": int(rng.choice([4, 8])), "broadcast_join": int(rng.integers(2)),
             "probe_slices": int(rng.integers(2)), "claim_weave": int(rng.integers(2)), "ring_full_tiles": int(rng.integers(2)),
             "probe_small_table": int(rng.integers(2))}
    per_pass = int(rng.choice([0, 0, 3, 5, 8, 10]))
    if per_pass and max(bits, 8) > 4 * per_pass:
        per_pass = 0                            # more than 4 passes is a configuration error
    dev.join_configure(bits, per_pass)
    for k, v in knobs.items():
        dev.set_tuning(k, v)
    try:
        j = o.JoinGpu(dev, left, right).Prepare()
        r = j.Run()
    finally:
        dev.join_configure(-1, 0)
        for k, v in {"scatter_variant": 0, "capacity_mode": 1, "claim_mode": 1, "direct_table": 1, "wc_geometry": 2, "line16": 0,
                     "ring_scatter": 3, "ring_tpt": 8, "broadcast_join": 1, "probe_slices": 1, "claim_weave": 1, "ring_full_tiles": 1,
                     "probe_small_table": 1}.items():
            dev.set_tuning(k, v)
    get = (lambda a: a) if space == "host" else dev.to_host
    got = sort_rows(get(r["fk"]), get(r["y"]), get(r["x"]))
    exp = sort_rows(*oracle.join(split_batches(fk, lrows), split_batches(y, lrows), split_batches(pk, rrows), split_batches(x, rrows),
                                 nr_partitions=4))
    assert got[0].size == fk.size, (knobs, bits, j.stats)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e, err_msg=f"{knobs} bits={bits} stats={j.stats}")


# ------------------------------------------------------------ error behaviour
def test_c_abi_rejects_misuse_with_status_codes(dev):
    """The ABI never throws or aborts: misuse comes back as a negative status
    plus a message, and the context stays usable."""
    from dpu_olap_amd import capi
    a = np.arange(1000, dtype=np.uint32)
    d = dev.to_device(a)
    out = dev.empty(1000, "uint32")

    def code(fn):
        with pytest.raises(capi.OlapError) as e:
            fn()
        assert e.value.message
        return e.value.code

    assert code(lambda: dev.filter_lt(99, [d.ptr], [1000], capi.MEM_DEVICE, 5, [out.ptr], capi.MEM_DEVICE)) == capi.ERR_INVALID    # dtype
    assert code(lambda: dev.filter_lt(capi.U32, [d.ptr], [1000], 7, 5, [out.ptr], capi.MEM_DEVICE)) == capi.ERR_INVALID          # memory space
    assert code(lambda: dev.filter_lt(capi.U32, [d.ptr], [1000], capi.MEM_DEVICE, 5, [0], capi.MEM_DEVICE)) == capi.ERR_INVALID   # null output
    assert code(lambda: dev.filter_lt(capi.U64, [d.ptr + 4], [100], capi.MEM_DEVICE, 5, [out.ptr], capi.MEM_DEVICE)) == capi.ERR_INVALID  # misaligned
    assert code(lambda: dev.partition_u32([d.ptr], None, [1000], capi.MEM_DEVICE, 3, out.ptr, out.ptr, 1000, capi.MEM_DEVICE)) == capi.ERR_INVALID  # P not 2^k
    assert code(lambda: dev.partition_u32([d.ptr], None, [1000], capi.MEM_DEVICE, 4, out.ptr, out.ptr, 10, capi.MEM_DEVICE)) == capi.ERR_CAPACITY
    assert code(lambda: dev.join_u32([d.ptr], [d.ptr], [1000], [d.ptr], [d.ptr], [1000], capi.MEM_DEVICE, out.ptr, out.ptr, out.ptr, 10,
                                     capi.MEM_DEVICE)) == capi.ERR_CAPACITY
    assert code(lambda: dev.join_tuples_u32(d.ptr + 8, 10, d.ptr, 10, out.ptr, out.ptr, out.ptr, 1000)) == capi.ERR_INVALID            # 16-byte alignment
    assert code(lambda: dev.join_presplit_u32(d.ptr, 10, [0, 5], [5, 99], d.ptr, 10, [0, 5], [5, 10], 2, 1, out.ptr, out.ptr, out.ptr,
                                              1000)) == capi.ERR_INVALID                                                               # chunk outside the array
    assert code(lambda: dev.join_configure(31, 0)) == capi.ERR_INVALID
    assert code(lambda: dev.set_tuning("no_such_knob", 1)) == capi.ERR_INVALID
    assert code(lambda: dev.join_set_hash_skip_bits(40)) == capi.ERR_INVALID
    # the context still works
    counts, total = dev.filter_lt(capi.U32, [d.ptr], [1000], capi.MEM_DEVICE, 10, [out.ptr], capi.MEM_DEVICE)
    assert total == 10 and int(counts[0]) == 10


@pytest.mark.parametrize("obits,gbits,lbits", [(1, 2, 6), (3, 2, 4), (2, 0, 7), (0, 3, 3)])
def test_split_tuples_grouped_bucket_order(dev, obits, gbits, lbits):
    """olap_split_tuples_u32: the same buckets as olap_partition_tuples_u32 with 2^(o+g+l) partitions
    (the top hash bits), laid out [group | owner | low] instead of [owner | group | low]."""
    import torch
    from dpu_olap_amd.dist_join import GpuLocalOps
    device = torch.device("cuda", 0)
    ops = GpuLocalOps(dev, device)
    rng = np.random.default_rng(obits * 100 + gbits * 10 + lbits)
    sizes = [70001, 0, 12345, 300000]
    keys = [torch.from_numpy(rng.integers(0, 1 << 32, size=n, dtype=np.uint32).view(np.int32)).to(device) for n in sizes]
    pay = [torch.from_numpy(rng.integers(0, 1 << 32, size=n, dtype=np.uint32).view(np.int32)).to(device) for n in sizes]
    bits = obits + gbits + lbits
    nat, noff = ops.partition_tuples(keys, pay, 1 << bits)
    grp, goff = ops.split_tuples(keys, pay, obits, gbits, lbits)
    nat, grp = nat.cpu().numpy().view(np.uint64), grp.cpu().numpy().view(np.uint64)
    assert goff[-1] == noff[-1] == sum(sizes)
    k = 0
    for g in range(1 << gbits):
        for o in range(1 << obits):
            for l in range(1 << lbits):
                b = (o << (gbits + lbits)) | (g << lbits) | l
                a = np.sort(grp[goff[k]:goff[k + 1]])
                e = np.sort(nat[noff[b]:noff[b + 1]])
                np.testing.assert_array_equal(a, e)
                h = oracle.wang_hash_np((a & np.uint64(0xFFFFFFFF)).astype(np.uint32))
                assert ((h >> np.uint32(32 - bits)) == b).all()
                k += 1


@pytest.mark.parametrize("seed", range(_FUZZ or 24))
def test_scan_ops_random_shapes_vs_oracle(dev, seed):
    """Seeded random batch shapes (empty, one-row, tile-boundary, ragged, a long batch), the four
    element types, host or device operands: filter / sum / take / partition against the oracle
    (numpy for the 64-bit and signed types, whose semantics SURVEY 8 pins to Arrow's)."""
    o = _ops()
    rng = np.random.default_rng(500 + seed)
    dt = [np.uint32, np.int32, np.uint64, np.int64][seed % 4]
    sizes = [int(x) for x in rng.choice([0, 1, 3, 255, 256, 1023, 4096, 4097, 16383, 65536, 65537, 100003], size=int(rng.integers(1, 8)))]
    if seed % 5 == 0:
        sizes[int(rng.integers(len(sizes)))] = int(rng.integers(300_000, 1_500_000))
    info = np.iinfo(dt)
    batches = [rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True) for n in sizes]
    thr = int(rng.integers(info.min, info.max, dtype=dt, endpoint=True)) if seed % 3 else int(info.min)   # also "nothing passes"
    space = "device" if seed % 2 else "host"
    wrap = (lambda a: a) if space == "host" else dev.to_device
    get = (lambda a: a) if space == "host" else dev.to_host
    ins = [wrap(b) for b in batches]
    # filter: order-preserving v < thr in the column's own type
    res = o.FilterGpu(dev, ins, threshold=thr).Prepare().GetResult()
    assert len(res) == len(batches)
    for r, b in zip(res, batches):
        np.testing.assert_array_equal(get(r), b[b < dt(thr)])
    # sum: 64-bit accumulator, wraps mod 2^64 (signed: two's complement)
    want = int(sum(int(b.astype(np.int64 if info.min < 0 else np.uint64).sum(dtype=np.int64 if info.min < 0 else np.uint64)) for b in batches))
    got = o.SumGpu(dev, ins).Prepare().Run()
    mod = 1 << 64
    assert (got - want) % mod == 0
    # take: per batch, no bounds check needed for in-range indices
    data = [b for b in batches if b.size]
    if data:
        idx = [rng.integers(0, d.size, size=int(rng.choice([0, 1, 7, 4096, 8192, 8193, 50000])), dtype=np.uint32) for d in data]
        out = o.TakeGpu(dev, [wrap(d) for d in data], [wrap(i) for i in idx]).Prepare().Run()
        for r, d, i in zip(out, data, idx):
            np.testing.assert_array_equal(get(r), d[i])
    # partition (u32 keys): bucket = top bits of the Wang hash; contents per bucket as a set of (key, payload)
    keys = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]
    pay = [rng.integers(0, 1 << 32, size=n, dtype=np.uint32) for n in sizes]
    P = int(rng.choice([2, 4, 32, 256, 1024, 4096]))
    parts = o.PartitionGpu(dev, [(wrap(k), wrap(p)) for k, p in zip(keys, pay)], P).Prepare().Run()
    allk, allp = np.concatenate(keys), np.concatenate(pay)
    bucket = oracle.wang_hash_np(allk) >> np.uint32(32 - (P.bit_length() - 1))
    ok_, _, hist_end = oracle.partition(allk, P)
    assert [int(h) for h in hist_end] == list(np.cumsum([pk.size for pk, _ in parts]))       # the oracle's END offsets
    for p in (0, P // 2, P - 1, int(rng.integers(P))):
        gk, gp = parts[p]
        sel = bucket == p
        a = sort_rows(gk, gp)
        e = sort_rows(allk[sel], allp[sel])
        np.testing.assert_array_equal(a[0], e[0])
        np.testing.assert_array_equal(a[1], e[1])


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_join_into_misaligned_result_slices(dev, golden, shift):
    """The result columns may be slices of a larger result (the per-group joins of the multi-GPU path write
    at arbitrary row offsets): every 4-byte alignment of the output pointers gives the same rows."""
    fk, y, pk, x = (golden[f"join_b_{k}"] for k in ("fk", "y", "pk", "x"))
    d = {k: dev.to_device(v) for k, v in (("fk", fk), ("y", y), ("pk", pk), ("x", x))}
    n = fk.size
    out = [dev.empty(n + 4, "uint32") for _ in range(3)]
    offs = [shift, shift, shift] if shift < 3 else [0, 1, 2]          # 3: mixed alignments
    from dpu_olap_amd import capi
    rows, st = dev.join_u32([d["fk"].ptr], [d["y"].ptr], [n], [d["pk"].ptr], [d["x"].ptr], [pk.size], capi.MEM_DEVICE,
                            out[0].ptr + 4 * offs[0], out[1].ptr + 4 * offs[1], out[2].ptr + 4 * offs[2], n, capi.MEM_DEVICE)
    assert rows == n
    got = [dev.to_host(o)[k:k + n] for o, k in zip(out, offs)]
    a, b_, c = sort_rows(*got)
    np.testing.assert_array_equal(a, golden["join_b_out_fk"])
    np.testing.assert_array_equal(b_, golden["join_b_out_y"])
    np.testing.assert_array_equal(c, golden["join_b_out_x"])
