"""Pins the CPU oracle (oracle/olap_oracle.c) against (1) the known-answer data
held by the reference's own tests (tests/golden/reference_kats.json) and (2)
vectors produced by Apache Arrow, the library behind the reference's *Native
operators (tests/golden/arrow_golden.npz, made by tests/golden/make_golden.py).
CPU only."""
import numpy as np
import pytest

import oracle
from conftest import sort_rows, split_batches

TYPES = ["u32", "i32", "u64", "i64"]


def test_wang_hash_kats(kats):
    for k, v in kats["wang_hash"]["values"].items():
        assert oracle.wang_hash(int(k)) == int(v, 16)
    for k, b in kats["wang_hash"]["bucket_P2"].items():
        assert oracle.bucket_of(int(k), 2) == b
    for k, b in kats["wang_hash"]["bucket_P2048"].items():
        assert oracle.bucket_of(int(k), 2048) == b


def test_wang_hash_is_a_bijection_on_a_sample_and_numpy_agrees():
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 1 << 32, size=100000, dtype=np.uint32)
    h = oracle.wang_hash_np(keys)
    for k in keys[:200]:
        assert oracle.wang_hash(int(k)) == int(oracle.wang_hash_np(np.array([k], dtype=np.uint32))[0])
    # distinct keys -> distinct hashes (every step of partition.c:20-28 is invertible mod 2^32)
    assert np.unique(h).size == np.unique(keys).size


def test_filter_kats(kats, golden):
    small = np.array(kats["small_column"], dtype=np.uint32)
    out = oracle.filter_lt(small, kats["filter_simple_test"]["threshold"])
    assert out.size == kats["filter_simple_test"]["expected_count"]
    np.testing.assert_array_equal(out, golden["kat_filter_small"])
    frt = kats["filter_result_test"]
    pat = np.arange(frt["rows"], dtype=np.uint32) + np.uint32(1 << 30)
    for i in frt["small_positions"]:
        pat[i] = i
    out = oracle.filter_lt(pat, 1 << 30)
    np.testing.assert_array_equal(out, np.array(frt["expected"], dtype=np.uint32))
    np.testing.assert_array_equal(out, golden["kat_filter_pattern"])
    np.testing.assert_array_equal(oracle.filter_lt(golden["edge_u32_in"], 1 << 30), golden["edge_u32_out"])


def test_take_sum_kats(kats, golden):
    small = np.array(kats["small_column"], dtype=np.uint32)
    idx = np.array(kats["take_indices"], dtype=np.uint32)
    np.testing.assert_array_equal(oracle.take(small, idx), np.array(kats["take_simple_test"]["expected"], dtype=np.uint32))
    np.testing.assert_array_equal(oracle.take(small, idx), golden["kat_take_small"])
    assert oracle.sum_(small) == kats["sum_simple_test"]["expected"] == int(golden["kat_sum_small"][0])


@pytest.mark.parametrize("t", TYPES)
def test_filter_sum_take_vs_arrow(golden, t):
    vals = golden[f"rand_{t}_values"]
    sizes = golden[f"rand_{t}_sizes"]
    thr = golden[f"rand_{t}_thr"][0]
    batches = split_batches(vals, sizes)
    outs = [oracle.filter_lt(b, thr.item()) for b in batches]
    np.testing.assert_array_equal(np.array([o.size for o in outs], dtype=np.uint64), golden[f"rand_{t}_filter_counts"])
    np.testing.assert_array_equal(np.concatenate(outs), golden[f"rand_{t}_filter_out"])
    assert oracle.sum_(vals) == int(golden[f"rand_{t}_sum"][0])
    data = batches[-1]
    np.testing.assert_array_equal(oracle.take(data, golden[f"rand_{t}_take_idx"]), golden[f"rand_{t}_take_out"])


def test_sum_wraps_like_arrow():
    v = np.array([(1 << 64) - 1, 5], dtype=np.uint64)
    assert oracle.sum_(v) == 4
    w = np.array([(1 << 63) - 1, 1], dtype=np.int64)
    assert oracle.sum_(w) == -(1 << 63)


def test_partition_kat(kats):
    p = kats["partition_simple_test"]
    pk = np.array(p["pk"], dtype=np.uint32)
    x = np.array(p["x"], dtype=np.uint32)
    ok, orow, hist = oracle.partition(pk, p["nr_partitions"])
    sizes = np.diff(np.concatenate([[0], hist]))
    assert sizes.tolist() == p["expected_sizes"]
    assert int(ok.astype(np.uint64).sum()) == p["expected_sum_pk"]
    assert int(oracle.take(x, orow).astype(np.uint64).sum()) == p["expected_sum_x"]
    np.testing.assert_array_equal(pk[orow], ok)


@pytest.mark.parametrize("P", [2, 32, 2048])
@pytest.mark.parametrize("n", [0, 1, 1000, 65536])
def test_partition_contract(P, n):
    rng = np.random.default_rng(P * 131 + n)
    keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    ok, orow, hist = oracle.partition(keys, P)
    assert hist[-1] == n if n else (hist == 0).all()
    starts = np.concatenate([[0], hist[:-1]]).astype(np.int64)
    b = oracle.wang_hash_np(ok) >> np.uint32(32 - int(np.log2(P)))
    for p in range(P):
        assert (b[starts[p] : hist[p]] == p).all()
    np.testing.assert_array_equal(keys[orow], ok)
    assert np.array_equal(np.sort(orow), np.arange(n, dtype=np.uint32))
    if n:
        # rotation (partition.c:139-147): bucket o_shift moves to the front
        s = P // 2
        ok2, orow2, hist2 = oracle.partition(keys, P, o_shift=s)
        np.testing.assert_array_equal(hist2, hist)
        rot = int(starts[s])
        np.testing.assert_array_equal(np.roll(ok, -rot), ok2)
        np.testing.assert_array_equal(np.roll(orow, -rot), orow2)


def test_partition_balance_like_reference_large_test(kats):
    """host/partition/partition_test.cc:60-92 at reduced batch count (same per-partition expectation)."""
    c = kats["partition_large_test"]
    rng = np.random.default_rng(42)
    nb = 8
    counts = np.zeros(c["nr_partitions"], dtype=np.int64)
    for _ in range(nb):
        keys = rng.integers(0, 1 << 32, size=c["batch_size"], dtype=np.uint32)
        _, _, hist = oracle.partition(keys, c["nr_partitions"])
        counts += np.diff(np.concatenate([[0], hist]))
    mean = nb * c["batch_size"] / c["nr_partitions"]
    assert counts.sum() == nb * c["batch_size"]
    assert (np.abs(counts - mean) / mean <= c["tolerance"]).all()


def test_partition_rejects_bad_P():
    with pytest.raises(ValueError):
        oracle.partition(np.zeros(4, dtype=np.uint32), 3)
    with pytest.raises(ValueError):
        oracle.partition(np.zeros(4, dtype=np.uint32), 1)


def test_join_simple_test_kat(kats, golden):
    j = kats["join_simple_test"]
    a = lambda k: [np.array(b, dtype=np.uint32) for b in j[k]]
    fk, y, x = oracle.join(a("left_fk"), a("left_v_l"), a("right_pk"), a("right_v_r"), nr_partitions=2)
    y, fk, x = sort_rows(y, fk, x)  # sorted by (v_l, fk), join_test.cc:76-77
    np.testing.assert_array_equal(fk, np.array(j["expected_sorted_fk"], dtype=np.uint32))
    np.testing.assert_array_equal(y, np.array(j["expected_sorted_v_l"], dtype=np.uint32))
    np.testing.assert_array_equal(x, np.array(j["expected_sorted_v_r"], dtype=np.uint32))
    gfk, gy, gx = sort_rows(golden["kat_join_fk"], golden["kat_join_y"], golden["kat_join_x"])
    fk, y, x = sort_rows(fk, y, x)
    np.testing.assert_array_equal(np.stack([fk, y, x]), np.stack([gfk, gy, gx]))


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("P", [2, 16, 256])
def test_join_vs_arrow(golden, case, P):
    lf = split_batches(golden[f"join_{case}_fk"], golden[f"join_{case}_left_rows"])
    ly = split_batches(golden[f"join_{case}_y"], golden[f"join_{case}_left_rows"])
    rp = split_batches(golden[f"join_{case}_pk"], golden[f"join_{case}_right_rows"])
    rx = split_batches(golden[f"join_{case}_x"], golden[f"join_{case}_right_rows"])
    fk, y, x = oracle.join(lf, ly, rp, rx, nr_partitions=P)
    assert fk.size == golden[f"join_{case}_fk"].size
    fk, y, x = sort_rows(fk, y, x)
    np.testing.assert_array_equal(fk, golden[f"join_{case}_out_fk"])
    np.testing.assert_array_equal(y, golden[f"join_{case}_out_y"])
    np.testing.assert_array_equal(x, golden[f"join_{case}_out_x"])
    # canonical index form: x == right_x[ridx] in original left order
    ridx = golden[f"join_{case}_ridx"]
    np.testing.assert_array_equal(golden[f"join_{case}_pk"][ridx], golden[f"join_{case}_fk"])
    cf, cy, cx = sort_rows(golden[f"join_{case}_fk"], golden[f"join_{case}_y"], golden[f"join_{case}_x"][ridx])
    np.testing.assert_array_equal(np.stack([cf, cy, cx]), np.stack([fk, y, x]))


def test_join_partition_major_layout_and_missing_key(golden):
    lf = [np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.uint32)]
    ly = [np.arange(8, dtype=np.uint32)]
    rp = [np.array([8, 7, 6, 5, 4, 3, 2, 1], dtype=np.uint32)]
    rx = [np.arange(10, 18, dtype=np.uint32)]
    fk, y, x = oracle.join(lf, ly, rp, rx, nr_partitions=4)
    b = oracle.wang_hash_np(fk) >> np.uint32(30)
    assert (np.diff(b.astype(np.int64)) >= 0).all()  # partition-major (join_dpu.cc:371-399)
    np.testing.assert_array_equal(x, 18 - fk)
    with pytest.raises(KeyError):
        oracle.join([np.array([99], dtype=np.uint32)], [np.array([0], dtype=np.uint32)], rp, rx, nr_partitions=2)


@pytest.mark.parametrize("case", ["c", "d"])
@pytest.mark.parametrize("P", [2, 64])
def test_join_u64_oracle_vs_arrow_golden(golden, case, P):
    """64-bit keys (the reference's HT_64BIT_KEYS table, hashtable.h:13-18 / hashtable.c:23-26): the oracle's join on
    uint64 (case c) and int64 (case d) keys equals Arrow's hash join on the same batches; hash KAT of the 64-bit
    hash function."""
    assert oracle.ht_hash_u64(0x123456789ABCDEF0) == ((0x123456789ABCDEF0 >> 33) ^ 0x123456789ABCDEF0 ^ (0x123456789ABCDEF0 << 11)) & 0xFFFFFFFF
    assert oracle.ht_hash_u64(0) == 0 and oracle.ht_hash_u64(1) == 0x801 and oracle.ht_hash_u64(1 << 33) == 1
    fk, y, pk, x = (golden[f"join_{case}_{k}"] for k in ("fk", "y", "pk", "x"))
    cut = lambda a, sizes: split_batches(a, sizes)
    ofk, oy, ox = oracle.join_u64(cut(fk, golden[f"join_{case}_left_rows"]), cut(y, golden[f"join_{case}_left_rows"]),
                                  cut(pk, golden[f"join_{case}_right_rows"]), cut(x, golden[f"join_{case}_right_rows"]), nr_partitions=P)
    got = sort_rows(ofk.view(fk.dtype), oy, ox)
    for g, k in zip(got, ("fk", "y", "x")):
        np.testing.assert_array_equal(g, golden[f"join_{case}_out_{k}"])
