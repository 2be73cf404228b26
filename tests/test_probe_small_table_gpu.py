"""GPU parity tests of the probing-table probe's two table sizes (dpu_olap_amd/csrc/probe_kernels.h k_join_probe<SLOTS>,
join.hip probe_small_table), through the C ABI: the 4096-slot table (four workgroups per CU) is taken when the build
partitions hold <= 2048 rows on average, the 8192-slot table otherwise -- and after a build partition of more than 3072
rows has overflowed the small one.  Rows must equal the oracle's join (hash_build.c:9-35, hash_probe.c:9-46) after the
canonical sort (host/join/join_test.cc:27-38), whatever the table."""
import numpy as np
import pytest

import oracle
from conftest import sort_rows, split_batches
from test_wang_inverse import wang_inv_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dpu_olap_amd import Device
    d = Device(0)
    yield d
    d.close()


def _join(dev, fk, y, pk, x, lrows, rrows, bits=-1):
    import dpu_olap_amd as o
    dev.join_configure(bits, 0)
    try:
        lb = [(dev.to_device(a), dev.to_device(b)) for a, b in zip(split_batches(fk, lrows), split_batches(y, lrows))]
        rb = [(dev.to_device(a), dev.to_device(b)) for a, b in zip(split_batches(pk, rrows), split_batches(x, rrows))]
        j = o.JoinGpu(dev, lb, rb).Prepare()
        out = {k: dev.to_host(v) for k, v in j.Run().items()}
    finally:
        dev.join_configure(-1, 0)
    return sort_rows(out["fk"], out["y"], out["x"]), j.stats


@pytest.mark.parametrize("log2_rows", [16, 20, 22])
def test_small_and_large_table_give_the_oracles_rows(dev, log2_rows):
    rng = np.random.default_rng(log2_rows)
    n = 1 << log2_rows
    pk = rng.permutation(np.arange(7, 7 + 3 * n, 3, dtype=np.uint32))
    pk[:2] = [0, 0xFFFFFFFF]  # (0xFFFFFFFF is the table's empty marker: it has a cell of its own)
    x = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    fk = pk[rng.integers(0, n, size=n + 12345)]
    y = rng.integers(0, 1 << 32, size=fk.size, dtype=np.uint32)
    lrows, rrows = [fk.size // 3, 0, fk.size - fk.size // 3], [n // 2, n // 2]
    exp = sort_rows(*oracle.join(split_batches(fk, lrows), split_batches(y, lrows), split_batches(pk, rrows), split_batches(x, rrows),
                                 nr_partitions=64))
    for small in (1, 0):
        dev.set_tuning("probe_small_table", small)
        try:
            got, st = _join(dev, fk, y, pk, x, lrows, rrows)
        finally:
            dev.set_tuning("probe_small_table", 1)
        assert st["passes"] >= 1 and st["missing_keys"] == 0 and st["duplicate_keys"] == 0
        assert st["table_slots"] == (4096 if small and n // st["nr_partitions"] <= 2048 else 8192), st
        assert st["max_build_rows"] <= st["table_slots"] // 4 * 3
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)


def test_partition_above_the_small_tables_capacity_takes_the_large_table(dev):
    """4000 build keys share the leading 10 hash bits (one partition of a 10-bit plan), 2^18 others are spread: the average
    partition holds 260 rows, so the probe starts with the 4096-slot table, finds the partition of > 3072 rows and the call
    repeats the plan with the 8192-slot table (6144 rows)."""
    rng = np.random.default_rng(5)
    hot = wang_inv_np((np.uint32(5) << np.uint32(22)) | rng.choice(1 << 22, size=4000, replace=False).astype(np.uint32))
    rest = rng.choice(1 << 32, size=1 << 18, replace=False).astype(np.uint32)
    rest = rest[(oracle.wang_hash_np(rest) >> np.uint32(22)) != 5]
    pk = rng.permutation(np.concatenate([hot, rest]))
    x = rng.integers(0, 1 << 32, size=pk.size, dtype=np.uint32)
    fk = np.concatenate([pk[rng.integers(0, pk.size, size=1 << 19)], hot])
    y = np.arange(fk.size, dtype=np.uint32)
    lrows, rrows = [fk.size], [pk.size]
    exp = sort_rows(*oracle.join([fk], [y], [pk], [x], nr_partitions=16))
    got, st = _join(dev, fk, y, pk, x, lrows, rrows, bits=10)
    assert st["table_slots"] == 8192 and 4000 <= st["max_build_rows"] <= 6144 and st["missing_keys"] == 0, st
    assert dev.get_counter("join_attempts") == 2
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
    got, st = _join(dev, fk, y, pk, x, lrows, rrows)  # (the automatic plan: the next call starts with the small table again)
    assert st["missing_keys"] == 0
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
