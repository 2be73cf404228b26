"""GPU parity of the filter's LDS-staged store path (csrc/scan_ops.hip k_filter_compact_lds; kernel_filter,
dpu/shared/kernels/filter.c:25,57-177: out = v[v < threshold], order preserved, per batch) against the oracle, bit-exact:
all four integer types, ragged / empty / long batches, unaligned input slices, every selectivity from none to all."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
TYPES = {"u32": np.uint32, "i32": np.int32, "u64": np.uint64, "i64": np.int64}


@pytest.fixture(scope="module")
def dev():
    from dpu_olap_amd import Device
    d = Device(0)
    d.set_tuning("filter_stores", 2)
    yield d
    d.set_tuning("filter_stores", 1)
    d.close()


def _check(dev, batches, thr, space):
    from dpu_olap_amd import FilterGpu
    ins = batches if space == "host" else [dev.to_device(b) for b in batches]
    f = FilterGpu(dev, ins, threshold=thr).Prepare()
    res = f.GetResult()
    if space == "device":
        res = [dev.to_host(r) for r in res]
    for b, r in zip(batches, res):
        np.testing.assert_array_equal(r, b[b < b.dtype.type(thr)])
    assert f.Run() == sum(int((b < b.dtype.type(thr)).sum()) for b in batches)


@pytest.mark.parametrize("t", list(TYPES))
@pytest.mark.parametrize("space", ["host", "device"])
def test_filter_lds_ragged_batches_and_selectivities(dev, t, space):
    dt = TYPES[t]
    rng = np.random.default_rng(5 + len(t))
    info = np.iinfo(dt)
    sizes = [65536, 1, 0, 4095, 4096, 4097, 65537, 300000, 7, 131072, 65536 * 3 + 5]
    batches = [rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True) for n in sizes]
    span = int(info.max) - int(info.min)
    for q in (0.0, 0.003, 0.25, 0.5, 0.97, 1.0):
        thr = min(int(info.min) + int(span * q), int(info.max))
        _check(dev, batches, thr, space)
    _check(dev, batches, int(info.max), space)   # everything but the maximum itself passes


@pytest.mark.parametrize("t", ["u32", "i64"])
def test_filter_lds_unaligned_slices_and_long_batches(dev, t):
    """Input batches that start 1..3 elements into their buffers (Arrow slices) and a few long batches (count + scan + compact over
    segments: a segment's output starts inside a 16-byte vector of its predecessor's)."""
    dt = TYPES[t]
    rng = np.random.default_rng(11)
    base = [rng.integers(0, 1 << 31, size=n + 3, dtype=np.int64).astype(dt) for n in (70000, 65536, 12345, 200001)]
    sl = [b[k + 1:] for k, b in enumerate(base[:3])] + [base[3][3:]]
    _check(dev, sl, 1 << 29, "host")
    _check(dev, sl, 1 << 29, "device")
    long_batches = [rng.integers(0, 1 << 31, size=n, dtype=np.int64).astype(dt) for n in (3_000_001, 2_500_000, 999_999)]
    _check(dev, long_batches, 1 << 29, "device")
    _check(dev, long_batches, 1 << 20, "device")


def test_filter_lds_reference_shape_vs_oracle(dev):
    """BM_Filter's shape at a reduced scale factor: 1024 batches x 64 Ki uniform u32, threshold 2^30 (filter_benchmark.cc:76,153)."""
    from dpu_olap_amd import FilterGpu
    from dpu_olap_amd import datagen as dg
    nb, rows = 1024, 65536
    v = [dg.uniform_u32(rows, dg.SEED, dg.STREAM_V, first_row=b * rows) for b in range(nb)]
    res = FilterGpu(dev, [dev.to_device(b) for b in v]).Prepare().GetResult()
    for b in (0, 1, 511, 1023):
        np.testing.assert_array_equal(dev.to_host(res[b]), oracle.filter_lt(v[b], 1 << 30))
    assert abs(sum(r.rows for r in res) / (nb * rows) - 0.25) < 0.001
