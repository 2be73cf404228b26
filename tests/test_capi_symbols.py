"""CPU-only: the C-ABI library builds, loads and exports every symbol that
include/olap_gpu.h declares; host-side logic (generator mirror, argument
plumbing) behaves.  No compute calls -- there is no GPU here."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from dpu_olap_amd import build, capi
    build.build()
    return capi.load_library()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "olap_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(olap_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    from dpu_olap_amd import capi
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/olap_gpu.h but not exported"
    assert sorted(capi.EXPORTS) == syms
    assert lib.olap_abi_version() == 1


def test_arrow_front_end_symbols_and_structs(lib):
    """include/olap_arrow.h: every declared entry point is exported, and the ctypes mirror of the
    Arrow C Data Interface structs has the ABI's size (80 / 72 bytes on LP64)."""
    import ctypes as C
    from dpu_olap_amd import arrow_cdata
    hdr = open(os.path.join(ROOT, "include", "olap_arrow.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    syms = sorted(set(re.findall(r"\b(olap_arrow_[a-z0-9_]+)\s*\(", hdr)))
    assert syms == sorted(arrow_cdata.EXPORTS) and len(syms) == 5
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/olap_arrow.h but not exported"
    assert C.sizeof(arrow_cdata.ArrowArray) == 80 and C.sizeof(arrow_cdata.ArrowSchema) == 72
    # round trip of the struct mirror through pyarrow (no GPU involved)
    import pyarrow as pa
    a = pa.array([1, 2, 3], type=pa.uint32()).slice(1)
    ex = arrow_cdata._Exported([a])
    assert ex.structs[0].length == 2 and ex.structs[0].offset == 1 and ex.structs[0].n_buffers == 2
    ex.release()
    assert not ex.structs[0].release


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dpu_olap_amd
    with pytest.raises(dpu_olap_amd.OlapError):
        dpu_olap_amd.Device(0)


def test_product_path_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "dpu_olap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "olap_oracle" not in txt, f
    assert "oracle" not in open(os.path.join(ROOT, "include", "olap_gpu.h")).read()


def test_generator_mirror_distributions():
    from dpu_olap_amd import datagen as dg
    v = dg.uniform_u32(1 << 16, dg.SEED, dg.STREAM_V)
    sel = (v < (1 << 30)).mean()
    assert abs(sel - 0.25) < 0.01  # filter selectivity of uniform u32
    assert np.array_equal(dg.uniform_u32(100, 42, 0, first_row=50), dg.uniform_u32(150, 42, 0)[50:])
    assert np.array_equal(dg.index_u32(5, 7), np.array([7, 8, 9, 10, 11], dtype=np.uint32))
    L, R = 3000, 2048
    fk = dg.fk_u32(4 * L, 42, dg.STREAM_FK, 0, L, R)
    for i in range(4):  # generator.cc:46-57: batch i draws from [i*R, (i+1)*R - 1]
        b = fk[i * L:(i + 1) * L]
        assert b.min() >= i * R and b.max() <= (i + 1) * R - 1
    assert np.unique(fk).size > 0.7 * 4 * R * (1 - np.exp(-L / R))
    # digests are order independent
    a, b, c = (dg.uniform_u32(1000, 1, s) for s in (0, 1, 2))
    p = np.random.default_rng(0).permutation(1000)
    assert dg.checksum_rows3(a, b, c) == dg.checksum_rows3(a[p], b[p], c[p])
    assert dg.checksum(a) == dg.checksum(a[p])
    assert dg.checksum_rows3(a, b, c) != dg.checksum_rows3(a, c, b)


def test_operator_argument_plumbing_without_gpu():
    from dpu_olap_amd import operators as ops
    cols = ops._columns([np.arange(4, dtype=np.uint32), np.arange(3, dtype=np.uint32)])
    ptrs, rows, mem, dt = ops._table(cols)
    assert rows == [4, 3] and mem == 0 and dt == "uint32" and all(ptrs)
    with pytest.raises(TypeError):
        ops._table([np.zeros(2, np.uint32), np.zeros(2, np.int64)])
    with pytest.raises(TypeError):
        ops._dtype_code("float32")
    import pyarrow as pa
    rb = pa.record_batch({"fk": pa.array([1, 2], pa.uint32()), "y": pa.array([3, 4], pa.uint32())})
    assert ops._columns([rb], 1)[0].tolist() == [3, 4]


def test_partition_operator_keeps_converted_batches_alive():
    """PartitionGpu.Run hands the C ABI bare addresses; when a host batch had to be converted to uint32 (int32 /
    int64 input) the converted copy must be referenced until the call returns.  Checked on CPU with a stand-in
    Device whose partition_u32 reads the memory behind the pointer tables, after a garbage collection."""
    import ctypes
    import gc

    import numpy as np

    import oracle
    from dpu_olap_amd import capi, operators

    class FakeDevice:
        def set_profiling(self, on): pass
        def reset_timers(self): pass

        def partition_u32(self, kp, pp, rows, mem, P, out_k, out_p, n, out_mem):
            gc.collect()
            junk = [np.full(70001, 0xDEADBEEF, dtype=np.uint32) for _ in range(8)]   # reuse freed blocks, if any
            ks = np.concatenate([np.ctypeslib.as_array(ctypes.cast(int(kp[b]), ctypes.POINTER(ctypes.c_uint32)), (int(rows[b]),)).copy()
                                 for b in range(len(rows))])
            ps = np.concatenate([np.ctypeslib.as_array(ctypes.cast(int(pp[b]), ctypes.POINTER(ctypes.c_uint32)), (int(rows[b]),)).copy()
                                 for b in range(len(rows))])
            del junk
            ok, orow, hist = oracle.partition(ks, P)
            ctypes.memmove(out_k, ok.ctypes.data, ok.nbytes)
            op = ps[orow]
            ctypes.memmove(out_p, op.ctypes.data, op.nbytes)
            return [0] + [int(h) for h in hist]

    rng = np.random.default_rng(5)
    keys = [rng.integers(0, 1 << 31, size=s).astype(np.int64) for s in (70001, 70001, 5)]
    pay = [rng.integers(0, 1 << 31, size=k.size).astype(np.int32) for k in keys]
    parts = operators.PartitionGpu(FakeDevice(), list(zip(keys, pay)), 8).Prepare().Run()
    allk, allp = np.concatenate(keys).astype(np.uint32), np.concatenate(pay).astype(np.uint32)
    gk, gp = np.concatenate([k for k, _ in parts]), np.concatenate([v for _, v in parts])
    order_g, order_e = np.lexsort((gp, gk)), np.lexsort((allp, allk))
    np.testing.assert_array_equal(gk[order_g], allk[order_e])
    np.testing.assert_array_equal(gp[order_g], allp[order_e])
