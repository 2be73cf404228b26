"""ctypes wrapper around oracle/libolap_oracle.so (the CPU restatement of the
reference's kernels).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by dpu_olap_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libolap_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "olap_oracle.c")
    hdr = os.path.join(_HERE, "olap_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libolap_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.oracle_wang_hash_u32.restype = C.c_uint32
        L.oracle_wang_hash_u32.argtypes = [C.c_uint32]
        L.oracle_bucket_of.restype = C.c_uint32
        L.oracle_bucket_of.argtypes = [C.c_uint32, C.c_uint32]
        for name, ct in (("u32", C.c_uint32), ("i32", C.c_int32), ("u64", C.c_uint64), ("i64", C.c_int64)):
            f = getattr(L, f"oracle_filter_lt_{name}")
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_uint64, ct, C.c_void_p]
            s = getattr(L, f"oracle_sum_{name}")
            s.restype = C.c_uint64 if name[0] == "u" else C.c_int64
            s.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_take_u32.restype = None
        L.oracle_take_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_take_u64.restype = None
        L.oracle_take_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_partition_u32.restype = C.c_int
        L.oracle_partition_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_join_u32.restype = C.c_uint64
        L.oracle_join_u32.argtypes = [
            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u32p, C.c_uint32,
            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u32p, C.c_uint32,
            C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        L.oracle_ht_hash_u64.restype = C.c_uint32
        L.oracle_ht_hash_u64.argtypes = [C.c_uint64]
        L.oracle_join_u64.restype = C.c_uint64
        L.oracle_join_u64.argtypes = L.oracle_join_u32.argtypes
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


_DT = {"u32": np.uint32, "i32": np.int32, "u64": np.uint64, "i64": np.int64}


def _kind(a: np.ndarray) -> str:
    for k, v in _DT.items():
        if a.dtype == v:
            return k
    raise TypeError(f"unsupported dtype {a.dtype}")


def _c(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a)


def wang_hash(key: int) -> int:
    return int(lib().oracle_wang_hash_u32(int(key) & 0xFFFFFFFF))


def wang_hash_np(keys: np.ndarray) -> np.ndarray:
    """Vectorised numpy restatement of partition.c:20-28 (used to cross-check the C one)."""
    k = keys.astype(np.uint32).copy()
    with np.errstate(over="ignore"):
        k += ~(k << np.uint32(15))
        k ^= k >> np.uint32(10)
        k += k << np.uint32(3)
        k ^= k >> np.uint32(6)
        k += ~(k << np.uint32(11))
        k ^= k >> np.uint32(16)
    return k


def bucket_of(key: int, nr_partitions: int) -> int:
    return int(lib().oracle_bucket_of(int(key) & 0xFFFFFFFF, nr_partitions))


def filter_lt(values: np.ndarray, threshold) -> np.ndarray:
    v = _c(values)
    k = _kind(v)
    out = np.empty_like(v)
    n = getattr(lib(), f"oracle_filter_lt_{k}")(v.ctypes.data, v.size, threshold, out.ctypes.data)
    return out[: int(n)].copy()


def take(data: np.ndarray, idx: np.ndarray) -> np.ndarray:
    d, i = _c(data), _c(idx.astype(np.uint32, copy=False))
    out = np.empty(i.size, dtype=d.dtype)
    if d.dtype.itemsize == 4:
        lib().oracle_take_u32(d.ctypes.data, i.ctypes.data, i.size, out.ctypes.data)
    elif d.dtype.itemsize == 8:
        lib().oracle_take_u64(d.ctypes.data, i.ctypes.data, i.size, out.ctypes.data)
    else:
        raise TypeError(d.dtype)
    return out


def sum_(values: np.ndarray) -> int:
    v = _c(values)
    return int(getattr(lib(), f"oracle_sum_{_kind(v)}")(v.ctypes.data, v.size))


def partition(keys: np.ndarray, nr_partitions: int, o_shift: int = 0):
    """-> (out_keys, out_rowid, hist_end) as kernel_partition leaves them."""
    k = _c(keys.astype(np.uint32, copy=False))
    ok = np.empty_like(k)
    orow = np.empty_like(k)
    hist = np.empty(nr_partitions, dtype=np.uint32)
    rc = lib().oracle_partition_u32(k.ctypes.data, k.size, nr_partitions, o_shift, ok.ctypes.data, orow.ctypes.data, hist.ctypes.data)
    if rc != 0:
        raise ValueError("nr_partitions must be a power of two >= 2 and o_shift < nr_partitions")
    return ok, orow, hist


def _ptr_array(cols):
    arr = (C.c_void_p * len(cols))()
    for i, c in enumerate(cols):
        arr[i] = c.ctypes.data
    return arr


def join(left_fk, left_y, right_pk, right_x, nr_partitions: int = 2, threads: int = 0):
    """Batched PK-FK join; each argument is a list of uint32 numpy arrays (one per
    batch).  Returns (fk, y, x) in the oracle's partition-major order."""
    lf = [_c(a.astype(np.uint32, copy=False)) for a in left_fk]
    ly = [_c(a.astype(np.uint32, copy=False)) for a in left_y]
    rp = [_c(a.astype(np.uint32, copy=False)) for a in right_pk]
    rx = [_c(a.astype(np.uint32, copy=False)) for a in right_x]
    lrows = np.array([a.size for a in lf], dtype=np.uint32)
    rrows = np.array([a.size for a in rp], dtype=np.uint32)
    total = int(lrows.sum())
    ofk = np.empty(max(total, 1), dtype=np.uint32)
    oy = np.empty(max(total, 1), dtype=np.uint32)
    ox = np.empty(max(total, 1), dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    n = lib().oracle_join_u32(
        _ptr_array(lf), _ptr_array(ly), lrows.ctypes.data_as(u32p), len(lf),
        _ptr_array(rp), _ptr_array(rx), rrows.ctypes.data_as(u32p), len(rp),
        nr_partitions, threads, ofk.ctypes.data, oy.ctypes.data, ox.ctypes.data,
    )
    if n == 0xFFFFFFFFFFFFFFFF:
        raise KeyError("oracle join: a foreign key has no matching primary key (reference asserts, hash_probe.c:32-33)")
    n = int(n)
    return ofk[:n], oy[:n], ox[:n]


def ht_hash_u64(key: int) -> int:
    """hashtable.c:23-26 (HT_64BIT_KEYS)."""
    return int(lib().oracle_ht_hash_u64(key & 0xFFFFFFFFFFFFFFFF))


def join_u64(left_fk, left_y, right_pk, right_x, nr_partitions: int = 2, threads: int = 0):
    """PK-FK join on 64-bit keys (uint64 / int64 bit patterns) with uint32 payloads; lists of numpy arrays, one per
    batch.  Returns (fk: uint64, y, x) in the oracle's partition-major order."""
    lf = [_c(a.view(np.uint64) if a.dtype == np.int64 else a.astype(np.uint64, copy=False)) for a in left_fk]
    rp = [_c(a.view(np.uint64) if a.dtype == np.int64 else a.astype(np.uint64, copy=False)) for a in right_pk]
    ly = [_c(a.astype(np.uint32, copy=False)) for a in left_y]
    rx = [_c(a.astype(np.uint32, copy=False)) for a in right_x]
    lrows = np.array([a.size for a in lf], dtype=np.uint32)
    rrows = np.array([a.size for a in rp], dtype=np.uint32)
    total = int(lrows.sum())
    ofk = np.empty(max(total, 1), dtype=np.uint64)
    oy = np.empty(max(total, 1), dtype=np.uint32)
    ox = np.empty(max(total, 1), dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    n = lib().oracle_join_u64(_ptr_array(lf), _ptr_array(ly), lrows.ctypes.data_as(u32p), len(lf), _ptr_array(rp), _ptr_array(rx),
                              rrows.ctypes.data_as(u32p), len(rp), nr_partitions, threads, ofk.ctypes.data, oy.ctypes.data, ox.ctypes.data)
    if n == 0xFFFFFFFFFFFFFFFF:
        raise KeyError("oracle join: a foreign key has no matching primary key (reference asserts, hash_probe.c:32-33)")
    n = int(n)
    return ofk[:n], oy[:n], ox[:n]


def max_threads() -> int:
    return int(lib().oracle_max_threads())
