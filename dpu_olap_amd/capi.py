"""ctypes binding of include/olap_gpu.h (libolap_gpu.so).

This is the only way Python reaches the hot path: there is NO CPU fallback.  If
the HIP library is missing or no GPU is visible, loading / creating a context
raises -- tests that need the GPU are marked `gpu`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libolap_gpu.so")

OLAP_OK = 0
ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_NO_MATCH, ERR_OVERFLOW, ERR_CAPACITY = -1, -2, -3, -4, -5, -6
U32, I32, U64, I64 = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1

DTYPE_BYTES = {U32: 4, I32: 4, U64: 8, I64: 8}
DTYPE_FROM_NUMPY = {"uint32": U32, "int32": I32, "uint64": U64, "int64": I64}
NUMPY_FROM_DTYPE = {U32: "uint32", I32: "int32", U64: "uint64", I64: "int64"}

# every symbol include/olap_gpu.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "olap_abi_version", "olap_create", "olap_destroy", "olap_last_error", "olap_device_info",
    "olap_set_stream", "olap_get_stream", "olap_synchronize", "olap_reserve_workspace",
    "olap_workspace_bytes", "olap_malloc", "olap_free", "olap_memcpy_h2d", "olap_memcpy_d2h",
    "olap_memset", "olap_set_profiling", "olap_reset_timers", "olap_get_timers", "olap_filter_lt",
    "olap_sum", "olap_take", "olap_partition_u32", "olap_partition_tuples_u32", "olap_split_tuples_u32", "olap_join_u32", "olap_join_u64", "olap_join_tuples_u32", "olap_join_presplit_u32", "olap_join_set_right_ready_event", "olap_join_set_unmatched", "olap_join_configure", "olap_join_set_hash_skip_bits", "olap_set_tuning", "olap_get_counter", "olap_phase_kernel",
    "olap_gen_uniform_u32", "olap_gen_uniform_u64", "olap_gen_index_u32", "olap_gen_fk_u32",
    "olap_gen_uniform_u32_at", "olap_checksum", "olap_checksum_rows3_u32",
    "olap_group_create", "olap_group_destroy", "olap_group_size", "olap_group_ctx", "olap_group_device", "olap_group_last_error",
    "olap_group_set_transport", "olap_group_last_transport", "olap_group_last_phase_ms", "olap_group_join_u32",
]


class OlapError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"olap error {code}: {message}")
        self.code = code
        self.message = message


class JoinStats(C.Structure):
    _fields_ = [
        ("passes", C.c_uint32),
        ("radix_bits", C.c_uint32 * 4),
        ("nr_partitions", C.c_uint32),
        ("table_slots", C.c_uint32),
        ("left_rows", C.c_uint64),
        ("right_rows", C.c_uint64),
        ("max_build_rows", C.c_uint64),
        ("duplicate_keys", C.c_uint64),
        ("missing_keys", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {
            "passes": self.passes,
            "radix_bits": list(self.radix_bits)[: self.passes],
            "nr_partitions": self.nr_partitions,
            "table_slots": self.table_slots,
            "left_rows": self.left_rows,
            "right_rows": self.right_rows,
            "max_build_rows": self.max_build_rows,
            "duplicate_keys": self.duplicate_keys,
            "missing_keys": self.missing_keys,
        }


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (SONAME libamdhip64.so.7, found through torch/lib's RPATH under the name
    "libamdhip64.so"); libolap_gpu.so asks for "libamdhip64.so.7".  If our library came
    first the loader would later add torch's copy as a SECOND runtime, which then sees no
    GPU ("No HIP GPUs are available").  Pre-loading torch's copy makes both resolve to the
    same object whatever the import order.  Without torch installed nothing happens and the
    system ROCm runtime is used (as the C++ host binaries do)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to the loader's default resolution


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libolap_gpu.so and declare the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("OLAP_GPU_LIB") or LIB_PATH  # OLAP_GPU_LIB: A/B builds in one run
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build the HIP extension first (python -m dpu_olap_amd.build). "
            "There is no CPU fallback for the operator hot path."
        )
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    vp, u64, u32, i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    pvp, pu64 = C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)
    sig = {
        "olap_abi_version": (i, []),
        "olap_create": (i, [i, pvp]),
        "olap_destroy": (None, [vp]),
        "olap_last_error": (C.c_char_p, [vp]),
        "olap_device_info": (i, [vp, C.c_char_p, C.c_size_t, C.POINTER(i), pu64, pu64]),
        "olap_set_stream": (i, [vp, vp]),
        "olap_get_stream": (vp, [vp]),
        "olap_synchronize": (i, [vp]),
        "olap_reserve_workspace": (i, [vp, u64]),
        "olap_workspace_bytes": (u64, [vp]),
        "olap_malloc": (i, [vp, u64, pvp]),
        "olap_free": (i, [vp, vp]),
        "olap_memcpy_h2d": (i, [vp, vp, vp, u64]),
        "olap_memcpy_d2h": (i, [vp, vp, vp, u64]),
        "olap_memset": (i, [vp, vp, i, u64]),
        "olap_set_profiling": (i, [vp, i]),
        "olap_reset_timers": (i, [vp]),
        "olap_get_timers": (i, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), pu64, i, C.POINTER(i)]),
        "olap_filter_lt": (i, [vp, i, pvp, pu64, u64, i, vp, pvp, i, pu64, pu64]),
        "olap_sum": (i, [vp, i, pvp, pu64, u64, i, pu64]),
        "olap_take": (i, [vp, i, pvp, pu64, pvp, pu64, u64, i, pvp, i]),
        "olap_partition_u32": (i, [vp, pvp, pvp, pu64, u64, i, u32, vp, vp, u64, i, pu64]),
        "olap_join_u32": (i, [vp, pvp, pvp, pu64, u64, pvp, pvp, pu64, u64, i, vp, vp, vp, u64, i, pu64,
                              C.POINTER(JoinStats)]),
        "olap_partition_tuples_u32": (i, [vp, pvp, pvp, pu64, u64, i, u32, vp, u64, pu64]),
        "olap_split_tuples_u32": (i, [vp, pvp, pvp, pu64, u64, i, u32, u32, u32, vp, u64, pu64]),
        "olap_join_tuples_u32": (i, [vp, vp, u64, vp, u64, vp, vp, vp, u64, pu64, C.POINTER(JoinStats)]),
        "olap_join_u64": (i, [vp, pvp, pvp, pu64, u64, pvp, pvp, pu64, u64, i, vp, vp, vp, u64, i, pu64, C.POINTER(JoinStats)]),
        "olap_join_presplit_u32": (i, [vp, vp, u64, pu64, pu64, vp, u64, pu64, pu64, u32, u32, vp, vp, vp, u64, pu64,
                                       C.POINTER(JoinStats)]),
        "olap_join_set_right_ready_event": (i, [vp, vp]),
        "olap_join_set_unmatched": (i, [vp, i]),
        "olap_join_configure": (i, [vp, i, i]),
        "olap_join_set_hash_skip_bits": (i, [vp, i]),
        "olap_set_tuning": (i, [vp, C.c_char_p, i]),
        "olap_get_counter": (i, [vp, C.c_char_p, C.POINTER(C.c_uint64)]),
        "olap_phase_kernel": (C.c_char_p, [vp, C.c_char_p]),
        "olap_group_create": (i, [i, C.POINTER(C.c_int), C.POINTER(vp)]),
        "olap_group_destroy": (None, [vp]),
        "olap_group_size": (i, [vp]),
        "olap_group_ctx": (vp, [vp, i]),
        "olap_group_device": (i, [vp, i]),
        "olap_group_last_error": (C.c_char_p, [vp]),
        "olap_group_set_transport": (i, [vp, i]),
        "olap_group_last_transport": (i, [vp]),
        "olap_group_last_phase_ms": (i, [vp, C.POINTER(C.c_double)]),
        "olap_group_join_u32": (i, [vp, pvp, pvp, pu64, u64, pvp, pvp, pu64, u64, i, vp, vp, vp, u64, pu64, pu64, C.POINTER(JoinStats)]),
        "olap_gen_uniform_u32": (i, [vp, vp, u64, u64, u32, u64]),
        "olap_gen_uniform_u64": (i, [vp, vp, u64, u64, u32, u64]),
        "olap_gen_index_u32": (i, [vp, vp, u64, u32]),
        "olap_gen_fk_u32": (i, [vp, vp, u64, u64, u32, u64, u64, u64]),
        "olap_gen_uniform_u32_at": (i, [vp, vp, vp, u64, u64, u32]),
        "olap_checksum": (i, [vp, vp, u64, i, pu64]),
        "olap_checksum_rows3_u32": (i, [vp, vp, vp, vp, u64, pu64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = L
    return L


def _ptr_table(ptrs: Sequence[int]):
    if isinstance(ptrs, C.Array):  # prebuilt table (hot loops over many batches)
        return ptrs
    if not len(ptrs):
        return (C.c_void_p * 1)()
    import numpy as np
    a = np.array([p or 0 for p in ptrs], dtype=np.uint64)      # thousands of batches per call: filled in C
    return (C.c_void_p * len(ptrs)).from_buffer(a)             # (the ctypes array keeps the numpy buffer alive)


def ptr_table(ptrs: Sequence[int]):
    """Prebuilt host pointer table (ctypes array) for repeated calls."""
    return _ptr_table(list(ptrs))


def u64_table(vals: Iterable[int]):
    return _u64_table(list(vals))


def _u64_table(vals: Iterable[int]):
    if isinstance(vals, C.Array):
        return vals
    import numpy as np
    a = np.array(list(vals), dtype=np.uint64)
    if not a.size:
        return (C.c_uint64 * 1)()
    return (C.c_uint64 * a.size).from_buffer(a)


class Context:
    """One olap_ctx == one GPU.  Plays the role of dpu::DpuSet in the reference
    (host/dpuext/dpuext.hpp:709-715)."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.olap_create(device_id, C.byref(h))
        if rc != OLAP_OK or not h:
            raise OlapError(rc, f"olap_create(device {device_id}) failed: no usable HIP device")
        self.h = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "h", None):
            self.lib.olap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- helpers
    def _check(self, rc: int):
        if rc != OLAP_OK:
            raise OlapError(rc, (self.lib.olap_last_error(self.h) or b"").decode())

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, tot, fr = C.c_int(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.olap_device_info(self.h, name, 256, C.byref(cus), C.byref(tot), C.byref(fr)))
        return {"name": name.value.decode(), "compute_units": cus.value, "total_mem": tot.value, "free_mem": fr.value}

    def set_stream(self, hip_stream: int | None):
        self._check(self.lib.olap_set_stream(self.h, hip_stream))

    def get_stream(self) -> int:
        return self.lib.olap_get_stream(self.h) or 0

    def synchronize(self):
        self._check(self.lib.olap_synchronize(self.h))

    def reserve_workspace(self, nbytes: int):
        self._check(self.lib.olap_reserve_workspace(self.h, nbytes))

    def workspace_bytes(self) -> int:
        return self.lib.olap_workspace_bytes(self.h)

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.olap_malloc(self.h, nbytes, C.byref(p)))
        return p.value or 0

    def free(self, dptr: int):
        self._check(self.lib.olap_free(self.h, dptr))

    def h2d(self, dst: int, src_host_ptr: int, nbytes: int):
        self._check(self.lib.olap_memcpy_h2d(self.h, dst, src_host_ptr, nbytes))

    def d2h(self, dst_host_ptr: int, src: int, nbytes: int):
        self._check(self.lib.olap_memcpy_d2h(self.h, dst_host_ptr, src, nbytes))

    def memset(self, dst: int, value: int, nbytes: int):
        self._check(self.lib.olap_memset(self.h, dst, value, nbytes))

    def set_profiling(self, on: bool):
        self._check(self.lib.olap_set_profiling(self.h, 1 if on else 0))

    def reset_timers(self):
        self._check(self.lib.olap_reset_timers(self.h))

    def timers(self) -> dict:
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_uint64 * cap)()
        n = C.c_int()
        self._check(self.lib.olap_get_timers(self.h, names, ms, launches, cap, C.byref(n)))
        return {names[k].decode(): {"ms": ms[k], "launches": launches[k]} for k in range(min(n.value, cap))}

    # -- operators (raw pointer level)
    def filter_lt(self, dtype: int, ptrs, rows, in_mem: int, threshold, out_ptrs, out_mem: int):
        nb = len(ptrs)
        ct = {U32: C.c_uint32, I32: C.c_int32, U64: C.c_uint64, I64: C.c_int64}.get(dtype, C.c_uint64)  # unknown codes are the ABI's to refuse
        thr = ct(threshold)
        out_rows = (C.c_uint64 * max(nb, 1))()
        total = C.c_uint64()
        self._check(self.lib.olap_filter_lt(self.h, dtype, _ptr_table(ptrs), _u64_table(rows), nb, in_mem,
                                            C.cast(C.byref(thr), C.c_void_p), _ptr_table(out_ptrs), out_mem,
                                            out_rows, C.byref(total)))
        # per-batch result lengths as a numpy view of the ctypes array (a Python list of 262144 entries would
        # cost more than the kernel at the reference's BM_Filter shape)
        import numpy as np
        return np.frombuffer(out_rows, dtype=np.uint64, count=nb).copy() if nb else np.empty(0, np.uint64), total.value

    def sum(self, dtype: int, ptrs, rows, in_mem: int) -> int:
        out = C.c_uint64()
        self._check(self.lib.olap_sum(self.h, dtype, _ptr_table(ptrs), _u64_table(rows), len(ptrs), in_mem, C.byref(out)))
        v = out.value
        if dtype in (I32, I64) and v >= 1 << 63:
            v -= 1 << 64
        return v

    def take(self, dtype: int, data_ptrs, data_rows, idx_ptrs, idx_rows, in_mem: int, out_ptrs, out_mem: int):
        self._check(self.lib.olap_take(self.h, dtype, _ptr_table(data_ptrs), _u64_table(data_rows), _ptr_table(idx_ptrs),
                                       _u64_table(idx_rows), len(data_ptrs), in_mem, _ptr_table(out_ptrs), out_mem))

    def partition_u32(self, key_ptrs, payload_ptrs, rows, in_mem: int, nr_partitions: int, out_keys: int,
                      out_payload: int, out_capacity: int, out_mem: int):
        offs = (C.c_uint64 * (nr_partitions + 1))()
        pay = _ptr_table(payload_ptrs) if payload_ptrs is not None else None
        self._check(self.lib.olap_partition_u32(self.h, _ptr_table(key_ptrs), pay, _u64_table(rows), len(key_ptrs), in_mem,
                                                nr_partitions, out_keys, out_payload, out_capacity, out_mem, offs))
        return [offs[k] for k in range(nr_partitions + 1)]

    def partition_tuples_u32(self, key_ptrs, payload_ptrs, rows, in_mem: int, nr_partitions: int, out_tuples: int,
                             out_capacity: int):
        offs = (C.c_uint64 * (nr_partitions + 1))()
        pay = _ptr_table(payload_ptrs) if payload_ptrs is not None else None
        self._check(self.lib.olap_partition_tuples_u32(self.h, _ptr_table(key_ptrs), pay, _u64_table(rows), len(key_ptrs),
                                                       in_mem, nr_partitions, out_tuples, out_capacity, offs))
        return [offs[k] for k in range(nr_partitions + 1)]

    def split_tuples_u32(self, key_ptrs, payload_ptrs, rows, in_mem: int, owner_bits: int, group_bits: int, low_bits: int,
                         out_tuples: int, out_capacity: int):
        """The multi-device split: buckets [owner | group | low] of the hash, laid out [group | owner | low]."""
        nb = 1 << (owner_bits + group_bits + low_bits)
        offs = (C.c_uint64 * (nb + 1))()
        pay = _ptr_table(payload_ptrs) if payload_ptrs is not None else None
        self._check(self.lib.olap_split_tuples_u32(self.h, _ptr_table(key_ptrs), pay, _u64_table(rows), len(key_ptrs), in_mem,
                                                   owner_bits, group_bits, low_bits, out_tuples, out_capacity, offs))
        return [offs[k] for k in range(nb + 1)]

    def join_tuples_u32(self, left_tuples: int, n_left: int, right_tuples: int, n_right: int, out_fk: int, out_y: int,
                        out_x: int, out_capacity: int):
        n = C.c_uint64()
        st = JoinStats()
        self._check(self.lib.olap_join_tuples_u32(self.h, left_tuples, n_left, right_tuples, n_right, out_fk, out_y,
                                                  out_x, out_capacity, C.byref(n), C.byref(st)))
        return n.value, st.as_dict()

    def join_presplit_u32(self, left_tuples: int, n_left: int, left_lo, left_hi, right_tuples: int, n_right: int,
                          right_lo, right_hi, n_segments: int, chunks_per_segment: int, out_fk: int, out_y: int,
                          out_x: int, out_capacity: int):
        """Join of tuple arrays pre-split into n_segments x chunks_per_segment chunks (segment-major
        host tables of tuple indices); see include/olap_gpu.h."""
        n = C.c_uint64()
        st = JoinStats()
        self._check(self.lib.olap_join_presplit_u32(self.h, left_tuples, n_left, _u64_table(left_lo), _u64_table(left_hi),
                                                    right_tuples, n_right, _u64_table(right_lo), _u64_table(right_hi),
                                                    n_segments, chunks_per_segment, out_fk, out_y, out_x, out_capacity,
                                                    C.byref(n), C.byref(st)))
        return n.value, st.as_dict()

    def join_configure(self, radix_bits_total: int = -1, max_bits_per_pass: int = 0):
        self._check(self.lib.olap_join_configure(self.h, radix_bits_total, max_bits_per_pass))

    def set_tuning(self, key: str, value: int):
        self._check(self.lib.olap_set_tuning(self.h, key.encode(), value))

    def phase_kernel(self, phase: str) -> str:
        """rocprofv3 name of the kernel last launched under a timer phase ("" if none)."""
        r = self.lib.olap_phase_kernel(self.h, phase.encode())
        return r.decode() if r else ""

    def get_counter(self, name: str) -> int:
        """Diagnostics of the last join: "join_attempts", "join_region_overflows"."""
        v = C.c_uint64()
        self._check(self.lib.olap_get_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def join_set_unmatched(self, drop: bool):
        """False (default): a foreign key without a match is an error, like the reference; True: inner join, the row is dropped."""
        self._check(self.lib.olap_join_set_unmatched(self.h, 1 if drop else 0))

    def join_set_right_ready_event(self, hip_event: int | None):
        """One-shot: the next join waits for this recorded hipEvent_t before reading its right table."""
        self._check(self.lib.olap_join_set_right_ready_event(self.h, hip_event))

    def join_set_hash_skip_bits(self, bits: int):
        self._check(self.lib.olap_join_set_hash_skip_bits(self.h, bits))

    def join_u32(self, lfk, ly, lrows, rpk, rx, rrows, in_mem: int, out_fk: int, out_y: int, out_x: int,
                 out_capacity: int, out_mem: int):
        n = C.c_uint64()
        st = JoinStats()
        lyt = _ptr_table(ly) if ly is not None else None   # None: out_y / out_x are row indices
        rxt = _ptr_table(rx) if rx is not None else None
        self._check(self.lib.olap_join_u32(self.h, _ptr_table(lfk), lyt, _u64_table(lrows), len(lfk),
                                           _ptr_table(rpk), rxt, _u64_table(rrows), len(rpk), in_mem,
                                           out_fk, out_y, out_x, out_capacity, out_mem, C.byref(n), C.byref(st)))
        return n.value, st.as_dict()

    def join_u64(self, lfk: Sequence[int], ly, lrows, rpk: Sequence[int], rx, rrows, in_mem: int, out_fk: int, out_y: int,
                 out_x: int, out_capacity: int, out_mem: int):
        """64-bit keys, 32-bit payloads (ly / rx None: selection vectors).  -> (rows, stats)"""
        n = C.c_uint64()
        st = JoinStats()
        lyt = _ptr_table(ly) if ly is not None else None
        rxt = _ptr_table(rx) if rx is not None else None
        self._check(self.lib.olap_join_u64(self.h, _ptr_table(lfk), lyt, _u64_table(lrows), len(lfk), _ptr_table(rpk), rxt,
                                           _u64_table(rrows), len(rpk), in_mem, out_fk, out_y, out_x, out_capacity, out_mem,
                                           C.byref(n), C.byref(st)))
        return n.value, st.as_dict()

    # -- generator / digests
    def gen_uniform_u32(self, out: int, n: int, seed: int, stream_id: int, first_row: int = 0):
        self._check(self.lib.olap_gen_uniform_u32(self.h, out, n, seed, stream_id, first_row))

    def gen_uniform_u64(self, out: int, n: int, seed: int, stream_id: int, first_row: int = 0):
        self._check(self.lib.olap_gen_uniform_u64(self.h, out, n, seed, stream_id, first_row))

    def gen_index_u32(self, out: int, n: int, first_value: int = 0):
        self._check(self.lib.olap_gen_index_u32(self.h, out, n, first_value))

    def gen_fk_u32(self, out: int, n: int, seed: int, stream_id: int, first_row: int, left_batch_rows: int,
                   right_batch_rows: int):
        self._check(self.lib.olap_gen_fk_u32(self.h, out, n, seed, stream_id, first_row, left_batch_rows, right_batch_rows))

    def gen_uniform_u32_at(self, out: int, rows_dev: int, n: int, seed: int, stream_id: int):
        self._check(self.lib.olap_gen_uniform_u32_at(self.h, out, rows_dev, n, seed, stream_id))

    def checksum(self, col: int, n: int, elem_bytes: int):
        out = (C.c_uint64 * 3)()
        self._check(self.lib.olap_checksum(self.h, col, n, elem_bytes, out))
        return out[0], out[1], out[2]

    def checksum_rows3_u32(self, a: int, b: int, c: int, n: int):
        out = (C.c_uint64 * 2)()
        self._check(self.lib.olap_checksum_rows3_u32(self.h, a, b, c, n, out))
        return out[0], out[1]


class _BorrowedContext(Context):
    """The context of one rank of a Group: owned (and destroyed) by the group."""

    def __init__(self, lib, h, device_id):
        self.lib, self.h, self.device_id = lib, C.c_void_p(h), device_id

    def close(self):
        self.h = None


TRANSPORT_AUTO, TRANSPORT_COPY, TRANSPORT_RCCL = 0, 1, 2


class Group:
    """olap_group: several devices (or several logical ranks on one device: loopback) behind one handle;
    the join spans all of them like JoinDpu spans the DpuSet (host/join/join_dpu.cc:82-142,168-400)."""

    def __init__(self, device_ids: Sequence[int]):
        self.lib = load_library()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.lib.olap_group_create(len(device_ids), ids, C.byref(h))
        if rc != OLAP_OK or not h:
            raise OlapError(rc, f"olap_group_create({list(device_ids)}) failed")
        self.h = h
        self.size = self.lib.olap_group_size(self.h)
        self.ranks = [_BorrowedContext(self.lib, self.lib.olap_group_ctx(self.h, r), device_ids[r]) for r in range(self.size)]

    def close(self):
        if getattr(self, "h", None):
            for r in self.ranks:
                r.close()
            self.lib.olap_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != OLAP_OK:
            raise OlapError(rc, (self.lib.olap_group_last_error(self.h) or b"").decode())

    def set_transport(self, transport: int):
        self._check(self.lib.olap_group_set_transport(self.h, transport))

    def last_transport(self) -> int:
        return self.lib.olap_group_last_transport(self.h)

    def last_phase_ms(self) -> dict:
        ms = (C.c_double * 4)()
        self._check(self.lib.olap_group_last_phase_ms(self.h, ms))
        return dict(zip(("split", "exchange", "join", "copy-out"), ms))

    def join_u32(self, lfk, ly, lrows, rpk, rx, rrows, in_mem: int, out_fk: int, out_y: int, out_x: int, out_capacity: int):
        """-> (rows, rows per rank, stats); the output columns are HOST arrays, rank-major."""
        n = C.c_uint64()
        per_rank = (C.c_uint64 * self.size)()
        st = JoinStats()
        self._check(self.lib.olap_group_join_u32(self.h, _ptr_table(lfk), _ptr_table(ly), _u64_table(lrows), len(lfk), _ptr_table(rpk),
                                                 _ptr_table(rx), _u64_table(rrows), len(rpk), in_mem, out_fk, out_y, out_x, out_capacity,
                                                 C.byref(n), per_rank, C.byref(st)))
        return n.value, [int(v) for v in per_rank], st.as_dict()
