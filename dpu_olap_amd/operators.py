"""Python mirror of the reference's operator interface for the hot path.

The reference's operators are duck-typed C++ classes used by the BM_* templates
and the gtests: ctor(device, batches...), `Status Prepare()`, `Result Run()`,
`Timers()` (host/filter/filter_dpu.h:14-29, host/take/take_dpu.h:14-30,
host/aggr/aggr_dpu.h:14-28, host/join/join_dpu.h:14-48).  These classes keep the
names, argument meaning and error behaviour (errors surface as OlapError, the
analogue of a non-ok arrow::Status) and call the HIP kernels through the C ABI
(include/olap_gpu.h) -- nothing here computes on the CPU.

Batches are what arrow::RecordBatchVector is to the reference: a list with one
entry per batch; an entry is a numpy array, a pyarrow Array/ChunkedArray (one
chunk), or -- for device-resident columns -- a DeviceArray.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np

from . import capi
from .capi import Context, OlapError  # noqa: F401  (re-exported)

FILTER_THRESHOLD = 1 << 30  # dpu/shared/kernels/filter.c:25


@dataclass
class DeviceArray:
    """A column chunk resident in HBM (device pointer + length + numpy dtype name)."""
    ptr: int
    rows: int
    dtype: str = "uint32"

    @property
    def nbytes(self) -> int:
        return self.rows * np.dtype(self.dtype).itemsize


class Device(Context):
    """dpu::DpuSet's stand-in: owns the GPU context (and HBM buffers it hands out)."""

    def __init__(self, device_id: int = 0):
        super().__init__(device_id)
        self._owned: list[int] = []

    def empty(self, rows: int, dtype="uint32") -> DeviceArray:
        dt = np.dtype(dtype)
        p = self.malloc(max(rows, 1) * dt.itemsize)
        self._owned.append(p)
        return DeviceArray(p, rows, dt.name)

    def to_device(self, a: np.ndarray) -> DeviceArray:
        a = np.ascontiguousarray(a)
        d = self.empty(a.size, a.dtype)
        if a.size:
            self.h2d(d.ptr, a.ctypes.data, a.nbytes)
        return d

    def to_host(self, d: DeviceArray, rows: int | None = None) -> np.ndarray:
        n = d.rows if rows is None else rows
        out = np.empty(n, dtype=d.dtype)
        if n:
            self.d2h(out.ctypes.data, d.ptr, out.nbytes)
        return out

    def release(self, d: DeviceArray):
        if d.ptr in self._owned:
            self._owned.remove(d.ptr)
            self.free(d.ptr)

    def close(self):
        if getattr(self, "h", None):
            for p in self._owned:
                try:
                    self.free(p)
                except Exception:
                    pass
            self._owned.clear()
        super().close()


def _as_column(x):
    """-> numpy array (host) or DeviceArray."""
    if isinstance(x, DeviceArray):
        return x
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x)
    # pyarrow Array / ChunkedArray without importing pyarrow here
    if hasattr(x, "to_numpy"):
        try:
            return np.ascontiguousarray(x.to_numpy(zero_copy_only=False))
        except TypeError:
            return np.ascontiguousarray(x.to_numpy())
    return np.ascontiguousarray(np.asarray(x))


def _columns(batches: Sequence, column=0) -> list:
    """Picks one column out of a vector of batches.  A batch may be a bare array
    (single-column batch), a tuple/list of arrays, a dict, or a pyarrow RecordBatch."""
    out = []
    for b in batches:
        if isinstance(b, (np.ndarray, DeviceArray)):
            out.append(_as_column(b))
        elif isinstance(b, dict):
            out.append(_as_column(b[column] if column in b else list(b.values())[column]))
        elif isinstance(b, (tuple, list)):
            out.append(_as_column(b[column]))
        elif hasattr(b, "column"):  # pyarrow.RecordBatch
            out.append(_as_column(b.column(column)))
        else:
            out.append(_as_column(b))
    return out


def _table(cols: list):
    """-> (ptrs, rows, mem, dtype_name); all chunks must live in the same space."""
    if not cols:
        return [], [], capi.MEM_HOST, "uint32"
    dev = [isinstance(c, DeviceArray) for c in cols]
    if any(dev) and not all(dev):
        raise ValueError("mixing host and device batches in one column is not supported")
    if all(dev):
        return [c.ptr for c in cols], [c.rows for c in cols], capi.MEM_DEVICE, cols[0].dtype
    dts = {c.dtype.name for c in cols}
    if len(dts) != 1:
        raise TypeError(f"batches disagree on dtype: {dts}")
    return [c.ctypes.data for c in cols], [c.size for c in cols], capi.MEM_HOST, cols[0].dtype.name


def _dtype_code(name: str) -> int:
    try:
        return capi.DTYPE_FROM_NUMPY[name]
    except KeyError:
        raise TypeError(f"unsupported column type {name}; the hot path covers uint32/int32/uint64/int64") from None


class _Op:
    def __init__(self, system: Device):
        self.system_ = system
        self._prepared = False

    def Prepare(self):
        """The reference loads the DPU binary here (e.g. host/filter/filter_dpu.cc:23-32),
        inside the timed region; kernels are linked in, so this only resets the timers."""
        self.system_.set_profiling(True)
        self.system_.reset_timers()
        self._prepared = True
        return self

    def Timers(self) -> dict:
        """name -> milliseconds of device time (the role of timer::Timers, host/timer/timer.h)."""
        return {k: v["ms"] for k, v in self.system_.timers().items()}

    def _need_prepare(self):
        if not self._prepared:
            raise OlapError(capi.ERR_INVALID, "Prepare() must be called before Run()")


class FilterGpu(_Op):
    """FilterDpu (host/filter/filter_dpu.{h,cc}): per batch, the values v < threshold in
    input order.  GetResult() -> one array per batch (the ChunkedArray's chunks, in batch
    order, filter_dpu.cc:95,162-166); Run() -> total number of results."""

    def __init__(self, system: Device, batches: Sequence, threshold: int = FILTER_THRESHOLD):
        super().__init__(system)
        self.cols = _columns(batches, 0)
        self.threshold = threshold

    def GetResult(self) -> list:
        self._need_prepare()
        ptrs, rows, mem, dt = _table(self.cols)
        code = _dtype_code(dt)
        if mem == capi.MEM_HOST:
            outs = [np.empty(r, dtype=dt) for r in rows]  # worst case, filter_dpu.cc:67-83
            counts, _ = self.system_.filter_lt(code, ptrs, rows, mem, self.threshold, [o.ctypes.data for o in outs], capi.MEM_HOST)
            return [o[:c] for o, c in zip(outs, counts.tolist())]
        outs = [self.system_.empty(r, dt) for r in rows]
        counts, _ = self.system_.filter_lt(code, ptrs, rows, mem, self.threshold, [o.ptr for o in outs], capi.MEM_DEVICE)
        return [DeviceArray(o.ptr, c, dt) for o, c in zip(outs, counts.tolist())]

    def Run(self) -> int:
        return int(sum((r.rows if isinstance(r, DeviceArray) else r.size) for r in self.GetResult()))


class SumGpu(_Op):
    """SumDpu (host/aggr/aggr_dpu.cc:31-89): sum of column 0 over all batches."""

    def __init__(self, system: Device, batches: Sequence):
        super().__init__(system)
        self.cols = _columns(batches, 0)

    def Run(self) -> int:
        self._need_prepare()
        ptrs, rows, mem, dt = _table(self.cols)
        return self.system_.sum(_dtype_code(dt), ptrs, rows, mem)


class TakeGpu(_Op):
    """TakeDpu (host/take/take_dpu.cc:34-104): per batch out[i] = data[idx[i]]; Run() ->
    one array per batch (the result Table's batches)."""

    def __init__(self, system: Device, batches: Sequence, indices_batches: Sequence):
        super().__init__(system)
        self.cols = _columns(batches, 0)
        self.idx = _columns(indices_batches, 0)
        if len(self.cols) != len(self.idx):
            raise ValueError("batches and indices_batches differ in length")

    def Run(self) -> list:
        self._need_prepare()
        dptrs, drows, mem, dt = _table(self.cols)
        idx = [i if isinstance(i, DeviceArray) else np.ascontiguousarray(i, dtype=np.uint32) for i in self.idx]
        iptrs, irows, imem, _ = _table(idx)
        if self.cols and mem != imem:
            raise ValueError("data and index batches must live in the same memory space")
        code = _dtype_code(dt)
        if mem == capi.MEM_HOST:
            outs = [np.empty(r, dtype=dt) for r in irows]
            self.system_.take(code, dptrs, drows, iptrs, irows, mem, [o.ctypes.data for o in outs], capi.MEM_HOST)
            return outs
        outs = [self.system_.empty(r, dt) for r in irows]
        self.system_.take(code, dptrs, drows, iptrs, irows, mem, [o.ptr for o in outs], capi.MEM_DEVICE)
        return outs


class PartitionGpu(_Op):
    """PartitionDpu (host/partition/partition_dpu.cc:50-135; upstream marks it
    non-functional, README.md:114-118): radix partition by the top bits of the Wang hash
    of the key column.  Run() -> list of (keys, payload) per partition."""

    def __init__(self, system: Device, batches: Sequence, nr_partitions: int, key_column=0, payload_column=1):
        super().__init__(system)
        self.keys = _columns(batches, key_column)
        self.payload = None
        if payload_column is not None:
            try:
                self.payload = _columns(batches, payload_column)
            except (IndexError, KeyError, TypeError):
                self.payload = None
        self.nr_partitions = nr_partitions

    @staticmethod
    def _u32_batches(cols):
        """Host batches as contiguous uint32 arrays (the reference's only element type, dpu/shared/kernels/common.h:3);
        the returned list owns whatever had to be copied."""
        return [c if isinstance(c, DeviceArray) else np.ascontiguousarray(c).astype(np.uint32, copy=False) for c in cols]

    def Run(self) -> list:
        self._need_prepare()
        # the uint32 views / copies the pointer tables point into must outlive the C call: _table() returns bare
        # addresses (int32 / int64 host batches are COPIED by the conversion)
        keys = self._u32_batches(self.keys)
        payload = self._u32_batches(self.payload) if self.payload is not None else None
        kp, rows, mem, _ = _table(keys)
        pp = None
        if payload is not None:
            pp, _, pmem, _ = _table(payload)
            if keys and pmem != mem:
                raise ValueError("key and payload batches must live in the same memory space")
        n = int(sum(rows))
        ok = np.empty(max(n, 1), dtype=np.uint32)
        op = np.empty(max(n, 1), dtype=np.uint32)
        self._keep = (keys, payload)
        if mem == capi.MEM_DEVICE:
            dk, dp = self.system_.empty(n, "uint32"), self.system_.empty(n, "uint32")
            offs = self.system_.partition_u32(kp, pp, rows, mem, self.nr_partitions, dk.ptr, dp.ptr, n, capi.MEM_DEVICE)
            ok, op = self.system_.to_host(dk), self.system_.to_host(dp)
            self.system_.release(dk)
            self.system_.release(dp)
        else:
            offs = self.system_.partition_u32(kp, pp, rows, mem, self.nr_partitions, ok.ctypes.data, op.ctypes.data, n, capi.MEM_HOST)
        return [(ok[offs[p]:offs[p + 1]], op[offs[p]:offs[p + 1]]) for p in range(self.nr_partitions)]


class JoinGpu(_Op):
    """JoinDpu (host/join/join_dpu.{h,cc}): inner PK-FK join, left {fk, y} with right
    {pk, x}; Run() -> {"fk", "y", "x"} (left columns then the right non-key column,
    join_dpu.cc:314-363), partition-major row order (sort before comparing, as
    host/join/join_test.cc:27-38 does).  A batch is a (key, payload) pair."""

    def __init__(self, system: Device, left_batches: Sequence, right_batches: Sequence, select: bool = False,
                 drop_unmatched: bool = False):
        """select=True: batches are bare key columns and the result's "y" / "x" are selection
        vectors -- the global left row index of every result row and the global right row index
        of its match (what kernel_hash_probe returns; gather further columns with TakeGpu)."""
        super().__init__(system)
        self.select = select
        self.drop_unmatched = drop_unmatched   # inner join: left rows without a match are dropped instead of raising
        if select:
            self.lfk, self.ly, self.rpk, self.rx = list(left_batches), None, list(right_batches), None
        else:
            self.lfk, self.ly = _columns(left_batches, 0), _columns(left_batches, 1)
            self.rpk, self.rx = _columns(right_batches, 0), _columns(right_batches, 1)
        self.stats: dict = {}

    @staticmethod
    def _u32(cols):
        return [c if isinstance(c, DeviceArray) else np.ascontiguousarray(c, dtype=np.uint32) for c in cols]

    def Run(self) -> dict:
        self._need_prepare()
        self.system_.join_set_unmatched(self.drop_unmatched)
        try:
            return self._run()
        finally:
            self.system_.join_set_unmatched(False)

    @staticmethod
    def _is64(cols) -> bool:
        return any(np.dtype(c.dtype).itemsize == 8 for c in cols)

    def _run64(self) -> dict:
        """64-bit keys (uint64 / int64 bit patterns), 32-bit payloads: olap_join_u64 (the reference's HT_64BIT_KEYS mode)."""
        k64 = lambda cols: [c if isinstance(c, DeviceArray) else np.ascontiguousarray(c).view(np.uint64) if np.dtype(c.dtype) == np.int64
                            else np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
        signed = any(str(np.dtype(c.dtype)) == "int64" for c in list(self.lfk) + list(self.rpk))
        lfk, rpk = k64(self.lfk), k64(self.rpk)
        ly, rx = ([], []) if self.select else (self._u32(self.ly), self._u32(self.rx))
        lp, lrows, lmem, _ = _table(lfk)
        rp, rrows, rmem, _ = _table(rpk)
        lyp = None if self.select else _table(ly)[0]
        rxp = None if self.select else _table(rx)[0]
        if lmem != rmem:
            raise ValueError("all join columns must live in the same memory space")
        n = int(sum(lrows))
        kdt = "int64" if signed else "uint64"
        if lmem == capi.MEM_HOST:
            out = {"fk": np.empty(max(n, 1), dtype=np.uint64), "y": np.empty(max(n, 1), dtype=np.uint32), "x": np.empty(max(n, 1), dtype=np.uint32)}
            rows, self.stats = self.system_.join_u64(lp, lyp, lrows, rp, rxp, rrows, lmem, out["fk"].ctypes.data, out["y"].ctypes.data,
                                                     out["x"].ctypes.data, n, capi.MEM_HOST)
            self._keep = (lfk, rpk, ly, rx)
            return {"fk": out["fk"][:rows].view(kdt), "y": out["y"][:rows], "x": out["x"][:rows]}
        out = {"fk": self.system_.empty(n, "uint64"), "y": self.system_.empty(n, "uint32"), "x": self.system_.empty(n, "uint32")}
        rows, self.stats = self.system_.join_u64(lp, lyp, lrows, rp, rxp, rrows, lmem, out["fk"].ptr, out["y"].ptr, out["x"].ptr, n,
                                                 capi.MEM_DEVICE)
        return {"fk": DeviceArray(out["fk"].ptr, rows, kdt), "y": DeviceArray(out["y"].ptr, rows, "uint32"),
                "x": DeviceArray(out["x"].ptr, rows, "uint32")}

    def _run(self) -> dict:
        if self._is64(self.lfk) or self._is64(self.rpk):
            return self._run64()
        lfk, rpk = self._u32(self.lfk), self._u32(self.rpk)
        ly, rx = ([], []) if self.select else (self._u32(self.ly), self._u32(self.rx))
        lp, lrows, lmem, _ = _table(lfk)
        rp, rrows, rmem, _ = _table(rpk)
        lyp, _, lymem, _ = (None, None, lmem, None) if self.select else _table(ly)
        rxp, _, rxmem, _ = (None, None, rmem, None) if self.select else _table(rx)
        mems = {m for m, c in ((lmem, lfk), (lymem, ly), (rmem, rpk), (rxmem, rx)) if c}
        if len(mems) > 1:
            raise ValueError("all join columns must live in the same memory space")
        mem = mems.pop() if mems else capi.MEM_HOST
        n = int(sum(lrows))
        if mem == capi.MEM_HOST:
            out = {k: np.empty(max(n, 1), dtype=np.uint32) for k in ("fk", "y", "x")}
            rows, self.stats = self.system_.join_u32(lp, lyp, lrows, rp, rxp, rrows, mem, out["fk"].ctypes.data,
                                                     out["y"].ctypes.data, out["x"].ctypes.data, n, capi.MEM_HOST)
            return {k: v[:rows] for k, v in out.items()}
        out = {k: self.system_.empty(n, "uint32") for k in ("fk", "y", "x")}
        rows, self.stats = self.system_.join_u32(lp, lyp, lrows, rp, rxp, rrows, mem, out["fk"].ptr, out["y"].ptr,
                                                 out["x"].ptr, n, capi.MEM_DEVICE)
        return {k: DeviceArray(v.ptr, rows, "uint32") for k, v in out.items()}
