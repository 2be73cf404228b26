"""pyarrow front end over include/olap_arrow.h (Arrow C Data Interface): pyarrow arrays go in
by their exported struct ArrowArray (no copy), results come back as pyarrow arrays that own
the library's buffers.  This is the binding a Python host of the reference would use in place
of dpuext's arrow_copy_to_dpus / arrow_copy_from_dpus (host/dpuext/arrow_utils.cc:47-73,
147-266); the same C entry points serve Go (arrow/cdata), Rust (arrow::ffi) and Java.
pyarrow is imported lazily: it is not a dependency of the package."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

from . import capi


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))), ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.CFUNCTYPE(None, C.POINTER(ArrowSchema))), ("private_data", C.c_void_p),
]
ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)), ("release", C.CFUNCTYPE(None, C.POINTER(ArrowArray))), ("private_data", C.c_void_p),
]

EXPORTS = ["olap_arrow_filter_lt", "olap_arrow_sum", "olap_arrow_take", "olap_arrow_join_u32", "olap_arrow_join_u64"]
_bound = False


def _lib():
    global _bound
    L = capi.load_library()
    if not _bound:
        vp, u64 = C.c_void_p, C.c_uint64
        pp = C.POINTER(C.POINTER(ArrowArray))
        L.olap_arrow_filter_lt.argtypes = [vp, C.POINTER(ArrowSchema), pp, u64, vp, C.POINTER(ArrowArray), C.POINTER(u64)]
        L.olap_arrow_sum.argtypes = [vp, C.POINTER(ArrowSchema), pp, u64, C.POINTER(u64)]
        L.olap_arrow_take.argtypes = [vp, C.POINTER(ArrowSchema), pp, pp, u64, C.POINTER(ArrowArray)]
        L.olap_arrow_join_u32.argtypes = [vp, pp, pp, u64, pp, pp, u64, C.POINTER(ArrowArray), C.POINTER(capi.JoinStats)]
        L.olap_arrow_join_u64.argtypes = [vp, pp, pp, u64, pp, pp, u64, C.POINTER(ArrowArray), C.POINTER(capi.JoinStats)]
        for name in EXPORTS:
            getattr(L, name).restype = C.c_int
        _bound = True
    return L


class _Exported:
    """struct ArrowArray views of pyarrow arrays, released when the call is over."""

    def __init__(self, arrays: Sequence):
        self.structs = (ArrowArray * max(len(arrays), 1))()
        self.keep = list(arrays)
        for k, a in enumerate(arrays):
            a._export_to_c(C.addressof(self.structs[k]))
        self.table = (C.POINTER(ArrowArray) * max(len(arrays), 1))(*[C.pointer(self.structs[k]) for k in range(len(arrays))])

    def release(self):
        for s in self.structs:
            if s.release:
                s.release(C.byref(s))


def _schema_of(pa_type):
    s = ArrowSchema()
    pa_type._export_to_c(C.addressof(s))
    return s


def _release_schema(s):
    if s.release:
        s.release(C.byref(s))


def _chunks(column) -> list:
    import pyarrow as pa
    if isinstance(column, pa.ChunkedArray):
        return list(column.chunks)
    if isinstance(column, pa.Array):
        return [column]
    return list(column)


def _import(structs, n: int, pa_type) -> list:
    import pyarrow as pa
    return [pa.Array._import_from_c(C.addressof(structs[k]), pa_type) for k in range(n)]


def filter_lt(ctx: capi.Context, column, threshold: int):
    """column: pa.ChunkedArray / list of pa.Array of one integer type -> pa.ChunkedArray of the values
    below `threshold`, one chunk per input chunk (FilterDpu::GetResult)."""
    import pyarrow as pa
    chunks = _chunks(column)
    typ = chunks[0].type if chunks else pa.uint32()
    ct = {"uint32": C.c_uint32, "int32": C.c_int32, "uint64": C.c_uint64, "int64": C.c_int64}[str(typ)]
    thr = ct(threshold)
    ins, schema = _Exported(chunks), _schema_of(typ)
    outs = (ArrowArray * max(len(chunks), 1))()
    total = C.c_uint64()
    try:
        ctx._check(_lib().olap_arrow_filter_lt(ctx.h, C.byref(schema), ins.table, len(chunks), C.cast(C.byref(thr), C.c_void_p),
                                               outs, C.byref(total)))
    finally:
        ins.release()
        _release_schema(schema)
    return pa.chunked_array(_import(outs, len(chunks), typ), type=typ), total.value


def sum_(ctx: capi.Context, column) -> int:
    import pyarrow as pa
    chunks = _chunks(column)
    typ = chunks[0].type if chunks else pa.uint32()
    ins, schema = _Exported(chunks), _schema_of(typ)
    out = C.c_uint64()
    try:
        ctx._check(_lib().olap_arrow_sum(ctx.h, C.byref(schema), ins.table, len(chunks), C.byref(out)))
    finally:
        ins.release()
        _release_schema(schema)
    v = out.value
    return v - (1 << 64) if str(typ) in ("int32", "int64") and v >= 1 << 63 else v


def take(ctx: capi.Context, data, indices):
    """Per chunk b: data[b][indices[b]] (uint32 indices, no bounds check) -> list of pa.Array."""
    import pyarrow as pa
    d, i = _chunks(data), _chunks(indices)
    if len(d) != len(i):
        raise ValueError("data and indices need the same number of chunks")
    typ = d[0].type if d else pa.uint32()
    dd, ii, schema = _Exported(d), _Exported(i), _schema_of(typ)
    outs = (ArrowArray * max(len(d), 1))()
    try:
        ctx._check(_lib().olap_arrow_take(ctx.h, C.byref(schema), dd.table, ii.table, len(d), outs))
    finally:
        dd.release()
        ii.release()
        _release_schema(schema)
    return _import(outs, len(d), typ)


def join(ctx: capi.Context, left_fk, left_y, right_pk, right_x):
    """PK-FK join of chunked columns -> (pa.Table{fk, y, x}, stats): uint32 keys (olap_arrow_join_u32) or uint64 / int64 keys
    (olap_arrow_join_u64; fk keeps the key column's type), uint32 payloads.  left_y / right_x None: y / x are the global row
    indices (selection vectors)."""
    import pyarrow as pa
    cols = [_chunks(c) if c is not None else None for c in (left_fk, left_y, right_pk, right_x)]
    ktype = cols[0][0].type if cols[0] else pa.uint32()
    wide = str(ktype) in ("uint64", "int64")
    ex = [_Exported(c) if c is not None else None for c in cols]
    outs = (ArrowArray * 3)()
    st = capi.JoinStats()
    tbl = lambda e: e.table if e is not None else None
    fn = _lib().olap_arrow_join_u64 if wide else _lib().olap_arrow_join_u32
    try:
        ctx._check(fn(ctx.h, tbl(ex[0]), tbl(ex[1]), len(cols[0]), tbl(ex[2]), tbl(ex[3]), len(cols[2]), outs, C.byref(st)))
    finally:
        for e in ex:
            if e is not None:
                e.release()
    if wide:
        fk = pa.Array._import_from_c(C.addressof(outs[0]), ktype)
        y = pa.Array._import_from_c(C.addressof(outs[1]), pa.uint32())
        x = pa.Array._import_from_c(C.addressof(outs[2]), pa.uint32())
    else:
        fk, y, x = _import(outs, 3, pa.uint32())
    return pa.table({"fk": fk, "y": y, "x": x}), st.as_dict()
