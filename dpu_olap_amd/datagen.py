"""Host (numpy) mirror of the device generator and digests in csrc/gen.hip.

Distributions follow the reference's generator (host/generator/generator.cc:22-71):
uniform u32 payload columns, pk = 0,1,2,... (MakeIndexColumn :59-71), fk of left
batch i uniform in [i*R, (i+1)*R-1] (MakeForeignKeyColumn :46-57).  The bit
stream is our own counter-based one -- value(seed, stream, row) -- identical on
host and device, so any slice of an SF=2048 column can be re-derived.
Bit-equality with Arrow's RandomArrayGenerator(42) is not required (the
reference publishes no numbers that depend on it)."""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
STREAM_V = 0    # filter / sum / take data column "v"
STREAM_IDX = 1  # take indices
STREAM_X = 2    # join right payload "x"
STREAM_Y = 3    # join left payload "y"
STREAM_FK = 4   # join left key "fk"
SEED = 42       # the reference seeds its RNG with 42 (filter_benchmark.cc:76, join_benchmark.cc:69)


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (common.h mix64)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gen64(seed: int, stream: int, rows: np.ndarray) -> np.ndarray:
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = np.uint64(seed) ^ (np.uint64(0xD1B54A32D192ED03) * np.uint64(stream + 1))
        base = mix64(np.array([k], dtype=np.uint64))[0]
        return mix64(base + rows)


def gen32(seed: int, stream: int, rows: np.ndarray) -> np.ndarray:
    return (gen64(seed, stream, rows) >> np.uint64(32)).astype(np.uint32)


def uniform_u32(n: int, seed: int, stream: int, first_row: int = 0) -> np.ndarray:
    return gen32(seed, stream, np.arange(first_row, first_row + n, dtype=np.uint64))


def uniform_u64(n: int, seed: int, stream: int, first_row: int = 0) -> np.ndarray:
    return gen64(seed, stream, np.arange(first_row, first_row + n, dtype=np.uint64))


def index_u32(n: int, first_value: int = 0) -> np.ndarray:
    return (np.arange(n, dtype=np.uint64) + np.uint64(first_value)).astype(np.uint32)


def fk_u32(n: int, seed: int, stream: int, first_row: int, left_batch_rows: int, right_batch_rows: int) -> np.ndarray:
    r = np.arange(first_row, first_row + n, dtype=np.uint64)
    batch = r // np.uint64(left_batch_rows)
    u = (gen32(seed, stream, r).astype(np.uint64) * np.uint64(right_batch_rows)) >> np.uint64(32)
    return (batch * np.uint64(right_batch_rows) + u).astype(np.uint32)


def checksum(col: np.ndarray):
    """(sum, xor, sum of mix64) mod 2^64 of a column -- olap_checksum."""
    col = np.ascontiguousarray(col)
    v = col.view(np.uint32).astype(np.uint64) if col.dtype.itemsize == 4 else col.view(np.uint64)
    with np.errstate(over="ignore"):
        s = int(np.add.reduce(v, dtype=np.uint64)) if v.size else 0
        x = int(np.bitwise_xor.reduce(v)) if v.size else 0
        m = int(np.add.reduce(mix64(v), dtype=np.uint64)) if v.size else 0
    return s, x, m


def checksum_rows3(a: np.ndarray, b: np.ndarray, c: np.ndarray):
    """Row-order independent digest of a 3-column u32 table -- olap_checksum_rows3_u32."""
    a64, b64, c64 = (np.asarray(t, dtype=np.uint32).astype(np.uint64) for t in (a, b, c))
    h = mix64(mix64((a64 << np.uint64(32)) | b64) ^ c64)
    with np.errstate(over="ignore"):
        s = int(np.add.reduce(h, dtype=np.uint64)) if h.size else 0
        x = int(np.bitwise_xor.reduce(h)) if h.size else 0
    return s, x
