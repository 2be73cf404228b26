#!/usr/bin/env python3
"""Generates tests/golden/*.npz with Apache Arrow (pyarrow) -- the library the
reference's "Native" operators delegate to (host/filter/filter_native.cc:59-62
`less(field_ref("v"), literal(1<<30))` + filter node; host/aggr/aggr_native.cc:68-73
"sum" aggregate; host/take/take_native.cc:27 `Take(..., NoBoundsCheck)`;
host/join/join_native.cc:31-40 `HashJoinNodeOptions{INNER, {"fk"}, {"pk"}}`).

The reference pins apache-arrow-8.0.0 (Dependencies.cmake:46-55); this image has
pyarrow 25.0.0.  The four operations are integer-exact and version-stable.

Nothing from /root/reference is imported, executed or copied: the script only
calls pyarrow on inputs it synthesises (numpy PCG64, fixed seeds) and on the
literal test inputs of the reference's own tests (transcribed as data in
reference_kats.json).  Run:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

HERE = os.path.dirname(os.path.abspath(__file__))
NP2PA = {"uint32": pa.uint32(), "int32": pa.int32(), "uint64": pa.uint64(), "int64": pa.int64()}


def arrow_filter_lt(v: np.ndarray, thr: int) -> np.ndarray:
    a = pa.array(v, type=NP2PA[str(v.dtype)])
    mask = pc.less(a, pa.scalar(thr, type=a.type))
    return pc.filter(a, mask).to_numpy(zero_copy_only=False).astype(v.dtype)


def arrow_sum(v: np.ndarray) -> int:
    a = pa.array(v, type=NP2PA[str(v.dtype)])
    return int(pc.sum(a).as_py())


def arrow_take(v: np.ndarray, idx: np.ndarray) -> np.ndarray:
    a = pa.array(v, type=NP2PA[str(v.dtype)])
    i = pa.array(idx, type=pa.uint32())
    return pc.take(a, i, boundscheck=False).to_numpy(zero_copy_only=False).astype(v.dtype)


def arrow_join(fk, y, pk, x):
    """Acero inner hash join fk = pk; returns rows sorted by (fk, y) like
    host/join/join_test.cc:27-38,116-120 does before comparing."""
    left = pa.table({"fk": pa.array(fk, pa.uint32()), "y": pa.array(y, pa.uint32())})
    right = pa.table({"pk": pa.array(pk, pa.uint32()), "x": pa.array(x, pa.uint32())})
    out = left.join(right, keys="fk", right_keys="pk", join_type="inner")
    out = out.sort_by([("fk", "ascending"), ("y", "ascending")])
    # canonical right row index for every left row (pc.index_in), left order
    ridx = pc.index_in(left["fk"], value_set=right["pk"].combine_chunks()).to_numpy(zero_copy_only=False)
    return (
        out["fk"].to_numpy().astype(np.uint32),
        out["y"].to_numpy().astype(np.uint32),
        out["x"].to_numpy().astype(np.uint32),
        ridx.astype(np.uint32),
    )


def main():
    rng = np.random.Generator(np.random.PCG64(42))
    out = {}

    # ---- reference KATs re-run through Arrow (reference_kats.json holds the literals)
    kats = json.load(open(os.path.join(HERE, "reference_kats.json")))
    small = np.array(kats["small_column"], dtype=np.uint32)
    out["kat_filter_small"] = arrow_filter_lt(small, 1 << 30)
    out["kat_sum_small"] = np.array([arrow_sum(small)], dtype=np.uint64)
    out["kat_take_small"] = arrow_take(small, np.array(kats["take_indices"], dtype=np.uint32))
    pat = np.arange(4096, dtype=np.uint32) + np.uint32(1 << 30)
    for i in kats["filter_result_test"]["small_positions"]:
        pat[i] = i
    out["kat_filter_pattern"] = arrow_filter_lt(pat, 1 << 30)
    js = kats["join_simple_test"]
    fk = np.array(sum(js["left_fk"], []), dtype=np.uint32)
    y = np.array(sum(js["left_v_l"], []), dtype=np.uint32)
    pk = np.array(sum(js["right_pk"], []), dtype=np.uint32)
    x = np.array(sum(js["right_v_r"], []), dtype=np.uint32)
    jf, jy, jx, jr = arrow_join(fk, y, pk, x)
    out["kat_join_fk"], out["kat_join_y"], out["kat_join_x"], out["kat_join_ridx"] = jf, jy, jx, jr

    # ---- boundary values around the predicate (unsigned compare!)
    edge = np.array([0, 2, 3, (1 << 30) - 1, 1 << 30, (1 << 31) + 5, (1 << 32) - 1], dtype=np.uint32)
    out["edge_u32_in"] = edge
    out["edge_u32_out"] = arrow_filter_lt(edge, 1 << 30)

    # ---- randomized filter / sum / take, all four integer types
    for name, dt, lo, hi, thr in (
        ("u32", np.uint32, 0, 1 << 32, 1 << 30),
        ("i32", np.int32, -(1 << 31), 1 << 31, -(1 << 30)),
        ("u64", np.uint64, 0, 1 << 64, 1 << 62),
        ("i64", np.int64, -(1 << 63), 1 << 63, -(1 << 62)),
    ):
        # ragged batches incl. an empty one and non-multiple-of-tile sizes
        sizes = [0, 1, 63, 64, 65, 1000, 4096, 4097, 5003]
        vals = [rng.integers(lo, hi, size=s, dtype=dt, endpoint=False) for s in sizes]
        allv = np.concatenate(vals)
        out[f"rand_{name}_values"] = allv
        out[f"rand_{name}_sizes"] = np.array(sizes, dtype=np.uint64)
        out[f"rand_{name}_thr"] = np.array([thr], dtype=dt)
        filt = [arrow_filter_lt(v, thr) for v in vals]
        out[f"rand_{name}_filter_out"] = np.concatenate(filt) if filt else np.empty(0, dt)
        out[f"rand_{name}_filter_counts"] = np.array([f.size for f in filt], dtype=np.uint64)
        # Arrow sums u32 into u64 and i32 into i64, u64/i64 wrap silently
        sdt = np.uint64 if name[0] == "u" else np.int64
        out[f"rand_{name}_sum"] = np.array([arrow_sum(allv) if allv.size else 0], dtype=sdt)
        data = vals[-1]
        idx = rng.integers(0, data.size, size=1237, dtype=np.uint32)
        out[f"rand_{name}_take_idx"] = idx
        out[f"rand_{name}_take_out"] = arrow_take(data, idx)

    # ---- randomized joins
    # (a) generator-shaped: pk = global row index, fk of batch i in [i*R, (i+1)*R)
    nb, R, Lr = 4, 2048, 3000
    pk = np.arange(nb * R, dtype=np.uint32)
    x = rng.integers(0, 1 << 32, size=nb * R, dtype=np.uint32)
    fk = np.concatenate([rng.integers(i * R, (i + 1) * R, size=Lr, dtype=np.uint32) for i in range(nb)])
    y = rng.integers(0, 1 << 32, size=nb * Lr, dtype=np.uint32)
    out["join_a_fk"], out["join_a_y"], out["join_a_pk"], out["join_a_x"] = fk, y, pk, x
    out["join_a_left_rows"] = np.array([Lr] * nb, dtype=np.uint32)
    out["join_a_right_rows"] = np.array([R] * nb, dtype=np.uint32)
    jf, jy, jx, jr = arrow_join(fk, y, pk, x)
    out["join_a_out_fk"], out["join_a_out_y"], out["join_a_out_x"], out["join_a_ridx"] = jf, jy, jx, jr
    # (b) sparse shuffled unique keys over the full u32 domain incl. 0 and 0xFFFFFFFF,
    # keys shuffled across ragged batches (JoinTest.SimpleTest style), L != R
    pk = np.unique(np.concatenate([rng.integers(0, 1 << 32, size=9000, dtype=np.uint32),
                                   np.array([0, 0xFFFFFFFF, 1 << 30, 1 << 31], dtype=np.uint32)]))
    rng.shuffle(pk)
    x = rng.integers(0, 1 << 32, size=pk.size, dtype=np.uint32)
    fk = pk[rng.integers(0, pk.size, size=12007)]
    fk[:4] = np.array([0xFFFFFFFF, 0, 0xFFFFFFFF, 1 << 31], dtype=np.uint32)
    y = rng.integers(0, 1 << 32, size=fk.size, dtype=np.uint32)
    out["join_b_fk"], out["join_b_y"], out["join_b_pk"], out["join_b_x"] = fk, y, pk, x
    out["join_b_left_rows"] = np.array([3000, 1, 0, 6006, 3000], dtype=np.uint32)
    rr = [pk.size // 3, pk.size // 3, pk.size - 2 * (pk.size // 3)]
    out["join_b_right_rows"] = np.array(rr, dtype=np.uint32)
    jf, jy, jx, jr = arrow_join(fk, y, pk, x)
    out["join_b_out_fk"], out["join_b_out_y"], out["join_b_out_x"], out["join_b_ridx"] = jf, jy, jx, jr

    # ---- joins on 64-bit keys (the reference's HT_64BIT_KEYS mode, hashtable.h:13-18): uint64 keys spread over the
    # whole domain incl. 0, 2^64 - 1 and keys that differ only above bit 32; int64 keys incl. negatives
    for name, kdt in (("c", np.uint64), ("d", np.int64)):
        if kdt is np.uint64:
            pk = np.unique(np.concatenate([rng.integers(0, 1 << 64, size=7000, dtype=np.uint64, endpoint=False),
                                           np.array([0, (1 << 64) - 1, 1 << 32, (1 << 32) + 5, 5, 1 << 63], dtype=np.uint64)]))
        else:
            pk = np.unique(np.concatenate([rng.integers(-(1 << 63), 1 << 63, size=7000, dtype=np.int64, endpoint=False),
                                           np.array([0, -1, -(1 << 63), (1 << 63) - 1, 1 << 32, -(1 << 32)], dtype=np.int64)]))
        rng.shuffle(pk)
        x = rng.integers(0, 1 << 32, size=pk.size, dtype=np.uint32)
        fk = pk[rng.integers(0, pk.size, size=10007)]
        fk[:6] = pk[:6]
        y = rng.integers(0, 1 << 32, size=fk.size, dtype=np.uint32)
        left = pa.table({"fk": pa.array(fk), "y": pa.array(y, pa.uint32())})
        right = pa.table({"pk": pa.array(pk), "x": pa.array(x, pa.uint32())})
        j = left.join(right, keys="fk", right_keys="pk", join_type="inner").sort_by([("fk", "ascending"), ("y", "ascending"), ("x", "ascending")])
        out[f"join_{name}_fk"], out[f"join_{name}_y"], out[f"join_{name}_pk"], out[f"join_{name}_x"] = fk, y, pk, x
        out[f"join_{name}_left_rows"] = np.array([4000, 0, 6007], dtype=np.uint32)
        out[f"join_{name}_right_rows"] = np.array([pk.size // 2, pk.size - pk.size // 2], dtype=np.uint32)
        out[f"join_{name}_out_fk"] = j["fk"].to_numpy().astype(kdt)
        out[f"join_{name}_out_y"] = j["y"].to_numpy().astype(np.uint32)
        out[f"join_{name}_out_x"] = j["x"].to_numpy().astype(np.uint32)

    path = os.path.join(HERE, "arrow_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; pyarrow", pa.__version__)


if __name__ == "__main__":
    main()
