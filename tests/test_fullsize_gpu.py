"""BASELINE.json's full-size configurations under -m gpu: the GPU parity record must not stop at
SF=64.  Each test runs the bench program of the config as the driver would (a fresh process that owns the
whole GPU) for ONE step at SF=2048 and checks the program's own verification:

  configs[3]  BM_Join u32 PK-FK SF=2048   bench.py: order-independent digest of the 2^32 output rows {fk, y, x}
                                          against the digest predicted from the inputs (pk = row index, so
                                          x = gen(fk)), row count, no missing / duplicate keys
  configs[1]  BM_Filter SF=2048           bench_ops.py: global selectivity + sampled batches bit-exact vs the
                                          generator's host mirror (uint32, and the same bits as int32)
  configs[2]  BM_Sum / BM_Take SF=2048    bench_ops.py: sum equal to an independent reduction kernel's digest
                                          (u32, u32 small batches, int64); take: sampled batches bit-exact

Sizes at which the oracle finishes in seconds are covered by tests/test_gpu_parity.py; here the checks are the
size-independent properties the domain offers."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_join_sf2048_digest():
    (d,) = _run([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--sf", "2048", "--no-cpu-baseline"], 900)
    assert d["config"]["sf"] == 2048 and d["config"]["rows_per_side"] == 1 << 32
    assert d["verified"] is True and d["rows_out"] == 1 << 32
    st = d["config"]["radix"]
    assert st["missing_keys"] == 0 and st["duplicate_keys"] == 0 and st["left_rows"] == st["right_rows"] == 1 << 32
    assert st["passes"] == 2 and st["nr_partitions"] == 1 << 18 and st["max_build_rows"] <= st["table_slots"]
    assert d["dtype"] == "u32" and d["roofline"]["bound"] == "hbm" and d["cold_first_step_ms"] > 0
    names = {k["phase"]: k["kernel"] for k in d["kernels"]}
    # pass 0 from the Arrow columns (hashes in place of the keys, straight-line loader), last pass with 6-byte tuples in the
    # capacity layout, probe over packed partitions (csrc/ring_claim.h, csrc/pack6.h)
    assert names["partition-scatter-claim"].startswith("olap::k_part_claim_ring<") and names["partition-scatter-claim"].endswith("true, 1>")
    assert names["partition-scatter-capacity"].startswith("olap::k_part_scatter_ring6<")
    assert names["join-probe"].startswith("olap::k_join_probe_direct6<") and names["join-probe"].endswith("true, true>")


@pytest.mark.parametrize("op,lines", [("filter", 2), ("sum", 3), ("take", 3)])
def test_scan_ops_sf2048(op, lines):
    out = _run([os.path.join(ROOT, "bench_ops.py"), "--sf", "2048", "--steps", "1", "--warmup", "0", "--only", op], 900)
    assert len(out) == lines, out
    for d in out:
        assert d["verified"] is True and d["sf"] == 2048 and d["n_gpus"] == 1 and d["value"] > 0, d
    if op == "filter":   # 2^34 rows; uniform u32 < 2^30: 25 %, the same bits as int32 (signed): 75 %
        assert [d["metric"].split("/")[3] for d in out] == ["uint32", "int32"]
        assert all("Batches:262144/Batch-Size:65536" in d["metric"] for d in out)
        assert abs(out[0]["selectivity"] - 0.25) < 1e-3 and abs(out[1]["selectivity"] - 0.75) < 1e-3
    if op == "sum":      # 2^32 rows in both batch shapes, then int64
        assert ["Batches:2048/Batch-Size:2097152/uint32" in out[0]["metric"], "Batches:65536/Batch-Size:65536/uint32" in out[1]["metric"],
                "Batches:2048/Batch-Size:2097152/int64" in out[2]["metric"]] == [True] * 3
        assert out[0]["result"] == out[1]["result"]   # the same 2^32 generator rows, cut differently
    if op == "take":     # 2^33 data rows, 2^30 indices
        assert all(d["indices"] == 1 << 30 for d in out)
