"""The C++ host side (dpu_olap_amd/host): Arrow operator classes over the C ABI, the Arrow
CPU "Native" operators and the Google-Benchmark-compatible harness.  CPU part: it builds,
the Native operators pass their known-answer tests, the benchmark registry lists the
reference's names/ArgNames and BM_FilterNative (BASELINE config 0, plumbing only) runs and
emits GB-style JSON.  GPU part: the whole C++ test binary (Gpu-vs-Native equality, the
reference's test cases) and the *Gpu benchmarks run on the MI355X."""
import json
import os
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "dpu_olap_amd", "host")
BENCH = os.path.join(HOST, "build", "olap-query-host")
TEST = os.path.join(HOST, "build", "olap-query-host_test")


@pytest.fixture(scope="module")
def host_build():
    from dpu_olap_amd import build
    build.build()            # libolap_gpu.so first (the host links against it)
    build.build_host()
    assert os.path.exists(BENCH) and os.path.exists(TEST)
    return True


def test_native_operators_and_registry(host_build):
    out = subprocess.run([TEST, "--gtest_filter=Native"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "TakeTest.SimpleTestNative" in out.stdout and "SumTest.SimpleTestNative" in out.stdout
    lst = subprocess.run([TEST, "--gtest_list_tests"], capture_output=True, text=True, timeout=60).stdout.split()
    for name in ("FilterTest.SimpleTest", "FilterTest.ResultTest", "FilterTest.LongerTest", "TakeTest.LargeTest",
                 "SumTest.LargeTest", "JoinTest.SimpleTest", "JoinTest.LargeTest", "PartitionTest.SimpleTest",
                 "PartitionTest.LargeTest"):
        assert name in lst  # the reference's test cases
    env = dict(os.environ, SF="2", NR_GPUS="1")
    names = subprocess.run([BENCH, "--benchmark_list_tests"], capture_output=True, text=True, env=env, timeout=60).stdout
    for frag in ("BM_FilterNative/Batches:256/Batch-Size:65536/Threads:", "BM_FilterGpu/Batches:256/Batch-Size:65536/GPUs:1",
                 "BM_TakeGpu/Batches:2/Batch-Size:4194304/Indices-Batch-Size:524288/GPUs:1",
                 "BM_SumGpu/Batches:64/Batch-Size:65536/GPUs:1", "BM_JoinNative/Batches:64/L-Batch-Size:65536/R-Batch-Size:65536/Partitioned:1",
                 "BM_JoinGpu/Batches:2/L-Batch-Size:2097152/R-Batch-Size:2097152/GPUs:1",
                 "BM_PartitionGpu/Batches:2/Batch-Size:2097152/GPUs:1/Partitions:2",
                 "BM_JoinGpuResident/Batches:2/L-Batch-Size:2097152/R-Batch-Size:2097152/GPUs:1",
                 "BM_FilterGpuResident/Batches:256/Batch-Size:65536/GPUs:1"):
        assert frag in names, frag


def test_bm_filter_native_plumbing_json(host_build, tmp_path):
    """BASELINE.json configs[0]: BM_FilterNative on the Arrow CPU path (reduced SF here)."""
    out_json = tmp_path / "bm.json"
    env = dict(os.environ, SF="1")
    r = subprocess.run([BENCH, "--benchmark_filter=BM_FilterNative", f"--benchmark_out={out_json}", "--benchmark_out_format=json"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.load(open(out_json))
    assert d["context"]["SF"] == "1" and "NR_GPUS" in d["context"]
    b = [x for x in d["benchmarks"] if "Batches:128" in x["name"]][0]
    assert not b.get("error_occurred") and b["iterations"] == 1 and b["time_unit"] == "ms"
    assert b["Processed (AVG)"] == 128 * 65536
    assert abs(b["Results (AVG)"] / b["Processed (AVG)"] - 0.25) < 0.01
    assert b["items_per_second"] > 0 and abs(b["bytes_per_second"] / b["items_per_second"] - 4) < 1e-6
    # name format understood by GB-JSON tooling (scripts/parse_results.py style: '/' separated, Key:Value)
    parts = dict(p.split(":") for p in b["name"].split("/") if ":" in p)
    assert parts["Batches"] == "128" and parts["Batch-Size"] == "65536"
    skipped = [x for x in d["benchmarks"] if "Batches:0" in x["name"]][0]
    assert skipped["error_occurred"]
    # the CSV flattening used by the sweep script (role of the reference's scripts/parse_results.py)
    import csv
    r = subprocess.run(["python3", os.path.join(ROOT, "scripts", "results_to_csv.py"), str(tmp_path)], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rows = list(csv.DictReader(open(tmp_path / "bm.csv")))
    assert rows and all(row["operator"] == "BM_FilterNative" for row in rows)
    assert rows[0]["Batches"] == "128" and rows[0]["Batch-Size"] == "65536" and float(rows[0]["real_time"]) > 0
    assert not any(row["Batches"] == "0" for row in rows)          # skipped rows are dropped


@pytest.mark.gpu
def test_cpp_gpu_vs_native_suite(host_build):
    out = subprocess.run([TEST], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "0 failed" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 4])
def test_cpp_suite_across_a_device_group_in_loopback(host_build, gpus):
    """The reference's test cases (FilterTest / TakeTest / SumTest / JoinTest ...) with the operators spanning NR_GPUS
    ranks like the *Dpu operators span the DpuSet: batches dealt to the ranks, and for the join the owner split, the
    exchange and the per-owner joins of olap_group_join_u32 -- as logical ranks on the one GPU of this box
    (OLAP_GROUP_LOOPBACK=1; PartitionTest runs on rank 0)."""
    env = dict(os.environ, NR_GPUS=str(gpus), OLAP_GROUP_LOOPBACK="1")
    out = subprocess.run([TEST], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "0 failed" in out.stdout


@pytest.mark.gpu
def test_cpp_benchmark_refuses_a_gpu_count_it_cannot_honour(host_build, tmp_path):
    """A line named GPUs:N ran on N ranks: on this one-GPU box NR_GPUS=2 is an error unless the loopback mode is asked
    for, and then the label says so."""
    out_json = tmp_path / "bm.json"
    args = [BENCH, "--benchmark_filter=BM_(Join|Sum)Gpu/", f"--benchmark_out={out_json}", "--benchmark_out_format=json"]
    r = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, SF="2", NR_GPUS="2"), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    d = json.load(open(out_json))
    real = [b for b in d["benchmarks"] if "Batches:0" not in b["name"]]
    assert real and all(b.get("error_occurred") and "olap_group_create(2 devices)" in b["error_message"] for b in real), real
    r = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, SF="2", NR_GPUS="2", OLAP_GROUP_LOOPBACK="1"), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    d = json.load(open(out_json))
    real = [b for b in d["benchmarks"] if "Batches:0" not in b["name"]]
    assert real and all(not b.get("error_occurred") and "GPUs:2" in b["name"] and "loopback" in b.get("label", "") for b in real), real


@pytest.mark.gpu
def test_cpp_gpu_benchmarks_run(host_build, tmp_path):
    out_json = tmp_path / "bm.json"
    env = dict(os.environ, SF="2")
    r = subprocess.run([BENCH, "--benchmark_filter=Gpu", f"--benchmark_out={out_json}", "--benchmark_out_format=json"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    d = json.load(open(out_json))
    ran = [b for b in d["benchmarks"] if not b.get("error_occurred")]
    names = " ".join(b["name"] for b in ran)
    for frag in ("BM_FilterGpu", "BM_TakeGpu", "BM_SumGpu", "BM_JoinGpu", "BM_PartitionGpu", "BM_FilterGpuResident", "BM_SumGpuResident",
                 "BM_TakeGpuResident", "BM_JoinGpuResident"):
        assert frag in names
    jr = [b for b in ran if b["name"].startswith("BM_JoinGpuResident")][0]      # device-resident: no copies, all rows joined
    assert jr["iterations"] == 3 and jr["gpu-work"] > 0 and "copy-to-gpu" not in jr
    fr = [b for b in ran if b["name"].startswith("BM_FilterGpuResident")][0]
    assert abs(fr["Results (AVG)"] / (256 * 65536) - 0.25) < 0.01
    f = [b for b in ran if "BM_FilterGpu" in b["name"]][0]
    assert abs(f["Results (AVG)"] / f["Processed (AVG)"] - 0.25) < 0.01
    assert f["gpu-work"] > 0 and f["copy-to-gpu"] > 0  # the reference's timer counters, renamed
