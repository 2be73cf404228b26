"""bench.py's multi-rank path end to end on the one GPU a gpurun box has: 2 and 4 ranks share
cuda:0 (OLAP_BENCH_SHARE_GPU=1) and exchange over gloo (RCCL refuses two ranks on one device),
launched exactly like the driver launches N > 1 (torch.distributed.run, one process per rank).
Everything else is the product path: device generator, fused owner split
(olap_partition_tuples_u32 with 512 buckets), counts + tuple exchange, olap_join_presplit_u32
on the received chunks, digest verification across ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world,sf", [(2, 16), (4, 16)])
def test_bench_multirank_on_one_gpu(world, sf):
    env = dict(os.environ, OLAP_BENCH_SHARE_GPU="1", OLAP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--sf", str(sf), "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE json line
    d = json.loads(lines[0])
    assert d["verified"] and d["n_gpus"] == world and d["rows_out"] == sf * (2 << 20)
    assert d["config"]["exchange"]["path"] == "fused" and d["scaling"] == "strong"
    assert d["config"]["radix"]["missing_keys"] == 0
    assert d["unit"] == "Mrows/s" and d["value"] > 0 and d["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["table", "group"])
def test_bench_exchange_path_over_rccl_at_world_size_one(order):
    """The multi-GPU route of bench.py -- fused split -> all_to_all_single over RCCL ("nccl") -> olap_join_presplit_u32
    per sub-bucket group -- executed on the ONE GPU of a gpurun box: a process group of one rank
    (init_process_group("nccl", device_id=...)), forced with --force-exchange.  A failing RCCL call is an error here
    (dist_join falls back to pairwise isend/irecv on gloo only).  SF=16, digest verified."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--sf", "16",
           "--no-cpu-baseline", "--force-exchange", "--exchange-order", order]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    ex = d["config"]["exchange"]
    assert d["verified"] and d["rows_out"] == 16 * (2 << 20)
    assert ex["backend"] == "nccl" and ex["path"] == "fused" and ex["forced_at_world_1"] and ex["order"] == order
    assert ex["groups"] == 4 and ex["left_slices"] == 2
    assert d["config"]["radix"]["missing_keys"] == 0 and "rank0_wall_ms_per_step" in d


@pytest.mark.gpu
def test_bench_exchange_over_rccl_in_rounds():
    """This RCCL delivers a message of more than 1 GiB only in part, silently (scripts/dbg_a2a.py; found at SF=256 with
    --force-exchange), so dist_join ships oversized messages in rounds of list-form all_to_alls over views of the same
    buffers.  Here the limit is lowered to 2^20 tuples: at SF=16 every group message (2^22 tuples) travels in four rounds."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--sf", "16",
           "--no-cpu-baseline", "--force-exchange"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OLAP_MAX_MESSAGE_ELEMS=str(1 << 20))
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["verified"] and d["rows_out"] == 16 * (2 << 20) and d["config"]["exchange"]["backend"] == "nccl"
    assert d["config"]["radix"]["missing_keys"] == 0


@pytest.mark.gpu
def test_bench_ops_two_ranks_on_one_gpu():
    """filter / sum / take shard by batch with no data-path communication: two ranks
    share cuda:0, meet only for the barrier, the max time and the final add; results verified
    per rank against the generator."""
    env = dict(os.environ, OLAP_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench_ops.py"), "--sf", "8", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 8                       # 2 filter + 3 sum + 3 take, printed by rank 0 only
    for d in lines:
        assert d["verified"] and d["n_gpus"] == 2 and "GPUs:2" in d["metric"] and d["value"] > 0
    f = lines[0]
    assert f["metric"].startswith("BM_FilterGpu/Batches:1024/") and abs(f["results"] / (1024 * 65536) - 0.25) < 0.01
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench_ops.py"), "--sf", "8", "--steps", "1", "--warmup", "0", "--only", "sum"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    single = [json.loads(l) for l in one.stdout.splitlines() if l.startswith("{")]
    assert [d["result"] for d in single] == [d["result"] for d in lines[2:5]]   # the sharded sums add up to the single-GPU sums
