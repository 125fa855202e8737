"""Self-arming N > 1 test: on a box with at least two GPUs the RCCL path ("nccl" backend, one GPU per rank) is validated
without anyone writing code that day; on a one-GPU box it skips and says why.

What it checks when it runs:
  * tests/multi_worker.py under torch.distributed.run with 2 ranks: batch.keypoints_batch over nccl equals a single-rank
    BatchPlan per frame, byte for byte (uneven shard, empty shard), batch.match_sharded with a MatchPlan per rank equals
    MatchPlan.match of the whole lists;
  * bench.py --gpus 2 (its own launcher) and bench.py --config c4 --gpus 2: the process group observed 2 ranks on the
    nccl backend, the exchange ran on device tensors, and the C4 keypoint count per frame equals the one-GPU run's.
The one-GPU rehearsals of the same code paths (gloo, both ranks on cuda:0) are tests/test_gpu_bench_contract.py and, on
CPU, tests/test_batch_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


needs_two = pytest.mark.skipif(n_gpus() < 2, reason="RCCL needs one GPU per rank: %d GPU(s) visible; rehearsed over gloo in "
                                                    "tests/test_gpu_bench_contract.py and tests/test_batch_gloo.py" % n_gpus())


def env():
    e = dict(os.environ, MASTER_ADDR="127.0.0.1")
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


def free_port():
    """a port nobody listens on right now (fixed ports collide with a lingering worker of an earlier run)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def bench(args, timeout=900):
    out = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=timeout, text=True, env=env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def run_worker(extra_env, port=None):
    port = port or free_port()
    e = env()
    e.update(extra_env)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                           "127.0.0.1", "--master-port", str(port), os.path.join("tests", "multi_worker.py")], cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True, env=e)


@needs_two
def test_sharded_paths_over_rccl_equal_one_rank():
    out = run_worker({})
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert "multi ok: 2 ranks over nccl" in out.stdout


def test_worker_rehearsal_on_one_gpu():
    """The same worker script with both ranks on cuda:0 and gloo collectives: what can be checked of it on a one-GPU box
    (its frame / shard / match logic and the host-staged exchange), so that the RCCL run above does not meet the script
    for the first time on the day a second GPU appears."""
    out = run_worker({"SIFT_MULTI_REHEARSAL": "1"})
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert "multi ok: 2 ranks over gloo" in out.stdout


def test_rccl_call_path_on_one_gpu():
    """RCCL itself on the GPU a one-GPU lease has: a world-size-1 "nccl" process group beside libsiftmi.so's HIP runtime, the
    device-tensor collectives of the exchange on records the descriptor kernels wrote (tests/nccl_ws1_worker.py)."""
    e = env()
    e.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT=str(free_port()))
    out = subprocess.run([sys.executable, os.path.join("tests", "nccl_ws1_worker.py")], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600, text=True, env=e)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert "rccl ws1 ok" in out.stdout


@needs_two
def test_bench_two_gpus_over_rccl():
    d = bench(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["config"]["backend"] == "nccl"
    assert "all_gather" in d["config"]["exchange"] and d["scaling"] == "weak"
    assert d["config"]["exchange_ms"] > 0 and d["config"]["exchange_bytes_per_rank"] > 3 * 144 * 5000      # the records of ALL timed steps
    assert abs(d["value"] - 2 * 4096 * 4096 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01


@needs_two
def test_bench_c4_two_gpus_over_rccl():
    d = bench(["--config", "c4", "--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["config"]["backend"] == "nccl"
    one = bench(["--config", "c4", "--steps", "1", "--warmup", "1"])
    assert abs(one["config"]["keypoints_per_image"] - d["config"]["keypoints_per_image"]) < 1e-6
