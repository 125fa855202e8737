"""The driver's contract for bench.py and __graft_entry__ on a GPU box: one JSON line with the agreed keys, roofline and
cpu_baseline objects; smoke() passes.  The N > 1 branch is rehearsed with two ranks sharing the GPU (RCCL refuses two
ranks on one device, so their collectives run over gloo): once through bench.py's own launcher (`--gpus 2`, no
torch.distributed.run), once under torch.distributed.run as the driver starts it, and once for the C4 workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, timeout=600):
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line, got %d" % len(lines)
    return json.loads(lines[0])


def check_common(d, n_gpus, steps, warmup, scaling="weak"):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "Mpix/s" and d["higher_is_better"] is True and d["scaling"] == scaling and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_pipeline", "achieved_pipeline"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    # SURVEY 8(d)'s whole-call figure under the same key: bytes_alg / kernel time, below the dominant kernel's own fraction
    assert 0.02 < r["frac_pipeline"] < r["frac"]
    if n_gpus > 1:
        m = d["ms_per_step_ranks"]
        assert len(m["per_rank"]) == n_gpus and m["min"] <= m["max"] and abs(m["max"] - d["ms_per_step"]) / d["ms_per_step"] < 0.02


def test_bench_single_gpu_line():
    d = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1"])
    check_common(d, 1, 3, 1)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # value is throughput of K steps of one 4096 x 4096 image: consistent with ms_per_step
    assert abs(d["value"] - 4096 * 4096 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    assert "blur_team_kernel" in d["roofline"]["kernel"] and "cpu" in c
    assert abs(d["roofline"]["frac_pipeline"] - d["roofline_pipeline"]["frac"]) < 1e-9 and "frac_rocprof" in d["roofline"]
    for key in ("pipelined", "host_to_host", "match_100k"):
        assert key in d and "error" not in d[key], (key, d.get(key))
    assert d["match_100k"]["pairs"] == d["match_100k"]["expected_pairs"]


def free_port():
    """a port nobody listens on right now (a fixed one collides with a lingering worker of an earlier run)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_two_ranks_rehearsal():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--share-gpu"])
    check_common(d, 2, 3, 1)
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only
    assert abs(d["value"] - 2 * 4096 * 4096 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` starts the two ranks itself (no torch.distributed.run) and reports the world size the
    process group observed."""
    d = run([sys.executable, "bench.py", "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1"])
    check_common(d, 2, 2, 1)
    assert d["config"]["world_size_observed"] == 2 and d["config"]["backend"] == "gloo" and "all_gather" in d["config"]["exchange"]


def test_bench_c4_sharded_batch():
    """BASELINE.json configs[3]: 64 x 2048^2 frames sharded over the ranks + all-gather of every frame's records."""
    d = run([sys.executable, "bench.py", "--config", "c4", "--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "1"], timeout=900)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["name"] == "c4" and d["config"]["images_per_step"] == 64
    assert 1500 < d["config"]["keypoints_per_image"] < 4000          # ~2.7 k per 2048^2 white-noise frame
    assert abs(d["value"] - 64 * 2048 * 2048 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    assert 0.0 < d["roofline"]["frac_pipeline"] < 1.0 and d["roofline"]["bound"] == "hbm" and len(d["ms_per_step_ranks"]["per_rank"]) == 2
    one = run([sys.executable, "bench.py", "--config", "c4", "--steps", "1", "--warmup", "1"], timeout=900)
    assert one["n_gpus"] == 1 and abs(one["config"]["keypoints_per_image"] - d["config"]["keypoints_per_image"]) < 1e-6


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout
