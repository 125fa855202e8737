"""The driver's contract for bench.py and __graft_entry__ on a GPU box: one JSON line with the agreed keys, roofline and
cpu_baseline objects; smoke() passes.  (The N > 1 branch is rehearsed with two gloo ranks sharing the GPU.)"""
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


def check_common(d, n_gpus, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "Mpix/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0


def test_bench_single_gpu_line():
    d = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1"])
    check_common(d, 1, 3, 1)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # value is throughput of K steps of one 4096 x 4096 image: consistent with ms_per_step
    assert abs(d["value"] - 4096 * 4096 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01


def test_bench_two_ranks_rehearsal():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--share-gpu"])
    check_common(d, 2, 3, 1)
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only
    assert abs(d["value"] - 2 * 4096 * 4096 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout
