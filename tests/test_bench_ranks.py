"""CPU checks of what `bench.py --gpus N` hands each rank it starts (no GPU, no process is spawned): the driver will run
that command on an 8-GPU node some day without anybody watching -- the environment of the ranks must be right by
construction.  Partitioning the path keeps: image i -> rank i mod N; counts, then padded records."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("n", [2, 4, 8])
def test_rank_envs(bench, n):
    base = {"PATH": "/usr/bin", "RANK": "7", "WORLD_SIZE": "99", "SOMETHING": "kept"}
    envs = bench.rank_envs(n, 29511, base)
    assert len(envs) == n
    assert [e["RANK"] for e in envs] == [str(r) for r in range(n)]
    assert [e["LOCAL_RANK"] for e in envs] == [str(r) for r in range(n)]          # one node: rank r drives GPU r
    for e in envs:
        assert e["WORLD_SIZE"] == str(n) and e["LOCAL_WORLD_SIZE"] == str(n)
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"    # one rendezvous for all of them
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                             # dmabuf IPC, or RCCL cannot share buffers
        assert e["OMP_NUM_THREADS"] == "1"
        assert e["SOMETHING"] == "kept" and e["PATH"] == "/usr/bin"
    assert base["RANK"] == "7"                                                    # the caller's mapping is not touched
    # the caller's own choices win where they are choices
    envs = bench.rank_envs(n, 1, dict(base, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="1"))
    assert all(e["OMP_NUM_THREADS"] == "4" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" for e in envs)


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_every_frame_has_exactly_one_owner(n):
    from sift_pyocl_amd.batch import shard_indices
    for frames in (64, 7, 1, 0):
        owned = [shard_indices(frames, r, n) for r in range(n)]
        assert sorted(i for o in owned for i in o) == list(range(frames))
        for r, o in enumerate(owned):
            assert all(i % n == r for i in o)                                     # image i -> rank i mod N


def test_default_flags_are_the_driver_contract(bench, monkeypatch):
    """`python bench.py` alone must mean N = 1 and a K / W that finish in minutes; --gpus / --steps / --warmup parse."""
    import argparse
    seen = {}
    real = argparse.ArgumentParser.parse_args

    def spy(self, *a, **k):
        ns = real(self, *a, **k)
        seen["ns"] = ns
        raise SystemExit(0)
    monkeypatch.setattr(argparse.ArgumentParser, "parse_args", spy)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["ns"].gpus == 1 and 1 <= seen["ns"].steps <= 100 and 0 <= seen["ns"].warmup <= 20 and seen["ns"].config == "c2"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit):
        bench.main()
    assert (seen["ns"].gpus, seen["ns"].steps, seen["ns"].warmup) == (8, 5, 2)
