"""bench.py --gpus N: the entry point the driver uses for the 1/2/4/8-GPU series.

* without a launcher environment it starts the N ranks itself (one process per GPU) and prints ONE JSON line with
  ``n_gpus`` = ranks that took part, the weak C2 headline, and BASELINE configs[4] (the CG2 cube split over the ranks,
  strong scaling) with exchange / overlap figures;
* with fewer than N visible devices it FAILS (it used to print ``n_gpus: 1``);
* on a one-GPU box the same entry point is rehearsed with all ranks on device 0 and the host-bounce wire
  (FDHIP_FORCE_DEVICE=0 FDHIP_DIST_BACKEND=gloo): everything but the RCCL transport is the production path.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=1500):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FDHIP_FORCE_DEVICE", "FDHIP_DIST_BACKEND"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_world_size_and_gpus_must_agree():
    r = _run(["--gpus", "4", "--steps", "1"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def test_single_gpu_workloads_refuse_gpus_n():
    r = _run(["--gpus", "2", "--workload", "c3"])
    assert r.returncode != 0 and "single-GPU configuration" in r.stderr


@pytest.mark.gpu
def test_gpus_n_fails_loudly_without_n_devices():
    import ctypes
    from firedrake_amd import _lib
    ndev = ctypes.c_int()
    _lib.call("fd_device_count", ctypes.byref(ndev))
    want = ndev.value + 1
    r = _run(["--gpus", str(want), "--steps", "1", "--warmup", "1", "--n", "8", "--cpu-sample", "0"])
    assert r.returncode != 0
    assert f"--gpus {want} but only {ndev.value} HIP device" in r.stderr
    assert "n_gpus" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,partition", [(2, "slabs"), (4, "blocks")])
def test_self_launched_ranks_rehearsal_on_one_device(world, partition):
    r = _run(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--n", "16", "--n5", "12", "--tile", "4,4,4",
              "--partition", partition, "--cpu-sample", "0", "--traffic", "off"],
             env={"FDHIP_FORCE_DEVICE": "0", "FDHIP_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == "weak"
    assert out["multi_gpu"]["wire"] == "host" and len(out["multi_gpu"]["per_rank"]) == world
    assert out["exchange_ms"] > 0 and out["multi_gpu"]["residual_kernel_only_ms"] > 0
    pg = out["config"]["workload"].split("partition ")[1]
    assert pg == {"slabs": f"1x1x{world}", "blocks": "1x2x2"}[partition]
    assert out["config"]["dofs_global"] == {"slabs": 17 * 17 * (16 * world + 1), "blocks": 17 * 33 * 33}[partition]
    c5 = out["strong_c5"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == world and c5["config"]["dofs_global"] == 25 ** 3
    assert c5["value"] > 0 and c5["exchange_ms"] > 0


@pytest.mark.gpu
def test_c5_workload_is_the_cube_split_over_the_ranks():
    r = _run(["--gpus", "2", "--workload", "c5", "--steps", "2", "--warmup", "1", "--n", "10", "--tile", "4,4,4",
              "--cpu-sample", "0"], env={"FDHIP_FORCE_DEVICE": "0", "FDHIP_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["dofs_global"] == 21 ** 3
    assert "UnitCubeMesh(10,10,10)" in out["config"]["workload"] and "CG2" in out["config"]["workload"]
