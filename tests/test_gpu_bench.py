"""bench.py end to end on the GPU box: the one-GPU line carries the contract's fields, and `--gpus 2`
started without a launcher spawns its two ranks itself (on a one-GPU box: both on GPU 0 over gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_self_launched():
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["unit"] == "Mrays/s" and line["value"] > 0
    assert line["scaling"] == "weak" and line["roofline"]["frac"] > 0
    assert "tile_partition" in line and "error" not in line["tile_partition"], line.get("tile_partition")


def test_bench_one_gpu_line():
    line = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert line["n_gpus"] == 1 and line["config"]["triangles"] == 262144 and line["dtype"] == "f32"
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
