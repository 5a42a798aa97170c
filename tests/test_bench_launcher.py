"""`python bench.py --gpus N` must start its own ranks (the driver runs it without a launcher):
N fresh child processes, spawned before the parent touches the GPU; rank 0's JSON line relayed;
non-zero exit when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
       "--batch", "4", "--steps", "8", "--warmup", "0", "--no-cpu-baseline", "--no-prof"]


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_launcher_spawns_ranks_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-container check")
    r = subprocess.run(CMD, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") == 2  # both ranks started, both refused a CPU run
    assert "ranks failed" in r.stderr


@pytest.mark.gpu
def test_launcher_two_ranks_on_one_gpu_prints_one_json_line():
    r = subprocess.run(CMD, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["config"]["global_batch"] == 8
    assert out["value"] > 0 and out["scaling"] == "weak"
