"""bench.py's launch contract: `python bench.py --gpus N` launches its own N ranks (the driver calls it both ways), the N > 1
branch (RCCL communicator, gather into rank 0's landing buffer, its JSON keys) is rehearsed at world size 1 on a real GPU, and
every line carries the parity block."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_spawns_two_ranks_and_fails_clearly_without_gpus():
    """On a box with fewer than 2 GPUs the self-launched ranks must stop with a message that says what is missing
    (not a SystemExit telling the caller to use another launcher, not a hang)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has 2+ GPUs: the launch would succeed")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], 300)
    assert r.returncode != 0
    err = r.stdout + r.stderr
    assert "GPU(s)" in err and "one process per GPU" in err, err[-2000:]


@pytest.mark.gpu
def test_force_dist_rehearses_the_multi_gpu_branch_at_world_size_1():
    r = _run(["--force-dist", "--steps", "3", "--warmup", "1"], 900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["forced_dist_rehearsal"] is True
    assert "configs[3]" in j["config"]["workload"] and j["config"]["index_vectors"] == 100000
    assert j["gather_ms"] is not None and j["gather_ms"] > 0 and j["gather_bytes_per_rank"] == 64 * 96000 * 4
    assert j["n1_same_workload_value"] > 0 and 0.5 < j["scaling_efficiency"] <= 1.25      # (two 3-step timings of the same work at world size 1: a sanity range, not a measurement)
    assert j["value"] > 3.2e6                       # north_star's 200x real time, by a wide margin
    p = j["parity"]
    assert p["knn_idx_equal"] and p["rms_vs_golden_max"] <= 1e-4 and p["ok"], p


@pytest.mark.gpu
def test_default_line_carries_parity_roofline_and_cpu_baseline_keys():
    r = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-stream"], 900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and "configs[1]" in j["config"]["workload"] and j["dtype"].startswith("f32")
    p = j["parity"]
    assert "timed" in p["checked"] and len(p["rms_vs_golden"]) == 4
    assert p["knn_idx_equal"] and p["knn_idx_mismatches"] == 0 and p["rms_vs_golden_max"] <= 1e-4 and p["ok"], p
    assert p["drawn_equals_injected_hash"] is True, "the timed instantiation (phases drawn in-kernel) must equal the injected restatement of its hash"
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and rf["launch_ms"] > 0
