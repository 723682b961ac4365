"""`python bench.py --gpus N` launches itself (one process per GPU under torch.distributed.run) and rank 0 prints ONE JSON line:
the form the driver's scaling run may use.  CPU: the launch path alone (`--dry-run`: rendezvous over gloo, a barrier, the MAX
all-reduce of the timing protocol).  GPU: the whole bench at world size 2 on the one leased device (gloo: RCCL refuses two ranks
on one GPU), end to end."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ, ISDF_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):      # as a user's shell: no launcher variables
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 only
    assert out.stdout.strip().splitlines()[-1] == lines[0]
    return json.loads(lines[0])


def test_bench_gpus_2_launches_itself_dry_run():
    j = _run(["--gpus", "2", "--steps", "7", "--warmup", "3", "--dry-run"], 180)
    assert j == {"dry_run": True, "n_gpus": 2, "steps": 7, "warmup": 3, "backend": "gloo"}


def test_bench_gpus_1_dry_run_needs_no_launcher():
    j = _run(["--gpus", "1", "--dry-run"], 60)
    assert j["n_gpus"] == 1 and j["dry_run"] is True


@pytest.mark.gpu
def test_bench_gpus_2_end_to_end_on_one_device():
    j = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--ramp-seconds", "0.05", "--no-cpu-baseline", "--no-accuracy"], 600)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["distributed"]["world_size"] == 2 and j["distributed"]["collectives_per_step"] == 1
    assert j["value"] > 0 and abs(j["value"] - 2 * 1e3 / j["ms_per_step"]) < 1e-2 * j["value"]      # whole-job rate: 2 batches per step
    assert j["roofline"]["frac"] > 0 and len(j["distributed"]["per_rank_chain_us"]) == 2
