"""bench.py plumbing that needs no GPU: the self-launching parent refuses to start ranks it has no devices for, and the
single-thread worker of the all-core cpu_baseline leg runs the oracle chain."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from mDeepFRI import _hip

BENCH = os.path.join(ROOT, "bench.py")


def test_parent_fails_loudly_when_devices_are_missing():
    if _hip.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 requested but only" in r.stderr and r.stdout.strip() == ""


def test_cpu_worker_runs_the_oracle_chain():
    r = subprocess.run([sys.executable, BENCH, "--cpu-worker", "9000,96,0.5,0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n"] >= 2 and out["t"] > 0


def test_dry_plan_prints_the_deal_of_the_strong_workloads():
    """`bench.py --gpus 8 --dry-plan`: CPU only, one JSON line, predicted imbalance of the padded rows below 1 % for both
    BASELINE configs (the same lengths and seeds the timed run uses)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-plan"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])["dry_plan"]
    assert set(plan) == {"configs3", "configs4"}
    assert sum(plan["configs3"]["proteins"]) == 100_000 and sum(plan["configs4"]["proteins"]) == 500_000
    for p in plan.values():
        assert p["world"] == 8 and p["imbalance"] < 0.01 and p["plan_seconds"] < 30
