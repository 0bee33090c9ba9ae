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


def test_trace_digests_on_a_synthetic_kernel_trace(tmp_path):
    """tools/kernel_overlap.py and tools/kernel_gaps.py read a rocprofv3 --kernel-trace directory (the numbers behind DESIGN.md section 5 rows
    14 and 22): a hand-made trace with two queues -- kernel a 0-10 and 20-30 us on queue 1, kernel b 5-25 us on queue 2 -- must come out as
    sums 20 / 20 us, union of both 30 us, and on queue 1 one gap of 10 us in front of the second a."""
    import subprocess
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    cols = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp"]
    rows = [("KERNEL_DISPATCH", "Agent 2", 1, 0, 1, 1, 7, "void mdf::k_a<6>(float*)", 1, 0, 10000),
            ("KERNEL_DISPATCH", "Agent 2", 2, 0, 1, 2, 8, "void mdf::k_b<7>(float*)", 2, 5000, 25000),
            ("KERNEL_DISPATCH", "Agent 2", 1, 0, 1, 3, 7, "void mdf::k_a<6>(float*)", 3, 20000, 30000)]
    with open(d / "1_kernel_trace.csv", "w") as f:
        f.write(",".join(f'"{c}"' for c in cols) + "\n")
        for r in rows:
            f.write(",".join(f'"{x}"' if isinstance(x, str) else str(x) for x in r) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_overlap.py"), str(tmp_path / "prof"), "void mdf::k_a", "void mdf::k_b"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert "launches      2" in lines[0] and "sum      0.02 ms" in lines[0] and "union      0.02 ms" in lines[0], lines
    assert "launches      1" in lines[1] and "union      0.03 ms" in lines[2], lines
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_gaps.py"), str(tmp_path / "prof"), "2"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "queue 1: 2 kernels" in out.stdout and "mean   10.00 us" in out.stdout and "before k_a<6>" in out.stdout, out.stdout
