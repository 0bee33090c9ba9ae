"""bench.py on the GPU box: the driver's command line shapes, including the self-launched multi-rank modes (two ranks on the one
GPU of the test box, gloo plumbing; the RCCL path needs a multi-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_default_shape_small():
    line = _run("--proteins", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--cpu-workers", "2", "--end-to-end", "1", "--query-stream", "2")
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["unit"] == "proteins/s" and line["value"] > 0
    assert line["verify"]["max_abs_err_vs_oracle"] < 1e-4
    assert line["roofline"]["bound"] == "mfma" and line["roofline_ax"]["bound"] == "hbm"
    assert set(line["by_length"]) == {"256", "1024"} and line["mixed"]["value"] > 0 and line["end_to_end"]["value"] > 0
    assert line["gcn_only"]["value"] > 0 and line["gcn_only"]["max_abs_diff_vs_fused_path"] < 1e-5
    assert line["gcn_only"].get("maps_in_hbm", {"value": float("inf")})["value"] > line["value"]      # present whenever a contact-stage launch was sampled
    assert line["query_stream"]["value"] > 0 and line["query_stream"]["queries"] == 8000 and line["query_stream"]["result_lines"] > 8000
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["process_pool"]["cores"] == 2 and "available" in cb["onnxruntime"]
    assert "cmap_stage_reference" in cb and cb["cmap_stage_reference"]["note"].startswith("constant")


def test_bare_python_launches_two_ranks_weak():
    line = _run("--gpus", "2", "--backend", "gloo", "--force-device", "0", "--proteins", "200", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["proteins_total"] == 400
    assert line["verify"]["max_abs_err_vs_oracle"] < 1e-4


@pytest.mark.parametrize("workload,count", [("configs3", 600), ("configs4", 900)])
def test_bare_python_launches_two_ranks_strong_filtered(workload, count):
    line = _run("--gpus", "2", "--backend", "gloo", "--force-device", "0", "--workload", workload, "--proteins", str(count), "--steps", "1",
                "--warmup", "1", "--cpu-seconds", "0", "--chunk-rows", "16384")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["proteins_total"] == count
    assert line["verify"]["max_abs_err_vs_oracle"] < 1e-4 and line["verify"]["gather_restores_input_order"] is True
    assert all(v > 0 for v in line["gathered_survivors"].values())


@pytest.mark.parametrize("workload,count,floor", [("configs3", 100_000, 15_000), ("configs4", 500_000, 25_000)])
def test_strong_workloads_at_full_size_one_gpu(workload, count, floor):
    """BASELINE.json configs[3] (100 000 mixed-length proteins) and configs[4] (500 000, proteome length histogram) at their STATED
    size on one GPU: contact-map alignment + three GO heads + GPU filter + the filtered gather plan (degenerate at N = 1), one
    timed step.  Heads at the operating point of trained ones (sparse_scores): the survivors are a few per cent of the terms,
    so the gathered payload is what the workload advertises -- far below the dense one.  The floors are ~1/3 of the rates
    measured in round 3 (profiles/r03_bench_configs{3,4}_n1.json)."""
    line = _run("--workload", workload, "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--verify", "6", timeout=1500)
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["proteins_total"] == count
    assert line["metric"] == f"proteins/sec (GCN+cmap), {workload}"
    assert line["verify"]["max_abs_err_vs_oracle"] < 1e-4 and line["verify"]["gather_restores_input_order"] is True
    assert line["value"] > floor, line["value"]
    for m, frac in line["survivor_fraction"].items():
        assert 0.002 < frac < 0.08, (m, frac)                 # compacted: a few per cent of the terms, not 90 %
    g = line["gather_bytes_per_step"]
    assert g["filtered"] < 0.25 * g["dense_equivalent"], g
    assert line["ranks"]["world_size"] == 1 and "cuda:0" in line["ranks"]["devices"][0]


def test_too_many_ranks_for_the_devices_is_refused():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "requested but only" in r.stderr
