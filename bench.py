#!/usr/bin/env python3
"""bench.py -- proteins/sec of the fused hot path (C-alpha coords + alignment + sequence -> GO scores) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 works both ways: under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as a bare
  `python bench.py --gpus N` -- then this process, which never touches the GPU, starts N fresh rank processes itself and
  fails loudly if fewer than N devices are visible.
One JSON line on rank 0.  A "step" is one pass of the hot path over the whole workload, inputs resident in HBM before the
timed region:

  --workload configs2  (default; BASELINE.json configs[2], the configuration the metric is quoted on) 10 000 synthetic L=512
                       proteins, MF+BP+CC heads, PER GPU (weak scaling); N > 1 ends every step with the RCCL gather of each rank's
                       (10 000, 2752) score block to rank 0 (DenseGatherPlan: one collective, no host work in the step)
  --workload configs3  BASELINE.json configs[3]: 100 000 proteins, L ~ U{128..1024} (seed 46), 5 % indels, dealt to the N ranks by
                       cost (STRONG scaling), every rank filters its scores on the GPU (score >= 0.1, results.tsv order) and ONE
                       gather per head moves only the survivors (sharding.FilteredGatherPlan)
  --workload configs4  BASELINE.json configs[4]: 500 000 proteins, lengths from the committed histogram of the reference's test
                       proteome (clipped to [30, 2048]), otherwise as configs3
  --workload mixed     configs[3]-shaped, 10 000 proteins per GPU (weak scaling)
  --workload cnn       the sequence-only CNN models (extra measurement)

Objects on the line (tier contract):
  roofline      dominant kernel (the H.W GEMM; BF16x6 on the bf16 matrix pipe, see csrc/gcn.hip): algorithmic fp32 flops per launch / mean launch duration measured with HIP events
                on the launch stream inside the timed region (mdf_timing_* hooks of the library)
  roofline_ax   the same for the A.X aggregation kernel against the HBM roofline (the north_star's named kernel)
  cpu_baseline  the oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded sample: one thread
                (the configuration the reference ships) and a pool of single-thread worker processes (the reference's own
                parallelism, pipeline.py:476-481); real onnxruntime-CPU on the exported synthetic weights when ORT is importable
  by_length / mixed / helix / f32_pipe / f16x3_pipe / gcn_only / end_to_end / query_stream   (default N = 1 run only) mini-runs at L = 256 and L = 1024, the configs[3]-shaped mix,
  protein-like helix-bundle traces (fewer contacts per residue than a random walk), the GCN alone on given contact maps, the
                PCIe-inclusive host-lists-in / host-arrays-out rate, and the stages either side of the path as one stream
                (sequences + candidate sets in -> aligner -> path -> filter -> results.tsv text out) -- never `value`
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)   # mdfri_testkit: synthetic workloads + weights (not part of the product package)

import numpy as np  # noqa: E402
from mDeepFRI._hip import DEFAULT_CHUNK_ROWS  # noqa: E402   (a constant of the module: importing it loads neither the library nor the HIP runtime)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s, no sparsity)
BF16X6_PRODUCTS = 6        # bf16 term products per fp32 product in k_gemm_bf16x6 (csrc/gcn.hip)
F16X3_PRODUCTS = 3         # fp16 term products per fp32 product in k_gemm_f16x3 (opt-in pipe, MDFRI_HW_PIPE=f16x3)
MODES = ("mf", "bp", "cc")
STRONG = {"configs3": (100000, 46), "configs4": (500000, 47)}   # workload -> (proteins, seed = 42 + 1-based config number)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--proteins", type=int, default=None,
                    help="proteins per GPU per step for the weak workloads (default 10000 = configs[2]); TOTAL proteins for configs3/configs4 "
                         "(default 100000 / 500000)")
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--chunk-rows", type=int, default=DEFAULT_CHUNK_ROWS,
                    help="residue rows per fused chunk (multiples of 32768 = full rounds of 256x256 GEMM tiles on 256 CUs); default = the library's MDF_DEFAULT_CHUNK_ROWS")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each cpu_baseline leg (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="single-thread worker processes of the all-core leg (0 = min(cores, 32))")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--timing-period", type=int, default=7,
                    help="bracket every n-th launch of each kernel class with HIP events (an event pair costs GPU time between "
                         "kernels: timing every launch lowers the step rate by ~6 %%).  Every GraphConv layer is a class of its own "
                         "(ax2 / ax3, gemm2 / gemm3) and the period is prime, so that a sample cannot lock onto one phase of the "
                         "engine's launch cycle (3 heads x 2 layers per chunk)")
    ap.add_argument("--workload", default="configs2", choices=["configs2", "configs3", "configs4", "mixed", "cnn"])
    ap.add_argument("--verify", type=int, default=4,
                    help="after the timed region, check this many proteins of the step against the oracle (untimed; 0 = skip)")
    ap.add_argument("--lm", action="store_true",
                    help="give every GO head the language-model branch of the released models (shared 2x512 LSTM + per-head "
                         "LM embedding; SURVEY.md section 8f row 1).  Not the BASELINE.json configuration: an extra measurement.")
    ap.add_argument("--lm-batch", type=int, default=0, help="--lm: proteins per LSTM group (0 = the engine's default, 16384: 10 000 proteins run as one group)")
    ap.add_argument("--no-board", action="store_true", help="do not sample board power / shader clock with rocm-smi while the steps run")
    ap.add_argument("--no-extras", action="store_true", help="N = 1 default run: skip the by_length / mixed / end_to_end mini-runs")
    ap.add_argument("--end-to-end", type=int, default=8, metavar="N",
                    help="batches of the host-to-host mini-runs end_to_end / host_pipeline / binding (0 = skip); their `steady` is the rate once the pipeline is full")
    ap.add_argument("--query-stream", type=int, default=6, metavar="N", help="batches of 4 000 queries of the query_stream mini-run (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--force-device", type=int, default=None, help="testing aid: put every rank on this device ordinal")
    ap.add_argument("--dry-plan", action="store_true",
                    help="CPU only: print, as one JSON line, how configs3 / configs4 would be dealt to --gpus ranks (proteins, padded rows, "
                         "chunks and predicted imbalance per rank) and exit")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)   # internal: one single-thread worker of the all-core leg
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: a parent that has made no GPU call starts the N rank processes (never a re-exec of a process that touched HIP)
# ---------------------------------------------------------------------------------------------------------------------
def visible_devices() -> int:
    """HIP devices visible to a rank, counted in a THROW-AWAY child process: the parent that starts the ranks must never open
    the GPU driver itself (on some ROCm wheels torch.cuda.device_count() falls through to hipGetDeviceCount), and the child
    sees exactly the HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES filtering the ranks will see."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_ranks(args) -> int:
    visible = args.gpus if args.force_device is not None else visible_devices()
    if args.force_device is None and visible < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested but only {visible} HIP device(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, deadline = [], 0.0
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:       # a dead rank would leave the others waiting in a collective
                        q.terminate()
                    deadline = time.time() + 20.0   # ... and one that sits in a collective may not see SIGTERM: SIGKILL after a grace period
            if rc != 0 and procs and time.time() > deadline:
                for q in procs:
                    q.kill()
            time.sleep(0.05)
    finally:
        for p in procs:
            p.kill()
        for p in procs:
            p.wait()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def make_fixed_length(seed, count, L):
    """configs[2] inputs: uniform 20-letter sequences, 3.8 A random-walk C-alpha traces rounded to 3 decimals, identity
    alignments.  Vectorised over the batch."""
    from mdfri_testkit import synthetic
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(synthetic.AA20.encode(), dtype=np.uint8)
    seq_bytes = letters[rng.integers(0, 20, size=(count, L))]
    seqs = [bytes(r).decode() for r in seq_bytes]
    v = rng.standard_normal((count, L, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True) + 1e-12
    xyz = np.round(np.cumsum(v * 3.8, axis=1), 3).astype(np.float32)
    return seqs, [xyz[i] for i in range(count)], seqs, seqs


def make_helix(seed, count, L):
    """Protein-like C-alpha traces (synthetic.helix_bundle_coords: ~8.6 contacts per residue at 6 A instead of the ~12.6 of a random
    walk; SURVEY.md section 8d asks for this second generator), otherwise as make_fixed_length."""
    from mdfri_testkit import synthetic
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(synthetic.AA20.encode(), dtype=np.uint8)
    seqs = [bytes(r).decode() for r in letters[rng.integers(0, 20, size=(count, L))]]
    return seqs, [synthetic.helix_bundle_coords(rng, L) for _ in range(count)], seqs, seqs


def make_mixed(seed, count):
    """configs[3]-shaped: L ~ U{128..1024}, 5 % indels, sorted by length as pipeline.py:529 sorts its work list."""
    from mdfri_testkit import synthetic
    L = synthetic.uniform_lengths(seed, count)
    order = np.argsort(L, kind="stable")
    return synthetic.bulk_proteins(seed, L, order, indel_rate=0.05)


def make_queries(seed, n_queries, n_db=1500, k=8):
    """Inputs of the stages in front of and behind the path (query_stream leg): a database of histogram-length sequences with random-walk
    structures, queries = mutated database members (ungapped, as a FASTA record holds them), k candidate targets each."""
    from mdfri_testkit import synthetic
    rng = np.random.default_rng(seed)
    lens = synthetic.histogram_lengths(seed + 1, n_db)
    db_seq = {f"T{j}": synthetic.random_sequence(rng, int(L)) for j, L in enumerate(lens)}
    db_xyz = {name: synthetic.random_walk_coords(rng, len(s)) for name, s in db_seq.items()}
    names = list(db_seq)
    qids, qseqs, cands = [], [], []
    for i in range(n_queries):
        home = names[int(rng.integers(0, n_db))]
        q, _, _ = synthetic.mutate_alignment(rng, db_seq[home], 0.04)
        qids.append(f"Q{i}")
        qseqs.append(q.replace("-", "") or "A")
        cands.append({n: db_seq[n] for n in [home] + [names[int(j)] for j in rng.integers(0, n_db, size=k - 1)]})
    return qids, qseqs, cands, db_xyz


def oracle_paths():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)


# ---------------------------------------------------------------------------------------------------------------------
# cpu_baseline
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_loop(seqs, coords, weights, budget_s, lm):
    oracle_paths()
    import cmap_oracle
    import gcn_oracle
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s or n < 2:
        i = n % len(seqs)
        cm = cmap_oracle.build_align_contact_map(coords[i], seqs[i], seqs[i], 6.0, 2)
        for m in MODES:
            if lm:
                import lm_oracle
                lm_oracle.gcn_lm_forward(weights[m], seqs[i], cm)   # (the oracle re-runs the shared LSTM per head, as the reference's three ONNX sessions do)
            else:
                gcn_oracle.gcn_forward(weights[m], seqs[i], cm)
        n += 1
    return n, time.perf_counter() - t0


def cpu_worker(spec):
    """One single-thread worker of the all-core leg (a fresh child process: BLAS pinned to one thread by its environment)."""
    seed, length, budget, lm = spec.split(",")
    from mdfri_testkit import synthetic
    weights = make_weights(bool(int(lm)))
    seqs, coords, _, _ = make_fixed_length(int(seed), 8, int(length))
    del synthetic
    n, t = _oracle_loop(seqs, coords, weights, float(budget), bool(int(lm)))
    print(json.dumps({"n": n, "t": t}), flush=True)


def make_weights(lm, sparse_scores=False):
    """sparse_scores: the operating point of trained heads (1-4 % of the terms pass score >= 0.1; synthetic.glorot_gcn_weights) --
    used by the workloads whose step ends in the FILTERED gather, so that they measure a compacted payload."""
    from mdfri_testkit import synthetic
    weights = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m], sparse_scores=sparse_scores) for i, m in enumerate(MODES)}
    if lm:
        lmw = synthetic.glorot_lm_weights(seed=1000)
        for i, m in enumerate(MODES):
            weights[m].update(lmw)
            weights[m]["W_lm"] = synthetic.glorot_uniform(np.random.default_rng(2000 + i), 512, 1024)  # LM_embedding is per head
    return weights


def ort_leg(seqs, coords, weights, budget_s):
    """Real onnxruntime-CPU, single thread, batch 1 -- the reference's shipped configuration (predict.pyx:62-73,98; SURVEY.md
    section 0.6) -- on the synthetic weights exported by mdfri_testkit.onnx_writer.  Only when `import onnxruntime` works here."""
    try:
        import onnxruntime as rt
    except Exception as e:
        return {"available": False, "probe": f"import onnxruntime: {type(e).__name__}"}
    try:
        oracle_paths()
        import cmap_oracle
        import gcn_oracle
        from mdfri_testkit import onnx_writer
        so = rt.SessionOptions()
        so.intra_op_num_threads = so.inter_op_num_threads = 1
        sess = {m: rt.InferenceSession(onnx_writer.deepfri_gcn_model(weights[m]), so, providers=["CPUExecutionProvider"]) for m in MODES}
        names = {m: [i.name for i in s.get_inputs()] for m, s in sess.items()}
        n, worst, t0 = 0, 0.0, time.perf_counter()
        while n < len(seqs) and (time.perf_counter() - t0 < budget_s or n < 2):
            L = len(seqs[n])
            cm = cmap_oracle.build_align_contact_map(coords[n], seqs[n], seqs[n], 6.0, 2)
            A = cm.reshape(1, L, L).astype(np.float32)
            S = cmap_oracle.seq2onehot(seqs[n]).reshape(1, L, 26)
            for m in MODES:
                y = sess[m].run(None, {names[m][0]: A, names[m][1]: S})[0][:, :, 0].reshape(-1)
                if n < 2:   # external check of the oracle (untimed cost is small): ORT vs oracle/gcn_oracle.py
                    worst = max(worst, float(np.max(np.abs(y - gcn_oracle.gcn_forward(weights[m], seqs[n], cm)))))
            n += 1
        dt = time.perf_counter() - t0
        return {"available": True, "version": rt.__version__, "value": n / dt, "unit": "proteins/s", "cores": 1, "kind": "reference-runtime",
                "sample": f"{n} proteins, contact map (C oracle) + 3 ONNX sessions each, intra_op=inter_op=1",
                "max_abs_diff_ort_vs_oracle": worst}
    except Exception as e:   # an ORT that cannot run the export is a finding, not a crash of the benchmark
        return {"available": True, "error": f"{type(e).__name__}: {str(e)[:300]}"}


def reference_cmap_leg(seqs, coords, budget_s):
    """The contact-map stage on the REAL reference kernels (oracle/_ref: contact_map_utils.pyx compiled by oracle/build_ref.py with
    the reference's flags), one thread, with the glue of bio_utils.py:196-227,348-385 restated in NumPy: pairwise_sqeuclidean ->
    D < thr^2 -> argwhere -> align_contact_map.  Falls back on the constant measured in the build container when the compiled
    reference did not travel to this box."""
    const = {"kind": "reference", "ms_per_protein": 1.39, "measured": "build container, round 1, 1 thread, L=512",
             "note": "constant: the compiled reference (oracle/_ref) is not on this box"}
    try:
        oracle_paths()
        import build_ref
        ref = build_ref.load()
        if ref is None:
            return const
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or n < 2:
            i = n % len(seqs)
            D = ref.pairwise_sqeuclidean(coords[i])
            sparse = np.argwhere((D < 6.0 ** 2).astype(np.int32) == 1).astype(np.int32)
            ref.align_contact_map(seqs[i], seqs[i], sparse, 2)
            n += 1
        dt = time.perf_counter() - t0
        return {"kind": "reference", "value": n / dt, "unit": "proteins/s", "cores": 1, "ms_per_protein": round(1e3 * dt / n, 3),
                "sample": f"{n} of the step's L={len(seqs[0])} proteins, {dt:.1f} s: pairwise_sqeuclidean + threshold + argwhere + align_contact_map "
                          "(the reference's own compiled contact_map_utils.pyx, one thread), contact-map stage only",
                "note": "measured live on this box's host"}
    except Exception as e:   # noqa: BLE001  -- a baseline leg never takes the headline line with it
        return dict(const, error=f"{type(e).__name__}: {e}"[:300])


def cpu_baseline(seqs, coords, weights, args):
    """Oracle chain (oracle/cmap_oracle.c + oracle/gcn_oracle.py) on this box's host cores."""
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        n1, t1 = _oracle_loop(seqs, coords, weights, args.cpu_seconds, args.lm)
    ncores = os.cpu_count() or 1
    workers = args.cpu_workers or min(ncores, 32)
    L = len(seqs[0])
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    spec = lambda k: f"{9000 + k},{L},{args.cpu_seconds * 0.6},{int(args.lm)}"  # noqa: E731
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", spec(k)], env=env, stdout=subprocess.PIPE, text=True)
             for k in range(workers)]
    rates, fails = [], 0
    for p in procs:
        out, _ = p.communicate()
        try:
            r = json.loads(out.strip().splitlines()[-1])
            rates.append(r["n"] / r["t"])
        except Exception:
            fails += 1
    pool = {"value": float(sum(rates)), "unit": "proteins/s", "cores": len(rates), "host_cores_visible": ncores,
            "sample": f"{len(rates)} single-thread worker processes (the reference's Pool of 1-thread workers, pipeline.py:476-481), "
                      f"{args.cpu_seconds * 0.6:.0f} s each, wall {time.perf_counter() - t0:.0f} s" + (f", {fails} workers failed" if fails else "")}
    return {"value": n1 / t1, "unit": "proteins/s", "cores": 1, "kind": "port",
            "sample": f"{n1} of the step's L={L} proteins, contact map + 3 GO heads each, {t1:.1f} s, numpy/BLAS pinned to 1 thread "
                      f"(= the configuration the reference ships: ORT intra_op=1, batch 1)",
            "process_pool": pool,
            "onnxruntime": ort_leg(seqs, coords, weights, args.cpu_seconds * 0.5),
            "cmap_stage_reference": reference_cmap_leg(seqs, coords, min(args.cpu_seconds * 0.25, 4.0)),
            "published_anchor": "reference weight_convert/inference_times.csv.gz: 0.13 s/protein/model/core at L~512 (ORT CPU)"}


# ---------------------------------------------------------------------------------------------------------------------
# timed runs
# ---------------------------------------------------------------------------------------------------------------------
class Ctx:
    pass


class BoardSampler:
    """Board power and shader clock of one GPU while the timed region runs, read by a thread of its own with `rocm-smi` (a process per
    sample, back to back, ~0.3 s each: nothing touches the measured stream).  The H.W GEMM sits on the board's power limit (profiles/r05_gemm_overlap_probe.txt),
    so the rate moves with the box: the line carries what THIS box held.  Missing tool / unparsable output: the fields are null."""

    def __init__(self, device_index: int = 0, off: bool = False):
        import shutil
        import threading
        self.off, self.started = off, False
        self.cmd = [shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi", "-d", str(device_index), "--showpower", "--showclocks", "--showmaxpower"]
        self.power, self.sclk, self.cap = [], [], None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                txt = subprocess.run(self.cmd, capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"(?:Current Socket|Average) Graphics Package Power \(W\):\s*([0-9.]+)", txt)
                if m:
                    self.power.append(float(m.group(1)))
                m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
                if m:
                    self.sclk.append(int(m.group(1)))
                m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", txt)
                if m:
                    self.cap = float(m.group(1))
            except Exception:
                return
            self._stop.wait(0.05)      # (a call takes ~0.25 s of its own)

    def __enter__(self):
        # not under a profiler: its preloaded library has initialised the GPU before Python starts, and the box refuses the exec of a child there
        if not self.off and not any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY")):
            self._thread.start()
            self.started = True
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.started:
            self._thread.join(timeout=10)

    def summary(self):
        def stat(v, nd):
            return {"mean": round(sum(v) / len(v), nd), "min": round(min(v), nd), "max": round(max(v), nd), "samples": len(v)} if v else None
        return {"board_power_w": stat(self.power, 1), "shader_clock_mhz": stat(self.sclk, 0), "power_cap_w": self.cap,
                "how": "rocm-smi --showpower --showclocks called back to back (~0.3 s each) by a separate thread while warmup + timed steps run"}


def timed_run(ctx, eng, db, steps, warmup, after=None, timing_period=7):
    """W warmup steps, then exactly K timed steps bracketed by barrier + synchronize; returns (seconds (max over ranks), last out,
    per-rank split).  The split comes from three events per step on the launch stream: forward issued -> forward done (`compute`)
    -> the step's gather done (`gather`; on a sending rank that includes waiting for the collective to be matched)."""
    import torch
    import torch.distributed as dist

    marks = []

    def step(timed=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        if timed:
            ev[0].record()
        out = eng.forward_alignments(db)
        if timed:
            ev[1].record()
        if after is not None:
            after(out)
        if timed:
            ev[2].record()
            marks.append(ev)
        return out

    def fence():
        if ctx.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = step()
    # fault injection for the launcher's test (tests/test_gpu_bench.py): this rank dies between warmup and the timed steps, with kernels of
    # its own in flight and the other ranks on their way into the barrier below -- the parent must take them all down and report the failure
    if os.environ.get("MDFRI_BENCH_FAIL_RANK") == str(ctx.rank):
        os._exit(17)
    fence()
    eng.check(db)  # invalid residues / CSR overflow would surface here
    ctx.lib.mdf_timing_reset()
    ctx.lib.mdf_timing_enable(max(1, timing_period) if timing_period > 0 else 0)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(timed=True)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.lib.mdf_timing_enable(0)
    mine = [sum(e[0].elapsed_time(e[1]) for e in marks) / steps, sum(e[1].elapsed_time(e[2]) for e in marks) / steps, 1e3 * elapsed / steps]
    split = {"compute_ms": [round(mine[0], 3)], "gather_ms": [round(mine[1], 3)], "step_ms": [round(mine[2], 3)]}
    if ctx.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.dev if ctx.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rows = [None] * ctx.world
        dist.all_gather_object(rows, mine)
        split = {"compute_ms": [round(r[0], 3) for r in rows], "gather_ms": [round(r[1], 3) for r in rows], "step_ms": [round(r[2], 3) for r in rows]}
    eng.check(db)
    return elapsed, out, split


def read_kernels(ctx, names):
    from mDeepFRI import _hip
    out = {}
    for k in names:
        n, ms = _hip.c_int64(0), _hip.ctypes.c_double(0.0)
        ctx.lib.mdf_timing_read(k.encode(), n, ms)
        out[k] = {"launches": int(n.value), "total_ms": round(ms.value, 3), "avg_us": round(1e3 * ms.value / max(n.value, 1), 2)}
    return out


def rooflines(ctx, eng, pk, kernels, lm):
    """roofline objects of the two named kernels from the library's sampled HIP-event timings of THIS run."""
    R = pk.chunks[0].rows                        # rows per launch (all chunks but the last are equal)
    rows_launch = sum(c.rows for c in pk.chunks) / len(pk.chunks)
    roof = roof_ax = None
    # HBM bytes per launch: a CONSTANT from the committed rocprofv3 PMC passes (profiles/traffic.json), reported only for the same rows
    # per launch AND the very kernels it was measured on: the file carries the library's mdf_version() (a hash of the GraphConv
    # kernels' source, csrc/Makefile); with any other build `traffic` is null and `traffic_source` says why
    src = "profiles/traffic.json (rocprofv3 --pmc pass of an earlier run of this command; not a counter of this run)"
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        lib_version = ctx.lib.mdf_version().decode()
        if traffic.get("rows_per_launch") != R:
            traffic, src = {}, f"dropped: profiles/traffic.json is for {traffic.get('rows_per_launch')} rows per launch, this run has {R}"
        elif traffic.get("library") != lib_version:
            traffic, src = {}, f"dropped: profiles/traffic.json was measured on '{traffic.get('library')}', this library is '{lib_version}' (re-run tools/refresh_profiles.sh)"
    except OSError:
        traffic, src = {}, "dropped: profiles/traffic.json not found"
    C = 512
    g, a = kernels.get("gemm", {}), kernels.get("ax", {})
    # `achieved` is the mean over ALL sampled launches of the kernel, every GraphConv layer pooled (the classes are sampled at the same
    # period and launched equally often); `per_layer` shows the layers on their own: layer 2 gathers rows the short K = 32 launch has
    # just written (Infinity-Cache resident), layer 3 rows written by a launch that streams 2 x 128 MiB
    if g.get("launches"):
        # with --lm the class also holds the unfolded layer-1 launch (K = 1024): mean over the three layers
        flops_launch = 2.0 * rows_launch * C * ((1024 + C + C) / 3.0 if lm else C)
        tf = flops_launch / (g["avg_us"] * 1e-6) / 1e12
        # the roofline of the pipe the kernel runs on (mdf_hw_pipe): BF16x6 executes six bf16 term products per fp32 product, so the
        # bf16 matrix peak / 6 bounds the ALGORITHMIC (fp32) flop rate of the method; the fp32 instruction's own peak is kept beside it
        pipe = ctx.lib.mdf_hw_pipe().decode()
        peak = MFMA_BF16_PEAK_TF / BF16X6_PRODUCTS if pipe == "bf16x6" else MFMA_BF16_PEAK_TF / F16X3_PRODUCTS if pipe == "f16x3" else MFMA_F32_PEAK_TF
        per_layer = {k: {"avg_us": kernels[k]["avg_us"], "timed_launches": kernels[k]["launches"],
                         "frac": round(2.0 * rows_launch * C * C / (kernels[k]["avg_us"] * 1e-6) / 1e12 / peak, 4)}
                     for k in ("gemm2", "gemm3") if kernels.get(k, {}).get("launches") and not lm}
        if pipe == "bf16x6":
            name = ("k_gemm_bf16x6 (H.W as BF16x6: fp32 operands split in registers into three bf16 terms, six term products per fp32 product on "
                    "v_mfma_f32_32x32x16_bf16, fp32 accumulate; 256x256x32 tiles, LDS-DMA staging, ELU+pool epilogue)")
        elif pipe == "f16x3":
            name = ("k_gemm_f16x3 (H.W as F16x3, opt-in: fp32 operands scaled by a power of two and split in registers into two fp16 terms, three term products "
                    "per fp32 product on v_mfma_f32_32x32x16_f16, fp32 accumulate; the split keeps 22 of 24 bits)")
        else:
            name = "k_gemm_f32 (H.W, 256x256x32 tiles, v_mfma_f32_32x32x2_f32, LDS-DMA staging, ELU+pool epilogue)"
        roof = {"kernel": name, "pipe": pipe,
                "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                "traffic": traffic.get("gemm_mean_bytes"), "traffic_source": src,
                "per_launch": {"rows": R, "flops": 2.0 * R * C * C, "avg_us": g["avg_us"], "timed_launches": g["launches"]},
                "per_layer": per_layer}
        if pipe == "bf16x6":
            roof["peak_note"] = (f"bf16 dense MFMA peak {MFMA_BF16_PEAK_TF:.0f} TFLOP/s / {BF16X6_PRODUCTS} term products per fp32 product; `achieved` counts the "
                                 "algorithmic fp32 flops (2.R.K.N), the matrix pipe executes six times that")
            roof["executed_bf16_tflops"] = round(tf * BF16X6_PRODUCTS, 1)
            roof["vs_f32_instruction_peak"] = round(tf / MFMA_F32_PEAK_TF, 4)
        if pipe == "f16x3":
            roof["peak_note"] = (f"fp16 dense MFMA peak {MFMA_BF16_PEAK_TF:.0f} TFLOP/s (the bf16 instruction's) / {F16X3_PRODUCTS} term products per fp32 product; "
                                 "`achieved` counts the algorithmic fp32 flops (2.R.K.N)")
            roof["executed_f16_tflops"] = round(tf * F16X3_PRODUCTS, 1)
            roof["vs_f32_instruction_peak"] = round(tf / MFMA_F32_PEAK_TF, 4)
    if a.get("launches"):
        # read Z (rows x 512 f32) once + write (rows x 512 f32) once per layer; CSR adjacency (4 B colidx
        # + 4 B val per entry + 4 B rowptr per row) added and stated
        nnz_per_row = float(eng.last_chunk_nnz()) / float(pk.chunks[-1].rows)
        bytes_row = 2 * 4 * C + 4 + 8 * nnz_per_row
        # with layer 1 made inside the layer-2 launch (mdf_layer1_form() == "fused") that launch is no longer an A.X alone -- it does not read
        # Z at all: the A.X roofline is then taken over the launches that ARE the named kernel (layer 3 and up), and the layer-2 launch is shown
        # beside the pair of launches it replaces
        fused_l1 = ctx.lib.mdf_layer1_form().decode() == "fused" and not lm and kernels.get("ax3", {}).get("launches")
        if fused_l1:
            a = kernels["ax3"]
        gbs = bytes_row * rows_launch / (a["avg_us"] * 1e-6) / 1e9
        per_layer = {k: {"avg_us": kernels[k]["avg_us"], "timed_launches": kernels[k]["launches"],
                         "frac": round(bytes_row * rows_launch / (kernels[k]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                     for k in ("ax2", "ax3") if kernels.get(k, {}).get("launches") and not lm}
        roof_ax = {"kernel": "k_aggregate_mfma (A.X as an exact block-sparse product on the bf16 matrix pipe: contact bits x the fp32 rows split into "
                             "three bf16 terms, fp32 accumulate; proteins outside its length classes: k_aggregate, CSR gather)",
                   "bound": "hbm", "achieved": round(gbs, 1),
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                   "traffic": (traffic.get("k_aggregate") or {}).get("bytes"), "traffic_source": src,
                   "per_launch": {"rows": R, "bytes": bytes_row * R, "nnz_per_row": round(nnz_per_row, 2), "avg_us": a["avg_us"],
                                  "timed_launches": a["launches"]},
                   "per_layer": per_layer}
        if fused_l1:
            # which proteins of THIS workload get their layer 1 inside the layer-2 launch: those the matrix pipe takes at one or two row blocks per
            # wave (mdf_agg_l1_fused: 80 .. 512 residues); the others' layer 1 is k_layer1 (`gemm1` class) in front of the plain form
            lq = np.asarray(pk.Lq)
            n_fused = int(sum(int(ctx.lib.mdf_agg_l1_fused(int(x))) != 0 for x in np.unique(lq) for _ in range(1)))
            fused_frac = float(np.mean([ctx.lib.mdf_agg_l1_fused(int(x)) != 0 for x in lq])) if len(lq) <= 200000 else None
            roof_ax["layer1_inside_layer2_launch"] = {"distinct_lengths_fused": n_fused, "distinct_lengths": int(len(np.unique(lq))),
                                                      "proteins_fused_fraction": None if fused_frac is None else round(fused_frac, 4)}
            roof_ax["layer1_form"] = "fused"
            roof_ax["note"] = ("`achieved` is over the layer-3 launches (the A.X kernel proper); the layer-2 launch (`per_layer.ax2`) also makes layer 1 "
                               "(H1 = elu(S.T1), never written) and replaces a k_layer1 launch + an A.X launch -- its `frac` against the A.X bytes is "
                               "kept for comparison only; MDFRI_L1_FUSE=0 runs the two-kernel form")
            per_layer["ax2"]["makes_layer1"] = bool(n_fused > 0)
    return roof, roof_ax


def verify(seqs, coords, q_alns, t_alns, weights, out, n, lm, index=None):
    """Parity spot check on the very outputs of the timed steps (oracle = checker, outside the timed region)."""
    oracle_paths()
    import cmap_oracle
    import gcn_oracle
    worst = 0.0
    for i in np.linspace(0, len(seqs) - 1, n).astype(int):
        cm = cmap_oracle.build_align_contact_map(coords[i], q_alns[i], t_alns[i], 6.0, 2)
        for m in MODES:
            if lm:
                import lm_oracle
                ref = lm_oracle.gcn_lm_forward(weights[m], seqs[i], cm, dtype=np.float64)
            else:
                ref = gcn_oracle.gcn_forward(weights[m], seqs[i], cm, dtype=np.float64)
            got = out[m][i if index is None else index[i]]
            worst = max(worst, float(np.max(np.abs(got.cpu().numpy().astype(np.float64) - ref))))
    return {"proteins": int(n), "heads": list(MODES), "max_abs_err_vs_oracle": worst, "tolerance": 1e-4, "oracle_dtype": "float64"}


def mini_run(ctx, eng, cols, chunk_rows, steps=2, warmup=1, lm=False):
    """A short untimed-contract run of another workload on the same engine: proteins/s + the two rooflines."""
    from mDeepFRI import batch
    pk = batch.PackedProteins.pack(*cols, max_rows=chunk_rows)
    db = eng.upload(pk)
    elapsed, _, _ = timed_run(ctx, eng, db, steps, warmup, timing_period=5)
    kernels = read_kernels(ctx, ("gemm", "gemm2", "gemm3", "ax", "ax2", "ax3"))
    roof, roof_ax = rooflines(ctx, eng, pk, kernels, lm)
    n = len(cols[0])
    return {"value": round(n * steps / elapsed, 1), "unit": "proteins/s", "proteins": n, "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3),
            "mean_length": round(float(np.mean([len(s) for s in cols[0]])), 1),
            "roofline": {k: roof[k] for k in ("achieved", "unit", "frac")} if roof else None,
            "roofline_ax": dict({k: roof_ax[k] for k in ("achieved", "unit", "frac")}, per_layer=roof_ax.get("per_layer")) if roof_ax else None}


def dry_plan(args):
    """`--dry-plan`: the deal of the strong-scaling workloads without a GPU -- what every rank would compute for itself."""
    from mDeepFRI import sharding
    from mdfri_testkit import synthetic
    out = {}
    for name in ([args.workload] if args.workload in STRONG else list(STRONG)):
        total, seed = STRONG[name]
        total = args.proteins or total
        lengths = synthetic.uniform_lengths(seed, total) if name == "configs3" else synthetic.histogram_lengths(seed, total)
        t0 = time.perf_counter()
        plan = sharding.plan_summary(lengths, args.gpus, args.chunk_rows)
        plan["plan_seconds"] = round(time.perf_counter() - t0, 3)
        plan["imbalance"] = round(plan["imbalance"], 6)
        out[name] = plan
    print(json.dumps({"dry_plan": out, "gpus": args.gpus, "chunk_rows": args.chunk_rows}), flush=True)


def main():
    args = parse()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.dry_plan:
        return dry_plan(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))

    cpu_leg = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_seconds > 0 and args.workload == "configs2":
        # the CPU legs run BEFORE this process makes its first GPU call: they start single-thread worker processes
        n_cpu = 256
        c_seqs, c_coords, _, _ = make_fixed_length(42 + 2, args.proteins or 10000, args.length)   # the first proteins of the step's workload
        cpu_leg = cpu_baseline(c_seqs[:n_cpu], c_coords[:n_cpu], make_weights(args.lm), args)
        del c_seqs, c_coords

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (set before the first HIP call: the host driver shares device memory between rank processes by dmabuf only)
    import torch
    import torch.distributed as dist

    ctx = Ctx()
    ctx.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.backend = args.backend
    if ctx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world}")
    if args.force_device is not None:
        local_rank = args.force_device
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {ctx.rank}: device {local_rank} requested but {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    ctx.dev = dev = torch.device(f"cuda:{local_rank}")
    if ctx.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=ctx.rank, world_size=ctx.world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=ctx.rank, world_size=ctx.world)
        ctx.world = dist.get_world_size()     # as the backend (RCCL) saw it

    from mDeepFRI import _hip, batch, sharding
    from mdfri_testkit import synthetic
    from mDeepFRI.output import filter_scores
    from mDeepFRI.predict import Predictor
    ctx.lib = _hip.lib()

    if args.workload == "cnn":
        return bench_cnn(args, ctx, local_rank)

    weights = make_weights(args.lm, sparse_scores=args.workload in STRONG)
    preds = {m: Predictor(f"synthetic-{m}", weights=weights[m], device=local_rank) for m in MODES}
    T_total = sum(p.n_terms for p in preds.values())
    eng = batch.HotPathEngine(preds, device=local_rank, max_rows=args.chunk_rows, **({"lm_batch": args.lm_batch} if args.lm_batch > 0 else {}))

    strong = args.workload in STRONG
    after, local_index = None, None
    if strong:
        total, seed = STRONG[args.workload]
        total = args.proteins or total
        lengths = synthetic.uniform_lengths(seed, total) if args.workload == "configs3" else synthetic.histogram_lengths(seed, total)
        mine = sharding.partition_by_cost(lengths, ctx.world)[ctx.rank]    # sorted by length inside the shard (pipeline.py:529)
        cols = synthetic.bulk_proteins(seed, lengths, mine, indel_rate=0.05, workers=min(32, max(1, (os.cpu_count() or 1) // ctx.world)))
        n_local, n_job = len(mine), total
        plans = {m: sharding.FilteredGatherPlan(mine, total, dev, dst=0) for m in MODES}
        gathered = {}

        def after(out):   # output stage on every rank, then ONE gather of the survivors per head
            for m in MODES:
                if n_local:
                    off, ti, kept = filter_scores(out[m], threshold=0.1, capacity_per_protein=preds[m].n_terms)
                else:      # a rank that owns nothing sends empty blocks
                    off, ti, kept = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.float32, device=dev)
                gathered[m] = plans[m].run(off, ti, kept, sizes_may_change=False)   # the same workload every step: sizes fixed by the first call
    else:
        n_local = args.proteins or 10000
        n_job = n_local * ctx.world
        if args.workload == "mixed":
            cols = make_mixed(42 + 4 + 1000 * ctx.rank, n_local)
        else:
            cols = make_fixed_length(42 + 2 + 1000 * ctx.rank, n_local, args.length)  # seed = 42 + config index (+rank)
        if ctx.world > 1:
            local_index = list(range(ctx.rank * n_local, (ctx.rank + 1) * n_local))
            plan = sharding.DenseGatherPlan(n_local, T_total, local_index, n_job, dev, dst=0)
            gathered = {}

            def after(out):
                gathered["all"] = plan.run(torch.cat([out[m] for m in MODES], dim=1))
    seqs, coords, q_alns, t_alns = cols
    if n_local == 0:
        # a rank that owns nothing (fewer proteins than ranks): it runs no kernel, and its gather plan sends empty blocks -- what
        # sharding.predict_sharded_filtered does for such a rank (tests/test_sharding_cpu.py at world size 8)
        class _NoWork:
            chunks, Lq = [], np.zeros(0, dtype=np.int32)

        class _Idle:
            modes = eng.modes

            def forward_alignments(self, db):
                return {m: torch.zeros((0, preds[m].n_terms), dtype=torch.float32, device=dev) for m in MODES}

            def check(self, db):
                return None

            def last_chunk_nnz(self):
                return 0
        pk, db, eng = _NoWork(), None, _Idle()
    else:
        pk = batch.PackedProteins.pack(seqs, coords, q_alns, t_alns, max_rows=args.chunk_rows)
        db = eng.upload(pk)

    # who is in the job: device name per rank and the world size as the backend reports it
    me = f"rank {ctx.rank}: {torch.cuda.get_device_name(dev)} (cuda:{local_rank})"
    if ctx.world > 1:
        names = [None] * ctx.world
        dist.all_gather_object(names, me)
        rank_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices": names}
    else:
        rank_info = {"world_size": 1, "backend": None, "devices": [me]}

    timing_period = 0 if args.no_kernel_timing else args.timing_period
    with BoardSampler(ctx.dev.index or 0, off=args.no_board or ctx.rank != 0) as board:   # (rank 0 only: one sampler per job)
        elapsed, out, split = timed_run(ctx, eng, db, args.steps, args.warmup, after=after, timing_period=timing_period)

    if strong:
        for m in MODES:
            plans[m].check()      # a rank that outgrew the planned payload in the last step is raised on every rank here
    if ctx.rank == 0:
        kernels = read_kernels(ctx, ("gemm", "gemm2", "gemm3", "gemm1", "ax", "ax2", "ax3", "cmap", "head") +
                               (("lstm", "lstm2", "embed") if args.lm else ())) if timing_period else {}
        roof, roof_ax = rooflines(ctx, eng, pk, kernels, args.lm)
        mean_len = float(np.mean([len(s) for s in seqs]))
        what = {
            "configs2": f"configs[2]: {n_local} synthetic L={args.length} proteins per GPU, GCN_MF+BP+CC (T={T_total}), fused cmap(6A, gen=2)+GCN, identity alignments",
            "mixed": f"configs[3]-shaped: {n_local} synthetic proteins per GPU, L~U[128,1024], 5% indels, GCN_MF+BP+CC (T={T_total}), fused cmap align(6A, gen=2)+GCN",
            "configs3": f"configs[3]: {n_job} synthetic proteins in total, L~U[128,1024] (seed 46), 5% indels, dealt to {ctx.world} rank(s) by cost, "
                        f"GCN_MF+BP+CC (T={T_total}), fused cmap align+GCN, GPU filter (score>=0.1) + one gather of the survivors per head",
            "configs4": f"configs[4]: {n_job} synthetic proteins in total, lengths from the reference test proteome's histogram clipped to [30,2048] (seed 47), "
                        f"5% indels, dealt to {ctx.world} rank(s) by cost, GCN_MF+BP+CC (T={T_total}), fused cmap align+GCN, GPU filter + one gather per head",
        }[args.workload]
        line = {
            "metric": ("proteins/sec (GCN+cmap) at L=512" if args.workload == "configs2" and args.length == 512 else
                       f"proteins/sec (GCN+cmap), {args.workload}" + (f" L={args.length}" if args.workload == "configs2" else "")) +
                      (" [with LSTM language model]" if args.lm else ""),
            "value": round(n_job * args.steps / elapsed, 1),
            "unit": "proteins/s",
            "n_gpus": ctx.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": ("inputs, outputs and every accumulation are fp32; with roofline.pipe = bf16x6 each fp32 product of the GraphConv H.W GEMMs is "
                           "formed from six bf16 term products of exactly split operands (error against float64 below the fp32 matrix instruction's: "
                           "profiles/r04_gemm_bf16x6_probe.txt, tests/test_gpu_gcn.py::test_bf16x6_products_are_at_least_as_accurate_as_the_fp32_instruction); "
                           "`f32_pipe` is the same run on the fp32 instruction") if ctx.lib.mdf_hw_pipe().decode() == "bf16x6" else None,
            "data": "synthetic",
            "config": {"workload": what, "proteins_total": n_job, "proteins_rank0": n_local, "mean_length_rank0": round(mean_len, 1),
                       "length": args.length if args.workload == "configs2" else None, "go_heads": list(MODES), "language_model": bool(args.lm),
                       "chunk_rows": args.chunk_rows,
                       "parallelism": (f"shard{ctx.world}+gather" if ctx.world > 1 else "single"), "backend": args.backend if ctx.world > 1 else None},
            "roofline": roof,
            "roofline_ax": roof_ax,
            "board": board.summary(),
            "kernels": kernels,
        }
        if kernels and args.lm:
            # the language-model branch's own kernels against the roofline of the pipe they run on: an LSTM time step is one
            # k_gemm_bf16x6<LSTM_*> launch per layer -- M = proteins of the group, N = 4 H, K = H (layer 1: the letter's table row rides in the
            # epilogue) or 2 H (layer 2: [x_t | h_{t-1}]) --, the embedding one launch per chunk (M = rows, N = embed, K = H)
            H, E = int(weights[MODES[0]]["lm_U1"].shape[0]), int(weights[MODES[0]]["W_lm"].shape[1])
            lm_b = args.lm_batch if args.lm_batch > 0 else 16384
            n_groups = max(1, -(-n_local // min(lm_b, 65535)))
            B_grp = n_local / n_groups
            lm_pipe = ctx.lib.mdf_hw_pipe().decode()
            peak = MFMA_F32_PEAK_TF if lm_pipe == "f32" else MFMA_BF16_PEAK_TF / (F16X3_PRODUCTS if lm_pipe == "f16x3" else BF16X6_PRODUCTS)
            rows_launch = sum(c.rows for c in pk.chunks) / len(pk.chunks)
            fl = {"lstm": 2.0 * B_grp * 4 * H * H, "lstm2": 2.0 * B_grp * 4 * H * 2 * H, "embed": 2.0 * rows_launch * E * H}
            line["lm_kernels"] = {k: {"avg_us": kernels[k]["avg_us"], "timed_launches": kernels[k]["launches"], "flops_per_launch": fl[k],
                                      "tflops": round(fl[k] / (kernels[k]["avg_us"] * 1e-6) / 1e12, 1),
                                      "frac": round(fl[k] / (kernels[k]["avg_us"] * 1e-6) / 1e12 / peak, 4)}
                                  for k in ("lstm", "lstm2", "embed") if kernels.get(k, {}).get("launches")}
            both = line["lm_kernels"].get("lstm"), line["lm_kernels"].get("lstm2")
            if all(both):   # the two layers' launches of a time step run under each other on two streams: their flops over the longer of the two
                t_step = max(both[0]["avg_us"], both[1]["avg_us"]) * 1e-6
                line["lm_kernels"]["recurrence"] = {"tflops": round((fl["lstm"] + fl["lstm2"]) / t_step / 1e12, 1),
                                                    "frac": round((fl["lstm"] + fl["lstm2"]) / t_step / 1e12 / peak, 4),
                                                    "note": "both layers of one time step together: the rate to hold against the H.W GEMM's (roofline.frac)"}
            line["lm_kernels"]["groups"] = {"count": n_groups, "proteins_per_group": round(B_grp, 1), "peak_tflops": round(peak, 1)}
        if kernels and not args.lm:
            # the sampled mean of every kernel class x the launches a step really makes: must add up to the step (nothing skipped, no idle stream)
            nC, nH = len(pk.chunks), len(MODES)
            per_step = {"gemm1": nC * nH, "ax2": nC * nH, "gemm2": nC * nH, "ax3": nC * nH, "gemm3": nC * nH, "cmap": nC, "head": nH}
            line["kernel_sum_ms_per_step"] = round(sum(kernels[k]["avg_us"] * n for k, n in per_step.items() if kernels.get(k, {}).get("launches")) / 1e3, 3)
            line["launches_per_step"] = per_step
        if strong:
            surv = {m: int(gathered[m][1].numel()) for m in MODES}
            line["gathered_survivors"] = surv
            line["gathered_survivors_per_protein"] = {m: round(surv[m] / n_job, 2) for m in MODES}
            line["survivor_fraction"] = {m: round(surv[m] / (n_job * preds[m].n_terms), 4) for m in MODES}
            # what the step's gathers move into rank 0: 8 B per surviving (term, score) pair + 8 B per protein of counts
            line["gather_bytes_per_step"] = {"filtered": int(8 * sum(surv.values()) + 8 * n_job * len(MODES)), "dense_equivalent": int(4 * n_job * T_total)}
        elif ctx.world > 1:
            line["gather_bytes_per_step"] = {"dense": int(4 * n_job * T_total)}
        rank_info["per_rank_ms_per_step"] = split     # forward (compute) and gather, per rank: an imbalance or a slow link shows here
        rank_info["compute_ms_min_max"] = [min(split["compute_ms"]), max(split["compute_ms"])]
        rank_info["gather_ms_min_max"] = [min(split["gather_ms"]), max(split["gather_ms"])]
        line["ranks"] = rank_info
        if args.verify > 0:
            line["verify"] = verify(seqs, coords, q_alns, t_alns, weights, out, args.verify, args.lm)
            if strong:   # and what rank 0 holds after the gather == filtering rank 0's own scores (its proteins sit at `mine`)
                off_g, ti_g, sc_g = (x.cpu().numpy() for x in gathered[MODES[0]])
                off_l, ti_l, sc_l = (x.cpu().numpy() for x in filter_scores(out[MODES[0]], threshold=0.1, capacity_per_protein=preds[MODES[0]].n_terms))
                ok = all(np.array_equal(ti_g[off_g[g]:off_g[g + 1]], ti_l[off_l[k]:off_l[k + 1]]) and
                         np.array_equal(sc_g[off_g[g]:off_g[g + 1]], sc_l[off_l[k]:off_l[k + 1]])
                         for k, g in list(enumerate(mine))[::max(1, len(mine) // 64)])
                line["verify"]["gather_restores_input_order"] = bool(ok)
                if not ok:
                    print(json.dumps(line), flush=True)
                    raise SystemExit("gathered survivors differ from the local filter")
            if not line["verify"]["max_abs_err_vs_oracle"] < 1e-4:
                print(json.dumps(line), flush=True)
                raise SystemExit(f"parity check failed: max |score - oracle| = {line['verify']['max_abs_err_vs_oracle']}")
        if ctx.world == 1 and args.workload == "configs2" and not args.no_extras and not args.lm:
            # the rest of the north_star's measurement list on the same engine, each a short run of its own (never `value`)
            # a leg that fails is reported in its own field and does not take the headline line with it
            def leg(name, fn):
                try:
                    line[name] = fn()
                except Exception as e:  # noqa: BLE001
                    line[name] = {"error": f"{type(e).__name__}: {e}"[:500]}

            leg("by_length", lambda: {str(L): mini_run(ctx, eng, make_fixed_length(42 + 2 + L, n_local, L), args.chunk_rows) for L in (256, 1024)})
            leg("mixed", lambda: mini_run(ctx, eng, make_mixed(42 + 4, n_local), args.chunk_rows))
            leg("helix", lambda: dict(mini_run(ctx, eng, make_helix(42 + 5, n_local, args.length), args.chunk_rows),
                                      note="same shape as the headline run, protein-like C-alpha traces (helix bundles) instead of random walks"))

            def thr10_leg():
                # the operating point of the released model files (`..._ca_10.0_...`, /root/reference/mDeepFRI/__init__.py:73,78): contacts at 10 A,
                # generated_contacts 2, protein-like traces -- three to four times the entries per row of the 6 A headline: an engine of its own
                # (threshold is a property of the engine), the same heads, one chunk geometry
                eng10 = batch.HotPathEngine(preds, device=local_rank, max_rows=args.chunk_rows, threshold=10.0, generated_contacts=2, nnz_per_row=96)
                res = mini_run(ctx, eng10, make_helix(42 + 7, n_local, args.length), args.chunk_rows)
                return dict(res, threshold_A=10.0, generated_contacts=2, nnz_per_row=round(float(eng10.last_chunk_nnz()) / float((n_local * args.length) % args.chunk_rows or args.chunk_rows), 2),
                            note="same shape as the headline run at the released models' 10 A threshold on helix-bundle traces; roofline_ax counts this "
                                 "density's own bytes per row (2 x 2 KiB + 4 + 8 x entries)")

            def host_batches(n_batches):
                # two distinct batches of host lists, taken in turn (generating 8 x 10 000 proteins would take longer than the leg itself)
                two = [make_fixed_length(777 + k, n_local, args.length) for k in range(2)]
                return [two[k & 1] for k in range(n_batches)]

            def steady(done, t0, per_batch):
                # done[k]: the wall clock at which batch k's score arrays were in the caller's hands.  `value` = all batches over all of the
                # time (fill and drain included); `steady` = the rate at which results come out once the pipeline is full (batches 1 .. n - 1
                # over the time between the first and the last completion); `fill_ms` = the first batch alone
                n = len(done)
                out = {"value": round(n * per_batch / (done[-1] - t0), 1), "unit": "proteins/s", "batches": n, "fill_ms": round(1e3 * (done[0] - t0), 1)}
                if n > 1:
                    out["steady"] = round((n - 1) * per_batch / (done[-1] - done[0]), 1)
                    out["steady_ms_per_batch"] = round(1e3 * (done[-1] - done[0]) / (n - 1), 2)
                return out

            def end_to_end_leg():
                from mDeepFRI.stream import AlignmentStream
                items = []
                for s2, c2, _, _ in host_batches(args.end_to_end):
                    items += [(a, b, a, a) for a, b in zip(s2, c2)]
                stream = AlignmentStream(eng, batch_size=n_local, max_rows=args.chunk_rows)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done, n_out = [], 0
                for _, res in stream.run(items):
                    n_out += res[MODES[0]].shape[0]
                    done.append(time.perf_counter())
                assert n_out == len(items)
                return dict(steady(done, t0, n_local),
                            note="PCIe-inclusive: host lists in -> host float32 score arrays out (packing thread + upload + compute + "
                                 "download, batches pipelined; mDeepFRI.stream.AlignmentStream); never `value`")

            def host_pipeline_leg(which):
                # the same through the library's own two-slot pipeline (mdf_engine_submit_alignments_host / mdf_engine_collect_host): `binding` =
                # the COMPILED reference-side module (tests/binding/predict.pyx BatchEngine: cdef extern calls, no torch, no ctypes), `ctypes` =
                # mDeepFRI.batch.HostPipeline.  Every batch's scores are compared bit for bit with the device-resident run's when the lists are the same
                batches = host_batches(args.end_to_end)
                if which == "binding":
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    import tempfile
                    import binding_loader
                    from mDeepFRI import weights as wfile
                    _, bp = binding_loader.load()
                    with tempfile.TemporaryDirectory() as td:
                        bpreds = []
                        for m in MODES:
                            wfile.save_mdfw(os.path.join(td, m + ".mdfw"), weights[m])
                            bpreds.append(bp.Predictor(os.path.join(td, m + ".mdfw")))
                    runner = bp.BatchEngine(bpreds, max_rows=args.chunk_rows)
                    as_dict = lambda res: dict(zip(MODES, res))  # noqa: E731
                else:
                    runner = batch.HostPipeline(eng)
                    as_dict = lambda res: res  # noqa: E731
                runner.submit(seqs, coords, q_alns, t_alns)      # the headline run's own lists: the same bits as its device-resident scores? (also sizes the buffers)
                ref = as_dict(runner.collect())
                same = all(np.array_equal(ref[m], out[m].cpu().numpy()) for m in MODES)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done = []
                for res in runner.run(batches):
                    done.append(time.perf_counter())
                return dict(steady(done, t0, n_local), bitwise_equal_to_device_resident=bool(same),
                            note=("compiled binding (tests/binding/predict.pyx BatchEngine)" if which == "binding" else "mDeepFRI.batch.HostPipeline (ctypes)") +
                                 ": host lists in -> host float32 arrays out through mdf_engine_submit_alignments_host / mdf_engine_collect_host "
                                 "(two batches in flight: packing and unpacking under the other batch's kernels); never `value`")

            def query_stream_leg():
                # the stages either side of the path as one stream (SURVEY 8f rows 3 and 4 around the path): host sequences + candidate sets
                # in, results.tsv text of all three heads out; heads at a trained-like operating point (a few % of the terms pass 0.1)
                from mDeepFRI.alignment import ScoringMatrix
                from mDeepFRI.output import results_text
                from mDeepFRI.stream import QueryStream
                w_sparse = make_weights(False, sparse_scores=True)
                eng_q = batch.HotPathEngine({m: Predictor(f"synthetic-sparse-{m}", weights=w_sparse[m], device=local_rank) for m in MODES},
                                            device=local_rank, max_rows=args.chunk_rows)
                nq = 4000 * args.query_stream
                qids, qseqs, cands, db_xyz = make_queries(42 + 6, nq)
                terms = {m: [f"GO:{k:07d}" for k in range(synthetic.GO_TERMS[m])] for m in MODES}
                qs = QueryStream(eng_q, db_xyz, batch_size=4000, max_rows=args.chunk_rows, scoring_matrix=ScoringMatrix.simple(), threshold=0.1)
                for rep in range(2):      # the first pass sizes the buffers
                    n_lines = n_bytes = 0
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for r in qs.run(qids, qseqs, cands):
                        ids = [r.batch.query_ids[i] for i in r.kept]
                        for m in MODES:
                            text = results_text(ids, "gcn", m, terms[m], terms[m], *r.gcn[m])
                            n_lines += text.count(b"\n")
                            n_bytes += len(text)
                    dt = time.perf_counter() - t0
                return {"value": round(nq / dt, 1), "unit": "proteins/s", "batches": args.query_stream, "queries": nq,
                        "candidates_per_query": 8, "mean_query_length": round(float(np.mean([len(q) for q in qseqs])), 1),
                        "result_lines": n_lines, "text_mb": round(n_bytes / 1e6, 1),
                        "note": "host sequences + candidate sets in -> best hit + alignment (GPU aligner) -> contact maps + GCN, 3 heads -> "
                                "score >= 0.1 filter -> results.tsv text out (mDeepFRI.stream.QueryStream: one stream, software pipeline "
                                "four batches deep; includes its fill and drain); never `value`"}

            def gcn_only_leg():
                # the other half of SURVEY 8(d)'s metric: the GCN alone, contact maps given.  (i) as the reference's API hands them over --
                # dense int32 (L, L) arrays in host memory, 1 MiB per protein at L = 512 (Predictor.forward_pass's shape, batched:
                # HotPathEngine.forward_dense): PCIe-bound; (ii) the fused step without its contact-map kernels (their sampled time
                # taken off the step): what the GCN kernels alone sustain on maps already in HBM.
                from mDeepFRI.alignment import AlignmentResult
                n = max(1, min(1024, n_local, (1 << 30) // (4 * args.length * args.length)))      # at most 1 GiB of maps in host memory
                alns = []
                for k in range(n):
                    a = AlignmentResult(query_name=f"p{k}", query_sequence=seqs[k], target_name=f"t{k}", target_sequence=seqs[k], alignment="M" * len(seqs[k]))
                    a.gapped_sequence, a.gapped_target, a.coords = q_alns[k], t_alns[k], coords[k]
                    alns.append(a)
                dense_rows = min(args.chunk_rows, batch.DENSE_CHUNK_ROWS)   # (1 MiB of map per protein at L = 512: chunks of 128 proteins keep the double-buffered upload busy)
                maps = [cm for _, cm in batch.build_align_contact_maps(alns, device=local_rank, max_rows=dense_rows)]
                pk_d = batch.PackedProteins.pack(seqs[:n], max_rows=dense_rows)
                db_d = eng.upload(pk_d)
                eng.forward_dense(db_d, maps)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out_d = eng.forward_dense(db_d, maps)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                err = float(max((out_d[m][:8] - out[m][:8]).abs().max().item() for m in MODES))
                res = {"value": round(n / dt, 1), "unit": "proteins/s", "proteins": n, "host_map_bytes": int(sum(m_.nbytes for m_ in maps)),
                       "max_abs_diff_vs_fused_path": err,
                       "note": "GCN alone on dense int32 (L, L) contact maps handed over from host memory (the reference API's format, batched: "
                               "forward_dense): PCIe-inclusive, 1 MiB per protein at L = 512; never `value`"}
                cm = kernels.get("cmap") if isinstance(kernels, dict) else None
                if cm and cm.get("launches"):
                    chunks_per_step = len(pk.chunks)
                    cmap_ms = cm["avg_us"] * chunks_per_step / 1e3
                    res["maps_in_hbm"] = {"value": round(n_local / ((1e3 * elapsed / args.steps - cmap_ms) / 1e3), 1), "unit": "proteins/s",
                                          "note": f"the fused step minus its contact-map kernels ({cmap_ms:.2f} ms of {1e3 * elapsed / args.steps:.2f} ms per step, sampled "
                                                  "launch times): the GCN kernels alone on maps already in HBM"}
                return res

            def f32_pipe_leg():
                # the same headline step with the H.W products on the fp32 matrix instruction (MDFRI_HW_PIPE=f32, read at library load: a child
                # process started before anything else of this one runs on the device again): the comparison the BF16x6 claim rests on
                torch.cuda.synchronize()
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
                env["MDFRI_HW_PIPE"] = "f32"
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-seconds", "0",
                                    "--proteins", str(n_local), "--length", str(args.length), "--chunk-rows", str(args.chunk_rows)],
                                   env=env, capture_output=True, text=True, timeout=900)
                sub = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                return {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "pipe": sub["roofline"]["pipe"],
                        "roofline_frac": sub["roofline"]["frac"], "gemm_avg_us": sub["roofline"]["per_launch"]["avg_us"],
                        "max_abs_err_vs_oracle": sub["verify"]["max_abs_err_vs_oracle"],
                        "note": "the same run with H.W on v_mfma_f32_32x32x2_f32 (k_gemm_f32; MDFRI_HW_PIPE=f32): its rate, its fraction of the fp32 "
                                "instruction's peak, and its error against the float64 oracle beside the default path's `verify`; never `value`"}

            def f16x3_pipe_leg():
                # the same headline step with the GraphConv H.W products as F16x3 (MDFRI_HW_PIPE=f16x3, opt-in: two fp16 terms per scaled operand, three
                # term products per fp32 product -- half the matrix work of the default, 22 of 24 operand bits): its rate and its error beside the default's
                torch.cuda.synchronize()
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
                env["MDFRI_HW_PIPE"] = "f16x3"
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-seconds", "0",
                                    "--proteins", str(n_local), "--length", str(args.length), "--chunk-rows", str(args.chunk_rows)],
                                   env=env, capture_output=True, text=True, timeout=900)
                sub = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                return {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "pipe": sub["roofline"]["pipe"],
                        "roofline_frac": sub["roofline"]["frac"], "gemm_avg_us": sub["roofline"]["per_launch"]["avg_us"],
                        "max_abs_err_vs_oracle": sub["verify"]["max_abs_err_vs_oracle"],
                        "note": "OPT-IN, never `value`: the same run with the GraphConv H.W products as F16x3 (k_gemm_f16x3; MDFRI_HW_PIPE=f16x3): operands scaled "
                                "by a power of two and split into two fp16 terms (22 of their 24 bits; BF16x6's three-term split is exact), three term products "
                                "instead of six; its rate, its fraction of fp16 peak / 3, and its error against the float64 oracle beside the default path's "
                                "`verify` (DESIGN.md section 4: why it is not the default)"}

            leg("f32_pipe", f32_pipe_leg)
            leg("f16x3_pipe", f16x3_pipe_leg)
            leg("gcn_only", gcn_only_leg)
            leg("thr10", thr10_leg)
            if args.end_to_end > 0:
                leg("end_to_end", end_to_end_leg)
                leg("binding", lambda: host_pipeline_leg("binding"))
                leg("host_pipeline", lambda: host_pipeline_leg("ctypes"))
            if args.query_stream > 0:
                leg("query_stream", query_stream_leg)
        line["cpu_baseline"] = cpu_leg
        if cpu_leg:
            line["gpu_over_cpu_1core"] = round(line["value"] / cpu_leg["value"], 1)
        print(json.dumps(line), flush=True)
    if ctx.world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_cnn(args, ctx, local_rank):
    """Extra measurement (not the BASELINE.json metric): the sequence-only CNN models the reference runs on proteins without a
    structural hit (pipeline.py:600-648), 3 heads, upstream DeepCNN default topology, same synthetic sequences."""
    import torch
    import torch.distributed as dist
    from mDeepFRI import _hip, batch
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    n_local = args.proteins or 10000
    weights = {m: synthetic.glorot_cnn_weights(seed=i, n_terms=synthetic.GO_TERMS[m]) for i, m in enumerate(MODES)}
    preds = {m: Predictor(f"synthetic-cnn-{m}", weights=weights[m], device=local_rank) for m in MODES}
    seqs = make_fixed_length(42 + 2 + 1000 * rank, n_local, args.length)[0]
    eng = batch.SequenceEngine(preds, device=local_rank)
    db = batch.DeviceBatch(batch.PackedProteins.pack(seqs, max_rows=1 << 20), dev)
    lib = ctx.lib

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.forward(db)
    fence()
    eng.check(db)
    lib.mdf_timing_reset()
    lib.mdf_timing_enable(0 if args.no_kernel_timing else 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.forward(db)
    fence()
    elapsed = time.perf_counter() - t0
    lib.mdf_timing_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n, ms = _hip.c_int64(0), _hip.ctypes.c_double(0.0)
        lib.mdf_timing_read(b"cnn", n, ms)
        line = {"metric": "proteins/sec (sequence-only CNN) at L=%d" % args.length, "value": round(world * n_local * args.steps / elapsed, 1),
                "unit": "proteins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{n_local} synthetic L={args.length} sequences per GPU, DeepCNN MF+BP+CC (4 Conv1D branches "
                                       f"120/100/80/60 x 5/10/15/20, BatchNorm, max pool, FuncPredictor)", "go_heads": list(MODES)},
                "kernels": {"cnn": {"launches": int(n.value), "total_ms": round(ms.value, 3), "avg_us": round(1e3 * ms.value / max(n.value, 1), 2)}}}
        if args.verify > 0:
            oracle_paths()
            import cnn_oracle
            worst = 0.0
            for i in np.linspace(0, n_local - 1, args.verify).astype(int):
                for m in MODES:
                    worst = max(worst, float(np.max(np.abs(out[m][i].cpu().numpy() - cnn_oracle.cnn_forward(weights[m], seqs[i])))))
            line["verify"] = {"proteins": int(args.verify), "max_abs_err_vs_oracle": worst, "tolerance": 1e-4}
            if not worst < 1e-4:
                raise SystemExit(f"parity check failed: {worst}")
        if world == 1 and args.cpu_seconds > 0:
            oracle_paths()
            import cnn_oracle
            from threadpoolctl import threadpool_limits
            k, t1 = 0, time.perf_counter()
            with threadpool_limits(limits=1):
                while k < len(seqs) and time.perf_counter() - t1 < args.cpu_seconds * 0.5:
                    for m in MODES:
                        cnn_oracle.cnn_forward(weights[m], seqs[k])
                    k += 1
            line["cpu_baseline"] = {"value": k / (time.perf_counter() - t1), "unit": "proteins/s", "cores": 1, "kind": "port",
                                    "sample": f"{k} sequences, 3 heads each, numpy oracle",
                                    "published_anchor": "reference weight_convert/inference_times.csv.gz: ~0.018 s/protein/model/core at L~512 (ORT CPU)"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
