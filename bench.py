#!/usr/bin/env python3
"""bench.py -- proteins/sec of the fused hot path (C-alpha coords + alignment + sequence -> GO scores) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched through torch.distributed.run)
One JSON line on rank 0.  A "step" is one pass of the hot path over the whole workload of BASELINE.json configs[2]
-- 10 000 synthetic L=512 proteins, three GO heads (MF+BP+CC) -- per GPU (weak scaling: every rank owns its own
10 000 proteins), inputs resident in HBM before the timed region, results left on the device; with N>1 each step
ends with the RCCL gather of the (10 000, 2752) score block of every rank to rank 0.

Extra objects on the line (tier contract):
  roofline      dominant kernel (H.W fp32-MFMA GEMM): algorithmic flops per launch / mean launch duration measured with
                HIP events on the launch stream inside the timed region (mdf_timing_* hooks of the library)
  roofline_ax   the same for the A.X aggregation kernel against the HBM roofline (the north_star's named kernel)
  cpu_baseline  the oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)
MODES = ("mf", "bp", "cc")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--proteins", type=int, default=10000, help="proteins per GPU per step (configs[2]: 10000)")
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--chunk-rows", type=int, default=65536, help="residue rows per fused chunk (multiples of 32768 = full rounds of 256x256 GEMM tiles on 256 CUs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--timing-period", type=int, default=8,
                    help="bracket every n-th launch of each kernel class with HIP events (an event pair costs GPU time between "
                         "kernels: timing every launch lowers the step rate by ~6 %%)")
    ap.add_argument("--workload", default="configs2", choices=["configs2", "mixed", "cnn"],
                    help="configs2 (headline): fixed length, identity alignments; mixed: configs[3]-style L~U[128,1024] with 5%% indels")
    ap.add_argument("--verify", type=int, default=4,
                    help="after the timed region, check this many proteins of the step against the oracle (untimed; 0 = skip)")
    ap.add_argument("--lm", action="store_true",
                    help="give every GO head the language-model branch of the released models (shared 2x512 LSTM + per-head "
                         "LM embedding; SURVEY.md section 8f row 1).  Not the BASELINE.json configuration: an extra measurement.")
    ap.add_argument("--end-to-end", type=int, default=0, metavar="N",
                    help="also stream N batches of --proteins from HOST lists through mDeepFRI.stream.AlignmentStream (packing, "
                         "PCIe upload, compute, PCIe download of the scores) and report the PCIe-inclusive rate next to `value`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--force-device", type=int, default=None, help="testing aid: put every rank on this device ordinal")
    return ap.parse_args()


def make_workload(seed, count, L):
    """configs[2] inputs: uniform 20-letter sequences, 3.8 A random-walk C-alpha traces rounded to 3 decimals,
    identity alignments.  Vectorised version of mDeepFRI.synthetic.synthetic_proteins."""
    from mDeepFRI import synthetic
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(synthetic.AA20.encode(), dtype=np.uint8)
    idx = rng.integers(0, 20, size=(count, L))
    seq_bytes = letters[idx]
    seqs = [bytes(r).decode() for r in seq_bytes]
    v = rng.standard_normal((count, L, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True) + 1e-12
    xyz = np.round(np.cumsum(v * 3.8, axis=1), 3).astype(np.float32)
    return seqs, [xyz[i] for i in range(count)]


def cpu_baseline(seqs, coords, weights, budget_s):
    """Oracle chain (oracle/cmap_oracle.c + oracle/gcn_oracle.py) on host cores: 1 thread = the configuration the
    reference ships, then all cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cmap_oracle
    import gcn_oracle
    from threadpoolctl import threadpool_limits

    def run(limit, budget):
        n, t0 = 0, time.perf_counter()
        with threadpool_limits(limits=limit):
            while n < len(seqs) and (time.perf_counter() - t0 < budget or n < 2):
                cm = cmap_oracle.build_align_contact_map(coords[n], seqs[n], seqs[n], 6.0, 2)
                for m in MODES:
                    if "W_lm" in weights[m]:
                        import lm_oracle
                        lm_oracle.gcn_lm_forward(weights[m], seqs[n], cm)   # (the oracle re-runs the shared LSTM per head, as the reference's three ONNX sessions do)
                    else:
                        gcn_oracle.gcn_forward(weights[m], seqs[n], cm)
                n += 1
        return n, time.perf_counter() - t0

    n1, t1 = run(1, budget_s * 0.7)
    ncores = os.cpu_count() or 1
    na, ta = run(None, budget_s * 0.3)
    # contact-map stage alone through the REAL reference kernels (oracle/_ref: the reference's contact_map_utils.pyx compiled
    # with its own flags by oracle/build_ref.py), glued as reference bio_utils.py:196-227,348-385 does, 1 thread
    ref_stage = None
    try:
        import build_ref
        ref = build_ref.load()
        if ref is not None:
            m, t0 = 0, time.perf_counter()
            while m < min(len(seqs), 200) and time.perf_counter() - t0 < 2.0:
                D = ref.pairwise_sqeuclidean(coords[m])
                sparse = np.argwhere((D < 6.0**2).astype(np.int32) == 1).astype(np.int32)
                ref.align_contact_map(seqs[m], seqs[m], sparse, 2)
                m += 1
            ref_stage = {"kind": "reference", "ms_per_protein": round(1e3 * (time.perf_counter() - t0) / max(m, 1), 3),
                         "sample": f"{m} proteins, pairwise_sqeuclidean + threshold/argwhere + align_contact_map, 1 thread"}
    except Exception as e:  # the compiled reference is optional on the GPU box
        ref_stage = {"kind": "reference", "error": str(e)[:200]}
    return {"cmap_stage_reference": ref_stage, "value": n1 / t1, "unit": "proteins/s", "cores": 1, "kind": "port",
            "sample": f"{n1} of the step's L={len(seqs[0])} proteins, contact map + 3 GO heads each, {t1:.1f} s, numpy/BLAS pinned to 1 thread",
            "all_cores": {"value": na / ta, "cores": ncores, "sample": f"{na} proteins, {ta:.1f} s, BLAS threads unrestricted"},
            "published_anchor": "reference weight_convert/inference_times.csv.gz: 0.13 s/protein/model/core at L~512 (ORT CPU, model incl. LSTM LM)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from mDeepFRI import _hip, batch, sharding, synthetic
    from mDeepFRI.predict import Predictor

    if args.workload == "cnn":
        return bench_cnn(args, rank, local_rank, world, dev)

    weights = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m]) for i, m in enumerate(MODES)}
    if args.lm:
        lm = synthetic.glorot_lm_weights(seed=1000)
        for i, m in enumerate(MODES):
            weights[m].update(lm)
            weights[m]["W_lm"] = synthetic.glorot_uniform(np.random.default_rng(2000 + i), 512, 1024)  # LM_embedding is per head
    preds = {m: Predictor(f"synthetic-{m}", weights=weights[m], device=local_rank) for m in MODES}
    T_total = sum(p.n_terms for p in preds.values())

    if args.workload == "mixed":   # configs[3] shape: ragged lengths, gapped alignments (sorted by length as pipeline.py:529)
        prots = synthetic.synthetic_proteins(42 + 3 + 1000 * rank, args.proteins, (128, 1024), indel_rate=0.05)
        prots.sort(key=lambda p: len(p["seq"]))
        seqs, coords = [p["seq"] for p in prots], [p["coords"] for p in prots]
        q_alns, t_alns = [p["q_aln"] for p in prots], [p["t_aln"] for p in prots]
    else:
        seqs, coords = make_workload(42 + 2 + 1000 * rank, args.proteins, args.length)  # seed = 42 + config index (+rank)
        q_alns = t_alns = seqs
    eng = batch.HotPathEngine(preds, device=local_rank, max_rows=args.chunk_rows)
    pk = batch.PackedProteins.pack(seqs, coords, q_alns, t_alns, max_rows=args.chunk_rows)
    db = eng.upload(pk)
    lib = _hip.lib()
    global_index = list(range(rank * args.proteins, (rank + 1) * args.proteins))

    def step():
        out = eng.forward_alignments(db)
        if world > 1:
            block = torch.cat([out[m] for m in MODES], dim=1)
            sharding.gather_scores(block, global_index, total=world * args.proteins, dst=0)
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.check(db)  # invalid residues / CSR overflow would surface here
    timing = not args.no_kernel_timing
    lib.mdf_timing_reset()
    lib.mdf_timing_enable(max(1, args.timing_period) if timing else 0)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.mdf_timing_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng.check(db)

    if rank == 0:
        def read(kind):
            n, ms = _hip.c_int64(0), _hip.ctypes.c_double(0.0)
            lib.mdf_timing_read(kind.encode(), n, ms)
            return int(n.value), float(ms.value)

        R = pk.chunks[0].rows                        # rows per launch (all chunks but the last are equal)
        rows_total = sum(c.rows for c in pk.chunks)
        roof, roof_ax, kernels = None, None, {}
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json); same rows per launch only
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if traffic.get("rows_per_launch") != R:
                traffic = {}
        except OSError:
            traffic = {}
        if timing:
            for kname in ("gemm", "gemm1", "ax", "cmap", "head") + (("lstm", "lstm2", "embed") if args.lm else ()):
                n, ms = read(kname)
                kernels[kname] = {"launches": n, "total_ms": round(ms, 3), "avg_us": round(1e3 * ms / max(n, 1), 2)}
            n_g, ms_g = read("gemm")
            n_a, ms_a = read("ax")
            # algorithmic work of one launch (mean rows per chunk; all chunks but the last are equal) / mean duration of
            # the launches that were bracketed with HIP events (every --timing-period-th one, on the launch stream)
            C = 512
            rows_launch = rows_total / len(pk.chunks)
            if n_g:
                # with --lm the class also holds the unfolded layer-1 launch (K = 1024): mean over the three layers
                flops_launch = 2.0 * rows_launch * C * ((1024 + C + C) / 3.0 if args.lm else C)
                tf = flops_launch / (ms_g / n_g * 1e-3) / 1e12
                roof = {"kernel": "k_gemm_f32 (H.W, 256x256x32 tiles, v_mfma_f32_32x32x2_f32, LDS-DMA staging, ELU+pool epilogue)",
                        "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": traffic.get("gemm_mean_bytes"),
                        "per_launch": {"rows": R, "flops": 2.0 * R * C * C, "avg_us": kernels["gemm"]["avg_us"],
                                       "timed_launches": n_g}}
            if n_a:
                # read Z (rows x 512 f32) once + write (rows x 512 f32) once per layer; CSR adjacency
                # (4 B colidx + 4 B val per nnz + 4 B rowptr per row) added and stated
                nnz_per_row = float(os.environ.get("MDFRI_BENCH_NNZ_PER_ROW", "0")) or eng_nnz_per_row(eng, db, pk)
                bytes_launch_rows = 2 * 4 * C + 4 + 8 * nnz_per_row
                gbs = bytes_launch_rows * rows_launch / (ms_a / n_a * 1e-3) / 1e9
                roof_ax = {"kernel": "k_aggregate<512> (A.X, CSR gather, one wave per residue row)", "bound": "hbm",
                           "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                           "traffic": (traffic.get("k_aggregate<512>") or {}).get("bytes"),
                           "per_launch": {"rows": R, "bytes": bytes_launch_rows * R, "nnz_per_row": round(nnz_per_row, 2),
                                          "avg_us": kernels["ax"]["avg_us"], "timed_launches": n_a}}
        line = {
            "metric": "proteins/sec (GCN+cmap) at L=512" + (" [with LSTM language model]" if args.lm else ""),
            "value": round(world * args.proteins * args.steps / elapsed, 1),
            "unit": "proteins/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"configs[2]: {args.proteins} synthetic L={args.length} proteins per GPU, GCN_MF+BP+CC "
                                    f"(T={T_total}), fused cmap(6A, gen=2)+GCN, identity alignments") if args.workload == "configs2" else
                                   (f"configs[3]-style: {args.proteins} synthetic proteins per GPU, L~U[128,1024], 5% indels, "
                                    f"GCN_MF+BP+CC (T={T_total}), fused cmap align(6A, gen=2)+GCN"),
                       "proteins_per_gpu": args.proteins, "length": args.length, "go_heads": list(MODES),
                       "language_model": bool(args.lm), "chunk_rows": args.chunk_rows, "parallelism": f"shard{world}+gather" if world > 1 else "single"},
            "roofline": roof,
            "roofline_ax": roof_ax,
            "kernels": kernels,
        }
        if args.verify > 0:
            # parity spot check on the very outputs of the timed steps (oracle = checker, outside the timed region)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import cmap_oracle
            import gcn_oracle
            pick = np.linspace(0, args.proteins - 1, args.verify).astype(int)
            worst = 0.0
            for i in pick:
                cm = cmap_oracle.build_align_contact_map(coords[i], q_alns[i], t_alns[i], 6.0, 2)
                for m in MODES:
                    if args.lm:
                        import lm_oracle
                        ref = lm_oracle.gcn_lm_forward(weights[m], seqs[i], cm)
                    else:
                        ref = gcn_oracle.gcn_forward(weights[m], seqs[i], cm)
                    worst = max(worst, float(np.max(np.abs(out[m][i].cpu().numpy() - ref))))
            line["verify"] = {"proteins": int(args.verify), "heads": list(MODES), "max_abs_err_vs_oracle": worst, "tolerance": 1e-4}
            if not worst < 1e-4:
                print(json.dumps(line), flush=True)
                raise SystemExit(f"parity check failed: max |score - oracle| = {worst}")
        if world == 1 and args.end_to_end > 0 and args.workload == "configs2":
            from mDeepFRI.stream import AlignmentStream
            items = []
            for k in range(args.end_to_end):
                s2, c2 = make_workload(777 + k, args.proteins, args.length)
                items += [(a, b, a, a) for a, b in zip(s2, c2)]
            stream = AlignmentStream(eng, batch_size=args.proteins, max_rows=args.chunk_rows)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_out = 0
            for _, res in stream.run(items):
                n_out += res[MODES[0]].shape[0]
            dt = time.perf_counter() - t0
            assert n_out == len(items)
            line["end_to_end"] = {"value": round(len(items) / dt, 1), "unit": "proteins/s", "batches": args.end_to_end,
                                  "note": "host lists in -> host float32 score arrays out: packing thread + PCIe upload + compute + "
                                          "PCIe download, batches pipelined (mDeepFRI.stream.AlignmentStream); never `value`"}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(seqs[:1024], coords[:1024], weights, args.cpu_seconds) if args.workload == "configs2" else None
            line["gpu_over_cpu_1core"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_cnn(args, rank, local_rank, world, dev):
    """Extra measurement (not the BASELINE.json metric): the sequence-only CNN models the reference runs on proteins without a
    structural hit (pipeline.py:600-648), 3 heads, upstream DeepCNN default topology, same synthetic sequences."""
    import torch
    import torch.distributed as dist
    from mDeepFRI import _hip, batch, synthetic
    from mDeepFRI.predict import Predictor
    weights = {m: synthetic.glorot_cnn_weights(seed=i, n_terms=synthetic.GO_TERMS[m]) for i, m in enumerate(MODES)}
    preds = {m: Predictor(f"synthetic-cnn-{m}", weights=weights[m], device=local_rank) for m in MODES}
    seqs, _ = make_workload(42 + 2 + 1000 * rank, args.proteins, args.length)
    eng = batch.SequenceEngine(preds, device=local_rank)
    db = batch.DeviceBatch(batch.PackedProteins.pack(seqs, max_rows=1 << 20), dev)
    lib = _hip.lib()

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
        line = {"metric": "proteins/sec (sequence-only CNN) at L=%d" % args.length, "value": round(world * args.proteins * args.steps / elapsed, 1),
                "unit": "proteins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.proteins} synthetic L={args.length} sequences per GPU, DeepCNN MF+BP+CC (4 Conv1D branches "
                                       f"120/100/80/60 x 5/10/15/20, BatchNorm, max pool, FuncPredictor)", "go_heads": list(MODES)},
                "kernels": {"cnn": {"launches": int(n.value), "total_ms": round(ms.value, 3), "avg_us": round(1e3 * ms.value / max(n.value, 1), 2)}}}
        if args.verify > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import cnn_oracle
            worst = 0.0
            for i in np.linspace(0, args.proteins - 1, args.verify).astype(int):
                for m in MODES:
                    worst = max(worst, float(np.max(np.abs(out[m][i].cpu().numpy() - cnn_oracle.cnn_forward(weights[m], seqs[i])))))
            line["verify"] = {"proteins": int(args.verify), "max_abs_err_vs_oracle": worst, "tolerance": 1e-4}
            if not worst < 1e-4:
                raise SystemExit(f"parity check failed: {worst}")
        if world == 1 and args.cpu_seconds > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
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


def eng_nnz_per_row(eng, db, pk):
    """Mean CSR entries per residue row of the last chunk processed (rowptr[R] / R), read back from the device."""
    rp = eng._bufs["rowptr"]
    last = pk.chunks[-1]
    return float(rp[last.rows].item()) / float(last.rows)


if __name__ == "__main__":
    main()
