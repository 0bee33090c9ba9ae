#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/.

  kernel stats :  python profiles/summarize_rocprof.py stats  <dir with *_kernel_stats.csv>   > profiles/rNN_kernel_stats.txt
  counters     :  python profiles/summarize_rocprof.py pmc    <dir with *_counter_collection.csv> ...  > profiles/rNN_pmc.txt
  per layer    :  python profiles/summarize_rocprof.py layers <dir with *_kernel_trace.csv>   >> profiles/rNN_kernel_stats.txt
                  (A.X and H.W launches split by GraphConv layer: a k_aggregate launch is layer 2 when the kernel in front of it is the
                  layer-1 kernel, layer 3 when it is an H.W GEMM; only full-size launches -- the modal grid -- are counted)
Counters are averaged per kernel name over all dispatches (one rocprofv3 --pmc pass per directory).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    name = name.replace("void ", "").replace("mdf::", "")
    return name.split("(")[0][:60]


def stats(d):
    for f in find(d, "*kernel_stats.csv"):
        print(f"# {os.path.relpath(f, d)}")
        rows = list(csv.DictReader(open(f)))
        print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for r in rows:
            print(f"{short(r['Name']):62s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} "
                  f"{float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")


def pmc(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    counters = sorted({c for k in acc.values() for c in k})
    print(f"{'kernel':62s} {'disp':>6s} " + " ".join(f"{c:>22s}" for c in counters))
    for k in sorted(acc):
        n = max(v[1] for v in acc[k].values())
        print(f"{k:62s} {n:6d} " + " ".join(f"{(acc[k][c][0]/acc[k][c][1] if acc[k][c][1] else float('nan')):22.1f}" for c in counters))


def layers(d):
    """Per-layer durations of the two named kernels from the kernel trace (the same split bench.py's `per_layer` reports from HIP events)."""
    for f in find(d, "*kernel_trace.csv"):
        rows = [r for r in csv.DictReader(open(f)) if r["Kind"] == "KERNEL_DISPATCH"]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        groups = defaultdict(list)
        prev = ""
        for r in rows:
            name = short(r["Kernel_Name"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid = int(r["Grid_Size_X"])
            if name.startswith("k_aggregate"):
                # the launch in front of a layer-2 aggregation is the layer-1 kernel (k_layer1; the K = 32 GEMM <4, ..> up to round 3), in
                # front of a layer-3 aggregation an H.W GEMM
                if name.startswith("k_aggregate_mfma") and ("true" in name.split("<", 1)[1] or "(bool)1" in name):
                    layer = "layer 2 (+ layer 1)"      # k_aggregate_mfma<.., true>: layer 1 is made inside this launch
                else:
                    layer = "layer 2" if prev.startswith("k_layer1") or "<4," in prev else ("layer 3" if prev.startswith(("k_gemm_f32", "k_gemm_bf16x6")) else "other")
                groups[("k_aggregate (A.X)", layer, name)].append((grid, dur))
            elif name.startswith(("k_gemm_f32<", "k_gemm_bf16x6<")):      # (not the _small forms: per-call API)
                kern, epi = name.split("<", 1)
                if epi.startswith("(Epilogue)0") or epi.startswith("0"):
                    groups[(f"{kern} (H.W)", "layer 2 (stores H2)", name)].append((grid, dur))
                elif epi.startswith("(Epilogue)1") or epi.startswith("1"):
                    groups[(f"{kern} (H.W)", "layer 3 (pool only)", name)].append((grid, dur))
            prev = name
        print(f"# per-layer split from {os.path.relpath(f, d)} (full-size launches = the modal launch grid of each group)")
        print(f"{'kernel':22s} {'layer':22s} {'launches':>8s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
        pooled = defaultdict(list)
        for (kern, layer, _), v in sorted(groups.items()):
            grids = [g for g, _ in v]
            mode = max(set(grids), key=grids.count)
            full = [t for g, t in v if g == mode]
            pooled[kern] += full
            print(f"{kern:22s} {layer:22s} {len(full):8d} {sum(full)/len(full):9.2f} {min(full):9.2f} {max(full):9.2f}")
        for kern, full in sorted(pooled.items()):
            print(f"{kern:22s} {'all layers pooled':22s} {len(full):8d} {sum(full)/len(full):9.2f} {min(full):9.2f} {max(full):9.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "layers":
        layers(sys.argv[2])
    else:
        pmc(sys.argv[2:])
