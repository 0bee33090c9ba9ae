#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/.

  kernel stats :  python profiles/summarize_rocprof.py stats  <dir with *_kernel_stats.csv>   > profiles/rNN_kernel_stats.txt
  counters     :  python profiles/summarize_rocprof.py pmc    <dir with *_counter_collection.csv> ...  > profiles/rNN_pmc.txt
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
