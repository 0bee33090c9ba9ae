"""Condense a rocprofv3 `--pmc` run (counter_collection.csv under DIR) into one row per kernel: dispatches and the MEAN of every counter."""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(int)
names = set()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mdf::", "")[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names.add(r["Counter_Name"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        n[k] += 1
names = sorted(names)
print("# " + "/".join(f.split("/")[-2:]))
print(f"{'kernel':<62} {'disp':>5} " + " ".join(f"{c:>22}" for c in names))
for k in sorted(acc):
    print(f"{k:<62} {n[k]:>5} " + " ".join(f"{acc[k][c] / n[k]:>22.1f}" for c in names))
