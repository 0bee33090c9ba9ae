import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print("# " + "/".join(f.split("/")[-2:]))
print(f"{'kernel':<64} {'calls':>5} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}")
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "").replace("mdf::", "")[:60]
    print(f"{n:<64} {int(r['Calls']):>5} {float(r['TotalDurationNs'])/1e6:>10.3f} {float(r['AverageNs'])/1e3:>9.2f} {float(r['MinNs'])/1e3:>9.2f} {float(r['MaxNs'])/1e3:>9.2f} {float(r['Percentage']):>6.2f}")
