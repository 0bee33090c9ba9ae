"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run: for every queue, the gaps end(i) -> start(i+1).

    python3 tools/kernel_gaps.py DIR [min_kernels]

Prints, per queue with at least `min_kernels` dispatches: kernels, busy time (union), span, idle share, and the gap distribution by the
kernel that FOLLOWS the gap (a launch-bound stream shows as a few us in front of every kernel).
"""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    min_k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    per_q = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per_q[row["Queue_Id"]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    for q, ks in sorted(per_q.items()):
        if len(ks) < min_k:
            continue
        ks.sort()
        busy = sum(e - s for s, e, _ in ks)
        span = ks[-1][1] - ks[0][0]
        gaps = defaultdict(list)
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = s1 - e0
            name = n1.split("(")[0].replace("void mdf::", "").replace("mdf::", "")[:40]
            gaps[name].append(g)
        allg = [g for v in gaps.values() for g in v]
        small = [g for g in allg if g < 50000]     # (gaps of 50 us and more are host-side pauses between steps, not launch overhead)
        print(f"queue {q}: {len(ks)} kernels, busy {busy * 1e-6:.2f} ms, span {span * 1e-6:.2f} ms, gaps < 50 us: {len(small)} totalling {sum(small) * 1e-6:.2f} ms "
              f"({100.0 * sum(small) / max(busy, 1):.2f} % of busy), mean {sum(small) / max(len(small), 1) * 1e-3:.2f} us")
        for name, v in sorted(gaps.items(), key=lambda kv: -sum(x for x in kv[1] if x < 50000)):
            vs = sorted(x for x in v if x < 50000)
            if vs:
                print(f"   before {name:42s} n {len(vs):6d}  mean {sum(vs) / len(vs) * 1e-3:7.2f} us  median {vs[len(vs) // 2] * 1e-3:7.2f}  p90 {vs[int(len(vs) * 0.9)] * 1e-3:7.2f}  total {sum(vs) * 1e-6:8.3f} ms")


if __name__ == "__main__":
    main()
