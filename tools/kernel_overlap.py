"""How much do the launches of two kernels overlap in time?  Reads a rocprofv3 --kernel-trace directory.

    python3 tools/kernel_overlap.py DIR 'k_gemm_bf16x6<6' 'k_gemm_bf16x6<7'

For each name prefix: launches, sum of durations, union of their intervals; then the union of both together -- equal to the larger
union when one kernel always runs under the other, to the sum of the unions when they never share the chip.
"""
import csv, glob, os, sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    d, names = sys.argv[1], sys.argv[2:]
    per = {n: [] for n in names}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            for n in names:
                if row["Kernel_Name"].startswith(n):
                    per[n].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Queue_Id"]))
    both = []
    for n in names:
        iv = [(s, e) for s, e, _ in per[n]]
        both += iv
        q = sorted({x[2] for x in per[n]})
        if iv:
            print(f"{n:28s} launches {len(iv):6d}  sum {sum(e - s for s, e in iv) * 1e-6:9.2f} ms  union {union(iv) * 1e-6:9.2f} ms  "
                  f"avg {sum(e - s for s, e in iv) / len(iv) * 1e-3:7.1f} us  first..last {(max(e for _, e in iv) - min(s for s, _ in iv)) * 1e-6:9.2f} ms  queues {q}")
    if both:
        print(f"{'all of the above':28s} union {union(both) * 1e-6:9.2f} ms  first..last {(max(e for _, e in both) - min(s for s, _ in both)) * 1e-6:9.2f} ms")


if __name__ == "__main__":
    main()
