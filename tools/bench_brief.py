"""Developer aid: run bench.py with the given extra arguments and print a one-line digest (value, ms/step, verify error)."""
import json
import os
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-seconds", "0", "--no-extras"] + sys.argv[1:], capture_output=True, text=True)
line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
if not line:
    print("bench failed:", out.stderr[-1500:])
    sys.exit(1)
d = json.loads(line[0])
k = d.get("kernels") or {}
print(d["value"], "proteins/s", d["ms_per_step"], "ms/step", "verify", (d.get("verify") or {}).get("max_abs_err_vs_oracle"),
      {n: v["avg_us"] for n, v in k.items()})
