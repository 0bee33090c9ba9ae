#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/exp7
mkdir -p "$O"
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench.py 2>&1 | tail -6
for w in configs2 configs3 configs4; do
  timeout 900 python3 bench.py --workload $w --cpu-seconds 0 --no-extras --steps 3 > "$O/$w.json" 2> "$O/$w.err"
  python3 -c "
import json,sys
for ln in open('$O/$w.json'):
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernels']
        print('$w', d['value'], d['ms_per_step'], {n:k[n]['avg_us'] for n in ('gemm1','ax2','gemm2','ax3','gemm3','cmap') if n in k}, d.get('kernel_sum_ms_per_step'))
"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- python3 bench.py --steps 3 --cpu-seconds 0 --no-extras --no-kernel-timing > "$O/prof.log" 2>&1
python3 tools/rocprof_summary.py "$O/prof" | head -24
python3 profiles/summarize_rocprof.py layers "$O/prof"
