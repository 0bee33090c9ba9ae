#!/bin/bash
set -u
O=gpurun_out/exp6
mkdir -p "$O"
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench.py 2>&1 | tail -8
for w in configs2 mixed configs3 configs4; do
  timeout 900 python3 bench.py --workload $w --cpu-seconds 0 --no-extras --steps 2 > "$O/$w.json" 2> "$O/$w.err"
  python3 -c "
import json,sys
for ln in open('$O/$w.json'):
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernels']
        print('$w', d['value'], d['ms_per_step'], {n:k[n]['avg_us'] for n in ('gemm1','ax2','gemm2','ax3','gemm3','cmap') if n in k}, d.get('verify'))
"
done
