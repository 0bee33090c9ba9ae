#!/bin/bash
set -u
O=gpurun_out/exp15
mkdir -p "$O"
for rep in 1 2; do
for v in 1 0; do
for w in configs3 configs4; do
  MDFRI_AX_MFMA=$v timeout 900 python3 bench.py --workload $w --cpu-seconds 0 --no-extras --steps 2 --verify 0 > "$O/t.json" 2> "$O/t.err"
  python3 -c "
import json
for ln in open('$O/t.json'):
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernels']
        print('$w mfma=$v', d['value'], d['ms_per_step'], 'ax2', k['ax2']['avg_us'], 'ax3', k['ax3']['avg_us'])
"
done
done
done
