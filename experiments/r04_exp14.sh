#!/bin/bash
set -u
O=gpurun_out/exp14
mkdir -p "$O"
for L in 128 192 256 320 384 448 512 640 768 1024; do
  for v in "MDFRI_AX_MFMA=0" "MDFRI_AX_MFMA=1 MDFRI_AX_C2=1" "MDFRI_AX_MFMA=1 MDFRI_AX_C2=1 MDFRI_AX_RB1=1"; do
    n=$((5000 * 512 / L))
    env $v timeout 300 python3 bench.py --length $L --proteins $n --cpu-seconds 0 --no-extras --steps 2 --verify 0 > "$O/t.json" 2> "$O/t.err"
    python3 -c "
import json
for ln in open('$O/t.json'):
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernels']
        print('L=$L', '$v'.replace('MDFRI_AX_',''), 'step', d['ms_per_step'], 'ax2', k['ax2']['avg_us'], 'ax3', k['ax3']['avg_us'])
"
  done
done
