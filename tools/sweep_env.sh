#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args...]  -- runs bench.py once per value of the environment knob
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --cpu-seconds 0 --steps 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('$VAR=$v', 'chunk', d['config']['chunk_rows'], 'value', d['value'], 'gemm_TF', d['roofline']['achieved'], 'ax_GBs', d['roofline_ax']['achieved'], 'ax_us', k['ax']['avg_us'], 'gemm_us', k['gemm']['avg_us'])"
done
