#!/bin/bash
# usage: tools/sweep_arg.sh --flag "v1 v2 ..." [bench args...]  -- runs bench.py once per value of a command-line flag
FLAG=$1; VALS=$2; shift 2
for v in $VALS; do
  python bench.py --cpu-seconds 0 $FLAG $v "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('$FLAG $v', 'value', d['value'], 'ms', d['ms_per_step'], 'gemm_TF', d['roofline']['achieved'] if d['roofline'] else None, 'ax_GBs', d['roofline_ax']['achieved'] if d['roofline_ax'] else None, 'gemm', k.get('gemm'))"
done
