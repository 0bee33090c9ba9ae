#!/bin/bash
# developer probe: step rate of the default workload against the residue rows per fused chunk
for c in "$@"; do
  timeout 300 python bench.py --cpu-seconds 0 --no-extras --steps 3 --chunk-rows $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print($c, d['value'], d['ms_per_step'], 'gemm', k['gemm']['avg_us'], 'ax', k['ax']['avg_us'], 'gemm1', k['gemm1']['avg_us'], 'cmap', k['cmap']['avg_us'])"
done
