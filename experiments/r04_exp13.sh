#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/exp13
mkdir -p "$O"
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench.py 2>&1 | tail -5
for v in 1 0; do
for w in configs2 configs3 configs4; do
  MDFRI_AX_MFMA=$v timeout 900 python3 bench.py --workload $w --cpu-seconds 0 --no-extras --steps 3 > "$O/${w}_mfma$v.json" 2> "$O/${w}_mfma$v.err"
  python3 -c "
import json,sys
for ln in open('$O/${w}_mfma$v.json'):
    if ln.startswith('{'):
        d=json.loads(ln); k=d['kernels']
        print('$w mfma=$v', d['value'], d['ms_per_step'], {n:k[n]['avg_us'] for n in ('gemm1','ax2','gemm2','ax3','gemm3','cmap') if n in k}, d['roofline_ax']['frac'], d['verify']['max_abs_err_vs_oracle'])
"
done
done
