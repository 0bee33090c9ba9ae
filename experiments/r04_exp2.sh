#!/bin/bash
# round 4, session 2: Infinity-Cache policy probe + store-policy knobs of the GraphConv chain
set -u
O=gpurun_out/exp2
mkdir -p "$O"
timeout 300 experiments/bin/mall_probe > "$O/mall_probe.txt" 2>&1
digest() { python3 -c "
import json,sys
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    d=json.loads(ln); k=d['kernels']
    print('$1', 'value', d['value'], 'ms', d['ms_per_step'], {n: k[n]['avg_us'] for n in ('gemm1','ax2','gemm2','ax3','gemm3','cmap') if n in k}, 'verify', d.get('verify',{}).get('max_abs_err_vs_oracle'))
"; }
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --no-extras --steps 4 --verify 0 $EXTRA 2>"$O/$name.err" | tee "$O/$name.json" | digest "$name"; }
EXTRA=""
run base X=0
run ax_nostore MDFRI_AX_NT=2
run c_nt1 MDFRI_GEMM_C_NT=1
run c_nt2 MDFRI_GEMM_C_NT=2
run c_nt3 MDFRI_GEMM_C_NT=3
run c_nt2_a1 MDFRI_GEMM_C_NT=2 MDFRI_GEMM_A_NT=1
run c_nt1_a2 MDFRI_GEMM_C_NT=1 MDFRI_GEMM_A_NT=2
cat "$O/mall_probe.txt"
