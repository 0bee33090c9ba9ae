#!/bin/bash
# round 4, session 1: does the layer-3 A.X gather find H2 in the Infinity Cache?  
#   A_NT bit 0: layer-2 GEMM reads its A operand (AH) non-temporally; bit 1: the same for the layer-3 GEMM
#   AX_NT: non-temporal output stores of the A.X kernel (default: on above 200 MiB of input + output)
set -u
O=gpurun_out/exp1
mkdir -p "$O"
digest() { python3 -c "
import json,sys
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    d=json.loads(ln); k=d['kernels']
    print('$1', 'value', d['value'], 'ms', d['ms_per_step'], {n: k[n]['avg_us'] for n in ('gemm1','ax2','gemm2','ax3','gemm3','cmap') if n in k}, 'verify', d.get('verify',{}).get('max_abs_err_vs_oracle'))
"; }
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --no-extras --steps 4 $EXTRA 2>"$O/$name.err" | tee "$O/$name.json" | digest "$name"; }
EXTRA=""
run base X=0
run a_nt1 MDFRI_GEMM_A_NT=1
run a_nt3 MDFRI_GEMM_A_NT=3
run a_nt2 MDFRI_GEMM_A_NT=2
run axnt0 MDFRI_AX_NT=0
run axnt0_a1 MDFRI_AX_NT=0 MDFRI_GEMM_A_NT=1
run axnt0_a3 MDFRI_AX_NT=0 MDFRI_GEMM_A_NT=3
EXTRA="--chunk-rows 32768"
run c32k X=0
run c32k_axnt1 MDFRI_AX_NT=1
run c32k_a3 MDFRI_GEMM_A_NT=3
run c32k_axnt1_a3 MDFRI_AX_NT=1 MDFRI_GEMM_A_NT=3
EXTRA="--chunk-rows 131072"
run c128k X=0
run c128k_a1 MDFRI_GEMM_A_NT=1
run c128k_a3 MDFRI_GEMM_A_NT=3
