#!/bin/bash
# Round 5, after the aggregation kernel's instruction diet: layer 3 on the matrix pipe (four row blocks per wave, MDFRI_AX_L3_MAX=1024) against the
# gather (default: above 832 residues) for 800-1 024 residues, and the fused layer-2 classes against gather + k_layer1 at 256 / 320 / 512 / 640 / 768.
set -u
O=gpurun_out/r05_len_sweep2
mkdir -p "$O"
run() {  # name, L, env...
  local name=$1 L=$2; shift 2
  local N=$((5000 * 512 / L))
  env "$@" timeout 200 python3 bench.py --cpu-seconds 0 --no-extras --no-board --steps 3 --length $L --proteins $N > "$O/$name.json" 2> "$O/$name.err"
  python3 - "$name" "$O/$name.json" <<'PY'
import json, sys
for ln in open(sys.argv[2]):
    if ln.startswith('{'):
        d = json.loads(ln); k = d['kernels']
        print(sys.argv[1], 'value', d['value'], {n: k[n]['avg_us'] for n in ('gemm1', 'ax2', 'ax3') if n in k})
        break
else:
    print(sys.argv[1], 'FAILED')
PY
}
for L in 800 864 928 992 1024; do
  run "l${L}_mfma" $L MDFRI_AX_L3_MAX=1024
  run "l${L}_default" $L MDFRI_AX_L3_MAX=832
done
for L in 128 192 256 320 384 512 640 768; do
  run "l${L}_default" $L MDFRI_AX_L3_MAX=832
  run "l${L}_gather" $L MDFRI_AX_MFMA=0
done
