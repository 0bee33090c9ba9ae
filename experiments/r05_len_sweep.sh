#!/bin/bash
# Round 5: layer-3 aggregation, matrix-pipe form (four row blocks per wave) against the CSR gather for proteins of 800-1 024 residues
# (bench.py --length L, 3 heads, chunks of 65 536 rows): us per launch of ax2 / ax3.  Run from the repository root on an MI355X.
set -u
O=gpurun_out/r05_len_sweep
mkdir -p "$O"
for L in 800 864 928 992 1024; do
  N=$((5000 * 512 / L))
  for M in 1 0; do
    MDFRI_AX_MFMA=$M timeout 200 python3 bench.py --cpu-seconds 0 --no-extras --steps 3 --length $L --proteins $N > "$O/l${L}_m$M.json" 2> "$O/l${L}_m$M.err"
    python3 - "$L" "$M" "$O/l${L}_m$M.json" <<'PY'
import json, sys
for ln in open(sys.argv[3]):
    if ln.startswith('{'):
        d = json.loads(ln); k = d['kernels']
        print('L', sys.argv[1], 'mfma' if sys.argv[2] == '1' else 'gather', 'value', d['value'], {n: k[n]['avg_us'] for n in ('gemm1', 'ax2', 'ax3') if n in k})
        break
PY
  done
done
