#!/bin/bash
# usage: experiments/bench_variants.sh OUTDIR [bench args] -- then lines "name VAR=val VAR=val ..." on stdin: one short bench.py run per
# line under those environment knobs, one digest line each (per-layer kernel times).  Run from the repository root (gpurun does).
set -u
O=$1; shift
mkdir -p "$O"
while read -r name envs; do
  [ -z "$name" ] && continue
  env $envs timeout 300 python3 bench.py --cpu-seconds 0 --no-extras --steps 4 "$@" 2>"$O/$name.err" > "$O/$name.json"
  python3 - "$name" "$O/$name.json" <<'PY'
import json, sys
for ln in open(sys.argv[2]):
    if ln.startswith('{'):
        d = json.loads(ln); k = d['kernels']
        print(sys.argv[1], 'value', d['value'], 'ms', d['ms_per_step'], {n: k[n]['avg_us'] for n in ('gemm1', 'ax2', 'gemm2', 'ax3', 'gemm3', 'cmap') if n in k},
              'verify', d.get('verify', {}).get('max_abs_err_vs_oracle'))
        break
else:
    print(sys.argv[1], 'FAILED', open(sys.argv[2].replace('.json', '.err')).read()[-400:])
PY
done
