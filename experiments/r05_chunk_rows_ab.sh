# residue rows per chunk: 65 536 (rounds 1-5 default) against larger chunks, on the workloads of the bench
cd "${GRAFT_REPO_ROOT}"
run() { timeout 600 python bench.py --cpu-seconds 0 --no-extras --steps 2 --verify 2 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$*', '->', d['value'], d['ms_per_step'], 'gemm', k['gemm']['avg_us'], 'ax', k['ax']['avg_us'], 'cmap', k['cmap']['avg_us'], d['verify'].get('max_abs_err_vs_oracle'))"; }
for c in 65536 262144; do
  run --chunk-rows $c --workload mixed
  run --chunk-rows $c --length 256
  run --chunk-rows $c --length 1024
  run --chunk-rows $c --workload configs3
  run --chunk-rows $c --lm
done
