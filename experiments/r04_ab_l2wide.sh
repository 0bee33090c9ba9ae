# A/B on one box: the layer-2 length classes of the matrix-pipe aggregation, narrow (the rule tuned for the two-kernel form) against wide (the
# layer-3 rule), with layer 1 made inside the layer-2 launch.  MDFRI_AGG_L2_WIDE existed for this probe only.
cd /root/repo
for w in 0 1 0 1; do export MDFRI_AGG_L2_WIDE=$w
python3 bench.py --workload configs3 --cpu-seconds 0 --steps 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w configs3', d['value'])"
python3 bench.py --workload mixed --cpu-seconds 0 --steps 2 --no-extras | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w mixed', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('gemm1','ax2','ax3')})"
done
for w in 0 1; do export MDFRI_AGG_L2_WIDE=$w
for L in 128 320 640 1024; do python3 bench.py --cpu-seconds 0 --no-extras --length $L --proteins $((5120000/L)) --steps 2 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w L=$L', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('gemm1','ax2','ax3')})"; done; done
MDFRI_AGG_L2_WIDE=1 python3 bench.py --workload configs4 --cpu-seconds 0 --steps 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=1 configs4', d['value'])"
MDFRI_AGG_L2_WIDE=0 python3 bench.py --workload configs4 --cpu-seconds 0 --steps 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=0 configs4', d['value'])"
