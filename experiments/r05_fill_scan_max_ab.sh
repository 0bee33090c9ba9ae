cd "${GRAFT_REPO_ROOT}"
for v in 16384 4096 16384 4096; do
MDFRI_FILL_SCAN_MAX=$v timeout 300 python bench.py --steps 5 --cpu-seconds 0 --no-extras > gpurun_out/r05_sc_$v.json 2>gpurun_out/r05_sc_$v.err
python - <<P
import json
d=json.load(open("gpurun_out/r05_sc_$v.json"))
print("$v", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k in("cmap","ax2","ax3","gemm2","gemm3")}, d["config"].get("chunk_rows"))
P
done
