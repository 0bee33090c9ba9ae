# A.X walking its protein list from the end (MDFRI_AX_REVERSE=1) against from the start, at the default chunk size
cd "${GRAFT_REPO_ROOT}"
for v in 1 0 1 0; do
MDFRI_AX_REVERSE=$v timeout 300 python bench.py --steps 5 --cpu-seconds 0 --no-extras > gpurun_out/r05_axrev_$v.json 2>gpurun_out/r05_axrev_$v.err
python - <<P
import json
d=json.load(open("gpurun_out/r05_axrev_$v.json"))
print("$v", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k in("cmap","ax2","ax3","gemm2","gemm3")}, d["roofline_ax"]["frac"], d["verify"]["max_abs_err_vs_oracle"])
P
done
