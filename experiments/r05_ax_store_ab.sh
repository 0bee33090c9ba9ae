# A.X output rows: non-temporal stores (shipped) against the default cache policy -- does the H.W GEMM that reads them next run faster?
cd "${GRAFT_REPO_ROOT}"
for v in nt default nt default; do
MDFRI_AX_STORE=$v timeout 300 python bench.py --steps 10 --cpu-seconds 0 --no-extras > gpurun_out/r05_axst_$v.json 2>gpurun_out/r05_axst_$v.err
python - <<P
import json
d=json.load(open("gpurun_out/r05_axst_$v.json"))
print("$v", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k in("cmap","ax2","ax3","gemm2","gemm3")}, d.get("board",{}).get("board_power_w",{}).get("mean"))
P
done
