# the heads' GraphConv stacks of a chunk on streams of their own (MDFRI_HEAD_STREAMS=1) against the one-stream form
cd "${GRAFT_REPO_ROOT}"
for v in 0 1 0 1; do
MDFRI_HEAD_STREAMS=$v timeout 300 python bench.py --steps 10 --cpu-seconds 0 --no-extras > gpurun_out/r05_hs_$v.json 2>gpurun_out/r05_hs_$v.err
python - <<P
import json
d=json.load(open("gpurun_out/r05_hs_$v.json"))
print("$v", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k in("cmap","ax2","ax3","gemm2","gemm3")}, d.get("board",{}).get("board_power_w",{}).get("mean"), d["verify"]["max_abs_err_vs_oracle"])
P
done
