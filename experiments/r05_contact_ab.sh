cd "${GRAFT_REPO_ROOT}"
for i in 1 2; do
timeout 300 python bench.py --steps 10 --cpu-seconds 0 --no-extras > gpurun_out/r05_b_$i.json 2>gpurun_out/r05_b_$i.err
python - <<P
import json
d=json.load(open("gpurun_out/r05_b_$i.json"))
print(d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k in("cmap","ax2","ax3","gemm2","gemm3")}, d.get("board",{}).get("board_power_w"))
P
done
MDFRI_CMAP_ROWS=old MDFRI_CMAP_FILL=words timeout 300 python bench.py --steps 10 --cpu-seconds 0 --no-extras > gpurun_out/r05_b_old.json 2>gpurun_out/r05_b_old.err
python - <<P
import json
d=json.load(open("gpurun_out/r05_b_old.json"))
print("old", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if k in("cmap","ax2","ax3","gemm2","gemm3")})
P
