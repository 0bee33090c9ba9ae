cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT}"
O=gpurun_out/r05_fill_abl; rm -rf $O; mkdir -p $O
for v in 0 1 2 3; do
  export MDFRI_FILL_ABLATE=$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --steps 1 --warmup 1 --proteins 4096 --cpu-seconds 0 --no-extras --no-board --no-kernel-timing --verify 0 > $O/line_$v.json 2>$O/err_$v.txt
  echo "== ablate $v: $(python3 tools/rocprof_summary.py $O/prof_$v | grep -E "cmap_fill")" | tee -a $O/abl.txt
  rm -rf $O/prof_$v
done
