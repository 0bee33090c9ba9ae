# contact-stage kernels: parity tests, then per-kernel durations from a kernel trace (MDFRI_CMAP_FILL=words: the round-4 fill kernel)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT}"
O=gpurun_out/r05_fill; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cmap.py tests/test_gpu_gcn.py tests/test_gpu_engine.py -m gpu -x -q --durations=5 > $O/tests.txt 2>&1; tail -12 $O/tests.txt
for v in ${VARIANTS:-rows words}; do
  export MDFRI_CMAP_FILL=$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --steps 2 --cpu-seconds 0 --no-extras --no-board --no-kernel-timing --verify 0 > $O/line_$v.json 2>$O/err_$v.txt
  echo "== $v"; python3 tools/rocprof_summary.py $O/prof_$v | grep -E "cmap|scan|seq_encode|agg_prepare|calls" | tee -a $O/stats_$v.txt
  rm -rf $O/prof_$v
done
