cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT}"
O=gpurun_out/r05_lm_trace; rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --lm --steps 1 --warmup 1 --cpu-seconds 0 --verify 0 --no-kernel-timing --no-extras --no-board > $O/line.json 2> $O/err.txt
python3 tools/kernel_overlap.py $O/prof 'void mdf::k_gemm_bf16x6<6' 'void mdf::k_gemm_bf16x6<7' > $O/overlap.txt 2>&1
python3 tools/kernel_overlap.py $O/prof 'k_gemm_bf16x6<6' 'k_gemm_bf16x6<7' 'mdf::k_gemm_bf16x6<(mdf::Epilogue)6' 'mdf::k_gemm_bf16x6<(mdf::Epilogue)7' >> $O/overlap.txt 2>&1
python3 tools/rocprof_summary.py $O/prof > $O/stats.txt 2>&1
find $O/prof -name "*kernel_trace.csv" | head -2 >> $O/overlap.txt
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); cut -d, -f8 "$f" | sort | uniq -c | sort -rn | head -20 >> $O/overlap.txt
rm -rf $O/prof
