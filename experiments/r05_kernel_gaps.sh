cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT}"
O=gpurun_out/r05_gaps; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --verify 0 --no-kernel-timing --no-extras --no-board > $O/line.json 2> $O/err.txt
python3 tools/kernel_gaps.py $O/prof 1000 > $O/gaps.txt 2>&1
rm -rf $O/prof
cat $O/gaps.txt
