#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/exp9
mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- python3 bench.py --steps 3 --cpu-seconds 0 --no-extras --no-kernel-timing > "$O/prof.log" 2>&1
python3 tools/rocprof_summary.py "$O/prof" | head -16
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof3" -- python3 bench.py --workload configs3 --proteins 20000 --steps 2 --cpu-seconds 0 --no-extras --no-kernel-timing > "$O/prof3.log" 2>&1
python3 tools/rocprof_summary.py "$O/prof3" | head -16
