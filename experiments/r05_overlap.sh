#!/bin/bash
# Round 5, VERDICT r4 #1: the BF16x6 GEMM's issue schedule against the power limit.  Run from the repository root on an MI355X
# (binaries built by `make -C experiments overlap`); writes gpurun_out/r05_overlap/*.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
O=gpurun_out/r05_overlap
mkdir -p "$O"
B=experiments/bin/gemm_overlap_probe
sample() { for i in $(seq 1 ${1:-6}); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done; }
{
echo "== micro"; timeout 300 $B micro
echo "== gemm (subtractions as inline asm: an s_nop behind each group)"; timeout 300 $B gemm
echo "== gemm (plain subtractions, -fno-slp-vectorize: no s_nop)"; timeout 300 ${B}_plain gemm
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
echo "== idle"; sample 2
for f in 0 1 2 3; do
  echo "== soak form $f (plain build), 6000 launches"; timeout 120 ${B}_plain soak $f 6000 > /tmp/soak$f.txt & sleep 2.0; sample 4; wait; cat /tmp/soak$f.txt
done
} > "$O/overlap.txt" 2>&1
# counters: matrix-pipe busy cycles and active cycles per kernel, one pass, program itself after `--`
for f in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$O/pmc$f" -- ${B}_plain soak $f 40 > "$O/pmc$f.log" 2>&1
  python3 tools/pmc_summary.py "$O/pmc$f" >> "$O/overlap_pmc.txt" 2>&1
done
cat "$O/overlap.txt"; cat "$O/overlap_pmc.txt"
