#!/bin/bash
# Board power and shader clock while one kernel runs back to back (rocm-smi sampled every 0.5 s): the fp32 GEMM, the shipped BF16x6 form (in-register
# split, experiments/bin/gemm_split_probe with -DNPROD=6) and the pre-split form (experiments/bin/gemm_split2_probe).  Run from the repository root on an MI355X.
sample() { for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done; }
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
echo "== idle"; sample | head -2
echo "== in-register BF16x6 probe (6 products), 4000 launches"; experiments/bin/gemm_split_probe_n6 65536 4000 > /tmp/p1.txt & sleep 2.5; sample; wait; grep "split kernel" /tmp/p1.txt
echo "== pre-split probe (9, 8, 6 products), 1500 launches each"; experiments/bin/gemm_split2_probe 65536 1500 > /tmp/p2.txt & sleep 2.5; sample; sample; wait; grep "last launch" /tmp/p2.txt
