#!/bin/bash
# Round 5: where the two matrix-pipe aggregation launches spend their wave cycles (counter passes of their own; program itself after `--`).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
O=gpurun_out/r05_ax_pmc
mkdir -p "$O"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|TA_[A-Z_0-9]+)\b" | sort -u > "$O/counters.txt"
wc -l "$O/counters.txt"
B="python3 bench.py --steps 1 --warmup 0 --proteins 2048 --cpu-seconds 0 --verify 0 --no-kernel-timing --no-extras --no-board"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$O/p$i" -- $B > "$O/p$i.log" 2>&1
  python3 tools/pmc_summary.py "$O/p$i" 2>&1 | grep -E "^#|^kernel|k_aggregate|k_gemm_bf16x6|k_layer1" >> "$O/ax_pmc.txt"
done
cat "$O/ax_pmc.txt"
