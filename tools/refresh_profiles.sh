#!/bin/bash
# Regenerate every measurement committed under profiles/ on an MI355X box (run from the repository root, e.g. through gpurun):
#   bash tools/refresh_profiles.sh            -> writes gpurun_out/refresh/*; copy what changed into profiles/ (names below)
# Counters are collected in passes of their own (rocprofv3 --pmc never together with the trace domains), the program itself after `--`.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
O=gpurun_out/refresh
rm -rf "$O"      # a scratch directory of an earlier call must not leak its traces into the summaries
mkdir -p "$O"
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_bench" -- python3 bench.py --steps 5 --cpu-seconds 0 --no-extras --no-kernel-timing --no-board > "$O/prof_bench.log" 2>&1
{ python3 tools/rocprof_summary.py "$O/prof_bench"; python3 profiles/summarize_rocprof.py layers "$O/prof_bench"; } > "$O/r_kernel_stats.txt"   # -> profiles/rNN_kernel_stats.txt
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc1" -- python3 bench.py --steps 1 --warmup 0 --proteins 2048 --cpu-seconds 0 --verify 0 --no-kernel-timing --no-extras --no-board > "$O/pmc1.log" 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc2" -- python3 bench.py --steps 1 --warmup 0 --proteins 2048 --cpu-seconds 0 --verify 0 --no-kernel-timing --no-extras --no-board > "$O/pmc2.log" 2>&1
{ python3 tools/pmc_summary.py "$O/pmc1"; python3 tools/pmc_summary.py "$O/pmc2"; } > "$O/r_pmc.txt"             # -> profiles/rNN_pmc.txt
python3 tools/make_traffic_json.py "$O/pmc1" "$O/pmc2" "${ROUND:-0}" > "$O/traffic.json"                          # -> profiles/traffic.json (stamped with mdf_version())
cp "$O/traffic.json" profiles/traffic.json     # (the box's copy of the repository: the bench line below then carries `traffic` of this very build)
$T python3 bench.py > "$O/r_bench_line.json" 2> "$O/bench.err"                                                   # -> profiles/rNN_bench_line.json
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_nw" -- python3 tools/nw_probe.py > "$O/nw.log" 2>&1
{ python3 tools/rocprof_summary.py "$O/prof_nw"; grep "host-to-host\|align_queries_arrays\|k_nw<score>\|turned" "$O/nw.log" | grep -v "^ "; } > "$O/r_nw_kernel_stats.txt"   # -> profiles/rNN_nw_kernel_stats.txt
$T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$O/pmc_nw1" -- python3 tools/nw_probe.py > "$O/pmc_nw1.log" 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_nw2" -- python3 tools/nw_probe.py > "$O/pmc_nw2.log" 2>&1
{ python3 tools/pmc_summary.py "$O/pmc_nw1"; python3 tools/pmc_summary.py "$O/pmc_nw2"; } > "$O/r_nw_pmc.txt"   # -> profiles/rNN_nw_pmc.txt
NB=8 $T rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_pipe" -- python3 tools/pipeline_example.py > "$O/pipe.log" 2>&1
{ python3 tools/rocprof_summary.py "$O/prof_pipe"; grep "stream of\|end to end\|  align\|  pack\|  upload\|  GPU filter" "$O/pipe.log"; } > "$O/r_pipeline_kernel_stats.txt"   # -> profiles/rNN_pipeline_kernel_stats.txt
$T python3 tools/format_rate.py > "$O/r_format_rate.txt" 2>&1                                                     # -> profiles/rNN_format_rate.txt
$T python3 tools/small_batch_rate.py > "$O/r_small_batch_rate.json" 2> "$O/small.err"                             # -> profiles/rNN_small_batch_rate.json
$T python3 bench.py --workload configs3 --cpu-seconds 0 --steps 2 > "$O/r_bench_configs3_n1.json" 2> "$O/c3.err"  # -> profiles/rNN_bench_configs3_n1.json
timeout 900 python3 bench.py --workload configs4 --cpu-seconds 0 --steps 2 > "$O/r_bench_configs4_n1.json" 2> "$O/c4.err"
# the headline step once more under the opt-in pipe (kernel trace + one counter pass)                                                  # -> profiles/rNN_f16x3_kernel_stats.txt
export MDFRI_HW_PIPE=f16x3
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_f16x3" -- python3 bench.py --steps 5 --cpu-seconds 0 --no-extras --no-kernel-timing --no-board > "$O/prof_f16x3.log" 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_f16x3" -- python3 bench.py --steps 1 --warmup 0 --proteins 2048 --cpu-seconds 0 --verify 0 --no-kernel-timing --no-extras --no-board > "$O/pmc_f16x3.log" 2>&1
unset MDFRI_HW_PIPE
{ python3 tools/rocprof_summary.py "$O/prof_f16x3"; python3 tools/pmc_summary.py "$O/pmc_f16x3"; } > "$O/r_f16x3_kernel_stats.txt"
ls -la "$O"/r_*
