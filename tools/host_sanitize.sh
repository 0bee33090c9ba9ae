#!/bin/bash
# The HOST code of the library (planner, aligner plans / orientation, text formatters, argument checks) under AddressSanitizer + UBSan on a
# box WITHOUT a GPU: builds a sanitized libmdfri_hip.so into /tmp (device code is compiled as usual, -fno-gpu-sanitize) and runs the CPU tests
# that call into it (TESTS="..." selects others).  It stays a CPU tool: on a GPU box the HSA runtime itself reads uninitialised memory under
# ASan's malloc fill and dies inside libhsa-runtime64.so before the library is reached; GPU sanitizers are not available on this pool.
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/mdfri_asan}"
mkdir -p "$OUT"
cd "$OUT"
for f in common.cpp cmap.hip gcn.hip output.hip cnn.hip nw.hip engine.hip; do
    x=""; [ "$f" = "common.cpp" ] && x="-x hip"
    c=""; [ "$f" = "cmap.hip" ] && c="-ffp-contract=off"
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
        -I"$ROOT/include" $c $x -c "$ROOT/metagenomic-deepfri_amd/csrc/$f" -o "${f%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o libmdfri_hip.so ./*.o
ASAN_RT="$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)"
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$ASAN_RT" MDFRI_HIP_LIB="$OUT/libmdfri_hip.so" \
    python -m pytest ${TESTS:-tests/test_output_format_cpu.py tests/test_aligner_host_cpu.py tests/test_engine_plan_cpu.py tests/test_abi_cpu.py} -x -q
