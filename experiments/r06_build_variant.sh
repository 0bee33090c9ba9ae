#!/bin/bash
# Round 6: a build of the library with extra -D flags on gcn.hip -> experiments/_r06/NAME/libmdfri_hip.so (A/B with tools/ax_ab.py)
#   bash experiments/r06_build_variant.sh prio1 -DMDF_AX_PRIO=1
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
C=metagenomic-deepfri_amd/csrc; O=experiments/_r06/$NAME; mkdir -p $O
make -s -C $C
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $C/gcn.hip -o $O/gcn.o
B=metagenomic-deepfri_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmdfri_hip.so -o $O/libmdfri_hip.so $B/common.o $B/cmap.o $O/gcn.o $B/output.o $B/cnn.o $B/nw.o $B/engine.o
ls -la $O/libmdfri_hip.so
