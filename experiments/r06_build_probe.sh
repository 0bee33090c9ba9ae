#!/bin/bash
# Round 6: the library with the aggregation kernel's in-kernel time stamps compiled in (MDF_AX_PROBE) -> experiments/_r06/probe/libmdfri_hip.so
set -e
cd "$(dirname "$0")/.."
C=metagenomic-deepfri_amd/csrc; O=experiments/_r06/probe; mkdir -p $O
make -s -C $C
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMDF_AX_PROBE ${AX_EXTRA} -c $C/gcn.hip -o $O/gcn.o
B=metagenomic-deepfri_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmdfri_hip.so -o $O/libmdfri_hip.so $B/common.o $B/cmap.o $O/gcn.o $B/output.o $B/cnn.o $B/nw.o $B/engine.o
ls -la $O/libmdfri_hip.so
