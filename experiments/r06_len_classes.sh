#!/bin/bash
# Round 6: the aggregation's length classes re-measured with the round-6 kernels.  Three builds of the library -- the shipped rule, layer 1 made
# inside the layer-2 launch for every matrix-pipe protein of at most 512 residues, the matrix pipe for every length 112..1024 (no gaps above the
# multiples of 256) -- side by side per length (tools/ax_ab.py: ax2 + gemm1 = layer 1 + layer-2 aggregation, ax3 = layer 3, ms per pass).
set -e
cd "$(dirname "$0")/.."
C=metagenomic-deepfri_amd/csrc; B=metagenomic-deepfri_amd/lib/obj
build() {   # name, extra flags
    O=experiments/_r06/$1; mkdir -p $O
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "${@:2}" -c $C/gcn.hip -o $O/gcn.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmdfri_hip.so -o $O/libmdfri_hip.so $B/common.o $B/cmap.o $O/gcn.o $B/output.o $B/cnn.o $B/nw.o $B/engine.o
}
if [ "$1" = build ]; then
    make -s -C $C
    build cls_fuse_all "-DMDF_AGG_FUSED_EXPERIMENT(L)=((L)<=512)"
    build cls_nogap "-DMDF_AGG_CLASS_EXPERIMENT(L)=((L)<=256?0:(L)<=512?1:2)" "-DMDF_AGG_FUSED_EXPERIMENT(L)=((L)<=512)"
    exit 0
fi
AX_AB_ROUNDS=1 AX_AB_SHAPES=${SHAPES:-L128,L160,L192,L224,L272,L288,L320,L352,L384,L416,L528,L544,L640,L768,L896,L1024} \
    python tools/ax_ab.py metagenomic-deepfri_amd/lib/libmdfri_hip.so experiments/_r06/cls_fuse_all/libmdfri_hip.so experiments/_r06/cls_nogap/libmdfri_hip.so
