#!/bin/bash
# runs tools/ax_pipe_probe.py for every built variant of csrc/tools/ax_pipe_probe.hip (see that file's AXP_* macros)
cd "$(dirname "$0")/.."
for v in "" _abl1 _abl2 _abl3 _nst2 _nst3u128 _nst4u128 _nst5u104; do
  lib=libax_pipe_probe$v.so
  [ -f metagenomic-deepfri_amd/lib/$lib ] || continue
  um=176; case $v in *u128) um=128;; *u104) um=104;; esac
  echo "== $lib (UMAX $um)"
  AXP_LIB=$lib AXP_UMAX=$um timeout 300 python tools/ax_pipe_probe.py 2>&1 | tail -4
done
