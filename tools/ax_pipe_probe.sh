#!/bin/bash
# runs tools/ax_pipe_probe.py for every built variant of csrc/tools/ax_pipe_probe.hip (see that file's AXP_* macros);
# a variant built with another UMAX carries it in its name (..._u144.so)
cd "$(dirname "$0")/.."
for f in metagenomic-deepfri_amd/lib/libax_pipe_probe*.so; do
  lib=$(basename $f)
  um=128; case $lib in *_u*) um=${lib##*_u}; um=${um%.so};; esac
  echo "== $lib (UMAX $um) AXL=${AXL:-512}"
  AXP_LIB=$lib AXP_UMAX=$um timeout 300 python tools/ax_pipe_probe.py 2>&1 | tail -4
done
