#!/bin/bash
# runs tools/l1ax_probe.py for every built variant of csrc/tools/l1ax_probe.hip
cd "$(dirname "$0")/.."
for f in metagenomic-deepfri_amd/lib/libl1ax_probe*.so; do
  echo "== $(basename $f)"
  LXP_LIB=$(basename $f) timeout 300 python tools/l1ax_probe.py 2>&1 | tail -3
done
