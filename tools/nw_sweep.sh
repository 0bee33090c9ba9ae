#!/bin/bash
# developer sweep of the score-mode launch shape of the aligner (tools/nw_probe.py): waves per workgroup x cooperative threshold
for w in ${WAVES:-4 8}; do for c in ${COOP:-16384 65536 131072 262144}; do
echo "waves=$w coop_min_cells=$c"; MDFRI_NW_SCORE_WAVES=$w MDFRI_NW_COOP_MIN_CELLS=$c timeout 300 python tools/nw_probe.py 2>&1 | grep "^k_nw"; done; done
