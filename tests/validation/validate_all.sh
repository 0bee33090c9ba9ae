#!/bin/bash
# The pin-day kit as ONE command:
#
#     bash tests/validation/validate_all.sh MODEL_DIR          # a directory holding the released DeepFRI-*.onnx files
#     bash tests/validation/validate_all.sh --self-test        # today: this build's own exported graphs (what CI runs), PyOpal skipped
#
# For every {GraphConv (gcn), CNN} x {mf, bp, cc, ec} file it finds under MODEL_DIR it runs validate_release.py (--ort when
# onnxruntime imports): the file's own graph under ONNX semantics / onnxruntime / the oracle on the mapped tensors / the HIP path, every
# pairwise max |delta| against 1e-4, and writes tests/golden/release_<kind>_<mode>.npz -- from then on tests/test_gpu_validation.py
# pins the GPU suite to the released file.  Then validate_opal.py names the aligner's tie rule (PyOpal + VTML80) or skips cleanly.
# The last lines are a table: file, kind, which embedding variant the graph turned out to be, verdict; and the detected TIE_RULE.
# Then the same files once more in a child process under MDFRI_HW_PIPE=f16x3 (the opt-in pipe: does the released model stay inside its activation
# range?) -- printed, informational, the exit code is the default pipe's.
# Exit 0 = every file PASSed (and the aligner check passed or was skipped), 1 = a mismatch, 2 = nothing could be decided.
# Test infrastructure (it uses oracle/ as the checker); nothing here is imported by the product.
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
PY="${PYTHON:-python3}"
if [ $# -lt 1 ]; then echo "usage: $0 MODEL_DIR | --self-test" >&2; exit 2; fi
exec "$PY" "$HERE/validate_all.py" "$@"
