"""Build the protein-length histogram behind BASELINE.json configs[4] ("L drawn from the empirical
length histogram of tests/data/GCA_000731455.1.proteins.fa.gz clipped to [30, 2048]").  Runs in the build container only
(reads /root/reference); the committed output is data -- 8-residue bins and their counts -- not reference source."""
import gzip
import json
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/mDeepFRI/tests/data/GCA_000731455.1.proteins.fa.gz"
lens, cur = [], 0
with gzip.open(src, "rt") as f:
    for line in f:
        if line.startswith(">"):
            if cur:
                lens.append(cur)
            cur = 0
        else:
            cur += len(line.strip().rstrip("*"))
if cur:
    lens.append(cur)
a = np.clip(np.array(lens), 30, 2048)
width = 8
edges = np.arange(30, 2048 + width + 1, width)
counts, _ = np.histogram(a, bins=edges)
out = {"source": "mDeepFRI/tests/data/GCA_000731455.1.proteins.fa.gz (reference v1.1.10)", "proteins": int(len(a)), "clip": [30, 2048],
       "bin_width": width, "first_bin_start": 30, "counts": [int(c) for c in counts], "mean_length": float(a.mean())}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mdfri_testkit", "data", "gca_000731455_length_hist.json")
json.dump(out, open(dst, "w"))
print(len(a), "proteins, mean", a.mean(), "->", dst)
