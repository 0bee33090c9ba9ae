"""Diagnosis aid: how long does `import torch` take AFTER libmdfri_hip has initialised the GPU (the order of the per-call API used
before the batch API)?  Prints timings; faulthandler dumps the Python stack if it takes longer than 350 s."""
import faulthandler, sys, time
faulthandler.dump_traceback_later(350, exit=True)
sys.path.insert(0, "metagenomic-deepfri_amd")
import numpy as np
from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
t0 = time.time()
pairwise_sqeuclidean(np.zeros((8, 3), np.float32))
print(f"library first GPU call: {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
import torch
print(f"import torch (after the library initialised HIP): {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
torch.zeros(4, device="cuda").sum().item()
print(f"torch first GPU op: {time.time() - t0:.1f} s", flush=True)
