import ctypes, sys, os
sys.path.insert(0, "metagenomic-deepfri_amd")
import torch
torch.cuda.init()
L = ctypes.CDLL(os.path.abspath("experiments/_r06/occ/libmdfri_hip.so"))
out = (ctypes.c_int * 8)()
n = L.mdf_debug_ax_occupancy(out)
print("resident workgroups per CU: <1,plain> <2,plain> <4,plain> <1,L1> <2,L1> =", list(out)[:n])
