"""Drop-in for the reference's mDeepFRI/contact_map.py (OO wrapper over the distance kernel), same validation
semantics (contact_map.py:12-13, 27, 55-62, 82-86); the arithmetic runs in HIP."""
import numpy as np

from . import _hip
from .contact_map_utils import pairwise_sqeuclidean


class CAlphaCoordinates:
    def __init__(self, structure_id: str, coords: np.ndarray):
        self.structure_id = structure_id
        self.coords = coords
        if coords.shape[1] != 3:
            raise ValueError("Coordinates are not 3D.")

    def calculate_distance_map(self, distance="sqeuclidean"):
        if distance == "sqeuclidean":
            distances = pairwise_sqeuclidean(np.ascontiguousarray(self.coords.astype(np.float32)))
        else:
            raise NotImplementedError("Distance metric not implemented.")
        return DistanceMap(distances)

    def calculate_contact_map(self, threshold=6.0):
        distance_map = self.calculate_distance_map()
        return distance_map.calculate_contacts(threshold**2)


class DistanceMap:
    def __init__(self, distance_map):
        self.distance_map = distance_map
        if not np.all(distance_map >= 0):
            raise ValueError("Distance map contains negative values.")
        if not np.all(np.diag(distance_map) == 0):
            raise ValueError("Distance map diagonal is not zero.")
        if not np.allclose(distance_map, distance_map.T):
            raise ValueError("Distance map is not symmetric.")

    def calculate_contacts(self, threshold):
        """(distance_map < threshold).astype(int32) -- strict '<', compared in the map's dtype (NumPy>=2 rule)."""
        dm = np.ascontiguousarray(self.distance_map)
        if dm.dtype != np.float32:
            # the kernel compares in float32, which is what the reference does for the float32 maps its own
            # distance kernel returns; other dtypes would change the comparison precision
            raise ValueError(f"Buffer dtype mismatch, expected 'float32' but got '{dm.dtype.name}'")
        out = np.empty(dm.shape, dtype=np.int32)
        _hip.check(_hip.lib().mdf_threshold_lt_i32(_hip.ptr(dm), dm.size, np.float32(threshold), _hip.ptr(out)))
        return ContactMap(out)


class ContactMap:
    def __init__(self, cmap):
        self.cmap = cmap
        if not np.allclose(cmap, cmap.T):
            raise ValueError("Contact map is not symmetric.")
        if not np.all(np.isin(cmap, [0, 1])):
            raise ValueError("Contact map values not in range [0, 1].")

    def sparsify(self):
        """np.argwhere(cmap == 1).astype(int32) -- row-major sorted (N,2)."""
        return argwhere_eq1(np.ascontiguousarray(self.cmap, dtype=np.int32))


def argwhere_eq1(cm: np.ndarray) -> np.ndarray:
    """np.argwhere(cm == 1).astype(int32) for a square int32 matrix, on the device."""
    n = cm.shape[0]
    L = _hip.lib()
    cap = max(64 * n, 1024)
    while True:
        pairs = np.empty((cap, 2), dtype=np.int32)
        cnt = _hip.c_int64(0)
        rc = L.mdf_argwhere_eq1_i32(_hip.ptr(cm), n, _hip.ptr(pairs), cap, cnt)
        if rc == _hip.MDF_ECAPACITY:
            cap = int(cnt.value)
            continue
        _hip.check(rc)
        return pairs[:cnt.value].copy()
