"""Object interface over the distance / threshold / argwhere kernels, with the public names and the observable behaviour of the
reference's mDeepFRI/contact_map.py (CAlphaCoordinates, DistanceMap, ContactMap; validation messages and exception types as
at contact_map.py:12-13, 27, 55-62, 82-86).  Every array operation runs in HIP through libmdfri_hip; nothing is computed
with NumPy on the host except the cheap input validation the reference also does."""
import numpy as np

from . import _hip
from .contact_map_utils import pairwise_sqeuclidean


def _reject_unless(ok, message: str):
    if not ok:
        raise ValueError(message)


def _mirror_image_equal(a) -> bool:
    return bool(np.allclose(a, np.transpose(a)))


def _sqeuclidean(coords):
    # the kernel takes a C-contiguous float32 (n, 3) buffer; the reference casts with astype(float32) at the same spot
    return pairwise_sqeuclidean(np.ascontiguousarray(coords, dtype=np.float32))


_METRICS = {"sqeuclidean": _sqeuclidean}


def argwhere_eq1(cm: np.ndarray) -> np.ndarray:
    """np.argwhere(cm == 1).astype(int32) of a square int32 matrix, computed on the device (row-major order).  The output
    capacity is guessed at 64 entries per row and grown once to the exact count the kernel reports if that was short."""
    n = cm.shape[0]
    lib = _hip.lib()
    capacity = max(64 * n, 1024)
    for _ in range(2):
        pairs = np.empty((capacity, 2), dtype=np.int32)
        found = _hip.c_int64(0)
        rc = lib.mdf_argwhere_eq1_i32(_hip.ptr(cm), n, _hip.ptr(pairs), capacity, found)
        if rc != _hip.MDF_ECAPACITY:
            _hip.check(rc)
            return pairs[:found.value].copy()
        capacity = int(found.value)
    _hip.check(rc)


class ContactMap:
    """Symmetric 0/1 matrix; `sparsify()` lists its ones as (row, col) pairs."""

    def __init__(self, cmap):
        self.cmap = cmap
        _reject_unless(_mirror_image_equal(cmap), "Contact map is not symmetric.")
        _reject_unless(np.isin(cmap, (0, 1)).all(), "Contact map values not in range [0, 1].")

    def sparsify(self):
        return argwhere_eq1(np.ascontiguousarray(self.cmap, dtype=np.int32))


class DistanceMap:
    """Non-negative symmetric matrix with a zero diagonal; `calculate_contacts(t)` thresholds it with a strict '<'."""

    def __init__(self, distance_map):
        self.distance_map = distance_map
        _reject_unless((distance_map >= 0).all(), "Distance map contains negative values.")
        _reject_unless((np.diagonal(distance_map) == 0).all(), "Distance map diagonal is not zero.")
        _reject_unless(_mirror_image_equal(distance_map), "Distance map is not symmetric.")

    def calculate_contacts(self, threshold):
        # reference contact_map.py:74: `(distance_map < threshold).astype(np.int32)` for a map of ANY dtype.  NumPy >= 2
        # gives a Python-float threshold the dtype of a floating-point map (float32 for the maps the distance kernel
        # returns, float16 -> the threshold is rounded to float16) and compares integer / bool maps in float64.  The device
        # has a float32 and a float64 comparison; float16 / integer values are exact in float64.
        d = np.asarray(self.distance_map)
        if d.dtype.kind not in "fiub":
            raise TypeError(f"'<' not supported between a distance map of dtype {d.dtype} and a threshold")
        flags = np.empty(d.shape, dtype=np.int32)
        if d.dtype == np.float32:
            d = np.ascontiguousarray(d)
            _hip.check(_hip.lib().mdf_threshold_lt_i32(_hip.ptr(d), d.size, np.float32(threshold), _hip.ptr(flags)))
        else:
            thr = float(d.dtype.type(threshold)) if d.dtype.kind == "f" and isinstance(threshold, (int, float)) else float(threshold)
            d = np.ascontiguousarray(d, dtype=np.float64)
            _hip.check(_hip.lib().mdf_threshold_lt_f64_i32(_hip.ptr(d), d.size, thr, _hip.ptr(flags)))
        return ContactMap(flags)


class CAlphaCoordinates:
    """C-alpha trace of one structure: (L, 3) coordinates -> DistanceMap -> ContactMap."""

    def __init__(self, structure_id: str, coords: np.ndarray):
        self.structure_id, self.coords = structure_id, coords
        _reject_unless(coords.shape[1] == 3, "Coordinates are not 3D.")

    def calculate_distance_map(self, distance="sqeuclidean"):
        metric = _METRICS.get(distance)
        if metric is None:
            raise NotImplementedError("Distance metric not implemented.")
        return DistanceMap(metric(self.coords))

    def calculate_contact_map(self, threshold=6.0):
        return self.calculate_distance_map().calculate_contacts(threshold**2)
