"""Import helper for the compiled reference-side binding (tests/binding/*.pyx -> tests/binding/_build/mdfri_binding/*.so)."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    """Build if needed (cython + g++; no GPU, no hipcc), then import -> (contact_map_utils, predict) compiled modules."""
    sys.path.insert(0, os.path.join(HERE, "binding"))
    import build_binding
    build_binding.build()
    from mDeepFRI import _hip
    _hip.lib()      # loads torch's HIP runtime first (see _hip._preload_torch_hip_runtime), then libmdfri_hip.so by SONAME
    parent = os.path.join(HERE, "binding", "_build")
    if parent not in sys.path:
        sys.path.insert(0, parent)
    return importlib.import_module("mdfri_binding.contact_map_utils"), importlib.import_module("mdfri_binding.predict")
