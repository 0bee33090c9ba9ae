#!/usr/bin/env python3
"""Build the REAL reference contact-map kernels into oracle/_ref/ (test infrastructure only).

TEST INFRASTRUCTURE -- never imported by the product package.

The reference's `mDeepFRI/contact_map_utils.pyx` is Cython -> C++ (reference `setup.py:258-275`).
This recipe cythonizes it *where it lies* under /root/reference and compiles the generated C++
with the reference's own flags (`setup.py:241-242`: `-O3 -fopenmp`, language c++, no -march,
no -ffast-math), writing ONLY into oracle/_ref/ (git-ignored).  The generated .cpp is deleted
after compilation so that no transcription of reference source stays in the tree; only the
compiled extension `oracle/_ref/ref_contact_map_utils*.so` remains.

Not built: `mDeepFRI/predict.pyx`.  It does `import onnxruntime` at module level
(reference `predict.pyx:10`) and onnxruntime is absent from this image; building/importing it
would need a stand-in for that library, so it is treated as unbuildable here.  `seq2onehot`
is pinned instead against the reference's own known-answer tests
(`mDeepFRI/tests/test_predict.py:9-33`) in tests/test_oracle_golden.py.

Usage:  python oracle/build_ref.py            (no-op when /root/reference is absent)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("MDF_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
MODULE = "ref_contact_map_utils"


def ref_so_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return os.path.join(OUT_DIR, MODULE + suffix)


def build(force: bool = False) -> str | None:
    pyx = os.path.join(REF_ROOT, "mDeepFRI", "contact_map_utils.pyx")
    if not os.path.exists(pyx):
        return None  # GPU box: reference tree is not present; use the prebuilt .so if it travelled
    so = ref_so_path()
    if os.path.exists(so) and not force and os.path.getmtime(so) >= os.path.getmtime(pyx):
        return so
    import numpy as np

    os.makedirs(OUT_DIR, exist_ok=True)
    cpp = os.path.join(OUT_DIR, MODULE + ".cpp")
    # compiler directives as in reference setup.py:205-215 (boundscheck etc. are set per function in the .pyx)
    subprocess.check_call([
        sys.executable, "-m", "cython", "-3", "--cplus", "--module-name", MODULE,
        "-X", "language_level=3", "-o", cpp, pyx
    ])
    inc = sysconfig.get_paths()["include"]
    try:
        subprocess.check_call([
            "g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-std=c++14",
            "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
            "-I", inc, "-I", np.get_include(), cpp, "-o", so, "-fopenmp", "-lstdc++"
        ])
    finally:
        if os.path.exists(cpp):
            os.remove(cpp)
    return so


def load():
    """Import the compiled reference module, or return None when it is not available."""
    so = ref_so_path()
    if not os.path.exists(so):
        return None
    import importlib.util

    spec = importlib.util.spec_from_file_location(MODULE, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print("reference build:", p if p else "skipped (no /root/reference)")
