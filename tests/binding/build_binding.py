#!/usr/bin/env python3
"""Compile the two Cython modules of tests/binding/ against include/mdfri.h and link them with libmdfri_hip.so: the
"reference-side binding" of INTEGRATION.md section B as a real build (what a maintainer's setup.py would do with
include_dirs / libraries=["mdfri_hip"] / runtime_library_dirs).  Output: tests/binding/_build/mdfri_binding/{contact_map_utils,
predict}*.so (git-ignored, travels to the GPU box with the snapshot).  Needs cython + g++ only -- hipcc is not involved: the
boundary is a plain C ABI."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "mdfri_binding")
LIB_DIR = os.path.join(ROOT, "metagenomic-deepfri_amd", "lib")
MODULES = ("contact_map_utils", "predict")


def so_path(name):
    return os.path.join(OUT, name + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force: bool = False):
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    init = os.path.join(OUT, "__init__.py")
    if not os.path.exists(init):
        open(init, "w").write('"""compiled by tests/binding/build_binding.py"""\n')
    lib = os.path.join(LIB_DIR, "libmdfri_hip.so")
    if not os.path.exists(lib):
        raise FileNotFoundError(f"{lib}: build the HIP library first (make -C metagenomic-deepfri_amd/csrc)")
    inc = sysconfig.get_paths()["include"]
    for name in MODULES:
        pyx, so = os.path.join(HERE, name + ".pyx"), so_path(name)
        if os.path.exists(so) and not force and os.path.getmtime(so) >= max(os.path.getmtime(pyx), os.path.getmtime(os.path.join(ROOT, "include", "mdfri.h"))):
            continue
        cpp = os.path.join(OUT, name + ".cpp")
        subprocess.check_call([sys.executable, "-m", "cython", "-3", "--cplus", "-o", cpp, pyx])
        try:
            subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++14", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                                   "-I", inc, "-I", np.get_include(), "-I", os.path.join(ROOT, "include"), cpp, "-o", so,
                                   "-L", LIB_DIR, "-lmdfri_hip", f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,$ORIGIN/../../../../metagenomic-deepfri_amd/lib"])
        finally:
            if os.path.exists(cpp):
                os.remove(cpp)
    return [so_path(n) for n in MODULES]


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv)))
