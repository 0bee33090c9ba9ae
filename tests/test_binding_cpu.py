"""The reference-side binding of INTEGRATION.md section B is real code: tests/binding/{contact_map_utils,predict}.pyx are
cythonized, compiled against include/mdfri.h and linked with libmdfri_hip.so (no hipcc, no GPU needed for that), and the compiled
modules expose the reference's signatures (contact_map_utils.pyx:17,44 incl. `threads`; predict.pyx:17,50-60,75)."""
import inspect
import subprocess

import numpy as np
import pytest

from mDeepFRI import _hip


@pytest.fixture(scope="module")
def compiled():
    import binding_loader
    return binding_loader.load()


def test_compiled_modules_link_against_the_c_abi(compiled):
    cmu, pr = compiled
    for mod in (cmu, pr):
        needed = subprocess.run(["readelf", "-d", mod.__file__], capture_output=True, text=True).stdout
        assert "libmdfri_hip.so" in needed
    assert callable(cmu.pairwise_sqeuclidean) and callable(cmu.align_contact_map) and callable(pr.seq2onehot)
    P = pr.Predictor
    assert {"model_path", "threads", "session", "input_names"} <= set(dir(P)) and callable(P.forward_pass) and callable(P._load_model)


def test_argument_binding_is_cythons_own(compiled):
    """Wrong dtype / layout are rejected by the Cython buffer protocol exactly as in the reference build (ValueError with the
    'Buffer dtype mismatch' text), before anything reaches the library."""
    cmu, pr = compiled
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        cmu.pairwise_sqeuclidean(np.zeros((3, 3), dtype=np.float64))
    with pytest.raises(ValueError, match="ndarray is not C-contiguous"):
        cmu.pairwise_sqeuclidean(np.asfortranarray(np.zeros((3, 4), dtype=np.float32)))
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        cmu.align_contact_map("AB", "AB", np.array([[0, 1]], dtype=np.int64))
    with pytest.raises(TypeError):
        pr.seq2onehot(b"ACD")
    assert pr.seq2onehot("").shape == (0, 26)
    # keyword `threads` of the .pyx signatures (the .pyi omits it)
    if _hip.device_count() == 0:
        with pytest.raises(RuntimeError, match="no HIP device"):
            cmu.pairwise_sqeuclidean(np.zeros((3, 3), dtype=np.float32), threads=2)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            cmu.align_contact_map("AB", "AB", np.array([[0, 1]], dtype=np.int32), generated_contacts=2, threads=1)
        with pytest.raises(RuntimeError, match="no HIP device"):
            pr.seq2onehot("ACD")
