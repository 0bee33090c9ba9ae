"""The two hot-path functions of the reference's mDeepFRI/bio_utils.py (calculate_contact_map :196-227,
build_align_contact_map :348-385).  Structure parsing / FoldComp access in that file are out of scope."""
from __future__ import annotations

import logging
from typing import Tuple

import numpy as np

from . import _hip
from .contact_map_utils import pairwise_sqeuclidean

logger = logging.getLogger(__name__)


def _coords_f32(coordinates) -> np.ndarray:
    # the reference hands the array straight to a `float[:, ::1]` argument: float32, 2-D, C-contiguous or ValueError
    if not isinstance(coordinates, np.ndarray) or coordinates.dtype != np.float32:
        raise ValueError(f"Buffer dtype mismatch, expected 'float32' but got '{getattr(coordinates, 'dtype', type(coordinates))}'")
    if coordinates.ndim != 2 or not coordinates.flags.c_contiguous:
        raise ValueError("coordinates must be a C-contiguous 2-D array")
    return coordinates


def calculate_contact_map(coordinates: np.ndarray, threshold=6.0, distance="sqeuclidean", mode="matrix") -> np.ndarray:
    """reference bio_utils.py:196-227: `(pairwise_sqeuclidean(coords) < threshold**2).astype(int32)`, optionally
    `np.argwhere(cmap == 1).astype(int32)`.  Fused on the device: the distance matrix never reaches the host."""
    if distance != "sqeuclidean":
        raise KeyError(distance)  # reference: distance_functions[distance]
    X = _coords_f32(coordinates)
    n = X.shape[0]
    L = _hip.lib()
    if X.shape[1] != 3:
        # generic-dimension input: unfused route through the same kernels
        D = pairwise_sqeuclidean(X)
        cm = np.empty((n, n), dtype=np.int32)
        _hip.check(L.mdf_threshold_lt_i32(_hip.ptr(D), D.size, np.float32(float(threshold)**2), _hip.ptr(cm)))
        if mode != "sparse":
            return cm
        from .contact_map import argwhere_eq1
        return argwhere_eq1(cm)
    cnt = _hip.c_int64(0)
    if mode == "sparse":
        cap = max(64 * n, 1024)
        while True:
            pairs = np.empty((cap, 2), dtype=np.int32)
            rc = L.mdf_calculate_contact_map(_hip.ptr(X), n, float(threshold), None, _hip.ptr(pairs), cap, cnt)
            if rc == _hip.MDF_ECAPACITY:
                cap = int(cnt.value)
                continue
            _hip.check(rc)
            return pairs[:cnt.value].copy()
    cm = np.empty((n, n), dtype=np.int32)
    _hip.check(L.mdf_calculate_contact_map(_hip.ptr(X), n, float(threshold), _hip.ptr(cm), None, 0, cnt))
    return cm


def build_align_contact_map(alignment, threshold: float = 6, generated_contacts: int = 2) -> Tuple[object, np.ndarray]:
    """reference bio_utils.py:348-385.  `alignment` is any object with the AlignmentResult attributes the reference
    reads: coords, gapped_sequence, gapped_target, target_name, query_name (reference alignment.py:106-150).
    Returns (alignment, int32 (Lq,Lq)) or (alignment, None) with a warning when coords is None."""
    coordinates = alignment.coords
    if coordinates is None:
        logger.warning(f"No coordinates found for {alignment.target_name}.")
        return (alignment, None)
    X = _coords_f32(coordinates)
    if X.shape[1] != 3:
        raise ValueError("Coordinates are not 3D.")
    q = alignment.gapped_sequence.encode("ascii")
    t = alignment.gapped_target.encode("ascii")
    if len(t) < len(q):
        raise ValueError("gapped_target is shorter than gapped_sequence")
    L = _hip.lib()
    lq = _hip.c_int64(0)
    _hip.check(L.mdf_align_len(q, t, len(q), lq))
    out = np.empty((lq.value, lq.value), dtype=np.int32)
    _hip.check(L.mdf_build_align_contact_map(_hip.ptr(X), X.shape[0], q, t, len(q), float(threshold),
                                             int(generated_contacts), _hip.ptr(out)))
    return (alignment, out)


# three-letter -> one-letter codes of the 20 standard amino acids plus the ambiguity / rare codes biotite's ProteinSequence knows
_THREE_TO_ONE = {"ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G", "HIS": "H", "ILE": "I",
                 "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S", "THR": "T", "TRP": "W", "TYR": "Y", "VAL": "V",
                 "ASX": "B", "GLX": "Z", "UNK": "X", "SEC": "U", "PYL": "O"}


# Modified residues -> the standard residue they derive from: the facts of pdbfixer's table, which the reference embeds as its
# module-level `substitutions` (reference bio_utils.py:48-193) and always applies at :252.  Kept here grouped by parent residue.
_MODIFIED_RESIDUES = {
    "ALA": "AIB ALM AYA BNN CHG CSD DAL DHA DNP FLA HAC MAA PRR TIH TPQ",
    "ARG": "ACL AGM ARM DAR HAR HMR",
    "ASN": "MEN",
    "ASP": "2AS ASA ASB ASK ASL ASQ BHD DAS DSP IAS",
    "CYS": "BCS BUC C5C C6C CAS CCS CEA CME CSO CSP CSS CSW CSX CY1 CY3 CYG CYM CYQ DCY EFC OCS PEC PR3 PYX SCH SCS SCY SHC SMC SOC",
    "GLN": "DGN",
    "GLU": "5HP CGU DGL GGL GMA PCA",
    "GLY": "GL3 GLZ GSC MPQ MSA NMC SAR",
    "HIS": "3AH DHI HIC HIP MHS NEM NEP",
    "ILE": "DIL IIL",
    "LEU": "BUG CLE DLE MK8 MLE NLE NLN NLP",
    "LYS": "5OW ALY DLY KCX LLP LLY LYM LYZ SHR TRG",
    "MET": "CXM FME MSE OMT",
    "PHE": "DAH DPN HPQ PHI PHL",
    "PRO": "DPR HYP",
    "SER": "DSN MIS OAS SAC SEL SEP SET SVA",
    "THR": "ALO BMT DTH TPO",
    "TRP": "DTR HTR LTR TPL TRO",
    "TYR": "DTY IYR PAQ PTR STY TYB TYI TYQ TYS TYY",
    "VAL": "DIV DVA MVA",
}
#: same name and meaning as the reference's module attribute: {modified three-letter name: standard three-letter name}
substitutions = {mod: std for std, mods in _MODIFIED_RESIDUES.items() for mod in mods.split()}


def get_residues_coordinates(structure, chain: str = "A", substitutions=None):
    """reference bio_utils.py:230-255: (one-letter residues, C-alpha coordinates) of one chain -- the C-alpha feed of the hot
    path.  `structure` is a biotite AtomArray or anything with the same per-atom NumPy attributes (chain_id, atom_name, hetero,
    res_name, coord); the selection is one vectorised mask (chain == `chain`, atom "CA", not hetero), no per-atom Python.
    Non-standard residue names are mapped through `substitutions` ({three-letter: standard three-letter}; default None = the
    module-level table above, which is what the reference always applies at bio_utils.py:252) before the one-letter
    conversion; unknown names raise, as biotite's ProteinSequence does.  Coordinates are returned as the structure stores them (float32 (L, 3) for biotite)."""
    chain_id = np.asarray(structure.chain_id)
    if chain not in set(chain_id.tolist()):
        raise ValueError(f"Chain {chain} not found in structure.")
    keep = (chain_id == chain) & (np.asarray(structure.atom_name) == "CA") & ~np.asarray(structure.hetero, dtype=bool)
    names = np.asarray(structure.res_name)[keep]
    uniq, inverse = np.unique(names, return_inverse=True)
    table = dict(_THREE_TO_ONE)
    subst = globals()["substitutions"] if substitutions is None else substitutions
    letters = []
    for n in uniq.tolist():
        std = subst.get(n, n)
        if std not in table:
            raise ValueError(f"'{n}' is not a known amino acid residue name")
        letters.append(table[std])
    residues = "".join(np.asarray(letters, dtype="<U1")[inverse].tolist()) if len(names) else ""
    return residues, np.asarray(structure.coord)[keep]
