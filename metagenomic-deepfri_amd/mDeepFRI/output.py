"""Output stage next to the hot path: the `score >= 0.1`, sort-descending filter the reference applies when it builds
results.tsv (reference pipeline.py:696-716 / 733-748), run on the GPU so that only a few dozen (term, score) pairs per
protein leave the device, plus the row formatter that reproduces the reference's lines byte for byte."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _hip

FINAL_OUTPUT_COLUMNS = 12  # query, net_type, mode, term, score, name + 6 alignment fields (pipeline.py:713-716)


def filter_scores(scores, threshold: float = 0.1, capacity_per_protein: int = 64):
    """scores: torch float32 CUDA tensor (B, T).  Returns (offsets int32 (B+1), term_idx int32 (N), kept float32 (N)) as
    torch tensors on the same device: protein p keeps term_idx[offsets[p]:offsets[p+1]], ordered by descending score
    with ties in term order (Python's stable `sorted(..., reverse=True)`).  Synchronises once (to size the result)."""
    import torch
    if not (scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 2 and scores.is_contiguous()):
        raise ValueError("scores must be a contiguous float32 CUDA tensor of shape (B, T)")
    L = _hip.lib()
    B, T = scores.shape
    dev = scores.device
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ws = torch.empty(L.mdf_filter_workspace_bytes(B), dtype=torch.uint8, device=dev)
    offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
    cap = max(int(B) * int(capacity_per_protein), 1024)
    with torch.cuda.device(dev):
        while True:
            status = torch.zeros(4, dtype=torch.int32, device=dev)
            term_idx = torch.empty(cap, dtype=torch.int32, device=dev)
            kept = torch.empty(cap, dtype=torch.float32, device=dev)
            _hip.check(L.mdf_filter_scores_dev(_hip.ptr(scores), B, T, float(threshold), _hip.ptr(offsets), _hip.ptr(term_idx),
                                               _hip.ptr(kept), cap, _hip.ptr(status), _hip.ptr(ws), ws.numel(), st))
            s = status.cpu().numpy()
            if s[0] == 0:
                break
            cap = int(s[1]) + 16
    n = int(offsets[-1].item())
    return offsets, term_idx[:n], kept[:n]


def results_rows(query_ids, net_type: str, mode_label: str, terms, gonames, offsets, term_idx, kept, alignment_data=None):
    """Lines of results.tsv for one GO head, formatted as reference pipeline.py:713-716:
    query_id, net_type, mode, term, f"{score:.4f}", go_name, then the six alignment fields (nan when unknown)."""
    off = np.asarray(offsets.cpu() if hasattr(offsets, "cpu") else offsets)
    ti = np.asarray(term_idx.cpu() if hasattr(term_idx, "cpu") else term_idx)
    sc = np.asarray(kept.cpu() if hasattr(kept, "cpu") else kept, dtype=np.float32)
    names = dict(zip(terms, gonames))
    lines = []
    for p, qid in enumerate(query_ids):
        aln = (alignment_data or {}).get(qid, [np.nan] * 6)
        aligned, target_id, database, target_identity, query_cov, target_cov = aln
        for k in range(off[p], off[p + 1]):
            term = terms[ti[k]]
            lines.append(f"{qid}\t{net_type}\t{mode_label}\t{term}\t{float(sc[k]):.4f}\t{names.get(term, 'Unknown')}"
                         f"\t{aligned}\t{target_id}\t{database}\t{target_identity}\t{query_cov}\t{target_cov}\n")
    return lines
