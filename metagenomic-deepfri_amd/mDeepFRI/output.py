"""Output stage next to the hot path: the `score >= 0.1`, sort-descending filter the reference applies when it builds
results.tsv (reference pipeline.py:696-716 / 733-748), run on the GPU so that only a few dozen (term, score) pairs per
protein leave the device, plus the row formatter that reproduces the reference's lines byte for byte."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _hip

FINAL_OUTPUT_COLUMNS = 12  # query, net_type, mode, term, score, name + 6 alignment fields (pipeline.py:713-716)


def filter_scores_async(scores, threshold: float = 0.1, capacity: int = 0):
    """The filter's launches on the current stream, no synchronisation: -> (offsets (B+1), term_idx (capacity), kept (capacity),
    status (4)) device tensors.  After a sync: status[0] != 0 means `capacity` was too small (status[1] = entries needed);
    otherwise the first offsets[-1] entries of term_idx / kept are the result (see filter_scores)."""
    import torch
    if not (scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 2 and scores.is_contiguous()):
        raise ValueError("scores must be a contiguous float32 CUDA tensor of shape (B, T)")
    L = _hip.lib()
    B, T = scores.shape
    dev = scores.device
    cap = max(int(capacity), 1024)
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ws = torch.empty(L.mdf_filter_workspace_bytes(B), dtype=torch.uint8, device=dev)
        offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
        status = torch.zeros(4, dtype=torch.int32, device=dev)
        term_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        kept = torch.empty(cap, dtype=torch.float32, device=dev)
        _hip.check(L.mdf_filter_scores_dev(_hip.ptr(scores), B, T, float(threshold), _hip.ptr(offsets), _hip.ptr(term_idx),
                                           _hip.ptr(kept), cap, _hip.ptr(status), _hip.ptr(ws), ws.numel(), st))
    return offsets, term_idx, kept, status


def filter_scores(scores, threshold: float = 0.1, capacity_per_protein: int = 64):
    """scores: torch float32 CUDA tensor (B, T).  Returns (offsets int32 (B+1), term_idx int32 (N), kept float32 (N)) as
    torch tensors on the same device: protein p keeps term_idx[offsets[p]:offsets[p+1]], ordered by descending score
    with ties in term order (Python's stable `sorted(..., reverse=True)`).  Synchronises once (to size the result)."""
    cap = int(scores.shape[0]) * int(capacity_per_protein)
    while True:
        offsets, term_idx, kept, status = filter_scores_async(scores, threshold, cap)
        s = status.cpu().numpy()
        if s[0] == 0:
            break
        cap = int(s[1]) + 16
    n = int(offsets[-1].item())
    return offsets, term_idx[:n], kept[:n]


def _concat(strings):
    """list of str -> (utf-8 bytes, int64 offsets (n + 1))."""
    enc = [x.encode("utf-8") for x in strings]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=len(enc)), out=off[1:])
    return b"".join(enc) or b"\0", off


def results_text(query_ids, net_type: str, mode_label: str, terms, gonames, offsets, term_idx, kept, alignment_data=None) -> bytes:
    """The text of results.tsv for one GO head (utf-8, the encoding the reference opens the file with), formatted as reference
    pipeline.py:713-716: query_id, net_type, mode, term, f"{score:.4f}", go_name, then the six alignment fields (nan when
    unknown) -- assembled by the library (`mdf_results_format_host`) from the filter's arrays, no per-line Python."""
    L = _hip.lib()
    off = np.ascontiguousarray(offsets.cpu() if hasattr(offsets, "cpu") else offsets, dtype=np.int32)
    ti = np.ascontiguousarray(term_idx.cpu() if hasattr(term_idx, "cpu") else term_idx, dtype=np.int32)
    sc = np.ascontiguousarray(kept.cpu() if hasattr(kept, "cpu") else kept, dtype=np.float32)
    query_ids = list(query_ids)
    B = len(query_ids)
    if len(off) != B + 1:
        raise ValueError("offsets must hold one entry per query plus one")
    terms = list(terms)
    names = dict(zip(terms, gonames))
    qid_b, qid_off = _concat(query_ids)
    term_b, term_off = _concat(terms)
    name_b, name_off = _concat([names.get(t, "Unknown") for t in terms])
    nan6 = "\t".join(["nan"] * 6)
    if alignment_data:
        tail_b, tail_off = _concat(["\t".join(map(str, alignment_data[q])) if q in alignment_data else nan6 for q in query_ids])
        tail_ptr = _hip.ptr(tail_off)
    else:
        tail_b, tail_ptr = nan6.encode() + b"\0", None
    nbytes, nlines = ctypes.c_int64(), ctypes.c_int64()
    args = (qid_b, _hip.ptr(qid_off), f"{net_type}\t{mode_label}".encode("utf-8"), term_b, _hip.ptr(term_off), name_b, _hip.ptr(name_off), tail_b, tail_ptr,
            _hip.ptr(off), _hip.ptr(ti), _hip.ptr(sc), B, len(terms))
    rc = L.mdf_results_format_host(*args, None, 0, ctypes.byref(nbytes), ctypes.byref(nlines))
    if rc not in (_hip.MDF_OK, _hip.MDF_ECAPACITY):
        _hip.check(rc)
    out = np.empty(max(nbytes.value, 1), dtype=np.uint8)
    _hip.check(L.mdf_results_format_host(*args, _hip.ptr(out), nbytes.value, ctypes.byref(nbytes), ctypes.byref(nlines)))
    return out[:nbytes.value].tobytes()


def results_rows(query_ids, net_type: str, mode_label: str, terms, gonames, offsets, term_idx, kept, alignment_data=None):
    """The same as a list of lines (str, each ending in a newline)."""
    return results_text(query_ids, net_type, mode_label, terms, gonames, offsets, term_idx, kept, alignment_data).decode("utf-8").splitlines(keepends=True)


def _matrix_body(query_ids, scores, net_type, threads):
    import csv
    import io
    s = scores.detach().cpu().numpy() if hasattr(scores, "detach") else np.asarray(scores)
    s = np.ascontiguousarray(s, dtype=np.float32)
    query_ids = [str(q) for q in query_ids]
    if s.ndim != 2 or s.shape[0] != len(query_ids):
        raise ValueError("scores must be (len(query_ids), T)")
    special = any(c in q for q in query_ids for c in '\t"\r\n') or any(c in net_type for c in '\t"\r\n') or any(q == "" for q in query_ids)
    prefixes = [_csv_row([q, net_type]) for q in query_ids] if special else [f"{q}\t{net_type}" for q in query_ids]
    pre_b, pre_off = _concat(prefixes)
    B, T = s.shape
    L = _hip.lib()
    nbytes = ctypes.c_int64()
    if B * T <= (1 << 20):       # small: one sizing pass, then a buffer of exactly that size
        rc = L.mdf_matrix_format_host(pre_b, _hip.ptr(pre_off), _hip.ptr(s), B, T, None, 0, int(threads), ctypes.byref(nbytes))
        if rc not in (_hip.MDF_OK, _hip.MDF_ECAPACITY):
            _hip.check(rc)
        out = np.empty(max(nbytes.value, 1), dtype=np.uint8)
    else:                        # large: no second pass over the digits, the bound is allocated (untouched pages cost nothing)
        out = np.empty(int(pre_off[-1]) + B * (T * 25 + 2) + 1, dtype=np.uint8)
    _hip.check(L.mdf_matrix_format_host(pre_b, _hip.ptr(pre_off), _hip.ptr(s), B, T, _hip.ptr(out), out.size, int(threads), ctypes.byref(nbytes)))
    return out[:nbytes.value]


def _csv_row(fields) -> str:
    import csv
    import io
    buf = io.StringIO()
    csv.writer(buf, delimiter="\t").writerow(fields)      # the reference's dialect: what gets quoted depends on its line terminator too
    return buf.getvalue()[:-2]


def prediction_matrix_text(query_ids, scores, net_type: str = "gcn", terms=None, threads: int = 0) -> bytes:
    """The text of the prediction matrix as reference pipeline.py writes it (header :566-571 when `terms` is given, rows :318-319:
    `csv.writer(delimiter="\t").writerow([query_id, net_type] + pred_vector.tolist())`), utf-8: every score is
    repr(float(np.float32(s))), rows end "\r\n".  scores: (B, T) float32 (numpy, or a torch tensor: copied to the host).  The rows are
    assembled by the library (`mdf_matrix_format_host`, rows dealt to `threads` host threads, 0 = as many as the machine has up to 32);
    only ids that need csv quoting go through the csv module.  `write_prediction_matrix` is the same without the final copies."""
    head = (_csv_row(["Protein", "Network_type"] + list(terms)) + "\r\n").encode("utf-8") if terms is not None else b""
    return head + _matrix_body(query_ids, scores, net_type, threads).tobytes()


def write_prediction_matrix(fh, query_ids, scores, net_type: str = "gcn", terms=None, threads: int = 0) -> int:
    """Write header (when `terms` is given) and rows to the binary file object `fh`; returns the bytes written."""
    head = (_csv_row(["Protein", "Network_type"] + list(terms)) + "\r\n").encode("utf-8") if terms is not None else b""
    body = _matrix_body(query_ids, scores, net_type, threads)
    fh.write(head)
    fh.write(memoryview(body))
    return len(head) + body.size
