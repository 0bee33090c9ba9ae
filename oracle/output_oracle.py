"""CPU restatement of the reference's results.tsv filter (mDeepFRI/pipeline.py:696-716, 733-748).  TEST INFRASTRUCTURE.

The reference writes every score through `pred_vector.tolist()` + csv (pipeline.py:318-319), reads the text back and
keeps `{term: float(s) ... if float(s) >= 0.1}`, `sorted(..., key=score, reverse=True)` (stable), formatted `:.4f`.
This module follows exactly that text round trip, so that the device filter is checked against the reference's
semantics (float32 -> repr -> float64 compare), not against a re-derivation."""
import csv
import io

import numpy as np


def matrix_text(query_ids, net_type, terms, scores):
    """The prediction-matrix TSV the reference writes (header pipeline.py:566-571, rows :318-319)."""
    buf = io.StringIO()
    w = csv.writer(buf, delimiter="\t")
    w.writerow(["Protein", "Network_type"] + list(terms))
    for qid, row in zip(query_ids, scores):
        w.writerow([qid, net_type] + np.asarray(row, dtype=np.float32).tolist())
    return buf.getvalue()


def results_lines(matrix_tsv, mode_label, gonames, alignment_data=None):
    """pipeline.py:684-716 on the text of one prediction matrix."""
    reader = csv.reader(matrix_tsv.strip().split("\n"), delimiter="\t")
    header = next(reader)
    terms = header[2:]
    term_to_name = {term: name for term, name in zip(terms, gonames)}
    out = []
    for row in reader:
        query_id, net_type, scores = row[0], row[1], row[2:]
        term_score = {terms[i]: float(scores[i]) for i in range(len(terms)) if float(scores[i]) >= 0.1}
        sorted_term_score = dict(sorted(term_score.items(), key=lambda item: item[1], reverse=True))
        for term, score in sorted_term_score.items():
            go_name = term_to_name.get(term, "Unknown")
            aln_info = (alignment_data or {}).get(query_id, [np.nan] * 6)
            aligned, target_id, database, target_identity, query_cov, target_cov = aln_info
            out.append(f"{query_id}\t{net_type}\t{mode_label}\t{term}\t{score:.4f}\t{go_name}"
                       f"\t{aligned}\t{target_id}\t{database}\t{target_identity}\t{query_cov}\t{target_cov}\n")
    return out
