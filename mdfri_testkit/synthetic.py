"""Seeded synthetic inputs for the hot path (SURVEY.md section 8d / BASELINE.md section 3).

Sequences are uniform over the 20 standard amino acids (the alphabet of reference
benchmark/time_benchmark_cpu.py:44-46), C-alpha traces are 3.8 A random walks rounded to 3 decimals
(PDB precision), alignments are the query mutated with random indels, weights are Glorot-uniform with
the topology of the shipped models (`GraphConv_gcd_512-512-512_fcd_1024`, reference
mDeepFRI/__init__.py:73,78).  There is no network for real structures or checkpoints.
"""
from __future__ import annotations

import numpy as np

AA20 = "ACDEFGHIKLMNPQRSTVWY"
# number of GO terms of the v1.0 MERGED heads; always read T from the weights in real use
GO_TERMS = {"mf": 489, "bp": 1943, "cc": 320, "ec": 538}


def random_sequence(rng: np.random.Generator, length: int) -> str:
    idx = rng.integers(0, len(AA20), size=length)
    return "".join(AA20[i] for i in idx)


def random_walk_coords(rng: np.random.Generator, length: int, step: float = 3.8) -> np.ndarray:
    """(L,3) float32 C-alpha trace: unit directions * 3.8 A, cumulative sum, rounded to 3 decimals."""
    v = rng.standard_normal((length, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
    xyz = np.cumsum(v * step, axis=0)
    return np.round(xyz, 3).astype(np.float32)


def helix_bundle_coords(rng: np.random.Generator, length: int) -> np.ndarray:
    """(L,3) float32 C-alpha trace of a protein-like chain ("compact-helix generator"): ideal alpha-helical
    segments (radius 2.3 A, 100 degrees and 1.5 A rise per residue, 3.8 A between consecutive C-alphas) of 6..24 residues, each
    with a random orientation, joined 3.8 A apart.  At 6 A a residue then sees itself and i+-1, i+-2, i+-3 plus the occasional
    contact between segments: ~7 entries per row instead of the ~12.6 of a random walk -- what real structures look like to the
    A.X kernel.  Rounded to 3 decimals (PDB precision)."""
    xyz = np.empty((length, 3), dtype=np.float64)
    pos = 0
    last = np.zeros(3)
    while pos < length:
        n = min(int(rng.integers(6, 25)), length - pos)
        k = np.arange(n)
        local = np.stack([2.3 * np.cos(np.deg2rad(100.0) * k), 2.3 * np.sin(np.deg2rad(100.0) * k), 1.5 * k], axis=1)
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))          # random orientation of the segment
        seg = (local - local[0]) @ q.T
        step = rng.standard_normal(3)
        step *= 3.8 / (np.linalg.norm(step) + 1e-12)
        xyz[pos:pos + n] = seg + (last + step if pos else 0.0)
        last = xyz[pos + n - 1]
        pos += n
    return np.round(xyz, 3).astype(np.float32)


def mutate_alignment(rng: np.random.Generator, query: str, indel_rate: float = 0.05):
    """Return (gapped_query, gapped_target, target_length) for a query aligned to a synthetic target.

    Each query residue is, with probability `indel_rate`, an insertion (target gap '-'); before each
    query residue, with probability `indel_rate`, the target has an extra residue (query gap '-').
    """
    q_cols, t_cols = [], []
    lt = 0
    for ch in query:
        if rng.random() < indel_rate:  # extra target residue: gap in the query
            q_cols.append("-")
            t_cols.append(AA20[rng.integers(0, 20)])
            lt += 1
        if rng.random() < indel_rate:  # query insertion: gap in the target
            q_cols.append(ch)
            t_cols.append("-")
        else:
            q_cols.append(ch)
            t_cols.append(ch)
            lt += 1
    return "".join(q_cols), "".join(t_cols), lt


def glorot_uniform(rng: np.random.Generator, fan_in: int, fan_out: int) -> np.ndarray:
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


def glorot_gcn_weights(seed: int = 0, n_terms: int = 489, embed: int = 1024, gc_dims=(512, 512, 512),
                       fc_dim: int = 1024, fc_gain: float = 0.1, sparse_scores: bool = False, embed_linear: bool = False,
                       embed_bias: bool = False) -> dict:
    """Random-init DeepFRI GCN weights (fp32).  Keys are the ones mDeepFRI.weights reads and writes.

    `fc_gain` scales the Glorot draw of W_fc: the sum-pooled features grow linearly with the protein
    length, and with untrained unit-gain weights the pair-softmax saturates to exactly 0/1 beyond
    L~256, which would make an absolute 1e-4 score check vacuous.  0.1 keeps |logit| ~ 2-8 for
    L = 256-1024, the regime trained heads operate in.

    `sparse_scores`: give the output layer the operating point of a TRAINED head -- most GO terms are off for most proteins.
    The score of term t is sigmoid(z[2t] - z[2t+1]); with Glorot weights and +-0.05 biases the difference is centred on 0
    (std ~ 0.0022 L), so 84-100 % of all terms pass the results.tsv filter `score >= 0.1` (reference pipeline.py:696-705)
    and a "filtered" gather carries MORE bytes than the dense one.  The switch adds a per-term prior N(9, 3^2) to the bias of
    channel 1 (drawn from its own generator: every other weight is unchanged): the pass rate becomes
    Phi(-(9 - 2.197) / sqrt(9 + (0.0022 L)^2)) = 1.2 % at L = 128, 1.7 % at L = 512, 3.5 % at L = 1024 -- a few dozen terms
    per protein, different ones per protein.  Used by bench.py's configs3 / configs4 and tools/pipeline_example.py."""
    rng = np.random.default_rng(seed)
    w = {"W_aa": glorot_uniform(rng, 26, embed)}
    prev = embed
    for k, c in enumerate(gc_dims, start=1):
        w[f"W_gc{k}"] = glorot_uniform(rng, prev, c)
        prev = c
    w["W_fc"] = (glorot_uniform(rng, int(sum(gc_dims)), fc_dim) * np.float32(fc_gain)).astype(np.float32)
    w["b_fc"] = rng.uniform(-0.05, 0.05, size=(fc_dim,)).astype(np.float32)
    w["W_out"] = glorot_uniform(rng, fc_dim, 2 * n_terms)
    w["b_out"] = rng.uniform(-0.05, 0.05, size=(2 * n_terms,)).astype(np.float32)
    if embed_linear:   # topology variant: AA_embedding without activation (weights.validate: "embed_linear")
        w["embed_linear"] = np.ones(1, dtype=np.float32)
    if embed_bias:     # topology variant: AA_embedding with a bias
        w["b_aa"] = np.random.default_rng([int(seed), 0xB1]).uniform(-0.05, 0.05, size=(embed,)).astype(np.float32)
    if sparse_scores:
        prior = np.random.default_rng([int(seed), 0x5A]).normal(9.0, 3.0, size=n_terms).astype(np.float32)
        w["b_out"][1::2] += prior
    return w


def glorot_lm_weights(seed: int = 1000, hidden: int = 512, embed: int = 1024) -> dict:
    """Random-init language-model branch: two LSTM layers (Keras defaults: Glorot kernel, orthogonal recurrent kernel,
    zero bias with unit forget bias) and the LM_embedding Dense.  Keys as mDeepFRI.weights."""
    rng = np.random.default_rng(seed)

    def orthogonal(n, m):
        q, r = np.linalg.qr(rng.standard_normal((max(n, m), min(n, m))))
        q = q * np.sign(np.diag(r))
        return (q if n >= m else q.T).astype(np.float32)[:n, :m]

    H = hidden
    w = {}
    for name, fan_in in (("1", 26), ("2", H)):
        w[f"lm_W{name}"] = glorot_uniform(rng, fan_in, 4 * H)
        w[f"lm_U{name}"] = np.concatenate([orthogonal(H, H) for _ in range(4)], axis=1)
        b = rng.uniform(-0.05, 0.05, size=(4 * H,)).astype(np.float32)
        b[H:2 * H] += 1.0
        w[f"lm_b{name}"] = b
    w["W_lm"] = glorot_uniform(rng, H, embed)
    w["b_lm"] = rng.uniform(-0.05, 0.05, size=(embed,)).astype(np.float32)
    return w


def glorot_cnn_weights(seed: int = 0, n_terms: int = 489, filters=(120, 100, 80, 60), kernel_lens=(5, 10, 15, 20)) -> dict:
    """Random-init sequence-only DeepCNN weights (upstream DeepCNN defaults: four parallel Conv1D branches over the one-hot
    sequence, BatchNorm with non-trivial moving statistics, FuncPredictor).  Keys as mDeepFRI.weights."""
    rng = np.random.default_rng(seed)
    w = {}
    for b, (F, k) in enumerate(zip(filters, kernel_lens), start=1):
        lim = np.sqrt(6.0 / (k * 26 + k * F))
        w[f"cnn_W{b}"] = rng.uniform(-lim, lim, size=(k, 26, F)).astype(np.float32)
        w[f"cnn_b{b}"] = rng.uniform(-0.05, 0.05, size=(F,)).astype(np.float32)
    C = int(sum(filters))
    w["bn_gamma"] = rng.uniform(0.5, 1.5, size=(C,)).astype(np.float32)
    w["bn_gamma"][::7] *= -1.0                       # negative scales happen in trained models; max-pool must see them after BN
    w["bn_beta"] = rng.uniform(-0.2, 0.2, size=(C,)).astype(np.float32)
    w["bn_mean"] = rng.uniform(-0.1, 0.1, size=(C,)).astype(np.float32)
    w["bn_var"] = rng.uniform(0.01, 0.2, size=(C,)).astype(np.float32)
    w["bn_eps"] = np.array([1e-3], dtype=np.float32)
    w["W_out"] = glorot_uniform(rng, C, 2 * n_terms)
    w["b_out"] = rng.uniform(-0.05, 0.05, size=(2 * n_terms,)).astype(np.float32)
    return w


def synthetic_proteins(seed: int, count: int, length, indel_rate: float = 0.0, coords: str = "walk"):
    """List of dicts {id, seq, coords, q_aln, t_aln}.  `length` is an int or a (lo, hi) inclusive range; `coords` is "walk"
    (3.8 A random walk, the BASELINE.json configs) or "helix" (helix_bundle_coords)."""
    make_coords = {"walk": random_walk_coords, "helix": helix_bundle_coords}[coords]
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        L = int(length) if np.isscalar(length) else int(rng.integers(length[0], length[1] + 1))
        seq = random_sequence(rng, L)
        if indel_rate > 0:
            q_aln, t_aln, lt = mutate_alignment(rng, seq, indel_rate)
        else:
            q_aln, t_aln, lt = seq, seq, L
        out.append({"id": f"syn{seed}_{i}", "seq": seq, "coords": make_coords(rng, lt), "q_aln": q_aln, "t_aln": t_aln})
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Large workloads (BASELINE.json configs[3] / configs[4]): vectorised per protein, and seeded per protein so that every rank of
# a sharded run can generate exactly its own shard (lengths are drawn first, from one stream, so that all ranks agree on them).
# ---------------------------------------------------------------------------------------------------------------------
_AA20_BYTES = np.frombuffer(AA20.encode(), dtype=np.uint8)


def uniform_lengths(seed: int, count: int, lo: int = 128, hi: int = 1024) -> np.ndarray:
    """configs[3]: L ~ U{lo..hi} (inclusive)."""
    return np.random.default_rng(seed).integers(lo, hi + 1, size=count).astype(np.int32)


def histogram_lengths(seed: int, count: int) -> np.ndarray:
    """configs[4]: lengths drawn from the empirical length histogram of the reference's test proteome
    (mDeepFRI/tests/data/GCA_000731455.1.proteins.fa.gz, clipped to [30, 2048]; data/gca_000731455_length_hist.json, made by
    tools/make_length_histogram.py): a bin by its frequency, then uniform inside the 8-residue bin."""
    import json
    import os
    h = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "gca_000731455_length_hist.json")))
    counts = np.asarray(h["counts"], dtype=np.float64)
    rng = np.random.default_rng(seed)
    bins = rng.choice(len(counts), size=count, p=counts / counts.sum())
    L = h["first_bin_start"] + bins * h["bin_width"] + rng.integers(0, h["bin_width"], size=count)
    return np.clip(L, h["clip"][0], h["clip"][1]).astype(np.int32)


def bulk_protein(seed: int, index: int, length: int, indel_rate: float = 0.0):
    """One synthetic protein, same distribution as synthetic_proteins(): (seq, coords (Lt,3) f32, gapped query, gapped target).
    The generator is seeded by (seed, index): independent of which rank asks and of the order in which proteins are made."""
    rng = np.random.default_rng([int(seed), int(index)])
    L = int(length)
    res = _AA20_BYTES[rng.integers(0, 20, size=L)]
    seq = res.tobytes().decode()
    if indel_rate > 0:
        extra = rng.random(L) < indel_rate          # an extra target residue (query gap) in front of residue i
        ins = rng.random(L) < indel_rate            # residue i is an insertion (target gap)
        pos = np.arange(L) + np.cumsum(extra)       # alignment column of residue i
        q = np.full(L + int(extra.sum()), 45, dtype=np.uint8)
        t = q.copy()
        q[pos] = res
        t[pos] = np.where(ins, 45, res)
        t[pos[extra] - 1] = _AA20_BYTES[rng.integers(0, 20, size=int(extra.sum()))]
        lt = len(q) - int(ins.sum())
        q_aln, t_aln = q.tobytes().decode(), t.tobytes().decode()
    else:
        q_aln = t_aln = seq
        lt = L
    v = rng.standard_normal((lt, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
    coords = np.round(np.cumsum(v * 3.8, axis=0), 3).astype(np.float32)
    return seq, coords, q_aln, t_aln


def _bulk_slice(job):
    seed, idx, lens, indel_rate = job
    return [bulk_protein(seed, i, l, indel_rate) for i, l in zip(idx, lens)]


def bulk_proteins(seed: int, lengths, indices, indel_rate: float = 0.0, workers: int = 0):
    """Columns (seqs, coords, q_alns, t_alns) for the proteins `indices` of a workload whose lengths are `lengths`.
    workers > 1: generated by that many SPAWNED worker processes (every protein has its own seeded generator, so the result
    does not depend on it); to be called before the calling process needs its cores for anything else."""
    indices = list(indices)
    cols = ([], [], [], [])
    if workers > 1 and len(indices) >= 4096:
        import multiprocessing as mp
        step = 2048
        jobs = [(seed, indices[k:k + step], [int(lengths[i]) for i in indices[k:k + step]], indel_rate) for k in range(0, len(indices), step)]
        with mp.get_context("spawn").Pool(workers) as pool:
            for part in pool.imap(_bulk_slice, jobs):
                for prot in part:
                    for c, x in zip(cols, prot):
                        c.append(x)
        return cols
    for i in indices:
        for c, x in zip(cols, bulk_protein(seed, i, lengths[i], indel_rate)):
            c.append(x)
    return cols
