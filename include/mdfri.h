/*
 * mdfri.h -- C ABI of libmdfri_hip.so: the MI355X (gfx950) implementation of Metagenomic-DeepFRI's
 * per-protein inference hot path (contact-map build/align + DeepFRI GCN forward).
 *
 * This is the drop-in boundary.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repository bioinf-mcb/Metagenomic-DeepFRI, v1.1.10).  The reference
 * binds native code through Cython `cpdef` functions; INTEGRATION.md shows the `cdef extern` stub a
 * maintainer adds to contact_map_utils.pyx / predict.pyx to call these symbols instead.
 *
 * Conventions
 *   - plain C types only: pointers + sizes, no torch / numpy types.
 *   - return value: 0 = success, negative = MDF_E* code; mdf_last_error() gives a thread-local message.
 *   - the caller allocates every output; nothing is retained past return (async entry points: until the
 *     stream is synchronised).
 *   - "host" entry points take host pointers and do H2D / kernel / D2H on the current device
 *     (they are the per-call API of the reference, which hands NumPy arrays across);
 *     "dev" entry points take device pointers and a hipStream_t (passed as void*), enqueue work and
 *     return without synchronising: they are the batched counterpart the reference lacks.
 *   - there is NO CPU fallback: without a GPU every compute entry point fails with MDF_ENODEVICE.
 *
 * Environment switches (ALL of them; each is read once per process or per object, so set it before the first call; round 6 removed the
 * other 23 developer knobs of rounds 1-5 with the kernel variants behind them -- experiments/r06_pruned_variants.patch)
 *   MDFRI_HW_PIPE=f32            the GraphConv / LSTM products on v_mfma_f32_32x32x2_f32 instead of BF16x6 (mdf_hw_pipe() reports the pipe;
 *                                another arithmetic: tests/test_gpu_gcn.py::test_bf16x6_products_are_at_least_as_accurate_as_the_fp32_instruction)
 *   MDFRI_HW_PIPE=f16x3          (opt-in, round 6) the GraphConv layers' H.W products, the LSTM time steps and the LM embedding from THREE fp16 term
 *                                products instead of six bf16 ones: half the matrix work, operands held to 22 of their 24 bits, GraphConv
 *                                activations beyond 8 190 become NaN scores (mdf_hw_pipe(); tests/test_gpu_lm.py passes under it;
 *                                another arithmetic: tests/test_gpu_gcn.py::test_f16x3_pipe_opt_in_is_fp32_class_and_bitwise_across_kernels)
 *   MDFRI_L1_FUSE=0              layer 1 by k_layer1 for every row instead of inside the layer-2 aggregation launch (mdf_layer1_form();
 *                                bit-identical: tests/test_gpu_engine.py::test_layer1_made_inside_the_aggregation_kernel_is_bit_identical)
 *   MDFRI_AX_MFMA=0              every protein through the CSR gather instead of the per-protein choice with the matrix-pipe aggregation
 *                                (another summation order, agreement to 1e-5: ...::test_gather_everywhere_knob_agrees_with_the_matrix_pipe_form)
 *   MDFRI_ENGINE_GRAPH=0         an engine never replays short batches as a hipGraph (bit-identical: ...::test_graph_replay_knob_is_bit_identical)
 *   MDFRI_LM_PERSISTENT_MAX_B=n  largest group of the one-launch LSTM, 0 = always the per-step GEMM form (bit-identical: tests/test_gpu_lm.py)
 *   MDFRI_NW_INT16=0             the aligner's 32-bit sweep for every pair instead of the packed 16-bit one where it qualifies (identical
 *                                alignments: tests/test_gpu_nw.py)
 *   (Python layer: MDFRI_HIP_LIB, MDFRI_DEVICE, MDFRI_NO_TORCH_HIP_PRELOAD -- mDeepFRI/_hip.py)
 */
#ifndef MDFRI_H
#define MDFRI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDF_OK 0
#define MDF_EINVAL (-1)     /* bad argument (NULL pointer, negative size, unsupported dimension)  */
#define MDF_ENODEVICE (-2)  /* no HIP device / HIP runtime error                                   */
#define MDF_ENOMEM (-3)     /* allocation failed                                                   */
#define MDF_EBADCHAR (-4)   /* residue letter outside the 26-letter alphabet (see mdf_seq2onehot)  */
#define MDF_ECAPACITY (-5)  /* caller-provided capacity too small (sparse outputs, CSR, workspace) */
#define MDF_EIO (-6)        /* model file unreadable / malformed                                   */

const char *mdf_last_error(void);
const char *mdf_version(void);
/* Number of visible HIP devices (0 without a GPU; never fails). */
int mdf_device_count(void);
int mdf_current_device(void); /* the calling thread's current HIP device (-1 without a device) */

/* ------------------------------------------------------------------------------------------------
 * Per-call API, host buffers.  Replaces the Cython functions one for one.
 * ---------------------------------------------------------------------------------------------- */

/* mDeepFRI/contact_map_utils.pyx:17-37  pairwise_sqeuclidean(float[:, ::1] X, int threads=1)
 * X: (n,m) C-contiguous f32.  D: (n,n) f32, D[i][j] = sum_k (X[i][k]-X[j][k])^2 accumulated in f32 in
 * ascending k with separately rounded multiply and add (the reference build has no FMA contraction);
 * diagonal exactly 0.  Bit-exact with the reference.  `threads` is accepted and ignored. */
int mdf_pairwise_sqeuclidean_f32(const float *X, int64_t n, int64_t m, float *D, int threads);

/* mDeepFRI/contact_map.py:64-75  DistanceMap.calculate_contacts  /  mDeepFRI/bio_utils.py:220
 * out[e] = (D[e] < thr) ? 1 : 0 for `count` elements; thr is the already squared threshold as f32
 * (NumPy>=2 compares an f32 array with a Python float in f32). */
int mdf_threshold_lt_i32(const float *D, int64_t count, float thr, int32_t *out);
/* The same comparison in float64: what NumPy does for a float64 distance map, and for an integer or float16 one compared
 * with a Python float (the binding converts those to float64, which is exact for them). */
int mdf_threshold_lt_f64_i32(const double *D, int64_t count, double thr, int32_t *out);

/* mDeepFRI/contact_map.py:88-95  ContactMap.sparsify  /  mDeepFRI/bio_utils.py:222-223
 * np.argwhere(cmap == 1).astype(int32): row-major sorted (N,2) index pairs of an (n,n) int32 matrix.
 * Writes at most `capacity` pairs and stores N in *n_pairs; returns MDF_ECAPACITY if N > capacity
 * (call again with a larger buffer; *n_pairs is valid). */
int mdf_argwhere_eq1_i32(const int32_t *cmap, int64_t n, int32_t *pairs, int64_t capacity, int64_t *n_pairs);

/* mDeepFRI/bio_utils.py:196-227  calculate_contact_map(coordinates, threshold, "sqeuclidean", mode)
 * coords (n,3) f32 -> fused on device: distances (never materialised on the host), strict `< threshold^2`.
 * Exactly one of cmap (n*n int32, mode="matrix") or pairs (mode="sparse", as mdf_argwhere_eq1_i32) is non-NULL. */
int mdf_calculate_contact_map(const float *coords, int64_t n, double threshold, int32_t *cmap,
                              int32_t *pairs, int64_t capacity, int64_t *n_pairs);

/* Query length of an alignment = number of non-gap ('-') bytes of q (mDeepFRI/contact_map_utils.pyx:64-80:
 * final query_idx).  Host-side shape helper so that the caller can size `out`. */
int mdf_align_len(const char *q_aln, const char *t_aln, int64_t La, int64_t *Lq);

/* mDeepFRI/contact_map_utils.pyx:44-117  align_contact_map(query_alignment, target_alignment,
 *     sparse_target_contact_map, generated_contacts=2, threads=1)
 * q_aln/t_aln: La ASCII bytes each; pairs: (N,2) int32 target contacts (any order, may be one-directional,
 * out-of-range entries are dropped as in the reference); out: (Lq,Lq) int32.  Bit-exact with the reference. */
int mdf_align_contact_map(const char *q_aln, const char *t_aln, int64_t La, const int32_t *pairs, int64_t N,
                          int generated_contacts, int32_t *out, int threads);

/* mDeepFRI/bio_utils.py:348-385  build_align_contact_map(alignment, threshold=6, generated_contacts=2)
 * Fused coords (Lt,3) + alignment -> (Lq,Lq) int32 without the (Lt,Lt) distance matrix or the pair list. */
int mdf_build_align_contact_map(const float *coords, int64_t Lt, const char *q_aln, const char *t_aln,
                                int64_t La, double threshold, int generated_contacts, int32_t *out);

/* mDeepFRI/predict.pyx:17-48  seq2onehot(str seq)
 * out: (L,26) f32 one-hot in the alphabet order "-DGULNTKHYWCPVSOIEFXQABZRM" (predict.pyx:26).
 * On an invalid byte returns MDF_EBADCHAR and stores its index in *bad_idx (the binding raises
 * ValueError(f"Invalid character in sequence: {seq[bad_idx]}") as predict.pyx:45-46 does). */
int mdf_seq2onehot(const char *seq, int64_t L, float *out, int64_t *bad_idx);

/* ------------------------------------------------------------------------------------------------
 * Model (replaces the onnxruntime.InferenceSession held by mDeepFRI/predict.pyx:50-73 Predictor)
 * ---------------------------------------------------------------------------------------------- */
typedef struct mdf_model mdf_model;

/* DeepFRI GCN weights, host pointers, fp32, row-major, Keras orientation (in_features x out_features).
 * Topology: AA_embedding Dense(26->embed, no bias) + relu; n_gc GraphConv layers (no bias, elu);
 * sum pooling of the concatenated GraphConv outputs; Dense(sum(gc_dims)->fc_dim, relu);
 * FuncPredictor Dense(fc_dim->2*n_terms) + softmax over pairs (see oracle/gcn_oracle.py). */
typedef struct {
    int32_t embed;        /* 1024 in the shipped models */
    int32_t n_gc;         /* number of GraphConv layers, 1..3 */
    int32_t gc_dims[3];   /* 512,512,512 */
    int32_t fc_dim;       /* 1024 */
    int32_t n_terms;      /* T: GO terms / EC numbers of this head */
    const float *W_aa;    /* (26, embed) */
    const float *W_gc[3]; /* (embed, gc0), (gc0, gc1), (gc1, gc2) */
    const float *W_fc;    /* (sum gc_dims, fc_dim) */
    const float *b_fc;    /* (fc_dim) */
    const float *W_out;   /* (fc_dim, 2*n_terms) */
    const float *b_out;   /* (2*n_terms) */
    /* language-model branch of the released models; lm_dim = 0: none.
     * X0 = relu(onehot.W_aa + lm_h.W_lm + b_lm), lm_h = LSTM2 output of mdf_lm_forward_dev. */
    int32_t lm_dim;       /* width of the language-model features (512) */
    const float *W_lm;    /* (lm_dim, embed) */
    const float *b_lm;    /* (embed) */
    /* Activation of the embedding: 0 = relu (the topology SURVEY.md section 3.3 states), 1 = none.  Upstream DeepFRI's model
     * builder is not in the reference tree and no released file exists offline; whether the no-language-model branch applies an
     * activation to `AA_embedding` is exactly the kind of detail only a released file settles, so it is DATA here: the ONNX reader
     * sets it from the graph (is there a Relu between the embedding and the first GraphConv?). */
    int32_t embed_linear;
} mdf_gcn_weights;

/* Upload weights to `device` and pre-pack them (transposed GEMM operands, folded embedding table). */
int mdf_model_create(const mdf_gcn_weights *w, int device, mdf_model **out);
/* Load a `.mdfw` container (format: metagenomic-deepfri_amd/mDeepFRI/weights.py). */
int mdf_model_load(const char *path, int device, mdf_model **out);
void mdf_model_free(mdf_model *m);
int mdf_model_num_terms(const mdf_model *m);
int mdf_model_feature_dim(const mdf_model *m); /* sum(gc_dims): width of the pooled feature vector */
int mdf_model_device(const mdf_model *m);
int mdf_model_lm_dim(const mdf_model *m);      /* 0: no language-model branch */
struct mdf_lm *mdf_model_lm(const mdf_model *m); /* the attached language model (mdf_model_attach_lm), or NULL */

/* ------------------------------------------------------------------------------------------------
 * LSTM language model feeding the GCN embedding of the released DeepFRI models.  The reference has no
 * code for it: it is part of the ONNX graph run at mDeepFRI/predict.pyx:98 (two stacked LSTM(512),
 * Keras gate order i,f,c,o; see oracle/lm_oracle.py for the exact arithmetic restated).
 * One mdf_lm is shared by all GO heads whose files carry the same frozen LM weights. */
typedef struct mdf_lm mdf_lm;
typedef struct {
    int32_t hidden;                    /* H: units of both LSTM layers, multiple of 64 */
    const float *W1, *U1, *b1;         /* LSTM1: kernel (26,4H), recurrent kernel (H,4H), bias (4H) */
    const float *W2, *U2, *b2;         /* LSTM2: kernel (H,4H),  recurrent kernel (H,4H), bias (4H) */
} mdf_lm_weights;
int mdf_lm_create(const mdf_lm_weights *w, int device, mdf_lm **out);
void mdf_lm_free(mdf_lm *lm);
int mdf_lm_hidden(const mdf_lm *lm);
/* Attach (or detach with NULL) the LM used by mdf_gcn_forward_host for a model with lm_dim > 0; not owned. */
int mdf_model_attach_lm(mdf_model *m, mdf_lm *lm);
size_t mdf_lm_workspace_bytes(const mdf_lm *lm, int32_t B, int32_t Lmax);
/* LSTM2 output for B proteins at once (device pointers unless noted).  Proteins must be ordered by
 * non-increasing length; protein b's residues are seq_idx[prot_row[b] .. +len[b]) (indices 0..25 from
 * mdf_seq_encode_dev) and its features are written to h_out[(prot_row[b]+t), 0..H) -- i.e. h_out is a
 * residue-row array (R, H) and rows that belong to no residue are left untouched.
 * len_host (HOST pointer) and len_dev hold the same B lengths.  Every time step is one launch per layer of
 * the MFMA GEMM over the still-active proteins, the LSTM cell fused in its epilogue. */
int mdf_lm_forward_dev(mdf_lm *lm, const uint8_t *seq_idx, const int64_t *prot_row, const int32_t *len_dev,
                       const int32_t *len_host, int32_t B, float *h_out, void *workspace, size_t workspace_bytes,
                       void *stream);

/* mDeepFRI/predict.pyx:75-102  Predictor.forward_pass(seqres, cmap)  -- GCN branch, one protein, host buffers.
 * seq: L ASCII residues; cmap: (L,L) C-contiguous, dtype by cmap_dtype (MDF_DT_*), cast to f32 as predict.pyx:88.
 * scores: (T) f32 = softmax(...)[:, 0] as predict.pyx:100.  On a bad residue: MDF_EBADCHAR + *bad_idx. */
#define MDF_DT_I32 0
#define MDF_DT_F32 1
#define MDF_DT_I64 2
#define MDF_DT_F64 3
#define MDF_DT_U8 4
int mdf_gcn_forward_host(mdf_model *m, const char *seq, int64_t L, const void *cmap, int cmap_dtype,
                         float *scores, int64_t *bad_idx);

/* ------------------------------------------------------------------------------------------------
 * Sequence-only CNN model: mDeepFRI/predict.pyx:91-100, Predictor.forward_pass(seqres) with cmap=None (the
 * reference feeds the one-hot sequence alone to the `DeepCNN-MERGED_{mode}.onnx` session; caller
 * pipeline.py:600-648).  n parallel Conv1D('same') branches over the one-hot sequence -> concat -> BatchNorm ->
 * relu -> global max pool -> FuncPredictor; arithmetic restated in oracle/cnn_oracle.py. */
typedef struct mdf_cnn mdf_cnn;
typedef struct {
    int32_t n_branch;
    const int32_t *kernel_len;      /* (n_branch) */
    const int32_t *filters;         /* (n_branch) */
    const int32_t *pad_left;        /* (n_branch) zeros left of the sequence, or NULL for TensorFlow 'same': (k-1)/2 */
    const float *const *W;          /* n_branch pointers, each (kernel_len, 26, filters)  -- Keras Conv1D kernel layout */
    const float *const *b;          /* n_branch pointers, each (filters) */
    const float *bn_gamma, *bn_beta, *bn_mean, *bn_var;   /* (sum filters) BatchNormalization, inference form */
    float bn_eps;
    int32_t n_terms;
    const float *W_out;             /* (sum filters, 2*n_terms) */
    const float *b_out;             /* (2*n_terms) */
} mdf_cnn_weights;
int mdf_cnn_create(const mdf_cnn_weights *w, int device, mdf_cnn **out);
void mdf_cnn_free(mdf_cnn *m);
int mdf_cnn_num_terms(const mdf_cnn *m);
int mdf_cnn_channels(const mdf_cnn *m);
/* One protein, host buffers (the per-call shape of predict.pyx:75-102).  scores: (T) f32. */
int mdf_cnn_forward_host(mdf_cnn *m, const char *seq, int64_t L, float *scores, int64_t *bad_idx);
/* B proteins in residue-row layout (see below): seq_idx (R) from mdf_seq_encode_dev, Lq / row_off device int32.
 * R = row_off[B] (total rows).  scores: (B, T) f32.  workspace: mdf_cnn_workspace_bytes(m, B, R). */
size_t mdf_cnn_workspace_bytes(const mdf_cnn *m, int32_t B, int64_t R);
/* The two halves of mdf_cnn_forward_dev, for callers that pool chunk by chunk and run the output layer once over all
 * proteins: pooled is (B, mdf_cnn_padded_channels) f32; mdf_cnn_pool_dev needs a workspace of 4*(R/32+1) bytes rounded up to 256. */
int mdf_cnn_padded_channels(const mdf_cnn *m);
int mdf_cnn_pool_dev(mdf_cnn *m, const uint8_t *seq_idx, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R,
                     float *pooled, void *workspace, size_t workspace_bytes, void *stream);
int mdf_cnn_head_dev(mdf_cnn *m, const float *pooled, int32_t B, float *scores, void *stream);
int mdf_cnn_forward_dev(mdf_cnn *m, const uint8_t *seq_idx, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R,
                        float *scores, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Batched device API (the counterpart of pipeline.py:476-481 Pool.map(build_align_contact_map) and
 * pipeline.py:292-319 _run_prediction_loop, for B proteins at once).
 *
 * Residue-row layout: protein p owns rows [row_off[p], row_off[p]+Lq[p]) of every per-residue array;
 * row_off[p] is a multiple of MDF_GROUP_ROWS (16) and row_off[B] (= R, total rows) a multiple of 128; rows between
 * Lq[p] and the next protein are padding (kept zero).  Use mdf_layout_rows() to build row_off.
 * A GROUP is MDF_GROUP_ROWS consecutive rows: the unit of the pooling partial sums (half a 32 x 32 MFMA tile).
 * All descriptor arrays are int32 on the device.
 * ---------------------------------------------------------------------------------------------- */
#define MDF_GROUP_ROWS 16
int mdf_group_rows(void);   /* = MDF_GROUP_ROWS, for callers that do not compile against this header */

/* Which matrix pipe the graph-convolution products H.W run on in this process: "bf16x6" (default: every fp32 operand split into three
 * bf16 terms, six term products per fp32 product accumulated in fp32 -- k_gemm_bf16x6, csrc/gcn.hip; error against float64 below the
 * fp32 instruction's), "f32" (v_mfma_f32_32x32x2_f32; environment MDFRI_HW_PIPE=f32) or "f16x3" (opt-in, MDFRI_HW_PIPE=f16x3: the GraphConv
 * layers' products, the LSTM time steps and the LM embedding -- not the GO heads -- from three fp16 term products: each operand times a power of two, split into two fp16 terms that hold
 * 22 of its 24 bits; the weights' scale is taken from their maximum at model load, the activations' is the constant 2^3, so an activation of
 * magnitude >= 8 190 turns into inf and the protein's scores into NaN, and one below 2^-5 is held to an absolute 2^-28 instead of 22 bits
 * (the LSTM's hidden state lies in (-1, 1): scale 2^13, no limit); k_gemm_f16x3, csrc/gcn.hip).  fp32 in, fp32 out in every case.
 * The variable is read ONCE, by the first product of the process: set it before the first call into the library (setting it later is
 * silently ignored; mdf_hw_pipe() tells which pipe is in use).  Non-finite and near-overflow inputs: the bf16x6 split rounds
 * hi = bf16(x) to nearest, so |x| > 0x1.fe fp127 (the top 2^-9 of the fp32 range) rounds hi to inf and x - hi to NaN, and an inf operand
 * gives inf - inf = NaN, where the fp32 instruction would return a finite product / inf.  Activations and weights of the GraphConv and
 * LSTM layers are many orders of magnitude below that range; a caller that feeds such values selects MDFRI_HW_PIPE=f32. */
const char *mdf_hw_pipe(void);

/* How layer 1 (H1 = elu(S . T1), the folded embedding) is computed on the fused engine path: "fused" (default: inside the layer-2
 * aggregation kernel for the proteins of the matrix-pipe classes, H1 never written; k_layer1 for the rows of the others) or "kernel"
 * (k_layer1 for every row; environment MDFRI_L1_FUSE=0 or MDFRI_AX_MFMA=0, read once).  Bit-identical results either way. */
const char *mdf_layer1_form(void);

/* Host helper: row_off[0..B] from Lq[0..B-1] as specified above.  Returns R (total rows) or a negative code. */
int64_t mdf_layout_rows(const int32_t *Lq, int32_t B, int32_t *row_off);

/* Residue letters -> alphabet index per row (the sparse form of seq2onehot, predict.pyx:17-48).
 * seqs: packed query sequences, protein p at bytes [seq_off[p], seq_off[p]+Lq[p]); seq_idx: (R) uint8, 255 on
 * padding rows.  *bad (ONE device int64, initialised to -1 by the caller) receives (protein << 32 | position) of the
 * FIRST invalid byte -- lowest protein, then lowest position, as the reference's serial scan reports it
 * (predict.pyx:36-46) -- by a 64-bit atomic minimum, or stays -1. */
int mdf_seq_encode_dev(const char *seqs, const int32_t *seq_off, const int32_t *Lq, const int32_t *row_off,
                       int32_t B, int64_t R, uint8_t *seq_idx, int64_t *bad, void *stream);

/* Fused contact-map stage for B proteins: C-alpha coords + gapped alignments -> normalised adjacency in CSR,
 * i.e. bio_utils.py:348-385 (a1+a2+a3) followed by GraphConv's normalisation
 *     A' = A - diag(A) + I,  d = 1/(1e-6 + sqrt(rowsum A')),  val[i][j] = (d[i]*A'[i][j])*d[j]
 * coords: packed (sum Lt,3) f32, protein p at residues [coord_off[p], coord_off[p+1]);
 * q_aln/t_aln: packed alignment bytes, protein p at [aln_off[p], aln_off[p+1]).
 * max_len: the longest query (max Lq) of the batch -- sizes the per-row contact-bit words of the workspace.
 * Outputs: rowptr (R+1) int32, colidx/val (nnz_cap) with colidx as GLOBAL row numbers.
 * status: device int32[4], zero-initialised by the caller: [0] != 0 -> CSR overflow (needed nnz in [1]); [2] != 0 -> a query
 * longer than max_len was met (its length in [2]): the result of that protein is invalid.
 * seq_idx / letter_sums (both or neither): with the residue indices of mdf_seq_encode_dev given, the layer-1 operand of
 * mdf_gcn_embed_dev (see mdf_letter_sums_dev) is written in the same pass as the CSR -- R x 32 f32 in the order of MDF_LSUM_INDEX.
 * The coordinates are read ONCE: a first kernel counts every row's contacts and stores the contact bits, the CSR is filled
 * from the bits.  workspace: mdf_cmap_workspace_bytes(B, R, max_len) bytes of device scratch (max_len = 0: the size the
 * dense-format and dense-to-CSR entry points below need). */
size_t mdf_cmap_workspace_bytes(int32_t B, int64_t R, int32_t max_len);
int mdf_cmap_csr_dev(const float *coords, const int32_t *coord_off, const char *q_aln, const char *t_aln,
                     const int32_t *aln_off, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R, int32_t max_len,
                     double threshold, int generated_contacts, int32_t *rowptr, int32_t *colidx, float *val,
                     int64_t nnz_cap, int32_t *status, const uint8_t *seq_idx, float *letter_sums, void *workspace,
                     size_t workspace_bytes, void *stream);
/* ... with the two offset arrays as (begin, end) PAIRS when off_stride == 2: protein p's residues are [coord_off[2 p], coord_off[2 p + 1]),
 * its alignment bytes [aln_off[2 p], aln_off[2 p + 1]) -- proteins visited in another order than the packed arrays store them (a plan that
 * orders a batch by length, mdf_plan_create).  off_stride == 1 is mdf_cmap_csr_dev. */
int mdf_cmap_csr_pairs_dev(const float *coords, const int32_t *coord_off, const char *q_aln, const char *t_aln,
                           const int32_t *aln_off, int32_t off_stride, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R, int32_t max_len,
                           double threshold, int generated_contacts, int32_t *rowptr, int32_t *colidx, float *val,
                           int64_t nnz_cap, int32_t *status, const uint8_t *seq_idx, float *letter_sums, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Same stage, reference output format: out[p] = (Lq[p],Lq[p]) int32 at element offset out_off[p] (int64, device).
 * The batched build_align_contact_map; bit-exact with the reference per protein. */
int mdf_cmap_dense_dev(const float *coords, const int32_t *coord_off, const char *q_aln, const char *t_aln,
                       const int32_t *aln_off, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R,
                       double threshold, int generated_contacts, int32_t *out, const int64_t *out_off,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Dense contact maps (what forward_pass receives, predict.pyx:82-90) -> the same normalised CSR.
 * cmaps: protein p is an (Lq[p],Lq[p]) matrix of cmap_dtype at element offset cmap_off[p] (int64, device). */
int mdf_dense_to_csr_dev(const void *cmaps, int cmap_dtype, const int64_t *cmap_off, const int32_t *Lq,
                         const int32_t *row_off, int32_t B, int64_t R, int32_t *rowptr, int32_t *colidx,
                         float *val, int64_t nnz_cap, int32_t *status, void *workspace, size_t workspace_bytes,
                         void *stream);
/* The same, additionally leaving every row's contact BITS (entry != 0) in the workspace (laid out with max_len = the longest query, see
 * mdf_cmap_ws_view) and telling, per protein, whether its map is binary: binary[p] (int32, device) is left at 1 when every entry off the
 * diagonal is 0 or 1, cleared to 0 otherwise -- what the matrix-pipe aggregation (mdf_agg_desc) needs of a dense map. */
int mdf_dense_to_csr_masks_dev(const void *cmaps, int cmap_dtype, const int64_t *cmap_off, const int32_t *Lq, const int32_t *row_off,
                               int32_t B, int64_t R, int32_t max_len, int32_t *rowptr, int32_t *colidx, float *val, int64_t nnz_cap,
                               int32_t *status, int32_t *binary, void *workspace, size_t workspace_bytes, void *stream);

/* Layer-1 operand, shared by every GO head: S[i][a] = sum of val over the CSR entries of row i whose column
 * residue is letter a  (= (Ahat . onehot)[i][a]); R x 32 f32, letters 26..31 zero.  With the embedding folded at model
 * load (relu(onehot W_aa) W_gc1 is a 26-row table), GraphConv layer 1 is elu(S . table).
 * Storage order (round 6): `letter_sums` is an operand between two stages of this library, and it is stored the way its hottest reader
 * -- the aggregation kernel that makes layer 1 from it on the fp32 matrix instruction -- takes it: per 16-row group (MDF_GROUP_ROWS) 512
 * floats, element (row, letter a) at MDF_LSUM_INDEX(row, a): the 16 bytes a lane of that kernel loads are the four letters it feeds to four
 * consecutive matrix instructions, a wave's load instruction covers 2 x 512 contiguous bytes (8 cache lines instead of 32: the row-major
 * form cost 160 of 500 us per launch in L1 line look-ups, profiles/r06_ax_timeline.txt).  Every producer (mdf_cmap_csr_dev,
 * mdf_letter_sums_dev) and consumer (mdf_gcn_embed_*_dev) of this library uses the macro. */
#define MDF_LSUM_INDEX(row, a) \
    ((size_t)((row) >> 4) * 512 + (size_t)((((a) >> 3) * 128) + ((((a) & 1) * 16 + ((row) & 15)) * 4) + (((a) >> 1) & 3)))
int mdf_letter_sums_dev(const uint8_t *seq_idx, const int32_t *rowptr, const int32_t *colidx, const float *val, int64_t R,
                        float *letter_sums, void *stream);

/* ---- the matrix-pipe form of the A.X aggregation -----------------------------------------------------------------------------
 * For a BINARY contact map (the fused path's always are; a dense map handed to forward_pass may hold any values) the aggregation
 *     out[i] = d_i * sum_j A'[i][j] * (d_j * H[j])
 * is an exact block-sparse product on the bf16 matrix pipe: A' is 0/1, d_j * H[j] is split into three bf16 terms whose sum is the fp32
 * value, products 1 * term are exact and the accumulation is fp32 -- the same fp32 sum as the CSR gather's, in another order, with
 * every H element crossing L1 once instead of once per neighbour.  It serves proteins of MDF_AGG_MIN_LEN .. MDF_AGG_MAX_LEN residues
 * (below, a workgroup per protein and channel slab is mostly overhead and the gather's working set is cache-sized anyway; above, the
 * accumulators of a protein's row blocks no longer fit a workgroup); the others, and non-binary maps, keep the CSR gather kernel.  Which kernel aggregates a protein depends on that protein alone (its length,
 * its map), never on the batch around it: batch == per call stays bitwise.
 * mdf_agg_desc says, for the R rows of one mdf_gcn_embed*_agg_dev call, which proteins go where.  The stack takes TWO of them (an array):
 * [0] for the aggregation in front of layer 2 (and a language-model head's layer 1), whose operand was just written and is cache-resident,
 * [1] for layer 3 and up, whose operand is not -- the matrix-pipe form gains more there, so more lengths are worth it (mdf_agg_class,
 * measured: profiles/r04_ax_mfma_by_length.txt).  masks / dinv / blk / row_off / Lq are the same in both. */
#ifndef MDF_AGG_MIN_LEN   /* (a build for a length sweep may state another: experiments/r06_len_classes.sh) */
#define MDF_AGG_MIN_LEN 80    /* round 6 (profiles/r06_len_classes.txt): 96 residues +5.7 %, 80 +2 % on the step against the gather; 64 -4 % (rounds 4-5: 112) */
#endif
#define MDF_AGG_MAX_LEN 1024
typedef struct mdf_agg_desc {
    const uint64_t *masks;      /* (R, W) device: bit j of word (r0+i, j/64) = A'[i][j] (diagonal set); rows >= Lq of a protein all zero.
                                 * What mdf_agg_prepare_dev read; the aggregation kernels read `tiles` below */
    int32_t W;
    const uint8_t *tiles;       /* device, from mdf_agg_prepare_dev: the same bits in the order the matrix-pipe kernel takes them (round 6).
                                 * Protein p (first row r0, nch = ceil(Lq / 256) column chunks) owns the bytes from r0 * tile_row_bytes on;
                                 * 16-row group g of the protein x column chunk c is a 512-byte tile at ((g * nch + c) * 512), byte
                                 * ((i % 16) * 2 + h) * 16 + cb of it = the bits of row 16 g + i % 16, columns 256 c + 16 cb + 8 h .. + 7:
                                 * a lane of the kernel (row, lane half h) gets the A-operand bytes of all 16 column blocks of a chunk in ONE
                                 * 16-byte load, and a wave's load covers 2 x 512 contiguous bytes (8 cache lines; the word form: 4 loads
                                 * of 16-32 lines each -- four fifths of the kernel's L1 line look-ups) */
    int32_t tile_row_bytes;     /* 32 * ceil(min(max_len, MDF_AGG_MAX_LEN) / 256): mdf_agg_tile_row_bytes(max_len) */
    const float *dinv;          /* (R) device: 1 / (1e-6 + sqrt(degree)), from mdf_agg_prepare_dev */
    const uint64_t *blk;        /* (B, 32) device: bit c of entry (p, b): rows [32b, 32b+32) of protein p have a contact in columns [16c, 16c+16) */
    const int32_t *row_off;     /* (B+1) device */
    const int32_t *Lq;          /* (B) device */
    const int32_t *plist;       /* device: indices (into row_off / Lq / blk) of the proteins the matrix-pipe kernel aggregates, by length class: */
    int32_t n_mf[3];            /* first n_mf[0] of at most 256 residues, then n_mf[1] of at most 512, then n_mf[2] of at most 1 024 */
    const int32_t *gate;        /* device, may be NULL: per protein (indexed like Lq), 0 = this protein's map is NOT binary: the matrix-pipe
                                 * kernel skips it and the CSR gather launch below takes it (per-call path: known on the device only) */
    const int32_t *csr_seg;     /* HOST: n_seg pairs (first row, row count) left to the CSR gather kernel */
    int32_t n_seg;
    const uint32_t *skip_groups;/* device, may be NULL: bit g (word g / 32) set = the MDF_GROUP_ROWS rows of group g belong to a listed protein.  When the
                                 * rows left to the gather fall into many segments (an UNSORTED batch: one launch per segment would be launch-bound)
                                 * the gather runs ONCE over all rows and skips the flagged groups */
    int32_t csr_gated;          /* 1: the CSR launches run only where gate[protein 0] == 0 (single-protein calls) */
    int64_t tail_row0;          /* rows [tail_row0, R) belong to no protein: the aggregate is zeroed there ... */
    int32_t tail_p;             /* ... by the workgroups of this listed protein (the last one of the rows, when it is on the list), or, < 0, by a
                                 * memset behind the launches (when the last protein is a long one the gather over its rows covers them) */
    /* descriptor [0] of the fused engine path only (round 5; zero elsewhere): which of the listed proteins get their LAYER-1 rows made inside
     * the layer-2 launch (mdf_agg_l1_fused) and which run the plain kernel on H1 rows that k_layer1 wrote.  plist then holds the n_mf[]
     * proteins of the first kind, class after class, FOLLOWED by n_plain[] of the second; csr_seg / skip_groups above describe the rows of
     * neither (the gather's), l1_seg / l1_skip the rows of everything but the first kind (k_layer1's) */
    int32_t n_plain[3];
    const int32_t *l1_seg;      /* HOST: n_l1_seg pairs (first row, row count) */
    int32_t n_l1_seg;
    const uint32_t *l1_skip;    /* device, may be NULL: group bitmap of the proteins of the first kind */
} mdf_agg_desc;

/* Length class of a protein for the aggregation of kind `resident` (1: the descriptor [0] above, 0: [1]): 0 / 1 / 2 = one / two / four
 * 32-row blocks per wave of the matrix-pipe kernel (at most 256 / 512 / 1 024 residues), -1 = the CSR gather.  Since round 5 the two kinds
 * list the same lengths (after the kernel's instruction diet the matrix-pipe form wins in front of layer 2 wherever it wins in front of
 * layer 3); what differs in front of layer 2 is whether layer 1 is made inside the launch: mdf_agg_l1_fused(L) != 0 (fused engine path
 * only; the per-call and dense-map paths run k_layer1 + the plain kernel, bit-identical).  Round 6: every length from MDF_AGG_MIN_LEN to
 * MDF_AGG_MAX_LEN takes the matrix pipe (no gaps above the multiples of 256), and every one of them has its layer 1 made inside the launch. */
int mdf_agg_class(int32_t L, int resident);
int mdf_agg_l1_fused(int32_t L);

/* dinv (R) and blk (B, 32) from the contact bits and the per-row degrees (counts: int32 (R), the number of set bits of a row) that
 * the contact stage leaves in its workspace (mdf_cmap_ws_view), and the contact bits once more as the byte tiles the matrix-pipe kernel
 * loads (mdf_agg_desc.tiles).  Proteins longer than MDF_AGG_MAX_LEN (or shorter than MDF_AGG_MIN_LEN) get no blk entry and no tiles. */
int mdf_agg_prepare_dev(const uint64_t *masks, int32_t W, const int32_t *counts, const int32_t *row_off, const int32_t *Lq, int32_t B,
                        int64_t R, float *dinv, uint64_t *blk, uint8_t *tiles, int32_t tile_row_bytes, void *stream);
/* ... `tiles`: R * mdf_agg_tile_row_bytes(max_len) bytes (max_len: the longest protein of the R rows), see mdf_agg_desc.tiles. */
int32_t mdf_agg_tile_row_bytes(int32_t max_len);
/* Pointers into a contact-stage workspace (laid out for R rows, max_len) after mdf_cmap_csr_dev / mdf_dense_to_csr_masks_dev ran on it. */
int mdf_cmap_ws_view(void *workspace, size_t workspace_bytes, int64_t R, int32_t max_len, const uint64_t **masks, int32_t *W,
                     const int32_t **counts);

/* GraphConv stack for R residue rows.  Output: per-group (MDF_GROUP_ROWS rows) partial sums of concat(H1,H2,H3):
 * partial (R/32, feature_dim) f32 -- the deterministic first level of the sum pooling (H3 itself never reaches HBM).
 * workspace: mdf_gcn_workspace_bytes(model, R) bytes. */
size_t mdf_gcn_workspace_bytes(const mdf_model *m, int64_t R);
int mdf_gcn_embed_dev(mdf_model *m, const float *letter_sums, const int32_t *rowptr, const int32_t *colidx, const float *val,
                      int64_t R, float *partial, void *workspace, size_t workspace_bytes, void *stream);
/* ... with the aggregation kernels chosen per protein as `agg` says (NULL: the CSR gather for every row, = mdf_gcn_embed_dev) */
int mdf_gcn_embed_agg_dev(mdf_model *m, const float *letter_sums, const int32_t *rowptr, const int32_t *colidx, const float *val,
                          int64_t R, const mdf_agg_desc *agg, float *partial, void *workspace, size_t workspace_bytes, void *stream);

/* The same stage for a model with a language-model branch (lm_dim > 0): X0 = relu(lm_h.W_lm + b_lm + W_aa[seq_idx]) on the
 * MFMA GEMM (table row added in the epilogue), then every GraphConv layer as A.X + H.W (layer 1 over `embed` channels).
 * seq_idx (R) from mdf_seq_encode_dev, lm_h (R, lm_dim) from mdf_lm_forward_dev.  workspace: mdf_gcn_workspace_bytes. */
int mdf_gcn_embed_lm_dev(mdf_model *m, const uint8_t *seq_idx, const float *lm_h, const int32_t *rowptr, const int32_t *colidx,
                         const float *val, int64_t R, float *partial, void *workspace, size_t workspace_bytes, void *stream);
int mdf_gcn_embed_lm_agg_dev(mdf_model *m, const uint8_t *seq_idx, const float *lm_h, const int32_t *rowptr, const int32_t *colidx,
                             const float *val, int64_t R, const mdf_agg_desc *agg, float *partial, void *workspace, size_t workspace_bytes,
                             void *stream);

/* Second level of the sum pooling: pooled[p] = sum of partial[g] over g in [grp_off[p], grp_off[p+1])  -> (B, feature_dim).
 * grp_off (B+1, int32, device) counts groups (row_off / MDF_GROUP_ROWS, plus the group base of the protein's chunk when the
 * partials of several chunks share one array). */
int mdf_gcn_pool_dev(mdf_model *m, const float *partial, const int32_t *grp_off, int32_t B, float *pooled, void *stream);

/* GO head for B pooled vectors: relu(g W_fc + b_fc) W_out + b_out -> pair softmax -> channel 0 (predict.pyx:100).
 * scores: (B, T) f32.  logits (optional, may be NULL): (B, 2T) pre-softmax, for tolerance studies.
 * workspace: mdf_head_workspace_bytes(model, B). */
size_t mdf_head_workspace_bytes(const mdf_model *m, int32_t B);
int mdf_gcn_head_dev(mdf_model *m, const float *pooled, int32_t B, float *scores, float *logits,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Batch engine: the whole batched hot path behind ONE call (`mdf_cmap_batch` /
 * `mdf_gcn_forward_batch`) -- the counterpart of pipeline.py:476-481 Pool.map(build_align_contact_map) followed by
 * pipeline.py:292-319 _run_prediction_loop, for B proteins and several GO heads at once.  The planner that cuts a batch
 * into chunks of residue rows, the per-chunk launch sequence (encode -> fused contact map/CSR -> GraphConv stack per head ->
 * pooling) and the GO heads run inside the library; a consumer of this header needs no other entry point.
 * An engine serialises its work on the stream it is called with and owns its device workspaces (grown on demand,
 * never shrunk); use one engine per stream / thread.
 * ---------------------------------------------------------------------------------------------- */

/* --- planner (host only; needs no GPU) ---
 * Proteins stay in input order.  A CHUNK is a run of consecutive proteins whose padded rows (32 per protein, see the
 * residue-row layout above) fit `max_rows`; a SEGMENT is a run of chunks whose per-group pool partials share one array
 * (at most max_segment_groups groups of MDF_GROUP_ROWS rows).  An immutable plan may be shared by any number of batches of the same
 * lengths. */
typedef struct mdf_plan mdf_plan;
/* mdf_plan_create VISITS the proteins shortest first (stable: equal lengths keep their input order), as the reference's work list does
 * (pipeline.py:529-533 sorts it by length): proteins of like length share chunks, which is what the per-length choice of the aggregation
 * kernel is built for (an unsorted mixed-length batch costs ~10 % of the GCN stage).  The order is internal: batches (mdf_batch_dev), dense
 * maps, scores, validation reports and every other per-protein array of the API stay in the CALLER's order; the chunk / segment tables below
 * and mdf_plan_order() speak of plan positions.  Results do not depend on it (a protein's scores are the same bits in any batch).
 * mdf_plan_create_ex with MDF_PLAN_KEEP_ORDER visits them in input order (a caller that has sorted already, tests). */
#define MDF_PLAN_KEEP_ORDER 1u
int mdf_plan_create(const int32_t *Lq, int32_t B, int32_t max_rows, int32_t max_segment_groups, mdf_plan **out);
int mdf_plan_create_ex(const int32_t *Lq, int32_t B, int32_t max_rows, int32_t max_segment_groups, uint32_t flags, mdf_plan **out);
/* plan position -> index in the caller's batch (B entries, owned by the plan); NULL (count 0) when the plan keeps the input order */
const int32_t *mdf_plan_order(const mdf_plan *plan, int64_t *count);
void mdf_plan_free(mdf_plan *plan);
int32_t mdf_plan_num_proteins(const mdf_plan *plan);
int32_t mdf_plan_num_chunks(const mdf_plan *plan);
int32_t mdf_plan_num_segments(const mdf_plan *plan);
int64_t mdf_plan_max_chunk_rows(const mdf_plan *plan);
/* chunk table, n_chunks x 6 int64: p0, p1 (proteins [p0, p1)), rows (R of the chunk), row_off_pos (start of the chunk's
 * p1-p0+1 row offsets in the row-offset array), segment, group_base (first group of the chunk in its segment) */
int mdf_plan_chunks(const mdf_plan *plan, int64_t *out);
/* segment table, n_segments x 4 int64: p0, p1, groups, grp_off_pos */
int mdf_plan_segments(const mdf_plan *plan, int64_t *out);
/* host arrays owned by the plan: row offsets of all chunks (B + n_chunks entries), group offsets of all segments
 * (B + n_segments entries); *count receives the length */
const int32_t *mdf_plan_chunk_row_off(const mdf_plan *plan, int64_t *count);
const int32_t *mdf_plan_grp_off(const mdf_plan *plan, int64_t *count);

/* Residue rows per fused chunk where the caller does not say (mdf_plan_create*, mdf_engine_config.max_rows = 0; the Python layer asks
 * mdf_default_chunk_rows()).  Rounds 1-5 ran 65 536; measured with the round-5 kernels (tools/chunk_sweep.sh, experiments/r05_chunk_rows_ab.sh):
 * 262 144 rows per launch take 1-3 % off a step on every workload of the bench (eight 256 x 256 tiles per GEMM workgroup instead of two,
 * a quarter of the launches), 1.6 GB of slabs per engine at 512 channels.  A chunk never exceeds the batch: small batches are unaffected. */
#define MDF_DEFAULT_CHUNK_ROWS 262144
int32_t mdf_default_chunk_rows(void);

/* --- engine --- */
typedef struct mdf_engine mdf_engine;
typedef struct {
    int32_t max_rows;            /* residue rows per fused chunk; 0 = MDF_DEFAULT_CHUNK_ROWS (multiples of 32768 are whole rounds of GEMM tiles) */
    int32_t nnz_per_row;         /* initial CSR capacity per residue row; 0 = 40 (6 A maps hold ~13, 10 A maps ~40) */
    double threshold;            /* contact threshold in Angstrom (cli.py:360-371 default 6.0) */
    int32_t generated_contacts;  /* contact_map_utils.pyx:44 generated_contacts (default 2) */
    int32_t max_segment_groups;  /* 0 = 1 << 20 */
    int32_t lm_batch;            /* proteins per LSTM group for heads with a language model; 0 = 16384 */
    double lm_workspace_gib;     /* LSTM time-major workspace budget; 0 = 48 */
    int32_t graph_max_chunks;    /* batches of at most this many chunks replay their launch sequence as ONE hipGraph from the
                                    third identical call on (same plan, buffers and outputs); 0 = 8, negative = never */
    int32_t pipeline_contact;    /* > 0 (engines without a language model): build the contact maps of chunk c+1 on a second,
                                    library-owned low-priority stream under the GraphConv stacks of chunk c (+0.8 % on the step; the
                                    aggregation kernel next to it runs 15 % slower); 0 = everything on the caller's stream (default) */
} mdf_engine_config;
/* models: n GO heads (GCN models of one device; heads with a language model must have it attached, mdf_model_attach_lm).
 * The models are not owned and must outlive the engine.  cfg may be NULL (all defaults, threshold 6.0, 2 generated contacts). */
int mdf_engine_create(mdf_model *const *models, int32_t n_models, int device, const mdf_engine_config *cfg, mdf_engine **out);
void mdf_engine_free(mdf_engine *e);
/* Raise the CSR capacity per row (after MDF_ECAPACITY from mdf_engine_check). */
int mdf_engine_set_nnz_per_row(mdf_engine *e, int32_t nnz_per_row);
int64_t mdf_engine_nnz_capacity(const mdf_engine *e);   /* CSR entries the current buffers hold (0 before the first call) */

/* One batch resident on the device (all pointers DEVICE memory owned by the caller, int32 descriptors as in the
 * residue-row layout above; the plan's own arrays are uploaded by the engine).  coords .. aln_off may be NULL for
 * batches that only ever take the dense-map path. */
typedef struct {
    int32_t B;
    const char *seqs;            /* packed query sequences */
    const int32_t *seq_off;      /* (B+1) */
    const int32_t *Lq;           /* (B) */
    const float *coords;         /* packed (sum Lt, 3) */
    const int32_t *coord_off;    /* (B+1) */
    const char *q_aln, *t_aln;   /* packed gapped strings */
    const int32_t *aln_off;      /* (B+1) */
    int32_t *status;             /* (n_chunks, 4) zero-initialised by the caller: see mdf_cmap_csr_dev */
    int64_t *bad;                /* (n_chunks) initialised to -1 by the caller: see mdf_seq_encode_dev */
} mdf_batch_dev;

/* The fused path: C-alpha coordinates + gapped alignments + sequences -> GO scores of every head.
 * scores[k]: DEVICE (B, T_k) f32 of head k (order of mdf_engine_create); logits: NULL, or per head NULL / DEVICE (B, 2 T_k).
 * Asynchronous on `stream`; call mdf_engine_check before trusting the result. */
int mdf_engine_forward_alignments(mdf_engine *e, const mdf_plan *plan, const mdf_batch_dev *batch, float *const *scores,
                                  float *const *logits, void *stream);
/* The reference-format path: one dense (Lq,Lq) contact map per protein (what build_align_contact_map returns) in HOST
 * memory, all of cmap_dtype (MDF_DT_I32 or MDF_DT_F32), uploaded chunk by chunk.  Synchronises `stream` before returning.
 * Footprint: the engine keeps TWO pinned host slots and TWO device slots of (sum of Lq^2 over a chunk's proteins) x 4 B each, about
 * chunk rows x Lq x 4 B: plan this path with MDF_DENSE_CHUNK_ROWS rows per chunk (128 MiB per slot at Lq = 512, 1 GiB at Lq = 4 096),
 * not with MDF_DEFAULT_CHUNK_ROWS (four times that, and the first chunk's host copy overlaps nothing); the Python layer
 * (HotPathEngine.forward_dense) re-plans a batch that was planned with larger chunks. */
#define MDF_DENSE_CHUNK_ROWS 65536
int mdf_engine_forward_dense(mdf_engine *e, const mdf_plan *plan, const mdf_batch_dev *batch, const void *const *cmaps_host,
                             int cmap_dtype, float *const *scores, float *const *logits, void *stream);
/* Synchronise `stream` and report what the asynchronous stages flagged, in the order the per-call API would raise it:
 * MDF_EBADCHAR  -- info[0] = protein, info[1] = position of the FIRST invalid residue of the batch (predict.pyx:36-46);
 * MDF_EINVAL    -- a query longer than the contact stage was sized for (info[2] = its length);
 * MDF_ECAPACITY -- CSR overflow, info[3] = entries the fullest chunk needs (raise mdf_engine_set_nnz_per_row and re-run with
 *                  re-initialised flags).  info may be NULL. */
int mdf_engine_check(mdf_engine *e, const mdf_plan *plan, const mdf_batch_dev *batch, void *stream, int64_t info[4]);
/* Language-model features (LSTM2 output) of every protein, packed (sum Lq, H) f32 in HOST memory, for LM `which` of the
 * engine's distinct language models -- inspection and tests. */
int mdf_engine_lm_features_host(mdf_engine *e, const mdf_plan *plan, const mdf_batch_dev *batch, int32_t which, float *out,
                                void *stream);
int32_t mdf_engine_num_lms(const mdf_engine *e);
/* Diagnostic: forward calls replayed as a hipGraph / issued launch by launch since the engine was made. */
int mdf_engine_graph_stats(const mdf_engine *e, int64_t *graph_launches, int64_t *eager_runs);
/* Diagnostic: synchronise `stream` and return the CSR entries of the chunk processed last (rowptr[R]); negative = error. */
int64_t mdf_engine_last_chunk_nnz(mdf_engine *e, void *stream);

/* The same for sequence-only CNN models (mdf_cnn): the batched counterpart of the reference's CNN loop over the proteins without
 * a structural hit (pipeline.py:600-648).  batch: seqs / seq_off / Lq / status / bad are used.  scores[k]: DEVICE (B, T_k). */
typedef struct mdf_seq_engine mdf_seq_engine;
int mdf_seq_engine_create(mdf_cnn *const *models, int32_t n_models, int device, mdf_seq_engine **out);
void mdf_seq_engine_free(mdf_seq_engine *e);
int mdf_seq_engine_forward(mdf_seq_engine *e, const mdf_plan *plan, const mdf_batch_dev *batch, float *const *scores, void *stream);
int mdf_seq_engine_check(mdf_seq_engine *e, const mdf_plan *plan, const mdf_batch_dev *batch, void *stream, int64_t info[4]);

/* Everything in one call with HOST buffers: plan, upload, fused forward, validation (one automatic retry with a larger
 * CSR capacity), download.  seqs / q_aln / t_aln are packed byte strings with per-protein lengths Lq / La / La; coords is
 * packed (sum Lt, 3) f32.  scores_host[k]: (B, T_k) f32.  Errors as mdf_engine_check (info may be NULL).
 * This is the single-call batched entry SURVEY.md section 8b lists (mdf_cmap_batch + mdf_gcn_forward_batch). */
int mdf_engine_run_alignments_host(mdf_engine *e, const char *seqs, const int32_t *Lq, int32_t B, const float *coords,
                                   const int32_t *Lt, const char *q_aln, const char *t_aln, const int32_t *La,
                                   float *const *scores_host, int64_t info[4]);
/* The same call as a pipeline of two slots (round 6): submit packs the batch into pinned staging, sends it over on a copy stream and
 * enqueues plan + fused forward + the download into pinned memory on engine-owned streams -- and returns a ticket; collect waits for
 * that batch, validates it (one automatic re-run with a larger CSR capacity) and copies the scores out, in the caller's order.  While
 * batch k computes, the caller packs batch k + 1 and unpacks batch k - 1: host lists -> host arrays at the rate of the device-resident
 * path.  At most TWO batches are in flight: a third submit fails with MDF_EINVAL until the oldest is collected; every ticket must be
 * collected (errors included: the slot is free afterwards).  mdf_engine_run_alignments_host == submit + collect: the same bits.  The
 * input arrays are copied before submit returns. */
int mdf_engine_submit_alignments_host(mdf_engine *e, const char *seqs, const int32_t *Lq, int32_t B, const float *coords,
                                      const int32_t *Lt, const char *q_aln, const char *t_aln, const int32_t *La, int64_t *ticket);
int mdf_engine_collect_host(mdf_engine *e, int64_t ticket, float *const *scores_host, int64_t info[4]);

/* Output stage next to the path (mDeepFRI/pipeline.py:696-705, 733-740: results.tsv keeps, per protein, the terms with
 * float(score) >= 0.1 sorted by score descending; Python's sort is stable, so equal scores keep term order).
 * scores: (B, T) f32.  offsets: (B+1) int32 out; term_idx / kept_scores: `capacity` entries, protein p's terms at
 * [offsets[p], offsets[p+1]).  status: device int32[4], zero-initialised: [0] != 0 -> capacity too small (needed count
 * in [1]).  T <= 8192.  workspace: mdf_filter_workspace_bytes(B). */
size_t mdf_filter_workspace_bytes(int32_t B);
int mdf_filter_scores_dev(const float *scores, int32_t B, int32_t T, float threshold, int32_t *offsets, int32_t *term_idx,
                          float *kept_scores, int64_t capacity, int32_t *status, void *workspace, size_t workspace_bytes,
                          void *stream);

/* The text of results.tsv for one GO head from the filter's arrays (host buffers), one line per kept (protein, term) as
 * mDeepFRI/pipeline.py:713-716 / 745-748 writes it:
 *     query_id \t <middle> \t term \t f"{score:.4f}" \t go_name \t <tail> \n
 * middle = "net_type\tmode label" (NUL-terminated); qid / term / name: concatenated strings with (count + 1) offsets; tail = the six
 * alignment fields of the protein joined by tabs -- per protein with tail_off (B + 1 offsets), or ONE NUL-terminated string for all
 * (tail_off == NULL: the reference's six "nan" when the query has no alignment record).  The score is printed as Python prints
 * float(np.float32) with ".4f" (correctly rounded, ties to even).  *bytes = size of the text, *lines = number of lines; MDF_ECAPACITY
 * when it exceeds `capacity` (call with out = NULL, capacity = 0 to size the buffer).  Host code: no device is touched. */
/* Rows of the prediction matrix (mDeepFRI/pipeline.py:318-319: csv.writer(delimiter="\t").writerow([query_id, net_type] + pred.tolist())):
 * prefix = B strings "query_id\tnet_type" (already csv-quoted where an id needs it) with B + 1 offsets; scores (B, T) f32; every score
 * printed as repr(float(np.float32(s))) -- the shortest digits that read back as the same double, CPython's layout --, rows end "\r\n"
 * (csv's default).  *bytes = size of the text; MDF_ECAPACITY when it exceeds `capacity` (out = NULL, capacity = 0 sizes the buffer;
 * B * (T * 25 + 2) + the prefixes always suffices).  threads: host threads the rows are dealt to (0 = up to 32, as the machine has).
 * Host code: no device is touched. */
int mdf_matrix_format_host(const char *prefix, const int64_t *prefix_off, const float *scores, int32_t B, int32_t T, char *out, int64_t capacity,
                           int threads, int64_t *bytes);
int mdf_results_format_host(const char *qid, const int64_t *qid_off, const char *middle, const char *term, const int64_t *term_off, const char *name,
                            const int64_t *name_off, const char *tail, const int64_t *tail_off, const int32_t *offsets, const int32_t *term_idx,
                            const float *kept, int32_t B, int32_t T, char *out, int64_t capacity, int64_t *bytes, int64_t *lines);

/* ------------------------------------------------------------------------------------------------
 * Alignment step in front of the path: global Needleman-Wunsch with affine gaps, the
 * arithmetic the reference obtains from PyOpal at mDeepFRI/alignment.py:164-250
 *     best_hit_database (:164-196)  aligner.align(query, database, mode="score", algorithm="nw")  -> mdf_nw_score_*
 *     align_pairwise    (:198-221)  aligner.align(query, [target], mode="full", algorithm="nw")    -> mdf_nw_align_*
 * Sequences are residue CODES (indices into the scoring matrix alphabet, < A <= 32), packed: sequence s occupies
 * codes[seq_off[s] .. seq_off[s] + seq_len[s]).  A pair p aligns query sequence pair_q[p] with target sequence pair_t[p].
 * matrix: (A, A) int32 row-major, matrix[q][t].  A gap of length n costs gap_open + (n - 1) * gap_extend (Opal's model).
 * Operations: 'M' match, 'X' mismatch, 'D' query residue against a target gap, 'I' target residue against a query gap --
 * the letters reference insert_gaps (alignment.py:38-62) consumes.  Which of several CO-OPTIMAL alignments is returned is set
 * by `tie_rule` (3 bits; oracle/nw_oracle.c): bit 0 = at H a gap move wins a tie against the diagonal, bit 1 = 'I' wins a tie
 * against 'D', bit 2 = extending a gap wins a tie against opening one; 0 (diagonal, then 'D', then 'I'; opening first) is the
 * default.  Integer work, bit-exact with that oracle for every rule.  PyOpal itself is not available offline: which rule
 * reproduces Opal's traceback is unpinned -- scores, best hits and unique optima do not depend on it.
 * ---------------------------------------------------------------------------------------------- */

/* Host helper: per-pair offsets into the three per-pair buffers, P + 1 entries each (entry P = total; any may be NULL):
 * bnd_off in int32 elements (2 * Lq per pair), trace_off in bytes, ops_off in bytes (capacity Lq + Lt per pair). */
int mdf_nw_plan(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P, int64_t *bnd_off, int64_t *trace_off,
                int64_t *ops_off);

/* Host helper: length of the leading run of LARGE pairs (>= 768 x 768 cells) of a pair list ordered by decreasing size -- the
 * n_long argument of mdf_nw_score_dev.  Large pairs get a whole workgroup each (their 64-column strips run as a pipeline sixteen
 * waves deep, the column between two strips handed over through LDS); with a single wave a 2 000 x 2 000 pair would outlive the
 * rest of the launch by milliseconds.  0 is always valid. */
int32_t mdf_nw_count_long(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P);
/* The same for mdf_nw_align_dev: full alignments are asked for far fewer pairs (one winner per query), so the launch lasts as
 * long as its longest serial chain and the workgroup form pays from 256 x 256 cells on. */
int32_t mdf_nw_count_long_align(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P);

/* Scores of P pairs (device pointers; bnd: int32 workspace of bnd_off[P] elements).  Pairs [0, n_long): one workgroup per pair;
 * the others: one wave per pair. */
int mdf_nw_score_dev(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t,
                     int32_t P, int32_t n_long, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, const int64_t *bnd_off,
                     int32_t *bnd, int32_t *scores, void *stream);

/* Full alignments of P pairs (device pointers).  trace: trace_off[P] bytes of workspace.  Pair p's operations are written to
 * ops[ops_off[p+1] - op_len[p] .. ops_off[p+1]) (right-aligned inside the pair's capacity); q_aln / t_aln (optional, same
 * layout) receive the gapped query / target strings spelled with `alphabet` (A letters, device) -- what insert_gaps would
 * build.  n_match[p] = number of 'M' (identity = n_match / op_len, alignment.py:214); coverages of a global alignment are 1. */
int mdf_nw_align_dev(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t,
                     int32_t P, int32_t n_long, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int tie_rule,
                     const char *alphabet, const int64_t *bnd_off, int32_t *bnd, const int64_t *trace_off, uint8_t *trace, const int64_t *ops_off, char *ops, char *q_aln,
                     char *t_aln, int32_t *op_len, int32_t *n_match, int32_t *scores, void *stream);

/* Score mode only: NW(q, t; S) = NW(t, q; S^T), so for a SYMMETRIC matrix a pair may be swept with either sequence as the rows; this turns
 * every pair of the list to the orientation that takes fewer steps on the device (in place; call it before mdf_nw_plan / mdf_nw_score_dev;
 * the host entries do it themselves).  Returns the number of pairs turned round (0 for an asymmetric matrix), negative on bad arguments. */
int32_t mdf_nw_orient_pairs(const int32_t *seq_len, int32_t *pair_q, int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int gap_open,
                            int gap_extend);

/* The same with host buffers (upload, run, download): n_seq sequences, P pairs; ops / q_aln / t_aln hold sum(Lq + Lt) bytes
 * laid out as mdf_nw_plan's ops_off says; alphabet is a NUL-terminated string of A letters. */
int mdf_nw_score_host(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const int32_t *pair_q,
                      const int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int32_t *scores);
int mdf_nw_align_host(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const int32_t *pair_q,
                      const int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int tie_rule,
                      const char *alphabet, char *ops, char *q_aln, char *t_aln, int32_t *op_len, int32_t *n_match, int32_t *scores);

/* Best hit of every query and its alignment in ONE call: the batched counterpart of the reference's per-query pool task
 * (alignment.py:223-250 `pairwise_against_database` = `best_hit_database` :164-196 + `align_pairwise` :198-221, mapped over the
 * queries by the pool of :266-320).  Host buffers in, host buffers out; everything between (launch order, the arg-max over a query's
 * candidates, the plans of the winners' alignments, packing in query order) happens inside.
 *   sequences: n_seq byte strings text[seq_off[s] .. seq_off[s] + seq_len[s]), offsets ascending.  lut != NULL: the bytes are residue
 *     LETTERS, translated on the device through the 256-entry table (255 = not in the alphabet: MDF_EBADCHAR, info[0] = sequence,
 *     info[1] = position of the first offender in sequence order); lut == NULL: they are residue codes < A.
 *   queries are sequences 0 .. nq-1; the candidates of query q are the sequences cand[first[q] .. first[q+1]) (at least one each).
 *   out, per query: best = position of the winner inside q's candidate list (the FIRST maximum, Python's max()), score, op_len (columns),
 *     n_match, aln_off (nq + 1); ops / q_aln / t_aln packed in query order, query q's columns at [aln_off[q], aln_off[q+1]).
 *     capacity = bytes available in each of ops / q_aln / t_aln (sum over q of Lq + the longest candidate always suffices); too small:
 *     MDF_ECAPACITY with info[2] = bytes needed.  cand_scores (optional, P = first[nq] ints): every candidate's score.
 *   max_trace_bytes: device memory the direction words of one alignment launch may take (the winners are aligned in groups).
 *   ws: what a caller keeps between calls (stream, growable device scratch, pinned staging) on the device it was created for; NULL: the
 *     calling thread's own on the current device (own stream; kept for the life of the process). */
typedef struct mdf_nw_workspace mdf_nw_workspace;
/* stream: the HIP stream the aligner's launches and copies go to (they then sit in THAT stream's order, e.g. between two GCN batches of
 * the same stream), or NULL for a non-blocking, highest-priority stream owned by the workspace. */
int mdf_nw_workspace_create(int device, void *stream, mdf_nw_workspace **out);
void mdf_nw_workspace_free(mdf_nw_workspace *ws);
/* The same call in three steps, for callers that keep several batches in flight (mDeepFRI.stream.QueryStream): `begin` and `align` only
 * ENQUEUE (upload + scores + arg-max; alignments of the winners + packing + the copies back into pinned memory), `align` first waits for
 * the scores of its `begin`, `finish` for the alignments of its `align` -- each for work enqueued one step earlier, so a caller that
 * interleaves the steps of consecutive batches never waits for the device.  The caller's input buffers are not read after `begin`
 * returns.  One call in flight per workspace; `abandon` drops it (after an error between the steps).  Arguments as below. */
int mdf_nw_best_hits_begin(mdf_nw_workspace *ws, const uint8_t *text, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const uint8_t *lut,
                           int32_t nq, const int32_t *cand, const int64_t *first, const int32_t *matrix, int32_t A, int gap_open, int gap_extend,
                           int tie_rule, const char *alphabet, int64_t max_trace_bytes, int want_cand_scores);
int mdf_nw_best_hits_align(mdf_nw_workspace *ws, int64_t *info);
int mdf_nw_best_hits_finish(mdf_nw_workspace *ws, int32_t *best, int32_t *score, int32_t *op_len, int32_t *n_match, int64_t *aln_off, char *ops, char *q_aln,
                            char *t_aln, int64_t capacity, int32_t *cand_scores, int64_t *info);
int mdf_nw_best_hits_abandon(mdf_nw_workspace *ws);
int mdf_nw_best_hits_host(mdf_nw_workspace *ws, const uint8_t *text, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const uint8_t *lut, int32_t nq,
                          const int32_t *cand, const int64_t *first, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int tie_rule,
                          const char *alphabet, int64_t max_trace_bytes, int32_t *best, int32_t *score, int32_t *op_len, int32_t *n_match,
                          int64_t *aln_off, char *ops, char *q_aln, char *t_aln, int64_t capacity, int32_t *cand_scores, int64_t *info);

/* Timing hook for bench.py: mdf_timing_enable(n), n = 0 off, n >= 1: the library brackets every n-th launch of each
 * kernel class with hipEvents on the stream it is launched on and accumulates count and milliseconds of the sampled
 * launches (read after a sync).  An event pair costs GPU time between kernels (~6 % of the step when every launch is
 * timed), hence the sampling.
 * kernel: "ax" (A.X aggregation, every GraphConv layer pooled), "ax2" / "ax3" (the launches of layer 2 / of layer 3 and up on their own: the
 * two layers find their input in different levels of the memory system), "gemm" (H.W fp32 MFMA, layers 2..3 pooled), "gemm2" / "gemm3"
 * (per layer), "gemm1" (layer-1 S.T1 GEMM, K=32), "cmap" (fused contact map), "head",
 * "lstm" / "lstm2" (one time step of language-model layer 1 / layer 2; the two run concurrently on two streams, so their
 * durations include the contention), "embed" (language-model embedding GEMM), "cnn" (sequence-only CNN: conv + max pool). */
int mdf_timing_enable(int on);
int mdf_timing_read(const char *kernel, int64_t *launches, double *total_ms);
int mdf_timing_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* MDFRI_H */
