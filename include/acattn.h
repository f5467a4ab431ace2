/*
 * acattn.h -- C ABI of the MI355X-native calibrated multi-head self-attention core of AC-TSR.
 *
 * The reference (AIM-SE/AC-TSR, a RecBole fork) has no native / FFI boundary: its hot path is a
 * chain of ATen ops dispatched from Python.  This header defines the boundary
 * a maintainer would bind instead: plain pointers + sizes, no torch types.  Every entry point names
 * the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all tensors fp32, contiguous, row-major, resident in device (HBM) memory of the current HIP
 *     device; ids / validity flags uint8;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work,
 *     never synchronise, never allocate (hipGraph-capturable);
 *   - inputs are borrowed for the duration of the enqueued work and never written;
 *     outputs are caller-allocated;
 *   - return value 0 = ok; negative = argument error (nothing enqueued), see acattn_last_error();
 *     positive = hipError_t from the launch.
 *
 * Shapes: B batch, L sequence length, H hidden size, nh heads, dh = H / nh.
 * Supported: dh in {16, 32, 64, 128}, 1 <= L <= 208.
 */
#ifndef ACATTN_H_
#define ACATTN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACATTN_ABI_VERSION 29

/* attention-mask encodings (recbole/model/abstract_recommender.py:136-143 builds the dense form) */
enum {
  ACATTN_MASK_STRUCTURED = 0, /* key_valid[B,L] (item_seq != 0) + `causal` flag; 0 / -10000 derived in-kernel */
  ACATTN_MASK_DENSE_LL = 1,   /* additive mask[B,L,L]  (the reference's [B,1,L,L]) */
  ACATTN_MASK_DENSE_L = 2     /* additive mask[B,L]    (the reference's [B,1,1,L], AcBERT4Rec acbert4rec.py:173) */
};

/* combine_option of AttackRTransformerLayer.combine_attention (recbole/model/layers.py:883-896) */
enum { ACATTN_COMBINE_FIXED = 0, ACATTN_COMBINE_GATE = 1, ACATTN_COMBINE_ANNEALING = 2 };

/* rich_calibrated_combine, used only when two_level == 0 (recbole/model/layers.py:929-936) */
enum { ACATTN_RICH_NONE = 0, ACATTN_RICH_FIXED = 1, ACATTN_RICH_TRAINABLE = 2 };

/* source of the layer's randomness (recbole/model/layers.py:672,736,917) */
enum {
  ACATTN_RNG_EXPLICIT = 0, /* noise / keep masks are read from the tensors below (parity mode) */
  ACATTN_RNG_COUNTER = 1   /* generated in-kernel from (seed, element index); reproducible, see acattn_rng_materialize */
};

#define ACATTN_NSTAT 8 /* floats per (b, head, query row) in `row_stats` */

typedef struct acattn_problem {
  int32_t B, L, H, n_heads;
  /* projected activations, [B,L,H] each */
  const float* q;  /* mixed_query_layer  = query(x)                      layers.py:687 */
  const float* k;  /* mixed_key_layer    = key(x)                        layers.py:688 */
  const float* v;  /* mixed_value_layer  = value(x)                      layers.py:689 */
  const float* qa; /* attack_query_transform(mixed_query)  (NULL if !adversarial)   layers.py:658 */
  const float* ka; /* attack_key_transform(mixed_key)      (NULL if !adversarial)   layers.py:659 */
  const float* gate_logits; /* gate(mixed_query) BEFORE the sigmoid, [B,L,L]; combine GATE only   layers.py:887 */
  /* attention mask */
  int32_t mask_mode;        /* ACATTN_MASK_* */
  int32_t causal;           /* STRUCTURED only: 1 = tril (SASRec), 0 = bidirectional */
  const uint8_t* key_valid; /* STRUCTURED: [B,L] */
  const float* mask;        /* DENSE_*: additive mask */
  /* spatial calibrator parameters (device pointers; NULL weight = term disabled)  layers.py:636-640 */
  const float* w_order; /* order_affine.weight    [2*dh] */
  const float* b_order; /* order_affine.bias      [1]    */
  const float* w_dist;  /* distance_affine.weight [2*dh] */
  const float* b_dist;  /* distance_affine.bias   [1]    */
  const float* scalar;  /* scalar                 [1]    */
  /* adversarial calibrator options */
  int32_t adversarial; /* 0 = spatial calibrator only: ctx_calibrated = after_spatial . V, nothing else written */
  int32_t combine_option; /* ACATTN_COMBINE_* */
  float anneal_rate;      /* ANNEALING: exp(-step/1e5), computed by the caller (layers.py:890) */
  int32_t two_level;      /* layers.py:911-914 */
  int32_t rich_combine;   /* ACATTN_RICH_* (two_level == 0 only) */
  const float* rich_ratio; /* TRAINABLE: rich_calibrated_combine_ratio [1] */
  /* randomness */
  int32_t rng_mode; /* ACATTN_RNG_* */
  float p_drop;     /* attn_dropout_prob; 0 = no dropout (eval mode) */
  const float* noise;         /* EXPLICIT: [B,nh,L,L] standard normal draws          layers.py:917 */
  const uint8_t* keep_after;  /* EXPLICIT: [B,nh,L,L] 0/1 keep mask or NULL           layers.py:736 (after_spatial) */
  const uint8_t* keep_before; /* EXPLICIT: [B,nh,L,L] or NULL                        layers.py:736 (before_spatial) */
  const uint8_t* keep_mask;   /* EXPLICIT: [B,nh,L,L] or NULL                        layers.py:672 */
  uint64_t seed;              /* COUNTER */
  const uint64_t* seed_device; /* COUNTER, optional: *seed_device (device memory) is added to `seed` when the kernel
                                  starts -- lets a captured hipGraph draw fresh randomness on every replay */
  /* ---- optional per-row products of the PRODUCER of q / k / gate (ABI 26) ------------------------------------------
   * Both are O(L H) work that belongs with the projections; the core re-derives them 2-4 times per sequence when they
   * are absent (once per head for the gate, once per query block for the key halves).  acattn_projections_fwd writes
   * both; acattn_spatial_affines() computes `affine` from q, k and the calibrator parameters for any other producer. */
  const float* affine;  /* NULL, or [B,nh,4,LP] with LP = 16*ceil(L/16): the rank-1 halves of the two spatial affines
                           of layers.py:705-708, affine(q_i || k_j) = q_i.w[:dh] + k_j.w[dh:] + b:
                             plane 0 [i] = -log2(e) * (q_i . w_order[:dh] + b_order)     (sigmoid(o) = 1/(1+exp2(p0+p2)))
                             plane 1 [i] =             q_i . w_dist[:dh]  + b_dist
                             plane 2 [j] = -log2(e) *  k_j . w_order[dh:]
                             plane 3 [j] =             k_j . w_dist[dh:]
                           entries L..LP-1 must be finite (zeros). */
  int32_t gate_is_prob; /* non-zero: `gate_logits` holds sigmoid(gate(mixed_query)) (layers.py:887: one sigmoid per
                           (b, i, j), shared by the heads) instead of the logits; d gate_logits of the backward is
                           still the gradient of the LOGITS */
} acattn_problem;

/* Non-finite inputs: a NaN in a valid row of q, k, v, qa, ka or the gate comes out as NaN in the context rows (and, for qa /
 * ka, the attack-mask rows) that row feeds, in every forward kernel -- the streaming kernel's translation units are built with
 * -fno-honor-nans (no canonicalising v_max in front of the row maxima), which does not change that
 * (tests/test_hip_onehop.py::test_a_nan_in_the_inputs_reaches_the_outputs_of_the_streaming_forward).  Rows of PADDED positions
 * (key_valid == 0) are never read by a masked soft-max here, whereas the reference's additive -10000 would propagate a NaN
 * sitting there. */
typedef struct acattn_fwd_out {
  float* ctx_attacked;   /* [B,L,H] head-merged perturbed_attention . V        layers.py:677-680 via :938 */
  float* ctx_calibrated; /* [B,L,H] head-merged combined attention . V         layers.py:677-680 via :942 */
  float* attack_mask;    /* [B,nh,L,L] M as returned by the layer (after dropout)  layers.py:915,951 */
  float* row_stats;      /* [B,nh,L,ACATTN_NSTAT] log-normalisers for the backward, or NULL */
  /* optional probability dumps, [B,nh,L,L] each, NULL = skip (return_all_attention_prob, layers.py:899-927) */
  float* after_spatial;
  float* before_spatial;
  float* perturbed_attention;
  float* calibrated_attention;
  /* ABI 26, optional (adversarial form only): [B, nh, ceil(L/16)] -- acattn_mask_penalty_rows of attack_mask, i.e.
   * sum (1 - M)^2 over each query block's rows and all L keys.  The long-sequence streaming forward forms the sums from
   * the M block it stages for its store; for every other kernel the entry point runs acattn_mask_penalty_rows behind the
   * launch.  Always filled when given. */
  float* penalty_part;
} acattn_fwd_out;

typedef struct acattn_bwd_io {
  /* forward results the backward re-uses */
  const float* attack_mask; /* [B,nh,L,L] as written by the forward */
  const float* row_stats;   /* [B,nh,L,ACATTN_NSTAT] */
  /* cotangents (NULL = zero) */
  const float* d_ctx_attacked;   /* [B,L,H] */
  const float* d_ctx_calibrated; /* [B,L,H] */
  const float* d_attack_mask;    /* [B,nh,L,L] */
  /* gradients, caller-allocated, fully overwritten */
  float* dq;  /* [B,L,H] */
  float* dk;  /* [B,L,H] */
  float* dv;  /* [B,L,H] */
  float* dqa; /* [B,L,H] */
  float* dka; /* [B,L,H] */
  float* dgate_logits; /* [B,nh,L,L] per-head partials of d gate_logits (combine GATE; the caller sums over heads) or NULL */
  /* per-(b,head) partial sums of the small parameters; the caller reduces over (B, nh) */
  float* dw_order_part; /* [B,nh,2*dh] */
  float* dw_dist_part;  /* [B,nh,2*dh] */
  float* dsmall_part;   /* [B,nh,4]: d b_order, d b_dist, d scalar, d rich_ratio */
  int32_t part_stride;  /* row stride (floats) of the three partial buffers; 0 = dense (2*dh, 2*dh, 4).  With a
                           common stride the three may be column ranges of ONE [B*nh, stride] buffer, which the
                           caller then reduces in a single pass */
  const uint32_t* active_qblocks; /* optional hint [B]: bit q set = some query row in [16q, 16q+16) of that sequence
                           has a non-zero cotangent (d_ctx_*); blocks with a clear bit are skipped and their dq, dqa,
                           gate partials written as zeros.  With d_attack_mask given no block is skipped, but one with a
                           clear bit only owes the soft-max of the mask scores (dqa, dka).  NULL = all active.
                           A hint, not a mask: a kernel may ignore it (the skipped work multiplies zeros anyway). */
  int32_t attack_only;  /* non-zero: the caller will read ONLY dqa and dka (pass 2 of the two-pass trainer through a
                           layer with no attack transform upstream, recbole/trainer/trainer.py:678-684); every other
                           output buffer must still be valid memory but may be left unwritten.  A hint like the above. */
  void* workspace;      /* optional device scratch of acattn_calibrated_attention_bwd_workspace_bytes(p) bytes (contents
                           irrelevant).  With it, long sequences (L > 64) take the streaming two-kernel backward
                           (acattn_bwd_stream.hip); NULL = the row-resident kernels only. */
  const int64_t* read_rows; /* the same hint as active_qblocks in its raw form: [B, n_read_rows] positions whose context
                           cotangents are the only non-zero ones (item_seq_len - 1 of abstract_recommender.py:130-134).
                           Consulted when active_qblocks is NULL; NULL = all active.  PRECONDITION: 0 <= position < L
                           (the reference's gather raises an index error otherwise); the kernels clamp a position
                           outside that range into it rather than touch another sequence's memory. */
  int32_t n_read_rows;
  const float* d_penalty_part; /* ABI 26, optional: [B, nh, ceil(L/16)] cotangent of acattn_mask_penalty_rows' sums.  The
                           kernels then add d M = 2 * d_penalty_part[b, head, query block] * (M - 1) to the mask cotangent
                           from the M tile they rebuild -- the mask penalty's gradient (acsasrec.py:131-137) without a
                           dense [B,nh,L,L] cotangent.  Like d_attack_mask it keeps every query block active. */
  int32_t dgate_summed; /* ABI 27, request: `dgate_logits` is [B,L,L] and receives the gate-logit gradient SUMMED over the heads
                           (the gate is shared by them, layers.py:887 unsqueeze(1)) instead of [B,nh,L,L] per-head partials.
                           Honoured by the one-row form of the backward only (one read position per sequence,
                           acattn_bwd_io.read_rows with n_read_rows == 1: of each sequence's L x L gate gradient ONE row is
                           non-zero, and writing nh dense partials of it -- 164 MB of zeros at B = 512, L = 200, 2 heads --
                           plus summing them was most of that launch): ask acattn_calibrated_attention_bwd_gate_summed()
                           first; a launch that cannot honour the request fails instead of misreading the buffer. */
  /* ABI 27, optional: a SECOND cotangent set evaluated in the same launch (the single-pass combined backward of the two-pass
   * protocol, recbole/trainer/trainer.py:672-686: the calibrated loss's cotangents above, the attacked loss's here).  The
   * backward is linear in its cotangents and both sets need the same rebuilt probability tiles; the second set is taken in
   * the form the attacked loss has in a layer with no attack transform upstream: d ctx_calibrated2 [B,L,H] and / or
   * d_penalty_part2 [B,nh,ceil(L/16)], no attacked-context and no dense mask cotangent, and of its gradients dqa2, dka2
   * ([B,L,H], fully overwritten) alone.  Honoured by the streaming kernels (needs `workspace`) when the FIRST set has no
   * d_ctx_attacked either and no block hints are given: ask acattn_calibrated_attention_bwd_pair_supported() first; a launch
   * that cannot honour a non-NULL dqa2 fails. */
  const float* d_ctx_calibrated2;
  const float* d_penalty_part2;
  float* dqa2;
  float* dka2;
} acattn_bwd_io;

/* Full-catalogue cross-entropy: ACSASRec._cal_loss for loss_type 'CE',
 * recbole/model/sequential_recommender/acsasrec.py:117-120, without materialising the [B, N] logits. */
typedef struct acattn_ce_problem {
  int32_t B, N, H;       /* rows (sequences), catalogue size, hidden size (64, 128 or 256) */
  const float* out;      /* [B,H] sequence representations (attacked_output / calibrated_output)  acsasrec.py:101-103 */
  const float* table;    /* [N,H] item_embedding.weight                                           acsasrec.py:117 */
  const int64_t* target; /* [B]   pos_items                                                      acsasrec.py:108 */
  /* backward only: how `coef` (d loss / d row_loss) is read.  coef_is_scalar: coef[0] holds for every row (the
   * cotangent of a mean, nn.CrossEntropyLoss's default reduction: acsasrec.py:119); coef_scale multiplies it
   * (1 / B of that mean), 0 = 1. */
  int32_t coef_is_scalar;
  float coef_scale;
} acattn_ce_problem;

/* Bytes of scratch both CE entry points need (caller-allocated device memory, contents irrelevant). */
int64_t acattn_full_sort_ce_workspace_bytes(const acattn_ce_problem* p);

/* lse[b] = logsumexp_n(out_b . table_n);  row_loss[b] = lse[b] - out_b . table_target(b).
 * mean(row_loss) == CrossEntropyLoss(out @ table^T, target)   (acsasrec.py:118-120). */
int acattn_full_sort_ce_fwd(const acattn_ce_problem* p, void* workspace, float* lse, float* row_loss, void* stream);

/* The forward plus dir[b] = softmax_b . table - table_target(b) = d row_loss[b] / d out_b in ONE sweep of the table
 * (flash-attention with K = V = the item table): for a loss whose table gradient is never taken -- the attacked loss
 * of the two-pass trainer (recbole/trainer/trainer.py:678-684) -- the backward is then d_out = coef[:, None] * dir, an
 * elementwise product, and the second sweep of acattn_full_sort_ce_bwd disappears.  Returns -100 (no error text)
 * when B is too large for its per-workgroup slabs (~96 MB); use _fwd and _bwd then. */
int acattn_full_sort_ce_fwd_dir(const acattn_ce_problem* p, void* workspace, float* lse, float* row_loss, float* dir,
                                void* stream);

/* Gradients of sum_b coef[b] * row_loss[b]: d_out [B,H] always; d_table [N,H] (fully overwritten) unless NULL. */
int acattn_full_sort_ce_bwd(const acattn_ce_problem* p, const float* lse, const float* coef, void* workspace,
                            float* d_out, float* d_table, void* stream);

/* [ABI 28] How the three products of the full-catalogue cross-entropy are evaluated at hidden 64 (acattn_ce_bf16.hip).
 * gfx950's fp32 matrix instruction runs at 1/16 of the bf16 one, so the default splits every fp32 operand exactly
 * into three bf16 numbers and evaluates a product as the six bf16 MFMAs whose dropped remainder is below 2^-23 of
 * |a||b| -- one fp32 rounding; results are as close to fp64 as the exact-fp32 kernels' (tests/test_hip_ce.py).
 *   ACATTN_CE_PRODUCTS_FP32 (0)    exact fp32 MFMA everywhere (round 3's kernels; also ACATTN_CE_PRODUCTS=fp32)
 *   ACATTN_CE_PRODUCTS_DEFAULT (1) the split sweeps for catalogues of more than 65,536 items (waves of six item tiles: one
 *                                  round of workgroups up to 98,304 items, one round + leftover tiles up to 102,400 --
 *                                  the benchmark's 100,000 --, several rounds beyond), the fp32 kernels below
 *   ACATTN_CE_PRODUCTS_ALL (2)     the split sweeps for every catalogue size (tests)
 * Process-wide; returns the previous mode; any other value only queries.  Non-finite inputs: an infinite operand splits
 * into inf + nan, so an infinite logit becomes NaN instead of inf (the loss is non-finite either way). */
#define ACATTN_CE_PRODUCTS_FP32 0
#define ACATTN_CE_PRODUCTS_DEFAULT 1
#define ACATTN_CE_PRODUCTS_ALL 2
int acattn_full_sort_ce_products(int mode);

/* y = LayerNorm(dropout(z) + residual) * gamma + beta   (the tail of both sub-blocks
 * of a layer -- recbole/model/layers.py:681-683 and :794-796).  Rows of H in {64, 128, 256} floats. */
#define ACATTN_LN_BWD_GRID 512 /* workgroups of the backward == rows of its dgamma/dbeta partial buffer */
typedef struct acattn_ln_problem {
  int32_t rows, H;
  int32_t residual_rows;  /* == rows, or a divisor of it when the residual is broadcast over a leading dimension */
  const float* z;         /* [rows,H] output of the dense layer (bias included) */
  const float* residual;  /* [residual_rows,H] */
  const float* gamma;     /* [H] LayerNorm.weight */
  const float* beta;      /* [H] LayerNorm.bias */
  float eps;              /* layer_norm_eps */
  float p_drop;           /* hidden_dropout_prob in training, 0 in eval */
  const uint8_t* keep;    /* optional explicit keep mask [rows,H]; NULL = counter RNG from (seed, seed_device) */
  uint64_t seed;
  const uint64_t* seed_device;
} acattn_ln_problem;

/* stats[rows,2] receives (mean, 1/std) per row for the backward. */
int acattn_dropout_add_layernorm_fwd(const acattn_ln_problem* p, float* y, float* stats, void* stream);

/* dz [rows,H] and dres [rows,H] may each be NULL; dgb_part is [ACATTN_LN_BWD_GRID, 2, H] partial sums of
 * (dgamma, dbeta), or NULL; the caller adds the partials (and folds dres when the residual was broadcast). */
int acattn_dropout_add_layernorm_bwd(const acattn_ln_problem* p, const float* dy, const float* stats, float* dz,
                                     float* dres, float* dgb_part, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * y = dropout(LayerNorm(item_embedding[idx] + position_embedding[position])): the front end of both models
 * (recbole/model/sequential_recommender/acsasrec.py:87-95, acbert4rec.py:163-171).  Rows of H in {64, 128, 256}. */
#define ACATTN_EMBED_BWD_CHUNKS 8 /* the backward runs L x 8 workgroups: leading dimension of its partial buffers */
typedef struct acattn_embed_problem {
  int32_t rows, L, H;     /* rows = B*L; row r sits at position r % L */
  int64_t n_table_rows;   /* rows of `table` (ids outside [0, n) are clamped, never dereferenced out of bounds) */
  const int64_t* idx;     /* [rows] item ids (item_seq, row-major [B,L]) */
  const float* table;     /* [n_table_rows,H] item_embedding.weight */
  const float* pos;       /* [>=L,H] position_embedding.weight, or NULL (use_position_embedding: False) */
  const float* gamma;     /* [H] LayerNorm.weight */
  const float* beta;      /* [H] LayerNorm.bias */
  float eps;              /* layer_norm_eps */
  float p_drop;           /* hidden_dropout_prob in training, 0 in eval (the dropout FOLLOWS the norm here) */
  const uint8_t* keep;    /* optional explicit keep mask [rows,H]; NULL = counter RNG from (seed, seed_device) */
  uint64_t seed;
  const uint64_t* seed_device;
  uint8_t* nonzero_out;   /* optional [rows]: receives idx != 0, the key-validity bytes of the structured mask
                             (abstract_recommender.py:137: attention_mask = item_seq != 0); forward only */
  int64_t hot_id_plus1;   /* ABI 27, backward only, 0 = none: id + 1 of ONE table row that a large share of the lookups hit
                             (AcBERT4Rec's mask token, acbert4rec.py:105-150: 20 % of all positions).  Its gradient rows are
                             summed inside each workgroup first and added with one atomic per column and workgroup: 20,480
                             float atomics per address on one row were 1.1 ms at B = 512, L = 200, H = 256. */
} acattn_embed_problem;

/* stats[rows,2] receives (mean, 1/std) per row for the backward. */
int acattn_embed_layernorm_fwd(const acattn_embed_problem* p, float* y, float* stats, void* stream);

/* d_table [n_table_rows,H] must be ZERO-INITIALISED by the caller: rows are accumulated with float atomics and rows
 * whose id == padding_idx are skipped (nn.Embedding(padding_idx=0): acsasrec.py:33); pass padding_idx = -1 for none.
 * d_pos_part [ACATTN_EMBED_BWD_CHUNKS, L, H] (or NULL) and dgb_part [ACATTN_EMBED_BWD_CHUNKS * L, 2, H] (or NULL)
 * receive partial sums of the position-embedding and (dgamma, dbeta) gradients; the caller adds the partials. */
int acattn_embed_layernorm_bwd(const acattn_embed_problem* p, const float* dy, const float* stats, int64_t padding_idx,
                               float* d_table, float* d_pos_part, float* dgb_part, void* stream);

/* out[bt, c] = sum_r x[bt, r, c]  (x is [batch, R, C] contiguous).  The reductions of the training step's backward:
 * bias gradients (sum over B*L rows), split-K slabs, per-(b,head) parameter partials, per-head gate gradients. */
int acattn_sum_rows(const float* x, float* out, int32_t batch, int32_t R, int32_t C, void* stream);
/* Two such reductions in one launch (independent shapes): the attention backward's parameter partials and its per-head
 * gate gradient (layers.py:887: the gate is shared by the heads) are summed together. */
int acattn_sum_rows_pair(const float* x1, float* out1, int32_t batch1, int32_t R1, int32_t C1, const float* x2, float* out2,
                         int32_t batch2, int32_t R2, int32_t C2, void* stream);

/* ABI 27: cross-entropy over MATERIALISED logits [rows, N] (the masked-slot loss of AcBERT4Rec, acbert4rec.py:201-209, at the
 * widths where the catalogue product is a library GEMM): `CrossEntropyLoss(reduction='none')` without the [rows, N]
 * log-probabilities torch writes in the forward and the zero-filled [rows, N] tensor its backward starts from.
 *   fwd: lse[r] = logsumexp_n logits[r, n]; row_loss[r] = lse[r] - logits[r, target[r]]  (NaN for a target outside [0, N))
 *   bwd: d_logits[r, n] = coef[r] * (exp(logits[r, n] - lse[r]) - [n == target[r]])       (one read, one write of [rows, N])
 * logits and d_logits contiguous fp32, target int64 [rows], lse / row_loss / coef fp32 [rows]. */
int acattn_dense_ce_fwd(const float* logits, int64_t rows, int64_t N, const int64_t* target, float* lse, float* row_loss,
                        void* stream);
int acattn_dense_ce_bwd(const float* logits, const float* lse, const int64_t* target, const float* coef, int64_t rows, int64_t N,
                        float* d_logits, void* stream);

/* ABI 27: the start of a (replayed) training step in ONE launch: up to ACATTN_MAX_COPIES device-to-device copies of the
 * batch tensors into the static buffers a captured hipGraph reads (recbole/trainer/trainer.py:661 interaction.to(device)
 * ends in such buffers here), `*counter += 1` when counter != NULL (the replay counter the in-kernel RNG adds to its
 * seeds), and last_row[b] = item_length[b] - 1 when both are given (the position the models read,
 * abstract_recommender.py:130-134; `item_length` is read from its SOURCE, so it may be one of the tensors being copied).
 * Sizes in bytes, any alignment; src[i] == dst[i] is skipped. */
#define ACATTN_MAX_COPIES 6
int acattn_step_inputs(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n_copies, int64_t* counter,
                       const int64_t* item_length, int64_t* last_row, int32_t n_rows, void* stream);

/* The mask penalty || 1 - M ||_2 over a whole attack-mask tensor (torch.norm(1 - attack_mask, p=2):
 * recbole/model/sequential_recommender/acsasrec.py:131-137, acbert4rec.py:229-232) and its gradient
 * d_m = d_norm * (m - 1) / norm.  m has n floats (16-byte aligned); `workspace` holds
 * ACATTN_PENALTY_WS_FLOATS floats; norm, d_norm are device scalars. */
#define ACATTN_PENALTY_WS_FLOATS 1024
int acattn_mask_penalty_fwd(const float* m, int64_t n, float* workspace, float* norm, void* stream);
int acattn_mask_penalty_bwd(const float* m, const float* norm, const float* d_norm, int64_t n, float* d_m, void* stream);

/* The attacked loss of the two-pass protocol assembled in one launch (acsasrec.py:129-137):
 *     loss = -mean_b row_loss_b + weight * mean_l || 1 - M_l ||_2
 * acattn_mask_penalty_partial: partial sums of (1 - m)^2 over one mask into `part` (ACATTN_PENALTY_WS_FLOATS floats).
 * acattn_attacked_loss_finish: row_loss [B] from acattn_full_sort_ce_fwd[_dir]; part [n_masks, ACATTN_PENALTY_WS_FLOATS],
 *   every mask of mask_numel elements; out [2 + n_masks] = (loss, mean row loss, norm_0, ...).  scale_buf (or NULL,
 *   n_scale floats, at most a few 100k: one workgroup walks it) is multiplied by -1/B in place: pass the CE direction
 *   so that d loss / d out = scale_buf * d_loss.
 * acattn_mask_penalty_bwd_scaled: d_m = d_loss * scale * (m - 1) / norm, scale = weight / n_masks. */
int acattn_mask_penalty_partial(const float* m, int64_t n, float* part, void* stream);
int acattn_attacked_loss_finish(const float* row_loss, int32_t B, const float* part, int32_t n_masks, int64_t mask_numel,
                                float weight, float* out, float* scale_buf, int32_t n_scale, void* stream);
int acattn_mask_penalty_bwd_scaled(const float* m, const float* norm, const float* d_loss, float scale, int64_t n, float* d_m,
                                   void* stream);
/* The penalty through the attention node instead of through M [round 3]:
 * acattn_mask_penalty_rows: pen[b, head, qb] = sum over the rows of query block qb (16 rows) and all L keys of (1 - M)^2,
 *   M [B,nh,L,L] -> pen [B, nh, ceil(L/16)].  The sum of pen is || 1 - M ||_2 ^ 2.
 * acattn_attacked_loss_finish_rows: acattn_attacked_loss_finish with one pen vector (count floats) per mask in place of the
 *   partial-sum workspace.
 * acattn_mask_penalty_drows: d_pen[l][:] = d_loss * scale / (2 * norm_l) for every mask l (count floats each), i.e. the
 *   cotangent of pen under loss = ... + scale * sum_l sqrt(sum pen_l): feed it to acattn_bwd_io.d_penalty_part. */
int acattn_mask_penalty_rows(const float* m, int32_t B, int32_t n_heads, int32_t L, float* pen, void* stream);
int acattn_attacked_loss_finish_rows(const float* row_loss, int32_t B, const float* const* pen, int32_t n_masks, int32_t count,
                                     float weight, float* out, float* scale_buf, int32_t n_scale, void* stream);
int acattn_mask_penalty_drows(const float* norms, const float* d_loss, float scale, int32_t count, float* const* d_pen,
                              int32_t n_masks, void* stream);
/* [ABI 29] acattn_mask_penalty_drows + d_out[:] = direction[:] * d_loss[0] (n_dir floats; the attacked loss's output
 * cotangent from the direction acattn_full_sort_ce_fwd_dir saved, acsasrec.py:129-137) in the same launch. */
int acattn_mask_penalty_drows_dir(const float* norms, const float* d_loss, float scale, int32_t count, float* const* d_pen,
                                  int32_t n_masks, const float* direction, float* d_out, int32_t n_dir, void* stream);
/* The same two for all masks of a model (one per layer, each of n elements, at most ACATTN_MAX_MASKS) in ONE launch each:
 * part [n_masks, ACATTN_PENALTY_WS_FLOATS]; norms [n_masks] (the out + 2 of acattn_attacked_loss_finish). */
#define ACATTN_MAX_MASKS 8
int acattn_mask_penalty_partial_multi(const float* const* m, int32_t n_masks, int64_t n, float* part, void* stream);
int acattn_mask_penalty_bwd_scaled_multi(const float* const* m, const float* norms, const float* d_loss, float scale, int64_t n,
                                         float* const* d_m, int32_t n_masks, void* stream);

/* Parameter gradients of y = x W^T + b (torch.nn.functional.linear as called for query/key/value, the attack
 * transforms, dense, the gate and the feed-forward pair: recbole/model/layers.py:687-690, 660-661, 681, 792-794, 863):
 *   dw[N,K] = sum_m dy[m,n] x[m,k]      db[N] = sum_m dy[m,n]  (db may be NULL)
 * x [M,K] and dy [M,N] contiguous fp32, M = B*L rows; one pass over both, fp32 MFMA, deterministic two-stage
 * reduction.  `workspace` is caller-owned scratch of acattn_linear_wgrad_workspace_bytes(M, K, N) bytes. */
int64_t acattn_linear_wgrad_workspace_bytes(int64_t M, int32_t K, int32_t N);
int acattn_linear_wgrad(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, void* workspace, float* dw,
                        float* db, void* stream);

/* The same for up to ACATTN_WGRAD_MAX_GROUP layers in ONE launch pair (the six projections of an encoder layer, or
 * dense + the feed-forward pair of a layer's tail, are differentiated together).  Item i has operands x[i] [M,K[i]],
 * dy[i] [M,N[i]] and results dw[i] [N[i],K[i]], db[i] [N[i]] (db[i] may be NULL); all items share M.  The arrays are
 * host arrays (of device pointers / sizes), read during the call.
 * `workspace`: sum_i acattn_linear_wgrad_workspace_bytes(M, K[i], N[i]) bytes. */
#define ACATTN_WGRAD_MAX_GROUP 8
int acattn_linear_wgrad_grouped(const float* const* x, const float* const* dy, const int32_t* K, const int32_t* N,
                                float* const* dw, float* const* db, int32_t n_items, int64_t M, void* workspace,
                                void* stream);

/* [ABI 29] The two stages of acattn_linear_wgrad_grouped separately.  A training step has one stage-2 launch per layer and
 * walk -- six 6-us launches at the benchmark shape for results that only the optimizer reads.  A trainer can launch stage 1
 * where the gradient is due (`_partial`: the items' partial sums go to `workspace`; `n_partials` and, per item, the offsets
 * in floats of its weight / bias partials inside `workspace` come back; want_bias[i] != 0 = bias partials wanted) and ONE
 * stage 2 for all of them when the backward walk is over (`_reduce_many`: up to ACATTN_WGRAD_MAX_REDUCE items, each with its
 * own partial pointers, sizes and partial count; db[i] may be NULL) -- which also takes up to ACATTN_SUMROWS_MAX_DEFER plain
 * row sums sum_out[i][c] = sum_r sum_x[i][r, c] (sum_x[i] contiguous [sum_R[i], sum_C[i]]: the LayerNorm / calibrator
 * parameter partials of the same walk, acattn_sum_rows with batch 1) as further slices of the same launch; either list may
 * be empty.  The workspaces must stay untouched in between. */
#define ACATTN_WGRAD_MAX_REDUCE 32
#define ACATTN_SUMROWS_MAX_DEFER 8
int acattn_linear_wgrad_grouped_partial(const float* const* x, const float* const* dy, const int32_t* K, const int32_t* N,
                                        const int32_t* want_bias, int32_t n_items, int64_t M, void* workspace, int32_t* n_partials,
                                        int64_t* w_offset, int64_t* b_offset, void* stream);
int acattn_linear_wgrad_reduce_many(const float* const* part_w, const float* const* part_b, const int32_t* K, const int32_t* N,
                                    const int32_t* n_partials, float* const* dw, float* const* db, int32_t n_items,
                                    const float* const* sum_x, float* const* sum_out, const int32_t* sum_R, const int32_t* sum_C,
                                    int32_t n_sums, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The position-wise tail of one branch of an encoder layer in one launch (forward) / one launch (input gradients):
 *     a   = LayerNorm(dropout(dense(ctx)) + x)                  recbole/model/layers.py:681-683 (cal_adjusted_outputs)
 *     out = LayerNorm(dropout(dense_2(gelu(dense_1(a)))) + a)   recbole/model/layers.py:790-798 (FeedForward, erf-GELU)
 * Replaces 3 GEMM + 2 dropout/add/LayerNorm + 1 GELU launches.  (H, I) in {(64, 256), (64, 128)}; weights are the
 * row-major nn.Linear parameters ([out, in]).  Dropout decisions are those of acattn_dropout_add_layernorm_* for
 * the same (seed, row, column). */
typedef struct acattn_tail_problem {
  int32_t rows, H, I;      /* rows = positions (B*L or the selected ones); H hidden_size; I inner_size */
  const float* ctx;        /* [rows,H] merged-head context (attention output before the dense) */
  const float* x;          /* [rows,H] the layer's input (residual of the first LayerNorm) */
  const float* wd;         /* [H,H] attack_attention.dense.weight */
  const float* bd;         /* [H] */
  const float* g1;         /* [H] attack_attention.LayerNorm.weight */
  const float* b1;         /* [H] */
  const float* w1;         /* [I,H] feed_forward.dense_1.weight */
  const float* bb1;        /* [I] */
  const float* w2;         /* [H,I] feed_forward.dense_2.weight */
  const float* bb2;        /* [H] */
  const float* g2;         /* [H] feed_forward.LayerNorm.weight */
  const float* b2;         /* [H] */
  float eps1, eps2;        /* layer_norm_eps of the two norms */
  float p1, p2;            /* hidden_dropout_prob of out_dropout / feed_forward.dropout in training, 0 in eval */
  const uint8_t* keep1;    /* optional explicit keep masks [rows,H]; NULL = counter RNG from (seed, seed_device) */
  const uint8_t* keep2;
  uint64_t seed1, seed2;
  const uint64_t* seed_device;
  /* Optional row selection: tail row r reads ctx / x at row (r / src_R) * src_L + src_index[r] (ctx, x are
   * [*, src_L, H]; src_index [rows] positions, e.g. item_seq_len - 1 of abstract_recommender.py:130-134 with
   * src_R = 1), and the backward ADDS the d_ctx / d_x rows there (a position may be picked twice): the caller
   * zero-fills those two.  NULL = identity. */
  const int64_t* src_index;
  int32_t src_R, src_L;
} acattn_tail_problem;

/* Tensors the forward writes and the backward reads (all required in both directions, `act` forward only). */
typedef struct acattn_tail_saved {
  float* h1;   /* [rows,H] dense(ctx) + bias */
  float* st1;  /* [rows,2] (mean, 1/std) of the first norm */
  float* a;    /* [rows,H] output of the first norm */
  float* act;  /* [rows,I] gelu(dense_1(a)): operand of dense_2's weight gradient */
  float* h3;   /* [rows,H] dense_2(act) + bias */
  float* st2;  /* [rows,2] */
  float* out;  /* [rows,H] the branch's output */
  /* ABI 26, optional (NULL = off), hidden 128 only: [rows,I] gelu'(dense_1(a)).  When the forward is given it, the
   * backward given the same pointer reads it instead of rebuilding dense_1(a) (a fourth of its matrix work at that
   * width).  Ignored at hidden 64 (the rebuild is cheaper than the traffic there). */
  float* gelu_grad;
} acattn_tail_saved;

typedef struct acattn_tail_bwd_io {
  const float* d_out;  /* [rows,H] */
  float* d_ctx;        /* [rows,H] or NULL */
  float* d_x;          /* [rows,H] or NULL */
  float* d_h1;         /* [rows,H], d_h2 [rows,I], d_h3 [rows,H]: cotangents of the three dense outputs, i.e. the */
  float* d_h2;         /*   operands of acattn_linear_wgrad_grouped with (ctx, a, act); each may be NULL */
  float* d_h3;
  float* dgb_part;     /* [acattn_layer_tail_bwd_partial_rows(rows), 4, H] partial sums of (dgamma1, dbeta1, dgamma2,
                          dbeta2), to be summed over the leading dimension by the caller; or NULL.  Hidden 128 / 256:
                          acattn_layer_tail_bwd_partial_rows_for(rows, H) rows */
  void* workspace;     /* ABI 26: device scratch of acattn_layer_tail_bwd_workspace_bytes(H, I) bytes: transposed weight
                          copies live in it during the launch (every width but hidden 64 at <= 4096 rows needs it) */
} acattn_tail_bwd_io;

int acattn_layer_tail_supported(int32_t H, int32_t I);
int acattn_layer_tail_fwd(const acattn_tail_problem* p, const acattn_tail_saved* saved, void* stream);
int acattn_layer_tail_bwd(const acattn_tail_problem* p, const acattn_tail_saved* saved, const acattn_tail_bwd_io* io,
                          void* stream);
int32_t acattn_layer_tail_bwd_partial_rows(int32_t rows);               /* hidden 64 */
int32_t acattn_layer_tail_bwd_partial_rows_for(int32_t rows, int32_t H); /* any supported hidden size */
int64_t acattn_layer_tail_bwd_workspace_bytes(int32_t H, int32_t I);
/* Measurement hook: rows per wave of the forward = 16 * nb (0 = chosen by size).  Returns the previous setting. */
int acattn_select_layer_tail_blocks(int nb);

/* ------------------------------------------------------------------------------------------------------------------
 * The six projections in front of the attention core in one launch (forward) / one launch (input gradients):
 *     mq, mk, mv = query(x), key(x), value(x)                               recbole/model/layers.py:687-689
 *     qa, ka     = attack_query_transform(mq), attack_key_transform(mk)     recbole/model/layers.py:658-659
 *     gate       = gate(mq)  [rows, G], G = seq_length                      recbole/model/layers.py:887
 * hidden_size 64, 128 or 256, G <= 256; weights are the row-major nn.Linear parameters ([out, in]). */
typedef struct acattn_proj_problem {
  int32_t rows, H, G;        /* G = 0 and wg = bg = NULL without the gate (combine_option != 'gate') */
  const float* x;            /* [rows,H] the layer's input */
  const float *wq, *bq;      /* attack_attention.query */
  const float *wk, *bk;      /* attack_attention.key */
  const float *wv, *bv;      /* attack_attention.value */
  const float *waq, *baq;    /* attack_attention.attack_query_transform */
  const float *wak, *bak;    /* attack_attention.attack_key_transform */
  const float *wg, *bg;      /* gate [G,H], [G]; or NULL */
  /* ABI 26, only read when acattn_proj_out.affine is set: the spatial calibrator's parameters (layers.py:636-640), the
   * head count and the sequence length (rows = B * L, row r sits at position r % L of sequence r / L) */
  const float *w_order, *b_order, *w_dist, *b_dist; /* [2*dh], [1], [2*dh], [1] */
  int32_t n_heads, L;
} acattn_proj_problem;

typedef struct acattn_proj_out {
  float *mq, *mk, *mv, *qa, *ka; /* [rows,H] each */
  float* gate;                   /* [rows,G] or NULL */
  /* ABI 26: what the attention core would otherwise re-derive per head / per query block (acattn_problem.affine,
   * .gate_is_prob) */
  float* affine;      /* NULL, or [B,nh,4,LP], LP = 16*ceil(L/16): entries [0, L) of every plane are written, the
                         padding entries are left alone (allocate the buffer zeroed once; they must be finite) */
  int32_t gate_prob;  /* non-zero: `gate` receives sigmoid(logits) instead of the logits */
} acattn_proj_out;

/* Cotangents in (each may be NULL = zero), gradients out (each may be NULL = not wanted):
 *   dmq_total = dmq + dqa . Waq + dgate . Wg     (cotangent operand of query's weight gradient; may alias dmq)
 *   dmk_total = dmk + dka . Wak                  (the same for key; may alias dmk)
 *   dx        = dmq_total . Wq + dmk_total . Wk + dmv . Wv */
typedef struct acattn_proj_bwd_io {
  const float *dmq, *dmk, *dmv, *dqa, *dka; /* [rows,H] */
  const float* dgate;                       /* [rows,G] */
  float *dmq_total, *dmk_total, *dx;        /* [rows,H] */
  /* [rows,H] or NULL: dx starts from it.  x also feeds the layer tails as the residual (layers.py:683): their d_x is
   * handed in here instead of being added to dx by a separate elementwise launch. */
  const float* dx_init;
  /* ABI 26: device scratch of acattn_projections_bwd_workspace_bytes(p) bytes (0 at hidden 64: may be NULL there); hidden
   * 128 / 256 keep transposed copies of the layer's weights in it for the duration of the launch */
  void* workspace;
} acattn_proj_bwd_io;

int acattn_projections_supported(int32_t H, int32_t G);
int64_t acattn_projections_bwd_workspace_bytes(const acattn_proj_problem* p);
int acattn_projections_fwd(const acattn_proj_problem* p, const acattn_proj_out* out, void* stream);
int acattn_projections_bwd(const acattn_proj_problem* p, const acattn_proj_bwd_io* io, void* stream);

/*
 * torch.optim.Adam's update (the reference's optimizer: recbole/trainer/trainer.py:590-615, `learner: adam`) for up to
 * ACATTN_ADAM_MAX_TENSORS parameters in one launch: amsgrad = False, maximize = False, weight decay as in Adam (added to
 * the gradient).  Per tensor: fp32 param / grad / exp_avg / exp_avg_sq of `numel` elements and the step counter as a
 * device float (torch's `capturable` state layout).  The counters hold the number of updates done BEFORE the call and
 * are incremented by the launch.  `done`: a device int32 that is zero before the first call (the launch leaves it zero).
 * The arithmetic reproduces ATen's fused kernel operation by operation (csrc/acattn_adam.hip).
 */
#define ACATTN_ADAM_MAX_TENSORS 64
typedef struct acattn_adam_group {
  int32_t n_tensors;
  float* param[ACATTN_ADAM_MAX_TENSORS];
  const float* grad[ACATTN_ADAM_MAX_TENSORS];
  float* exp_avg[ACATTN_ADAM_MAX_TENSORS];
  float* exp_avg_sq[ACATTN_ADAM_MAX_TENSORS];
  float* step[ACATTN_ADAM_MAX_TENSORS];
  int64_t numel[ACATTN_ADAM_MAX_TENSORS];
} acattn_adam_group;
int acattn_adam_step(const acattn_adam_group* g, double lr, double beta1, double beta2, double eps, double weight_decay,
                     int32_t* done, void* stream);

/* ABI 27: 1 when acattn_calibrated_attention_bwd(p, io) with io->dgate_summed = 1 will write the head-summed [B,L,L] gate
 * gradient (see acattn_bwd_io.dgate_summed), 0 when the launch needs the per-head [B,nh,L,L] buffer.  Validates nothing
 * else; every other field of `io` as for the launch itself (pointers are only tested for NULL). */
int acattn_calibrated_attention_bwd_gate_summed(const acattn_problem* p, const acattn_bwd_io* io);
/* ABI 27: 1 when the launch will evaluate the second cotangent set of `io` (see acattn_bwd_io.dqa2), else 0. */
int acattn_calibrated_attention_bwd_pair_supported(const acattn_problem* p, const acattn_bwd_io* io);

/* ABI version of the loaded library (== ACATTN_ABI_VERSION of the header it was built from). */
int acattn_abi_version(void);

/* Thread-local description of the last negative return value. */
const char* acattn_last_error(void);

/* Bytes of algorithmic HBM traffic of one forward call (DESIGN.md "Contract A"/"A'"): what bench.py prices. */
int64_t acattn_fwd_algorithmic_bytes(const acattn_problem* p);

/*
 * Fused calibrated attention, forward.  Replaces, for all heads of all sequences in one launch:
 *   raw scores + spatial calibrator + two softmaxes   recbole/model/layers.py:695-740
 *   attack-mask scores + softmax                       recbole/model/layers.py:661-672
 *   noise, perturbed / calibrated / combined probs     recbole/model/layers.py:917-936
 *   the two P.V products and head merge                recbole/model/layers.py:677-680
 */
int acattn_calibrated_attention_fwd(const acattn_problem* p, const acattn_fwd_out* out, void* stream);

/* The affine planes of acattn_problem.affine ([B,nh,4,16*ceil(L/16)], padding written as zeros) from p->q, p->k and the
 * spatial calibrator's parameters (recbole/model/layers.py:705-708 in rank-1 form), for a producer of q / k other than
 * acattn_projections_fwd.  Reads only B, L, H, n_heads, q, k, w_order, b_order, w_dist, b_dist of *p. */
int acattn_spatial_affines(const acattn_problem* p, float* affine, void* stream);

/* Backward of the above (what autograd derives for the same op chain in the reference). May be
 * called any number of times for one forward (the trainer walks the graph twice:
 * recbole/trainer/trainer.py:677,684). */
int acattn_calibrated_attention_bwd(const acattn_problem* p, const acattn_bwd_io* io, void* stream);

/* Bytes of `acattn_bwd_io.workspace` for this problem (5 row scalars of the chained soft-max backward per query row). */
int64_t acattn_calibrated_attention_bwd_workspace_bytes(const acattn_problem* p);

/* Materialise the COUNTER-mode randomness for (seed, shape) so a run can be replayed in EXPLICIT mode. */
int acattn_rng_materialize(int32_t B, int32_t n_heads, int32_t L, uint64_t seed, float p_drop, float* noise,
                           uint8_t* keep_after, uint8_t* keep_mask, uint8_t* keep_before, void* stream);

/* Diagnostics / measurement: pin the forward kernel acattn_calibrated_attention_fwd dispatches to (process-wide).
 * Every kernel computes the same function; one that does not cover the problem (options, L, rng mode) is skipped
 * and the automatic choice applies.  Returns the previous setting.  The setting is PER HOST THREAD (thread_local): it
 * affects the launches the calling thread issues afterwards and nobody else's. */
enum {
  ACATTN_FWD_AUTO = 0,    /* streaming kernel where it applies (training configuration, L <= 208), else general */
  ACATTN_FWD_STREAM = 1,  /* acattn_fwd_stream.hip: one wave per 16-row query block, no LDS staging (L <= 208) */
  ACATTN_FWD_STAGED = 2,  /* acattn_fwd_dma.hip / acattn_fwd_fast.hip: K, Ka, V, gate logits staged in LDS (L <= 64) */
  ACATTN_FWD_GENERAL = 3  /* acattn_fwd.hip: every option, explicit randomness, probability dumps */
};
int acattn_select_forward_kernel(int which);

/* The same for acattn_calibrated_attention_bwd. */
enum {
  ACATTN_BWD_AUTO = 0,    /* training configuration: row-resident for L <= 64, streaming (io.workspace given) beyond */
  ACATTN_BWD_STREAM = 1,  /* acattn_bwd_stream.hip: row kernel + key kernel, tiles rebuilt from the saved normalisers */
  ACATTN_BWD_ROW = 2      /* acattn_bwd_fast.hip / acattn_bwd.hip: a query block's whole row in registers */
};
int acattn_select_backward_kernel(int which);

#ifdef __cplusplus
}
#endif
#endif /* ACATTN_H_ */
