// C-ABI entry points of libacattn.so (include/acattn.h): argument validation, dispatch, error text.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "acattn_common.h"

namespace {
thread_local char g_err[256] = "";

int fail(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

int check_problem(const acattn_problem* p) {
  if (!p) return fail("problem is NULL");
  if (p->B < 1 || p->L < 1 || p->H < 1 || p->n_heads < 1) return fail("B, L, H, n_heads must be positive");
  if (p->H % p->n_heads != 0)
    return fail("The hidden size is not a multiple of the number of attention heads");  // layers.py:618-622
  const int dh = p->H / p->n_heads;
  if (dh != 16 && dh != 32 && dh != 64 && dh != 128) return fail("unsupported head size: dh must be 16, 32, 64 or 128");
  if (p->L > 208) return fail("unsupported sequence length: L must be <= 208");
  if (!p->q || !p->k || !p->v) return fail("q, k, v must be non-NULL");
  if (p->mask_mode == ACATTN_MASK_STRUCTURED) {
    if (!p->key_valid) return fail("structured mask needs key_valid");
  } else if (p->mask_mode == ACATTN_MASK_DENSE_LL || p->mask_mode == ACATTN_MASK_DENSE_L) {
    if (!p->mask) return fail("dense mask mode needs mask");
  } else {
    return fail("unknown mask_mode");
  }
  if (p->w_order && !p->b_order) return fail("w_order given without b_order");
  if (p->w_dist && (!p->b_dist || !p->scalar)) return fail("w_dist given without b_dist / scalar");
  if (p->p_drop < 0.f || p->p_drop >= 1.f) return fail("p_drop must lie in [0, 1)");
  if (p->rng_mode != ACATTN_RNG_EXPLICIT && p->rng_mode != ACATTN_RNG_COUNTER) return fail("unknown rng_mode");
  if (p->rng_mode == ACATTN_RNG_EXPLICIT && p->p_drop > 0.f) {
    if (!p->keep_after) return fail("explicit dropout needs keep_after");
    if (p->adversarial && !p->keep_mask) return fail("explicit dropout needs keep_mask");
    if (p->adversarial && !p->two_level && !p->keep_before) return fail("explicit dropout needs keep_before");
  }
  if (p->adversarial) {
    if (!p->qa || !p->ka) return fail("adversarial calibrator needs qa and ka");
    if (p->rng_mode == ACATTN_RNG_EXPLICIT && !p->noise) return fail("explicit rng mode needs noise");
    if (p->combine_option == ACATTN_COMBINE_GATE) {
      if (!p->gate_logits) return fail("combine_option gate needs gate_logits");
    } else if (p->combine_option != ACATTN_COMBINE_FIXED && p->combine_option != ACATTN_COMBINE_ANNEALING) {
      return fail("unknown combine_option");  // layers.py:894-895 raises KeyError
    }
    if (!p->two_level) {
      if (p->rich_combine != ACATTN_RICH_FIXED && p->rich_combine != ACATTN_RICH_TRAINABLE)
        return fail("unknown rich_calibrated_combine");  // layers.py:935-936 raises KeyError
      if (p->rich_combine == ACATTN_RICH_TRAINABLE && !p->rich_ratio) return fail("trainable rich combine needs rich_ratio");
    }
  }
  return 0;
}
}  // namespace

void acattn_set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

extern "C" {

int acattn_abi_version(void) { return ACATTN_ABI_VERSION; }

int acattn_select_forward_kernel(int which) { return acattn_fwd_kernel_choice(which); }

int acattn_select_backward_kernel(int which) { return acattn_bwd_kernel_choice(which); }

const char* acattn_last_error(void) { return g_err; }

int64_t acattn_fwd_algorithmic_bytes(const acattn_problem* p) {
  if (!p) return -1;
  const int64_t t_lh = 4LL * p->L * p->H, t_ll = 4LL * p->L * p->L;
  if (!p->adversarial) return (int64_t)p->B * 4 * t_lh;  // contract A': read q,k,v, write one context
  // contract A: read q,k,v,qa,ka (+ gate logits), write two contexts + M (noise generated in-kernel)
  int64_t per_seq = 7 * t_lh + (int64_t)p->n_heads * t_ll;
  if (p->combine_option == ACATTN_COMBINE_GATE) per_seq += t_ll;
  return (int64_t)p->B * per_seq;
}

int acattn_calibrated_attention_fwd(const acattn_problem* p, const acattn_fwd_out* out, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!out) return fail("out is NULL");
  if (!out->ctx_calibrated) return fail("ctx_calibrated must be non-NULL");
  if (p->adversarial && (!out->ctx_attacked || !out->attack_mask))
    return fail("adversarial forward needs ctx_attacked and attack_mask outputs");
  acattn_penalty_written_set(false);
  int rc = acattn_launch_fwd(*p, *out, (hipStream_t)stream);
  // acattn_fwd_out.penalty_part: the long-sequence streaming kernel has formed the sums itself; behind every other kernel
  // they are taken from the mask it wrote
  if (rc == 0 && p->adversarial && out->penalty_part && !acattn_penalty_written())
    rc = acattn_launch_penalty_rows(out->attack_mask, p->B, p->n_heads, p->L, out->penalty_part, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_spatial_affines(const acattn_problem* p, float* affine, void* stream) {
  if (!p || !affine) return fail("problem and affine must be non-NULL");
  if (p->B < 1 || p->L < 1 || p->H < 1 || p->n_heads < 1 || p->H % p->n_heads != 0 || (p->H / p->n_heads) % 4 != 0)
    return fail("spatial affines: bad shape");
  if (!p->q || !p->k || !p->w_order || !p->b_order || !p->w_dist || !p->b_dist)
    return fail("spatial affines need q, k and both affines' parameters");
  const int rc = acattn_launch_spatial_affines(*p, affine, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_calibrated_attention_bwd(const acattn_problem* p, const acattn_bwd_io* io, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!io) return fail("io is NULL");
  if (!p->adversarial) return fail("backward is defined for the adversarial (full) operator only");
  if (!io->attack_mask || !io->row_stats) return fail("backward needs the forward's attack_mask and row_stats");
  if (!io->dq || !io->dk || !io->dv || !io->dqa || !io->dka) return fail("dq, dk, dv, dqa, dka must be non-NULL");
  if (p->combine_option == ACATTN_COMBINE_GATE && !io->dgate_logits) return fail("gate combine needs dgate_logits");
  if (!io->dw_order_part || !io->dw_dist_part || !io->dsmall_part) return fail("parameter partial buffers must be non-NULL");
  if (io->part_stride != 0 && io->part_stride < 2 * (p->H / p->n_heads)) return fail("part_stride is smaller than a partial row");
  const int rc = acattn_launch_bwd(*p, *io, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_calibrated_attention_bwd_gate_summed(const acattn_problem* p, const acattn_bwd_io* io) {
  if (!p || !io || p->B < 1 || p->L < 1 || p->H < 1 || p->n_heads < 1) return 0;
  return acattn_bwd_gate_summed(*p, *io) ? 1 : 0;
}

int acattn_calibrated_attention_bwd_pair_supported(const acattn_problem* p, const acattn_bwd_io* io) {
  if (!p || !io || p->B < 1 || p->L < 1 || p->H < 1 || p->n_heads < 1) return 0;
  return acattn_bwd_pair_supported(*p, *io) ? 1 : 0;
}

int64_t acattn_calibrated_attention_bwd_workspace_bytes(const acattn_problem* p) {
  if (!p || p->B < 1 || p->L < 1 || p->n_heads < 1) return -1;
  return acattn_bwd_stream_ws_bytes(*p);
}

int acattn_rng_materialize(int32_t B, int32_t n_heads, int32_t L, uint64_t seed, float p_drop, float* noise,
                           uint8_t* keep_after, uint8_t* keep_mask, uint8_t* keep_before, void* stream) {
  if (B < 1 || n_heads < 1 || L < 1) return fail("B, n_heads, L must be positive");
  const int rc = acattn_launch_rng(B, n_heads, L, seed, p_drop, noise, keep_after, keep_mask, keep_before,
                                   (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

static int check_ce(const acattn_ce_problem* p) {
  if (!p) return fail("ce problem is NULL");
  if (p->B < 1 || p->N < 1) return fail("B and N must be positive");
  if (p->H != 64 && p->H != 128 && p->H != 256) return fail("unsupported hidden size for the fused cross-entropy: H must be 64, 128 or 256");
  if (!p->out || !p->table || !p->target) return fail("out, table, target must be non-NULL");
  return 0;
}

int64_t acattn_full_sort_ce_workspace_bytes(const acattn_ce_problem* p) {
  if (!p || (p->H != 64 && p->H != 128 && p->H != 256)) return -1;
  return acattn_ce_ws_bytes(*p);
}

int acattn_full_sort_ce_fwd(const acattn_ce_problem* p, void* workspace, float* lse, float* row_loss, void* stream) {
  if (int rc = check_ce(p)) return rc;
  if (!workspace || !lse || !row_loss) return fail("workspace, lse, row_loss must be non-NULL");
  const int rc = acattn_launch_ce_fwd(*p, workspace, lse, row_loss, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_full_sort_ce_fwd_dir(const acattn_ce_problem* p, void* workspace, float* lse, float* row_loss, float* dir,
                                void* stream) {
  if (int rc = check_ce(p)) return rc;
  if (!workspace || !lse || !row_loss || !dir) return fail("workspace, lse, row_loss, dir must be non-NULL");
  const int rc = acattn_launch_ce_fwd_dir(*p, workspace, lse, row_loss, dir, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_full_sort_ce_bwd(const acattn_ce_problem* p, const float* lse, const float* coef, void* workspace,
                            float* d_out, float* d_table, void* stream) {
  if (int rc = check_ce(p)) return rc;
  if (!workspace || !lse || !coef || !d_out) return fail("workspace, lse, coef, d_out must be non-NULL");
  const int rc = acattn_launch_ce_bwd(*p, lse, coef, workspace, d_out, d_table, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_full_sort_ce_products(int mode) { return acattn_ce_products_choice(mode); }

static int check_ln(const acattn_ln_problem* p) {
  if (!p) return fail("ln problem is NULL");
  if (p->rows < 1) return fail("rows must be positive");
  if (p->H != 64 && p->H != 128 && p->H != 256) return fail("unsupported hidden size for the fused LayerNorm: H must be 64, 128 or 256");
  if (p->residual_rows < 1 || p->rows % p->residual_rows != 0) return fail("residual_rows must divide rows");
  if (!p->z || !p->residual || !p->gamma || !p->beta) return fail("z, residual, gamma, beta must be non-NULL");
  if (p->p_drop < 0.f || p->p_drop >= 1.f) return fail("p_drop must lie in [0, 1)");
  return 0;
}

int acattn_dropout_add_layernorm_fwd(const acattn_ln_problem* p, float* y, float* stats, void* stream) {
  if (int rc = check_ln(p)) return rc;
  if (!y || !stats) return fail("y and stats must be non-NULL");
  const int rc = acattn_launch_ln_fwd(*p, y, stats, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_dropout_add_layernorm_bwd(const acattn_ln_problem* p, const float* dy, const float* stats, float* dz,
                                     float* dres, float* dgb_part, void* stream) {
  if (int rc = check_ln(p)) return rc;
  if (!dy || !stats) return fail("dy and stats must be non-NULL");
  const int rc = acattn_launch_ln_bwd(*p, dy, stats, dz, dres, dgb_part, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

static int check_proj(const acattn_proj_problem* p) {
  if (!p) return fail("problem must be non-NULL");
  if (p->rows < 1) return fail("rows must be positive");
  if (!acattn_proj_supported(p->H, p->G)) return fail("projections: hidden_size must be 64, 128 or 256 and the gate at most 256 wide");
  if (!p->x || !p->wq || !p->bq || !p->wk || !p->bk || !p->wv || !p->bv || !p->waq || !p->baq || !p->wak || !p->bak)
    return fail("projections: input and parameters must be non-NULL");
  if ((p->wg != nullptr) != (p->bg != nullptr) || (p->wg != nullptr) != (p->G > 0))
    return fail("projections: gate weight, bias and width go together");
  return 0;
}

int acattn_projections_supported(int32_t H, int32_t G) { return acattn_proj_supported(H, G) ? 1 : 0; }

int64_t acattn_projections_bwd_workspace_bytes(const acattn_proj_problem* p) {
  if (!p || !acattn_proj_supported(p->H, p->G)) return -1;
  return acattn_proj_bwd_ws_bytes(*p);
}

int acattn_projections_fwd(const acattn_proj_problem* p, const acattn_proj_out* out, void* stream) {
  if (int rc = check_proj(p)) return rc;
  if (!out || !out->mq || !out->mk || !out->mv || !out->qa || !out->ka) return fail("projections: outputs must be non-NULL");
  if ((p->wg != nullptr) != (out->gate != nullptr)) return fail("projections: gate output goes with the gate parameters");
  if (out->affine) {
    if (!p->w_order || !p->b_order || !p->w_dist || !p->b_dist) return fail("projections: affine planes need the spatial calibrator's parameters");
    if (p->n_heads < 1 || p->H % p->n_heads != 0 || (p->H / p->n_heads) % 16 != 0) return fail("projections: affine planes need a head size that is a multiple of 16");
    if (p->L < 1 || p->rows % p->L != 0) return fail("projections: affine planes need rows = B * L");
  }
  const int rc = acattn_launch_proj_fwd(*p, *out, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_projections_bwd(const acattn_proj_problem* p, const acattn_proj_bwd_io* io, void* stream) {
  if (int rc = check_proj(p)) return rc;
  if (!io) return fail("io must be non-NULL");
  if (!io->dmq_total && !io->dmk_total && !io->dx) return fail("projections backward: nothing to compute");
  const int rc = acattn_launch_proj_bwd(*p, *io, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

static int check_tail(const acattn_tail_problem* p, const acattn_tail_saved* s) {
  if (!p || !s) return fail("problem and saved must be non-NULL");
  if (p->rows < 1) return fail("rows must be positive");
  if (!acattn_tail_supported(p->H, p->I)) return fail("layer tail: (hidden_size, inner_size) must be (64, 256), (64, 128), (128, 512), (128, 256) or (256, 1024)");
  if (!p->ctx || !p->x || !p->wd || !p->bd || !p->g1 || !p->b1 || !p->w1 || !p->bb1 || !p->w2 || !p->bb2 || !p->g2 || !p->b2)
    return fail("layer tail: inputs and parameters must be non-NULL");
  if (!(p->p1 >= 0.f && p->p1 < 1.f) || !(p->p2 >= 0.f && p->p2 < 1.f)) return fail("dropout probabilities must be in [0, 1)");
  if (!s->h1 || !s->st1 || !s->a || !s->h3 || !s->st2) return fail("layer tail: saved tensors must be non-NULL");
  if (p->src_index && (p->src_R < 1 || p->src_L < 1)) return fail("layer tail: src_index needs src_R, src_L >= 1");
  return 0;
}

int acattn_layer_tail_supported(int32_t H, int32_t I) { return acattn_tail_supported(H, I) ? 1 : 0; }

int32_t acattn_layer_tail_bwd_partial_rows(int32_t rows) { return acattn_tail_bwd_partial_rows(rows); }
int32_t acattn_layer_tail_bwd_partial_rows_for(int32_t rows, int32_t H) { return acattn_tail_bwd_partial_rows_h(rows, H); }
int64_t acattn_layer_tail_bwd_workspace_bytes(int32_t H, int32_t I) {
  return acattn_tail_supported(H, I) ? acattn_tail_bwd_ws_bytes(H, I) : -1;
}

int acattn_select_layer_tail_blocks(int nb) {
  if (nb < 0 || nb > 2) return fail("rows per wave: 0 (automatic), 1 or 2 blocks of 16");
  return acattn_select_tail_nb(nb);
}

int acattn_layer_tail_fwd(const acattn_tail_problem* p, const acattn_tail_saved* saved, void* stream) {
  if (int rc = check_tail(p, saved)) return rc;
  if (!saved->act || !saved->out) return fail("layer tail forward: act and out must be non-NULL");
  const int rc = acattn_launch_tail_fwd(*p, *saved, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_layer_tail_bwd(const acattn_tail_problem* p, const acattn_tail_saved* saved, const acattn_tail_bwd_io* io,
                          void* stream) {
  if (int rc = check_tail(p, saved)) return rc;
  if (!io || !io->d_out) return fail("layer tail backward: d_out must be non-NULL");
  const int rc = acattn_launch_tail_bwd(*p, *saved, *io, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_dense_ce_fwd(const float* logits, int64_t rows, int64_t N, const int64_t* target, float* lse, float* row_loss,
                        void* stream) {
  if (!logits || !target || !lse || !row_loss) return fail("dense CE: logits, target, lse, row_loss must be non-NULL");
  if (rows < 1 || N < 1) return fail("dense CE: rows and N must be positive");
  const int rc = acattn_launch_dense_ce_fwd(logits, rows, N, target, lse, row_loss, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}
int acattn_dense_ce_bwd(const float* logits, const float* lse, const int64_t* target, const float* coef, int64_t rows, int64_t N,
                        float* d_logits, void* stream) {
  if (!logits || !lse || !target || !coef || !d_logits) return fail("dense CE backward: every pointer must be non-NULL");
  if (rows < 1 || N < 1) return fail("dense CE: rows and N must be positive");
  const int rc = acattn_launch_dense_ce_bwd(logits, lse, target, coef, rows, N, d_logits, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_step_inputs(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n_copies, int64_t* counter,
                       const int64_t* item_length, int64_t* last_row, int32_t n_rows, void* stream) {
  if (n_copies < 0 || n_copies > ACATTN_MAX_COPIES) return fail("step inputs: 0 .. ACATTN_MAX_COPIES copies");
  if (n_copies > 0 && (!src || !dst || !bytes)) return fail("step inputs: src, dst, bytes must be non-NULL");
  for (int k = 0; k < n_copies; ++k)
    if (bytes[k] > 0 && (!src[k] || !dst[k])) return fail("step inputs: NULL copy operand");
  if ((item_length == nullptr) != (last_row == nullptr)) return fail("step inputs: item_length and last_row come together");
  if (last_row && n_rows < 1) return fail("step inputs: n_rows must be positive");
  const int rc = acattn_launch_step_inputs(src, dst, bytes, n_copies, counter, item_length, last_row, n_rows, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_sum_rows(const float* x, float* out, int32_t batch, int32_t R, int32_t C, void* stream) {
  if (!x || !out) return fail("x and out must be non-NULL");
  if (batch < 1 || R < 1 || C < 1) return fail("batch, R, C must be positive");
  const int rc = acattn_launch_sum_rows(x, out, batch, R, C, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_sum_rows_pair(const float* x1, float* out1, int32_t batch1, int32_t R1, int32_t C1, const float* x2, float* out2,
                         int32_t batch2, int32_t R2, int32_t C2, void* stream) {
  if (!x1 || !out1 || !x2 || !out2) return fail("x and out must be non-NULL");
  if (batch1 < 1 || R1 < 1 || C1 < 1 || batch2 < 1 || R2 < 1 || C2 < 1) return fail("batch, R, C must be positive");
  const int rc = acattn_launch_sum_rows_pair(x1, out1, batch1, R1, C1, x2, out2, batch2, R2, C2, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

static int check_embed(const acattn_embed_problem* p) {
  if (!p) return fail("embed problem is NULL");
  if (p->rows < 1 || p->L < 1 || p->rows % p->L != 0) return fail("rows must be a positive multiple of L");
  if (p->H != 64 && p->H != 128 && p->H != 256) return fail("unsupported hidden size for the fused embedding front end: H must be 64, 128 or 256");
  if (p->n_table_rows < 1) return fail("n_table_rows must be positive");
  if (!p->idx || !p->table || !p->gamma || !p->beta) return fail("idx, table, gamma, beta must be non-NULL");
  if (p->p_drop < 0.f || p->p_drop >= 1.f) return fail("p_drop must lie in [0, 1)");
  return 0;
}

int acattn_embed_layernorm_fwd(const acattn_embed_problem* p, float* y, float* stats, void* stream) {
  if (const int rc = check_embed(p)) return rc;
  if (!y || !stats) return fail("y and stats must be non-NULL");
  const int rc = acattn_launch_embed_fwd(*p, y, stats, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_embed_layernorm_bwd(const acattn_embed_problem* p, const float* dy, const float* stats, int64_t padding_idx,
                               float* d_table, float* d_pos_part, float* dgb_part, void* stream) {
  if (const int rc = check_embed(p)) return rc;
  if (!dy || !stats) return fail("dy and stats must be non-NULL");
  const int rc = acattn_launch_embed_bwd(*p, dy, stats, padding_idx, d_table, d_pos_part, dgb_part, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_fwd(const float* m, int64_t n, float* workspace, float* norm, void* stream) {
  if (!m || !workspace || !norm) return fail("m, workspace and norm must be non-NULL");
  if (n < 1) return fail("n must be positive");
  if (((uintptr_t)m & 15) != 0) return fail("m must be 16-byte aligned");
  const int rc = acattn_launch_penalty_fwd(m, n, workspace, norm, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_bwd(const float* m, const float* norm, const float* d_norm, int64_t n, float* d_m, void* stream) {
  if (!m || !norm || !d_norm || !d_m) return fail("m, norm, d_norm and d_m must be non-NULL");
  if (n < 1) return fail("n must be positive");
  if ((((uintptr_t)m | (uintptr_t)d_m) & 15) != 0) return fail("m and d_m must be 16-byte aligned");
  const int rc = acattn_launch_penalty_bwd(m, norm, d_norm, n, d_m, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_partial(const float* m, int64_t n, float* part, void* stream) {
  if (!m || !part) return fail("m and part must be non-NULL");
  if (n < 1) return fail("n must be positive");
  if (((uintptr_t)m & 15) != 0) return fail("m must be 16-byte aligned");
  const int rc = acattn_launch_penalty_partial(m, n, part, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_attacked_loss_finish(const float* row_loss, int32_t B, const float* part, int32_t n_masks, int64_t mask_numel,
                                float weight, float* out, float* scale_buf, int32_t n_scale, void* stream) {
  if (!row_loss || !part || !out) return fail("row_loss, part and out must be non-NULL");
  if (B < 1 || n_masks < 1 || mask_numel < 1) return fail("B, n_masks and mask_numel must be positive");
  if (n_scale < 0 || (n_scale > 0 && !scale_buf)) return fail("scale_buf must be given with n_scale > 0");
  const int rc = acattn_launch_attacked_loss_finish(row_loss, B, part, n_masks, mask_numel, weight, out, scale_buf, n_scale,
                                                    (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_rows(const float* m, int32_t B, int32_t n_heads, int32_t L, float* pen, void* stream) {
  if (!m || !pen) return fail("m and pen must be non-NULL");
  if (B < 1 || n_heads < 1 || L < 1) return fail("B, n_heads and L must be positive");
  const int rc = acattn_launch_penalty_rows(m, B, n_heads, L, pen, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_attacked_loss_finish_rows(const float* row_loss, int32_t B, const float* const* pen, int32_t n_masks, int32_t count,
                                     float weight, float* out, float* scale_buf, int32_t n_scale, void* stream) {
  if (!row_loss || !pen || !out) return fail("row_loss, pen and out must be non-NULL");
  if (B < 1 || count < 1 || n_masks < 1 || n_masks > ACATTN_MAX_MASKS) return fail("B, count positive, 1 <= n_masks <= ACATTN_MAX_MASKS");
  for (int l = 0; l < n_masks; ++l)
    if (!pen[l]) return fail("every pen vector must be non-NULL");
  if (n_scale < 0 || (n_scale > 0 && !scale_buf)) return fail("scale_buf must be given with n_scale > 0");
  const int rc = acattn_launch_attacked_loss_finish_rows(row_loss, B, pen, n_masks, count, weight, out, scale_buf, n_scale,
                                                         (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_drows(const float* norms, const float* d_loss, float scale, int32_t count, float* const* d_pen,
                              int32_t n_masks, void* stream) {
  if (!norms || !d_loss || !d_pen) return fail("norms, d_loss and d_pen must be non-NULL");
  if (count < 1 || n_masks < 1 || n_masks > ACATTN_MAX_MASKS) return fail("count positive, 1 <= n_masks <= ACATTN_MAX_MASKS");
  for (int l = 0; l < n_masks; ++l)
    if (!d_pen[l]) return fail("every d_pen vector must be non-NULL");
  const int rc = acattn_launch_penalty_drows(norms, d_loss, scale, count, d_pen, n_masks, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_drows_dir(const float* norms, const float* d_loss, float scale, int32_t count, float* const* d_pen,
                                  int32_t n_masks, const float* direction, float* d_out, int32_t n_dir, void* stream) {
  if (!norms || !d_loss || !d_pen || !direction || !d_out) return fail("norms, d_loss, d_pen, direction and d_out must be non-NULL");
  if (count < 1 || n_dir < 1 || n_masks < 1 || n_masks > ACATTN_MAX_MASKS) return fail("count, n_dir positive, 1 <= n_masks <= ACATTN_MAX_MASKS");
  for (int l = 0; l < n_masks; ++l)
    if (!d_pen[l]) return fail("every d_pen vector must be non-NULL");
  const int rc = acattn_launch_penalty_drows(norms, d_loss, scale, count, d_pen, n_masks, (hipStream_t)stream, direction, d_out, n_dir);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_bwd_scaled(const float* m, const float* norm, const float* d_loss, float scale, int64_t n, float* d_m,
                                   void* stream) {
  if (!m || !norm || !d_loss || !d_m) return fail("m, norm, d_loss and d_m must be non-NULL");
  if (n < 1) return fail("n must be positive");
  if ((((uintptr_t)m | (uintptr_t)d_m) & 15) != 0) return fail("m and d_m must be 16-byte aligned");
  const int rc = acattn_launch_penalty_bwd_scaled(m, norm, d_loss, scale, n, d_m, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_partial_multi(const float* const* m, int32_t n_masks, int64_t n, float* part, void* stream) {
  if (!m || !part) return fail("m and part must be non-NULL");
  if (n < 1 || n_masks < 1 || n_masks > ACATTN_MAX_MASKS) return fail("n must be positive, 1 <= n_masks <= ACATTN_MAX_MASKS");
  for (int l = 0; l < n_masks; ++l)
    if (!m[l] || ((uintptr_t)m[l] & 15) != 0) return fail("every mask must be non-NULL and 16-byte aligned");
  const int rc = acattn_launch_penalty_partial_multi(m, n_masks, n, part, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_mask_penalty_bwd_scaled_multi(const float* const* m, const float* norms, const float* d_loss, float scale, int64_t n,
                                         float* const* d_m, int32_t n_masks, void* stream) {
  if (!m || !norms || !d_loss || !d_m) return fail("m, norms, d_loss and d_m must be non-NULL");
  if (n < 1 || n_masks < 1 || n_masks > ACATTN_MAX_MASKS) return fail("n must be positive, 1 <= n_masks <= ACATTN_MAX_MASKS");
  for (int l = 0; l < n_masks; ++l)
    if (!m[l] || !d_m[l] || (((uintptr_t)m[l] | (uintptr_t)d_m[l]) & 15) != 0)
      return fail("every m and d_m must be non-NULL and 16-byte aligned");
  const int rc = acattn_launch_penalty_bwd_scaled_multi(m, norms, d_loss, scale, n, d_m, n_masks, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_adam_step(const acattn_adam_group* g, double lr, double beta1, double beta2, double eps, double weight_decay,
                     int32_t* done, void* stream) {
  if (!g || !done) return fail("group and done must be non-NULL");
  if (g->n_tensors < 1 || g->n_tensors > ACATTN_ADAM_MAX_TENSORS) return fail("1 <= n_tensors <= ACATTN_ADAM_MAX_TENSORS");
  int64_t blocks = 0;
  for (int t = 0; t < g->n_tensors; ++t) {
    if (!g->param[t] || !g->grad[t] || !g->exp_avg[t] || !g->exp_avg_sq[t] || !g->step[t])
      return fail("param, grad, exp_avg, exp_avg_sq and step of every tensor must be non-NULL");
    if (g->numel[t] < 1) return fail("numel must be positive");
    blocks += (g->numel[t] + 4095) / 4096;
  }
  if (blocks >= (1ll << 31)) return fail("too many elements for one launch");
  if (!(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1)) return fail("betas must lie in [0, 1)");
  const int rc = acattn_launch_adam_step(*g, lr, beta1, beta2, eps, weight_decay, done, (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int64_t acattn_linear_wgrad_workspace_bytes(int64_t M, int32_t K, int32_t N) {
  if (M < 1 || K < 1 || N < 1) return fail("M, K, N must be positive");
  return acattn_linear_wgrad_ws_bytes(M, K, N);
}

int acattn_linear_wgrad_grouped(const float* const* x, const float* const* dy, const int32_t* K, const int32_t* N,
                                float* const* dw, float* const* db, int32_t n_items, int64_t M, void* workspace,
                                void* stream) {
  if (!x || !dy || !K || !N || !dw || !db || !workspace) return fail("x, dy, K, N, dw, db and workspace must be non-NULL");
  if (n_items < 1 || n_items > ACATTN_WGRAD_MAX_GROUP) return fail("n_items must lie in [1, ACATTN_WGRAD_MAX_GROUP]");
  if (M < 1) return fail("M must be positive");
  for (int i = 0; i < n_items; ++i) {
    if (!x[i] || !dy[i] || !dw[i]) return fail("every item needs x, dy and dw");
    if (K[i] < 1 || N[i] < 1) return fail("K and N must be positive");
    if (M * (int64_t)std::max(K[i], N[i]) >= (1LL << 40)) return fail("matrix too large");
  }
  const int rc = acattn_launch_linear_wgrad(x, dy, (const int*)K, (const int*)N, dw, db, n_items, M, workspace,
                                            (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_linear_wgrad_grouped_partial(const float* const* x, const float* const* dy, const int32_t* K, const int32_t* N,
                                        const int32_t* want_bias, int32_t n_items, int64_t M, void* workspace, int32_t* n_partials,
                                        int64_t* w_offset, int64_t* b_offset, void* stream) {
  if (!x || !dy || !K || !N || !want_bias || !workspace || !n_partials || !w_offset || !b_offset)
    return fail("x, dy, K, N, want_bias, workspace, n_partials, w_offset and b_offset must be non-NULL");
  if (n_items < 1 || n_items > ACATTN_WGRAD_MAX_GROUP) return fail("n_items must lie in [1, ACATTN_WGRAD_MAX_GROUP]");
  if (M < 1) return fail("M must be positive");
  float* dw[ACATTN_WGRAD_MAX_GROUP];
  float* db[ACATTN_WGRAD_MAX_GROUP];
  for (int i = 0; i < n_items; ++i) {
    if (!x[i] || !dy[i]) return fail("every item needs x and dy");
    if (K[i] < 1 || N[i] < 1) return fail("K and N must be positive");
    if (M * (int64_t)std::max(K[i], N[i]) >= (1LL << 40)) return fail("matrix too large");
    dw[i] = nullptr;                                   // (stage 1 writes partials only; db != NULL = "bias partials wanted")
    db[i] = want_bias[i] ? (float*)workspace : nullptr;
  }
  int P = 0;
  long long wo[ACATTN_WGRAD_MAX_GROUP], bo[ACATTN_WGRAD_MAX_GROUP];
  const int rc = acattn_launch_linear_wgrad(x, dy, (const int*)K, (const int*)N, dw, db, n_items, M, workspace,
                                            (hipStream_t)stream, &P, wo, bo);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  *n_partials = P;
  for (int i = 0; i < n_items; ++i) {
    w_offset[i] = wo[i];
    b_offset[i] = bo[i];
  }
  return rc;
}

int acattn_linear_wgrad_reduce_many(const float* const* part_w, const float* const* part_b, const int32_t* K, const int32_t* N,
                                    const int32_t* n_partials, float* const* dw, float* const* db, int32_t n_items,
                                    const float* const* sum_x, float* const* sum_out, const int32_t* sum_R, const int32_t* sum_C,
                                    int32_t n_sums, void* stream) {
  if (n_items < 0 || n_items > ACATTN_WGRAD_MAX_REDUCE) return fail("n_items must lie in [0, ACATTN_WGRAD_MAX_REDUCE]");
  if (n_sums < 0 || n_sums > ACATTN_SUMROWS_MAX_DEFER) return fail("n_sums must lie in [0, ACATTN_SUMROWS_MAX_DEFER]");
  if (n_items + n_sums < 1) return fail("nothing to reduce");
  if (n_items > 0 && (!part_w || !part_b || !K || !N || !n_partials || !dw || !db)) return fail("every weight-gradient array must be non-NULL");
  if (n_sums > 0 && (!sum_x || !sum_out || !sum_R || !sum_C)) return fail("every row-sum array must be non-NULL");
  for (int i = 0; i < n_sums; ++i)
    if (!sum_x[i] || !sum_out[i] || sum_R[i] < 1 || sum_C[i] < 1) return fail("every row sum needs x, out and positive sizes");
  for (int i = 0; i < n_items; ++i) {
    if (!part_w[i] || !dw[i]) return fail("every item needs its partials and dw");
    if (db[i] && !part_b[i]) return fail("db given without bias partials");
    if (K[i] < 1 || N[i] < 1 || n_partials[i] < 1) return fail("K, N and n_partials must be positive");
  }
  const int rc = acattn_launch_linear_wgrad_reduce_many(part_w, part_b, (const int*)K, (const int*)N, (const int*)n_partials, dw, db,
                                                        n_items, sum_x, sum_out, (const int*)sum_R, (const int*)sum_C, n_sums,
                                                        (hipStream_t)stream);
  if (rc > 0) snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
  return rc;
}

int acattn_linear_wgrad(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, void* workspace, float* dw,
                        float* db, void* stream) {
  return acattn_linear_wgrad_grouped(&x, &dy, &K, &N, &dw, &db, 1, M, workspace, stream);
}

}  // extern "C"
