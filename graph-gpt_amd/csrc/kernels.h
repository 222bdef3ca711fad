// Host launchers of the HBM-bound kernels (kernels.hip) and the attention core (attention.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gget.h"
#include "common.h"

// Small fp32 accumulators (norm weights, LayerScale, ...) exist in kAccumCopies copies, `stride` elements apart: a block
// adds into copy blockIdx % copies (an atomic add of 512 blocks into ONE 768-float vector serialises in L2: that was
// the whole cost of the RMSNorm backward), and the bf16 conversion sums the copies.
constexpr int kAccumCopies = 8;   // (round 3: 32 / 128 / 512 copies leave the C1 step unchanged or slower - 7.30 / 7.31 / 7.36 / 7.57 ms)
constexpr uint64_t kAccumCopyMax = 8192;   // parameters up to this many elements are replicated
struct GgetSegment {
  uint64_t src;    // element offset into the fp32 scratch (copy 0)
  uint64_t dst;    // element offset into the bf16 gradient array
  uint64_t count;  // elements (multiple of 4)
  uint64_t copies; // number of fp32 copies to sum
  uint64_t stride; // elements between copies
};

int k_embed_fwd(const int64_t* ids, const void* emb, const void* gate, void* out, int T, int F, int ldF, int d,
                hipStream_t st, ElemDropArg E = ElemDropArg{0, 1.f, 0});
int k_elem_dropout(void* x, long T, int n, unsigned stream, ElemDropArg E, hipStream_t st);
#define GGET_DROP_STREAM_RAW 60u      /* raw_embed_dropout (48-50: embed / MLP, 51 + layer: the MLP score head) */
// raw-embedding inputs (config.embed_dim > 0; modeling_pretrain.py:131-149): out[t,:] = bf16(raw[t,:]), or the mask token on rows whose
// labels are all set (first_only: whose first label is set); flag[t] = 1 on those rows.  labels == nullptr: no row is replaced.
int k_raw_blend(const float* raw, const int64_t* labels, int n, bool first_only, const void* tok, void* out, int32_t* flag, int T, int e,
                hipStream_t st, const int32_t* rows_map = nullptr, int n_logical = 0);
// d emb_mask_token += sum over the flagged rows of dx[t,:]  (fp32 accumulator, copy 0)
int k_raw_tok_grad(const void* dx, const int32_t* flag, float* dtok, int T, int e, hipStream_t st);
// stack_method = "long": x[t,:] *= min(1, bf16(1 / (non-zero ids of token t + 1e-7))) in place (forward value and its gradient);
// w[b] = 1 / (labelled cells of sample b + 1e-7) - modeling_helpers.py:106-110, :327-342
int k_embed_long_ratio(const int64_t* ids, void* x, int T, int F, int ldF, int d, hipStream_t st);
int k_sample_mask_wgt(const int64_t* labels, float* w, int B, int cells, hipStream_t st);
// sort_ws: int32 scratch of k_embed_bwd_ws_elems(T*F, V) elements (device-side counting sort of the cells by id)
int k_embed_bwd(const int64_t* ids, const void* dx, const void* emb, const void* gate, float* demb, float* dgate, int T,
                int F, int ldF, int d, int V, int pad_id, int32_t* sort_ws, hipStream_t st, ElemDropArg E = ElemDropArg{0, 1.f, 0});
inline size_t k_embed_bwd_ws_elems(size_t ncell, size_t V) { return 3 * V + 1 + 2 * ncell; }
// count matrix of the dense embedding backward: cnt[t][v] = #{f: ids[t][f] == v, v != pad} as bf16, row pitch ldc; the
// caller clears it first.  Used when k_embed_dense_ok(): dE = cnt^T dX is then one split-K GEMM (engine.hip: embed_bwd)
int k_embed_count(const int64_t* ids, void* cnt, int T, int F, int ldF, int ldc, int pad_id, hipStream_t st);
inline bool k_embed_dense_ok(int V, bool gated) { return !gated && ((V + 63) / 64) * 64 <= 1024; }
constexpr int kEmbDenseSplit = 8;   // K slices (fp32 slabs of V*d each) of that GEMM
int k_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int T, int d, float eps, hipStream_t st);
// dw_accum: fp32 [copies][copy_stride] accumulators (see GgetSegment); copies = 1 for a plain vector
void k_set_deterministic(int on);   // reproducible summation order of the RMSNorm weight gradients (gget_debug_set key 4)
void k_set_rms_wide(int on);   // 1 (default): short RMSNorm backward launches run one 16-wave block per CU (kernels.hip rmsnorm_bwd_kernel)
void k_set_ce_parts(int on);    // 1 (default): ce_rows_kernel leaves one partial loss sum per block where the caller has the slots (no same-address atomics)
int k_get_deterministic();
int k_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                  float* dw_accum, int T, int d, hipStream_t st, int copies = 1, uint64_t copy_stride = 0);
int k_rope(void* qkv, const float* cos_tab, const float* sin_tab, const int64_t* position_ids, int T, int S, int H,
           int inverse, hipStream_t st);
int k_rope_table(float* cos_tab, float* sin_tab, int max_pos, float theta, hipStream_t st);
// rope_range > 0: per-token angle tables [B*S][32] from positions rescaled to [0, range) per row, and the identity ids [B*S]
int k_rope_range_table(const int64_t* pos, float* cos_t, float* sin_t, int64_t* ids, int B, int S, float range, float theta,
                       hipStream_t st);
int k_geglu_fwd(const void* gu, void* h, int T, int ff, hipStream_t st);
int k_geglu_bwd(const void* gu, const void* dh, void* dgu, int T, int ff, hipStream_t st);
int k_lengths(const int64_t* mask, const int64_t* ids, int ldF, int pad_id, int32_t* key_len, int32_t* pool_row, int B,
              int S, hipStream_t st);
// Var-len (padding-free) token layout of a right-padded batch: cu[b] = first compact row of sample b (exclusive scan of key_len, cu[B] =
// total), compact ids [t_rows][F] / row positions / sample index per row, pad2c[b*S+s] = compact row or -1; rows [tc, t_rows) are pad
// tokens.  pool_row (task head, may be NULL) is moved to the compact rows; status[0] = 1 if sum(key_len) != tc.
int k_varlen_plan(const int64_t* ids, int ldF, int F, const int64_t* pos, int32_t* key_len, int32_t* pool_row, int32_t* cu,
                  int64_t* ids_c, int64_t* pos_c, int32_t* row_b, int32_t* pad2c, int32_t* c2p, int32_t* status, int B, int S, int tc,
                  int t_rows, int pad_id, hipStream_t st, int32_t* long_list = nullptr);      // c2p[r] = logical row b * S + s of compact row r
// out = clamp(pos, 0, max_pos - 1); *flag = 1 (sticky) if anything was clamped
int k_clamp_positions(const int64_t* pos, int64_t* out, int32_t* flag, long n, int max_pos, hipStream_t st);
int k_head_slot_sort(const int32_t* sel_src, const int32_t* row_idx, const int32_t* lm_count, int32_t* slot_state, int32_t* a_tok,
                     int32_t* a_cell, int32_t* c_l, int32_t* cellpos, int32_t* tile_off, int32_t* total_p, int cap, int n, long slot_elems,
                     int tile_rows, hipStream_t st);
int k_head_cell_sum(const void* dxs, const int32_t* cellpos, const int32_t* cnt, const int32_t* l_off, const int32_t* pad2c, void* dhid,
                    int TP, int d, int pad_row, hipStream_t st);
int k_poison_loss(const int32_t* flag, float* loss, hipStream_t st);
int k_sum_lengths(const int32_t* key_len, int B, int32_t* out, hipStream_t st, int32_t* host_out = nullptr);   // host_out: pinned host word the kernel stores to
int k_head_compact(const int64_t* labels, int T, int n, int32_t* cnt, int32_t* m_off, int32_t* l_off, int32_t* counts,
                   int32_t* row_idx, int32_t* sel_src, int32_t* sel_label, int32_t* sel_tok, int32_t* blk_tot, int32_t* slot_hist,
                   hipStream_t st);     // slot_hist: nullptr, or the slot-sorted head's state (k_head_slot_sort follows in the same forward)
size_t k_head_compact_ws_bytes(int T);
int k_gather_rows(const void* src, const int32_t* idx, const int32_t* count, void* dst, int cap, int d, int scatter,
                  hipStream_t st);
int k_gather_rows_remap(const void* src, int32_t* idx, const int32_t* count, const int32_t* pad2c, void* dst, int cap, int d, int pad_row,
                        int32_t* status, hipStream_t st);     // var-len remap of idx (in place) + gather, one launch
int k_ce_fwd_bwd(const void* logits, int ld, const int32_t* labels, const int32_t* sel_tok, const float* sample_wgt, int S,
                 const int32_t* n_rows_dev, int n_rows_cap, int V, float* loss_sum, void* dlogits, float scale_base,
                 int mean_over_rows, float* loss_out, hipStream_t st, float focal_gamma = 0.f, float* loss_part = nullptr,
                 int loss_part_cap = 0);   // loss_part: optional [loss_part_cap >= 2048] floats - one partial loss sum per block instead of atomics
int k_score_fwd(const void* hidden, const int32_t* pool_row, const void* w, const void* bias, float* logits,
                void* pooled_h, int B, int C, int d, hipStream_t st);
// token-level head (loss_type = "token_ce"): score on every row (fp32 logits rounded as bf16), cross-entropy with ignore_index -100
// (stat: 4 floats of scratch - loss sum, labelled rows, 1 / rows), and the backward (dW / dbias accumulate, dhidden is overwritten)
int k_tok_score_fwd(const void* hidden, const void* w, const void* bias, float* logits, int T, int C, int d, hipStream_t st);
int k_tok_ce(const float* logits, const int64_t* labels, float* dl, float* stat, float* loss_out, int T, int C, hipStream_t st,
             const int32_t* rows_map = nullptr, int n_logical = 0);   // rows_map: compact row -> logical row of the labels (var-len layout)
int k_copy_from_host(const void* src_host_mapped, void* dst, size_t bytes, hipStream_t st);
int k_rows_to_grid(const void* src, const int32_t* pad2c, void* out, long n_pos, int d, hipStream_t st);
int k_scatter_rows_map_f32(const float* src, const int32_t* rows_map, float* dst, int T, int C, int n_logical, hipStream_t st);
int k_tok_score_bwd(const float* dl, const float* stat, const void* hidden, const void* w, float* dw, float* dbias, void* dhidden, int T,
                    int C, int d, hipStream_t st);
int k_pool_rows(const void* hidden, const int32_t* pool_row, void* out, int B, int d, hipStream_t st);
int k_head_linear_fwd(const void* x, void* a, const void* w, const void* bias, void* y, float* y32, int B, int Din, int Dout,
                      int layer, ElemDropArg E, hipStream_t st);
int k_head_linear_bwd(const float* dy, const void* x, const void* a, const void* w, float* dw, float* dbias, float* dx, int B,
                      int Din, int Dout, int layer, ElemDropArg E, hipStream_t st);
int k_scatter_rows_f32(const float* src, const int32_t* pool_row, void* dhidden, int B, int d, hipStream_t st);
int k_auc_loss(const float* logits, const int64_t* labels, int B, int C, int num_neg, unsigned seed, float* loss_out,
               float* dlogits, int32_t* lists, hipStream_t st);
int k_task_loss(const float* logits, const void* labels, const float* sample_wgt, int problem, int B, int C,
                float* loss_out, float* dlogits, hipStream_t st);
int k_score_bwd(const float* dlogits, const void* hidden, const int32_t* pool_row, const void* w, float* dw, float* dbias,
                void* dhidden, int B, int C, int d, hipStream_t st);
// ws: k_grad_sqnorm_ws_bytes() of scratch; ws[0] receives sum(g^2) (deterministic reduction order, two launches)
int k_grad_sqnorm(const void* g, size_t n, float* ws, hipStream_t st);
inline size_t k_grad_sqnorm_ws_bytes() { return (16 + 1024) * sizeof(float); }
// the same over a list of chunks of the gradient array (offset / count pairs, counts multiples of 8, at most 1024 chunks) plus `nextra`
// ready-made partial sums (the weight-gradient kernels' per-tile sums): ws[0] = sum of both, fixed order
struct GgetSqChunk { uint64_t off; uint64_t cnt; };
int k_grad_sqnorm_chunks(const void* g, const GgetSqChunk* chunks_dev, int nchunks, const float* extra, int nextra, float* ws, hipStream_t st);
int k_adamw(float* master, float* m, float* v, const void* grad, void* param, size_t n, float lr, float beta1, float beta2,
            float eps, float wd, int step, float max_norm, float grad_scale, const float* sqnorm, float* gnorm_out,
            hipStream_t st, bool skip_nonfinite = false);   // skip_nonfinite: leave everything untouched when the gradient norm is inf / NaN
constexpr int kZeroRanges = 6;
struct GgetZeroRanges {
  void* ptr[kZeroRanges];
  size_t bytes[kZeroRanges];
  int n;
  void add(void* p, size_t b) { if (b && n < kZeroRanges) { ptr[n] = p; bytes[n] = b; ++n; } }
};
int k_zero_ranges(const GgetZeroRanges& R, hipStream_t st);   // every range cleared by one launch (kernels.hip)
int k_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t st);
int k_slab_reduce(const float* slabs, long slab_stride, int nslab, void* dst, size_t n, hipStream_t st, bool f32_out = false);  // dst: bf16, or fp32 (overwritten)
int k_convert_segments(const float* scratch, void* grads, const GgetSegment* segs_dev, int nseg, hipStream_t st);

// attention.hip
// cos_tab/sin_tab ([max_pos][32] fp32) non-null => RoPE is applied to q,k on load (unless qk_rotated: the projection GEMM
// already rotated them) and undone on dq,dk, so dqkv is always the gradient of the UN-rotated projections
int k_attn_fwd(const void* qkv, const int32_t* key_len, void* out, float* lse, int B, int S, int H, int causal,
               const float* cos_tab, const float* sin_tab, const int64_t* position_ids, float dropout_p,
               unsigned dropout_seed, hipStream_t st, const int32_t* key_lo = nullptr, const int32_t* key_hi = nullptr,
               const int32_t* row_base = nullptr, const int32_t* long_list = nullptr);
// S <= 32: attention (all heads of a sample), the o projection + residual add and the RMSNorm behind it in one launch, one workgroup
// per sample (attention.hip: attn_oproj_fwd_kernel).  qkv rotated [rows, 3d]; writes attn_out [rows, d] (kept for the backward), lse,
// x_mid = x_in + attn_out Wo^T, xn = rmsnorm(x_mid) * nw, rstd [rows].  *taken = 1 when it ran (S <= 32, H in {2, 4, 8, 12, 16},
// GGET_ATTN_OPROJ != 0), else the caller runs k_attn_fwd + the GEMM + k_rmsnorm_fwd.
// ... `wo` is the FRAGMENT-MAJOR copy of the o weight written by k_pack_wo (fwd); the backward counterpart below reads the fragment-major
// copy of the transposed weight (bwd).  k_pack_wo packs `layers` [d][d] weights that sit `layer_stride` elements apart.
int k_pack_wo(const void* w0, size_t layer_stride, void* fwd, void* bwd, int d, int layers, hipStream_t st);
// RMSNorm backward of post_attention_layernorm (dxn, x_mid, nw, rstd, dres -> dx_mid, dw_accum as k_rmsnorm_bwd), the o projection's
// dgrad and the attention backward (-> dqkv, dq / dk rotated back with cos_tab / sin_tab / position_ids) of one decoder layer, one
// workgroup per sample; *taken as above (also 0 in the reproducible mode: the norm weight gradient is summed with atomics).
int k_attn_oproj_bwd(const void* dxn, const void* x_mid, const void* nw, const float* rstd, const void* dres, void* dx_mid, float* dw_accum,
                     int copies, uint64_t copy_stride, const void* wot_packed, const void* qkv, const float* lse, const int32_t* key_len,
                     const int32_t* row_base, void* dqkv, int B, int S, int H, int causal, const float* cos_tab, const float* sin_tab,
                     const int64_t* position_ids, float dropout_p, unsigned dropout_seed, int t_rows, hipStream_t st, int* taken,
                     void* dattn_long = nullptr, const int32_t* long_list = nullptr);   // t_rows: rows of the token-major buffers (var-len: the pad rows behind the last sample get a zero dx_mid)
int k_attn_oproj_fwd(const void* qkv, const int32_t* key_len, const int32_t* row_base, void* attn_out, float* lse, const void* wo,
                     const void* x_in, void* x_mid, const void* nw, void* xn, float* rstd, int B, int S, int H, int causal, float eps,
                     float dropout_p, unsigned dropout_seed, hipStream_t st, int* taken);
int k_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len, void* dqkv,
               float* delta_ws, int B, int S, int H, int causal, const float* cos_tab, const float* sin_tab,
               const int64_t* position_ids, int qk_rotated, float dropout_p, unsigned dropout_seed, hipStream_t st,
               const int32_t* key_lo = nullptr, const int32_t* key_hi = nullptr, const int32_t* row_base = nullptr,
               void* dq_ws = nullptr, size_t dq_slab_stride = 0, const int32_t* long_list = nullptr);   // dq_ws: bf16 [ceil(S / 256)][dq_slab_stride >= rows * H * 64] - S >= 256 (GGET_ATTN_FUSED_MIN_S) then runs the fused one-pass backward (attention.hip)
// row_base ([B] int32, needs key_len): var-len (padding-free) token layout - sample b owns rows [row_base[b], row_base[b] + key_len[b]) of
// qkv / out / dout / dqkv instead of [b * S, b * S + S); lse / delta / position ids stay [B,S]-indexed.
// packed rows: inclusive key range [lo, hi] of every token from the block-diagonal mask [B,S,S] (first / last 1 of its row)
int k_ranges_from_mask3d(const int64_t* mask3d, int32_t* key_lo, int32_t* key_hi, int B, int S, hipStream_t st);
int k_smtp2d(const int64_t* ids_in, int ld_in, const int64_t* node_idx, int ld_node, int64_t* ids_out, int64_t* labels_out,
             int B, int S, int F, float rate, float power, float replace_rate, int vocab, int global_mask, unsigned seed,
             hipStream_t st);
int k_token_sample(const void* logits, int ld, int R, int V, int mode, float temperature, float top_p, int top_k, float alg_temp,
                   unsigned seed, float* conf, int64_t* tok, hipStream_t st);
int k_unmask_origin(int64_t* x, const int64_t* cand, int B, int N, float p_transfer, unsigned seed, int mask_id, hipStream_t st);
int k_token_confidence(const void* logits, int ld, int R, int V, int mode, float* conf, int64_t* tok, hipStream_t st);
int k_smtp_rows(const int64_t* ids_in, const int32_t* lengths, int64_t* ids_out, int64_t* labels_out, float* wgt_out, int B, int S,
                int F, double umr_min, double umr_max, double power, unsigned seed, hipStream_t st);
