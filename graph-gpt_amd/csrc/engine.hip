// libgget_hip.so - engine object behind the C ABI of include/gget.h.
// Owns no memory: the caller (PyTorch in our binding) hands in the parameter / optimizer / gradient /
// workspace arenas; the engine lays tensors out inside them and enqueues HIP kernels on the caller's
// stream.  Forward/backward structure follows the reference call stack:
//   GraphGPTPretrainBase.forward  modeling_pretrain.py:152-266   -> gget_forward_pretrain
//   GraphGPTTaskModel.forward     modeling_finetune.py:236-326   -> gget_forward_task
//   hf LlamaModel / LlamaDecoderLayer.forward :367-418 / :295-325 -> layer_forward / layer_backward
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <functional>
#include <vector>

#include "common.h"
#include "gemm.h"
#include "kernels.h"
#include <rccl/rccl.h>   // types and enums only: the library is bound with dlopen (see the collective section)

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void gget_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* gget_last_error(void) { return g_err; }
extern "C" int gget_version(void) { return 101; }

struct PathDropArg { float rate; unsigned seed; int S; const int32_t* row_b; };   // row_b: sample index of every row (var-len token layout) or NULL (row / S)

extern int g_gemm_lds_headroom;   // gemm.hip: >= 2 while a collective's kernel shares the chip (data-parallel runs)
namespace {

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
constexpr int kWgSplit = 3;    // K slices of the attention-projection wgrad (72 output tiles -> 216 blocks)
constexpr int kSqTilesPerLayer = 256;   // tiles of a layer's grouped weight-gradient launch that may leave norm partials (one per CU)
constexpr int kLmSplit = 16;   // K slices of the lm_head wgrad at most (few output tiles, K = Lm ~ 4e4); see fit_split
// K slices of a split-K launch whose output has only `tiles` tiles: as many as asked for, but tiles x slices must fit in ONE round of
// the CUs (one 96-144 KiB block per CU) - 18 tiles x 16 slices = 288 blocks ran a second round for 32 of them (lm_head weight
// gradient 88.6 -> measured below; profiles/r03_step_experiments.txt)
int fit_split(int tiles, int want) {
  const int num_cu = gget_gemm_num_cu();   // (minus the CUs a data-parallel run leaves to its collectives)
  int s = want;
  while (s > 1 && tiles * s > num_cu) --s;
  return s;
}

struct ParamRec {
  std::string name;
  int ndim;
  int64_t shape[2];
  uint64_t off;     // elements into flat arrays
  uint64_t count;   // elements (unpadded)
  int layer;        // -1 embeddings, 0..L-1, L = final norm + heads
  bool accum32;     // gradient accumulated in fp32 scratch (atomics), converted to bf16 afterwards
  uint64_t off32;   // element offset in the fp32 scratch
};

struct LayerOff {
  uint64_t ln1, wqkv, wo, lam1, ln2, wgu, wdown, lam2;
  uint64_t ln1_32, lam1_32, ln2_32, lam2_32;
};

struct Plan {
  std::vector<ParamRec> params;
  uint64_t n_params = 0;    // padded flat length
  uint64_t lm_pad_off = 0, lm_pad_count = 0;   // zero rows behind lm_head.weight (see make_plan)
  uint64_t n_scratch32 = 0; // fp32 accumulation elements
  std::vector<LayerOff> layers;
  uint64_t emb = 0, emb32 = 0, gate = 0, gate32 = 0, normf = 0, normf32 = 0;
  uint64_t raw_ln = 0, raw_ln32 = 0, raw_tok = 0, raw_tok32 = 0, raw_proj = 0;   // raw-embedding inputs (embed_dim > 0)
  uint64_t ntp = 0, lm = 0, lm32 = 0, score = 0, score32 = 0, sbias = 0, sbias32 = 0;
  bool has_gate = false, has_ntp = false, has_ls = false;
  // MLP score head (config.head_mlp_layers > 0): n_lin = layers + 1 Linears of widths head_dim[0] = d, ..., head_dim[n_lin] = num_labels
  int n_lin = 0;
  int head_dim[6] = {0, 0, 0, 0, 0, 0};
  uint64_t head_w[5] = {0}, head_w32[5] = {0}, head_b[5] = {0}, head_b32[5] = {0};
  bool has_res = false;  // residual adds run as their own kernels (LayerScale and/or DropPath) instead of GEMM epilogues
};

void add_param(Plan& pl, const std::string& name, int64_t r, int64_t c, int layer, bool accum32, uint64_t* off_out,
               uint64_t* off32_out) {
  ParamRec p;
  p.name = name;
  p.ndim = c > 0 ? 2 : 1;
  p.shape[0] = r;
  p.shape[1] = c > 0 ? c : 0;
  p.count = (uint64_t)r * (uint64_t)(c > 0 ? c : 1);
  p.off = pl.n_params;
  p.layer = layer;
  p.accum32 = accum32;
  p.off32 = 0;
  pl.n_params += align_up(p.count, 128);
  if (accum32) {
    p.off32 = pl.n_scratch32;
    pl.n_scratch32 += align_up(p.count, 128) * (p.count <= kAccumCopyMax ? kAccumCopies : 1);
  }
  if (off_out) *off_out = p.off;
  if (off32_out) *off32_out = p.off32;
  pl.params.push_back(p);
}

// Parameter table = reference state-dict keys ("Checkpoint / resume" row).
Plan make_plan(const gget_config_t& c) {
  Plan pl;
  const int64_t d = c.hidden_size, ff = c.intermediate_size, V = c.vocab_size, F = c.stacked_feat, L = c.num_layers;
  pl.has_gate = c.gated_agg != 0;
  pl.has_ls = c.layer_scale_init > 0.f;
  pl.has_res = pl.has_ls || c.path_pdrop > 0.f || c.mlp_pdrop > 0.f;   // (mlp_dropout sits between down_proj and the residual add)
  pl.has_ntp = c.kind == GGET_KIND_PRETRAIN && c.next_n_token > 1;
  add_param(pl, "model.embed_tokens.weight", V, d, -1, true, &pl.emb, &pl.emb32);
  if (pl.has_gate) add_param(pl, "stacked_feat_agg.weight", F, d, -1, true, &pl.gate, &pl.gate32);
  if (c.embed_dim > 0) {   // modeling_pretrain.py:69-84 / modeling_finetune.py:76-85
    add_param(pl, "embed_layernorm.weight", c.embed_dim, 0, -1, true, &pl.raw_ln, &pl.raw_ln32);
    if (c.kind == GGET_KIND_PRETRAIN) add_param(pl, "emb_mask_token", c.embed_dim, 0, -1, true, &pl.raw_tok, &pl.raw_tok32);
    add_param(pl, "embed_proj.weight", d, c.embed_dim, -1, false, &pl.raw_proj, nullptr);
  }
  pl.layers.resize(L);
  for (int i = 0; i < L; ++i) {
    LayerOff& lo = pl.layers[i];
    const std::string p = "model.layers." + std::to_string(i) + ".";
    add_param(pl, p + "input_layernorm.weight", d, 0, i, true, &lo.ln1, &lo.ln1_32);
    add_param(pl, p + "self_attn.q_proj.weight", d, d, i, false, &lo.wqkv, nullptr);
    add_param(pl, p + "self_attn.k_proj.weight", d, d, i, false, nullptr, nullptr);
    add_param(pl, p + "self_attn.v_proj.weight", d, d, i, false, nullptr, nullptr);
    add_param(pl, p + "self_attn.o_proj.weight", d, d, i, false, &lo.wo, nullptr);
    if (pl.has_ls) add_param(pl, p + "lambda_1", d, 0, i, true, &lo.lam1, &lo.lam1_32);
    add_param(pl, p + "post_attention_layernorm.weight", d, 0, i, true, &lo.ln2, &lo.ln2_32);
    add_param(pl, p + "mlp.gate_proj.weight", ff, d, i, false, &lo.wgu, nullptr);
    add_param(pl, p + "mlp.up_proj.weight", ff, d, i, false, nullptr, nullptr);
    add_param(pl, p + "mlp.down_proj.weight", d, ff, i, false, &lo.wdown, nullptr);
    if (pl.has_ls) add_param(pl, p + "lambda_2", d, 0, i, true, &lo.lam2, &lo.lam2_32);
  }
  add_param(pl, "model.norm.weight", d, 0, (int)L, true, &pl.normf, &pl.normf32);
  if (c.kind == GGET_KIND_PRETRAIN) {
    if (pl.has_ntp) add_param(pl, "n_token_proj.weight", (int64_t)c.next_n_token * d, d, (int)L, false, &pl.ntp, nullptr);
    add_param(pl, "lm_head.weight", V, d, (int)L, false, &pl.lm, nullptr);
    // zero rows behind lm_head up to round_up(V, 64): its dgrad GEMM then runs over K = Vp without a K tail (the pad
    // columns of dlogits are zero as well); not a parameter - no name, no bucket, gradient and AdamW state stay zero
    const int64_t Vp = (int64_t)align_up((uint64_t)V, 64);
    pl.lm_pad_off = pl.n_params;
    pl.lm_pad_count = (uint64_t)(Vp - V) * d;
    pl.n_params += align_up(pl.lm_pad_count, 128);
  } else {
    if (c.head_mlp_layers > 0) {   // `MLP` head (src/utils/modules_utils.py:8-34): state-dict keys score.mlp_modules.<i>.weight / .bias
      pl.n_lin = c.head_mlp_layers + 1;
      pl.head_dim[0] = (int)d;
      for (int i = 0; i < c.head_mlp_layers; ++i) pl.head_dim[i + 1] = c.head_mlp[i];
      pl.head_dim[pl.n_lin] = c.num_labels;
      for (int i = 0; i < pl.n_lin; ++i) {
        const std::string p = "score.mlp_modules." + std::to_string(i) + ".";
        add_param(pl, p + "weight", pl.head_dim[i + 1], pl.head_dim[i], (int)L, true, &pl.head_w[i], &pl.head_w32[i]);
        if (c.score_bias) add_param(pl, p + "bias", pl.head_dim[i + 1], 0, (int)L, true, &pl.head_b[i], &pl.head_b32[i]);
      }
    } else {
      add_param(pl, "score.weight", c.num_labels, d, (int)L, true, &pl.score, &pl.score32);
      if (c.score_bias) add_param(pl, "score.bias", c.num_labels, 0, (int)L, true, &pl.sbias, &pl.sbias32);
    }
  }
  return pl;
}

// bump allocator over the workspace arena
struct Bump {
  uint64_t off = 0;
  uint64_t take(uint64_t bytes) {
    const uint64_t o = off;
    off += align_up(bytes, 256);
    return o;
  }
};

struct LayerWs {
  uint64_t rstd1, xn1, qkv, lse, attn, araw, xmid, rstd2, xn2, gu, h, mraw;
};

struct Ws {
  uint64_t key_len, pool_row, cos_tab, sin_tab, key_lo, key_hi;
  std::vector<uint64_t> xres;  // L+1 residual-stream snapshots
  std::vector<LayerWs> lw;
  uint64_t rstd_f, hidden;
  uint64_t dxa, dxb, dxc, dxn, dqkv, dattn, dgu, dh, delta, dq_acc, dscaled, dscaled2;
  uint64_t scratch32, loss_sum, loss_part, sqnorm, counts, segs, emb_sort, emb_cnt, emb_slab, wg32, lm_slab;
  uint64_t sq_chunks, sq_tiles;   // gradient-norm shortcut: chunk table of everything but the layers' weight matrices; per-tile sums of those
  // pre-train head
  uint64_t cnt, hc_tot, m_off, l_off, row_idx, sel_src, sel_label, sel_tok, Hm, Pp, Hl, logits, dlogits, dHl, dP, dHm;
  // slot-sorted n_token_proj (kernels.hip k_head_slot_sort): sorted cell lists (token row / dP row / cell index per sorted position), the
  // sorted position of every cell, weight offset per 128-row tile, slot counters, padded total
  uint64_t ss_tok, ss_cell, ss_l, ss_pos, ss_tile, ss_state, ss_total;
  // task head
  uint64_t tlogits, tdlogits, pooled_h, auc_lists;
  uint64_t raw_x, raw_xn, raw_rstd, raw_dxn, raw_dx, raw_flag;   // raw-embedding inputs: blended bf16 [T,e], normalised [T,e], 1/rms [T], their gradients, mask flags [T]
  uint64_t rr_cos, rr_sin, rr_ids;   // rope_range: per-token angle tables [T][32] fp32 and the identity position list [T] int64
  uint64_t tok_stat;   // token-level head: loss sum, labelled rows, 1 / rows
  uint64_t long_wgt;   // stack_method = "long": per-sample loss weights (fp32 [max_batch])
  // var-len token layout: first compact row of every sample [max_batch + 1], per compact row its sample index / position / ids, the
  // padded -> compact row map [max_tokens], a status word
  uint64_t vl_cu, vl_rowb, vl_pos, vl_ids, vl_pad2c, vl_c2p, vl_status;   // (vl_c2p: compact -> padded row map [max_tokens])
  uint64_t vl_long;   // [3 max_batch + 1] int32: count, indices, first rows and row counts of the batch's 33 .. 64-row samples (varlen_scan_kernel)
  uint64_t wo_pack = 0, wot_pack = 0;   // fragment-major copies of every layer's o weight / its transpose (S <= 32 per-sample kernels), [L][d][d] bf16
  uint64_t pos_safe;   // position ids clamped into the RoPE table (int64 [max_tokens]); the sticky "clamped" flag is vl_status[1]
  uint64_t sk_ws;      // stream-K GEMM launches: flags + one fp32 partial tile per block (gemm.h)
  uint64_t head_x[6] = {0}, head_a[5] = {0}, head_d[2] = {0};   // MLP head: layer inputs / activations (bf16), fp32 gradient ping-pong
  uint64_t total;
};

Ws make_ws(const gget_config_t& c, const Plan& pl) {
  Ws w;
  Bump b;
  const uint64_t T = c.max_tokens, d = c.hidden_size, ff = c.intermediate_size, H = c.num_heads, L = c.num_layers;
  const uint64_t Bm = c.max_batch;
  w.key_len = b.take(Bm * 4);
  w.key_lo = b.take(T * 4);   // packed rows: per-token key range (block-diagonal mask)
  w.key_hi = b.take(T * 4);
  w.pool_row = b.take(Bm * 4);
  w.cos_tab = b.take((uint64_t)c.max_position * 32 * 4);
  w.sin_tab = b.take((uint64_t)c.max_position * 32 * 4);
  w.xres.resize(L + 1);
  for (uint64_t i = 0; i <= L; ++i) w.xres[i] = b.take(T * d * 2);
  w.lw.resize(L);
  for (uint64_t i = 0; i < L; ++i) {
    LayerWs& l = w.lw[i];
    l.rstd1 = b.take(T * 4);
    l.xn1 = b.take(T * d * 2);
    l.qkv = b.take(T * 3 * d * 2);
    l.lse = b.take(T * H * 4);
    l.attn = b.take(T * d * 2);
    l.araw = pl.has_res ? b.take(T * d * 2) : 0;
    l.xmid = b.take(T * d * 2);
    l.rstd2 = b.take(T * 4);
    l.xn2 = b.take(T * d * 2);
    l.gu = b.take(T * 2 * ff * 2);
    l.h = b.take(T * ff * 2);
    l.mraw = pl.has_res ? b.take(T * d * 2) : 0;
  }
  w.rstd_f = b.take(T * 4);
  w.hidden = b.take(T * d * 2);
  w.dxa = b.take(T * d * 2);
  w.dxb = b.take(T * d * 2);
  w.dxc = b.take(T * d * 2);
  w.dxn = b.take(T * d * 2);
  w.dqkv = b.take(T * 3 * d * 2);
  w.dattn = b.take(T * d * 2);
  w.dgu = b.take(T * 2 * ff * 2);
  w.dh = b.take(T * ff * 2);
  w.delta = b.take(T * H * 4);
  {   // long sequences: the key blocks' dQ partials of the fused attention backward, bf16 [ceil(S / 256)][T][d]
    const uint64_t Smax = c.max_tokens / (c.max_batch > 0 ? c.max_batch : 1);
    w.dq_acc = Smax >= 256 ? b.take(((Smax + 255) / 256) * T * d * 2) : 0;
  }
  w.dscaled = pl.has_res ? b.take(T * d * 2) : 0;
  w.dscaled2 = pl.has_res ? b.take(T * d * 2) : 0;
  w.scratch32 = b.take(pl.n_scratch32 * 4);
  w.loss_sum = b.take(256);
  w.loss_part = b.take(2048 * sizeof(float));   // one partial loss sum per block of the cross-entropy launch
  w.sqnorm = b.take(k_grad_sqnorm_ws_bytes());
  w.sq_chunks = b.take(1024 * sizeof(GgetSqChunk));
  w.sq_tiles = b.take((uint64_t)c.num_layers * kSqTilesPerLayer * sizeof(float));
  w.counts = b.take(256);
  w.segs = b.take((uint64_t)pl.params.size() * sizeof(GgetSegment));
  w.wg32 = b.take(kWgSplit * 4 * d * d * 4);   // split-K slabs of the q|k|v|o wgrad
  w.long_wgt = b.take((uint64_t)c.max_batch * 4);
  w.vl_cu = b.take((Bm + 1) * 4);
  w.vl_long = b.take((3 * Bm + 1) * 4);
  w.vl_rowb = b.take(T * 4);
  w.vl_pos = b.take(T * 8);
  w.vl_ids = b.take(T * (uint64_t)c.stacked_feat * 8);
  w.vl_pad2c = b.take(T * 4);
  w.vl_c2p = b.take(T * 4);
  if (d % 64 == 0 && !pl.has_res) {
    w.wo_pack = b.take((uint64_t)c.num_layers * d * d * 2);
    w.wot_pack = b.take((uint64_t)c.num_layers * d * d * 2);
  }
  w.vl_status = b.take(256);
  w.pos_safe = b.take(T * 8);
  w.sk_ws = b.take(gget_gemm_streamk_bytes());
  if (c.embed_dim > 0) {
    const uint64_t e = c.embed_dim;
    w.raw_x = b.take(T * e * 2); w.raw_xn = b.take(T * e * 2); w.raw_rstd = b.take(T * 4);
    w.raw_dxn = b.take(T * e * 2); w.raw_dx = b.take(T * e * 2); w.raw_flag = b.take(T * 4);
  }
  w.rr_cos = b.take(T * 32 * 4);
  w.rr_sin = b.take(T * 32 * 4);
  w.rr_ids = b.take(T * 8);
  w.emb_sort = b.take(k_embed_bwd_ws_elems(T * (uint64_t)c.stacked_feat, (uint64_t)c.vocab_size) * 4);
  w.emb_cnt = k_embed_dense_ok(c.vocab_size, pl.has_gate) ? b.take(T * align_up(c.vocab_size, 64) * 2) : 0;   // bf16 [T][Vp] count matrix
  w.emb_slab = k_embed_dense_ok(c.vocab_size, pl.has_gate) ? b.take((uint64_t)kEmbDenseSplit * c.vocab_size * d * 4) : 0;
  if (c.kind == GGET_KIND_PRETRAIN) {
    const uint64_t n = c.next_n_token, Vp = align_up(c.vocab_size, 64);
    w.cnt = b.take(T * 4);
    w.hc_tot = b.take(k_head_compact_ws_bytes((int)T));
    w.m_off = b.take(T * 4);
    w.l_off = b.take(T * 4);
    w.row_idx = b.take(T * 4);
    w.sel_src = b.take(T * n * 4);
    w.sel_label = b.take(T * n * 4);
    w.sel_tok = b.take(T * n * 4);
    w.Hm = b.take(T * d * 2);
    w.dHm = b.take(T * d * 2);
    if (n > 1) {
      const uint64_t cap_p = T * n + n * 256;           // sorted cells + one partial tile of pad rows per slot
      w.ss_tok = b.take(cap_p * 4);
      w.ss_cell = b.take(cap_p * 4);
      w.ss_l = b.take(cap_p * 4);
      w.ss_pos = b.take(T * n * 4);
      w.ss_tile = b.take((cap_p / 128 + 1) * 4);
      w.ss_state = b.take(512);
      w.ss_total = b.take(256);
      w.Pp = b.take(cap_p * d * 2);                     // (Pp doubles as dXs, the per-cell input gradients of the sorted backward)
      w.Hl = b.take(T * n * d * 2);
      w.dHl = b.take(T * n * d * 2);
      w.dP = b.take(T * n * d * 2);
    } else {
      w.Pp = w.Hl = w.Hm;
      w.dHl = w.dP = w.dHm;
    }
    w.lm_slab = b.take((uint64_t)kLmSplit * c.vocab_size * d * 4);
    w.logits = b.take(T * n * Vp * 2);
    w.dlogits = b.take(T * n * Vp * 2);
  } else {
    // (rows = tokens for the token-level head, loss_type = "token_ce": logits / dlogits of every row)
    w.tlogits = b.take(std::max(Bm, T) * c.num_labels * 4);
    w.tdlogits = b.take(std::max(Bm, T) * c.num_labels * 4);
    w.tok_stat = b.take(256);
    w.auc_lists = b.take(Bm * 2 * 4);
    w.pooled_h = b.take(Bm * d * 2);
    if (pl.n_lin > 0) {
      uint64_t dmax = 0;
      for (int i = 0; i <= pl.n_lin; ++i) dmax = std::max<uint64_t>(dmax, (uint64_t)pl.head_dim[i]);
      w.head_x[0] = w.pooled_h;
      for (int i = 1; i <= pl.n_lin; ++i) w.head_x[i] = b.take(Bm * pl.head_dim[i] * 2);
      for (int i = 0; i < pl.n_lin; ++i) w.head_a[i] = b.take(Bm * pl.head_dim[i] * 2);
      w.head_d[0] = b.take(Bm * dmax * 4);
      w.head_d[1] = b.take(Bm * dmax * 4);
    }
  }
  w.total = b.off;
  return w;
}

int check_cfg(const gget_config_t* c) {
  GGET_REQUIRE(c != nullptr, "null config");
  GGET_REQUIRE(c->hidden_size > 0 && c->hidden_size % 64 == 0, "hidden_size must be a positive multiple of 64");
  GGET_REQUIRE(c->num_heads * 64 == c->hidden_size, "num_heads*64 must equal hidden_size (head_dim is 64)");
  GGET_REQUIRE(c->intermediate_size > 0 && c->intermediate_size % 64 == 0, "intermediate_size must be a multiple of 64");
  GGET_REQUIRE(c->hidden_size <= 2048, "hidden_size > 2048 not supported by the norm kernels");
  GGET_REQUIRE(c->vocab_size > 1 && c->num_layers >= 1 && c->stacked_feat >= 1, "bad vocab/layers/stacked_feat");
  GGET_REQUIRE(c->kind == GGET_KIND_PRETRAIN || c->kind == GGET_KIND_TASK, "bad kind");
  GGET_REQUIRE(c->max_tokens > 0 && c->max_batch > 0 && c->max_position > 0, "bad capacities");
  if (c->kind == GGET_KIND_PRETRAIN) GGET_REQUIRE(c->next_n_token >= 1, "next_n_token must be >= 1");
  else GGET_REQUIRE(c->num_labels >= 1, "num_labels must be >= 1");
  GGET_REQUIRE(c->embed_dim >= 0 && c->embed_dim % 64 == 0 && c->embed_dim <= 2048, "embed_dim must be 0 or a multiple of 64 up to 2048");
  GGET_REQUIRE(c->head_mlp_layers >= 0 && c->head_mlp_layers <= 4, "the MLP score head holds at most 4 hidden layers");
  for (int i = 0; i < c->head_mlp_layers; ++i) GGET_REQUIRE(c->head_mlp[i] > 0, "bad MLP head width");
  return 0;
}

}  // namespace

struct gget_engine {
  gget_config_t cfg;
  Plan plan;
  Ws ws;
  bf16_t* P;
  float* master;
  float* am;
  float* av;
  bf16_t* G;
  unsigned char* W;
  const float* cos_tab;
  const float* sin_tab;
  // bucket -> [first,last) params, segments
  std::vector<std::pair<uint64_t, uint64_t>> bucket_range;  // elements
  std::vector<std::pair<int, int>> bucket_segs;              // [first, count) into the device segment table
  // state of the last forward
  // T = rows of the token-major activation buffers: B * S (padded layout) or round_up(real tokens, 64) (var-len layout, see
  // backbone_forward); TP = B * S always (the index space of ids / labels / the SMTP head's selections)
  int B = 0, S = 0, T = 0, TP = 0;
  bool varlen = false;            // the last forward ran on the compacted (padding-free) token layout
  bool wo_packed = false;         // the last forward built the fragment-major o weights (S <= 32 per-sample kernels)
  int tc = 0;                     // its number of real tokens
  long tc_next = -1;              // real-token count of the NEXT forward's batch (gget_set_token_count); -1 = unknown -> padded layout,
                                  // GGET_TOKENS_AUTO = count on the device and read back; consumed (reset) at the ENTRY of every forward
  // gget_set_option(GGET_OPT_NORM_FROM_BACKWARD): the squared gradient norm of the layers' weight matrices comes from partial sums
  // their weight-gradient launches left behind (sq_layers = layers of the last backward that did), the rest from a chunked pass
  bool opt_norm_from_backward = false;
  bool opt_skip_nonfinite = false;     // GGET_OPT_SKIP_NONFINITE_STEP: gget_adamw_step leaves the parameters alone when the gradient norm is inf / NaN
  int sq_layers = 0;
  int n_sq_chunks = 0;
  bool emb_cnt_cleared = false;   // this backward's first launch cleared the embedding gradient's count matrix (gget_backward_begin)
  bool head_sorted_fwd = false;   // the last pre-train forward ran the slot-sorted n_token_proj (its lists feed the backward)
  bool tc_from_caller = false;    // the last var-len forward ran on a caller's count (a wrong one poisons the loss, see poison_loss)
  int32_t* host_word = nullptr;   // pinned host word the counted total lands in
  int32_t* host_word_dev = nullptr;   // ... and its device-side address (sum_lengths_kernel stores to it)
  const int64_t* pos_rows = nullptr;   // position of every ROW (GEMM RoPE epilogue): pos_cur, or the compacted positions
  const int32_t* row_base() const { return varlen ? wsp<int32_t>(ws.vl_cu) : nullptr; }
  const int32_t* long_list() const { return varlen ? wsp<int32_t>(ws.vl_long) : nullptr; }
  const int64_t* ids = nullptr;
  const int64_t* pos = nullptr;
  const float* sample_wgt = nullptr;
  bool have_labels = false;
  int problem = 0;
  int auc_num_neg = 1;
  float focal_gamma = 0.f;        // focal loss on the SMTP head (config.focal_gamma)
  bool stack_long = false;        // config.stack_method == "long" (gget_set_stack_method)
  float rope_range = 0.f;         // config.rope_range (gget_set_rope_range)
  const float* raw_next = nullptr;  // inputs_raw_embeds of the NEXT forward (gget_set_raw_embeds)
  bool raw_first_label_only = false;   // smtp_inside: the mask-token rule looks at labels[:, :, 0] only (modeling_pretrain.py:136-137)
  bool ls2_done = false;          // the LayerScale / DropPath backward of the next layer_backward's down branch is already done (fused)
  bool raw_used = false;          // the last forward consumed raw embeddings (its backward owes their gradients)
  const float* cos_cur = nullptr; // angle tables / position list of the last forward (the per-token ones under rope_range)
  const float* sin_cur = nullptr;
  const int64_t* pos_cur = nullptr;
  // in-step kernel probe (gget_debug_probe): HIP events around the grouped weight-gradient launch [0] and the gate|up + GEGLU launch [1]
  // of every layer, on the stream they are launched on - the bench reads the launch durations INSIDE a step from them
  bool probe = false;
  std::vector<hipEvent_t> probe_ev[2];
  hipEvent_t probe_event(int which, int idx) {
    auto& v = probe_ev[which];
    while ((int)v.size() <= idx) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; v.push_back(e); }
    return v[idx];
  }
  // GGET_TOKENS_AUTO: work of the forward that does not depend on the token count (the SMTP head's row / cell compaction, the packed copies
  // of the o weights) is enqueued BEHIND the 4-byte read-back and BEFORE the host waits for it, so that the device has ~35 us of kernels to
  // run while the host wakes up and enqueues the layer stack (the step trace showed the device idle for 25-37 us there)
  std::function<int(hipStream_t)> presync_work;
  bool presync_done = false;
  hipEvent_t count_event = nullptr;
  unsigned auc_seed = 0;
  bool fwd_valid = false;
  float attn_drop_p = 0.f;        // attention dropout of the NEXT forward (training mode); 0 = off
  float path_drop_p = 0.f;        // stochastic-depth rate of the last layer (layer l: p*l/(L-1))
  unsigned attn_drop_seed = 0;
  float embed_drop_p = 0.f;       // embed_dropout / mlp dropouts of the NEXT forward (training mode); 0 = off
  float mlp_drop_p = 0.f;
  float head_drop_p = 0.f;        // dropout inside the MLP score head
  ElemDropArg head_drop() const { return elem_drop(head_drop_p, attn_drop_seed ^ 0x2545F491u); }
  static ElemDropArg elem_drop(float p, unsigned seed, const int32_t* rows = nullptr) {
    if (p <= 0.f) return ElemDropArg{0, 1.f, 0, nullptr};
    return ElemDropArg{(unsigned)(p * 16777216.0f), 1.0f / (1.0f - p), seed, rows};
  }
  // token-wise masks are keyed by the logical [B,S] row: on the var-len layout through the compact -> padded row map (vl_c2p)
  const int32_t* drop_rows() const { return varlen ? wsp<int32_t>(ws.vl_c2p) : nullptr; }
  ElemDropArg embed_drop() const { return elem_drop(embed_drop_p, attn_drop_seed ^ 0x5BD1E995u, drop_rows()); }
  ElemDropArg mlp_drop(int layer) const { return elem_drop(mlp_drop_p, attn_drop_seed + 0x7F4A7C15u * (unsigned)(layer + 1), drop_rows()); }
  PathDropArg path_drop(int layer, int which) const {
    const int L = cfg.num_layers;
    const float rate = (path_drop_p > 0.f && L > 1) ? path_drop_p * (float)layer / (float)(L - 1) : 0.f;
    return PathDropArg{rate, attn_drop_seed ^ (0xD6E8FEB8u * (unsigned)(layer * 2 + which + 1)), S, varlen ? wsp<int32_t>(ws.vl_rowb) : nullptr};
  }
  bf16_t* dx_cur = nullptr;  // gradient w.r.t. the residual stream entering the next backward stage
  bool packed = false;       // last forward used a 3-D block-diagonal attention mask (per-token key ranges)
  bool defer_convert = false;
  // data-parallel exchange (gget_comm_*): RCCL communicator of this rank and an fp32 staging buffer for the largest bucket
  void* comm = nullptr;
  bool comm_loopback = false;     // gget_comm_init_loopback: no peers, an all-reduce multiplies by comm_world
  int comm_rank = 0, comm_world = 1;
  float* comm_f32 = nullptr;
  uint64_t comm_f32_elems = 0;
  const int32_t* klo() const { return packed ? wsp<int32_t>(ws.key_lo) : nullptr; }
  const int32_t* khi() const { return packed ? wsp<int32_t>(ws.key_hi) : nullptr; }

  template <typename Tp>
  Tp* wsp(uint64_t off) const { return reinterpret_cast<Tp*>(W + off); }
  int bucket_of_layer(int layer) const {  // completion order: heads(L) first, then L-1..0, then embeddings(-1)
    const int L = cfg.num_layers;
    if (layer == L) return 0;
    if (layer < 0) return L + 1;
    return L - layer;
  }
};

// the engine's GEMM launches may run stream-K through the handle's workspace slice (zeroed at creation); op-level calls of the same
// thread afterwards must not see it
struct StreamKScope {
  explicit StreamKScope(gget_engine* h) { gget_gemm_streamk_workspace(h->W + h->ws.sk_ws); }
  ~StreamKScope() { gget_gemm_streamk_workspace(nullptr); }
};

// ================================================================================================
// creation / introspection
// ================================================================================================
extern "C" int gget_query_sizes(const gget_config_t* cfg, gget_sizes_t* out) {
  if (int e = check_cfg(cfg)) return e;
  GGET_REQUIRE(out != nullptr, "null out");
  Plan pl = make_plan(*cfg);
  Ws w = make_ws(*cfg, pl);
  out->n_params = pl.n_params;
  out->param_bf16_bytes = pl.n_params * 2;
  out->master_bytes = pl.n_params * 4;
  out->adam_bytes = pl.n_params * 8;
  out->grad_bf16_bytes = pl.n_params * 2;
  out->workspace_bytes = w.total;
  return 0;
}

extern "C" int gget_create(const gget_config_t* cfg, const gget_buffers_t* bufs, gget_handle_t* out) {
  if (int e = check_cfg(cfg)) return e;
  GGET_REQUIRE(bufs && out, "null argument");
  GGET_REQUIRE(bufs->param_bf16_dev && bufs->grad_bf16_dev && bufs->workspace_dev, "missing device arenas");
  gget_engine* h = new gget_engine();
  h->cfg = *cfg;
  h->plan = make_plan(*cfg);
  h->ws = make_ws(*cfg, h->plan);
  h->P = static_cast<bf16_t*>(bufs->param_bf16_dev);
  h->master = static_cast<float*>(bufs->master_dev);
  h->am = static_cast<float*>(bufs->adam_m_dev);
  h->av = static_cast<float*>(bufs->adam_v_dev);
  h->G = static_cast<bf16_t*>(bufs->grad_bf16_dev);
  h->W = static_cast<unsigned char*>(bufs->workspace_dev);
  // stale rows behind a device-side row count are multiplied by zeros in the head weight gradient: they must be finite
  GGET_HIP_CHECK(hipMemset(h->W, 0, h->ws.total));
  if (h->plan.lm_pad_count) {   // the pad rows must read as zeros whatever the caller's arenas held
    const uint64_t o = h->plan.lm_pad_off, n = h->plan.lm_pad_count;
    GGET_HIP_CHECK(hipMemset(h->P + o, 0, n * 2));
    GGET_HIP_CHECK(hipMemset(h->G + o, 0, n * 2));
    if (h->master) GGET_HIP_CHECK(hipMemset(h->master + o, 0, n * 4));
    if (h->am) GGET_HIP_CHECK(hipMemset(h->am + o, 0, n * 4));
    if (h->av) GGET_HIP_CHECK(hipMemset(h->av + o, 0, n * 4));
  }
  // gradient buckets in completion order + fp32->bf16 conversion segments per bucket
  const int L = cfg->num_layers;
  const int nb = L + 2;
  h->bucket_range.assign(nb, {UINT64_MAX, 0});
  std::vector<std::vector<GgetSegment>> segs(nb);
  for (const ParamRec& p : h->plan.params) {
    const int b = h->bucket_of_layer(p.layer);
    auto& r = h->bucket_range[b];
    r.first = std::min(r.first, p.off);
    r.second = std::max(r.second, p.off + align_up(p.count, 128));
    if (p.accum32)
      segs[b].push_back(GgetSegment{p.off32, p.off, align_up(p.count, 128),
                                    (uint64_t)(p.count <= kAccumCopyMax ? kAccumCopies : 1), align_up(p.count, 128)});
  }
  if (h->plan.lm_pad_count) {   // the zero rows behind lm_head travel with the head bucket: the buckets tile [0, n_params)
    auto& r = h->bucket_range[h->bucket_of_layer(L)];
    r.second = std::max(r.second, h->plan.lm_pad_off + align_up(h->plan.lm_pad_count, 128));
  }
  std::vector<GgetSegment> flat;
  h->bucket_segs.resize(nb);
  for (int b = 0; b < nb; ++b) {
    h->bucket_segs[b] = {(int)flat.size(), (int)segs[b].size()};
    flat.insert(flat.end(), segs[b].begin(), segs[b].end());
  }
  if (!flat.empty()) {
    hipError_t e = hipMemcpy(h->W + h->ws.segs, flat.data(), flat.size() * sizeof(GgetSegment), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      gget_set_error("segment table upload failed: %s", hipGetErrorString(e));
      delete h;
      return 1;
    }
  }
  {
    // chunk table of every gradient that is NOT one of the layers' seven projection matrices (those carry per-tile partial sums)
    uint64_t other = 0;
    auto is_layer_matrix = [&](const ParamRec& p) { return p.layer >= 0 && p.layer < L && !p.accum32 && p.ndim == 2; };
    for (const ParamRec& p : h->plan.params)
      if (!is_layer_matrix(p)) other += align_up(p.count, 128);
    const uint64_t chunk = std::max<uint64_t>(32768, align_up((other + 899) / 900, 128));
    std::vector<GgetSqChunk> chunks;
    for (const ParamRec& p : h->plan.params) {
      if (is_layer_matrix(p)) continue;
      const uint64_t n = align_up(p.count, 128);
      for (uint64_t o = 0; o < n; o += chunk) chunks.push_back(GgetSqChunk{p.off + o, std::min(chunk, n - o)});
    }
    if (h->plan.lm_pad_count) chunks.push_back(GgetSqChunk{h->plan.lm_pad_off, align_up(h->plan.lm_pad_count, 128)});   // (zeros; kept for symmetry with the full pass)
    if (chunks.size() <= 1024) {
      h->n_sq_chunks = (int)chunks.size();
      if (!chunks.empty() &&
          hipMemcpy(h->W + h->ws.sq_chunks, chunks.data(), chunks.size() * sizeof(GgetSqChunk), hipMemcpyHostToDevice) != hipSuccess) {
        gget_set_error("norm chunk table upload failed");
        delete h;
        return 1;
      }
    } else {
      h->n_sq_chunks = -1;   // (never with the chunk size above; the full pass is used then)
    }
  }
  if (bufs->rope_cos_dev && bufs->rope_sin_dev) {
    h->cos_tab = bufs->rope_cos_dev;
    h->sin_tab = bufs->rope_sin_dev;
  } else {
    float* ct = h->wsp<float>(h->ws.cos_tab);
    float* st = h->wsp<float>(h->ws.sin_tab);
    if (k_rope_table(ct, st, cfg->max_position, cfg->rope_theta, nullptr)) { delete h; return 1; }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { gget_set_error("rope table: %s", hipGetErrorString(e)); delete h; return 1; }
    h->cos_tab = ct;
    h->sin_tab = st;
  }
  *out = h;
  return 0;
}

extern "C" int gget_comm_destroy(gget_handle_t h);
extern "C" int gget_destroy(gget_handle_t h) {
  if (h) gget_comm_destroy(h);
  if (h && h->host_word) (void)hipHostFree(h->host_word);
  if (h && h->count_event) (void)hipEventDestroy(h->count_event);
  delete h;
  return 0;
}

extern "C" int gget_param_count(gget_handle_t h) { return h ? (int)h->plan.params.size() : -1; }

extern "C" int gget_param_info(gget_handle_t h, int index, gget_param_info_t* out) {
  GGET_REQUIRE(h && out, "null argument");
  GGET_REQUIRE(index >= 0 && index < (int)h->plan.params.size(), "param index %d out of range", index);
  const ParamRec& p = h->plan.params[index];
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", p.name.c_str());
  out->ndim = p.ndim;
  out->shape[0] = p.shape[0];
  out->shape[1] = p.shape[1];
  out->offset = p.off;
  out->layer = p.layer;
  return 0;
}

extern "C" int gget_bucket_count(gget_handle_t h) { return h ? (int)h->bucket_range.size() : -1; }

extern "C" int gget_bucket_range(gget_handle_t h, int bucket, uint64_t* offset, uint64_t* count) {
  GGET_REQUIRE(h && offset && count, "null argument");
  GGET_REQUIRE(bucket >= 0 && bucket < (int)h->bucket_range.size(), "bucket %d out of range", bucket);
  const auto& r = h->bucket_range[bucket];
  *offset = r.first;
  *count = r.second - r.first;
  return 0;
}

extern "C" int gget_set_dropout_ex(gget_handle_t h, float embed_p, float mlp_p, float head_p) {
  GGET_REQUIRE(h != nullptr, "null handle");
  GGET_REQUIRE(embed_p >= 0.f && embed_p < 1.f && mlp_p >= 0.f && mlp_p < 1.f && head_p >= 0.f && head_p < 1.f,
               "dropout probabilities must be in [0, 1)");
  h->head_drop_p = head_p;
  GGET_REQUIRE(mlp_p == 0.f || h->cfg.mlp_pdrop > 0.f, "MLP dropout needs a handle created with config.mlp_pdrop > 0");
  h->embed_drop_p = embed_p;
  h->mlp_drop_p = mlp_p;
  return 0;
}

extern "C" int gget_set_raw_embeds(gget_handle_t h, const float* raw_embeds_dev, int first_label_only) {
  GGET_REQUIRE(h != nullptr, "null handle");
  GGET_REQUIRE(h->cfg.embed_dim > 0 || raw_embeds_dev == nullptr, "the handle was created with embed_dim = 0");
  h->raw_next = raw_embeds_dev;
  h->raw_first_label_only = first_label_only != 0;
  return 0;
}

extern "C" int gget_set_rope_range(gget_handle_t h, float rope_range) {
  GGET_REQUIRE(h != nullptr, "null handle");
  GGET_REQUIRE(rope_range >= 0.f, "rope_range must be >= 0");
  h->rope_range = rope_range;
  return 0;
}

extern "C" int gget_set_stack_method(gget_handle_t h, int stack_long) {
  GGET_REQUIRE(h != nullptr, "null handle");
  h->stack_long = stack_long != 0;
  return 0;
}

extern "C" int gget_set_focal_gamma(gget_handle_t h, float gamma) {
  GGET_REQUIRE(h != nullptr, "null handle");
  GGET_REQUIRE(gamma >= 0.f, "focal_gamma must be >= 0");
  h->focal_gamma = gamma;
  return 0;
}

extern "C" int gget_set_auc(gget_handle_t h, int num_neg, uint32_t seed) {
  GGET_REQUIRE(h != nullptr, "null handle");
  GGET_REQUIRE(num_neg >= 1, "num_neg must be >= 1");
  h->auc_num_neg = num_neg;
  h->auc_seed = seed;
  return 0;
}

extern "C" int gget_set_token_count(gget_handle_t h, int64_t n_real_tokens) {
  GGET_REQUIRE(h != nullptr, "null handle");
  h->tc_next = n_real_tokens > 0 ? (long)n_real_tokens : (n_real_tokens == GGET_TOKENS_AUTO ? (long)GGET_TOKENS_AUTO : -1);
  return 0;
}

extern "C" int gget_set_option(gget_handle_t h, int option, int value) {
  GGET_REQUIRE(h != nullptr, "null handle");
  switch (option) {
    case GGET_OPT_NORM_FROM_BACKWARD: h->opt_norm_from_backward = value != 0; return 0;
    case GGET_OPT_SKIP_NONFINITE_STEP: h->opt_skip_nonfinite = value != 0; return 0;
  }
  gget_set_error("set_option: unknown option %d", option);
  return 2;
}

extern "C" int gget_deferred_status(gget_handle_t h, int32_t out[2], void* stream) {
  GGET_REQUIRE(h && out, "null argument");
  int32_t* st = h->wsp<int32_t>(h->ws.vl_status);
  GGET_HIP_CHECK(hipMemcpyAsync(&out[0], st + 1, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  GGET_HIP_CHECK(hipMemcpyAsync(&out[1], st + 2, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  GGET_HIP_CHECK(hipMemsetAsync(st + 1, 0, 8, (hipStream_t)stream));
  GGET_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" int gget_position_status(gget_handle_t h, int32_t* clamped_out, void* stream) {
  GGET_REQUIRE(h && clamped_out, "null argument");
  int32_t* flag = h->wsp<int32_t>(h->ws.vl_status) + 1;
  GGET_HIP_CHECK(hipMemcpyAsync(clamped_out, flag, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  GGET_HIP_CHECK(hipMemsetAsync(flag, 0, 4, (hipStream_t)stream));
  GGET_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" int gget_varlen_status(gget_handle_t h, int32_t out[3], void* stream) {
  GGET_REQUIRE(h && out, "null argument");
  out[0] = h->varlen ? 1 : 0;
  out[1] = h->T;
  out[2] = 0;
  if (h->varlen) {
    GGET_HIP_CHECK(hipMemcpyAsync(&out[2], h->wsp<int32_t>(h->ws.vl_status), 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    GGET_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  }
  return 0;
}

extern "C" int gget_set_dropout(gget_handle_t h, float attention_p, float path_p, uint32_t seed) {
  GGET_REQUIRE(h, "null handle");
  GGET_REQUIRE(attention_p >= 0.f && attention_p < 1.f && path_p >= 0.f && path_p < 1.f, "dropout probability out of range");
  GGET_REQUIRE(path_p == 0.f || h->plan.has_res, "path dropout needs a handle created with config.path_pdrop > 0");
  h->attn_drop_p = attention_p;
  h->path_drop_p = path_p;
  h->attn_drop_seed = seed;
  return 0;
}

extern "C" int gget_sync_params(gget_handle_t h, void* stream) {
  GGET_REQUIRE(h && h->master, "sync_params needs the fp32 master arena");
  h->wo_packed = false;      // the bf16 weights change: a backward that follows must not read the fragment-major o-weight copies of the last forward
  return k_f32_to_bf16(h->master, h->P, h->plan.n_params, (hipStream_t)stream);
}

// ================================================================================================
// forward
// ================================================================================================
namespace {

int gemm_nt(const void* A, const void* Bw, void* C, const void* R, int M, int N, int K, int lda, int ldb, int ldc,
            const int* m_dev, hipStream_t st) {
  return gget_gemm_single(GGET_GEMM_NT, R ? GGET_EPI_RESIDUAL : GGET_EPI_NONE, A, Bw, C, R, M, N, K, lda, ldb, ldc, m_dev,
                          nullptr, 1, st);
}
int gemm_nn(const void* A, const void* Bw, void* C, int M, int N, int K, int lda, int ldb, int ldc, const int* m_dev,
            hipStream_t st, const int* c_rows = nullptr) {
  return gget_gemm_single(GGET_GEMM_NN, GGET_EPI_NONE, A, Bw, C, nullptr, M, N, K, lda, ldb, ldc, m_dev, nullptr, 1, st, false, c_rows);
}
// Slot-sorted n_token_proj (kernels.hip: "Slot-sorted SMTP head"): on by default; GGET_HEAD_SORTED=0 / gget_debug_set(8, 1) = the dense
// projection of every selected row through all n slots (rounds 1-3)
int g_head_dense = 0;
int g_head_tile = 0;   // 0: 256-row tiles when d % 256 == 0, else 128 ; 128 / 256: forced (gget_debug_set key 9)
inline int head_tile_rows(int d) { return g_head_tile == 128 ? 128 : ((d % 256) == 0 ? 256 : 128); }
bool head_scatter_fused();
inline bool head_scatter_fused_decl() { return head_scatter_fused(); }
bool head_sorted() {
  static const int off = getenv("GGET_HEAD_SORTED") != nullptr && atoi(getenv("GGET_HEAD_SORTED")) == 0;
  return !off && !g_head_dense && head_scatter_fused_decl();
}
bool head_scatter_fused() {
  static const int off = getenv("GGET_NO_HEAD_SCATTER_FUSION") != nullptr;
  return !off;
}

// Gated-GELU MLP with the activation fused into the GEMMs around it (hf LlamaMLP.forward :174-176):
//   forward : gu = xn2 W_gu^T (kept for the backward), h = bf16(gelu(gate)) * up, one launch
//   backward: dgu = geglu'(dy W_down ; gu), one launch - dh is never materialised
// Available when ff % 128 == 0 (the forward tile holds 128 gate + 128 up columns); otherwise GEMM + element-wise kernel.
bool geglu_fusable(int d, int ff) {
  static const int off = getenv("GGET_NO_GEGLU_FUSION") != nullptr;
  return !off && ff % 128 == 0 && d % 64 == 0;
}
int gateup_geglu(const bf16_t* x, const bf16_t* wgu, bf16_t* gu, bf16_t* hh, int T, int d, int ff, hipStream_t st) {
  if (!geglu_fusable(d, ff)) {
    if (int e = gemm_nt(x, wgu, gu, nullptr, T, 2 * ff, d, d, d, 2 * ff, nullptr, st)) return e;
    return k_geglu_fwd(gu, hh, T, ff, st);
  }
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = 1;
  GemmProblem& p = g.p[0];
  p.A = x; p.B = wgu; p.C = gu; p.C2 = hh;
  p.M = T; p.N = 2 * ff; p.K = d; p.lda = d; p.ldb = d; p.ldc = 2 * ff; p.ldc2 = ff; p.ff = ff;
  return gget_gemm_launch(GGET_GEMM_NT, GGET_EPI_GEGLU_FWD, g, 1, st);
}
int down_dgrad_geglu(const bf16_t* dy, const bf16_t* wdown, const bf16_t* gu, bf16_t* dgu, bf16_t* dh_scratch, int T, int d, int ff,
                     hipStream_t st) {
  if (!geglu_fusable(d, ff)) {
    if (int e = gemm_nn(dy, wdown, dh_scratch, T, ff, d, d, ff, ff, nullptr, st)) return e;
    return k_geglu_bwd(gu, dh_scratch, dgu, T, ff, st);
  }
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = 1;
  GemmProblem& p = g.p[0];
  p.A = dy; p.B = wdown; p.C = dgu; p.G = gu;
  p.M = T; p.N = ff; p.K = d; p.lda = d; p.ldb = ff; p.ldc = 2 * ff; p.ldg = 2 * ff; p.ff = ff;
  return gget_gemm_launch(GGET_GEMM_NN, GGET_EPI_GEGLU_BWD, g, 1, st);
}

// Residual add as its own kernel: out = res + keep_b * (lam * y)
//   lam  : LayerScale vector (utils_graphgpt.py:153-166) or nullptr (= 1)
//   keep_b: DropPath / stochastic depth per SAMPLE (utils_graphgpt.py:64-66 = transformers BeitDropPath): 0 with
//           probability `rate`, else 1/(1-rate); counter-based so backward regenerates it.
typedef PathDropArg PathDrop;
__device__ __forceinline__ float path_keep(const PathDrop& D, long t) {
  if (D.rate <= 0.f) return 1.f;
  unsigned x = D.seed + (unsigned)(D.row_b ? D.row_b[t] : t / D.S) * 0x85EBCA77u;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / 16777216.0f) < D.rate ? 0.f : 1.0f / (1.0f - D.rate);
}
__global__ void __launch_bounds__(256) ls_fwd_kernel(const bf16_t* __restrict__ res, const bf16_t* __restrict__ y,
                                                     const bf16_t* __restrict__ lam, bf16_t* __restrict__ out, long T, int d,
                                                     PathDrop D, ElemDropArg E) {
  const int cpr = d >> 3;
  const long total = T * cpr;
  for (long w = (long)blockIdx.x * 256 + threadIdx.x; w < total; w += (long)gridDim.x * 256) {
    const int c = (int)(w % cpr);
    const float keep = path_keep(D, w / cpr);
    float r[8], v[8], l[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    unpack8(*reinterpret_cast<const uint4*>(res + w * 8), r);
    unpack8(*reinterpret_cast<const uint4*>(y + w * 8), v);
    if (lam) unpack8(*reinterpret_cast<const uint4*>(lam + c * 8), l);
    if (E.thresh) {   // mlp_dropout on the down projection's output (utils_graphgpt.py:79), rounded like the bf16 module does
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = bf2f(f2bf(v[e] * elem_drop_mul(E, GGET_DROP_STREAM_MLP_OUT, elem_row(E, w / cpr), (unsigned)(c * 8 + e))));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] += keep * bf2f(f2bf(l[e] * v[e]));
    *reinterpret_cast<uint4*>(out + w * 8) = pack8(r);
  }
}
// ls_fwd_kernel and the RMSNorm that always follows it (rmsnorm_fwd_kernel, kernels.hip) in one pass over the rows: the residual stream
// is written (out) and normalised (xn, rstd) without being read back - same arithmetic and rounding points as the two kernels.
// A wave owns a row at a time; NCH = 16-byte chunks per lane.
template <int NCH>
__global__ void __launch_bounds__(256) ls_rmsnorm_fwd_kernel(const bf16_t* __restrict__ res, const bf16_t* __restrict__ y,
                                                             const bf16_t* __restrict__ lam, bf16_t* __restrict__ out,
                                                             const bf16_t* __restrict__ w, bf16_t* __restrict__ xn,
                                                             float* __restrict__ rstd_out, int T, int d, float eps, PathDrop D,
                                                             ElemDropArg E) {
  const int lane = threadIdx.x & 63;
  const int nchunk = d >> 3;
  const int stride = gridDim.x * 4;
  float wv[NCH][8], lv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) { wv[i][e] = 0.f; lv[i][e] = 1.f; }
    if (c < nchunk) {
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wv[i]);
      if (lam) unpack8(*reinterpret_cast<const uint4*>(lam + c * 8), lv[i]);
    }
  }
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint4 cr[NCH], cy[NCH], nr[NCH], ny[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    const bool ok = c < nchunk && row < T;
    cr[i] = ok ? *reinterpret_cast<const uint4*>(res + (size_t)row * d + c * 8) : make_uint4(0, 0, 0, 0);
    cy[i] = ok ? *reinterpret_cast<const uint4*>(y + (size_t)row * d + c * 8) : make_uint4(0, 0, 0, 0);
  }
  for (; row < T; row += stride) {
    const int nrow = row + stride;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      const bool ok = c < nchunk && nrow < T;
      nr[i] = ok ? *reinterpret_cast<const uint4*>(res + (size_t)nrow * d + c * 8) : make_uint4(0, 0, 0, 0);
      ny[i] = ok ? *reinterpret_cast<const uint4*>(y + (size_t)nrow * d + c * 8) : make_uint4(0, 0, 0, 0);
    }
    const float keep = path_keep(D, row);
    float v[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      float r[8], yv[8];
      unpack8(cr[i], r);
      unpack8(cy[i], yv);
      if (E.thresh) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          yv[e] = bf2f(f2bf(yv[e] * elem_drop_mul(E, GGET_DROP_STREAM_MLP_OUT, elem_row(E, row), (unsigned)(c * 8 + e))));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] += keep * bf2f(f2bf(lv[i][e] * yv[e]));
      const uint4 pk = pack8(r);                    // the residual stream is a bf16 tensor: the norm sees the rounded values
      if (c < nchunk) *reinterpret_cast<uint4*>(out + (size_t)row * d + c * 8) = pk;
      unpack8(pk, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (lane == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = wv[i][e] * bf2f(f2bf(v[i][e] * rstd));
        *reinterpret_cast<uint4*>(xn + (size_t)row * d + c * 8) = pack8(o);
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) { cr[i] = nr[i]; cy[i] = ny[i]; }
  }
}
int ls_rmsnorm_fwd(const bf16_t* res, const bf16_t* y, const bf16_t* lam, bf16_t* out, const bf16_t* w, bf16_t* xn, float* rstd, int T,
                   int d, float eps, PathDrop D, ElemDropArg E, hipStream_t st) {
  const int grid = (int)std::min<long>(2048, ((long)T + 3) / 4);
  if (d <= 1024) hipLaunchKernelGGL(ls_rmsnorm_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, res, y, lam, out, w, xn, rstd, T, d, eps, D, E);
  else hipLaunchKernelGGL(ls_rmsnorm_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, res, y, lam, out, w, xn, rstd, T, d, eps, D, E);
  GGET_LAUNCH_CHECK();
  return 0;
}
__global__ void __launch_bounds__(256) ls_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                     const bf16_t* __restrict__ lam, bf16_t* __restrict__ dscaled,
                                                     float* __restrict__ dlam, int T, int d, PathDrop D, int copies,
                                                     uint64_t copy_stride, ElemDropArg E) {
  // thread = (row lane, 8-channel chunk); a block walks rows blockIdx.x*RL + lane, stride gridDim.x*RL.  The dlam partials
  // of the block's row lanes are summed through LDS and added to accumulator copy blockIdx.x % copies (GgetSegment).
  extern __shared__ float ls_lds[];   // [RL][d]
  const int CH = d >> 3, RL = 256 / CH;
  const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
  const bool active = rl < RL;
  float l[8] = {1, 1, 1, 1, 1, 1, 1, 1}, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lam && active) unpack8(*reinterpret_cast<const uint4*>(lam + c * 8), l);
  if (active) {
    for (int t = blockIdx.x * RL + rl; t < T; t += gridDim.x * RL) {
      const float keep = path_keep(D, t);
      float g[8], v[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + (size_t)t * d + c * 8), g);
      unpack8(*reinterpret_cast<const uint4*>(y + (size_t)t * d + c * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float em = elem_drop_mul(E, GGET_DROP_STREAM_MLP_OUT, elem_row(E, t), (unsigned)(c * 8 + e));
        o[e] = l[e] * keep * g[e] * em;
        acc[e] += keep * g[e] * bf2f(f2bf(v[e] * em));
      }
      *reinterpret_cast<uint4*>(dscaled + (size_t)t * d + c * 8) = pack8(o);
    }
  }
  if (dlam) {
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) ls_lds[rl * d + c * 8 + e] = acc[e];
    }
    __syncthreads();
    float* dst = dlam + (size_t)(blockIdx.x % copies) * copy_stride;
    for (int j = threadIdx.x; j < d; j += 256) {
      float sum = 0.f;
      for (int r = 0; r < RL; ++r) sum += ls_lds[r * d + j];
      unsafeAtomicAdd(dst + j, sum);
    }
  }
}

// RMSNorm backward (rmsnorm_bwd_kernel, kernels.hip) and the LayerScale / DropPath backward that consumes its result (ls_bwd_kernel) in
// one pass over the rows: dx = dres + rstd (dy w - xhat mean(dy w xhat)) is written once and, rounded to bf16 as the separate kernel would
// read it, scaled into the gradient of the branch behind it (dsc = lam keep dx, dlam += keep dx y).  Same arithmetic as the two kernels.
template <int NCH>
__global__ void __launch_bounds__(256) rmsnorm_bwd_ls_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                             const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                             float* __restrict__ dw_accum, const bf16_t* __restrict__ y,
                                                             const bf16_t* __restrict__ lam, bf16_t* __restrict__ dsc,
                                                             float* __restrict__ dlam_accum, int T, int d, int copies,
                                                             uint64_t copy_stride, PathDrop D, ElemDropArg E) {
  extern __shared__ float red_lds[];  // [2][4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  float dwp[NCH][8], dlp[NCH][8], wv[NCH][8], lv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dwp[i][e] = 0.f; dlp[i][e] = 0.f; wv[i][e] = 0.f; lv[i][e] = 1.f; }
    if (c < nchunk) {
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wv[i]);
      if (lam) unpack8(*reinterpret_cast<const uint4*>(lam + c * 8), lv[i]);
    }
  }
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  uint4 xr[NCH], dr[NCH], rr[NCH], yr[NCH];
  float rstd = 0.f;
  auto fetch = [&](int r) {
    rstd = rstd_in[r];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        xr[i] = *reinterpret_cast<const uint4*>(x + (size_t)r * d + c * 8);
        dr[i] = *reinterpret_cast<const uint4*>(dy + (size_t)r * d + c * 8);
        rr[i] = dres ? *reinterpret_cast<const uint4*>(dres + (size_t)r * d + c * 8) : make_uint4(0, 0, 0, 0);
        yr[i] = *reinterpret_cast<const uint4*>(y + (size_t)r * d + c * 8);
      }
    }
  };
  if (row < T) fetch(row);
  for (; row < T; row += stride) {
    const float rs = rstd;
    float xh[NCH][8], g[NCH][8], res[NCH][8], yv[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float xv[8], dv[8];
        unpack8(xr[i], xv);
        unpack8(dr[i], dv);
        unpack8(rr[i], res[i]);
        unpack8(yr[i], yv[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = xv[e] * rs;
          g[i][e] = dv[e] * wv[i][e];
          dot += g[i][e] * xh[i][e];
          dwp[i][e] += dv[e] * xh[i][e];
        }
      }
    }
    if (row + stride < T) fetch(row + stride);
    dot = wave_sum(dot) / (float)d;
    const float keep = path_keep(D, row);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float o[8], gq[8], sc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = res[i][e] + rs * (g[i][e] - xh[i][e] * dot);
        const uint4 pk = pack8(o);
        *reinterpret_cast<uint4*>(dx + (size_t)row * d + c * 8) = pk;
        unpack8(pk, gq);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float em = elem_drop_mul(E, GGET_DROP_STREAM_MLP_OUT, elem_row(E, row), (unsigned)(c * 8 + e));
          sc[e] = lv[i][e] * keep * gq[e] * em;
          dlp[i][e] += keep * gq[e] * bf2f(f2bf(yv[i][e] * em));
        }
        *reinterpret_cast<uint4*>(dsc + (size_t)row * d + c * 8) = pack8(sc);
      }
    }
  }
  float* dl_lds = red_lds + 4 * d;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red_lds[wave * d + c * 8 + e] = dwp[i][e]; dl_lds[wave * d + c * 8 + e] = dlp[i][e]; }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += 256) {
    const size_t off = (size_t)(blockIdx.x % copies) * copy_stride + j;
    unsafeAtomicAdd(dw_accum + off, red_lds[j] + red_lds[d + j] + red_lds[2 * d + j] + red_lds[3 * d + j]);
    if (dlam_accum) unsafeAtomicAdd(dlam_accum + off, dl_lds[j] + dl_lds[d + j] + dl_lds[2 * d + j] + dl_lds[3 * d + j]);
  }
}
int g_ls_norm_bwd_wide = getenv("GGET_LS_NORM_BWD_WIDE") ? atoi(getenv("GGET_LS_NORM_BWD_WIDE")) : 0;
// The same pass for d = 256 PCH with every lane busy and a third of the registers: lane l owns the 4 channels [256 p + 4 l, +4) of each of
// the PCH pieces of a row (8-byte loads, 512 contiguous bytes per wave instruction), norm weight and LayerScale vector stay packed, the
// residual / branch operands are unpacked where they are used, and there is no software prefetch - ~100 registers = 5 waves per SIMD keep
// 12 loads per lane in flight each instead of 2 waves x 16 (the form above: 230 registers; at d = 768 its second chunk idles half of the
// lanes).  C3 (T = 41 088, d = 768): 104 -> see profiles/r05_step_experiments.txt item 7.  Same expressions per element; the fp32 order of
// the row's dot product differs (other lane assignment), i.e. results equal the form above up to single bf16 rounding flips.
template <int PCH>
__global__ void __launch_bounds__(256) rmsnorm_bwd_ls4_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                              const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                              const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                              float* __restrict__ dw_accum, const bf16_t* __restrict__ y,
                                                              const bf16_t* __restrict__ lam, bf16_t* __restrict__ dsc,
                                                              float* __restrict__ dlam_accum, int T, int copies, uint64_t copy_stride,
                                                              PathDrop D, ElemDropArg E) {
  constexpr int d = 256 * PCH;
  extern __shared__ float red_lds[];  // [2][4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto un4 = [](const uint2& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  };
  uint2 wq[PCH], lq[PCH];
  float dwp[PCH][4], dlp[PCH][4];
#pragma unroll
  for (int p = 0; p < PCH; ++p) {
    wq[p] = *reinterpret_cast<const uint2*>(w + p * 256 + lane * 4);
    lq[p] = lam ? *reinterpret_cast<const uint2*>(lam + p * 256 + lane * 4) : make_uint2(0x3F803F80u, 0x3F803F80u);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dwp[p][e] = 0.f; dlp[p][e] = 0.f; }
  }
  const int stride = gridDim.x * 4;
  // `keepv`: an empty asm the compiler must assume rewrites the packed words - the unpacked copies are recomputed where they are used
  // (one shift / and each) instead of living in registers across the loop (norm weight, LayerScale vector) or across the row's reduction
  // (x, dy): that is what takes the kernel from 149 to <= 102 registers
  auto keepv = [](uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); };
  for (int row0 = blockIdx.x * 4 + wave; row0 < T; row0 += stride) {
    const int row = __builtin_amdgcn_readfirstlane(row0);     // (wave-uniform: scalar row base, 32-bit lane offsets)
    const size_t rb = (size_t)row * d;
    const bf16_t *xrow = x + rb, *dyrow = dy + rb, *yrow = y + rb, *rrow = dres ? dres + rb : nullptr;
    bf16_t *dxrow = dx + rb, *dscrow = dsc + rb;
    const unsigned lo = lane * 4;
    uint2 xr[PCH], dr[PCH], rr[PCH], yr[PCH];
#pragma unroll
    for (int p = 0; p < PCH; ++p) {
      xr[p] = *reinterpret_cast<const uint2*>(xrow + lo + p * 256);
      dr[p] = *reinterpret_cast<const uint2*>(dyrow + lo + p * 256);
    }
#pragma unroll
    for (int p = 0; p < PCH; ++p) {
      rr[p] = rrow ? *reinterpret_cast<const uint2*>(rrow + lo + p * 256) : make_uint2(0, 0);
      yr[p] = *reinterpret_cast<const uint2*>(yrow + lo + p * 256);
    }
    const float rs = rstd_in[row];
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < PCH; ++p) {
      float xv[4], dv[4], wv[4];
      keepv(wq[p]);
      un4(xr[p], xv);
      un4(dr[p], dv);
      un4(wq[p], wv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = xv[e] * rs, g = dv[e] * wv[e];
        dot += g * xh;
        dwp[p][e] += dv[e] * xh;
      }
    }
    dot = wave_sum(dot) / (float)d;
    const float keep = path_keep(D, row);
    const unsigned erow = E.thresh ? elem_row(E, row) : 0u;
#pragma unroll
    for (int p = 0; p < PCH; ++p) {
      float xv[4], dv[4], wv[4], res[4], o[4], gq[4], sc[4], lv[4], yv[4];
      keepv(xr[p]);
      keepv(dr[p]);
      keepv(wq[p]);
      keepv(lq[p]);
      un4(xr[p], xv);
      un4(dr[p], dv);
      un4(wq[p], wv);
      un4(rr[p], res);
      un4(lq[p], lv);
      un4(yr[p], yv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = xv[e] * rs, g = dv[e] * wv[e];
        o[e] = res[e] + rs * (g - xh * dot);
      }
      const uint2 pk = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
      *reinterpret_cast<uint2*>(dxrow + lo + p * 256) = pk;
      un4(pk, gq);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float em = elem_drop_mul(E, GGET_DROP_STREAM_MLP_OUT, erow, (unsigned)(p * 256 + lane * 4 + e));
        sc[e] = lv[e] * keep * gq[e] * em;
        dlp[p][e] += keep * gq[e] * bf2f(f2bf(yv[e] * em));
      }
      *reinterpret_cast<uint2*>(dscrow + lo + p * 256) = make_uint2(pack2bf(sc[0], sc[1]), pack2bf(sc[2], sc[3]));
    }
  }
  float* dl_lds = red_lds + 4 * d;
#pragma unroll
  for (int p = 0; p < PCH; ++p) {
    *reinterpret_cast<float4*>(red_lds + wave * d + p * 256 + lane * 4) = make_float4(dwp[p][0], dwp[p][1], dwp[p][2], dwp[p][3]);
    *reinterpret_cast<float4*>(dl_lds + wave * d + p * 256 + lane * 4) = make_float4(dlp[p][0], dlp[p][1], dlp[p][2], dlp[p][3]);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += 256) {
    const size_t off = (size_t)(blockIdx.x % copies) * copy_stride + j;
    unsafeAtomicAdd(dw_accum + off, red_lds[j] + red_lds[d + j] + red_lds[2 * d + j] + red_lds[3 * d + j]);
    if (dlam_accum) unsafeAtomicAdd(dlam_accum + off, dl_lds[j] + dl_lds[d + j] + dl_lds[2 * d + j] + dl_lds[3 * d + j]);
  }
}
int rmsnorm_bwd_ls(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const float* rstd, const bf16_t* dres, bf16_t* dx, float* dw_accum,
                   const bf16_t* y, const bf16_t* lam, bf16_t* dsc, float* dlam_accum, int T, int d, PathDrop D, ElemDropArg E,
                   hipStream_t st) {
  if (T == 0) return 0;
  const int grid = (int)std::min<long>(4096, ((long)T + 15) / 16);
  const size_t lds = (size_t)8 * d * sizeof(float);
  const uint64_t cs = align_up((uint64_t)d, 128);
  if (!g_ls_norm_bwd_wide && (d == 512 || d == 768 || d == 1024)) {   // (gget_debug_set(11, 1): the 16-byte-chunk form for every width)
    // 5 workgroups of 4 waves per CU in one round; every wave walks rows with a grid stride (fewer partial-sum atomics per row as well)
    const int g4 = (int)std::min<long>(1280, ((long)T + 15) / 16);
#define GGET_LS4(P) hipLaunchKernelGGL(rmsnorm_bwd_ls4_kernel<P>, dim3(g4), dim3(256), lds, st, dy, x, w, rstd, dres, dx, dw_accum, y, lam, dsc, \
                                      dlam_accum, T, kAccumCopies, cs, D, E)
    if (d == 512) GGET_LS4(2); else if (d == 768) GGET_LS4(3); else GGET_LS4(4);
#undef GGET_LS4
    GGET_LAUNCH_CHECK();
    return 0;
  }
  if (d <= 1024)
    hipLaunchKernelGGL(rmsnorm_bwd_ls_kernel<2>, dim3(grid), dim3(256), lds, st, dy, x, w, rstd, dres, dx, dw_accum, y, lam, dsc, dlam_accum, T,
                       d, kAccumCopies, cs, D, E);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_ls_kernel<4>, dim3(grid), dim3(256), lds, st, dy, x, w, rstd, dres, dx, dw_accum, y, lam, dsc, dlam_accum, T,
                       d, kAccumCopies, cs, D, E);
  GGET_LAUNCH_CHECK();
  return 0;
}

bool ls_norm_fused() {
  static const int off = getenv("GGET_NO_LS_NORM_FUSION") != nullptr;
  return !off;
}
int layer_forward(gget_engine* h, int i, hipStream_t st) {
  const gget_config_t& c = h->cfg;
  const int T = h->T, d = c.hidden_size, ff = c.intermediate_size, H = c.num_heads;
  const LayerOff& lo = h->plan.layers[i];
  const LayerWs& lw = h->ws.lw[i];
  bf16_t* x_in = h->wsp<bf16_t>(h->ws.xres[i]);
  bf16_t* x_out = h->wsp<bf16_t>(h->ws.xres[i + 1]);
  bf16_t* xn1 = h->wsp<bf16_t>(lw.xn1);
  bf16_t* qkv = h->wsp<bf16_t>(lw.qkv);
  bf16_t* attn = h->wsp<bf16_t>(lw.attn);
  bf16_t* xmid = h->wsp<bf16_t>(lw.xmid);
  bf16_t* xn2 = h->wsp<bf16_t>(lw.xn2);
  bf16_t* gu = h->wsp<bf16_t>(lw.gu);
  bf16_t* hh = h->wsp<bf16_t>(lw.h);
  // (LayerScale / DropPath path: the previous layer's residual kernel already normalised this layer's input - see below)
  if (!(h->plan.has_res && i > 0 && ls_norm_fused()))
    if (int e = k_rmsnorm_fwd(x_in, h->P + lo.ln1, xn1, h->wsp<float>(lw.rstd1), T, d, c.rms_eps, st)) return e;
  {
    // q|k|v projection with RoPE applied to the q and k columns in the GEMM epilogue (fp32 accumulators, one rounding):
    // qkv holds ROTATED q,k; attention consumes them as they are, its backward rotates dq,dk back.
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = 1;
    GemmProblem& p = g.p[0];
    p.A = xn1; p.B = h->P + lo.wqkv; p.C = qkv;
    p.M = T; p.N = 3 * d; p.K = d; p.lda = d; p.ldb = d; p.ldc = 3 * d;
    p.rope_cos = h->cos_cur; p.rope_sin = h->sin_cur; p.rope_pos = h->pos_rows; p.rope_S = h->S; p.rope_cols = 2 * d;
    if (int e = gget_gemm_launch(GGET_GEMM_NT, GGET_EPI_ROPE, g, 1, st)) return e;
  }
  if (!h->plan.has_res && !h->klo()) {
    // S <= 32 (the graph sequences of the headline workload): attention, o projection + residual and post_attention_layernorm of a
    // sample in ONE workgroup (attention.hip: attn_oproj_fwd_kernel) instead of three latency-bound launches
    int taken = 0;
    if (h->wo_packed)
    if (int e = k_attn_oproj_fwd(qkv, h->wsp<int32_t>(h->ws.key_len), h->row_base(), attn, h->wsp<float>(lw.lse),
                                 h->wsp<bf16_t>(h->ws.wo_pack) + (size_t)i * d * d, x_in, xmid,
                                 h->P + lo.ln2, xn2, h->wsp<float>(lw.rstd2), h->B, h->S, H, c.causal, c.rms_eps, h->attn_drop_p,
                                 h->attn_drop_seed + 0x9E37u * (unsigned)i, st, &taken))
      return e;
    if (taken) {
      if (h->probe) GGET_HIP_CHECK(hipEventRecord(h->probe_event(1, 2 * i), st));
      if (int e = gateup_geglu(xn2, h->P + lo.wgu, gu, hh, T, d, ff, st)) return e;
      if (h->probe) GGET_HIP_CHECK(hipEventRecord(h->probe_event(1, 2 * i + 1), st));
      return gemm_nt(hh, h->P + lo.wdown, x_out, xmid, T, d, ff, ff, ff, d, nullptr, st);
    }
  }
  if (int e = k_attn_fwd(qkv, h->wsp<int32_t>(h->ws.key_len), attn, h->wsp<float>(lw.lse), h->B, h->S, H, c.causal,
                         nullptr, nullptr, nullptr, h->attn_drop_p, h->attn_drop_seed + 0x9E37u * (unsigned)i, st, h->klo(),
                         h->khi(), h->row_base(), h->long_list()))
    return e;
  if (h->plan.has_res) {
    bf16_t* araw = h->wsp<bf16_t>(lw.araw);
    bf16_t* mraw = h->wsp<bf16_t>(lw.mraw);
    const bf16_t* lam1 = h->plan.has_ls ? h->P + lo.lam1 : nullptr;
    const bf16_t* lam2 = h->plan.has_ls ? h->P + lo.lam2 : nullptr;
    const PathDrop pd1 = h->path_drop(i, 0), pd2 = h->path_drop(i, 1);
    const int g = (int)std::min<long>(4096, ((long)T * (d / 8) + 255) / 256);
    if (int e = gemm_nt(attn, h->P + lo.wo, araw, nullptr, T, d, d, d, d, d, nullptr, st)) return e;
    // the residual kernels also produce what the RMSNorm behind them would: the residual stream is written and normalised in one pass
    // (the second one normalises for the NEXT layer, or with the final norm)
    const bool fused = ls_norm_fused();
    if (fused) {
      if (int e = ls_rmsnorm_fwd(x_in, araw, lam1, xmid, h->P + lo.ln2, xn2, h->wsp<float>(lw.rstd2), T, d, c.rms_eps, pd1,
                                 ElemDropArg{0, 1.f, 0}, st))
        return e;
    } else {
      hipLaunchKernelGGL(ls_fwd_kernel, dim3(g), dim3(256), 0, st, x_in, araw, lam1, xmid, (long)T, d, pd1, ElemDropArg{0, 1.f, 0});
      if (int e = k_rmsnorm_fwd(xmid, h->P + lo.ln2, xn2, h->wsp<float>(lw.rstd2), T, d, c.rms_eps, st)) return e;
    }
    if (int e = gateup_geglu(xn2, h->P + lo.wgu, gu, hh, T, d, ff, st)) return e;
    const ElemDropArg md = h->mlp_drop(i);
    if (int e = k_elem_dropout(hh, T, ff, GGET_DROP_STREAM_MLP_ACT, md, st)) return e;     // mlp_act_dropout (utils_graphgpt.py:78)
    if (int e = gemm_nt(hh, h->P + lo.wdown, mraw, nullptr, T, d, ff, ff, ff, d, nullptr, st)) return e;
    if (fused) {
      const bool last = i + 1 == c.num_layers;
      const bf16_t* nw = last ? h->P + h->plan.normf : h->P + h->plan.layers[i + 1].ln1;
      bf16_t* nxn = last ? h->wsp<bf16_t>(h->ws.hidden) : h->wsp<bf16_t>(h->ws.lw[i + 1].xn1);
      float* nrs = last ? h->wsp<float>(h->ws.rstd_f) : h->wsp<float>(h->ws.lw[i + 1].rstd1);
      return ls_rmsnorm_fwd(xmid, mraw, lam2, x_out, nw, nxn, nrs, T, d, c.rms_eps, pd2, md, st);
    }
    hipLaunchKernelGGL(ls_fwd_kernel, dim3(g), dim3(256), 0, st, xmid, mraw, lam2, x_out, (long)T, d, pd2, md);
    GGET_LAUNCH_CHECK();
    return 0;
  }
  if (int e = gemm_nt(attn, h->P + lo.wo, xmid, x_in, T, d, d, d, d, d, nullptr, st)) return e;
  if (int e = k_rmsnorm_fwd(xmid, h->P + lo.ln2, xn2, h->wsp<float>(lw.rstd2), T, d, c.rms_eps, st)) return e;
  if (h->probe) GGET_HIP_CHECK(hipEventRecord(h->probe_event(1, 2 * i), st));
  if (int e = gateup_geglu(xn2, h->P + lo.wgu, gu, hh, T, d, ff, st)) return e;
  if (h->probe) GGET_HIP_CHECK(hipEventRecord(h->probe_event(1, 2 * i + 1), st));
  if (int e = gemm_nt(hh, h->P + lo.wdown, x_out, xmid, T, d, ff, ff, ff, d, nullptr, st)) return e;
  return 0;
}

bool varlen_enabled() {
  static const int off = getenv("GGET_NO_VARLEN") != nullptr;
  return !off;
}
int backbone_forward(gget_engine* h, long tc_hint, const int64_t* ids, int ldF, const int64_t* mask, const int64_t* pos, int B, int S,
                     hipStream_t st, bool mask_is_3d = false, const int64_t* labels = nullptr, bool allow_varlen = true) {
  const gget_config_t& c = h->cfg;
  GGET_REQUIRE(B > 0 && S > 0, "empty batch");
  GGET_REQUIRE((long)B * S <= c.max_tokens && B <= c.max_batch, "batch %dx%d exceeds capacity (%d tokens, %d rows)", B, S,
               c.max_tokens, c.max_batch);
  GGET_REQUIRE(S <= c.max_position, "sequence length %d exceeds max_position %d", S, c.max_position);
  h->B = B; h->S = S; h->T = h->TP = B * S;
  if (pos && !(h->rope_range > 0.f)) {   // (rope_range: angles are evaluated per token from the rescaled positions - no table to overrun)
    if (int e = k_clamp_positions(pos, h->wsp<int64_t>(h->ws.pos_safe), h->wsp<int32_t>(h->ws.vl_status) + 1, (long)B * S, c.max_position, st))
      return e;
    pos = h->wsp<int64_t>(h->ws.pos_safe);
  }
  h->ids = ids; h->pos = pos;
  h->cos_cur = h->cos_tab; h->sin_cur = h->sin_tab; h->pos_cur = pos;
  // Var-len (padding-free) token layout: when the number of real tokens of the right-padded batch is known - from the caller
  // (gget_set_token_count(n): the host knows sum(attention_mask) from its collator) or counted here (GGET_TOKENS_AUTO: the key lengths
  // are summed on the device and the total read back: 4 bytes and ONE stream synchronisation, which is what the reference's step pays
  // on every `.to(device)` of a batch tensor anyway, training_utils.py:17-26) - the real tokens are compacted sample after sample and
  // every token-wise kernel and GEMM of the layer stack runs on round_up(real, 64) rows instead of B * S (PCQM4M-v2 batches are ~30 %
  // padding, ogbl-ppa ~37 %).  Logical [B,S] quantities (labels, lse, dropout coordinates, loss normalisers) are unchanged: the
  // element dropouts hash the logical row (ElemDropArg::rows = vl_c2p), the per-token rope_range tables are indexed through the compact
  // position list (= the logical row); raw-embedding inputs, the token-level head's labels / logits and the cells of full-logit inference
  // go through the same maps (vl_c2p / vl_pad2c).  Packed rows carry no padding to begin with; hidden-state accessors need the padded grid.
  h->varlen = false;
  h->tc_from_caller = false;
  h->tc = B * S;
  const int d = c.hidden_size;
  h->packed = mask_is_3d;
  auto pack_o_weights = [&]() -> int {
    // S <= 32 without LayerScale / DropPath: the layers run the per-sample attention + o projection kernels (attention.hip), which read
    // fragment-major copies of the o weights.  They are rebuilt at the start of EVERY forward (one launch, ~10 us: the bf16 weights are the
    // caller's memory and change under AdamW, gget_sync_params or a caller's own writes); the backward of this forward uses the same copies.
    // 32 < S <= 64: only the BACKWARD's per-sample kernel runs (var-len layout, every sample by its own row count); the copies are built
    // whenever that layout can still be chosen for this forward
    h->wo_packed = false;
    const bool may_varlen = allow_varlen && varlen_enabled() && (tc_hint > 0 || tc_hint == GGET_TOKENS_AUTO) && mask != nullptr;
    // One workgroup per sample, one per CU: B samples are ceil(B / CUs) ROUNDS of the kernel whatever the last round holds.  Up to one
    // round they always pay; beyond, only when the last round is nearly full (B = 288 / 320 / 384 on 256 CUs: 8.20 / 8.51 / 9.90 ms per
    // step against 7.95 / 8.32 / 9.73 with the three launches, profiles/r06_step_experiments.txt item 8)
    const int ncu = gget_gemm_num_cu() > 0 ? gget_gemm_num_cu() : 256;
    const int rounds = (B + ncu - 1) / ncu;
    const bool rounds_ok = B <= ncu || (long)B * 100 >= (long)rounds * ncu * 85;
    if (!h->plan.has_res && !mask_is_3d && (S <= 32 || (S <= 64 && may_varlen)) && rounds_ok && h->ws.wo_pack && c.num_layers > 0) {
      const size_t stride = c.num_layers > 1 ? h->plan.layers[1].wo - h->plan.layers[0].wo : 0;
      bool regular = true;
      for (int i = 0; i < c.num_layers; ++i) regular = regular && h->plan.layers[i].wo == h->plan.layers[0].wo + (size_t)i * stride;
      if (regular) {
        if (int e = k_pack_wo(h->P + h->plan.layers[0].wo, stride, h->wsp<bf16_t>(h->ws.wo_pack), h->wsp<bf16_t>(h->ws.wot_pack), d, c.num_layers, st))
          return e;
        h->wo_packed = true;
      }
    }
    return 0;
  };
  bool early_done = false;
  auto early_work = [&]() -> int {     // token-count independent launches (see gget_engine::presync_work)
    if (early_done) return 0;
    early_done = true;
    if (int e = pack_o_weights()) return e;
    if (h->presync_work) {
      if (int e = h->presync_work(st)) return e;
      h->presync_done = true;
    }
    return 0;
  };
  if (mask_is_3d) {
    GGET_REQUIRE(mask != nullptr && c.kind == GGET_KIND_PRETRAIN, "a 3-D (packed) attention mask needs the pre-train model and a mask");
    if (int e = k_ranges_from_mask3d(mask, h->wsp<int32_t>(h->ws.key_lo), h->wsp<int32_t>(h->ws.key_hi), B, S, st)) return e;
  } else if (int e = k_lengths(mask, c.kind == GGET_KIND_TASK ? ids : nullptr, ldF, c.pad_token_id,
                               h->wsp<int32_t>(h->ws.key_len),
                               c.kind == GGET_KIND_TASK ? h->wsp<int32_t>(h->ws.pool_row) : nullptr, B, S, st)) {
    return e;
  }
  if (allow_varlen && varlen_enabled() && (tc_hint > 0 || tc_hint == GGET_TOKENS_AUTO) && !mask_is_3d && mask != nullptr) {
    long tc = tc_hint;
    if (tc_hint == GGET_TOKENS_AUTO) {
      int32_t* dst = h->wsp<int32_t>(h->ws.vl_status) + 3;
      if (!h->host_word) {     // (coherent: a store from a running kernel is visible to the polling host)
        GGET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->host_word), 64, hipHostMallocMapped | hipHostMallocCoherent));
        GGET_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->host_word_dev), h->host_word, 0));
      }
      // Round 6: sum_lengths_kernel stores the count into the pinned word itself (system-scope release) and the host POLLS the word -
      // no device-to-host copy packet and no event in the stream (a 4 us blit kernel plus a 5.7 us bubble behind the event's barrier
      // packet in every step before).  GGET_COUNT_COPY=1: the copy + event of rounds 3 - 5.
      static const bool count_copy = getenv("GGET_COUNT_COPY") && atoi(getenv("GGET_COUNT_COPY")) != 0;
      constexpr int32_t kNoCount = INT32_MIN;
      if (!count_copy) __atomic_store_n(h->host_word, kNoCount, __ATOMIC_RELEASE);     // (before the launch: the launch orders it)
      if (int e = k_sum_lengths(h->wsp<int32_t>(h->ws.key_len), B, dst, st, count_copy ? nullptr : h->host_word_dev)) return e;
      if (count_copy) {
        GGET_HIP_CHECK(hipMemcpyAsync(h->host_word, dst, 4, hipMemcpyDeviceToHost, st));
        if (!h->count_event) GGET_HIP_CHECK(hipEventCreateWithFlags(&h->count_event, hipEventDisableTiming));
        GGET_HIP_CHECK(hipEventRecord(h->count_event, st));
      }
      if (int e = early_work()) return e;                      // (runs on the device while the host waits for the 4 bytes)
      if (count_copy) {
        GGET_HIP_CHECK(hipEventSynchronize(h->count_event));
        tc = *h->host_word;
      } else {
        int32_t v = kNoCount;
        const auto t0 = std::chrono::steady_clock::now();
        for (long spin = 0; (v = __atomic_load_n(h->host_word, __ATOMIC_ACQUIRE)) == kNoCount; ++spin) {
#if defined(__x86_64__)
          __builtin_ia32_pause();     // (a polite spin: the sibling hyper-thread may be a data-loader worker)
#endif
          if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            // (a stream that is held up - another process on the GPU, a debugger - or a failed launch: wait for the stream, then look again)
            GGET_HIP_CHECK(hipStreamSynchronize(st));
            v = __atomic_load_n(h->host_word, __ATOMIC_ACQUIRE);
            GGET_REQUIRE(v != kNoCount, "the token count never arrived from the device");
            break;
          }
        }
        tc = v;
      }
    }
    const long t_rows = (tc + 63) / 64 * 64;
    if (tc > 0 && t_rows < (long)B * S) {
      h->varlen = true;
      h->tc_from_caller = tc_hint > 0;
      h->tc = (int)tc;
      h->T = (int)t_rows;
    }
  }
  if (int e = early_work()) return e;
  if (h->rope_range > 0.f && pos) {
    if (int e = k_rope_range_table(pos, h->wsp<float>(h->ws.rr_cos), h->wsp<float>(h->ws.rr_sin), h->wsp<int64_t>(h->ws.rr_ids), B, S,
                                   h->rope_range, c.rope_theta > 0.f ? c.rope_theta : 10000.0f, st))
      return e;
    h->cos_cur = h->wsp<float>(h->ws.rr_cos); h->sin_cur = h->wsp<float>(h->ws.rr_sin); h->pos_cur = h->wsp<int64_t>(h->ws.rr_ids);
  }
  h->pos_rows = h->pos_cur;
  if (h->varlen) {
    const Ws& w = h->ws;
    // (h->pos_cur: the caller's positions, or with rope_range the identity list into the per-token angle tables - a compact row then
    //  reads the table row of its logical token)
    if (int e = k_varlen_plan(ids, ldF, c.stacked_feat, h->pos_cur, h->wsp<int32_t>(w.key_len),
                              c.kind == GGET_KIND_TASK ? h->wsp<int32_t>(w.pool_row) : nullptr, h->wsp<int32_t>(w.vl_cu),
                              h->wsp<int64_t>(w.vl_ids), h->wsp<int64_t>(w.vl_pos), h->wsp<int32_t>(w.vl_rowb), h->wsp<int32_t>(w.vl_pad2c),
                              h->wsp<int32_t>(w.vl_c2p), h->wsp<int32_t>(w.vl_status), B, S, h->tc, h->T, c.pad_token_id, st,
                              h->wsp<int32_t>(w.vl_long)))
      return e;
    h->ids = ids = h->wsp<int64_t>(w.vl_ids);
    ldF = c.stacked_feat;
    h->pos_rows = h->wsp<int64_t>(w.vl_pos);
  }
  if (int e = k_embed_fwd(ids, h->P + h->plan.emb, h->plan.has_gate ? h->P + h->plan.gate : nullptr,
                          h->wsp<bf16_t>(h->ws.xres[0]), h->T, c.stacked_feat, ldF, d, st, h->embed_drop()))
    return e;
  if (h->stack_long)
    if (int e = k_embed_long_ratio(ids, h->wsp<bf16_t>(h->ws.xres[0]), h->T, c.stacked_feat, ldF, d, st)) return e;
  h->raw_used = false;
  if (c.embed_dim > 0) {
    // raw-embedding inputs (modeling_pretrain.py:131-149, modeling_helpers.py:127-139): [rows that carry a label -> emb_mask_token,]
    // RMSNorm, dropout, embed_proj, added to the stacked token embeddings (the sum is rounded once, in the GEMM epilogue)
    GGET_REQUIRE(h->raw_next != nullptr, "this model was built with embed_dim > 0: hand the raw embeddings over with gget_set_raw_embeds");
    const Ws& w = h->ws;
    const int e_ = c.embed_dim;
    const bool blend = c.kind == GGET_KIND_PRETRAIN && labels != nullptr;
    if (int e = k_raw_blend(h->raw_next, blend ? labels : nullptr, c.next_n_token, h->raw_first_label_only,
                            blend ? h->P + h->plan.raw_tok : nullptr, h->wsp<bf16_t>(w.raw_x), h->wsp<int32_t>(w.raw_flag), h->T, e_, st,
                            h->varlen ? h->wsp<int32_t>(w.vl_c2p) : nullptr, B * S))     // (var-len: raw rows / labels through the logical row)
      return e;
    if (int e = k_rmsnorm_fwd(h->wsp<bf16_t>(w.raw_x), h->P + h->plan.raw_ln, h->wsp<bf16_t>(w.raw_xn), h->wsp<float>(w.raw_rstd), h->T, e_,
                              c.rms_eps, st))
      return e;
    if (int e = k_elem_dropout(h->wsp<bf16_t>(w.raw_xn), h->T, e_, GGET_DROP_STREAM_RAW, h->embed_drop(), st)) return e;
    bf16_t* x0 = h->wsp<bf16_t>(w.xres[0]);
    if (int e = gemm_nt(h->wsp<bf16_t>(w.raw_xn), h->P + h->plan.raw_proj, x0, x0, h->T, d, e_, e_, e_, d, nullptr, st)) return e;
    h->raw_used = true;
    h->raw_next = nullptr;
  }
  for (int i = 0; i < c.num_layers; ++i)
    if (int e = layer_forward(h, i, st)) return e;
  if (h->plan.has_res && ls_norm_fused()) return 0;   // (the last layer's residual kernel applied the final norm)
  return k_rmsnorm_fwd(h->wsp<bf16_t>(h->ws.xres[c.num_layers]), h->P + h->plan.normf, h->wsp<bf16_t>(h->ws.hidden),
                       h->wsp<float>(h->ws.rstd_f), h->T, d, c.rms_eps, st);
}

}  // namespace

static int forward_pretrain_impl(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev,
                                 bool mask_is_3d, const int64_t* labels_dev, const float* sample_wgt_dev,
                                 const int64_t* position_ids_dev, int B, int S, float* loss_dev, void* stream) {
  GGET_REQUIRE(h != nullptr, "null argument");
  const long tc_hint = h->tc_next;   // the token count belongs to THIS call whatever happens below (a failing forward must not leave it
  h->tc_next = -1;                   // to the next batch)
  GGET_REQUIRE(input_ids_dev, "null argument");
  GGET_REQUIRE(h->cfg.kind == GGET_KIND_PRETRAIN, "handle was not created as a pre-train model");
  hipStream_t st = (hipStream_t)stream;
  const gget_config_t& c = h->cfg;
  StreamKScope sk_scope(h);
  h->fwd_valid = false;
  const Ws& w = h->ws;
  const int d = c.hidden_size, n = c.next_n_token, V = c.vocab_size;
  const int Vp = (int)align_up(V, 64);
  int32_t* counts = h->wsp<int32_t>(w.counts);
  // row / cell compaction of the head: reads the labels only (padded coordinates), so it is handed to the backbone as token-count
  // independent work (gget_engine::presync_work) - with GGET_TOKENS_AUTO it runs while the host waits for the count
  // (the slot-sorted n_token_proj below: its per-slot cell counts are taken by the same launches)
  const bool sorted_head = h->plan.has_ntp && head_sorted() && d % 192 == 0 && n <= 32;
  auto head_compact = [&](hipStream_t s_) -> int {
    return k_head_compact(labels_dev, B * S, n, h->wsp<int32_t>(w.cnt), h->wsp<int32_t>(w.m_off), h->wsp<int32_t>(w.l_off), counts,
                          h->wsp<int32_t>(w.row_idx), h->wsp<int32_t>(w.sel_src), h->wsp<int32_t>(w.sel_label), h->wsp<int32_t>(w.sel_tok),
                          h->wsp<int32_t>(w.hc_tot), sorted_head ? h->wsp<int32_t>(w.ss_state) : nullptr, s_);
  };
  h->presync_work = head_compact;
  h->presync_done = false;
  const int rc_bb = backbone_forward(h, tc_hint, input_ids_dev, c.stacked_feat, attention_mask_dev, position_ids_dev, B, S, st, mask_is_3d, labels_dev);
  h->presync_work = nullptr;       // (captures this frame)
  if (rc_bb) {
    if (sorted_head && h->presync_done)     // (the slot counts were taken and nobody will consume them)
      (void)hipMemsetAsync(h->wsp<int32_t>(w.ss_state), 0, 512, st);
    return rc_bb;
  }
  const int T = h->TP;             // (capacities of the head: the padded token space)
  if (!h->presync_done)
    if (int e = head_compact(st)) return e;
  if (h->varlen) {  // the selected rows live at their compact positions (sel_tok / sel_label keep the padded coordinates the loss weights need)
    // (full-logit inference - labels == NULL, generation - selects every cell of the padded grid: the cells of padded positions read a
    //  pad-token row, their logits are defined but meaningless exactly like the reference's, and no flag is raised)
    if (int e = k_gather_rows_remap(h->wsp<bf16_t>(w.hidden), h->wsp<int32_t>(w.row_idx), counts, h->wsp<int32_t>(w.vl_pad2c),
                                    h->wsp<bf16_t>(w.Hm), T, d, h->T > h->tc ? h->tc : 0, labels_dev ? h->wsp<int32_t>(w.vl_status) : nullptr, st))
      return e;
  } else if (int e = k_gather_rows(h->wsp<bf16_t>(w.hidden), h->wsp<int32_t>(w.row_idx), counts, h->wsp<bf16_t>(w.Hm), T, d, 0, st))
    return e;
  h->head_sorted_fwd = false;
  if (sorted_head) {
    // n_token_proj on the labelled cells only (modeling_helpers.py:263-301 computes all n slots of every selected row and drops the
    // unlabelled half): cells sorted by slot, one GEMM with a weight block per row tile, rows gathered from `hidden` and scattered
    // straight to Hl's (m, f) order - the dense [M, n d] product and its gather are gone
    if (int e = k_head_slot_sort(h->wsp<int32_t>(w.sel_src), h->wsp<int32_t>(w.row_idx), counts + 1, h->wsp<int32_t>(w.ss_state),
                                 h->wsp<int32_t>(w.ss_tok), h->wsp<int32_t>(w.ss_cell), h->wsp<int32_t>(w.ss_l), h->wsp<int32_t>(w.ss_pos),
                                 h->wsp<int32_t>(w.ss_tile), h->wsp<int32_t>(w.ss_total), T * n, n, (long)d * d, head_tile_rows(d), st))
      return e;
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = 1;
    GemmProblem& p = g.p[0];
    p.A = h->wsp<bf16_t>(w.hidden); p.B = h->P + h->plan.ntp; p.C = h->wsp<bf16_t>(w.Hl);
    p.M = T * n + n * 256; p.N = d; p.K = d; p.lda = d; p.ldb = d; p.ldc = d;
    p.m_dev = h->wsp<int32_t>(w.ss_total);
    p.a_rows = h->wsp<int32_t>(w.ss_tok); p.b_tile_off = h->wsp<int32_t>(w.ss_tile); p.c_rows = h->wsp<int32_t>(w.ss_l);
    p.b_tile_rows = head_tile_rows(d);
    if (int e = gget_gemm_launch(GGET_GEMM_NT, GGET_EPI_NONE, g, 1, st)) return e;
    h->head_sorted_fwd = true;
  } else if (h->plan.has_ntp) {
    if (int e = gemm_nt(h->wsp<bf16_t>(w.Hm), h->P + h->plan.ntp, h->wsp<bf16_t>(w.Pp), nullptr, T, n * d, d, d, d, n * d,
                        counts, st))
      return e;
    if (int e = k_gather_rows(h->wsp<bf16_t>(w.Pp), h->wsp<int32_t>(w.sel_src), counts + 1, h->wsp<bf16_t>(w.Hl), T * n, d, 0,
                              st))
      return e;
  }
  if (int e = gemm_nt(h->wsp<bf16_t>(w.Hl), h->P + h->plan.lm, h->wsp<bf16_t>(w.logits), nullptr, T * n, V, d, d, d, Vp,
                      counts + 1, st))
    return e;
  h->have_labels = labels_dev != nullptr;
  if (h->stack_long && labels_dev) {   // the per-feature-level weights replace whatever the caller passed (modeling_helpers.py:368-374)
    if (int e = k_sample_mask_wgt(labels_dev, h->wsp<float>(w.long_wgt), B, S * n, st)) return e;
    sample_wgt_dev = h->wsp<float>(w.long_wgt);
  }
  h->sample_wgt = sample_wgt_dev;
  if (labels_dev) {
    const int mean_rows = sample_wgt_dev == nullptr;
    const float base = 1.0f / (float)((long)B * S * n);  // dLM normaliser, modeling_pretrain.py:230-236
    if (int e = k_ce_fwd_bwd(h->wsp<bf16_t>(w.logits), Vp, h->wsp<int32_t>(w.sel_label), h->wsp<int32_t>(w.sel_tok),
                             sample_wgt_dev, S, counts + 1, T * n, V, h->wsp<float>(w.loss_sum), h->wsp<bf16_t>(w.dlogits),
                             base, mean_rows, loss_dev, st, mean_rows ? h->focal_gamma : 0.f, h->wsp<float>(w.loss_part), 2048))   // (the dLM-weighted loss has no focal form)
      return e;
    if (h->varlen && h->tc_from_caller && loss_dev)
      if (int e = k_poison_loss(h->wsp<int32_t>(w.vl_status), loss_dev, st)) return e;
  }
  h->fwd_valid = true;
  return 0;
}

extern "C" int gget_forward_pretrain(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev,
                                     const int64_t* labels_dev, const float* sample_wgt_dev,
                                     const int64_t* position_ids_dev, int B, int S, float* loss_dev, void* stream) {
  return forward_pretrain_impl(h, input_ids_dev, attention_mask_dev, false, labels_dev, sample_wgt_dev, position_ids_dev, B, S,
                               loss_dev, stream);
}

extern "C" int gget_forward_pretrain_packed(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask3d_dev,
                                            const int64_t* labels_dev, const float* sample_wgt_dev,
                                            const int64_t* position_ids_dev, int B, int S, float* loss_dev, void* stream) {
  GGET_REQUIRE(attention_mask3d_dev != nullptr, "the packed forward needs the [B,S,S] mask");
  return forward_pretrain_impl(h, input_ids_dev, attention_mask3d_dev, true, labels_dev, sample_wgt_dev, position_ids_dev, B, S,
                               loss_dev, stream);
}

extern "C" int gget_forward_task(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev,
                                 const int64_t* position_ids_dev, const void* task_labels_dev, const float* sample_wgt_dev,
                                 int problem_type, int B, int S, float* loss_dev, float* task_logits_dev,
                                 void* task_hidden_dev, void* stream) {
  GGET_REQUIRE(h != nullptr, "null argument");
  const long tc_hint = h->tc_next;   // (consumed at entry: see forward_pretrain_impl)
  h->tc_next = -1;
  GGET_REQUIRE(input_ids_dev, "null argument");
  GGET_REQUIRE(h->cfg.kind == GGET_KIND_TASK, "handle was not created as a task model");
  hipStream_t st = (hipStream_t)stream;
  const gget_config_t& c = h->cfg;
  StreamKScope sk_scope(h);
  h->fwd_valid = false;
  if (int e = backbone_forward(h, tc_hint, input_ids_dev, c.stacked_feat, attention_mask_dev, position_ids_dev, B, S, st, false, nullptr))
    return e;
  const Ws& w = h->ws;
  const int d = c.hidden_size, C = c.num_labels;
  float* lg = h->wsp<float>(w.tlogits);
  GGET_REQUIRE(problem_type != GGET_PROBLEM_TOKEN_CE || h->plan.n_lin == 0, "the token-level head is the Linear `score` (no MLP head)");
  if (h->plan.n_lin > 0) {
    const Plan& pl = h->plan;
    if (int e = k_pool_rows(h->wsp<bf16_t>(w.hidden), h->wsp<int32_t>(w.pool_row), h->wsp<bf16_t>(w.pooled_h), B, d, st)) return e;
    for (int i = 0; i < pl.n_lin; ++i)
      if (int e = k_head_linear_fwd(h->wsp<bf16_t>(w.head_x[i]), h->wsp<bf16_t>(w.head_a[i]), h->P + pl.head_w[i],
                                    c.score_bias ? h->P + pl.head_b[i] : nullptr, h->wsp<bf16_t>(w.head_x[i + 1]),
                                    i + 1 == pl.n_lin ? lg : nullptr, B, pl.head_dim[i], pl.head_dim[i + 1], i, h->head_drop(), st))
        return e;
  } else if (problem_type == GGET_PROBLEM_TOKEN_CE) {
    // token-level task (loss_type = "token_ce", modeling_finetune.py:162-164, :198-202): `score` on EVERY row; the hidden state
    // handed back is still the last token's (:291-296)
    if (int e = k_tok_score_fwd(h->wsp<bf16_t>(w.hidden), h->P + h->plan.score, c.score_bias ? h->P + h->plan.sbias : nullptr, lg, h->T, C,
                                d, st))
      return e;
    if (int e = k_pool_rows(h->wsp<bf16_t>(w.hidden), h->wsp<int32_t>(w.pool_row), h->wsp<bf16_t>(w.pooled_h), B, d, st)) return e;
  } else if (int e = k_score_fwd(h->wsp<bf16_t>(w.hidden), h->wsp<int32_t>(w.pool_row), h->P + h->plan.score,
                                 c.score_bias ? h->P + h->plan.sbias : nullptr, lg, h->wsp<bf16_t>(w.pooled_h), B, C, d, st))
    return e;
  const int rows = problem_type == GGET_PROBLEM_TOKEN_CE ? h->T : B;   // rows of the logits
  const bool tok_vl = problem_type == GGET_PROBLEM_TOKEN_CE && h->varlen;   // token-level logits of compact rows: back to [B,S,C] order
  if (task_logits_dev && tok_vl) {
    GGET_HIP_CHECK(hipMemsetAsync(task_logits_dev, 0, (size_t)B * S * C * 4, st));      // (padded positions: zeros)
    if (int e = k_scatter_rows_map_f32(lg, h->wsp<int32_t>(w.vl_c2p), task_logits_dev, h->T, C, B * S, st)) return e;
  } else if (task_logits_dev) GGET_HIP_CHECK(hipMemcpyAsync(task_logits_dev, lg, (size_t)rows * C * 4, hipMemcpyDeviceToDevice, st));
  if (task_hidden_dev)
    GGET_HIP_CHECK(hipMemcpyAsync(task_hidden_dev, h->wsp<bf16_t>(w.pooled_h), (size_t)B * d * 2, hipMemcpyDeviceToDevice, st));
  h->have_labels = task_labels_dev != nullptr;
  h->problem = problem_type;
  if (task_labels_dev) {
    GGET_REQUIRE(loss_dev != nullptr, "loss_dev is required when task labels are given");
    if (problem_type == GGET_PROBLEM_AUC) {
      GGET_REQUIRE(C >= 2, "the AUC loss reads logits[:, 1] - logits[:, 0]");
      GGET_REQUIRE((long)B * h->auc_num_neg <= 8192, "AUC loss: positives x num_neg is limited to 8192 pairs");
      if (int e = k_auc_loss(lg, (const int64_t*)task_labels_dev, B, C, h->auc_num_neg, h->auc_seed, loss_dev,
                             h->wsp<float>(w.tdlogits), h->wsp<int32_t>(w.auc_lists), st))
        return e;
    } else if (problem_type == GGET_PROBLEM_TOKEN_CE) {
      if (int e = k_tok_ce(lg, (const int64_t*)task_labels_dev, h->wsp<float>(w.tdlogits), h->wsp<float>(w.tok_stat), loss_dev, rows, C, st,
                           tok_vl ? h->wsp<int32_t>(w.vl_c2p) : nullptr, B * S))
        return e;
    } else if (int e = k_task_loss(lg, task_labels_dev, sample_wgt_dev, problem_type, B, C, loss_dev, h->wsp<float>(w.tdlogits), st))
      return e;
    if (h->varlen && h->tc_from_caller)
      if (int e = k_poison_loss(h->wsp<int32_t>(w.vl_status), loss_dev, st)) return e;
  }
  h->fwd_valid = true;
  return 0;
}

// ================================================================================================
// backward
// ================================================================================================
namespace {

int convert_bucket(gget_engine* h, int bucket, hipStream_t st) {
  if (h->defer_convert) return 0;   // monolithic backward: one conversion launch over all buckets at the end
  const auto& bs = h->bucket_segs[bucket];
  return k_convert_segments(h->wsp<float>(h->ws.scratch32), h->G,
                            reinterpret_cast<const GgetSegment*>(h->W + h->ws.segs) + bs.first, bs.second, st);
}

int layer_backward(gget_engine* h, int i, hipStream_t st) {
  const gget_config_t& c = h->cfg;
  const int T = h->T, d = c.hidden_size, ff = c.intermediate_size, H = c.num_heads;
  const LayerOff& lo = h->plan.layers[i];
  const LayerWs& lw = h->ws.lw[i];
  const Ws& w = h->ws;
  float* s32 = h->wsp<float>(w.scratch32);
  bf16_t* bufs[3] = {h->wsp<bf16_t>(w.dxa), h->wsp<bf16_t>(w.dxb), h->wsp<bf16_t>(w.dxc)};
  bf16_t* dx_out = h->dx_cur;
  int oi = 0;
  while (oi < 3 && bufs[oi] != dx_out) ++oi;
  GGET_REQUIRE(oi < 3, "backward stages called out of order");
  bf16_t* dx_mid = bufs[(oi + 1) % 3];
  bf16_t* dx_in = bufs[(oi + 2) % 3];
  bf16_t* x_in = h->wsp<bf16_t>(w.xres[i]);
  bf16_t* xmid = h->wsp<bf16_t>(lw.xmid);
  bf16_t* dxn = h->wsp<bf16_t>(w.dxn);
  bf16_t* dqkv = h->wsp<bf16_t>(w.dqkv);
  bf16_t* dattn = h->wsp<bf16_t>(w.dattn);
  bf16_t* dgu = h->wsp<bf16_t>(w.dgu);
  bf16_t* dh = h->wsp<bf16_t>(w.dh);
  const bf16_t* dy_down = dx_out;  // gradient of the down_proj output
  const bf16_t* dy_o = nullptr;    // gradient of the o_proj output
  // LayerScale / DropPath backward: d/8 <= 256 chunks per row (d <= 2048 is checked at create), 512 blocks
  const int ls_rl = 256 / (d / 8);
  dim3 lsgrid((unsigned)std::min(512, (T + ls_rl - 1) / ls_rl));
  const size_t ls_lds_bytes = (size_t)ls_rl * d * sizeof(float);
  const bool fuse_ls = h->plan.has_res && ls_norm_fused();
  if (h->plan.has_res) {
    // x_out = xmid + keep_b * lam2 * mraw  =>  d mraw = keep_b * lam2 * dx_out, dlam2 += sum keep_b * dx_out * mraw
    // (already done by the RMSNorm backward that produced dx_out when the two are fused - see the end of this function)
    bf16_t* dsc = h->wsp<bf16_t>(w.dscaled);
    if (!h->ls2_done)
      hipLaunchKernelGGL(ls_bwd_kernel, lsgrid, dim3(256), ls_lds_bytes, st, dx_out, h->wsp<bf16_t>(lw.mraw),
                         h->plan.has_ls ? h->P + lo.lam2 : nullptr, dsc, h->plan.has_ls ? s32 + lo.lam2_32 : nullptr, T, d,
                         h->path_drop(i, 1), kAccumCopies, align_up((uint64_t)d, 128), h->mlp_drop(i));
    h->ls2_done = false;
    dy_down = dsc;   // stays alive for the grouped wgrad at the end of the layer (the o_proj branch has its own buffer)
  }
  // MLP: dh = dy_down W_down ; dgu = geglu'(dh) ; dxn2 = dgu W_gu
  if (h->mlp_drop_p > 0.f) {   // mlp_act_dropout sits between the gated product and down_proj: mask dh before the GEGLU backward
    if (int e = gemm_nn(dy_down, h->P + lo.wdown, dh, T, ff, d, d, ff, ff, nullptr, st)) return e;
    if (int e = k_elem_dropout(dh, T, ff, GGET_DROP_STREAM_MLP_ACT, h->mlp_drop(i), st)) return e;
    if (int e = k_geglu_bwd(h->wsp<bf16_t>(lw.gu), dh, dgu, T, ff, st)) return e;
  } else if (int e = down_dgrad_geglu(dy_down, h->P + lo.wdown, h->wsp<bf16_t>(lw.gu), dgu, dh, T, d, ff, st)) return e;
  if (int e = gemm_nn(dgu, h->P + lo.wgu, dxn, T, d, 2 * ff, 2 * ff, d, d, nullptr, st)) return e;
  bool front_fused = false;
  // (g_gemm_lds_headroom >= 2: a collective's kernel shares the chip with the backward - data-parallel runs.  The per-sample backward kernel
  //  fills 151 of the 160 KiB of a CU's LDS with ONE workgroup per sample: a foreign workgroup on a CU would push the sample's workgroup into
  //  a second round, the mechanism behind the GEMM launch menu's headroom rule.  The three-launch form runs then; the
  //  forward - which never overlaps a collective - stays fused.)
  // (Both rules are opt-in since round 5 - GGET_DP_LDS_HEADROOM, GGET_DP_RESERVE_CUS; a data-parallel rank keeps the single-GPU menu by
  //  default, this kernel included: tools/dp_standin.py.  With CUs RESERVED for the collective - g_gemm_cu_reserve - the collective's
  //  workgroups have their own CUs: the kernel may run whenever its one-workgroup-per-sample grid fits the CUs that are left.)
  const bool ao_bwd_ok = g_gemm_cu_reserve > 0 ? h->B <= gget_gemm_num_cu() : g_gemm_lds_headroom < 2;
  if (!h->plan.has_res && !h->klo() && h->wo_packed && ao_bwd_ok) {
    // S <= 32, or S <= 64 on the var-len layout (every sample by its own row count): RMSNorm backward of post_attention_layernorm, the o
    // projection's dgrad and the attention backward of a sample in ONE workgroup (attention.hip: attn_oproj_bwd_kernel); dattn is never
    // materialised, except for the rows of 33 .. 64-row samples (their attention backward is a second, sparse launch)
    int taken = 0;
    if (int e = k_attn_oproj_bwd(dxn, xmid, h->P + lo.ln2, h->wsp<float>(lw.rstd2), dx_out, dx_mid, s32 + lo.ln2_32, kAccumCopies,
                                 align_up((uint64_t)d, 128), h->wsp<bf16_t>(w.wot_pack) + (size_t)i * d * d, h->wsp<bf16_t>(lw.qkv),
                                 h->wsp<float>(lw.lse), h->wsp<int32_t>(w.key_len), h->row_base(), dqkv, h->B, h->S, H, c.causal, h->cos_cur,
                                 h->sin_cur, h->pos_cur, h->attn_drop_p, h->attn_drop_seed + 0x9E37u * (unsigned)i, T, st, &taken,
                                 dattn, h->long_list()))     // (dattn: only the rows of 33 .. 64-row samples are written, for their own launch)
      return e;
    front_fused = taken != 0;
  }
  if (front_fused) {
  } else if (fuse_ls) {
    if (int e = rmsnorm_bwd_ls(dxn, xmid, h->P + lo.ln2, h->wsp<float>(lw.rstd2), dx_out, dx_mid, s32 + lo.ln2_32, h->wsp<bf16_t>(lw.araw),
                               h->plan.has_ls ? h->P + lo.lam1 : nullptr, h->wsp<bf16_t>(w.dscaled2),
                               h->plan.has_ls ? s32 + lo.lam1_32 : nullptr, T, d, h->path_drop(i, 0), ElemDropArg{0, 1.f, 0}, st))
      return e;
  } else if (int e = k_rmsnorm_bwd(dxn, xmid, h->P + lo.ln2, h->wsp<float>(lw.rstd2), dx_out, dx_mid, s32 + lo.ln2_32, T, d, st, kAccumCopies, align_up((uint64_t)d, 128)))
    return e;
  dy_o = dx_mid;
  if (fuse_ls) {
    dy_o = h->wsp<bf16_t>(w.dscaled2);
  } else if (h->plan.has_res) {
    bf16_t* dsc = h->wsp<bf16_t>(w.dscaled2);
    hipLaunchKernelGGL(ls_bwd_kernel, lsgrid, dim3(256), ls_lds_bytes, st, dx_mid, h->wsp<bf16_t>(lw.araw),
                       h->plan.has_ls ? h->P + lo.lam1 : nullptr, dsc, h->plan.has_ls ? s32 + lo.lam1_32 : nullptr, T, d,
                       h->path_drop(i, 0), kAccumCopies, align_up((uint64_t)d, 128), ElemDropArg{0, 1.f, 0});
    dy_o = dsc;
  }
  // attention: dattn = dy_o W_o ; (dq,dk,dv) ; inverse RoPE ; dxn1 = dqkv W_qkv
  if (!front_fused)
  if (int e = gemm_nn(dy_o, h->P + lo.wo, dattn, T, d, d, d, d, d, nullptr, st)) return e;
  if (!front_fused)
  if (int e = k_attn_bwd(h->wsp<bf16_t>(lw.qkv), h->wsp<bf16_t>(lw.attn), dattn, h->wsp<float>(lw.lse),
                         h->wsp<int32_t>(w.key_len), dqkv, h->wsp<float>(w.delta), h->B, h->S, H, c.causal, h->cos_cur,
                         h->sin_cur, h->pos_cur, /*qk_rotated=*/1, h->attn_drop_p, h->attn_drop_seed + 0x9E37u * (unsigned)i, st,
                         h->klo(), h->khi(), h->row_base(),
                         /* the slabs were sized for ceil(max_tokens / max_batch / 256) key blocks: a call with fewer, longer rows keeps the two-kernel form */
                         (w.dq_acc && (uint64_t)(h->S + 255) / 256 <= (c.max_tokens / (c.max_batch > 0 ? c.max_batch : 1) + 255) / 256) ? h->wsp<bf16_t>(w.dq_acc) : nullptr,
                         (size_t)c.max_tokens * d, h->long_list()))
    return e;
  if (int e = gemm_nn(dqkv, h->P + lo.wqkv, dxn, T, d, 3 * d, 3 * d, d, d, nullptr, st)) return e;
  // (fused with the LayerScale backward of the layer below, this RMSNorm backward writes w.dscaled - which this layer's down_proj weight
  //  gradient still reads - so it runs behind the weight gradients then)
  const bool fuse_next = fuse_ls && i > 0;
  if (!fuse_next)
    if (int e = k_rmsnorm_bwd(dxn, x_in, h->P + lo.ln1, h->wsp<float>(lw.rstd1), dx_mid, dx_in, s32 + lo.ln1_32, T, d, st, kAccumCopies, align_up((uint64_t)d, 128)))
      return e;
  // weight gradients of the layer (dW = dY^T X, K = T).  All four dY / X pairs are still alive here.
  const GemmProblem wg_gu{dgu, h->wsp<bf16_t>(lw.xn2), h->G + lo.wgu, nullptr, 2 * ff, d, T, 2 * ff, d, d, nullptr, nullptr, 0, 0};
  const GemmProblem wg_dn{dy_down, h->wsp<bf16_t>(lw.h), h->G + lo.wdown, nullptr, d, ff, T, d, ff, ff, nullptr, nullptr, 0, 0};
  // Data-parallel runs leave CUs to their collectives (g_gemm_cu_reserve): the 256 tiles of the grouped launch below would then need a second
  // round for a few of them (twice the time).  As many problems as fit the CUs that are left stay in the one-tile-per-CU group (in the order
  // gate|up 128, down 64, q|k|v 48, o 16 tiles at d = 768); the rest - q|k|v and o, contiguous in the gradient array - take the split-K slab
  // path (128 x 128 or 256 x 128 tiles x K slices fitted to one round, fp32 slabs summed by slab_reduce).
  const long t192[4] = {(long)2 * ff * d / (192 * 192), (long)d * ff / (192 * 192), (long)3 * d * d / (192 * 192), (long)d * d / (192 * 192)};
  int n_grouped = 4;
  if (d % 192 == 0 && ff % 192 == 0 && g_gemm_cu_reserve > 0) {
    long acc_t = 0;
    n_grouped = 0;
    for (int q = 0; q < 4 && acc_t + t192[q] <= gget_gemm_num_cu(); ++q) { acc_t += t192[q]; ++n_grouped; }
    if (n_grouped < 2) n_grouped = 4;      // (nothing sensible to shed: the plain plan, whatever rounds it takes)
  }
  if (d % 192 == 0 && ff % 192 == 0) {
    // one persistent launch of 192x192 tiles over gate|up, down, q|k|v and o: (2ff*d + d*ff + 4d*d) / 192^2 tiles, which for
    // d = 768 is exactly 256 - every CU owns one tile and walks the full K, no split-K slabs, no reduce pass
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = n_grouped;
    g.p[0] = wg_gu;
    g.p[1] = wg_dn;
    g.p[2] = GemmProblem{dqkv, h->wsp<bf16_t>(lw.xn1), h->G + lo.wqkv, nullptr, 3 * d, d, T, 3 * d, d, d, nullptr, nullptr, 0, 0};
    g.p[3] = GemmProblem{dy_o, h->wsp<bf16_t>(lw.attn), h->G + lo.wo, nullptr, d, d, T, d, d, d, nullptr, nullptr, 0, 0};
    const long wg_tiles = ((long)2 * ff * d + (long)d * ff + (long)4 * d * d) / (192 * 192);
    g.sq_partials = n_grouped == 4 && h->opt_norm_from_backward && wg_tiles <= kSqTilesPerLayer ? h->wsp<float>(w.sq_tiles) + (size_t)i * kSqTilesPerLayer : nullptr;
    if (h->probe) GGET_HIP_CHECK(hipEventRecord(h->probe_event(0, 2 * i), st));
    if (int e = gget_gemm_launch(GGET_GEMM_TN, GGET_EPI_NONE, g, 1, st)) return e;
    if (h->probe) GGET_HIP_CHECK(hipEventRecord(h->probe_event(0, 2 * i + 1), st));
    if (g.sq_written) {
      if (wg_tiles < kSqTilesPerLayer)   // (slots no tile writes must read as zero)
        GGET_HIP_CHECK(hipMemsetAsync(h->wsp<float>(w.sq_tiles) + (size_t)i * kSqTilesPerLayer + wg_tiles, 0, (kSqTilesPerLayer - wg_tiles) * sizeof(float), st));
      ++h->sq_layers;
    }
    if (n_grouped < 4) {
      // the shed problems: q|k|v (when n_grouped == 2) and o, one split-K launch into fp32 slabs of 4 d^2 (q|k|v|o are contiguous in the
      // gradient array); the workspace holds kWgSplit such slabs
      float* wg = h->wsp<float>(w.wg32);
      const bool with_qkv = n_grouped == 2;
      const long slab = with_qkv ? (long)4 * d * d : (long)d * d;
      const int max_slabs = with_qkv ? kWgSplit : 4 * kWgSplit;
      const int rows_out = with_qkv ? 4 * d : d;
      const int t_big = ((rows_out + 255) / 256) * ((d + 127) / 128), t_small = ((rows_out + 127) / 128) * ((d + 127) / 128);
      int split = fit_split(t_big, max_slabs);
      if (t_big * split < 160) split = fit_split(t_small, max_slabs);           // (gemm.hip launch_shape: 128 x 128 tiles below 160 tile-slices)
      while (split > 1 && (T + 63) / 64 < 4 * split) --split;                   // (at least four 64-deep K-tiles per slice)
      GemmGroup go;
      memset(&go, 0, sizeof(go));
      if (with_qkv) {
        go.count = 2;
        go.p[0] = GemmProblem{dqkv, h->wsp<bf16_t>(lw.xn1), wg, nullptr, 3 * d, d, T, 3 * d, d, d, nullptr, nullptr, 0, 0, slab};
        go.p[1] = GemmProblem{dy_o, h->wsp<bf16_t>(lw.attn), wg + (size_t)3 * d * d, nullptr, d, d, T, d, d, d, nullptr, nullptr, 0, 0, slab};
      } else {
        go.count = 1;
        go.p[0] = GemmProblem{dy_o, h->wsp<bf16_t>(lw.attn), wg, nullptr, d, d, T, d, d, d, nullptr, nullptr, 0, 0, slab};
      }
      if (int e = gget_gemm_launch(GGET_GEMM_TN, GGET_EPI_SLAB_F32, go, split, st)) return e;
      const int ktiles = (T + 63) / 64, per = (ktiles + split - 1) / split;      // (the kernel's slicing: trailing slices may be empty)
      if (int e = k_slab_reduce(wg, slab, (ktiles + per - 1) / per, h->G + (with_qkv ? lo.wqkv : lo.wo), (size_t)slab, st)) return e;
    }
  } else {
    {
      GemmGroup g;
      memset(&g, 0, sizeof(g));
      g.count = 2;
      g.p[0] = wg_gu;
      g.p[1] = wg_dn;
      if (int e = gget_gemm_launch(GGET_GEMM_TN, GGET_EPI_NONE, g, 1, st)) return e;
    }
    // wgrad of q|k|v and o: only 72 output tiles of 256x128 with K = T, so K is cut into kWgSplit slices (216 blocks);
    // every slice writes its own fp32 slab with wide stores (fp32 atomics are ~3x slower: one TA op per 4 bytes) and a
    // small kernel sums the slabs into the bf16 gradient array (q|k|v|o are contiguous there).
    float* wg = h->wsp<float>(w.wg32);
    const long slab = (long)4 * d * d;
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = 2;
    g.p[0] = GemmProblem{dqkv, h->wsp<bf16_t>(lw.xn1), wg, nullptr, 3 * d, d, T, 3 * d, d, d, nullptr, nullptr, 0, 0, slab};
    g.p[1] = GemmProblem{dy_o, h->wsp<bf16_t>(lw.attn), wg + (size_t)3 * d * d, nullptr, d, d, T, d, d, d, nullptr, nullptr, 0, 0, slab};
    const int split = T >= kWgSplit * 1024 ? kWgSplit : 1;
    if (int e = gget_gemm_launch(GGET_GEMM_TN, GGET_EPI_SLAB_F32, g, split, st)) return e;
    if (int e = k_slab_reduce(wg, slab, split, h->G + lo.wqkv, (size_t)4 * d * d, st)) return e;
  }
  if (fuse_next) {
    const LayerOff& lp = h->plan.layers[i - 1];
    if (int e = rmsnorm_bwd_ls(dxn, x_in, h->P + lo.ln1, h->wsp<float>(lw.rstd1), dx_mid, dx_in, s32 + lo.ln1_32,
                               h->wsp<bf16_t>(h->ws.lw[i - 1].mraw), h->plan.has_ls ? h->P + lp.lam2 : nullptr, h->wsp<bf16_t>(w.dscaled),
                               h->plan.has_ls ? s32 + lp.lam2_32 : nullptr, T, d, h->path_drop(i - 1, 1), h->mlp_drop(i - 1), st))
      return e;
    h->ls2_done = true;
  }
  h->dx_cur = dx_in;
  return convert_bucket(h, h->bucket_of_layer(i), st);
}

}  // namespace

// which form of the embedding backward runs (embed_bwd below): the count-matrix product, or the sorted scatter-add
// (embedding dropout masks every (cell, channel) on its own: the count-matrix product cannot express it)
static bool embed_dense_path(int T, int d, int V, bool gated, const ElemDropArg& E) {
  static const bool sorted_only = getenv("GGET_EMBED_SORTED") != nullptr;   // A/B knob
  return !sorted_only && E.thresh == 0 && k_embed_dense_ok(V, gated) && T > 0 && ((long)V * d) % 4 == 0;
}

extern "C" int gget_backward_begin(gget_handle_t h, float loss_scale, void* stream) {
  GGET_REQUIRE(h, "null handle");
  GGET_REQUIRE(h->fwd_valid && h->have_labels, "backward needs a preceding forward with labels");
  GGET_REQUIRE(loss_scale == 1.0f, "loss scaling is not used on the bf16 path (pass 1.0)");
  h->sq_layers = 0;
  StreamKScope sk_scope(h);
  hipStream_t st = (hipStream_t)stream;
  const gget_config_t& c = h->cfg;
  const Ws& w = h->ws;
  const int d = c.hidden_size;
  float* s32 = h->wsp<float>(w.scratch32);
  // everything the backward needs cleared, in ONE launch (five runtime fill kernels of 4 - 15 us each before): the fp32 accumulators of the
  // small parameters, the gradient of the final-norm output (the head scatters the selected rows into it), ...
  GgetZeroRanges zr;
  zr.n = 0;
  zr.add(s32, (size_t)h->plan.n_scratch32 * 4);
  bf16_t* dhid = h->wsp<bf16_t>(w.dxb);  // gradient w.r.t. the final-norm output
  zr.add(dhid, (size_t)h->T * d * 2);
  if (h->varlen && h->T > h->tc)   // var-len layout: the <= 63 pad rows behind the last sample belong to no attention problem - their
    // q|k|v gradient rows are written by nobody and must read as zeros in the weight gradients (K = T) and the dgrad below them
    zr.add(h->wsp<bf16_t>(w.dqkv) + (size_t)h->tc * 3 * d, (size_t)(h->T - h->tc) * 3 * d * 2);
  int T = h->TP;   // the head works in the padded token index space (capacities only: the counts are on the device)
  if (c.kind == GGET_KIND_PRETRAIN) {
    // ... the scatter target of the lm_head dgrad and the split-K slabs of its weight gradient (both only ever written by the launches below)
    if (h->plan.has_ntp && head_scatter_fused())
      zr.add(h->wsp<bf16_t>(w.dP), (size_t)(h->varlen ? h->T : T) * c.next_n_token * d * 2);
    zr.add(h->wsp<float>(w.lm_slab), (size_t)kLmSplit * c.vocab_size * d * sizeof(float));
  }
  // ... and the bf16 count matrix of the embedding gradient (embed_bwd at the end of the backward: its own fill launch before)
  h->emb_cnt_cleared = false;
  if (h->ws.emb_cnt && embed_dense_path(h->T, d, c.vocab_size, h->plan.has_gate, h->embed_drop()) && zr.n < kZeroRanges) {
    zr.add(h->wsp<unsigned char>(h->ws.emb_cnt), (size_t)h->T * align_up((uint64_t)c.vocab_size, 64) * 2);
    h->emb_cnt_cleared = true;
  }
  if (int e = k_zero_ranges(zr, st)) return e;
  if (c.kind == GGET_KIND_PRETRAIN) {
    const int n = c.next_n_token, V = c.vocab_size, Vp = (int)align_up(V, 64);
    int32_t* counts = h->wsp<int32_t>(w.counts);
    bf16_t* dlog = h->wsp<bf16_t>(w.dlogits);
    // lm_head: dHl = dlogits W_lm ; dW_lm = dlogits^T Hl (split-K, fp32 slabs: only a few output tiles)
    // With n_token_proj the rows of dHl are scattered into dP (row sel_src[i] of the [M n, d] view) - fused into the GEMM's epilogue
    // (GemmProblem::c_rows), so dHl is never materialised and the separate scatter launch disappears; dP is cleared first.
    const bool fuse_sc = h->plan.has_ntp && head_scatter_fused();
    if (fuse_sc) {      // (dP was cleared by the launch above)
      if (int e = gemm_nn(dlog, h->P + h->plan.lm, h->wsp<bf16_t>(w.dP), T * n, d, Vp, Vp, d, d, counts + 1, st, h->wsp<int32_t>(w.sel_src)))
        return e;
    } else if (int e = gemm_nn(dlog, h->P + h->plan.lm, h->wsp<bf16_t>(w.dHl), T * n, d, Vp, Vp, d, d, counts + 1, st)) return e;   // K = Vp: zero pads on both sides
    {
      // K = Lm (device-side count, ~5e4) over only 6x6 output tiles: kLmSplit K-slices, one fp32 slab each, then a sum.
      // Slices that fall beyond a short Lm write nothing, so the slabs are cleared first.
      float* slabs = h->wsp<float>(w.lm_slab);
      const long slab = (long)V * d;
      GemmGroup g;     // (the slabs were cleared by the launch at the top)
      memset(&g, 0, sizeof(g));
      g.count = 1;
      g.p[0] = GemmProblem{dlog, h->wsp<bf16_t>(w.Hl), slabs, nullptr, V, d, T * n, Vp, d, d, nullptr, counts + 1, 0, 0, slab};
      // (256 x 128 tiles from 160 tile-slices on, 128 x 128 below - gemm.hip launch_shape)
      int lm_split = fit_split(((V + 255) / 256) * ((d + 127) / 128), kLmSplit);
      if (((V + 255) / 256) * ((d + 127) / 128) * lm_split < 160) lm_split = fit_split(((V + 127) / 128) * ((d + 127) / 128), kLmSplit);
      if (int e = gget_gemm_launch(GGET_GEMM_TN, GGET_EPI_SLAB_F32, g, lm_split, st)) return e;
      if (int e = k_slab_reduce(slabs, slab, lm_split, h->G + h->plan.lm, (size_t)V * d, st)) return e;
    }
    if (h->plan.has_ntp) {
      bf16_t* dP = h->wsp<bf16_t>(w.dP);
      // (rows M.. are read as zeros by the K-tail of the weight gradient; M <= real tokens <= h->T on the var-len layout)
      if (!fuse_sc) {
        GGET_HIP_CHECK(hipMemsetAsync(dP, 0, (size_t)(h->varlen ? h->T : T) * n * d * 2, st));
        if (int e = k_gather_rows(h->wsp<bf16_t>(w.dHl), h->wsp<int32_t>(w.sel_src), counts + 1, dP, T * n, d, 1, st)) return e;
      }
      // dHm = dP W_ntp, its rows scattered into the gradient of the final-norm output (row row_idx[i]) by the same fused epilogue
      if (h->head_sorted_fwd) {
        // slot-sorted form: one input-gradient row per labelled cell, dXs[p] = dP[cell(p)] W_f(p) (rows gathered from dP, the weight
        // block per row tile), then every selected token sums its cells (fp32, one rounding) into its row of d hidden
        bf16_t* dxs = h->wsp<bf16_t>(w.Pp);
        GemmGroup g;
        memset(&g, 0, sizeof(g));
        g.count = 1;
        GemmProblem& p = g.p[0];
        p.A = dP; p.B = h->P + h->plan.ntp; p.C = dxs;
        p.M = T * n + n * 256; p.N = d; p.K = d; p.lda = d; p.ldb = d; p.ldc = d;
        p.m_dev = h->wsp<int32_t>(w.ss_total);
        p.a_rows = h->wsp<int32_t>(w.ss_cell); p.b_tile_off = h->wsp<int32_t>(w.ss_tile);
        p.b_tile_rows = head_tile_rows(d);
        if (int e = gget_gemm_launch(GGET_GEMM_NN, GGET_EPI_NONE, g, 1, st)) return e;
        if (int e = k_head_cell_sum(dxs, h->wsp<int32_t>(w.ss_pos), h->wsp<int32_t>(w.cnt), h->wsp<int32_t>(w.l_off),
                                    h->varlen ? h->wsp<int32_t>(w.vl_pad2c) : nullptr, dhid, T, d, h->T > h->tc ? h->tc : 0, st))
          return e;
      } else if (head_scatter_fused()) {
        if (int e = gemm_nn(dP, h->P + h->plan.ntp, dhid, T, d, n * d, n * d, d, d, counts, st, h->wsp<int32_t>(w.row_idx))) return e;
      } else if (int e = gemm_nn(dP, h->P + h->plan.ntp, h->wsp<bf16_t>(w.dHm), T, d, n * d, n * d, d, d, counts, st)) return e;
      if (int e = gget_gemm_single(GGET_GEMM_TN, GGET_EPI_NONE, dP, h->wsp<bf16_t>(w.Hm), h->G + h->plan.ntp, nullptr, n * d, d,
                                   T, n * d, d, d, nullptr, counts, 1, st, /*k_pad_zero=*/true))   // dP is cleared above, Hm is finite
        return e;
    }
    if (!(h->plan.has_ntp && (head_scatter_fused() || h->head_sorted_fwd)))
      if (int e = k_gather_rows(h->wsp<bf16_t>(w.dHm), h->wsp<int32_t>(w.row_idx), counts, dhid, T, d, 1, st)) return e;
  } else if (h->plan.n_lin > 0) {
    const Plan& pl = h->plan;
    const float* dy = h->wsp<float>(w.tdlogits);
    for (int i = pl.n_lin - 1; i >= 0; --i) {
      float* dx = h->wsp<float>(w.head_d[i & 1]);
      if (int e = k_head_linear_bwd(dy, h->wsp<bf16_t>(w.head_x[i]), h->wsp<bf16_t>(w.head_a[i]), h->P + pl.head_w[i],
                                    s32 + pl.head_w32[i], c.score_bias ? s32 + pl.head_b32[i] : nullptr, dx, h->B, pl.head_dim[i],
                                    pl.head_dim[i + 1], i, h->head_drop(), st))
        return e;
      dy = dx;
    }
    if (int e = k_scatter_rows_f32(dy, h->wsp<int32_t>(w.pool_row), dhid, h->B, d, st)) return e;
  } else if (h->problem == GGET_PROBLEM_TOKEN_CE) {
    if (int e = k_tok_score_bwd(h->wsp<float>(w.tdlogits), h->wsp<float>(w.tok_stat), h->wsp<bf16_t>(w.hidden), h->P + h->plan.score,
                                s32 + h->plan.score32, c.score_bias ? s32 + h->plan.sbias32 : nullptr, dhid, h->T, c.num_labels, d, st))   // (h->T: the rows of the token-major buffers)
      return e;
  } else {
    if (int e = k_score_bwd(h->wsp<float>(w.tdlogits), h->wsp<bf16_t>(w.hidden), h->wsp<int32_t>(w.pool_row),
                            h->P + h->plan.score, s32 + h->plan.score32, c.score_bias ? s32 + h->plan.sbias32 : nullptr, dhid,
                            h->B, c.num_labels, d, st))
      return e;
  }
  bf16_t* dx = h->wsp<bf16_t>(w.dxa);
  h->ls2_done = false;
  T = h->T;   // from here on: rows of the token-major buffers
  if (h->plan.has_res && ls_norm_fused()) {     // ... fused with the LayerScale backward of the last layer (layer_backward)
    const int li = c.num_layers - 1;
    const LayerOff& lp = h->plan.layers[li];
    if (int e = rmsnorm_bwd_ls(dhid, h->wsp<bf16_t>(w.xres[c.num_layers]), h->P + h->plan.normf, h->wsp<float>(w.rstd_f), nullptr, dx,
                               s32 + h->plan.normf32, h->wsp<bf16_t>(h->ws.lw[li].mraw), h->plan.has_ls ? h->P + lp.lam2 : nullptr,
                               h->wsp<bf16_t>(w.dscaled), h->plan.has_ls ? s32 + lp.lam2_32 : nullptr, T, d, h->path_drop(li, 1),
                               h->mlp_drop(li), st))
      return e;
    h->ls2_done = true;
  } else if (int e = k_rmsnorm_bwd(dhid, h->wsp<bf16_t>(w.xres[c.num_layers]), h->P + h->plan.normf, h->wsp<float>(w.rstd_f), nullptr,
                                   dx, s32 + h->plan.normf32, T, d, st, kAccumCopies, align_up((uint64_t)d, 128)))
    return e;
  h->dx_cur = dx;
  return convert_bucket(h, 0, st);
}

extern "C" int gget_backward_layer(gget_handle_t h, int layer, void* stream) {
  GGET_REQUIRE(h && h->fwd_valid && h->dx_cur, "backward_layer before backward_begin");
  GGET_REQUIRE(layer >= 0 && layer < h->cfg.num_layers, "layer %d out of range", layer);
  StreamKScope sk_scope(h);
  return layer_backward(h, layer, (hipStream_t)stream);
}

// Embedding backward (SURVEY row A1 backward).  Small un-gated vocabularies: dE = C^T dX with the bf16 count matrix C
// (k_embed_count), one split-K GEMM with fp32 atomics into the accumulator - the sorted scatter-add below spends its time in
// same-address atomics when half of the cells hold the <mask> id.  Otherwise: counting sort by id + segmented sums.
int embed_bwd(const int64_t* ids, const void* dx, const void* emb, const void* gate, float* demb, float* dgate, int T, int F,
              int ldF, int d, int V, int pad_id, int32_t* sort_ws, void* cnt_ws, void* slab_ws, hipStream_t st, ElemDropArg E,
              bool cnt_cleared = false) {
  if (cnt_ws && slab_ws && embed_dense_path(T, d, V, gate != nullptr, E)) {
    const int ldc = (int)align_up((uint64_t)V, 64);
    if (!cnt_cleared) GGET_HIP_CHECK(hipMemsetAsync(cnt_ws, 0, (size_t)T * ldc * 2, st));
    if (int e = k_embed_count(ids, cnt_ws, T, F, ldF, ldc, pad_id, st)) return e;
    // 6 x 6 output tiles at V = 756, d = 768: K = T is cut into <= kEmbDenseSplit slices, one fp32 slab each (every slice is
    // non-empty: nslab is recomputed from the 64-row K-tiles), then one pass sums the slabs into the accumulator
    const int ktiles = (T + 63) / 64;
    int split = ktiles / 4 < 1 ? 1 : (ktiles / 4 > kEmbDenseSplit ? kEmbDenseSplit : ktiles / 4);
    {   // one round of the CUs (see fit_split): 36 tiles of 128 x 128 x 8 slices were 288 blocks
      const int t_big = ((V + 255) / 256) * ((d + 127) / 128), t_small = ((V + 127) / 128) * ((d + 127) / 128);
      split = t_big * split >= 160 ? fit_split(t_big, split) : fit_split(t_small, split);
    }
    const int per = (ktiles + split - 1) / split;
    const int nslab = (ktiles + per - 1) / per;
    float* slabs = static_cast<float*>(slab_ws);
    if (int e = gget_gemm_single(GGET_GEMM_TN, GGET_EPI_SLAB_F32, cnt_ws, dx, slabs, nullptr, V, d, T, ldc, d, d, nullptr, nullptr,
                                 split, st))
      return e;
    return k_slab_reduce(slabs, (long)V * d, nslab, demb, (size_t)V * d, st, /*f32_out=*/true);
  }
  return k_embed_bwd(ids, dx, emb, gate, demb, dgate, T, F, ldF, d, V, pad_id, sort_ws, st, E);
}

extern "C" int gget_backward_end(gget_handle_t h, void* stream) {
  GGET_REQUIRE(h && h->fwd_valid && h->dx_cur, "backward_end before backward_begin");
  StreamKScope sk_scope(h);
  hipStream_t st = (hipStream_t)stream;
  const gget_config_t& c = h->cfg;
  float* s32 = h->wsp<float>(h->ws.scratch32);
  if (h->raw_used) {
    // backward of the raw-embedding branch (it received the same gradient as the token embeddings): dW_proj = dx^T xn,
    // d xn = dx W_proj (dropout mask again), RMSNorm backward, and the labelled rows' gradient summed into emb_mask_token
    const Ws& w = h->ws;
    const int T = h->T, d = c.hidden_size, e_ = c.embed_dim;
    if (int e = gget_gemm_single(GGET_GEMM_TN, GGET_EPI_NONE, h->dx_cur, h->wsp<bf16_t>(w.raw_xn), h->G + h->plan.raw_proj, nullptr, d, e_, T,
                                 d, e_, e_, nullptr, nullptr, 1, st))
      return e;
    if (int e = gemm_nn(h->dx_cur, h->P + h->plan.raw_proj, h->wsp<bf16_t>(w.raw_dxn), T, e_, d, d, e_, e_, nullptr, st)) return e;
    if (int e = k_elem_dropout(h->wsp<bf16_t>(w.raw_dxn), T, e_, GGET_DROP_STREAM_RAW, h->embed_drop(), st)) return e;
    if (int e = k_rmsnorm_bwd(h->wsp<bf16_t>(w.raw_dxn), h->wsp<bf16_t>(w.raw_x), h->P + h->plan.raw_ln, h->wsp<float>(w.raw_rstd), nullptr,
                              h->wsp<bf16_t>(w.raw_dx), s32 + h->plan.raw_ln32, T, e_, st, kAccumCopies, align_up((uint64_t)e_, 128)))
      return e;
    if (c.kind == GGET_KIND_PRETRAIN)
      if (int e = k_raw_tok_grad(h->wsp<bf16_t>(w.raw_dx), h->wsp<int32_t>(w.raw_flag), s32 + h->plan.raw_tok32, T, e_, st)) return e;
  }
  if (h->stack_long)   // d(e * ratio) / de = ratio
    if (int e = k_embed_long_ratio(h->ids, h->dx_cur, h->T, c.stacked_feat, c.stacked_feat, c.hidden_size, st)) return e;
  if (int e = embed_bwd(h->ids, h->dx_cur, h->P + h->plan.emb, h->plan.has_gate ? h->P + h->plan.gate : nullptr,
                        s32 + h->plan.emb32, h->plan.has_gate ? s32 + h->plan.gate32 : nullptr, h->T, c.stacked_feat,
                        c.stacked_feat, c.hidden_size, c.vocab_size, c.pad_token_id, h->wsp<int32_t>(h->ws.emb_sort),
                        k_embed_dense_ok(c.vocab_size, h->plan.has_gate) ? h->wsp<unsigned char>(h->ws.emb_cnt) : nullptr,
                        k_embed_dense_ok(c.vocab_size, h->plan.has_gate) ? h->wsp<unsigned char>(h->ws.emb_slab) : nullptr, st,
                        h->embed_drop(), h->emb_cnt_cleared))
    return e;
  h->emb_cnt_cleared = false;
  h->dx_cur = nullptr;
  return convert_bucket(h, c.num_layers + 1, st);
}

extern "C" int gget_backward(gget_handle_t h, float loss_scale, void* stream) {
  GGET_REQUIRE(h, "null handle");
  // no bucket has to be final before the end: the small-parameter gradients (fp32 accumulators -> bf16) of all L+2
  // buckets are converted by ONE launch instead of L+2
  h->defer_convert = true;
  int rc = gget_backward_begin(h, loss_scale, stream);
  for (int i = h->cfg.num_layers - 1; i >= 0 && rc == 0; --i) rc = gget_backward_layer(h, i, stream);
  if (rc == 0) rc = gget_backward_end(h, stream);
  h->defer_convert = false;
  if (rc) return rc;
  int nseg = 0;
  for (const auto& bs : h->bucket_segs) nseg += bs.second;
  return k_convert_segments(h->wsp<float>(h->ws.scratch32), h->G, reinterpret_cast<const GgetSegment*>(h->W + h->ws.segs), nseg,
                            (hipStream_t)stream);
}

extern "C" int gget_adamw_step(gget_handle_t h, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float max_grad_norm, float grad_scale, int step, float* gnorm_dev, void* stream) {
  GGET_REQUIRE(h && h->master && h->am && h->av, "adamw needs master/m/v arenas");
  GGET_REQUIRE(step >= 1, "step is 1-based");
  hipStream_t st = (hipStream_t)stream;
  h->wo_packed = false;      // (see gget_sync_params; the next forward rebuilds the copies)
  float* sq = h->wsp<float>(h->ws.sqnorm);
  const bool need_norm = max_grad_norm > 0.f || gnorm_dev != nullptr || h->opt_skip_nonfinite;
  if (need_norm) {
    // the shortcut holds only while the gradient array is exactly what the last backward wrote: the caller promised that
    // (GGET_OPT_NORM_FROM_BACKWARD), grad_scale != 1 means an exchange happened anyway, and every layer must have left its partials
    if (h->opt_norm_from_backward && grad_scale == 1.0f && h->sq_layers == h->cfg.num_layers && h->n_sq_chunks >= 0) {
      if (int e = k_grad_sqnorm_chunks(h->G, h->wsp<GgetSqChunk>(h->ws.sq_chunks), h->n_sq_chunks, h->wsp<float>(h->ws.sq_tiles),
                                       h->cfg.num_layers * kSqTilesPerLayer, sq, st))
        return e;
    } else if (int e = k_grad_sqnorm(h->G, h->plan.n_params, sq, st)) return e;
  }
  return k_adamw(h->master, h->am, h->av, h->G, h->P, h->plan.n_params, lr, beta1, beta2, eps, weight_decay, step,
                 max_grad_norm, grad_scale, need_norm ? sq : nullptr, gnorm_dev, st, h->opt_skip_nonfinite);
}

extern "C" int gget_head_counts(gget_handle_t h, int32_t counts[2], void* stream) {
  GGET_REQUIRE(h && counts, "null argument");
  GGET_REQUIRE(h->cfg.kind == GGET_KIND_PRETRAIN, "head counts exist only for the pre-train head");
  GGET_HIP_CHECK(hipMemcpyAsync(counts, h->wsp<int32_t>(h->ws.counts), 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  GGET_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" int gget_head_logits(gget_handle_t h, const void** logits_dev, int32_t* ld) {
  GGET_REQUIRE(h && logits_dev && ld, "null argument");
  GGET_REQUIRE(h->cfg.kind == GGET_KIND_PRETRAIN, "head logits exist only for the pre-train head");
  *logits_dev = h->wsp<bf16_t>(h->ws.logits);
  *ld = (int32_t)align_up(h->cfg.vocab_size, 64);
  return 0;
}

extern "C" int gget_hidden_states(gget_handle_t h, const void** hidden_dev) {
  GGET_REQUIRE(h && hidden_dev, "null argument");
  GGET_REQUIRE(!h->varlen, "hidden_states: the last forward ran on the var-len token layout (rows are compacted): run that forward on the padded grid - no token count at the C ABI, num_tokens=None with a host-side mask or GGET_VARLEN=0 through the model classes, or ask the model for output_hidden_states=True");
  *hidden_dev = h->wsp<bf16_t>(h->ws.hidden);
  return 0;
}

// residual stream entering decoder layer `layer` (layer == num_layers: leaving the last one) of the last forward - the quantity
// hf LlamaModel's `output_hidden_states=True` returns as hidden_states[layer] (modeling_llama.py :401-414).  Padded layout only.
extern "C" int gget_layer_hidden_states(gget_handle_t h, int layer, const void** hidden_dev) {
  GGET_REQUIRE(h && hidden_dev, "null argument");
  GGET_REQUIRE(layer >= 0 && layer <= h->cfg.num_layers, "layer_hidden_states: layer %d out of range", layer);
  GGET_REQUIRE(!h->varlen, "layer_hidden_states: the last forward ran on the var-len token layout (rows are compacted): run that forward on the padded grid - no token count at the C ABI, num_tokens=None with a host-side mask or GGET_VARLEN=0 through the model classes, or ask the model for output_hidden_states=True");
  *hidden_dev = h->wsp<bf16_t>(h->ws.xres[layer]);
  return 0;
}

// The same two quantities in the reference's [B,S,d] layout WHATEVER layout the forward ran on: layer = -1 the final-normed hidden states
// (gget_hidden_states), 0 .. num_layers the residual stream entering that layer (gget_layer_hidden_states).  After a var-len forward the
// compact rows are spread over the grid through the padded -> compact map and the positions behind a sample's tokens read as zero
// (the padded layout computes values there that nothing downstream uses).  out_dev: bf16 [B * S * d] of the last forward's B, S.
extern "C" int gget_hidden_states_grid(gget_handle_t h, int layer, void* out_dev, void* stream) {
  GGET_REQUIRE(h && out_dev, "null argument");
  GGET_REQUIRE(layer >= -1 && layer <= h->cfg.num_layers, "hidden_states_grid: layer %d out of range", layer);
  GGET_REQUIRE(h->B > 0 && h->S > 0, "hidden_states_grid: no forward has run");
  const bf16_t* src = layer < 0 ? h->wsp<bf16_t>(h->ws.hidden) : h->wsp<bf16_t>(h->ws.xres[layer]);
  const int d = h->cfg.hidden_size;
  const long n_pos = (long)h->B * h->S;
  if (!h->varlen) {
    GGET_HIP_CHECK(hipMemcpyAsync(out_dev, src, (size_t)n_pos * d * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
  }
  return k_rows_to_grid(src, h->wsp<int32_t>(h->ws.vl_pad2c), out_dev, n_pos, d, (hipStream_t)stream);
}

// ================================================================================================
// operator-level entry points
// ================================================================================================
// Stand-in for a collective's kernel in co-residency measurements (tools/coresidency.py): `blocks` workgroups of 256 threads
// with `lds_bytes` of LDS each stay resident for ~`microseconds`, streaming a little memory (a ring step's copy / reduce
// traffic) while they wait.  No reference counterpart: a measurement aid like gget_debug_set.
__global__ void __launch_bounds__(256) occupy_kernel(float* buf, size_t n, long long ticks) {
  extern __shared__ float occ_lds[];
  const long long t0 = wall_clock64();
  float acc = 0.f;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) % n;
  while (wall_clock64() - t0 < ticks) {
    acc += buf[i];
    i = (i + 65536) % n;
    if (threadIdx.x == 0) occ_lds[0] = acc;
  }
  if (acc == 1.2345e-30f) buf[0] = acc;
}
// ... with the REGISTER footprint of RCCL's kernel (gget_debug_set(16, 1)): rcclGenericKernel allocates 261 - 280 registers per lane (its
// gfx950 code object) - one wave of it per SIMD leaves no room for the two waves of a GEMM workgroup, whatever LDS is
// free.  The clobbers make the compiler allocate 256 vector + 8 accumulation registers; the loop is the same.
__global__ void __launch_bounds__(256) occupy_fat_kernel(float* buf, size_t n, long long ticks) {
  extern __shared__ float occ_lds[];
  asm volatile("" ::: "v255", "a7");
  const long long t0 = wall_clock64();
  float acc = 0.f;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) % n;
  while (wall_clock64() - t0 < ticks) {
    acc += buf[i];
    i = (i + 65536) % n;
    if (threadIdx.x == 0) occ_lds[0] = acc;
  }
  if (acc == 1.2345e-30f) buf[0] = acc;
}
int g_occupy_fat = 0;
extern "C" int gget_debug_occupy(void* scratch, uint64_t scratch_bytes, int blocks, int lds_bytes, int microseconds, void* stream) {
  GGET_REQUIRE(scratch && scratch_bytes >= 4096 && blocks > 0 && lds_bytes >= 4 && lds_bytes <= 160 * 1024, "debug_occupy: bad arguments");
  static bool attr = false;
  if (!attr) {
    GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_fat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  if (g_occupy_fat) {
    hipLaunchKernelGGL(occupy_fat_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (float*)scratch, scratch_bytes / 4,
                       (long long)microseconds * 100);
    GGET_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (float*)scratch, scratch_bytes / 4,
                     (long long)microseconds * 100);   // wall_clock64 ticks at 100 MHz
  GGET_LAUNCH_CHECK();
  return 0;
}

extern int g_attn_oproj_off;
extern int g_gemm_variant;
extern int g_gemm_lds_headroom;
extern int g_gemm_split_last;
extern int g_gemm_stagger_ticks;
extern int g_gemm_ablate_set;
extern "C" int gget_debug_probe(gget_handle_t h, int enable, float* avg_ms_out /* [2] or NULL */) {
  GGET_REQUIRE(h != nullptr, "null handle");
  if (avg_ms_out) {   // mean launch duration over the layers of the last forward / backward that ran with the probe on
    for (int w = 0; w < 2; ++w) {
      double sum = 0.0;
      int n = 0;
      for (size_t k = 0; k + 1 < h->probe_ev[w].size(); k += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(h->probe_ev[w][k + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->probe_ev[w][k], h->probe_ev[w][k + 1]) == hipSuccess) {
          sum += ms;
          ++n;
        }
      }
      avg_ms_out[w] = n ? (float)(sum / n) : 0.f;
    }
  }
  h->probe = enable != 0;
  return 0;
}
extern "C" int gget_debug_set(int key, int value) {
  switch (key) {
    case 1: g_gemm_variant = value; return 0;
    case 2: g_gemm_lds_headroom = value; return 0;
    case 4: k_set_deterministic(value); return 0;
    case 3: g_gemm_split_last = value; return 0;
    case 5: g_gemm_stagger_ticks = value; return 0;
    case 7: g_gemm_ablate_set = value > 0 ? value : -1; return 0;
    case 8: g_head_dense = value; return 0;
    case 9: g_head_tile = value; return 0;
    case 10: g_attn_oproj_off = value; return 0;
    case 11: g_ls_norm_bwd_wide = value; return 0;
    case 13: k_set_rms_wide(value); return 0;
    case 14: k_set_ce_parts(value); return 0;
    case 15: g_gemm_cu_reserve = value > 0 ? value : 0; return 0;
    case 16: g_occupy_fat = value; return 0;
  }
  gget_set_error("debug_set: unknown key %d", key);
  return 2;
}
extern "C" int gget_op_gemm(int mode, int epilogue, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                            int lda, int ldb, int ldc, int split_k, void* stream) {
  return gget_gemm_single(mode, epilogue, A, B, C, R, M, N, K, lda, ldb, ldc, nullptr, nullptr, split_k, (hipStream_t)stream);
}
extern "C" int gget_op_gemm_streamk(int mode, int epilogue, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                                    int lda, int ldb, int ldc, void* streamk_ws, void* stream) {
  GGET_REQUIRE(streamk_ws != nullptr, "gemm_streamk: null workspace");
  gget_gemm_streamk_workspace(streamk_ws);
  const int rc = gget_gemm_single(mode, epilogue, A, B, C, R, M, N, K, lda, ldb, ldc, nullptr, nullptr, 1, (hipStream_t)stream);
  gget_gemm_streamk_workspace(nullptr);
  return rc;
}
extern "C" uint64_t gget_op_gemm_streamk_bytes(void) { return gget_gemm_streamk_bytes(); }
extern "C" int gget_op_gemm_grouped(int mode, int count, const void* const* A, const void* const* B, void* const* Cs, const int* M,
                                   const int* N, const int* K, const int* lda, const int* ldb, const int* ldc, void* stream) {
  GGET_REQUIRE(count >= 1 && count <= GGET_MAX_GROUP && A && B && Cs && M && N && K && lda && ldb && ldc, "gemm_grouped: bad arguments");
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = count;
  for (int i = 0; i < count; ++i) {
    GemmProblem& p = g.p[i];
    p.A = static_cast<const bf16_t*>(A[i]); p.B = static_cast<const bf16_t*>(B[i]); p.C = Cs[i];
    p.M = M[i]; p.N = N[i]; p.K = K[i]; p.lda = lda[i]; p.ldb = ldb[i]; p.ldc = ldc[i];
  }
  return gget_gemm_launch(mode, GGET_EPI_NONE, g, 1, (hipStream_t)stream);
}
extern "C" int gget_op_qkv_rope(const void* x, const void* wqkv, void* qkv, const float* cos_tab, const float* sin_tab,
                                const int64_t* position_ids, int T, int S, int d, void* stream) {
  GGET_REQUIRE(x && wqkv && qkv && cos_tab && sin_tab, "qkv_rope: null argument");
  GGET_REQUIRE(d > 0 && d % 64 == 0 && T > 0 && S > 0, "qkv_rope: bad shape T %d S %d d %d", T, S, d);
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = 1;
  GemmProblem& p = g.p[0];
  p.A = static_cast<const bf16_t*>(x); p.B = static_cast<const bf16_t*>(wqkv); p.C = qkv;
  p.M = T; p.N = 3 * d; p.K = d; p.lda = d; p.ldb = d; p.ldc = 3 * d;
  p.rope_cos = cos_tab; p.rope_sin = sin_tab; p.rope_pos = position_ids; p.rope_S = S; p.rope_cols = 2 * d;
  return gget_gemm_launch(GGET_GEMM_NT, GGET_EPI_ROPE, g, 1, (hipStream_t)stream);
}
extern "C" int gget_op_smtp2d(const int64_t* ids_in, int ld_in, const int64_t* node_idx, int ld_node, int64_t* ids_out,
                              int64_t* labels_out, int B, int S, int F, float smtp_2d_rate, float power, float replace_rate,
                              int vocab, int global_2d_mask, uint32_t seed, void* stream) {
  GGET_REQUIRE(ids_in && node_idx && ids_out && labels_out, "smtp2d: null argument");
  GGET_REQUIRE(B >= 0 && S >= 0 && F >= 1 && ld_in >= F && ld_node >= 1 && vocab >= 2, "smtp2d: bad shape");
  return k_smtp2d(ids_in, ld_in, node_idx, ld_node, ids_out, labels_out, B, S, F, smtp_2d_rate, power, replace_rate, vocab,
                  global_2d_mask, seed, (hipStream_t)stream);
}
extern "C" int gget_op_token_sample(const void* logits, int ld, int R, int V, int mode, float temperature, float top_p, int top_k,
                                   float alg_temp, uint32_t seed, float* conf, int64_t* tok, void* stream) {
  GGET_REQUIRE(logits && conf && tok && R >= 0 && V >= 1 && ld >= V && mode >= 0 && mode <= 2, "token_sample: bad arguments");
  return k_token_sample(logits, ld, R, V, mode, temperature, top_p, top_k, alg_temp, seed, conf, tok, (hipStream_t)stream);
}
extern "C" int gget_op_unmask_origin(int64_t* x, const int64_t* cand, int B, int N, float p_transfer, uint32_t seed, int mask_token_id,
                                     void* stream) {
  GGET_REQUIRE(x && cand && B >= 0 && N >= 0, "unmask_origin: bad arguments");
  return k_unmask_origin(x, cand, B, N, p_transfer, seed, mask_token_id, (hipStream_t)stream);
}
extern "C" int gget_op_token_confidence(const void* logits, int ld, int R, int V, int mode, float* conf, int64_t* tok, void* stream) {
  GGET_REQUIRE(logits && conf && tok, "token_confidence: null argument");
  GGET_REQUIRE(R >= 0 && V >= 2 && ld >= V && mode >= 0 && mode <= 2, "token_confidence: bad arguments (R %d V %d ld %d mode %d)", R, V, ld, mode);
  return k_token_confidence(logits, ld, R, V, mode, conf, tok, (hipStream_t)stream);
}
extern "C" int gget_op_smtp_rows(const int64_t* ids_in, const int32_t* lengths, int64_t* ids_out, int64_t* labels_out,
                                 float* wgt_out, int B, int S, int F, double umr_min, double umr_max, double power, uint32_t seed,
                                 void* stream) {
  GGET_REQUIRE(ids_in && lengths && ids_out && labels_out, "smtp_rows: null argument");
  GGET_REQUIRE(B >= 0 && S >= 1 && F >= 1 && (long)S * F < (1l << 20), "smtp_rows: bad shape (S*F must be < 2^20)");
  GGET_REQUIRE(0.0 <= umr_min && umr_min <= umr_max && umr_max <= 1.0 && power > 0.0, "smtp_rows: bad schedule");
  return k_smtp_rows(ids_in, lengths, ids_out, labels_out, wgt_out, B, S, F, umr_min, umr_max, power, seed, (hipStream_t)stream);
}
extern "C" int gget_op_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int T, int d, float eps, void* stream) {
  return k_rmsnorm_fwd(x, w, y, rstd, T, d, eps, (hipStream_t)stream);
}
extern "C" int gget_op_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                                   float* dw_accum, int T, int d, void* stream) {
  return k_rmsnorm_bwd(dy, x, w, rstd, dres, dx, dw_accum, T, d, (hipStream_t)stream);
}
extern "C" int gget_op_embed_fwd(const int64_t* ids, const void* emb, const void* gate, void* out, int T, int F, int ldF, int d,
                                 void* stream) {
  return k_embed_fwd(ids, emb, gate, out, T, F, ldF, d, (hipStream_t)stream);
}
extern "C" int gget_op_embed_bwd(const int64_t* ids, const void* dx, const void* emb, const void* gate, float* demb_accum,
                                 float* dgate_accum, int T, int F, int ldF, int d, int V, int pad_id, void* stream) {
  int32_t* ws = nullptr;
  GGET_HIP_CHECK(hipMalloc(&ws, k_embed_bwd_ws_elems((size_t)T * F, (size_t)V) * sizeof(int32_t)));  // test-only entry point
  void* cnt = nullptr;   // (GGET_EMBED_SORTED=1 forces the sorted scatter-add)
  if (k_embed_dense_ok(V, gate != nullptr))
    GGET_HIP_CHECK(hipMalloc(&cnt, (size_t)T * align_up((uint64_t)V, 64) * 2 + (size_t)kEmbDenseSplit * V * d * 4));
  void* slab = cnt ? static_cast<unsigned char*>(cnt) + (size_t)T * align_up((uint64_t)V, 64) * 2 : nullptr;
  const int rc = embed_bwd(ids, dx, emb, gate, demb_accum, dgate_accum, T, F, ldF, d, V, pad_id, ws, cnt, slab, (hipStream_t)stream,
                           ElemDropArg{0, 1.f, 0});
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(ws);
  if (cnt) (void)hipFree(cnt);
  return rc;
}
extern "C" int gget_op_rope(void* qkv, const float* cos_tab, const float* sin_tab, const int64_t* position_ids, int B, int S,
                            int H, int inverse, void* stream) {
  return k_rope(qkv, cos_tab, sin_tab, position_ids, B * S, S, H, inverse, (hipStream_t)stream);
}
extern "C" int gget_op_attn_fwd(const void* qkv, const int32_t* key_len, void* out, float* lse, int B, int S, int H, int causal,
                                const float* cos_tab, const float* sin_tab, const int64_t* position_ids, float dropout_p,
                                uint32_t dropout_seed, void* stream) {
  return k_attn_fwd(qkv, key_len, out, lse, B, S, H, causal, cos_tab, sin_tab, position_ids, dropout_p, dropout_seed,
                    (hipStream_t)stream);
}
extern "C" int gget_op_attn_oproj_fwd(const void* qkv, const int32_t* key_len, const int32_t* row_base, void* attn_out, float* lse,
                                      const void* wo, const void* x_in, void* x_mid, const void* norm_w, void* xn, float* rstd, int B, int S,
                                      int H, int causal, float eps, float dropout_p, uint32_t dropout_seed, void* stream, int32_t* taken) {
  GGET_REQUIRE(qkv && attn_out && wo && x_in && x_mid && norm_w && xn && taken, "null argument");
  int t = 0;
  const int rc = k_attn_oproj_fwd(qkv, key_len, row_base, attn_out, lse, wo, x_in, x_mid, norm_w, xn, rstd, B, S, H, causal, eps, dropout_p,
                                  dropout_seed, (hipStream_t)stream, &t);
  *taken = t;
  return rc;
}
extern "C" int gget_op_pack_wo(const void* w, uint64_t layer_stride, void* fwd, void* bwd, int d, int layers, void* stream) {
  GGET_REQUIRE(w && fwd && bwd, "null argument");
  return k_pack_wo(w, (size_t)layer_stride, fwd, bwd, d, layers, (hipStream_t)stream);
}
extern "C" int gget_op_attn_oproj_bwd(const void* dxn, const void* x_mid, const void* norm_w, const float* rstd, const void* dres, void* dx_mid,
                                      float* dw_accum, int copies, uint64_t copy_stride, const void* wot_packed, const void* qkv, const float* lse,
                                      const int32_t* key_len, const int32_t* row_base, void* dqkv, int B, int S, int H, int causal,
                                      const float* cos_tab, const float* sin_tab, const int64_t* position_ids, float dropout_p,
                                      uint32_t dropout_seed, int t_rows, void* stream, int32_t* taken, void* dattn_long) {
  GGET_REQUIRE(dxn && x_mid && norm_w && rstd && dres && dx_mid && dw_accum && wot_packed && qkv && lse && dqkv && taken, "null argument");
  int t = 0;
  const int rc = k_attn_oproj_bwd(dxn, x_mid, norm_w, rstd, dres, dx_mid, dw_accum, copies, copy_stride, wot_packed, qkv, lse, key_len, row_base,
                                  dqkv, B, S, H, causal, cos_tab, sin_tab, position_ids, dropout_p, dropout_seed, t_rows, (hipStream_t)stream, &t,
                                  dattn_long, nullptr);
  *taken = t;
  return rc;
}
extern "C" int gget_op_attn_fwd_ranges(const void* qkv, const int32_t* key_lo, const int32_t* key_hi, void* out, float* lse, int B,
                                       int S, int H, int causal, float dropout_p, uint32_t dropout_seed, void* stream) {
  GGET_REQUIRE(key_lo && key_hi, "attn_fwd_ranges: null ranges");
  return k_attn_fwd(qkv, nullptr, out, lse, B, S, H, causal, nullptr, nullptr, nullptr, dropout_p, dropout_seed,
                    (hipStream_t)stream, key_lo, key_hi);
}
extern "C" int gget_op_attn_bwd_ranges(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_lo,
                                       const int32_t* key_hi, void* dqkv, float* delta_ws, int B, int S, int H, int causal,
                                       float dropout_p, uint32_t dropout_seed, void* stream) {
  GGET_REQUIRE(key_lo && key_hi, "attn_bwd_ranges: null ranges");
  return k_attn_bwd(qkv, out, dout, lse, nullptr, dqkv, delta_ws, B, S, H, causal, nullptr, nullptr, nullptr, 0, dropout_p,
                    dropout_seed, (hipStream_t)stream, key_lo, key_hi);
}
extern "C" int gget_op_ranges_from_mask3d(const int64_t* mask3d, int32_t* key_lo, int32_t* key_hi, int B, int S, void* stream) {
  GGET_REQUIRE(mask3d && key_lo && key_hi, "ranges_from_mask3d: null argument");
  return k_ranges_from_mask3d(mask3d, key_lo, key_hi, B, S, (hipStream_t)stream);
}
extern "C" int gget_op_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len,
                                void* dqkv, float* delta_ws, int B, int S, int H, int causal, const float* cos_tab,
                                const float* sin_tab, const int64_t* position_ids, float dropout_p, uint32_t dropout_seed,
                                void* stream) {
  return k_attn_bwd(qkv, out, dout, lse, key_len, dqkv, delta_ws, B, S, H, causal, cos_tab, sin_tab, position_ids,
                    /*qk_rotated=*/0, dropout_p, dropout_seed, (hipStream_t)stream);
}
extern "C" int gget_op_copy_from_host(const void* src_pinned_host, void* dst_dev, uint64_t bytes, void* stream) {
  GGET_REQUIRE(src_pinned_host && dst_dev, "copy_from_host: null argument");
  return k_copy_from_host(src_pinned_host, dst_dev, (size_t)bytes, (hipStream_t)stream);
}
extern "C" int gget_op_attn_fwd_varlen(const void* qkv, const int32_t* key_len, const int32_t* row_base, void* out, float* lse, int B, int S,
                                       int H, int causal, const float* cos_tab, const float* sin_tab, const int64_t* position_ids,
                                       int qk_rotated, float dropout_p, uint32_t dropout_seed, void* stream) {
  GGET_REQUIRE(qkv && key_len && row_base && out, "attn_fwd_varlen: null argument");
  // (rotated q / k: the forward reads them as they are)
  return k_attn_fwd(qkv, key_len, out, lse, B, S, H, causal, qk_rotated ? nullptr : cos_tab, qk_rotated ? nullptr : sin_tab,
                    qk_rotated ? nullptr : position_ids, dropout_p, dropout_seed, (hipStream_t)stream, nullptr, nullptr, row_base);
}
extern "C" int gget_op_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len,
                                       const int32_t* row_base, void* dqkv, float* delta_ws, int B, int S, int H, int causal,
                                       const float* cos_tab, const float* sin_tab, const int64_t* position_ids, int qk_rotated,
                                       float dropout_p, uint32_t dropout_seed, void* stream) {
  GGET_REQUIRE(qkv && dout && key_len && row_base && dqkv, "attn_bwd_varlen: null argument");
  return k_attn_bwd(qkv, out, dout, lse, key_len, dqkv, delta_ws, B, S, H, causal, cos_tab, sin_tab, position_ids, qk_rotated, dropout_p,
                    dropout_seed, (hipStream_t)stream, nullptr, nullptr, row_base);
}
extern "C" int gget_op_attn_bwd_fused(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len,
                                      const int32_t* key_lo, const int32_t* key_hi, void* dqkv, float* delta_ws, void* dq_ws, int B, int S,
                                      int H, int causal, float dropout_p, uint32_t dropout_seed, void* stream) {
  GGET_REQUIRE(dq_ws && (!key_lo == !key_hi), "attn_bwd_fused: null workspace / half a key range");
  return k_attn_bwd(qkv, out, dout, lse, key_lo ? nullptr : key_len, dqkv, delta_ws, B, S, H, causal, nullptr, nullptr, nullptr, 0, dropout_p,
                    dropout_seed, (hipStream_t)stream, key_lo, key_hi, nullptr, dq_ws, (size_t)B * S * H * 64);
}
extern "C" int gget_op_gateup_geglu(const void* x, const void* wgu, void* gu, void* h, int T, int d, int ff, void* stream) {
  GGET_REQUIRE(x && wgu && gu && h && T > 0, "gateup_geglu: null argument");
  return gateup_geglu((const bf16_t*)x, (const bf16_t*)wgu, (bf16_t*)gu, (bf16_t*)h, T, d, ff, (hipStream_t)stream);
}
extern "C" int gget_op_down_dgrad_geglu(const void* dy, const void* wdown, const void* gu, void* dgu, void* dh_scratch, int T, int d,
                                        int ff, void* stream) {
  GGET_REQUIRE(dy && wdown && gu && dgu && T > 0, "down_dgrad_geglu: null argument");
  GGET_REQUIRE(dh_scratch || geglu_fusable(d, ff), "down_dgrad_geglu: this shape needs the [T][ff] dh scratch buffer");
  return down_dgrad_geglu((const bf16_t*)dy, (const bf16_t*)wdown, (const bf16_t*)gu, (bf16_t*)dgu, (bf16_t*)dh_scratch, T, d, ff,
                          (hipStream_t)stream);
}
extern "C" int gget_op_geglu_fwd(const void* gu, void* h, int T, int ff, void* stream) {
  return k_geglu_fwd(gu, h, T, ff, (hipStream_t)stream);
}
extern "C" int gget_op_geglu_bwd(const void* gu, const void* dh, void* dgu, int T, int ff, void* stream) {
  return k_geglu_bwd(gu, dh, dgu, T, ff, (hipStream_t)stream);
}
extern "C" int gget_op_ce_fwd_bwd(const void* logits, int ld, const int32_t* labels, const float* row_wgt,
                                  const int32_t* n_rows_dev, int n_rows_cap, int V, float* loss_sum, void* dlogits,
                                  float grad_scale_base, int mean_over_rows, void* stream) {
  GGET_REQUIRE(row_wgt == nullptr, "per-row weights go through the engine path (sample_wgt)");
  return k_ce_fwd_bwd(logits, ld, labels, nullptr, nullptr, 1, n_rows_dev, n_rows_cap, V, loss_sum, dlogits, grad_scale_base,
                      mean_over_rows, nullptr, (hipStream_t)stream);
}


// ================================================================================================
// data-parallel gradient exchange over RCCL (reference: DDP all-reduce opt_utils.py:13,
// DeepSpeed ZeRO-2 reduce-scatter ds_config2_pt.json:29-32, communicator setup misc_utils.py:519-526)
// ================================================================================================
// The library is bound with dlopen("librccl.so.1") at the first gget_comm_* call: inside a PyTorch process that is the copy
// torch already loaded (one RCCL per process), and a caller that never exchanges gradients never needs RCCL at all.
namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "symbols missing";     // dlerror() text captured once, at the failing dlopen (a second dlerror() call returns NULL)
};

RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
      const char* e = dlerror();
      snprintf(api.why, sizeof(api.why), "%s", e ? e : "dlopen failed");
    }
    if (lib) {
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
      api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
    }
  }
  return &api;
}

#define GGET_RCCL_CHECK(expr)                                                                          \
  do {                                                                                                 \
    ncclResult_t _r = (expr);                                                                          \
    if (_r != ncclSuccess) {                                                                           \
      gget_set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__);  \
      return 1;                                                                                        \
    }                                                                                                  \
  } while (0)

__global__ void __launch_bounds__(256) bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, uint64_t n) {
  // n is a multiple of 128 (flat-arena alignment): 8 elements per thread, 16-byte loads, two 16-byte stores
  for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (uint64_t)gridDim.x * 256 * 8) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(src + i), f);
    *reinterpret_cast<float4*>(dst + i) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(dst + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
}

}  // namespace

extern "C" int gget_comm_unique_id(void* out_bytes) {
  GGET_REQUIRE(out_bytes, "comm_unique_id: null argument");
  GGET_REQUIRE(rccl()->ok, "RCCL (librccl.so.1) could not be loaded: %s", rccl()->why);
  static_assert(sizeof(ncclUniqueId) == GGET_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  GGET_RCCL_CHECK(rccl()->GetUniqueId(&id));
  memcpy(out_bytes, &id, sizeof(id));
  return 0;
}

extern "C" int gget_comm_init(gget_handle_t h, int rank, int world, const void* unique_id_bytes) {
  GGET_REQUIRE(h && unique_id_bytes && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d world %d)", rank, world);
  GGET_REQUIRE(h->comm == nullptr, "comm_init: this handle already has a communicator");
  GGET_REQUIRE(rccl()->ok, "RCCL (librccl.so.1) could not be loaded: %s", rccl()->why);
  ncclUniqueId id;
  memcpy(&id, unique_id_bytes, sizeof(id));
  ncclComm_t c = nullptr;
  GGET_RCCL_CHECK(rccl()->CommInitRank(&c, world, id, rank));
  h->comm = c;
  h->comm_rank = rank;
  h->comm_world = world;
  // a collective's kernel will share the chip with the compute stream: keep LDS headroom on every CU
  if (world > 1 && g_gemm_cu_reserve == 0 && g_gemm_lds_headroom < 2 && getenv("GGET_DP_LDS_HEADROOM") && atoi(getenv("GGET_DP_LDS_HEADROOM"))) g_gemm_lds_headroom = 2;   // (opt-in since round 5: DESIGN.md section 6)
  return 0;
}

extern "C" int gget_comm_destroy(gget_handle_t h) {
  if (!h) return 0;
  if (h->comm) {
    rccl()->CommDestroy(static_cast<ncclComm_t>(h->comm));
    h->comm = nullptr;
  }
  if (h->comm_f32) {
    (void)hipFree(h->comm_f32);
    h->comm_f32 = nullptr;
    h->comm_f32_elems = 0;
  }
  h->comm_world = 1;
  h->comm_loopback = false;
  return 0;
}

extern "C" int gget_comm_move(gget_handle_t dst, gget_handle_t src) {
  // A handle that is re-created with larger capacities (a bigger batch arrived) must keep exchanging gradients with the SAME
  // communicator: creating a new one is a collective, and only the ranks whose batch grew would enter it.
  GGET_REQUIRE(dst && src && dst != src, "comm_move: bad arguments");
  GGET_REQUIRE(dst->comm == nullptr && !dst->comm_loopback, "comm_move: the destination handle already has a communicator");
  GGET_REQUIRE(dst->plan.n_params == src->plan.n_params, "comm_move: the two handles hold different models");
  dst->comm = src->comm;
  dst->comm_loopback = src->comm_loopback;
  src->comm_loopback = false;
  dst->comm_rank = src->comm_rank;
  dst->comm_world = src->comm_world;
  dst->comm_f32 = src->comm_f32;
  dst->comm_f32_elems = src->comm_f32_elems;
  src->comm = nullptr;
  src->comm_f32 = nullptr;
  src->comm_f32_elems = 0;
  src->comm_world = 1;
  return 0;
}

__global__ void __launch_bounds__(256) scale_bf16_kernel(bf16_t* __restrict__ g, uint64_t n, float mul) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) g[i] = f2bf(bf2f(g[i]) * mul);
}

extern "C" int gget_comm_init_loopback(gget_handle_t h, int world) {
  GGET_REQUIRE(h && world >= 1, "comm_init_loopback: bad arguments (world %d)", world);
  GGET_REQUIRE(h->comm == nullptr && !h->comm_loopback, "comm_init_loopback: this handle already has a communicator");
  h->comm_loopback = true;
  h->comm_rank = 0;
  h->comm_world = world;
  if (world > 1 && g_gemm_cu_reserve == 0 && g_gemm_lds_headroom < 2 && getenv("GGET_DP_LDS_HEADROOM") && atoi(getenv("GGET_DP_LDS_HEADROOM"))) g_gemm_lds_headroom = 2;   // (opt-in since round 5: DESIGN.md section 6)   // (the launch menu of a data-parallel run, as gget_comm_init)
  return 0;
}

extern "C" int gget_allreduce_range_async(gget_handle_t h, uint64_t offset, uint64_t count, int fp32_accumulate, void* side_stream) {
  GGET_REQUIRE(h && (h->comm || h->comm_loopback), "allreduce_grads: call gget_comm_init first");
  GGET_REQUIRE(offset <= h->plan.n_params && count <= h->plan.n_params - offset, "allreduce_range: [%llu, +%llu) leaves the gradient array",
               (unsigned long long)offset, (unsigned long long)count);
  if (count == 0) return 0;
  hipStream_t st = (hipStream_t)side_stream;
  bf16_t* g = h->G + offset;
  const uint64_t n = count;
  if (h->comm_loopback) {   // world identical contributions: the sum is world x (exact in bf16 for a power of two)
    const int grid = (int)std::min<uint64_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid), dim3(256), 0, st, g, n, (float)h->comm_world);
    GGET_LAUNCH_CHECK();
    return 0;
  }
  ncclComm_t c = static_cast<ncclComm_t>(h->comm);
  if (!fp32_accumulate) {
    GGET_RCCL_CHECK(rccl()->AllReduce(g, g, n, ncclBfloat16, ncclSum, c, st));
    return 0;
  }
  // fp32 reduction: a bf16 ring sum rounds after every hop (world - 1 roundings); widening the bucket first makes the sum
  // exact up to the single final rounding, at twice the bytes on the wire.  Staging buffer owned by the communicator.
  if (h->comm_f32_elems < n) {
    if (h->comm_f32) GGET_HIP_CHECK(hipFree(h->comm_f32));
    h->comm_f32 = nullptr;
    uint64_t cap = 0;
    for (const auto& r : h->bucket_range) cap = std::max<uint64_t>(cap, r.second - r.first);
    cap = std::max<uint64_t>(cap, n);
    GGET_HIP_CHECK(hipMalloc(&h->comm_f32, cap * sizeof(float)));
    h->comm_f32_elems = cap;
  }
  const int grid = (int)std::min<uint64_t>(4096, (n / 8 + 255) / 256);
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid), dim3(256), 0, st, g, h->comm_f32, n);
  GGET_LAUNCH_CHECK();
  GGET_RCCL_CHECK(rccl()->AllReduce(h->comm_f32, h->comm_f32, n, ncclFloat32, ncclSum, c, st));
  return k_f32_to_bf16(h->comm_f32, g, n, st);
}

extern "C" int gget_allreduce_grads_async(gget_handle_t h, int bucket, int fp32_accumulate, void* side_stream) {
  GGET_REQUIRE(h && (h->comm || h->comm_loopback), "allreduce_grads: call gget_comm_init first");
  GGET_REQUIRE(bucket >= -1 && bucket < (int)h->bucket_range.size(), "allreduce_grads: bucket %d out of range", bucket);
  const uint64_t lo = bucket < 0 ? 0 : h->bucket_range[bucket].first;
  const uint64_t hi = bucket < 0 ? h->plan.n_params : h->bucket_range[bucket].second;
  return gget_allreduce_range_async(h, lo, hi - lo, fp32_accumulate, side_stream);
}
