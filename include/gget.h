/*
 * gget.h - C ABI of libgget_hip.so, the MI355X (gfx950) engine for the GraphGPT
 * Graph-Eulerian-Transformer hot path.
 *
 * The reference (alibaba/graph-gpt) has no FFI layer: its operator API for this path is the
 * Python nn.Module surface.  This header is the boundary a binding for that
 * surface sits on; every entry point cites the reference interface it replaces
 * (paths relative to the reference repo root).  The build's own ctypes binding is
 * graph-gpt_amd/_lib.py; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; gget_last_error() gives the text;
 *     nothing throws across the ABI;
 *   - all pointers named *_dev are device (HBM) pointers owned by the caller; the engine never
 *     allocates, frees or copies across PCIe behind the caller's back (the caller - PyTorch in
 *     our binding - owns memory and streams);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no host sync unless
 *     the function says so;
 *   - not thread-safe per handle; one handle per GPU per process (one process per GPU).
 *   - integer batch tensors use the reference's dtypes: int64 ids/labels/masks.
 */
#ifndef GGET_H_
#define GGET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGET_KIND_PRETRAIN 0 /* GraphGPTPretrainBase (src/models/graphgpt/modeling_pretrain.py:57) */
#define GGET_KIND_TASK 1     /* GraphGPTTaskModel    (src/models/graphgpt/modeling_finetune.py:64) */

#define GGET_PROBLEM_SINGLE_LABEL 0 /* CrossEntropy on pooled logits (modeling_finetune.py:209-214) */
#define GGET_PROBLEM_REGRESSION_L1 1 /* L1Loss  (modeling_finetune.py:183-197) */
#define GGET_PROBLEM_REGRESSION_MSE 2 /* MSELoss */
#define GGET_PROBLEM_TOKEN_CE 5 /* loss_type = "token_ce" (node-level tasks, modeling_finetune.py:162-164, :198-202): `score` on EVERY row, task_labels int64 [B,S] with -100 = unlabelled, mean cross-entropy over the labelled rows; task_logits_dev is then f32 [B,S,num_labels] (the reference hands the all-row logits back as `task_logits`), with or without labels */
#define GGET_PROBLEM_AUC 4 /* pairwise squared-hinge AUC surrogate on logit[:,1]-logit[:,0] (src/utils/loss_utils.py:25-53, modeling_finetune.py:203-207); see gget_set_auc */
#define GGET_PROBLEM_MULTI_LABEL 3 /* BCEWithLogitsLoss on the non-NaN entries of float labels [B,num_labels] (modeling_finetune.py:227-230) */

/* Field meaning = GraphGPTConfig (src/models/graphgpt/configuration_graphgpt.py:26-110). */
typedef struct gget_config_t {
  int32_t kind;           /* GGET_KIND_* */
  int32_t vocab_size;
  int32_t hidden_size;    /* d, multiple of 64 */
  int32_t intermediate_size; /* ff, multiple of 64 */
  int32_t num_layers;
  int32_t num_heads;      /* d / 64 (head_dim is 64: src/utils/modules_utils.py:37-42) */
  int32_t stacked_feat;   /* F */
  int32_t next_n_token;   /* pre-train: F; 1 => n_token_proj is Identity */
  int32_t gated_agg;      /* stacked_feat_agg_method == "gated" */
  int32_t causal;         /* causal_attention */
  int32_t max_position;   /* max_position_embeddings */
  int32_t num_labels;     /* fine-tune head width */
  int32_t score_bias;     /* problem_type == "regression" */
  int32_t pad_token_id;
  float rms_eps;
  float rope_theta;
  float layer_scale_init; /* >0 => lambda_1 / lambda_2 per layer (utils_graphgpt.py:95-104) */
  int32_t max_tokens;     /* capacity: max B*S of any forward call */
  int32_t max_batch;      /* capacity: max B */
  float path_pdrop;       /* >0 => stochastic depth is available (rate linspace(0,path_pdrop,L), utils_graphgpt.py:184) */
  float mlp_pdrop;        /* >0 => MLP dropouts are available (utils_graphgpt.py:69-80): the residual adds run as their own kernels */
  int32_t head_mlp_layers;/* fine-tune: hidden layers of the `MLP` score head (len(config.mlp), src/utils/modules_utils.py:8-34); 0 = Linear */
  int32_t head_mlp[4];    /* their widths */
  int32_t embed_dim;      /* >0 => raw-embedding inputs [B,S,embed_dim] are projected and added to the token embeddings (config.embed_dim; modeling_pretrain.py:69-84, :131-149; modeling_helpers.py:127-139); multiple of 64 */
} gget_config_t;

/* Arena sizes the caller must provide (all 256-byte aligned device buffers). */
typedef struct gget_sizes_t {
  uint64_t n_params;        /* parameter elements (flat) */
  uint64_t param_bf16_bytes;/* compute copy of the parameters, bf16 [n_params] */
  uint64_t master_bytes;    /* fp32 master weights [n_params] */
  uint64_t adam_bytes;      /* fp32 m and v, 2*[n_params] */
  uint64_t grad_bf16_bytes; /* bf16 gradients [n_params] (what the DP all-reduce moves) */
  uint64_t workspace_bytes; /* activations + scratch for max_tokens */
} gget_sizes_t;

typedef struct gget_buffers_t {
  void* param_bf16_dev;
  void* master_dev;
  void* adam_m_dev;
  void* adam_v_dev;
  void* grad_bf16_dev;
  void* workspace_dev;
  const float* rope_cos_dev; /* [max_position][32] fp32, hf LlamaRotaryEmbedding tables; NULL => engine computes */
  const float* rope_sin_dev;
} gget_buffers_t;

typedef struct gget_param_info_t {
  char name[96];      /* reference state-dict key, e.g. "model.layers.3.mlp.down_proj.weight" */
  int32_t ndim;
  int64_t shape[2];
  uint64_t offset;    /* element offset into the flat param / master / adam / grad arrays */
  int32_t layer;      /* decoder layer index, -1 embeddings, num_layers = final norm + heads */
} gget_param_info_t;

typedef struct gget_engine* gget_handle_t;

const char* gget_last_error(void);
int gget_version(void);

/* replaces: GraphGPTPretrainBase.__init__/GraphGPTTaskModel.__init__ (modeling_pretrain.py:58-117,
 * modeling_finetune.py:67-105): computes the flat parameter layout and workspace need.
 * The flat arrays hold the named parameters (gget_param_info) at 128-element aligned offsets; the gaps, and for the
 * pre-train model round_up(V,64)-V zero rows behind lm_head.weight (its dgrad GEMM runs over K = round_up(V,64)), belong to
 * no parameter: the caller provides all arenas ZERO-FILLED (gradients and AdamW state of the gaps then stay zero).
 * gget_create additionally clears the workspace and the lm_head pad rows of every arena it is given (synchronously). */
int gget_query_sizes(const gget_config_t* cfg, gget_sizes_t* out);
int gget_create(const gget_config_t* cfg, const gget_buffers_t* bufs, gget_handle_t* out);
int gget_destroy(gget_handle_t h);

/* replaces: nn.Module.named_parameters()/state_dict() key+shape enumeration. */
int gget_param_count(gget_handle_t h);
int gget_param_info(gget_handle_t h, int index, gget_param_info_t* out);
/* number of DP gradient buckets (embeddings | one per decoder layer | final norm + heads) and their
 * element ranges in the flat gradient array, in the order backward completes them. */
int gget_bucket_count(gget_handle_t h);
int gget_bucket_range(gget_handle_t h, int bucket, uint64_t* offset, uint64_t* count);

/* replaces: `attention_dropout` / `path_pdrop` of the config + model.train()/eval() (reference launch scripts set
 * attention_dropout 0.1, e.g. examples/graph_lvl/pcqm4m_v2_pretrain.sh:20; ogbl-ppa fine-tuning adds path dropout 0.2,
 * examples/edge_lvl/ppa_supervised.sh:21-25): probabilities and RNG seed used by the NEXT forward and its backward;
 * zeros (default) are evaluation behaviour.  path_p is the LAST layer's rate (layer l uses path_p*l/(L-1)) and needs a
 * handle created with config.path_pdrop > 0. */
int gget_set_dropout(gget_handle_t h, float attention_p, float path_p, uint32_t seed);

/* Var-len (padding-free) token layout.  replaces: nothing the reference has as an operator - it runs every token-wise module over the
 * padded [B,S] grid and masks attention (src/models/graphgpt/modeling_helpers.py:38-64); its only remedy against padding is the
 * collator-side `pack_tokens` option (src/data/tokenizer.py:359-415, served by gget_forward_pretrain_packed).  n_real_tokens =
 * sum(attention_mask) of the NEXT gget_forward_pretrain / gget_forward_task batch, one of
 *     n > 0              the caller's count (the host knows it from its collator): no device->host traffic at all;
 *     GGET_TOKENS_AUTO   counted by the engine: the key lengths are summed on the device and the total is read back before the
 *                        launches are sized - 4 bytes and ONE host wait per forward (the counting kernel stores the total into a
 *                        pinned, coherent host word and the host polls it: no copy packet and no event in the stream since round 6;
 *                        the launches that do not need the count - head compaction, packed o weights - are enqueued in front of the
 *                        wait).  This is what the model classes
 *                        pass for a device-resident mask, i.e. for a call shaped exactly like the reference's step, which has just
 *                        synchronised on `.to(device)` of every batch tensor (src/utils/training_utils.py:17-26): the stream is
 *                        drained, the read costs a kernel launch + a 4-byte store over the host link (measured: bench.py `layouts`);
 *     anything else      unknown -> padded layout (the default of a bare C-ABI forward).
 * The count is consumed at the ENTRY of the next forward call, whether that call succeeds or not.
 * In the var-len layout the engine compacts the real tokens once (sample b owns rows [cu[b], cu[b] + len[b]))
 * and runs embedding, every GEMM / norm / residual of the layer stack, attention (per-sample row offsets) and the backward on
 * round_up(n_real_tokens, 64) rows instead of B*S.  Results are those of the padded layout (pad rows never influence real rows; the
 * loss, its normalisers, the dropout streams and all [B,S]-shaped inputs / outputs keep their logical coordinates: element-dropout hashes
 * and rope_range tables are keyed by the logical row, full-logit inference keeps the [B S F, V] cell order of its logits, the token-level
 * head writes task_logits [B,S,C] with zeros at padded positions, raw-embedding inputs are read at the logical row).  The engine falls back
 * to the padded layout by itself for packed rows only.  gget_hidden_states / gget_layer_hidden_states (pointers into the workspace) refuse after a var-len forward; gget_hidden_states_grid copies out of either layout.
 * A caller's count that DISAGREES with the mask cannot go unnoticed: the samples are cut at the count (no kernel leaves the rows of
 * the step), the step's loss is NaN, and a sticky device flag is raised that gget_deferred_status reports; the same flag is raised by
 * a label != -100 at a padded position (the reference's collator pads labels with -100; such a row does not exist in the compact
 * layout - its loss term is taken from a pad-token row instead of the padded grid's row).
 * gget_varlen_status: out[0] = 1 if the last forward ran var-len, out[1] = rows it ran on, out[2] = 1 if sum(key lengths) on the
 * device differed from n_real_tokens; synchronises the stream. */
#define GGET_TOKENS_AUTO (-2)
int gget_set_token_count(gget_handle_t h, int64_t n_real_tokens);
/* position_ids of a forward index the RoPE table precomputed for config.max_position rows (the reference evaluates the rotary embedding
 * per call, hf LlamaRotaryEmbedding.forward :111-127, and accepts any position): the engine reads them through a copy clamped to
 * [0, max_position) and raises a sticky device-side flag when it had to clamp.  *clamped_out = that flag (then cleared); synchronises
 * the stream - call it when convenient (end of an epoch, a logging step), not per step. */
int gget_position_status(gget_handle_t h, int32_t* clamped_out, void* stream);
int gget_varlen_status(gget_handle_t h, int32_t out[3], void* stream);
/* The sticky device-side input guards in one read (then cleared; synchronises the stream - call it where the loop synchronises anyway):
 * out[0] = position_ids were clamped into the RoPE table (as gget_position_status), out[1] = a var-len step ran on a token count the
 * mask contradicted, or met a label at a padded position (see gget_set_token_count).  replaces: the IndexError / shape error the
 * reference's nn.Embedding / rotary embedding raise synchronously for such inputs. */
int gget_deferred_status(gget_handle_t h, int32_t out[2], void* stream);

/* replaces: `embed_pdrop` / `mlp_pdrop` of the config + model.train()/eval(): nn.Dropout on the gathered token embeddings
 * (modeling_helpers.py:96-98) and the two dropouts of the decoder MLP - on act(gate)*up and on down_proj's output
 * (utils_graphgpt.py:69-80) - for the NEXT forward and its backward, keyed by the seed of gget_set_dropout; zeros (default)
 * are evaluation behaviour.  mlp_p > 0 needs a handle created with config.mlp_pdrop > 0.  head_p: `config.dropout`, the
 * dropout between activation and Linear inside the MLP score head (src/utils/modules_utils.py:27-33). */
int gget_set_dropout_ex(gget_handle_t h, float embed_p, float mlp_p, float head_p);

/* replaces: the `inputs_raw_embeds` argument of the model forwards (config.embed_dim > 0): fp32 [B,S,embed_dim] on the device, consumed by the
 * NEXT gget_forward_* call (then forgotten).  Pre-train (modeling_pretrain.py:131-149): rows that carry a label are replaced by the learned
 * `emb_mask_token` (ALL of its next_n_token labels set - or, first_label_only != 0 = the smtp_inside rule, its first label), then RMSNorm (`embed_layernorm`), dropout (embed_pdrop of gget_set_dropout_ex) and `embed_proj`, added to the stacked token
 * embeddings; fine-tune (modeling_helpers.py:127-139): the same without the mask token.  A forward of a handle created with embed_dim > 0
 * fails without it, as the reference's does. */
int gget_set_raw_embeds(gget_handle_t h, const float* raw_embeds_dev, int first_label_only);

/* replaces: `config.rope_range` (configuration_graphgpt.py:42; utils_graphgpt.reset_pos_ids :574-581 through resolve_forward_defaults,
 * modeling_common.py:185-203): when > 0 and position ids are passed to a forward, the positions of every row are rescaled to
 * float(p) * rope_range / float(max_s p + 1) before the rotary embedding.  The engine then evaluates the angles per token (the
 * precomputed integer-position table does not apply).  Default 0 = off; forwards without position ids are not affected, as in the
 * reference. */
int gget_set_rope_range(gget_handle_t h, float rope_range);

/* replaces: `config.stack_method` ("short" | "long", configuration_graphgpt.py:50; examples/node_lvl/proteins_supervised.sh:31 runs
 * "long").  stack_long != 0: (i) the stacked embedding of every token is multiplied by min(1, 1 / (its non-zero feature ids + 1e-7))
 * (_get_stacked_inputs_embeds, modeling_helpers.py:106-110) in forward and backward, both model kinds; (ii) the SMTP head weighs
 * every labelled cell of sample b by 1 / (labelled cells of b + 1e-7) and sums / (B S F) - the per-feature-level path
 * (_prepare_for_stacked_feat_labels_per_feat_lvl :327-342 -> _get_dlm_ce_loss :180-198, modeling_pretrain.py:230-236); a
 * sample_wgt passed to the forward is ignored then, as in the reference (:368-374).  Default 0 = "short". */
int gget_set_stack_method(gget_handle_t h, int stack_long);

/* replaces: `config.focal_gamma` (configs/training/base.yaml:60): > 0 turns the SMTP head's mean cross-entropy into the focal loss
 * of utils_graphgpt.FocalLoss (:340-376) - every row weighted by (1 - p_target)^gamma, the weight detached as in the reference.
 * Applies to gget_forward_pretrain[_packed] without sample_wgt (_get_ce_loss, modeling_helpers.py:158-160). */
int gget_set_focal_gamma(gget_handle_t h, float gamma);

/* replaces: `config.num_neg` + the torch RNG behind `torch.randperm` in auc_loss (src/utils/loss_utils.py:25-43): negatives per
 * positive and the seed of the counter-hash permutation the NEXT gget_forward_task(problem_type = GGET_PROBLEM_AUC) draws its
 * negative samples with (idx = perm(P * num_neg) % N_neg; perm = rank of the hashed keys). */
int gget_set_auc(gget_handle_t h, int num_neg, uint32_t seed);

/* replaces: load_state_dict + `.to(bfloat16)`: refresh the bf16 compute copy from the fp32 master. */
int gget_sync_params(gget_handle_t h, void* stream);

/* replaces: GraphGPTPretrainBase.forward (modeling_pretrain.py:152-266).
 *   input_ids i64 [B,S,F]; attention_mask i64 [B,S] (1 real / 0 right padding);
 *   labels i64 [B,S,next_n] (-100 = not predicted) or NULL (=> logits for every cell, no loss);
 *   sample_wgt f32 [B] or NULL (dLM weighting, modeling_pretrain.py:230-236);
 *   position_ids i64 [B,S] or NULL (=> arange(S), hf LlamaModel.forward :389-392).
 *   loss_dev: f32[1] device scalar (head1_loss).  Logits stay in the workspace: gget_head_logits. */
int gget_forward_pretrain(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev,
                          const int64_t* labels_dev, const float* sample_wgt_dev, const int64_t* position_ids_dev,
                          int B, int S, float* loss_dev, void* stream);

/* replaces: GraphGPTTaskModel.forward + calculate_task_loss (modeling_finetune.py:236-326, :167-234).
 *   task_labels: i64 [B] (single label) or f32 [B] (regression), NULL => no loss;
 *   task_logits_dev f32 [B,num_labels] (pooled "last" row, returned as .float());
 *   task_hidden_dev bf16 [B,d] or NULL. */
/* replaces: the same forward on PACKED rows (reference GraphsMapDataset.pack_token_seq, src/data/tokenizer.py:359-415;
 * block-diagonal mask built at src/utils/tokenizer_utils.py:349-355 and expanded by _expand_mask_from_3d_mask,
 * modeling_helpers.py:51-64): attention_mask3d is int64 [B,S,S]; a token attends exactly the contiguous key range of its
 * own graph (first .. last non-zero of its mask row), all-zero rows are padding.  backward / adamw as usual. */
int gget_forward_pretrain_packed(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask3d_dev,
                                 const int64_t* labels_dev, const float* sample_wgt_dev, const int64_t* position_ids_dev,
                                 int B, int S, float* loss_dev, void* stream);
int gget_forward_task(gget_handle_t h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev,
                      const int64_t* position_ids_dev, const void* task_labels_dev, const float* sample_wgt_dev,
                      int problem_type, int B, int S, float* loss_dev, float* task_logits_dev,
                      void* task_hidden_dev, void* stream);

/* replaces: loss.backward() / engine.backward(loss) (src/utils/training_utils.py:44,66).
 * Full backward of the last forward; bf16 gradients of every parameter land in grad_bf16_dev.
 * The staged form lets the caller overlap the DP all-reduce of bucket k with the backward of the
 * next layers (one HIP event per bucket on the caller's side):
 *   gget_backward_begin  -> head + final norm   (bucket 0 complete)
 *   gget_backward_layer(i) for i = L-1..0       (bucket L-i complete)
 *   gget_backward_end    -> embedding scatter   (last bucket complete)
 * The backward reads the SAME bf16 weights the forward used (the caller's arena): do not write parameters between a forward and its
 * backward.  gget_sync_params / gget_adamw_step in between are tolerated - the per-sample kernels' fragment-major o-weight copies of that
 * forward are dropped and the backward takes the three-launch form on the live weights - but the gradients then mix two weight versions. */
int gget_backward(gget_handle_t h, float loss_scale, void* stream);
int gget_backward_begin(gget_handle_t h, float loss_scale, void* stream);
int gget_backward_layer(gget_handle_t h, int layer, void* stream);
int gget_backward_end(gget_handle_t h, void* stream);

/* Engine options (no reference counterpart: promises of the caller that let the engine skip work).
 * GGET_OPT_NORM_FROM_BACKWARD = 1: "nothing rewrites the bf16 gradient array between gget_backward* and gget_adamw_step" (a single-rank
 *   step; any exchange rewrites it).  gget_adamw_step then takes the squared norm of the decoder layers' weight-gradient matrices
 *   (93 % of the base model's parameters) from per-tile partial sums their weight-gradient launches left behind and reads only the
 *   remaining tensors again - the clip + AdamW step (reference: torch.nn.utils.clip_grad_norm_ + AdamW, src/utils/training_utils.py
 *   :72-82, DeepSpeed's gradient_clipping) loses its 244 MB norm pass.  Ignored (full pass) whenever grad_scale != 1 or a layer's launch
 *   left no partials.  The norm is the same sum in another (fixed) order: equal to fp32 rounding. */
#define GGET_OPT_NORM_FROM_BACKWARD 1
/* GGET_OPT_SKIP_NONFINITE_STEP = 1: gget_adamw_step touches nothing (weights, Adam moments, bf16 copy) when the global gradient norm is
 *   inf / NaN - what torch.cuda.amp.GradScaler.step does on the reference's DDP branch (src/utils/training_utils.py:46-86: scale,
 *   backward, unscale_, clip, scaler.step, scaler.update).  The norm still reaches gnorm_dev, so the caller can tell (and keep its
 *   1-based step count for the skipped step).  Off by default: DeepSpeed's bf16 optimizer, the path the engine reproduces, has no such guard. */
#define GGET_OPT_SKIP_NONFINITE_STEP 2
int gget_set_option(gget_handle_t h, int option, int value);

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange.  replaces: the DDP gradient all-reduce (src/utils/opt_utils.py:13),
 * DeepSpeed ZeRO-2's reduce-scatter/all-gather (examples/ds_config2_pt.json:29-32) and the communicator bootstrap of
 * set_dist_env (src/utils/misc_utils.py:507-539, init_process_group at :519-526).  One process per GPU; rank 0 obtains a
 * unique id and hands it to the other ranks out of band (launcher store, MPI, torch.distributed broadcast ...).
 * gget_allreduce_grads_async enqueues the SUM all-reduce of one gradient bucket (gget_bucket_range; -1 = the whole flat
 * array) on `side_stream`: the caller makes that stream wait for the backward stage that finalises the bucket
 * (gget_backward_begin / _layer / _end) and makes the compute stream wait for it before gget_adamw_step, whose grad_scale
 * = 1/world turns the sum into the mean.  fp32_accumulate != 0 widens the bucket to fp32 for the reduction (one rounding
 * instead of world-1) at twice the wire bytes.  RCCL is bound lazily (dlopen of librccl.so.1).
 * ------------------------------------------------------------------------------------------ */
#define GGET_UNIQUE_ID_BYTES 128
int gget_comm_unique_id(void* out_bytes /* GGET_UNIQUE_ID_BYTES */);
int gget_comm_init(gget_handle_t h, int rank, int world, const void* unique_id_bytes);
int gget_comm_destroy(gget_handle_t h);
/* hands the communicator of `src` (and its staging buffer) to `dst`, a handle of the same model created with other capacities
 * (the host re-creates the handle when a larger batch arrives): no collective, so ranks may do it independently. */
int gget_comm_move(gget_handle_t dst, gget_handle_t src);
int gget_allreduce_grads_async(gget_handle_t h, int bucket, int fp32_accumulate, void* side_stream);
/* the same collective over an arbitrary element range [offset, offset + count) of the flat gradient array: several consecutive
 * buckets in ONE collective (fewer, larger messages - xGMI rings are per-link bound; the host picks the coalescing, e.g.
 * GGET_DP_BUCKET_MB in graph-gpt_amd/training.py).  The range must lie inside [0, n_params). */
int gget_allreduce_range_async(gget_handle_t h, uint64_t offset, uint64_t count, int fp32_accumulate, void* side_stream);
/* A communicator that needs no peers (tests / dry runs of the exchange schedule on one GPU): the handle behaves as rank 0 of `world`
 * ranks that all hold THIS rank's gradients, i.e. every "all-reduce" multiplies the range by `world` on the side stream (the same
 * stream / event protocol as the RCCL collective; with grad_scale = 1/world the step equals the single-rank step).  No reference
 * counterpart - RCCL refuses two ranks on one device, and the exchange schedule (bucket ranges, stream waits, the 1/world folded into
 * AdamW) must be testable beyond world 1 on a one-GPU box. */
int gget_comm_init_loopback(gget_handle_t h, int world);

/* replaces: clip_grad_norm_ + AdamW.step / FusedAdam (training_utils.py:68-80, opt_utils.py:18-24,
 * examples/ds_config2_pt.json:11-19).  grad_scale multiplies every gradient first (1/world after a
 * sum all-reduce); max_grad_norm <= 0 disables clipping; step is 1-based.
 * gnorm_dev (f32[1], may be NULL) receives the pre-clip global L2 norm. */
int gget_adamw_step(gget_handle_t h, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float max_grad_norm, float grad_scale, int step, float* gnorm_dev, void* stream);

/* Head bookkeeping of the last pre-train forward.  counts[0] = M (positions with >=1 masked
 * feature), counts[1] = Lm (masked feature tokens).  Synchronises `stream`.
 * logits: bf16 [Lm][ld] with ld = round_up(V,64) (pad columns are zero). */
int gget_head_counts(gget_handle_t h, int32_t counts[2], void* stream);
int gget_head_logits(gget_handle_t h, const void** logits_dev, int32_t* ld);
/* final hidden states bf16 [B*S][d] of the last forward (outputs[0] of the backbone) */
int gget_hidden_states(gget_handle_t h, const void** hidden_dev);
/* residual stream bf16 [B*S][d] ENTERING decoder layer `layer` (layer = num_layers: leaving the last layer, before the final norm) of
 * the last forward.  replaces: `output_hidden_states=True` of the reference's backbone (hf LlamaModel.forward modeling_llama.py
 * :401-414 collects exactly these tensors); used by the per-layer error budget of tests/test_error_budget.py.  Padded layout only. */
int gget_layer_hidden_states(gget_handle_t h, int layer, const void** hidden_dev);
/* Both of the above as a COPY in the reference's [B,S,d] layout, after a forward on either token layout: layer = -1 the final-normed
 * hidden states (`outputs.hidden_states[-1]` of modeling_pretrain.py:264 / modeling_finetune.py:323), 0 .. num_layers the residual
 * stream entering that layer (hf LlamaModel.forward :401-414).  After a var-len forward the rows go back to their (b, s) positions;
 * positions behind a sample's tokens read as zero.  out_dev: bf16 [B * S * hidden] for the last forward's B and S. */
int gget_hidden_states_grid(gget_handle_t h, int layer, void* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (the individual HIP kernels), used by the parity tests and by
 * callers that only want one op.  bf16 tensors unless stated; row-major; ld* in elements.
 * ------------------------------------------------------------------------------------------ */
#define GGET_GEMM_NT 0 /* C[M,N] = A[M,K] * B[N,K]^T   (forward  y = x W^T) */
#define GGET_GEMM_NN 1 /* C[M,N] = A[M,K] * B[K,N]     (dgrad    dx = dy W) */
#define GGET_GEMM_TN 2 /* C[M,N] = A[K,M]^T * B[K,N]   (wgrad    dW = dy^T x) */
#define GGET_EPI_NONE 0
#define GGET_EPI_RESIDUAL 1 /* C = A*B + R (R bf16 [M,N], ldr = ldc) */
#define GGET_EPI_ATOMIC_F32 2 /* C is fp32, C += A*B with atomics (split-K) */
#define GGET_EPI_ROPE 4 /* C = rope(A*B) on columns [0, rope_cols): RoPE fused into the q|k|v projection (engine only) */
#define GGET_EPI_SLAB_F32 3 /* C is fp32 [split_k][M][ldc]: slice s of K writes slab s (reduced by the caller) */
#define GGET_EPI_GEGLU_FWD 5 /* gate|up projection with the gated-GELU product fused into the epilogue (gget_op_gateup_geglu) */
#define GGET_EPI_GEGLU_BWD 6 /* dh = dy W_down with the gated-GELU backward fused into the epilogue (gget_op_down_dgrad_geglu) */
int gget_op_gemm(int mode, int epilogue, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                 int lda, int ldb, int ldc, int split_k, void* stream);
/* the same contraction with stream-K allowed (csrc/gemm.h): `streamk_ws` = gget_op_gemm_streamk_bytes() of device memory zeroed ONCE by
 * the caller and reusable by later calls on the same stream; launches whose tile count is no multiple of the CU count then cut the
 * K range of their boundary tiles between neighbouring workgroups (fp32 partial tiles through the workspace).  The engine does this
 * by itself with a slice of its workspace arena. */
int gget_op_gemm_streamk(int mode, int epilogue, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                         int lda, int ldb, int ldc, void* streamk_ws, void* stream);
uint64_t gget_op_gemm_streamk_bytes(void);
/* up to 4 independent GEMMs of one mode in ONE persistent launch (plain bf16 epilogue): how the engine issues the four
 * weight gradients of a decoder layer (dW = dY^T X for gate|up, down, q|k|v, o: hf LlamaMLP / LlamaAttention Linear backward) -
 * exactly 256 tiles of 192x192 for d = 768, one per CU */
int gget_op_gemm_grouped(int mode, int count, const void* const* A, const void* const* B, void* const* C, const int* M,
                         const int* N, const int* K, const int* lda, const int* ldb, const int* ldc, void* stream);
/* measurement knob (tools/ and tests; no reference counterpart).  key 1 = bit mask that switches GEMM kernel variants OFF, so that
 * two selections can be timed interleaved in one process or pinned to the same summation order (0 = the shipped selection): 1 K-split
 * kernel for one-round N = d launches, 2 K-split kernel for the grouped weight gradients, 4 the 192-row tiles, 8 the split of the
 * last round, 16 the 64- / 96-row tiles of the K-split kernel, 32 two co-resident workgroups per CU for the dh + GEGLU' launch.  key 2 = LDS headroom of the 128x192 tile (0: 4-slot ring).
 * key 3 = 1: split the K range of the last, partial round's tiles among the idle workgroups (off by default).
 * key 4 = 1 (also env GGET_DETERMINISTIC=1): reproducible mode of the pre-train step - the RMSNorm weight gradients, the one sum of that
 * gradient path added with fp32 atomics, are summed in block order instead; two runs then produce bit-identical parameters.
 * key 10 = 1: the per-sample kernels of S <= 32 off (the three launches each replaces run).  key 11 = 1: the fused RMSNorm + LayerScale
 * backward in its 16-byte-chunk form for every width (0: the 8-byte, all-lanes form for d = 512 / 768 / 1024).
 * key 13 = 0: the RMSNorm backward of short launches (<= 64 rows per CU) in 4-wave blocks as everywhere else (1, default: one 16-wave block
 * per CU, same bits).  key 14 = 0: the engine's cross-entropy launch adds its loss with one atomic per block (1, default: one partial sum
 * per block, summed in block order by the finalising launch).  key 15 = R: every GEMM launch plan (tile shapes, persistent grids, split-K
 * fits) counts the device's CUs minus R - data-parallel runs leave R CUs to the collective library's workgroups, which cannot share a CU
 * with a GEMM workgroup (0, default: the whole chip).  key 16 = 1: gget_debug_occupy's stand-in takes the register footprint of RCCL's
 * kernel (264 registers per lane) besides the LDS asked for. */
int gget_debug_set(int key, int value);
/* measurement aid: with enable != 0 the engine brackets, with HIP events on the launch stream, the grouped weight-gradient launch
 * (avg_ms_out[0]) and the gate|up + GEGLU launch (avg_ms_out[1]) of every layer of the following forward / backward calls;
 * avg_ms_out (may be NULL) receives the mean durations recorded so far.  bench.py uses it for the roofline of the dominant kernel
 * as it runs INSIDE a step. */
int gget_debug_probe(gget_handle_t h, int enable, float* avg_ms_out);
/* measurement aid behind bench.py's time-weighted GEMM figure (no reference counterpart).  enable = 1: start recording every GEMM
 * launch of the process between HIP events; enable = 0: stop and report the sum of the algorithmic FLOPs (2 M N K) and of the
 * durations of the recorded launches, how many were summed and how many were left out (row / K counts that live on the device) */
int gget_debug_gemm_probe(int enable, double* flops_out, double* ms_out, int* launches_out, int* skipped_out);
/* measurement aid (tools/coresidency.py; no reference counterpart): occupies `blocks` CU slots (256 threads, lds_bytes of LDS
 * each) for ~microseconds on `stream`, as a stand-in for a collective's kernel running beside the compute stream */
int gget_debug_occupy(void* scratch, uint64_t scratch_bytes, int blocks, int lds_bytes, int microseconds, void* stream);
/* replaces: q_proj/k_proj/v_proj + apply_rotary_pos_emb (hf LlamaAttention.forward :253-262, :138-160) as ONE GEMM:
 * qkv[T,3d] = x[T,d] * wqkv[3d,d]^T with RoPE applied to the q|k columns in the fp32 accumulators (position of row t is
 * position_ids[t], or t % S when position_ids is NULL; cos/sin tables [max_position][32] fp32). */
int gget_op_qkv_rope(const void* x, const void* wqkv, void* qkv, const float* cos_tab, const float* sin_tab,
                     const int64_t* position_ids, int T, int S, int d, void* stream);
/* replaces: prepare_for_2d_smtp_inputs_labels (src/models/graphgpt/modeling_helpers.py:399-449) as called by
 * GraphGPTPretrainBase.forward when config.smtp_inside (modeling_pretrain.py:175-189): per-sample mask rate r ~ U, a cell
 * (node, feature) is masked when U > r^power (looked up through node_idx), masked ids -> 1, labels = original id or -100,
 * optional replacement of masked cells by id + round(10 * N(0,1)) mod vocab.  ids_in [B,S,ld_in] (first F columns are
 * used), node_idx [B,S] with element stride ld_node, outputs dense [B,S,F].  Draws are a counter hash of `seed`
 * (graph-gpt_amd/smtp.py is the bit-exact Python twin). */
int gget_op_smtp2d(const int64_t* ids_in, int ld_in, const int64_t* node_idx, int ld_node, int64_t* ids_out, int64_t* labels_out,
                   int B, int S, int F, float smtp_2d_rate, float power, float replace_rate, int vocab, int global_2d_mask,
                   uint32_t seed, void* stream);
/* replaces: sample_tokens at temperature 0 (src/utils/generation_utils.py:45-82) inside the unmasking loop
 * (_batch_unmask_without_for_loop :139-237): per row of bf16 logits [R, ld] (V valid columns) the arg-max token and its
 * confidence: mode 0 max softmax probability ("maskgit_plus"), 1 top1 - top2 probability ("topk_margin"),
 * 2 sum p log(p + 1e-10) ("entropy").  Ties resolve to the lowest index. */
#define GGET_CONF_MAXPROB 0
#define GGET_CONF_MARGIN 1
#define GGET_CONF_NEG_ENTROPY 2
int gget_op_token_confidence(const void* logits, int ld, int R, int V, int mode, float* conf, int64_t* tok, void* stream);
/* replaces: sample_tokens with every option (src/utils/generation_utils.py:22-82: temperature, top-p, top-k, categorical
 * sampling, margin / entropy confidence) and the Gumbel-max perturbation of the confidence ranking (:199-209, alg_temp).
 * temperature 0 = arg-max; top_p outside (0,1) and top_k 0 = filter off; alg_temp 0 = no perturbation.  Draws: counter hash
 * of (seed, stream, row) (graph-gpt_amd/generation.py holds the Python twin); V <= 8192. */
int gget_op_token_sample(const void* logits, int ld, int R, int V, int mode, float temperature, float top_p, int top_k,
                         float alg_temp, uint32_t seed, float* conf, int64_t* tok, void* stream);
/* replaces: the alg = "origin" update of _batch_unmask_without_for_loop (src/utils/generation_utils.py:150-162): x[b][n] takes
 * cand[b][n] where it is <mask> and the cell's own uniform draw is below p_transfer. */
int gget_op_unmask_origin(int64_t* x, const int64_t* cand, int B, int N, float p_transfer, uint32_t seed, int mask_token_id,
                          void* stream);
/* replaces: the collator's SMTP masking (prepare_inputs_for_pretrain_mlm, src/utils/tokenizer_utils.py:259-271 polynomial
 * schedule + _mask_stacked_input_ids_v2 :112-148, mtp (1,0,0)) for a right-padded batch ids [B,S,F] with lengths [B]:
 * per sample t = umr_min + (umr_max - umr_min) U, exactly ceil(len*F*(1 - t^power)) of its cells are masked (id -> 1
 * unless pad, label = original id; all other labels -100); optional wgt_out[b] = power / t (dlm_wgt).  Draws are a
 * counter hash of `seed` (graph-gpt_amd/smtp.py holds the bit-exact twin). */
int gget_op_smtp_rows(const int64_t* ids_in, const int32_t* lengths, int64_t* ids_out, int64_t* labels_out, float* wgt_out, int B,
                      int S, int F, double umr_min, double umr_max, double power, uint32_t seed, void* stream);
int gget_op_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int T, int d, float eps, void* stream);
int gget_op_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                        void* dx, float* dw_accum, int T, int d, void* stream);
int gget_op_embed_fwd(const int64_t* ids, const void* emb, const void* gate, void* out, int T, int F, int ldF,
                      int d, void* stream);
int gget_op_embed_bwd(const int64_t* ids, const void* dx, const void* emb, const void* gate, float* demb_accum,
                      float* dgate_accum, int T, int F, int ldF, int d, int V, int pad_id, void* stream);
int gget_op_rope(void* qkv, const float* cos_tab, const float* sin_tab, const int64_t* position_ids, int B, int S,
                 int H, int inverse, void* stream);
/* cos_tab/sin_tab ([max_pos][32] fp32, hf LlamaRotaryEmbedding tables) non-NULL: RoPE (hf apply_rotary_pos_emb
 * :138-160) is applied to q,k inside the kernels and undone on dq,dk; qkv / dqkv are the UN-rotated projections.
 * dropout_p > 0: attention dropout on the softmax output (hf eager_attention_forward :210), counter-based mask
 * keyed by (dropout_seed, batch*head, query, key) - see drop_mul() in csrc/attention.hip. */
int gget_op_attn_fwd(const void* qkv, const int32_t* key_len, void* out, float* lse, int B, int S, int H,
                     int causal, const float* cos_tab, const float* sin_tab, const int64_t* position_ids,
                     float dropout_p, uint32_t dropout_seed, void* stream);
int gget_op_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len,
                     void* dqkv, float* delta_ws, int B, int S, int H, int causal, const float* cos_tab,
                     const float* sin_tab, const int64_t* position_ids, float dropout_p, uint32_t dropout_seed,
                     void* stream);
/* replaces: the `.to(device)` calls at the top of the reference's step (src/utils/training_utils.py:17-26) for a batch that sits in PINNED
 * host memory (hipHostMalloc / torch pin_memory: device-mapped): a kernel on `stream` reads it over the host link into dst_dev.  Stays in
 * the compute queue - an in-stream hipMemcpyAsync from pinned memory is a copy-engine job whose dependencies the runtime resolves on the
 * host (four per step: the C1 step 6.7 -> 12 - 20 ms).  Both pointers 16-byte aligned; the host buffer must stay untouched until the
 * kernel has run (record an event behind it).  graph-gpt_amd/training.py DevicePrefetcher is the caller. */
int gget_op_copy_from_host(const void* src_pinned_host, void* dst_dev, uint64_t bytes, void* stream);
/* The two above on the padding-free (var-len) token layout: row_base (int32 [B]) = first row of sample b in the token-major buffers
 * (qkv / out / dout / dqkv hold the samples' real rows back to back, key_len[b] of them for sample b); lse / delta keep their [B, H, S]
 * indexing, S = the padded width the collator produced (8 * ceil(longest graph / 8), reference src/data/collator.py:70-111).  Every
 * sample is processed by its OWN row count: blocks behind a sample's rows exit at once, and the backward for S <= 64 is one launch whose
 * waves take one 32-row tile or - a sample of 33 .. 64 rows - 2 x 2 tiles (csrc/attention.hip: attn_bwd_small_kernel).  qk_rotated != 0:
 * q and k in qkv are rotated already (the engine's layout); the tables then only rotate dq / dk back. */
int gget_op_attn_fwd_varlen(const void* qkv, const int32_t* key_len, const int32_t* row_base, void* out, float* lse, int B, int S, int H,
                            int causal, const float* cos_tab, const float* sin_tab, const int64_t* position_ids, int qk_rotated,
                            float dropout_p, uint32_t dropout_seed, void* stream);
int gget_op_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len,
                            const int32_t* row_base, void* dqkv, float* delta_ws, int B, int S, int H, int causal, const float* cos_tab,
                            const float* sin_tab, const int64_t* position_ids, int qk_rotated, float dropout_p, uint32_t dropout_seed,
                            void* stream);
/* S <= 32 (graph sequences of PCQM4M-v2): attention of every head of a sample, the o projection + residual add and the RMSNorm behind
 * it in ONE launch, one workgroup per sample (csrc/attention.hip: attn_oproj_fwd_kernel).  replaces, for one decoder layer:
 * hf LlamaAttention.forward :243-281 (attention on rotated q / k + o_proj), LlamaDecoderLayer.forward :305-316 (residual add,
 * post_attention_layernorm = LlamaRMSNorm.forward :62-67).  wo_packed: the FRAGMENT-MAJOR copy of o_proj.weight [d][d] written by
 * gget_op_pack_wo (its `fwd` output; the 64 lanes' 16-byte pieces of one 16 x 32 MFMA operand are contiguous - in the weight's row-major
 * layout a fragment load is 64 separate 16-byte requests and the launch runs at 10 B/clk per CU).
 * qkv [rows, 3 * 64 H] bf16 with q / k ALREADY rotated (the engine's layout);
 * row_base (int32 [B], may be NULL): first row of sample b in the token-major buffers (var-len layout; then key_len[b] rows belong to it),
 * NULL = padded layout, sample b at rows [b S, b S + S).  Outputs: attn_out [rows, d] bf16, lse fp32 [B, H, S] (natural log), x_mid =
 * x_in + attn_out Wo^T (bf16, residual added on the fp32 accumulator: one rounding), xn = norm_w * bf16(x_mid * rstd) and rstd fp32 [rows].
 * *taken = 1 when the fused form ran; 0 when the shape is not covered (S > 32, H not in {2, 4, 8, 12}, GGET_ATTN_OPROJ=0) - nothing
 * is written then and the caller runs gget_op_attn_fwd, a GEMM and gget_op_rmsnorm_fwd. */
int gget_op_attn_oproj_fwd(const void* qkv, const int32_t* key_len, const int32_t* row_base, void* attn_out, float* lse, const void* wo_packed,
                           const void* x_in, void* x_mid, const void* norm_w, void* xn, float* rstd, int B, int S, int H, int causal,
                           float eps, float dropout_p, uint32_t dropout_seed, void* stream, int32_t* taken);
/* fwd[((T KS + s) 64 + lane) 8 + e] = w[16 T + lane % 16][32 s + 8 (lane / 16) + e], KS = d / 32; bwd: the same of w transposed.
 * `layers` weights, `layer_stride` elements apart in w, are packed back to back ([layers][d][d] each output).  d % 64 == 0. */
int gget_op_pack_wo(const void* w, uint64_t layer_stride, void* fwd, void* bwd, int d, int layers, void* stream);
/* The backward counterpart of gget_op_attn_oproj_fwd, one workgroup per sample (csrc/attention.hip: attn_oproj_bwd_kernel): RMSNorm
 * backward of post_attention_layernorm - dx_mid = dres + rstd (dxn w - xhat mean(dxn w xhat)), dw_accum[j] += sum_rows dxn xhat (fp32,
 * `copies` replicas `copy_stride` floats apart, as gget_op_rmsnorm_bwd's) - then dattn = dx_mid Wo (never written to memory) and the
 * attention backward of every head: dqkv [rows, 3 d] with dq / dk rotated BACK by cos_tab / sin_tab / position_ids (NULL tables: no
 * rotation).  wot_packed = gget_op_pack_wo's `bwd` output.  t_rows = rows of the token-major buffers (var-len layout: dx_mid of the pad
 * rows behind the last sample is zeroed).  *taken as above; also 0 in the reproducible mode (gget_debug_set(4, 1)).
 * Var-len layout (row_base != NULL) with 32 < S <= 64: every sample by its own row count.  A sample of 33 .. 64 rows takes the first two
 * steps in its workgroup, over two row tiles; its dattn rows go to dattn_long (bf16 [t_rows, d]; rows of other samples are not touched;
 * NULL: the launch is not taken for S > 32) and a second launch on the same stream (attn_bwd_long_kernel) does its attention backward. */
int gget_op_attn_oproj_bwd(const void* dxn, const void* x_mid, const void* norm_w, const float* rstd, const void* dres, void* dx_mid,
                           float* dw_accum, int copies, uint64_t copy_stride, const void* wot_packed, const void* qkv, const float* lse,
                           const int32_t* key_len, const int32_t* row_base, void* dqkv, int B, int S, int H, int causal, const float* cos_tab,
                           const float* sin_tab, const int64_t* position_ids, float dropout_p, uint32_t dropout_seed, int t_rows,
                           void* stream, int32_t* taken, void* dattn_long);
/* attention with a per-token inclusive key range [key_lo, key_hi] (int32 [B,S]; packed rows) instead of one length per
 * batch row; gget_op_ranges_from_mask3d derives the ranges from a block-diagonal int64 [B,S,S] mask. */
int gget_op_attn_fwd_ranges(const void* qkv, const int32_t* key_lo, const int32_t* key_hi, void* out, float* lse, int B, int S,
                            int H, int causal, float dropout_p, uint32_t dropout_seed, void* stream);
int gget_op_attn_bwd_ranges(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_lo,
                            const int32_t* key_hi, void* dqkv, float* delta_ws, int B, int S, int H, int causal,
                            float dropout_p, uint32_t dropout_seed, void* stream);
int gget_op_ranges_from_mask3d(const int64_t* mask3d, int32_t* key_lo, int32_t* key_hi, int B, int S, void* stream);
/* gget_op_attn_bwd / _ranges (no RoPE) through the ONE-PASS long-sequence backward (S >= 256; shorter rows run the usual kernels and leave
 * the workspace alone): S, dP and the softmax backward are evaluated once, dK / dV and dQ come out of the same pass (5 matmuls of hf
 * eager_attention_forward's autograd graph, modeling_llama.py:191-214, instead of the 7 of the two-kernel form).  dq_ws: bf16
 * [ceil(S / 256)][B * S][H * 64] scratch (contents irrelevant on entry): every block of 256 keys writes its dQ partials into its own
 * slab, the last launch sums the slabs in fp32 in block order - reproducible - into the q part of dqkv.
 * key_lo / key_hi both NULL: one key length per row (key_len, may be NULL = S). */
int gget_op_attn_bwd_fused(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len,
                           const int32_t* key_lo, const int32_t* key_hi, void* dqkv, float* delta_ws, void* dq_ws, int B, int S,
                           int H, int causal, float dropout_p, uint32_t dropout_seed, void* stream);
/* replaces: LlamaMLP.forward (hf :174-176) up to the down projection, as ONE GEMM with the gated-GELU product in its
 * epilogue: gu[T,2ff] = x[T,d] * wgu[2ff,d]^T (gate | up pre-activations, kept for the backward),
 * h[T,ff] = bf16(gelu(gate)) * up.  Falls back to GEMM + gget_op_geglu_fwd when ff % 128 != 0. */
int gget_op_gateup_geglu(const void* x, const void* wgu, void* gu, void* h, int T, int d, int ff, void* stream);
/* backward counterpart: dgu[T,2ff] = (dh * up * gelu'(gate) | dh * gelu(gate)) with dh = dy[T,d] * wdown[d,ff] computed in
 * the same launch and never stored (dh_scratch [T,ff] is only used by the un-fused fallback, may be NULL when ff % 128 == 0) */
int gget_op_down_dgrad_geglu(const void* dy, const void* wdown, const void* gu, void* dgu, void* dh_scratch, int T, int d, int ff,
                             void* stream);
int gget_op_geglu_fwd(const void* gu, void* h, int T, int ff, void* stream);
int gget_op_geglu_bwd(const void* gu, const void* dh, void* dgu, int T, int ff, void* stream);
int gget_op_ce_fwd_bwd(const void* logits, int ld, const int32_t* labels, const float* row_wgt, const int32_t* n_rows_dev,
                       int n_rows_cap, int V, float* loss_sum, void* dlogits, float grad_scale_base, int mean_over_rows,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GGET_H_ */
