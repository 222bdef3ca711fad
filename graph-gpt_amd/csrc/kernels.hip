// HBM-bound kernels of the Graph Eulerian Transformer path (everything that is not a GEMM or the
// attention core): stacked-token embedding gather / scatter-add, RMSNorm, RoPE, GEGLU, SMTP head
// compaction + cross-entropy, fine-tune score head, AdamW.  All bf16 I/O with fp32 math, 16-byte
// vector accesses, wave64 reductions.  Reference sites are cited per kernel.
#include "common.h"
#include "kernels.h"

static int g_deterministic = getenv("GGET_DETERMINISTIC") ? atoi(getenv("GGET_DETERMINISTIC")) : 0;   // k_set_deterministic
static int g_rms_bwd_wide = getenv("GGET_RMS_WIDE") ? atoi(getenv("GGET_RMS_WIDE")) : 1;   // k_set_rms_wide (gget_debug_set key 13)
static int g_ce_parts = getenv("GGET_CE_PARTS") ? atoi(getenv("GGET_CE_PARTS")) : 1;         // k_set_ce_parts (gget_debug_set key 14)
static float* g_det_scratch = nullptr;
static size_t g_det_bytes = 0;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stg16(bf16_t* p, const uint4& v) {   // non-temporal (see stc16 in gemm.hip)
  typedef unsigned u4nt __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u4nt*>(p));
}

// ---------------------------------------------------------------------------------------------
// K1  stacked-token embedding: E[t,:] = sum_f W[ids[t,f],:] (* G[f,:])
// reference: _get_stacked_inputs_embeds (modeling_helpers.py:89-114), StackedFeatAggregation
// (modeling_common.py:127-135).  One 128-thread block per token, 8 channels per thread.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) embed_fwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ emb,
                                                        const bf16_t* __restrict__ gate, bf16_t* __restrict__ out,
                                                        int T, int F, int ldF, int d, ElemDropArg E) {
  const int t = blockIdx.x;
  const int64_t* row = ids + (size_t)t * ldF;
  for (int c = threadIdx.x; c * 8 < d; c += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // four gathered rows in flight at a time (unconditional loads of a clamped feature index; the sum keeps the feature order):
    // one row per loop trip was a chain of F dependent L2 round trips per thread
    for (int f0 = 0; f0 < F; f0 += 4) {
      uint4 r[4], gr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int fu = min(f0 + u, F - 1);
        r[u] = ldg16(emb + (size_t)row[fu] * d + c * 8);
        if (gate) gr[u] = ldg16(gate + (size_t)fu * d + c * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + u;
        if (f >= F) break;
        float v[8];
        unpack8(r[u], v);
        if (E.thresh) {   // embed_dropout acts on the gathered rows, before the stacking (modeling_helpers.py:96-101)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = bf2f(f2bf(v[e] * elem_drop_mul(E, GGET_DROP_STREAM_EMBED, elem_row(E, t) * (unsigned)F + (unsigned)f, (unsigned)(c * 8 + e))));
        }
        if (gate) {
          float gv[8];
          unpack8(gr[u], gv);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e] * gv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
      }
    }
    stg16(out + (size_t)t * d + c * 8, pack8(acc));
  }
}

// stack_method = "long" (modeling_helpers.py:106-110): the stacked embedding of a token is divided by the number of its non-zero
// feature ids - ratio = min(1, bf16(1 / (nnz + 1e-7))), the reference's fp32 reciprocal cast to the activation dtype and
// clamped, and the product rounded to bf16 once more.  The multiplier is also the backward of the product (autograd rounds
// grad * ratio to bf16 the same way), so this one in-place kernel serves the forward and the gradient.
__global__ void __launch_bounds__(128) embed_long_ratio_kernel(const int64_t* __restrict__ ids, bf16_t* __restrict__ x, int F,
                                                               int ldF, int d) {
  const int t = blockIdx.x;
  const int64_t* row = ids + (size_t)t * ldF;
  int nnz = 0;
  for (int f = 0; f < F; ++f) nnz += row[f] != 0;
  const float ratio = fminf(bf2f(f2bf(1.0f / ((float)nnz + 1e-7f))), 1.0f);
  if (ratio == 1.0f) return;
  for (int c = threadIdx.x; c * 8 < d; c += blockDim.x) {
    float v[8];
    unpack8(ldg16(x + (size_t)t * d + c * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= ratio;
    stg16(x + (size_t)t * d + c * 8, pack8(v));
  }
}

// ... and its loss weights (_prepare_for_stacked_feat_labels_per_feat_lvl, modeling_helpers.py:327-342): every masked cell of
// sample b weighs 1 / (masked cells of b + 1e-7).  One block per sample.
__global__ void __launch_bounds__(kBlock) sample_mask_wgt_kernel(const int64_t* __restrict__ labels, float* __restrict__ w,
                                                                 int cells) {
  __shared__ int part[kBlock / GGET_WAVE];
  const int64_t* row = labels + (size_t)blockIdx.x * cells;
  int n = 0;
  for (int i = threadIdx.x; i < cells; i += kBlock) n += row[i] != -100;
  n = (int)wave_sum((float)n);   // (exact: at most 2^24 cells per sample)
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int i = 0; i < kBlock / GGET_WAVE; ++i) tot += part[i];
    w[blockIdx.x] = 1.0f / ((float)tot + 1e-7f);
  }
}

// raw-embedding inputs (config.embed_dim > 0): the `inputs_raw_embeds.to(dtype)` cast and the mask-token blend of
// GraphGPTPretrainBase.prepare_inputs_embeds (modeling_pretrain.py:131-143): embed_mask = (labels == -100).sum(-1).bool() keeps the raw row
// when ANY of the token's labels is unset (smtp_inside: when its first label is unset); the other rows become emb_mask_token.
__global__ void __launch_bounds__(128) raw_blend_kernel(const float* __restrict__ raw, const int64_t* __restrict__ labels, int n,
                                                        int first_only, const bf16_t* __restrict__ tok, bf16_t* __restrict__ out,
                                                        int32_t* __restrict__ flag, int e, const int32_t* __restrict__ rows_map, int n_logical) {
  // var-len token layout (rows_map): out / flag rows are the compact rows, raw and labels are read at the logical row rows_map[t]; the
  // pad rows that round the row count up (logical row >= n_logical) are zeros, unflagged
  const int t = blockIdx.x;
  const int lt = rows_map ? rows_map[t] : t;
  if (lt >= n_logical) {
    if (threadIdx.x == 0) flag[t] = 0;
    for (int j = threadIdx.x; j < e; j += blockDim.x) out[(size_t)t * e + j] = 0;
    return;
  }
  bool masked = false;
  if (labels) {
    const int64_t* lr = labels + (size_t)lt * n;
    int unset = 0;
    for (int f = 0; f < (first_only ? 1 : n); ++f) unset += lr[f] == -100;
    masked = unset == 0;
  }
  if (threadIdx.x == 0) flag[t] = masked ? 1 : 0;
  for (int j = threadIdx.x; j < e; j += blockDim.x) out[(size_t)t * e + j] = masked ? tok[j] : f2bf(raw[(size_t)lt * e + j]);
}
__global__ void __launch_bounds__(kBlock) raw_tok_grad_kernel(const bf16_t* __restrict__ dx, const int32_t* __restrict__ flag,
                                                              float* __restrict__ dtok, int T, int e) {
  const int t0 = blockIdx.x * 256, t1 = min(T, t0 + 256);
  for (int j = threadIdx.x; j < e; j += kBlock) {
    float s = 0.f;
    for (int t = t0; t < t1; ++t)
      if (flag[t]) s += bf2f(dx[(size_t)t * e + j]);
    if (s != 0.f) unsafeAtomicAdd(dtok + j, s);
  }
}

// backward of K1: dW[v,:] = sum over cells (t,f) with ids[t,f]==v of dx[t,:] (* G[f,:]).
// SMTP batches hit a few hundred vocabulary rows with ~10^5 cells (half of them the <mask> row), so a direct
// atomic scatter serialises on hot rows.  Instead: counting sort of the cells by id on the device
// (histogram -> single-block scan -> fill), then every block walks 128 consecutive SORTED cells, accumulates each
// run of equal ids in registers and flushes one fp32 atomic per run: ~(cells/128 + distinct ids) x d atomics.
// Both passes pre-aggregate per block in LDS (vocabularies up to kEmbLdsV ids) so the hot <mask> row costs one
// global atomic per block instead of one per cell; larger vocabularies aggregate the hot id per wave with a ballot.
constexpr int kEmbLdsV = 8192;   // 2*V*4 B of dynamic LDS in the fill pass stays <= 64 KiB
constexpr int kEmbCells = 4096;  // cells per block
template <bool LDS_PATH>
__global__ void __launch_bounds__(256) embed_hist_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ hist,
                                                         long ncell, int F, int ldF, int pad_id, int V, int hot_id) {
  extern __shared__ int lh[];
  const long c0 = (long)blockIdx.x * kEmbCells;
  if (LDS_PATH) {
    for (int v = threadIdx.x; v < V; v += 256) lh[v] = 0;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < kEmbCells; i += 256) {
    const long c = c0 + i;
    const bool in = c < ncell;
    const int id = in ? (int)ids[(c / F) * ldF + (c % F)] : pad_id;
    if (LDS_PATH) {
      if (id != pad_id) atomicAdd(lh + id, 1);
    } else {
      const unsigned long long hot = __ballot(id == hot_id && id != pad_id);
      if (id == hot_id) {
        if ((threadIdx.x & 63) == __builtin_ctzll(hot)) atomicAdd(hist + id, (int)__builtin_popcountll(hot));
      } else if (id != pad_id) {
        atomicAdd(hist + id, 1);
      }
    }
  }
  if (LDS_PATH) {
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += 256)
      if (lh[v]) atomicAdd(hist + v, lh[v]);
  }
}
// exclusive scan of hist[V] -> offs[V] (+ total in offs[V]); cursor[v] = offs[v]
__global__ void __launch_bounds__(1024) embed_scan_kernel(const int32_t* __restrict__ hist, int32_t* __restrict__ offs,
                                                          int32_t* __restrict__ cursor, int V) {
  __shared__ int sm[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < V; base += 1024) {
    const int v = base + tid;
    const int c = v < V ? hist[v] : 0;
    sm[tid] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int a = 0;
      if (tid >= o) a = sm[tid - o];
      __syncthreads();
      sm[tid] += a;
      __syncthreads();
    }
    if (v < V) { const int e = carry + sm[tid] - c; offs[v] = e; cursor[v] = e; }
    __syncthreads();
    if (tid == 1023) carry += sm[1023];
    __syncthreads();
  }
  if (tid == 0) offs[V] = carry;
}
template <bool LDS_PATH>
__global__ void __launch_bounds__(256) embed_fill_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ cursor,
                                                         int32_t* __restrict__ cell_sorted, int32_t* __restrict__ id_sorted,
                                                         long ncell, int F, int ldF, int pad_id, int V, int hot_id) {
  extern __shared__ int lh[];  // LDS_PATH: local counts [V] then global bases [V]
  const long c0 = (long)blockIdx.x * kEmbCells;
  if (LDS_PATH) {
    int* base = lh + V;
    for (int v = threadIdx.x; v < V; v += 256) lh[v] = 0;
    __syncthreads();
    int rank[kEmbCells / 256], idv[kEmbCells / 256];
#pragma unroll
    for (int k = 0; k < kEmbCells / 256; ++k) {
      const long c = c0 + threadIdx.x + k * 256;
      idv[k] = c < ncell ? (int)ids[(c / F) * ldF + (c % F)] : pad_id;
      rank[k] = idv[k] != pad_id ? atomicAdd(lh + idv[k], 1) : 0;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += 256)
      if (lh[v]) base[v] = atomicAdd(cursor + v, lh[v]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kEmbCells / 256; ++k) {
      if (idv[k] == pad_id) continue;
      const int pos = base[idv[k]] + rank[k];
      cell_sorted[pos] = (int)(c0 + threadIdx.x + k * 256);
      id_sorted[pos] = idv[k];
    }
  } else {
    for (int i = threadIdx.x; i < kEmbCells; i += 256) {
      const long c = c0 + i;
      const int id = c < ncell ? (int)ids[(c / F) * ldF + (c % F)] : pad_id;
      const unsigned long long hot = __ballot(id == hot_id && id != pad_id);
      int pos = -1;
      if (id == hot_id && id != pad_id) {
        const int lane = threadIdx.x & 63;
        const int leader = __builtin_ctzll(hot);
        int b0 = 0;
        if (lane == leader) b0 = atomicAdd(cursor + id, (int)__builtin_popcountll(hot));
        b0 = __shfl(b0, leader, 64);
        pos = b0 + (int)__builtin_popcountll(hot & ((1ull << lane) - 1ull));
      } else if (id != pad_id) {
        pos = atomicAdd(cursor + id, 1);
      }
      if (pos >= 0) { cell_sorted[pos] = (int)c; id_sorted[pos] = id; }
    }
  }
}
constexpr int kEmbSeg = 128;
__global__ void __launch_bounds__(128) embed_reduce_kernel(const int32_t* __restrict__ cell_sorted,
                                                           const int32_t* __restrict__ id_sorted,
                                                           const int32_t* __restrict__ offs, int V,
                                                           const bf16_t* __restrict__ dx, const bf16_t* __restrict__ gate,
                                                           float* __restrict__ demb, int F, int d, ElemDropArg E) {
  const int n = offs[V];
  const int beg = blockIdx.x * kEmbSeg;
  if (beg >= n) return;
  const int end = min(n, beg + kEmbSeg);
  for (int c = threadIdx.x; c * 8 < d; c += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cur = id_sorted[beg];
    // a run of equal ids that lies inside this segment is the ONLY contribution of this launch to its row of the accumulator: a
    // plain 32-byte read-add-store; only the runs cut by a segment boundary need atomics (large vocabularies: most runs are a few cells, and
    // 768 fp32 atomics per run were the whole cost of this kernel at V = 41 245)
    auto flush = [&](int id, const float (&a)[8]) {
      float* dst = demb + (size_t)id * d + c * 8;
      if (offs[id] >= beg && offs[id + 1] <= end) {      // (read-add-store: the entry point accumulates into demb)
        float4 lo = *reinterpret_cast<const float4*>(dst), up = *reinterpret_cast<const float4*>(dst + 4);
        lo.x += a[0]; lo.y += a[1]; lo.z += a[2]; lo.w += a[3];
        up.x += a[4]; up.y += a[5]; up.z += a[6]; up.w += a[7];
        *reinterpret_cast<float4*>(dst) = lo;
        *reinterpret_cast<float4*>(dst + 4) = up;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) unsafeAtomicAdd(dst + e, a[e]);
      }
    };
    // the rows of one segment are independent loads: fetch them eight at a time before the (ordered) accumulation
    for (int j0 = beg; j0 < end; j0 += 8) {
      int idv[8], cellv[8];
      uint4 raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = min(j0 + u, end - 1);
        idv[u] = id_sorted[j];
        cellv[u] = cell_sorted[j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[u] = ldg16(dx + (size_t)(cellv[u] / F) * d + c * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u >= end) break;
        if (idv[u] != cur) {
          flush(cur, acc);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = 0.f;
          cur = idv[u];
        }
        float g[8];
        unpack8(raw[u], g);
        if (E.thresh) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] *= elem_drop_mul(E, GGET_DROP_STREAM_EMBED, elem_row(E, cellv[u] / F) * (unsigned)F + (unsigned)(cellv[u] % F), (unsigned)(c * 8 + e));
        }
        if (gate) {
          float gv[8];
          unpack8(ldg16(gate + (size_t)(cellv[u] % F) * d + c * 8), gv);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += g[e] * gv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += g[e];
        }
      }
    }
    flush(cur, acc);
  }
}
// Small vocabularies, plain (un-gated) stacking: the scatter-add is the dense product dE = C^T dX with the count matrix
// C[t][v] = #{f : ids[t][f] == v} (pad id excluded) - small integers, exact in bf16.  One thread per cell writes the count of
// its id inside its token (duplicates all write the same value, so no read-modify-write); the caller clears C.
__global__ void __launch_bounds__(kBlock) embed_count_kernel(const int64_t* __restrict__ ids, bf16_t* __restrict__ cnt, int T,
                                                             int F, int ldF, int ldc, int pad_id) {
  const long cell = (long)blockIdx.x * kBlock + threadIdx.x;   // one thread per cell; the token's other ids come from L1
  if (cell >= (long)T * F) return;
  const int t = (int)(cell / F), f = (int)(cell - (long)t * F);
  const int64_t* row = ids + (size_t)t * ldF;
  const int id = (int)row[f];
  if (id == pad_id) return;
  int c = 0;
  for (int g = 0; g < F; ++g) c += (int)row[g] == id ? 1 : 0;
  cnt[(size_t)t * ldc + id] = f2bf((float)c);
}

// gated stacking only: dG[f,:] = sum_t dx[t,:] * W[ids[t,f],:]  (block-local accumulation, one atomic per block)
constexpr int kEmbTok = 64;
__global__ void __launch_bounds__(128) embed_dgate_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ dx,
                                                          const bf16_t* __restrict__ emb, float* __restrict__ dgate, int T,
                                                          int F, int ldF, int d, ElemDropArg E) {
  const int t0 = blockIdx.x * kEmbTok;
  const int f = blockIdx.y;
  for (int c = threadIdx.x; c * 8 < d; c += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int tt = 0; tt < kEmbTok && t0 + tt < T; ++tt) {
      const int t = t0 + tt;
      float g[8], ev[8];
      unpack8(ldg16(dx + (size_t)t * d + c * 8), g);
      unpack8(ldg16(emb + (size_t)ids[(size_t)t * ldF + f] * d + c * 8), ev);
      if (E.thresh) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          ev[e] = bf2f(f2bf(ev[e] * elem_drop_mul(E, GGET_DROP_STREAM_EMBED, elem_row(E, t) * (unsigned)F + (unsigned)f, (unsigned)(c * 8 + e))));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += g[e] * ev[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) unsafeAtomicAdd(dgate + (size_t)f * d + c * 8 + e, acc[e]);
  }
}

// ---------------------------------------------------------------------------------------------
// K4  RMSNorm (hf LlamaRMSNorm.forward :62-67): y = w * bf16(x * rsqrt(mean(x^2)+eps)); one wave per row.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxChunksPerLane = 4;  // d <= 2048
// NCH = 16-byte chunks per lane (2 covers d <= 1024).  A wave walks its rows with the NEXT row's loads already in flight
// while it reduces / scales / stores the current one (one row at a time the kernel ran at 3.2-3.8 TB/s).
template <int NCH>
__global__ void __launch_bounds__(kBlock) rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                             int T, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int nchunk = d >> 3;
  const int stride = gridDim.x * (kBlock / 64);
  int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  uint4 cur[NCH], nxt[NCH];
  float wv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    cur[i] = make_uint4(0, 0, 0, 0);
    if (c < nchunk) {
      unpack8(ldg16(w + c * 8), wv[i]);
      if (row < T) cur[i] = ldg16(x + (size_t)row * d + c * 8);
    }
  }
  for (; row < T; row += stride) {
    const int nrow = row + stride;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      nxt[i] = (c < nchunk && nrow < T) ? ldg16(x + (size_t)nrow * d + c * 8) : make_uint4(0, 0, 0, 0);
    }
    float v[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      unpack8(cur[i], v[i]);      // (chunks beyond the row are zero)
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = wv[i][e] * bf2f(f2bf(v[i][e] * rstd));
        stg16(y + (size_t)row * d + c * 8, pack8(o));
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) cur[i] = nxt[i];
  }
}

// backward: dx = dres + rstd*(dy*w - xhat*mean(dy*w*xhat)),  dw += dy*xhat  (fp32 accumulators)
// NCH = 16-byte chunks per lane (2 covers d <= 1024: fewer live registers => more waves to hide HBM latency).
template <int NCH>
__global__ void __launch_bounds__(kBlock) rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                             const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                             float* __restrict__ dw_accum, int T, int d, int copies,
                                                             uint64_t copy_stride, float* __restrict__ dw_part) {
  extern __shared__ float dw_lds[];  // [4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  float dwp[NCH][8];
  float wv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) dwp[i][e] = 0.f;
    if (c < nchunk) unpack8(ldg16(w + c * 8), wv[i]);
  }
  // software-pipelined over rows: the loads of the next row are in flight while the current one is reduced and stored
  const int stride = gridDim.x * (kBlock / 64);
  int row = blockIdx.x * (kBlock / 64) + wave;
  uint4 xr[NCH], dr[NCH], rr[NCH];
  float rstd = 0.f;
  auto fetch = [&](int r) {
    rstd = rstd_in[r];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        xr[i] = ldg16(x + (size_t)r * d + c * 8);
        dr[i] = ldg16(dy + (size_t)r * d + c * 8);
        rr[i] = dres ? ldg16(dres + (size_t)r * d + c * 8) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  if (row < T) fetch(row);
  for (; row < T; row += stride) {
    const float rs = rstd;
    float xh[NCH][8], g[NCH][8], res[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float xv[8], dv[8];
        unpack8(xr[i], xv);
        unpack8(dr[i], dv);
        unpack8(rr[i], res[i]);
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
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = res[i][e] + rs * (g[i][e] - xh[i][e] * dot);
        stg16(dx + (size_t)row * d + c * 8, pack8(o));
      }
    }
  }
  // block-level reduction of the dw partials, then one atomic per channel per block
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      // Two planes of d / 2 floats per wave: channels [8 c, 8 c + 4) in plane 0 at float 4 c, [8 c + 4, 8 c + 8) in plane 1 - neighbouring lanes
      // store neighbouring 16 bytes.  (Eight scalar stores 32 bytes apart per lane were 8-way bank conflicts: 0.75 conflict cycles per
      // active LDS cycle in the round-4 PMC table; two 16-byte stores at that pitch still 2-way.)
      *reinterpret_cast<float4*>(dw_lds + wave * d + c * 4) = make_float4(dwp[i][0], dwp[i][1], dwp[i][2], dwp[i][3]);
      *reinterpret_cast<float4*>(dw_lds + wave * d + (d >> 1) + c * 4) = make_float4(dwp[i][4], dwp[i][5], dwp[i][6], dwp[i][7]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < d; t += kBlock) {
    const int hd = d >> 1, pl = t >= hd ? 1 : 0, ix = t - pl * hd;      // thread -> (plane, float): consecutive threads read consecutive floats
    const int j = (ix >> 2) * 8 + pl * 4 + (ix & 3);                     // ... of channel j
    const float s = dw_lds[t] + dw_lds[d + t] + dw_lds[2 * d + t] + dw_lds[3 * d + t];
    if (dw_part) dw_part[(size_t)blockIdx.x * d + j] = s;     // reproducible mode: summed in block order by ordered_colsum_kernel
    else unsafeAtomicAdd(dw_accum + (size_t)(blockIdx.x % copies) * copy_stride + j, s);
  }
}

// SHORT launches (round 5).  The step's shapes are short - T = 5 696 rows at the headline, 22 rows per CU: with 4-wave blocks of 4 rows per
// wave only ~6 waves per CU had loads in flight (25 KB per CU where HBM latency x bandwidth wants ~47 KB: 3.2 TB/s), and fewer rows per wave
// multiplied the per-block atomics of the weight gradient.  Here: ONE block of NW = 16 waves per CU, the rows cut into contiguous ranges
// per block and dealt round-robin to its waves (1 - 2 rows each: everything a CU will read, 72 KB, is requested at once), one block-level
// reduction of the weight-gradient partials = 256 x d atomics per launch instead of 356 x d.  To fit 16 waves per CU (<= 128 registers)
// a row's operands stay PACKED in two register sets (the row being worked on, the row in flight) and are unpacked where they are used -
// once for the reductions, once more for the result; same expressions, same bits as rmsnorm_bwd_kernel.
template <int NCH, int NW>
__global__ void __launch_bounds__(NW * 64) rmsnorm_bwd_wide_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                                  const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                                  float* __restrict__ dw_accum, int T, int d, int copies,
                                                                  uint64_t copy_stride, float* __restrict__ dw_part) {
  extern __shared__ float dw_lds[];  // [NW][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  float dwp[NCH][8];
  uint4 wp[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) dwp[i][e] = 0.f;
    wp[i] = c < nchunk ? ldg16(w + c * 8) : make_uint4(0, 0, 0, 0);
  }
  struct RowRegs { uint4 x[NCH], d[NCH], r[NCH]; float rs; };
  RowRegs ra, rb;
  const unsigned lo = lane * 8;     // (32-bit lane offsets on a wave-uniform row base: the addresses stay out of the vector registers)
  auto fetch = [&](int r0, RowRegs& q) {
    const int r = __builtin_amdgcn_readfirstlane(r0);
    const size_t rbase = (size_t)r * d;
    const bf16_t *xrow = x + rbase, *dyrow = dy + rbase, *rrow = dres ? dres + rbase : nullptr;
    q.rs = rstd_in[r];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        q.x[i] = ldg16(xrow + lo + i * 512);
        q.d[i] = ldg16(dyrow + lo + i * 512);
        q.r[i] = rrow ? ldg16(rrow + lo + i * 512) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  // (`keep4`: an empty asm the compiler must assume rewrites the packed words, so that it unpacks them AGAIN where they are used instead
  //  of keeping the floats of the first pass alive across the reduction / the loop - that is what spilled)
  auto keep4 = [](uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
  auto work = [&](int row0, RowRegs& q) {
    const int row = __builtin_amdgcn_readfirstlane(row0);
    const float rs = q.rs;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float xv[8], dv[8], wv[8];
        keep4(wp[i]);
        unpack8(q.x[i], xv);
        unpack8(q.d[i], dv);
        unpack8(wp[i], wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = xv[e] * rs;
          const float g = dv[e] * wv[e];
          dot += g * xh;
          dwp[i][e] += dv[e] * xh;
        }
      }
    }
    dot = wave_sum(dot) / (float)d;
    bf16_t* dxrow = dx + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float xv[8], dv[8], wv[8], res[8], o[8];
        keep4(q.x[i]);
        keep4(q.d[i]);
        keep4(wp[i]);
        unpack8(q.x[i], xv);
        unpack8(q.d[i], dv);
        unpack8(wp[i], wv);
        unpack8(q.r[i], res);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = xv[e] * rs;
          float g = dv[e] * wv[e];
          asm volatile("" : "+v"(g));   // (the ROUNDED product, as in rmsnorm_bwd_kernel where it is shared with the first pass: without this
                                        //  the compiler may fuse THIS multiply into the subtraction instead of xh * dot - single bf16 flips)
          o[e] = res[e] + rs * (g - xh * dot);
        }
        stg16(dxrow + lo + i * 512, pack8(o));
      }
    }
  };
  // block b owns rows [b T / G, (b + 1) T / G), wave w of it rows w, w + NW, ... of the range: every CU the same share
  int row = (int)((long)blockIdx.x * T / gridDim.x) + wave;
  const int r_end = (int)((long)(blockIdx.x + 1) * T / gridDim.x);
  if (row < r_end) fetch(row, ra);
  while (row < r_end) {
    if (row + NW < r_end) fetch(row + NW, rb);
    work(row, ra);
    row += NW;
    if (row >= r_end) break;
    if (row + NW < r_end) fetch(row + NW, ra);
    work(row, rb);
    row += NW;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {   // (plane layout of the partials: see rmsnorm_bwd_kernel)
      *reinterpret_cast<float4*>(dw_lds + wave * d + c * 4) = make_float4(dwp[i][0], dwp[i][1], dwp[i][2], dwp[i][3]);
      *reinterpret_cast<float4*>(dw_lds + wave * d + (d >> 1) + c * 4) = make_float4(dwp[i][4], dwp[i][5], dwp[i][6], dwp[i][7]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < d; t += NW * 64) {
    const int hd = d >> 1, pl = t >= hd ? 1 : 0, ix = t - pl * hd;
    const int j = (ix >> 2) * 8 + pl * 4 + (ix & 3);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NW; q += 4) s += (dw_lds[q * d + t] + dw_lds[(q + 1) * d + t]) + (dw_lds[(q + 2) * d + t] + dw_lds[(q + 3) * d + t]);
    if (dw_part) dw_part[(size_t)blockIdx.x * d + j] = s;
    else unsafeAtomicAdd(dw_accum + (size_t)(blockIdx.x % copies) * copy_stride + j, s);
  }
}

// reproducible mode (k_set_deterministic): dst[j] += part[0][j] + part[1][j] + ... in block order (one thread per column)
// A fixed order, not the sequential one: the rows are cut into segments of `per` (one block each), the four waves of a block interleave a
// segment's rows and every lane keeps 8 running sums (32 coalesced row reads in flight per block); lane sums, wave sums and - in a
// second launch over the segment sums - the segments are folded in fixed trees.
__global__ void __launch_bounds__(kBlock) ordered_colsum_kernel(const float* __restrict__ part, int nblk, int d, int per, float* __restrict__ out,
                                                                int accumulate) {
  __shared__ float ws[kBlock / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int b_lo = blockIdx.y * per, b_hi = min(b_lo + per, nblk);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < d) {
    for (int b0 = b_lo + wave * 8; b0 < b_hi; b0 += (kBlock / 64) * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + u < b_hi) a[u] += part[(size_t)(b0 + u) * d + j];
    }
  }
  ws[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (wave == 0 && j < d) {
    const float t = (ws[0][lane] + ws[1][lane]) + (ws[2][lane] + ws[3][lane]);
    if (accumulate) out[j] += t;
    else out[(size_t)blockIdx.y * d + j] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// K6  RoPE in place on the q and k thirds of qkv [T,3d]
// reference: hf apply_rotary_pos_emb :138-160 / rotate_half :130-135 (half-split pairing j <-> j+32).
// inverse=1 applies the transpose rotation (backward).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cos_tab,
                                                      const float* __restrict__ sin_tab,
                                                      const int64_t* __restrict__ position_ids, int T, int S, int H,
                                                      int inverse) {
  const int d = H * 64;
  const long total = (long)T * 2 * H * 4;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const int j8 = (int)(w & 3);
    long r = w >> 2;
    const int h = (int)(r % H); r /= H;
    const int sec = (int)(r & 1);
    const int t = (int)(r >> 1);
    const int pos = position_ids ? (int)position_ids[t] : (t % S);
    bf16_t* p = qkv + (size_t)t * 3 * d + sec * d + h * 64 + j8 * 8;
    float a[8], b[8], c[8], s[8];
    unpack8(ldg16(p), a);
    unpack8(ldg16(p + 32), b);
    const float4* ct = reinterpret_cast<const float4*>(cos_tab + (size_t)pos * 32 + j8 * 8);
    const float4* stb = reinterpret_cast<const float4*>(sin_tab + (size_t)pos * 32 + j8 * 8);
    const float4 c0 = ct[0], c1 = ct[1], s0 = stb[0], s1 = stb[1];
    c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
    s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
    float oa[8], ob[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sn = inverse ? -s[e] : s[e];
      oa[e] = a[e] * c[e] - b[e] * sn;
      ob[e] = b[e] * c[e] + a[e] * sn;
    }
    stg16(p, pack8(oa));
    stg16(p + 32, pack8(ob));
  }
}

// hf LlamaRotaryEmbedding.forward :111-127: fp32 freqs = pos * inv_freq, cos/sin in fp32
__global__ void rope_table_kernel(float* cos_tab, float* sin_tab, int max_pos, float theta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_pos * 32) return;
  const int pos = i >> 5, j = i & 31;
  const float inv_freq = 1.0f / powf(theta, (float)(2 * j) / 64.0f);
  const float fr = (float)pos * inv_freq;
  cos_tab[i] = (float)cos((double)fr);
  sin_tab[i] = (float)sin((double)fr);
}

// config.rope_range > 0 (utils_graphgpt.reset_pos_ids :574-581, applied whenever position ids are passed): the positions of a row are
// rescaled into [0, range) before the rotary embedding - p' = float(p) * range / float(max_s p + 1) - so they are no longer integers
// and the precomputed table does not apply: this kernel writes the angles of every TOKEN ([T][32] cos / sin, same evaluation as
// rope_table_kernel) and the identity position list that addresses them.  One block per row.
__global__ void __launch_bounds__(kBlock) rope_range_table_kernel(const int64_t* __restrict__ pos, float* __restrict__ cos_t,
                                                                  float* __restrict__ sin_t, int64_t* __restrict__ ids, int S,
                                                                  float range, float theta) {
  __shared__ float red[kBlock / 64];
  const int b = blockIdx.x;
  const int64_t* row = pos + (size_t)b * S;
  float mx = -INFINITY;
  for (int s = threadIdx.x; s < S; s += kBlock) mx = fmaxf(mx, (float)row[s]);     // (positions < 2^24: exact in fp32)
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < kBlock / 64; ++i) mx = fmaxf(mx, red[i]);
  const float den = mx + 1.0f;
  for (int i = threadIdx.x; i < S * 32; i += kBlock) {
    const int s = i >> 5, j = i & 31;
    const float scaled = (float)row[s] * range / den;
    const float inv_freq = 1.0f / powf(theta, (float)(2 * j) / 64.0f);
    const float fr = scaled * inv_freq;
    cos_t[((size_t)b * S + s) * 32 + j] = (float)cos((double)fr);
    sin_t[((size_t)b * S + s) * 32 + j] = (float)sin((double)fr);
    if (j == 0) ids[(size_t)b * S + s] = (int64_t)b * S + s;
  }
}

// ---------------------------------------------------------------------------------------------
// K9  GEGLU: h = bf16(gelu(g)) * u   (hf LlamaMLP.forward :174-176, exact-erf GELU)
// gu is [T, 2ff] with gate columns [0,ff) and up columns [ff,2ff).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) geglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ h, long T,
                                                           int ff) {
  const int cpr = ff >> 3;
  const long total = T * cpr;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const long t = w / cpr;
    const int c = (int)(w % cpr);
    float g[8], u[8], o[8];
    unpack8(ldg16(gu + t * 2 * ff + c * 8), g);
    unpack8(ldg16(gu + t * 2 * ff + ff + c * 8), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(gelu_erf(g[e]))) * u[e];
    stg16(h + t * ff + c * 8, pack8(o));
  }
}

__global__ void __launch_bounds__(kBlock) geglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dh,
                                                           bf16_t* __restrict__ dgu, long T, int ff) {
  const int cpr = ff >> 3;
  const long total = T * cpr;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const long t = w / cpr;
    const int c = (int)(w % cpr);
    float g[8], u[8], dv[8], dg[8], du[8];
    unpack8(ldg16(gu + t * 2 * ff + c * 8), g);
    unpack8(ldg16(gu + t * 2 * ff + ff + c * 8), u);
    unpack8(ldg16(dh + t * ff + c * 8), dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gv, gd;
      gelu_erf_both(g[e], gv, gd);
      dg[e] = dv[e] * u[e] * gd;
      du[e] = dv[e] * gv;
    }
    stg16(dgu + t * 2 * ff + c * 8, pack8(dg));
    stg16(dgu + t * 2 * ff + ff + c * 8, pack8(du));
  }
}

// ---------------------------------------------------------------------------------------------
// K2  key lengths from the right-padded attention mask (replaces the [B,1,S,S] additive mask of
// _update_causal_mask, modeling_helpers.py:38-48); also the "last" pooling row of the task head
// (_get_sequence_len, modeling_helpers.py:78-86: (in_ != pad).sum(-1) - 1).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) lengths_kernel(const int64_t* __restrict__ mask, const int64_t* __restrict__ ids,
                                                     int ldF, int pad_id, int32_t* __restrict__ key_len,
                                                     int32_t* __restrict__ pool_row, int B, int S) {
  const int b = blockIdx.x;
  int n = 0, m = 0;
  for (int s = threadIdx.x; s < S; s += 64) {
    if (mask) n += mask[(size_t)b * S + s] != 0;
    if (ids) m += ids[((size_t)b * S + s) * ldF] != pad_id;
  }
  n = (int)wave_sum((float)n);
  m = (int)wave_sum((float)m);
  if (threadIdx.x == 0) {
    if (key_len) key_len[b] = mask ? n : S;
    if (pool_row) pool_row[b] = b * S + ((m - 1 + S) % S);
  }
}

// ---------------------------------------------------------------------------------------------
// K11  SMTP head compaction (modeling_helpers.py:263-301): positions with >=1 masked feature -> M rows,
// masked (row,f) cells -> Lm rows, in (b,s,f) row-major order, without leaving the device.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Two launches (round 6; before: flags -> a single-block scan over all T tokens, 12 us at T = 8192 -> fill):
// (1) label count of every token and the (selected rows, labelled cells) totals of every kBlock-token block;
__global__ void __launch_bounds__(kBlock) head_count_kernel(const int64_t* __restrict__ labels, int32_t* __restrict__ cnt,
                                                            int32_t* __restrict__ blk_tot, int T, int n) {
  __shared__ int sm[kBlock / 64], sl[kBlock / 64];
  const int t = blockIdx.x * kBlock + threadIdx.x;
  int c = 0;
  if (t < T) {
    c = n;
    if (labels) {
      c = 0;
      for (int f = 0; f < n; ++f) c += labels[(size_t)t * n + f] != -100;
    }
    cnt[t] = c;
  }
  const int wm = wave_sum_i32(c > 0), wl = wave_sum_i32(c);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = wm; sl[threadIdx.x >> 6] = wl; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = 0, l = 0;
    for (int w = 0; w < kBlock / 64; ++w) { m += sm[w]; l += sl[w]; }
    blk_tot[2 * blockIdx.x] = m;
    blk_tot[2 * blockIdx.x + 1] = l;
  }
}
// (2) every block sums the totals of the blocks before it (T / kBlock pairs at most), scans its own kBlock tokens - exclusive offsets of
// (cnt > 0) and cnt in token order, counts[0] = M, counts[1] = Lm from the last block - and writes its rows / cells.
__global__ void __launch_bounds__(kBlock) head_fill_kernel(const int64_t* __restrict__ labels, const int32_t* __restrict__ cnt,
                                                           const int32_t* __restrict__ blk_tot, int32_t* __restrict__ m_off,
                                                           int32_t* __restrict__ l_off, int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ row_idx, int32_t* __restrict__ sel_src,
                                                           int32_t* __restrict__ sel_label, int32_t* __restrict__ sel_tok,
                                                           int32_t* __restrict__ slot_hist, int T, int n) {
  __shared__ int pm_s[kBlock / 64], pl_s[kBlock / 64], wm_s[kBlock / 64], wl_s[kBlock / 64], hist[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (slot_hist && threadIdx.x < 32) hist[threadIdx.x] = 0;     // (ordered before its use by the barrier below)
  int pm = 0, pl = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += kBlock) { pm += blk_tot[2 * j]; pl += blk_tot[2 * j + 1]; }
  pm = wave_sum_i32(pm);
  pl = wave_sum_i32(pl);
  const int t = blockIdx.x * kBlock + threadIdx.x;
  const int c = t < T ? cnt[t] : 0;
  int im = c > 0, il = c;   // inclusive scan inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int am = __shfl_up(im, o, 64), al = __shfl_up(il, o, 64);
    if (lane >= o) { im += am; il += al; }
  }
  if (lane == 63) { wm_s[wave] = im; wl_s[wave] = il; }
  if (lane == 0) { pm_s[wave] = pm; pl_s[wave] = pl; }
  __syncthreads();
  int bm = 0, bl = 0;
  for (int w = 0; w < kBlock / 64; ++w) { bm += pm_s[w]; bl += pl_s[w]; }
  for (int w = 0; w < wave; ++w) { bm += wm_s[w]; bl += wl_s[w]; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) { counts[0] = bm + im; counts[1] = bl + il; }
  if (t < T) {
    const int m = bm + im - (c > 0);
    int j = bl + il - c;
    m_off[t] = m;
    l_off[t] = j;
    if (c > 0) {
      row_idx[m] = t;
      for (int f = 0; f < n; ++f) {
        const int64_t lab = labels ? labels[(size_t)t * n + f] : 0;
        if (!labels || lab != -100) {
          sel_src[j] = m * n + f;
          sel_label[j] = (int32_t)lab;
          sel_tok[j] = t;
          if (slot_hist) atomicAdd(&hist[f], 1);
          ++j;
        }
      }
    }
  }
  if (slot_hist) {     // cells per slot (the slot-sorted head below; n <= 32): one global atomic per slot and block
    __syncthreads();
    if ((int)threadIdx.x < n && hist[threadIdx.x]) atomicAdd(&slot_hist[threadIdx.x], hist[threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------------
// Slot-sorted SMTP head (round 4).  The reference projects every selected row through ALL n slots of n_token_proj and then keeps the
// labelled cells (modeling_helpers.py:263-301: `proj(hidden_states[mask_m])`, `hidden_states[mask.reshape(-1)]`) - half of that product
// is thrown away (SMTP masks ~50 % of the cells of a selected row).  Here the labelled cells are sorted by slot, every slot padded to a
// multiple of 128 rows, so that ONE GEMM over the sorted cells with a per-row-tile weight block W_f computes exactly the kept rows:
//   Hl[l] = hidden[tok(l)] . W_f(l)^T         (forward:  A rows gathered, C rows scattered back to the cells' (m, f) order)
//   dXs[p] = dP[cell(p)] . W_f(p)             (backward: one row per cell; head_cell_sum adds a token's cells into d hidden)
// slot_state: [0, n) cell counts, [n, 2n) fill cursors, [2n, 3n + 1) padded slot starts.
// ---------------------------------------------------------------------------------------------
// ONE launch (round 6; histogram / plan / fill before): the per-slot cell counts come from head_fill_kernel (slot_state[0, n)); every
// block derives the padded slot starts from them; block 0 writes what the one-block plan kernel wrote (the per-row-tile weight offsets,
// the defaults of the pad rows, the padded total); positions inside a slot: every block ranks the cells of its chunk per slot in LDS
// and reserves ONE range per slot from the global cursors (13 global atomics per block; one per cell - 37 k atomics on 13 addresses -
// took 83 us); the LAST block to finish (ticket slot_state[3 n + 1]) clears counts, cursors and the ticket for the next forward.
constexpr int kFillItems = 4;
__global__ void __launch_bounds__(kBlock) slot_fill_kernel(const int32_t* __restrict__ sel_src, const int32_t* __restrict__ row_idx,
                                                           const int32_t* __restrict__ count, int32_t* __restrict__ slot_state,
                                                           int32_t* __restrict__ a_tok, int32_t* __restrict__ a_cell,
                                                           int32_t* __restrict__ c_l, int32_t* __restrict__ cellpos,
                                                           int32_t* __restrict__ tile_off, int32_t* __restrict__ total_p, int cap, int n,
                                                           long slot_elems, int kSlotPad) {     // kSlotPad = row-tile height of the consuming GEMM
  __shared__ int cnt_s[32], base_s[32], start[64], cells[64];
  const int lm = min(cap, *count);
  const int l0 = blockIdx.x * (kBlock * kFillItems);
  if (threadIdx.x < 32) cnt_s[threadIdx.x] = 0;
  if (threadIdx.x < 64) {     // padded slot starts: the n counts in one round trip, an exclusive scan across the first wave (n <= 32)
    const int lane = threadIdx.x;
    const int cf = lane < n ? slot_state[lane] : 0;
    const int pc = (cf + kSlotPad - 1) / kSlotPad * kSlotPad;
    int inc = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    cells[lane] = cf;
    start[lane] = inc - pc;     // (start[n] = the padded total)
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *total_p = start[n];
    if ((int)threadIdx.x <= n) slot_state[2 * n + threadIdx.x] = start[threadIdx.x];
    for (int f = 0; f < n; ++f) {
      const int lo = start[f] + cells[f], hi = start[f + 1];
      for (int p = lo + threadIdx.x; p < hi; p += kBlock) { a_tok[p] = 0; a_cell[p] = 0; c_l[p] = -1; }     // pad rows: read row 0, store nothing
      for (int t = start[f] / kSlotPad + threadIdx.x; t < start[f + 1] / kSlotPad; t += kBlock) tile_off[t] = (int32_t)((long)f * slot_elems);
    }
  }
  if (l0 < lm) {
    int cell[kFillItems], rank[kFillItems], tok[kFillItems];
#pragma unroll
    for (int i = 0; i < kFillItems; ++i) {
      const int l = l0 + i * kBlock + threadIdx.x;
      cell[i] = l < lm ? sel_src[l] : -1;
    }
#pragma unroll
    for (int i = 0; i < kFillItems; ++i) {
      tok[i] = cell[i] >= 0 ? row_idx[cell[i] / n] : 0;     // (requested before the cursor atomics: one dependent round trip less)
      rank[i] = cell[i] >= 0 ? atomicAdd(&cnt_s[cell[i] % n], 1) : 0;
    }
    __syncthreads();
    if ((int)threadIdx.x < n)
      base_s[threadIdx.x] = start[threadIdx.x] + (cnt_s[threadIdx.x] ? atomicAdd(&slot_state[n + threadIdx.x], cnt_s[threadIdx.x]) : 0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kFillItems; ++i) {
      if (cell[i] < 0) continue;
      const int l = l0 + i * kBlock + threadIdx.x;
      const int p = base_s[cell[i] % n] + rank[i];     // (order inside a slot is arbitrary: every row of the products is independent)
      a_tok[p] = tok[i];
      a_cell[p] = cell[i];
      c_l[p] = l;
      cellpos[l] = p;
    }
  }
  __syncthreads();     // (every read of the counts by this block lies before its ticket)
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&slot_state[3 * n + 1], 1) == (int)gridDim.x - 1) {
      for (int f = 0; f < 2 * n; ++f) slot_state[f] = 0;
      slot_state[3 * n + 1] = 0;
    }
  }
}
// d hidden[row_idx[m]] = sum over the labelled cells of selected row m of dXs[cellpos[l]] (fp32 sum in slot order, one bf16 rounding);
// one wave per selected row, cells l_off[t] .. l_off[t] + cnt[t] of token t = row order of the compaction
__global__ void __launch_bounds__(kBlock) head_cell_sum_kernel(const bf16_t* __restrict__ dxs, const int32_t* __restrict__ cellpos,
                                                               const int32_t* __restrict__ cnt, const int32_t* __restrict__ l_off,
                                                               const int32_t* __restrict__ pad2c, bf16_t* __restrict__ dhid, int TP, int d,
                                                               int pad_row) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (t >= TP) return;
  const int c = cnt[t];
  if (c == 0) return;
  const int l0 = l_off[t];
  int row = t;
  if (pad2c) { row = pad2c[t]; if (row < 0) row = pad_row; }       // var-len layout: the token's compact row (gather_rows_remap_kernel's rule)
  // the token's cell rows (<= 32 of them): lane j holds the row of cell j, broadcast below; four row loads in flight per lane
  const int myrow = lane < c ? cellpos[l0 + lane] : 0;
  if ((d >> 3) <= 128) {
    // d <= 1024: both channel passes of a cell batch in one go - eight row loads in flight per lane instead of four, half as many
    // dependent round trips per token (24.8 -> 23.1 us at the headline shape: the launch is bound elsewhere); same sums in the same order
    const bool on1 = lane + 64 < (d >> 3), on0 = lane < (d >> 3);
    float a0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < c; j += 4) {
      uint4 q0[4], q1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = __shfl(myrow, min(j + u, c - 1), 64);
        const bf16_t* src = dxs + (size_t)r * d + lane * 8;
        q0[u] = on0 ? ldg16(src) : make_uint4(0u, 0u, 0u, 0u);
        q1[u] = on1 ? ldg16(src + 512) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j + u < c) {
          float v[8];
          unpack8(q0[u], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) a0[e] += v[e];
          unpack8(q1[u], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) a1[e] += v[e];
        }
      }
    }
    if (on0) stg16(dhid + (size_t)row * d + lane * 8, pack8(a0));
    if (on1) stg16(dhid + (size_t)row * d + lane * 8 + 512, pack8(a1));
    return;
  }
  // the trip count is wave-uniform (ch0, not ch): __shfl reads from the lanes that hold the cell rows, which must be active in the
  // last partial pass over the channels too (d/8 % 64 != 0: d = 576, 1152, ...) - loads and the store are predicated instead
  for (int ch0 = 0; ch0 < (d >> 3); ch0 += 64) {
    const int ch = ch0 + lane;
    const bool on = ch < (d >> 3);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < c; j += 4) {
      uint4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = __shfl(myrow, min(j + u, c - 1), 64);
        q[u] = on ? ldg16(dxs + (size_t)r * d + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j + u < c) {
          float v[8];
          unpack8(q[u], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
      }
    }
    if (on) stg16(dhid + (size_t)row * d + ch * 8, pack8(acc));
  }
}

// dst[i,:] = src[idx[i],:]  (i < *count)   /   scatter: dst[idx[i],:] = src[i,:]
__global__ void __launch_bounds__(kBlock) gather_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ count, bf16_t* __restrict__ dst,
                                                             int cap, int d, int scatter) {
  const int n = min(cap, *count);
  const int cpr = d >> 3;
  const long total = (long)n * cpr;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const long i = w / cpr;
    const int c = (int)(w % cpr);
    const long r = idx[i];
    if (scatter) stg16(dst + r * d + c * 8, ldg16(src + i * d + c * 8));
    else stg16(dst + i * d + c * 8, ldg16(src + r * d + c * 8));
  }
}

// SMTP head on the var-len layout: the selected rows (padded token indices, head_fill_kernel) -> rows of the compact layout, and the
// gather of those rows, in one launch (a remap launch + gather_rows_kernel before round 6).  One wave per row i reads idx[i], maps it
// through pad2c, gathers the row and writes the mapped index back - only this wave touches idx[i], and its store follows its loads in
// program order.  A label != -100 at a PADDED position (the reference's collator never writes one: labels are padded with -100)
// selects a row the compact layout does not hold: it is sent to `pad_row` (a pad-token row behind the real tokens when the row count
// was rounded up, else row 0) and the sticky flag status[2] is raised - gget_deferred_status reports that this step's loss differs
// from the padded layout's.
__global__ void __launch_bounds__(kBlock) gather_rows_remap_kernel(const bf16_t* __restrict__ src, int32_t* __restrict__ idx,
                                                                   const int32_t* __restrict__ count, const int32_t* __restrict__ pad2c,
                                                                   bf16_t* __restrict__ dst, int cap, int d, int pad_row,
                                                                   int32_t* __restrict__ status) {
  const int n = min(cap, *count);
  const int lane = threadIdx.x & 63, cpr = d >> 3;
  bool bad = false;
  for (int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < n; i += gridDim.x * (kBlock / 64)) {
    int r = pad2c[idx[i]];
    if (r < 0) { bad = true; r = pad_row; }
    for (int c = lane; c < cpr; c += 64) stg16(dst + (size_t)i * d + c * 8, ldg16(src + (size_t)r * d + c * 8));
    if (lane == 0) idx[i] = r;
  }
  if (status && bad && lane == 0) status[2] = 1;     // (status == nullptr: inference selects EVERY cell of the padded grid on purpose)
}

// ---------------------------------------------------------------------------------------------
// K14  cross-entropy over [rows, V] bf16 logits in fp32 (_get_ce_loss :145-177 / _get_dlm_ce_loss
// :180-198) fused with its backward: dlogits = (softmax - onehot) * w_row * scale.
// One wave per row, block-level loss reduction, one atomic per block.
// ---------------------------------------------------------------------------------------------
// focal loss (utils_graphgpt.FocalLoss :340-376, chosen by config.focal_gamma > 0 in _get_ce_loss :158-160): the row's CE is
// weighted by (1 - p_t)^gamma with p_t = exp(log p_t) DETACHED (the reference wraps it in Variable(logpt.data.exp())), so the
// same factor scales the row's gradient.
__device__ __forceinline__ float focal_weight(float logpt, float gamma) {
  if (gamma <= 0.f) return 1.f;
  const float om = fmaxf(1.f - __expf(logpt), 0.f);
  return om > 0.f ? __builtin_amdgcn_exp2f(gamma * __log2f(om)) : 0.f;
}
__global__ void __launch_bounds__(kBlock) ce_fwd_bwd_kernel(const bf16_t* __restrict__ logits, int ld,
                                                            const int32_t* __restrict__ labels,
                                                            const int32_t* __restrict__ sel_tok,
                                                            const float* __restrict__ sample_wgt, int S,
                                                            const int32_t* __restrict__ n_rows_dev, int n_rows_cap, int V,
                                                            float* __restrict__ loss_sum, bf16_t* __restrict__ dlogits,
                                                            float scale_base, int mean_over_rows, float focal_gamma) {
  __shared__ float part[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_rows = min(n_rows_cap, n_rows_dev ? *n_rows_dev : n_rows_cap);
  const float scale = mean_over_rows ? (n_rows > 0 ? 1.0f / (float)n_rows : 0.f) : scale_base;
  float local = 0.f;
  for (int row = blockIdx.x * (kBlock / 64) + wave; row < n_rows; row += gridDim.x * (kBlock / 64)) {
    const bf16_t* lp = logits + (size_t)row * ld;
    float mx = -INFINITY;
    for (int j = lane; j < V; j += 64) mx = fmaxf(mx, bf2f(lp[j]));
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < V; j += 64) se += __expf(bf2f(lp[j]) - mx);
    se = wave_sum(se);
    const int y = labels[row];
    const float lse = mx + __logf(se);
    const float w = (sample_wgt ? sample_wgt[sel_tok[row] / S] : 1.0f) * focal_weight(bf2f(lp[y]) - lse, focal_gamma);
    if (lane == 0) local += w * (lse - bf2f(lp[y]));
    if (dlogits) {
      bf16_t* dp = dlogits + (size_t)row * ld;
      const float inv = 1.0f / se;
      const float ws = w * scale;
      for (int j = lane; j < ld; j += 64) {
        float gval = 0.f;
        if (j < V) gval = (__expf(bf2f(lp[j]) - mx) * inv - (j == y ? 1.0f : 0.0f)) * ws;
        dp[j] = f2bf(gval);
      }
    }
  }
  if (lane == 0) part[wave] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < kBlock / 64; ++i) s += part[i];
    if (s != 0.f) unsafeAtomicAdd(loss_sum, s);
  }
}

// Rows of at most 64 * 8 * CH columns (ld % 8 == 0, rows 16-byte aligned): a lane keeps its CH 16-byte chunks of the row in
// registers, so the logits are read once and the gradient is written with whole-line stores (the generic kernel above walks
// the row three times with 2-byte accesses).  Same arithmetic per element.
template <int CH>
__global__ void __launch_bounds__(kBlock) ce_rows_kernel(const bf16_t* __restrict__ logits, int ld,
                                                         const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ sel_tok,
                                                         const float* __restrict__ sample_wgt, int S,
                                                         const int32_t* __restrict__ n_rows_dev, int n_rows_cap, int V,
                                                         float* __restrict__ loss_sum, bf16_t* __restrict__ dlogits,
                                                         float scale_base, int mean_over_rows, float focal_gamma,
                                                         float* __restrict__ loss_part) {
  __shared__ float part[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_rows = min(n_rows_cap, n_rows_dev ? *n_rows_dev : n_rows_cap);
  const float scale = mean_over_rows ? (n_rows > 0 ? 1.0f / (float)n_rows : 0.f) : scale_base;
  const int nch = ld >> 3;
  float local = 0.f;
  // software-pipelined over the wave's rows: the 16-byte chunks and the label of the NEXT row are in flight while the current row is
  // reduced, exponentiated and stored (a row is one dependent chain otherwise: chunks -> max -> exp -> sum -> store, with the label's
  // own load -> logit-at-label load behind it; 45 -> measured below, profiles/r03_step_experiments.txt)
  const int rstride = gridDim.x * (kBlock / 64);
  int row = blockIdx.x * (kBlock / 64) + wave;
  uint4 nx[CH];
  int ny = 0;
  auto fetch = [&](int r) {
    const bf16_t* q = logits + (size_t)r * ld;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int ch = lane + 64 * t;
      nx[t] = ch < nch ? ldg16(q + ch * 8) : make_uint4(0, 0, 0, 0);
    }
    ny = labels[r];
  };
  if (row < n_rows) fetch(row);
  for (; row < n_rows; row += rstride) {
    const bf16_t* lp = logits + (size_t)row * ld;
    float x[CH][8];
    float mx = -INFINITY;
    const int y = ny;
    const float xy = bf2f(lp[y]);
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int ch = lane + 64 * t;
      unpack8(nx[t], x[t]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (ch >= nch || ch * 8 + e >= V) x[t][e] = -INFINITY;
        mx = fmaxf(mx, x[t][e]);
      }
    }
    if (row + rstride < n_rows) fetch(row + rstride);
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[t][e] = __expf(x[t][e] - mx); se += x[t][e]; }   // exp(-inf) = 0 on the pad columns
    se = wave_sum(se);
    const float w = (sample_wgt ? sample_wgt[sel_tok[row] / S] : 1.0f) * focal_weight(xy - (mx + __logf(se)), focal_gamma);
    if (lane == 0) local += w * (mx + __logf(se) - xy);
    if (dlogits) {
      bf16_t* dp = dlogits + (size_t)row * ld;
      const float inv = 1.0f / se, ws = w * scale;
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int ch = lane + 64 * t;
        if (ch >= nch) continue;
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = ch * 8 + e;
          gv[e] = j < V ? (x[t][e] * inv - (j == y ? 1.0f : 0.0f)) * ws : 0.f;
        }
        stg16(dp + ch * 8, pack8(gv));
      }
    }
  }
  if (lane == 0) part[wave] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < kBlock / 64; ++i) s += part[i];
    // (one atomic per block on ONE address: 2048 of them cost the headline launch 13 of its 38 us - the engine hands in a slot per block)
    if (loss_part) loss_part[blockIdx.x] = s;
    else if (s != 0.f) unsafeAtomicAdd(loss_sum, s);
  }
}

// loss_part != nullptr: the blocks of ce_rows_kernel left one partial sum each (n_part of them) - summed here in a fixed order, the total
// also goes to loss_sum[0] (what the atomics of the other form accumulate)
__global__ void __launch_bounds__(256) finalize_loss_kernel(float* loss_sum, const int32_t* n_rows_dev, float scale_base, int mean_over_rows,
                                                            float* loss_out, const float* __restrict__ loss_part, int n_part) {
  __shared__ float red[4];
  if (loss_part) {
    float a = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) a += loss_part[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss_sum[0] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss_out) {
    const float sc = mean_over_rows ? (*n_rows_dev > 0 ? 1.0f / (float)(*n_rows_dev) : 0.f) : scale_base;
    loss_out[0] = loss_sum[0] * sc;
  }
}

// ---------------------------------------------------------------------------------------------
// K15  fine-tune head on the pooled row only (modeling_finetune.py:281-296 computes `score` on all
// rows then indexes; algebraically identical): logits[b,c] = bf16(h[row_b] . W[c] + bias[c]).float()
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) score_fwd_kernel(const bf16_t* __restrict__ hidden, const int32_t* __restrict__ pool_row,
                                                       const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                       float* __restrict__ logits, bf16_t* __restrict__ pooled_h, int B,
                                                       int C, int d) {
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  const bf16_t* hp = hidden + (size_t)pool_row[b] * d;
  float acc = 0.f;
  for (int j = threadIdx.x; j < d; j += 64) {
    acc += bf2f(hp[j]) * bf2f(w[(size_t)c * d + j]);
    if (c == 0 && pooled_h) pooled_h[(size_t)b * d + j] = hp[j];
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) logits[b * C + c] = bf2f(f2bf(acc + (bias ? bf2f(bias[c]) : 0.f)));
}

// ---------------------------------------------------------------------------------------------
// Token-level head (loss_type = "token_ce", modeling_finetune.py:162-164, :198-202): `score` and the cross-entropy on EVERY row.
// T is tens of thousands of rows and C tens of classes: too many rows for the pooled-row kernels above, too narrow for the GEMM
// tiles (K = C in the backward).  A wave owns kTokRows consecutive rows and keeps them in registers; the weight rows come from L1.
// ---------------------------------------------------------------------------------------------
constexpr int kTokRows = 4;
constexpr int kTokMaxCols = 16;   // d <= 64 * kTokMaxCols
// logits[t,c] = bf16(h[t] . W[c] + bias[c])   (fp32 out, rounded as the reference's bf16 `score` output)
__global__ void __launch_bounds__(kBlock) tok_score_fwd_kernel(const bf16_t* __restrict__ hidden, const bf16_t* __restrict__ w,
                                                               const bf16_t* __restrict__ bias, float* __restrict__ logits, int T, int C,
                                                               int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * (kBlock / 64) + wave) * kTokRows;
  if (t0 >= T) return;
  const int nc = d >> 6;
  float hv[kTokRows][kTokMaxCols];
#pragma unroll
  for (int r = 0; r < kTokRows; ++r)
#pragma unroll
    for (int k = 0; k < kTokMaxCols; ++k)
      hv[r][k] = (k < nc && t0 + r < T) ? bf2f(hidden[(size_t)(t0 + r) * d + k * 64 + lane]) : 0.f;
  for (int c0 = 0; c0 < C; c0 += 64) {      // lane l keeps class c0 + l of every row
    float out[kTokRows] = {0.f, 0.f, 0.f, 0.f};
    for (int c = c0; c < min(C, c0 + 64); ++c) {
      float wv[kTokMaxCols];
#pragma unroll
      for (int k = 0; k < kTokMaxCols; ++k) wv[k] = k < nc ? bf2f(w[(size_t)c * d + k * 64 + lane]) : 0.f;
      const float bc = bias ? bf2f(bias[c]) : 0.f;
#pragma unroll
      for (int r = 0; r < kTokRows; ++r) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < kTokMaxCols; ++k) a = fmaf(hv[r][k], wv[k], a);
        a = wave_sum(a);
        if (lane == c - c0) out[r] = bf2f(f2bf(a + bc));
      }
    }
#pragma unroll
    for (int r = 0; r < kTokRows; ++r)
      if (t0 + r < T && c0 + lane < C) logits[(size_t)(t0 + r) * C + c0 + lane] = out[r];
  }
}
// per-row cross-entropy with ignore_index = -100 (label < 0): dl[t,:] = softmax - onehot for labelled rows (NOT yet divided by
// their number), zeros otherwise; stat[0] += sum of row losses, stat[1] += labelled rows (fp32 atomics; the count is exact)
// rows_map (var-len token layout): logits / dl rows are the compact rows, labels are indexed by the logical row rows_map[t] (rows behind
// the padded grid - the pad rows that round the row count up - carry no label)
__global__ void __launch_bounds__(kBlock) tok_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                        float* __restrict__ dl, float* __restrict__ stat, int T, int C,
                                                        const int32_t* __restrict__ rows_map, int n_logical) {
  __shared__ float red[2][kBlock / 64];
  float ls = 0.f, cnt = 0.f;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < T; t += gridDim.x * kBlock) {
    const int lt = rows_map ? rows_map[t] : t;
    const int y = lt < n_logical ? (int)labels[lt] : -100;
    const float* lp = logits + (size_t)t * C;
    float* dp = dl + (size_t)t * C;
    if (y < 0) {
      for (int c = 0; c < C; ++c) dp[c] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(lp[c] - mx);
    ls += mx + __logf(se) - lp[y];
    cnt += 1.f;
    const float inv = 1.0f / se;
    for (int c = 0; c < C; ++c) dp[c] = __expf(lp[c] - mx) * inv - (c == y ? 1.f : 0.f);
  }
  ls = wave_sum(ls); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ls; red[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < kBlock / 64; ++i) { a += red[0][i]; b += red[1][i]; }
    if (b > 0.f) { unsafeAtomicAdd(stat, a); unsafeAtomicAdd(stat + 1, b); }
  }
}
__global__ void tok_ce_final_kernel(float* __restrict__ stat, float* __restrict__ loss_out) {
  const float n = stat[1];
  stat[2] = n > 0.f ? 1.0f / n : 0.f;            // the mean's factor, read by the backward kernels
  loss_out[0] = n > 0.f ? stat[0] / n : __builtin_nanf("");   // (torch's mean over no labelled row is nan too)
}
// dhidden[t,:] = bf16( sum_c bf16(dl[t,c] / n) W[c,:] )   (the reference's gradient is a bf16 tensor at both points)
__global__ void __launch_bounds__(kBlock) tok_score_bwd_dx_kernel(const float* __restrict__ dl, const float* __restrict__ stat,
                                                                  const bf16_t* __restrict__ w, bf16_t* __restrict__ dhidden, int T,
                                                                  int C, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * (kBlock / 64) + wave) * kTokRows;
  if (t0 >= T) return;
  const int nc = d >> 6;
  const float inv_n = stat[2];
  float acc[kTokRows][kTokMaxCols];
#pragma unroll
  for (int r = 0; r < kTokRows; ++r)
#pragma unroll
    for (int k = 0; k < kTokMaxCols; ++k) acc[r][k] = 0.f;
  for (int c = 0; c < C; ++c) {
    float g[kTokRows];
#pragma unroll
    for (int r = 0; r < kTokRows; ++r) g[r] = t0 + r < T ? bf2f(f2bf(dl[(size_t)(t0 + r) * C + c] * inv_n)) : 0.f;
#pragma unroll
    for (int k = 0; k < kTokMaxCols; ++k) {
      if (k < nc) {
        const float wv = bf2f(w[(size_t)c * d + k * 64 + lane]);
#pragma unroll
        for (int r = 0; r < kTokRows; ++r) acc[r][k] = fmaf(g[r], wv, acc[r][k]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kTokRows; ++r)
#pragma unroll
    for (int k = 0; k < kTokMaxCols; ++k)
      if (k < nc && t0 + r < T) dhidden[(size_t)(t0 + r) * d + k * 64 + lane] = f2bf(acc[r][k]);
}
// dW[c,:] += sum_t bf16(dl[t,c] / n) h[t,:], dbias[c] += sum_t ...: a block owns kTokSlab rows x kTokCls classes, one fp32 atomic per
// (class, column) per block
constexpr int kTokSlab = 512, kTokCls = 8;
__global__ void __launch_bounds__(kBlock) tok_score_bwd_dw_kernel(const float* __restrict__ dl, const float* __restrict__ stat,
                                                                  const bf16_t* __restrict__ hidden, float* __restrict__ dw,
                                                                  float* __restrict__ dbias, int T, int C, int d) {
  const int tA = blockIdx.x * kTokSlab, tB = min(T, tA + kTokSlab);
  const int c0 = blockIdx.y * kTokCls;
  const float inv_n = stat[2];
  for (int j0 = 0; j0 < d; j0 += kBlock) {
    const int j = j0 + threadIdx.x;
    float acc[kTokCls], bs[kTokCls];
#pragma unroll
    for (int q = 0; q < kTokCls; ++q) { acc[q] = 0.f; bs[q] = 0.f; }
    for (int t = tA; t < tB; ++t) {
      const float hvv = j < d ? bf2f(hidden[(size_t)t * d + j]) : 0.f;
#pragma unroll
      for (int q = 0; q < kTokCls; ++q) {
        const float g = c0 + q < C ? bf2f(f2bf(dl[(size_t)t * C + c0 + q] * inv_n)) : 0.f;
        acc[q] = fmaf(g, hvv, acc[q]);
        bs[q] += g;
      }
    }
#pragma unroll
    for (int q = 0; q < kTokCls; ++q) {
      if (c0 + q < C && j < d && acc[q] != 0.f) unsafeAtomicAdd(dw + (size_t)(c0 + q) * d + j, acc[q]);
      if (dbias && j == 0 && c0 + q < C) unsafeAtomicAdd(dbias + c0 + q, bs[q]);
    }
  }
}

// task loss + dlogits (calculate_task_loss, modeling_finetune.py:167-234); single block.
__global__ void __launch_bounds__(kBlock) task_loss_kernel(const float* __restrict__ logits, const void* __restrict__ labels,
                                                           const float* __restrict__ sample_wgt, int problem, int B, int C,
                                                           float* __restrict__ loss_out, float* __restrict__ dlogits) {
  __shared__ float red[kBlock];
  __shared__ float wsum_s;
  float wsum = 0.f;
  if (problem == GGET_PROBLEM_SINGLE_LABEL && sample_wgt) {
    float p = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) p += sample_wgt[b];
    red[threadIdx.x] = p;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) wsum_s = red[0];
    __syncthreads();
    wsum = wsum_s;
    __syncthreads();
  }
  if (problem == GGET_PROBLEM_MULTI_LABEL) {   // mean over the labelled (non-NaN) entries: count them first
    float p = 0.f;
    for (int i = threadIdx.x; i < B * C; i += kBlock) {
      const float y = reinterpret_cast<const float*>(labels)[i];
      p += (y == y) ? 1.f : 0.f;
    }
    red[threadIdx.x] = p;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) wsum_s = red[0];
    __syncthreads();
    wsum = wsum_s;
    __syncthreads();
  }
  float local = 0.f;
  for (int b = threadIdx.x; b < B; b += kBlock) {
    if (problem == GGET_PROBLEM_MULTI_LABEL) {
      // BCEWithLogitsLoss (pos_weight None) on labelled entries, modeling_finetune.py:227-230:
      // l = max(x, 0) - x*y + log(1 + exp(-|x|)),  dl/dx = sigmoid(x) - y
      const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
      for (int c = 0; c < C; ++c) {
        const float y = reinterpret_cast<const float*>(labels)[b * C + c];
        const float x = logits[b * C + c];
        if (y == y) {
          local += (fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)))) * inv;
          dlogits[b * C + c] = (1.0f / (1.0f + __expf(-x)) - y) * inv;
        } else {
          dlogits[b * C + c] = 0.f;
        }
      }
    } else if (problem == GGET_PROBLEM_SINGLE_LABEL) {
      const int y = (int)reinterpret_cast<const int64_t*>(labels)[b];
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[b * C + c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += __expf(logits[b * C + c] - mx);
      const float lse = mx + __logf(se);
      const float wgt = sample_wgt ? sample_wgt[b] / wsum : 1.0f / (float)B;
      local += wgt * (lse - logits[b * C + y]);
      for (int c = 0; c < C; ++c)
        dlogits[b * C + c] = (__expf(logits[b * C + c] - mx) / se - (c == y ? 1.f : 0.f)) * wgt;
    } else {
      // regression on num_labels == 1 (squeeze) or elementwise over C
      for (int c = 0; c < C; ++c) {
        const float y = reinterpret_cast<const float*>(labels)[b * C + c];
        const float df = logits[b * C + c] - y;
        const float inv = 1.0f / (float)(B * C);
        if (problem == GGET_PROBLEM_REGRESSION_L1) {
          local += fabsf(df) * inv;
          dlogits[b * C + c] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv;
        } else {
          local += df * df * inv;
          dlogits[b * C + c] = 2.f * df * inv;
        }
      }
    }
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) loss_out[0] = red[0];
}

// MLP score head (src/utils/modules_utils.py:8-34, chosen by `len(config.mlp) > 0` at modeling_finetune.py:88-97), on the pooled
// rows only: for every Linear i:  x = Linear_i(dropout(act(x))) - the activation comes BEFORE each linear, the first included.
// Row counts are batch sizes: plain kernels.  bf16 at the points where the reference's bf16 module rounds.
#define GGET_DROP_STREAM_HEAD 51u
__global__ void __launch_bounds__(kBlock) pool_rows_kernel(const bf16_t* __restrict__ hidden, const int32_t* __restrict__ pool_row,
                                                           bf16_t* __restrict__ out, int B, int d) {
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < d; j += kBlock) out[(size_t)b * d + j] = hidden[(size_t)pool_row[b] * d + j];
}
__global__ void __launch_bounds__(kBlock) head_act_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ a, int B, int D,
                                                          unsigned layer, ElemDropArg E) {
  const long n = (long)B * D;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
    const float g = bf2f(f2bf(gelu_erf(bf2f(x[i]))));
    a[i] = f2bf(g * elem_drop_mul(E, GGET_DROP_STREAM_HEAD + layer, (unsigned)(i / D), (unsigned)(i % D)));
  }
}
// y[b,o] = bf16(a[b,:] . w[o,:] + bias[o]); one 64-thread block per (b, o); y32 (optional) = the same value as fp32
__global__ void __launch_bounds__(64) head_linear_fwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ y,
                                                             float* __restrict__ y32, int B, int Din, int Dout) {
  const int b = blockIdx.x / Dout, o = blockIdx.x % Dout;
  float acc = 0.f;
  for (int j = threadIdx.x; j < Din; j += 64) acc += bf2f(a[(size_t)b * Din + j]) * bf2f(w[(size_t)o * Din + j]);
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    const bf16_t r = f2bf(acc + (bias ? bf2f(bias[o]) : 0.f));
    y[(size_t)b * Dout + o] = r;
    if (y32) y32[(size_t)b * Dout + o] = bf2f(r);
  }
}
// backward of one Linear + the activation in front of it: blocks [0, B): dx[b,:] = (dy[b,:] W) * keep * gelu'(x[b,:]);
// blocks [B, B + Dout): dW[o,:] += sum_b dy[b,o] a[b,:], dbias[o] += sum_b dy[b,o]
__global__ void __launch_bounds__(kBlock) head_linear_bwd_kernel(const float* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                 const bf16_t* __restrict__ a, const bf16_t* __restrict__ w,
                                                                 float* __restrict__ dw, float* __restrict__ dbias,
                                                                 float* __restrict__ dx, int B, int Din, int Dout, unsigned layer,
                                                                 ElemDropArg E) {
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < Din; j += kBlock) {
      float s = 0.f;
      for (int o = 0; o < Dout; ++o) s += dy[(size_t)b * Dout + o] * bf2f(w[(size_t)o * Din + j]);
      const float keep = elem_drop_mul(E, GGET_DROP_STREAM_HEAD + layer, (unsigned)b, (unsigned)j);
      dx[(size_t)b * Din + j] = s * keep * gelu_erf_grad(bf2f(x[(size_t)b * Din + j]));
    }
  } else {
    const int o = blockIdx.x - B;
    for (int j = threadIdx.x; j < Din; j += kBlock) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dy[(size_t)b * Dout + o] * bf2f(a[(size_t)b * Din + j]);
      dw[(size_t)o * Din + j] += s;
    }
    if (threadIdx.x == 0 && dbias) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dy[(size_t)b * Dout + o];
      dbias[o] += s;
    }
  }
}
__global__ void __launch_bounds__(kBlock) scatter_rows_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ pool_row,
                                                                  bf16_t* __restrict__ dhidden, int B, int d) {
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < d; j += kBlock) dhidden[(size_t)pool_row[b] * d + j] = f2bf(src[(size_t)b * d + j]);
}

// backward of the score head: dW[c,:] += sum_b dl[b,c] h[row_b,:], dbias[c] += sum_b dl[b,c],
// dhidden[row_b,:] = sum_c dl[b,c] W[c,:] (dhidden pre-zeroed).  grid = B + C * kScoreSplit blocks.
constexpr int kScoreSplit = 16;
__global__ void __launch_bounds__(kBlock) score_bwd_kernel(const float* __restrict__ dlogits, const bf16_t* __restrict__ hidden,
                                                           const int32_t* __restrict__ pool_row, const bf16_t* __restrict__ w,
                                                           float* __restrict__ dw, float* __restrict__ dbias,
                                                           bf16_t* __restrict__ dhidden, int B, int C, int d) {
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += kBlock) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += dlogits[b * C + c] * bf2f(w[(size_t)c * d + j]);
      dhidden[(size_t)pool_row[b] * d + j] = f2bf(s);
    }
  } else {
    // the batch is cut into kScoreSplit slices per class (one block walking 256+ pooled rows alone took 0.2 ms)
    const int c = (blockIdx.x - B) / kScoreSplit, sl = (blockIdx.x - B) % kScoreSplit;
    const int per = (B + kScoreSplit - 1) / kScoreSplit, b0 = sl * per, b1 = min(B, b0 + per);
    if (b0 >= b1) return;
    for (int j = threadIdx.x; j < d; j += kBlock) {
      float s = 0.f;
      for (int b = b0; b < b1; ++b) s += dlogits[b * C + c] * bf2f(hidden[(size_t)pool_row[b] * d + j]);
      unsafeAtomicAdd(dw + (size_t)c * d + j, s);
    }
    if (threadIdx.x == 0 && dbias) {
      float s = 0.f;
      for (int b = b0; b < b1; ++b) s += dlogits[b * C + c];
      unsafeAtomicAdd(dbias + c, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K16  optimizer: global grad norm, clip, AdamW on fp32 master weights, refresh of the bf16 copy
// reference: torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (training_utils.py:68-80,
// opt_utils.py:18-24) == DeepSpeed FusedAdam adam_w_mode (examples/ds_config2_pt.json:11-19).
// ---------------------------------------------------------------------------------------------
// Deterministic: every block stores its partial sum and a one-block pass adds the partials in a fixed order - data-parallel
// replicas must derive bit-identical clip factors from identical gradients (an atomic accumulation gives last-bit
// differences between ranks and the replicas would drift apart; a last-block-done counter costs 1024 contended returning
// atomics, ~35 us).  ws: [0] result, [16 .. 16 + blocks) partials.
constexpr int kSqnormBlocks = 1024;
__global__ void __launch_bounds__(kBlock) grad_sqnorm_kernel(const bf16_t* __restrict__ g, size_t n, float* __restrict__ ws) {
  __shared__ float part[kBlock / 64];
  float s = 0.f;
  const size_t nv = n >> 3;
  const size_t stride = (size_t)gridDim.x * kBlock;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < nv; i += 4 * stride) {   // four loads in flight per thread
    uint4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = ldg16(g + (i + u * stride) * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      unpack8(q[u], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e] * v[e];
    }
  }
  for (; i < nv; i += stride) {
    float v[8];
    unpack8(ldg16(g + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] * v[e];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kBlock / 64; ++i) t += part[i];
    ws[16 + blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(kBlock) grad_sqnorm_final_kernel(float* __restrict__ ws, int nblocks) {
  __shared__ float part[kBlock];
  float t = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) t += ws[16 + i];
  part[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[0] = part[0];
}

// one block per chunk (see k_grad_sqnorm_chunks): ws[16 + chunk] = sum of squares of the chunk
__global__ void __launch_bounds__(kBlock) grad_sqnorm_chunks_kernel(const bf16_t* __restrict__ g, const GgetSqChunk* __restrict__ chunks,
                                                                    float* __restrict__ ws) {
  __shared__ float part[kBlock / 64];
  const GgetSqChunk c = chunks[blockIdx.x];
  const bf16_t* p = g + c.off;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < (c.cnt >> 3); i += kBlock) {
    float v[8];
    unpack8(ldg16(p + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] * v[e];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / 64; ++w) t += part[w];
    ws[16 + blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(kBlock) grad_sqnorm_final2_kernel(float* __restrict__ ws, int nblocks, const float* __restrict__ extra,
                                                                    int nextra) {
  __shared__ float part[kBlock];
  float t = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) t += ws[16 + i];
  for (int i = threadIdx.x; i < nextra; i += kBlock) t += extra[i];
  part[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[0] = part[0];
}

template <int UNR>
__global__ void __launch_bounds__(kBlock) adamw_kernel(float* __restrict__ master, float* __restrict__ m_, float* __restrict__ v_,
                                                       const bf16_t* __restrict__ grad, bf16_t* __restrict__ param, size_t n,
                                                       float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                       float bc2_sqrt, float max_norm, float grad_scale,
                                                       const float* __restrict__ sqnorm, float* __restrict__ gnorm_out, int skip_nonfinite) {
  float coef = grad_scale;
  if (sqnorm) {
    const float nrm = sqrtf(sqnorm[0]) * grad_scale;
    if (max_norm > 0.f) coef *= fminf(1.0f, max_norm / (nrm + 1e-6f));
    if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) gnorm_out[0] = nrm;
    // GradScaler's rule (torch.cuda.amp.GradScaler.step: the optimizer step is skipped when the unscaled gradients hold an inf / NaN;
    // reference src/utils/training_utils.py:76-82, the DDP branch): nothing is touched, the caller reads the norm and leaves its step count
    if (skip_nonfinite && !(nrm < INFINITY)) return;      // (uniform: every thread reads the same scalar; NaN fails the comparison too)
  }
  const size_t nv = n >> 2;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x; i0 < nv; i0 += stride * UNR)
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const size_t i = i0 + u * stride;
    if (i >= nv) break;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v gr = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(grad + i * 4));
    float g[4] = {__uint_as_float(gr.x << 16), __uint_as_float(gr.x & 0xffff0000u), __uint_as_float(gr.y << 16),
                  __uint_as_float(gr.y & 0xffff0000u)};
    f4v w = __builtin_nontemporal_load(reinterpret_cast<f4v*>(master) + i);
    f4v mm = __builtin_nontemporal_load(reinterpret_cast<f4v*>(m_) + i);
    f4v vv = __builtin_nontemporal_load(reinterpret_cast<f4v*>(v_) + i);
    float wp[4] = {w.x, w.y, w.z, w.w}, mp[4] = {mm.x, mm.y, mm.z, mm.w}, vp[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = g[e] * coef;
      mp[e] = beta1 * mp[e] + (1.f - beta1) * ge;
      vp[e] = beta2 * vp[e] + (1.f - beta2) * ge * ge;
      wp[e] *= (1.f - lr * wd);
      const float denom = sqrtf(vp[e]) / bc2_sqrt + eps;
      wp[e] -= (lr / bc1) * (mp[e] / denom);
    }
    __builtin_nontemporal_store(f4v{wp[0], wp[1], wp[2], wp[3]}, reinterpret_cast<f4v*>(master) + i);
    __builtin_nontemporal_store(f4v{mp[0], mp[1], mp[2], mp[3]}, reinterpret_cast<f4v*>(m_) + i);
    __builtin_nontemporal_store(f4v{vp[0], vp[1], vp[2], vp[3]}, reinterpret_cast<f4v*>(v_) + i);
    uint2 o;
    o.x = pack2bf(wp[0], wp[1]);
    o.y = pack2bf(wp[2], wp[3]);
    __builtin_nontemporal_store(u2v{o.x, o.y}, reinterpret_cast<u2v*>(param + i * 4));
  }
}

__global__ void __launch_bounds__(kBlock) f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  const size_t nv = n >> 2;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (size_t)gridDim.x * kBlock) {
    const float4 w = reinterpret_cast<const float4*>(src)[i];
    uint2 o;
    o.x = pack2bf(w.x, w.y);
    o.y = pack2bf(w.z, w.w);
    *reinterpret_cast<uint2*>(dst + i * 4) = o;
  }
}

// sum of `nslab` fp32 slabs (split-K partial products) -> bf16; rows beyond *rows_dev (if given) are left as they are
template <bool F32_OUT>
__global__ void __launch_bounds__(kBlock) slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int nslab,
                                                             void* __restrict__ dst_, size_t n) {
  const size_t nv = n >> 2;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (size_t)gridDim.x * kBlock) {
    float4 a = reinterpret_cast<const float4*>(slabs)[i];
    for (int s = 1; s < nslab; ++s) {
      const float4 b = reinterpret_cast<const float4*>(slabs + (size_t)s * slab_stride)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (F32_OUT) {
      reinterpret_cast<float4*>(dst_)[i] = a;
    } else {
      uint2 o;
      o.x = pack2bf(a.x, a.y);
      o.y = pack2bf(a.z, a.w);
      *reinterpret_cast<uint2*>(static_cast<bf16_t*>(dst_) + i * 4) = o;
    }
  }
}

// fp32 accumulation segments (embedding, norm weights, ...) -> bf16 gradient array
__global__ void __launch_bounds__(kBlock) convert_segments_kernel(const float* __restrict__ scratch, bf16_t* __restrict__ grads,
                                                                  const GgetSegment* __restrict__ segs) {
  const GgetSegment s = segs[blockIdx.y];
  const size_t nv = s.count >> 2;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (size_t)gridDim.x * kBlock) {
    float4 w = reinterpret_cast<const float4*>(scratch + s.src)[i];
    for (uint64_t c = 1; c < s.copies; ++c) {
      const float4 u = reinterpret_cast<const float4*>(scratch + s.src + c * s.stride)[i];
      w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w;
    }
    uint2 o;
    o.x = pack2bf(w.x, w.y);
    o.y = pack2bf(w.z, w.w);
    *reinterpret_cast<uint2*>(grads + s.dst + i * 4) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// In-model SMTP masking
// reference: prepare_for_2d_smtp_inputs_labels + _get_gaussian_rnd_tokens (src/models/graphgpt/modeling_helpers.py:399-468)
// One thread per (sample, position, feature) cell.  torch's Philox stream cannot be reproduced, so the draws are a
// counter hash of (seed, stream, sample, cell) - graph-gpt_amd/smtp.py holds the Python twin the parity tests feed to the
// oracle:  stream 0 sample mask, 1 mask rate of the sample, 2 per-(node, feature) draw, 3 replace draw, 16..27 the twelve
// uniforms whose sum (Irwin-Hall, variance 1) stands in for randn; everything after the draws is integer-exact.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned smtp_rng(unsigned seed, unsigned stream, unsigned a, unsigned b) {
  unsigned x = seed ^ (stream * 0x9E3779B1u);
  x += a * 0x85EBCA77u + b * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x >> 8;   // 24 bits: u = value * 2^-24 is exact in fp32
}
__global__ void __launch_bounds__(kBlock) smtp2d_kernel(const int64_t* __restrict__ ids_in, int ld_in,
                                                        const int64_t* __restrict__ node_idx, int ld_node,
                                                        int64_t* __restrict__ ids_out, int64_t* __restrict__ labels_out,
                                                        int B, int S, int F, float rate, float power, float replace_rate,
                                                        int vocab, int global_mask, unsigned seed, int mask_id) {
  const long total = (long)B * S * F;
  const float inv24 = 1.0f / 16777216.0f;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const int f = (int)(w % F);
    const long bs = w / F;
    const int sidx = (int)(bs % S), b = (int)(bs / S);
    const int64_t id = ids_in[bs * ld_in + f];
    const int ni = (int)node_idx[bs * ld_node];
    const bool sample = (float)smtp_rng(seed, 0, b, 0) * inv24 < rate;
    const float mr = (float)smtp_rng(seed, 1, b, 0) * inv24;
    const float thr = power == 1.0f ? mr : powf(mr, power);
    bool m = (float)smtp_rng(seed, 2, b, (unsigned)(ni * F + f)) * inv24 > thr;
    if (!global_mask) m = m && sample;
    m = m && id > 0;
    int64_t out = m ? (int64_t)mask_id : id;
    if (replace_rate > 0.f && m) {
      const unsigned cell = (unsigned)(sidx * F + f);
      if ((float)smtp_rng(seed, 3, b, cell) * inv24 < replace_rate) {
        long long s12 = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) s12 += smtp_rng(seed, 16 + k, b, cell);
        const long long num = 10 * s12 - 60ll * 16777216ll;       // 10 * (sum - 6) in units of 2^-24
        long long q = num >> 24;
        const long long r = num & 16777215ll;
        if (r > 8388608ll || (r == 8388608ll && (q & 1))) ++q;     // round half to even, like torch.round
        long long t = (id + q) % vocab;
        if (t < 0) t += vocab;                                     // python-style modulo
        out = t;
      }
    }
    ids_out[w] = out;
    labels_out[w] = m ? id : (int64_t)-100;
  }
}

// ---------------------------------------------------------------------------------------------
// Unmasking confidence of the MaskGIT / dLLM generation loop
// reference: sample_tokens at temperature 0 (src/utils/generation_utils.py:45-82): probs = softmax(logits),
// x0 = argmax, confidence = max prob | top1 - top2 prob (margin) | sum p log(p + 1e-10) (negative entropy).
// One wave per row of bf16 logits [R, ld] (V valid columns); fp32 arithmetic; ties -> lowest index (torch.max).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) token_confidence_kernel(const bf16_t* __restrict__ logits, int ld, int R, int V, int mode,
                                                                  float* __restrict__ conf, int64_t* __restrict__ tok) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < R; row += gridDim.x * (kBlock / 64)) {
    const bf16_t* x = logits + (size_t)row * ld;
    float m1 = -INFINITY, m2 = -INFINITY;   // largest and second largest value seen by this lane
    int i1 = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
      const float v = bf2f(x[c]);
      if (v > m1) { m2 = m1; m1 = v; i1 = c; } else if (v > m2) { m2 = v; }
    }
    // wave reduction of (max, argmax with lowest index on ties, second max)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om1 = __shfl_xor(m1, o, 64), om2 = __shfl_xor(m2, o, 64);
      const int oi1 = __shfl_xor(i1, o, 64);
      if (om1 > m1 || (om1 == m1 && oi1 < i1)) { m2 = fmaxf(m1, om2); m1 = om1; i1 = oi1; }
      else { m2 = fmaxf(m2, om1); }
    }
    float se = 0.f;
    for (int c = lane; c < V; c += 64) se += __expf(bf2f(x[c]) - m1);
    se = wave_sum(se);
    float out;
    if (mode == 2) {
      float ent = 0.f;
      for (int c = lane; c < V; c += 64) {
        const float p = __expf(bf2f(x[c]) - m1) / se;
        ent += p * __logf(p + 1e-10f);
      }
      out = wave_sum(ent);
    } else if (mode == 1) {
      out = 1.0f / se - __expf(m2 - m1) / se;
    } else {
      out = 1.0f / se;
    }
    if (lane == 0) { conf[row] = out; tok[row] = i1; }
  }
}

// ---------------------------------------------------------------------------------------------
// Candidate sampling of the generation loop with every option of the reference's sample_tokens
// (src/utils/generation_utils.py:22-82): logits / temperature -> top-p filter -> top-k filter -> softmax -> categorical
// sample (temperature > 0) or arg-max -> confidence (probability of the candidate | top1 - top2 | sum p log(p + 1e-10)),
// optionally perturbed for the Gumbel-max ranking of _batch_unmask_without_for_loop (:199-209): conf / alg_temp + g.
// One wave per row, the row lives in LDS as fp32 (scaled logits and probabilities).  Draws are the 24-bit counter hash of
// (seed, stream, row): stream 32 the categorical uniform (inverse CDF in index order: first c with cumsum p > u, where the
// reference calls torch's multinomial - same distribution, different draw), stream 33 the Gumbel uniform.
// Filters restated without a sort: top-p keeps c iff the probability mass ranked strictly ahead of it (larger p, or equal p
// and lower index = a stable descending sort) is <= top_p; top-k keeps c iff fewer than k values are strictly larger
// (ties at the k-th value all survive, like `logits < topk(...)[..., -1]`).  O(V^2 / 64) per lane: vocabularies <= 8192.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) token_sample_kernel(const bf16_t* __restrict__ logits, int ld, int R, int V, int mode,
                                                              float temperature, float top_p, int top_k, float alg_temp,
                                                              unsigned seed, float* __restrict__ conf, int64_t* __restrict__ tok,
                                                              int waves_per_block) {
  extern __shared__ float ts_lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv >= waves_per_block) return;
  float* s = ts_lds + (size_t)wv * 2 * V;
  float* p = s + V;
  const float inv24 = 1.0f / 16777216.0f;
  const float NEG = -3.4028234663852886e38f;   // torch.finfo(float32).min, what masked_fill writes
  for (int row = blockIdx.x * waves_per_block + wv; row < R; row += gridDim.x * waves_per_block) {
    const bf16_t* x = logits + (size_t)row * ld;
    const float invt = temperature > 0.f ? 1.0f / temperature : 1.0f;
    for (int c = lane; c < V; c += 64) s[c] = temperature > 0.f ? bf2f(x[c]) / temperature : bf2f(x[c]);
    (void)invt;
    __builtin_amdgcn_wave_barrier();
    auto softmax_to_p = [&]() {
      float m = -INFINITY;
      for (int c = lane; c < V; c += 64) m = fmaxf(m, s[c]);
      m = wave_max(m);
      float z = 0.f;
      for (int c = lane; c < V; c += 64) { const float e = __expf(s[c] - m); p[c] = e; z += e; }
      z = wave_sum(z);
      for (int c = lane; c < V; c += 64) p[c] = p[c] / z;
      __builtin_amdgcn_wave_barrier();
    };
    if (top_p > 0.f && top_p < 1.f) {
      softmax_to_p();
      for (int c = lane; c < V; c += 64) {
        const float pc = p[c];
        float ahead = 0.f;
        for (int i = 0; i < V; ++i) {
          const float pi = p[i];
          ahead += (pi > pc || (pi == pc && i < c)) ? pi : 0.f;
        }
        if (ahead > top_p) s[c] = NEG;   // the first token is always kept (nothing ahead of it)
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (top_k > 0) {
      const int kk = min(top_k, V);
      for (int c = lane; c < V; c += 64) p[c] = s[c];   // snapshot: the filter compares unfiltered values
      __builtin_amdgcn_wave_barrier();
      for (int c = lane; c < V; c += 64) {
        const float sc = p[c];
        int larger = 0;
        for (int i = 0; i < V; ++i) larger += p[i] > sc;
        if (larger >= kk) s[c] = NEG;
      }
      __builtin_amdgcn_wave_barrier();
    }
    softmax_to_p();
    // candidate
    int x0;
    float cf;
    float m1 = -1.f, m2 = -1.f;   // largest and second largest probability, arg-max with the lowest index
    int i1 = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
      const float v = p[c];
      if (v > m1) { m2 = m1; m1 = v; i1 = c; } else if (v > m2) { m2 = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om1 = __shfl_xor(m1, o, 64), om2 = __shfl_xor(m2, o, 64);
      const int oi1 = __shfl_xor(i1, o, 64);
      if (om1 > m1 || (om1 == m1 && oi1 < i1)) { m2 = fmaxf(m1, om2); m1 = om1; i1 = oi1; }
      else { m2 = fmaxf(m2, om1); }
    }
    if (temperature > 0.f) {
      const float u = (float)smtp_rng(seed, 32, (unsigned)row, 0) * inv24;
      float run = 0.f;
      int found = -1, last_pos = -1;
      for (int base = 0; base < V && found < 0; base += 64) {
        const int c = base + lane;
        const float v = c < V ? p[c] : 0.f;
        float pre = v;   // inclusive prefix over the 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float t = __shfl_up(pre, o, 64);
          if (lane >= o) pre += t;
        }
        const unsigned long long hit = __ballot(c < V && run + pre > u);
        const unsigned long long pos = __ballot(c < V && v > 0.f);
        if (pos) last_pos = base + 63 - __builtin_clzll(pos);
        if (hit) found = base + __builtin_ctzll(hit);
        run += __shfl(pre, 63, 64);
      }
      if (found < 0) {   // u beyond the accumulated mass (rounding): the last token with probability
        found = last_pos >= 0 ? last_pos : i1;
      }
      x0 = found;
      cf = p[x0];
    } else {
      x0 = i1;
      cf = m1;
    }
    if (mode == 1) cf = m1 - m2;
    if (mode == 2) {
      float ent = 0.f;
      for (int c = lane; c < V; c += 64) { const float v = p[c]; ent += v * __logf(v + 1e-10f); }
      cf = wave_sum(ent);
    }
    if (alg_temp > 0.f) {
      const float u2 = (float)smtp_rng(seed, 33, (unsigned)row, 0) * inv24;
      cf = cf / alg_temp - __logf(-__logf(u2 + 1e-9f) + 1e-9f);
    }
    if (lane == 0) { conf[row] = cf; tok[row] = x0; }
    __builtin_amdgcn_wave_barrier();
  }
}

// alg = "origin" update of the generation loop (generation_utils.py:150-162): a masked cell takes its candidate when its own
// uniform draw (stream 34 of the counter hash, indexed by (sample, cell)) falls below p_transfer.
__global__ void __launch_bounds__(kBlock) unmask_origin_kernel(int64_t* __restrict__ x, const int64_t* __restrict__ cand, int B, int N,
                                                               float p_transfer, unsigned seed, int mask_id) {
  const long total = (long)B * N;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const unsigned b = (unsigned)(w / N), n = (unsigned)(w % N);
    const float u = (float)smtp_rng(seed, 34, b, n) * (1.0f / 16777216.0f);
    if (x[w] == mask_id && u < p_transfer) x[w] = cand[w];
  }
}

// first / last non-zero of every row of a [B,S,S] int64 mask: one wave per row (block-diagonal packing masks,
// reference src/utils/tokenizer_utils.py:349-355).  An all-zero row (padding) gives lo = 0, hi = -1 (attends nothing).
__global__ void __launch_bounds__(kBlock) ranges_from_mask3d_kernel(const int64_t* __restrict__ mask, int32_t* __restrict__ lo,
                                                                    int32_t* __restrict__ hi, int rows, int S) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < rows; row += gridDim.x * (kBlock / 64)) {
    const int64_t* m = mask + (size_t)row * S;
    int first = S, last = -1;
    for (int c = lane; c < S; c += 64) {
      if (m[c] != 0) { first = min(first, c); last = max(last, c); }
    }
    first = (int)-wave_max(-(float)first);
    last = (int)wave_max((float)last);
    if (lane == 0) { lo[row] = last >= 0 ? first : 0; hi[row] = last; }
  }
}

// ---------------------------------------------------------------------------------------------
// Collator-side SMTP masking on the device
// reference: prepare_inputs_for_pretrain_mlm (src/utils/tokenizer_utils.py:259-271, polynomial schedule) +
// _mask_stacked_input_ids_v2 (:112-148, mask_token_precent (1,0,0)): per sample t = umr_min + (umr_max-umr_min) U,
// alpha = 1 - t^power, EXACTLY k = ceil(len*F*alpha) of the sample's own len*F cells are chosen uniformly at random:
// here the k cells with the smallest keys (24-bit counter hash of (seed, sample, cell), ties broken by the cell index);
// the k-th smallest key is found by bisection over the 44-bit key space, one block per sample.
// Chosen cell: label = original id, id -> <mask> unless it is the pad id; every other label = -100.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long smtp_cell_key(unsigned seed, unsigned b, unsigned cell) {
  return ((unsigned long long)smtp_rng(seed, 8, b, cell) << 20) | cell;
}
__global__ void __launch_bounds__(kBlock) smtp_rows_kernel(const int64_t* __restrict__ ids_in, const int32_t* __restrict__ lengths,
                                                           int64_t* __restrict__ ids_out, int64_t* __restrict__ labels_out,
                                                           float* __restrict__ wgt_out, int S, int F, double umr_min,
                                                           double umr_max, double power, unsigned seed, int mask_id, int pad_id) {
  __shared__ int cnt_s[kBlock / 64];
  __shared__ int total_s;
  const int b = blockIdx.x;
  const int len = min(max(lengths[b], 0), S);
  const int n = len * F;
  const double r = (double)smtp_rng(seed, 9, b, 0) * (1.0 / 16777216.0);
  const double t = umr_min + (umr_max - umr_min) * r;
  const double alpha = 1.0 - pow(t, power);
  const int k = (int)ceil((double)n * alpha);
  if (wgt_out && threadIdx.x == 0) wgt_out[b] = (float)(power / t);
  // smallest key value v with #{cells: key <= v} >= k
  unsigned long long lo = 0, hi = (1ull << 44) - 1;
  if (k > 0) {
    while (lo < hi) {
      const unsigned long long mid = lo + ((hi - lo) >> 1);
      int c = 0;
      for (int cell = threadIdx.x; cell < n; cell += kBlock) c += smtp_cell_key(seed, b, cell) <= mid ? 1 : 0;
      c = (int)wave_sum((float)c);
      if ((threadIdx.x & 63) == 0) cnt_s[threadIdx.x >> 6] = c;
      __syncthreads();
      if (threadIdx.x == 0) { int tsum = 0; for (int w = 0; w < kBlock / 64; ++w) tsum += cnt_s[w]; total_s = tsum; }
      __syncthreads();
      if (total_s >= k) hi = mid; else lo = mid + 1;
      __syncthreads();
    }
  }
  const long base = (long)b * S * F;
  for (int cell = threadIdx.x; cell < S * F; cell += kBlock) {
    const int64_t id = ids_in[base + cell];
    const bool chosen = k > 0 && cell < n && smtp_cell_key(seed, b, cell) <= lo;
    ids_out[base + cell] = (chosen && id != pad_id) ? (int64_t)mask_id : id;
    labels_out[base + cell] = chosen ? id : (int64_t)-100;
  }
}

inline int grid_for(long work_items, int per_block = kBlock, int cap = 4096) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

// ================================================================================================
// host launchers
// ================================================================================================
// in-place element dropout of a [T][n] bf16 matrix (mlp_act_dropout on the gated activations; the same call masks dh in backward)
__global__ void __launch_bounds__(kBlock) elem_dropout_kernel(bf16_t* __restrict__ x, long T, int n, unsigned stream, ElemDropArg E) {
  const int cpr = n >> 3;
  const long total = T * cpr;
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < total; w += (long)gridDim.x * kBlock) {
    const unsigned t = elem_row(E, w / cpr), c = (unsigned)(w % cpr) * 8;
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(x + w * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= elem_drop_mul(E, stream, t, c + e);
    *reinterpret_cast<uint4*>(x + w * 8) = pack8(v);
  }
}
int k_elem_dropout(void* x, long T, int n, unsigned stream, ElemDropArg E, hipStream_t st) {
  if (T == 0 || E.thresh == 0) return 0;
  GGET_REQUIRE(n % 8 == 0, "elem_dropout: width must be a multiple of 8");
  const int g = (int)std::min<long>(4096, (T * (n / 8) + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(elem_dropout_kernel, dim3(g), dim3(kBlock), 0, st, (bf16_t*)x, T, n, stream, E);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_embed_fwd(const int64_t* ids, const void* emb, const void* gate, void* out, int T, int F, int ldF, int d,
                hipStream_t st, ElemDropArg E) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(T), dim3(128), 0, st, ids, (const bf16_t*)emb, (const bf16_t*)gate,
                     (bf16_t*)out, T, F, ldF, d, E);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_raw_blend(const float* raw, const int64_t* labels, int n, bool first_only, const void* tok, void* out, int32_t* flag, int T, int e,
                hipStream_t st, const int32_t* rows_map, int n_logical) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(raw_blend_kernel, dim3(T), dim3(128), 0, st, raw, labels, n, first_only ? 1 : 0, (const bf16_t*)tok, (bf16_t*)out, flag, e,
                     rows_map, rows_map ? n_logical : T);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_raw_tok_grad(const void* dx, const int32_t* flag, float* dtok, int T, int e, hipStream_t st) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(raw_tok_grad_kernel, dim3((T + 255) / 256), dim3(kBlock), 0, st, (const bf16_t*)dx, flag, dtok, T, e);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_embed_long_ratio(const int64_t* ids, void* x, int T, int F, int ldF, int d, hipStream_t st) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(embed_long_ratio_kernel, dim3(T), dim3(128), 0, st, ids, (bf16_t*)x, F, ldF, d);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_sample_mask_wgt(const int64_t* labels, float* w, int B, int cells, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(sample_mask_wgt_kernel, dim3(B), dim3(kBlock), 0, st, labels, w, cells);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_embed_bwd(const int64_t* ids, const void* dx, const void* emb, const void* gate, float* demb, float* dgate, int T,
                int F, int ldF, int d, int V, int pad_id, int32_t* sort_ws, hipStream_t st, ElemDropArg E) {
  if (T == 0) return 0;
  // sort_ws: hist[V] | offs[V+1] | cursor[V] | cell_sorted[T*F] | id_sorted[T*F]
  const long ncell = (long)T * F;
  int32_t* hist = sort_ws;
  int32_t* offs = hist + V;
  int32_t* cursor = offs + V + 1;
  int32_t* cell_sorted = cursor + V;
  int32_t* id_sorted = cell_sorted + ncell;
  GGET_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)V * sizeof(int32_t), st));
  const int g = (int)((ncell + kEmbCells - 1) / kEmbCells);
  const int hot_id = 1;  // <mask> token id of the SMTP collator (tokenizer_utils.py: mask id 1)
  if (V <= kEmbLdsV) {
    hipLaunchKernelGGL(embed_hist_kernel<true>, dim3(g), dim3(256), V * sizeof(int), st, ids, hist, ncell, F, ldF, pad_id,
                       V, hot_id);
  } else {
    hipLaunchKernelGGL(embed_hist_kernel<false>, dim3(g), dim3(256), 0, st, ids, hist, ncell, F, ldF, pad_id, V, hot_id);
  }
  hipLaunchKernelGGL(embed_scan_kernel, dim3(1), dim3(1024), 0, st, hist, offs, cursor, V);
  if (V <= kEmbLdsV) {
    hipLaunchKernelGGL(embed_fill_kernel<true>, dim3(g), dim3(256), 2 * V * sizeof(int), st, ids, cursor, cell_sorted,
                       id_sorted, ncell, F, ldF, pad_id, V, hot_id);
  } else {
    hipLaunchKernelGGL(embed_fill_kernel<false>, dim3(g), dim3(256), 0, st, ids, cursor, cell_sorted, id_sorted, ncell, F,
                       ldF, pad_id, V, hot_id);
  }
  hipLaunchKernelGGL(embed_reduce_kernel, dim3((int)((ncell + kEmbSeg - 1) / kEmbSeg)), dim3(128), 0, st, cell_sorted,
                     id_sorted, offs, V, (const bf16_t*)dx, (const bf16_t*)gate, demb, F, d, E);
  if (gate)
    hipLaunchKernelGGL(embed_dgate_kernel, dim3((T + kEmbTok - 1) / kEmbTok, F), dim3(128), 0, st, ids, (const bf16_t*)dx,
                       (const bf16_t*)emb, dgate, T, F, ldF, d, E);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_embed_count(const int64_t* ids, void* cnt, int T, int F, int ldF, int ldc, int pad_id, hipStream_t st) {
  if (T == 0) return 0;
  GGET_REQUIRE(F <= 256, "embed_count: counts above 256 are not exact in bf16 (F=%d)", F);
  hipLaunchKernelGGL(embed_count_kernel, dim3((int)(((long)T * F + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ids, (bf16_t*)cnt, T, F, ldF,
                     ldc, pad_id);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int T, int d, float eps, hipStream_t st) {
  GGET_REQUIRE(d % 8 == 0 && d <= 64 * 8 * kMaxChunksPerLane, "rmsnorm: d=%d unsupported", d);
  if (T == 0) return 0;
  if (d <= 1024) hipLaunchKernelGGL(rmsnorm_fwd_kernel<2>, dim3(grid_for(T, 4, 2048)), dim3(kBlock), 0, st, (const bf16_t*)x,
                                    (const bf16_t*)w, (bf16_t*)y, rstd, T, d, eps);
  else hipLaunchKernelGGL(rmsnorm_fwd_kernel<kMaxChunksPerLane>, dim3(grid_for(T, 4, 2048)), dim3(kBlock), 0, st, (const bf16_t*)x,
                          (const bf16_t*)w, (bf16_t*)y, rstd, T, d, eps);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                  float* dw_accum, int T, int d, hipStream_t st, int copies, uint64_t copy_stride) {
  GGET_REQUIRE(d % 8 == 0 && d <= 64 * 8 * kMaxChunksPerLane, "rmsnorm: d=%d unsupported", d);
  if (T == 0) return 0;
  if (copies < 1) copies = 1;
  static int rpw = 0;
  if (!rpw) { const char* e = getenv("GGET_RMS_ROWS"); rpw = e ? atoi(e) : 4; }
  const int grid = grid_for(T, 4 * rpw, 4096);  // rows per wave: amortises the dw atomics, pipelined row loads
  // Reproducible mode: the per-block partials of the weight gradient go to a scratch matrix and are summed in block order (the
  // fp32 atomics into the replicated accumulators are the one unordered sum of the pre-train gradient path: DESIGN.md section 5).
  float* part = nullptr;
  if (g_deterministic) {
    const size_t need = (size_t)(4096 + 64) * d * sizeof(float);   // per-block partials [4096][d] + segment sums [64][d]
    if (g_det_bytes < need) {
      if (g_det_scratch) GGET_HIP_CHECK(hipFree(g_det_scratch));
      GGET_HIP_CHECK(hipMalloc(&g_det_scratch, need));
      g_det_bytes = need;
    }
    part = g_det_scratch;
  }
  // short launches (<= 4 rows per wave of a 16-wave block per CU): the one-block-per-CU form (kernel comment); GGET_RMS_WIDE=0 / gget_debug_set(13, 0)
  // = the 4-wave blocks everywhere
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    GGET_HIP_CHECK(hipGetDevice(&dev));
    GGET_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int wgrid = grid;
  if (d <= 1024 && g_rms_bwd_wide && T <= n_cu * 16 * 4) {
    wgrid = (T + 15) / 16 < n_cu ? (T + 15) / 16 : n_cu;
    static bool attr_done = false;
    if (!attr_done) {
      GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rmsnorm_bwd_wide_kernel<2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024 * 4));
      attr_done = true;
    }
    hipLaunchKernelGGL((rmsnorm_bwd_wide_kernel<2, 16>), dim3(wgrid), dim3(1024), 16 * d * sizeof(float), st, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw_accum, T, d, copies, copy_stride, part);
  } else if (d <= 1024)
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<2>, dim3(grid), dim3(kBlock), 4 * d * sizeof(float), st, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw_accum, T, d, copies, copy_stride, part);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, dim3(grid), dim3(kBlock), 4 * d * sizeof(float), st, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw_accum, T, d, copies, copy_stride, part);
  if (part) {
    constexpr int kPer = 64;
    const int nseg = (wgrid + kPer - 1) / kPer;
    float* seg = part + (size_t)4096 * d;
    hipLaunchKernelGGL(ordered_colsum_kernel, dim3((d + 63) / 64, nseg), dim3(kBlock), 0, st, part, wgrid, d, kPer, seg, 0);
    hipLaunchKernelGGL(ordered_colsum_kernel, dim3((d + 63) / 64, 1), dim3(kBlock), 0, st, seg, nseg, d, nseg, dw_accum, 1);
  }
  GGET_LAUNCH_CHECK();
  return 0;
}

void k_set_deterministic(int on) { g_deterministic = on; }
void k_set_rms_wide(int on) { g_rms_bwd_wide = on; }
void k_set_ce_parts(int on) { g_ce_parts = on; }
int k_get_deterministic() { return g_deterministic; }

int k_rope(void* qkv, const float* cos_tab, const float* sin_tab, const int64_t* position_ids, int T, int S, int H,
           int inverse, hipStream_t st) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for((long)T * 8 * H)), dim3(kBlock), 0, st, (bf16_t*)qkv, cos_tab, sin_tab,
                     position_ids, T, S, H, inverse);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_rope_table(float* cos_tab, float* sin_tab, int max_pos, float theta, hipStream_t st) {
  hipLaunchKernelGGL(rope_table_kernel, dim3((max_pos * 32 + 255) / 256), dim3(256), 0, st, cos_tab, sin_tab, max_pos,
                     theta);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_rope_range_table(const int64_t* pos, float* cos_t, float* sin_t, int64_t* ids, int B, int S, float range, float theta,
                       hipStream_t st) {
  if (B * S == 0) return 0;
  hipLaunchKernelGGL(rope_range_table_kernel, dim3(B), dim3(kBlock), 0, st, pos, cos_t, sin_t, ids, S, range, theta);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_geglu_fwd(const void* gu, void* h, int T, int ff, hipStream_t st) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for((long)T * (ff / 8))), dim3(kBlock), 0, st, (const bf16_t*)gu,
                     (bf16_t*)h, (long)T, ff);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_geglu_bwd(const void* gu, const void* dh, void* dgu, int T, int ff, hipStream_t st) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for((long)T * (ff / 8))), dim3(kBlock), 0, st, (const bf16_t*)gu,
                     (const bf16_t*)dh, (bf16_t*)dgu, (long)T, ff);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_lengths(const int64_t* mask, const int64_t* ids, int ldF, int pad_id, int32_t* key_len, int32_t* pool_row, int B,
              int S, hipStream_t st) {
  hipLaunchKernelGGL(lengths_kernel, dim3(B), dim3(64), 0, st, mask, ids, ldF, pad_id, key_len, pool_row, B, S);
  GGET_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Var-len (padding-free) token layout.  The reference runs every token-wise op over the padded [B,S] grid (modeling_helpers.py:38-64
// builds the additive mask; a third of a PCQM4M-v2 batch is padding); here the real tokens of a right-padded batch are compacted once,
// sample after sample, and the whole layer stack runs on T = round_up(sum(len), 64) rows.  cu[b] = first row of sample b.
// ---------------------------------------------------------------------------------------------
// one block: exclusive scan of key_len over the batch (any B), cu[B] = total; the pooled row of the task head (lengths_kernel: b*S + p)
// moves to cu[b] + min(p, len - 1); status[0] = 1 when the total differs from the caller's token count (status[2]: the same, sticky -
// gget_deferred_status reads and clears it)
// long_list (may be NULL; int32 [3 B + 1]): [0] = number n of samples of 33 .. 64 rows, [1 .. n] their indices in ascending order, [1 + B ..]
// their first rows, [1 + 2 B ..] their row counts - the work list of the launch that takes such samples apart from the one-tile ones
// (attention.hip: attn_bwd_long_kernel; one read gives a block its whole work item)
__global__ void __launch_bounds__(1024) varlen_scan_kernel(int32_t* __restrict__ key_len, int32_t* __restrict__ cu,
                                                           int32_t* __restrict__ pool_row, int32_t* __restrict__ status, int B, int S,
                                                           int expect_total, int32_t* __restrict__ long_list) {
  __shared__ int wsum[16];
  __shared__ int lsum[16];
  __shared__ int carry_s, lcarry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry_s = 0; lcarry_s = 0; }
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    const int v = b < B ? key_len[b] : 0;
    int x = v;   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int base = carry_s;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int len = 0;
    if (b < B) {
      // (a caller's count SMALLER than the mask's total - flagged below, the step's results are invalid - must still not send any
      //  kernel behind the rows the step runs on: samples are cut at expect_total)
      const int start = min(base + x - v, expect_total);
      len = min(v, expect_total - start);
      cu[b] = start;
      if (len != v) key_len[b] = len;
      if (pool_row) {
        const int pidx = pool_row[b] - b * S;
        pool_row[b] = min(start + max(min(pidx, len - 1), 0), max(expect_total - 1, 0));
      }
    }
    // the 33 .. 64-row samples, in batch order (ballot + prefix count: deterministic)
    const bool is_long = len > 32 && len <= 64;
    const unsigned long long bal = __ballot(is_long);
    if (lane == 0) lsum[wave] = __popcll(bal);
    __syncthreads();
    if (long_list && is_long) {
      int lpos = lcarry_s;
      for (int w = 0; w < wave; ++w) lpos += lsum[w];
      lpos += __popcll(bal & ((1ull << lane) - 1ull));
      long_list[1 + lpos] = b;
      long_list[1 + B + lpos] = cu[b];        // (this thread wrote cu[b] above)
      long_list[1 + 2 * B + lpos] = len;
    }
    __syncthreads();
    if (tid == 1023) {
      carry_s = base + x;
      int lt = lcarry_s;
      for (int w = 0; w < 16; ++w) lt += lsum[w];
      lcarry_s = lt;
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (long_list) long_list[0] = lcarry_s;
    cu[B] = min(carry_s, expect_total);
    status[0] = carry_s == expect_total ? 0 : 1;
    if (carry_s != expect_total) status[2] = 1;
  }
}
// one thread per padded token (b, s) plus one per tail row: compact ids / positions / sample index of its row, the padded -> compact map
__global__ void __launch_bounds__(kBlock) varlen_fill_kernel(const int64_t* __restrict__ ids, int ldF, int F, const int64_t* __restrict__ pos,
                                                             const int32_t* __restrict__ key_len, const int32_t* __restrict__ cu,
                                                             int64_t* __restrict__ ids_c, int64_t* __restrict__ pos_c,
                                                             int32_t* __restrict__ row_b, int32_t* __restrict__ pad2c,
                                                             int32_t* __restrict__ c2p, int B, int S, int tc, int t_rows, int64_t pad_id) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  const long TP = (long)B * S;
  if (i < TP) {
    const int b = (int)(i / S), sq = (int)(i % S);
    if (sq < key_len[b]) {   // (key_len was cut by the scan kernel where a too-small caller's count would overrun the rows of the step)
      const int r = cu[b] + sq;
      for (int f = 0; f < F; ++f) ids_c[(size_t)r * F + f] = ids[(size_t)i * ldF + f];
      pos_c[r] = pos ? pos[i] : (int64_t)sq;
      row_b[r] = b;
      pad2c[i] = r;
      c2p[r] = (int)i;        // the row's logical coordinate b * S + s (element-dropout hashes, rope_range tables)
    } else {
      pad2c[i] = -1;
    }
  } else if (i < TP + (t_rows - tc)) {   // tail rows [tc, t_rows): pad tokens of sample 0, position 0 (finite activations, zero gradients)
    const int r = tc + (int)(i - TP);
    for (int f = 0; f < F; ++f) ids_c[(size_t)r * F + f] = pad_id;
    pos_c[r] = 0;
    row_b[r] = 0;
    c2p[r] = (int)i;          // (behind the padded grid: a coordinate no real token has)
  }
}
// sum of the key lengths of a batch (one block): the real-token count the host reads back when it asked the engine to count
// (gget_set_token_count(GGET_TOKENS_AUTO))
// host_out (may be NULL): a word of pinned host memory - the kernel stores the count there itself and the host reads it behind an event
// with the system-scope fence (no device-to-host copy packet in the stream: a 4 us blit kernel + a 5.6 us bubble before round 6)
__global__ void __launch_bounds__(1024) sum_lengths_kernel(const int32_t* __restrict__ key_len, int B, int32_t* __restrict__ out,
                                                           int32_t* __restrict__ host_out) {
  __shared__ int tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  int v = 0;
  for (int b = threadIdx.x; b < B; b += 1024) v += key_len[b];
  if (v) atomicAdd(&tot, v);
  __syncthreads();
  if (threadIdx.x == 0) {
    *out = tot;
    if (host_out) __hip_atomic_store(host_out, tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// position ids handed to a forward index the precomputed RoPE table [max_pos][32]: a copy clamped to [0, max_pos) keeps every table read
// in bounds, and a sticky flag tells the host (at a time of its choosing, not per step) that a clamp happened - the reference evaluates
// the rotary embedding on the fly and accepts any position (hf LlamaRotaryEmbedding.forward :111-127), so an out-of-range position is a
// configuration error of the caller (max_position_embeddings too small), reported instead of silently diverging
__global__ void __launch_bounds__(kBlock) clamp_positions_kernel(const int64_t* __restrict__ pos, int64_t* __restrict__ out,
                                                                 int32_t* __restrict__ flag, long n, int max_pos) {
  bool bad = false;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
    const int64_t v = pos[i];
    bad |= v < 0 || v >= max_pos;
    out[i] = v < 0 ? 0 : (v >= max_pos ? max_pos - 1 : v);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1;
}
int k_clamp_positions(const int64_t* pos, int64_t* out, int32_t* flag, long n, int max_pos, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(clamp_positions_kernel, dim3(grid_for(n, kBlock, 1024)), dim3(kBlock), 0, st, pos, out, flag, n, max_pos);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_varlen_plan(const int64_t* ids, int ldF, int F, const int64_t* pos, int32_t* key_len, int32_t* pool_row, int32_t* cu,
                  int64_t* ids_c, int64_t* pos_c, int32_t* row_b, int32_t* pad2c, int32_t* c2p, int32_t* status, int B, int S, int tc,
                  int t_rows, int pad_id, hipStream_t st, int32_t* long_list) {
  hipLaunchKernelGGL(varlen_scan_kernel, dim3(1), dim3(1024), 0, st, key_len, cu, pool_row, status, B, S, tc, long_list);
  const long n = (long)B * S + (t_rows - tc);
  hipLaunchKernelGGL(varlen_fill_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ids, ldF, F, pos, key_len, cu,
                     ids_c, pos_c, row_b, pad2c, c2p, B, S, tc, t_rows, (int64_t)pad_id);
  GGET_LAUNCH_CHECK();
  return 0;
}
// Host -> device hand-over of a collated batch as a KERNEL: the source is pinned (device-mapped) host memory, read over the host link
// with 16-byte loads (1.8 MB for a C1 batch: ~35 us).  An in-stream hipMemcpyAsync from pinned memory goes to the copy engine and every
// such copy is a cross-engine dependency the runtime resolves on the host: four of them per step took the C1 step from 6.7 to 12 - 20 ms
// (tools/prefetch_probe.py); a kernel stays in the compute queue.
__global__ void __launch_bounds__(kBlock) copy_from_host_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t n) {
  const size_t n16 = n / 16;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock)
    reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  if (blockIdx.x == 0)
    for (size_t i = n16 * 16 + threadIdx.x; i < n; i += kBlock) dst[i] = src[i];
}
int k_copy_from_host(const void* src_host_mapped, void* dst, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  GGET_REQUIRE((((uintptr_t)src_host_mapped | (uintptr_t)dst) & 15) == 0, "copy_from_host: 16-byte aligned buffers");
  hipLaunchKernelGGL(copy_from_host_kernel, dim3(grid_for((long)(bytes / 16 + 1), kBlock, 256)), dim3(kBlock), 0, st, (const unsigned char*)src_host_mapped,
                     (unsigned char*)dst, bytes);
  GGET_LAUNCH_CHECK();
  return 0;
}

// compact rows -> the padded [B,S] grid (hidden-state accessors after a var-len forward): out[i] = src[pad2c[i]], zero where the padded
// position holds no token; one thread per 16-byte piece
__global__ void __launch_bounds__(kBlock) rows_to_grid_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ pad2c,
                                                              bf16_t* __restrict__ out, long n_pos, int d) {
  const int cpr = d / 8;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n_pos * cpr; i += (long)gridDim.x * kBlock) {
    const long pos = i / cpr;
    const int c = (int)(i % cpr);
    const int r = pad2c[pos];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r >= 0) v = *reinterpret_cast<const uint4*>(src + (size_t)r * d + c * 8);
    *reinterpret_cast<uint4*>(out + (size_t)pos * d + c * 8) = v;
  }
}
int k_rows_to_grid(const void* src, const int32_t* pad2c, void* out, long n_pos, int d, hipStream_t st) {
  if (n_pos == 0) return 0;
  hipLaunchKernelGGL(rows_to_grid_kernel, dim3(grid_for(n_pos * (d / 8), kBlock, 4096)), dim3(kBlock), 0, st, (const bf16_t*)src, pad2c, (bf16_t*)out,
                     n_pos, d);
  GGET_LAUNCH_CHECK();
  return 0;
}

// a var-len step that ran on a caller's token count the mask contradicts computed garbage: its loss becomes NaN (and the sticky flag
// status[2] stays up for gget_deferred_status) instead of a plausible number
__global__ void poison_loss_kernel(const int32_t* __restrict__ flag, float* __restrict__ loss) {
  if (*flag) *loss = __builtin_nanf("");
}
int k_poison_loss(const int32_t* flag, float* loss, hipStream_t st) {
  hipLaunchKernelGGL(poison_loss_kernel, dim3(1), dim3(1), 0, st, flag, loss);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_sum_lengths(const int32_t* key_len, int B, int32_t* out, hipStream_t st, int32_t* host_out) {
  hipLaunchKernelGGL(sum_lengths_kernel, dim3(1), dim3(1024), 0, st, key_len, B, out, host_out);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_head_slot_sort(const int32_t* sel_src, const int32_t* row_idx, const int32_t* lm_count, int32_t* slot_state, int32_t* a_tok,
                     int32_t* a_cell, int32_t* c_l, int32_t* cellpos, int32_t* tile_off, int32_t* total_p, int cap, int n, long slot_elems,
                     int tile_rows, hipStream_t st) {
  if (n > 32) { gget_set_error("slot-sorted head: next_n_token %d > 32", n); return 2; }
  // (slot_state[0, n) holds the cells per slot, counted by k_head_compact with slot_hist = slot_state; [n, 2 n) and the ticket are zero:
  //  cleared with the workspace at creation and by the last block of every slot_fill_kernel)
  hipLaunchKernelGGL(slot_fill_kernel, dim3((cap + kBlock * kFillItems - 1) / (kBlock * kFillItems)), dim3(kBlock), 0, st, sel_src, row_idx, lm_count, slot_state, a_tok,
                     a_cell, c_l, cellpos, tile_off, total_p, cap, n, slot_elems, tile_rows);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_head_cell_sum(const void* dxs, const int32_t* cellpos, const int32_t* cnt, const int32_t* l_off, const int32_t* pad2c, void* dhid,
                    int TP, int d, int pad_row, hipStream_t st) {
  hipLaunchKernelGGL(head_cell_sum_kernel, dim3((TP + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, (const bf16_t*)dxs, cellpos, cnt,
                     l_off, pad2c, (bf16_t*)dhid, TP, d, pad_row);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_head_compact(const int64_t* labels, int T, int n, int32_t* cnt, int32_t* m_off, int32_t* l_off, int32_t* counts,
                   int32_t* row_idx, int32_t* sel_src, int32_t* sel_label, int32_t* sel_tok, int32_t* blk_tot, int32_t* slot_hist,
                   hipStream_t st) {
  const int g = (T + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(head_count_kernel, dim3(g), dim3(kBlock), 0, st, labels, cnt, blk_tot, T, n);
  hipLaunchKernelGGL(head_fill_kernel, dim3(g), dim3(kBlock), 0, st, labels, cnt, blk_tot, m_off, l_off, counts, row_idx, sel_src,
                     sel_label, sel_tok, n <= 32 ? slot_hist : nullptr, T, n);
  GGET_LAUNCH_CHECK();
  return 0;
}
size_t k_head_compact_ws_bytes(int T) { return (size_t)((T + kBlock - 1) / kBlock) * 8; }

int k_gather_rows(const void* src, const int32_t* idx, const int32_t* count, void* dst, int cap, int d, int scatter,
                  hipStream_t st) {
  if (cap == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)cap * (d / 8))), dim3(kBlock), 0, st, (const bf16_t*)src,
                     idx, count, (bf16_t*)dst, cap, d, scatter);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_gather_rows_remap(const void* src, int32_t* idx, const int32_t* count, const int32_t* pad2c, void* dst, int cap, int d, int pad_row,
                        int32_t* status, hipStream_t st) {
  if (cap == 0) return 0;
  hipLaunchKernelGGL(gather_rows_remap_kernel, dim3(grid_for((long)cap * 64)), dim3(kBlock), 0, st, (const bf16_t*)src, idx, count, pad2c,
                     (bf16_t*)dst, cap, d, pad_row, status);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_ce_fwd_bwd(const void* logits, int ld, const int32_t* labels, const int32_t* sel_tok, const float* sample_wgt, int S,
                 const int32_t* n_rows_dev, int n_rows_cap, int V, float* loss_sum, void* dlogits, float scale_base,
                 int mean_over_rows, float* loss_out, hipStream_t st, float focal_gamma, float* loss_part, int loss_part_cap) {
  const bool vec = (ld % 8) == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0 && getenv("GGET_CE_GENERIC") == nullptr;
  const dim3 grid(grid_for(n_rows_cap, 4 * 8, 2048));
  // per-block partial sums instead of same-address atomics: the row-in-registers kernels, when the caller has a slot per block
  // (gget_debug_set(14, 0): atomics everywhere)
  const bool parts = g_ce_parts && loss_part && n_rows_cap > 0 && vec && ld <= 2048 && (int)grid.x <= loss_part_cap;
  if (!parts) GGET_HIP_CHECK(hipMemsetAsync(loss_sum, 0, sizeof(float), st));
  if (n_rows_cap > 0) {
    float* lp = parts ? loss_part : nullptr;
#define GGET_CE_ARGS (const bf16_t*)logits, ld, labels, sel_tok, sample_wgt, S, n_rows_dev, n_rows_cap, V, loss_sum, \
                     (bf16_t*)dlogits, scale_base, mean_over_rows, focal_gamma
    if (vec && ld <= 512) hipLaunchKernelGGL(ce_rows_kernel<1>, grid, dim3(kBlock), 0, st, GGET_CE_ARGS, lp);
    else if (vec && ld <= 1024) hipLaunchKernelGGL(ce_rows_kernel<2>, grid, dim3(kBlock), 0, st, GGET_CE_ARGS, lp);
    else if (vec && ld <= 2048) hipLaunchKernelGGL(ce_rows_kernel<4>, grid, dim3(kBlock), 0, st, GGET_CE_ARGS, lp);
    else hipLaunchKernelGGL(ce_fwd_bwd_kernel, grid, dim3(kBlock), 0, st, GGET_CE_ARGS);
#undef GGET_CE_ARGS
  }
  if (loss_out || parts)
    hipLaunchKernelGGL(finalize_loss_kernel, dim3(1), dim3(256), 0, st, loss_sum, n_rows_dev, scale_base, mean_over_rows, loss_out,
                       parts ? loss_part : nullptr, (int)grid.x);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_tok_score_fwd(const void* hidden, const void* w, const void* bias, float* logits, int T, int C, int d, hipStream_t st) {
  GGET_REQUIRE(d % 64 == 0 && d <= 64 * kTokMaxCols, "token-level head: d=%d unsupported", d);
  if (T == 0) return 0;
  const int rows_per_block = (kBlock / 64) * kTokRows;
  hipLaunchKernelGGL(tok_score_fwd_kernel, dim3((T + rows_per_block - 1) / rows_per_block), dim3(kBlock), 0, st, (const bf16_t*)hidden,
                     (const bf16_t*)w, (const bf16_t*)bias, logits, T, C, d);
  GGET_LAUNCH_CHECK();
  return 0;
}
// dst[rows_map[t], :] = src[t, :] for the rows inside the logical grid (token-level logits of the var-len layout back to [B,S,C] order)
__global__ void __launch_bounds__(kBlock) scatter_rows_map_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows_map,
                                                                      float* __restrict__ dst, long T, int C, int n_logical) {
  for (long w = (long)blockIdx.x * kBlock + threadIdx.x; w < T * C; w += (long)gridDim.x * kBlock) {
    const long t = w / C;
    const int lt = rows_map[t];
    if (lt < n_logical) dst[(size_t)lt * C + (w % C)] = src[w];
  }
}
int k_scatter_rows_map_f32(const float* src, const int32_t* rows_map, float* dst, int T, int C, int n_logical, hipStream_t st) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(scatter_rows_map_f32_kernel, dim3(grid_for((long)T * C, kBlock, 2048)), dim3(kBlock), 0, st, src, rows_map, dst, (long)T, C,
                     n_logical);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_tok_ce(const float* logits, const int64_t* labels, float* dl, float* stat, float* loss_out, int T, int C, hipStream_t st,
             const int32_t* rows_map, int n_logical) {
  GGET_HIP_CHECK(hipMemsetAsync(stat, 0, 4 * sizeof(float), st));
  if (T > 0)
    hipLaunchKernelGGL(tok_ce_kernel, dim3(grid_for(T, kBlock, 1024)), dim3(kBlock), 0, st, logits, labels, dl, stat, T, C, rows_map,
                       rows_map ? n_logical : T);
  hipLaunchKernelGGL(tok_ce_final_kernel, dim3(1), dim3(1), 0, st, stat, loss_out);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_tok_score_bwd(const float* dl, const float* stat, const void* hidden, const void* w, float* dw, float* dbias, void* dhidden, int T,
                    int C, int d, hipStream_t st) {
  if (T == 0) return 0;
  const int rows_per_block = (kBlock / 64) * kTokRows;
  hipLaunchKernelGGL(tok_score_bwd_dx_kernel, dim3((T + rows_per_block - 1) / rows_per_block), dim3(kBlock), 0, st, dl, stat,
                     (const bf16_t*)w, (bf16_t*)dhidden, T, C, d);
  hipLaunchKernelGGL(tok_score_bwd_dw_kernel, dim3((T + kTokSlab - 1) / kTokSlab, (C + kTokCls - 1) / kTokCls), dim3(kBlock), 0, st, dl,
                     stat, (const bf16_t*)hidden, dw, dbias, T, C, d);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_score_fwd(const void* hidden, const int32_t* pool_row, const void* w, const void* bias, float* logits,
                void* pooled_h, int B, int C, int d, hipStream_t st) {
  hipLaunchKernelGGL(score_fwd_kernel, dim3(B * C), dim3(64), 0, st, (const bf16_t*)hidden, pool_row, (const bf16_t*)w,
                     (const bf16_t*)bias, logits, (bf16_t*)pooled_h, B, C, d);
  GGET_LAUNCH_CHECK();
  return 0;
}

// AUC surrogate loss (src/utils/loss_utils.py:25-53, selected by loss_type == "auc" at modeling_finetune.py:203-207):
// y = logit[:,1] - logit[:,0]; every positive sample is paired with num_neg negatives drawn as
// idx = randperm(P * num_neg) % N_neg; loss = mean (1 - (y_pos - y_neg))^2.  torch's randperm cannot be reproduced: the
// permutation here is the rank of the 24-bit counter hash of (seed, stream 40, i) with ties broken by i
// (graph-gpt_amd/modeling.py:auc_pairs is the Python twin the parity test feeds to the oracle).  One block; P * num_neg <= 8192.
constexpr int kAucMaxPairs = 8192;
__global__ void __launch_bounds__(kBlock) auc_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int B,
                                                          int C, int num_neg, unsigned seed, float* __restrict__ loss_out,
                                                          float* __restrict__ dlogits, int32_t* __restrict__ lists /* [2B] */) {
  __shared__ unsigned keys[kAucMaxPairs];
  __shared__ float red[kBlock];
  __shared__ int cnt_s[kBlock + 1];
  const int tid = threadIdx.x;
  // ordered lists of the positive / negative samples (chunk per thread, exclusive scan over the threads)
  const int per = (B + kBlock - 1) / kBlock, b0 = min(B, tid * per), b1 = min(B, b0 + per);
  int np_ = 0;
  for (int b = b0; b < b1; ++b) np_ += labels[b] != 0;
  cnt_s[tid + 1] = np_;
  if (tid == 0) cnt_s[0] = 0;
  __syncthreads();
  if (tid == 0) for (int i = 1; i <= kBlock; ++i) cnt_s[i] += cnt_s[i - 1];
  __syncthreads();
  const int P = cnt_s[kBlock], N = B - P;
  {
    int ip = cnt_s[tid], in = b0 - ip;
    for (int b = b0; b < b1; ++b) {
      if (labels[b] != 0) lists[ip++] = b; else lists[B + in++] = b;
    }
  }
  for (int i = tid; i < B * C; i += kBlock) dlogits[i] = 0.f;
  const int cnt = P * num_neg;
  for (int i = tid; i < cnt; i += kBlock) keys[i] = smtp_rng(seed, 40, (unsigned)i, 0);
  __syncthreads();
  float local = 0.f;
  if (cnt > 0 && N > 0) {
    const float inv = 1.0f / (float)cnt;
    for (int i = tid; i < cnt; i += kBlock) {
      const unsigned ki = keys[i];
      int rank = 0;
      for (int j = 0; j < cnt; ++j) rank += (keys[j] < ki) || (keys[j] == ki && j < i);
      const int bp = lists[i / num_neg], bn = lists[B + rank % N];
      const float yp = logits[bp * C + 1] - logits[bp * C], yn = logits[bn * C + 1] - logits[bn * C];
      const float t = 1.f - (yp - yn);
      local += t * t * inv;
      const float g = 2.f * t * inv;           // d loss / d yn = +g, d loss / d yp = -g
      atomicAdd(&dlogits[bp * C + 1], -g); atomicAdd(&dlogits[bp * C], g);
      atomicAdd(&dlogits[bn * C + 1], g); atomicAdd(&dlogits[bn * C], -g);
    }
  }
  red[tid] = local;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  // (no positive or no negative sample: torch's mean over an empty tensor is NaN - so is this)
  if (tid == 0) loss_out[0] = (cnt > 0 && N > 0) ? red[0] : __builtin_nanf("");
}

int k_auc_loss(const float* logits, const int64_t* labels, int B, int C, int num_neg, unsigned seed, float* loss_out,
               float* dlogits, int32_t* lists, hipStream_t st) {
  hipLaunchKernelGGL(auc_loss_kernel, dim3(1), dim3(kBlock), 0, st, logits, labels, B, C, num_neg, seed, loss_out, dlogits, lists);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_task_loss(const float* logits, const void* labels, const float* sample_wgt, int problem, int B, int C,
                float* loss_out, float* dlogits, hipStream_t st) {
  hipLaunchKernelGGL(task_loss_kernel, dim3(1), dim3(kBlock), 0, st, logits, labels, sample_wgt, problem, B, C, loss_out,
                     dlogits);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_pool_rows(const void* hidden, const int32_t* pool_row, void* out, int B, int d, hipStream_t st) {
  hipLaunchKernelGGL(pool_rows_kernel, dim3(B), dim3(kBlock), 0, st, (const bf16_t*)hidden, pool_row, (bf16_t*)out, B, d);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_head_linear_fwd(const void* x, void* a, const void* w, const void* bias, void* y, float* y32, int B, int Din, int Dout,
                      int layer, ElemDropArg E, hipStream_t st) {
  hipLaunchKernelGGL(head_act_kernel, dim3(grid_for((long)B * Din, kBlock, 1024)), dim3(kBlock), 0, st, (const bf16_t*)x, (bf16_t*)a, B,
                     Din, (unsigned)layer, E);
  hipLaunchKernelGGL(head_linear_fwd_kernel, dim3(B * Dout), dim3(64), 0, st, (const bf16_t*)a, (const bf16_t*)w, (const bf16_t*)bias,
                     (bf16_t*)y, y32, B, Din, Dout);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_head_linear_bwd(const float* dy, const void* x, const void* a, const void* w, float* dw, float* dbias, float* dx, int B,
                      int Din, int Dout, int layer, ElemDropArg E, hipStream_t st) {
  hipLaunchKernelGGL(head_linear_bwd_kernel, dim3(B + Dout), dim3(kBlock), 0, st, dy, (const bf16_t*)x, (const bf16_t*)a,
                     (const bf16_t*)w, dw, dbias, dx, B, Din, Dout, (unsigned)layer, E);
  GGET_LAUNCH_CHECK();
  return 0;
}
int k_scatter_rows_f32(const float* src, const int32_t* pool_row, void* dhidden, int B, int d, hipStream_t st) {
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3(B), dim3(kBlock), 0, st, src, pool_row, (bf16_t*)dhidden, B, d);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_score_bwd(const float* dlogits, const void* hidden, const int32_t* pool_row, const void* w, float* dw, float* dbias,
                void* dhidden, int B, int C, int d, hipStream_t st) {
  hipLaunchKernelGGL(score_bwd_kernel, dim3(B + C * kScoreSplit), dim3(kBlock), 0, st, dlogits, (const bf16_t*)hidden, pool_row,
                     (const bf16_t*)w, dw, dbias, (bf16_t*)dhidden, B, C, d);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_grad_sqnorm(const void* g, size_t n, float* ws, hipStream_t st) {
  const int blocks = grid_for((long)(n / 8), kBlock, kSqnormBlocks);   // (2048 .. 8192 blocks measured slower)
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(blocks), dim3(kBlock), 0, st, (const bf16_t*)g, n, ws);
  hipLaunchKernelGGL(grad_sqnorm_final_kernel, dim3(1), dim3(kBlock), 0, st, ws, blocks);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_grad_sqnorm_chunks(const void* g, const GgetSqChunk* chunks_dev, int nchunks, const float* extra, int nextra, float* ws, hipStream_t st) {
  if (nchunks > 0)
    hipLaunchKernelGGL(grad_sqnorm_chunks_kernel, dim3(nchunks), dim3(kBlock), 0, st, (const bf16_t*)g, chunks_dev, ws);
  hipLaunchKernelGGL(grad_sqnorm_final2_kernel, dim3(1), dim3(kBlock), 0, st, ws, nchunks, extra, nextra);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_adamw(float* master, float* m, float* v, const void* grad, void* param, size_t n, float lr, float beta1, float beta2,
            float eps, float wd, int step, float max_norm, float grad_scale, const float* sqnorm, float* gnorm_out,
            hipStream_t st, bool skip_nonfinite) {
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  // one pass, two float4 groups per thread (grid up to 65536 blocks): the grid-stride form with 4096 blocks ran at 5.1 TB/s of
  // state traffic, this one at 6.0 (profiles/r02_adamw_sweep.txt); loads and the fp32 state stores are non-temporal
  hipLaunchKernelGGL(adamw_kernel<2>, dim3(grid_for((long)(n / 8), kBlock, 65536)), dim3(kBlock), 0, st, master, m, v,
                     (const bf16_t*)grad, (bf16_t*)param, n, lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2), max_norm,
                     grad_scale, sqnorm, gnorm_out, skip_nonfinite ? 1 : 0);
  GGET_LAUNCH_CHECK();
  return 0;
}

// Up to kZeroRanges device ranges cleared by ONE launch (the backward's accumulators, the head's scatter targets and split-K slabs: five
// hipMemsetAsync calls = five runtime fill kernels of 4 - 15 us each at the start of every backward before).  16-byte non-temporal stores
// over the aligned body of every range (plain stores: the non-temporal form ran at 4.4 TB/s, the runtime's fill at 7.8), byte stores for what is left at its ends.
__global__ void __launch_bounds__(kBlock) zero_ranges_kernel(GgetZeroRanges R) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, stride = (size_t)gridDim.x * kBlock;
  for (int r = 0; r < R.n; ++r) {
    unsigned char* p = reinterpret_cast<unsigned char*>(R.ptr[r]);
    const size_t bytes = R.bytes[r];
    const size_t head = std::min<size_t>(bytes, (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15);
    const size_t nv = (bytes - head) >> 4, tail0 = head + (nv << 4);
    uint4* v = reinterpret_cast<uint4*>(p + head);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    size_t i = tid;
    for (; i + 3 * stride < nv; i += 4 * stride) { v[i] = z; v[i + stride] = z; v[i + 2 * stride] = z; v[i + 3 * stride] = z; }
    for (; i < nv; i += stride) v[i] = z;
    if (tid < head) p[tid] = 0;
    if (tid < bytes - tail0) p[tail0 + tid] = 0;
  }
}
int k_zero_ranges(const GgetZeroRanges& R, hipStream_t st) {
  size_t total = 0;
  for (int r = 0; r < R.n; ++r) total += R.bytes[r];
  if (total == 0) return 0;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(grid_for((long)(total / 64 + 1), kBlock, 8192)), dim3(kBlock), 0, st, R);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for((long)(n / 4), kBlock, 4096)), dim3(kBlock), 0, st, src,
                     (bf16_t*)dst, n);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_slab_reduce(const float* slabs, long slab_stride, int nslab, void* dst, size_t n, hipStream_t st, bool f32_out) {
  if (f32_out)
    hipLaunchKernelGGL(slab_reduce_kernel<true>, dim3(grid_for((long)(n / 4), kBlock, 2048)), dim3(kBlock), 0, st, slabs,
                       slab_stride, nslab, dst, n);
  else
    hipLaunchKernelGGL(slab_reduce_kernel<false>, dim3(grid_for((long)(n / 4), kBlock, 2048)), dim3(kBlock), 0, st, slabs,
                       slab_stride, nslab, dst, n);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_convert_segments(const float* scratch, void* grads, const GgetSegment* segs_dev, int nseg, hipStream_t st) {
  if (nseg == 0) return 0;
  hipLaunchKernelGGL(convert_segments_kernel, dim3(64, nseg), dim3(kBlock), 0, st, scratch, (bf16_t*)grads, segs_dev);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_smtp2d(const int64_t* ids_in, int ld_in, const int64_t* node_idx, int ld_node, int64_t* ids_out, int64_t* labels_out,
             int B, int S, int F, float rate, float power, float replace_rate, int vocab, int global_mask, unsigned seed,
             hipStream_t st) {
  if ((long)B * S * F == 0) return 0;
  hipLaunchKernelGGL(smtp2d_kernel, dim3(grid_for((long)B * S * F)), dim3(kBlock), 0, st, ids_in, ld_in, node_idx, ld_node,
                     ids_out, labels_out, B, S, F, rate, power, replace_rate, vocab, global_mask, seed, 1);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_token_confidence(const void* logits, int ld, int R, int V, int mode, float* conf, int64_t* tok, hipStream_t st) {
  if (R == 0) return 0;
  hipLaunchKernelGGL(token_confidence_kernel, dim3(grid_for(R, kBlock / 64, 4096)), dim3(kBlock), 0, st, (const bf16_t*)logits,
                     ld, R, V, mode, conf, tok);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_token_sample(const void* logits, int ld, int R, int V, int mode, float temperature, float top_p, int top_k, float alg_temp,
                   unsigned seed, float* conf, int64_t* tok, hipStream_t st) {
  if (R == 0) return 0;
  GGET_REQUIRE(V >= 1 && V <= 8192, "token_sample: vocabulary %d out of range (1..8192: the row is filtered in LDS)", V);
  GGET_REQUIRE(temperature >= 0.f && alg_temp >= 0.f && top_k >= 0, "token_sample: negative temperature / alg_temp / top_k");
  const int waves = V <= 1024 ? kBlock / 64 : 1;
  const size_t lds = (size_t)waves * 2 * V * sizeof(float);
  hipLaunchKernelGGL(token_sample_kernel, dim3(grid_for(R, waves, 8192)), dim3(kBlock), lds, st, (const bf16_t*)logits, ld, R, V,
                     mode, temperature, top_p, top_k, alg_temp, seed, conf, tok, waves);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_unmask_origin(int64_t* x, const int64_t* cand, int B, int N, float p_transfer, unsigned seed, int mask_id, hipStream_t st) {
  if ((long)B * N == 0) return 0;
  hipLaunchKernelGGL(unmask_origin_kernel, dim3(grid_for((long)B * N)), dim3(kBlock), 0, st, x, cand, B, N, p_transfer, seed, mask_id);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_ranges_from_mask3d(const int64_t* mask3d, int32_t* key_lo, int32_t* key_hi, int B, int S, hipStream_t st) {
  if (B * S == 0) return 0;
  hipLaunchKernelGGL(ranges_from_mask3d_kernel, dim3(grid_for((long)B * S, kBlock / 64, 4096)), dim3(kBlock), 0, st, mask3d,
                     key_lo, key_hi, B * S, S);
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_smtp_rows(const int64_t* ids_in, const int32_t* lengths, int64_t* ids_out, int64_t* labels_out, float* wgt_out, int B, int S,
                int F, double umr_min, double umr_max, double power, unsigned seed, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(smtp_rows_kernel, dim3(B), dim3(kBlock), 0, st, ids_in, lengths, ids_out, labels_out, wgt_out, S, F, umr_min,
                     umr_max, power, seed, 1, 0);
  GGET_LAUNCH_CHECK();
  return 0;
}
