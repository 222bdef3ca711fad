// Self-attention core of the Graph Eulerian Transformer on gfx950 MFMA (32x32x16 bf16).
// reference: hf eager_attention_forward :191-214 / SDPA as selected by LlamaAttention.forward :243-281:
//   P = softmax_fp32(Q K^T * dh^-1/2 + mask) ; O = bf16(P) V, bidirectional or causal, keys >= key_len[b]
// masked (the reference's additive [B,1,S,S] mask is never materialised: right padding => a length).
// qkv is [T,3d] (q | k | v, head h at column h*64), q/k UN-rotated: RoPE is applied on the operand loads.
//
// Formulation (one wave per 32-row tile, dh = 64): scores are computed TRANSPOSED,
//   S^T[key][query] = K_tile Q_tile^T, so a lane owns ONE query column: the softmax statistics
// (m, l, lse, delta) are lane-local scalars, and the bf16 P^T accumulator registers are directly the
// B operand of the next MFMA  O^T[dh][query] += V^T[dh][key] P^T[key][query].  The only transposed
// operand (V^T, K^T, Q^T, dO^T: contraction index strided in memory) is read from a 4 KB LDS tile with
// gfx950's ds_read_b64_tr_b16.  Backward = delta kernel + dQ kernel (per query tile) + dK/dV kernel
// (per key tile); no atomics, P recomputed from the saved log-sum-exp.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr float kScale = 0.125f;  // head_dim^-0.5, head_dim = 64
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kScaleL2 = kScale * kLog2e;   // scores are carried in the log2 domain: p = exp2(s * kScaleL2 - m2)

// [32 rows][128 B] tile: the 16-byte chunk index is XORed with a 3-bit function of the row so that both access patterns
// are bank-conflict free: ds_read_b128 of one chunk over 16 consecutive rows (rows of equal parity need 8 distinct chunk
// slots: f is a bijection of (row>>1)&7) and ds_read_b64_tr_b16 over 4 consecutive rows x 64 B (rows r and r+2 must fall
// into different 64-byte halves: bit 2 of f = bit 0 of row>>1).  PMC before: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// 0.32-0.45 with the old single-bit swizzle.
__device__ __forceinline__ int swz(int row, int byte_in_row) {
  const int x = (row >> 1) & 7;
  const int f = ((x & 1) << 2) | (x >> 1);
  return row * 128 + ((((byte_in_row >> 4) ^ f) << 4) | (byte_in_row & 15));
}

// RoPE fused into the operand loads (hf apply_rotary_pos_emb :138-160, half-split pairing j <-> j+32): q and k stay
// un-rotated in HBM; they are rotated in registers on the way into the MFMAs and dq/dk are rotated back on the way out.
struct Rope {
  const float* cos_tab;  // [max_pos][32] fp32, nullptr => no rotation (plain attention op)
  const float* sin_tab;
  const int64_t* pos;    // [B,S] or nullptr => position = index in the sequence
  int S;
};
__device__ __forceinline__ int rope_pos(const Rope& R, int b, int s) {
  return R.pos ? (int)R.pos[(size_t)b * R.S + s] : s;
}
// lo holds dh [j0, j0+8), up holds dh [j0+32, j0+40); rotate both in place by the angle table row `pos`
__device__ __forceinline__ void rope_pair(uint4& lo, uint4& up, const Rope& R, int pos, int j0) {
  float a[8], b[8], c[8], sn[8];
  unpack8(lo, a);
  unpack8(up, b);
  const float4* ct = reinterpret_cast<const float4*>(R.cos_tab + (size_t)pos * 32 + j0);
  const float4* st = reinterpret_cast<const float4*>(R.sin_tab + (size_t)pos * 32 + j0);
  const float4 c0 = ct[0], c1 = ct[1], s0 = st[0], s1 = st[1];
  c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
  sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
  float oa[8], ob[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    oa[e] = a[e] * c[e] - b[e] * sn[e];
    ob[e] = b[e] * c[e] + a[e] * sn[e];
  }
  lo = pack8(oa);
  up = pack8(ob);
}

// [32 rows][64 dh] bf16 tile -> LDS (rows >= row_lim are zero-filled so masked probabilities never meet NaNs)
__device__ __forceinline__ void load_tile(unsigned char* lds, const bf16_t* __restrict__ base, int r0, int row_lim,
                                          size_t pitch, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    const int row = c >> 3, ch = c & 7;
    const int gr = r0 + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < row_lim) v = *reinterpret_cast<const uint4*>(base + (size_t)gr * pitch + ch * 8);
    *reinterpret_cast<uint4*>(lds + swz(row, ch * 16)) = v;
  }
}
// same, rotating every row by its position (each lane fetches its chunk and the chunk 32 channels away).
// Cooperative: the NT threads of the block split the tile's 256 16-byte chunks.
template <int NT>
__device__ __forceinline__ void load_tile_coop(unsigned char* lds, const bf16_t* __restrict__ base, int r0, int row_lim,
                                               size_t pitch, int tid, const Rope& R, int b) {
#pragma unroll
  for (int i = 0; i < 256 / NT; ++i) {
    const int c = tid + i * NT;
    const int row = c >> 3, ch = c & 7;
    const int gr = r0 + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < row_lim) {
      const bf16_t* rp = base + (size_t)gr * pitch;
      if (R.cos_tab) {
        uint4 lo = *reinterpret_cast<const uint4*>(rp + (ch & 3) * 8);
        uint4 up = *reinterpret_cast<const uint4*>(rp + (ch & 3) * 8 + 32);
        rope_pair(lo, up, R, rope_pos(R, b, gr), (ch & 3) * 8);
        v = (ch & 4) ? up : lo;
      } else {
        v = *reinterpret_cast<const uint4*>(rp + ch * 8);
      }
    }
    *reinterpret_cast<uint4*>(lds + swz(row, ch * 16)) = v;
  }
}
// Register-prefetched variant of load_tile_coop: `tile_fetch` only issues the global loads of a tile (they stay in flight
// while the block computes on the tile already in LDS), `tile_commit` rotates (if asked) and writes them to LDS later.
template <int NT>
struct TilePref {
  uint4 lo[256 / NT];
  uint4 up[256 / NT];   // the chunk 32 channels away (RoPE partner); only loaded when a rotation is applied
};
template <int NT>
__device__ __forceinline__ void tile_fetch(TilePref<NT>& p, const bf16_t* __restrict__ base, int r0, int row_lim, size_t pitch,
                                           int tid, const Rope& R) {
#pragma unroll
  for (int i = 0; i < 256 / NT; ++i) {
    const int c = tid + i * NT;
    const int row = c >> 3, ch = c & 7;
    const int gr = r0 + row;
    p.lo[i] = make_uint4(0, 0, 0, 0);
    p.up[i] = make_uint4(0, 0, 0, 0);
    if (gr < row_lim) {
      const bf16_t* rp = base + (size_t)gr * pitch;
      if (R.cos_tab) {
        p.lo[i] = *reinterpret_cast<const uint4*>(rp + (ch & 3) * 8);
        p.up[i] = *reinterpret_cast<const uint4*>(rp + (ch & 3) * 8 + 32);
      } else {
        p.lo[i] = *reinterpret_cast<const uint4*>(rp + ch * 8);
      }
    }
  }
}
template <int NT>
__device__ __forceinline__ void tile_commit(unsigned char* lds, TilePref<NT>& p, int r0, int row_lim, int tid, const Rope& R,
                                            int b) {
#pragma unroll
  for (int i = 0; i < 256 / NT; ++i) {
    const int c = tid + i * NT;
    const int row = c >> 3, ch = c & 7;
    const int gr = r0 + row;
    uint4 v = p.lo[i];
    if (R.cos_tab && gr < row_lim) {
      uint4 lo = p.lo[i], up = p.up[i];
      rope_pair(lo, up, R, rope_pos(R, b, gr), (ch & 3) * 8);
      v = (ch & 4) ? up : lo;
    }
    *reinterpret_cast<uint4*>(lds + swz(row, ch * 16)) = v;
  }
}
__device__ __forceinline__ void load_tile_rope(unsigned char* lds, const bf16_t* __restrict__ base, int r0, int row_lim,
                                               size_t pitch, int lane, const Rope& R, int b) {
  if (!R.cos_tab) { load_tile(lds, base, r0, row_lim, pitch, lane); return; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    const int row = c >> 3, ch = c & 7;
    const int gr = r0 + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < row_lim) {
      const bf16_t* rp = base + (size_t)gr * pitch;
      uint4 lo = *reinterpret_cast<const uint4*>(rp + (ch & 3) * 8);
      uint4 up = *reinterpret_cast<const uint4*>(rp + (ch & 3) * 8 + 32);
      rope_pair(lo, up, R, rope_pos(R, b, gr), (ch & 3) * 8);
      v = (ch & 4) ? up : lo;
    }
    *reinterpret_cast<uint4*>(lds + swz(row, ch * 16)) = v;
  }
}

// MFMA operand whose "row/col" index is the tile row (lane&31) and whose k-chunk is dh [16s + 8*hi, +8)
__device__ __forceinline__ bf16x8_t frag_rows(const unsigned char* lds, int s, int lane) {
  const uint4 v = *reinterpret_cast<const uint4*>(lds + swz(lane & 31, (2 * s + (lane >> 5)) * 16));
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t frag_global(const bf16_t* __restrict__ base, int row, int row_lim, size_t pitch, int s,
                                                int lane) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < row_lim) v = *reinterpret_cast<const uint4*>(base + (size_t)row * pitch + 16 * s + (lane >> 5) * 8);
  return __builtin_bit_cast(bf16x8_t, v);
}
// Attention dropout (hf eager_attention_forward :210: dropout on the softmax output, training only; every reference
// launch script sets attention_dropout=0.1).  Counter-based: the keep decision of element (batch*head, query, key) is a
// hash of (seed, coordinates), so forward and the two backward kernels regenerate the same mask without storing it.
struct Drop {
  unsigned thresh;   // drop when hash24 < thresh  (thresh = p * 2^24); 0 => no dropout
  float inv_keep;    // 1 / (1 - p)
  unsigned seed;
};
// hash(seed, bh, q, k) = mix((seed ^ bh*C0) + q*C1 + k*C2), mix = one multiply between two xor-shifts: the softmax loop is
// VALU-bound (head_dim 64) and integer multiplies are quarter rate, so the per-element cost is kept to 2 adds + 1 multiply:
// callers pass the partial sum of everything that is fixed for the lane (drop_base) and add the moving coordinate's term.
__device__ __forceinline__ unsigned drop_base(const Drop& D, unsigned bh, unsigned q_or_0, unsigned k_or_0) {
  return (D.seed ^ (bh * 0x9E3779B1u)) + q_or_0 * 0x85EBCA77u + k_or_0 * 0xC2B2AE3Du;
}
__device__ __forceinline__ float drop_mul_x(const Drop& D, unsigned x) {
  if (D.thresh == 0) return 1.f;
  x ^= x >> 16; x *= 0x045D9F3Bu; x ^= x >> 16;
  return (x >> 8) < D.thresh ? 0.f : D.inv_keep;
}
__device__ __forceinline__ float drop_mul(const Drop& D, unsigned bh, unsigned q, unsigned k) {
  return drop_mul_x(D, drop_base(D, bh, q, k));
}

// Which keys a query may attend: right padding gives one length per batch row (key_len[b], keys [0, len)); packed rows
// (several graphs back to back, block-diagonal [B,S,S] mask of reference src/utils/tokenizer_utils.py:349-355 /
// modeling_helpers.py:51-64) give every token the inclusive key range [lo, hi] of its own graph.
struct KeyRange {
  const int32_t* key_len;   // [B] or nullptr (=> S); ignored when lo/hi are given
  const int32_t* lo;        // [B,S] or nullptr
  const int32_t* hi;        // [B,S]
};
// wave-uniform min / max of small non-negative integers (exact in fp32)
__device__ __forceinline__ int wave_imin(int v) { return (int)-wave_max(-(float)v); }
__device__ __forceinline__ int wave_imax(int v) { return (int)wave_max((float)v); }

// the 4 dh-fragments of one row (dh chunk [16s + 8hi, +8), s = 0..3), rotated: chunks s and s+2 are 32 channels apart
__device__ __forceinline__ void frags_global_rope(bf16x8_t (&f)[4], const bf16_t* __restrict__ base, int row, int row_lim,
                                                  size_t pitch, int lane, const Rope& R, int b) {
  uint4 v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    v[s] = make_uint4(0, 0, 0, 0);
    if (row < row_lim) v[s] = *reinterpret_cast<const uint4*>(base + (size_t)row * pitch + 16 * s + (lane >> 5) * 8);
  }
  if (R.cos_tab && row < row_lim) {
    const int pos = rope_pos(R, b, row);
    rope_pair(v[0], v[2], R, pos, (lane >> 5) * 8);
    rope_pair(v[1], v[3], R, pos, 16 + (lane >> 5) * 8);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) f[s] = __builtin_bit_cast(bf16x8_t, v[s]);
}
// transposed operand: MFMA row = dh (dhb*32 + lane&31), k = tile rows {16j + 4hi + (e&3) + 8(e>>2)}
__device__ __forceinline__ bf16x8_t frag_tr(const unsigned char* lds, int dhb, int j, int lane) {
  const int li = lane & 15, g = lane >> 4, hi = lane >> 5;
  const int colb = (dhb * 32 + (g & 1) * 16 + (li & 3) * 4) * 2;
  const int ra = 16 * j + 4 * hi + (li >> 2);
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(lds + swz(ra, colb)));
  const bf16x4_t up = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(lds + swz(ra + 8, colb)));
  bf16x8_t o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
  return o;
}
// accumulator registers [8j, 8j+8) -> bf16 B operand (k = tile rows in the same order as frag_tr)
__device__ __forceinline__ bf16x8_t acc_to_b(const f32x16_t& a, int j) {
  uint4 v;
  v.x = pack2bf(a[8 * j + 0], a[8 * j + 1]);
  v.y = pack2bf(a[8 * j + 2], a[8 * j + 3]);
  v.z = pack2bf(a[8 * j + 4], a[8 * j + 5]);
  v.w = pack2bf(a[8 * j + 6], a[8 * j + 7]);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16_t zero16() {
  f32x16_t z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// store a transposed accumulator pair (O^T[dh][row]) as row-major bf16 [row][64 dh]; lane owns row (lane&31)
__device__ __forceinline__ void store_t(bf16_t* __restrict__ dst_row, const f32x16_t& a0, const f32x16_t& a1, float mul,
                                        int hi) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    uint2 o;
    o.x = pack2bf(a0[4 * rr + 0] * mul, a0[4 * rr + 1] * mul);
    o.y = pack2bf(a0[4 * rr + 2] * mul, a0[4 * rr + 3] * mul);
    *reinterpret_cast<uint2*>(dst_row + 8 * rr + 4 * hi) = o;
    o.x = pack2bf(a1[4 * rr + 0] * mul, a1[4 * rr + 1] * mul);
    o.y = pack2bf(a1[4 * rr + 2] * mul, a1[4 * rr + 3] * mul);
    *reinterpret_cast<uint2*>(dst_row + 32 + 8 * rr + 4 * hi) = o;
  }
}

// gradient of a rotated row back to the un-rotated one: x = R(-theta) x'   (a0 = dh 0..31, a1 = dh 32..63)
__device__ __forceinline__ void unrope_acc(f32x16_t& a0, f32x16_t& a1, const Rope& R, int pos, int hi) {
  if (!R.cos_tab) return;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int j0 = 8 * rr + 4 * hi;
    const float4 c = *reinterpret_cast<const float4*>(R.cos_tab + (size_t)pos * 32 + j0);
    const float4 sn = *reinterpret_cast<const float4*>(R.sin_tab + (size_t)pos * 32 + j0);
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = a0[4 * rr + e], up = a1[4 * rr + e];
      a0[4 * rr + e] = lo * cc[e] + up * ss[e];
      a1[4 * rr + e] = up * cc[e] - lo * ss[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Block = NW waves = NW consecutive 32-query tiles of one (batch, head); every 32-key K tile and V tile is brought into
// LDS ONCE per block (K rotated on the way in when R is given) and shared by the NW waves: K as the A operand of
// S^T = K Q^T (ds_read_b128), V through the transposing read for O^T += V^T P^T.
template <int NW, bool PK>
__global__ void __launch_bounds__(NW * 64, NW == 1 ? 3 : 2) attn_fwd_kernel(const bf16_t* __restrict__ qkv, KeyRange KR,
                                                           bf16_t* __restrict__ out, float* __restrict__ lse, int B, int S,
                                                           int H, int causal, Rope R, Drop D) {
  __shared__ __attribute__((aligned(16))) unsigned char kt[4096];
  __shared__ __attribute__((aligned(16))) unsigned char vt[4096];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const int d = H * 64;
  const size_t pitch = (size_t)3 * d;
  const bf16_t* qb = qkv + (size_t)b * S * pitch + h * 64;
  const bf16_t* kb = qb + d;
  const bf16_t* vb = qb + 2 * d;
  const int qrow = q0 + l31;
  const Rope Rnone{nullptr, nullptr, nullptr, S};
  // per-lane inclusive key range [qlo, qhi]; wave-uniform union [ulo, uhi] (tiles outside are skipped) and intersection
  // [ilo, ihi] (tiles inside need no mask)
  constexpr bool packed = PK;   // per-token key ranges (KR.lo / KR.hi) instead of one key length per batch row
  int qlo = 0, qhi = (KR.key_len ? KR.key_len[b] : S) - 1;
  int ulo = 0, uhi = qhi, ilo = 0, ihi = qhi;
  if (packed) {
    const bool v = qrow < S;
    qlo = v ? KR.lo[(size_t)b * S + qrow] : 0;
    qhi = v ? KR.hi[(size_t)b * S + qrow] : -1;
    ulo = wave_imin(qhi >= qlo ? qlo : S); uhi = wave_imax(qhi >= qlo ? qhi + 1 : 0) - 1;
    ilo = wave_imax(v ? qlo : 0); ihi = wave_imin(v ? qhi + 1 : S) - 1;
  }
  const int klen = packed ? S : qhi + 1;     // block-level upper bound of the key loop

  bf16x8_t qf[4];
  frags_global_rope(qf, qb, qrow, S, pitch, lane, R, b);
  f32x16_t o0 = zero16(), o1 = zero16();
  float m = -INFINITY, l = 0.f;
  const unsigned dbase = drop_base(D, b * H + h, qrow, 0);
  const int q_end_blk = min(S, (int)(blockIdx.x + 1) * NW * 32);       // one past the block's last query row
  const int kend_blk = causal ? min(klen, q_end_blk) : klen;
  const int kend = (q0 < S) ? min(uhi + 1, causal ? q0 + 32 : S) : 0;  // this wave's own key range
  constexpr bool PF = NW > 1;   // single-wave blocks see one tile (S <= 32): nothing to prefetch, registers are tight
  TilePref<NW * 64> pk, pv;
  if (PF && kend_blk > 0) {
    tile_fetch<NW * 64>(pk, kb, 0, S, pitch, tid, R);
    tile_fetch<NW * 64>(pv, vb, 0, S, pitch, tid, Rnone);
  }
  for (int k0 = 0; k0 < kend_blk; k0 += 32) {
    __syncthreads();  // previous tile fully consumed
    if constexpr (PF) {
      tile_commit<NW * 64>(kt, pk, k0, S, tid, R, b);
      tile_commit<NW * 64>(vt, pv, k0, S, tid, Rnone, b);
    } else {
      load_tile_coop<NW * 64>(kt, kb, k0, S, pitch, tid, R, b);
      load_tile_coop<NW * 64>(vt, vb, k0, S, pitch, tid, Rnone, b);
    }
    __syncthreads();
    if (PF && k0 + 32 < kend_blk) {   // next tile's loads fly while this one is computed
      tile_fetch<NW * 64>(pk, kb, k0 + 32, S, pitch, tid, R);
      tile_fetch<NW * 64>(pv, vb, k0 + 32, S, pitch, tid, Rnone);
    }
    if (k0 >= kend || k0 + 31 < ulo) continue;
    f32x16_t sc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, s, lane), qf[s], sc, 0, 0, 0);
    // The loop is VALU-bound (head_dim 64: ~0.25 MFMA cycles but several VALU cycles per score), so: scores stay raw
    // until one fma + exp2 (scale and log2(e) folded, running max kept in the log2 domain); the key / causal mask is only
    // evaluated on tiles that touch the sequence end or the diagonal; O is rescaled only when some lane's max moved.
    const bool edge = (k0 < ilo) || (k0 + 31 > ihi) || (causal && k0 + 31 > q0) || (q0 + 32 > S);
    float mx = -INFINITY;
    if (edge) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + acc_row(r, hi);
        const bool ok = key >= qlo && key <= qhi && (!causal || key <= qrow);
        sc[r] = ok ? sc[r] : -INFINITY;
        mx = fmaxf(mx, sc[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx * kScaleL2);     // kScaleL2 > 0: max commutes with the scaling
    const bool dead = m_new == -INFINITY;
    const float alpha = dead ? 1.f : exp2f(m - m_new);
    const float nm = dead ? 0.f : -m_new;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = dead ? 0.f : exp2f(fmaf(sc[r], kScaleL2, nm));
      rs += p;                                                                    // softmax normaliser: before dropout
      sc[r] = p * drop_mul_x(D, dbase + (unsigned)(k0 + acc_row(r, hi)) * 0xC2B2AE3Du);   // what multiplies V
    }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = m_new;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    const bf16x8_t pb0 = acc_to_b(sc, 0), pb1 = acc_to_b(sc, 1);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt, 0, 0, lane), pb0, o0, 0, 0, 0);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt, 0, 1, lane), pb1, o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt, 1, 0, lane), pb0, o1, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt, 1, 1, lane), pb1, o1, 0, 0, 0);
  }
  if (qrow < S) {
    const float inv = l > 0.f ? 1.f / l : 0.f;   // (the keep scale 1/(1-p) is already folded into the dropped P)
    store_t(out + ((size_t)b * S + qrow) * d + h * 64, o0, o1, inv, hi);
    if (hi == 0 && lse) lse[((size_t)b * H + h) * S + qrow] = l > 0.f ? (m + log2f(l)) * (1.0f / kLog2e) : 0.f;   // natural log
  }
}

// dQ^T[dh][q] = sum_keys K^T[dh][key] dS^T[key][q],  dS^T = P^T (dP^T - delta_q) * scale
// Block = NW query tiles; the K tile (rotated if Rin) and the V tile are shared through LDS.
template <int NW, bool PK>
__global__ void __launch_bounds__(NW * 64, NW == 1 ? 3 : 2) attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse, float* __restrict__ delta,
                                                              KeyRange KR, bf16_t* __restrict__ dqkv,
                                                              int B, int S, int H, int causal, Rope Rin, Rope R, Drop D) {
  __shared__ __attribute__((aligned(16))) unsigned char kt[4096];
  __shared__ __attribute__((aligned(16))) unsigned char vt[4096];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const int d = H * 64;
  const size_t pitch = (size_t)3 * d;
  const bf16_t* qb = qkv + (size_t)b * S * pitch + h * 64;
  const bf16_t* kb = qb + d;
  const bf16_t* vb = qb + 2 * d;
  const bf16_t* dob = dout + (size_t)b * S * d + h * 64;
  const int qrow = q0 + l31;
  const Rope Rnone{nullptr, nullptr, nullptr, S};
  constexpr bool packed = PK;      // see attn_fwd_kernel
  int qlo = 0, qhi = (KR.key_len ? KR.key_len[b] : S) - 1;
  int ulo = 0, uhi = qhi, ilo = 0, ihi = qhi;
  if (packed) {
    const bool v = qrow < S;
    qlo = v ? KR.lo[(size_t)b * S + qrow] : 0;
    qhi = v ? KR.hi[(size_t)b * S + qrow] : -1;
    ulo = wave_imin(qhi >= qlo ? qlo : S); uhi = wave_imax(qhi >= qlo ? qhi + 1 : 0) - 1;
    ilo = wave_imax(v ? qlo : 0); ihi = wave_imin(v ? qhi + 1 : S) - 1;
  }
  const int klen = packed ? S : qhi + 1;
  bf16x8_t qf[4], dof[4];
  frags_global_rope(qf, qb, qrow, S, pitch, lane, Rin, b);
#pragma unroll
  for (int s = 0; s < 4; ++s) dof[s] = frag_global(dob, qrow, S, (size_t)d, s, lane);
  const size_t sidx = ((size_t)b * H + h) * S + min(qrow, S - 1);
  // delta_q = sum_dh dO*O of this query row (the lane holds half of the row's dO already; the other half sits in lane^32);
  // stored for the dK/dV kernel that follows on the same stream
  float dl = 0.f;
  {
    const bf16_t* ob = out + (size_t)b * S * d + h * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float a[8], gg[8];
      unpack8(__builtin_bit_cast(uint4, frag_global(ob, qrow, S, (size_t)d, s, lane)), a);
      unpack8(__builtin_bit_cast(uint4, dof[s]), gg);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += a[e] * gg[e];
    }
    dl += __shfl_xor(dl, 32, 64);
    if (hi == 0 && qrow < S) delta[sidx] = dl;
  }
  const float nlse2 = -lse[sidx] * kLog2e, ndl = -dl;
  const unsigned dbase = drop_base(D, b * H + h, qrow, 0);
  f32x16_t a0 = zero16(), a1 = zero16();
  const int q_end_blk = min(S, (int)(blockIdx.x + 1) * NW * 32);
  const int kend_blk = causal ? min(klen, q_end_blk) : klen;
  const int kend = (q0 < S) ? min(uhi + 1, causal ? q0 + 32 : S) : 0;
  constexpr bool PF = NW > 1;
  TilePref<NW * 64> pk, pv;
  if (PF && kend_blk > 0) {
    tile_fetch<NW * 64>(pk, kb, 0, S, pitch, tid, Rin);
    tile_fetch<NW * 64>(pv, vb, 0, S, pitch, tid, Rnone);
  }
  for (int k0 = 0; k0 < kend_blk; k0 += 32) {
    __syncthreads();
    if constexpr (PF) {
      tile_commit<NW * 64>(kt, pk, k0, S, tid, Rin, b);
      tile_commit<NW * 64>(vt, pv, k0, S, tid, Rnone, b);
    } else {
      load_tile_coop<NW * 64>(kt, kb, k0, S, pitch, tid, Rin, b);
      load_tile_coop<NW * 64>(vt, vb, k0, S, pitch, tid, Rnone, b);
    }
    __syncthreads();
    if (PF && k0 + 32 < kend_blk) {
      tile_fetch<NW * 64>(pk, kb, k0 + 32, S, pitch, tid, Rin);
      tile_fetch<NW * 64>(pv, vb, k0 + 32, S, pitch, tid, Rnone);
    }
    if (k0 >= kend || k0 + 31 < ulo) continue;
    f32x16_t dp = zero16(), sc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(vt, s, lane), dof[s], dp, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, s, lane), qf[s], sc, 0, 0, 0);
    }
    const bool edge = (k0 < ilo) || (k0 + 31 > ihi) || (causal && k0 + 31 > q0) || (q0 + 32 > S);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row(r, hi);
      float p = exp2f(fmaf(sc[r], kScaleL2, nlse2));
      if (edge) p = (key >= qlo && key <= qhi && (!causal || key <= qrow) && qrow < S) ? p : 0.f;
      sc[r] = p * fmaf(dp[r], drop_mul_x(D, dbase + (unsigned)key * 0xC2B2AE3Du), ndl) * kScale;
    }
    const bf16x8_t ds0 = acc_to_b(sc, 0), ds1 = acc_to_b(sc, 1);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 0, 0, lane), ds0, a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 0, 1, lane), ds1, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 1, 0, lane), ds0, a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 1, 1, lane), ds1, a1, 0, 0, 0);
  }
  if (qrow < S) {
    unrope_acc(a0, a1, R, rope_pos(R, b, qrow), hi);
    store_t(dqkv + ((size_t)b * S + qrow) * pitch + h * 64, a0, a1, 1.f, hi);
  }
}

// dV^T[dh][key] = sum_q dO^T[dh][q] P[q][key] ; dK^T[dh][key] = sum_q Q^T[dh][q] dS[q][key]
// Block = NW key tiles; every 32-query Q tile (rotated if Rin), dO tile and their lse/delta are shared through LDS.
template <int NW, bool PK>
__global__ void __launch_bounds__(NW * 64, NW == 1 ? 3 : 2) attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               KeyRange KR, bf16_t* __restrict__ dqkv,
                                                               int B, int S, int H, int causal, Rope Rin, Rope R, Drop D) {
  __shared__ __attribute__((aligned(16))) unsigned char qt[4096];
  __shared__ __attribute__((aligned(16))) unsigned char dot_[4096];
  __shared__ float lse_s[32], dl_s[32];
  __shared__ int qlo_s[32], qhi_s[32];   // packed rows: inclusive key range of each query of the tile
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int k0 = (blockIdx.x * NW + wave) * 32;
  const int d = H * 64;
  const size_t pitch = (size_t)3 * d;
  const bf16_t* qb = qkv + (size_t)b * S * pitch + h * 64;
  const bf16_t* kb = qb + d;
  const bf16_t* vb = qb + 2 * d;
  const bf16_t* dob = dout + (size_t)b * S * d + h * 64;
  constexpr bool packed = PK;
  const int klen = packed ? S : (KR.key_len ? KR.key_len[b] : S);
  const int krow = k0 + l31;
  const Rope Rnone{nullptr, nullptr, nullptr, S};
  bf16x8_t kf[4], vf[4];
  frags_global_rope(kf, kb, krow, S, pitch, lane, Rin, b);
#pragma unroll
  for (int s = 0; s < 4; ++s) vf[s] = frag_global(vb, krow, S, pitch, s, lane);
  f32x16_t dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const unsigned dbase = drop_base(D, b * H + h, 0, krow);
  const bool key_ok = krow < klen;                 // right-padded rows: one key length per batch row
  const int kblk0 = blockIdx.x * NW * 32;          // first key of the block
  const int qstart = causal ? kblk0 : 0;           // queries before the block's first key never see it
  if (kblk0 < klen) {
    TilePref<NW * 64> pq, pdo;
    float p_lse = 0.f, p_dl = 0.f;
    int p_lo = 0, p_hi = klen - 1;
    auto fetch_q = [&](int q0) {
      tile_fetch<NW * 64>(pq, qb, q0, S, pitch, tid, Rin);
      tile_fetch<NW * 64>(pdo, dob, q0, S, (size_t)d, tid, Rnone);
      if (tid < 32) {
        const int q = min(q0 + tid, S - 1);
        p_lse = lse[((size_t)b * H + h) * S + q];
        p_dl = delta[((size_t)b * H + h) * S + q];
        if (packed) {
          const bool v = q0 + tid < S;
          p_lo = v ? KR.lo[(size_t)b * S + q] : 0;
          p_hi = v ? KR.hi[(size_t)b * S + q] : -1;
        }
      }
    };
    constexpr bool PF = NW > 1;
    if (PF && qstart < S) fetch_q(qstart);
    for (int q0 = qstart; q0 < S; q0 += 32) {
      __syncthreads();
      if constexpr (PF) {
        tile_commit<NW * 64>(qt, pq, q0, S, tid, Rin, b);
        tile_commit<NW * 64>(dot_, pdo, q0, S, tid, Rnone, b);
        if (tid < 32) {
          lse_s[tid] = -p_lse * kLog2e; dl_s[tid] = -p_dl;
          if (packed) { qlo_s[tid] = p_lo; qhi_s[tid] = p_hi; }
        }
      } else {
        load_tile_coop<NW * 64>(qt, qb, q0, S, pitch, tid, Rin, b);
        load_tile_coop<NW * 64>(dot_, dob, q0, S, (size_t)d, tid, Rnone, b);
        if (tid < 32) {
          const int q = min(q0 + tid, S - 1);
          lse_s[tid] = -lse[((size_t)b * H + h) * S + q] * kLog2e;
          dl_s[tid] = -delta[((size_t)b * H + h) * S + q];
          if (packed) {
            const bool v = q0 + tid < S;
            qlo_s[tid] = v ? KR.lo[(size_t)b * S + q] : 0;
            qhi_s[tid] = v ? KR.hi[(size_t)b * S + q] : -1;
          }
        }
      }
      __syncthreads();
      if (PF && q0 + 32 < S) fetch_q(q0 + 32);
      if (k0 >= klen || (causal && q0 + 31 < k0)) continue;   // this wave's keys are padding / all in the future
      // packed rows: union / intersection of the tile's query ranges decide skipping and masking for this wave's 32 keys
      bool edge = (k0 + 32 > klen) || (q0 + 32 > S) || (causal && k0 + 31 > q0);
      if (packed) {
        const int lo = qlo_s[l31], hi_ = qhi_s[l31];
        const int ulo = wave_imin(hi_ >= lo ? lo : S), uhi = wave_imax(hi_ >= lo ? hi_ + 1 : 0) - 1;
        if (uhi < k0 || ulo > k0 + 31) continue;
        const int ilo = wave_imax(lo), ihi = wave_imin(hi_ + 1) - 1;
        edge = edge || ilo > k0 || ihi < k0 + 31;
      }
      f32x16_t sc = zero16(), dp = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(qt, s, lane), kf[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dot_, s, lane), vf[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = acc_row(r, hi);
        const int q = q0 + qi;
        float p = exp2f(fmaf(sc[r], kScaleL2, lse_s[qi]));          // lse_s holds -lse * log2(e)
        if (edge) {
          const bool in_range = packed ? (krow >= qlo_s[qi] && krow <= qhi_s[qi]) : key_ok;
          p = (in_range && q < S && (!causal || krow <= q)) ? p : 0.f;
        }
        const float dm = drop_mul_x(D, dbase + (unsigned)q * 0x85EBCA77u);
        sc[r] = p * dm;                                  // dropped probabilities: what multiplied V in forward
        dp[r] = p * fmaf(dp[r], dm, dl_s[qi]) * kScale;  // dl_s holds -delta
      }
      const bf16x8_t p0 = acc_to_b(sc, 0), p1 = acc_to_b(sc, 1);
      const bf16x8_t s0 = acc_to_b(dp, 0), s1 = acc_to_b(dp, 1);
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 0, 0, lane), p0, dv0, 0, 0, 0);
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 0, 1, lane), p1, dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 1, 0, lane), p0, dv1, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 1, 1, lane), p1, dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 0, 0, lane), s0, dk0, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 0, 1, lane), s1, dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 1, 0, lane), s0, dk1, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 1, 1, lane), s1, dk1, 0, 0, 0);
    }
  }
  if (krow < S) {
    bf16_t* row = dqkv + ((size_t)b * S + krow) * pitch + h * 64;
    unrope_acc(dk0, dk1, R, rope_pos(R, b, krow), hi);
    store_t(row + d, dk0, dk1, 1.f, hi);
    store_t(row + 2 * d, dv0, dv1, 1.f, hi);
  }
}

// S <= 32: one wave holds the whole (batch, head) problem, so the backward is ONE launch: the K, Q and dO tiles go to
// LDS once (V stays in registers), then the dQ part (scores transposed, lane = query; it also yields the softmax-backward row
// term delta) and the dK/dV part (lane = key) run back to back on the same tiles.  Same arithmetic as attn_bwd_dq_kernel +
// attn_bwd_dkv_kernel except that delta comes from P and dP in registers instead of rowsum(dO * O).
__global__ void __launch_bounds__(64, 3) attn_bwd_small_kernel(const bf16_t* __restrict__ qkv,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                               const int32_t* __restrict__ key_len, bf16_t* __restrict__ dqkv,
                                                               int B, int S, int H, int causal, Rope Rin, Rope R, Drop D) {
  __shared__ __attribute__((aligned(16))) unsigned char kt[4096];
  __shared__ __attribute__((aligned(16))) unsigned char qt[4096];
  __shared__ __attribute__((aligned(16))) unsigned char dot_[4096];
  __shared__ float lse_s[32], dl_s[32];
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int d = H * 64;
  const size_t pitch = (size_t)3 * d;
  const bf16_t* qb = qkv + (size_t)b * S * pitch + h * 64;
  const bf16_t* kb = qb + d;
  const bf16_t* vb = qb + 2 * d;
  const bf16_t* dob = dout + (size_t)b * S * d + h * 64;
  const int klen = key_len ? key_len[b] : S;
  const Rope Rnone{nullptr, nullptr, nullptr, S};
  load_tile_coop<64>(kt, kb, 0, S, pitch, lane, Rin, b);
  // V is only ever read row-wise (operand rows = keys): its fragments come straight from global memory, which keeps the
  // block at 12 KiB of LDS = 12 single-wave blocks per CU, i.e. B*H = 3072 problems of the headline shape in ONE round
  bf16x8_t vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) vf[s] = frag_global(vb, l31, S, pitch, s, lane);
  load_tile_coop<64>(qt, qb, 0, S, pitch, lane, Rin, b);
  load_tile_coop<64>(dot_, dob, 0, S, (size_t)d, lane, Rnone, b);
  const float nlse2 = -lse[((size_t)b * H + h) * S + min(l31, S - 1)] * kLog2e;
  if (hi == 0) lse_s[l31] = nlse2;
  __syncthreads();
  const unsigned bh = b * H + h;
  {   // ---------------- dQ^T[dh][q] = K^T dS^T   (lane owns query l31)
    const int qrow = l31;
    const unsigned dbase = drop_base(D, bh, qrow, 0);
    f32x16_t dp = zero16(), sc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s], frag_rows(dot_, s, lane), dp, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kt, s, lane), frag_rows(qt, s, lane), sc, 0, 0, 0);
    }
    // the whole key range of a query sits in this lane and its partner (lane ^ 32), so the softmax-backward row term
    // delta = sum_k P~[q,k] dP[q,k] (P~ = dropped P; equal to rowsum(dO * O), which the long-sequence kernels read from the
    // saved output) comes straight from the registers: no read of O
    float dl = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, hi);
      const bool ok = key < klen && (!causal || key <= qrow) && qrow < S;
      const float p = ok ? exp2f(fmaf(sc[r], kScaleL2, nlse2)) : 0.f;
      const float t = dp[r] * drop_mul_x(D, dbase + (unsigned)key * 0xC2B2AE3Du);
      dl = fmaf(p, t, dl);
      sc[r] = p;
      dp[r] = t;
    }
    dl += __shfl_xor(dl, 32, 64);
    if (hi == 0) dl_s[l31] = -dl;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = sc[r] * (dp[r] - dl) * kScale;
    const bf16x8_t ds0 = acc_to_b(sc, 0), ds1 = acc_to_b(sc, 1);
    f32x16_t a0 = zero16(), a1 = zero16();
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 0, 0, lane), ds0, a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 0, 1, lane), ds1, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 1, 0, lane), ds0, a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, 1, 1, lane), ds1, a1, 0, 0, 0);
    if (qrow < S) {
      unrope_acc(a0, a1, R, rope_pos(R, b, qrow), hi);
      store_t(dqkv + ((size_t)b * S + qrow) * pitch + h * 64, a0, a1, 1.f, hi);
    }
  }
  __syncthreads();   // dl_s (written above) is read per query below
  {   // ---------------- dV^T = dO^T P, dK^T = Q^T dS   (lane owns key l31)
    const int krow = l31;
    const bool key_ok = krow < klen;
    const unsigned dbase = drop_base(D, bh, 0, krow);
    f32x16_t sc = zero16(), dp = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(qt, s, lane), frag_rows(kt, s, lane), sc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dot_, s, lane), vf[s], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = acc_row(r, hi);
      const bool ok = key_ok && q < S && (!causal || krow <= q);
      const float p = ok ? exp2f(fmaf(sc[r], kScaleL2, lse_s[q])) : 0.f;
      const float dm = drop_mul_x(D, dbase + (unsigned)q * 0x85EBCA77u);
      sc[r] = p * dm;
      dp[r] = p * fmaf(dp[r], dm, dl_s[q]) * kScale;
    }
    const bf16x8_t p0 = acc_to_b(sc, 0), p1 = acc_to_b(sc, 1);
    const bf16x8_t s0 = acc_to_b(dp, 0), s1 = acc_to_b(dp, 1);
    f32x16_t dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 0, 0, lane), p0, dv0, 0, 0, 0);
    dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 0, 1, lane), p1, dv0, 0, 0, 0);
    dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 1, 0, lane), p0, dv1, 0, 0, 0);
    dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot_, 1, 1, lane), p1, dv1, 0, 0, 0);
    dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 0, 0, lane), s0, dk0, 0, 0, 0);
    dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 0, 1, lane), s1, dk0, 0, 0, 0);
    dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 1, 0, lane), s0, dk1, 0, 0, 0);
    dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 1, 1, lane), s1, dk1, 0, 0, 0);
    if (krow < S) {
      bf16_t* row = dqkv + ((size_t)b * S + krow) * pitch + h * 64;
      unrope_acc(dk0, dk1, R, rope_pos(R, b, krow), hi);
      store_t(row + d, dk0, dk1, 1.f, hi);
      store_t(row + 2 * d, dv0, dv1, 1.f, hi);
    }
  }
}

// waves (= 32-row tiles) per block: long sequences share each K/V (or Q/dO) tile among 4 waves through LDS
int attn_waves(int S) { return S >= 128 ? 4 : (S >= 64 ? 2 : 1); }

Drop make_drop(float p, unsigned seed) {
  Drop d;
  d.thresh = p > 0.f ? (unsigned)(p * 16777216.0f) : 0u;
  d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  d.seed = seed;
  return d;
}

}  // namespace

int k_attn_fwd(const void* qkv, const int32_t* key_len, void* out, float* lse, int B, int S, int H, int causal,
               const float* cos_tab, const float* sin_tab, const int64_t* position_ids, float dropout_p,
               unsigned dropout_seed, hipStream_t st, const int32_t* key_lo, const int32_t* key_hi) {
  const KeyRange KR{key_len, key_lo, key_hi};
  if (B == 0 || S == 0) return 0;
  const Rope R{cos_tab, sin_tab, position_ids, S};
  const Drop D = make_drop(dropout_p, dropout_seed);
  const int nw = attn_waves(S);
  dim3 grid((S + 32 * nw - 1) / (32 * nw), H, B);
#define GGET_ATTN_FWD(NW, PK) hipLaunchKernelGGL((attn_fwd_kernel<NW, PK>), grid, dim3(NW * 64), 0, st, (const bf16_t*)qkv, KR, \
                                             (bf16_t*)out, lse, B, S, H, causal, R, D)
  if (key_lo) { if (nw == 4) GGET_ATTN_FWD(4, true); else if (nw == 2) GGET_ATTN_FWD(2, true); else GGET_ATTN_FWD(1, true); }
  else { if (nw == 4) GGET_ATTN_FWD(4, false); else if (nw == 2) GGET_ATTN_FWD(2, false); else GGET_ATTN_FWD(1, false); }
#undef GGET_ATTN_FWD
  GGET_LAUNCH_CHECK();
  return 0;
}

int k_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_len, void* dqkv,
               float* delta_ws, int B, int S, int H, int causal, const float* cos_tab, const float* sin_tab,
               const int64_t* position_ids, int qk_rotated, float dropout_p, unsigned dropout_seed, hipStream_t st,
               const int32_t* key_lo, const int32_t* key_hi) {
  const KeyRange KR{key_len, key_lo, key_hi};
  if (B == 0 || S == 0) return 0;
  const Rope R{cos_tab, sin_tab, position_ids, S};     // rotation of dq, dk back to the un-rotated projections
  const Rope Rin = qk_rotated ? Rope{nullptr, nullptr, nullptr, S} : R;   // q,k in memory are already rotated?
  const Drop D = make_drop(dropout_p, dropout_seed);
  static int small = -1;
  if (small < 0) { const char* e = getenv("GGET_ATTN_SMALL"); small = e ? atoi(e) : 1; }
  if (S <= 32 && !key_lo && small) {
    hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(1, H, B), dim3(64), 0, st, (const bf16_t*)qkv,
                       (const bf16_t*)dout, lse, key_len, (bf16_t*)dqkv, B, S, H, causal, Rin, R, D);
    GGET_LAUNCH_CHECK();
    return 0;
  }
  const int nw = attn_waves(S);
  dim3 grid((S + 32 * nw - 1) / (32 * nw), H, B);
#define GGET_ATTN_BWD(NW, PK)                                                                                                \
  do {                                                                                                                     \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NW, PK>), grid, dim3(NW * 64), 0, st, (const bf16_t*)qkv, (const bf16_t*)out,     \
                       (const bf16_t*)dout, lse,   \
                       delta_ws, KR, (bf16_t*)dqkv, B, S, H, causal, Rin, R, D);                                      \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<NW, PK>), grid, dim3(NW * 64), 0, st, (const bf16_t*)qkv, (const bf16_t*)dout, lse,  \
                       delta_ws, KR, (bf16_t*)dqkv, B, S, H, causal, Rin, R, D);                                      \
  } while (0)
  if (key_lo) { if (nw == 4) GGET_ATTN_BWD(4, true); else if (nw == 2) GGET_ATTN_BWD(2, true); else GGET_ATTN_BWD(1, true); }
  else { if (nw == 4) GGET_ATTN_BWD(4, false); else if (nw == 2) GGET_ATTN_BWD(2, false); else GGET_ATTN_BWD(1, false); }
#undef GGET_ATTN_BWD
  GGET_LAUNCH_CHECK();
  return 0;
}
