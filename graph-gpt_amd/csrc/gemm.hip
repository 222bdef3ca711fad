// bf16 MFMA GEMM for gfx950 (MI355X) - the contraction behind every Linear of the Graph Eulerian
// Transformer (reference: hf LlamaAttention q/k/v/o_proj :253-280, LlamaMLP :174-176,
// n_token_proj / lm_head modeling_pretrain.py:88-93,218) and their dgrad / wgrad.
//
// One kernel template, three operand layouts:
//   NT  C[M,N] = A[M,K] B[N,K]^T     forward            (both operands K-contiguous)
//   NN  C[M,N] = A[M,K] B[K,N]       dgrad  dx = dy W   (B is N-contiguous)
//   TN  C[M,N] = A[K,M]^T B[K,N]     wgrad  dW = dy^T x (both operands M/N-contiguous)
// K-contiguous tiles live in LDS as [rows][64] with a 16-byte-chunk XOR swizzle and are read with
// ds_read_b128; M/N-contiguous tiles live as [64 k][rows] with a 32-byte-window XOR swizzle and are
// read with gfx950's transposing ds_read_b64_tr_b16, so no operand is ever transposed in HBM.
// MFMA: v_mfma_f32_16x16x32_bf16, operands swapped (D = Btile * Atile^T) so that a lane owns 4
// consecutive N of one row of C; v_permlane16_swap + a DPP row_ror:8 exchange regroup them so that a
// store instruction writes 8 rows x 128 B (whole cache lines).
//
// Two kernels:
//  * gemm_persist_kernel - static shapes with K % 64 == 0: one block per CU walks an XCD-contiguous,
//    super-tiled list of output tiles and streams ALL their K-tiles through one 3/4-slot LDS ring
//    (global_load_lds_dwordx4: uniform 64-bit K-origin + per-lane 32-bit offsets computed once per
//    tile; counted s_waitcnt vmcnt; ONE barrier per K-tile; the next K-tile's DMA pieces are issued
//    between the MFMA groups).  Tile shapes 256x256x32, 256x128x64, 128x192x64, 192x192x64 and
//    128x128x64 are chosen per launch so that tiles / CUs is integral where possible.
//  * gemm_kernel - one 256x128 / 128x128 tile per block with the same ring: device-side row counts /
//    reduction lengths (SMTP head), partial last K-tile (register path), split-K into fp32 slabs.
// The DMA is issued from inline asm: hipcc (ROCm 7.2) otherwise drains vmcnt(0) in front of the next
// ds_read and nothing overlaps.  Measurements behind these choices: profiles/r01_gemm_*.txt.
// Diagnostics (timing experiments only): GGET_GEMM_ABLATE (1 no DMA, 2 no MFMA, 4 no C store on the
// one-tile kernel; 32 = persistent kernel without epilogue), GGET_GEMM_NO_PERSIST, GGET_GEMM_NO_256,
// GGET_GEMM_192=0, GGET_GEMM_SUPER=<rows per L2 super-tile>.
#include <stdlib.h>
#include <string.h>
#include <deque>

#include "common.h"
#include "gemm.h"

#define LDS_AS __attribute__((address_space(3)))

int g_gemm_lds_headroom = 1;   // 1 (default): the 128x192 / 192x128 tile kernels run a 3-slot ring and leave 40 KiB of LDS free, 0: 4 slots = all 160 KiB,
                               // 2 (set when a communicator is attached: world > 1): also no launch with two LDS-filling blocks per CU
                               // (gget_debug_set key 2, env GGET_GEMM_LDS_HEADROOM; measured in profiles/r02_coresidency.txt)
int g_gemm_split_last = 0;   // split the K range of the last, partial round's tiles among the idle blocks (gget_debug_set key 3, env
                            // GGET_GEMM_SPLIT_LAST).  Off by default: on the C1 shapes the hand-over of the partial tiles costs more than the
                            // shorter last round returns (profiles/r03_gemm_varlen_shapes.txt); correct and tested (tests/test_gpu_ops.py)
int g_gemm_variant = 0;   // measurement knob (gget_debug_set key 1): selects experimental kernel variants for in-process A/B timing
int g_gemm_ablate_set = -1;     // >= 0: replaces GGET_GEMM_ABLATE at run time (gget_debug_set key 7; in-process A/B of the experiment bits)
int g_gemm_stagger_ticks = 0;   // MODE 2 launches (two workgroups per CU): start delay of a CU's second workgroup in 100 MHz ticks (gget_debug_set
                                // key 5, env GGET_GEMM_STAGGER); 0 = both start together (round 3)


// CUs the GEMM launches leave FREE (gget_debug_set key 15; data-parallel runs set it to the number of channels they allow the collective
// library).  Why: an RCCL workgroup (rcclGenericKernel: 256 threads, 261 - 280 registers per lane, 19.7 KiB of LDS - read from the library's
// gfx950 code object) cannot share a CU with ANY 8-wave GEMM workgroup of this file (2 waves x 130 - 216 registers per SIMD), and these
// launches assign their tiles statically: one workgroup that finds its CU taken starts when the others exit and the launch takes twice as
// long (profiles/r02_coresidency.txt measured exactly that with one foreign workgroup).  With R CUs left free the collective's <= R
// workgroups and the GEMM's (CUs - R) never compete.
int g_gemm_cu_reserve = 0;
int gget_gemm_num_cu() {
  static int dev_cus = 0;
  if (!dev_cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    dev_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("GGET_GEMM_NUM_CU")) dev_cus = atoi(e);   // measurement knob: pretend the chip has fewer CUs (tools/halfchip.py)
  }
  int n = dev_cus - (g_gemm_cu_reserve > 0 ? g_gemm_cu_reserve : 0);
  if (g_gemm_cu_reserve > 0) n &= ~7;      // (the XCD permutation of the persistent kernels wants a multiple of 8)
  return n < 8 ? 8 : n;
}

namespace {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int mc_swz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB), lane-linear.
// M0 (LDS base) is written in the same statement that uses it and restored afterwards.
__device__ __forceinline__ void glds16(const void* gsrc, const unsigned char* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(LDS_AS const void*)lds_dst);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(dst)
      : "memory");
}
// same, address = uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: nothing to compute per K-tile
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, const unsigned char* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(LDS_AS const void*)lds_dst);
  const unsigned long long b = (unsigned long long)(uintptr_t)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sb), "s"(dst)
      : "memory");
}
// same with the LDS destination already a wave-uniform byte offset: {s_mov m0, global_load_lds} and nothing else per piece.
// M0 is NOT restored.  It is a reserved register the compiler does not track through inline asm (a clobber entry is
// ignored with a warning), so this is only valid in kernels whose compiler-generated code never uses M0 - true for the
// persistent GEMM (no LDS-direct / movrel / GWS / sendmsg; checked in the ISA: every M0 access there is one of these moves).
__device__ __forceinline__ void glds16m(const unsigned char* sbase, unsigned voff, unsigned lds_off) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_off)
      : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ROWS, bool MC, int NTHREADS, int BK = 64>
struct TileIO {
  static_assert(BK == 64 || BK == 32, "K-tile depth");
  static constexpr int NWAVES = NTHREADS / 64;
  static constexpr int CHUNKS = ROWS * BK / 8;  // 16-byte chunks per tile
  static constexpr int PER_THREAD = CHUNKS / NTHREADS;
  static constexpr int RB = BK * 2;             // bytes per row of a K-contiguous tile
  static constexpr int CR = BK / 8;             // chunks per row of a K-contiguous tile
  static constexpr int RPP = 1024 / RB;         // rows per DMA piece of a K-contiguous tile
  static constexpr int ROWB = ROWS * 2;         // bytes per k-row of an M/N-contiguous tile
  static constexpr int CPR = ROWS / 8;          // chunks per k-row of an M/N-contiguous tile
  static constexpr int NPIECE = ROWS * BK * 2 / 1024;
  // LDS-DMA instructions per wave per tile.  A piece count that is no multiple of the wave count (96-row tiles: 12 pieces, 8 waves)
  // is rounded up and the surplus slots fetch pieces 0.. a second time (same data to the same LDS bytes): every wave then issues the
  // same number of instructions, which is what the counted vmcnt waits need; the DMA is not what bounds these tiles.
  static constexpr int PIECES = (NPIECE + NWAVES - 1) / NWAVES;
  static constexpr bool WRAP = NPIECE % NWAVES != 0;
  static_assert(ROWS * BK * 2 % 1024 == 0, "tile/thread shape");
  static_assert(!MC || ROWS == 128 || ROWS == 192 || ROWS == 256, "M/N-contiguous tile widths");
  static constexpr int NWIN = ROWS / 16;        // 32-byte windows per k-row of an M/N-contiguous tile

  // M/N-contiguous tiles: a k-row is NWIN 32-byte windows (16 rows each); the window holding rows [16w, 16w+16) of
  // k-row kr sits at LDS window mc_lds_win(kr, w) so that one ds_read_b64_tr_b16 (8 k-rows x one window per 32 lanes)
  // touches 8 distinct 32-byte bank windows.  256/128-row tiles (512/256-byte k-rows): XOR with 3 bits of kr.
  // 192-row tiles (384-byte k-rows, 12 windows): rotation by ((kr>>3)&1) + 2*((kr>>1)&1), found by exhaustive search.
  __device__ __forceinline__ static int mc_lds_win(int kr, int w) {
    if (ROWS == 192) { const int t = w + ((kr >> 3) & 1) + ((kr & 2)); return t >= NWIN ? t - NWIN : t; }
    return w ^ mc_swz(kr);
  }
  __device__ __forceinline__ static int mc_src_win(int kr, int wl) {   // inverse: which rows the LDS window wl holds
    if (ROWS == 192) { const int t = wl - ((kr >> 3) & 1) - ((kr & 2)); return t < 0 ? t + NWIN : t; }
    return wl ^ mc_swz(kr);
  }

  // swizzle of the 16-byte chunk index inside a K-contiguous row (conflict-free ds_read_b128 of 16 rows x 1 chunk):
  // 128-byte rows: chunk ^= row & 7 ; 64-byte rows (4 rows per 256-byte bank row): chunk ^= {0,3,2,1}[(row >> 2) & 3]
  __device__ __forceinline__ static int kc_swz(int row) {
    return BK == 64 ? (row & 7) : ((4 - ((row >> 2) & 3)) & 3);
  }

  // global -> LDS by DMA; out-of-range rows are clamped (they only feed C rows/columns that are never stored),
  // the K range must be a full BK-deep tile.
  __device__ __forceinline__ static void glds(unsigned char* lds, const bf16_t* __restrict__ base, int ld, int row0,
                                              int row_lim, int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) glds_piece(lds, base, ld, row0, row_lim, k0, wave, lane, i);
  }
  __device__ __forceinline__ static void glds_piece(unsigned char* lds, const bf16_t* __restrict__ base, int ld, int row0,
                                                    int row_lim, int k0, int wave, int lane, int i) {
    static_assert(!WRAP, "wrapped piece lists are issued by the K-split kernel only");
    {
      const int q = wave + i * NWAVES;
      const bf16_t* p;
      if (!MC) {
        const int r = q * RPP + lane / CR;
        const int lc = (lane % CR) ^ kc_swz(r);
        const int gr = min(row0 + r, row_lim - 1);
        p = base + (size_t)gr * ld + k0 + lc * 8;
      } else {
        const int c = q * 64 + lane;   // 16-byte chunk of the tile image; a DMA instruction covers 64 of them
        const int kr = c / CPR;
        const int sl = c % CPR;
        const int lw = mc_src_win(kr, sl >> 1);
        // a partial last chunk (row_lim % 8 != 0) is still read in full: the leading dimension covers it
        const int gm = min(row0 + lw * 16 + (sl & 1) * 8, ((row_lim + 7) & ~7) - 8);
        p = base + (size_t)(k0 + kr) * ld + gm;
      }
      glds16(p, lds + q * 1024);
    }
  }
  // The same DMA split into a per-tile part (the lane's byte offset from the K-origin of the operand, < 4 GiB) and a
  // per-K-tile part (the uniform K-origin): the persistent kernel computes the offsets once per output tile.
  __device__ __forceinline__ static int piece_index(int wave, int i) {
    const int q = wave + i * NWAVES;
    return WRAP && q >= NPIECE ? q - NPIECE : q;
  }
  __device__ __forceinline__ static unsigned piece_off(int ld, int row0, int row_lim, int wave, int lane, int i, const int* rows = nullptr) {
    const int q = piece_index(wave, i);
    if (!MC) {
      const int r = q * RPP + lane / CR;
      const int lc = (lane % CR) ^ kc_swz(r);
      int gr = min(row0 + r, row_lim - 1);
      if (rows) gr = rows[gr];     // fused row gather (GemmProblem::a_rows)
      return ((unsigned)gr * (unsigned)ld + (unsigned)(lc * 8)) * 2u;
    } else {
      const int c = q * 64 + lane;
      const int kr = c / CPR;
      const int sl = c % CPR;
      const int lw = mc_src_win(kr, sl >> 1);
      const int gm = min(row0 + lw * 16 + (sl & 1) * 8, ((row_lim + 7) & ~7) - 8);
      return ((unsigned)kr * (unsigned)ld + (unsigned)gm) * 2u;
    }
  }
  // EPI_GEGLU_FWD: the B tile of the gate|up projection interleaves gate and up rows wave-wise, so that a wave's accumulators
  // hold gate columns [c, c + W/2) and the SAME up columns: LDS row r of the tile (W = rows per wave) comes from weight row
  // (r % W < W/2 ? 0 : ff) + n0/2 + (r / W) * (W/2) + r % (W/2)   (n0 = tile origin in the 2 ff wide output).
  template <int W>
  __device__ __forceinline__ static unsigned piece_off_geglu(int ld, int n0, int ff, int wave, int lane, int i) {
    static_assert(!MC, "gate|up weights are K-contiguous");
    const int q = wave + i * NWAVES;
    const int r = q * RPP + lane / CR;
    const int lc = (lane % CR) ^ kc_swz(r);
    const int rw = r % W;
    const int gr = (rw < W / 2 ? 0 : ff) + (n0 >> 1) + (r / W) * (W / 2) + (rw % (W / 2));
    return ((unsigned)gr * (unsigned)ld + (unsigned)(lc * 8)) * 2u;
  }
  __device__ __forceinline__ static const bf16_t* k_origin(const bf16_t* base, int ld, int k0) {
    return MC ? base + (size_t)k0 * ld : base + k0;
  }
  __device__ __forceinline__ static void glds_at(unsigned char* lds, const bf16_t* korg, unsigned voff, int wave, int i) {
    glds16s(korg, voff, lds + (wave + i * NWAVES) * 1024);
  }
  // global -> registers (zero fill outside [row_lim) x [k_lim)) : used for a partial last K tile only
  __device__ __forceinline__ static void load(uint4 (&r)[PER_THREAD], const bf16_t* __restrict__ base, int ld,
                                              int row0, int row_lim, int k0, int k_lim, int tid) {
    static_assert(CHUNKS % NTHREADS == 0, "register path: chunks per thread");
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int c = tid + i * NTHREADS;
      int gr, gk;
      const bf16_t* p;
      if (!MC) {
        gr = row0 + c / CR;
        gk = k0 + (c % CR) * 8;
        p = base + (size_t)gr * ld + gk;
      } else {
        gk = k0 + c / CPR;
        gr = row0 + (c % CPR) * 8;
        p = base + (size_t)gk * ld + gr;
      }
      const bool ok = (gr < row_lim) && (gk < k_lim);
      r[i] = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    }
  }
  // registers -> LDS (swizzled)
  __device__ __forceinline__ static void store(const uint4 (&r)[PER_THREAD], unsigned char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int c = tid + i * NTHREADS;
      int off;
      if (!MC) {
        const int row = c / CR, ch = c % CR;
        off = row * RB + ((ch ^ kc_swz(row)) << 4);
      } else {
        const int krow = c / CPR, mc = c % CPR;
        off = krow * ROWB + ((((mc >> 1) ^ mc_swz(krow))) << 5) + ((mc & 1) << 4);
      }
      *reinterpret_cast<uint4*>(lds + off) = r[i];
    }
  }
  // LDS -> MFMA fragment for 16-row sub-tile `sub` (index inside the block tile), 32-deep k-step kk (< BK/32)
  __device__ __forceinline__ static bf16x8_t frag(const unsigned char* lds, int sub, int kk, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    if (!MC) {
      const int row = sub * 16 + l15;
      const int ch = kk * 4 + g;
      const uint4 v = *reinterpret_cast<const uint4*>(lds + row * RB + ((ch ^ kc_swz(row)) << 4));
      return __builtin_bit_cast(bf16x8_t, v);
    } else {
      const int kr0 = kk * 32 + g * 8 + (l15 >> 2);
      const int kr1 = kr0 + 4;
      const int inw = (l15 & 3) * 8;
      const unsigned char* p0 = lds + kr0 * ROWB + (mc_lds_win(kr0, sub) << 5) + inw;
      const unsigned char* p1 = lds + kr1 * ROWB + (mc_lds_win(kr1, sub) << 5) + inw;
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p0));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p1));
      bf16x8_t out;
      out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
      out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
      return out;
    }
  }
};

constexpr int kStages = 3;
// LDS ring depth of the persistent kernel for a given stage size: 4 slots when they fit in 160 KiB, else 3
constexpr int persist_slots(int stage_bytes) { return (160 * 1024 / stage_bytes) >= 4 ? 4 : 3; }
constexpr int kSuper = 8;  // row panels per L2 super-tile

// Epilogue of one wave's 64x64 sub-tile.  After the swapped MFMA a lane owns C[m][n..n+3] (m = lane&15, n-group = lane>>4)
// of each 16x16 accumulator; lanes l and l^16 trade halves of two neighbouring accumulators so that every lane ends
// up with 8 consecutive columns => 16-byte stores, 64-byte row segments, half the store instructions (the C store is
// issue-bound, not bandwidth-bound).
// ILV (RoPE epilogue on a 192-wide block tile, two waves across N): the wave's six accumulators are not six adjacent
// 16-column groups but the even (wave 0) or odd (wave 1) 16-column groups of each 32-channel half of three heads, so
// that a channel c < 32 and its partner c + 32 still sit in the same lane: accumulator j covers columns
// nw + (j/2)*64 + (j%2)*32 .. +15 with nw = tile origin + 16*wave, i.e. head j/2, channels chan0 + (j%2)*32 .. +15.
// 16-byte C store with the non-temporal hint.  Every output of these GEMMs is consumed by a LATER kernel from HBM / the memory-side
// cache anyway (12-100 MB per launch against 4 MB of L2 per XCD); written with the default policy the lines displace the A / B panels the
// running kernel and its neighbours re-read.  Measured in the step (stand-alone loops do not show it): C1 8.54 -> 8.43 ms, C3 55.3 -> 54.1 ms.
__device__ __forceinline__ void stc16(bf16_t* p, const uint4& v) {
  typedef unsigned u4nt __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u4nt*>(p));
}
// GEGLU' epilogue (EPI_GEGLU_BWD, see store_tile) of ONE 16-row block x ONE 64-column group: a0..a3 = the four dh accumulators of the
// group, gv[hf] / uv[hf] = the lane's 8 gate / up pre-activations (row mrow + (lane & 15), columns n64 + hf * 32 + c0 * 8).  Writes
// d gate / d up of the block with whole-line stores.
__device__ __forceinline__ void geglu_bwd_block(const f32x4_t& a0, const f32x4_t& a1, const f32x4_t& a2, const f32x4_t& a3,
                                                const uint4 (&gv)[2], const uint4 (&uv)[2], const GemmProblem& P, int M, int mrow,
                                                int n64, int lane) {
  const int l15 = lane & 15, gq = lane >> 4;
  const bool low = (l15 & 8) == 0;
  const int c0 = 2 * (gq & 1) + (gq >> 1);
  const int ma = mrow + (l15 & 7), mb = ma + 8;
  auto ror8 = [](unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true); };
  uint4 pk[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const f32x4_t& x = hf ? a2 : a0;
    const f32x4_t& y = hf ? a3 : a1;
    float a[8], gt[8], u[8], dg[8], du[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[e]), __float_as_uint(y[e]), false, false);
      a[e] = __uint_as_float(r[0]);
      a[4 + e] = __uint_as_float(r[1]);
    }
    unpack8(gv[hf], gt);
    unpack8(uv[hf], u);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dv = bf2f(f2bf(a[e]));
      float gval, gd;
      gelu_erf_both(gt[e], gval, gd);
      dg[e] = dv * u[e] * gd;
      du[e] = dv * gval;
    }
    pk[hf][0] = pack8(dg);
    pk[hf][1] = pack8(du);
  }
  bf16_t* C = reinterpret_cast<bf16_t*>(P.C);
  const int nst = n64 + (low ? c0 : c0 + 4) * 8;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const uint4 x = pk[0][o], y = pk[1][o];
    const uint4 xr = make_uint4(ror8(x.x), ror8(x.y), ror8(x.z), ror8(x.w));
    const uint4 yr = make_uint4(ror8(y.x), ror8(y.y), ror8(y.z), ror8(y.w));
    const uint4 pa = low ? x : yr;
    const uint4 pb = low ? xr : y;
    const int col = nst + (o == 1 ? P.ff : 0);
    if (ma < M) stc16(C + (size_t)ma * P.ldc + col, pa);
    if (mb < M) stc16(C + (size_t)mb * P.ldc + col, pb);
  }
}

template <int EPI, int MI, int NJ, bool ILV = false>
__device__ __forceinline__ void store_tile(f32x4_t (&acc)[MI][NJ], const GemmProblem& P, int M, int N, int mw, int nw, int lane,
                                           int kslice = 0, int chan0 = 0) {
  const int l15 = lane & 15, gq = lane >> 4;
  auto col_of = [](int j) { return ILV ? (j >> 1) * 64 + (j & 1) * 32 : j * 16; };
  if (EPI == GGET_EPI_ATOMIC_F32) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mw + i * 16 + l15;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = nw + col_of(j) + gq * 4;
        if (n >= N) continue;
        float* c = reinterpret_cast<float*>(P.C) + (size_t)m * P.ldc + n;
        unsafeAtomicAdd(c + 0, acc[i][j][0]); unsafeAtomicAdd(c + 1, acc[i][j][1]);
        unsafeAtomicAdd(c + 2, acc[i][j][2]); unsafeAtomicAdd(c + 3, acc[i][j][3]);
      }
    }
    return;
  }
  if (EPI == GGET_EPI_ROPE) {
    // hf apply_rotary_pos_emb :138-160 on the fp32 accumulators: channel c (< 32) of a head and its partner c + 32 sit
    // in the same lane and register of two accumulators of this wave (j and j+2 when the wave's 64 columns are one head).
    static_assert(EPI != GGET_EPI_ROPE || ILV || NJ == 4, "RoPE epilogue: a wave owns one 64-column head");
    constexpr int NP = ILV ? NJ / 2 : 2;
    // Table reads first, math second, in batches of IB row blocks (<= 48 registers of cos / sin in flight): with per-row early-outs
    // between them the reads were MI * NP dependent L2 round trips per tile - +30 % on the launch at T = 8192, +42 % at 65 536
    // (tools/rope_ab.py).  Rows beyond M read the last row's angles (never stored); columns beyond rope_cols (the v block) keep
    // their values.
    constexpr int IB = MI * NP > 8 ? 2 : MI;
    static_assert(MI % IB == 0, "RoPE epilogue: row blocks per batch");
    // a wave whose columns all lie in the v block (a third of the q|k|v tiles) has nothing to rotate: it skips the position and table
    // reads as well (wave-uniform; the reads were unconditional: 16 KiB of cos / sin rows per wave and tile)
    const bool any_live = (ILV ? nw - chan0 : nw) < P.rope_cols;
#pragma unroll
    for (int i0 = 0; i0 < MI && any_live; i0 += IB) {
      int pos[IB];
      if (P.rope_pos) {   // (the test outside the loop: inside it every position load sat behind its own branch and vmcnt(0))
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) pos[ii] = (int)P.rope_pos[min(mw + (i0 + ii) * 16 + l15, M - 1)];
      } else {
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) pos[ii] = min(mw + (i0 + ii) * 16 + l15, M - 1) % P.rope_S;
      }
      float4 cq[IB][NP], sq[IB][NP];
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int ch = (ILV ? chan0 : p * 16) + gq * 4;
          cq[ii][p] = *reinterpret_cast<const float4*>(P.rope_cos + (size_t)pos[ii] * 32 + ch);
          sq[ii][p] = *reinterpret_cast<const float4*>(P.rope_sin + (size_t)pos[ii] * 32 + ch);
        }
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const int i = i0 + ii;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int ja = ILV ? 2 * p : p, jb = ILV ? 2 * p + 1 : p + 2;
          const int headcol = ILV ? nw - chan0 + p * 64 : nw;
          const bool live = headcol < P.rope_cols;
          const float cc[4] = {cq[ii][p].x, cq[ii][p].y, cq[ii][p].z, cq[ii][p].w};
          const float ss[4] = {sq[ii][p].x, sq[ii][p].y, sq[ii][p].z, sq[ii][p].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = acc[i][ja][e], b = acc[i][jb][e];
            // explicit contraction: which product the compiler folds into the fma moved single bf16 roundings of q / k whenever
            // the surrounding code changed, and the big-weight parity case turns one such flip into 1e-4 of loss
            acc[i][ja][e] = live ? fmaf(a, cc[e], -(b * ss[e])) : a;
            acc[i][jb][e] = live ? fmaf(b, cc[e], a * ss[e]) : b;
          }
        }
      }
    }
  }
  if constexpr (EPI == GGET_EPI_GEGLU_FWD || EPI == GGET_EPI_GEGLU_BWD) {
    // Gated-GELU fused into the two GEMMs around it (hf LlamaMLP.forward :174-176, hidden_act = exact-erf GELU).
    //  FWD (gate|up projection, B rows interleaved by piece_off_geglu): accumulators j < NJ/2 are gate columns nw.., j >= NJ/2
    //      the same up columns; writes gu (bf16 pre-activations, backward input) and h = bf16(gelu(bf16 gate)) * bf16 up -
    //      the same rounding points as the un-fused path (separate bf16 tensors in the reference module).
    //  BWD (dh = dy W_down, N = ff): dh is rounded to bf16 (the reference's gradient tensor), then
    //      d gate = dh * up * gelu'(gate), d up = dh * gelu(gate) go straight to dgu; dh is never written.
    // Stores use the whole-line regrouping below (the launcher guarantees 64-column alignment of every destination).
    static_assert(!ILV && NJ % 4 == 0, "GEGLU epilogue: 64-column wave groups");
    constexpr bool FWD = EPI == GGET_EPI_GEGLU_FWD;
    constexpr int NG = FWD ? NJ / 8 : NJ / 4;   // 64-column groups (of gate columns) this wave owns
    constexpr int UP = FWD ? NJ / 2 : 0;        // accumulator index of the first up column group
    constexpr int NOUT = FWD ? 3 : 2;
    const bool low = (l15 & 8) == 0;
    const int c0 = 2 * (gq & 1) + (gq >> 1);
    const int ff = P.ff;
    auto ror8 = [](unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true); };
    auto xchg = [&](int i, int ja, float* v) {   // the lane's 8 consecutive columns of the accumulator pair (ja, ja + 1), row l15
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][ja][e]), __float_as_uint(acc[i][ja + 1][e]), false, false);
        v[e] = __uint_as_float(r[0]);
        v[4 + e] = __uint_as_float(r[1]);
      }
    };
    bf16_t* C = reinterpret_cast<bf16_t*>(P.C);
    bf16_t* C2 = reinterpret_cast<bf16_t*>(P.C2);
    if constexpr (!FWD) {
      // gate / up operands of one 16-row block at a time, loaded unconditionally (rows clamped: what a clamped row yields is never
      // stored) - behind `if (m < M)` the compiler branched around the loads.  Prefetching the next block's pieces while the current one
      // is computed measured equal inside the C3 / C4 steps and cost the 128-register two-workgroups-per-CU kernel 32 spills: not kept.
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int jq = 0; jq < NG; ++jq) {
          uint4 gv[2], uv[2];
          // whole-line loads (8 rows x 128 B per instruction, the mapping of the stores) and the inverse of the stores' regrouping:
          // a low lane (row < 8 of the block) gets its second piece from lane ^ 8's first load, a high lane its first from lane ^ 8's second
          {
            const int ra = min(mw + i * 16 + (l15 & 7), M - 1), rb = min(mw + i * 16 + (l15 & 7) + 8, M - 1);
            const int nl = nw + jq * 64 + (low ? c0 : c0 + 4) * 8;
            const uint4 ga = *reinterpret_cast<const uint4*>(P.G + (size_t)ra * P.ldg + nl), gb = *reinterpret_cast<const uint4*>(P.G + (size_t)rb * P.ldg + nl);
            const uint4 ua = *reinterpret_cast<const uint4*>(P.G + (size_t)ra * P.ldg + ff + nl), ub = *reinterpret_cast<const uint4*>(P.G + (size_t)rb * P.ldg + ff + nl);
            auto r4 = [&](const uint4& v) { return make_uint4(ror8(v.x), ror8(v.y), ror8(v.z), ror8(v.w)); };
            const uint4 gar = r4(ga), gbr = r4(gb), uar = r4(ua), ubr = r4(ub);
            gv[0] = low ? ga : gbr; gv[1] = low ? gar : gb;
            uv[0] = low ? ua : ubr; uv[1] = low ? uar : ub;
          }
          geglu_bwd_block(acc[i][4 * jq], acc[i][4 * jq + 1], acc[i][4 * jq + 2], acc[i][4 * jq + 3], gv, uv, P, M, mw + i * 16, nw + jq * 64, lane);
        }
      }
      return;
    } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ma = mw + i * 16 + (l15 & 7), mb = ma + 8;
#pragma unroll
      for (int jq = 0; jq < NG; ++jq) {
        uint4 pk[2][NOUT];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int jp = 2 * jq + hf;                 // 32-column group: accumulators 2 jp, 2 jp + 1 (gate columns nw + jp * 32 + c0 * 8 ..)
          float a[8];
          xchg(i, 2 * jp, a);
          if constexpr (FWD) {
            float u[8], gr[8], ur[8], hv[8];
            xchg(i, UP + 2 * jp, u);
            const uint4 gb = pack8(a), ub = pack8(u);
            unpack8(gb, gr);
            unpack8(ub, ur);
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = bf2f(f2bf(gelu_erf(gr[e]))) * ur[e];
            pk[hf][0] = gb;
            pk[hf][1] = ub;
            pk[hf][2] = pack8(hv);
          }
        }
        const int nst = nw + jq * 64 + (low ? c0 : c0 + 4) * 8;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          const uint4 x = pk[0][o], y = pk[1][o];
          const uint4 xr = make_uint4(ror8(x.x), ror8(x.y), ror8(x.z), ror8(x.w));
          const uint4 yr = make_uint4(ror8(y.x), ror8(y.y), ror8(y.z), ror8(y.w));
          const uint4 pa = low ? x : yr;
          const uint4 pb = low ? xr : y;
          bf16_t* base = (FWD && o == 2) ? C2 : C;
          const int ld = (FWD && o == 2) ? P.ldc2 : P.ldc;
          const int col = nst + (o == 1 ? ff : 0);
          if (ma < M) stc16(base + (size_t)ma * ld + col, pa);
          if (mb < M) stc16(base + (size_t)mb * ld + col, pb);
        }
      }
    }
    }
    return;
  }
  const bool odd = gq & 1;
  // Full-line fast path (bf16 C, 64-column groups aligned to 128 B): the exchange below leaves a lane with 16 bytes of
  // row (lane & 15) - an instruction would then touch 16 rows x 64 B.  One more exchange between lanes l and l^8
  // (DPP row_ror:8 on the packed words) regroups two such pieces into rows (lane & 7) and 8 + (lane & 7), so that every
  // store instruction writes 8 rows x 128 B = whole cache lines: the C store is what a K = 768 GEMM loses most time in
  // (the write path takes ~9 B/clk/CU when all CUs store at once), and whole lines cut it by a third to a half.
  if constexpr ((EPI == GGET_EPI_NONE || EPI == GGET_EPI_RESIDUAL || EPI == GGET_EPI_ROPE) && !ILV && (NJ % 4 == 0 || NJ == 6)) {
    // NJ == 6 (96-column wave tile of the 128x192 block tile): the wave whose tile starts mid-line stores its first 32
    // columns the plain way and regroups the other 64, the other wave the reverse
    const int lead = (NJ == 6 && (nw & 63) == 32) ? 1 : 0;
    // (a ragged width - N % 64 != 0, plain stores only - keeps the whole-line path: the chunk that straddles N is cut to 8 B)
    if (((N & 63) == 0 || EPI == GGET_EPI_NONE) && (P.ldc & 63) == 0 && ((nw & 63) == 0 || lead) && ((uintptr_t)P.C & 127) == 0) {
      const bool low = (l15 & 8) == 0;
      const int c0 = 2 * (gq & 1) + (gq >> 1);   // 16-byte chunk (of the 8 in a 64-column group) this lane holds for the first jp
      // residual pieces of ONE 16-row block at a time, loaded unconditionally (row / column clamped: what a clamped piece adds lands in
      // values that are never stored).  One load behind each `if (m < M && n < N)` was a branch + vmcnt(0) per piece; all pieces of the
      // wave tile at once measured slower INSIDE the step (profiles/r03_step_experiments.txt item 15).
      uint4 rrow[EPI == GGET_EPI_RESIDUAL ? NJ / 2 : 1];
      auto fetch_res = [&](int i) {
        if constexpr (EPI == GGET_EPI_RESIDUAL) {
          // (the same pieces as whole lines + the inverse of the stores' regrouping, as in the GEGLU' epilogue: step 7.14 -> 7.32 ms - item 21)
          const int m = min(mw + i * 16 + l15, M - 1);
#pragma unroll
          for (int jp = 0; jp < NJ / 2; ++jp) rrow[jp] = *reinterpret_cast<const uint4*>(P.R + (size_t)m * P.ldc + min(nw + jp * 32 + c0 * 8, N - 8));
        }
      };
      auto piece = [&](int i, int jp, int m) {   // epilogue math of (i, jp): the lane's 8 columns of row m, packed
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * jp][e]), __float_as_uint(acc[i][2 * jp + 1][e]), false, false);
          v[e] = __uint_as_float(r[0]);
          v[4 + e] = __uint_as_float(r[1]);
        }
        if constexpr (EPI == GGET_EPI_RESIDUAL) {
          float r[8];
          unpack8(rrow[jp], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        return pack8(v);
      };
      auto ror8 = [](unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true); };
      bf16_t* C = reinterpret_cast<bf16_t*>(P.C);
      const int* crows = EPI == GGET_EPI_NONE ? P.c_rows : nullptr;   // fused row scatter (see GemmProblem::c_rows)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fetch_res(i);
        const int m = mw + i * 16 + l15;          // the row whose values this lane holds before the regrouping
        const int ma = mw + i * 16 + (l15 & 7), mb = ma + 8;
        // (a negative c_rows entry = row not stored: folded into the row test by moving the row behind M)
        const int ia = crows && ma < M ? crows[ma] : ma, ib = crows && mb < M ? crows[mb] : mb, im = crows && m < M ? crows[m] : m;
        const size_t ra = (size_t)max(ia, 0), rb = (size_t)max(ib, 0), rm = (size_t)max(im, 0);
        const int ma_ = ia < 0 ? M : ma, mb_ = ib < 0 ? M : mb, m_ = im < 0 ? M : m;
#pragma unroll
        for (int jq = 0; jq < NJ / 4; ++jq) {
          const int ja = 2 * jq + lead;           // group = (ja, ja + 1): 64 columns from nw + 32 * ja
          const uint4 x = lead ? piece(i, 2 * jq + 1, m) : piece(i, 2 * jq, m);
          const uint4 y = lead ? piece(i, 2 * jq + 2 < NJ / 2 ? 2 * jq + 2 : NJ / 2 - 1, m) : piece(i, 2 * jq + 1, m);
          const uint4 xr = make_uint4(ror8(x.x), ror8(x.y), ror8(x.z), ror8(x.w));
          const uint4 yr = make_uint4(ror8(y.x), ror8(y.y), ror8(y.z), ror8(y.w));
          // first instruction: rows 0..7 of the 16 (low lanes keep their first piece, high lanes carry the second piece of
          // row-8); second instruction: rows 8..15
          const uint4 pa = low ? x : yr;
          const uint4 pb = low ? xr : y;
          const int n = nw + ja * 32 + (low ? c0 : c0 + 4) * 8;
          if (n + 8 <= N) {
            if (ma_ < M) stc16(C + ra * P.ldc + n, pa);
            if (mb_ < M) stc16(C + rb * P.ldc + n, pb);
          } else if (n < N) {
            if (ma_ < M) *reinterpret_cast<uint2*>(C + ra * P.ldc + n) = make_uint2(pa.x, pa.y);
            if (mb_ < M) *reinterpret_cast<uint2*>(C + rb * P.ldc + n) = make_uint2(pb.x, pb.y);
          }
        }
        if constexpr (NJ == 6) {   // the remaining 32 columns: 16 rows x 64 B per instruction
          const int jl = lead ? 0 : 2;
          const uint4 z = lead ? piece(i, 0, m) : piece(i, NJ / 2 - 1, m);
          const int n = nw + jl * 32 + c0 * 8;
          if (m_ < M && n + 8 <= N) stc16(C + rm * P.ldc + n, z);
          else if (m_ < M && n < N) *reinterpret_cast<uint2*>(C + rm * P.ldc + n) = make_uint2(z.x, z.y);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mw + i * 16 + l15;
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) {
      // even 16-lane rows keep accumulator 2jp and take the partner row's quarter of it, odd rows keep 2jp+1:
      // v_permlane16_swap (odd rows of the first operand <-> even rows of the second) does the whole exchange in one
      // VALU instruction per register - even rows end up with {own X, X of row+1}, odd rows with {Y of row-1, own Y},
      // i.e. 8 consecutive columns in (first, second) order on every lane.  (A __shfl_xor here is a ds_bpermute through
      // the LDS crossbar: 64 dependent ones per 256x256 tile cost more than the stores themselves.)
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * jp][e]), __float_as_uint(acc[i][2 * jp + 1][e]), false, false);
        v[e] = __uint_as_float(r[0]);
        v[4 + e] = __uint_as_float(r[1]);
      }
      const int n = nw + (odd ? col_of(2 * jp + 1) : col_of(2 * jp)) + (gq & 2) * 4;
      if (m >= M || n >= N) continue;
      if (EPI == GGET_EPI_NONE && P.c_rows && P.c_rows[m] < 0) continue;
      if (EPI == GGET_EPI_SLAB_F32) {
        float* fp = reinterpret_cast<float*>(P.C) + (size_t)kslice * P.slab_stride + (size_t)m * P.ldc + n;
        *reinterpret_cast<float4*>(fp) = make_float4(v[0], v[1], v[2], v[3]);
        if (n + 8 <= N) *reinterpret_cast<float4*>(fp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        continue;
      }
      bf16_t* cp = reinterpret_cast<bf16_t*>(P.C) + (EPI == GGET_EPI_NONE && P.c_rows ? (size_t)P.c_rows[m] : (size_t)m) * P.ldc + n;
      if (n + 8 <= N) {
        if (EPI == GGET_EPI_RESIDUAL) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(P.R + (size_t)m * P.ldc + n), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        stc16(cp, pack8(v));
      } else {  // N % 8 == 4 : only the first half of this chunk exists
        if (EPI == GGET_EPI_RESIDUAL) {
          const uint2 r = *reinterpret_cast<const uint2*>(P.R + (size_t)m * P.ldc + n);
          v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
          v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(cp) = o;
      }
    }
  }
}

// (tile id inside one problem) -> (m0, n0): 8-row super-tiles walked column-wise (L2 reuse of A and B panels)
__device__ __forceinline__ void tile_origin(const GemmProblem& P, int M, int lt, int BM, int BN, int kSup, int& m0, int& n0) {
  const int tiles_m = (M + BM - 1) / BM;
  const int per_super = kSup * P.tiles_n;
  const int sup = lt / per_super, rem = lt - sup * per_super;
  const int rows_here = min(kSup, tiles_m - sup * kSup);
  m0 = (sup * kSup + rem % rows_here) * BM;
  n0 = (rem / rows_here) * BN;
}

template <int WM, int WN, bool A_MC, bool B_MC, int EPI>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN) / 4) gemm_kernel(const GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = WM * 64, BN = WN * 64, NT = WM * WN * 64;
  using TA = TileIO<BM, A_MC, NT>;
  using TB = TileIO<BN, B_MC, NT>;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int PIECES = TA::PIECES + TB::PIECES;  // LDS-DMA instructions per wave per K-tile
  constexpr int MI = 4, NJ = 4;

  // ---- tile id: XCD-contiguous, then 8-row super-tiles walked column-wise inside the problem
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7;
  const int xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GGET_MAX_GROUP; ++i)
    if (i < g.count && tile >= g.p[i].tile_begin) pi = i;
  const GemmProblem& P = g.p[pi];
  const int M = P.m_dev ? *P.m_dev : P.M;
  const int K = P.k_dev ? *P.k_dev : P.K;
  const int N = P.N;
  int m0, n0;
  tile_origin(P, P.M, tile - P.tile_begin, BM, BN, g.super, m0, n0);
  if (m0 >= M) return;
  // split-K slice of this block (gridDim.y slices, 64-aligned)
  const int ktiles = (K + 63) >> 6;
  const int per = (ktiles + gridDim.y - 1) / gridDim.y;
  const int kbeg = blockIdx.y * per * 64;
  const int kend = min(K, kbeg + per * 64);
  if (kbeg >= kend) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const unsigned char* a_l) {
    const unsigned char* b_l = a_l + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = TA::frag(a_l, wm * MI + i, kk, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = TB::frag(b_l, wn * NJ + j, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  // per-lane DMA offsets of this block's tile (fixed), K-origin advanced per K-tile (uniform)
  unsigned offA[TA::PIECES], offB[TB::PIECES];
#pragma unroll
  for (int i = 0; i < TA::PIECES; ++i) offA[i] = TA::piece_off(P.lda, m0, M, wave, lane, i);
#pragma unroll
  for (int i = 0; i < TB::PIECES; ++i) offB[i] = TB::piece_off(P.ldb, n0, N, wave, lane, i);
  auto issue = [&](int t, int slot) {
    unsigned char* s = smem + slot * STAGE;
    const bf16_t* ka = TA::k_origin(P.A, P.lda, kbeg + t * 64);
    const bf16_t* kb = TB::k_origin(P.B, P.ldb, kbeg + t * 64);
#pragma unroll
    for (int i = 0; i < TA::PIECES; ++i) TA::glds_at(s, ka, offA[i], wave, i);
#pragma unroll
    for (int i = 0; i < TB::PIECES; ++i) TB::glds_at(s + A_BYTES, kb, offB[i], wave, i);
  };

  const int nfull = (kend - kbeg) >> 6;
  const bool tail = ((kend - kbeg) & 63) != 0;
  const int abl = g.ablate;  // diagnostics only (GGET_GEMM_ABLATE): 1 = no DMA, 2 = no MFMA/ds_read, 4 = no C store
  // ---- pipelined main loop over the full K tiles (3-slot ring, 2 tiles in flight)
  if (nfull > 0 && !(abl & 1)) issue(0, 0);
  if (nfull > 1 && !(abl & 1)) issue(1, 1);
  int slot = 0;
  for (int t = 0; t < nfull; ++t) {
    if (t + 1 < nfull) vm_wait<PIECES>(); else vm_wait<0>();   // this wave's pieces of tile t have landed
    __syncthreads();                                           // everyone's have; tile t-1 is fully consumed
    if (t + 2 < nfull && !(abl & 1)) issue(t + 2, slot == 0 ? 2 : slot - 1); // refill the slot tile t-1 lived in
    if (!(abl & 2)) compute(smem + slot * STAGE);
    slot = slot == 2 ? 0 : slot + 1;
  }
  // ---- partial last K tile: zero-filling register path (rare: vocab-sized or data-dependent K)
  if (tail) {
    uint4 ra[TA::PER_THREAD], rb[TB::PER_THREAD];
    const int kt = kbeg + nfull * 64;
    TA::load(ra, P.A, P.lda, m0, M, kt, kend, tid);
    TB::load(rb, P.B, P.ldb, n0, N, kt, kend, tid);
    __syncthreads();
    TA::store(ra, smem, tid);
    TB::store(rb, smem + A_BYTES, tid);
    __syncthreads();
    compute(smem);
  }

  // ---- epilogue
  if (abl & 4) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 1.2345e-30f) reinterpret_cast<bf16_t*>(P.C)[0] = 1;  // keep the accumulators live
    return;
  }
  // a width that is not a multiple of 4 (odd vocabularies) is stored up to the next multiple: the caller's ldc covers
  // the pad columns (checked by the launcher) and their values (the clamped last B row again) are never read
  store_tile<EPI, MI, NJ>(acc, P, M, (N + 3) & ~3, m0 + wm * 64, n0 + wn * 64, lane, blockIdx.y);
}


// Persistent variant (static shapes, K % 64 == 0, no split-K): one block per CU walks its list of output tiles and
// treats all their K-tiles as ONE stream through the LDS ring, so the DMA of the next tile's first K-tiles is already
// in flight while the current tile finishes and its C stores drain: prologue and epilogue of every tile but the
// first/last are hidden (at K = 768 they are ~30 % of a non-persistent tile's life).
// Counted waits stay valid with stores in flight: vmcnt <= PIECES means >= (stores + PIECES) older operations have
// retired, and loads retire in order among themselves, so the oldest PIECES loads (the tile being waited for) are in.
//
// SK (aligned split-K of the LAST, partial round; single problem, host-side shapes): a launch whose tile count is no multiple of the CU
// count - the var-len token layout: M = the batch's real tokens - would leave most CUs idle in its last round.  Full rounds run as
// always; the rem = tiles % G tiles of the last round share the idle CUs:
//   rem <= G/2 : every tile's K range is cut into s = min(G / rem, 4) equal parts, part p of tile t on block p * rem + t;
//   rem >  G/2 : block t < rem computes K-tiles [0, a) of tile t, the H = G - rem helper blocks each take the tails [a, nk) of up to
//                q = ceil(rem / H) tiles (t = h, h + H, ...), a = ceil(nk q / (q + 1)) balances the two.
// Blocks that work on the same K range at the same time sit next to each other, so the operand panels keep their L2 reuse (a first
// version handed every block one contiguous range of (tile, K-tile) units: perfectly balanced, and 40 % SLOWER - neighbouring blocks
// were at different K offsets of the same panels and every K-slice was fetched once per block).  The block with a tile's K-tile 0 owns
// it; the others write fp32 partial accumulators to their slot of the workspace and publish a flag (agent-scope release, Guideline 16
// of the CDNA guide); the owner adds them (acquire) and runs the epilogue.
//
// MODE 2 (the dh + GEGLU' launch, whose epilogue - 96 KiB of gate|up read, 96 KiB of d gate|up written and an erf-GELU value + derivative
// per element for every 192x128 tile - takes as long as its K = 768 loop): TWO blocks per CU (2-slot rings of 40 KiB, <= 128 VGPRs),
// grid = 2 x CUs, so that one block's epilogue runs under the other's K-loop with its own VMEM counter and barrier.  -5 % on the launch
// (cold 74.5 -> 70.5 us), -0.02 ms on the C1 step.  Measured and dropped (profiles/r03_step_experiments.txt, item 11): the same epilogue
// run one 16-row block at a time inside the NEXT tile's K-loop of a one-block-per-CU kernel (operands prefetched by LDS-DMA into a
// landing area) - the in-order VMEM queue makes every K-tile wait for the HBM-latency operand loads issued before it: -2 % / nothing.
// (Round 4: the LDS-DMA issued by the first four waves only - what pays in the K-split kernel below - measured +0.3 % here: every wave
//  keeps issuing its own share.  profiles/r04_step_experiments.txt item 5.)
template <int BM, int BN, int BK, int WM, int WN, bool A_MC, bool B_MC, int EPI, int NSLOT_ = 0, bool SK = false, int MODE = 0>
__global__ void __launch_bounds__(WM * WN * 64, MODE == 2 ? 4 : ((WM * WN) >= 8 ? 2 : 1)) gemm_persist_kernel(const GemmGroup g, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = WM * WN * 64;
  using TA = TileIO<BM, A_MC, NT, BK>;
  using TB = TileIO<BN, B_MC, NT, BK>;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NSLOT = NSLOT_ > 0 ? NSLOT_ : persist_slots(STAGE);   // 256x128x64: 3 x 48 KiB ; 256x256x32, 128x128x64: 4 x 32 KiB
  constexpr int PIECES = TA::PIECES + TB::PIECES;
  constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
  constexpr int KSH = BK == 64 ? 6 : 5;
  constexpr bool ILV = EPI == GGET_EPI_ROPE && BN == 192;   // see store_tile
  static_assert(!ILV || WN == 2, "interleaved RoPE tile: two waves across N");
  static_assert(MODE == 0 || MODE == 2, "persistent kernel modes");

  const int G = gridDim.x;                                     // multiple of 8
  const int perm = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // XCD-contiguous inside every round
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // a row count that lives on the device (rows selected by the masking, single-problem launches only): the tile list is
  // rebuilt from it here, the host sized the grid for the capacity
  const bool dyn_m = g.p[0].m_dev != nullptr;
  const int m_dyn = dyn_m ? *g.p[0].m_dev : 0;
  if (dyn_m) total_tiles = ((m_dyn + BM - 1) / BM) * g.p[0].tiles_n;
  // a device-side K (weight gradient over the selected rows, single problem): whole K-tiles up to round_up(K, BK) - the
  // caller guarantees that the rows between K and the capacity are finite on one side and zero on the other
  const bool dyn_k = g.p[0].k_dev != nullptr;
  const int k_tiles_dyn = dyn_k ? max(1, (min(*g.p[0].k_dev, g.p[0].K) + BK - 1) >> KSH) : 0;
  struct Ctx { int pi, m0, n0, nk, M, kb, owner, nwait, slot; };
  // SK pieces: kb = first K-tile, nk = K-tiles of the piece; owner = this block runs the epilogue after adding `nwait` partials (slots
  // slot, slot + rem, ...); otherwise it writes its accumulators to `slot`
  const int nkf = SK ? (g.p[0].K >> KSH) : 1;
  auto tile_at = [&](int r, Ctx& c) -> bool {
    if constexpr (SK) {
      const int R = g.sk_rounds, rem = g.sk_rem, parts = g.sk_parts;
      int tile, kb, ke;
      c.owner = 1; c.nwait = 0; c.slot = 0;
      if (r < R) {
        tile = r * G + perm; kb = 0; ke = nkf;
      } else if (parts > 1) {                       // equal K parts
        if (r > R || perm >= rem * parts) return false;
        const int part = perm / rem, t = perm - part * rem;
        tile = R * G + t;
        kb = (int)((long)nkf * part / parts); ke = (int)((long)nkf * (part + 1) / parts);
        c.owner = part == 0; c.nwait = parts - 1; c.slot = part == 0 ? t : (part - 1) * rem + t;
      } else {                                      // owners + helpers
        const int H = G - rem, a = g.sk_a;
        if (perm < rem) {
          if (r > R) return false;
          tile = R * G + perm; kb = 0; ke = a;
          c.nwait = 1; c.slot = perm;
        } else {
          const int t = (perm - rem) + (r - R) * H;
          if (t >= rem) return false;
          tile = R * G + t; kb = a; ke = nkf;
          c.owner = 0; c.slot = t;
        }
      }
      // (the divisions above run on the vector unit: tell the compiler again that the results are wave-uniform - the K origins they
      //  lead to feed the scalar operand of the LDS-DMA instruction)
      tile = __builtin_amdgcn_readfirstlane(tile);
      kb = __builtin_amdgcn_readfirstlane(kb);
      ke = __builtin_amdgcn_readfirstlane(ke);
      c.slot = __builtin_amdgcn_readfirstlane(c.slot);
      c.owner = __builtin_amdgcn_readfirstlane(c.owner);
      c.pi = 0;
      c.M = g.p[0].M;
      tile_origin(g.p[0], c.M, tile, BM, BN, g.super, c.m0, c.n0);
      c.kb = kb;
      c.nk = ke - kb;
      return true;
    }
    c.kb = 0;
    int tile = r * G + perm;
    if constexpr (MODE == 2) {
      // the last, partial round goes to the blocks dispatched first (one per CU, spread over the XCDs), not to the first XCDs
      const int full = total_tiles / G;
      if (r == full) tile = full * G + (int)blockIdx.x;
      if (r > full || (r == full && (int)blockIdx.x >= total_tiles - full * G)) return false;
    }
    if (tile >= total_tiles) return false;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GGET_MAX_GROUP; ++i)
      if (i < g.count && tile >= g.p[i].tile_begin) pi = i;
    c.pi = pi;
    c.M = dyn_m ? m_dyn : g.p[pi].M;
    tile_origin(g.p[pi], c.M, tile - g.p[pi].tile_begin, BM, BN, g.super, c.m0, c.n0);
    c.nk = dyn_k ? k_tiles_dyn : (g.p[pi].K >> KSH);
    return true;
  };

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if constexpr (MODE == 2) {
    // Anti-phase start of the two workgroups of a CU: launched together and given equal work they run in lockstep - both in their
    // K-loops (sharing the matrix pipe), then both in their epilogues (matrix pipe idle) - and the second workgroup buys nothing.  The
    // workgroup that arrives second on its CU (arrival counter keyed by the hardware CU id) waits about one tile's K-loop before it
    // touches memory; from then on one workgroup's epilogue runs beside the other's K-loop.
    if (g.stagger_ticks > 0) {
      if (tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4);     // HW_REG_HW_ID[15:8] = {SE, SH, CU}
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
        *reinterpret_cast<volatile unsigned*>(smem) = atomicAdd(g.cu_slots + ((xcc & 7u) << 8) + (hw & 255u), 1u) & 1u;
      }
      __syncthreads();
      const unsigned second = *reinterpret_cast<volatile unsigned*>(smem);
      __syncthreads();   // (the word is read before the first DMA piece may land on it)
      if (second) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)g.stagger_ticks) __builtin_amdgcn_s_sleep(32);
      }
    }
  }

  Ctx ic, cc;
  int ir = 0, ik = 0, cr = 0, ck = 0;
  bool ivalid = tile_at(0, ic);
  bool cvalid = ivalid;
  cc = ic;
  int islot = 0, cslot = 0, inflight = 0;
  // issue side: operand bases / leading dims of the tile being streamed and the per-lane DMA offsets, refreshed only
  // when the stream moves on to the next output tile (a K-tile's DMA is then PIECES x {M0, global_load_lds})
  // Everything a piece needs is wave-uniform scalar state kept across K-tiles: the K-origin pointers of the two operands
  // (advanced by a constant byte stride per K-tile) and the LDS byte offset of this wave's first piece in the current slot;
  // a piece is then one s_add (constant piece offset), s_mov m0 and the global_load_lds.
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(LDS_AS const void*)smem) + (unsigned)wave * 1024u;
  const unsigned char* kA = nullptr;
  const unsigned char* kB = nullptr;
  long strideA = 0, strideB = 0;
  unsigned islot_off = lds0;
  unsigned offA[TA::PIECES], offB[TB::PIECES];
  auto load_issue_tile = [&]() {
    if (!ivalid) return;
    const GemmProblem& P = g.p[ic.pi];
    strideA = A_MC ? (long)BK * P.lda * 2 : (long)BK * 2;
    strideB = B_MC ? (long)BK * P.ldb * 2 : (long)BK * 2;
    kA = reinterpret_cast<const unsigned char*>(P.A) + (SK ? ic.kb * strideA : 0);
    kB = reinterpret_cast<const unsigned char*>(P.B) + (SK ? ic.kb * strideB : 0);
    if (P.b_tile_off)     // (slot-sorted head: one weight block per row tile; the value is wave-uniform - tell the compiler, kB feeds an SGPR pair)
      kB += (long)__builtin_amdgcn_readfirstlane(P.b_tile_off[__builtin_amdgcn_readfirstlane(ic.m0) / BM]) * 2;
#pragma unroll
    for (int i = 0; i < TA::PIECES; ++i) offA[i] = TA::piece_off(P.lda, ic.m0, ic.M, wave, lane, i, A_MC ? nullptr : P.a_rows);
#pragma unroll
    for (int i = 0; i < TB::PIECES; ++i) {
      if constexpr (EPI == GGET_EPI_GEGLU_FWD) offB[i] = TB::template piece_off_geglu<BN / WN>(P.ldb, ic.n0, P.ff, wave, lane, i);
      else offB[i] = TB::piece_off(P.ldb, ic.n0, P.N, wave, lane, i);
    }
  };
  load_issue_tile();
  auto issue_piece = [&](int q) {       // piece q of the K-tile (ic, ik) into slot islot
    if (q < TA::PIECES) glds16m(kA, offA[q < TA::PIECES ? q : 0], islot_off + (unsigned)(q * TA::NWAVES * 1024));
    else glds16m(kB, offB[q >= TA::PIECES ? q - TA::PIECES : 0], islot_off + (unsigned)(A_BYTES + (q - TA::PIECES) * TB::NWAVES * 1024));
  };
  auto issue_advance = [&]() {
    islot = islot == NSLOT - 1 ? 0 : islot + 1;
    islot_off = islot == 0 ? lds0 : islot_off + (unsigned)STAGE;
    ++inflight;
    kA += strideA;
    kB += strideB;
    if (++ik == ic.nk) { ik = 0; ++ir; ivalid = tile_at(ir, ic); load_issue_tile(); }
  };
  auto issue_next = [&]() {
    if (!ivalid) return;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) issue_piece(q);
    issue_advance();
  };
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) issue_next();
  // wait until this wave's pieces of the OLDEST K-tile in flight have landed: the (inflight - 1) younger K-tiles may
  // stay outstanding (stores in flight only make the count conservative, see above)
  // (steady state: NSLOT - 1 K-tiles in flight; in the tail of the stream everything is waited for)
  auto wait_oldest = [&]() {
    if (inflight == NSLOT - 1) vm_wait<(NSLOT - 2) * PIECES>();
    else vm_wait<0>();
  };
  auto finish_tile = [&]() {
    const GemmProblem& P = g.p[cc.pi];
    bool store = true;
    if constexpr (SK) {
      typedef __attribute__((address_space(1))) unsigned gu32;
      constexpr int SLOT4 = BM * BN / 4;                       // float4 per slot; thread t holds float4 (i * NJ + j) * NT + t
      if (!cc.owner) {
        // a later part of a tile another block owns: hand the partial accumulators over (payload, every wave drains its stores,
        // barrier, ONE lane's agent-scope release, the flag)
        float4* slot = reinterpret_cast<float4*>(g.sk_partial) + (size_t)cc.slot * SLOT4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            slot[(i * NJ + j) * NT + tid] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store((gu32*)(g.sk_flags + cc.slot), g.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        store = false;
      } else {
        // the owner of a split tile: add the other parts (in slot order: the sum is deterministic), then the usual epilogue
        for (int w = 0; w < cc.nwait; ++w) {
          const int sl = cc.slot + w * g.sk_rem;
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load((gu32*)(g.sk_flags + sl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.sk_epoch) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1u << 24)) {   // bounded: a block that never becomes resident must not hang the device
                __hip_atomic_store((gu32*)(g.sk_flags + kStreamKBlocks), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
          const float4* slot = reinterpret_cast<const float4*>(g.sk_partial) + (size_t)sl * SLOT4;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const float4 v = slot[(i * NJ + j) * NT + tid];
              acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
            }
        }
      }
    }
    if (store && (g.ablate != 32 || acc[0][0][0] == 123.456f))   // GGET_GEMM_ABLATE=32: persistent kernel without the epilogue (timing only)
      store_tile<EPI, MI, NJ, ILV>(acc, P, cc.M, (P.N + 3) & ~3, cc.m0 + wm * (MI * 16),
                                   EPI == GGET_EPI_GEGLU_FWD ? (cc.n0 >> 1) + wn * (NJ * 8) : cc.n0 + wn * (ILV ? 16 : NJ * 16), lane, 0, wn * 16);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    ck = 0;
    ++cr;
    cvalid = tile_at(cr, cc);
  };
  while (cvalid) {
    wait_oldest();
    __syncthreads();   // publishes that tile; also every wave is done with the slot consumed last iteration
    // ... which is the slot refilled here.  The DMA pieces of K-tile t+NSLOT-1 go between the MFMA groups of this K-tile
    // (a piece is {s_add, s_mov m0, global_load_lds}), so their issue overlaps the matrix pipe: +5..20 % on NT / NN when
    // introduced; on the TN kernel it only paid (-1 % of the step) once the pieces had become that cheap.  Measured slower
    // (profiles/r01_gemm_kloop_stamps.txt): a ping-pong schedule with the two waves of a SIMD half a K-tile apart, the
    // barrier in the middle of the MFMA stream, two K-tiles per barrier with a shorter prefetch distance.
    const bool did = ivalid;
    --inflight;
    {
      const unsigned char* a_l = smem + cslot * STAGE;
      const unsigned char* b_l = a_l + A_BYTES;
      constexpr int KK = BK / 32, NG = KK * MI;
      bf16x8_t af[KK][MI], bf[KK][NJ];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[kk][i] = TA::frag(a_l, wm * MI + i, kk, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[kk][j] = TB::frag(b_l, ILV ? (j >> 1) * 4 + (j & 1) * 2 + wn : wn * NJ + j, kk, lane);
      }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
          const int grp = kk * MI + i;
          // (2-slot ring: the K-tile issued here is waited for at the top of the NEXT iteration - all its pieces go right behind the
          //  first MFMA group so that they have the rest of the iteration to land, not one MFMA group: C1 step 7.41 -> 7.36 ms;
          //  spread over the first half of the groups 7.39, over the first two 7.37)
          constexpr int NGI = NSLOT == 2 ? 1 : NG;
          // (2-slot ring, measured in round 4: the upper four waves issuing their burst behind a LATER group than the lower four - so that
          //  the two waves of a SIMD do not sit in DMA issue together - is monotonically slower, 7.035 / 7.059 / 7.072 / 7.089 / 7.114 ms for
          //  groups 0..4: what the later issue loses in landing time outweighs it)
          if (did && grp < NGI) {
#pragma unroll
            for (int q = grp * PIECES / NGI; q < (grp + 1) * PIECES / NGI; ++q) issue_piece(q);
          }
        }
    }
    if (did) issue_advance();
    cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
    if (++ck == cc.nk) finish_tile();
  }
}

// In-block K split for the launches with ONE output tile per CU (every N = d GEMM of the step at T = 8192: o / down projections,
// the dgrads into the residual stream, and the grouped weight gradients): the 8 waves are 2 (M) x 2 (N) x 2 (K) instead of
// 4 x 2, so a wave owns a sub-tile twice as large (64x96 of 128x192, 96x96 of 192x192) but only one 32-deep half of every
// 64-deep K-tile.  Same MFMA count per wave, 10 instead of 16 (12 instead of 18) fragment reads per K-tile: the 128x192 tile
// is bound by LDS fragment traffic (profiles/r02_gemm_structure_experiments.txt: +17 % / +8 % in the pipe micro-benchmark).
// At the end the two K halves of a sub-tile (waves w and w ^ 4, same SIMD) are added through LDS - the ring is free by
// then, one tile per block - each wave keeps one row half, and the usual epilogue stores it.
// DW = waves that issue the LDS-DMA.  4 (default since round 4): only the first wave of every SIMD, twice the pieces each - the other
// four never enter the vector-memory queue during the K-loop and keep issuing MFMAs while the first four sit in DMA issue (a DMA piece
// costs its wave 60-180 cycles of issue when the queue is busy).  Bit-identical results, C1 step -1.0 ... -1.2 % on two boxes (7.173 ->
// 7.100, 7.402 -> 7.314 ms, same process, alternated: profiles/r04_step_experiments.txt item 5).  8 = every wave its own share
// (rounds 2-3; g_gemm_variant bit 7).
template <int BM, int BN, bool A_MC, bool B_MC, int EPI, int NSLOT_ = 0, int DW = 8>
__global__ void __launch_bounds__(512, 2) gemm_ks_kernel(const GemmGroup g, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BK = 64, NT = DW * 64;     // (threads that take part in the DMA: TileIO's piece lists are cut for them)
  using TA = TileIO<BM, A_MC, NT, BK>;
  using TB = TileIO<BN, B_MC, NT, BK>;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NSLOT = NSLOT_ > 0 ? NSLOT_ : persist_slots(STAGE);
  constexpr int PIECES = TA::PIECES + TB::PIECES;
  // row blocks of a wave's sub-tile: after the K loop wave (wk = 0) keeps blocks [0, H0), its K partner (wk = 1) blocks [H0, MI)
  constexpr int MI = BM / 2 / 16, NJ = BN / 2 / 16, H0 = (MI + 1) / 2, H1 = MI - H0, HT = H0 * NJ;
  static_assert(H1 >= 1, "at least two row blocks per wave");
  static_assert(8 * HT * 64 * 16 <= NSLOT * STAGE, "accumulator exchange fits in the ring");
  static_assert(!TB::WRAP, "only the A tile may have a wrapped piece list");

  const int G = gridDim.x;
  const int tile = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // XCD-contiguous
  if (tile >= total_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GGET_MAX_GROUP; ++i)
    if (i < g.count && tile >= g.p[i].tile_begin) pi = i;
  const GemmProblem& P = g.p[pi];
  int m0, n0;
  tile_origin(P, P.M, tile - P.tile_begin, BM, BN, g.super, m0, n0);
  const int nk = P.K >> 6;

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int pw = DW == 8 ? wave : (wave & (DW - 1));     // this wave's index in the piece lists (only the issuing waves use it)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(LDS_AS const void*)smem) + (unsigned)pw * 1024u;
  const unsigned char* kA = reinterpret_cast<const unsigned char*>(P.A);
  const unsigned char* kB = reinterpret_cast<const unsigned char*>(P.B);
  const long strideA = A_MC ? (long)BK * P.lda * 2 : (long)BK * 2;
  const long strideB = B_MC ? (long)BK * P.ldb * 2 : (long)BK * 2;
  unsigned offA[TA::PIECES], offB[TB::PIECES];
#pragma unroll
  for (int i = 0; i < TA::PIECES; ++i) offA[i] = TA::piece_off(P.lda, m0, P.M, pw, lane, i);
#pragma unroll
  for (int i = 0; i < TB::PIECES; ++i) offB[i] = TB::piece_off(P.ldb, n0, P.N, pw, lane, i);
  int islot = 0, cslot = 0, issued = 0;
  unsigned islot_off = lds0;
  // (a wrapped A piece - TileIO::WRAP - lands where the piece it repeats lives: a wave-uniform correction of the LDS offset)
  int adjA[TA::PIECES];
#pragma unroll
  for (int i = 0; i < TA::PIECES; ++i) adjA[i] = (TA::piece_index(pw, i) - (pw + i * TA::NWAVES)) * 1024;
  auto issue_piece = [&](int q) {
    if (q < TA::PIECES) glds16m(kA, offA[q < TA::PIECES ? q : 0], islot_off + (unsigned)(q * TA::NWAVES * 1024 + (TA::WRAP ? adjA[q < TA::PIECES ? q : 0] : 0)));
    else glds16m(kB, offB[q >= TA::PIECES ? q - TA::PIECES : 0], islot_off + (unsigned)(A_BYTES + (q - TA::PIECES) * TB::NWAVES * 1024));
  };
  auto issue_advance = [&]() {
    islot = islot == NSLOT - 1 ? 0 : islot + 1;
    islot_off = islot == 0 ? lds0 : islot_off + (unsigned)STAGE;
    kA += strideA;
    kB += strideB;
    ++issued;
  };
  // The issuing waves are the FIRST four (dispatched first = oldest on their SIMDs: the issue arbiter serves them first).  Measured
  // alternatives, same box (profiles/r04_step_experiments.txt item 5): the last four instead - no gain over all eight; priority 1 for the
  // non-issuing waves - the gain is gone; all pieces of a K-tile up front instead of between the MFMA groups - half the gain.
  const bool dma_wave = DW == 8 || wave < DW;
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) {
    if (issued < nk) {
      if (dma_wave) {
#pragma unroll
        for (int q = 0; q < PIECES; ++q) issue_piece(q);
      }
      issue_advance();
    }
  }
  for (int t = 0; t < nk; ++t) {
    // this wave's pieces of K-tile t have landed (issued - t - 1 younger K-tiles may stay in flight)
    if (dma_wave) {
      if (issued - t == NSLOT - 1) vm_wait<(NSLOT - 2) * PIECES>();
      else vm_wait<0>();
    }
    __syncthreads();
    const bool did = issued < nk;
    const unsigned char* a_l = smem + cslot * STAGE;
    const unsigned char* b_l = a_l + A_BYTES;
    bf16x8_t af[MI], bf[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = TA::frag(a_l, wm * MI + i, wk, lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[j] = TB::frag(b_l, wn * NJ + j, wk, lane);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
      if (did && dma_wave) {
#pragma unroll
        for (int q = i * PIECES / MI; q < (i + 1) * PIECES / MI; ++q) issue_piece(q);
      }
    }
    if (did) issue_advance();
    cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
  }
  // ---- add the two K halves: wave w keeps its row blocks (see H0 / H1) of the sub-tile and gives the others to wave w ^ 4
  __syncthreads();
  float4* xch = reinterpret_cast<float4*>(smem);
  const int partner = wave ^ 4;
  if (wk == 0) {
#pragma unroll
    for (int ii = 0; ii < H1; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4_t v = acc[H0 + ii][j];
        xch[(size_t)(wave * HT + ii * NJ + j) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
      }
  } else {
#pragma unroll
    for (int ii = 0; ii < H0; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4_t v = acc[ii][j];
        xch[(size_t)(wave * HT + ii * NJ + j) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
      }
  }
  __syncthreads();
  // (weight gradients: the sum of squares of the stored bf16 values of this tile, for the gradient norm - see GemmGroup::sq_partials;
  //  the tiles of these launches are whole - M, N multiples of the tile - so every accumulator is a stored value)
  constexpr bool SQ = A_MC && B_MC && EPI == GGET_EPI_NONE;
  float sq = 0.f;
  auto sq_add = [&](const f32x4_t& v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float r = bf2f(f2bf(v[e])); sq = fmaf(r, r, sq); }
  };
  if (wk == 0) {
    f32x4_t own[H0][NJ];
#pragma unroll
    for (int ii = 0; ii < H0; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float4 o = xch[(size_t)(partner * HT + ii * NJ + j) * 64 + lane];
        const f32x4_t mine = acc[ii][j];
        own[ii][j] = f32x4_t{mine[0] + o.x, mine[1] + o.y, mine[2] + o.z, mine[3] + o.w};
        if constexpr (SQ) sq_add(own[ii][j]);
      }
    store_tile<EPI, H0, NJ>(own, P, P.M, (P.N + 3) & ~3, m0 + wm * (MI * 16), n0 + wn * (NJ * 16), lane);
  } else {
    f32x4_t own[H1][NJ];
#pragma unroll
    for (int ii = 0; ii < H1; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float4 o = xch[(size_t)(partner * HT + ii * NJ + j) * 64 + lane];
        const f32x4_t mine = acc[H0 + ii][j];
        own[ii][j] = f32x4_t{mine[0] + o.x, mine[1] + o.y, mine[2] + o.z, mine[3] + o.w};
        if constexpr (SQ) sq_add(own[ii][j]);
      }
    store_tile<EPI, H1, NJ>(own, P, P.M, (P.N + 3) & ~3, m0 + wm * (MI * 16) + H0 * 16, n0 + wn * (NJ * 16), lane);
  }
  if constexpr (SQ) {
    if (g.sq_partials) {   // (uniform over the block; fixed reduction order: lanes by DPP, waves 0..7)
      float* red = reinterpret_cast<float*>(smem + NSLOT * STAGE);   // 64 bytes behind the ring (launch_ks_cfg)
      sq = wave_sum(sq);
      if (lane == 0) red[wave] = sq;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w];
        g.sq_partials[tile] = t;
      }
    }
  }
}

// ---- the same in-block K split on v_mfma_f32_32x32x16_bf16 (round 4; weight gradients: both operands M/N-contiguous) -----------------
// A wave owns (BM/2) x (BN/2) of the block tile as 32x32 accumulator blocks (192x192: 3 x 3 blocks = 144 accumulator registers) and the
// two 16-deep k-steps [32 wk, 32 wk + 32) of every 64-deep K-tile: per k-step 3 + 3 fragments (two transposing LDS reads each - the
// window rotation of TileIO<192, MC> is conflict-free for this lane pattern as well: the 32 lanes of an LDS cycle read 2 windows x 4
// k-rows, (4 kr + window + rotation) mod 8 all distinct) feed 9 MFMAs of 32 cycles - the same LDS bytes per FLOP as the 16x16x32 form,
// half the matrix instructions, and the 32x32 shape issues back to back at the pipe's full rate (32 cycles for 32 K FLOP; 16x16x32
// measures ~17 for 16 K).  MEASURED (profiles/r04_step_experiments.txt item 1): same bits out, 15 % slower launch - kept as an
// experiment behind g_gemm_variant bit 6, not the default.  Fragment of block `sb` (32 rows), k-step `ks` (of the K-tile): lane l holds row l % 32, k = 8 (l / 32) .. + 7.
template <int ROWS>
__device__ __forceinline__ bf16x8_t frag32_mc(const unsigned char* lds, int sb, int ks, int lane) {
  using T = TileIO<ROWS, true, 512, 64>;
  const int l15 = lane & 15, g = lane >> 4;
  const int w = sb * 2 + (g & 1);
  const int kr0 = ks * 16 + (g >> 1) * 8 + (l15 >> 2), kr1 = kr0 + 4;
  const int inw = (l15 & 3) * 8;
  const unsigned char* p0 = lds + kr0 * T::ROWB + (T::mc_lds_win(kr0, w) << 5) + inw;
  const unsigned char* p1 = lds + kr1 * T::ROWB + (T::mc_lds_win(kr1, w) << 5) + inw;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p0));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p1));
  bf16x8_t out;
  out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
  out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
  return out;
}
// bf16 store of one 32x32 accumulator block (operands swapped: lane l holds row m = l % 32 and, in register quad q, the columns
// 8 q + 4 (l / 32) .. + 3): v_permlane32_swap between the two lanes of a row pairs the quads so that every lane owns 8 consecutive
// columns (16 bytes) of two 16-column groups
__device__ __forceinline__ void store32_block(const f32x16_t& a, bf16_t* C, int ldc, int M, int N, int mrow, int ncol, int lane) {
  const int m = mrow + (lane & 31);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * p + e]), __float_as_uint(a[8 * p + 4 + e]), false, false);
      v[e] = __uint_as_float(r[0]);
      v[4 + e] = __uint_as_float(r[1]);
    }
    const int n = ncol + 16 * p + 8 * (lane >> 5);
    if (m < M && n + 8 <= N) stc16(C + (size_t)m * ldc + n, pack8(v));
    else if (m < M && n < N) {
      const uint4 pk = pack8(v);
      *reinterpret_cast<uint2*>(C + (size_t)m * ldc + n) = make_uint2(pk.x, pk.y);   // (N % 8 == 4: the first half of the chunk)
    }
  }
}
template <int BM, int BN, int NSLOT_ = 0>
__global__ void __launch_bounds__(512, 2) gemm_ks32_kernel(const GemmGroup g, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BK = 64, NT = 512;
  using TA = TileIO<BM, true, NT, BK>;
  using TB = TileIO<BN, true, NT, BK>;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NSLOT = NSLOT_ > 0 ? NSLOT_ : persist_slots(STAGE);
  constexpr int PIECES = TA::PIECES + TB::PIECES;
  constexpr int MI = BM / 2 / 32, NJ = BN / 2 / 32, H0 = (MI + 1) / 2, H1 = MI - H0;
  static_assert(BM % 64 == 0 && BN % 64 == 0 && H1 >= 1, "wave tile = whole 32x32 blocks, at least two row blocks");
  static_assert(!TA::WRAP && !TB::WRAP, "piece lists");
  // accumulator exchange: the wk = 1 waves hand over H0 row blocks, the wk = 0 waves H1 (float4 units of 1 KiB per wave)
  constexpr int SEND1 = H0 * NJ * 4, SEND0 = H1 * NJ * 4;
  static_assert(4 * (SEND0 + SEND1) * 1024 <= NSLOT * STAGE, "accumulator exchange fits in the ring");

  const int G = gridDim.x;
  const int tile = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // XCD-contiguous
  if (tile >= total_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GGET_MAX_GROUP; ++i)
    if (i < g.count && tile >= g.p[i].tile_begin) pi = i;
  const GemmProblem& P = g.p[pi];
  int m0, n0;
  tile_origin(P, P.M, tile - P.tile_begin, BM, BN, g.super, m0, n0);
  const int nk = P.K >> 6;

  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(LDS_AS const void*)smem) + (unsigned)wave * 1024u;
  const unsigned char* kA = reinterpret_cast<const unsigned char*>(P.A);
  const unsigned char* kB = reinterpret_cast<const unsigned char*>(P.B);
  const long strideA = (long)BK * P.lda * 2, strideB = (long)BK * P.ldb * 2;
  unsigned offA[TA::PIECES], offB[TB::PIECES];
#pragma unroll
  for (int i = 0; i < TA::PIECES; ++i) offA[i] = TA::piece_off(P.lda, m0, P.M, wave, lane, i);
#pragma unroll
  for (int i = 0; i < TB::PIECES; ++i) offB[i] = TB::piece_off(P.ldb, n0, P.N, wave, lane, i);
  int islot = 0, cslot = 0, issued = 0;
  unsigned islot_off = lds0;
  auto issue_piece = [&](int q) {
    if (q < TA::PIECES) glds16m(kA, offA[q < TA::PIECES ? q : 0], islot_off + (unsigned)(q * TA::NWAVES * 1024));
    else glds16m(kB, offB[q >= TA::PIECES ? q - TA::PIECES : 0], islot_off + (unsigned)(A_BYTES + (q - TA::PIECES) * TB::NWAVES * 1024));
  };
  auto issue_advance = [&]() {
    islot = islot == NSLOT - 1 ? 0 : islot + 1;
    islot_off = islot == 0 ? lds0 : islot_off + (unsigned)STAGE;
    kA += strideA;
    kB += strideB;
    ++issued;
  };
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) {
    if (issued < nk) {
#pragma unroll
      for (int q = 0; q < PIECES; ++q) issue_piece(q);
      issue_advance();
    }
  }
  for (int t = 0; t < nk; ++t) {
    if (issued - t == NSLOT - 1) vm_wait<(NSLOT - 2) * PIECES>();
    else vm_wait<0>();
    __syncthreads();
    const bool did = issued < nk;
    const unsigned char* a_l = smem + cslot * STAGE;
    const unsigned char* b_l = a_l + A_BYTES;
    bf16x8_t af[2][MI], bf[2][NJ];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[s][i] = frag32_mc<BM>(a_l, wm * MI + i, wk * 2 + s, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[s][j] = frag32_mc<BN>(b_l, wn * NJ + j, wk * 2 + s, lane);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
        if (did) {
          constexpr int NG = 2 * MI;
          const int grp = s * MI + i;
#pragma unroll
          for (int q = grp * PIECES / NG; q < (grp + 1) * PIECES / NG; ++q) issue_piece(q);
        }
      }
    if (did) issue_advance();
    cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
  }
  // ---- add the two K halves: wave (wk = 0) keeps row blocks [0, H0), its partner (wave ^ 4) the blocks [H0, MI)
  __syncthreads();
  float4* xch = reinterpret_cast<float4*>(smem);
  // 1 KiB units: the wk = 1 waves' sends first (SEND1 units each), then the wk = 0 waves' (SEND0 each)
  auto unit_of = [&](int w, int u) { return (w >= 4 ? (w - 4) * SEND1 : 4 * SEND1 + w * SEND0) + u; };
  if (wk == 0) {
#pragma unroll
    for (int ii = 0; ii < H1; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16_t& v = acc[H0 + ii][j];
          xch[(size_t)unit_of(wave, (ii * NJ + j) * 4 + q) * 64 + lane] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
  } else {
#pragma unroll
    for (int ii = 0; ii < H0; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16_t& v = acc[ii][j];
          xch[(size_t)unit_of(wave, (ii * NJ + j) * 4 + q) * 64 + lane] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
  }
  __syncthreads();
  const int partner = wave ^ 4;
  bf16_t* C = reinterpret_cast<bf16_t*>(P.C);
  const int Nst = (P.N + 3) & ~3;
  if (wk == 0) {
#pragma unroll
    for (int ii = 0; ii < H0; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x16_t own = acc[ii][j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 o = xch[(size_t)unit_of(partner, (ii * NJ + j) * 4 + q) * 64 + lane];
          own[4 * q] += o.x; own[4 * q + 1] += o.y; own[4 * q + 2] += o.z; own[4 * q + 3] += o.w;
        }
        store32_block(own, C, P.ldc, P.M, Nst, m0 + wm * (MI * 32) + ii * 32, n0 + wn * (NJ * 32) + j * 32, lane);
      }
  } else {
#pragma unroll
    for (int ii = 0; ii < H1; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x16_t own = acc[H0 + ii][j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 o = xch[(size_t)unit_of(partner, (ii * NJ + j) * 4 + q) * 64 + lane];
          own[4 * q] += o.x; own[4 * q + 1] += o.y; own[4 * q + 2] += o.z; own[4 * q + 3] += o.w;
        }
        store32_block(own, C, P.ldc, P.M, Nst, m0 + wm * (MI * 32) + (H0 + ii) * 32, n0 + wn * (NJ * 32) + j * 32, lane);
      }
  }
}

template <int BM, int BN, int NSLOT_ = 0>
int launch_ks32_cfg(GemmGroup& g, int total, hipStream_t st) {
  constexpr int STG = (BM + BN) * 64 * 2;
  constexpr int SM = (NSLOT_ > 0 ? NSLOT_ : persist_slots(STG)) * STG;
  const int G = (total + 7) & ~7;
  static bool attr0 = false;
  if (!attr0) {
    GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ks32_kernel<BM, BN, NSLOT_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SM));
    attr0 = true;
  }
  hipLaunchKernelGGL((gemm_ks32_kernel<BM, BN, NSLOT_>), dim3(G), dim3(512), SM, st, g, total);
  GGET_LAUNCH_CHECK();
  return 0;
}

template <int BM, int BN, bool A_MC, bool B_MC, int EPI, int NSLOT_ = 0, int DW = 8>
int launch_ks_cfg(GemmGroup& g, int total, hipStream_t st) {
  if constexpr (DW == 8 && BM != 64) {
    if (!(g_gemm_variant & 128)) return launch_ks_cfg<BM, BN, A_MC, B_MC, EPI, NSLOT_, 4>(g, total, st);
  }
  constexpr int STG = (BM + BN) * 64 * 2;
  constexpr int SM = (NSLOT_ > 0 ? NSLOT_ : persist_slots(STG)) * STG + (A_MC && B_MC && EPI == GGET_EPI_NONE ? 64 : 0);   // (+ the norm partials' scratch)
  static_assert(SM <= 160 * 1024, "LDS ring");
  const int G = (total + 7) & ~7;
  static bool attr0 = false;
  if (!attr0) {
    GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ks_kernel<BM, BN, A_MC, B_MC, EPI, NSLOT_, DW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SM));
    attr0 = true;
  }
  hipLaunchKernelGGL((gemm_ks_kernel<BM, BN, A_MC, B_MC, EPI, NSLOT_, DW>), dim3(G), dim3(512), SM, st, g, total);
  GGET_LAUNCH_CHECK();
  return 0;
}

// launch one persistent configuration: one block per CU (grid rounded to the 8 XCDs)
template <int BM, int BN, int BK, int WM, int WN, bool A_MC, bool B_MC, int EPI, int NSLOT_ = 0, bool SK = false, int MODE = 0>
int launch_persist_cfg(GemmGroup& g, int total, int num_cu, hipStream_t st) {
  constexpr int STG = (BM + BN) * BK * 2;
  constexpr int SM = (NSLOT_ > 0 ? NSLOT_ : persist_slots(STG)) * STG;
  if (MODE == 2) num_cu *= 2;
  static_assert(SM <= 160 * 1024, "LDS ring");
  static_assert(!SK || (size_t)BM * BN * 4 <= kStreamKSlotBytes, "stream-K slot");
  if constexpr (MODE == 2) {
    static unsigned* cu_slots = nullptr;   // 8 XCC x 256 hardware CU ids, zeroed once (the counters only ever count up)
    if (!cu_slots && g_gemm_stagger_ticks > 0) {
      GGET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&cu_slots), 2048 * sizeof(unsigned)));
      GGET_HIP_CHECK(hipMemset(cu_slots, 0, 2048 * sizeof(unsigned)));
    }
    g.cu_slots = cu_slots;
    g.stagger_ticks = cu_slots ? g_gemm_stagger_ticks : 0;
  }
  int G = total < num_cu && !SK ? total : num_cu;
  G = (G + 7) & ~7;  // the XCD permutation needs a multiple of 8 (idle blocks exit at once)
  static bool attr0 = false;
  if (!attr0) {
    GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_persist_kernel<BM, BN, BK, WM, WN, A_MC, B_MC, EPI, NSLOT_, SK, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SM));
    attr0 = true;
  }
  hipLaunchKernelGGL((gemm_persist_kernel<BM, BN, BK, WM, WN, A_MC, B_MC, EPI, NSLOT_, SK, MODE>), dim3(G), dim3(WM * WN * 64), SM, st, g, total);
  GGET_LAUNCH_CHECK();
  return 0;
}

// The last round's split pays when it shortens the launch by more than the hand-over of the partial tiles costs (one write and one
// read of BM x BN fp32 per split tile: a few K-tiles' worth).  Fills g.sk_* and returns true when the launch should run split.
thread_local void* t_streamk_ws = nullptr;
unsigned g_streamk_epoch = 0;
inline bool streamk_plan(GemmGroup& g, int tiles, int num_cu, int bk) {
  if (!g_gemm_split_last || !t_streamk_ws || g.count != 1 || (g_gemm_variant & 8)) return false;
  const GemmProblem& p = g.p[0];
  if (p.m_dev || p.k_dev || num_cu > 256 || num_cu % 8 != 0) return false;
  const int nk = p.K / bk, G = num_cu;
  const int R = tiles / G, rem = tiles % G;
  if (rem == 0 || nk < 8) return false;
  int parts = 1, a = nk, last;
  if (rem <= G / 2) {
    parts = G / rem < 4 ? G / rem : 4;
    last = (nk + parts - 1) / parts;
  } else {
    const int H = G - rem, q = (rem + H - 1) / H;
    a = (nk * q + q) / (q + 1);            // ceil(nk q / (q + 1))
    if (a >= nk) return false;
    last = a > q * (nk - a) ? a : q * (nk - a);
  }
  const long plain = (long)(R + 1) * nk, split = (long)R * nk + last + 4;
  if (split > plain - plain / 16) return false;
  g.sk_rounds = R; g.sk_rem = rem; g.sk_parts = parts; g.sk_a = a;
  g.sk_flags = static_cast<unsigned*>(t_streamk_ws);
  g.sk_partial = reinterpret_cast<float*>(static_cast<unsigned char*>(t_streamk_ws) + kStreamKFlagBytes);
  g.sk_epoch = ++g_streamk_epoch ? g_streamk_epoch : ++g_streamk_epoch;   // never 0 (the flags start zeroed)
  return true;
}

template <int WM, int WN, bool A_MC, bool B_MC, int EPI>
int launch_t(GemmGroup& g, int split_k, hipStream_t st) {
  constexpr int BM = WM * 64, BN = WN * 64;
  int total = 0;
  for (int i = 0; i < g.count; ++i) {
    GemmProblem& p = g.p[i];
    const int tm = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.tile_begin = total;
    total += tm * p.tiles_n;
  }
  if (total == 0) return 0;
  constexpr int SMEM = kStages * (BM + BN) * 128;
  bool persist = split_k <= 1 && EPI != GGET_EPI_ATOMIC_F32 && EPI != GGET_EPI_SLAB_F32 && (!g.ablate || g.ablate >= 8) && getenv("GGET_GEMM_NO_PERSIST") == nullptr;
  for (int i = 0; i < g.count; ++i)
    persist = persist && (g.p[i].m_dev == nullptr || (g.count == 1 && !A_MC && getenv("GGET_GEMM_NO_DYN") == nullptr)) &&
              (g.p[i].k_dev == nullptr || (g.count == 1 && g.p[i].k_pad_zero && getenv("GGET_GEMM_NO_DYN") == nullptr)) &&
              (g.p[i].K % 64) == 0 && g.p[i].K >= 64;
  if (persist) {
    const int num_cu = gget_gemm_num_cu();
    // wide tile for the widest forward GEMM: 256x256x32 (8 waves of 128x64, 4-slot ring) moves 2/3 of the bytes per
    // FLOP of the 256x128 tile
    if constexpr (WM == 4 && WN == 2 && !A_MC && EPI != GGET_EPI_SLAB_F32 && EPI != GGET_EPI_ATOMIC_F32 && EPI != GGET_EPI_GEGLU_FWD) {
      long t256 = 0;
      bool ok256 = getenv("GGET_GEMM_NO_256") == nullptr;
      for (int i = 0; i < g.count; ++i) {
        t256 += (long)((g.p[i].M + 255) / 256) * ((g.p[i].N + 255) / 256);
        ok256 = ok256 && (g.p[i].N % 256) == 0;
      }
      // measured (profiles/r01_gemm_tiles.txt): pays only with >= ~2.5 tiles per CU (FFN gate|up), loses on N = 2304 / 3072
      // ... or, from 1.5 tiles per CU on, when its rounds x tile area come within 10 % of the default tile's (the 256x256 tile moves
      // 2/3 of the operand bytes per FLOP: the N = d GEMMs of a 41 472-row batch - ogbl-ppa fine-tune - are 486 tiles = two rounds
      // against 972 = four rounds of 256x128: dxn2 386 -> 343 us, down 216 -> 198, dxn1 143 -> 134, o 70 -> 64; tools/gemm_bench.py)
      const long r256 = (t256 + num_cu - 1) / num_cu, r_cur = (total + num_cu - 1) / num_cu;
      const bool near = 2 * t256 >= 3L * num_cu && r256 * 256 * 256 * 9 <= r_cur * BM * BN * 10;
      // Round 5 experiment (OFF by default; GGET_GEMM_ONE_ROUND=1 or g_gemm_variant bit 8 turns it on): ONE partial round of 256x256 tiles
      // instead of two rounds of the default tile when they are the same work per CU - the q|k|v projection on the var-len rows, 5696 x
      // 2304, is 207 tiles of 256x256 = 81 % of the CUs once, against 414 of 256x128 = two rounds, the second 62 % full.  Measured in the
      // C1 step, alternated in one process: +0.22 ms (7.000 -> 7.223 ms; profiles/r05_step_experiments.txt item 2) - a 256x256 tile with the
      // RoPE epilogue (128 accumulator registers per lane rotated and stored behind a 12-K-tile loop) costs more than the second round.
      static const int one_round_on = getenv("GGET_GEMM_ONE_ROUND") ? atoi(getenv("GGET_GEMM_ONE_ROUND")) : 0;
      const bool one_round = (one_round_on || (g_gemm_variant & 256)) && t256 <= num_cu && t256 * 10 >= (long)num_cu * 7 && r_cur >= 2 &&
                             (long)256 * 256 <= r_cur * BM * BN;
      if (ok256 && ((EPI != GGET_EPI_ROPE && (t256 >= (5 * num_cu) / 2 || near)) || one_round)) {
        int tot2 = 0;
        for (int i = 0; i < g.count; ++i) {
          GemmProblem& p = g.p[i];
          p.tiles_n = (p.N + 255) / 256;
          p.tile_begin = tot2;
          tot2 += ((p.M + 255) / 256) * p.tiles_n;
        }
        // row-major B (NT): 64-deep K-tiles through a 2-slot ring (2 x 64 KiB) - half as many barriers / waits / DMA
        // bookkeeping rounds per FLOP as 32-deep tiles through 4 slots (gate|up 91 -> 86 us); the NN form would spill
        if constexpr (!B_MC) {
          static const int bk64 = getenv("GGET_GEMM_BK64") ? atoi(getenv("GGET_GEMM_BK64")) : 1;
          if (bk64) return launch_persist_cfg<256, 256, 64, 2, 4, A_MC, B_MC, EPI, 2>(g, tot2, num_cu, st);
        }
        return launch_persist_cfg<256, 256, 32, 2, 4, A_MC, B_MC, EPI>(g, tot2, num_cu, st);
      }
    }
    // 128x192 tile for the N = d outputs (o/down projections and the dgrads into the residual stream): M/128 x N/192
    // tiles fill the chip where 256x128 leaves a quarter of the CUs idle (8192 x 768: 256 tiles vs 192)
    if constexpr (!A_MC && (EPI == GGET_EPI_NONE || EPI == GGET_EPI_RESIDUAL || EPI == GGET_EPI_ROPE)) {
      static int use192 = -1;
      if (use192 < 0) { const char* e = getenv("GGET_GEMM_192"); use192 = e ? atoi(e) : 1; }
      bool ok = use192 != 0;
      long t192 = 0;
      for (int i = 0; i < g.count; ++i) {
        // a ragged last column of tiles only for row-major B (NT: its DMA clamps the B rows) and plain stores
        ok = ok && ((g.p[i].N % 192) == 0 || (!B_MC && EPI != GGET_EPI_ROPE && g.p[i].N > 384));
        t192 += (long)((g.p[i].M + 127) / 128) * ((g.p[i].N + 191) / 192);
      }
      // rounds x tile area: take 128x192 when it needs less per-CU work than the default tile
      const long cur_rounds = (total + num_cu - 1) / num_cu, r192 = (t192 + num_cu - 1) / num_cu;
      // ... or when the K-split arrangement below fits ONE round at no more than 1.3x the default tiling's per-CU work: the default tiling
      // of an N = d launch happens to fit one round of 128x128 tiles from M <= 5 376 down (42 x 6 = 252 tiles), and the area rule then
      // kept it although it runs 66 us where the 96x192 K-split tiles run 47 (dxn2 at M = 5 376, profiles/r06_step_experiments.txt item 2:
      // a batch a few dozen tokens shorter than the headline's 5 696 rows cost +0.35 ms per step).  Only for launches that fill a good
      // part of the chip either way (>= 30 % of the CUs; the batch-size sweep of the same file: B = 128 -> T ~ 2 880 ran 5.71 ms against
      // 5.40 at B = 160 while the guard stood at 60 %.  Down to 1/8 of the CUs B = 64 gains as well, 5.11 -> 4.51 ms, but launches that
      // small are the tiny-batch fixtures of the parity suite, whose few-row loss statistics sit within a rounding flip of their bounds:
      // small launches keep the tiling they were validated with).
      bool ks_first = false;
      if constexpr (EPI != GGET_EPI_ROPE) {
        if (ok && !(g_gemm_variant & 1) && cur_rounds == 1 && total * 10 >= (long)num_cu * 3 && !(g_gemm_variant & 512)) {
          bool can = true;
          for (int i = 0; i < g.count; ++i) can = can && !g.p[i].m_dev && !g.p[i].k_dev && (g.p[i].N % 192) == 0 && g.p[i].K >= 1536;
          if (can)
            for (int cand : {64, 96, 128}) {
              long t = 0;
              for (int i = 0; i < g.count; ++i) t += (long)((g.p[i].M + cand - 1) / cand) * (g.p[i].N / 192);
              if (t <= num_cu) { ks_first = (long)cand * 192 * 10 <= (long)BM * BN * 13; break; }
            }
        }
      }
      if (ok && (r192 * 128 * 192 < cur_rounds * BM * BN || ks_first)) {
        int tot3 = 0;
        for (int i = 0; i < g.count; ++i) {
          GemmProblem& p = g.p[i];
          p.tiles_n = (p.N + 191) / 192;
          p.tile_begin = tot3;
          tot3 += ((p.M + 127) / 128) * p.tiles_n;
        }
        if constexpr (EPI != GGET_EPI_ROPE) {
          // one tile per CU, nothing device-sized: the K-split arrangement of the tile (g_gemm_variant bit 0 turns it off)
          bool ks = tot3 <= num_cu && !(g_gemm_variant & 1);
          // (K >= 1536: with only 12 K-tiles the accumulator exchange costs what the lighter fragment traffic saves - measured again in
          //  round 3 with the 96-row tiles: K = 768 launches through this kernel left the C1 step unchanged, 7.44-7.48 ms either way)
          for (int i = 0; i < g.count; ++i)
            ks = ks && !g.p[i].m_dev && !g.p[i].k_dev && (g.p[i].N % 192) == 0 && g.p[i].K >= 1536;
          if (ks) {
            // Rows per tile: the smallest of 64 / 96 / 128 whose tiles still fit in ONE round - more, smaller tiles keep more CUs
            // busy for a shorter time (var-len token layout: 5 696 x 768 is 180 tiles of 128 rows = 70 % of the CUs, 240 tiles of 96
            // rows = 94 % at 3/4 of the work each; the padded 8 192 x 768 stays at 256 tiles of 128 rows).  g_gemm_variant bit 4:
            // 128 rows only.
            int bm = 128;
            if (!(g_gemm_variant & 16))
              for (int cand : {64, 96}) {
                long t = 0;
                for (int i = 0; i < g.count; ++i) t += (long)((g.p[i].M + cand - 1) / cand) * (g.p[i].N / 192);
                if (t <= num_cu) { bm = cand; break; }
              }
            int tot = 0;
            for (int i = 0; i < g.count; ++i) {
              GemmProblem& p = g.p[i];
              p.tiles_n = p.N / 192;
              p.tile_begin = tot;
              tot += ((p.M + bm - 1) / bm) * p.tiles_n;
            }
            // g_gemm_lds_headroom (gget_debug_set key 2 / GGET_GEMM_LDS_HEADROOM): 3 ring slots instead of 4, so that a
            // collective's workgroup (a few KiB of LDS) can share the CU with the GEMM block
            if (bm == 64) return launch_ks_cfg<64, 192, A_MC, B_MC, EPI, 3>(g, tot, st);
            if (bm == 96) return launch_ks_cfg<96, 192, A_MC, B_MC, EPI, 3>(g, tot, st);
            if (g_gemm_lds_headroom) return launch_ks_cfg<128, 192, A_MC, B_MC, EPI, 3>(g, tot, st);
            return launch_ks_cfg<128, 192, A_MC, B_MC, EPI>(g, tot, st);
          }
        }
        if constexpr (EPI != GGET_EPI_ROPE) {
          if (streamk_plan(g, tot3, num_cu, 64)) {
            return launch_persist_cfg<128, 192, 64, 4, 2, A_MC, B_MC, EPI, 3, true>(g, tot3, num_cu, st);
          }
        }
        if (g_gemm_lds_headroom) return launch_persist_cfg<128, 192, 64, 4, 2, A_MC, B_MC, EPI, 3>(g, tot3, num_cu, st);
        return launch_persist_cfg<128, 192, 64, 4, 2, A_MC, B_MC, EPI>(g, tot3, num_cu, st);
      }
    }
    // 192x192 tile for the weight gradients (TN, K = T): for d = 768 the four wgrads of a decoder layer
    // (gate|up 6144x768, down 768x3072, q|k|v 2304x768, o 768x768) are exactly 256 such tiles - one per CU, full K,
    // no split-K slabs.  Taken whenever it needs less per-CU work than the 256x128 tiling.
    if constexpr (A_MC && B_MC && EPI == GGET_EPI_NONE) {
      static int use192 = -1;
      if (use192 < 0) { const char* e = getenv("GGET_GEMM_192"); use192 = e ? atoi(e) : 1; }
      bool ok = use192 != 0;
      long t192 = 0;
      for (int i = 0; i < g.count; ++i) {
        ok = ok && (g.p[i].N % 192) == 0 && (g.p[i].M % 192) == 0;
        t192 += (long)(g.p[i].M / 192) * (g.p[i].N / 192);
      }
      const long cur_rounds = (total + num_cu - 1) / num_cu, r192 = (t192 + num_cu - 1) / num_cu;
      if (ok && r192 * 192 * 192 < cur_rounds * BM * BN) {
        int tot4 = 0;
        for (int i = 0; i < g.count; ++i) {
          GemmProblem& p = g.p[i];
          p.tiles_n = p.N / 192;
          p.tile_begin = tot4;
          tot4 += (p.M / 192) * p.tiles_n;
        }
        if (tot4 <= num_cu && !(g_gemm_variant & 2)) {
          bool ks = true;
          for (int i = 0; i < g.count; ++i) ks = ks && !g.p[i].m_dev && !g.p[i].k_dev;
          // g_gemm_variant bit 6: the 32x32x16 MFMA form (gemm_ks32_kernel, round 4) - bit-identical results, measured SLOWER in the step
          // (7.285 against 7.119 ms, same box, alternated: profiles/r04_step_experiments.txt item 1), so the 16x16x32 form stays the default
          if (ks && (g_gemm_variant & 64)) return launch_ks32_cfg<192, 192>(g, tot4, st);
          if (ks) {
            g.sq_written = g.sq_partials != nullptr;
            return launch_ks_cfg<192, 192, true, true, EPI>(g, tot4, st);
          }
        }
        return launch_persist_cfg<192, 192, 64, 4, 2, true, true, EPI>(g, tot4, num_cu, st);
      }
    }
    // 192-row variant of the 256x128 tile (waves 4 x 2, 48 x 64 each): a row count that is no multiple of 256 x (CUs / column tiles)
    // - the var-len token layout: M = real tokens, e.g. 5 696 instead of 8 192 - leaves the last round of 256-row tiles mostly
    // empty; rounds x rows decides (8192 x 3072: 3 x 256 against 5 x 192 -> 256; 5696 x 3072: 3 x 256 against 3 x 192 -> 192)
    if constexpr (WM == 4 && WN == 2 && !A_MC && (EPI == GGET_EPI_GEGLU_BWD || EPI == GGET_EPI_NONE || EPI == GGET_EPI_RESIDUAL)) {
      bool ok = !(g_gemm_variant & 4);
      long t192 = 0;
      for (int i = 0; i < g.count; ++i) {
        ok = ok && !g.p[i].m_dev && !g.p[i].k_dev && (g.p[i].N % BN) == 0;
        t192 += (long)((g.p[i].M + 191) / 192) * (g.p[i].N / BN);
      }
      const long cur_rounds = (total + num_cu - 1) / num_cu, r192 = (t192 + num_cu - 1) / num_cu;
      // (a 192-row tile moves 8 % more operand bytes per FLOP than the 256-row one: it has to save more than that in rounds x rows -
      //  gate|up at T = 41 472: 21 x 192 against 16 x 256 is 1.5 % less work and ran 463 us against 432; tools/gemm_gu_ab.py)
      if (ok && r192 * 192 * 100 < cur_rounds * 256 * 92) {
        int tot5 = 0;
        for (int i = 0; i < g.count; ++i) {
          GemmProblem& p = g.p[i];
          p.tiles_n = p.N / BN;
          p.tile_begin = tot5;
          tot5 += ((p.M + 191) / 192) * p.tiles_n;
        }
        // the GEGLU' launch: two blocks per CU (kernel comment, MODE 2); g_gemm_variant bit 5: one block per CU as everywhere else
        // (g_gemm_lds_headroom 2 = a collective's kernel shares the chip, data-parallel runs: the two blocks fill a CU's LDS and a foreign
        //  workgroup on a CU would push one of them into a second round - one block per CU with a 3-slot ring then)
        if constexpr (EPI == GGET_EPI_GEGLU_BWD && BN == 128) {
          if (!(g_gemm_variant & 32) && g_gemm_lds_headroom < 2) return launch_persist_cfg<192, BN, 64, WM, WN, A_MC, B_MC, EPI, 2, false, 2>(g, tot5, num_cu, st);
        }
        if (g_gemm_lds_headroom) return launch_persist_cfg<192, BN, 64, WM, WN, A_MC, B_MC, EPI, 3>(g, tot5, num_cu, st);
        return launch_persist_cfg<192, BN, 64, WM, WN, A_MC, B_MC, EPI>(g, tot5, num_cu, st);
      }
    }
    return launch_persist_cfg<BM, BN, 64, WM, WN, A_MC, B_MC, EPI>(g, total, num_cu, st);
  }
  static bool attr_done = false;
  if (!attr_done) {
    GGET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<WM, WN, A_MC, B_MC, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  dim3 grid(total, split_k > 0 ? split_k : 1, 1);
  hipLaunchKernelGGL((gemm_kernel<WM, WN, A_MC, B_MC, EPI>), grid, dim3(WM * WN * 64), SMEM, st, g);
  GGET_LAUNCH_CHECK();
  return 0;
}

// Tile choice: 256x128 (8 waves, 1 block/CU, 0.75x the LDS-fill traffic per FLOP of 128x128) when it still yields
// at least ~1.5 tiles per CU; 128x128 (4 waves) otherwise, so the narrow N=d GEMMs keep every CU busy.
template <bool A_MC, bool B_MC, int EPI>
int launch_shape(GemmGroup& g, int split_k, hipStream_t st) {
  long tiles_big = 0;
  for (int i = 0; i < g.count; ++i) tiles_big += (long)((g.p[i].M + 255) / 256) * ((g.p[i].N + 127) / 128);
  tiles_big *= (split_k > 1 ? split_k : 1);
  if (tiles_big >= 160) return launch_t<4, 2, A_MC, B_MC, EPI>(g, split_k, st);
  return launch_t<2, 2, A_MC, B_MC, EPI>(g, split_k, st);
}

template <bool A_MC, bool B_MC>
int launch_mode(GemmGroup& g, int epi, int split_k, hipStream_t st) {
  switch (epi) {
    case GGET_EPI_NONE: return launch_shape<A_MC, B_MC, GGET_EPI_NONE>(g, split_k, st);
    case GGET_EPI_RESIDUAL: return launch_shape<A_MC, B_MC, GGET_EPI_RESIDUAL>(g, split_k, st);
    case GGET_EPI_ATOMIC_F32: return launch_shape<A_MC, B_MC, GGET_EPI_ATOMIC_F32>(g, split_k, st);
    case GGET_EPI_SLAB_F32: return launch_shape<A_MC, B_MC, GGET_EPI_SLAB_F32>(g, split_k, st);
    case GGET_EPI_ROPE:
      if constexpr (!A_MC && !B_MC) return launch_shape<false, false, GGET_EPI_ROPE>(g, split_k, st);
      break;
    case GGET_EPI_GEGLU_BWD:
      if constexpr (!A_MC && B_MC) return launch_shape<false, true, GGET_EPI_GEGLU_BWD>(g, split_k, st);
      break;
    case GGET_EPI_GEGLU_FWD:
      if constexpr (!A_MC && !B_MC) {
        // one configuration: 256x256x64 tiles (128 gate + 128 up columns), 8 waves as 4 (M) x 2 (N), each wave 64 rows x
        // (64 gate + 64 up) columns, 2-slot ring
        int total = 0;
        for (int i = 0; i < g.count; ++i) {
          GemmProblem& p = g.p[i];
          p.tiles_n = p.N / 256;
          p.tile_begin = total;
          total += ((p.M + 255) / 256) * p.tiles_n;
        }
        if (total == 0) return 0;
        const int num_cu = gget_gemm_num_cu();
        // (192-row tiles when they need fewer rounds x rows than 256-row tiles - see launch_t)
        long t192 = 0;
        for (int i = 0; i < g.count; ++i) t192 += (long)((g.p[i].M + 191) / 192) * (g.p[i].N / 256);
        if (!(g_gemm_variant & 4) && ((t192 + num_cu - 1) / num_cu) * 192 * 100 < ((total + num_cu - 1) / num_cu) * 256 * 92) {   // (margin: see launch_t)
          int tot2 = 0;
          for (int i = 0; i < g.count; ++i) {
            GemmProblem& p = g.p[i];
            p.tile_begin = tot2;
            tot2 += ((p.M + 191) / 192) * p.tiles_n;
          }
          return launch_persist_cfg<192, 256, 64, 4, 2, false, false, GGET_EPI_GEGLU_FWD, 2>(g, tot2, num_cu, st);
        }
        return launch_persist_cfg<256, 256, 64, 4, 2, false, false, GGET_EPI_GEGLU_FWD, 2>(g, total, num_cu, st);
      }
      break;
  }
  gget_set_error("gemm: unknown epilogue %d", epi);
  return 2;
}

}  // namespace

void gget_gemm_streamk_workspace(void* ws) { t_streamk_ws = ws; }

struct GemmProbeRec { hipEvent_t e0, e1; double flops; bool dyn; };
static std::deque<GemmProbeRec> g_probe;   // (deque: records keep their address while launches are appended)
static bool g_probe_on = false;

int gget_gemm_launch(int mode, int epi, GemmGroup& g, int split_k, hipStream_t st) {
  GGET_REQUIRE(g.count >= 1 && g.count <= GGET_MAX_GROUP, "gemm: bad group size %d", g.count);
  g.sk_partial = nullptr; g.sk_flags = nullptr; g.sk_epoch = 0;
  g.sk_rounds = g.sk_rem = g.sk_a = 0; g.sk_parts = 1;
  g.cu_slots = nullptr; g.stagger_ticks = 0;
  g.sq_written = 0;
  static int ablate = -1;
  if (ablate < 0) {
    const char* e = getenv("GGET_GEMM_ABLATE");
    ablate = e ? atoi(e) : 0;
    if (const char* v = getenv("GGET_GEMM_VARIANT")) g_gemm_variant = atoi(v);
    if (const char* v = getenv("GGET_GEMM_LDS_HEADROOM")) g_gemm_lds_headroom = atoi(v);
    if (const char* v = getenv("GGET_GEMM_SPLIT_LAST")) g_gemm_split_last = atoi(v);   // same knob as gget_debug_set(1, .), for whole-step A/B
    if (const char* v = getenv("GGET_GEMM_STAGGER")) g_gemm_stagger_ticks = atoi(v);
  }
  g.ablate = g_gemm_ablate_set >= 0 ? g_gemm_ablate_set : ablate;
  static int super = -1;
  if (super < 0) { const char* e = getenv("GGET_GEMM_SUPER"); super = e ? atoi(e) : 0; }
  g.super = super > 0 ? super : (mode == GGET_GEMM_TN ? 1 : kSuper);
  GGET_REQUIRE(split_k <= 1 || epi == GGET_EPI_ATOMIC_F32 || epi == GGET_EPI_SLAB_F32,
               "gemm: split-K needs the fp32 atomic or slab epilogue");
  for (int i = 0; i < g.count; ++i) {
    const GemmProblem& p = g.p[i];
    GGET_REQUIRE(p.c_rows == nullptr || (epi == GGET_EPI_NONE && split_k <= 1), "gemm: the fused row scatter needs the plain bf16 epilogue");
    if (epi == GGET_EPI_GEGLU_FWD || epi == GGET_EPI_GEGLU_BWD) {
      const bool fwd = epi == GGET_EPI_GEGLU_FWD;
      GGET_REQUIRE(mode == (fwd ? GGET_GEMM_NT : GGET_GEMM_NN) && split_k <= 1 && !p.m_dev && !p.k_dev, "gemm: GEGLU epilogue: wrong mode");
      GGET_REQUIRE(p.ff > 0 && p.N == (fwd ? 2 * p.ff : p.ff) && p.N % (fwd ? 256 : 64) == 0 && (p.K % 64) == 0,
                   "gemm: GEGLU epilogue needs ff %% %d == 0 and K %% 64 == 0 (ff %d N %d K %d)", fwd ? 128 : 64, p.ff, p.N, p.K);
      GGET_REQUIRE((p.ldc % 64) == 0 && ((uintptr_t)p.C & 127) == 0 && p.ldc >= 2 * p.ff, "gemm: GEGLU epilogue: gu / dgu must be [M][>= 2 ff], 128-byte aligned rows");
      if (fwd) GGET_REQUIRE(p.C2 && (p.ldc2 % 64) == 0 && ((uintptr_t)p.C2 & 127) == 0 && p.ldc2 >= p.ff, "gemm: GEGLU forward: h must be [M][>= ff], 128-byte aligned rows");
      else GGET_REQUIRE(p.G && (p.ldg % 8) == 0 && ((uintptr_t)p.G & 15) == 0 && p.ldg >= 2 * p.ff, "gemm: GEGLU backward: gu must be [M][>= 2 ff]");
    }
    GGET_REQUIRE((p.lda % 8) == 0 && (p.ldb % 8) == 0 && (p.ldc % 8) == 0,
                 "gemm: leading dims must be multiples of 8 (lda %d ldb %d ldc %d)", p.lda, p.ldb, p.ldc);
    // N % 4 != 0 (e.g. the 41 245-entry ogbl-ppa vocabulary): C is written up to the next multiple of 4, which must fit in
    // ldc; only the forward layout (B = [N,K], rows clamped) supports it
    GGET_REQUIRE((p.N % 4) == 0 || (mode == GGET_GEMM_NT && p.ldc >= ((p.N + 3) & ~3) && epi != GGET_EPI_RESIDUAL),
                 "gemm: N = %d must be a multiple of 4 (or NT mode with ldc >= N rounded up to 4)", p.N);
    // the LDS-DMA addresses are a uniform 64-bit K-origin plus an unsigned 32-bit per-lane byte offset
    GGET_REQUIRE(((size_t)p.M + 64) * (size_t)p.lda * 2 < (1ull << 32) && ((size_t)p.N + 64) * (size_t)p.ldb * 2 < (1ull << 32),
                 "gemm: operand spans more than 4 GiB (M %d lda %d N %d ldb %d)", p.M, p.lda, p.N, p.ldb);
  }
  // measurement aid (gget_debug_gemm_probe): bracket the launch with events and keep its algorithmic FLOPs.  Launches whose row or
  // K count lives on the device (the SMTP head) have no host-side FLOP count: counted separately, left out of both sums.
  GemmProbeRec* rec = nullptr;
  if (g_probe_on) {
    double fl = 0;
    bool dyn = false;
    for (int i = 0; i < g.count; ++i) {
      fl += 2.0 * g.p[i].M * g.p[i].N * g.p[i].K;
      dyn = dyn || g.p[i].m_dev || g.p[i].k_dev;
    }
    g_probe.emplace_back();
    rec = &g_probe.back();
    rec->flops = fl;
    rec->dyn = dyn;
    GGET_HIP_CHECK(hipEventCreate(&rec->e0));
    GGET_HIP_CHECK(hipEventCreate(&rec->e1));
    GGET_HIP_CHECK(hipEventRecord(rec->e0, st));
  }
  int rc = 2;
  if (g.count == 1 && (g.p[0].b_tile_off || g.p[0].a_rows)) {
    // slot-sorted head GEMM: a fixed configuration (128-row tiles = the granule the caller padded its slots to)
    GemmProblem& p = g.p[0];
    GGET_REQUIRE((mode == GGET_GEMM_NT || mode == GGET_GEMM_NN) && epi == GGET_EPI_NONE && split_k <= 1 && p.m_dev && !p.k_dev &&
                 (p.N % 192) == 0 && (p.K % 64) == 0 && p.K >= 64,
                 "gemm: a_rows / b_tile_off need a single NT / NN problem with the plain epilogue, a device-side row count, N %% 192 == 0 and K %% 64 == 0");
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      GGET_HIP_CHECK(hipGetDevice(&dev));
      GGET_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const bool big = p.b_tile_rows == 256 && (p.N % 256) == 0;
    p.tiles_n = big ? p.N / 256 : p.N / 192;
    p.tile_begin = 0;
    const int total = big ? ((p.M + 255) / 256) * p.tiles_n : ((p.M + 127) / 128) * p.tiles_n;
    if (total <= 0) rc = 0;
    else if (big && mode == GGET_GEMM_NT) rc = launch_persist_cfg<256, 256, 64, 2, 4, false, false, GGET_EPI_NONE, 2>(g, total, ncu, st);
    else if (big) rc = launch_persist_cfg<256, 256, 32, 2, 4, false, true, GGET_EPI_NONE>(g, total, ncu, st);
    else if (mode == GGET_GEMM_NT) rc = launch_persist_cfg<128, 192, 64, 4, 2, false, false, GGET_EPI_NONE, 3>(g, total, ncu, st);
    else rc = launch_persist_cfg<128, 192, 64, 4, 2, false, true, GGET_EPI_NONE, 3>(g, total, ncu, st);
    if (rec) GGET_HIP_CHECK(hipEventRecord(rec->e1, st));
    return rc;
  }
  switch (mode) {
    case GGET_GEMM_NT: rc = launch_mode<false, false>(g, epi, split_k, st); break;
    case GGET_GEMM_NN: rc = launch_mode<false, true>(g, epi, split_k, st); break;
    case GGET_GEMM_TN: rc = launch_mode<true, true>(g, epi, split_k, st); break;
    default: gget_set_error("gemm: unknown mode %d", mode);
  }
  if (rec) GGET_HIP_CHECK(hipEventRecord(rec->e1, st));
  return rc;
}

// Measurement aid behind bench.py's time-weighted GEMM figure (no reference counterpart).  enable = 1: start recording every GEMM
// launch of this process (previous records dropped); enable = 0: stop, wait for the recorded launches and report the sum of their
// algorithmic FLOPs (2 M N K per problem), the sum of their durations (HIP events on the launch stream: includes the launch gaps
// the events themselves open, so the figure is a lower bound of the in-step rate), the number of launches summed and the number left
// out (device-sized row / K counts).
extern "C" int gget_debug_gemm_probe(int enable, double* flops_out, double* ms_out, int* launches_out, int* skipped_out) {
  if (enable) {
    for (auto& r : g_probe) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_probe.clear();
    g_probe_on = true;
    return 0;
  }
  g_probe_on = false;
  double fl = 0, ms = 0;
  int n = 0, skipped = 0;
  for (auto& r : g_probe) {
    GGET_HIP_CHECK(hipEventSynchronize(r.e1));
    float t = 0.f;
    GGET_HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
    if (r.dyn) { ++skipped; }
    else { fl += r.flops; ms += t; ++n; }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_probe.clear();
  if (flops_out) *flops_out = fl;
  if (ms_out) *ms_out = ms;
  if (launches_out) *launches_out = n;
  if (skipped_out) *skipped_out = skipped;
  return 0;
}

int gget_gemm_single(int mode, int epi, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                     int lda, int ldb, int ldc, const int* m_dev, const int* k_dev, int split_k, hipStream_t st, bool k_pad_zero,
                     const int* c_rows) {
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = 1;
  GemmProblem& p = g.p[0];
  p.A = static_cast<const bf16_t*>(A);
  p.B = static_cast<const bf16_t*>(B);
  p.C = C;
  p.R = static_cast<const bf16_t*>(R);
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.m_dev = m_dev; p.k_dev = k_dev;
  p.k_pad_zero = k_pad_zero ? 1 : 0;
  p.slab_stride = (long)M * ldc;  // EPI_SLAB_F32 through the op-level entry: dense [split_k][M][ldc] slabs
  p.c_rows = c_rows;
  return gget_gemm_launch(mode, epi, g, split_k, st);
}
