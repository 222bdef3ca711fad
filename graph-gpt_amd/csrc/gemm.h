// Host-side descriptors for the grouped bf16 MFMA GEMM (gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gget.h"
#include "common.h"

#define GGET_MAX_GROUP 4

struct GemmProblem {
  const bf16_t* A;
  const bf16_t* B;
  void* C;            // bf16 (or fp32 for the atomic epilogue)
  const bf16_t* R;    // residual, same shape/ld as C (EPI_RESIDUAL)
  int M, N, K;
  int lda, ldb, ldc;
  const int* m_dev;   // optional: rows of C read from device memory (head compaction counts)
  const int* k_dev;   // optional: reduction length read from device memory
  int tiles_n, tile_begin;  // filled by the launcher
  long slab_stride;         // EPI_SLAB_F32: elements between the fp32 slabs of consecutive K slices
  // EPI_ROPE: rotate columns [0, rope_cols) head-wise (64-wide heads, pair j <-> j+32) by the row's position
  const float* rope_cos;    // [max_pos][32] fp32
  const float* rope_sin;
  const int64_t* rope_pos;  // [M] or nullptr => position = row % rope_S
  int rope_S, rope_cols;
  int k_pad_zero;           // with k_dev: rows k_dev..round_up(k_dev, 64) of A are zero and those of B finite (whole K-tiles allowed)
  // EPI_GEGLU_FWD (NT, B = [gate rows | up rows] = [2 ff][K]): C = gu [M][2 ff] (pre-activations, kept for the backward),
  //   C2 = h [M][ff] = bf16(gelu(gate)) * up ; N = 2 ff.  EPI_GEGLU_BWD (the dh = dy W_down GEMM, N = ff): the epilogue reads
  //   G = gu [M][2 ff] and writes C = dgu [M][2 ff] = (dh * up * gelu'(gate) | dh * gelu(gate)); dh itself is never stored.
  void* C2;
  const bf16_t* G;
  int ldc2, ldg, ff;
  // EPI_NONE only: row m of the product is stored at row c_rows[m] of C (a scatter fused into the epilogue: the SMTP head's
  // dHl -> dP and dHm -> d hidden, which were separate gather_rows launches); nullptr = row m
  const int* c_rows;
  // Slot-sorted SMTP head (engine.hip, round 4; single-problem NT / NN launches through the 128 x 192 persistent tile):
  //  a_rows     - row m of the product reads row a_rows[m] of A (a gather fused into the LDS-DMA's per-lane source offsets);
  //  b_tile_off - the B operand of the 128-row tile mt starts b_tile_off[mt] ELEMENTS behind B (one n_token_proj slot per row tile);
  //  with c_rows, a NEGATIVE entry means "row not stored" (the pad rows that fill a slot's last tile).
  const int* a_rows;
  const int* b_tile_off;
  int b_tile_rows;          // rows per tile of b_tile_off: 128 (128 x 192 tiles) or 256 (256 x 256 tiles, N % 256 == 0)
};

struct GemmGroup {
  GemmProblem p[GGET_MAX_GROUP];
  int count;
  int super;   // row panels per L2 super-tile (tile order)
  int ablate;  // diagnostics (env GGET_GEMM_ABLATE): bit0 skip LDS-DMA, bit1 skip MFMA, bit2 skip the C store
  // stream-K launches (filled by the launcher from the workspace registered with gget_gemm_streamk_workspace): one fp32 tile slot and
  // one flag word per block, the epoch that marks this launch's flags
  float* sk_partial;
  unsigned* sk_flags;
  unsigned sk_epoch;
  int sk_rounds, sk_rem, sk_parts, sk_a;   // full rounds, tiles of the last round, K parts per tile (1: owners + helpers), owner K-tiles
  // two-workgroups-per-CU launches (gemm_persist_kernel MODE 2): the second workgroup to arrive on a CU starts `stagger_ticks` (100 MHz
  // ticks) late, so that one workgroup's epilogue meets the other's K-loop instead of its epilogue; cu_slots = one arrival counter per
  // (XCC, SE, SH, CU), monotonic (parity = arrival order)
  unsigned* cu_slots;
  int stagger_ticks;
  // grouped weight gradients (TN, gemm_ks_kernel<192,192>): when set, tile t of the launch also leaves the sum of squares of the bf16
  // values it stored in sq_partials[t] (the gradient-norm pass then skips these matrices: engine.hip gget_adamw_step); the launcher
  // sets sq_written when the kernel that does so was the one launched
  float* sq_partials;
  int sq_written;
};

// Split-K of the last round (gemm.hip: gemm_persist_kernel<..., SK = true>): launches whose tile count is no multiple of the CU count
// split the K range of the last round's tiles among the otherwise idle blocks; the blocks exchange fp32 partial tiles through this workspace.  `ws` = device memory,
// zero-initialised once by the caller, >= gget_gemm_streamk_bytes(); the launcher of the calling thread uses it until it is replaced
// (nullptr: stream-K off).  Launches that share a workspace must be ordered on one stream.
constexpr size_t kStreamKBlocks = 512;                                  // >= CUs
constexpr size_t kStreamKSlotBytes = (size_t)128 * 192 * 4;             // the largest tile that runs stream-K
constexpr size_t kStreamKFlagBytes = (kStreamKBlocks + 64) * sizeof(unsigned);
inline size_t gget_gemm_streamk_bytes() { return kStreamKFlagBytes + kStreamKBlocks * kStreamKSlotBytes; }
void gget_gemm_streamk_workspace(void* ws);

// CUs the launcher plans for: the device's, minus g_gemm_cu_reserve (gget_debug_set key 15: CUs left free for a collective's workgroups in
// data-parallel runs - gemm.hip has the reason), or GGET_GEMM_NUM_CU.  Every tile plan, persistent grid and split-K fit uses it.
extern int g_gemm_cu_reserve;
int gget_gemm_num_cu();

// mode: GGET_GEMM_NT/NN/TN, epi: GGET_EPI_*; problems of one group share mode and epilogue.
int gget_gemm_launch(int mode, int epi, GemmGroup& g, int split_k, hipStream_t st);
int gget_gemm_single(int mode, int epi, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                     int lda, int ldb, int ldc, const int* m_dev, const int* k_dev, int split_k, hipStream_t st, bool k_pad_zero = false,
                     const int* c_rows = nullptr);
