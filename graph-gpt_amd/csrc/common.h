// Shared device helpers for the gfx950 kernels of libgget_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define GGET_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }

// round-to-nearest-even f32 -> bf16 (NaN kept quiet)
// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction; NaN -> quiet NaN), which clang emits for a float -> __bf16 cast.
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2_t));
}

__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// Wave-wide all-reduce without the LDS crossbar (a __shfl_xor is a ds_bpermute: ~100 cycles of dependent latency each,
// six of them per reduction): four DPP steps reduce inside each 16-lane row, then gfx950's v_permlane16_swap /
// v_permlane32_swap fold the four rows.  Every lane ends up with the full result.
typedef unsigned hw_u32x2_t __attribute__((ext_vector_type(2)));
template <typename Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
  auto dpp = [](float x, auto ctrl) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
  v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
  hw_u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_allreduce(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_allreduce(v, [](float a, float b) { return fmaxf(a, b); });
}

// exact-erf GELU, the reference's hidden_act="gelu" (transformers/activations.py GELUActivation), evaluated with the
// Abramowitz-Stegun 7.1.26 form of erfc: Phi(x) = 1/2 erfc(-x/sqrt 2), erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) e^{-z^2},
// t = 1 / (1 + p z), |error| <= 1.5e-7 absolute on erf (7.5e-8 on Phi) - three orders of magnitude below the bf16 rounding every
// result of it receives.  One v_rcp_f32, one v_exp_f32 and ~10 plain VALU operations per element instead of the ~100
// instructions (with divergent range branches) of the device library's erff: the activation runs in GEMM epilogues, where
// its VALU time is not hidden by anything.  e^{-x^2/2} is also the Gaussian density the derivative needs.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& gauss) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  gauss = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // e^{-x^2/2}
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = 0.5f * (p * t) * gauss;   // 1/2 erfc(|x| / sqrt 2)
  cdf = x >= 0.f ? 1.0f - q : q;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, g;
  gelu_parts(x, cdf, g);
  return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, g;
  gelu_parts(x, cdf, g);
  return fmaf(x * 0.39894228040143267794f, g, cdf);
}
// value and derivative together (one evaluation of the shared parts)
__device__ __forceinline__ void gelu_erf_both(float x, float& val, float& grad) {
  float cdf, g;
  gelu_parts(x, cdf, g);
  val = x * cdf;
  grad = fmaf(x * 0.39894228040143267794f, g, cdf);
}

// Element-wise dropout (nn.Dropout on a tensor: embed_dropout - modeling_helpers.py:97-98, mlp_act_dropout / mlp_dropout -
// utils_graphgpt.py:69-80).  Counter-based like the attention / path dropouts: element (a, b) of stream `stream` is dropped when
// the 24-bit hash of (seed, stream, a, b) - the same hash as smtp_rng in kernels.hip, graph-gpt_amd/smtp.py:_rng24 is its
// Python twin - falls below thresh = p * 2^24; kept elements are scaled by 1 / (1 - p) and rounded to bf16 as the reference's
// bf16 module does.  Backward regenerates the mask from the coordinates.
struct ElemDropArg {
  unsigned thresh;    // 0 => off
  float inv_keep;
  unsigned seed;
  // var-len token layout: the mask is keyed by the LOGICAL row b * S + s of the padded [B,S] grid, so that the compact layout draws
  // the same random stream as the padded one (rows[t] = logical row of compact row t; nullptr: the rows are the logical rows)
  const int32_t* rows;
};
#define GGET_DROP_STREAM_EMBED 48u
#define GGET_DROP_STREAM_MLP_ACT 49u
#define GGET_DROP_STREAM_MLP_OUT 50u
#ifdef __HIPCC__
__device__ __forceinline__ unsigned elem_row(const ElemDropArg& E, long t) { return E.rows ? (unsigned)E.rows[t] : (unsigned)t; }
__device__ __forceinline__ float elem_drop_mul(const ElemDropArg& E, unsigned stream, unsigned a, unsigned b) {
  if (E.thresh == 0) return 1.f;
  unsigned x = E.seed ^ (stream * 0x9E3779B1u);
  x += a * 0x85EBCA77u + b * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return (x >> 8) < E.thresh ? 0.f : E.inv_keep;
}
#endif

// ---- host side error plumbing (engine.cpp owns the storage) ----
void gget_set_error(const char* fmt, ...);
#define GGET_HIP_CHECK(expr)                                                                  \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      gget_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
#define GGET_LAUNCH_CHECK() GGET_HIP_CHECK(hipGetLastError())
#define GGET_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      gget_set_error(__VA_ARGS__);   \
      return 2;                      \
    }                                \
  } while (0)
