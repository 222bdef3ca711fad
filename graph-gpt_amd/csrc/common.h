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

// exact (erf) GELU, the reference's hidden_act="gelu" (transformers/activations.py GELUActivation)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

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
