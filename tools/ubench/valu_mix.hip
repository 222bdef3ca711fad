// Which VALU instructions overlap with MFMAs on a gfx950 SIMD?  Each wave runs ITER x 8 x { one v_mfma_f32_32x32x16_bf16
// (two accumulator chains), NV instructions of ONE kind on registers the MFMAs never touch }, W waves per SIMD.  For every kind:
// VALU alone, MFMA alone, both interleaved - and how much of the MFMA time disappeared under the VALU work.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int KIND> __device__ __forceinline__ void valu(float (&v)[16], int j) {
  float& x = v[j & 15];
  float& y = v[(j + 5) & 15];
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y));
  if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if (KIND == 3) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); x = __uint_as_float(r); }
  if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<double*>(&v[(2 * j) & 14])) : "v"(*reinterpret_cast<double*>(&v[(2 * j + 6) & 14])));
  if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));
  if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
  if (KIND == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
}

template <int KIND, int NV, bool MFMA, bool VALU>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f16v a = {0}, b = {0};
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.01f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MFMA) {
        if (m & 1) b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b, 0, 0, 0);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (VALU) {
#pragma unroll
        for (int j = 0; j < NV; ++j) valu<KIND>(v, m * NV + j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + b[i] + v[i];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND, int NV, bool MFMA, bool VALU>
float run(int W, int iters, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NV, MFMA, VALU>), dim3(256 * W), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NV, MFMA, VALU>), dim3(256 * W), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f / iters * 2.4e9f;
}

template <int KIND, int NV> void kind(const char* name, float* d) {
  const int iters = 10000;
  for (int W : {1, 2, 3, 4}) {
    const float m = run<KIND, NV, true, false>(W, iters, d), v = run<KIND, NV, false, true>(W, iters, d), both = run<KIND, NV, true, true>(W, iters, d);
    printf("%-22s x%2d per MFMA, W=%d: mfma %5.0f | valu %5.0f | both %5.0f | hidden %4.0f%% of the smaller\n", name, NV, W, m, v, both,
           100.f * (m + v - both) / (m < v ? m : v));
  }
}

int main() {
  float* d; (void)hipMalloc(&d, 4);
  printf("cycles at 2.4 GHz per iteration (8 MFMAs + 8 x NV VALU) of the W waves sharing a SIMD\n");
  kind<0, 4>("v_fma_f32", d);
  kind<0, 6>("v_fma_f32", d);
  kind<0, 8>("v_fma_f32", d);
  kind<0, 12>("v_fma_f32", d);
  kind<0, 16>("v_fma_f32", d);
  kind<4, 6>("v_pk_fma_f32", d);
  kind<1, 3>("v_exp_f32", d);
  kind<1, 6>("v_exp_f32", d);
  kind<2, 6>("v_max3_f32", d);
  kind<3, 6>("v_cvt_pk_bf16_f32", d);
  kind<5, 3>("v_mul_lo_u32", d);
  kind<7, 12>("v_xor_b32", d);
  return 0;
}
