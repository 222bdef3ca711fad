// Does a gfx950 SIMD overlap plain VALU work of one wave with the MFMAs of another (or of the same wave)?
// Each wave runs ITER x { NM mfma_f32_32x32x16_bf16 (two accumulator chains), NV independent v_fma_f32, NE v_exp_f32 }.
// Grid: 256 CUs x W waves per SIMD.  Prints cycles per iteration per wave and the sum of the separately measured parts.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mfma.hip -o valu_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 (as the builtin wants them)
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int NM, int NV, int NE, int ORDER>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ __bf16 lds[8 * 512];
  for (int i = threadIdx.x; i < 8 * 512; i += 256) lds[i] = (__bf16)(i * 1e-4f);
  __syncthreads();
  f16v a = {0}, b = {0};
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  float e[8];
  for (int i = 0; i < 8; ++i) e[i] = threadIdx.x * 1e-4f + i * 0.1f;
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {   // blocks: all MFMAs, then all VALU (what the attention tile does)
#pragma unroll
      for (int m = 0; m < NM; m += 2) {
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
#pragma unroll
      for (int j = 0; j < NE; ++j) e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7]) * 0.5f;
      __builtin_amdgcn_sched_barrier(0);
    } else if (ORDER == 2) {   // interleaved, the A operand of every MFMA freshly read from LDS (ds_read_b128 two MFMAs ahead)
      bf8 f0 = *reinterpret_cast<const bf8*>(&lds[(threadIdx.x & 63) * 8]);
      bf8 f1 = *reinterpret_cast<const bf8*>(&lds[512 + (threadIdx.x & 63) * 8]);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const bf8 cur = (m & 1) ? f1 : f0;
        if (m & 1) b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, y, b, 0, 0, 0);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, y, a, 0, 0, 0);
        if (m & 1) f1 = *reinterpret_cast<const bf8*>(&lds[((m + 2) & 7) * 512 + (threadIdx.x & 63) * 8]);
        else f0 = *reinterpret_cast<const bf8*>(&lds[((m + 2) & 7) * 512 + (threadIdx.x & 63) * 8]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
#pragma unroll
        for (int j = 0; j < NE / NM; ++j) e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7]) * 0.5f;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (ORDER == 3) {   // interleaved, the VALU work reads the accumulator the MFMA before last wrote (softmax on MFMA outputs)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (m & 1) b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b, 0, 0, 0);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f16v& src = (m & 1) ? a : b;     // the OTHER accumulator: written one MFMA earlier
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) v[j & 15] = __builtin_fmaf(src[j & 15], 1.0001f, v[j & 15]);
#pragma unroll
        for (int j = 0; j < NE / NM; ++j) e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7]) * 0.5f;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {            // interleaved inside the wave: one MFMA, then its share of the VALU work
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (m & 1) b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b, 0, 0, 0);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
#pragma unroll
        for (int j = 0; j < NE / NM; ++j) e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7]) * 0.5f;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + b[i] + v[i];
  for (int i = 0; i < 8; ++i) s += e[i];
  if (s == 12345.678f) out[0] = s;
}

template <int NM, int NV, int NE, int ORDER>
float run(int W, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g(256 * W), b(256);
  hipLaunchKernelGGL((k<NM, NV, NE, ORDER>), g, b, 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, NE, ORDER>), g, b, 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f / iters * 2.4e9f;     // cycles (at 2.4 GHz) per iteration of the W waves sharing a SIMD
}

int main(int argc, char** argv) {
  float* d; hipMalloc(&d, 4);
  const int iters = 20000;
  if (argc > 1) {   // one configuration, one launch pair (for rocprofv3 --pmc): argv[1] = mfma | fma | exp | blocked | inter, W = 4
    const int c = atoi(argv[1]);
    float r = c == 0 ? run<8, 0, 0, 0>(4, iters, d) : c == 1 ? run<0, 96, 0, 0>(4, iters, d) : c == 2 ? run<0, 0, 16, 0>(4, iters, d)
              : c == 3 ? run<8, 96, 16, 0>(4, iters, d) : run<8, 96, 16, 1>(4, iters, d);
    printf("config %d: %.0f cycles\n", c, r);
    return 0;
  }
  printf("cycles at 2.4 GHz per iteration per SIMD (W waves each doing one iteration)\n");
  for (int W : {1, 2, 4}) {
    float m = run<8, 0, 0, 0>(W, iters, d), v = run<0, 96, 0, 0>(W, iters, d), e = run<0, 0, 16, 0>(W, iters, d);
    float all0 = run<8, 96, 16, 0>(W, iters, d), all1 = run<8, 96, 16, 1>(W, iters, d);
    float all2 = run<8, 96, 16, 2>(W, iters, d), all3 = run<8, 96, 16, 3>(W, iters, d);
    printf("W=%d: 8 mfma %.0f | 96 fma %.0f | 16 exp(+mul) %.0f | all, blocked %.0f | all, interleaved %.0f | sum of parts %.0f | interleaved + LDS-fed A "
           "operands %.0f | interleaved, VALU reads MFMA outputs %.0f\n", W, m, v, e, all0, all1, m + v + e, all2, all3);
  }
  return 0;
}
