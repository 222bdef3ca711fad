// Per-CU store bandwidth as a function of how many CUs store at once (MI355X): is the C-store burst of a GEMM epilogue
// limited by the chip (HBM / fabric) or by the CU's own write path?  One 512-thread block per CU; a block is active when
// blockIdx.x % stride == 0 (stride 1: all 256 CUs).  Each active block writes `per_block` bytes of whole 128-byte lines
// (16 B per lane, 8 rows x 128 B per wave instruction, the GEMM epilogue's pattern: rows `pitch` bytes apart).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) store_kernel(unsigned char* dst, size_t per_block, int stride, int pitch, int reps) {
  if (blockIdx.x % stride) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* base = dst + (size_t)(blockIdx.x / stride) * per_block;
  const uint4 v = make_uint4(tid, tid * 3, tid * 5, tid * 7);
  for (int r = 0; r < reps; ++r) {
    // tile of 256 rows x pitch bytes; wave w owns rows [32 w, 32 w + 32); an instruction = 8 rows x 128 B
    for (size_t col = 0; col + 128 <= (size_t)pitch; col += 128)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const size_t row = (size_t)wave * 32 + rr * 8 + (lane >> 3);
        *reinterpret_cast<uint4*>(base + ((row * pitch + col + (lane & 7) * 16) % per_block)) = v;
      }
  }
}
int main() {
  const size_t total = 1u << 30;
  unsigned char* dst;
  hipMalloc(&dst, total);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("active CUs | tile 256 rows x pitch | per-CU GB/s | chip TB/s\n");
  for (int pitch : {512, 12288}) {
    for (int stride : {1, 2, 4, 8, 16, 32}) {
      const int active = 256 / stride;
      const size_t per_block = (size_t)256 * pitch;          // one tile
      const int reps = pitch == 512 ? 64 : 4;                // tiles written back to back (same addresses: stays in L2/MALL?)
      // distinct tile per repetition would need more memory; use a larger per_block window instead
      const size_t window = per_block;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(store_kernel, dim3(256), dim3(512), 0, 0, dst, window, stride, pitch, 1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(store_kernel, dim3(256), dim3(512), 0, 0, dst, window, stride, pitch, 1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      (void)reps;
      const double bytes = (double)per_block * active;
      printf("%10d | 256 x %5d B | %8.1f | %6.2f   (%.1f us)\n", active, pitch, per_block / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3);
    }
  }
  return 0;
}
