// Sustained v_mfma_f32_32x32x2_f32 rate on this part (no memory traffic): what "100 % of the
// fp32 MFMA roofline" can mean in practice under the card's power/clock management.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 1.2345f) out[0] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 2; ++wpc)
    for (int iters : {2000, 20000, 200000}) {
      const int blocks = 256 * wpc;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 4 * 4096.0;
      printf("wg/cu %d iters %6d: %.3f ms  %.1f TFLOP/s  (=> %.2f GHz at 64 cycles/MFMA)\n", wpc, iters, ms,
             flop / ms / 1e9, (double)wpc * iters * 4 * 64 / (ms * 1e6));
    }
  return 0;
}
