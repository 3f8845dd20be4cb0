// How fast do DEPENDENT v_mfma_f32_32x32x2_f32 chains run?  NCH independent accumulators per wave, W waves per SIMD.
// (The BIGLU tails are chains of 16-32 dependent MFMAs; the gather loop has 4 independent accumulators.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a[NCH];
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) a[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) s += a[c][r];
  if (s == 1.2345f) out[0] = s;
}
template <int NCH>
void run(float* out, int wpc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000 / NCH, blocks = 256 * wpc;
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * NCH * 4096.0;
  printf("chains/wave %d  waves/SIMD %d : %.1f TFLOP/s\n", NCH, wpc, flop / ms / 1e9);
}
int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    run<1>(out, wpc);
    run<2>(out, wpc);
    run<4>(out, wpc);
  }
  return 0;
}
