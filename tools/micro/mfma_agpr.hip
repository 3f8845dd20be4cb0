// Does the operand register class change the cost of v_mfma_f32_32x32x16_bf16?  6 dependent MFMAs per iteration, one wave
// per SIMD.  Variants: accumulator in VGPRs / AGPRs, B operand in VGPRs / AGPRs, A operand from registers / re-read from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_diag[3];
#define MF_VVV(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF_AVV(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MF_AVA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b))
#define MF_AAA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b))
template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  __shared__ f32x4 lds[1024];
  f32x16 acc, acc2;
  for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
  f32x4 a1 = {1.f * threadIdx.x, 2.f, 3.f, 4.f}, a2 = a1 * 2.f, a3 = a1 * 3.f, b1 = a1 * 0.5f, b2 = a1 * 0.25f, b3 = a1 * 0.125f;
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = a1 * (float)i;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (V == 4 || V == 5) { a1 = lds[threadIdx.x & 63]; a2 = lds[64 + (threadIdx.x & 63)]; a3 = lds[128 + (threadIdx.x & 63)]; }
    if (V == 0) { MF_VVV(acc, a1, b3); MF_VVV(acc, a3, b1); MF_VVV(acc, a2, b2); MF_VVV(acc, a1, b2); MF_VVV(acc, a2, b1); MF_VVV(acc, a1, b1); }
    if (V == 1) { MF_AVV(acc, a1, b3); MF_AVV(acc, a3, b1); MF_AVV(acc, a2, b2); MF_AVV(acc, a1, b2); MF_AVV(acc, a2, b1); MF_AVV(acc, a1, b1); }
    if (V == 2 || V == 4) { MF_AVA(acc, a1, b3); MF_AVA(acc, a3, b1); MF_AVA(acc, a2, b2); MF_AVA(acc, a1, b2); MF_AVA(acc, a2, b1); MF_AVA(acc, a1, b1); }
    if (V == 3) { MF_AAA(acc, a1, b3); MF_AAA(acc, a3, b1); MF_AAA(acc, a2, b2); MF_AAA(acc, a1, b2); MF_AAA(acc, a2, b1); MF_AAA(acc, a1, b1); }
    if (V == 5) {   // two accumulators alternating, as the K loop does (L, R with the same B)
      MF_AVA(acc, a1, b3); MF_AVA(acc, a3, b1); MF_AVA(acc, a2, b2); MF_AVA(acc, a1, b2); MF_AVA(acc, a2, b1); MF_AVA(acc, a1, b1);
      MF_AVA(acc2, a1, b3); MF_AVA(acc2, a3, b1); MF_AVA(acc2, a2, b2); MF_AVA(acc2, a1, b2); MF_AVA(acc2, a2, b1); MF_AVA(acc2, a1, b1);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  if (s == 1.2345f) out[0] = s;
  if ((threadIdx.x & 63) == 0) { atomicAdd(&g_diag[0], c1 - c0); atomicAdd(&g_diag[1], t1 - t0); atomicAdd(&g_diag[2], 1ull); }
}
template <int V>
void run(float* out, const char* what) {
  const int iters = 20000;
  unsigned long long z[3] = {0, 0, 0}, r[3];
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z));
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_diag), sizeof(r));
  printf("%-72s %7.1f cycles per MFMA, clock %.0f MHz\n", what, (double)r[0] / r[2] / iters / (V == 5 ? 12 : 6), (double)r[0] / r[1] * 100.0);
}
int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  run<0>(out, "acc VGPR, A VGPR, B VGPR");
  run<1>(out, "acc AGPR, A VGPR, B VGPR");
  run<2>(out, "acc AGPR, A VGPR, B AGPR");
  run<3>(out, "acc AGPR, A AGPR, B AGPR");
  run<4>(out, "acc AGPR, A VGPR re-read from LDS each iteration, B AGPR");
  run<5>(out, "two accumulators in turn, A from LDS, B AGPR");
  return 0;
}
