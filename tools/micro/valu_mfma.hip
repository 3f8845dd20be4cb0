// What does one wave alone on a SIMD pay for the vector instructions of an exact bf16 split beside bf16 MFMAs?
// Variants per loop iteration (one wave per SIMD, 4 waves per CU, every CU busy):
//   0: split8 only (44 vector instructions on 8 live values)
//   1: six dependent v_mfma_f32_32x32x16_bf16 only
//   2: both, source order (split, then MFMAs)
//   3: both, interleaved by sched_group_barrier: 1 MFMA, 8 vector instructions, ...
//   4: split8 x2 only;  5: 12 MFMAs on two accumulators;  6: variant 3 with two splits per 6 MFMAs
// Reports shader cycles per iteration (s_memtime) and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(const uint4& a, const uint4& b, f32x16 c) {
  union { uint4 u; bf16x8 v; } A, B;
  A.u = a; B.u = b;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, c, 0, 0, 0);
}
__device__ __forceinline__ void split8(const float (&x)[8], uint4& p1, uint4& p2, uint4& p3) {
  uint32_t q1[4], q2[4], q3[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    const float rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u);
    const float sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
    q1[i] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    q2[i] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    q3[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
  }
  p1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
  p2 = make_uint4(q2[0], q2[1], q2[2], q2[3]);
  p3 = make_uint4(q3[0], q3[1], q3[2], q3[3]);
}
__device__ unsigned long long g_diag[3];
template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  f32x16 acc, acc2;
  for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
  float x[8], y[8];
  for (int e = 0; e < 8; ++e) { x[e] = threadIdx.x * 1e-3f + e; y[e] = blockIdx.x * 1e-3f - e; }
  uint4 w1 = make_uint4(threadIdx.x, 2, 3, 4), w2 = w1, w3 = w1;
  uint4 p1 = w1, p2 = w1, p3 = w1, r1 = w1, r2 = w1, r3 = w1;
  uint4 q1 = w1, q2 = w1, q3 = w1;   // planes of the PREVIOUS iteration: what the MFMAs of this iteration read
  unsigned sink = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (V == 0 || V == 2 || V == 3 || V == 4 || V == 6) {
      split8(x, p1, p2, p3);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = x[e] * 1.0001f + 0.5f;
    }
    if (V == 4 || V == 6) {
      split8(y, r1, r2, r3);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = y[e] * 1.0001f + 0.25f;
    }
    if (V == 1 || V == 2 || V == 3 || V == 5 || V == 6) {
      acc = mf(w1, q3, acc); acc = mf(w3, q1, acc); acc = mf(w2, q2, acc);
      acc = mf(w1, q2, acc); acc = mf(w2, q1, acc); acc = mf(w1, q1, acc);
    }
    if (V == 5) {
      acc2 = mf(w1, q3, acc2); acc2 = mf(w3, q1, acc2); acc2 = mf(w2, q2, acc2);
      acc2 = mf(w1, q2, acc2); acc2 = mf(w2, q1, acc2); acc2 = mf(w1, q1, acc2);
    }
    if (V == 3 || V == 6) {
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, V == 6 ? 18 : 9, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (V == 0 || V == 4) {   // no MFMA reads the planes: keep them alive with one instruction per register (12 / 24 more)
      sink ^= p1.x ^ p1.y ^ p1.z ^ p1.w ^ p2.x ^ p2.y ^ p2.z ^ p2.w ^ p3.x ^ p3.y ^ p3.z ^ p3.w;
      if (V == 4) sink ^= r1.x ^ r1.y ^ r1.z ^ r1.w ^ r2.x ^ r2.y ^ r2.z ^ r2.w ^ r3.x ^ r3.y ^ r3.z ^ r3.w;
    }
    if (V == 6) { q1 = make_uint4(p1.x ^ r1.x, p1.y ^ r1.y, p1.z ^ r1.z, p1.w ^ r1.w); q2 = make_uint4(p2.x ^ r2.x, p2.y ^ r2.y, p2.z ^ r2.z, p2.w ^ r2.w); q3 = make_uint4(p3.x ^ r3.x, p3.y ^ r3.y, p3.z ^ r3.z, p3.w ^ r3.w); }
    else { q1 = p1; q2 = p2; q3 = p3; }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  float s = __uint_as_float(sink);
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  for (int e = 0; e < 8; ++e) s += x[e] + y[e];
  if (s == 1.2345f) out[0] = s;
  if ((threadIdx.x & 63) == 0) { atomicAdd(&g_diag[0], c1 - c0); atomicAdd(&g_diag[1], t1 - t0); atomicAdd(&g_diag[2], 1ull); }
}
template <int V>
void run(float* out, const char* what, int blocks_per_cu = 1) {
  const int iters = 20000;
  unsigned long long z[3] = {0, 0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z));
    hipLaunchKernelGGL(k<V>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
  }
  unsigned long long r[3];
  (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_diag), sizeof(r));
  printf("%-64s waves/SIMD %d: %7.1f cycles/iteration, clock %.0f MHz\n", what, blocks_per_cu, (double)r[0] / r[2] / iters, (double)r[0] / r[1] * 100.0);
}
int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  for (int w = 1; w <= 2; ++w) {
    run<0>(out, "split8 + 16 keep-alive VALU (60 vector instructions)", w);
    run<1>(out, "6 dependent MFMA 32x32x16 bf16", w);
    run<2>(out, "split8 then 6 MFMA (source order)", w);
    run<3>(out, "split8 + 6 MFMA interleaved 1:9", w);
    run<4>(out, "2 x split8 (120 vector instructions)", w);
    run<5>(out, "12 MFMA on two accumulators", w);
    run<6>(out, "2 x split8 + 6 MFMA interleaved 1:18", w);
  }
  return 0;
}
