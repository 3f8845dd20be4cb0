// gru3.hip — the bidirectional GRU of the DB-AIAT transformer layers (model/dbaiat.py:45,83; H = 64, d_model 32) with the
// input projection fused (csrc/aia.hip: gru_kernel<64, true>), on the bf16 matrix cores with exact three-way operand
// splits (gconv_common.h: six bf16 products per fp32 multiply-add, fp32 accumulation, fp32-level accuracy).
//
// What bounded the fp32 kernel (1.57 ms per layer at B=32, T=401; 3.9 us per sequential step): six 32-row gate tiles x
// (32 + 16) dependent v_mfma_f32_32x32x2_f32 = 18.4k matrix-pipe cycles per step on the workgroup's ONE CU (4.6k per SIMD
// if spread evenly; 6.1k as distributed), i.e. the fp32 matrix rate of a single CU, not latency.  The same products as
// v_mfma_f32_32x32x16_bf16 on split operands are 6 x (4 + 2) x 6 = 216 instructions of 32 cycles = 6.9k pipe cycles.
//   * the state h [64 x 32 lines] lives in LDS ALREADY SPLIT, in B-fragment order (three bf16 planes, written by the gate
//     threads that produce it: each owns four units of one line = 8 bytes of a fragment per plane), so the matrix waves
//     read operands, not values; the fp32 state for z * h stays in the registers of the thread that owns the item;
//   * waves 0-3: the r and z tiles (W_ih x + W_hh h in one accumulator); waves 4, 5: W_hn h of the n tiles; waves 6, 7:
//     W_in x of the n tiles - and the input: they fetch x_{t+2}, split x_{t+1} and leave its planes in LDS for everyone;
//   * W_hh / W_ih tiles stay in registers as split A fragments for all S steps.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gconv_common.h"
#include "pdse.h"
#include "pdse_internal.h"

namespace {

__device__ __forceinline__ float g3_exp(const float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float g3_sigmoid(const float x) { return __builtin_amdgcn_rcpf(1.0f + g3_exp(-x)); }
#ifdef G3_LIBM_TANH
__device__ __forceinline__ float g3_tanh(const float x) { return tanhf(x); }
#else
// tanh x = 1 - 2 / (1 + e^{2x}): exp2 + rcp (absolute error ~1e-7; e^{2x} = inf gives 1, 0 gives -1)
__device__ __forceinline__ float g3_tanh(const float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + g3_exp(2.0f * x)); }
#endif

__device__ __forceinline__ f32x16 g3_mfma6(const uint4 (&a)[3], const uint4 (&b)[3], f32x16 acc) {
  acc = mfma_bf16(a[0], b[2], acc);
  acc = mfma_bf16(a[2], b[0], acc);
  acc = mfma_bf16(a[1], b[1], acc);
  acc = mfma_bf16(a[0], b[1], acc);
  acc = mfma_bf16(a[1], b[0], acc);
  acc = mfma_bf16(a[0], b[0], acc);
  return acc;
}

// four values -> their three bf16 planes, 8 bytes each (element i in half i & 1 of dword i >> 1)
__device__ __forceinline__ void g3_split4(const float (&x)[4], uint2 (&p)[3]) {
  uint32_t q[3][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t a1 = __float_as_uint(a) & 0xffff0000u, b1 = __float_as_uint(b) & 0xffff0000u;
    const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
    const uint32_t a2 = __float_as_uint(ra) & 0xffff0000u, b2 = __float_as_uint(rb) & 0xffff0000u;
    const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
    q[0][i] = (a1 >> 16) | b1;
    q[1][i] = (a2 >> 16) | b2;
    q[2][i] = (__float_as_uint(sa) >> 16) | (__float_as_uint(sb) & 0xffff0000u);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = make_uint2(q[k][0], q[k][1]);
}

constexpr int H = 64, G3 = 192;

__global__ __launch_bounds__(512) void gru3_kernel(const pdse_gru_desc d) {
  __shared__ uint4 hsP[4][3][64];       // h: [K block][plane][lane half hh * 32 + line], element j = unit 16 kb + 8 hh + j
  __shared__ uint4 xP[2][2][3][64];     // x_t planes, double-buffered by step parity: [buf][K block][plane][lane]
  __shared__ float gh[G3][33];          // rows 0..127: W_i x + W_h h + b_i + b_h (r, z); rows 128..191: W_hn h + b_hn
  __shared__ float gxn[H][33];          // W_in x + b_in
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, hh = lane >> 5;
  const int dir = blockIdx.y;
  const int S = d.axis == 0 ? d.F : d.T;
  const int per_b = d.axis == 0 ? d.T : d.F;
  const int nlines = d.B * per_b;
  const int64_t plane = (int64_t)d.T * d.F;
  const int64_t ss = d.axis == 0 ? 1 : d.F;

  // ---- weights: split A fragments, in registers for the whole sequence
  uint4 wh[4][3], wi[2][3];
  const bool has_h = wave < 6, has_i = wave < 4 || wave >= 6;
  const int tile = wave < 6 ? wave : wave - 2;   // waves 6, 7: the x half of the n tiles 4, 5
  if (has_h) {
    const uint4* A = reinterpret_cast<const uint4*>(d.whh) + ((size_t)(dir * 6 + tile) * 4 * 3) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wh[kb][p] = A[(size_t)(kb * 3 + p) * 64];
  }
  if (has_i) {
    const uint4* A = reinterpret_cast<const uint4*>(d.wih) + ((size_t)(dir * 6 + tile) * 2 * 3) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wi[kb][p] = A[(size_t)(kb * 3 + p) * 64];
  }
  // biases of this lane's 16 accumulator rows
  float bias[16];
  {
    const float* bh = d.bhh + dir * G3;
    const float* bi = d.bih + dir * G3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * tile + PDSE_KR(r) + 4 * hh;
      bias[r] = wave < 4 ? bh[row] + bi[row] : (wave < 6 ? bh[row] : bi[row]);
    }
  }

  // ---- gate items of this thread: units 4g..4g+3 of line l
  const int g = tid >> 5, l = tid & 31;
  const int Lg = blockIdx.x * 32 + l;
  const bool live = Lg < nlines;
  int64_t ybase;
  {
    const int b = live ? Lg / per_b : 0, w = live ? Lg - b * per_b : 0;
    const int64_t pos = d.axis == 0 ? (int64_t)w * d.F : (int64_t)w;
    ybase = (int64_t)b * 2 * H * plane + pos + (int64_t)(dir * H + 4 * g) * plane;
  }
  float hreg[4] = {0.f, 0.f, 0.f, 0.f};
  char* const hs_w = reinterpret_cast<char*>(&hsP[g >> 2][0][((g >> 1) & 1) * 32 + l]) + (g & 1) * 8;   // + plane * 1024 bytes

  // ---- the input: waves 6, 7 fetch K block (wave - 6) of x: channels 16 kb + 8 hh + j of line `col`
  const bool xw = wave >= 6;
  int64_t xbase = 0;
  bool xlive = false;
  if (xw) {
    const int Lx = blockIdx.x * 32 + col;
    xlive = Lx < nlines;
    const int b = xlive ? Lx / per_b : 0, w = xlive ? Lx - b * per_b : 0;
    xbase = (int64_t)b * (H / 2) * plane + (d.axis == 0 ? (int64_t)w * d.F : (int64_t)w) + (int64_t)(16 * (wave - 6) + 8 * hh) * plane;
  }
  float xr[8];
  auto fetch = [&](const int step) {
    const int sq = dir ? S - 1 - step : step;
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[j] = (xlive && step < S) ? d.x[xbase + (int64_t)j * plane + (int64_t)sq * ss] : 0.f;
  };
  auto publish = [&](const int buf) {   // xr -> planes of K block (wave - 6)
    uint4 p[3];
    split8(xr, p[0], p[1], p[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) xP[buf][wave - 6][k][lane] = p[k];
  };

  for (int i = tid; i < 4 * 3 * 64; i += 512) (&hsP[0][0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
  if (xw) {
    fetch(0);
    publish(0);
    fetch(1);
  }
  __syncthreads();

  for (int step = 0; step < S; ++step) {
    const int sq = dir ? S - 1 - step : step;
    const int buf = step & 1;
    if (wave < 6 || xw) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bias[r];
      if (has_i) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          uint4 b3[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) b3[p] = xP[buf][kb][p][lane];
          acc = g3_mfma6(wi[kb], b3, acc);
        }
      }
      if (has_h) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          uint4 b3[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) b3[p] = hsP[kb][p][lane];
          acc = g3_mfma6(wh[kb], b3, acc);
        }
      }
      if (xw) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gxn[32 * (tile - 4) + PDSE_KR(r) + 4 * hh][col] = acc[r];
        publish(buf ^ 1);   // x_{t+1}: read from the next step on (its buffer was last read during step t-1)
        fetch(step + 2);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) gh[32 * tile + PDSE_KR(r) + 4 * hh][col] = acc[r];
      }
    }
    __syncthreads();
    float hn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = 4 * g + k;
      const float r = g3_sigmoid(gh[u][l]);
      const float z = g3_sigmoid(gh[H + u][l]);
      const float n = g3_tanh(gxn[u][l] + r * gh[2 * H + u][l]);
      hn[k] = (1.f - z) * n + z * hreg[k];
      hreg[k] = hn[k];
    }
    uint2 hp[3];
    g3_split4(hn, hp);
    // every matrix wave has read hsP before it reached the barrier above
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(hs_w + p * 1024) = hp[p];
    if (live) {
#pragma unroll
      for (int k = 0; k < 4; ++k) d.y[ybase + (int64_t)k * plane + (int64_t)sq * ss] = hn[k];
    }
    __syncthreads();
  }
}

}  // namespace

// called by pdse_gru_launch (aia.hip) when the descriptor carries split weights (d->split)
int pdse_gru3_launch(const pdse_gru_desc* d, hipStream_t s) {
  if (d->H != 64 || !d->x || !d->wih || !d->bih) {
    pdse_set_error("bigru: the split-bf16 kernel is the fused H == 64 form (x, wih, bih)");
    return 1;
  }
  const int nlines = d->B * (d->axis == 0 ? d->T : d->F);
  hipLaunchKernelGGL(gru3_kernel, dim3((nlines + 31) / 32, 2), dim3(512), 0, s, *d);
  return pdse_check_launch("bigru (split-bf16)");
}
