// gconv_common.h — helpers shared by the two gather-GEMM kernels: accumulator row map,
// activations and the three epilogues (LINEAR / GLU / BIGLU chain).
#ifndef PDSE_GCONV_COMMON_H
#define PDSE_GCONV_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// row of accumulator register r on lane-half h (32x32 C/D layout, cdna_hip_programming.md §3)
__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float act_f(float y, int act, float slope) {
  switch (act) {
    case PDSE_ACT_PRELU: return y > 0.f ? y : slope * y;
    case PDSE_ACT_ELU: return y > 0.f ? y : expm1f(y);
    case PDSE_ACT_SIGMOID: return sigmoid_f(y);
    default: return y;
  }
}


// acc0/acc1: MT accumulator tiles of this wave; (b, t, j) its output position on this lane.
template <int EPI, int MT>
__device__ __forceinline__ void gconv_epilogue(const pdse_gconv_desc& d, f32x16* acc0, f32x16* acc1, const int b,
                                               const int t, const int j, const bool pvalid, const int lane,
                                               const int h, const int mt0, const int mtiles) {
  const int64_t obase = (int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off;

  if constexpr (EPI == PDSE_EPI_LINEAR || EPI == PDSE_EPI_GLU) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (mt0 + m >= mtiles) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * (mt0 + m) + rho(r, h);
        if (pvalid && co < d.Cout) {
          float y = acc0[m][r];
          if (d.bias0) y += d.bias0[(int64_t)b * d.bias0_sb + co];
          if constexpr (EPI == PDSE_EPI_GLU) {
            float g = acc1[m][r];
            if (d.bias1) g += d.bias1[(int64_t)b * d.bias1_sb + co];
            y = y * sigmoid_f(g);
          }
          if (d.post_scale) y = y * d.post_scale[co] + d.post_shift[co];
          y = act_f(y, d.act, d.act_slope);
          const int64_t idx = obase + (int64_t)(co / d.out_cr) * d.out_sc_hi + (int64_t)(co % d.out_cr) * d.out_sc_lo;
          if (d.resid) y += d.resid[idx];
          d.out[idx] = y;
        }
      }
    }
  } else {
    // BiConvGLU / BiConvTransGLU tail, register to register (model/diff3.py:316-326, :345-351)
    f32x16 L = acc0[0], R = acc1[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = rho(r, h);
      L[r] += d.bias0[c];
      R[r] += d.bias1[c];
    }
    f32x16 mL, mR;
#pragma unroll
    for (int r = 0; r < 16; ++r) mL[r] = mR[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mL = __builtin_amdgcn_mfma_f32_32x32x2f32(d.wlc[r * 64 + lane], L[r], mL, 0, 0, 0);
      mR = __builtin_amdgcn_mfma_f32_32x32x2f32(d.wrc[r * 64 + lane], R[r], mR, 0, 0, 0);
    }
    f32x16 G;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = rho(r, h);
      const float ml = sigmoid_f(mL[r] + d.blc[c]);
      const float mr = sigmoid_f(mR[r] + d.brc[c]);
      G[r] = L[r] * mr + R[r] * ml;
    }
    if (d.C2 == 1) {
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) part += d.wc2[rho(r, h)] * G[r];
      float y = part + __shfl_xor(part, 32) + d.bc2[0];
      if (d.post_scale) y = y * d.post_scale[0] + d.post_shift[0];
      y = act_f(y, d.act, d.act_slope);
      if (pvalid && h == 0) d.out[obase] = y;
    } else {
      const int tiles2 = (d.C2 + 31) >> 5;
      for (int m2 = 0; m2 < tiles2; ++m2) {
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          O = __builtin_amdgcn_mfma_f32_32x32x2f32(d.wc2[(m2 * 16 + r) * 64 + lane], G[r], O, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = 32 * m2 + rho(r, h);
          if (pvalid && co < d.C2) {
            float y = O[r] + d.bc2[co];
            if (d.post_scale) y = y * d.post_scale[co] + d.post_shift[co];
            y = act_f(y, d.act, d.act_slope);
            d.out[obase + (int64_t)(co / d.out_cr) * d.out_sc_hi + (int64_t)(co % d.out_cr) * d.out_sc_lo] = y;
          }
        }
      }
    }
  }
}
#endif
