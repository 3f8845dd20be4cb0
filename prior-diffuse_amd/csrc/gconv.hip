// gconv.hip — gather-GEMM convolution on the gfx950 fp32 matrix cores.
//
// One kernel family serves every dense contraction of the sampling path (SURVEY.md §8 A2,
// A3, A6, A7): Conv2d / ConvTranspose2d (per output phase) / dilated Conv1d / Linear /
// the STFT and inverse-STFT bases.  Orientation is chosen for CDNA4, not translated from
// the reference's NCHW cuDNN calls:
//
//   * output channels are the MFMA M dimension (weights = A operand, pre-packed on the
//     host in fragment order, one coalesced 256-B load per k-step, L2-resident);
//   * 32 consecutive output positions (t, j) are the N dimension and sit on the lanes, so
//     activation loads and stores run along the contiguous F axis of [B,C,T,F];
//   * v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (≡ an fmaf chain), the
//     fp32 matrix rate of the chip (MI355X_MICROARCH.md § Matrix cores).
//
// The accumulator layout (column on the lane, rows in the 16 registers) is exactly the B
// operand layout of the next MFMA, so a chain of 1x1 convolutions (the BiConvGLU tail:
// l_conv / r_conv masks, gating, closing 1x1, BatchNorm, PReLU — model/diff3.py:316-326)
// runs register-to-register with no LDS and no cross-lane traffic: k-step r of the next
// product takes acc[r] as B and a weight fragment whose k order follows rho(r, h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

template <int EPI, int MT, bool CIN1>
__global__ __launch_bounds__(256) void gconv_kernel(const pdse_gconv_desc d) {
  constexpr bool DUAL = (EPI != PDSE_EPI_LINEAR);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int P = d.Tout * d.Fout;
  const int p = (blockIdx.x * 4 + wave) * 32 + col;
  const bool pvalid = p < P;
  const int t = pvalid ? p / d.Fout : 0;
  const int j = pvalid ? p - t * d.Fout : 0;
  const int mtiles = (d.Cout + 31) >> 5;
  const int mt0 = blockIdx.z * MT;

  f32x16 acc0[MT], acc1[DUAL ? MT : 1];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[m][r] = 0.f;
      if (DUAL) acc1[m][r] = 0.f;
    }
  }

  const int ksteps = d.ksteps;
  const float* wp0 = d.w0 + (size_t)mt0 * ksteps * 64 + lane;
  const float* wp1 = DUAL ? d.w1 + (size_t)mt0 * ksteps * 64 + lane : nullptr;
  const int xf = d.xf_mode;

  if constexpr (!CIN1) {
    const int C0 = d.in0.C;
    int ks = 0;
    for (int tap = 0; tap < d.ntaps; ++tap) {
      const int dt = d.taps[2 * tap], df = d.taps[2 * tap + 1];
      const int tin = t + dt, fin = j * d.sf_in + df;
      const bool fok = pvalid && fin >= 0 && fin < d.Fin;
      const bool inb = fok && tin >= 0 && tin < d.Tin;
      const bool isp = fok && tin == -1 && d.padrow != nullptr;
#pragma unroll 1
      for (int s = 0; s < 2; ++s) {
        const pdse_src& S = s ? d.in1 : d.in0;
        const int cpairs = S.C >> 1;
        if (cpairs == 0) continue;
        const int cbase = s ? C0 : 0;
        const int64_t sc2 = 2 * S.sc;
        const float* pp = S.ptr + (inb ? (int64_t)b * S.sb + (int64_t)tin * S.st + (int64_t)fin * S.sf + (int64_t)h * S.sc
                                       : (int64_t)0);
        const float* prow = isp ? d.padrow + (int64_t)b * d.padrow_sb + cbase + h : nullptr;
        const int sact = S.act;
#pragma unroll 4
        for (int cp = 0; cp < cpairs; ++cp, ++ks) {
          float v = 0.f;
          if (inb) {
            v = pp[(int64_t)cp * sc2];
            if (sact) v = act_f(v, sact, 0.f);
          } else if (isp) {
            v = prow[2 * cp];
          }
          float v0 = v, v1 = v;
          if (xf && inb) {
            const int ci = cbase + 2 * cp + h;
            float u = v > 0.f ? v : d.xf_slope0 * v;
            v0 = u * d.xf_scale0[ci] + d.xf_shift0[ci];
            if (xf == 2) {
              float u1 = v > 0.f ? v : d.xf_slope1 * v;
              v1 = u1 * d.xf_scale1[ci] + d.xf_shift1[ci];
            } else {
              v1 = v0;
            }
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            if (mt0 + m < mtiles) {
              const float a0 = wp0[((size_t)m * ksteps + ks) * 64];
              acc0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, v0, acc0[m], 0, 0, 0);
              if (DUAL) {
                const float a1 = wp1[((size_t)m * ksteps + ks) * 64];
                acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, v1, acc1[m], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  } else {
    // Cin == 1: k enumerates the taps, lane-half h takes tap 2*ks + h
    const pdse_src& S = d.in0;
    const float* base = S.ptr + (int64_t)b * S.sb;
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const int tap = 2 * ks + h;
      float v = 0.f;
      if (pvalid && tap < d.ntaps) {
        const int tin = t + d.taps[2 * tap], fin = j * d.sf_in + d.taps[2 * tap + 1];
        if (tin >= 0 && tin < d.Tin && fin >= 0 && fin < d.Fin) {
          v = base[(int64_t)tin * S.st + (int64_t)fin * S.sf];
          if (S.act) v = act_f(v, S.act, 0.f);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (mt0 + m < mtiles) {
          const float a0 = wp0[((size_t)m * ksteps + ks) * 64];
          acc0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, v, acc0[m], 0, 0, 0);
          if (DUAL) {
            const float a1 = wp1[((size_t)m * ksteps + ks) * 64];
            acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, v, acc1[m], 0, 0, 0);
          }
        }
      }
    }
  }

  gconv_epilogue<EPI, MT>(d, tail_from_desc(d), acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles);
}

#define PDSE_REQUIRE(cond, msg)    \
  do {                             \
    if (!(cond)) {                 \
      pdse_set_error("gconv: " msg); \
      return 1;                    \
    }                              \
  } while (0)

int pdse_gconv_launch(const pdse_gconv_desc* d, hipStream_t s) {
  PDSE_REQUIRE(d != nullptr, "null descriptor");
  PDSE_REQUIRE(d->in0.ptr && d->out && d->w0 && d->taps, "null pointer (in0/out/w0/taps)");
  PDSE_REQUIRE(d->B > 0 && d->B <= 65535 && d->Tout > 0 && d->Fout > 0, "bad output extents");
  PDSE_REQUIRE(d->Cout > 0 && d->ntaps > 0 && d->out_cr > 0, "bad Cout/ntaps/out_cr");
  PDSE_REQUIRE((int64_t)d->Tout * d->Fout < (1ll << 30), "too many positions per batch item");
  const int Cin = d->in0.C + d->in1.C;
  if (d->cin1) {
    PDSE_REQUIRE(Cin == 1 && d->in1.C == 0, "cin1 path needs exactly one input channel");
    PDSE_REQUIRE(d->ksteps == (d->ntaps + 1) / 2, "ksteps != ceil(ntaps/2)");
    PDSE_REQUIRE(d->xf_mode == 0 && d->padrow == nullptr, "cin1 path has no load transform / pad row");
  } else {
    PDSE_REQUIRE(Cin >= 2 && (d->in0.C % 2) == 0 && (d->in1.C % 2) == 0, "channel counts must be even");
    PDSE_REQUIRE(d->in1.C == 0 || d->in1.ptr, "in1 has channels but no pointer");
    if (d->korder == 0) PDSE_REQUIRE(d->ksteps == d->ntaps * (Cin / 2), "ksteps != ntaps*Cin/2");
    else PDSE_REQUIRE(d->ksteps >= d->ntaps * (Cin / 2), "ksteps < ntaps*Cin/2");
    PDSE_REQUIRE(d->korder < 3 || d->epi != PDSE_EPI_BIGLU, "korder 3 / 4 / 5 have LINEAR / GLU epilogues only");
  }
  PDSE_REQUIRE((d->in0.blk == 0 && d->in1.blk == 0) || d->korder == 3 || d->korder == 4 || d->korder == 5, "channel-blocked sources (pdse_src.blk) are read by the korder 3 kernel only");
  if (d->xf_mode) {
    PDSE_REQUIRE(d->xf_scale0 && d->xf_shift0, "xf_mode set without scale/shift");
    PDSE_REQUIRE(d->xf_mode != 2 || (d->xf_scale1 && d->xf_shift1), "xf_mode 2 without second set");
  }
  PDSE_REQUIRE((d->post_scale == nullptr) == (d->post_shift == nullptr), "post_scale/post_shift must come together");
  {  // the epilogues read per-channel operands four floats at a time
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    PDSE_REQUIRE(al16(d->bias0) && al16(d->bias1) && al16(d->post_scale) && al16(d->post_shift) && al16(d->blc) &&
                     al16(d->brc) && al16(d->bc2) && al16(d->bias0_t0) && al16(d->bias1_t0),
                 "per-channel operands (biases, folded BatchNorm) must be 16-byte aligned");
    PDSE_REQUIRE((d->bias0_sb & 3) == 0 && (d->bias1_sb & 3) == 0, "bias batch strides must be multiples of 4 floats");
  }
  PDSE_REQUIRE(d->resid == nullptr || d->act == PDSE_ACT_NONE, "a residual input excludes an activation (x + f(..) is the last op)");
  if (d->nx_n != 0) {
    PDSE_REQUIRE(d->nx_n > 0 && d->nx_n <= 3 && d->nx_w, "nx: 1..3 chained tiles and their weights");
    PDSE_REQUIRE(d->epi == PDSE_EPI_BIGLU && (d->korder == 1 || d->korder == 2) && d->C2 == 64 && d->out_cr == 1,
                 "nx: chained 1x1 tiles need the pipelined BIGLU kernel with a 64-channel block output");
    for (int i = 0; i < d->nx_n; ++i) PDSE_REQUIRE(d->nx_bias[i] && d->nx_out[i], "nx: bias / output pointer missing");
    PDSE_REQUIRE(d->nx_row0 < d->nx_n, "nx_row0 out of range");
    PDSE_REQUIRE(d->w2 == nullptr || (d->nx_n == 1 && d->nx_keep == 0 && d->nx_row0 < 0),
                 "nx: a dual-phase launch chains one tile and does not keep its own output");
    PDSE_REQUIRE(d->nx_keep == 0 || d->out, "nx_keep without an output pointer");
  }
  PDSE_REQUIRE((d->bias0_t0 == nullptr) == (d->bias1_t0 == nullptr), "bias0_t0 / bias1_t0 must come together");
  PDSE_REQUIRE(d->bias0_t0 == nullptr || (d->nx_n > 0 && d->w2 == nullptr),
               "frame-0 biases are a feature of the chained single-phase BIGLU tail");
  if (d->korder == 1) {
    PDSE_REQUIRE(!d->cin1, "korder 1 needs Cin >= 2");
    return pdse_gconv2_launch(d, s);
  }
  if (d->korder >= 3 && d->korder <= 5) return pdse_gconv4_launch(d, s);   // split-bf16 (3) / plain bf16 (4) / f16x2 (5) GEMM-shaped convolutions (gconv4.hip)
  if (d->korder == 2) {   // split-bf16 BIGLU blocks (gconv3.hip)
    PDSE_REQUIRE(d->resid == nullptr, "BIGLU has no residual input");
    PDSE_REQUIRE(d->act == PDSE_ACT_NONE || d->act == PDSE_ACT_PRELU, "BIGLU stages end in PReLU or no activation");
    return pdse_gconv3_launch(d, s);
  }
  PDSE_REQUIRE(d->korder == 0, "unknown korder");
  const int mtiles = (d->Cout + 31) / 32;
  const int P = d->Tout * d->Fout;
  const int gx = ((P + 31) / 32 + 3) / 4;
  dim3 block(256);
  auto grid = [&](int mt) { return dim3(gx, d->B, (mtiles + mt - 1) / mt); };

#define LAUNCH(EPI, MT, C1)                                                          \
  hipLaunchKernelGGL((gconv_kernel<EPI, MT, C1>), grid(MT), block, 0, s, *d)

  switch (d->epi) {
    case PDSE_EPI_LINEAR: {
      if (d->cin1) {
        if (mtiles >= 4) LAUNCH(PDSE_EPI_LINEAR, 4, true);
        else if (mtiles >= 2) LAUNCH(PDSE_EPI_LINEAR, 2, true);
        else LAUNCH(PDSE_EPI_LINEAR, 1, true);
      } else {
        if (mtiles >= 4) LAUNCH(PDSE_EPI_LINEAR, 4, false);
        else if (mtiles >= 2) LAUNCH(PDSE_EPI_LINEAR, 2, false);
        else LAUNCH(PDSE_EPI_LINEAR, 1, false);
      }
      break;
    }
    case PDSE_EPI_GLU: {
      PDSE_REQUIRE(d->w1 != nullptr && !d->cin1, "GLU needs w1 and Cin >= 2");
      if (mtiles >= 4) LAUNCH(PDSE_EPI_GLU, 4, false);
      else if (mtiles >= 2) LAUNCH(PDSE_EPI_GLU, 2, false);
      else LAUNCH(PDSE_EPI_GLU, 1, false);
      break;
    }
    case PDSE_EPI_BIGLU: {
      PDSE_REQUIRE(d->w1 && !d->cin1 && d->Cout == 32, "BIGLU needs w1 and a 32-channel gate pair");
      PDSE_REQUIRE(d->bias0 && d->bias1 && d->wlc && d->wrc && d->blc && d->brc && d->wc2 && d->bc2,
                   "BIGLU chain pointers missing");
      PDSE_REQUIRE(d->C2 == 1 || (d->C2 > 0 && d->C2 % 32 == 0), "BIGLU C2 must be 1 or a multiple of 32");
      PDSE_REQUIRE(d->resid == nullptr, "BIGLU has no residual input");
      PDSE_REQUIRE(d->act == PDSE_ACT_NONE || d->act == PDSE_ACT_PRELU, "BIGLU stages end in PReLU or no activation");
      LAUNCH(PDSE_EPI_BIGLU, 1, false);
      break;
    }
    default:
      pdse_set_error("gconv: unknown epilogue");
      return 1;
  }
#undef LAUNCH
  return pdse_check_launch("gconv");
}
