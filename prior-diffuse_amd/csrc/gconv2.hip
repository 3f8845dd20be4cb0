// gconv2.hip — the pipelined form of the gather-GEMM convolution (korder 1).
//
// Same math, layout and epilogues as gconv.hip; what changes is the K loop, rebuilt around
// what the first profile showed (profiles/r01_kernel_stats_v1_baseline.csv): every wave was
// parked on one global load per MFMA.
//
//   * taps are the INNER, fully unrolled dimension (template NT): the per-lane gather
//     offsets and validity of every tap are computed once, before the loop, and live in
//     registers; the K order becomes (channel pair, tap);
//   * the loop walks channel-pair chunks of CP pairs = CP*NT k-steps; the activation values
//     and all A fragments of chunk i+1 are issued before the MFMAs of chunk i (two register
//     sets, ping-pong), so loads overlap the 64-cycle MFMAs instead of preceding each one;
//   * the launcher splits output-channel tiles over more waves when a layer has few
//     positions (TCM: 13 tiles per utterance), trading activation re-reads for occupancy.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

template <int N, int CP>
struct Chunk {
  float v[N];       // raw activation per k-step (lane = position, half = channel parity)
  float a0[N][4];   // A fragments, up to 4 M tiles
  float a1[N][4];
  float xs0[CP], xh0[CP], xs1[CP], xh1[CP];  // load-transform parameters (TCM only, XF != 0)
};

// XF: 0 no load transform; 1 one PReLU->BN set for both accumulators; 2 one set per accumulator
template <int EPI, int MT, int NT, int CP, bool SRC2, int XF>
__global__ __launch_bounds__(256) void gconv2_kernel(const pdse_gconv_desc d) {
  constexpr bool DUAL = (EPI != PDSE_EPI_LINEAR);
  constexpr int N = NT * CP;
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

  // ---- per-tap gather state, computed once
  int off0[NT], off1[SRC2 ? NT : 1];
  unsigned inb_mask = 0, isp_mask = 0;
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) {
    const int dt = d.taps[2 * tap], df = d.taps[2 * tap + 1];
    const int tin = t + dt, fin = j * d.sf_in + df;
    const bool fok = pvalid && fin >= 0 && fin < d.Fin;
    const bool inb = fok && tin >= 0 && tin < d.Tin;
    if (inb) inb_mask |= 1u << tap;
    if (fok && tin == -1 && d.padrow != nullptr) isp_mask |= 1u << tap;
    off0[tap] = inb ? (int)((int64_t)b * d.in0.sb + (int64_t)tin * d.in0.st + (int64_t)fin * d.in0.sf + (int64_t)h * d.in0.sc) : 0;
    if (SRC2)
      off1[tap] = inb ? (int)((int64_t)b * d.in1.sb + (int64_t)tin * d.in1.st + (int64_t)fin * d.in1.sf + (int64_t)h * d.in1.sc) : 0;
  }
  const int ksteps = d.ksteps;
  const float* wp0 = d.w0 + (size_t)mt0 * ksteps * 64 + lane;
  const float* wp1 = DUAL ? d.w1 + (size_t)mt0 * ksteps * 64 + lane : nullptr;
  const int cps0 = d.in0.C >> 1;
  const int cps1 = SRC2 ? (d.in1.C >> 1) : 0;
  const int nch0 = (cps0 + CP - 1) / CP, nch1 = (cps1 + CP - 1) / CP;
  const int nchunks = nch0 + nch1;
  // frame -1 of the encoder reads the folded time bias; without a pad row the pointer still
  // names readable memory (never selected: isp_mask is 0) so the load below needs no branch
  const float* prow = d.padrow ? d.padrow + (int64_t)b * d.padrow_sb + h : d.w0;

  // chunk q -> (source s, first pair cp0); pairs are numbered globally: gp = (s ? cps0 : 0) + cp.
  // Straight-line code: every load is unconditional from a clamped (always readable) address
  // and masked afterwards, so the whole chunk is one scheduling region of independent loads.
  auto issue = [&](Chunk<N, CP>& c, const int q) {
    const bool s1 = SRC2 && q >= nch0;
    const int cp0 = (s1 ? q - nch0 : q) * CP;
    const int cps = s1 ? cps1 : cps0;
    const int gbase = s1 ? cps0 : 0;
    const float* sp = s1 ? d.in1.ptr : d.in0.ptr;
    const int sc2 = (int)(2 * (s1 ? d.in1.sc : d.in0.sc));
#pragma unroll
    for (int cc = 0; cc < CP; ++cc) {
      const bool live = cp0 + cc < cps;
      const int cp = live ? cp0 + cc : cps - 1;
      const int gp = gbase + cp;
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int i = cc * NT + tap;
        const int o = (SRC2 && s1) ? off1[tap] : off0[tap];
        float v = sp[o + cp * sc2];
        v = ((inb_mask >> tap) & 1u) ? v : 0.f;
        if constexpr (EPI == PDSE_EPI_BIGLU) {
          const float pv = prow[2 * gp];
          v = ((isp_mask >> tap) & 1u) ? pv : v;
        }
        c.v[i] = live ? v : 0.f;
        const int ks = gp * NT + tap;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int mc = (mt0 + m < mtiles) ? m : 0;   // tiles past Cout re-read tile 0; never stored
          c.a0[i][m] = wp0[((size_t)mc * ksteps + ks) * 64];
          if (DUAL) c.a1[i][m] = wp1[((size_t)mc * ksteps + ks) * 64];
        }
      }
      if constexpr (XF != 0) {
        const int ci = 2 * gp + h;
        c.xs0[cc] = d.xf_scale0[ci];
        c.xh0[cc] = d.xf_shift0[ci];
        if constexpr (XF == 2) {
          c.xs1[cc] = d.xf_scale1[ci];
          c.xh1[cc] = d.xf_shift1[ci];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the chunk's loads together, ahead of the other chunk's MFMAs
  };

  auto consume = [&](Chunk<N, CP>& c, const int q) {
    // ELU on load exists only for the GCRN decoder's skip source (gcrn.py:152-155)
    bool elu_src = false;
    if constexpr (SRC2 && EPI == PDSE_EPI_GLU) elu_src = (q >= nch0) ? d.in1.act == PDSE_ACT_ELU : d.in0.act == PDSE_ACT_ELU;
#pragma unroll
    for (int cc = 0; cc < CP; ++cc) {
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int i = cc * NT + tap;
        float v = c.v[i];
        if constexpr (SRC2 && EPI == PDSE_EPI_GLU) {
          const float e = expm1f(fminf(v, 0.f));   // elu(0) = 0 keeps masked lanes at zero
          v = (elu_src && v < 0.f) ? e : v;
        }
        float v0 = v, v1 = v;
        if constexpr (XF != 0) {
          const bool inb = (inb_mask >> tap) & 1u;
          const float u = v > 0.f ? v : d.xf_slope0 * v;
          v0 = inb ? u * c.xs0[cc] + c.xh0[cc] : 0.f;   // zero padding is applied AFTER PReLU->BN (diff3.py:221-231)
          if constexpr (XF == 2) {
            const float u1 = v > 0.f ? v : d.xf_slope1 * v;
            v1 = inb ? u1 * c.xs1[cc] + c.xh1[cc] : 0.f;
          } else {
            v1 = v0;
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a0[i][m], v0, acc0[m], 0, 0, 0);
          if (DUAL) acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a1[i][m], v1, acc1[m], 0, 0, 0);
        }
      }
    }
  };

  // ---- ping-pong over the chunks: loads of q+1 are in flight while q is multiplied
  Chunk<N, CP> ca, cb;
  issue(ca, 0);
  int q = 0;
  for (; q + 1 < nchunks; q += 2) {
    issue(cb, q + 1);
    consume(ca, q);
    if (q + 2 < nchunks) issue(ca, q + 2);
    consume(cb, q + 1);
  }
  if (q < nchunks) consume(ca, q);

  gconv_epilogue<EPI, MT>(d, acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles);
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
static int pick_mt(const pdse_gconv_desc* d, int mtiles, int max_mt) {
  // prefer re-using each activation fragment for several channel tiles, but not at the price
  // of leaving SIMDs idle: want >= 2 waves per SIMD (2048 waves) before widening a wave's tile
  const long long tiles = (long long)d->B * (((long long)d->Tout * d->Fout + 31) / 32);
  int mt = 1;
  for (int cand = 2; cand <= max_mt; cand *= 2) {
    if (cand > mtiles) break;
    const long long waves = tiles * ((mtiles + cand - 1) / cand);
    if (waves >= 2048) mt = cand;
  }
  return mt;
}

template <int EPI, int NT, int CP, bool SRC2, int XF>
static void launch_mt(const pdse_gconv_desc* d, hipStream_t s, int mt, int gx, int mtiles) {
  const dim3 block(256);
  if constexpr (EPI == PDSE_EPI_BIGLU) {
    hipLaunchKernelGGL((gconv2_kernel<EPI, 1, NT, CP, SRC2, XF>), dim3(gx, d->B, 1), block, 0, s, *d);
  } else {
    if (mt >= 4)
      hipLaunchKernelGGL((gconv2_kernel<EPI, 4, NT, (CP > 1 ? CP / 2 : 1), SRC2, XF>), dim3(gx, d->B, (mtiles + 3) / 4), block, 0, s, *d);
    else if (mt == 2)
      hipLaunchKernelGGL((gconv2_kernel<EPI, 2, NT, CP, SRC2, XF>), dim3(gx, d->B, (mtiles + 1) / 2), block, 0, s, *d);
    else
      hipLaunchKernelGGL((gconv2_kernel<EPI, 1, NT, CP, SRC2, XF>), dim3(gx, d->B, mtiles), block, 0, s, *d);
  }
}

// Supported (epilogue, taps, two-source) combinations; keep in sync with packing.v2_supported().
int pdse_gconv2_launch(const pdse_gconv_desc* d, hipStream_t s) {
  const int mtiles = (d->Cout + 31) / 32;
  const int P = d->Tout * d->Fout;
  const int gx = ((P + 31) / 32 + 3) / 4;
  const bool two = d->in1.C > 0;
  const int nt = d->ntaps;
  // int32 gather offsets: every addressed element must sit below 2^31
  const long long span0 = (long long)d->B * d->in0.sb, span1 = two ? (long long)d->B * d->in1.sb : 0;
  if (span0 >= (1ll << 31) || span1 >= (1ll << 31) || d->in0.sc * 2 >= (1ll << 31)) {
    pdse_set_error("gconv2: input too large for 32-bit gather offsets");
    return 1;
  }
  // GLU keeps two accumulator sets: 4 tiles each would leave one wave per SIMD (268 VGPRs)
  const int mt = pick_mt(d, mtiles, d->epi == PDSE_EPI_GLU ? 2 : 4);
  const int xf = d->xf_mode;
#define GO(EPI, NT, CP, SRC2, XF)                              \
  do {                                                         \
    launch_mt<EPI, NT, CP, SRC2, XF>(d, s, mt, gx, mtiles);    \
    return pdse_check_launch("gconv2");                        \
  } while (0)
  if (d->epi == PDSE_EPI_LINEAR) {
    if (nt == 1 && !two && xf == 0) GO(PDSE_EPI_LINEAR, 1, 8, false, 0);
    if (nt == 1 && !two && xf == 1) GO(PDSE_EPI_LINEAR, 1, 8, false, 1);
    if (nt == 1 && two && xf == 0) GO(PDSE_EPI_LINEAR, 1, 8, true, 0);
    if (nt == 4 && !two && xf == 0) GO(PDSE_EPI_LINEAR, 4, 2, false, 0);
  } else if (d->epi == PDSE_EPI_GLU) {
    if (nt == 1 && two && xf == 0) GO(PDSE_EPI_GLU, 1, 4, true, 0);
    if (nt == 2 && two && xf == 0) GO(PDSE_EPI_GLU, 2, 2, true, 0);
    if (nt == 3 && !two && xf == 0) GO(PDSE_EPI_GLU, 3, 2, false, 0);
    if (nt == 5 && !two && xf == 2) GO(PDSE_EPI_GLU, 5, 1, false, 2);
  } else if (d->epi == PDSE_EPI_BIGLU && !two && xf == 0) {
    if (nt == 2) GO(PDSE_EPI_BIGLU, 2, 4, false, 0);
    if (nt == 4) GO(PDSE_EPI_BIGLU, 4, 2, false, 0);
    if (nt == 6) GO(PDSE_EPI_BIGLU, 6, 1, false, 0);
    if (nt == 10) GO(PDSE_EPI_BIGLU, 10, 1, false, 0);
  }
#undef GO
  pdse_set_error("gconv2: no pipelined instantiation for this (epilogue, taps, sources); pack with korder 0");
  return 1;
}
