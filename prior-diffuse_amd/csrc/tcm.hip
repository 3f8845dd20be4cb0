// tcm.hip — one dilated residual block of the eps-net's temporal convolution modules
// (model/diff3.py:215-257) per launch, over [B,256,T]:
//
//     h   = conv1(x)                                   1x1, 256 -> 64   (done by the PREVIOUS launch)
//     g   = main(BN(PReLU(h))) * sigmoid(mask(BN(PReLU(h))))          k = 5, dilation d, 64 -> 64
//     x'  = conv2(BN(PReLU(g))) + x                    1x1, 64 -> 256
//     h'  = conv1_next(x')                             1x1, 256 -> 64   (the NEXT block's conv1)
//
// The three-launch form (gconv.hip) is latency-bound: T = 401 gives 13 position tiles per item, so
// each 1x1 launch is a few hundred waves walking a long dependent MFMA chain.  Here a workgroup
// of 4 waves owns 32 frames of one utterance for the whole block and splits every GEMM four ways:
//   A  dilated branches: wave (mi, kh) = output tile mi, K half kh (80 k-steps, main + mask share
//      each gathered operand); halves meet in LDS, gate -> PReLU -> BN, result back to LDS (B operand
//      of conv2, rows conflict-free);
//   B  conv2: wave w produces output channels 64w..64w+63 (+ residual), stores x';
//   C  next conv1: those accumulators ARE the B operand of the next 1x1 (accumulator row order
//      rho(r,h); the host packs the weight rows in that order), K split over the four waves, summed
//      through LDS in a fixed order (deterministic).
// Only h (64 channels) carries a halo, so nothing is recomputed; h ping-pongs between two buffers
// (other workgroups gather from it), x may be updated in place.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gconv_common.h"
#include "pdse.h"
#include "pdse_internal.h"

#define REQ(cond, msg)     \
  do {                     \
    if (!(cond)) {         \
      pdse_set_error(msg); \
      return 1;            \
    }                      \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float prelu_f(float v, float slope) { return v > 0.f ? v : slope * v; }

// Addressing: every global access is (wave-uniform base pointer) + (one of a few 32-bit lane offsets), so the
// compiler keeps the bases in SGPRs and no per-access 64-bit vector arithmetic or address registers are needed.
__global__ __launch_bounds__(256, 2) void tcm_block_kernel(const pdse_tcm_desc d) {
  __shared__ float xfs[64][4];          // per input channel: scale/shift of the main and of the mask branch
  __shared__ float gpar[64][4];         // per gate channel: main bias, mask bias, BN scale, BN shift (conv2 input)
  __shared__ float bc2s[256], bn1s[64]; // conv2 / next-conv1 biases
  __shared__ float part[2][2][32][33];  // K-half 1 partial sums: [mi][main|mask][row][col]
  __shared__ float gl[64][33];          // BN(PReLU(gate)) : B operand of conv2
  __shared__ float red[4][64][33];      // next conv1: per-wave partial sums
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * 32, T = d.T;
  const int t = t0 + col;
  const bool tlive = t < T;
  const bool chain = d.h_out != nullptr;

  if (threadIdx.x < 64) {
    const int c = threadIdx.x;
    xfs[c][0] = d.xf[c * 2];
    xfs[c][1] = d.xf[c * 2 + 1];
    xfs[c][2] = d.xf[128 + c * 2];
    xfs[c][3] = d.xf[128 + c * 2 + 1];
  } else if (threadIdx.x < 128) {
    const int c = threadIdx.x - 64;
    gpar[c][0] = d.bmain[c];
    gpar[c][1] = d.bmask[c];
    gpar[c][2] = d.xf2[2 * c];
    gpar[c][3] = d.xf2[2 * c + 1];
  } else if (threadIdx.x < 192) {
    const int c = threadIdx.x - 128;
    bn1s[c] = chain ? d.bn1[c] : 0.f;
  }
  bc2s[threadIdx.x] = d.bc2[threadIdx.x];

  // ------------------------------------------------------------------ A: dilated branches
  const int mi = wave & 1, kh = wave >> 1;
  const float* hb = d.h + (size_t)b * 64 * T;
  int loff[5];        // lane offsets of the five taps: channel parity row + clamped frame
  bool tv[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int tt = t + (k - 2) * d.dil;
    tv[k] = tt >= 0 && tt < T;
    loff[k] = hh * T + min(max(tt, 0), T - 1);
  }
  float raw[80];
#pragma unroll
  for (int j = 0; j < 80; ++j) {
    // k-step ks = 80*kh + j: tap = ks >> 5, channel pair = ks & 31 (tap index differs between the two K halves)
    const float* u0 = hb + (size_t)(2 * (j & 31)) * T;                   // kh == 0: ks = j
    const float* u1 = hb + (size_t)(2 * ((80 + j) & 31)) * T;            // kh == 1: ks = 80 + j
    raw[j] = kh == 0 ? u0[loff[j >> 5]] : u1[loff[(80 + j) >> 5]];
  }
  const f32x4* WA = (const f32x4*)d.wbr + ((size_t)(mi * 2 + kh) * 20) * 2 * 64 + lane;
  f32x16 am, ak;
#pragma unroll
  for (int r = 0; r < 16; ++r) am[r] = ak[r] = 0.f;
  // conv2 weights and the residual: requested near the end of phase A so they arrive during the reduction
  const f32x4* W2 = (const f32x4*)d.wc2 + ((size_t)(2 * wave) * 8) * 64 + lane;
  f32x4 w2[2][8];
  const float* xb = d.x + ((size_t)b * 256 + 64 * wave) * T;
  float xres[2][16];
  const int lrow = 4 * hh * T + (tlive ? t : T - 1);   // lane offset of accumulator rows: rho(r,h) = (r&3) + 8*(r>>2) + 4*h
  __syncthreads();   // parameter tables
  constexpr int PF = 4;                 // weight groups in flight (one group = 4 k-steps = 8 MFMAs: an L2 round trip is longer)
  f32x4 wq[PF + 1][2];
#pragma unroll
  for (int g = 0; g < PF; ++g) {
    wq[g][0] = WA[(size_t)g * 128];
    wq[g][1] = WA[(size_t)g * 128 + 64];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 20; ++g) {
    if (g + PF < 20) {
      wq[(g + PF) % (PF + 1)][0] = WA[(size_t)(g + PF) * 128];
      wq[(g + PF) % (PF + 1)][1] = WA[(size_t)(g + PF) * 128 + 64];
    }
    if (g == 14) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) w2[q][gg] = W2[(size_t)(q * 8 + gg) * 64];
    }
    if (g == 17) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) xres[q][r] = (xb + (size_t)(32 * q + (r & 3) + 8 * (r >> 2)) * T)[lrow];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch PF groups ahead (hipcc sinks the loads to their use otherwise)
    const f32x4 wm = wq[g % (PF + 1)][0], wk = wq[g % (PF + 1)][1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * g + e;
      const int c = 2 * ((80 * kh + j) & 31) + hh;      // kh is wave-uniform
      const bool ok = kh == 0 ? tv[j >> 5] : tv[(80 + j) >> 5];
      const float v = raw[j];
      const f32x4 p = *(const f32x4*)&xfs[c][0];
      const float bm = ok ? p[0] * prelu_f(v, d.slope_main) + p[1] : 0.f;
      const float bk = ok ? p[2] * prelu_f(v, d.slope_mask) + p[3] : 0.f;
      am = __builtin_amdgcn_mfma_f32_32x32x2f32(wm[e], bm, am, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[e], bk, ak, 0, 0, 0);
    }
  }
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      part[mi][0][rho(r, hh)][col] = am[r];
      part[mi][1][rho(r, hh)][col] = ak[r];
    }
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rho(r, hh), c = 32 * mi + row;
      const f32x4 gp = *(const f32x4*)&gpar[c][0];
      const float m = am[r] + part[mi][0][row][col] + gp[0];
      const float k = ak[r] + part[mi][1][row][col] + gp[1];
      const float gte = m * sigmoid_f(k);
      gl[c][col] = gp[2] * prelu_f(gte, d.slope2) + gp[3];
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ B: conv2 + residual
  // next conv1 weights: in flight during conv2
  f32x4 wn[2][2][4];
  if (chain) {
    const f32x4* WN = (const f32x4*)d.wn1 + ((size_t)wave * 16) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int g = 0; g < 4; ++g) wn[q][mo][g] = WN[(size_t)((q * 2 + mo) * 4 + g) * 64];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 a2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) a2[q][r] = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float bv = gl[2 * (4 * g + e) + hh][col];
      a2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[0][g][e], bv, a2[0], 0, 0, 0);
      a2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[1][g][e], bv, a2[1], 0, 0, 0);
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const f32x4 bb = *(const f32x4*)&bc2s[64 * wave + 32 * q + 8 * r4 + 4 * hh];   // rows (r&3) = 0..3 are consecutive
#pragma unroll
      for (int i = 0; i < 4; ++i) a2[q][4 * r4 + i] += bb[i] + xres[q][4 * r4 + i];
    }
  if (tlive) {   // one predicated region for all 32 stores
    float* xo = d.x_out + ((size_t)b * 256 + 64 * wave) * T;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) (xo + (size_t)(32 * q + (r & 3) + 8 * (r >> 2)) * T)[lrow] = a2[q][r];
  }
  if (!chain) return;   // uniform over the grid

  // ------------------------------------------------------------------ C: next block's conv1
  f32x16 a1[2];
#pragma unroll
  for (int mo = 0; mo < 2; ++mo)
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[mo][r] = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bv = a2[q][4 * g + e];
        a1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wn[q][0][g][e], bv, a1[0], 0, 0, 0);
        a1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wn[q][1][g][e], bv, a1[1], 0, 0, 0);
      }
#pragma unroll
  for (int mo = 0; mo < 2; ++mo)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][32 * mo + rho(r, hh)][col] = a1[mo][r];
  __syncthreads();
  float* ho = d.h_out + (size_t)b * 64 * T + t0;
  const int cl = threadIdx.x & 31, c0 = threadIdx.x >> 5;   // rows c0, c0 + 8, ...
  if (t0 + cl < T) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + 8 * i;
      const float s = ((red[0][c][cl] + red[1][c][cl]) + (red[2][c][cl] + red[3][c][cl])) + bn1s[c];
      ho[(size_t)c * T + cl] = s;
    }
  }
}

int pdse_tcm_launch(const pdse_tcm_desc* d, hipStream_t s) {
  REQ(d && d->x && d->h && d->x_out && d->wbr && d->bmain && d->bmask && d->xf && d->wc2 && d->bc2 && d->xf2,
      "tcm: null pointer");
  REQ(!d->h_out || (d->wn1 && d->bn1), "tcm: the chained conv1 needs its weights and bias");
  REQ(d->h_out != d->h, "tcm: h_out must not alias h (other workgroups gather from h)");
  REQ(d->B > 0 && d->B <= 65535 && d->T > 0 && d->dil > 0, "tcm: bad sizes");
  hipLaunchKernelGGL(tcm_block_kernel, dim3((d->T + 31) / 32, d->B), dim3(256), 0, s, *d);
  return pdse_check_launch("tcm");
}
