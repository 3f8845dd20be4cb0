// gconv3.hip — the BiConvGLU / BiConvTransGLU blocks of the eps-net on the bf16 matrix cores with EXACT three-way
// bf16 operand splits (korder 2): same descriptors, same fp32 [B,C,T,F] tensors in HBM and the same epilogues as
// gconv2.hip; only the arithmetic of the contractions changes.
//
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate, and these blocks are bound by it
// (profiles/r01_*: 0.53-0.68 of the fp32 matrix peak).  An fp32 number is the exact sum of three bf16 numbers
// (3 x 8 significand bits, split by truncation: packing.split_bf16x3 / split8()), so a product a b is the sum of nine
// bf16 x bf16 products, each exact in fp32.  The six leading ones (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2) are issued as
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped ones are below 2^-23 |a b|, i.e. about one fp32
// rounding of the product.  Six bf16 MFMAs (32 cycles each, K = 16) replace eight fp32 MFMAs (64 cycles each, K = 2):
// 2.67x fewer matrix-pipe cycles at fp32-level accuracy (tests: the same goldens and tolerances as the fp32 path).
//
// Structure (Cin = 32, Cout = 32 per gated branch - every eps-net block but the composed encoder stage 1):
//   * a workgroup of 8 waves shares ONE LDS image of all weight fragments of the launch (gather weights of both
//     branches and both output phases, the chained 1x1 tails, biases, folded BatchNorm): 108-134 KB, filled once by
//     LDS-DMA (global_load_lds_dwordx4) while the tile's activation loads are in flight; fragments reach the matrix cores through ds_read_b128 (the fp32 kernel re-reads its A fragments
//     from L1/L2 in every wave: at 1.5x the bytes per weight that path would bound this kernel);
//   * K order (tap, channel): lane (position, half h) gathers channels 16q + 8h .. +7 of its position for tap `tap` -
//     eight 4-byte loads per (tap, q), the same number of vector-memory instructions per K as the fp32 kernel - splits
//     them in registers (~6 VALU operations per value) and feeds 12 MFMAs (L and R branch);
//   * all taps of a tile are requested before the K loop (16 NT registers) and overlap the LDS image fill;
//   * the BIGLU tail, the chained next-stage 1x1 tiles and the stores are gconv_common.h's, instantiated on the
//     split-bf16 tail image (accumulator tiles are re-split in registers and used as B operands, k order rho_bf16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

constexpr int popc3(int m) { return m ? (m & 1) + popc3(m >> 1) : 0; }
constexpr int rank3(int m, int tap) { return popc3(m & ((1 << tap) - 1)); }
constexpr int nth_bit(int m, int n) {   // position of the n-th set bit of m
  int pos = 0;
  for (; pos < 31; ++pos)
    if ((m >> pos) & 1) {
      if (n == 0) break;
      --n;
    }
  return pos;
}


// one 16-byte LDS-DMA: LDS destination = wave-uniform base + lane * 16 (the fragment areas are lane-linear)
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}
#define S3_FLOATS 640   // float operands behind the fragment areas (see image layout in the kernel)

// fragment counts (blocks of 3 planes x 64 lanes, 192 uint4 = 3 KB each)
__host__ __device__ constexpr int s3_blocks(int nt, int p1mask, int c2, int nx_n) {
  return 2 * (2 * nt) + 2 * (2 * popc3(p1mask)) + 2 + 2 + (c2 == 1 ? 0 : 4) + 4 * nx_n;
}

// LDS image: [gL NB][gR NB][gL1 NB1][gR1 NB1][lc 2][rc 2][c2 4 | 0][nx 4 nx_n] fragment blocks (192 uint4 each), then 512
// float operands: bl 0, br 32, bl0 64, br0 96, blc 128, brc 160, bc2 192 (64), ps 256 (64), pt 320 (64), nxb 384 (96), wc2v 480
struct s3_layout {
  int o_gR, o_gL1, o_gR1, o_lc, o_rc, o_c2, o_nx, o_f, c2b;
};
__device__ __forceinline__ s3_layout s3_make_layout(const int NB, const int NB1, const pdse_gconv_desc& d) {
  s3_layout L;
  L.c2b = d.C2 == 1 ? 0 : 4;
  L.o_gR = NB * 192;
  L.o_gL1 = 2 * NB * 192;
  L.o_gR1 = L.o_gL1 + NB1 * 192;
  L.o_lc = L.o_gR1 + NB1 * 192;
  L.o_rc = L.o_lc + 384;
  L.o_c2 = L.o_rc + 384;
  L.o_nx = L.o_c2 + L.c2b * 192;
  L.o_f = L.o_nx + d.nx_n * 768;
  return L;
}

// Fragment areas by LDS-DMA (global_load_lds_dwordx4: no registers, nothing to wait for until the barrier), one 1 KB chunk
// = one wave instruction.  The areas sit in the image in the order of the table, so chunk c of the image is chunk
// (c - first chunk of its area) of one source array; wave w takes chunks w, w + nwaves, ... in ONE loop (a loop per area
// made hipcc drain the DMA queue between areas).
__device__ __forceinline__ void s3_fill(uint4* img, const pdse_gconv_desc& d, const s3_layout& L, const int NB, const int NB1,
                                        const int lane, const int wave, const int nwaves) {
  const uint4* const srcs[8] = {reinterpret_cast<const uint4*>(d.w0), reinterpret_cast<const uint4*>(d.w1),
                                reinterpret_cast<const uint4*>(d.w2), reinterpret_cast<const uint4*>(d.w3),
                                reinterpret_cast<const uint4*>(d.wlc), reinterpret_cast<const uint4*>(d.wrc),
                                reinterpret_cast<const uint4*>(d.wc2), reinterpret_cast<const uint4*>(d.nx_w)};
  const int cnt[8] = {NB * 3, NB * 3, NB1 * 3, NB1 * 3, 6, 6, L.c2b * 3, d.nx_n * 12};
  const int total = L.o_f >> 6;
  for (int c = __builtin_amdgcn_readfirstlane(wave); c < total; c += nwaves) {
    int cc = c;
    const uint4* src = nullptr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (src == nullptr) {
        if (cc < cnt[k]) src = srcs[k] + cc * 64;
        else cc -= cnt[k];
      }
    }
    glds16(src + lane, img + c * 64);
  }
}

// float operands, one slot per thread (512 slots, one load each, issued behind the DMA and the activation requests)
__device__ __forceinline__ void s3_float_operands(uint4* img, const pdse_gconv_desc& d, const s3_layout& L, const int b, const int tid) {
  if (tid >= 512) return;
  const int k = tid & 31, c64 = (tid - 192) & 63;
  const bool bn = d.post_scale != nullptr;
  float v;
  if (tid < 32) v = d.bias0[(int64_t)b * d.bias0_sb + k];
  else if (tid < 64) v = d.bias1[(int64_t)b * d.bias1_sb + k];
  else if (tid < 96) v = (d.bias0_t0 ? d.bias0_t0 : d.bias0)[(int64_t)b * d.bias0_sb + k];
  else if (tid < 128) v = (d.bias1_t0 ? d.bias1_t0 : d.bias1)[(int64_t)b * d.bias1_sb + k];
  else if (tid < 160) v = d.blc[k];
  else if (tid < 192) v = d.brc[k];
  else if (tid < 256) v = c64 < d.C2 ? d.bc2[c64] : 0.f;
  else if (tid < 320) v = (bn && c64 < d.C2) ? d.post_scale[c64] : 1.0f;
  else if (tid < 384) v = (bn && c64 < d.C2) ? d.post_shift[c64] : 0.0f;
  else if (tid < 480) {
    const int i = (tid - 384) >> 5;
    v = i < d.nx_n ? d.nx_bias[i][(int64_t)b * d.nx_bias_sb[i] + k] : 0.f;
  } else v = d.C2 == 1 ? d.wc2[k] : 0.f;
  reinterpret_cast<float*>(img + L.o_f)[tid] = v;
}

__device__ __forceinline__ pdse_tail_s3 s3_tail(const uint4* img, const s3_layout& L) {
  const float* f = reinterpret_cast<const float*>(img + L.o_f);
  return pdse_tail_s3{img + L.o_lc, img + L.o_rc, img + L.o_c2, img + L.o_nx, f + 480, f, f + 32, f + 128, f + 160,
                      f + 192, f + 256, f + 320, f + 384, f + 64, f + 96};
}

// ---------------------------------------------------------------------------------------------------------------
// Encoder stage 1 with conv1 (and Preprocess) composed into the gather weights (nets._biconvglu_composed): two sources
// of two channels (x, x_init), ten taps -> K = 40 (three 16-deep blocks, the last half zero).  Block q, lane half h holds
// taps 4q + 2h and 4q + 2h + 1 x (x ch 0, x ch 1, x_init ch 0, x_init ch 1).  The gather is tiny (24 loads per lane); the
// time goes into the BIGLU tail and its three chained tiles at 401 x 79 positions, which is what moves to bf16 MFMAs.
// The image is 78 KB: 16 waves (four per SIMD) share it.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024, 4) void gconv3_in4_kernel(const pdse_gconv_desc d) {
  constexpr int NB = 3, WV = 16;
  extern __shared__ uint4 img[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int P = d.Tout * d.Fout;
  const int p = (blockIdx.x * WV + wave) * 32 + col;
  const bool pvalid = p < P;
  const int t = pvalid ? p / d.Fout : 0;
  const int j = pvalid ? p - t * d.Fout : 0;
  const s3_layout L = s3_make_layout(NB, 0, d);
  s3_fill(img, d, L, NB, 0, lane, wave, WV);

  // slot s = 2q + w: tap 4q + 2h + w (taps >= 10: the zero half of block 2)
  float raw[6][4];
  unsigned live = 0;
#pragma unroll
  for (int s_ = 0; s_ < 6; ++s_) {
    const int ta = 4 * (s_ >> 1) + (s_ & 1), tb = ta + 2;           // candidate taps of lane half 0 / 1 (compile time)
    const bool has_b = tb < 10, has_a = ta < 10;
    const int dt = h ? (has_b ? d.tap_dt[tb < 10 ? tb : 0] : 0) : (has_a ? d.tap_dt[ta < 10 ? ta : 0] : 0);
    const int df = h ? (has_b ? d.tap_df[tb < 10 ? tb : 0] : 0) : (has_a ? d.tap_df[ta < 10 ? ta : 0] : 0);
    const bool has = h ? has_b : has_a;
    const int tin = t + dt, fin = j * d.sf_in + df;
    const bool inb = has && pvalid && fin >= 0 && fin < d.Fin && tin >= 0 && tin < d.Tin;
    if (inb) live |= 1u << s_;
    const unsigned o0 = inb ? (unsigned)((int64_t)b * d.in0.sb + (int64_t)tin * d.in0.st + (int64_t)fin * d.in0.sf) : 0u;
    const unsigned o1 = inb ? (unsigned)((int64_t)b * d.in1.sb + (int64_t)tin * d.in1.st + (int64_t)fin * d.in1.sf) : 0u;
    raw[s_][0] = d.in0.ptr[o0];
    raw[s_][1] = (d.in0.ptr + d.in0.sc)[o0];
    raw[s_][2] = d.in1.ptr[o1];
    raw[s_][3] = (d.in1.ptr + d.in1.sc)[o1];
  }
  s3_float_operands(img, d, L, b, tid);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  f32x16 aL, aR;
#pragma unroll
  for (int r = 0; r < 16; ++r) aL[r] = aR[r] = 0.f;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = ((live >> (2 * q + (e >> 2))) & 1u) ? raw[2 * q + (e >> 2)][e & 3] : 0.f;
    uint4 b1, b2, b3;
    split8(x, b1, b2, b3);
    const int blk = q * 192 + lane;
    aL = mfma6(img + blk, b1, b2, b3, aL);
    aR = mfma6(img + L.o_gR + blk, b1, b2, b3, aR);
  }
  biglu_nx_epilogue(d, s3_tail(img, L), aL, aR, b, t, j, pvalid, lane, h);
}

// WV: waves per workgroup.  8 (two per SIMD, up to 256 registers each): every tap's activations are requested before the
// K loop.  16 (single-phase blocks only; four per SIMD, 128 registers): one tap in flight, the other waves cover its latency.
// PF: request every tap's activations before the K loop (16 NT registers); false: one tap in flight.
// PERSIST: the workgroup walks rounds blockIdx.x, blockIdx.x + gridDim.x, ... of its batch item (image filled once); false:
// one round per workgroup - the 16-wave form has no registers left for the loop-carried state (32 spills, 181 -> 200 us).
__device__ long long* g_trace3 = nullptr;   // PDSE_S3_TRACE=1 (diagnostic): [workgroup][wave][8] clock sums of the persistent loop

template <int NT, int P1MASK, bool NX, int WV, bool PF = (WV != 16), bool PERSIST = (WV != 16)>
__global__ __launch_bounds__(64 * WV, WV / 4) void gconv3_kernel(const pdse_gconv_desc d) {
#ifdef PDSE_DIAG
  long long* const trace = g_trace3;
#else
  long long* const trace = nullptr;
#endif
  const long long c_start = trace ? clock64() : 0;
  constexpr int NB = 2 * NT, NT1 = popc3(P1MASK), NB1 = 2 * NT1;
  extern __shared__ uint4 img[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int P = d.Tout * d.Fout;

  // The workgroup is PERSISTENT over the position tiles of its batch item: the LDS image (weights of the launch, the
  // item's biases) is filled once, then rounds blockIdx.x, blockIdx.x + gridDim.x, ... of WV tiles each are walked with
  // no barrier in between, so the waves drift apart and one wave's VALU-heavy tail runs beside another's MFMA-heavy K
  // loop; the next round's activations are requested before the current round's tail.  (One round per workgroup left
  // ~45 % of a round to fill latency, launch gaps and barrier skew: the image allows one workgroup per CU.)
  const s3_layout LY = s3_make_layout(NB, NB1, d);
  const int o_gR = LY.o_gR, o_gL1 = LY.o_gL1, o_gR1 = LY.o_gR1;
  s3_fill(img, d, LY, NB, NB1, lane, wave, WV);
  s3_float_operands(img, d, LY, b, tid);
  __syncthreads();   // the image is complete (DMA + float operands)
  const pdse_tail_s3 tl = s3_tail(img, LY);
  const int nrounds = ((P + 31) / 32 + WV - 1) / WV;

  // position of this lane in round rd and its per-tap gather state
  struct pos_t {
    int t, j;
    bool pvalid;
    unsigned inb_mask;
    unsigned off[NT];   // per-lane element offset of (tap, this half's first channel): 32-bit, added to a scalar channel base
  };
  auto locate = [&](const int rd, pos_t& ps) {
    const int p = (rd * WV + wave) * 32 + col;
    ps.pvalid = p < P;
    ps.t = ps.pvalid ? p / d.Fout : 0;
    ps.j = ps.pvalid ? p - ps.t * d.Fout : 0;
    ps.inb_mask = 0;
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int dt = d.tap_dt[tap], df = d.tap_df[tap];   // kernel arguments (scalar registers), not the device table
      const int tin = ps.t + dt, fin = ps.j * d.sf_in + df;
      const bool inb = ps.pvalid && fin >= 0 && fin < d.Fin && tin >= 0 && tin < d.Tin;
      if (inb) ps.inb_mask |= 1u << tap;
      ps.off[tap] = (unsigned)((inb ? (int64_t)b * d.in0.sb + (int64_t)tin * d.in0.st + (int64_t)fin * d.in0.sf : 0) +
                               (int64_t)(8 * h) * d.in0.sc);
    }
  };
  // channel c of the 16q + e enumeration: a wave-uniform base (scalar registers) + the lane's 32-bit offset - the
  // global_load saddr form, no per-load 64-bit address arithmetic or address registers
  auto chan = [&](const int c) -> const float* { return d.in0.ptr + (int64_t)c * d.in0.sc; };

  f32x16 aL, aR, aL1, aR1;
  auto zero_acc = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) aL[r] = aR[r] = aL1[r] = aR1[r] = 0.f;
  };
  auto block = [&](const float (&x)[8], const int tap, const int q) {   // one 16-channel K block of one tap
    uint4 b1, b2, b3;
    split8(x, b1, b2, b3);
    const int blk = (tap * 2 + q) * 192 + lane;
    aL = mfma6(img + blk, b1, b2, b3, aL);
    aR = mfma6(img + o_gR + blk, b1, b2, b3, aR);
    if constexpr (P1MASK != 0) {
      if ((P1MASK >> tap) & 1) {   // folds after unrolling
        const int blk1 = (rank3(P1MASK, tap) * 2 + q) * 192 + lane;
        aL1 = mfma6(img + o_gL1 + blk1, b1, b2, b3, aL1);
        aR1 = mfma6(img + o_gR1 + blk1, b1, b2, b3, aR1);
      }
    }
  };
  auto request_all = [&](const pos_t& ps, float (&raw)[NT][16]) {   // [tap][q * 8 + e]: channels 16q + 8h + e
#pragma unroll
    for (int tap = 0; tap < NT; ++tap)
#pragma unroll
      for (int e = 0; e < 16; ++e) raw[tap][e] = chan((e >> 3) * 16 + (e & 7))[ps.off[tap]];
  };
  auto kloop_all = [&](const unsigned inb_mask, const float (&raw)[NT][16]) {
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const bool inb = (inb_mask >> tap) & 1u;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = inb ? raw[tap][q * 8 + e] : 0.f;
        block(x, tap, q);
      }
    }
  };
  auto kloop_tap = [&](const pos_t& ps) {   // one tap in flight
    float cur[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cur[e] = chan((e >> 3) * 16 + (e & 7))[ps.off[0]];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const bool inb = (ps.inb_mask >> tap) & 1u;
      float x0[8], x1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x0[e] = inb ? cur[e] : 0.f;
        x1[e] = inb ? cur[8 + e] : 0.f;
      }
      if (tap + 1 < NT) {   // the next tap's requests go out before this tap's matrix work
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = chan((e >> 3) * 16 + (e & 7))[ps.off[tap + 1]];
      }
      block(x0, tap, 0);
      block(x1, tap, 1);
    }
  };

  auto epilogue = [&](const int t, const int j, const bool pvalid) {
    if constexpr (P1MASK != 0 && NX) {
      // Two-phase block chained into the next stage's conv1 (gconv_common.h: biglu_dual_epilogue<true>), with the block
      // output streamed into the chained tile one 32-channel half at a time instead of being kept whole (16 registers less)
      const int64_t scz = d.nx_sc[0], nbin = d.nx_sf[0] >> 1;
      const int64_t offz = (int64_t)b * d.nx_sb[0] + (int64_t)t * d.nx_st[0] + (int64_t)j * d.nx_sf[0] + d.nx_off[0] + (int64_t)(4 * h) * scz;
      float* const zb = d.nx_out[0] + offz;
      const float* const ab = d.nx_add[0] ? d.nx_add[0] + offz : nullptr;
      const bool two = pvalid && j < d.Fout1;
      f32x16 Z0, Z1, Yt;
#pragma unroll
      for (int r = 0; r < 16; ++r) Z0[r] = (ab && pvalid) ? ab[(int64_t)PDSE_KR(r) * scz] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      biglu_tail_values(d, tl, aL, aR, lane, h, [&](const int m2, const int r, const float v) {
        Yt[r] = v;
        if (r == 15) Z0 = chain_s3(tl.nxw + m2 * 384 + lane, Yt, Z0);
      });
#pragma unroll
      for (int r = 0; r < 16; ++r) Z1[r] = (ab && two) ? ab[(int64_t)PDSE_KR(r) * scz + nbin] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      biglu_tail_values(d, tl, aL1, aR1, lane, h, [&](const int m2, const int r, const float v) {
        Yt[r] = v;
        if (r == 15) Z1 = chain_s3(tl.nxw + m2 * 384 + lane, Yt, Z1);
      });
      const f32x16 pb = ld16(tl.nxb + 4 * h);
      if (nbin == 1 && two) {
#pragma unroll
        for (int r = 0; r < 16; ++r) store_pair(zb + (int64_t)PDSE_KR(r) * scz, Z0[r] + pb[r], Z1[r] + pb[r]);
      } else if (pvalid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[(int64_t)PDSE_KR(r) * scz] = Z0[r] + pb[r];
        if (two) {
#pragma unroll
          for (int r = 0; r < 16; ++r) zb[(int64_t)PDSE_KR(r) * scz + nbin] = Z1[r] + pb[r];
        }
      }
    } else if constexpr (P1MASK == 27) {
      // last decoder stage (kernel (2,5), ONE output channel - validated at launch): one value per position and phase;
      // the generic two-phase epilogue keeps a 64-channel copy of the even phase, which at 256 registers spilled
      float ve = 0.f, vo = 0.f;
      biglu_tail_values(d, tl, aL, aR, lane, h, [&](const int, const int, const float v) { ve = v; });
      __builtin_amdgcn_sched_barrier(0);
      biglu_tail_values(d, tl, aL1, aR1, lane, h, [&](const int, const int, const float v) { vo = v; });
      float* const po = d.out + ((int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off);
      const int64_t bin = d.out_sf >> 1;
      const bool both = pvalid && j < d.Fout1;
      if (h == 0) {
        if (both && bin == 1) store_pair(po, ve, vo);
        else if (pvalid) {
          po[0] = ve;
          if (both) po[bin] = vo;
        }
      }
    } else if constexpr (P1MASK != 0) {
      biglu_dual_epilogue<NX>(d, tl, aL, aR, aL1, aR1, b, t, j, pvalid, lane, h);
    } else {
      biglu_nx_epilogue(d, tl, aL, aR, b, t, j, pvalid, lane, h);
    }
  };

  if constexpr (PERSIST && PF) {
    // software pipeline over the rounds: round r+1's activations are requested between round r's K loop and its tail
    pos_t ps;
    float raw[NT][16];
    int rd = blockIdx.x;
    const long long c_fill = trace ? clock64() : 0;
    long long c_k = 0, c_r = 0, c_e = 0, n_r = 0;
    if (rd < nrounds) {
      locate(rd, ps);
      request_all(ps, raw);
    }
    while (rd < nrounds) {
      const long long c0 = trace ? clock64() : 0;
      zero_acc();
      __builtin_amdgcn_sched_barrier(0);   // every request of this round is out before its K loop starts
      kloop_all(ps.inb_mask, raw);
      const long long c1 = trace ? clock64() : 0;
      const int t = ps.t, j = ps.j;
      const bool pvalid = ps.pvalid;
      rd += gridDim.x;
      if (rd < nrounds) {
        locate(rd, ps);
        request_all(ps, raw);
      }
      __builtin_amdgcn_sched_barrier(0);   // ... and the next round's before this round's tail
      const long long c2 = trace ? clock64() : 0;
      epilogue(t, j, pvalid);
      if (trace) {
        const long long c3 = clock64();
        c_k += c1 - c0;
        c_r += c2 - c1;
        c_e += c3 - c2;
        ++n_r;
      }
    }
    if (trace && lane == 0) {
      long long* q = trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WV + wave) * 8;
      q[0] = c_fill - c_start;
      q[1] = c_k;
      q[2] = c_r;
      q[3] = c_e;
      q[4] = n_r;
      q[5] = clock64() - c_start;
    }
  } else {
    auto round = [&](const int rd) {
      pos_t ps;
      locate(rd, ps);
      zero_acc();
      if constexpr (PF) {
        float raw[NT][16];
        request_all(ps, raw);
        __builtin_amdgcn_sched_barrier(0);
        kloop_all(ps.inb_mask, raw);
      } else {
        kloop_tap(ps);
      }
      epilogue(ps.t, ps.j, ps.pvalid);
    };
    if constexpr (PERSIST) {
      for (int rd = blockIdx.x; rd < nrounds; rd += gridDim.x) round(rd);
    } else {
      round(blockIdx.x);   // one round per workgroup (grid.x == nrounds)
    }
  }
}

template <int NT, int P1MASK, bool NX, int WV, bool PF = (WV != 16), bool PERSIST = (WV != 16)>
static int launch3(const pdse_gconv_desc* d, hipStream_t s) {
  const int P = d->Tout * d->Fout;
  const int rounds = ((P + 31) / 32 + WV - 1) / WV;
  // one workgroup per CU holds the image: about 256 persistent workgroups in all, each walking its share of its item's rounds
  static const int wgs = getenv("PDSE_S3_WGS") ? atoi(getenv("PDSE_S3_WGS")) : 256;
  int gx = (wgs + d->B - 1) / d->B;
  if (gx > rounds || !PERSIST) gx = rounds;
  if (gx < 1) gx = 1;
  const dim3 grid(gx, d->B, 1), block(64 * WV);
  static const bool tracing = PDSE_DIAG_ENV("PDSE_S3_TRACE") != nullptr;
  static long long* tbuf = nullptr;
  if (tracing && !tbuf) {
    (void)hipMalloc(&tbuf, (size_t)1 << 22);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace3), &tbuf, sizeof(tbuf));
  }
  const size_t lds = (size_t)s3_blocks(NT, P1MASK, d->C2, d->nx_n) * 192 * sizeof(uint4) + S3_FLOATS * sizeof(float);
  if (lds > 160 * 1024) {
    pdse_set_error("gconv3: LDS image too large");
    return 1;
  }
  const void* fn = (const void*)gconv3_kernel<NT, P1MASK, NX, WV, PF, PERSIST>;
  static unsigned long long attr_mask = 0;   // per instantiation and device
  if (pdse_lds_attr(fn, &attr_mask, "gconv3 lds attribute")) return 1;
  if (tracing) (void)hipMemsetAsync(tbuf, 0, (size_t)gx * d->B * WV * 64, s);
  hipLaunchKernelGGL((gconv3_kernel<NT, P1MASK, NX, WV, PF, PERSIST>), grid, block, lds, s, *d);
  if (tracing && PERSIST && PF) {   // diagnostic: averages over all waves of the persistent loop's phases, in shader clocks
    (void)hipStreamSynchronize(s);
    const size_t nw = (size_t)gx * d->B * WV;
    long long* h = (long long*)malloc(nw * 64);
    (void)hipMemcpy(h, tbuf, nw * 64, hipMemcpyDeviceToHost);
    double sum[6] = {0};
    for (size_t i = 0; i < nw; ++i)
      for (int k = 0; k < 6; ++k) sum[k] += (double)h[i * 8 + k];
    fprintf(stderr, "gconv3 trace NT %d P1MASK %d NX %d C2 %d %dx%d: fill %.0f | per round: kloop %.0f request %.0f tail %.0f | rounds %.2f total %.0f\n",
            NT, P1MASK, (int)NX, d->C2, d->Tout, d->Fout, sum[0] / nw, sum[1] / sum[4], sum[2] / sum[4], sum[3] / sum[4], sum[4] / nw, sum[5] / nw);
    free(h);
  }
  return pdse_check_launch("gconv3");
}

// korder 2: BIGLU, one source of 32 channels, 32 + 32 output channels, C2 in {64, 1}; validated by pdse_gconv_launch
static int launch3_in4(const pdse_gconv_desc* d, hipStream_t s) {
  const int P = d->Tout * d->Fout;
  const dim3 grid(((P + 31) / 32 + 15) / 16, d->B, 1), block(1024);
  const size_t lds = (size_t)s3_blocks(0, 0, 64, d->nx_n) * 192 * sizeof(uint4) + 6 * 192 * sizeof(uint4) + S3_FLOATS * sizeof(float);
  static unsigned long long attr_mask = 0;
  if (pdse_lds_attr((const void*)gconv3_in4_kernel, &attr_mask, "gconv3 lds attribute")) return 1;
  hipLaunchKernelGGL(gconv3_in4_kernel, grid, block, lds, s, *d);
  return pdse_check_launch("gconv3");
}

int pdse_gconv3_launch(const pdse_gconv_desc* d, hipStream_t s) {
  if (d->epi == PDSE_EPI_BIGLU && d->in0.C == 2 && d->in1.C == 2 && d->ntaps == 10 && d->w2 == nullptr) {
    // encoder stage 1, conv1 composed into the gather weights: (x, x_init) read directly
    const long long sp0 = (long long)d->B * d->in0.sb, sp1 = (long long)d->B * d->in1.sb;
    if (d->Cout != 32 || d->C2 != 64 || d->xf_mode != 0 || d->out_cr != 1 || d->padrow != nullptr || d->cin1 || !d->in1.ptr ||
        sp0 >= (1ll << 31) || sp1 >= (1ll << 31) || d->nx_n > 3 || !d->bias0 || !d->bias1 || !d->w1 || !d->wlc || !d->wrc ||
        !d->blc || !d->brc || !d->wc2 || !d->bc2) {
      pdse_set_error("gconv3: composed encoder stage 1 needs two 2-channel sources, 10 taps, C2 == 64, all BIGLU operands");
      return 1;
    }
    return launch3_in4(d, s);
  }
  const long long span0 = (long long)d->B * d->in0.sb;
  if (d->epi != PDSE_EPI_BIGLU || d->in1.C != 0 || d->in0.C != 32 || d->Cout != 32 || d->xf_mode != 0 || d->out_cr != 1 ||
      !(d->C2 == 64 || d->C2 == 1) || d->padrow != nullptr || d->cin1 || span0 >= (1ll << 31) || d->in0.sc * 40 >= (1ll << 31)) {
    pdse_set_error("gconv3: split-bf16 kernels cover BIGLU blocks with one 32-channel source, C2 in {64, 1}, out_cr 1");
    return 1;
  }
  if (!d->bias0 || !d->bias1 || !d->w1 || !d->wlc || !d->wrc || !d->blc || !d->brc || !d->wc2 || !d->bc2) {
    pdse_set_error("gconv3: null operand");
    return 1;
  }
  // Waves per workgroup of the single-phase (encoder) blocks: 16 (1024 threads, four waves per SIMD at <= 128 registers,
  // one tap in flight) when a batch item has enough position tiles to fill such workgroups (measured 229 -> 180 us on
  // the 401 x 39 stage), else 8.  PDSE_S3_WAVES = 8 | 16 overrides (tuning / diagnostics).
  static const int wv_env = getenv("PDSE_S3_WAVES") ? atoi(getenv("PDSE_S3_WAVES")) : 0;
  const int tiles = (d->Tout * d->Fout + 31) / 32;
  const bool wv16 = wv_env == 16 || (wv_env != 8 && tiles >= 100);
  if (d->w2 != nullptr) {
    if (d->w3 == nullptr || (d->out_sf & 1) || d->nx_n > 1 || (d->nx_n == 1 && d->C2 != 64)) {
      pdse_set_error("gconv3: dual phase needs w2 and w3, an even out_sf, at most one chained tile");
      return 1;
    }
    // transposed (two-phase) blocks always run 8 waves: a 16-wave form (the phases one after the other on one accumulator
    // pair to fit 128 registers, odd-phase taps gathered and split a second time) measured the same - 213 vs 211 us on
    // the 401 x 40 stage, 340 vs 346 us on the last stage - and was dropped
    static const bool pf_off = getenv("PDSE_S3_PF") && atoi(getenv("PDSE_S3_PF")) == 0;   // tuning: one tap in flight
    if (d->ntaps == 4 && d->p1mask == 5) {
      if (d->nx_n) return pf_off ? launch3<4, 5, true, 8, false>(d, s) : launch3<4, 5, true, 8>(d, s);
      return launch3<4, 5, false, 8, false>(d, s);
    }
    if (d->ntaps == 6 && d->p1mask == 27 && d->nx_n == 0 && d->C2 == 1)
      return pf_off ? launch3<6, 27, false, 8, false>(d, s) : launch3<6, 27, false, 8>(d, s);
  } else {
    if (d->C2 != 64 || d->nx_n > 3) {
      pdse_set_error("gconv3: single phase needs C2 == 64 and at most three chained tiles");
      return 1;
    }
    if (d->ntaps == 6) return wv16 ? launch3<6, 0, false, 16>(d, s) : launch3<6, 0, false, 8>(d, s);
    if (d->ntaps == 4) return launch3<4, 0, false, 8>(d, s);
  }
  pdse_set_error("gconv3: no split-bf16 instantiation for this (taps, phases)");
  return 1;
}
