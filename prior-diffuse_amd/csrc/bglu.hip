// bglu.hip - BiConvGLU / BiConvTransGLU blocks of the eps-net on PLANE tensors (include/pdse.h: pdse_bglu_desc).
// Reference semantics: model/diff3.py:307-326 (BiConvGLU), :329-351 (BiConvTransGLU), the BatchNorm2d + PReLU behind
// every stage (:122-141, :172-203) and the next stage's 1x1 conv1 (:146-149, :343-345).
//
// Round 3 redesign of csrc/gconv3.hip, from measurements of that kernel (profiles/r03_*):
//   * one tile of 32 positions cost 288 bf16 MFMAs (9.2k matrix-pipe cycles) and ~2450 vector instructions (9.8k issue
//     cycles), and the time per tile on a SIMD was their SUM plus ~5k of waits, scalar work and ~70 exec-mask branches:
//     vector instructions of one wave do not hide behind matrix instructions of the other wave of the SIMD, only behind
//     matrix instructions of their own stream (MI355X_MICROARCH.md: <= 5 vector instructions per 32x32x16 gap);
//   * so the loop is software-pipelined by hand: ONE wave per SIMD (4-wave workgroups, 512 registers), and every
//     iteration issues the K loop of tile i+1 - matrix instructions and LDS reads only, because the input arrives as
//     the producer's bf16 split planes - together with the tail of tile i - vector instructions mostly - in one
//     basic block (no exec-mask branches: out-of-range taps are zero margins, lanes beyond the last position compute
//     on position 0 and only the final stores are predicated);
//   * vector work removed from the tail: BatchNorm folded into conv2, -log2 e into l_conv / r_conv, biases seed the
//     accumulators from LDS, PReLU is mul + max.
// NP = 3: exact three-way bf16 split of every operand, six products per multiply-add (fp32-equivalent);
// NP = 1: plain bf16 operands, one product (the opt-in bf16 mode; its own tolerance).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

// Cache policy of the plane / skip stores (gconv_common.h: bstore16).  0 = write-back through the XCD's L2.  Measured: written
// through (17 = sc0 sc1) or non-temporal (2) the same launches take 3-7 x as long (encoder 401x79: 183 -> 1203 / 1352 us;
// decoder 401x40: 153 -> 566 / 1033 us) - the L2 assembling full lines before they go to HBM is worth that much here.
#ifndef BGLU_ST_AUX
#define BGLU_ST_AUX 0
#endif

namespace {

#ifdef BGLU_DIAG   // diagnostic build only (tools/time_bglu.py --diag): shader-clock and 100 MHz stamps around the loop
__device__ unsigned long long g_bglu_diag[4];
__device__ unsigned long long g_bglu_slots[97];   // BGLU_SLOTSTAMP: shader cycles per slot of the 8-wave form, summed over waves and tiles; [96]: tiles
#endif

constexpr int popc(int m) { return m ? (m & 1) + popc(m >> 1) : 0; }
constexpr int rank_of(int m, int tap) { return popc(m & ((1 << tap) - 1)); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}

// acc += A B: A = NP fragment planes at w[0], w[64], .. (this lane's entry), B = NP planes of the activation
template <int NP>
__device__ __forceinline__ f32x16 mm(const uint4* w, const uint4 (&b)[NP], f32x16 acc) {
  if constexpr (NP == 3) {
    const uint4 a1 = w[0], a2 = w[64], a3 = w[128];
    acc = mfma_bf16(a1, b[2], acc);   // smallest terms first
    acc = mfma_bf16(a3, b[0], acc);
    acc = mfma_bf16(a2, b[1], acc);
    acc = mfma_bf16(a1, b[1], acc);
    acc = mfma_bf16(a2, b[0], acc);
    acc = mfma_bf16(a1, b[0], acc);
  } else if constexpr (NP == 2) {   // f16x2: a2 b1, a1 b2, a1 b1
    const uint4 a1 = w[0], a2 = w[64];
    acc = mfma_f16(a2, b[0], acc);
    acc = mfma_f16(a1, b[1], acc);
    acc = mfma_f16(a1, b[0], acc);
  } else {
    acc = mfma_bf16(w[0], b[0], acc);
  }
  return acc;
}

// the same with the A fragments already in registers (read from LDS one slot ahead)
template <int NP>
__device__ __forceinline__ f32x16 mmf(const uint4 (&a)[NP], const uint4 (&b)[NP], f32x16 acc) {
  if constexpr (NP == 3) {
    acc = mfma_bf16(a[0], b[2], acc);
    acc = mfma_bf16(a[2], b[0], acc);
    acc = mfma_bf16(a[1], b[1], acc);
    acc = mfma_bf16(a[0], b[1], acc);
    acc = mfma_bf16(a[1], b[0], acc);
    acc = mfma_bf16(a[0], b[0], acc);
  } else if constexpr (NP == 2) {
    acc = mfma_f16(a[1], b[0], acc);
    acc = mfma_f16(a[0], b[1], acc);
    acc = mfma_f16(a[0], b[0], acc);
  } else {
    acc = mfma_bf16(a[0], b[0], acc);
  }
  return acc;
}

// round-to-nearest-even bf16 of two floats, packed (hipcc: v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16(const float a, const float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  union { bf16x2 h; uint32_t u; } c;
  c.h = __builtin_convertvector(v, bf16x2);
  return c.u;
}

// sc: the power of two that takes the values to the plane exponent (NP == 2 only; gconv_common.h: split8h)
template <int NP>
__device__ __forceinline__ void split8p(const float (&x)[8], uint4 (&p)[NP], const float sc = 1.0f) {
#if defined(BGLU_DIAG) && defined(BGLU_NO_SPLIT)   // timing ablation: no split arithmetic (results wrong)
  p[0] = make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
  if constexpr (NP == 3) {
    p[1] = make_uint4(__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7]));
    p[2] = p[0];
  }
  return;
#endif
  if constexpr (NP == 3) {
    split8(x, p[0], p[1], p[2]);
  } else if constexpr (NP == 2) {
    split8h(x, sc, p[0], p[1]);
  } else {
    p[0] = make_uint4(pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(x[4], x[5]), pack_bf16(x[6], x[7]));
  }
}

// a 32-channel accumulator tile as the B operand of the next contraction: two K blocks of 8 registers
template <int NP>
__device__ __forceinline__ void split16p(const f32x16& X, uint4 (&p)[2][NP], const float sc = 1.0f) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = X[8 * s + j];
    split8p<NP>(x, p[s], sc);
  }
}

template <int NP>
__device__ __forceinline__ f32x16 chain(const uint4* w, const uint4 (&p)[2][NP], f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) acc = mm<NP>(w + s * NP * 64, p[s], acc);
  return acc;
}

__device__ __forceinline__ float sigm2(const float m) {   // sigmoid of a pre-activation that is already scaled by -log2 e
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m));
}

__device__ __forceinline__ float vmax(const float a, const float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

#define BGLU_FLOATS 512
// float operands behind the fragment areas: bl 0, br 32, bl0 64, br0 96, blc 128, brc 160, bc2 192 (64), nxb 256 (96), wc2v 352
enum { F_BL = 0, F_BR = 32, F_BL0 = 64, F_BR0 = 96, F_BLC = 128, F_BRC = 160, F_BC2 = 192, F_NXB = 256, F_WC2V = 352 };

template <int NT, int P1MASK, int C2, int NXN, bool IN4, int NP, bool PIPE>
struct bglu_cfg {
  static constexpr int WV = PIPE ? 4 : 8;
  static constexpr int NB = IN4 ? 3 : 2 * NT;
  static constexpr int NT1 = popc(P1MASK), NB1 = 2 * NT1;
  static constexpr bool DUAL = P1MASK != 0;
  static constexpr int BS = NP * 64;   // uint4 per fragment block
  static constexpr int o_gL = 0, o_gR = NB * BS, o_gL1 = 2 * NB * BS, o_gR1 = o_gL1 + NB1 * BS, o_lc = o_gR1 + NB1 * BS,
                       o_rc = o_lc + 2 * BS, o_c2 = o_rc + 2 * BS, o_nx = o_c2 + (C2 == 64 ? 4 * BS : 0),
                       o_f = o_nx + NXN * 4 * BS;
  static constexpr size_t lds_bytes = (size_t)o_f * sizeof(uint4) + BGLU_FLOATS * sizeof(float);
};

// PIPE true: 4 waves (one per SIMD, 512 registers), the K loop of tile i+1 interleaved with the tail of tile i.
// PIPE false: 8 waves (two per SIMD, 256 registers), K loop and tail of the same tile one after the other - a wave issues at
// most one instruction per four cycles, so two waves per SIMD double the issue rate and cover each other's stalls.
template <int NT, int P1MASK, int C2, int NXN, bool IN4, int NP, bool PIPE, bool SPR = false>
__global__ __launch_bounds__(PIPE ? 256 : 512, PIPE ? 1 : 2) void bglu_kernel(const pdse_bglu_desc d) {
  using CF = bglu_cfg<NT, P1MASK, C2, NXN, IN4, NP, PIPE>;
  constexpr int WV = CF::WV, NB = CF::NB, NB1 = CF::NB1, BS = CF::BS;
  constexpr bool DUAL = CF::DUAL;
  extern __shared__ uint4 img[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int P = d.Tout * d.Fout;
  // NP == 2 (f16x2): the powers of two between the exponents of the planes (2^PDSE_F16_ACT_EXP), of the four weight groups
  // (d.qexp: gather, l_conv / r_conv, conv2, chained tiles) and of the accumulators (plane exponent + weight exponent)
  constexpr int PE = NP == 2 ? PDSE_F16_ACT_EXP : 0;
  const int qG = NP == 2 ? d.qexp[0] : 0, qLC = NP == 2 ? d.qexp[1] : 0, qC2 = NP == 2 ? d.qexp[2] : 0, qNX = NP == 2 ? d.qexp[3] : 0;
  [[maybe_unused]] const float s_in = pow2i(PE), s_G = pow2i(-qG), s_LC = pow2i(-(PE + qLC)), s_C2 = pow2i(-qC2), s_NX = pow2i(-qNX),
                               s_C2out = pow2i(-(PE + qC2)), s_NXout = pow2i(-(PE + qNX)), s_NXin = pow2i(PE + qNX);

  // ---- the LDS image: every weight fragment of the launch by LDS-DMA (1 KB per wave instruction), then the floats
  {
    const uint4* const srcs[8] = {reinterpret_cast<const uint4*>(d.w0), reinterpret_cast<const uint4*>(d.w1),
                                  reinterpret_cast<const uint4*>(d.w2), reinterpret_cast<const uint4*>(d.w3),
                                  reinterpret_cast<const uint4*>(d.wlc), reinterpret_cast<const uint4*>(d.wrc),
                                  reinterpret_cast<const uint4*>(d.wc2), reinterpret_cast<const uint4*>(d.nx_w)};
    const int cnt[8] = {NB * NP, NB * NP, NB1 * NP, NB1 * NP, 2 * NP, 2 * NP, C2 == 64 ? 4 * NP : 0, NXN * 4 * NP};
    constexpr int total = CF::o_f >> 6;
    for (int c = __builtin_amdgcn_readfirstlane(wave); c < total; c += WV) {
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
    float* const fo = reinterpret_cast<float*>(img + CF::o_f);
    for (int s = tid; s < BGLU_FLOATS; s += 64 * WV) {
      const int k = s & 31;
      float v = 0.f;
      if (s < 32) v = d.bias0[(int64_t)b * d.bias_sb + k];
      else if (s < 64) v = d.bias1[(int64_t)b * d.bias_sb + k];
      else if (s < 96) v = (d.bias0_t0 ? d.bias0_t0 : d.bias0)[(int64_t)b * d.bias_sb + k];
      else if (s < 128) v = (d.bias1_t0 ? d.bias1_t0 : d.bias1)[(int64_t)b * d.bias_sb + k];
      else if (s < 160) v = d.blc[k];
      else if (s < 192) v = d.brc[k];
      else if (s < 256) v = (s - 192) < C2 ? d.bc2[s - 192] : 0.f;
      else if (s < 352) {
        const int i = (s - 256) >> 5;
        v = (i < NXN && d.nx_bias[i]) ? d.nx_bias[i][(int64_t)b * d.nx_bias_sb[i] + k] : 0.f;
      } else if (s < 384) v = (C2 == 1) ? d.wc2v[k] : 0.f;
      if constexpr (NP == 2) {   // biases seed accumulators: they carry the accumulator's exponent (exact: powers of two)
        v *= s < 128 ? pow2i(PE + qG) : s < 192 ? pow2i(PE + qLC) : s < 256 ? (C2 == 64 ? pow2i(PE + qC2) : 1.0f)
             : s < 352 ? pow2i(PE + qNX) : pow2i(-(PE + qG));   // wc2v (C2 == 1) multiplies G, which is at the gather exponent
      }
      fo[s] = v;
    }
  }
  __syncthreads();
  const float* const fop = reinterpret_cast<const float*>(img + CF::o_f);
  const int ntiles = (P + 31) >> 5;
  const int nrounds = (ntiles + WV - 1) / WV;

  // ---- per-tile lane state
  struct pos_t {
    int t, j;
    bool valid;
    uint32_t vin;   // lane part of the input address: BYTE offset into the item's hp (planes) / unused (IN4)
  };
  const int Fp = d.hp_Fp;
  auto locate = [&](const int rd, pos_t& ps) {
    const int p = (rd * WV + wave) * 32 + col;
    ps.valid = p < P;
    const int pp = ps.valid ? p : 0;   // branch-free: lanes without a position work on position 0
    ps.t = pp / d.Fout;
    ps.j = pp - ps.t * d.Fout;
    // hp_par (sf_in == 2, validated): bin 2 j + df + f0 lives at ((df + f0) & 1) * Fh + j + ((df + f0) >> 1) - the lane part is j
    ps.vin = (uint32_t)(ps.t * (4 * NP * Fp) + h * (NP * Fp) + (d.hp_par ? ps.j : ps.j * d.sf_in)) << 4;
  };
  const int Fh = (Fp + 1) >> 1;
  // (tap, K block q, plane pl) of a lane = buffer resource of this item's hp + the lane's byte offset (pos_t::vin) + a
  // wave-uniform byte offset in a scalar register
  auto soff_in = [&](const int tap, const int q, const int pl) -> int {
    const int bin0 = d.tap_df[tap] + d.hp_f0;
    return (((d.tap_dt[tap] + d.hp_t0) * 4 + 2 * q) * (NP * Fp) + pl * Fp + (d.hp_par ? (bin0 & 1) * Fh + (bin0 >> 1) : bin0)) << 4;
  };
  // input operands of one tile: planes [tap][q][plane] (or, stage 1, the raw fp32 gathers [slot][4])
  struct in_t {
    uint4 pl[IN4 ? 1 : NT][2][NP];
    float raw[IN4 ? 6 : 1][4];
    unsigned live;
  };
  auto request_tap = [&](const pos_t& ps, in_t& in, const int tap) {
#if defined(BGLU_DIAG) && defined(BGLU_HALF_REQ)   // timing ablation (results wrong): every other tap re-uses its neighbour's registers
    if (tap & 1) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) in.pl[tap][q][pl] = in.pl[tap - 1][q][pl];
      return;
    }
#endif
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.hp + (int64_t)b * d.hp_sb, (uint32_t)d.hp_Tp * 4u * NP * (uint32_t)Fp * 16u);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) in.pl[tap][q][pl] = bload16(r_in, ps.vin, soff_in(tap, q, pl));
  };
  auto request_one = [&](const pos_t& ps, in_t& in, const int tap, const int q, const int pl) {   // one 16-byte load (spread schedules)
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.hp + (int64_t)b * d.hp_sb, (uint32_t)d.hp_Tp * 4u * NP * (uint32_t)Fp * 16u);
    in.pl[tap][q][pl] = bload16(r_in, ps.vin, soff_in(tap, q, pl));
  };
  auto request_in4 = [&](const pos_t& ps, in_t& in) {
    // slot s = 2q + w: tap 4q + 2h + w of the ten (2,5) taps, channels (x 0, x 1, x_init 0, x_init 1); taps >= 10: zero
    in.live = 0;
    const __amdgpu_buffer_rsrc_t r_x0 = make_rsrc(d.x0.ptr, 0xfffffffcu), r_x1 = make_rsrc(d.x1.ptr, 0xfffffffcu);
#pragma unroll
    for (int s_ = 0; s_ < 6; ++s_) {
      const int ta = 4 * (s_ >> 1) + (s_ & 1), tb = ta + 2;
      const bool has_b = tb < 10, has_a = ta < 10;
      const int dt = h ? (has_b ? d.tap_dt[tb < 10 ? tb : 0] : 0) : (has_a ? d.tap_dt[ta < 10 ? ta : 0] : 0);
      const int df = h ? (has_b ? d.tap_df[tb < 10 ? tb : 0] : 0) : (has_a ? d.tap_df[ta < 10 ? ta : 0] : 0);
      const bool has = h ? has_b : has_a;
      const int tin = ps.t + dt, fin = ps.j * d.sf_in + df;
      const bool inb = has && ps.valid && fin >= 0 && fin < d.Fin && tin >= 0 && tin < d.Tin;
      if (inb) in.live |= 1u << s_;
      const uint32_t o0 = inb ? (unsigned)((int64_t)b * d.x0.sb + (int64_t)tin * d.x0.st + (int64_t)fin * d.x0.sf) : 0u;
      const uint32_t o1 = inb ? (unsigned)((int64_t)b * d.x1.sb + (int64_t)tin * d.x1.st + (int64_t)fin * d.x1.sf) : 0u;
      in.raw[s_][0] = bload4(r_x0, o0 << 2, 0);
      in.raw[s_][1] = bload4(r_x0, o0 << 2, (int)(d.x0.sc << 2));
      in.raw[s_][2] = bload4(r_x1, o1 << 2, 0);
      in.raw[s_][3] = bload4(r_x1, o1 << 2, (int)(d.x1.sc << 2));
    }
  };
  auto request_all = [&](const pos_t& ps, in_t& in) {
    if constexpr (IN4) {
      request_in4(ps, in);
    } else {
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) request_tap(ps, in, tap);
    }
  };

  struct acc_t {
    f32x16 L, R, L1, R1;
  };
  // accumulators start from the gather biases (frame 0 of the composed stage 1 has its own)
  auto seed = [&](const pos_t& ps, acc_t& a) {
    const bool f0 = ps.t == 0;
    a.L = ld16(fop + (f0 ? F_BL0 : F_BL) + 4 * h);
    a.R = ld16(fop + (f0 ? F_BR0 : F_BR) + 4 * h);
    if constexpr (DUAL) {
      a.L1 = a.L;
      a.R1 = a.R;
    }
  };
  auto kblock = [&](acc_t& a, const uint4 (&bp)[NP], const int tap, const int q) {
    const int blk = (tap * 2 + q) * BS + lane;
    a.L = mm<NP>(img + CF::o_gL + blk, bp, a.L);
    a.R = mm<NP>(img + CF::o_gR + blk, bp, a.R);
    if constexpr (DUAL) {
      if ((P1MASK >> tap) & 1) {
        const int blk1 = (rank_of(P1MASK, tap) * 2 + q) * BS + lane;
        a.L1 = mm<NP>(img + CF::o_gL1 + blk1, bp, a.L1);
        a.R1 = mm<NP>(img + CF::o_gR1 + blk1, bp, a.R1);
      }
    }
  };
  auto kloop_in4 = [&](acc_t& a, const in_t& in) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ((in.live >> (2 * q + (e >> 2))) & 1u) ? in.raw[2 * q + (e >> 2)][e & 3] : 0.f;
      uint4 bp[NP];
      split8p<NP>(x, bp, s_in);
      const int blk = q * BS + lane;
      a.L = mm<NP>(img + CF::o_gL + blk, bp, a.L);
      a.R = mm<NP>(img + CF::o_gR + blk, bp, a.R);
    }
  };

  // ---- stores.  Plane and skip-half stores are UNCONDITIONAL (lanes without a position write to the dump item B of the
  // target tensor, which is allocated with B + 1 items), so they leave the tail as soon as their values exist and the iteration stays one basic block; only
  // the few fp32 stores of the stages without a dump frame (last decoder stage, encoder stage 5) are predicated.
  const int nFp = d.nx_Fp;
  // dump targets: item B of the tensor (allocated with B + 1 items); all offsets are bytes from the start of the tensor
  const uint32_t item_hp = NXN > 0 ? (uint32_t)(d.nx_hp_sb * 2) : 0u;
  const __amdgpu_buffer_rsrc_t r_nx = make_rsrc(NXN > 0 ? d.nx_hp : nullptr, NXN > 0 ? (uint32_t)(d.B + 1) * item_hp : 0u);
  auto store_planes = [&](const uint4 (&zp)[2][NP], const bool ok, const int t, const int bin) {
    // (frame t, bin) of the next stage's hp: lane half h owns groups g = 2q + h
    const int bi = bin + d.nx_f0;
    const int bpos = d.nx_par ? (bi & 1) * ((nFp + 1) >> 1) + (bi >> 1) : bi;
    const uint32_t o = ok ? (uint32_t)b * item_hp + ((uint32_t)(((t + d.nx_t0) * 4 + h) * (NP * nFp) + bpos) << 4)
                          : (uint32_t)d.B * item_hp + ((uint32_t)(h * (NP * nFp)) << 4);
#if defined(BGLU_DIAG) && defined(BGLU_ONE_STORER)   // timing ablation (results wrong): wave 7 issues the plane stores of all eight waves
    if (wave != 7) return;
#pragma unroll
    for (int k_ = 0; k_ < 8; ++k_)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) bstore16<BGLU_ST_AUX>(zp[q][pl], r_nx, o + (uint32_t)(k_ * 4 * NP * nFp * 16), ((2 * q * NP + pl) * nFp) << 4);
    return;
#endif
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bstore16<BGLU_ST_AUX>(zp[q][pl], r_nx, o, ((2 * q * NP + pl) * nFp) << 4);
  };
  auto plane_off = [&](const bool ok, const int t, const int bin) -> uint32_t {
    const int bi = bin + d.nx_f0;
    const int bpos = d.nx_par ? (bi & 1) * ((nFp + 1) >> 1) + (bi >> 1) : bi;
    return ok ? (uint32_t)b * item_hp + ((uint32_t)(((t + d.nx_t0) * 4 + h) * (NP * nFp) + bpos) << 4)
              : (uint32_t)d.B * item_hp + ((uint32_t)(h * (NP * nFp)) << 4);
  };
  auto store_planes_q = [&](const uint4 (&zq)[NP], const uint32_t o, const int q) {   // K block q of a tile: its NP planes
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) bstore16<BGLU_ST_AUX>(zq[pl], r_nx, o, ((2 * q * NP + pl) * nFp) << 4);
  };
  const __amdgpu_buffer_rsrc_t r_sk0 = make_rsrc(NXN > 1 ? d.nx_out[0] : nullptr, NXN > 1 ? (uint32_t)((d.B + 1) * d.nx_sb[0] * 4) : 0u);
  const __amdgpu_buffer_rsrc_t r_sk1 = make_rsrc(NXN > 2 ? d.nx_out[1] : nullptr, NXN > 2 ? (uint32_t)((d.B + 1) * d.nx_sb[1] * 4) : 0u);
  // fp32 skip halves in groups of four channels, [B + 1 (dump item)][8 groups][T][F][4]: accumulator rows 4q..4q+3 of lane
  // half h are channels 8q + 4h..+3 = group 2q + h, 16 contiguous bytes - four 16-byte stores per tile where
  // [B][32][T][F] took sixteen 4-byte ones (a 4-byte and a 16-byte wave access cost the addresser about the same)
  auto store_skip = [&](const f32x16& z, const int i, const pos_t& ps) {
    const int jp = d.skip_Fh ? (ps.j & 1) * d.skip_Fh + (ps.j >> 1) : ps.j;   // bins split by parity (pdse_bglu_desc.skip_Fh)
    const uint32_t o = (uint32_t)(((int64_t)(ps.valid ? b : d.B) * d.nx_sb[i] + (ps.valid ? (int64_t)ps.t * d.nx_st[i] + (int64_t)jp * d.nx_sf[i] : 0) +
                                   (int64_t)h * d.nx_sc[i]) << 2);
    // NP == 2: HBM holds true values - all sixteen are re-scaled into registers of their own BEFORE the first store.  (Scaled four at
    // a time into one temporary, hipcc re-used that temporary right behind each 16-byte store, and on gfx950 a v_pk_mul_f32 issued
    // directly behind a buffer_store_dwordx4 with a scalar offset overwrote dword 1 of the store's data: LLVM models that hazard
    // for stores without a scalar offset only.  tests/test_isa_guards.py checks every kernel for the pattern.)
    f32x16 zs = z;
    if constexpr (NP == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) zs[r] = z[r] * s_NXout;
      asm volatile("" : "+v"(zs));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bstore16<BGLU_ST_AUX>(make_uint4(__float_as_uint(zs[4 * q]), __float_as_uint(zs[4 * q + 1]), __float_as_uint(zs[4 * q + 2]), __float_as_uint(zs[4 * q + 3])),
               i == 0 ? r_sk0 : r_sk1, o, (int)((2 * q * d.nx_sc[i]) << 2));
  };
  auto stores_masked = [&](const pos_t& ps, const auto& o0, const auto& o1) {
    if (!ps.valid) return;
    const int t = ps.t, j = ps.j;
    const bool two = DUAL && j < d.Fout1;
    float* const po = d.out + ((int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off);
    if constexpr (C2 == 1) {
      if (h == 0) {
        if constexpr (DUAL) {
          const int64_t bin = d.out_sf >> 1;
          if (two && bin == 1) store_pair(po, o0.v, o1.v);
          else {
            po[0] = o0.v;
            if (two) po[bin] = o1.v;
          }
        } else {
          po[0] = o0.v;
        }
      }
    } else if constexpr (NXN == 0) {   // no chained tile: the 64-channel block output itself (encoder stage 5 -> TCM)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 16; ++r) po[(int64_t)(32 * m2 + 4 * h + PDSE_KR(r)) * d.out_sc] = (m2 ? o0.O1 : o0.O0)[r] * (NP == 2 ? s_C2out : 1.0f);
    }
  };

  // ---- the tail of one output phase as separately schedulable pieces (vocabulary of csrc/bglu_sched.inc).
  // PReLU slope <= 1 (validated at launch): PReLU(v) = max(v, slope v).
  const float slope = d.slope;
  struct ph_t {
    f32x16 L, R, mL, mR, G, O0, O1, Z0, Z1, Z2;
    uint4 lp[2][NP], rp[2][NP], gp[2][NP], yp[2][2][NP], zp[2][NP];
    float v;
    uint32_t zo, so;   // spread schedules: byte offsets of the addend loads / the plane stores of this phase
  };
  auto split_half = [&](const f32x16& X, const int s_, uint4 (&p)[NP], const float sc = 1.0f) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = X[8 * s_ + j];
    split8p<NP>(x, p, sc);
  };
  // addend of chained tile 0 (decoders: the encoder's skip half of this conv1; its bias is part of it).  Lanes without a
  // valid position (or without an odd bin) read a position that exists.
  const __amdgpu_buffer_rsrc_t r_add = make_rsrc((DUAL && NXN > 0) ? d.nx_add : nullptr, (DUAL && NXN > 0) ? (uint32_t)(d.B * d.add_sb * 4) : 0u);
  auto zseed = [&](const pos_t& ps, const int ph, f32x16& z) {
    const bool two = ph == 0 || ps.j < d.Fout1;
    const int bin0 = 2 * ps.j + (two ? ph : 0);
    const int bin = d.skip_Fh ? (two ? ph : 0) * d.skip_Fh + ps.j : bin0;   // parity-split bins: phase ph reads a contiguous run
    const uint32_t o = (uint32_t)(((int64_t)b * d.add_sb + (int64_t)ps.t * d.add_st + (int64_t)bin * d.add_sf + (int64_t)h * d.add_sc) << 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // groups of four channels (store_skip's layout): four 16-byte loads
      const uint4 v = bload16(r_add, o, (int)((2 * q * d.add_sc) << 2));
      z[4 * q] = __uint_as_float(v.x), z[4 * q + 1] = __uint_as_float(v.y), z[4 * q + 2] = __uint_as_float(v.z), z[4 * q + 3] = __uint_as_float(v.w);
      if constexpr (NP == 2) z[4 * q] *= s_NXin, z[4 * q + 1] *= s_NXin, z[4 * q + 2] *= s_NXin, z[4 * q + 3] *= s_NXin;   // seeds an accumulator
    }
  };

  auto zoff = [&](const pos_t& ps, const int ph) -> uint32_t {
    const bool two = ph == 0 || ps.j < d.Fout1;
    const int bin0 = 2 * ps.j + (two ? ph : 0);
    const int bin = d.skip_Fh ? (two ? ph : 0) * d.skip_Fh + ps.j : bin0;
    return (uint32_t)(((int64_t)b * d.add_sb + (int64_t)ps.t * d.add_st + (int64_t)bin * d.add_sf + (int64_t)h * d.add_sc) << 2);
  };
  auto zload = [&](const uint32_t o, const int q, f32x16& z) {
    const uint4 v = bload16(r_add, o, (int)((2 * q * d.add_sc) << 2));
    z[4 * q] = __uint_as_float(v.x), z[4 * q + 1] = __uint_as_float(v.y), z[4 * q + 2] = __uint_as_float(v.z), z[4 * q + 3] = __uint_as_float(v.w);
    if constexpr (NP == 2) z[4 * q] *= s_NXin, z[4 * q + 1] *= s_NXin, z[4 * q + 2] *= s_NXin, z[4 * q + 3] *= s_NXin;
  };

#define FRAG(buf, ptr)                                         \
  {                                                            \
    _Pragma("unroll") for (int pl_ = 0; pl_ < NP; ++pl_) fr[buf][pl_] = (ptr)[64 * pl_]; \
  }
#if defined(BGLU_DIAG) && defined(BGLU_NO_MM)   // timing ablation: no matrix instructions in the loop (results wrong)
#define MM(buf, acc_, bp_) acc_[0] += __uint_as_float(fr[buf][0].x ^ bp_[0].x)
#else
#define MM(buf, acc_, bp_) acc_ = mmf<NP>(fr[buf], bp_, acc_)
#endif
#define GL(tap, q) (img + CF::o_gL + ((tap) * 2 + (q)) * BS + lane)
#define GR(tap, q) (img + CF::o_gR + ((tap) * 2 + (q)) * BS + lane)
#define GL1(rk, q) (img + CF::o_gL1 + ((rk) * 2 + (q)) * BS + lane)
#define GR1(rk, q) (img + CF::o_gR1 + ((rk) * 2 + (q)) * BS + lane)
#define GLI(q) (img + CF::o_gL + (q) * BS + lane)
#define GRI(q) (img + CF::o_gR + (q) * BS + lane)
#define LCW(s_) (img + CF::o_lc + (s_) * BS + lane)
#define RCW(s_) (img + CF::o_rc + (s_) * BS + lane)
#define C2W(m2, s_) (img + CF::o_c2 + ((m2) * 2 + (s_)) * BS + lane)
#define NXW(i, m2, s_) (img + CF::o_nx + ((i) * 4 + (m2) * 2 + (s_)) * BS + lane)
#define V_SL(S, s_)                                      \
  {                                                      \
    if ((s_) == 0) S.mL = ld16(fop + F_BLC + 4 * h);     \
    split_half(S.L, s_, S.lp[s_], s_G);                  \
  }
#define V_SR(S, s_)                                      \
  {                                                      \
    if ((s_) == 0) S.mR = ld16(fop + F_BRC + 4 * h);     \
    split_half(S.R, s_, S.rp[s_], s_G);                  \
  }
#define V_SG(S, lo, hi)                                                                                       \
  {                                                                                                           \
    _Pragma("unroll") for (int r = lo; r < hi; ++r) S.G[r] = S.L[r] * sigm2(NP == 2 ? S.mR[r] * s_LC : S.mR[r]) + S.R[r] * sigm2(NP == 2 ? S.mL[r] * s_LC : S.mL[r]); \
  }
#define V_SPG(S, s_)                                          \
  {                                                           \
    if ((s_) == 0) {                                          \
      S.O0 = ld16(fop + F_BC2 + 4 * h);                       \
      S.O1 = ld16(fop + F_BC2 + 32 + 4 * h);                  \
    }                                                         \
    split_half(S.G, s_, S.gp[s_], s_G);                       \
  }
#define V_PR(S, m2)                                                                                        \
  {                                                                                                        \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) S.O##m2[r] = vmax(S.O##m2[r], slope * S.O##m2[r]);       \
  }
#define V_SY(S, m2, s_)                                                     \
  {                                                                         \
    if ((m2) == 0 && (s_) == 0) {                                           \
      if constexpr (!DUAL && NXN > 0) S.Z0 = ld16(fop + F_NXB + 4 * h);     \
      if constexpr (NXN > 1) S.Z1 = ld16(fop + F_NXB + 32 + 4 * h);         \
      if constexpr (NXN > 2) S.Z2 = ld16(fop + F_NXB + 64 + 4 * h);         \
    }                                                                       \
    split_half(S.O##m2, s_, S.yp[m2][s_], s_C2);                            \
  }
#define V_ZSEED(S, ph) zseed(pc, ph, S.Z0);
#define V_SZ(S, s_) split_half(S.Z0, s_, S.zp[s_], s_NX);
#define V_ST0(S, ph)                                                                                   \
  {                                                                                                    \
    if constexpr (DUAL) {                                                                              \
      store_planes(S.zp, pc.valid && ((ph) == 0 || pc.j < d.Fout1), pc.t, 2 * pc.j + (ph));            \
    } else {                                                                                           \
      store_planes(S.zp, pc.valid, pc.t, pc.j);                                                        \
      if (d.nx_row0) { /* uniform: the explicit pad frame of the next encoder stage = the folded bias */ \
        uint4 bp_[2][NP];                                                                              \
        split16p<NP>(ld16(fop + F_NXB + 4 * h), bp_, s_NX);                                            \
        store_planes(bp_, pc.valid && pc.t == 0, -1, pc.j);                                            \
      }                                                                                                \
    }                                                                                                  \
  }
#define V_STSKIP(S, i) store_skip(S.Z##i, (i)-1, pc);
#define V_ZOFF(S, ph) S.zo = zoff(pc, ph);
#define V_ZLD(S, q) zload(S.zo, q, S.Z0);
#define V_ST0Q(S, ph, q)                                                                                \
  {                                                                                                    \
    if ((q) == 0) S.so = DUAL ? plane_off(pc.valid && ((ph) == 0 || pc.j < d.Fout1), pc.t, 2 * pc.j + (ph)) : plane_off(pc.valid, pc.t, pc.j); \
    store_planes_q(S.zp[q], S.so, q);                                                                  \
    if constexpr (!DUAL) {                                                                             \
      if ((q) == 1 && d.nx_row0) {                                                                     \
        uint4 bp_[2][NP];                                                                              \
        split16p<NP>(ld16(fop + F_NXB + 4 * h), bp_, s_NX);                                            \
        store_planes(bp_, pc.valid && pc.t == 0, -1, pc.j);                                            \
      }                                                                                                \
    }                                                                                                  \
  }
#define V_DOT(S, ph)                                                              \
  {                                                                               \
    const f32x16 vw_ = ld16(fop + F_WC2V + 4 * h);                                \
    float part_ = 0.f;                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) part_ += vw_[r] * S.G[r];      \
    const float v_ = part_ + __shfl_xor(part_, 32) + fop[F_BC2];                  \
    S.v = vmax(v_, slope * v_);                                                   \
  }
#define V_KEEP(S)
#define V_KSPLIT(q)                                                                                                               \
  {                                                                                                                               \
    float x_[8];                                                                                                                  \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) x_[e] = ((in.live >> (2 * (q) + (e >> 2))) & 1u) ? in.raw[2 * (q) + (e >> 2)][e & 3] : 0.f; \
    split8p<NP>(x_, kb[q], s_in);                                                                                                 \
  }
#define REQ(tap) request_tap(p_req, in, tap);
#define REQ_CUR(tap) request_tap(pc, in, tap);
#define REQ_IN4() request_in4(p_req, in);
#define REQ1(tap, q, pl) request_one(p_req, in, tap, q, pl);
#define REQ_CUR1(tap, q, pl) request_one(pc, in, tap, q, pl);
#define V_TAKE()          \
  {                       \
    SA.L = acc.L;         \
    SA.R = acc.R;         \
    if constexpr (DUAL) { \
      SB.L = acc.L1;      \
      SB.R = acc.R1;      \
    }                     \
  }
#ifndef BGLU_VPER
#define BGLU_VPER 9   // vector instructions placed behind each MFMA of a slot (tuning: tools/time_bglu.py with PDSE_LIB builds)
#endif
#define FRAG_FENCE __builtin_amdgcn_sched_barrier(0);
#if defined(BGLU_DIAG) && defined(BGLU_NO_V)
#undef V_SL
#undef V_SR
#undef V_SG
#undef V_SPG
#undef V_PR
#undef V_SY
#undef V_SZ
#undef V_ST0
#undef V_STSKIP
#undef V_DOT
#undef V_KSPLIT
#define V_SL(S, s_)
#define V_SR(S, s_)
#define V_SG(S, lo, hi)
#define V_SPG(S, s_)
#define V_PR(S, m2)
#define V_SY(S, m2, s_)
#define V_SZ(S, s_)
#undef V_ZSEED
#define V_ZSEED(S, ph)
#undef V_ZOFF
#undef V_ZLD
#undef V_ST0Q
#define V_ZOFF(S, ph)
#define V_ZLD(S, q)
#define V_ST0Q(S, ph, q)
#define V_ST0(S, ph)
#define V_STSKIP(S, i)
#define V_DOT(S, ph)
#define V_KSPLIT(q)
#endif
#if defined(BGLU_DIAG) && defined(BGLU_NO_ST)   // timing ablation: no plane / skip stores
#undef V_ST0
#undef V_STSKIP
#define V_ST0(S, ph) if (d.slope == 12345.f) { store_planes(S.zp, pc.valid, pc.t, DUAL ? 2 * pc.j + (ph) : pc.j); }
#define V_STSKIP(S, i) if (d.slope == 12345.f) { store_skip(S.Z##i, (i)-1, pc); }
#endif
#if defined(BGLU_DIAG) && defined(BGLU_NO_ZSEED)   // timing ablation: no addend loads
#undef V_ZSEED
#define V_ZSEED(S, ph)
#endif
#if defined(BGLU_DIAG) && defined(BGLU_NO_REQ)
#undef REQ
#undef REQ_CUR
#undef REQ_IN4
#define REQ(tap)
#define REQ_CUR(tap)
#define REQ_IN4()
#endif
#if defined(BGLU_DIAG) && defined(BGLU_NO_FRAG)
#undef FRAG
#define FRAG(buf, ptr)
#endif
#define SLOT_BEGIN {
  // one slot: the LDS reads of the next slot's fragments first, then each MFMA followed by the vector instructions
  // that fit its shadow; nothing crosses the slot boundary
#define SLOT_END(hasm, hasv)                                                       \
  if constexpr (NP == 3) {                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
      __builtin_amdgcn_sched_group_barrier(0x002, BGLU_VPER, 0);                   \
    }                                                                              \
  } else if constexpr (NP == 2) {                                                  \
    _Pragma("unroll") for (int g_ = 0; g_ < 3; ++g_) {                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
      __builtin_amdgcn_sched_group_barrier(0x002, BGLU_VPER, 0);                   \
    }                                                                              \
  } else {                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
  }                                                                                \
  __builtin_amdgcn_sched_barrier(0);                                               \
  DG_SLOT(__COUNTER__)                                                             \
  }
#if defined(BGLU_DIAG) && defined(BGLU_SLOTSTAMP)
  // per-slot shader-clock stamps (s_memtime waits for lgkmcnt(0): the slot's own LDS reads have been consumed by then)
  unsigned dg_slot[96];
#pragma unroll
  for (int k_ = 0; k_ < 96; ++k_) dg_slot[k_] = 0;
  unsigned long long dg_prev = 0;
  unsigned dg_tiles = 0;
#define DG_SLOT(c_)                                                                 \
  if constexpr (!PIPE) {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
    dg_slot[(c_) - dg_base - 1] += (unsigned)(t_ - dg_prev);                        \
    dg_prev = t_;                                                                   \
  }
#else
#define DG_SLOT(c_)
#endif

  // ---- the software pipeline over this workgroup's rounds r_k = blockIdx.x + k gridDim.x
  const int stride = gridDim.x;
  int rd = blockIdx.x;
  if (rd >= nrounds) return;
  if constexpr (PIPE) {
  pos_t pc, pn;
  in_t in;
  acc_t accs[2];   // accumulators of the current / the next tile; the roles swap every iteration (the loop is unrolled twice:
                   // as one body with "acc = accn" at its end hipcc copied all 64 accumulator registers per iteration)
  locate(rd, pc);
  request_all(pc, in);
  seed(pc, accs[0]);
  if constexpr (IN4) {
    kloop_in4(accs[0], in);
  } else {
#pragma unroll
    for (int tap = 0; tap < NT; ++tap)
#pragma unroll
      for (int q = 0; q < 2; ++q) kblock(accs[0], in.pl[tap][q], tap, q);
  }
  locate(rd + stride, pn);
  request_all(pn, in);
  constexpr int SCHED = IN4 ? 5 : (DUAL ? (C2 == 64 ? 1 : 2) : (NXN == 3 ? 3 : 4));
#ifdef BGLU_DIAG
  const unsigned long long dg_c0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned dg_n = 0;
#endif
  // one iteration: tail of the current tile (vector instructions) beside the K loop of the next tile (matrix instructions),
  // the requests of the tile after that behind each consumed tap - in the order of bglu_sched.inc.  Returns true after
  // the last tile of this wave.
  auto iteration = [&](auto PAR) __attribute__((always_inline)) -> bool {
    constexpr int I = decltype(PAR)::value;
    acc_t& acc = accs[I];
    acc_t& accn = accs[1 - I];
    pos_t pnn;
    locate(rd + 2 * stride, pnn);
    const pos_t& p_req = pnn;
    ph_t SA, SB;
    uint4 fr[2][NP];
    uint4 kb[IN4 ? 3 : 1][NP];
    SA.L = acc.L;
    SA.R = acc.R;
    if constexpr (DUAL) {
      SB.L = acc.L1;
      SB.R = acc.R1;
    }
    seed(pn, accn);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SCHED == 1) {
#define BGLU_SCHED 1
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
    } else if constexpr (SCHED == 2) {
#define BGLU_SCHED 2
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
    } else if constexpr (SCHED == 3) {
#define BGLU_SCHED 3
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
    } else if constexpr (SCHED == 4) {
#define BGLU_SCHED 4
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
    } else {
#define BGLU_SCHED 5
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
    }
    if constexpr (C2 == 1 || NXN == 0) stores_masked(pc, SA, SB);
    rd += stride;
#ifdef BGLU_DIAG
    ++dg_n;
    if (rd >= nrounds && lane == 0) {
      atomicAdd(&g_bglu_diag[0], __builtin_amdgcn_s_memtime() - dg_c0);
      atomicAdd(&g_bglu_diag[1], __builtin_amdgcn_s_memrealtime() - dg_r0);
      atomicAdd(&g_bglu_diag[2], (unsigned long long)dg_n);
      atomicAdd(&g_bglu_diag[3], 1ull);
    }
#endif
    pc = pn;
    pn = pnn;
    return rd >= nrounds;
  };
  while (true) {
    if (iteration(std::integral_constant<int, 0>{})) break;
    if (iteration(std::integral_constant<int, 1>{})) break;
  }
    return;
  } else {
    // ---- 8 waves: K loop, then tail, per tile; the first two taps of the next tile are in flight during the tail
    pos_t pc, pn;
    in_t in;
    acc_t acc;
    locate(rd, pc);
    if constexpr (IN4) {
      request_in4(pc, in);
    } else {
      request_tap(pc, in, 0);
      request_tap(pc, in, 1);
    }
    constexpr int SCHED = ((SPR && !IN4) ? 20 : 10) + (IN4 ? 5 : (DUAL ? (C2 == 64 ? 1 : 2) : (NXN == 3 ? 3 : 4)));
    while (true) {
      locate(rd + stride, pn);
      const pos_t& p_req = pn;
      ph_t SA, SB;
      uint4 fr[2][NP];
      uint4 kb[IN4 ? 3 : 1][NP];
      seed(pc, acc);
      __builtin_amdgcn_sched_barrier(0);
#if defined(BGLU_DIAG) && defined(BGLU_SLOTSTAMP)
      dg_prev = __builtin_amdgcn_s_memtime();
      ++dg_tiles;
#endif
      if constexpr (SCHED == 11) {
#define BGLU_SCHED 11
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#include "bglu_sched.inc"
#undef BGLU_SCHED
      } else if constexpr (SCHED == 12) {
#define BGLU_SCHED 12
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#include "bglu_sched.inc"
#undef BGLU_SCHED
      } else if constexpr (SCHED == 13) {
#define BGLU_SCHED 13
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#include "bglu_sched.inc"
#undef BGLU_SCHED
      } else if constexpr (SCHED == 14) {
#define BGLU_SCHED 14
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#include "bglu_sched.inc"
#undef BGLU_SCHED
      } else if constexpr (SCHED == 15) {
#define BGLU_SCHED 15
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#include "bglu_sched.inc"
#undef BGLU_SCHED
      } else if constexpr (SCHED == 21) {
#define BGLU_SCHED 21
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
      } else if constexpr (SCHED == 22) {
#define BGLU_SCHED 22
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
      } else if constexpr (SCHED == 23) {
#define BGLU_SCHED 23
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
      } else {
#define BGLU_SCHED 24
      [[maybe_unused]] constexpr int dg_base = __COUNTER__;
#ifdef BGLU_FORMS
#include "bglu_sched_forms.inc"
#endif
#undef BGLU_SCHED
      }
      if constexpr (C2 == 1 || NXN == 0) stores_masked(pc, SA, SB);
      rd += stride;
      if (rd >= nrounds) break;
      pc = pn;
    }
#if defined(BGLU_DIAG) && defined(BGLU_SLOTSTAMP)
    if (lane == 0) {
#pragma unroll
      for (int k_ = 0; k_ < 96; ++k_)
        if (dg_slot[k_]) atomicAdd(&g_bglu_slots[k_], (unsigned long long)dg_slot[k_]);
      atomicAdd(&g_bglu_slots[96], (unsigned long long)dg_tiles);
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-wave form (round 4).  What round 3 measured on the 8-wave form (profiles/r03_bglu_forms.txt): every unit of the CU busy
// a quarter to a third of the time and the launch taking about the SUM - a wave spends 70 % of a tile queueing at the
// vector-memory pipeline (loads 7k cycles away under load, stores sharing the in-order vmcnt), and with two waves per SIMD
// nothing runs meanwhile.  The remedy named there was more independent waves per CU; what stood in the way was the
// 256-register tail of the slot schedule, which keeps both phases' tails and all taps of a tile live at once to feed one
// wave's matrix pipe.  This form trades that instruction-level overlap for thread-level overlap: 16 waves (four per SIMD,
// <= 128 registers) share the same LDS weight image, and each wave runs a tile strictly in sequence - K loop of phase 0
// (two taps in a register ring), tail of phase 0 with its values consumed as soon as they exist (one half-tile split
// live at a time), then the same for phase 1 - so a wave that waits for memory, for LDS fragments or for an MFMA result
// leaves the SIMD to three others.  Same descriptor, same image, same plane tensors, same roundings per element
// (identical mm / split / gate expressions): results are bit-identical to the 8-wave form.
// ---------------------------------------------------------------------------------------------------------------
constexpr int nth_tap(int m, int k) { return (m & 1) ? (k == 0 ? 0 : 1 + nth_tap(m >> 1, k - 1)) : 1 + nth_tap(m >> 1, k); }

template <int NT, int P1MASK, int C2, int NXN, bool IN4, int NP, int WV>
__global__ __launch_bounds__(64 * WV) void bglu16_kernel(const pdse_bglu_desc d) {
  using CF = bglu_cfg<NT, P1MASK, C2, NXN, IN4, NP, false>;
  constexpr int NB = CF::NB, NB1 = CF::NB1, NT1 = CF::NT1, BS = CF::BS;
  constexpr bool DUAL = CF::DUAL;
  static_assert(IN4 || (NT % 2 == 0 && NT1 % 2 == 0), "the two-slot tap ring assumes an even number of taps per phase");
  extern __shared__ uint4 img[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int P = d.Tout * d.Fout;

  // ---- the LDS image (as bglu_kernel)
  {
    const uint4* const srcs[8] = {reinterpret_cast<const uint4*>(d.w0), reinterpret_cast<const uint4*>(d.w1),
                                  reinterpret_cast<const uint4*>(d.w2), reinterpret_cast<const uint4*>(d.w3),
                                  reinterpret_cast<const uint4*>(d.wlc), reinterpret_cast<const uint4*>(d.wrc),
                                  reinterpret_cast<const uint4*>(d.wc2), reinterpret_cast<const uint4*>(d.nx_w)};
    const int cnt[8] = {NB * NP, NB * NP, NB1 * NP, NB1 * NP, 2 * NP, 2 * NP, C2 == 64 ? 4 * NP : 0, NXN * 4 * NP};
    constexpr int total = CF::o_f >> 6;
    for (int c = __builtin_amdgcn_readfirstlane(wave); c < total; c += WV) {
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
    float* const fo = reinterpret_cast<float*>(img + CF::o_f);
    for (int s = tid; s < BGLU_FLOATS; s += 64 * WV) {
      const int k = s & 31;
      float v = 0.f;
      if (s < 32) v = d.bias0[(int64_t)b * d.bias_sb + k];
      else if (s < 64) v = d.bias1[(int64_t)b * d.bias_sb + k];
      else if (s < 96) v = (d.bias0_t0 ? d.bias0_t0 : d.bias0)[(int64_t)b * d.bias_sb + k];
      else if (s < 128) v = (d.bias1_t0 ? d.bias1_t0 : d.bias1)[(int64_t)b * d.bias_sb + k];
      else if (s < 160) v = d.blc[k];
      else if (s < 192) v = d.brc[k];
      else if (s < 256) v = (s - 192) < C2 ? d.bc2[s - 192] : 0.f;
      else if (s < 352) {
        const int i = (s - 256) >> 5;
        v = (i < NXN && d.nx_bias[i]) ? d.nx_bias[i][(int64_t)b * d.nx_bias_sb[i] + k] : 0.f;
      } else if (s < 384) v = (C2 == 1) ? d.wc2v[k] : 0.f;
      fo[s] = v;
    }
  }
  __syncthreads();
  const float* const fop = reinterpret_cast<const float*>(img + CF::o_f);
  const int ntiles = (P + 31) >> 5;
  const int nrounds = (ntiles + WV - 1) / WV;

  struct pos_t {
    int t, j;
    bool valid;
    uint32_t vin;
  };
  const int Fp = d.hp_Fp;
  auto locate = [&](const int rd, pos_t& ps) {
    const int p = (rd * WV + wave) * 32 + col;
    ps.valid = p < P;
    const int pp = ps.valid ? p : 0;
    ps.t = pp / d.Fout;
    ps.j = pp - ps.t * d.Fout;
    ps.vin = (uint32_t)(ps.t * (4 * NP * Fp) + h * (NP * Fp) + (d.hp_par ? ps.j : ps.j * d.sf_in)) << 4;
  };
  const int Fh = (Fp + 1) >> 1;
  // byte offset of (tap, K block q, plane pl) = a per-tap base + a per-(q, pl) part, added where the load is issued: kept as
  // one loop-invariant scalar per load (36 of them in the six-tap kernels) they overflow the scalar registers into vector lanes
  auto tap_base = [&](const int tap) -> int {
    const int bin0 = d.tap_df[tap] + d.hp_f0;
    int tb = __builtin_amdgcn_readfirstlane((((d.tap_dt[tap] + d.hp_t0) * 4) * (NP * Fp) + (d.hp_par ? (bin0 & 1) * Fh + (bin0 >> 1) : bin0)) << 4);
    asm volatile("" : "+s"(tb));   // opaque: the sums below stay inside the tile loop
    return tb;
  };
  // one tap's planes [q][plane] into a ring slot
  auto request_tap = [&](const pos_t& ps, uint4 (&sl)[2][NP], const int tap) {
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.hp + (int64_t)b * d.hp_sb, (uint32_t)d.hp_Tp * 4u * NP * (uint32_t)Fp * 16u);
    const int tb = tap_base(tap);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) sl[q][pl] = bload16(r_in, ps.vin, tb + (((2 * q) * NP + pl) * Fp << 4));
  };
  struct raw_t {
    float raw[IN4 ? 6 : 1][4];
    unsigned live;
  };
  auto request_in4 = [&](const pos_t& ps, raw_t& in) {
    in.live = 0;
    const __amdgpu_buffer_rsrc_t r_x0 = make_rsrc(d.x0.ptr, 0xfffffffcu), r_x1 = make_rsrc(d.x1.ptr, 0xfffffffcu);
#pragma unroll
    for (int s_ = 0; s_ < 6; ++s_) {
      const int ta = 4 * (s_ >> 1) + (s_ & 1), tb = ta + 2;
      const bool has_b = tb < 10, has_a = ta < 10;
      const int dt = h ? (has_b ? d.tap_dt[tb < 10 ? tb : 0] : 0) : (has_a ? d.tap_dt[ta < 10 ? ta : 0] : 0);
      const int df = h ? (has_b ? d.tap_df[tb < 10 ? tb : 0] : 0) : (has_a ? d.tap_df[ta < 10 ? ta : 0] : 0);
      const bool has = h ? has_b : has_a;
      const int tin = ps.t + dt, fin = ps.j * d.sf_in + df;
      const bool inb = has && ps.valid && fin >= 0 && fin < d.Fin && tin >= 0 && tin < d.Tin;
      if (inb) in.live |= 1u << s_;
      const uint32_t o0 = inb ? (unsigned)((int64_t)b * d.x0.sb + (int64_t)tin * d.x0.st + (int64_t)fin * d.x0.sf) : 0u;
      const uint32_t o1 = inb ? (unsigned)((int64_t)b * d.x1.sb + (int64_t)tin * d.x1.st + (int64_t)fin * d.x1.sf) : 0u;
      in.raw[s_][0] = bload4(r_x0, o0 << 2, 0);
      in.raw[s_][1] = bload4(r_x0, o0 << 2, (int)(d.x0.sc << 2));
      in.raw[s_][2] = bload4(r_x1, o1 << 2, 0);
      in.raw[s_][3] = bload4(r_x1, o1 << 2, (int)(d.x1.sc << 2));
    }
  };
  auto seed = [&](const pos_t& ps, f32x16& L, f32x16& R) {
    const bool f0 = ps.t == 0;
    L = ld16(fop + (f0 ? F_BL0 : F_BL) + 4 * h);
    R = ld16(fop + (f0 ? F_BR0 : F_BR) + 4 * h);
  };

  // ---- stores (as bglu_kernel)
  const int nFp = d.nx_Fp;
  const uint32_t item_hp = NXN > 0 ? (uint32_t)(d.nx_hp_sb * 2) : 0u;
  const __amdgpu_buffer_rsrc_t r_nx = make_rsrc(NXN > 0 ? d.nx_hp : nullptr, NXN > 0 ? (uint32_t)(d.B + 1) * item_hp : 0u);
  auto store_planes = [&](const uint4 (&zp)[2][NP], const bool ok, const int t, const int bin) {
    const int bi = bin + d.nx_f0;
    const int bpos = d.nx_par ? (bi & 1) * ((nFp + 1) >> 1) + (bi >> 1) : bi;
    const uint32_t o = ok ? (uint32_t)b * item_hp + ((uint32_t)(((t + d.nx_t0) * 4 + h) * (NP * nFp) + bpos) << 4)
                          : (uint32_t)d.B * item_hp + ((uint32_t)(h * (NP * nFp)) << 4);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bstore16<BGLU_ST_AUX>(zp[q][pl], r_nx, o, ((2 * q * NP + pl) * nFp) << 4);
  };
  const __amdgpu_buffer_rsrc_t r_sk0 = make_rsrc(NXN > 1 ? d.nx_out[0] : nullptr, NXN > 1 ? (uint32_t)((d.B + 1) * d.nx_sb[0] * 4) : 0u);
  const __amdgpu_buffer_rsrc_t r_sk1 = make_rsrc(NXN > 2 ? d.nx_out[1] : nullptr, NXN > 2 ? (uint32_t)((d.B + 1) * d.nx_sb[1] * 4) : 0u);
  auto store_skip = [&](const f32x16& z, const int i, const pos_t& ps) {
    const int jp = d.skip_Fh ? (ps.j & 1) * d.skip_Fh + (ps.j >> 1) : ps.j;
    // 32-bit arithmetic: pdse_bglu_launch validates (B + 1) * nx_sb * 4 < 2^32
    const uint32_t o = ((uint32_t)(ps.valid ? b : d.B) * (uint32_t)d.nx_sb[i] + (ps.valid ? (uint32_t)ps.t * (uint32_t)d.nx_st[i] + (uint32_t)jp * (uint32_t)d.nx_sf[i] : 0u) +
                        (uint32_t)h * (uint32_t)d.nx_sc[i]) << 2;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bstore16<BGLU_ST_AUX>(make_uint4(__float_as_uint(z[4 * q]), __float_as_uint(z[4 * q + 1]), __float_as_uint(z[4 * q + 2]), __float_as_uint(z[4 * q + 3])),
               i == 0 ? r_sk0 : r_sk1, o, (int)((2 * q * (uint32_t)d.nx_sc[i]) << 2));
  };
  const __amdgpu_buffer_rsrc_t r_add = make_rsrc((DUAL && NXN > 0) ? d.nx_add : nullptr, (DUAL && NXN > 0) ? (uint32_t)(d.B * d.add_sb * 4) : 0u);
  auto zseed = [&](const pos_t& ps, const int ph, f32x16& z) {
    const bool two = ph == 0 || ps.j < d.Fout1;
    const int bin0 = 2 * ps.j + (two ? ph : 0);
    const int bin = d.skip_Fh ? (two ? ph : 0) * d.skip_Fh + ps.j : bin0;
    const uint32_t o = ((uint32_t)b * (uint32_t)d.add_sb + (uint32_t)ps.t * (uint32_t)d.add_st + (uint32_t)bin * (uint32_t)d.add_sf + (uint32_t)h * (uint32_t)d.add_sc) << 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = bload16(r_add, o, (int)((2 * q * (uint32_t)d.add_sc) << 2));
      z[4 * q] = __uint_as_float(v.x), z[4 * q + 1] = __uint_as_float(v.y), z[4 * q + 2] = __uint_as_float(v.z), z[4 * q + 3] = __uint_as_float(v.w);
    }
  };
  auto split_half = [&](const f32x16& X, const int s_, uint4 (&p)[NP]) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = X[8 * s_ + j];
    split8p<NP>(x, p);
  };
  const float slope = d.slope;

  // ---- the tail of one output phase, values consumed as soon as they exist.  Returns the phase's output value for C2 == 1.
  auto tail = [&](const pos_t& pc, auto PH, const f32x16& L, const f32x16& R) __attribute__((always_inline)) -> float {
    constexpr int ph = decltype(PH)::value;
    f32x16 mL = ld16(fop + F_BLC + 4 * h), mR = ld16(fop + F_BRC + 4 * h);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      uint4 p[NP];
      split_half(L, s_, p);
      mL = mm<NP>(img + CF::o_lc + s_ * BS + lane, p, mL);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      uint4 p[NP];
      split_half(R, s_, p);
      mR = mm<NP>(img + CF::o_rc + s_ * BS + lane, p, mR);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 G;
#pragma unroll
    for (int r = 0; r < 16; ++r) G[r] = L[r] * sigm2(mR[r]) + R[r] * sigm2(mL[r]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (C2 == 1) {
      const f32x16 vw_ = ld16(fop + F_WC2V + 4 * h);
      float part_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) part_ += vw_[r] * G[r];
      const float v_ = part_ + __shfl_xor(part_, 32) + fop[F_BC2];
      return vmax(v_, slope * v_);
    } else {
      f32x16 Z0;
      if constexpr (DUAL && NXN > 0) zseed(pc, ph, Z0);           // the addend: in flight during conv2
      f32x16 O0 = ld16(fop + F_BC2 + 4 * h), O1 = ld16(fop + F_BC2 + 32 + 4 * h);
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        uint4 p[NP];
        split_half(G, s_, p);
        O0 = mm<NP>(img + CF::o_c2 + s_ * BS + lane, p, O0);
        __builtin_amdgcn_sched_barrier(0);
        O1 = mm<NP>(img + CF::o_c2 + (2 + s_) * BS + lane, p, O1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        O0[r] = vmax(O0[r], slope * O0[r]);
        O1[r] = vmax(O1[r], slope * O1[r]);
      }
      if constexpr (NXN == 0) {
        // the 64-channel block output itself (encoder stage 5 -> TCM): lane offset + a scalar channel offset per store
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, pc.valid ? 0xfffffffcu : 0u);   // lanes without a position: out of range, dropped
        const uint32_t o = (uint32_t)(((int64_t)b * d.out_sb + (int64_t)pc.t * d.out_st + (int64_t)pc.j * d.out_sf + d.out_off + (int64_t)(4 * h) * d.out_sc) << 2);
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int r = 0; r < 16; ++r) bstore4<0>((m2 ? O1 : O0)[r], r_out, o, (int)(((32 * m2 + PDSE_KR(r)) * d.out_sc) << 2));
      } else {
        uint4 yp[2][2][NP];
        split_half(O0, 0, yp[0][0]);
        split_half(O0, 1, yp[0][1]);
        split_half(O1, 0, yp[1][0]);
        split_half(O1, 1, yp[1][1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NXN; ++i) {
          f32x16 Z;
          if constexpr (DUAL) Z = Z0;                              // DUAL: one chained tile, seeded by the encoder's skip half
          else Z = ld16(fop + F_NXB + 32 * i + 4 * h);
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
              Z = mm<NP>(img + CF::o_nx + (i * 4 + m2 * 2 + s_) * BS + lane, yp[m2][s_], Z);
              __builtin_amdgcn_sched_barrier(0);
            }
          if (i == 0) {
            uint4 zp[2][NP];
            split16p<NP>(Z, zp);
            if constexpr (DUAL) {
              store_planes(zp, pc.valid && (ph == 0 || pc.j < d.Fout1), pc.t, 2 * pc.j + ph);
            } else {
              store_planes(zp, pc.valid, pc.t, pc.j);
              if (d.nx_row0) {   // uniform: the explicit pad frame of the next encoder stage = the folded bias
                uint4 bp_[2][NP];
                split16p<NP>(ld16(fop + F_NXB + 4 * h), bp_);
                store_planes(bp_, pc.valid && pc.t == 0, -1, pc.j);
              }
            }
          } else {
            store_skip(Z, i - 1, pc);
          }
        }
      }
      return 0.f;
    }
  };

  const int stride = gridDim.x;
  int rd = blockIdx.x;
  if (rd >= nrounds) return;
  pos_t pc, pn;
  locate(rd, pc);
  uint4 ring[2][2][NP];   // two taps: [slot][q][plane]
  raw_t raw;
  if constexpr (IN4) request_in4(pc, raw);
  else request_tap(pc, ring[0], 0);
  while (true) {
    locate(rd + stride, pn);
    f32x16 L, R;
    seed(pc, L, R);
    if constexpr (IN4) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ((raw.live >> (2 * q + (e >> 2))) & 1u) ? raw.raw[2 * q + (e >> 2)][e & 3] : 0.f;
        uint4 kb[NP];
        split8p<NP>(x, kb);
        L = mm<NP>(img + CF::o_gL + q * BS + lane, kb, L);
        __builtin_amdgcn_sched_barrier(0);
        R = mm<NP>(img + CF::o_gR + q * BS + lane, kb, R);
        __builtin_amdgcn_sched_barrier(0);
      }
      request_in4(pn, raw);                                        // the next tile's gathers: in flight during this tile's tail
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if (i + 1 < NT) request_tap(pc, ring[(i + 1) & 1], i + 1);
        else if constexpr (DUAL) request_tap(pc, ring[0], nth_tap(P1MASK, 0));   // phase 1's first tap: in flight during tail 0
        else request_tap(pn, ring[0], 0);                                         // the next tile's first tap
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          L = mm<NP>(img + CF::o_gL + (i * 2 + q) * BS + lane, ring[i & 1][q], L);
          __builtin_amdgcn_sched_barrier(0);
          R = mm<NP>(img + CF::o_gR + (i * 2 + q) * BS + lane, ring[i & 1][q], R);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    const float v0 = tail(pc, std::integral_constant<int, 0>{}, L, R);
    float v1 = 0.f;
    if constexpr (DUAL) {
      seed(pc, L, R);
#pragma unroll
      for (int k = 0; k < NT1; ++k) {
        if (k + 1 < NT1) request_tap(pc, ring[(k + 1) & 1], nth_tap(P1MASK, k + 1));
        else request_tap(pn, ring[0], 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          L = mm<NP>(img + CF::o_gL1 + (k * 2 + q) * BS + lane, ring[k & 1][q], L);
          __builtin_amdgcn_sched_barrier(0);
          R = mm<NP>(img + CF::o_gR1 + (k * 2 + q) * BS + lane, ring[k & 1][q], R);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      v1 = tail(pc, std::integral_constant<int, 1>{}, L, R);
    }
    if constexpr (C2 == 1) {
      if (pc.valid && h == 0) {
        float* const po = d.out + ((int64_t)b * d.out_sb + (int64_t)pc.t * d.out_st + (int64_t)pc.j * d.out_sf + d.out_off);
        if constexpr (DUAL) {
          const bool two = pc.j < d.Fout1;
          const int64_t bin = d.out_sf >> 1;
          if (two && bin == 1) store_pair(po, v0, v1);
          else {
            po[0] = v0;
            if (two) po[bin] = v1;
          }
        } else {
          po[0] = v0;
        }
      }
    }
    rd += stride;
    if (rd >= nrounds) break;
    pc = pn;
  }
}

// fp32 [B, 32, T, F] -> planes (the first decoder stage's standalone conv1 output)
template <int NP>
__global__ __launch_bounds__(256) void planes_kernel(const pdse_planes_desc d) {
  const int64_t n = (int64_t)d.B * d.T * d.F * 4;   // one thread per (b, t, g, f): 8 channels
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = (int)(i % d.F);
  const int g = (int)((i / d.F) & 3);
  const int64_t bt = i / ((int64_t)4 * d.F);
  const int t = (int)(bt % d.T), b = (int)(bt / d.T);
  const int q = g >> 1, h = g & 1;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 16 * q + 8 * (e >> 2) + 4 * h + (e & 3);
    x[e] = d.in[(int64_t)b * d.in_sb + (int64_t)c * d.in_sc + (int64_t)t * d.in_st + (int64_t)f * d.in_sf];
  }
  uint4 p[NP];
  split8p<NP>(x, p, NP == 2 ? pow2i(PDSE_F16_ACT_EXP) : 1.0f);
  uint4* const base = reinterpret_cast<uint4*>(d.hp + (int64_t)b * d.hp_sb) + ((int64_t)(t + d.hp_t0) * 4 + g) * (NP * d.hp_Fp) + f + d.hp_f0;
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) base[pl * d.hp_Fp] = p[pl];
}

// The same planes, of a 1x1 convolution computed here (pdse_planes_desc.w; F == 4): a wave owns 8 frames x 4 bins = 32 positions and all
// 32 output channels on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one fmaf chain in k order per output - the
// arithmetic of the gather-GEMM launch this replaces); the accumulator registers of lane half h ARE plane groups h and 2 + h, so the
// split and the 64-byte-per-frame plane stores are those of the block kernels' tails.  (As a vector kernel - one thread per position,
// wave-uniform weights from scalar loads or broadcast LDS reads - the launch took 52-55 us: 256 dependent scalar loads, or an LDS
// array busy with 1024 broadcast reads, per thread.)
template <int NP>
__global__ __launch_bounds__(256) void planes_conv_kernel(const pdse_planes_desc d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
  const int di = blockIdx.z, b = blockIdx.y;
  const int t = (blockIdx.x * 4 + wave) * 8 + (col >> 2), f = col & 3;
  const bool live = t < d.T;
  const int64_t xo = (int64_t)b * d.in_sb + (int64_t)(live ? t : d.T - 1) * d.in_st + (int64_t)f * d.in_sf;
  const float* const w = d.w[di] + col;                 // A operand: lane (row = output channel, k = 2 i + h)
  f32x16 acc;
  if (d.bias[di]) {
    acc = ld16(d.bias[di] + (int64_t)b * d.bias_sb + 4 * h);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
  const int K = d.cin0 + d.cin1;                        // multiples of 32 per source (validated)
  float av[2][16], bv[2][16];
  auto request = [&](const int k0, const int s_) {      // the 32 loads of a chunk of 32 input channels
    const float* const src = (k0 < d.cin0 ? d.in + (int64_t)k0 * d.in_sc : d.in1 + (int64_t)(k0 - d.cin0) * d.in_sc) + xo;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      av[s_][i] = w[(k0 + 2 * i + h) * 32];
      bv[s_][i] = src[(int64_t)(2 * i + h) * d.in_sc];
    }
  };
  request(0, 0);
  for (int k0 = 0; k0 < K; k0 += 64) {                  // two chunks per iteration: the next chunk's loads fly under this chunk's MFMAs
    if (k0 + 32 < K) request(k0 + 32, 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][i], bv[0][i], acc, 0, 0, 0);
    if (k0 + 32 >= K) break;
    if (k0 + 64 < K) request(k0 + 64, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][i], bv[1][i], acc, 0, 0, 0);
  }
  if (!live) return;
  uint4 p[2][NP];
  split16p<NP>(acc, p, NP == 2 ? pow2i(PDSE_F16_ACT_EXP) : 1.0f);
  uint4* const base = reinterpret_cast<uint4*>((di ? d.hp1 : d.hp) + (int64_t)b * d.hp_sb) + ((int64_t)(t + d.hp_t0) * 4 + h) * (NP * d.hp_Fp) + f + d.hp_f0;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) base[(2 * q * NP + pl) * d.hp_Fp] = p[q][pl];
}

template <int NT, int P1MASK, int C2, int NXN, bool IN4, int NP, bool PIPE, bool SPR = false>
int launch_(const pdse_bglu_desc* d, hipStream_t s) {
  using CF = bglu_cfg<NT, P1MASK, C2, NXN, IN4, NP, PIPE>;
  const int P = d->Tout * d->Fout;
  const int rounds = (((P + 31) >> 5) + CF::WV - 1) / CF::WV;
  static const int wgs = PDSE_DIAG_ENV("PDSE_BGLU_WGS") ? atoi(PDSE_DIAG_ENV("PDSE_BGLU_WGS")) : 256;
  int gx = (wgs + d->B - 1) / d->B;
  if (gx > rounds) gx = rounds;
  if (gx < 1) gx = 1;
  if (CF::lds_bytes > 160 * 1024) {
    pdse_set_error("bglu: LDS image too large");
    return 1;
  }
  const void* fn = (const void*)bglu_kernel<NT, P1MASK, C2, NXN, IN4, NP, PIPE, SPR>;
  static unsigned long long attr_mask = 0;   // per instantiation and device
  if (pdse_lds_attr(fn, &attr_mask, "bglu lds attribute")) return 1;
#ifdef BGLU_DIAG
  unsigned long long z[4] = {0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bglu_diag), z, sizeof(z));
  static unsigned long long zs[97];
  for (int k = 0; k < 97; ++k) zs[k] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bglu_slots), zs, sizeof(zs));
#endif
  hipLaunchKernelGGL((bglu_kernel<NT, P1MASK, C2, NXN, IN4, NP, PIPE, SPR>), dim3(gx, d->B, 1), dim3(64 * CF::WV), CF::lds_bytes, s, *d);
#ifdef BGLU_DIAG
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_bglu_diag), sizeof(z));
  (void)hipMemcpyFromSymbol(zs, HIP_SYMBOL(g_bglu_slots), sizeof(zs));
  if (zs[96]) {
    fprintf(stderr, "bglu slots NT %d P1 %d C2 %d NXN %d NP %d (cycles per slot and tile, %llu tiles):", NT, P1MASK, C2, NXN, NP, zs[96]);
    double tot = 0;
    for (int k = 0; k < 96; ++k)
      if (zs[k]) {
        fprintf(stderr, " %d:%.0f", k, (double)zs[k] / (double)zs[96]);
        tot += (double)zs[k] / (double)zs[96];
      }
    fprintf(stderr, " | sum %.0f\n", tot);
  }
  if (z[3]) fprintf(stderr, "bglu diag NT %d P1 %d C2 %d NXN %d NP %d: %.0f cycles per iteration, %.2f iterations per wave, clock %.0f MHz\n", NT, P1MASK, C2, NXN, NP,
                    (double)z[0] / (double)z[2], (double)z[2] / (double)z[3], (double)z[0] / (double)z[1] * 100.0);
#endif
  return pdse_check_launch("bglu");
}

int g_bglu_form = -1;   // pdse_bglu_set_form: -1 = chosen per geometry (default), 0 = 8 waves, 1 = 4 waves pipelined, 2 = 16 waves, 3 = 12 waves

#ifdef BGLU_FORMS
template <int NT, int P1MASK, int C2, int NXN, bool IN4, int NP, int WV>
int launch16_(const pdse_bglu_desc* d, hipStream_t s) {
  using CF = bglu_cfg<NT, P1MASK, C2, NXN, IN4, NP, false>;
  const int P = d->Tout * d->Fout;
  const int rounds = (((P + 31) >> 5) + WV - 1) / WV;
  int gx = (256 + d->B - 1) / d->B;
  if (gx > rounds) gx = rounds;
  if (gx < 1) gx = 1;
  if (CF::lds_bytes > 160 * 1024) {
    pdse_set_error("bglu: LDS image too large");
    return 1;
  }
  const void* fn = (const void*)bglu16_kernel<NT, P1MASK, C2, NXN, IN4, NP, WV>;
  static unsigned long long attr_mask = 0;   // per instantiation and device
  if (pdse_lds_attr(fn, &attr_mask, "bglu16 lds attribute")) return 1;
  hipLaunchKernelGGL((bglu16_kernel<NT, P1MASK, C2, NXN, IN4, NP, WV>), dim3(gx, d->B, 1), dim3(64 * WV), CF::lds_bytes, s, *d);
  return pdse_check_launch("bglu16");
}

#endif

template <int NT, int P1MASK, int C2, int NXN, bool IN4, int NP>
int launch(const pdse_bglu_desc* d, hipStream_t s) {
  // measured (profiles/r03_bglu_forms.txt, B=32, T=401): the 8-wave form is 10-25 % faster than the pipelined 4-wave form
  // on every geometry and for both plane counts: a wave issues at most one instruction per four cycles whatever its type,
  // so two waves per SIMD double the issue rate, which the interleaving inside one wave does not make up for
#ifdef BGLU_FORMS   // the forms that were measured and not kept (profiles/r03_bglu_forms.txt, r04_bglu_forms.txt): diagnostic builds only
  const int form = NP == 2 ? -1 : g_bglu_form;   // the experimental forms know the bf16 plane counts only
  if (form == 1) return launch_<NT, P1MASK, C2, NXN, IN4, NP, true>(d, s);
  if (form == 2) return launch16_<NT, P1MASK, C2, NXN, IN4, NP, 16>(d, s);
  if (form == 3) return launch16_<NT, P1MASK, C2, NXN, IN4, NP, 12>(d, s);
  if (form == 4) return launch_<NT, P1MASK, C2, NXN, IN4, NP, false, !IN4>(d, s);
#endif
  return launch_<NT, P1MASK, C2, NXN, IN4, NP, false>(d, s);
}

template <int NP>
int dispatch(const pdse_bglu_desc* d, hipStream_t s) {
  if (d->x0.ptr) {   // encoder stage 1
    if (d->nx_n == 3) return launch<10, 0, 64, 3, true, NP>(d, s);
  } else if (d->p1mask) {
    if (d->ntaps == 4 && d->p1mask == 5 && d->C2 == 64 && d->nx_n == 1) return launch<4, 5, 64, 1, false, NP>(d, s);
    if (d->ntaps == 6 && d->p1mask == 27 && d->C2 == 1 && d->nx_n == 0) return launch<6, 27, 1, 0, false, NP>(d, s);
  } else {
    if (d->ntaps == 6 && d->C2 == 64 && d->nx_n == 3) return launch<6, 0, 64, 3, false, NP>(d, s);
    if (d->ntaps == 6 && d->C2 == 64 && d->nx_n == 0) return launch<6, 0, 64, 0, false, NP>(d, s);
  }
  pdse_set_error("bglu: no instantiation for this (taps, phases, C2, chained tiles)");
  return 1;
}

}   // namespace

int pdse_bglu_launch(const pdse_bglu_desc* d, hipStream_t s) {
  if (!d || !d->w0 || !d->w1 || !d->wlc || !d->wrc || !d->bias0 || !d->bias1 || !d->blc || !d->brc || !d->bc2) {
    pdse_set_error("bglu: null operand");
    return 1;
  }
  if (!(d->slope <= 1.0f)) {
    pdse_set_error("bglu: PReLU slope must be <= 1 (max form); use the korder 2 kernels otherwise");
    return 1;
  }
  if (d->B < 1 || d->Tout < 1 || d->Fout < 1 || !(d->np == 3 || d->np == 2 || d->np == 1) || !(d->C2 == 64 || d->C2 == 1) || d->nx_n < 0 ||
      d->nx_n > 3 || d->ntaps < 1 || d->ntaps > 10) {
    pdse_set_error("bglu: bad geometry (np in {1, 2, 3}, C2 in {64, 1}, nx_n <= 3, ntaps <= 10)");
    return 1;
  }
  if (d->p1mask && d->Fout1 > d->Fout) {
    pdse_set_error("bglu: Fout1 > Fout");
    return 1;
  }
  if ((d->C2 == 64 && !d->wc2) || (d->C2 == 1 && (!d->wc2v || !d->out)) || (d->C2 == 64 && d->nx_n == 0 && !d->out) || (d->nx_n > 0 && (!d->nx_w || !d->nx_hp)) ||
      (d->nx_n > 1 && !d->nx_out[0]) || (d->nx_n > 2 && !d->nx_out[1]) || (d->p1mask && (!d->w2 || !d->w3))) {
    pdse_set_error("bglu: an operand of the selected form is missing");
    return 1;
  }
  if (!d->x0.ptr) {
    // plane input: every tap must stay inside the margins of hp, and the lane offsets must fit 32 bits
    if (!d->hp || d->hp_Fp < 1) {
      pdse_set_error("bglu: null hp");
      return 1;
    }
    const long long frame = 4ll * d->np * d->hp_Fp;
    if ((long long)d->hp_Tp * frame * 16 >= (1ll << 32)) {
      pdse_set_error("bglu: hp item exceeds 32-bit lane offsets");
      return 1;
    }
    if (d->hp_par && (d->hp_par != 1 || d->sf_in != 2)) {
      pdse_set_error("bglu: parity-split input planes (hp_par = 1) are for stride-2 taps (sf_in == 2)");
      return 1;
    }
    for (int i = 0; i < d->ntaps; ++i) {
      const int tlo = d->tap_dt[i] + d->hp_t0, thi = d->Tout - 1 + d->tap_dt[i] + d->hp_t0;
      const int flo = d->tap_df[i] + d->hp_f0, fhi = (d->Fout - 1) * d->sf_in + d->tap_df[i] + d->hp_f0;
      if (tlo < 0 || thi >= d->hp_Tp || flo < 0 || fhi >= d->hp_Fp) {
        pdse_set_error("bglu: a tap leaves the margins of hp");
        return 1;
      }
    }
  } else if (!d->x1.ptr || d->ntaps != 10 || d->p1mask || (long long)d->B * d->x0.sb * 4 >= (1ll << 32) || (long long)d->B * d->x1.sb * 4 >= (1ll << 32)) {
    pdse_set_error("bglu: encoder stage 1 needs two fp32 sources and the ten (2,5) taps");
    return 1;
  }
  if (d->nx_n > 0) {
    if (d->nx_items < d->B + 1) {
      pdse_set_error("bglu: nx_hp / nx_out need B + 1 allocated items (nx_items): item B is the dump target of lanes beyond the last position");
      return 1;
    }
    const long long frame = 4ll * d->np * d->nx_Fp;
    int maxbin = d->Fout - 1;
    if (d->p1mask) maxbin = 2 * (d->Fout - 1) > 2 * (d->Fout1 - 1) + 1 ? 2 * (d->Fout - 1) : 2 * (d->Fout1 - 1) + 1;
    if (d->nx_t0 + d->Tout > d->nx_Tp || d->nx_t0 - (d->nx_row0 ? 1 : 0) < 0 || d->nx_f0 < 0 || maxbin + d->nx_f0 >= d->nx_Fp ||
        (long long)d->nx_Tp * frame >= (1ll << 31) || (d->p1mask != 0) != (d->nx_add != nullptr) ||
        (long long)(d->B + 1) * d->nx_hp_sb * 2 >= (1ll << 32)) {
      pdse_set_error("bglu: the chained tile does not fit its hp tensor");
      return 1;
    }
    if (d->nx_add && (long long)d->B * d->add_sb * 4 >= (1ll << 32)) {
      pdse_set_error("bglu: addend exceeds 32-bit lane offsets");
      return 1;
    }
    auto c4_ok = [](const void* p, const int64_t sb, const int64_t sc, const int64_t st, const int64_t sf) {
      return ((sb | sc | st | sf) & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    };
    if (d->nx_add && !c4_ok(d->nx_add, d->add_sb, d->add_sc, d->add_st, d->add_sf)) {
      pdse_set_error("bglu: the addend is kept in groups of four channels: strides in multiples of 4 floats, 16-byte aligned base");
      return 1;
    }
    for (int i = 0; i + 1 < d->nx_n; ++i) {
      if ((long long)(d->B + 1) * d->nx_sb[i] * 4 >= (1ll << 32)) {
        pdse_set_error("bglu: skip tensor exceeds 32-bit lane offsets");
        return 1;
      }
      if (!c4_ok(d->nx_out[i], d->nx_sb[i], d->nx_sc[i], d->nx_st[i], d->nx_sf[i])) {
        pdse_set_error("bglu: skip tensors are kept in groups of four channels: strides in multiples of 4 floats, 16-byte aligned base");
        return 1;
      }
    }
  }
  if (d->np == 2) {
    for (int i = 0; i < 4; ++i)
      if (d->qexp[i] < -40 || d->qexp[i] > 40) {
        pdse_set_error("bglu: qexp (f16x2 weight exponents) out of range");
        return 1;
      }
    return dispatch<2>(d, s);
  }
  return d->np == 3 ? dispatch<3>(d, s) : dispatch<1>(d, s);
}

int pdse_bglu_set_form(int form) {
#ifdef BGLU_FORMS
  const int top = 4;
#else
  const int top = 0;   // the product library holds the 8-wave form only
#endif
  if (form < -1 || form > top) return -2;
  const int prev = g_bglu_form;
  g_bglu_form = form;
  return prev;
}

int pdse_planes_launch(const pdse_planes_desc* d, hipStream_t s) {
  if (!d || !d->in || !d->hp || d->B < 1 || d->T < 1 || d->F < 1 || !(d->np == 3 || d->np == 2 || d->np == 1)) {
    pdse_set_error("planes: bad argument");
    return 1;
  }
  if (d->hp_t0 < 0 || d->hp_f0 < 0 || d->hp_t0 + d->T > d->hp_Tp || d->hp_f0 + d->F > d->hp_Fp) {
    pdse_set_error("planes: the tensor does not fit hp");
    return 1;
  }
  if (d->w[0]) {   // the planes of a 1x1 convolution computed here
    if (d->nd < 1 || d->nd > 2 || d->cin0 < 32 || d->cin1 < 0 || (d->cin0 & 31) || (d->cin1 & 31) || d->cin0 + d->cin1 > 128 || (d->cin1 > 0 && !d->in1) || (d->nd == 2 && (!d->w[1] || !d->hp1)) ||
        d->B > 65535) {
      pdse_set_error("planes: the convolution form takes 1 or 2 weight sets, up to 128 input channels (multiples of 32) over one or two sources");
      return 1;
    }
    if (d->F != 4 || (reinterpret_cast<uintptr_t>(d->bias[0]) & 15) || (reinterpret_cast<uintptr_t>(d->bias[1]) & 15) || (d->bias_sb & 3)) {
      pdse_set_error("planes: the convolution form is built for F == 4 (decoder stage 5) and 16-byte aligned bias rows");
      return 1;
    }
    const dim3 cgrid((unsigned)((d->T + 31) / 32), (unsigned)d->B, (unsigned)d->nd), cblock(256);
    if (d->np == 3) hipLaunchKernelGGL(planes_conv_kernel<3>, cgrid, cblock, 0, s, *d);
    else if (d->np == 2) hipLaunchKernelGGL(planes_conv_kernel<2>, cgrid, cblock, 0, s, *d);
    else hipLaunchKernelGGL(planes_conv_kernel<1>, cgrid, cblock, 0, s, *d);
    return pdse_check_launch("planes (conv)");
  }
  const int64_t n = (int64_t)d->B * d->T * d->F * 4;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (d->np == 3) hipLaunchKernelGGL(planes_kernel<3>, grid, block, 0, s, *d);
  else if (d->np == 2) hipLaunchKernelGGL(planes_kernel<2>, grid, block, 0, s, *d);
  else hipLaunchKernelGGL(planes_kernel<1>, grid, block, 0, s, *d);
  return pdse_check_launch("planes");
}
