// gconv_common.h — helpers shared by the two gather-GEMM kernels: accumulator row map,
// activations and the three epilogues (LINEAR / GLU / BIGLU chain).
#ifndef PDSE_GCONV_COMMON_H
#define PDSE_GCONV_COMMON_H
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "pdse.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// row of accumulator register r on lane-half h (32x32 C/D layout, cdna_hip_programming.md §3)
__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Epilogue transcendentals use the hardware exp2/rcp (v_exp_f32 / v_rcp_f32, ~1 ulp each):
// the ablation of profiles/r01_ablation.txt showed the epilogue, not the MFMA loop, bounding
// the block kernels — accurate expf + IEEE division cost ~35 VALU instructions per sigmoid.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }

template <int ACT>
__device__ __forceinline__ float act_c(float y, float slope) {
  if constexpr (ACT == PDSE_ACT_PRELU) return y > 0.f ? y : slope * y;
  else if constexpr (ACT == PDSE_ACT_ELU) return y > 0.f ? y : fast_exp(y) - 1.0f;
  else if constexpr (ACT == PDSE_ACT_SIGMOID) return sigmoid_f(y);
  else return y;
}

// stand-ins for absent per-channel operands (bias, folded BatchNorm): keeps the epilogue free of pointer tests
__device__ const float pdse_zeros[32] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                         0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
__device__ const float pdse_ones[32] __attribute__((aligned(16))) = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                        1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};

__device__ __forceinline__ float act_f(float y, int act, float slope) {
  switch (act) {
    case PDSE_ACT_PRELU: return y > 0.f ? y : slope * y;
    case PDSE_ACT_ELU: return y > 0.f ? y : fast_exp(y) - 1.0f;
    case PDSE_ACT_SIGMOID: return sigmoid_f(y);
    default: return y;
  }
}

// register r of a 32x32 accumulator holds row KR(r) + 4*h
#define PDSE_KR(r) (((r) & 3) + 8 * ((r) >> 2))

// The 16 per-channel operands of a lane's accumulator rows, KR(r) + 4*h for r = 0..15, are four runs of four
// consecutive floats (KR(4q + i) = 8q + i): p must already include the 4*h (and tile) offset and be 16-byte aligned.
// One ds_read_b128 / global_load_dwordx4 per run instead of sixteen dependent 4-byte reads - with the scalar form
// the BIGLU tail issued ~100 LDS reads per phase, each waited for individually (~12k cycles per tile).
__device__ __forceinline__ f32x16 ld16(const float* p) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 x = *reinterpret_cast<const float4*>(p + 8 * q);
    v[4 * q] = x.x;
    v[4 * q + 1] = x.y;
    v[4 * q + 2] = x.z;
    v[4 * q + 3] = x.w;
  }
  return v;
}

// Channel-indexed vectors (bias, folded BN) and the output are addressed as
//   base(lane half, tile) + KR(r) * stride
// with KR(r) a compile-time constant and the stride wave-uniform, so the per-element address
// work is one scalar multiply + one vector add instead of an integer divide/modulo pair.
// CR1: out_cr == 1 (every layer but the STFT, whose channel index splits into (re/im, bin)).
// Small per-layer operands of the BIGLU tail.  The pipelined kernel copies them into LDS once
// per workgroup at launch (they are the same for every wave): fetched from global memory inside
// the tail, each of the ~70 fragment loads exposed its full latency in front of a dependent MFMA.
struct pdse_tail {
  const float* wlc;
  const float* wrc;
  const float* wc2;
  const float* bl;
  const float* br;
  const float* blc;
  const float* brc;
  const float* bc2;
  const float* ps;  // nullptr: no folded BatchNorm
  const float* pt;
  const float* nxw; // chained next-stage 1x1 tiles [nx_n][2][16][64]
  const float* nxb; // their biases of this workgroup's batch item [nx_n][32]
  const float* bl0; // bl / br for output frame 0 (pdse.h: bias0_t0 / bias1_t0; copies of bl / br when absent)
  const float* br0;
};
// LDS image: [wlc 1024][wrc 1024][wc2 2048][bl 32][br 32][blc 32][brc 32][bc2 64][ps 64][pt 64][nxw 3*2048][nxb 3*32]
#define PDSE_TAIL_NXW (1024 + 1024 + 2048 + 4 * 32 + 3 * 64)
#define PDSE_TAIL_NXB (PDSE_TAIL_NXW + 3 * 2048)
#define PDSE_TAIL_B0 (PDSE_TAIL_NXB + 3 * 32)
#define PDSE_TAIL_FLOATS (PDSE_TAIL_B0 + 2 * 32)

__device__ __forceinline__ pdse_tail tail_from_desc(const pdse_gconv_desc& d) {
  return pdse_tail{d.wlc, d.wrc, d.wc2, d.bias0, d.bias1, d.blc, d.brc, d.bc2, d.post_scale, d.post_shift, nullptr, nullptr,
                   d.bias0, d.bias1};
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 operands (csrc/gconv3.hip, packing.split_bf16x3): an fp32 value is the exact sum of three bf16 numbers
// (8 + 8 + 8 significand bits, by truncation), and a product is evaluated as its six leading cross terms on the bf16
// matrix cores with fp32 accumulation.  Fragments are 8 bf16 per lane (one uint4) per plane.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, const f32x16 c) {
  union { uint4 u; bf16x8 v; } A, B;
  A.u = a;
  B.u = b;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, c, 0, 0, 0);
}

// x[0..7] -> three planes of 8 bf16 (element j in the low / high half of dword j >> 1), x == p1 + p2 + p3 exactly.
// 11 VALU instructions per pair of values: two masked residuals each, and one v_perm_b32 per plane picks the high
// halves of the pair (the bf16 of a truncation IS the high half, masked or not); 15 per pair with shift / or packing.
// Two thirds of the VALU instructions of a block kernel's tail are this function, yet a quarter fewer of them moved
// the tail by 5 % only (22.4k -> 21.1k cycles per round, PDSE_S3_TRACE): the tail is a serial chain of split ->
// fragment read -> six dependent MFMAs, not an issue-bound stream.
__device__ __forceinline__ void split8(const float (&x)[8], uint4& p1, uint4& p2, uint4& p3) {
#ifdef PDSE_ABLATE_SPLIT   // diagnostic build only (wrong results): the cost of the splits
  p1 = make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
  p2 = make_uint4(__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7]));
  p3 = make_uint4(p1.x ^ p2.x, p1.y ^ p2.y, p1.z ^ p2.z, p1.w ^ p2.w);
  return;
#endif
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

// ---------------------------------------------------------------------------------------------------------------
// Split-f16 operands (round 4; include/pdse.h "f16x2"): an fp32 value x, scaled by a power of two into the fp16 range,
// is carried as hi = RN16(x), lo = RN16(x - hi): 11 + 11 significand bits and the sign of lo, |x - hi - lo| <= 2^-23 |x|
// (half an fp32 ulp at worst) while lo is a normal fp16, i.e. for |x| >= 2^-2 in scaled units; below that the absolute
// error is <= 2^-25 of the scaled unit.  A product is its three leading cross terms (a1 b1, a1 b2, a2 b1; what is dropped
// is a2 b2 <= 2^-22 |a b|) on the f16 matrix cores with fp32 accumulation: half the matrix instructions and two thirds
// of the operand bytes of the three-plane bf16 split.  Every product of two fp16 numbers is exact in fp32.
// Scaling is by exact powers of two: activations by 2^PDSE_F16_ACT_EXP, every weight matrix by its own 2^q chosen on
// the host (packing.f16_wexp); accumulators therefore hold (true value) * 2^(P + q) and are re-scaled where they are
// split for the next contraction (one multiply, merged with the split's own).
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_f16(const uint4& a, const uint4& b, const f32x16 c) {
  union { uint4 u; f16x8 v; } A, B;
  A.u = a;
  B.u = b;
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(A.v, B.v, c, 0, 0, 0);
}

// 2^e as a float (|e| < 127)
__device__ __forceinline__ float pow2i(const int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }

// Conversions are IEEE (MODE.FP16_OVFL stays 0): a scaled value beyond the fp16 range becomes an infinity in its planes and a
// non-finite number in every sum it enters - it surfaces at the end of the path (SamplerPipeline.check) instead of passing for a value.

// x[0..7] * sc -> two planes of 8 fp16 (element j in the low / high half of dword j >> 1): v_pk_mul, v_cvt_pk_f16_f32,
// two v_cvt_f32_f16, v_pk_add, v_cvt_pk_f16_f32 per pair of values (the bf16 three-way split takes eleven)
__device__ __forceinline__ void split8h(const float (&x)[8], const float sc, uint4& p1, uint4& p2) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  uint32_t q1[4], q2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 t = {x[2 * i] * sc, x[2 * i + 1] * sc};
    union { f16x2 h; uint32_t u; } hi, lo;
    hi.h = __builtin_convertvector(t, f16x2);
    const f32x2 r = t - __builtin_convertvector(hi.h, f32x2);
    lo.h = __builtin_convertvector(r, f16x2);
    q1[i] = hi.u;
    q2[i] = lo.u;
  }
  p1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
  p2 = make_uint4(q2[0], q2[1], q2[2], q2[3]);
}

// acc += A B with A given as three fragment planes at w[0], w[64], w[128] (uint4 units, this lane's entry) and B as
// the three planes of an exact split: smallest terms first
__device__ __forceinline__ f32x16 mfma6(const uint4* w, const uint4& b1, const uint4& b2, const uint4& b3, f32x16 acc) {
  const uint4 a1 = w[0], a2 = w[64], a3 = w[128];
  acc = mfma_bf16(a1, b3, acc);
  acc = mfma_bf16(a3, b1, acc);
  acc = mfma_bf16(a2, b2, acc);
  acc = mfma_bf16(a1, b2, acc);
  acc = mfma_bf16(a2, b1, acc);
  acc = mfma_bf16(a1, b1, acc);
  return acc;
}

// Y = W X for a 32-channel accumulator tile X used as the B operand (two k-blocks of 8 registers each);
// w: LDS fragments [2 blocks][3 planes][64 lanes] of this output tile, already offset by the lane
__device__ __forceinline__ f32x16 chain_s3(const uint4* w, const f32x16& X, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = X[8 * s + j];
    uint4 b1, b2, b3;
    split8(x, b1, b2, b3);
    acc = mfma6(w + s * 192, b1, b2, b3, acc);
  }
  return acc;
}

// The same with the operand split once and used for several output tiles: p[k-block][plane]
__device__ __forceinline__ void split16(const f32x16& X, uint4 (&p)[2][3]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = X[8 * s + j];
    split8(x, p[s][0], p[s][1], p[s][2]);
  }
}

__device__ __forceinline__ f32x16 chain_s3p(const uint4* w, const uint4 (&p)[2][3], f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) acc = mfma6(w + s * 192, p[s][0], p[s][1], p[s][2], acc);
  return acc;
}

// LDS image of the split-bf16 BIGLU tail (gconv3.hip fills it): fragment areas in uint4 units, then the float operands
struct pdse_tail_s3 {
  const uint4* wlc;   // [2][3][64]
  const uint4* wrc;
  const uint4* wc2;   // [2 tiles][2][3][64]   (C2 == 64)
  const uint4* nxw;   // [nx_n][4][3][64]
  const float* wc2v;  // [32]                   (C2 == 1)
  const float* bl;
  const float* br;
  const float* blc;
  const float* brc;
  const float* bc2;
  const float* ps;
  const float* pt;
  const float* nxb;
  const float* bl0;
  const float* br0;
};

template <int EPI, int MT, bool CR1>
__device__ __forceinline__ void gconv_epilogue_impl(const pdse_gconv_desc& d, const pdse_tail& tl, f32x16* acc0,
                                                    f32x16* acc1, const int b, const int t, const int j,
                                                    const bool pvalid, const int lane, const int h, const int mt0,
                                                    const int mtiles, const int64_t extra_off) {
  float* const obase =
      d.out + ((int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off + extra_off);
  const int64_t cstep = d.out_sc_hi;
  auto chan_off = [&](const int co) -> int64_t {
    if constexpr (CR1) return (int64_t)co * cstep;
    else return (int64_t)(co / d.out_cr) * d.out_sc_hi + (int64_t)(co % d.out_cr) * d.out_sc_lo;
  };

  if constexpr (EPI == PDSE_EPI_LINEAR || EPI == PDSE_EPI_GLU) {
    // Everything that is uniform over the launch is resolved OUTSIDE the element loops: the activation and the
    // residual are dispatched once (epilogue_tiles<ACT, RES>), a missing bias / BatchNorm reads zeros / ones from a
    // constant block, and a full channel tile stores without per-element guards.  The straightforward form
    // (`switch (d.act)`, `if (bias)`, `if (co < Cout)` per element) compiled to ~20 branches per output element.
    const float* rbase = d.resid ? d.resid + (obase - d.out) : nullptr;
    // 16-byte stores need every stride in multiples of 4 floats and an aligned base (else: the element-wise path below)
    const bool blocked8 = !CR1 && d.out_cr == 8 && d.out_sc_lo == 1 &&
                          !((d.out_sb | d.out_sc_hi | d.out_st | d.out_sf | d.out_off | extra_off) & 3) &&
                          (reinterpret_cast<uintptr_t>(d.out) & 15) == 0;
    auto tiles = [&](auto act_tag, auto res_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      constexpr bool RES = decltype(res_tag)::value;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (mt0 + m >= mtiles) continue;
        const int c0 = 32 * (mt0 + m) + 4 * h;
        const float* pb0 = d.bias0 ? d.bias0 + (int64_t)b * d.bias0_sb + c0 : pdse_zeros + 4 * h;
        const float* pb1 = (EPI == PDSE_EPI_GLU && d.bias1) ? d.bias1 + (int64_t)b * d.bias1_sb + c0 : pdse_zeros + 4 * h;
        const float* ps = d.post_scale ? d.post_scale + c0 : pdse_ones + 4 * h;
        const float* pt = d.post_scale ? d.post_shift + c0 : pdse_zeros + 4 * h;
        auto value = [&](const int r) {
          float y = acc0[m][r] + pb0[PDSE_KR(r)];
          if constexpr (EPI == PDSE_EPI_GLU) y = y * sigmoid_f(acc1[m][r] + pb1[PDSE_KR(r)]);
          y = y * ps[PDSE_KR(r)] + pt[PDSE_KR(r)];
          return act_c<ACT>(y, d.act_slope);
        };
        if (!CR1 && !RES && blocked8 && 32 * (mt0 + m) + 32 <= d.Cout) {
          // channel-blocked output [B][C/8][T][F][8] (out_cr = 8, out_sc_lo = 1; what pdse_src.blk = 8 reads): a lane's rows
          // 4q..4q+3 are channels 8 (4 tile + q) + 4h..+3 = 16 contiguous bytes of block 4 tile + q
          if (pvalid) {
            float* po = obase + (int64_t)(4 * (mt0 + m)) * cstep + 4 * h;
            f32x16 y = acc0[m];
            if (d.bias0) y += ld16(pb0);
            if constexpr (EPI == PDSE_EPI_GLU) {
              f32x16 g = acc1[m];
              if (d.bias1) g += ld16(pb1);
#pragma unroll
              for (int r = 0; r < 16; ++r) y[r] *= sigmoid_f(g[r]);
            }
            if (d.post_scale) y = y * ld16(ps) + ld16(pt);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v4;
#pragma unroll
              for (int i = 0; i < 4; ++i) v4[i] = act_c<ACT>(y[4 * q + i], d.act_slope);
              *reinterpret_cast<f32x4*>(po + (int64_t)q * cstep) = v4;
            }
          }
        } else if (32 * (mt0 + m) + 32 <= d.Cout && CR1) {   // full tile, plain channel stride: one predicated region
          if (pvalid) {
            float* po = obase + (int64_t)c0 * cstep;
            const float* pr = RES ? rbase + (int64_t)c0 * cstep : nullptr;
            if constexpr (EPI == PDSE_EPI_GLU) {
              // per-channel operands as 16-byte reads folded into the tile in place (measured: GCRN's gated
              // convolutions 63/93/157/256 -> 54/86/146/243 us; the same form on the 4-tile LINEAR launches of the LSTM
              // input projections cost 425 -> 500 us, so LINEAR keeps the element-wise reads)
              f32x16 y = acc0[m], g = acc1[m];
              if (d.bias0) y += ld16(pb0);
              if (d.bias1) g += ld16(pb1);
#pragma unroll
              for (int r = 0; r < 16; ++r) y[r] *= sigmoid_f(g[r]);
              if (d.post_scale) y = y * ld16(ps) + ld16(pt);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float v = act_c<ACT>(y[r], d.act_slope);
                if constexpr (RES) v += pr[(int64_t)PDSE_KR(r) * cstep];
                po[(int64_t)PDSE_KR(r) * cstep] = v;
              }
            } else if (!RES && cstep == 1 && ((uintptr_t)po & 15) == 0) {
              // channels innermost (the LSTM gate buffer): a lane's rows 4q..4q+3 are 16 contiguous bytes - four stores
              // per tile instead of sixteen, each still touching one line per lane
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f32x4 v4;
#pragma unroll
                for (int i = 0; i < 4; ++i) v4[i] = value(4 * q + i);
                *reinterpret_cast<f32x4*>(po + 8 * q) = v4;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float y = value(r);
                if constexpr (RES) y += pr[(int64_t)PDSE_KR(r) * cstep];
                po[(int64_t)PDSE_KR(r) * cstep] = y;
              }
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = c0 + PDSE_KR(r);
            if (pvalid && co < d.Cout) {
              float y = value(r);
              const int64_t o = CR1 ? (int64_t)c0 * cstep + (int64_t)PDSE_KR(r) * cstep : chan_off(co);
              if constexpr (RES) y += rbase[o];
              obase[o] = y;
            }
          }
        }
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (rbase) {   // a residual input never comes with an activation (validated at launch)
      tiles(std::integral_constant<int, PDSE_ACT_NONE>{}, T{});
    } else {
      switch (d.act) {
        case PDSE_ACT_PRELU: tiles(std::integral_constant<int, PDSE_ACT_PRELU>{}, F{}); break;
        case PDSE_ACT_ELU: tiles(std::integral_constant<int, PDSE_ACT_ELU>{}, F{}); break;
        case PDSE_ACT_SIGMOID: tiles(std::integral_constant<int, PDSE_ACT_SIGMOID>{}, F{}); break;
        default: tiles(std::integral_constant<int, PDSE_ACT_NONE>{}, F{}); break;
      }
    }
  } else {
    // BiConvGLU / BiConvTransGLU tail, register to register (model/diff3.py:316-326, :345-351)
    f32x16 L = acc0[0], R = acc1[0];
    const float slope = d.act == PDSE_ACT_PRELU ? d.act_slope : 1.0f;   // PReLU or nothing (validated at launch)
    L += ld16(tl.bl + 4 * h);
    R += ld16(tl.br + 4 * h);
    const f32x16 vblc = ld16(tl.blc + 4 * h), vbrc = ld16(tl.brc + 4 * h);
    f32x16 mL, mR;
#pragma unroll
    for (int r = 0; r < 16; ++r) mL[r] = mR[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mL = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.wlc[r * 64 + lane], L[r], mL, 0, 0, 0);
      mR = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.wrc[r * 64 + lane], R[r], mR, 0, 0, 0);
    }
    f32x16 G;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ml = sigmoid_f(mL[r] + vblc[r]);
      const float mr = sigmoid_f(mR[r] + vbrc[r]);
      G[r] = L[r] * mr + R[r] * ml;
    }
    if (d.C2 == 1) {
      const f32x16 vw = ld16(tl.wc2 + 4 * h);
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) part += vw[r] * G[r];
      float y = part + __shfl_xor(part, 32) + tl.bc2[0];
      if (tl.ps) y = y * tl.ps[0] + tl.pt[0];
      y = y > 0.f ? y : slope * y;
      if (pvalid && h == 0) obase[0] = y;
    } else {
      const int tiles2 = (d.C2 + 31) >> 5;
      for (int m2 = 0; m2 < tiles2; ++m2) {
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          O = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.wc2[(m2 * 16 + r) * 64 + lane], G[r], O, 0, 0, 0);
        const int c0 = 32 * m2 + 4 * h;
        const float* pb = tl.bc2 + c0;
        const float* ps = tl.ps ? tl.ps + c0 : nullptr;
        const float* pt = tl.ps ? tl.pt + c0 : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = c0 + PDSE_KR(r);
          if (pvalid && co < d.C2) {
            float y = O[r] + pb[PDSE_KR(r)];
            if (ps) y = y * ps[PDSE_KR(r)] + pt[PDSE_KR(r)];
            y = y > 0.f ? y : slope * y;
            const int64_t o = CR1 ? (int64_t)c0 * cstep + (int64_t)PDSE_KR(r) * cstep : chan_off(co);
#if defined(PDSE_ABLATE) && (PDSE_ABLATE & 8)
            if (y == 1234.5f) obase[o] = y;   // diagnostic: tail without its stores
#else
            obase[o] = y;
#endif
          }
        }
      }
    }
  }
}

// BIGLU tail without the stores: y[m2][r] = act(bn(Wc2 (L*mR + R*mL) + bc2)) for C2 <= 64 channels
// (C2 == 1: y[0][0] on every lane).  Used by the dual-phase transposed conv, which stores the even
// and the odd output bin of a lane with ONE 8-byte store (two strided 4-byte stores are not merged
// on their way to HBM: profiles/r01_pmc_traffic_v5.json showed 2x WRITE_SIZE).
// Per-element control flow is kept out of the tail on purpose: the activation of a BIGLU stage is PReLU or nothing
// (PReLU with slope 1), and the LDS tail image always carries a BatchNorm pair (identity when the stage has none).
// With `switch (d.act)` / `if (ps)` per element the tail ran ~1000 scalar branches per 32-position tile and took
// 2.4x as long per MFMA as the gather loop (profiles/r01_ablation_v8.txt).
template <typename Sink>   // sink(m2, r, value) is called once per output element, in (m2, r) order
__device__ __forceinline__ void biglu_tail_values(const pdse_gconv_desc& d, const pdse_tail& tl, const f32x16& accL,
                                                  const f32x16& accR, const int lane, const int h, Sink&& sink,
                                                  const bool frame0 = false) {
  f32x16 L = accL, R = accR;
  const float slope = d.act == PDSE_ACT_PRELU ? d.act_slope : 1.0f;   // validated: BIGLU stages use PReLU or nothing
  L += ld16((frame0 ? tl.bl0 : tl.bl) + 4 * h);
  R += ld16((frame0 ? tl.br0 : tl.br) + 4 * h);
  const f32x16 vblc = ld16(tl.blc + 4 * h), vbrc = ld16(tl.brc + 4 * h);
  f32x16 mL, mR;
#pragma unroll
  for (int r = 0; r < 16; ++r) mL[r] = mR[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    mL = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.wlc[r * 64 + lane], L[r], mL, 0, 0, 0);
    mR = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.wrc[r * 64 + lane], R[r], mR, 0, 0, 0);
  }
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float ml = sigmoid_f(mL[r] + vblc[r]);
    const float mr = sigmoid_f(mR[r] + vbrc[r]);
    G[r] = L[r] * mr + R[r] * ml;
  }
  if (d.C2 == 1) {
    const f32x16 vw = ld16(tl.wc2 + 4 * h);
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += vw[r] * G[r];
    float v = part + __shfl_xor(part, 32) + tl.bc2[0];
    v = v * tl.ps[0] + tl.pt[0];
    sink(0, 0, v > 0.f ? v : slope * v);
  } else {
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
      f32x16 O;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        O = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.wc2[(m2 * 16 + r) * 64 + lane], G[r], O, 0, 0, 0);
      const int c0 = 32 * m2 + 4 * h;
      const f32x16 vb = ld16(tl.bc2 + c0), vs = ld16(tl.ps + c0), vt = ld16(tl.pt + c0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (O[r] + vb[r]) * vs[r] + vt[r];
        sink(m2, r, v > 0.f ? v : slope * v);
      }
    }
  }
}

// One chained 1x1 tile: z = z0 + Wn_i y over the 64 channels of the block output held in accumulator order.
__device__ __forceinline__ f32x16 nx_tile_acc(const pdse_tail& tl, const int i, const float (&Y)[2][16], const int lane,
                                              f32x16 Z) {
  const float* w = tl.nxw + (size_t)i * 2048 + lane;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
    for (int r = 0; r < 16; ++r) Z = __builtin_amdgcn_mfma_f32_32x32x2f32(w[(m2 * 16 + r) * 64], Y[m2][r], Z, 0, 0, 0);
  return Z;
}

__device__ __forceinline__ f32x16 nx_tile(const pdse_tail& tl, const int i, const float (&Y)[2][16], const int lane) {
  f32x16 Z;
#pragma unroll
  for (int r = 0; r < 16; ++r) Z[r] = 0.f;
  const float* w = tl.nxw + (size_t)i * 2048 + lane;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
    for (int r = 0; r < 16; ++r) Z = __builtin_amdgcn_mfma_f32_32x32x2f32(w[(m2 * 16 + r) * 64], Y[m2][r], Z, 0, 0, 0);
  return Z;
}

struct nx_operand_f32 {
  float Y[2][16];
};
__device__ __forceinline__ nx_operand_f32 nx_operand(const pdse_tail&, const float (&Y)[2][16]) {
  nx_operand_f32 o;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
    for (int r = 0; r < 16; ++r) o.Y[m2][r] = Y[m2][r];
  return o;
}
__device__ __forceinline__ f32x16 nx_tile(const pdse_tail& tl, const int i, const nx_operand_f32& o, const int lane) {
  return nx_tile(tl, i, o.Y, lane);
}

// ---- the same tail on split-bf16 operands (overloads selected by the tail image type)
template <typename Sink>
__device__ __forceinline__ void biglu_tail_values(const pdse_gconv_desc& d, const pdse_tail_s3& tl, const f32x16& accL,
                                                  const f32x16& accR, const int lane, const int h, Sink&& sink,
                                                  const bool frame0 = false) {
  f32x16 L = accL, R = accR;
  const float slope = d.act == PDSE_ACT_PRELU ? d.act_slope : 1.0f;
  L += ld16((frame0 ? tl.bl0 : tl.bl) + 4 * h);
  R += ld16((frame0 ? tl.br0 : tl.br) + 4 * h);
  f32x16 mL = ld16(tl.blc + 4 * h), mR = ld16(tl.brc + 4 * h);    // the biases seed the accumulators
  mL = chain_s3(tl.wlc + lane, L, mL);
  mR = chain_s3(tl.wrc + lane, R, mR);
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) G[r] = L[r] * sigmoid_f(mR[r]) + R[r] * sigmoid_f(mL[r]);
  if (d.C2 == 1) {
    const f32x16 vw = ld16(tl.wc2v + 4 * h);
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += vw[r] * G[r];
    float v = part + __shfl_xor(part, 32) + tl.bc2[0];
    v = v * tl.ps[0] + tl.pt[0];
    sink(0, 0, v > 0.f ? v : slope * v);
  } else {
    uint4 gp[2][3];     // the gate is split once for both output tiles
    split16(G, gp);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
      const int c0 = 32 * m2 + 4 * h;
      f32x16 O = ld16(tl.bc2 + c0);
      O = chain_s3p(tl.wc2 + m2 * 384 + lane, gp, O);
      const f32x16 vs = ld16(tl.ps + c0), vt = ld16(tl.pt + c0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = O[r] * vs[r] + vt[r];
        sink(m2, r, v > 0.f ? v : slope * v);
      }
    }
  }
}

__device__ __forceinline__ f32x16 nx_tile_acc(const pdse_tail_s3& tl, const int i, const float (&Y)[2][16], const int lane,
                                              f32x16 Z) {
  const uint4* w = tl.nxw + (size_t)i * 768 + lane;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
    f32x16 X;
#pragma unroll
    for (int r = 0; r < 16; ++r) X[r] = Y[m2][r];
    Z = chain_s3(w + m2 * 384, X, Z);
  }
  return Z;
}

__device__ __forceinline__ f32x16 nx_tile(const pdse_tail_s3& tl, const int i, const float (&Y)[2][16], const int lane) {
  f32x16 Z;
#pragma unroll
  for (int r = 0; r < 16; ++r) Z[r] = 0.f;
  return nx_tile_acc(tl, i, Y, lane, Z);
}

// the block output split once for every chained tile of the launch (up to three: next stage's H, two skip halves)
struct nx_operand_s3 {
  uint4 p[2][2][3];
};
__device__ __forceinline__ nx_operand_s3 nx_operand(const pdse_tail_s3&, const float (&Y)[2][16]) {
  nx_operand_s3 o;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
    f32x16 X;
#pragma unroll
    for (int r = 0; r < 16; ++r) X[r] = Y[m2][r];
    split16(X, o.p[m2]);
  }
  return o;
}
__device__ __forceinline__ f32x16 nx_tile(const pdse_tail_s3& tl, const int i, const nx_operand_s3& o, const int lane) {
  f32x16 Z;
#pragma unroll
  for (int r = 0; r < 16; ++r) Z[r] = 0.f;
  const uint4* w = tl.nxw + (size_t)i * 768 + lane;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) Z = chain_s3p(w + m2 * 384, o.p[m2], Z);
  return Z;
}

// Single-phase BIGLU tail with chained next-stage 1x1 convolutions (pdse.h: nx_*); out_cr == 1, C2 == 64.
template <typename TL>
__device__ __forceinline__ void biglu_nx_epilogue(const pdse_gconv_desc& d, const TL& tl, const f32x16& a0,
                                                  const f32x16& a1, const int b, const int t, const int j,
                                                  const bool pvalid, const int lane, const int h) {
  float Y[2][16];
  float* const obase = d.out + ((int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off);
  const int64_t cstep = d.out_sc_hi;
  const bool keep = (d.nx_keep != 0 || d.nx_n == 0) && pvalid;   // no chained tile: the block output itself is the result
  biglu_tail_values(d, tl, a0, a1, lane, h, [&](const int m2, const int r, const float v) {
    Y[m2][r] = v;
    if (keep) obase[(int64_t)(32 * m2 + 4 * h) * cstep + (int64_t)PDSE_KR(r) * cstep] = v;
  }, t == 0);
  const auto yop = nx_operand(tl, Y);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= d.nx_n) break;
    const f32x16 Z = nx_tile(tl, i, yop, lane);
    const int64_t sc = d.nx_sc[i];
    const int64_t rowbase = (int64_t)b * d.nx_sb[i] + (int64_t)j * d.nx_sf[i] + (int64_t)(4 * h) * sc;
    const int64_t off = rowbase + (int64_t)t * d.nx_st[i] + d.nx_off[i];
    float* const zb = d.nx_out[i] + off;
    const float* const ab = d.nx_add[i] ? d.nx_add[i] + off : nullptr;
    const f32x16 pb = ld16(tl.nxb + 32 * i + 4 * h);
    if (pvalid) {
      if (ab) {
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[(int64_t)PDSE_KR(r) * sc] = Z[r] + pb[r] + ab[(int64_t)PDSE_KR(r) * sc];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[(int64_t)PDSE_KR(r) * sc] = Z[r] + pb[r];
      }
      if (i == d.nx_row0 && t == 0) {   // explicit pad frame of the next encoder stage: conv1(0 + tp) = the folded bias
        float* const z0 = d.nx_out[i] + rowbase;
#pragma unroll
        for (int r = 0; r < 16; ++r) z0[(int64_t)PDSE_KR(r) * sc] = pb[r];
      }
    }
  }
}

// one 8-byte store of two neighbouring bins; the address is only 4-byte aligned (odd row lengths),
// which global_store_dwordx2 accepts
__device__ __forceinline__ void store_pair(float* p, const float a, const float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// Dual-phase epilogue (out_cr == 1, C2 == 64 or 1): even bin at obase, odd bin one bin stride later.
// NX: the launch chains the next stage's conv1 (its own register budget: a separate instantiation).
template <bool NX, typename TL>
__device__ __forceinline__ void biglu_dual_epilogue(const pdse_gconv_desc& d, const TL& tl, const f32x16& a0,
                                                    const f32x16& a1, const f32x16& a2, const f32x16& a3, const int b,
                                                    const int t, const int j, const bool pvalid, const int lane,
                                                    const int h) {
  if constexpr (NX) {
    // the 64-channel output of either phase only feeds the chained next-stage conv1 (nx_keep == 0): nothing of it is
    // stored.  The addend (the encoder's skip half + time bias) is requested BEFORE the tail whose MFMAs hide its
    // latency, one phase at a time (16 registers each), and seeds the chained tile's accumulator.
    const int64_t sc = d.nx_sc[0], nbin = d.nx_sf[0] >> 1;
    const int64_t off = (int64_t)b * d.nx_sb[0] + (int64_t)t * d.nx_st[0] + (int64_t)j * d.nx_sf[0] + d.nx_off[0] + (int64_t)(4 * h) * sc;
    float* const zb = d.nx_out[0] + off;
    const float* const ab = d.nx_add[0] ? d.nx_add[0] + off : nullptr;
    const f32x16 pb = ld16(tl.nxb + 4 * h);
    const bool two = pvalid && j < d.Fout1;
    float Y[2][16];
    f32x16 Z0, Z1;
#pragma unroll
    for (int r = 0; r < 16; ++r) Z0[r] = (ab && pvalid) ? ab[(int64_t)PDSE_KR(r) * sc] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    biglu_tail_values(d, tl, a0, a1, lane, h, [&](const int m2, const int r, const float v) { Y[m2][r] = v; });
    Z0 = nx_tile_acc(tl, 0, Y, lane, Z0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Z1[r] = (ab && two) ? ab[(int64_t)PDSE_KR(r) * sc + nbin] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    biglu_tail_values(d, tl, a2, a3, lane, h, [&](const int m2, const int r, const float v) { Y[m2][r] = v; });
    Z1 = nx_tile_acc(tl, 0, Y, lane, Z1);
    if (nbin == 1 && two) {             // the common case: neighbouring bins, one 8-byte store per channel row
#pragma unroll
      for (int r = 0; r < 16; ++r) store_pair(zb + (int64_t)PDSE_KR(r) * sc, Z0[r] + pb[r], Z1[r] + pb[r]);
    } else if (pvalid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) zb[(int64_t)PDSE_KR(r) * sc] = Z0[r] + pb[r];
      if (two) {
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[(int64_t)PDSE_KR(r) * sc + nbin] = Z1[r] + pb[r];
      }
    }
    return;
  }
  // even bins first (kept in registers), then the odd tail streams its values straight into the paired stores
  float ye[2][16];
  biglu_tail_values(d, tl, a0, a1, lane, h, [&](const int m2, const int r, const float v) { ye[m2][r] = v; });
  float* const obase = d.out + ((int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off);
  const int64_t bin = d.out_sf >> 1;
  const bool both = pvalid && j < d.Fout1;
  const int64_t cstep = d.out_sc_hi;
  const bool one = d.C2 == 1;
  biglu_tail_values(d, tl, a2, a3, lane, h, [&](const int m2, const int r, const float vo) {
    if (one && h != 0) return;
    float* p = one ? obase : obase + (int64_t)(32 * m2 + 4 * h) * cstep + (int64_t)PDSE_KR(r) * cstep;
    const float ve = ye[m2][r];
    if (both && bin == 1) store_pair(p, ve, vo);
    else if (pvalid) {
      p[0] = ve;
      if (both) p[bin] = vo;
    }
  });
}

// acc0/acc1: MT accumulator tiles of this wave; (b, t, j) its output position on this lane.
template <int EPI, int MT>
__device__ __forceinline__ void gconv_epilogue(const pdse_gconv_desc& d, const pdse_tail& tl, f32x16* acc0,
                                               f32x16* acc1, const int b, const int t, const int j,
                                               const bool pvalid, const int lane, const int h, const int mt0,
                                               const int mtiles, const int64_t extra_off = 0) {
  if (d.out_cr == 1)
    gconv_epilogue_impl<EPI, MT, true>(d, tl, acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles, extra_off);
  else
    gconv_epilogue_impl<EPI, MT, false>(d, tl, acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles, extra_off);
}

// Raw buffer access: address = resource base (4 scalar registers) + 32-bit lane byte offset + scalar byte offset, so no
// 64-bit per-lane address lives in vector registers (with flat pointers hipcc kept one 64-bit address per load of a tile -
// 72 registers - and spilled; every spill reload is a scratch load + s_waitcnt vmcnt(0), which drains all prefetches), and
// reads beyond the resource return zero instead of faulting.
typedef int v4i_t __attribute__((ext_vector_type(4)));
#define PDSE_RSRC_FLAGS 0x00020000
// The base and the size pass through readfirstlane: a resource that hipcc does not PROVE wave-uniform (it kept the one of the
// plane loads in vector registers) makes every access a waterfall loop - four readfirstlane, two compares, a saveexec and a
// branch per load, 36 of them per tile, each splitting the basic block the slot schedule lives in.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, const uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  void* const q = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)__builtin_amdgcn_readfirstlane(bytes), PDSE_RSRC_FLAGS);
}
__device__ __forceinline__ uint4 bload16(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
  const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
  return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
}
// AUX: cache policy bits of the load (16 = sc1: served by L2 / memory, never by this CU's L1 - data another workgroup of the same
// launch has written)
template <int AUX>
__device__ __forceinline__ uint4 bload16a(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
  const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, AUX);
  return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
}
__device__ __forceinline__ float bload4(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
// AUX: cache policy bits of the store (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void bstore16(const uint4& x, const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
  v4i_t v;
  v.x = (int)x.x; v.y = (int)x.y; v.z = (int)x.z; v.w = (int)x.w;
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void bstore4(const float x, const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x), r, (int)voff, soff, AUX);
}

#endif
