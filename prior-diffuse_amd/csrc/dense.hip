// dense.hip — one layer of DB-AIAT's dilated dense block (model/dbaiat.py:605-631) as ONE launch:
//   pad -> Conv2d(64 i -> 64, kernel (2,3), dilation (2^(i-1),1)) -> LayerNorm over the F bins -> PReLU(64)
// on the channel-blocked concatenation buffer D [B][G][T + tpad][F + 2][8] fp32 (G = 40 groups of 8 channels: [out4, out3, out2,
// out1, x]; zero pads: tpad rows in front - the causal time padding - and one bin on each side).  Before this file a layer was a
// gather convolution (csrc/gconv4.hip, fp32 NCHW output) plus a row LayerNorm launch that read that output back and wrote it again;
// every tap of every layer re-split its fp32 operands in registers (eleven VALU instructions per pair, six taps, up to four layers).
//
// Arithmetic: the same exact three-way bf16 split as gconv4.hip (six bf16 products per fp32 multiply-add, fp32 accumulation), or
// plain bf16 operands (np = 1: the opt-in bf16 mode).
//
// Workgroup = 8 waves, 16 tiles of 32 positions: R rows (t0, t0 + dil, ..., t0 + (R-1) dil) of one batch item, all F bins, all 64
// output channels - whole rows, so that the LayerNorm is an epilogue; rows dil apart, so that the time tap -dil of row rho + 1 is
// the slab row rho reads at tap 0 (R + 1 input rows per R output rows).
//   * B operand (activations): per K chunk = (16 channels, one time tap) every wave loads the fp32 entries its two tiles need -
//     32 positions plus one halo bin per side of the up to two rows a tile touches - ONCE, splits them and writes the planes to
//     its own LDS region; the three bin taps then read the region at lane offsets -1, 0, +1 (ds_read_b128, no VALU work).
//     Regions are private to a wave and double-buffered: the loads of chunk c + 1 are issued before chunk c's matrix work
//     and split under its last step.
//   * A operand (weights, [K step][2 channel tiles][np planes][64 lanes] uint4): streamed through a double-buffered LDS ring by
//     LDS-DMA, one barrier per chunk (three K steps = 72 MFMAs per wave).
//   * epilogue: accumulators + bias -> LDS [position][64 + 4], row statistics (shifted sums, fixed order: results do not depend on
//     how the batch is cut), normalise, PReLU, 32-byte entries into D's output groups.
// Work items are mapped so that each XCD (blockIdx % 8) walks a contiguous range of (item, row block): the rows a block shares
// with its neighbours are in that XCD's L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

#define REQ(cond, msg)     \
  do {                     \
    if (!(cond)) {         \
      pdse_set_error(msg); \
      return 1;            \
    }                      \
  } while (0)

#define DN_SLOTS 36   // B slots per (tile, plane, channel half): 32 positions + a halo bin on each side of up to two row segments
#define DN_SROW 68    // floats per staged position in the epilogue (64 channels + 4: conflict-free 16-byte writes)
#define DN_NPOS 512   // positions per workgroup (8 waves x 2 tiles x 32)

typedef unsigned dn_u32x4 __attribute__((ext_vector_type(4)));
typedef float dn_f32x4 __attribute__((ext_vector_type(4)));

// Loads and LDS reads of the main loop are inline asm: with an LDS-DMA in flight hipcc waits for vmcnt(0) in front of every
// compiler-visible LDS read and load result (see gconv4.hip).  What orders them: s_waitcnt written here, and the chunk barrier.
// Nothing may name a destination register between the request and the wait (tests/test_isa_guards.py).
template <int OFF>
__device__ __forceinline__ void dn_gload16(dn_f32x4& v, const void* sbase, const unsigned off) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(off), "s"(sbase), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void dn_lds16(const unsigned addr, dn_u32x4& v) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ void dn_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}
__device__ __forceinline__ uint4 dn_u4(const dn_u32x4& a) { return make_uint4(a[0], a[1], a[2], a[3]); }

template <int NP>
struct dn_ops {   // the operands of one K step: A fragments of the two channel tiles, B fragments of the wave's two position tiles
  dn_u32x4 a[2][NP], b[2][NP];
};

// ring reads of K step S of the chunk in the ring buffer at `ra`, region reads at bin tap S (df = S - 1)
template <int NP, int S, int... I>
__device__ __forceinline__ void dn_fetch_(const unsigned ra, const unsigned rb0, const unsigned rb1, dn_ops<NP>& o,
                                          std::integer_sequence<int, I...>) {
  (dn_lds16<((S * 2 + I / NP) * NP + I % NP) * 1024>(ra, o.a[I / NP][I % NP]), ...);
  (dn_lds16<(I % NP) * (2 * DN_SLOTS * 16) + S * 16>(I / NP ? rb1 : rb0, o.b[I / NP][I % NP]), ...);
}
template <int NP, int S>
__device__ __forceinline__ void dn_fetch(const unsigned ra, const unsigned rb0, const unsigned rb1, dn_ops<NP>& o) {
  dn_fetch_<NP, S>(ra, rb0, rb1, o, std::make_integer_sequence<int, 2 * NP>());
}
template <int NP>
__device__ __forceinline__ void dn_wait(dn_ops<NP>& o) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      asm volatile("" : "+v"(o.a[i][p]));
      asm volatile("" : "+v"(o.b[i][p]));
    }
}
// 4 accumulators (position tile k, channel tile m); the six products smallest first, the accumulators interleaved so that
// dependent MFMAs are three instructions apart
__device__ __forceinline__ void dn_mm(const dn_ops<3>& o, f32x16 (&acc)[2][2]) {
  constexpr int pa[6] = {0, 2, 1, 0, 1, 0}, pb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[k][m] = mfma_bf16(dn_u4(o.a[m][pa[q]]), dn_u4(o.b[k][pb[q]]), acc[k][m]);
}
// f16x2 (gconv_common.h: split8h): a2 b1, a1 b2, a1 b1 on the f16 matrix cores
__device__ __forceinline__ void dn_mm(const dn_ops<2>& o, f32x16 (&acc)[2][2]) {
  constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[k][m] = mfma_f16(dn_u4(o.a[m][pa[q]]), dn_u4(o.b[k][pb[q]]), acc[k][m]);
}
__device__ __forceinline__ void dn_mm(const dn_ops<1>& o, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[k][m] = mfma_bf16(dn_u4(o.a[m][0]), dn_u4(o.b[k][0]), acc[k][m]);
}
__device__ __forceinline__ uint32_t dn_pack_bf16(const float a, const float b) {   // round to nearest even (v_cvt_pk_bf16_f32)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  union { bf16x2 h; uint32_t u; } c;
  c.h = __builtin_convertvector(v, bf16x2);
  return c.u;
}

struct dn_geom {   // launch geometry the host derives from the descriptor
  int R, nblk, nwork;
};

template <int NP>
__global__ __launch_bounds__(512, 1) void dense_kernel(const pdse_dense_desc d, const dn_geom gm) {
  constexpr int REGION = NP * 2 * DN_SLOTS * 16;   // bytes of one (wave, tile) region
  constexpr int BBUF = 8 * 2 * REGION;             // one buffer of all regions
  constexpr int ACH = 3 * 2 * NP * 1024;           // weight bytes of one chunk (3 K steps x 2 channel tiles x NP planes)
  constexpr int PSTR = 2 * DN_SLOTS * 16;          // plane stride inside a region
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][BBUF] regions | [2][ACH] ring; the epilogue re-uses it
  const unsigned lbase = (unsigned)(uintptr_t)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;

  // work item: XCD x (= blockIdx % 8) walks items / row blocks x * chunkw ... in order
  const int chunkw = (gm.nwork + 7) >> 3;
  const int wid = (int)(blockIdx.x & 7) * chunkw + (int)(blockIdx.x >> 3);
  if (wid >= gm.nwork) return;
  const int b = wid / gm.nblk, qb = wid - b * gm.nblk;
  const int qh = qb / d.dil, ql = qb - qh * d.dil;
  const int R = gm.R, F = d.F, T = d.T, dil = d.dil;
  const int t0 = qh * R * dil + ql;   // row rho of the block is frame t0 + rho * dil
  const int Fp = F + 2, Tp = T + d.tpad;
  const int npos = R * F;
  const unsigned gsb = (unsigned)Tp * Fp * 32;   // bytes between channel groups

  // ---- geometry of the wave's two tiles
  int rhoA[2], fa[2], n1[2];
  unsigned rb[2];   // region read address of this lane at bin tap -1, plane 0, buffer 0
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p0 = (wave * 2 + k) * 32;
    rhoA[k] = p0 / F;
    fa[k] = p0 - rhoA[k] * F;
    n1[k] = min(32, F - fa[k]);
    const int lslot = col < n1[k] ? col + 1 : col + 3;
    rb[k] = lbase + (wave * 2 + k) * REGION + (h * DN_SLOTS + lslot - 1) * 16;
  }
  // ---- staging entries: 2 tiles x 2 channel halves x 36 slots = 144 entries of 32 bytes (8 fp32 channels of one bin) = 3 rounds of
  // 48 lanes; lanes 48..63 repeat the entries of lanes 0..15 (same loads, same LDS writes: no divergent code in the main loop)
  unsigned voff[3], wr[3];
  const int lane48 = lane < 48 ? lane : lane - 48;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int e = r * 48 + lane48;
    const int k = e >= 2 * DN_SLOTS ? 1 : 0, rem = e - k * 2 * DN_SLOTS;
    const int hh = rem >= DN_SLOTS ? 1 : 0, s = rem - hh * DN_SLOTS;
    const int ra_ = k ? rhoA[1] : rhoA[0], fa_ = k ? fa[1] : fa[0], n1_ = k ? n1[1] : n1[0];
    const bool segA = s < n1_ + 2;
    const int rho = segA ? ra_ : ra_ + 1;
    const int f = segA ? fa_ - 1 + s : s - n1_ - 3;   // -1 .. F: the halo bins of a row are D's zero pads
    const int t = t0 + rho * dil;
    const bool ok = rho < R && t < T && f <= F;
    voff[r] = ok ? (unsigned)((t * Fp + f + 1) * 32) + (unsigned)hh * gsb : 0u;   // 0: a pad bin (zero)
    wr[r] = (unsigned)((wave * 2 + k) * REGION + (hh * DN_SLOTS + s) * 16);
  }

  dn_f32x4 sv[3][2];
  // chunk ci = 2 kb + kt: input groups g_in + 2 kb (+ h), time tap (kt - 1) dil
  auto stage_load = [&](const int ci) {
    const int kb = ci >> 1, kt = ci & 1;
    const char* cb = reinterpret_cast<const char*>(d.D) +
                     ((((int64_t)b * d.G + d.g_in + 2 * kb) * Tp + d.tpad + (kt - 1) * dil) * Fp) * 32;
    // wave-uniform by construction; said so to the compiler (an "s" operand), halves zero-extended (readfirstlane returns int)
    cb = reinterpret_cast<const char*>((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)cb) |
                                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)cb >> 32)) << 32));
    dn_gload16<0>(sv[0][0], cb, voff[0]);
    dn_gload16<16>(sv[0][1], cb, voff[0]);
    dn_gload16<0>(sv[1][0], cb, voff[1]);
    dn_gload16<16>(sv[1][1], cb, voff[1]);
    dn_gload16<0>(sv[2][0], cb, voff[2]);
    dn_gload16<16>(sv[2][1], cb, voff[2]);
  };
  auto dma = [&](const int ci, const int buf) {
    const char* src = reinterpret_cast<const char*>(d.w) + (size_t)ci * ACH + lane * 16;
    unsigned char* dst = lds + 2 * BBUF + buf * ACH;
    for (int pc = wave; pc < 6 * NP; pc += 8) dn_glds16(src + pc * 1024, dst + pc * 1024);
  };
  auto landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      asm volatile("" : "+v"(sv[r][0]));
      asm volatile("" : "+v"(sv[r][1]));
    }
  };
  auto split_write = [&](const int buf) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      {
        const float x[8] = {sv[r][0][0], sv[r][0][1], sv[r][0][2], sv[r][0][3], sv[r][1][0], sv[r][1][1], sv[r][1][2], sv[r][1][3]};
        unsigned char* dst = lds + buf * BBUF + wr[r];
        if constexpr (NP == 3) {
          uint4 p1, p2, p3;
          split8(x, p1, p2, p3);
          *reinterpret_cast<uint4*>(dst) = p1;
          *reinterpret_cast<uint4*>(dst + PSTR) = p2;
          *reinterpret_cast<uint4*>(dst + 2 * PSTR) = p3;
        } else if constexpr (NP == 2) {
          uint4 p1, p2;
          split8h(x, pow2i(PDSE_F16_ACT_EXP), p1, p2);
          *reinterpret_cast<uint4*>(dst) = p1;
          *reinterpret_cast<uint4*>(dst + PSTR) = p2;
        } else {
          *reinterpret_cast<uint4*>(dst) = make_uint4(dn_pack_bf16(x[0], x[1]), dn_pack_bf16(x[2], x[3]), dn_pack_bf16(x[4], x[5]),
                                                      dn_pack_bf16(x[6], x[7]));
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][m][r] = 0.f;

  const int nkb = d.cin >> 4;
  const unsigned ra0 = lbase + 2 * BBUF + lane * 16;
  stage_load(0);
  dma(0, 0);
  landed();
  split_write(0);
  __syncthreads();
  dn_ops<NP> X, Y;
#pragma clang loop unroll(disable)
  for (int kb = 0; kb < nkb; ++kb) {
    // ---- chunk 2 kb (time tap -dil) in buffer 0; chunk 2 kb + 1 is staged into buffer 1
    stage_load(2 * kb + 1);
    dma(2 * kb + 1, 1);
    dn_fetch<NP, 0>(ra0, rb[0], rb[1], X);
    dn_wait(X);
    dn_fetch<NP, 1>(ra0, rb[0], rb[1], Y);
    dn_mm(X, acc);
    dn_wait(Y);
    dn_fetch<NP, 2>(ra0, rb[0], rb[1], X);
    dn_mm(Y, acc);
    dn_wait(X);
    landed();
    split_write(1);
    dn_mm(X, acc);
    __syncthreads();
    // ---- chunk 2 kb + 1 (time tap 0) in buffer 1
    const bool more = kb + 1 < nkb;
    if (more) {
      stage_load(2 * kb + 2);
      dma(2 * kb + 2, 0);
    }
    dn_fetch<NP, 0>(ra0 + ACH, rb[0] + BBUF, rb[1] + BBUF, Y);
    dn_wait(Y);
    dn_fetch<NP, 1>(ra0 + ACH, rb[0] + BBUF, rb[1] + BBUF, X);
    dn_mm(Y, acc);
    dn_wait(X);
    dn_fetch<NP, 2>(ra0 + ACH, rb[0] + BBUF, rb[1] + BBUF, Y);
    dn_mm(X, acc);
    dn_wait(Y);
    if (more) {
      landed();
      split_write(0);
    }
    dn_mm(Y, acc);
    __syncthreads();
  }

  // ---- epilogue: bias, staged [position][DN_SROW], LayerNorm over the bins of every (row, channel), PReLU, 32-byte entries
  float* const st = reinterpret_cast<float*>(lds);
  float* const part = st + DN_NPOS * DN_SROW;   // [parts][R * 64][2]
  const float us = NP == 2 ? pow2i(-(PDSE_F16_ACT_EXP + d.wexp)) : 1.0f;
  float* const stats = part + 1024;             // [R * 64][2] (mean, rstd)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (wave * 2 + k) * 32 + col;
    if (p < npos) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c0 = 32 * m + 8 * j + 4 * h;
          const float4 bv = *reinterpret_cast<const float4*>(d.bias + c0);
          float4 v;
          v.x = acc[k][m][4 * j + 0] * us + bv.x;   // us: 1, or (f16x2) the power of two that undoes the operand scaling
          v.y = acc[k][m][4 * j + 1] * us + bv.y;
          v.z = acc[k][m][4 * j + 2] * us + bv.z;
          v.w = acc[k][m][4 * j + 3] * us + bv.w;
          *reinterpret_cast<float4*>(st + p * DN_SROW + c0) = v;
        }
    }
  }
  __syncthreads();
  const int npairs = R * 64, parts = 512 / npairs;   // R <= 8
  {
    const int pr = tid / npairs, pc = tid - pr * npairs;   // pc = rho * 64 + c: lanes of a wave read consecutive channels
    if (pr < parts) {
      const int rho = pc >> 6, c = pc & 63;
      const float* row = st + (rho * F) * DN_SROW + c;
      const float x0 = row[0];
      const int f0 = pr * F / parts, f1 = (pr + 1) * F / parts;
      float s = 0.f, q = 0.f;
#pragma unroll 8
      for (int f = f0; f < f1; ++f) {
        const float e = row[f * DN_SROW] - x0;
        s += e;
        q = fmaf(e, e, q);
      }
      part[(pr * npairs + pc) * 2] = s;
      part[(pr * npairs + pc) * 2 + 1] = q;
    }
  }
  __syncthreads();
  if (tid < npairs) {
    float s = 0.f, q = 0.f;
    for (int pr = 0; pr < parts; ++pr) {
      s += part[(pr * npairs + tid) * 2];
      q += part[(pr * npairs + tid) * 2 + 1];
    }
    const float x0 = st[((tid >> 6) * F) * DN_SROW + (tid & 63)];
    const float ms = s / (float)F;                       // mean - x0
    const float var = fmaxf(q / (float)F - ms * ms, 0.f);
    stats[tid * 2] = x0 + ms;
    stats[tid * 2 + 1] = 1.0f / sqrtf(var + d.eps);
  }
  __syncthreads();
  float* const obase = const_cast<float*>(d.D) + (((int64_t)b * d.G + d.g_out) * Tp + d.tpad) * (int64_t)Fp * 8;
  for (int i = tid; i < 8 * npos; i += 512) {
    const int g = i / npos, pos = i - g * npos;
    const int rho = pos / F, f = pos - rho * F;
    const int t = t0 + rho * dil;
    if (t >= T) continue;
    const float4 xa = *reinterpret_cast<const float4*>(st + pos * DN_SROW + 8 * g);
    const float4 xb = *reinterpret_cast<const float4*>(st + pos * DN_SROW + 8 * g + 4);
    const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    const float ga = d.gamma[f], be = d.beta[f];
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 mr = *reinterpret_cast<const float2*>(stats + (rho * 64 + 8 * g + e) * 2);
      const float v = (x[e] - mr.x) * mr.y * ga + be;
      y[e] = v > 0.f ? v : d.slope[8 * g + e] * v;
    }
    float* o = obase + ((int64_t)g * Tp + t) * (int64_t)Fp * 8 + (f + 1) * 8;
    *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
  }
}

static int dense_geom(const pdse_dense_desc* d, dn_geom* g) {
  g->R = DN_NPOS / d->F;
  if (g->R > 8) g->R = 8;
  g->nblk = ((d->T + g->R * d->dil - 1) / (g->R * d->dil)) * d->dil;
  const long long nwork = (long long)g->nblk * d->B;
  if (nwork >= (1ll << 28)) return 1;
  g->nwork = (int)nwork;
  return 0;
}

template <int NP>
static int dense_launch_(const pdse_dense_desc* d, hipStream_t s) {
  dn_geom g;
  REQ(dense_geom(d, &g) == 0, "dense: too many row blocks");
  const size_t lds = (size_t)(DN_NPOS * DN_SROW + 1024 + 1024) * sizeof(float);   // the epilogue's image (the main loop's is smaller)
  static_assert(2 * (8 * 2 * 3 * 2 * DN_SLOTS * 16) + 2 * (3 * 2 * 3 * 1024) <= (DN_NPOS * DN_SROW + 2048) * 4, "main-loop LDS image");
  static unsigned long long attr_mask = 0;
  if (pdse_lds_attr((const void*)dense_kernel<NP>, &attr_mask, "dense lds attribute")) return 1;
  const unsigned grid = 8u * (unsigned)((g.nwork + 7) / 8);
  hipLaunchKernelGGL((dense_kernel<NP>), dim3(grid), dim3(512), lds, s, *d, g);
  return pdse_check_launch("dense");
}

int pdse_dense_launch(const pdse_dense_desc* d, hipStream_t s) {
  REQ(d && d->D && d->w && d->bias && d->gamma && d->beta && d->slope, "dense: null pointer");
  REQ(d->B > 0 && d->T > 0 && d->F >= DN_SLOTS && d->F <= 192, "dense: bad sizes (36 <= F <= 192)");
  REQ(d->np == 1 || d->np == 3 || (d->np == 2 && d->wexp >= -40 && d->wexp <= 40), "dense: np is 3 (exact split), 2 (f16x2, wexp within +-40) or 1 (plain bf16)");
  REQ(d->dil >= 1 && d->dil <= d->tpad, "dense: dilation exceeds the time padding of the buffer");
  REQ(d->cin > 0 && (d->cin & 15) == 0, "dense: input channels in multiples of 16");
  const int gi0 = d->g_in, gi1 = d->g_in + d->cin / 8, go0 = d->g_out, go1 = d->g_out + 8;
  REQ(gi0 >= 0 && gi1 <= d->G && go0 >= 0 && go1 <= d->G, "dense: channel groups outside the buffer");
  REQ(go1 <= gi0 || gi1 <= go0, "dense: output groups overlap the input groups");
  REQ((reinterpret_cast<uintptr_t>(d->D) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->w) & 15) == 0 &&
          (reinterpret_cast<uintptr_t>(d->bias) & 15) == 0,
      "dense: buffer, weights and bias are 16-byte aligned");
  // 32-bit lane offsets: two channel groups of one item
  REQ(2ll * (d->T + d->tpad) * (d->F + 2) * 32 < (1ll << 32), "dense: item too large for 32-bit lane offsets");
  if (d->np == 2) return dense_launch_<2>(d, s);
  return d->np == 3 ? dense_launch_<3>(d, s) : dense_launch_<1>(d, s);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row LayerNorm + PReLU (dbaiat.py:498 inp_norm / inp_prelu) - or a plain re-layout (gamma == NULL) - from a channel-major
// [B,C,T,F] tensor into channel-blocked 32-byte entries (the dense buffer's layout).  Workgroup = 8 waves = the 8 channels of a
// group x RB_R frames: every wave normalises its channel's rows (the arithmetic of aia.hip's rowln_kernel), the entries are
// assembled in LDS and written whole.
// ---------------------------------------------------------------------------------------------------------------------------
#define RB_R 4
__global__ __launch_bounds__(512) void rowlnb_kernel(const pdse_rowlnb_desc d) {
  __shared__ float ls[RB_R][8][192];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tb = blockIdx.x * RB_R, g = blockIdx.y, b = blockIdx.z;
  const int c = 8 * g + w;
  const bool norm = d.gamma != nullptr;
  float v[RB_R][3];
#pragma unroll
  for (int r = 0; r < RB_R; ++r) {
    const int t = min(tb + r, d.T - 1);
    const float* x = d.in + (int64_t)b * d.in_sb + (int64_t)c * d.in_sc + (int64_t)t * d.in_st;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = k * 64 + lane;
      v[r][k] = i < d.F ? x[i] : 0.f;
    }
  }
  float g3[3] = {1.f, 1.f, 1.f}, b3[3] = {0.f, 0.f, 0.f};
  float slope = 1.f;
  if (norm) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = k * 64 + lane;
      g3[k] = i < d.F ? d.gamma[i] : 0.f;
      b3[k] = i < d.F ? d.beta[i] : 0.f;
    }
    slope = d.slope[c];
  }
#pragma unroll
  for (int r = 0; r < RB_R; ++r) {
    float mean = 0.f, rstd = 1.f;
    if (norm) {
      float sum = (v[r][0] + v[r][1]) + v[r][2];
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      mean = sum / (float)d.F;
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float e = (k * 64 + lane) < d.F ? v[r][k] - mean : 0.f;
        sq += e * e;
      }
      for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
      rstd = 1.0f / sqrtf(sq / (float)d.F + d.eps);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = k * 64 + lane;
      if (i < d.F) {
        float y = v[r][k];
        if (norm) {
          y = (y - mean) * rstd * g3[k] + b3[k];
          y = y > 0.f ? y : slope * y;
        }
        ls[r][w][i] = y;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RB_R * 2 * d.F; i += 512) {
    const int r = i / (2 * d.F), rem = i - r * 2 * d.F;
    const int f = rem >> 1, hh = rem & 1;
    if (tb + r >= d.T) break;
    const float4 y = make_float4(ls[r][4 * hh][f], ls[r][4 * hh + 1][f], ls[r][4 * hh + 2][f], ls[r][4 * hh + 3][f]);
    *reinterpret_cast<float4*>(d.out + (int64_t)b * d.out_sb + (int64_t)g * d.out_sg + (int64_t)(tb + r) * d.out_st + f * 8 + 4 * hh) = y;
  }
}

int pdse_rowlnb_launch(const pdse_rowlnb_desc* d, hipStream_t s) {
  REQ(d && d->in && d->out, "rowlnb: null pointer");
  REQ(d->gamma == nullptr || (d->beta && d->slope), "rowlnb: gamma without beta / slope");
  REQ(d->B > 0 && d->C > 0 && (d->C & 7) == 0 && d->T > 0 && d->F > 0 && d->F <= 192, "rowlnb: bad sizes (C in multiples of 8, F <= 192)");
  REQ(d->B <= 65535 && d->C / 8 <= 65535, "rowlnb: grid too large");
  REQ((reinterpret_cast<uintptr_t>(d->out) & 15) == 0 && !((d->out_sb | d->out_sg | d->out_st) & 3), "rowlnb: output entries are 16-byte aligned");
  hipLaunchKernelGGL(rowlnb_kernel, dim3((d->T + RB_R - 1) / RB_R, d->C / 8, d->B), dim3(512), 0, s, *d);
  return pdse_check_launch("rowlnb");
}
