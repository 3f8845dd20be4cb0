// tcm2.hip — the fused TCM residual block of tcm.hip (model/diff3.py:215-257) in split-bf16 arithmetic
// (gconv_common.h: an fp32 value is the exact sum of three bf16, a product its six leading cross terms on the bf16
// matrix cores with fp32 accumulation), restructured around what bounded the fp32 kernel: its waves spent 58 % of
// their life in s_waitcnt (4 waves, each walking a long chain of weight loads with 4 groups in flight).
//
//   * 8 waves per workgroup;
//   * the 64-channel bottleneck tensor h travels between launches ALREADY transformed and split: the producer (phase
//     C of the previous block) applies both branches' PReLU + BatchNorm and stores the three bf16 planes in the
//     B-operand order of v_mfma_f32_32x32x16_bf16, frames innermost:
//         hs[B][2 branch][4 kb][2 kg][3 plane][T + 128][8]      (frame t at index t + 64; the margins stay zero)
//     so a tap outside [0,T) is an address, not a select, a wave's load of one fragment plane is two runs of 512
//     contiguous bytes, and the dilated gather costs no VALU work at all;
//   * what bounds the block is the CU's read path from L2 (measured here: 32 B/clk per CU): a workgroup reads all of
//     a block's weights (432 KB as bf16 planes) whatever its frame count, and the first version of this kernel (one
//     output tile per wave, 32 frames per workgroup, h rows of 768 B gathered at 16 B per lane) pulled 960 KB + its
//     line overfetch per CU through that path in phase A alone: 30,000 of its 52,000 cycles, the same 37 us per
//     block as the fp32 kernel.  Hence: every operand fragment is loaded by exactly one wave of the workgroup, and a
//     workgroup covers 64 frames (NT = 2) once 32-frame workgroups would outnumber the CUs;
//   * A  wave (branch, kq): both output tiles of one branch over a quarter of K = 5 taps x 64 channels (5 K blocks;
//        each A fragment feeds NT frame tiles, each B fragment both output tiles), operands two K blocks ahead in
//        registers; the quarters meet in LDS in two steps (fixed order);
//     G  gate -> PReLU -> BN by all 512 threads, split once, written as the B operand of conv2 (rows padded to 144 B);
//     B  conv2: wave w = output channels 32w..32w+31 (+ bias + residual), x' stored;
//     C  next block's conv1: that accumulator tile is split in registers and used as the B operand (K order
//        rho_bf16, packing.pack_s3_chain); eight partial sums meet in LDS in a fixed order; the result is transformed
//        for the NEXT block's two branches, split and stored as hs_out (512 contiguous bytes per 32 lanes).
//   * MODE 1 is phase C alone on x (the conv1 of the first block of the stack).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

namespace {

// Cache policy of the x' and hs stores: 17 = sc0 sc1 (written through to memory while the workgroup is still running, instead
// of staying dirty in the XCD's L2 until the end-of-kernel release).  Measured per forward (18 blocks, B=32, T=401):
// 0: 496 us, 2 (nt): 517, 17: 479, 19: 551.
#ifndef TCM2_ST_AUX
#define TCM2_ST_AUX 17
#endif
constexpr int HS_PAD = 64;         // zero frames in front of frame 0 of hs (2 x the largest dilation), and behind frame T-1
constexpr int GL_ROW = 144;        // bytes of one frame of one plane of the conv2 operand in LDS (128 + 16 pad)

__device__ __forceinline__ float prelu2(float v, float slope) { return v > 0.f ? v : slope * v; }

// NP = 3: exact three-way splits, six products; NP = 1: plain bf16 operands, one product (the opt-in bf16 mode, round 3);
// NP = 2 (round 4): fp16 hi + lo of the power-of-two scaled operands, three f16 products (gconv_common.h: split8h)
template <int NP>
__device__ __forceinline__ f32x16 mfma6r(const uint4 (&a)[NP], const uint4 (&b)[NP], f32x16 acc) {
  if constexpr (NP == 2) {
    acc = mfma_f16(a[1], b[0], acc);
    acc = mfma_f16(a[0], b[1], acc);
    acc = mfma_f16(a[0], b[0], acc);
  } else if constexpr (NP == 3) {
    acc = mfma_bf16(a[0], b[2], acc);
    acc = mfma_bf16(a[2], b[0], acc);
    acc = mfma_bf16(a[1], b[1], acc);
    acc = mfma_bf16(a[0], b[1], acc);
    acc = mfma_bf16(a[1], b[0], acc);
    acc = mfma_bf16(a[0], b[0], acc);
  } else {
    acc = mfma_bf16(a[0], b[0], acc);
  }
  return acc;
}

__device__ __forceinline__ uint32_t pack_bf16_rne(const float a, const float b) {   // v_cvt_pk_bf16_f32
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  union { bf16x2 h; uint32_t u; } c;
  c.h = __builtin_convertvector(v, bf16x2);
  return c.u;
}
template <int NP>
__device__ __forceinline__ void split8n(const float (&x)[8], uint4 (&p)[NP], const float sc = 1.0f) {
  if constexpr (NP == 3) split8(x, p[0], p[1], p[2]);
  else if constexpr (NP == 2) split8h(x, sc, p[0], p[1]);
  else p[0] = make_uint4(pack_bf16_rne(x[0], x[1]), pack_bf16_rne(x[2], x[3]), pack_bf16_rne(x[4], x[5]), pack_bf16_rne(x[6], x[7]));
}

template <int NP>
__device__ __forceinline__ void split4n(const float (&x)[4], uint2 (&p)[NP], const float sc = 1.0f);
// four values -> their three bf16 planes, 8 bytes each (element i in half i & 1 of dword i >> 1)
__device__ __forceinline__ void split4(const float (&x)[4], uint2& p1, uint2& p2, uint2& p3) {
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
  p1 = make_uint2(q[0][0], q[0][1]);
  p2 = make_uint2(q[1][0], q[1][1]);
  p3 = make_uint2(q[2][0], q[2][1]);
}

// four values * sc -> fp16 hi / lo planes, 8 bytes each
__device__ __forceinline__ void split4h(const float (&x)[4], const float sc, uint2& p1, uint2& p2) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  uint32_t q1[2], q2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x2 t = {x[2 * i] * sc, x[2 * i + 1] * sc};
    union { f16x2 h; uint32_t u; } hi, lo;
    hi.h = __builtin_convertvector(t, f16x2);
    const f32x2 r = t - __builtin_convertvector(hi.h, f32x2);
    lo.h = __builtin_convertvector(r, f16x2);
    q1[i] = hi.u;
    q2[i] = lo.u;
  }
  p1 = make_uint2(q1[0], q1[1]);
  p2 = make_uint2(q2[0], q2[1]);
}

template <int NP>
__device__ __forceinline__ void split4n(const float (&x)[4], uint2 (&p)[NP], const float sc) {
  if constexpr (NP == 3) split4(x, p[0], p[1], p[2]);
  else if constexpr (NP == 2) split4h(x, sc, p[0], p[1]);
  else p[0] = make_uint2(pack_bf16_rne(x[0], x[1]), pack_bf16_rne(x[2], x[3]));
}

__device__ long long* g_trace = nullptr;   // PDSE_TCM2_TRACE=1 (diagnostic): [workgroup][wave][8] clock stamps
#define STAMP(i)                                                                                       \
  do {                                                                                                 \
    if (trace && lane == 0) trace[((blockIdx.y * gridDim.x + blockIdx.x) * 8 * TW + wv) * 8 + (i)] = (i) == 0 ? wall_clock64() : clock64(); \
  } while (0)

// NT = frame tiles of 32 per wave, TW = teams of 8 waves per workgroup (workgroup = 32 NT TW frames of one utterance).
// The teams of a workgroup run the same instruction stream on neighbouring frames between the same barriers, so
// their weight requests reach the CU's L1 together.
// One residual block for the frame tile(s) of this workgroup.  HSA: cache policy of the hs loads - 0 for one launch per block, 16
// (sc1) inside the persistent stack kernel, where hs was written by other workgroups of the SAME launch (tcm2s_kernel below).
struct tcm2_nowait {
  __device__ __forceinline__ void operator()() const {}
};

// hs_ready(): called once, by every thread, after the block's first weight requests have gone out and before the first hs request
// (the stack kernel waits for the neighbours' progress counters there, with the weight fetch already in flight).
template <int MODE, int NT, int TW, int NP, int HSA, typename Wait = tcm2_nowait>
__device__ __forceinline__ void tcm2_block(const pdse_tcm2_desc& d, float* const par, float (*const part_)[4][64][33],
                                           char (*const gls_)[NP * 32 * GL_ROW], const Wait hs_ready = Wait()) {
  int tid_ = threadIdx.x & 511;                                                  // within the team
  if constexpr (HSA != 0) asm volatile("" : "+v"(tid_));   // stack kernel: per-lane indices are recomputed per block, not kept live (and spilled) across the block loop
  const int tid = tid_;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = wv >> 3, wave = wv & 7;
  auto part = part_ + team * NT;
  auto gls = gls_ + team * NT;
  const int col = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, t0 = (blockIdx.x * TW + team) * (32 * NT), T = d.T, TP = T + 2 * HS_PAD;
  const bool chain = d.hs_out != nullptr;
  // NP == 2 (f16x2): planes hold value * 2^PDSE_F16_ACT_EXP, the weights of the branches / conv2 / the next conv1 were scaled by
  // 2^qexp[0..2] on the host; an accumulator holds (true value) * 2^(PE + q) and is brought back where it meets fp32 operands
  constexpr int PE = NP == 2 ? PDSE_F16_ACT_EXP : 0;
  [[maybe_unused]] const float s_pl = pow2i(PE), s_A = pow2i(-(PE + (NP == 2 ? d.qexp[0] : 0))), s_2 = pow2i(-(PE + (NP == 2 ? d.qexp[1] : 0))),
                               s_N = pow2i(-(PE + (NP == 2 ? d.qexp[2] : 0)));
#ifdef PDSE_DIAG
  long long* trace = (HSA == 0) ? g_trace : nullptr;
#else
  long long* trace = nullptr;
#endif
  STAMP(0);
  STAMP(1);

  // the parameter table is requested first and written to LDS behind the first batch of operand requests
  const float par_a = d.par[tid], par_b = d.par[512 + (tid < 320 ? tid : 0)];
  auto store_par = [&]() {
    if (TW > 1 && team) return;
    par[tid] = par_a;
    if (tid < 320) par[512 + tid] = par_b;
  };

  bool tlive[NT];
  int lrow[NT];        // lane offset of accumulator rows rho(r,h) in [256][T]
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int t = t0 + 32 * n + col;
    tlive[n] = t < T;
    lrow[n] = 4 * hh * T + (tlive[n] ? t : T - 1);
  }
  f32x16 a2[NT];
  if constexpr (MODE == 0) {
    // -------------------------------------------------------------- A: dilated branches
    const int br = wave & 1, kq = wave >> 1;
    const int plane = TP * 16;                                 // bytes of one fragment plane of one (kb, kg)
    // raw buffer access (gconv_common.h): the slab of (utterance, branch) and the weights are resources in scalar
    // registers, a lane holds ONE 32-bit byte offset per frame tile, and the tap is a scalar offset.  (Until round 3 the
    // five per-tap lane offsets sat in an array indexed by the wave-uniform tap: hipcc put it in scratch, and each
    // scratch load's s_waitcnt made the wave wait for the other slot's operands before it asked for the next ones.)
    const __amdgpu_buffer_rsrc_t r_hs = make_rsrc((const char*)d.hs + (size_t)(b * 2 + br) * (8 * NP) * plane, (uint32_t)(8 * NP * plane));
    const __amdgpu_buffer_rsrc_t r_wa = make_rsrc((const uint4*)d.wbr + (size_t)(br * 2 * 20 + 5 * kq) * NP * 64,
                                                  (uint32_t)((2 * 20 - 5 * kq) * NP * 1024));   // [br][2 mi][20][3][64]
    int roff[NT];       // lane offset of tap 0: this lane half's planes + frame - 2 dil (the margins are zeros; a lane
                        // without a frame reads those of frame T-1 and its column is never stored)
#pragma unroll
    for (int n = 0; n < NT; ++n) roff[n] = hh * NP * plane + ((tlive[n] ? t0 + 32 * n + col : T - 1) - 2 * d.dil + HS_PAD) * 16;
    constexpr int D = 2;   // K blocks in flight (a slot is requested again as soon as its MFMAs have issued)
    uint4 qa[D][2][NP], qb[D][NT][NP];
    // workgroups walk their five K blocks in an order rotated by the frame tile index: neighbours in time do not ask L2
    // for the same weight lines at the same moment (-6 % on phase A); a function of the frame tile only, so an
    // utterance's result does not depend on its place in the batch
    const int js = (int)(blockIdx.x % 5);
    auto request_w = [&](int i) {
      const int slot = i % D;
      const int j = (i + js) % 5;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int p = 0; p < NP; ++p) qa[slot][mi][p] = bload16(r_wa, lane * 16, ((mi * 20 + j) * NP + p) * 1024);
    };
    auto request_h = [&](int i) {
      const int slot = i % D;
      const int j = (i + js) % 5;
      const int kbi = 5 * kq + j;                              // wave-uniform: tap = kbi >> 2, channel block = kbi & 3
      const int hq = (kbi & 3) * (2 * NP) * plane + (kbi >> 2) * d.dil * 16;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int p = 0; p < NP; ++p) qb[slot][n][p] = bload16a<HSA>(r_hs, roff[n], hq + p * plane);
    };
    auto request = [&](int i) {
      request_w(i);
      request_h(i);
    };
    if constexpr (HSA != 0) {   // stack kernel: the weights of the first K blocks fly while the neighbours' counters are awaited
#pragma unroll
      for (int j = 0; j < D; ++j) request_w(j);
      __builtin_amdgcn_sched_barrier(0);
      hs_ready();
#pragma unroll
      for (int j = 0; j < D; ++j) {
        request_h(j);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      request(j);
      __builtin_amdgcn_sched_barrier(0);   // slot 0's operands are requested first (hipcc interleaved the two slots)
    }
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][n][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[mi][n] = mfma6r<NP>(qa[j % D][mi], qb[j % D][n], acc[mi][n]);
      __builtin_amdgcn_sched_barrier(0);
      if (j + D < 5) request(j + D);
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(2);
    // conv2 weights and the residual: in flight across the gate phase
    const __amdgpu_buffer_rsrc_t r_w2 = make_rsrc((const uint4*)d.wc2 + (size_t)wave * 4 * NP * 64, (uint32_t)(4 * NP * 1024));
    uint4 w2[4][NP];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int p = 0; p < NP; ++p) w2[kb][p] = bload16(r_w2, lane * 16, (kb * NP + p) * 1024);
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.x + ((size_t)b * 256 + 32 * wave) * T, (uint32_t)(32 * T * 4));
    float xres[NT][16];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) xres[n][r] = bload4(r_x, lrow[n] * 4, ((r & 3) + 8 * (r >> 2)) * T * 4);
    __builtin_amdgcn_sched_barrier(0);
    store_par();
    // K quarters 2, 3 -> LDS; quarters 0, 1 add theirs: part[n][2 br + (kq & 1)] = quarter (kq & 1) + quarter (kq & 1) + 2
    if (kq >= 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[n][br * 2 + (kq & 1)][32 * mi + rho(r, hh)][col] = acc[mi][n][r];
    }
    __syncthreads();
    if (kq < 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* pr = &part[n][br * 2 + kq][32 * mi + rho(r, hh)][col];
            *pr = acc[mi][n][r] + *pr;
          }
    }
    __syncthreads();   // partial sums and the parameter table
    STAMP(3);

    // -------------------------------------------------------------- G: gate, PReLU, BN, split
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int f = tid & 31, cg = tid >> 5;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * cg + i;
        const f32x4 gp = *(const f32x4*)&par[4 * c];
        float m = part[n][0][c][f] + part[n][1][c][f];
        float k = part[n][2][c][f] + part[n][3][c][f];
        if constexpr (NP == 2) m *= s_A, k *= s_A;
        m += gp[0];
        k += gp[1];
        v[i] = gp[2] * prelu2(m * sigmoid_f(k), d.slope2) + gp[3];
      }
      uint2 pp[NP];
      split4n<NP>(v, pp, s_pl);
      char* g = gls[n] + f * GL_ROW + cg * 8;
#pragma unroll
      for (int p = 0; p < NP; ++p) *(uint2*)(g + p * 32 * GL_ROW) = pp[p];
    }
    __syncthreads();
    STAMP(4);

    // -------------------------------------------------------------- B: conv2 + bias + residual
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[n][r] = 0.f;
      const char* gb = gls[n] + col * GL_ROW + hh * 16;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        uint4 bp[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) bp[p] = *(const uint4*)(gb + p * 32 * GL_ROW + kb * 32);
        a2[n] = mfma6r<NP>(w2[kb], bp, a2[n]);
      }
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const f32x4 bb = *(const f32x4*)&par[256 + 32 * wave + 8 * r4 + 4 * hh];   // rows (r&3) = 0..3 are consecutive
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (NP == 2) a2[n][4 * r4 + i] = a2[n][4 * r4 + i] * s_2 + (bb[i] + xres[n][4 * r4 + i]);
          else a2[n][4 * r4 + i] += bb[i] + xres[n][4 * r4 + i];
        }
    }
  } else {
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.x + ((size_t)b * 256 + 32 * wave) * T, (uint32_t)(32 * T * 4));
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[n][r] = bload4(r_x, lrow[n] * 4, ((r & 3) + 8 * (r >> 2)) * T * 4);
    store_par();
  }

  STAMP(5);
  // next conv1's weights: requested before x' is stored
  uint4 wn[2][2][NP];
  if (chain) {
    const __amdgpu_buffer_rsrc_t r_wn = make_rsrc((const uint4*)d.wn1 + (size_t)(2 * wave) * NP * 64, (uint32_t)((32 - 2 * wave) * NP * 1024));   // [2 mo][16 blocks][3][64]
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) wn[mo][s][p] = bload16(r_wn, lane * 16, ((mo * 16 + s) * NP + p) * 1024);
  }
  if constexpr (MODE == 0) {
    const __amdgpu_buffer_rsrc_t r_xo = make_rsrc(d.x_out + ((size_t)b * 256 + 32 * wave) * T, (uint32_t)(32 * T * 4));
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (tlive[n]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bstore4<TCM2_ST_AUX>(a2[n][r], r_xo, lrow[n] * 4, ((r & 3) + 8 * (r >> 2)) * T * 4);
      }
  }
  if (!chain) return;   // uniform over the grid

  // ---------------------------------------------------------------- C: next block's conv1 on the accumulator tile
  f32x16 a1[NT][2];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[n][mo][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = a2[n][8 * s + j];
      uint4 bp[NP];
      split8n<NP>(xv, bp, s_pl);
      a1[n][0] = mfma6r<NP>(wn[0][s], bp, a1[n][0]);
      a1[n][1] = mfma6r<NP>(wn[1][s], bp, a1[n][1]);
    }
  STAMP(6);
  // part is free: every thread passed the barrier behind G
  if (wave >= 4) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[n][wave - 4][32 * mo + rho(r, hh)][col] = a1[n][mo][r];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* pr = &part[n][wave][32 * mo + rho(r, hh)][col];
          *pr = a1[n][mo][r] + *pr;
        }
  }
  __syncthreads();
  // 8 channels (one fragment half) x one frame per thread: 16-byte stores, 512 contiguous bytes per 32 lanes
  const __amdgpu_buffer_rsrc_t r_ho = make_rsrc((char*)d.hs_out + (size_t)(b * 2) * (8 * NP) * (TP * 16), (uint32_t)(16 * NP * TP * 16));
  for (int item = tid; item < NT * 256; item += 512) {
    const int n = item >> 8, cg = (item >> 5) & 7, f = item & 31;
    const int t = t0 + 32 * n + f;
    if (t < T) {
      float vm[8], vk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = 8 * cg + i;
        float s = (part[n][0][c][f] + part[n][1][c][f]) + (part[n][2][c][f] + part[n][3][c][f]);
        if constexpr (NP == 2) s *= s_N;
        s += par[512 + c];
        const f32x4 xp = *(const f32x4*)&par[576 + 4 * c];
        vm[i] = xp[0] * prelu2(s, d.slope_main_next) + xp[1];
        vk[i] = xp[2] * prelu2(s, d.slope_mask_next) + xp[3];
      }
      const int plane = TP * 16;
      const uint32_t ho = (uint32_t)(cg * NP * plane + (t + HS_PAD) * 16);
      uint4 pq[NP];
      split8n<NP>(vm, pq, s_pl);
#pragma unroll
      for (int p = 0; p < NP; ++p) bstore16<TCM2_ST_AUX>(pq[p], r_ho, ho, p * plane);
      split8n<NP>(vk, pq, s_pl);
#pragma unroll
      for (int p = 0; p < NP; ++p) bstore16<TCM2_ST_AUX>(pq[p], r_ho, ho, (8 * NP + p) * plane);
    }
  }
  STAMP(7);
}

template <int MODE, int NT, int TW, int NP = 3>
__global__ __launch_bounds__(512 * TW, NT == 1 ? 4 : 2) void tcm2_kernel(const pdse_tcm2_desc d) {
  // par: [64][4] main bias, mask bias, BN scale, BN shift of the gate | [256] conv2 bias | [64] next conv1 bias |
  //      [64][4] next block's input transforms: main scale, shift, mask scale, shift
  __shared__ __attribute__((aligned(16))) float par[832];
  __shared__ float part_[TW * NT][4][64][33];                                    // A: [2 branch + kh]; C: partial sums of four waves
  __shared__ __attribute__((aligned(16))) char gls_[TW * NT][NP * 32 * GL_ROW];   // conv2's B operand: [plane][frame][64 + 8 bf16]
  tcm2_block<MODE, NT, TW, NP, 0>(d, par, part_, gls_);
}

// ---------------------------------------------------------------------------------------------------------------
// [r4] The 18 residual blocks of the TCM stack (model/diff3.py:215-277) as ONE launch (the first block's conv1 stays the small
// mode-1 launch in front of it: with both block forms in one kernel hipcc spilled 14 registers at the 128 that two workgroups
// per CU allow).
// One launch per block leaves ~9 us of every 25 us outside the waves (launch boundary, ramp, the end-of-kernel write-back), 19
// times per forward.  What a block needs from other workgroups is only the bottleneck tensor hs of the frame tiles within
// +-2 (dilation <= 32, 32-frame tiles) of the SAME utterance; the residual stream x is private to a tile.  So every workgroup
// (utterance b, tile f) walks the blocks itself: before block i it waits until tiles f-2..f+2 of its utterance have published
// block i-1 (progress counters, one per (b, f): the same wait also keeps a neighbour from still READING the hs buffer this block
// is about to overwrite - the ping-pong), and after its stores have drained it publishes block i.
//   * hs is written through (sc0 sc1 stores, as before) and, in this kernel, read with sc1 loads (never from the CU's L1); the
//     counter is an sc1 store behind every wave's s_waitcnt vmcnt(0) and the workgroup barrier, polled with sc1 loads by the
//     five lanes that own a neighbour each (MI355X_MICROARCH.md, valid forms: payload and flag sc1 on both sides, no fence);
//   * the counters are zeroed by a memset node in front of the launch; every wait is bounded (a workgroup that gives up records
//     the block in d.status and leaves, its neighbours follow), so a launch cannot hang;
//   * liveness needs the 13 workgroups of an utterance to become resident, not the whole grid: their dependencies never leave
//     the utterance, so a partially resident launch (other kernels still draining, other batches in flight) keeps completing
//     utterances and freeing slots.
// Same arithmetic, same order as one launch per block: results are bit-identical (tests/test_gpu_round4.py).
// ---------------------------------------------------------------------------------------------------------------
#define TCM2S_SPINS 60000   // polls of one neighbour counter before a workgroup gives up (~50-100 ms)

template <int NP>
__global__ __launch_bounds__(512, 4) void tcm2s_kernel(const pdse_tcm2s_desc s) {
  __shared__ __attribute__((aligned(16))) float par[832];
  __shared__ float part_[1][4][64][33];
  __shared__ __attribute__((aligned(16))) char gls_[1][NP * 32 * GL_ROW];
  __shared__ int dead;
  typedef __attribute__((address_space(1))) int gint;
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, ntiles = gridDim.x, b = blockIdx.y;
  gint* const flags = (gint*)(s.flags + (size_t)b * ntiles);
  if (tid == 0) dead = 0;
  for (int i = 0; i < s.n; ++i) {
    bool gave_up = false;
    auto wait = [&]() {
      if (i > 0 && tid < 5) {
        const int f = tile - 2 + tid;
        if (f >= 0 && f < ntiles && f != tile) {
          int spins = 0;
          while (__hip_atomic_load(flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) {
            if (++spins > TCM2S_SPINS) {
              dead = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
        }
      }
      __syncthreads();   // the neighbours' block i - 1 is visible to every wave
      gave_up = dead != 0;
    };
    tcm2_block<0, 1, 1, NP, 16>(s.blk[i], par, part_, gls_, wait);
    // a workgroup that gave up still ran the block (on whatever hs held): every barrier of the block was reached by all waves;
    // it records the failure and leaves before publishing, so its neighbours give up at their next wait
    if (gave_up) {
      if (tid == 0) atomicCAS(s.status, 0, i + 1);
      return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its x' and hs stores have left
    __syncthreads();                                    // ... and the block's LDS use is over before the next block's begins
    if (tid == 0) __hip_atomic_store(flags + tile, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int NT, int TW, int NP>
int launch_tcm2(const pdse_tcm2_desc* d, hipStream_t s) {
  const dim3 grid((d->T + 32 * NT * TW - 1) / (32 * NT * TW), d->B);
  static long long* tbuf = nullptr;
  const size_t nst = (size_t)grid.x * grid.y * 64 * TW;
  // (diagnostic builds only) the trace buffer holds 65536 * 64 stamps: larger launches are not traced
  static const bool tracing_env = PDSE_DIAG_ENV("PDSE_TCM2_TRACE") != nullptr;
  const bool tracing = tracing_env && nst <= (size_t)65536 * 64;
  if (tracing) {
    if (!tbuf) {
      (void)hipMalloc(&tbuf, 65536 * 64 * sizeof(long long));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &tbuf, sizeof(tbuf));
    }
    (void)hipMemsetAsync(tbuf, 0, nst * sizeof(long long), s);
  }
  if (d->mode == 1) hipLaunchKernelGGL((tcm2_kernel<1, NT, TW, NP>), grid, dim3(512 * TW), 0, s, *d);
  else hipLaunchKernelGGL((tcm2_kernel<0, NT, TW, NP>), grid, dim3(512 * TW), 0, s, *d);
  if (tracing) {   // diagnostic: per-phase shader-clock averages over all waves, and the spread of start times (100 MHz clock)
    (void)hipStreamSynchronize(s);
    long long* h = (long long*)malloc(nst * sizeof(long long));
    (void)hipMemcpy(h, tbuf, nst * sizeof(long long), hipMemcpyDeviceToHost);
    double sum[8] = {0};
    long long w0 = -1, w1 = 0;
    const size_t nw = nst / 8;
    for (size_t i = 0; i < nw; ++i) {
      const long long* q = h + i * 8;
      if (w0 < 0 || q[0] < w0) w0 = q[0];
      if (q[0] > w1) w1 = q[0];
      for (int k = 2; k < 8; ++k) sum[k] += q[k] ? (double)(q[k] - q[1]) : 0.0;
    }
    fprintf(stderr, "tcm2 trace mode %d NT %d TW %d dil %d: start spread %.2f us; cycles since wave start:", d->mode, NT, TW, d->dil, (w1 - w0) * 0.01);
    for (int k = 2; k < 8; ++k) fprintf(stderr, " %.0f", sum[k] / nw);
    fprintf(stderr, "\n");
    free(h);
  }
  return pdse_check_launch("tcm2");
}

}  // namespace

int pdse_tcm2s_launch(const pdse_tcm2s_desc* d, hipStream_t s) {
  REQ(d && d->flags && d->status, "tcm2s: null pointer");
  REQ(d->n >= 1 && d->n <= PDSE_TCM2S_MAX, "tcm2s: 1 .. PDSE_TCM2S_MAX blocks");
  const int B = d->blk[0].B, T = d->blk[0].T, np = d->blk[0].np ? d->blk[0].np : 3;
  REQ(B > 0 && B <= 65535 && T > 0 && (np == 3 || np == 2 || np == 1), "tcm2s: bad sizes");
  for (int i = 0; i < d->n; ++i) {
    const pdse_tcm2_desc& k = d->blk[i];
    REQ(k.x && k.par && k.B == B && k.T == T && (k.np ? k.np : 3) == np, "tcm2s: blocks of one stack share B, T and the plane count");
    REQ(k.mode == 0, "tcm2s: residual blocks only (mode 0; the first block's conv1 is its own pdse_tcm2_bf16x3 launch)");
    REQ(k.dil > 0 && 2 * k.dil <= HS_PAD, "tcm2s: dilation <= 32");
    REQ(!k.hs_out || k.wn1, "tcm2s: the chained conv1 needs its weights");
    REQ(k.hs && k.x_out && k.wbr && k.wc2 && k.hs_out != k.hs, "tcm2s: bad residual block");
    REQ(i == 0 || k.hs == d->blk[i - 1].hs_out, "tcm2s: block i reads the hs that block i - 1 wrote (the wait protocol assumes it)");
    // the wait in front of block i covers the neighbours' block i - 1 only: block i may overwrite nothing but the buffer block
    // i - 1 READ, i.e. the one block i - 2 wrote (two buffers, strictly alternating)
    REQ(!k.hs_out || k.hs_out == (i >= 2 ? d->blk[i - 2].hs_out : (i == 1 ? d->blk[0].hs : k.hs_out)), "tcm2s: the hs buffers must alternate strictly");
    REQ(k.x != k.x_out, "tcm2s: x and x_out of a block differ (other tiles never read them, but a tile's waves do)");
  }
  const int ntiles = (T + 31) / 32;
  if (pdse_check_hip(hipMemsetAsync(d->flags, 0, (size_t)B * ntiles * sizeof(int), s), "tcm2s: memset")) return 1;
  const dim3 grid(ntiles, B);
  if (np == 1) hipLaunchKernelGGL((tcm2s_kernel<1>), grid, dim3(512), 0, s, *d);
  else if (np == 2) hipLaunchKernelGGL((tcm2s_kernel<2>), grid, dim3(512), 0, s, *d);
  else hipLaunchKernelGGL((tcm2s_kernel<3>), grid, dim3(512), 0, s, *d);
  return pdse_check_launch("tcm2s");
}

int pdse_tcm2_launch(const pdse_tcm2_desc* d, hipStream_t s) {
  REQ(d && d->x && d->par, "tcm2: null pointer");
  REQ(d->mode == 0 || d->mode == 1, "tcm2: mode is 0 (residual block) or 1 (input convolution only)");
  REQ(d->B > 0 && d->B <= 65535 && d->T > 0 && d->dil > 0 && 2 * d->dil <= HS_PAD, "tcm2: bad sizes (dilation <= 32)");
  REQ(!d->hs_out || d->wn1, "tcm2: the chained conv1 needs its weights");
  if (d->mode == 1) {
    REQ(d->hs_out, "tcm2: mode 1 writes hs_out");
  } else {
    REQ(d->hs && d->x_out && d->wbr && d->wc2, "tcm2: null pointer");
    REQ(d->hs_out != d->hs, "tcm2: hs_out must not alias hs (other workgroups gather from hs)");
  }
  // workgroup shape 10 NT + TW.  Measured at B=32, T=401 (us per forward, 18 blocks): 11 -> 592, 21 (64 frames, two
  // tiles per wave) -> 730, 12 (64 frames, two teams of 8 waves) -> 769: the shapes that halve the weight bytes per
  // CU lose more to their longer dependent chains / simultaneous identical requests.  PDSE_TCM2_SHAPE: ablation.
  REQ(d->np == 0 || d->np == 3 || d->np == 2 || d->np == 1, "tcm2: np is 3 (exact splits; 0 means 3), 2 (f16x2) or 1 (plain bf16)");
  if (d->np == 1) return launch_tcm2<1, 1, 1>(d, s);   // the opt-in bf16 mode
  if (d->np == 2) {
    REQ(d->qexp[0] >= -40 && d->qexp[0] <= 40 && d->qexp[1] >= -40 && d->qexp[1] <= 40 && d->qexp[2] >= -40 && d->qexp[2] <= 40, "tcm2: qexp out of range");
    return launch_tcm2<1, 1, 2>(d, s);
  }
  static const int force = PDSE_DIAG_ENV("PDSE_TCM2_SHAPE") ? atoi(PDSE_DIAG_ENV("PDSE_TCM2_SHAPE")) : 0;
  const int shape = force ? force : 11;
  switch (shape) {
    case 12: return launch_tcm2<1, 2, 3>(d, s);
    case 21: return launch_tcm2<2, 1, 3>(d, s);
    default: return launch_tcm2<1, 1, 3>(d, s);
  }
}
