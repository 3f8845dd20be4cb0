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
//   * the second profile (profiles/r01_pmc_sq_v2.txt) showed the MFMA pipe 40 % busy with the
//     waves stalled at issue: two global_load_dword per MFMA saturate the vector-memory path.
//     Weight fragments are therefore packed 4 k-steps deep and fetched with one 16-byte load
//     per lane, and the encoder's pad frame is materialised in the input instead of being
//     patched in by a second load per tap (0.75 load instructions per MFMA instead of 2);
//   * the launcher splits output-channel tiles over more waves when a layer has few
//     positions (TCM: 13 tiles per utterance), trading activation re-reads for occupancy.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

#ifndef PDSE_WAVES_PER_EU
#define PDSE_WAVES_PER_EU 2   // at least two resident waves per SIMD: caps the allocation at 256 VGPRs (the dual-phase tail reached 272)
#endif
#ifndef PDSE_ABLATE
#define PDSE_ABLATE 0   // diagnostic builds: 1 no activation loads, 2 no weight loads, 4 plain epilogue
#endif

constexpr int popc(int m) { return m ? (m & 1) + popc(m >> 1) : 0; }
constexpr int rank_of(int m, int tap) { return popc(m & ((1 << tap) - 1)); }   // index of `tap` among the set bits

template <int N, int CP, int MT, int N1>
struct Chunk {
  float v[N];           // raw activation per k-step (lane = position, half = channel parity)
  float4 a0[N / 4][MT];  // A fragments: one 16-byte load covers 4 consecutive k-steps
  float4 a1[N / 4][MT];
  float4 a2[N1 > 0 ? N1 / 4 : 1];   // odd-bin phase of a dual-phase transposed conv (MT == 1)
  float4 a3[N1 > 0 ? N1 / 4 : 1];
  float xs0[CP], xh0[CP], xs1[CP], xh1[CP];  // load-transform parameters (TCM only, XF != 0)
};

__device__ __forceinline__ float f4get(const float4& q, const int i) {
  return i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w));
}

// XF: 0 no load transform; 1 one PReLU->BN set for both accumulators; 2 one set per accumulator
// PP: ping-pong two register sets (layers with few waves per SIMD need the ILP); otherwise one
//     set per chunk and the other resident waves hide the load latency.
// P1MASK != 0: dual-phase transposed conv — taps in the mask also feed the odd output bins.
template <int EPI, int MT, int NT, int CP, bool SRC2, int XF, bool PP, int P1MASK = 0, bool NX = false>
__global__ __launch_bounds__(256, PDSE_WAVES_PER_EU) void gconv2_kernel(const pdse_gconv_desc d) {
  constexpr bool DUAL = (EPI != PDSE_EPI_LINEAR);
  constexpr int N = NT * CP;
  constexpr int NT1 = popc(P1MASK), N1 = NT1 * CP;
  static_assert(N % 4 == 0 && N1 % 4 == 0, "a chunk must hold whole 4-k-step weight groups");
  static_assert(P1MASK == 0 || (EPI == PDSE_EPI_BIGLU && MT == 1), "dual phase is a BIGLU feature");
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
  // A workgroup covers four 32-position tiles; launches with few positions per batch item (P = 401: 13 tiles in
  // 4 workgroups) would run up to three fully masked waves through the whole K loop.  They leave here - except in
  // the BIGLU kernels, whose waves meet at a barrier before the tail.
  if constexpr (EPI != PDSE_EPI_BIGLU) {
    if ((blockIdx.x * 4 + wave) * 32 >= P) return;
  }

  // BIGLU: copy the tail's operands (chain fragments, biases, folded BN) to LDS, once per
  // workgroup; the copy is in flight during the tap set-up and is fenced just before the tail
  __shared__ float tail_lds[EPI == PDSE_EPI_BIGLU ? PDSE_TAIL_FLOATS : 1];
  if constexpr (EPI == PDSE_EPI_BIGLU) {
    const int tid = threadIdx.x;
    const int w2n = d.C2 == 1 ? 32 : ((d.C2 + 31) >> 5) * 1024;
    for (int i = tid; i < 1024; i += 256) {
      tail_lds[i] = d.wlc[i];
      tail_lds[1024 + i] = d.wrc[i];
    }
    for (int i = tid; i < w2n; i += 256) tail_lds[2048 + i] = d.wc2[i];
    if (tid < 32) {
      const float bl = d.bias0[(int64_t)b * d.bias0_sb + tid], br = d.bias1[(int64_t)b * d.bias1_sb + tid];
      tail_lds[4096 + tid] = bl;
      tail_lds[4128 + tid] = br;
      tail_lds[PDSE_TAIL_B0 + tid] = d.bias0_t0 ? d.bias0_t0[(int64_t)b * d.bias0_sb + tid] : bl;
      tail_lds[PDSE_TAIL_B0 + 32 + tid] = d.bias1_t0 ? d.bias1_t0[(int64_t)b * d.bias1_sb + tid] : br;
      tail_lds[4160 + tid] = d.blc[tid];
      tail_lds[4192 + tid] = d.brc[tid];
    }
    if (tid < d.C2) {
      tail_lds[4224 + tid] = d.bc2[tid];
      tail_lds[4288 + tid] = d.post_scale ? d.post_scale[tid] : 1.0f;   // no BatchNorm: identity pair (branch-free tail)
      tail_lds[4352 + tid] = d.post_scale ? d.post_shift[tid] : 0.0f;
    }
    if (d.nx_n > 0) {   // chained next-stage 1x1 tiles + their biases for this batch item
      for (int i = tid; i < d.nx_n * 2048; i += 256) tail_lds[PDSE_TAIL_NXW + i] = d.nx_w[i];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < d.nx_n && tid < 32) tail_lds[PDSE_TAIL_NXB + 32 * i + tid] = d.nx_bias[i][(int64_t)b * d.nx_bias_sb[i] + tid];
    }
  }

  f32x16 acc0[MT], acc1[DUAL ? MT : 1];
  f32x16 acc2[1], acc3[1];   // odd-bin phase
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[m][r] = 0.f;
      if (DUAL) acc1[m][r] = 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[0][r] = acc3[0][r] = 0.f;

  // ---- per-tap gather state, computed once
  int off0[NT], off1[SRC2 ? NT : 1];
  unsigned inb_mask = 0;
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) {
    const int dt = d.tap_dt[tap], df = d.tap_df[tap];   // kernel arguments (scalar registers), not the device table
    const int tin = t + dt, fin = j * d.sf_in + df;
    const bool inb = pvalid && fin >= 0 && fin < d.Fin && tin >= 0 && tin < d.Tin;
    if (inb) inb_mask |= 1u << tap;
    off0[tap] = inb ? (int)((int64_t)b * d.in0.sb + (int64_t)tin * d.in0.st + (int64_t)fin * d.in0.sf + (int64_t)h * d.in0.sc) : 0;
    if (SRC2)
      off1[tap] = inb ? (int)((int64_t)b * d.in1.sb + (int64_t)tin * d.in1.st + (int64_t)fin * d.in1.sf + (int64_t)h * d.in1.sc) : 0;
  }
  const int kgroups = d.ksteps >> 2;   // host pads ksteps to a multiple of 4
  const float4* wq0 = reinterpret_cast<const float4*>(d.w0) + (size_t)mt0 * kgroups * 64 + lane;
  const float4* wq1 = DUAL ? reinterpret_cast<const float4*>(d.w1) + (size_t)mt0 * kgroups * 64 + lane : nullptr;
  const float4* wq2 = P1MASK ? reinterpret_cast<const float4*>(d.w2) + lane : nullptr;
  const float4* wq3 = P1MASK ? reinterpret_cast<const float4*>(d.w3) + lane : nullptr;
  const int cps0 = d.in0.C >> 1;
  const int cps1 = SRC2 ? (d.in1.C >> 1) : 0;
  const int nch0 = (cps0 + CP - 1) / CP, nch1 = (cps1 + CP - 1) / CP;
  const int nchunks = nch0 + nch1;
  // chunks are laid out back to back in the packed weights: chunk q starts at k-group q*N/4
  // (the host packs source 0's pairs padded to a multiple of CP, then source 1's)

  // Straight-line code: every load is unconditional from a clamped (always readable) address
  // and masked afterwards, so the whole chunk is one scheduling region of independent loads.
  auto issue = [&](Chunk<N, CP, MT, N1>& c, const int q) {
    const bool s1 = SRC2 && q >= nch0;
    const int cp0 = (s1 ? q - nch0 : q) * CP;
    const int cps = s1 ? cps1 : cps0;
    const int gbase = s1 ? cps0 : 0;
    const float* sp = s1 ? d.in1.ptr : d.in0.ptr;
    const int sc2 = (int)(2 * (s1 ? d.in1.sc : d.in0.sc));
#pragma unroll
    for (int g4 = 0; g4 < N / 4; ++g4) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mc = (mt0 + m < mtiles) ? m : 0;   // tiles past Cout re-read tile 0; never stored
#if PDSE_ABLATE & 2
        c.a0[g4][m] = make_float4(q * 1e-9f, 0.1f, 0.2f, 0.3f);   // diagnostic: no weight loads
        if (DUAL) c.a1[g4][m] = make_float4(q * 2e-9f, 0.1f, 0.2f, 0.3f);
#else
        c.a0[g4][m] = wq0[((size_t)mc * kgroups + (size_t)q * (N / 4) + g4) * 64];
        if (DUAL) c.a1[g4][m] = wq1[((size_t)mc * kgroups + (size_t)q * (N / 4) + g4) * 64];
#endif
      }
    }
    if constexpr (P1MASK != 0) {
#pragma unroll
      for (int g4 = 0; g4 < N1 / 4; ++g4) {
        c.a2[g4] = wq2[((size_t)q * (N1 / 4) + g4) * 64];
        c.a3[g4] = wq3[((size_t)q * (N1 / 4) + g4) * 64];
      }
    }
#pragma unroll
    for (int cc = 0; cc < CP; ++cc) {
      // pairs past the end of a source are clamped to its last pair: the host packs ZERO weight
      // rows there, so the (finite) value read is multiplied away — no uniform branch, which
      // hipcc would turn into per-load control flow with a vmcnt(0) at every join
      const int cp = min(cp0 + cc, cps - 1);
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int o = (SRC2 && s1) ? off1[tap] : off0[tap];
#if PDSE_ABLATE & 1
        c.v[cc * NT + tap] = __builtin_amdgcn_readfirstlane(o) * 1e-9f + 0.5f;   // diagnostic: no activation loads
#else
        c.v[cc * NT + tap] = sp[o + cp * sc2];   // raw: masking happens at use, or the wait would sit here
#endif
      }
      if constexpr (XF != 0) {
        const int ci = 2 * (gbase + cp) + h;
        c.xs0[cc] = d.xf_scale0[ci];
        c.xh0[cc] = d.xf_shift0[ci];
        if constexpr (XF == 2) {
          c.xs1[cc] = d.xf_scale1[ci];
          c.xh1[cc] = d.xf_shift1[ci];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the chunk's loads together, ahead of the MFMAs
  };

  auto consume = [&](Chunk<N, CP, MT, N1>& c, const int q) {
    // ELU on load exists only for the GCRN decoder's skip source (gcrn.py:152-155)
    bool elu_src = false;
    if constexpr (SRC2 && EPI == PDSE_EPI_GLU) elu_src = (q >= nch0) ? d.in1.act == PDSE_ACT_ELU : d.in0.act == PDSE_ACT_ELU;
#pragma unroll
    for (int cc = 0; cc < CP; ++cc) {
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int i = cc * NT + tap;
        const bool inb = (inb_mask >> tap) & 1u;
        float v = inb ? c.v[i] : 0.f;
        if constexpr (SRC2 && EPI == PDSE_EPI_GLU) {
          const float e = expm1f(fminf(v, 0.f));   // elu(0) = 0 keeps masked lanes at zero
          v = (elu_src && v < 0.f) ? e : v;
        }
        float v0 = v, v1 = v;
        if constexpr (XF != 0) {
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
          acc0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(c.a0[i >> 2][m], i & 3), v0, acc0[m], 0, 0, 0);
          if (DUAL) acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(c.a1[i >> 2][m], i & 3), v1, acc1[m], 0, 0, 0);
        }
        if constexpr (P1MASK != 0) {
          if ((P1MASK >> tap) & 1) {   // folds after unrolling
            const int i1 = cc * NT1 + rank_of(P1MASK, tap);
            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(c.a2[i1 >> 2], i1 & 3), v0, acc2[0], 0, 0, 0);
            acc3[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(c.a3[i1 >> 2], i1 & 3), v0, acc3[0], 0, 0, 0);
          }
        }
      }
    }
  };

  if constexpr (PP) {
    // ping-pong over the chunks: loads of q+1 are in flight while q is multiplied
    Chunk<N, CP, MT, N1> ca, cb;
    issue(ca, 0);
    int q = 0;
    for (; q + 1 < nchunks; q += 2) {
      issue(cb, q + 1);
      consume(ca, q);
      if (q + 2 < nchunks) issue(ca, q + 2);
      consume(cb, q + 1);
    }
    if (q < nchunks) consume(ca, q);
  } else {
    Chunk<N, CP, MT, N1> ca;
    for (int q = 0; q < nchunks; ++q) {
      issue(ca, q);
      consume(ca, q);
    }
  }

#if PDSE_ABLATE & 4
  if (pvalid) d.out[(int64_t)b * d.out_sb + (int64_t)t * d.out_st + (int64_t)j * d.out_sf + d.out_off] = acc0[0][0] + (DUAL ? acc1[0][3] : 0.f);
#else
  if constexpr (EPI == PDSE_EPI_BIGLU) {
    __syncthreads();   // the tail operands staged at launch are in LDS by now (fencing right after the copy
                       // instead measured the same: the barrier is not what the tail costs)
    float* const sw = tail_lds;
    const pdse_tail tl{sw, sw + 1024, sw + 2048, sw + 4096, sw + 4128, sw + 4160, sw + 4192, sw + 4224,
                       sw + 4288, sw + 4352, sw + PDSE_TAIL_NXW, sw + PDSE_TAIL_NXB,
                       sw + PDSE_TAIL_B0, sw + PDSE_TAIL_B0 + 32};
    if constexpr (P1MASK != 0)   // even + odd bins of this lane, paired stores
      biglu_dual_epilogue<NX>(d, tl, acc0[0], acc1[0], acc2[0], acc3[0], b, t, j, pvalid, lane, h);
    else if (d.nx_n > 0)         // block output chained into the next stage's 1x1 convolutions
      biglu_nx_epilogue(d, tl, acc0[0], acc1[0], b, t, j, pvalid, lane, h);
    else
      gconv_epilogue<EPI, MT>(d, tl, acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles);
  } else {
    gconv_epilogue<EPI, MT>(d, tail_from_desc(d), acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles);
  }
#endif
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

// PP1/PP2/PP4: ping-pong at 1, 2, 4 channel tiles per wave (register budget decides)
template <int EPI, int NT, int CP, bool SRC2, int XF, bool PP1, bool PP2, bool PP4>
static void launch_mt(const pdse_gconv_desc* d, hipStream_t s, int mt, int gx, int mtiles) {
  const dim3 block(256);
  if constexpr (EPI == PDSE_EPI_BIGLU) {
    hipLaunchKernelGGL((gconv2_kernel<EPI, 1, NT, CP, SRC2, XF, PP1>), dim3(gx, d->B, 1), block, 0, s, *d);
  } else {
    if constexpr (EPI != PDSE_EPI_GLU) {   // GLU keeps two accumulator sets: pick_mt() never widens it beyond 2 tiles
      if (mt >= 4) {
        hipLaunchKernelGGL((gconv2_kernel<EPI, 4, NT, CP, SRC2, XF, PP4>), dim3(gx, d->B, (mtiles + 3) / 4), block, 0, s, *d);
        return;
      }
    }
    if (mt >= 2)
      hipLaunchKernelGGL((gconv2_kernel<EPI, 2, NT, CP, SRC2, XF, PP2>), dim3(gx, d->B, (mtiles + 1) / 2), block, 0, s, *d);
    else
      hipLaunchKernelGGL((gconv2_kernel<EPI, 1, NT, CP, SRC2, XF, PP1>), dim3(gx, d->B, mtiles), block, 0, s, *d);
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
  if (d->padrow != nullptr || (d->ksteps & 3) != 0) {
    pdse_set_error("gconv2: korder 1 needs ksteps % 4 == 0 and no pad row (materialise frame -1 in the input)");
    return 1;
  }
  // (CP must match packing.V2_CP: the host pads each source's pairs to a multiple of CP)
#define GO(EPI, NT, CP, SRC2, XF, PP1, PP2, PP4)                              \
  do {                                                                        \
    launch_mt<EPI, NT, CP, SRC2, XF, PP1, PP2, PP4>(d, s, mt, gx, mtiles);    \
    return pdse_check_launch("gconv2");                                       \
  } while (0)
  if (d->epi == PDSE_EPI_LINEAR) {
    if (nt == 1 && !two && xf == 0) GO(PDSE_EPI_LINEAR, 1, 8, false, 0, true, true, true);
    if (nt == 1 && !two && xf == 1) GO(PDSE_EPI_LINEAR, 1, 8, false, 1, true, true, true);
    if (nt == 1 && two && xf == 0) GO(PDSE_EPI_LINEAR, 1, 8, true, 0, true, true, true);
    if (nt == 4 && !two && xf == 0) GO(PDSE_EPI_LINEAR, 4, 2, false, 0, true, true, true);
    if (nt == 3 && !two && xf == 0) GO(PDSE_EPI_LINEAR, 3, 4, false, 0, true, true, true);   // dbaiat (1,3) convs
    if (nt == 6 && !two && xf == 0) GO(PDSE_EPI_LINEAR, 6, 2, false, 0, true, true, true);   // dbaiat dense blocks
  } else if (d->epi == PDSE_EPI_GLU) {
    if (nt == 1 && two && xf == 0) GO(PDSE_EPI_GLU, 1, 4, true, 0, true, true, true);
    if (nt == 2 && two && xf == 0) GO(PDSE_EPI_GLU, 2, 2, true, 0, true, true, true);
    if (nt == 3 && !two && xf == 0) GO(PDSE_EPI_GLU, 3, 4, false, 0, true, false, false);
    if (nt == 5 && !two && xf == 2) GO(PDSE_EPI_GLU, 5, 4, false, 2, false, false, false);
  } else if (d->epi == PDSE_EPI_BIGLU && !two && xf == 0 && d->C2 <= 64 && d->w2 != nullptr) {
    // dual-phase transposed conv: kernel (2,3) -> 4 union taps, odd bins use taps {0,2};
    //                             kernel (2,5) -> 6 union taps, odd bins use taps {0,1,3,4}
    const dim3 block(256), grid(gx, d->B, 1);
    if (d->w3 == nullptr || (d->out_sf & 1) || (d->ksteps1 & 3) || d->out_cr != 1 || !(d->C2 == 1 || d->C2 == 64)) {
      pdse_set_error("gconv2: dual phase needs w2 and w3, an even out_sf, ksteps1 % 4 == 0, out_cr 1, C2 in {1, 64}");
      return 1;
    }
    if (nt == 4 && d->p1mask == 5) {
      if (d->nx_n > 0)
        hipLaunchKernelGGL((gconv2_kernel<PDSE_EPI_BIGLU, 1, 4, 2, false, 0, true, 5, true>), grid, block, 0, s, *d);
      else
        hipLaunchKernelGGL((gconv2_kernel<PDSE_EPI_BIGLU, 1, 4, 2, false, 0, true, 5>), grid, block, 0, s, *d);
      return pdse_check_launch("gconv2");
    }
    if (nt == 6 && d->p1mask == 27 && d->nx_n == 0) {
      hipLaunchKernelGGL((gconv2_kernel<PDSE_EPI_BIGLU, 1, 6, 2, false, 0, false, 27>), grid, block, 0, s, *d);
      return pdse_check_launch("gconv2");
    }
  } else if (d->epi == PDSE_EPI_BIGLU && two && xf == 0 && d->C2 <= 64 && d->w2 == nullptr) {
    // encoder stage 1 with conv1 composed into the gather weights: (x, x_init) read directly
    if (nt == 10) GO(PDSE_EPI_BIGLU, 10, 2, true, 0, false, false, false);
  } else if (d->epi == PDSE_EPI_BIGLU && !two && xf == 0 && d->C2 <= 64) {   // tail image in LDS holds C2 <= 64
    if (nt == 2) GO(PDSE_EPI_BIGLU, 2, 4, false, 0, true, true, true);
    if (nt == 4) GO(PDSE_EPI_BIGLU, 4, 2, false, 0, true, true, true);
    if (nt == 6) GO(PDSE_EPI_BIGLU, 6, 2, false, 0, true, true, true);
    if (nt == 10) GO(PDSE_EPI_BIGLU, 10, 2, false, 0, false, false, false);   // ping-pong measured equal, 40 more VGPRs
  }
#undef GO
  pdse_set_error("gconv2: no pipelined instantiation for this (epilogue, taps, sources); pack with korder 0");
  return 1;
}
