// gconv4.hip — the GEMM-shaped gather convolutions (korder 3) on the bf16 matrix cores with exact three-way bf16
// operand splits: GCRN's gated (transposed) convolutions and LSTM input projection, DB-AIAT's dilated dense blocks.
// Same descriptors, same fp32 tensors in HBM and the same LINEAR / GLU epilogues as gconv2.hip (gconv_common.h); only
// the contraction changes (arithmetic: see gconv3.hip / packing.split_bf16x3 - six bf16 products per fp32
// multiply-add, fp32 accumulation, fp32-level accuracy at 16/6 of the fp32 MFMA rate).
//
// Unlike the eps-net blocks (gconv3.hip: K <= 192, all weights of a launch in one LDS image) these layers have K up to
// 1024 and up to 2 x 256 output channels, i.e. up to 1.5 MB of split weights: a GEMM main loop.
//   * workgroup = 8 waves = 8 tiles of 32 positions of one batch item x MT channel tiles (x 2 branches for GLU);
//   * K is walked in chunks of G4_CH 16-channel blocks of one (source, tap); the chunk's A fragments
//     ([block][branch][tile][3 planes][64 lanes] uint4) are streamed into a double-buffered LDS ring by LDS-DMA
//     (global_load_lds_dwordx4) while the previous chunk is multiplied; one barrier per chunk;
//   * B operand: lane (position, half h) gathers channels 16 cb + 8h .. +7 of its position (eight 4-byte loads per block,
//     scalar channel base + 32-bit lane offset), ELU-on-load for GCRN's skip source, exact split in registers; the next
//     chunk's requests are issued before the current chunk's matrix work;
//   * every (branch, tile) accumulator of the wave reuses the split B operand: 6 x branches x MT MFMAs per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pdse.h"
#include "pdse_internal.h"

#include "gconv_common.h"

// 16-channel K blocks per LDS chunk.  The ring (2 buffers x G4_CH x fragments x 3 KB) decides how many workgroups share
// a CU: with 4 blocks the four-fragment launches (GLU with two channel tiles, LINEAR with four) need 96 KB, i.e. ONE
// workgroup = 8 waves per CU although their registers allow two; 3 blocks = 72 KB lets two workgroups cover each
// other's chunk barriers and epilogues.
#ifndef G4_CH
#define G4_CH 3
#endif

__device__ __forceinline__ void glds16_g4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}

// One gathered activation: global_load_dword (scalar base + 32-bit lane offset) written as inline asm.  While an LDS-DMA is
// in flight hipcc puts s_waitcnt vmcnt(0) in front of the first use of ANY compiler-visible load result
// (cdna_hip_programming.md §5, "Pipelining across barriers"), which made every K block wait for the next chunk's DMA and
// gather.  The compiler does not see these loads; what orders them is the chunk barrier: a chunk's values are requested
// one barrier before they are used, and __syncthreads() drains vmcnt (the DMA is compiler-visible) before it releases.
__device__ __forceinline__ float g4_load(const float* sbase, const unsigned byte_off) {
  float v;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(sbase) : "memory");
  return v;
}
// channel-blocked sources (pdse_src.blk = 8): the lane's eight channels of a K block are 32 contiguous bytes
typedef float g4_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ g4_f32x4 g4_load16(const float* sbase, const unsigned byte_off) {
  g4_f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(sbase) : "memory");
  return v;
}
// The gathered values of one chunk.  A load written as inline asm returns before its data: between the request and the
// chunk barrier NOTHING may touch the destination registers - not even a copy - so every load writes the registers the
// values are read from later: scalars for plain sources, whole 16-byte vectors for channel-blocked ones (extracting the
// elements of a vector right behind its load made hipcc copy them at once, i.e. before they had landed, and re-use the
// vector's registers while the load was still in flight).
template <bool BLK>
struct g4_raw {
  float v[G4_CH][8];
  __device__ __forceinline__ float get(const int i, const int e) const { return v[i][e]; }
};
template <>
struct g4_raw<true> {
  g4_f32x4 v[G4_CH][2];
  __device__ __forceinline__ float get(const int i, const int e) const { return v[i][e >> 2][e & 3]; }
};
// no instruction: pins the point after which the values may be read (register-only uses could otherwise be scheduled above
// the barrier that makes them valid)
__device__ __forceinline__ void g4_landed(g4_raw<false>& raw) {
#pragma unroll
  for (int i = 0; i < G4_CH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(raw.v[i][e]));
}
__device__ __forceinline__ void g4_landed(g4_raw<true>& raw) {
#pragma unroll
  for (int i = 0; i < G4_CH; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(raw.v[i][e]));
}

// A fragments are read from the ring with ds_read_b128 written as inline asm too: the compiler cannot tell the ring
// buffer being read from the one an LDS-DMA in flight is filling, and put s_waitcnt vmcnt(0) in front of every K
// block's ring reads - each chunk then waited for the NEXT chunk's DMA and gathers before its first MFMA.  The chunk
// barrier orders the DMA of a buffer before its reads; the wait below orders the reads before their use.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // a register quad the inline asm can name (uint4 is a struct)
// the NP planes of one fragment (NP = 3: exact three-way split; NP = 1: plain bf16, the opt-in bf16 mode - korder 4)
__device__ __forceinline__ void g4_lds_read3(const unsigned addr, u32x4 (&a)[3]) {
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048"
               : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2])
               : "v"(addr)
               : "memory");
}
__device__ __forceinline__ void g4_lds_read3(const unsigned addr, u32x4 (&a)[2]) {   // f16x2: hi and lo planes
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(a[0]), "=&v"(a[1]) : "v"(addr) : "memory");
}
__device__ __forceinline__ void g4_lds_read3(const unsigned addr, u32x4 (&a)[1]) {
  asm volatile("ds_read_b128 %0, %1" : "=&v"(a[0]) : "v"(addr) : "memory");
}
template <int N, int NP>
__device__ __forceinline__ void g4_lds_wait(u32x4 (&a)[N][NP]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(a[i][p]));
}
__device__ __forceinline__ f32x16 g4_mfma6(const u32x4 (&a)[3], const uint4 (&b)[3], f32x16 acc) {
  const uint4 a1 = make_uint4(a[0][0], a[0][1], a[0][2], a[0][3]), a2 = make_uint4(a[1][0], a[1][1], a[1][2], a[1][3]),
              a3 = make_uint4(a[2][0], a[2][1], a[2][2], a[2][3]);
  acc = mfma_bf16(a1, b[2], acc);
  acc = mfma_bf16(a3, b[0], acc);
  acc = mfma_bf16(a2, b[1], acc);
  acc = mfma_bf16(a1, b[1], acc);
  acc = mfma_bf16(a2, b[0], acc);
  acc = mfma_bf16(a1, b[0], acc);
  return acc;
}
__device__ __forceinline__ f32x16 g4_mfma6(const u32x4 (&a)[2], const uint4 (&b)[2], f32x16 acc) {   // a2 b1, a1 b2, a1 b1 (gconv_common.h)
  const uint4 a1 = make_uint4(a[0][0], a[0][1], a[0][2], a[0][3]), a2 = make_uint4(a[1][0], a[1][1], a[1][2], a[1][3]);
  acc = mfma_f16(a2, b[0], acc);
  acc = mfma_f16(a1, b[1], acc);
  acc = mfma_f16(a1, b[0], acc);
  return acc;
}
__device__ __forceinline__ f32x16 g4_mfma6(const u32x4 (&a)[1], const uint4 (&b)[1], f32x16 acc) {
  return mfma_bf16(make_uint4(a[0][0], a[0][1], a[0][2], a[0][3]), b[0], acc);
}
__device__ __forceinline__ uint32_t g4_pack_bf16(const float a, const float b) {   // round to nearest even (v_cvt_pk_bf16_f32)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  union { bf16x2 h; uint32_t u; } c;
  c.h = __builtin_convertvector(v, bf16x2);
  return c.u;
}
__device__ __forceinline__ void g4_split(const float (&x)[8], uint4 (&b)[3]) { split8(x, b[0], b[1], b[2]); }
__device__ __forceinline__ void g4_split(const float (&x)[8], uint4 (&b)[2]) { split8h(x, pow2i(PDSE_F16_ACT_EXP), b[0], b[1]); }
__device__ __forceinline__ void g4_split(const float (&x)[8], uint4 (&b)[1]) {
  b[0] = make_uint4(g4_pack_bf16(x[0], x[1]), g4_pack_bf16(x[2], x[3]), g4_pack_bf16(x[4], x[5]), g4_pack_bf16(x[6], x[7]));
}

struct g4_chunk {
  int s, tap, cb0, n, kb0;   // source, tap, first 16-channel block, blocks in the chunk, first K block in the packed weights
};

__device__ long long* g_trace4 = nullptr;   // PDSE_G4_TRACE=1 (diagnostic): [workgroup][wave][8] clock sums

// BLK: every source is channel-blocked (pdse_src.blk = 8) / none is
template <int EPI, int MT, bool BLK, int NP = 3>
__global__ __launch_bounds__(512, 2) void gconv4_kernel(const pdse_gconv_desc d) {
  constexpr int FB = 64 * NP;   // uint4 entries per fragment (NP planes of 64 lanes)
#ifdef PDSE_DIAG
  long long* const trace = g_trace4;
#else
  long long* const trace = nullptr;
#endif
  const long long c_start = trace ? clock64() : 0;
  long long c_req = 0, c_cmp = 0, c_bar = 0;
  constexpr bool DUAL = (EPI != PDSE_EPI_LINEAR);
  constexpr int NBR = DUAL ? 2 : 1, FR = NBR * MT;   // fragments (3 planes each) per K block and workgroup
  extern __shared__ uint4 ring[];                     // [2 buffers][G4_CH blocks][FR][FB]
  const unsigned ring_base = (unsigned)(uintptr_t)ring;   // LDS byte address (the low half of the flat address)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int P = d.Tout * d.Fout;
  const int p = (blockIdx.x * 8 + wave) * 32 + col;
  const bool pvalid = p < P;
  const int t = pvalid ? p / d.Fout : 0;
  const int j = pvalid ? p - t * d.Fout : 0;
  const int mtiles = (d.Cout + 31) >> 5;
  const int mt0 = blockIdx.z * MT;

  const int cbn0 = d.in0.C >> 4, cbn1 = d.in1.C >> 4;                    // 16-channel blocks per source
  const int cps0 = (cbn0 + G4_CH - 1) / G4_CH, cps1 = (cbn1 + G4_CH - 1) / G4_CH;   // chunks per (source, tap)
  const int nch0 = d.ntaps * cps0, nch = nch0 + d.ntaps * cps1;

  // Everything about a chunk is wave-uniform; readfirstlane says so to the compiler (without it the source select and the
  // channel bases were treated as per-lane values and every gather load sat in a waterfall loop)
  auto uni = [](const int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto decode = [&](const int ci) {
    g4_chunk c;
    c.s = uni(ci >= nch0);
    const int q = c.s ? ci - nch0 : ci, cps = c.s ? cps1 : cps0, cbn = c.s ? cbn1 : cbn0;
    c.tap = uni(q / cps);
    c.cb0 = uni((q - c.tap * cps) * G4_CH);
    c.n = uni(min(G4_CH, cbn - c.cb0));
    c.kb0 = uni((c.s ? d.ntaps * cbn0 : 0) + c.tap * cbn + c.cb0);
    return c;
  };
  // A fragments of a chunk -> ring buffer `buf`: one 1 KB piece (= one plane of one fragment) per wave instruction
  auto dma = [&](const g4_chunk& c, const int buf) {
    const int npieces = c.n * FR * NP;
    for (int pc = __builtin_amdgcn_readfirstlane(wave); pc < npieces; pc += 8) {
      const int i = pc / (FR * NP), rem = pc - i * (FR * NP), f = rem / NP, part = rem - f * NP;
      const int br = f / MT, m = f - br * MT;
      if (mt0 + m < mtiles) {   // tiles past Cout keep stale fragments: their accumulators are never stored
        const uint4* src = reinterpret_cast<const uint4*>(br ? d.w1 : d.w0) + ((size_t)(c.kb0 + i) * mtiles + mt0 + m) * FB +
                           part * 64 + lane;
        glds16_g4(src, ring + ((buf * G4_CH + i) * FR + f) * FB + part * 64);
      }
    }
  };
  // B operand of a chunk: raw[i][e] = channel 16 (cb0 + i) + 8h + e of this lane's position at the chunk's tap
  auto gather = [&](const g4_chunk& c, g4_raw<BLK>& raw, bool& inb) {
    const bool s1 = c.s != 0;   // field-by-field scalar selects (a reference to one of two kernel-argument structs is not)
    const float* const sptr = s1 ? d.in1.ptr : d.in0.ptr;
    const int64_t ssb = s1 ? d.in1.sb : d.in0.sb, ssc = s1 ? d.in1.sc : d.in0.sc, sst = s1 ? d.in1.st : d.in0.st,
                  ssf = s1 ? d.in1.sf : d.in0.sf;
    int dt = 0, df = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {   // kernel-argument table, scalar compares instead of a dynamically indexed load
      dt = c.tap == k ? d.tap_dt[k] : dt;
      df = c.tap == k ? d.tap_df[k] : df;
    }
    const int tin = t + dt, fin = j * d.sf_in + df;
    inb = pvalid && fin >= 0 && fin < d.Fin && tin >= 0 && tin < d.Tin;
    if constexpr (BLK) {   // [B][C/8][T][F][8]: block 2 (cb0 + i) + h, two 16-byte loads per K block
      const unsigned off = 4u * (unsigned)((inb ? (int64_t)b * ssb + (int64_t)tin * sst + (int64_t)fin * ssf : 0) + (int64_t)h * ssc);
#pragma unroll
      for (int i = 0; i < G4_CH; ++i) {
        if (i < c.n) {
          const float* base = sptr + (int64_t)(2 * (c.cb0 + i)) * ssc;
          raw.v[i][0] = g4_load16(base, off);
          raw.v[i][1] = g4_load16(base + 4, off);
        }
      }
    } else {
    const unsigned off = 4u * (unsigned)((inb ? (int64_t)b * ssb + (int64_t)tin * sst + (int64_t)fin * ssf : 0) + (int64_t)(8 * h) * ssc);
#pragma unroll
    for (int i = 0; i < G4_CH; ++i) {
      if (i < c.n) {   // wave-uniform
        const float* base = sptr + (int64_t)(16 * (c.cb0 + i)) * ssc;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw.v[i][e] = g4_load(base + (int64_t)e * ssc, off);
      }
    }
    }
  };

  f32x16 acc0[MT], acc1[DUAL ? MT : 1];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[m][r] = 0.f;
      if (DUAL) acc1[m][r] = 0.f;
    }
  }
  // One chunk.  The ring reads of K block i+1 are issued fragment by fragment right behind the six MFMAs that were the last
  // readers of that fragment's registers in block i, so they fly under the remaining MFMAs of block i instead of in front of
  // block i+1's (an MFMA reads its A / B operands when it issues; the LDS data returns ~100 cycles later).  Before round 3's
  // end every block waited for its twelve ring reads with an idle matrix pipe: 5.3k cycles per chunk for 3.1k of MFMAs.
  auto compute = [&](const g4_chunk& c, const g4_raw<BLK>& raw, const bool inb, const int buf) {
    const bool elu = uni((c.s ? d.in1.act : d.in0.act) == PDSE_ACT_ELU) != 0;   // GCRN re-applies ELU to the skip half (gcrn.py:152-155)
    u32x4 af[FR][NP];
    const unsigned wa0 = ring_base + (unsigned)((((buf * G4_CH) * FR) * FB + lane) * 16);
#pragma unroll
    for (int f = 0; f < FR; ++f) g4_lds_read3(wa0 + f * (FB * 16), af[f]);
#pragma unroll
    for (int i = 0; i < G4_CH; ++i) {
      if (i < c.n) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = inb ? raw.get(i, e) : 0.f;
        if (elu) {   // wave-uniform; exp2-based like the epilogues' ELU (act_c); elu(0) = 0 keeps masked lanes zero
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = x[e] < 0.f ? fast_exp(x[e]) - 1.0f : x[e];
        }
        uint4 bp[NP];
        g4_split(x, bp);         // runs under the LDS latency of the block's fragments
        g4_lds_wait(af);
        const bool more = i + 1 < c.n;   // wave-uniform
        const unsigned wn = ring_base + (unsigned)((((buf * G4_CH + i + 1) * FR) * FB + lane) * 16);
#pragma unroll
        for (int f = 0; f < FR; ++f) {
          const int m = f % MT;
          if (f < MT) acc0[m] = g4_mfma6(af[f], bp, acc0[m]);
          else acc1[m] = g4_mfma6(af[f], bp, acc1[m]);
          // (i + 1 < G4_CH is a compile-time fact once the loop is unrolled: no read is emitted behind the last possible block,
          // whose destination registers would be dead - free for the compiler to re-use - while the read is still in flight)
          if (i + 1 < G4_CH && more) g4_lds_read3(wn + f * (FB * 16), af[f]);
        }
      }
    }
  };

  g4_raw<BLK> rawA, rawB;
  bool inbA = false, inbB = false;
  g4_chunk cA = decode(0), cB = cA;
  gather(cA, rawA, inbA);
  dma(cA, 0);
  __syncthreads();   // DMA and gather of chunk 0 have landed (the fence waits vmcnt(0) while LDS-DMA is in flight)
  g4_landed(rawA);
  const long long c_pro = trace ? clock64() : 0;
  for (int ci = 0; ci < nch; ci += 2) {
    const bool hasB = ci + 1 < nch;
    const long long k0 = trace ? clock64() : 0;
    if (hasB) {      // chunk ci+1: requests and DMA go out before chunk ci's matrix work
      cB = decode(ci + 1);
      gather(cB, rawB, inbB);
      dma(cB, 1);
    }
    const long long k1 = trace ? clock64() : 0;
    compute(cA, rawA, inbA, 0);
    const long long k2 = trace ? clock64() : 0;
    __syncthreads();   // buffer 1 complete; every wave is done reading buffer 0
    const long long k3 = trace ? clock64() : 0;
    c_req += k1 - k0, c_cmp += k2 - k1, c_bar += k3 - k2;
    if (!hasB) break;
    g4_landed(rawB);
    const bool hasA = ci + 2 < nch;
    if (hasA) {
      cA = decode(ci + 2);
      gather(cA, rawA, inbA);
      dma(cA, 0);
    }
    const long long k4 = trace ? clock64() : 0;
    compute(cB, rawB, inbB, 1);
    const long long k5 = trace ? clock64() : 0;
    __syncthreads();
    const long long k6 = trace ? clock64() : 0;
    c_req += k4 - k3, c_cmp += k5 - k4, c_bar += k6 - k5;
    g4_landed(rawA);
  }
  const long long c_loop = trace ? clock64() : 0;
  if constexpr (NP == 2) {   // f16x2: the accumulators hold (true value) * 2^(activation exponent + weight exponent)
    const float us = pow2i(-(PDSE_F16_ACT_EXP + d.wexp));
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[m][r] *= us;
        if (DUAL) acc1[m][r] *= us;
      }
  }
  gconv_epilogue<EPI, MT>(d, tail_from_desc(d), acc0, acc1, b, t, j, pvalid, lane, h, mt0, mtiles);
  if (trace && lane == 0) {
    long long* q = trace + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 8;
    q[0] = c_pro - c_start;
    q[1] = c_req;
    q[2] = c_cmp;
    q[3] = c_bar;
    q[4] = nch;
    q[5] = clock64() - c_loop;
    q[6] = clock64() - c_start;
  }
}

template <int EPI, int MT, bool BLK, int NP>
static int launch4b(const pdse_gconv_desc* d, hipStream_t s, const int mtiles) {
  const int P = d->Tout * d->Fout;
  const dim3 grid(((P + 31) / 32 + 7) / 8, d->B, (mtiles + MT - 1) / MT), block(512);
  constexpr int FR = (EPI == PDSE_EPI_LINEAR ? 1 : 2) * MT;
  const size_t lds = (size_t)2 * G4_CH * FR * 64 * NP * sizeof(uint4);
  static unsigned long long attr_mask = 0;   // per instantiation and device
  if (lds > 64 * 1024 && pdse_lds_attr((const void*)gconv4_kernel<EPI, MT, BLK, NP>, &attr_mask, "gconv4 lds attribute")) return 1;
  static const bool tracing = PDSE_DIAG_ENV("PDSE_G4_TRACE") != nullptr;
  static long long* tbuf = nullptr;
  const size_t nw = (size_t)grid.x * grid.y * grid.z * 8;
  if (tracing) {
    if (!tbuf) {
      (void)hipMalloc(&tbuf, (size_t)1 << 24);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace4), &tbuf, sizeof(tbuf));
    }
    (void)hipMemsetAsync(tbuf, 0, nw * 64, s);
  }
  hipLaunchKernelGGL((gconv4_kernel<EPI, MT, BLK, NP>), grid, block, lds, s, *d);
  if (tracing && nw * 64 <= ((size_t)1 << 24)) {   // diagnostic: per-wave averages in shader clocks
    (void)hipStreamSynchronize(s);
    long long* h = (long long*)malloc(nw * 64);
    (void)hipMemcpy(h, tbuf, nw * 64, hipMemcpyDeviceToHost);
    double sum[7] = {0};
    for (size_t i = 0; i < nw; ++i)
      for (int k = 0; k < 7; ++k) sum[k] += (double)h[i * 8 + k];
    fprintf(stderr, "gconv4 trace EPI %d MT %d taps %d cin %d+%d cout %d %dx%d grid %ux%ux%u: prologue %.0f | per chunk: request %.0f compute %.0f barrier %.0f | chunks %.1f | epilogue %.0f total %.0f\n",
            EPI, MT, d->ntaps, d->in0.C, d->in1.C, d->Cout, d->Tout, d->Fout, grid.x, grid.y, grid.z, sum[0] / nw, sum[1] / sum[4], sum[2] / sum[4], sum[3] / sum[4],
            sum[4] / nw, sum[5] / nw, sum[6] / nw);
    free(h);
  }
  return pdse_check_launch("gconv4");
}

template <int EPI, int MT>
static int launch4(const pdse_gconv_desc* d, hipStream_t s, const int mtiles) {
  if (d->korder == 5)   // two planes: fp16 hi + lo of the power-of-two scaled operands, three f16 products (f16x2)
    return d->in0.blk ? launch4b<EPI, MT, true, 2>(d, s, mtiles) : launch4b<EPI, MT, false, 2>(d, s, mtiles);
  if (d->korder == 4)   // one plane: plain bf16 operands, one product (the opt-in bf16 mode)
    return d->in0.blk ? launch4b<EPI, MT, true, 1>(d, s, mtiles) : launch4b<EPI, MT, false, 1>(d, s, mtiles);
  return d->in0.blk ? launch4b<EPI, MT, true, 3>(d, s, mtiles) : launch4b<EPI, MT, false, 3>(d, s, mtiles);
}

// korder 3: LINEAR / GLU, one or two sources of a multiple of 16 channels, no load transform; validated by pdse_gconv_launch
int pdse_gconv4_launch(const pdse_gconv_desc* d, hipStream_t s) {
  const bool two = d->in1.C > 0;
  // lane offsets are 32-bit BYTE offsets from a scalar channel base
  const long long span0 = 4 * ((long long)d->B * d->in0.sb + 40ll * d->in0.sc), span1 = two ? 4 * ((long long)d->B * d->in1.sb + 40ll * d->in1.sc) : 0;
  auto blk_ok = [](const pdse_src& x) {   // 16-byte loads: every stride a multiple of 4 floats, the base 16-byte aligned
    return x.blk == 0 || (x.blk == 8 && !((x.sb | x.sc | x.st | x.sf) & 3) && (reinterpret_cast<uintptr_t>(x.ptr) & 15) == 0);
  };
  if (!blk_ok(d->in0) || (two && !blk_ok(d->in1)) || (two && d->in0.blk != d->in1.blk)) {
    pdse_set_error("gconv4: a channel-blocked source has blk = 8, strides in multiples of 4 floats and a 16-byte aligned base; "
                   "two sources are both blocked or both plain");
    return 1;
  }
  if (!(d->epi == PDSE_EPI_LINEAR || d->epi == PDSE_EPI_GLU) || (d->in0.C & 15) || (d->in1.C & 15) || d->in0.C == 0 || d->cin1 ||
      d->xf_mode != 0 || d->padrow != nullptr || d->ntaps > 12 || span0 >= (1ll << 32) || span1 >= (1ll << 32) ||
      (d->epi == PDSE_EPI_GLU && !d->w1) || (d->korder == 5 && (d->wexp < -40 || d->wexp > 40))) {
    pdse_set_error("gconv4: split-bf16 GEMM convolutions need LINEAR / GLU, channel counts in multiples of 16, <= 12 taps, "
                   "no load transform / pad row, 32-bit gather offsets");
    return 1;
  }
  const int mtiles = (d->Cout + 31) / 32;
  if (d->epi == PDSE_EPI_GLU) return mtiles >= 2 ? launch4<PDSE_EPI_GLU, 2>(d, s, mtiles) : launch4<PDSE_EPI_GLU, 1>(d, s, mtiles);
  if (mtiles >= 4) return launch4<PDSE_EPI_LINEAR, 4>(d, s, mtiles);
  if (mtiles >= 2) return launch4<PDSE_EPI_LINEAR, 2>(d, s, mtiles);
  return launch4<PDSE_EPI_LINEAR, 1>(d, s, mtiles);
}
