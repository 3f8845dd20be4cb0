// aia.hip — the non-convolutional operators of the DB-AIAT prior (model/dbaiat.py):
// LayerNorm over bins + per-channel PReLU, LayerNorm over channels, the attention core,
// a persistent per-line bidirectional GRU, GroupNorm + layer update, AHAM merge.
// All tensors are channel-major [B,C,T,F] fp32; the convolutions and Linear layers of the
// model run on the gather-GEMM kernels (gconv.hip / gconv2.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"
#include "pdse_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REQ(cond, msg)     \
  do {                     \
    if (!(cond)) {         \
      pdse_set_error(msg); \
      return 1;            \
    }                      \
  } while (0)

__device__ __forceinline__ float aia_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float aia_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + aia_exp(-x)); }

// ---------------------------------------------------------------------------------------
// LayerNorm over the F bins of a (b,c,t) row, then PReLU with the channel's slope
// (dbaiat.py:498 inp_norm/inp_prelu, :627-628 DenseBlock, :500 enc_norm1, :545 dec_norm1).
// One wavefront per row (F <= 192: three values per lane), HBM-bound.
// ---------------------------------------------------------------------------------------
// RL_R rows per wavefront with all their loads issued first: one row per wave left 644 bytes in flight per wave
// (3.4 TB/s over the 1.06 GB of a [32,64,401,161] launch).
#define RL_R 4
__global__ __launch_bounds__(256) void rowln_kernel(const pdse_rowln_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RL_R;
  const int64_t rows = (int64_t)d.B * d.C * d.T;
  if (row0 >= rows) return;
  float v[RL_R][3];
#pragma unroll
  for (int r = 0; r < RL_R; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
    const float* x = d.in + row * d.F;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = k * 64 + lane;
      v[r][k] = i < d.F ? x[i] : 0.f;
    }
  }
  float g3[3], b3[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = k * 64 + lane;
    g3[k] = i < d.F ? d.gamma[i] : 0.f;
    b3[k] = i < d.F ? d.beta[i] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < RL_R; ++r) {
    const int64_t row = row0 + r;
    if (row >= rows) break;          // wave-uniform
    const int64_t b = row / ((int64_t)d.C * d.T), ct = row - b * (int64_t)d.C * d.T;
    const int c = (int)(ct / d.T);
    float* o = d.out + b * d.out_sb + ct * d.F;
    float sum = (v[r][0] + v[r][1]) + v[r][2];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)d.F;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float e = (k * 64 + lane) < d.F ? v[r][k] - mean : 0.f;
      sq += e * e;
    }
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)d.F + d.eps);
    const float slope = d.slope[c];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = k * 64 + lane;
      if (i < d.F) {
        const float y = (v[r][k] - mean) * rstd * g3[k] + b3[k];
        o[i] = y > 0.f ? y : slope * y;
      }
    }
  }
}

int pdse_rowln_launch(const pdse_rowln_desc* d, hipStream_t s) {
  REQ(d && d->in && d->gamma && d->beta && d->slope && d->out, "rowln: null pointer");
  REQ(d->B > 0 && d->C > 0 && d->T > 0 && d->F > 0 && d->F <= 192, "rowln: bad sizes (F <= 192)");
  const int64_t rows = (int64_t)d->B * d->C * d->T;
  const int64_t blocks = (rows + 4 * RL_R - 1) / (4 * RL_R);
  REQ(blocks < (1ll << 31), "rowln: too many rows");
  hipLaunchKernelGGL(rowln_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *d);
  return pdse_check_launch("rowln");
}

// ---------------------------------------------------------------------------------------
// LayerNorm over the C (<= 64) channels of every position (the d_model axis of the
// transformer layers, dbaiat.py:76,81,87).  One thread per position, coalesced across threads.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chln_kernel(const pdse_chln_desc d) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (q >= d.plane) return;
  const float* x = d.in + (int64_t)b * d.C * d.plane + q;
  float* o = d.out + (int64_t)b * d.C * d.plane + q;
  float v[64];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    v[c] = c < d.C ? x[(int64_t)c * d.plane] : 0.f;
    sum += v[c];
  }
  const float mean = sum / (float)d.C;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    const float e = c < d.C ? v[c] - mean : 0.f;
    sq += e * e;
  }
  const float rstd = 1.0f / sqrtf(sq / (float)d.C + d.eps);
#pragma unroll
  for (int c = 0; c < 64; ++c)
    if (c < d.C) o[(int64_t)c * d.plane] = (v[c] - mean) * rstd * d.gamma[c] + d.beta[c];
}

int pdse_chln_launch(const pdse_chln_desc* d, hipStream_t s) {
  REQ(d && d->in && d->gamma && d->beta && d->out, "chln: null pointer");
  REQ(d->B > 0 && d->B <= 65535 && d->C > 0 && d->C <= 64 && d->plane > 0, "chln: bad sizes (C <= 64)");
  hipLaunchKernelGGL(chln_kernel, dim3((unsigned)((d->plane + 255) / 256), d->B), dim3(256), 0, s, *d);
  return pdse_check_launch("chln");
}

// ---------------------------------------------------------------------------------------
// Attention core: one workgroup per (line, b); K and V of the line live in LDS (16-byte aligned
// rows), each thread owns (query position, head) pairs and runs an online softmax over the keys.
// E = 32, 4 heads of 8 (dbaiat.py:123-126); sequence 80 (bins) or T (frames).
// ---------------------------------------------------------------------------------------
// One workgroup per (line, batch item, HEAD) since round 3: K and V of one head are 2 x S x HD floats (25.7 KB at S = 401,
// HD = 8), so four to five workgroups share a CU and every thread owns ONE query; keys are taken four at a time so the
// running maximum is rescaled once per block instead of once per key.  (Until then a workgroup of 1024 threads held all four
// heads of a line - 100 KB of LDS, one workgroup per CU - and walked its 4 S (query, head) items in rounds of 1024: the
// second round of a 401-position line ran 56 % full.  Measured at B=32: S = 401 1.45 -> see profiles/r03_prior_per_launch.txt.)
#define ATTN_THREADS 1024
// HD = head dimension (8: d_model 32, 16: d_model 64; always 4 heads).
// Sequences longer than one LDS image (S > 2048: never on this path) would be walked in key chunks of CH positions:
// K and V of a chunk are staged, every thread advances the online softmax of its query over the chunk
// and keeps (m, l, acc) in registers across chunks.  Chunks are multiples of four keys, so the rescale grouping - and
// with it every rounding - is the one a single image would give.
template <int HD>
__global__ __launch_bounds__(ATTN_THREADS) void attn_kernel(const pdse_attn_desc d, const int CH) {
  extern __shared__ float kv[];  // [CH][HD] keys, then [CH][HD] values
  constexpr int W = HD;
  const int E = d.E;
  const int b = blockIdx.y, line = blockIdx.x, c0 = blockIdx.z * W;   // blockIdx.z: head
  const int S = d.axis == 0 ? d.F : d.T;
  const int NT = blockDim.x;
  const int64_t plane = (int64_t)d.T * d.F;
  const int64_t base = (int64_t)b * 3 * E * plane + (d.axis == 0 ? (int64_t)line * d.F : (int64_t)line);
  const int64_t ss = d.axis == 0 ? 1 : d.F;
  float* ks = kv;
  float* vs = kv + (size_t)CH * W;
  const bool single = CH >= S;
  auto stage = [&](const int k0, const int n) {
    for (int i = threadIdx.x; i < n * W; i += NT) {
      const int e = i / n, sp = i - e * n;  // consecutive threads -> consecutive sequence positions
      ks[sp * W + e] = d.qkv[base + (int64_t)(E + c0 + e) * plane + (int64_t)(k0 + sp) * ss];
      vs[sp * W + e] = d.qkv[base + (int64_t)(2 * E + c0 + e) * plane + (int64_t)(k0 + sp) * ss];
    }
  };
  if (single) {
    stage(0, S);
    __syncthreads();
  }
  const int64_t obase = (int64_t)b * E * plane + (d.axis == 0 ? (int64_t)line * d.F : (int64_t)line);
  const int rounds = (S + NT - 1) / NT;   // uniform trip count: the chunk loop holds barriers
  for (int round = 0; round < rounds; ++round) {
    const int idx = round * NT + threadIdx.x;
    const bool valid = idx < S;
    const int sq = valid ? idx : 0;
    float q[HD], acc[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) {
      q[e] = valid ? d.qkv[base + (int64_t)(c0 + e) * plane + (int64_t)sq * ss] : 0.f;
      acc[e] = 0.f;
    }
    float m = -1e30f, l = 0.f;
    const float* kh = ks;
    const float* vh = vs;
    for (int k0 = 0; k0 < S; k0 += CH) {
      const int n = min(CH, S - k0);
      if (!single) {
        __syncthreads();          // every reader of the previous chunk (or round) is done with the image
        stage(k0, n);
        __syncthreads();
      }
      if (!valid) continue;
      int sp = 0;
      for (; sp + 4 <= n; sp += 4) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float a = 0.f;
#pragma unroll
          for (int e4 = 0; e4 < HD; e4 += 4) {
            const float4 kq = *reinterpret_cast<const float4*>(kh + (sp + u) * W + e4);
            a += q[e4] * kq.x + q[e4 + 1] * kq.y + q[e4 + 2] * kq.z + q[e4 + 3] * kq.w;
          }
          sc[u] = a;
        }
        const float mn = fmaxf(fmaxf(m, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
        const float corr = aia_exp(m - mn);
        l *= corr;
#pragma unroll
        for (int e = 0; e < HD; ++e) acc[e] *= corr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float pw = aia_exp(sc[u] - mn);
          l += pw;
#pragma unroll
          for (int e4 = 0; e4 < HD; e4 += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(vh + (sp + u) * W + e4);
            acc[e4] += pw * v0.x; acc[e4 + 1] += pw * v0.y; acc[e4 + 2] += pw * v0.z; acc[e4 + 3] += pw * v0.w;
          }
        }
        m = mn;
      }
      for (; sp < n; ++sp) {
        const float* kr = kh + sp * W;
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) s1 += q[e] * kr[e];
        const float mn = fmaxf(m, s1);
        const float corr = aia_exp(m - mn), pw = aia_exp(s1 - mn);
        const float* vr = vh + sp * W;
        l = l * corr + pw;
#pragma unroll
        for (int e = 0; e < HD; ++e) acc[e] = acc[e] * corr + pw * vr[e];
        m = mn;
      }
    }
    if (valid) {
      const float inv = 1.0f / l;
#pragma unroll
      for (int e = 0; e < HD; ++e) d.out[obase + (int64_t)(c0 + e) * plane + (int64_t)sq * ss] = acc[e] * inv;
    }
  }
}

int pdse_attn_launch(const pdse_attn_desc* d, hipStream_t s) {
  REQ(d && d->qkv && d->out, "attention: null pointer");
  REQ(d->B > 0 && d->B <= 65535 && d->T > 0 && d->F > 0 && d->heads == 4 && (d->E == 32 || d->E == 64),
      "attention: 4 heads, d_model 32 or 64 (dbaiat.py:123-126, :186-189)");
  REQ(d->axis == 0 || d->axis == 1, "attention: axis 0 (bins) or 1 (frames)");
  const int S = d->axis == 0 ? d->F : d->T, lines = d->axis == 0 ? d->T : d->F;
  const int HD = d->E / 4;
  // one LDS image of the head's K and V when the sequence fits, else key chunks (a multiple of four keys).  The chunk follows from
  // the LDS budget: 2 CH HD floats <= 64 KB (two workgroups per CU at the least) - 2048 keys at head dimension 8 (d_model 32),
  // 1024 at 16 (d_model 64: 2048 keys would ask for 256 KB, more than a CU has)
  const int cap = (int)((64 * 1024) / (2 * HD * sizeof(float))) & ~3;
  const int CH = S <= cap ? S : cap;
  const size_t lds = (size_t)2 * CH * HD * sizeof(float);
  REQ(lds <= 160 * 1024, "attention: LDS image of K and V exceeds the CU");
  const void* fn = d->E == 32 ? (const void*)attn_kernel<8> : (const void*)attn_kernel<16>;
  if (lds > 64 * 1024)
    if (pdse_check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "attention lds attribute"))
      return 1;
  const int nt = S >= ATTN_THREADS ? ATTN_THREADS : (S + 63) / 64 * 64;   // one query per thread, whole waves
  const dim3 grid(lines, d->B, 4);
  if (d->E == 32)
    hipLaunchKernelGGL(attn_kernel<8>, grid, dim3(nt), lds, s, *d, CH);
  else
    hipLaunchKernelGGL(attn_kernel<16>, grid, dim3(nt), lds, s, *d, CH);
  return pdse_check_launch("attention");
}

// ---------------------------------------------------------------------------------------
// Bidirectional GRU, hidden H = 64 (d_model 32, dbaiat.py:45,83) or 128 (d_model 64, :186-189), persistent over
// the sequence: a workgroup owns 32 lines of one direction for all S steps, so the recurrence needs no
// inter-workgroup synchronisation.  Waves 0..3H/32-1 hold one 32-row tile of W_hh each in registers (A operand)
// and multiply it with the state h [H x 32 lines] kept in LDS (B operand, conflict-free rows); all 8H threads
// then apply the gates (r,z,n order, n = tanh(x_n + r * (W_hn h + b_hn))).
// ---------------------------------------------------------------------------------------
// FUSED: the input projection W_ih x_t runs inside the step (16 more MFMAs per wave; x_t is requested at the top of
// the step and consumed after the 32 W_hh MFMAs that hide its latency).  r and z rows accumulate both products in
// one tile, the n rows keep W_in x + b_in apart from W_hn h + b_hn (n = tanh(x_n + r * h_n)).  Measured at B=32,
// T=401: 1.55 ms per layer against 1.13 + 0.76 ms for recurrence + separate projection launch.  (Parking x_{t+1} in
// LDS through the two waves without a W_hh tile saved the 32 extra registers but put the 16 W_ih MFMAs in front of
// the recurrent ones: 1.81 ms.)
template <int H, bool FUSED>
__global__ __launch_bounds__(8 * H) void gru_kernel(const pdse_gru_desc d) {
  constexpr int G3 = 3 * H, NT = 8 * H, KS = H / 2, MW = G3 / 32, KI = H / 4;
  extern __shared__ float gru_lds[];
  float (*hs)[32] = reinterpret_cast<float (*)[32]>(gru_lds);             // [H][32]
  float (*gh)[33] = reinterpret_cast<float (*)[33]>(gru_lds + H * 32);     // [3H][33]
  float (*gxn)[33] = reinterpret_cast<float (*)[33]>(gru_lds + H * 32 + G3 * 33);   // FUSED: [H][33] W_in x + b_in
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int dir = blockIdx.y;
  const int S = d.axis == 0 ? d.F : d.T;
  const int per_b = d.axis == 0 ? d.T : d.F;
  const int nlines = d.B * per_b;
  const int64_t plane = (int64_t)d.T * d.F;
  const int64_t ss = d.axis == 0 ? 1 : d.F;

  float a[KS];
  float ai[FUSED ? KI : 1];
  if (wave < MW) {
    const float* A = d.whh + ((size_t)(dir * MW + wave) * KS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = A[(size_t)ks * 64];
    if constexpr (FUSED) {
      const float* AI = d.wih + ((size_t)(dir * MW + wave) * KI) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KI; ++ks) ai[ks] = AI[(size_t)ks * 64];
    }
  }
  for (int i = threadIdx.x; i < H * 32; i += NT) (&hs[0][0])[i] = 0.f;

  // gate work of this thread: 4 (unit, line) items, lines fastest
  int64_t gbase[4], ybase[4];
  bool live[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int it = threadIdx.x + NT * k;
    const int unit = it >> 5, ln = it & 31;
    const int Lg = blockIdx.x * 32 + ln;
    live[k] = Lg < nlines;
    const int b = live[k] ? Lg / per_b : 0, w = live[k] ? Lg - b * per_b : 0;
    const int64_t pos = d.axis == 0 ? (int64_t)w * d.F : (int64_t)w;
    gbase[k] = (int64_t)b * 6 * H * plane + pos + (int64_t)(dir * G3 + unit) * plane;
    ybase[k] = (int64_t)b * 2 * H * plane + pos + (int64_t)(dir * H + unit) * plane;
  }
  // FUSED: this lane's line as a B-operand column: input channel 2*ks + hh of line `col`
  int64_t xbase = 0;
  bool xlive = false;
  if constexpr (FUSED) {
    const int Lg = blockIdx.x * 32 + col;
    xlive = Lg < nlines;
    const int b = xlive ? Lg / per_b : 0, w = xlive ? Lg - b * per_b : 0;
    xbase = (int64_t)b * (H / 2) * plane + (d.axis == 0 ? (int64_t)w * d.F : (int64_t)w) + (int64_t)hh * plane;
  }
  const float* bh = d.bhh + dir * G3;
  const float* bi = FUSED ? d.bih + dir * G3 : nullptr;
  __syncthreads();

  for (int step = 0; step < S; ++step) {
    const int sq = dir ? S - 1 - step : step;
    float xr[4], xz[4], xn[4];
    float xin[FUSED ? KI : 1];
    if constexpr (FUSED) {
      if (wave < MW) {
#pragma unroll
        for (int ks = 0; ks < KI; ++ks) xin[ks] = xlive ? d.x[xbase + (int64_t)(2 * ks) * plane + (int64_t)sq * ss] : 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t o = gbase[k] + (int64_t)sq * ss;
        xr[k] = live[k] ? d.gx[o] : 0.f;
        xz[k] = live[k] ? d.gx[o + (int64_t)H * plane] : 0.f;
        xn[k] = live[k] ? d.gx[o + (int64_t)2 * H * plane] : 0.f;
      }
    }
    if (wave < MW) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], hs[2 * ks + hh][col], acc, 0, 0, 0);
      if constexpr (FUSED) {
        const bool ngate = wave >= 2 * (H / 32);          // tiles of the n rows
        if (ngate) {
          f32x16 ax;
#pragma unroll
          for (int r = 0; r < 16; ++r) ax[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KI; ++ks) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[ks], xin[ks], ax, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh;
            gxn[row - 2 * H][col] = ax[r] + bi[row];
            gh[row][col] = acc[r] + bh[row];
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < KI; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[ks], xin[ks], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh;
            gh[row][col] = acc[r] + (bh[row] + bi[row]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh;
          gh[row][col] = acc[r] + bh[row];
        }
      }
    }
    __syncthreads();
    float hn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int it = threadIdx.x + NT * k;
      const int u = it >> 5, l = it & 31;
      const float r = aia_sigmoid((FUSED ? 0.f : xr[k]) + gh[u][l]);
      const float z = aia_sigmoid((FUSED ? 0.f : xz[k]) + gh[H + u][l]);
      const float n = tanhf((FUSED ? gxn[u][l] : xn[k]) + r * gh[2 * H + u][l]);
      hn[k] = (1.f - z) * n + z * hs[u][l];
    }
    __syncthreads();   // every MFMA wave has consumed hs, every gate thread has read gh
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int it = threadIdx.x + NT * k;
      hs[it >> 5][it & 31] = hn[k];
      if (live[k]) d.y[ybase[k] + (int64_t)sq * ss] = hn[k];
    }
    __syncthreads();
  }
}

int pdse_gru_launch(const pdse_gru_desc* d, hipStream_t s) {
  REQ(d && d->whh && d->bhh && d->y, "bigru: null pointer");
  REQ(d->B > 0 && d->T > 0 && d->F > 0 && (d->H == 64 || d->H == 128), "bigru: hidden size 64 or 128 (dbaiat.py:45)");
  REQ(d->axis == 0 || d->axis == 1, "bigru: axis 0 (bins) or 1 (frames)");
  const int nlines = d->B * (d->axis == 0 ? d->T : d->F);
  const bool fused = d->x != nullptr;
  REQ(!fused || (d->H == 64 && d->wih && d->bih), "bigru: the fused input projection needs H == 64, wih and bih");
  REQ(d->split == 0 || d->split == 1, "bigru: split is 0 or 1");
  if (d->split) return pdse_gru3_launch(d, s);   // split-bf16 operands (csrc/gru3.hip)
  REQ(fused || d->gx, "bigru: gx missing");
  const size_t lds = (size_t)(d->H * 32 + 3 * d->H * 33 + (fused ? d->H * 33 : 0)) * sizeof(float);
  const dim3 grid((nlines + 31) / 32, 2);
  if (d->H == 64) {
    if (fused) hipLaunchKernelGGL((gru_kernel<64, true>), grid, dim3(512), lds, s, *d);
    else hipLaunchKernelGGL((gru_kernel<64, false>), grid, dim3(512), lds, s, *d);
  } else {
    if (pdse_check_hip(hipFuncSetAttribute((const void*)gru_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "bigru lds attribute"))
      return 1;
    hipLaunchKernelGGL((gru_kernel<128, false>), grid, dim3(1024), lds, s, *d);
  }
  return pdse_check_launch("bigru");
}

// ---------------------------------------------------------------------------------------
// GroupNorm(1, C, eps 1e-8) of the row and column branches + the AIA layer update
// (dbaiat.py:142,147-148).  Two launches: fixed-order partial sums (deterministic, no float
// atomics), then every workgroup of the apply pass folds the 64 partials in double.
// ---------------------------------------------------------------------------------------
#define GN_PARTS 64
__global__ __launch_bounds__(256) void gn_stats_kernel(const pdse_gncomb_desc d) {
  __shared__ float red[4][4];
  const int b = blockIdx.y, part = blockIdx.x;
  const int64_t n = (int64_t)d.C * d.plane;
  const float* r = d.row + (int64_t)b * n;
  const float* c = d.col + (int64_t)b * n;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < n; i += (int64_t)GN_PARTS * 256) {
    const float a = r[i], e = c[i];
    s[0] += a;
    s[1] += a * a;
    s[2] += e;
    s[3] += e * e;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 4)
    d.stats[((size_t)b * GN_PARTS + part) * 4 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const pdse_gncomb_desc d) {
  __shared__ float st[4];
  const int b = blockIdx.y;
  const int64_t n = (int64_t)d.C * d.plane;
  if (threadIdx.x < 4) {
    double acc = 0.0;
    for (int p = 0; p < GN_PARTS; ++p) acc += (double)d.stats[((size_t)b * GN_PARTS + p) * 4 + threadIdx.x];
    st[threadIdx.x] = (float)(acc / (double)n);
  }
  __syncthreads();
  const float mr = st[0], mc = st[2];
  const float rr = 1.0f / sqrtf(fmaxf(st[1] - mr * mr, 0.f) + d.eps);
  const float rc = 1.0f / sqrtf(fmaxf(st[3] - mc * mc, 0.f) + d.eps);
  const int64_t off = (int64_t)b * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / d.plane);
    const float gr = (d.row[off + i] - mr) * rr * d.g_row[c] + d.b_row[c];
    const float gc = (d.col[off + i] - mc) * rc * d.g_col[c] + d.b_col[c];
    d.out[off + i] = d.base[off + i] + d.k1 * gr + d.k2 * gc;
  }
}

int pdse_gncomb_launch(const pdse_gncomb_desc* d, hipStream_t s) {
  REQ(d && d->base && d->row && d->col && d->g_row && d->b_row && d->g_col && d->b_col && d->stats && d->out,
      "gn_combine: null pointer");
  REQ(d->B > 0 && d->B <= 65535 && d->C > 0 && d->plane > 0, "gn_combine: bad sizes");
  hipLaunchKernelGGL(gn_stats_kernel, dim3(GN_PARTS, d->B), dim3(256), 0, s, *d);
  const int64_t n = (int64_t)d->C * d->plane;
  int bx = (int)((n + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(bx, d->B), dim3(256), 0, s, *d);
  return pdse_check_launch("gn_combine");
}

// ---------------------------------------------------------------------------------------
// AHAM (dbaiat.py:266-288): channel means of the 4 layer outputs, softmax over the layers of
// conv1(mean) (conv1 is linear, so it commutes with the average pool), weighted merge.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aham_mean_kernel(const pdse_aham_desc d) {
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y, i = blockIdx.z;
  const float* x = d.x[i] + ((int64_t)b * d.C + c) * d.plane;
  float s = 0.f;
  for (int64_t q = threadIdx.x; q < d.plane; q += 256) s += x[q];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) d.means[((size_t)i * d.B + b) * d.C + c] = (red[0] + red[1] + red[2] + red[3]) / (float)d.plane;
}

__global__ __launch_bounds__(256) void aham_merge_kernel(const pdse_aham_desc d) {
  __shared__ float w[4];
  const int b = blockIdx.y;
  if (threadIdx.x < 4) {
    float y = d.bias;
    for (int c = 0; c < d.C; ++c) y += d.w[c] * d.means[((size_t)threadIdx.x * d.B + b) * d.C + c];
    w[threadIdx.x] = y;
  }
  __syncthreads();
  const float mx = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
  const float e0 = expf(w[0] - mx), e1 = expf(w[1] - mx), e2 = expf(w[2] - mx), e3 = expf(w[3] - mx);
  const float inv = 1.0f / (e0 + e1 + e2 + e3);
  const int64_t n = (int64_t)d.C * d.plane, off = (int64_t)b * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x3 = d.x[3][off + i];
    d.out[off + i] = x3 + inv * (e0 * d.x[0][off + i] + e1 * d.x[1][off + i] + e2 * d.x[2][off + i] + e3 * x3);
  }
}

int pdse_aham_launch(const pdse_aham_desc* d, hipStream_t s) {
  REQ(d && d->x[0] && d->x[1] && d->x[2] && d->x[3] && d->w && d->means && d->out, "aham: null pointer");
  REQ(d->B > 0 && d->B <= 65535 && d->C > 0 && d->C <= 65535 && d->plane > 0, "aham: bad sizes");
  hipLaunchKernelGGL(aham_mean_kernel, dim3(d->C, d->B, 4), dim3(256), 0, s, *d);
  const int64_t n = (int64_t)d->C * d->plane;
  int bx = (int)((n + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(aham_merge_kernel, dim3(bx, d->B), dim3(256), 0, s, *d);
  return pdse_check_launch("aham");
}

// ---------------------------------------------------------------------------------------
// [N][R][Cc] -> [N][Cc][R] through a padded 32x32 LDS tile: both the read and the write are
// 128-byte row segments.  HBM-bound (one read + one write of the tensor).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const pdse_transpose_desc d) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* src = d.in + (size_t)n * d.R * d.Cc;
  float* dst = d.out + (size_t)n * d.R * d.Cc;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < d.R && c < d.Cc) tile[ty + 8 * k][tx] = src[(size_t)r * d.Cc + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < d.R && c < d.Cc) dst[(size_t)c * d.R + r] = tile[tx][ty + 8 * k];
  }
}

int pdse_transpose_launch(const pdse_transpose_desc* d, hipStream_t s) {
  REQ(d && d->in && d->out && d->in != d->out, "transpose: null or aliased pointers");
  REQ(d->N > 0 && d->N <= 65535 && d->R > 0 && d->Cc > 0 && (d->R + 31) / 32 <= 65535, "transpose: bad sizes");
  hipLaunchKernelGGL(transpose_kernel, dim3((d->Cc + 31) / 32, (d->R + 31) / 32, d->N), dim3(256), 0, s, *d);
  return pdse_check_launch("transpose");
}

// ---------------------------------------------------------------------------------------
// Dual-branch prior: magnitude of the input (mode 0) and the masked-magnitude / decoded real-imag
// recombination (mode 1), dbaiat.py:389, :403-411.  Elementwise, HBM-bound.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crm_kernel(const pdse_crm_desc d) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (q >= d.plane) return;
  const float re = d.x[((int64_t)b * 2) * d.plane + q], im = d.x[((int64_t)b * 2 + 1) * d.plane + q];
  const float mag = sqrtf(re * re + im * im);
  if (d.mode == 0) {
    d.out[(int64_t)b * d.plane + q] = mag;
    return;
  }
  const float o = d.o[(int64_t)b * d.plane + q];
  const float m1 = 1.0f / (1.0f + expf(-(d.a1 * o + d.b1)));
  const float m2 = tanhf(d.a2 * o + d.b2);
  const float mask = 1.0f / (1.0f + expf(-(d.a3 * (m1 * m2) + d.b3)));
  const float cs = mag > 0.f ? re / mag : 1.f, sn = mag > 0.f ? im / mag : 0.f;
  const float mo = mask * mag;
  d.out[((int64_t)b * 2) * d.plane + q] = mo * cs + d.ri[((int64_t)b * 2) * d.plane + q];
  d.out[((int64_t)b * 2 + 1) * d.plane + q] = mo * sn + d.ri[((int64_t)b * 2 + 1) * d.plane + q];
}

int pdse_crm_launch(const pdse_crm_desc* d, hipStream_t s) {
  REQ(d && d->x && d->out && (d->mode == 0 || (d->mode == 1 && d->o && d->ri)), "crm: null pointer or bad mode");
  REQ(d->B > 0 && d->B <= 65535 && d->plane > 0, "crm: bad sizes");
  hipLaunchKernelGGL(crm_kernel, dim3((unsigned)((d->plane + 255) / 256), d->B), dim3(256), 0, s, *d);
  return pdse_check_launch("crm");
}
