// lstm.hip — recurrent part of the grouped LSTM of the GCRN prior (model/gcrn.py:6-40).
//
// The input projections of all T frames are one batched gather-GEMM (gconv.hip); what is
// left is the strictly sequential h_{t-1} · W_hh^T per frame.  v1 runs one launch per frame
// (the launch boundary is the inter-workgroup barrier; the whole prior is replayed from a
// hipGraph): a workgroup owns 8 hidden units x 4 gates = 32 gate columns (MFMA M) of one
// group for a 32-item batch tile (MFMA N, on the lanes), its 8 waves split K = H and are
// summed through LDS, then thread (unit, item) applies the cell update.  h is kept
// transposed [H][Bp] (ping-pong) so the B operand is a coalesced 128-B row per k.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include "pdse.h"
#include "pdse_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// 512 threads = 8 waves; wave w multiplies k-steps [w*KS/8, (w+1)*KS/8).  All of a wave's
// operands (KS/8 weight fragments + KS/8 rows of h) are requested before the first MFMA:
// the frame is latency-bound, so the loads must overlap each other, not the arithmetic.
template <int KPW>  // k-steps per wave
__global__ __launch_bounds__(512) void lstm_step_kernel(const pdse_lstm_desc d, const int t, const int pp) {
  __shared__ float red[8][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int slice = blockIdx.x, g = blockIdx.y, bt = blockIdx.z;
  const int H = d.H, Bp = d.Bp, G = d.G;
  const float* hprev = d.hT + ((size_t)(pp * G + g) * H) * Bp + bt * 32;
  float* hnext = d.hT + ((size_t)((pp ^ 1) * G + g) * H) * Bp + bt * 32;
  const int KS = H / 2;
  const float* A = d.whh + ((size_t)(g * (H / 8) + slice) * KS) * 64 + lane;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (t > 0) {  // h_{-1} = 0
    float av[KPW], bv[KPW];
    const int ks0 = wave * KPW;
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      av[i] = A[(size_t)(ks0 + i) * 64];
      bv[i] = hprev[(size_t)(2 * (ks0 + i) + h) * Bp + col];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep all 2*KPW loads in flight together (hipcc re-serialises them otherwise)
#pragma unroll
    for (int i = 0; i < KPW; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
  }
  // gate pre-activations of this frame, requested before the reduction barrier
  const int u = threadIdx.x >> 5 & 7, bb = threadIdx.x & 31;   // threads 0..255 finish the cell update
  const int b = bt * 32 + bb;
  const int hu = slice * 8 + u;
  float gxv[4] = {0.f, 0.f, 0.f, 0.f};
  float c_old = 0.f;
  const size_t ci = ((size_t)g * H + hu) * Bp + b;
  if (threadIdx.x < 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gxv[q] = d.gx[(((size_t)g * d.T + t) * (4 * H) + (size_t)q * H + hu) * Bp + b];
    if (t > 0) c_old = d.cst[ci];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][col] = acc[r];
  __syncthreads();
  if (threadIdx.x < 256) {
    float gate[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = q * 8 + u;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[w][i][bb];
      gate[q] = s + gxv[q];
    }
    const float c = sigm(gate[1]) * c_old + sigm(gate[0]) * tanhf(gate[2]);
    const float hv = sigm(gate[3]) * tanhf(c);
    d.cst[ci] = c;
    hnext[(size_t)hu * Bp + bb] = hv;
    if (b < d.B) d.y[(int64_t)b * d.y_sb + (int64_t)t * d.y_st + (int64_t)hu * d.y_su + (int64_t)g * d.y_sg] = hv;
  }
}

int pdse_lstm_launch(const pdse_lstm_desc* d, hipStream_t s) {
  if (!d || !d->gx || !d->whh || !d->hT || !d->cst || !d->y) {
    pdse_set_error("lstm: null pointer");
    return 1;
  }
  if (d->B <= 0 || d->T <= 0 || d->G <= 0 || d->H != 512 || d->Bp % 32 != 0 || d->Bp < d->B) {
    pdse_set_error("lstm: bad sizes (H == 512 as in gcrn.py:9-16, Bp % 32 == 0, Bp >= B)");
    return 1;
  }
  const dim3 grid(d->H / 8, d->G, d->Bp / 32), block(512);
  for (int t = 0; t < d->T; ++t) hipLaunchKernelGGL(lstm_step_kernel<32>, grid, block, 0, s, *d, t, t & 1);
  return pdse_check_launch("lstm");
}

// =======================================================================================
// Both layers of the grouped LSTM + LayerNorm 1 as a LAYER WAVEFRONT (model/gcrn.py:22-35).
//
// The per-frame launch above leaves the chain at 2 x T dependent launches.  What is sequential is only
// h_t -> h_{t+1} inside a layer; layer 2 can work on frame t-2 while layer 1 works on frame t.  One launch
// per wavefront step s = 0 .. T+1 carries three stages, each a 32-row x K=512 x 32-item matvec per workgroup:
//   A  layer 1, frame s:        gates = gx1[s] + W_hh1 h1[s-1]           -> h1[s], LayerNorm partial sums
//   B  layer-2 input, frame s-1: gx2 = rs_b (W'_ih2 h1[s-1] - mu_b R) + C  (LayerNorm 1 folded, below)
//   C  layer 2, frame s-2:      gates = gx2 + W_hh2 h2[s-3]               -> h2[s-2] -> y
// so the chain is T + 2 launches instead of 2 T, the [B,T,1024] LayerNorm output, the layer-2 input projection
// GEMMs and their 210 MB gate buffer disappear (gx2 lives for two frames), and all 256 CUs have work
// (3 x 128 workgroups at B = 32).
//
// LayerNorm folded into stage B:  W_ih LN(y) = rs (W' y) - rs mu (W' 1) + (W_ih beta),  W' = W_ih diag(gamma),
// with mu_b, rs_b = rsqrt(var_b + eps) over the 1024 interleaved outputs of both groups (stack(dim=-1) + flatten:
// feature j = 2 u + g).  Stage A writes, per workgroup, the sums of h and h^2 over its 8 units for every batch
// item; stage B adds the 128 partials in a fixed order (no atomics: results do not depend on scheduling).
//
// Operand layout: the state h is kept [H/8][2][Bp][4] (unit 8 kq + 2 i + hh at [kq][hh][b][i]), so that the B
// operand of four consecutive k-steps is ONE 16-byte load per lane (the per-frame kernel issues four 4-byte loads),
// and the weights are packed four k-steps deep like the gather-GEMM's (packing.pack_a4).  Stage B walks K in the
// order (source group g', kq, i, hh) of that layout; the host permutes W'_ih2's columns to match.
// =======================================================================================
__device__ __forceinline__ float sigm_g(float x) { return 1.0f / (1.0f + expf(-x)); }

// index of hidden unit u, batch item b in the state layout [H/8][2][Bp][4]
__device__ __forceinline__ size_t hidx(const int u, const int b, const int Bp) {
  return (((size_t)(u >> 3) * 2 + (u & 1)) * Bp + b) * 4 + ((u & 7) >> 1);
}

__device__ long long* g_trace_l = nullptr;   // PDSE_GLSTM_TRACE=1 (diagnostic): [workgroup][wave][8] clock stamps of step T/2
#define LSTAMP(i)                                                                                               \
  do {                                                                                                          \
    if (trace && lane == 0)                                                                                     \
      trace[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 8 + (i)] =  \
          (i) == 0 ? wall_clock64() : clock64();                                                                \
  } while (0)

// NS = slices of 8 hidden units (32 gate rows) per workgroup.  NS = 1 (rounds 2-4): 3 x 128 workgroups per step, each pulling its
// 64 KB weight slice AND the 64 KB state of its group through one CU's L2 port - 48 MB per step, half of it the state, and 128 of
// the 256 CUs hold two workgroups (A alone 6.9 us, A + C 11.3: the stages queue for the same ports).  NS = 2 (round 4): the two
// slices of a workgroup share ONE fetch of the state (36 MB per step), 3 x 64 workgroups sit on a CU of their own, every summation
// keeps its order (reduction over the 8 waves per row, LayerNorm partials per 8 units): results are bit-identical to NS = 1.
template <int NS>
__global__ __launch_bounds__(512) void glstm_wave_kernel(const pdse_glstm_desc d, const int s, const int stage_mask) {
  extern __shared__ __attribute__((aligned(16))) float lds_[];
  float (*red)[32 * NS][33] = reinterpret_cast<float (*)[32 * NS][33]>(lds_);                    // [8 waves][rows][items + 1]
  float (*stat)[32][2] = reinterpret_cast<float (*)[32][2]>(lds_ + 8 * 32 * NS * 33);             // [16][32][2]
  float (*gxs)[32][8 * NS] = reinterpret_cast<float (*)[32][8 * NS]>(lds_ + 8 * 32 * NS * 33 + 16 * 32 * 2);   // stage A: gate pre-activations [gate][item][unit]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int slice = blockIdx.x * NS, g = blockIdx.y;
  const int nbt = d.Bp >> 5;
  const int stage = blockIdx.z / nbt, bt = blockIdx.z - stage * nbt;
  const int t = s - stage;
  if (t < 0 || t >= d.T || !((stage_mask >> stage) & 1)) return;
#ifdef PDSE_DIAG
  long long* const trace = (s == d.T / 2) ? g_trace_l : nullptr;
#else
  long long* const trace = nullptr;
#endif
  LSTAMP(0);
  LSTAMP(1);
  const int H = d.H, Bp = d.Bp, G = d.G;
  const size_t hsz = (size_t)H * Bp;               // one group's state
  const int par = t & 1;

  // ---- operands of this stage's matvec
  const float* Aw;                                  // [H/8 groups of 4 k-steps][64 lanes][4]
  const float* hsrc0;                               // state read by k-groups 0..31 of this workgroup's K range
  const float* hsrc1;                               // ... by k-groups 32..63
  int kq0, kq1;                                     // first state row block of each half
  bool skip = false;
  if (stage == 0) {                                 // W_hh1 h1[t-1]
    Aw = d.whh1;
    hsrc0 = hsrc1 = d.hT1 + ((size_t)par * G + g) * hsz;
    kq0 = 0, kq1 = 32;
    skip = t == 0;
  } else if (stage == 1) {                          // W'_ih2 y1[t]: features of chunk g = units 256g..256g+255 of BOTH groups
    Aw = d.wih2;
    const float* base = d.hT1 + (size_t)((t + 1) & 1) * G * hsz;   // h1[t] was written to parity (t & 1) ^ 1
    hsrc0 = base, hsrc1 = base + hsz;               // source group g' = 0, then g' = 1
    kq0 = kq1 = 32 * g;
  } else {                                          // W_hh2 h2[t-1]
    Aw = d.whh2;
    hsrc0 = hsrc1 = d.hT2 + ((size_t)par * G + g) * hsz;
    kq0 = 0, kq1 = 32;
    skip = t == 0;
  }
  const float4* A4 = reinterpret_cast<const float4*>(Aw) + ((size_t)(g * (H / 8) + slice) * (H / 8)) * 64 + lane;

  f32x16 acc[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  if (!skip) {
    // wave w multiplies k-groups 8w .. 8w+7 (32 k-steps); all loads are requested before the first MFMA
    float4 av[NS][8], bv[8];
    const int q0 = wave * 8;
    const float* hs = q0 < 32 ? hsrc0 : hsrc1;
    const int kq = (q0 < 32 ? kq0 : kq1) + (q0 & 31);
    const float4* B4 = reinterpret_cast<const float4*>(hs) + ((size_t)kq * 2 + h) * Bp + bt * 32 + col;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int n = 0; n < NS; ++n) av[n][i] = A4[((size_t)n * (H / 8) + q0 + i) * 64];   // slice + n: the next (H/8) k-groups of the packed weights
      bv[i] = B4[(size_t)i * 2 * Bp];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (trace) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      LSTAMP(2);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n][i].x, bv[i].x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n][i].y, bv[i].y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n][i].z, bv[i].z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n][i].w, bv[i].w, acc[n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue operands, requested before the reduction barrier
  constexpr int NFIN = 256 * NS;                                 // (unit, batch item) threads
  const int uu = (threadIdx.x >> 5) & (8 * NS - 1), bb = threadIdx.x & 31;
  const int sl = uu >> 3, u = uu & 7;                            // slice within the workgroup, unit within the slice
  const int b = bt * 32 + bb;
  const int hu = (slice + sl) * 8 + u;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  float4 gx4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float c_old = 0.f;
  const size_t ci = ((size_t)g * H + hu) * Bp + b;
  if (threadIdx.x < NFIN) {
    if (stage == 0) {
      // gx1 [G][T][Bp][4H] (gate rows innermost, pdse.h): thread i fetches four consecutive units of one gate and item
      // as ONE 16-byte load (one instruction per thread, 32 lines per wave instead of 4 x 32) and the values reach
      // their (unit, item) threads through LDS behind the reduction barrier
      const int lq = threadIdx.x / (64 * NS), lb = (threadIdx.x / (2 * NS)) & 31, lu = (threadIdx.x & (2 * NS - 1)) * 4;
      gx4 = *reinterpret_cast<const float4*>(d.gx1 + (((size_t)g * d.T + t) * Bp + bt * 32 + lb) * (4 * H) + (size_t)lq * H + slice * 8 + lu);
      if (t > 0) c_old = d.cst1[ci];
    } else if (stage == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = d.gx2[(((size_t)par * G + g) * (4 * H) + (size_t)q * H + hu) * Bp + b];
      if (t > 0) c_old = d.cst2[ci];
    }
  }
  if (stage == 1) {
    // LayerNorm statistics of frame t: 2 groups x H/8 slices partial (sum, sumsq) pairs per batch item, fixed order
    const int p = threadIdx.x >> 5;                  // 16 parts of 8 entries
    const float2* P2 = reinterpret_cast<const float2*>(d.part) + ((size_t)par * G * (H / 8)) * Bp + b;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 v = P2[(size_t)(p * 8 + e) * Bp];
      s1 += v.x;
      s2 += v.y;
    }
    stat[p][bb][0] = s1;
    stat[p][bb][1] = s2;
  }
  LSTAMP(3);
#pragma unroll
  for (int n = 0; n < NS; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][32 * n + (r & 3) + 8 * (r >> 2) + 4 * h][col] = acc[n][r];
  if (stage == 0 && threadIdx.x < NFIN) {
    const int lq = threadIdx.x / (64 * NS), lb = (threadIdx.x / (2 * NS)) & 31, lu = (threadIdx.x & (2 * NS - 1)) * 4;
    *reinterpret_cast<float4*>(&gxs[lq][lb][lu]) = gx4;
  }
  __syncthreads();
  if (stage == 0 && threadIdx.x < NFIN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pre[q] = gxs[q][bb][uu];
  }
  LSTAMP(4);
  const bool fin = threadIdx.x < NFIN;              // (unit, item) threads; every wave still reaches the barrier below
  float gate[4] = {0.f, 0.f, 0.f, 0.f};
  if (fin) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 32 * sl + q * 8 + u;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += red[w][i][bb];
      gate[q] = sum;
    }
  }
  if (stage == 1) {
    if (!fin) return;                               // no further barrier in this stage
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      s1 += (double)stat[p][bb][0];
      s2 += (double)stat[p][bb][1];
    }
    const double n = (double)(G * H);
    const double mu = s1 / n;
    const double var = fmax(s2 / n - mu * mu, 0.0);                // biased variance, like nn.LayerNorm
    const float rs = (float)(1.0 / sqrt(var + (double)d.eps)), muf = (float)mu;
    float* out = d.gx2 + (((size_t)par * G + g) * (4 * H)) * Bp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t row = (size_t)q * H + hu;
      out[row * Bp + b] = rs * (gate[q] - muf * d.r2[(size_t)g * 4 * H + row]) + d.c2[(size_t)g * 4 * H + row];
    }
    return;
  }
  float hv = 0.f;
  if (fin) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gate[q] += pre[q];
    const float c = sigm_g(gate[1]) * c_old + sigm_g(gate[0]) * tanhf(gate[2]);
    hv = sigm_g(gate[3]) * tanhf(c);
    if (stage == 0) {
      d.cst1[ci] = c;
      d.hT1[((size_t)(par ^ 1) * G + g) * hsz + hidx(hu, b, Bp)] = hv;
      stat[uu][bb][0] = hv;                         // stat[] is unused in stages 0 / 2: staging for the partial sums
    } else {
      d.cst2[ci] = c;
      d.hT2[((size_t)(par ^ 1) * G + g) * hsz + hidx(hu, b, Bp)] = hv;
      if (b < d.B) d.y[(int64_t)b * d.y_sb + (int64_t)t * d.y_st + (int64_t)hu * d.y_su + (int64_t)g * d.y_sg] = hv;
    }
  }
  LSTAMP(5);
  if (stage != 0) return;
  __syncthreads();
  // LayerNorm partial sums over the 8 units of each slice (the same summation order at every batch size and for every NS)
  if (threadIdx.x < 32 * NS) {
    const int ps = threadIdx.x >> 5;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = stat[8 * ps + k][bb][0];
      s1 += v;
      s2 += v * v;
    }
    float2* P2 = reinterpret_cast<float2*>(d.part) + (((size_t)par * G + g) * (H / 8) + slice + ps) * Bp + b;
    *P2 = make_float2(s1, s2);
  }
}


int pdse_glstm_launch(const pdse_glstm_desc* d, hipStream_t s) {
  if (!d || !d->gx1 || !d->whh1 || !d->wih2 || !d->r2 || !d->c2 || !d->whh2 || !d->hT1 || !d->cst1 || !d->hT2 ||
      !d->cst2 || !d->gx2 || !d->part || !d->y) {
    pdse_set_error("glstm: null pointer");
    return 1;
  }
  if (d->B <= 0 || d->T <= 0 || d->G != 2 || d->H != 512 || d->Bp % 32 != 0 || d->Bp < d->B || (d->Bp / 32) * 3 > 65535) {
    pdse_set_error("glstm: bad sizes (H == 512, G == 2 as in gcrn.py:9-16, Bp % 32 == 0, Bp >= B)");
    return 1;
  }
  if (d->slices < 0 || d->slices > 2) {
    pdse_set_error("glstm: slices is 0 / 1 (one slice of 8 units per workgroup) or 2");
    return 1;
  }
  const int NS = d->slices == 2 ? 2 : 1;
  const dim3 grid(d->H / 8 / NS, d->G, 3 * (d->Bp / 32)), block(512);
  const size_t lds = (size_t)(8 * 32 * NS * 33 + 16 * 32 * 2 + 4 * 32 * 8 * NS) * sizeof(float);
  static unsigned long long attr_mask = 0;
  if (NS == 2 && pdse_lds_attr((const void*)glstm_wave_kernel<2>, &attr_mask, "glstm lds attribute")) return 1;
  // PDSE_GLSTM_MASK (diagnostic, tools/time_glstm.py): run only some stages to time them apart - results are then wrong
  static const int mask = PDSE_DIAG_ENV("PDSE_GLSTM_MASK") ? atoi(PDSE_DIAG_ENV("PDSE_GLSTM_MASK")) : 7;
  static const bool tracing = PDSE_DIAG_ENV("PDSE_GLSTM_TRACE") != nullptr;
  static long long* tbuf = nullptr;
  const size_t nw = (size_t)grid.x * grid.y * grid.z * 8;
  if (tracing) {
    if (!tbuf) {
      (void)hipMalloc(&tbuf, nw * 64);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_l), &tbuf, sizeof(tbuf));
    }
    (void)hipMemsetAsync(tbuf, 0, nw * 64, s);
  }
  for (int st = 0; st < d->T + 2; ++st) {
    if (NS == 2) hipLaunchKernelGGL(glstm_wave_kernel<2>, grid, block, lds, s, *d, st, mask);
    else hipLaunchKernelGGL(glstm_wave_kernel<1>, grid, block, lds, s, *d, st, mask);
  }
  if (tracing) {   // diagnostic: stamps of wavefront step T/2 - start spread (100 MHz clock) and shader clocks since the wave started
    (void)hipStreamSynchronize(s);
    long long* h = (long long*)malloc(nw * 64);
    (void)hipMemcpy(h, tbuf, nw * 64, hipMemcpyDeviceToHost);
    for (int stg = 0; stg < 3; ++stg) {
      double sum[8] = {0};
      long long w0 = -1, w1 = 0;
      size_t n = 0;
      const size_t per = nw / 3;
      for (size_t i = stg * per; i < (stg + 1) * per; ++i) {
        const long long* q = h + i * 8;
        if (!q[0]) continue;
        ++n;
        if (w0 < 0 || q[0] < w0) w0 = q[0];
        if (q[0] > w1) w1 = q[0];
        for (int k = 2; k < 6; ++k) sum[k] += q[k] ? (double)(q[k] - q[1]) : 0.0;
      }
      if (n) fprintf(stderr, "glstm trace stage %d (%zu waves): start spread %.2f us; cycles since wave start: loads landed %.0f, mfma issued %.0f, after barrier %.0f, end %.0f\n",
                     stg, n, (w1 - w0) * 0.01, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n);
    }
    free(h);
  }
  return pdse_check_launch("glstm");
}
