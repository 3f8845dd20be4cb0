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
#include <stdint.h>

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
