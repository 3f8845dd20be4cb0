// lstmp.hip - the grouped LSTM of the GCRN prior (model/gcrn.py:6-40) as ONE persistent launch, for the batches the
// reference's own entry point runs (generate_wav: B = 1; here B <= 8).
//
// Why another form (csrc/lstm.hip keeps the layer wavefront for B > 8 and for several batches in flight): the wavefront is
// T + 2 dependent launches of ~12 us that re-fetch 24 MB of weights each, whatever the batch - at B = 1 it is half of a
// file's 9 ms.  At B <= 8 the state of a frame is tiny (2 x 1024 x B floats), so the weights can stay on chip and the step
// becomes an exchange of the state:
//   * 256 workgroups, each owning 4 hidden units of layer 1 and 4 of layer 2 of one group: their 16 + 16 + 16 gate rows
//     (W_hh1 | W_ih2 diag(gamma_ln1) | W_hh2, K = 512 each = 96 KB) live in registers for all frames (48 per thread);
//   * two-stage wavefront: at step s layer 1 works on frame s, layer 2 on frame s - 1 (LayerNorm 1 folded into its input
//     projection exactly as in glstm_wave_kernel: W_ih LN(y) = rs (W' y - mu W' 1) + W_ih beta);
//   * every step ends with each workgroup publishing its 8 x B new state values as 8-byte {tag = step + 1, value}
//     granules (one sc1 store each: the data is the flag, MI355X_MICROARCH.md "R2"), and begins with every thread polling
//     its share of the 1536 x B granules the workgroup needs (all of h1 for the LayerNorm and both matvecs, h2 of its
//     group) with sc1 loads until the tags match - no fence, no flag, no atomics, results independent of arrival order;
//   * a 2-slot ring by step parity is enough: a workgroup can only publish step s + 1 after it has read every step-s
//     granule, i.e. after every workgroup has finished reading the step s - 1 granules of the same slot;
//   * the granule buffer is zeroed by a memset node in front of the launch (tags restart at 1 in every call);
//   * every poll is bounded: on a timeout the workgroup records the step in d.status and leaves, the others follow at
//     their next poll - a launch that cannot have its 256 workgroups resident at once ends with an error code, it does
//     not hang.  The host only selects this form for a plan that owns the GPU while it runs (one batch in flight).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"
#include "pdse_internal.h"

// No implicit contraction: an item's arithmetic must not depend on the batch it is in (with contraction left to hipcc the
// packed two-item forms of a * a + c * c round differently from the scalar ones); every fused multiply-add is an fmaf.
#pragma clang fp contract(off)

namespace {

constexpr int H = 512, G = 2, NWG = 256, UPW = 4;   // hidden units, groups, workgroups, units per workgroup and layer

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

// gates on v_exp_f32 / v_rcp_f32 (the cell update sits on the critical path of every step: the library expf / tanhf are
// ~300 instructions per cell, a third of a step's compute chain; csrc/gru3.hip uses the same forms)
__device__ __forceinline__ float sigm_p(const float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_p(const float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(const float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over each row of 16 lanes in a fixed tree (quad, quad pair, half row, row): every lane of the row ends with the total
__device__ __forceinline__ float row_sum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float rl(const float v, const int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

__device__ __forceinline__ u64 gran_load(const u64* p) {
  return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dwordx2 sc1
}
__device__ __forceinline__ void gran_store(u64* p, const unsigned tag, const float v) {
  __hip_atomic_store((gu64*)p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define PDSE_LSTMP_SPINS 400000   // polls of one step before a workgroup gives up (~0.3 s)

template <int BQ>
__global__ __launch_bounds__(512, BQ == 1 ? 4 : 2) void glstm_persist_kernel(const pdse_glstmp_desc d) {
  __shared__ __attribute__((aligned(16))) float x1[2 * H * BQ];   // h1 of the previous frame, feature f = 2 u + g' (stack(dim=-1) + flatten), [f][b]
  __shared__ __attribute__((aligned(16))) float x2[H * BQ];       // h2 of this group two frames back, [u][b]
  __shared__ float red[3][16][BQ];                                // gate-row sums: W_hh1 h1 | W'_ih2 y1 | W_hh2 h2
  __shared__ float lnp[8][BQ][2];                                 // LayerNorm partial sums per wave
  __shared__ int dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x, g = wg >> 7, u0 = (wg & 127) * UPW;
  const int row = tid >> 5, l = tid & 31;          // row = ui * 4 + q: unit u0 + ui, gate q
  const int T = d.T, B = d.B, Bp = d.Bp;

  // ---- this thread's 48 weights: 16 k-steps (k = l + 32 m) of its gate row in each of the three matrices
  float w1[16], w2i[16], w2h[16];
  {
    const size_t ro = ((size_t)(g * H + u0) * 4 + row) * H + l;   // [G][H units][4 gates][H]: rows of a workgroup are contiguous
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      w1[m] = d.w1[ro + 32 * m];
      w2i[m] = d.w2i[ro + 32 * m];
      w2h[m] = d.w2h[ro + 32 * m];
    }
  }
  // ---- gate threads: (layer, unit, item)
  const bool gate_thr = tid < 2 * UPW * BQ;
  const int glayer = tid / (UPW * BQ), gui = (tid / BQ) % UPW, gb = tid % BQ;
  const int gu = u0 + gui;
  const bool item = gb < B;
  float c_state = 0.f;
  float gxn[4] = {0.f, 0.f, 0.f, 0.f};
  float r2v[4] = {0.f, 0.f, 0.f, 0.f}, c2v[4] = {0.f, 0.f, 0.f, 0.f};
  const float* gxp = d.gx1 + ((size_t)g * T * Bp + gb) * (4 * H) + gu;   // + t * Bp * 4H + q * H
  if (gate_thr && glayer == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r2v[q] = d.r2[(size_t)g * 4 * H + q * H + gu];
      c2v[q] = d.c2[(size_t)g * 4 * H + q * H + gu];
    }
  }
  if (gate_thr && glayer == 0 && item) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gxn[q] = gxp[(size_t)q * H];
  }
  if (tid == 0) dead = 0;
  __syncthreads();

  u64* const G1 = d.gran;                                  // [2 slots][2 H features][BQ]
  u64* const G2 = d.gran + (size_t)2 * 2 * H * BQ;         // [2 slots][G][H][BQ]

#pragma clang loop unroll(disable)
  for (int s = 0; s <= T; ++s) {
    // gate pre-activations of layer 1's NEXT frame: in flight during this step
    float gx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gx[q] = gxn[q];
    if (gate_thr && glayer == 0 && item && s + 1 < T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) gxn[q] = gxp[((size_t)(s + 1) * Bp) * (4 * H) + (size_t)q * H];
    }
    // ---- gather the state published at the end of step s - 1
    float v1[2 * BQ], v2[BQ];
    if (s > 0) {
      const unsigned tag = (unsigned)s;
      const u64* const p1 = G1 + ((size_t)((s - 1) & 1) * 2 * H + 2 * tid) * BQ;
      const u64* const p2 = G2 + (((size_t)((s - 1) & 1) * G + g) * H + tid) * BQ;
      // every granule is polled until ITS tag matches; arrived ones are not read again (at B = 8 a pass over all 24 would
      // put 25 MB of sc1 loads on the fabric, chip-wide, per pass)
      constexpr unsigned ALL = (1u << (3 * BQ)) - 1u;
      unsigned got = 0;
      int spins = 0;
      while (true) {
        u64 x[3 * BQ];
#pragma unroll
        for (int e = 0; e < 3 * BQ; ++e)
          if (!((got >> e) & 1u)) x[e] = gran_load(e < 2 * BQ ? p1 + e : p2 + (e - 2 * BQ));
#pragma unroll
        for (int e = 0; e < 3 * BQ; ++e)
          if (!((got >> e) & 1u) && (unsigned)(x[e] >> 32) == tag) {
            got |= 1u << e;
            if (e < 2 * BQ) v1[e] = __uint_as_float((unsigned)x[e]);
            else v2[e - 2 * BQ] = __uint_as_float((unsigned)x[e]);
          }
        if (got == ALL) break;
        if (++spins > PDSE_LSTMP_SPINS) {
          dead = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 2 * BQ; ++e) v1[e] = 0.f;
#pragma unroll
      for (int e = 0; e < BQ; ++e) v2[e] = 0.f;
    }
    // LayerNorm 1 partial sums of this thread's two features, per item; fixed order: lanes by xor tree, waves 0..7
#pragma unroll
    for (int b = 0; b < BQ; ++b) {
      const float a = v1[b], c = v1[BQ + b];
      const float s1 = row_sum(a + c), s2 = row_sum(a * a + c * c);
      const float t1 = (rl(s1, 0) + rl(s1, 16)) + (rl(s1, 32) + rl(s1, 48)), t2 = (rl(s2, 0) + rl(s2, 16)) + (rl(s2, 32) + rl(s2, 48));
      if (lane == 0) {
        lnp[wave][b][0] = t1;
        lnp[wave][b][1] = t2;
      }
    }
#pragma unroll
    for (int e = 0; e < 2 * BQ; ++e) x1[2 * tid * BQ + e] = v1[e];
#pragma unroll
    for (int e = 0; e < BQ; ++e) x2[tid * BQ + e] = v2[e];
    __syncthreads();
    if (dead) {
      if (tid == 0) atomicCAS(d.status, 0, s + 1);
      return;
    }
    // ---- the three matvecs of this workgroup's 16 gate rows per layer
    float a1[BQ], aI[BQ], aH[BQ];
#pragma unroll
    for (int b = 0; b < BQ; ++b) a1[b] = aI[b] = aH[b] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int k = l + 32 * m;
#pragma unroll
      for (int b = 0; b < BQ; ++b) {
        a1[b] = fmaf(w1[m], x1[(2 * k + g) * BQ + b], a1[b]);          // h1 of this group: features 2 u + g
        aI[b] = fmaf(w2i[m], x1[(H * g + k) * BQ + b], aI[b]);         // chunk g of the interleaved layer-1 output
        aH[b] = fmaf(w2h[m], x2[k * BQ + b], aH[b]);
      }
      // one k-step at a time: hipcc otherwise keeps the LDS values of every k-step live and finishes the items one after the other
#pragma unroll
      for (int b = 0; b < BQ; ++b) asm volatile("" : "+v"(a1[b]), "+v"(aI[b]), "+v"(aH[b]));
    }
#pragma unroll
    for (int b = 0; b < BQ; ++b) {   // lanes 0-31 hold gate row 2 wave, lanes 32-63 gate row 2 wave + 1
      const float r1 = row_sum(a1[b]), rI = row_sum(aI[b]), rH = row_sum(aH[b]);
      const float r1a = rl(r1, 0) + rl(r1, 16), r1b = rl(r1, 32) + rl(r1, 48);
      const float rIa = rl(rI, 0) + rl(rI, 16), rIb = rl(rI, 32) + rl(rI, 48);
      const float rHa = rl(rH, 0) + rl(rH, 16), rHb = rl(rH, 32) + rl(rH, 48);
      if (lane == 0) {
        red[0][2 * wave][b] = r1a, red[0][2 * wave + 1][b] = r1b;
        red[1][2 * wave][b] = rIa, red[1][2 * wave + 1][b] = rIb;
        red[2][2 * wave][b] = rHa, red[2][2 * wave + 1][b] = rHb;
      }
    }
    __syncthreads();
    // ---- cell updates and publication
    if (gate_thr) {
      const unsigned tag = (unsigned)(s + 1);
      if (glayer == 0) {
        if (s < T) {
          float pre[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pre[q] = gx[q] + red[0][gui * 4 + q][gb];
          const float c = sigm_p(pre[1]) * c_state + sigm_p(pre[0]) * tanh_p(pre[2]);
          const float hv = sigm_p(pre[3]) * tanh_p(c);
          c_state = c;
          gran_store(G1 + ((size_t)(s & 1) * 2 * H + 2 * gu + g) * BQ + gb, tag, item ? hv : 0.f);
        }
      } else {
        float hv = 0.f;
        if (s > 0) {
          float s1 = 0.f, s2 = 0.f;                                       // the 8 wave partials in a fixed order
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            s1 += lnp[w][gb][0];
            s2 += lnp[w][gb][1];
          }
          const float muf = s1 * (1.0f / (G * H));
          const float var = fmaxf(s2 * (1.0f / (G * H)) - muf * muf, 0.f);   // biased variance, like nn.LayerNorm
          const float rs = __builtin_amdgcn_rsqf(var + d.eps);
          float pre[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pre[q] = (rs * (red[1][gui * 4 + q][gb] - muf * r2v[q]) + c2v[q]) + red[2][gui * 4 + q][gb];
          const float c = sigm_p(pre[1]) * c_state + sigm_p(pre[0]) * tanh_p(pre[2]);
          hv = sigm_p(pre[3]) * tanh_p(c);
          c_state = c;
          if (item) d.y[(int64_t)gb * d.y_sb + (int64_t)(s - 1) * d.y_st + (int64_t)gu * d.y_su + (int64_t)g * d.y_sg] = hv;
        }
        if (s < T) gran_store(G2 + (((size_t)(s & 1) * G + g) * H + gu) * BQ + gb, tag, item ? hv : 0.f);   // step 0: h2[-1] = 0
      }
    }
  }
}

template <int BQ>
int launch(const pdse_glstmp_desc* d, hipStream_t s) {
  // all 256 workgroups have to be resident at once (they wait for each other): check what this device admits
  static int cap[64];
  int dev = 0;
  if (pdse_check_hip(hipGetDevice(&dev), "glstmp: get device")) return 1;
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cap[dev]) {
    int per = 0, cus = 0;
    if (pdse_check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, glstm_persist_kernel<BQ>, 512, 0), "glstmp: occupancy")) return 1;
    if (pdse_check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "glstmp: CU count")) return 1;
    cap[dev] = per * cus > 0 ? per * cus : -1;
  }
  if (cap[dev] < NWG) {
    pdse_set_error("glstmp: this device cannot hold the 256 co-resident workgroups of the persistent LSTM");
    return 1;
  }
  if (pdse_check_hip(hipMemsetAsync(d->gran, 0, (size_t)4 * 2 * H * BQ * sizeof(u64), s), "glstmp: memset")) return 1;
  hipLaunchKernelGGL(glstm_persist_kernel<BQ>, dim3(NWG), dim3(512), 0, s, *d);
  return pdse_check_launch("glstmp");
}

}   // namespace

int pdse_glstmp_launch(const pdse_glstmp_desc* d, hipStream_t s) {
  if (!d || !d->gx1 || !d->w1 || !d->w2i || !d->w2h || !d->r2 || !d->c2 || !d->gran || !d->status || !d->y) {
    pdse_set_error("glstmp: null pointer");
    return 1;
  }
  if (d->B < 1 || d->B > 8 || d->T < 1 || d->G != G || d->H != H || d->Bp < d->B) {
    pdse_set_error("glstmp: bad sizes (1 <= B <= 8, H == 512, G == 2 as in gcrn.py:9-16, Bp >= B)");
    return 1;
  }
  if ((reinterpret_cast<uintptr_t>(d->gran) & 15) != 0) {
    pdse_set_error("glstmp: granule buffer must be 16-byte aligned");
    return 1;
  }
  if (d->B == 1) return launch<1>(d, s);
  if (d->B == 2) return launch<2>(d, s);
  if (d->B <= 4) return launch<4>(d, s);
  return launch<8>(d, s);
}
