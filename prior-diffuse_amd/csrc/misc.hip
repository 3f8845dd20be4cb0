// misc.hip — the HBM-bound and small kernels around the convolution stack:
// diffusion-step embedding, reverse-step update, sqrt/square companding, waveform
// front-end, overlap-add back-end, --sigma mask, LayerNorm.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pdse.h"
#include "pdse_internal.h"

#define REQ(cond, msg)          \
  do {                          \
    if (!(cond)) {              \
      pdse_set_error(msg);      \
      return 1;                 \
    }                           \
  } while (0)

// ---------------------------------------------------------------------------------------
// Diffusion-step embedding (model/diff3.py:62-95) and the 15 folded per-stage time biases.
// One workgroup per batch item; the three matrices are stored transposed so that thread j streams
// column j with coalesced reads.  Tiny (1.3 MFLOP per item): latency only.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x * (1.0f / (1.0f + expf(-x))); }

// blockDim = max(512, NF rounded up to a wavefront) so that every folded output has its own thread; the first 512
// threads run the two MLP layers.  The dot products keep one running sum (the summation order the golden vectors of
// the time embedding were checked with).
__global__ __launch_bounds__(1024) void time_embed_kernel(const pdse_time_desc d) {
  __shared__ float x0[128];
  __shared__ float y1[512];
  __shared__ float y2[512];
  const int b = blockIdx.x, j = threadIdx.x;
  const float t = d.t[b];
  if (j < 128) {
    // low + (high - low) * (t - floor(t)); an integral t reads one table row exactly
    float fl = floorf(t), ce = ceilf(t);
    int lo = (int)fl, hi = (int)ce;
    lo = min(max(lo, 0), d.max_steps - 1);
    hi = min(max(hi, 0), d.max_steps - 1);
    const float low = d.table[lo * 128 + j], high = d.table[hi * 128 + j];
    x0[j] = low + (high - low) * (t - fl);
  }
  __syncthreads();
  if (j < 512) {
    float s = d.b1[j];
    for (int i = 0; i < 128; ++i) s += d.p1T[i * 512 + j] * x0[i];
    y1[j] = silu_f(s);
  }
  __syncthreads();
  if (j < 512) {
    float s = d.b2[j];
    for (int i = 0; i < 512; ++i) s += d.p2T[i * 512 + j] * y1[i];
    s = silu_f(s);
    y2[j] = s;
    if (d.temb) d.temb[(size_t)b * 512 + j] = s;
  }
  __syncthreads();
  for (int o = j; o < d.NF; o += blockDim.x) {
    float a = d.bf[o];
    for (int i = 0; i < 512; ++i) a += d.wfT[(size_t)i * d.NF + o] * y2[i];
    d.out[(size_t)b * d.NF + o] = a;
  }
}

int pdse_time_launch(const pdse_time_desc* d, hipStream_t s) {
  REQ(d && d->t && d->table && d->p1T && d->b1 && d->p2T && d->b2 && d->out, "time_embed: null pointer");
  REQ(d->B > 0 && d->max_steps > 0 && d->NF >= 0, "time_embed: bad sizes");
  REQ(d->NF == 0 || (d->wfT && d->bf), "time_embed: folded weights missing");
  const int threads = d->NF <= 512 ? 512 : (d->NF >= 1024 ? 1024 : (d->NF + 63) / 64 * 64);
  hipLaunchKernelGGL(time_embed_kernel, dim3(d->B), dim3(threads), 0, s, *d);
  return pdse_check_launch("time_embed");
}

// ---------------------------------------------------------------------------------------
// Reverse-step arithmetic (trainer/complex_ddpm_trainer.py:942, :977, :995-996).  Every
// product/sum is rounded separately (no fma contraction) so the update is bit-identical to
// the reference's chain of fp32 tensor ops.  HBM-bound: 16-byte accesses when aligned.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float ew_one(int op, float a, float b, float c, float s0, float s1, float s2) {
  // plain operators under contract(off): the __f*_rn wrappers of the HIP headers carry the
  // translation unit's default contraction and still fuse into v_fma_f32
#pragma clang fp contract(off)
  switch (op) {
    case PDSE_EW_DIV: return a / s0;
    case PDSE_EW_UPDATE: {
      const float p = s1 * b;
      const float q = a - p;
      return s0 * q;
    }
    case PDSE_EW_UPDATE_FINAL: {
      const float p = s1 * b;
      const float q = a - p;
      const float r = s0 * q;
      const float u = r + c;
      return u * s2;
    }
    case PDSE_EW_ADD_MUL: {
      const float u = a + b;
      return u * s0;
    }
    default: return a;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void ew_kernel(const pdse_ew_desc d) {
  const int64_t nv = d.n / VEC;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    if constexpr (VEC == 4) {
      const float4 a = reinterpret_cast<const float4*>(d.a)[i];
      float4 b = make_float4(0, 0, 0, 0), c = make_float4(0, 0, 0, 0);
      if (d.b) b = reinterpret_cast<const float4*>(d.b)[i];
      if (d.c) c = reinterpret_cast<const float4*>(d.c)[i];
      float4 o;
      o.x = ew_one(d.op, a.x, b.x, c.x, d.s0, d.s1, d.s2);
      o.y = ew_one(d.op, a.y, b.y, c.y, d.s0, d.s1, d.s2);
      o.z = ew_one(d.op, a.z, b.z, c.z, d.s0, d.s1, d.s2);
      o.w = ew_one(d.op, a.w, b.w, c.w, d.s0, d.s1, d.s2);
      reinterpret_cast<float4*>(d.out)[i] = o;
    } else {
      d.out[i] = ew_one(d.op, d.a[i], d.b ? d.b[i] : 0.f, d.c ? d.c[i] : 0.f, d.s0, d.s1, d.s2);
    }
  }
  if constexpr (VEC == 4) {
    // scalar tail
    const int64_t i = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n) d.out[i] = ew_one(d.op, d.a[i], d.b ? d.b[i] : 0.f, d.c ? d.c[i] : 0.f, d.s0, d.s1, d.s2);
  }
}

int pdse_ew_launch(const pdse_ew_desc* d, hipStream_t s) {
  REQ(d && d->a && d->out && d->n > 0, "ew: null pointer or empty");
  REQ(d->op >= PDSE_EW_DIV && d->op <= PDSE_EW_ADD_MUL, "ew: unknown op");
  REQ(d->op != PDSE_EW_UPDATE || d->b, "ew: UPDATE needs b");
  REQ(d->op != PDSE_EW_UPDATE_FINAL || (d->b && d->c), "ew: UPDATE_FINAL needs b and c");
  REQ(d->op != PDSE_EW_ADD_MUL || d->b, "ew: ADD_MUL needs b");
  const uintptr_t al = (uintptr_t)d->a | (uintptr_t)d->b | (uintptr_t)d->c | (uintptr_t)d->out;
  const bool vec = (al & 15) == 0;
  const int64_t work = vec ? (d->n + 3) / 4 : d->n;
  int64_t blocks = (work + 255) / 256;
  if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (blocks < 1) blocks = 1;
  if (vec)
    hipLaunchKernelGGL(ew_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, *d);
  else
    hipLaunchKernelGGL(ew_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, *d);
  return pdse_check_launch("ew");
}

// ---------------------------------------------------------------------------------------
// Companding of [B,2,T,F]: phase kept, magnitude raised to 0.5 (front-end, :931-937) or 2
// (back-end, :1004-1008).  cos/sin(atan2(im,re)) is evaluated as re/|X|, im/|X| (1,0 at 0).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compand_kernel(const pdse_compand_desc d) {
  const int64_t total = (int64_t)d.B * d.plane;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
#pragma clang fp contract(off)
    const int64_t b = i / d.plane, q = i - b * d.plane;
    const int64_t ire = (2 * b) * d.plane + q, iim = ire + d.plane;
    const float re = d.in[ire], im = d.in[iim];
    const float rr = re * re, ii = im * im;
    const float mag = sqrtf(rr + ii);
    float cr = 1.f, sr = 0.f;
    if (mag > 0.f) {
      cr = re / mag;
      sr = im / mag;
    }
    const float m2 = d.mode == 0 ? sqrtf(mag) : mag * mag;
    if (d.F > 0) {
      const int64_t t = q / d.F, f = q - t * d.F;
      float* o = d.out + b * d.out_sb + t * d.out_st + f;
      o[0] = m2 * cr;
      o[d.out_sc] = m2 * sr;
    } else {
      d.out[ire] = m2 * cr;
      d.out[iim] = m2 * sr;
    }
  }
}

int pdse_compand_launch(const pdse_compand_desc* d, hipStream_t s) {
  REQ(d && d->in && d->out && d->B > 0 && d->plane > 0, "compand: bad descriptor");
  REQ(d->mode == 0 || d->mode == 1, "compand: mode must be 0 or 1");
  REQ(d->F >= 0 && (d->F == 0 || d->plane % d->F == 0), "compand: plane must be a multiple of F");
  int64_t blocks = ((int64_t)d->B * d->plane + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(compand_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *d);
  return pdse_check_launch("compand");
}

// ---------------------------------------------------------------------------------------
// Waveform front-end (:922-923 + torch.stft center=True reflect padding): one workgroup per
// utterance computes c = sqrt(sum x^2 / L), then writes reflect_pad(x / c).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wavprep_kernel(const pdse_wavprep_desc d) {
  __shared__ float red[16];
  __shared__ float cval;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = d.wav + (size_t)b * d.L;
  float c = 1.f;
  if (d.normalize) {
    const int len = d.lens ? min(max(d.lens[b], 1), d.L) : d.L;
    float ss = 0.f;
    for (int i = tid; i < len; i += 1024) ss += x[i] * x[i];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
      for (int w = 0; w < 16; ++w) tot += red[w];
      cval = sqrtf(tot / (float)len);
    }
    __syncthreads();
    c = cval;
  }
  if (tid == 0 && d.c) d.c[b] = c;
  const int Lp = d.L + 2 * d.pad;
  float* o = d.xpad + (size_t)b * Lp;
  for (int i = tid; i < Lp; i += 1024) {
    int k = i - d.pad;
    if (k < 0) k = -k;
    if (k >= d.L) k = 2 * (d.L - 1) - k;
    o[i] = x[k] / c;
  }
}

int pdse_wavprep_launch(const pdse_wavprep_desc* d, hipStream_t s) {
  REQ(d && d->wav && d->xpad && d->B > 0 && d->L > 0 && d->pad >= 0, "wavprep: bad descriptor");
  REQ(d->pad < d->L, "wavprep: reflect padding needs pad < L");
  hipLaunchKernelGGL(wavprep_kernel, dim3(d->B), dim3(1024), 0, s, *d);
  return pdse_check_launch("wavprep");
}

// ---------------------------------------------------------------------------------------
// Overlap-add + envelope normalisation + centre trim + rescale (torch.istft, :1010-1016).
// frames[b][n][t] already carry the synthesis window; every output sample sums two frames.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ola_kernel(const pdse_ola_desc d) {
  const int64_t total = (int64_t)d.B * d.L;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int half = d.n_fft / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i / d.L);
    const int n = (int)(i - (int64_t)b * d.L) + half;  // index in the un-trimmed signal
    float acc = 0.f, env = 0.f;
    // frames t with 0 <= n - t*hop < n_fft
    const int t_hi = n / d.hop;
    for (int t = t_hi; t >= 0 && n - t * d.hop < d.n_fft; --t) {
      if (t < d.T) {
        const int r = n - t * d.hop;
        acc += d.frames[((size_t)b * d.n_fft + r) * d.T + t];
        env += d.win2[r];
      }
    }
    float y = env > 1e-11f ? acc / env : 0.f;
    if (d.c) y *= d.c[b];
    d.out[i] = y;
  }
}

int pdse_ola_launch(const pdse_ola_desc* d, hipStream_t s) {
  REQ(d && d->frames && d->win2 && d->out, "ola: null pointer");
  REQ(d->B > 0 && d->T > 0 && d->L > 0 && d->n_fft > 0 && d->hop > 0, "ola: bad sizes");
  int64_t blocks = ((int64_t)d->B * d->L + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ola_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *d);
  return pdse_check_launch("ola");
}

// ---------------------------------------------------------------------------------------
// --sigma mask (:951-956): per-(b,ch) plane m = |init| / max|init| / 2 + 0.5; out = a*sqrt(m).
// |x| >= 0, so the float bit pattern orders like an unsigned integer: atomicMax on the bits.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sigma_max_kernel(const pdse_sigma_desc d) {
  const int pl = blockIdx.y;
  const float* x = d.init + (size_t)pl * d.plane;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.plane; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(d.maxbuf) + pl, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void sigma_apply_kernel(const pdse_sigma_desc d) {
  const int pl = blockIdx.y;
  const float mx = d.maxbuf[pl];
  const size_t base = (size_t)pl * d.plane;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.plane; i += (int64_t)gridDim.x * 256) {
#pragma clang fp contract(off)
    float m = fabsf(d.init[base + i]) / mx;
    m = m * 0.5f;
    m = m + 0.5f;
    d.out[base + i] = d.a[base + i] * sqrtf(m);
  }
}

int pdse_sigma_launch(const pdse_sigma_desc* d, hipStream_t s) {
  REQ(d && d->init && d->a && d->out && d->maxbuf, "sigma: null pointer");
  REQ(d->nplanes > 0 && d->nplanes <= 65535 && d->plane > 0, "sigma: bad sizes");
  if (pdse_check_hip(hipMemsetAsync(d->maxbuf, 0, sizeof(float) * d->nplanes, s), "sigma memset")) return 1;
  int bx = (int)((d->plane + 255) / 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(sigma_max_kernel, dim3(bx, d->nplanes), dim3(256), 0, s, *d);
  hipLaunchKernelGGL(sigma_apply_kernel, dim3(bx, d->nplanes), dim3(256), 0, s, *d);
  return pdse_check_launch("sigma");
}

// ---------------------------------------------------------------------------------------
// LayerNorm over rows of N <= 1024 (gcrn.py:31,35); one wavefront per row, values held in
// registers, wave-shuffle reductions, strided (transposing) store.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_kernel(const pdse_ln_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t rows = (int64_t)d.B * d.T;
  if (row >= rows) return;
  const int b = (int)(row / d.T), t = (int)(row - (int64_t)b * d.T);
  const float* x = d.in + row * d.N;
  float v[16];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = k * 64 + lane;
    v[k] = i < d.N ? x[i] : 0.f;
    sum += v[k];
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)d.N;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = k * 64 + lane;
    const float e = i < d.N ? v[k] - mean : 0.f;
    sq += e * e;
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)d.N + d.eps);
  const int64_t ob = (int64_t)b * d.osb + (int64_t)t * d.os_t;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = k * 64 + lane;
    if (i < d.N) {
      const float y = (v[k] - mean) * rstd * d.gamma[i] + d.beta[i];
      const int c = i / d.r;
      d.out[ob + (d.blk ? (int64_t)(c >> 3) * d.os_hi + (c & 7) : (int64_t)c * d.os_hi) + (int64_t)(i % d.r) * d.os_lo] = y;
    }
  }
}

int pdse_ln_launch(const pdse_ln_desc* d, hipStream_t s) {
  REQ(d && d->in && d->gamma && d->beta && d->out, "layernorm: null pointer");
  REQ(d->B > 0 && d->T > 0 && d->N > 0 && d->N <= 1024 && d->r > 0, "layernorm: bad sizes (N <= 1024)");
  REQ(d->blk == 0 || d->blk == 8, "layernorm: blk is 0 or 8");
  const int64_t rows = (int64_t)d->B * d->T;
  hipLaunchKernelGGL(ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *d);
  return pdse_check_launch("layernorm");
}

// ---------------------------------------------------------------------------------------
// q-sample of the training step (trainer/complex_ddpm_trainer.py:707-727, prior-grad branch):
// per-item scalars, separately rounded operations (bit-exact with the reference's tensor ops).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qsample_kernel(const pdse_qsample_desc d) {
  const int b = blockIdx.y;
  const float a = d.a[b], s = d.s[b];
  const int64_t off = (int64_t)b * d.plane;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.plane; i += (int64_t)gridDim.x * 256) {
#pragma clang fp contract(off)
    float t1, t2;
    if (d.mode == 0) {          // a * (label - init) + s * noise
      const float diff = d.label[off + i] - d.init[off + i];
      t1 = a * diff;
      t2 = s * d.noise[off + i];
    } else if (d.mode == 1) {   // a * label + s * (noise + init)
      t1 = a * d.label[off + i];
      const float ni = d.noise[off + i] + d.init[off + i];
      t2 = s * ni;
    } else {                    // a * label + s * noise
      t1 = a * d.label[off + i];
      t2 = s * d.noise[off + i];
    }
    d.out[off + i] = t1 + t2;
  }
}

int pdse_qsample_launch(const pdse_qsample_desc* d, hipStream_t s) {
  REQ(d && d->label && d->noise && d->a && d->s && d->out, "qsample: null pointer");
  REQ(d->mode >= 0 && d->mode <= 2 && (d->mode == 2 || d->init), "qsample: mode 0 / 1 need init");
  REQ(d->B > 0 && d->B <= 65535 && d->plane > 0, "qsample: bad sizes");
  int bx = (int)((d->plane + 255) / 256);
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(qsample_kernel, dim3(bx, d->B), dim3(256), 0, s, *d);
  return pdse_check_launch("qsample");
}

// ---------------------------------------------------------------------------------------
// GCRN: last decoder stage (gated ConvTranspose 32 -> 1, BN, ELU) + Linear(161,161) over the bins
// (model/gcrn.py:158-163).
//
// [r4] Until round 3 a persistent 512-thread workgroup handled ONE (b, t) row per iteration: five barriers per row, the Linear on
// the vector units (84 FMAs per thread and row from a register-resident column): 152-175 us per launch for 1.5 GFLOP and 131 MB.
// Now a workgroup owns a tile of 32 rows:
//   * the transposed convolution + gate + BatchNorm + ELU two rows per iteration (thread = (output bin, which of the two rows),
//     all 32 channels of its bin from an LDS copy of the row, ELU applied to the skip half on the way in; the next two rows' 20
//     loads per thread are in flight meanwhile; one barrier per iteration thanks to a double-buffered copy), its 161 values per
//     row written straight into the A-operand layout of the Linear;
//   * the Linear on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32): rows = M, output bins = N, six waves take one
//     32-bin tile each, K = 161 padded to 168 as 21 groups of four k-steps (one ds_read_b128 of y and one 16-byte load of the
//     packed matrix per group, packing.pack_a4); the accumulator layout puts 32 consecutive output bins of one row on the lanes
//     of each store.
// ---------------------------------------------------------------------------------------
#define GL_F 161
#define GL_FI 80
#define GL_KQ 21   // K groups of four k-steps (8 inputs each): 168 >= 161
#define GL_YB 132  // floats per (kq, h) block of ys: 32 rows x 4 + 4 of skew
__device__ __forceinline__ float gl_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__global__ __launch_bounds__(512, 4) void gcrnlast_kernel(const pdse_gcrnlast_desc d) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(16))) float xs[2][2][32 * GL_FI];                               // [buffer][row of the pair][channel][bin], ELU applied to the skip half
  // y of the 32 rows: input k = 2 (4 kq + i) + h at block (2 kq + h), [row][i]; blocks 4 floats apart from a multiple of 32 banks, so that
  // the 161 threads of a row (i, h and kq vary, the row does not) do not all write into four banks
  __shared__ __attribute__((aligned(16))) float ys[GL_KQ * 2 * GL_YB];
  __shared__ __attribute__((aligned(16))) float4 wl4[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = tid & 255, rr = tid >> 8;                               // conv: output bin, row of the pair
  const bool act = o < GL_F;
  const bool odd = (o & 1) != 0;
  // wl4[parity][c] = (main weight of x[j], of x[j-1], gate weight of x[j], of x[j-1]) of channel c for even / odd output bins
  // (even bin 2j <- tap 0 at j, tap 2 at j-1; odd bin 2j+1 <- tap 1 at j): one 16-byte LDS read per channel
  if (tid < 64) {
    const int c = tid & 31, par = tid >> 5;
    wl4[par][c] = par ? make_float4(d.w1[3 * c + 1], 0.f, d.w2[3 * c + 1], 0.f) : make_float4(d.w1[3 * c], d.w1[3 * c + 2], d.w2[3 * c], d.w2[3 * c + 2]);
  }
  // inputs 161 .. 167 of the Linear do not exist: their slots stay zero (so does the packed matrix there)
  for (int e = tid; e < GL_KQ * 2 * GL_YB; e += 512) ys[e] = 0.f;
  const int64_t plane = (int64_t)d.T * GL_FI;
  const int rows = d.B * d.T, ntiles = (rows + 31) >> 5;
  // this thread's ten (channel, bin) items of its row: element e = o + 256 i, channel e / 80 (a row beyond the last reads row 0).
  // (Measured and not kept: two pairs of rows in flight - no change, the iteration is not waiting for its loads; 16-byte loads with
  // per-thread offsets computed once - 84 instead of 75 us per launch.)
  auto fetch = [&](const int row, float (&v)[10]) {
    const int rw = row < rows ? row : 0;
    const int b = rw / d.T, t = rw - b * d.T;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int e = o + 256 * i;
      const int c = e / GL_FI, f = e - c * GL_FI;
      v[i] = c < 16 ? d.in0[((int64_t)b * 16 + c) * plane + (int64_t)t * GL_FI + f]
                    : d.in1[((int64_t)b * 16 + (c - 16)) * plane + (int64_t)t * GL_FI + f];
    }
  };
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // uniform trip count per workgroup
    const int row0 = tile * 32;
    float nxt[10];
    fetch(row0 + rr, nxt);
    __syncthreads();                                                 // the previous tile's ys / xs are free (and the staging above is done)
    for (int it = 0; it < 16; ++it) {
      float* const xb = xs[it & 1][rr];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int e = o + 256 * i;
        const float v = nxt[i];
        xb[e] = e < 16 * GL_FI ? v : (v > 0.f ? v : gl_exp(v) - 1.0f);   // ELU on the skip half
      }
      if (it + 1 < 16) fetch(row0 + 2 * (it + 1) + rr, nxt);         // the next pair of rows: in flight during this pair's arithmetic
      __syncthreads();                                               // (the other buffer was last read two iterations ago)
      if (act) {
        // ConvTranspose (1,3) stride 2: even bin 2j <- tap 0 at j, tap 2 at j-1; odd bin 2j+1 <- tap 1 at j
        const int j = o >> 1;
        const bool has0 = odd || j < GL_FI, has2 = !odd && j >= 1;     // bin 160 = 2 * 80 has no tap 0, bin 0 no tap 2
        const int j0 = has0 ? j : GL_FI - 1, j2 = has2 ? j - 1 : 0;
        float m = 0.f, g = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
          const float4 w = wl4[odd ? 1 : 0][k];
          const float x0 = has0 ? xb[k * GL_FI + j0] : 0.f, x2 = has2 ? xb[k * GL_FI + j2] : 0.f;
          m += x0 * w.x + x2 * w.y;
          g += x0 * w.z + x2 * w.w;
        }
        float y = (m + d.b1) * __builtin_amdgcn_rcpf(1.0f + gl_exp(-(g + d.b2)));
        y = y * d.bn_scale + d.bn_shift;
        y = y > 0.f ? y : gl_exp(y) - 1.0f;
        ys[(2 * (o >> 3) + (o & 1)) * GL_YB + (2 * it + rr) * 4 + ((o >> 1) & 3)] = y;   // k = o = 2 (4 kq + i) + h
      }
    }
    __syncthreads();
    // ---- Linear(161,161): wave w < 6 owns output bins 32 w .. 32 w + 31 of all 32 rows
    if (wave < 6) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float4* const B4 = reinterpret_cast<const float4*>(d.fcp) + (size_t)wave * GL_KQ * 64 + lane;   // [6 tiles][21][64 lanes][4]
      const float* const A1 = ys + (lane >> 5) * GL_YB + (lane & 31) * 4;                                   // + kq * 2 * GL_YB
#pragma unroll 7
      for (int kq = 0; kq < GL_KQ; ++kq) {
        const float4 bv = B4[(size_t)kq * 64];
        const float4 av = *reinterpret_cast<const float4*>(A1 + kq * 2 * GL_YB);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
      }
      const int ob = 32 * wave + (lane & 31);
      if (ob < GL_F) {
        const float bias = d.fcb[ob];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // accumulator row of register r
          if (row < rows) {
            const int b = row / d.T, t = row - b * d.T;
            d.out[(int64_t)b * d.out_sb + (int64_t)t * GL_F + ob] = acc[r] + bias;
          }
        }
      }
    }
  }
}

int pdse_gcrnlast_launch(const pdse_gcrnlast_desc* d, hipStream_t s) {
  REQ(d && d->in0 && d->in1 && d->w1 && d->w2 && d->fcT && d->fcp && d->fcb && d->out, "gcrn_last: null pointer");
  REQ(d->B > 0 && d->T > 0 && (int64_t)d->B * d->T < (1ll << 31), "gcrn_last: bad sizes");
  const int ntiles = (d->B * d->T + 31) / 32;
  hipLaunchKernelGGL(gcrnlast_kernel, dim3(ntiles < 512 ? ntiles : 512), dim3(512), 0, s, *d);   // two workgroups per CU
  return pdse_check_launch("gcrn_last");
}

// ---------------------------------------------------------------------------------------
// Masked complex MSE of the validation loop (utils/loss.py:34-44).  HBM-bound: two streamed reads, fixed-order
// double-precision partial sums (32 workgroups per utterance, then one workgroup over all partials).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* lds) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = ((lds[0] + lds[1]) + lds[2]) + lds[3];
  return r;   // valid on thread 0
}

__global__ __launch_bounds__(256) void maskloss_partial_kernel(const pdse_maskloss_desc d) {
  __shared__ double lds[4];
  const int b = blockIdx.y;
  const int fr = min(max(d.frames[b], 0), d.T);
  const int64_t plane = (int64_t)d.T * d.F, live = (int64_t)fr * d.F;   // the first `fr` frames of every channel plane
  double acc = 0.0;
  for (int c = 0; c < d.C; ++c) {
    const int64_t base = ((int64_t)b * d.C + c) * plane;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < live; i += (int64_t)gridDim.x * 256) {
      const float e = d.esti[base + i] - d.label[base + i];   // fp32 difference like the reference, squared in double
      acc += (double)e * (double)e;
    }
  }
  const double tot = block_sum_256(acc, lds);
  if (threadIdx.x == 0) d.partial[(int64_t)b * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void maskloss_final_kernel(const pdse_maskloss_desc d) {
  __shared__ double lds[4];
  const int n = d.B * PDSE_MASKLOSS_BLOCKS;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += d.partial[i];
  const double tot = block_sum_256(acc, lds);
  if (threadIdx.x == 0) {
    double frames = 0.0;
    for (int b = 0; b < d.B; ++b) frames += (double)min(max(d.frames[b], 0), d.T);
    d.out[0] = (float)(tot / (frames * (double)d.C * (double)d.F));
  }
}

int pdse_maskloss_launch(const pdse_maskloss_desc* d, hipStream_t s) {
  REQ(d && d->esti && d->label && d->frames && d->partial && d->out, "masked_mse: null pointer");
  REQ(d->B > 0 && d->B <= 65535 && d->C > 0 && d->T > 0 && d->F > 0, "masked_mse: bad sizes");
  hipLaunchKernelGGL(maskloss_partial_kernel, dim3(PDSE_MASKLOSS_BLOCKS, d->B), dim3(256), 0, s, *d);
  hipLaunchKernelGGL(maskloss_final_kernel, dim3(1), dim3(256), 0, s, *d);
  return pdse_check_launch("masked_mse");
}
