/* pdse.h — C-ABI of libpdse.so: the MI355X (gfx950) reverse-diffusion sampling path
 * of Prior-DiffuSE.
 *
 * The reference has NO plugin / operator / FFI layer: its boundary is
 * the Python surface
 *     ComplexDDPMTrainer.inference_schedule / generate_wav
 *                                   (trainer/complex_ddpm_trainer.py:105, :903)
 *     self.model(feat) -> X_init              (trainer/complex_ddpm_trainer.py:941)
 *     self.model_ddpm(audio, init, t) -> eps  (trainer/complex_ddpm_trainer.py:968)
 * i.e. nn.Module.__call__ on contiguous fp32 [B,2,T,161] tensors.  This header is the
 * build-defined C boundary underneath that surface: every entry point takes raw device
 * pointers (tensor.data_ptr()), explicit sizes/strides and a hipStream_t, returns an
 * int status (0 = ok) and never throws; the message of the last failure on the calling
 * thread is available from pdse_last_error().
 *
 * Ownership: the caller owns every buffer named in a descriptor (inputs, outputs,
 * packed weights, workspaces).  The library owns only plan objects.  No global mutable
 * state; direct launches run on the device current on the calling thread, plans on the device
 * they were bound to with pdse_plan_set_device (restoring the caller's current device).
 * Threading: one host thread per plan; different plans are independent.
 *
 * Data layout: activations are "channel-major" [B, C, T, F] fp32 with F innermost (the
 * reference's own NCHW layout), addressed through explicit element strides so that the
 * same kernels serve Conv2d / ConvTranspose2d / Conv1d / Linear / STFT bases.
 */
#ifndef PDSE_H
#define PDSE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDSE_ABI_VERSION 8

typedef void* pdse_stream_t; /* hipStream_t */

enum pdse_act { PDSE_ACT_NONE = 0, PDSE_ACT_PRELU = 1, PDSE_ACT_ELU = 2, PDSE_ACT_SIGMOID = 3 };

/* epilogue of the gather-GEMM convolution kernel */
enum pdse_epi {
  PDSE_EPI_LINEAR = 0, /* y = act(affine(acc0 + bias0)) (+ resid)                              */
  PDSE_EPI_GLU = 1,    /* y = act(affine((acc0+bias0) * sigmoid(acc1+bias1)))  gcrn.py:43-84   */
  PDSE_EPI_BIGLU = 2   /* L,R = acc0,acc1 (+bias); mL = s(Wlc L), mR = s(Wrc R);
                          y = act(affine(Wc2 (L*mR + R*mL) + bc2))           diff3.py:316-326 */
};

/* One input source of a (possibly channel-concatenated) convolution input. */
typedef struct pdse_src {
  const float* ptr;
  int64_t sb, sc, st, sf; /* element strides of batch, channel, frame (t), bin (f) */
  int32_t C;              /* channels taken from this source                       */
  int32_t act;            /* pdse_act applied on load (ELU on GCRN skip tensors)    */
  /* blk 8 (korder 3 only, ABI 5): the tensor is stored in blocks of 8 channels, [B][C/8][T][F][8] - channel c lives at
     (c >> 3)*sc + (c & 7), sc / st / sf being the element strides of a block, a frame and a bin (multiples of 4), so a
     lane's 8 channels of one K block are 32 contiguous bytes (two 16-byte loads instead of eight 4-byte gathers).  It is
     what a launch with out_cr = 8, out_sc_lo = 1 writes.  0: plain strides, channel c at c*sc. */
  int32_t blk;
  int32_t pad_;
} pdse_src;

/* Gather-GEMM convolution:
 *   acc[co](b,t,j) = sum_tap sum_ci W[tap][ci][co] * IN(b, ci, t + dt[tap], j*sf_in + df[tap])
 * IN is zero outside [0,Tin)x[0,Fin), except that frame -1 reads padrow[b][ci] when padrow
 * is set (the ε-net encoder adds the time bias to the already padded tensor, diff3.py:146-147).
 * Replaces, per layer: nn.Conv2d / nn.ConvTranspose2d (one launch per output phase) /
 * nn.Conv1d / nn.Linear calls of model/diff3.py, model/gcrn.py, and the torch.stft /
 * torch.istft DFTs of trainer/complex_ddpm_trainer.py:926-930, :1010-1015.
 * Weights are pre-packed on the host into MFMA A-fragment order:
 *   w[mtile][kstep][lane] = W[k = 2*kstep + (lane>>5)][co = 32*mtile + (lane&31)],
 *   k = tap*Cin + ci  (korder 0)   or   kstep = (ci/2)*ntaps + tap  (korder 1).
 */
typedef struct pdse_gconv_desc {
  pdse_src in0, in1; /* in1.C == 0: single source */
  int32_t Tin, Fin;
  const float* padrow; /* [B][Cin] value of frame -1, or NULL */
  int64_t padrow_sb;
  const int32_t* taps; /* device, [ntaps][2] = {dt, df} */
  int32_t ntaps;
  int32_t sf_in; /* input bin = j*sf_in + df */
  /* optional transform on load: v = prelu(v, slope) * scale[ci] + shift[ci]   (diff3.py:221-246,
     PReLU -> BatchNorm1d in front of a zero-padded Conv1d).  xf_mode 0: none, 1: set 0 feeds
     both accumulators, 2: set 0 feeds acc0 and set 1 feeds acc1. */
  const float* xf_scale0;
  const float* xf_shift0;
  const float* xf_scale1;
  const float* xf_shift1;
  float xf_slope0, xf_slope1;
  int32_t xf_mode;
  int32_t cin1; /* 1: Cin == 1, k enumerates taps (STFT / Linear-over-bins) */
  const float* w0;
  const float* w1;
  int32_t ksteps; /* ceil(K/2) */
  int32_t Cout;
  const float* bias0;
  const float* bias1;
  int64_t bias0_sb, bias1_sb; /* batch stride of the bias (0: shared) */
  int32_t epi;                /* pdse_epi */
  int32_t act;                /* pdse_act */
  float act_slope;
  int32_t C2; /* BIGLU: channels after the closing 1x1 (64 or 1) */
  const float* post_scale; /* folded eval-mode BatchNorm: y*scale[co] + shift[co], or NULL */
  const float* post_shift;
  const float* wlc; /* BIGLU chain fragments, [16][64] each; wc2 [C2 tiles][16][64] or [32] when C2==1 */
  const float* wrc;
  const float* blc;
  const float* brc;
  const float* wc2;
  const float* bc2;
  const float* resid; /* added after the activation, addressed like out, or NULL */
  float* out;
  int64_t out_sb, out_sc_hi, out_sc_lo, out_st, out_sf, out_off;
  int32_t out_cr; /* co -> (co / out_cr)*out_sc_hi + (co % out_cr)*out_sc_lo */
  int32_t B, Tout, Fout;
  /* K order of the packed weights: 0: k = tap*Cin + ci (generic kernel); 1: k-step =
     (ci/2)*ntaps + tap, k = 2*kstep + (ci&1) (pipelined kernel: taps innermost, unrolled);
     2: split-bf16 BIGLU blocks (csrc/gconv3.hip; Cin = 32, Cout = 32, C2 in {64, 1}): every weight is stored as its
     exact three-way bf16 split (packing.split_bf16x3) in v_mfma_f32_32x32x16_bf16 A-fragment order -
       w0..w3  [ntaps*2 blocks][3 planes][64 lanes][8 bf16]: block tap*2 + q, lane (row, h), element j =
               W[k = tap*32 + 16q + 8h + j][row]
       wlc, wrc [2][3][64][8]; wc2 [2 tiles][2][3][64][8] (C2 == 64; C2 == 1: the fp32 vector [32] as before);
       nx_w [nx_n][4][3][64][8]: chained tiles, k order of an accumulator tile used as B operand
               (packing.rho_bf16).  Activations, biases, BatchNorm and every output stay fp32.
     3: split-bf16 GEMM-shaped convolutions (csrc/gconv4.hip; LINEAR / GLU, channel counts in multiples of 16, no load
     transform): w0 (w1) [K blocks][ceil(Cout/32) tiles][3 planes][64 lanes][8 bf16], K blocks of 16 channels in the
     order (source, tap, channel block); lane (row, h), element j = W[tap*Cin + cbase + 16cb + 8h + j][32 tile + row].
     4 (ABI 6): the same kernel on ONE plane - plain bf16 operands (weights and gathered activations rounded to nearest even),
     one product, fp32 accumulation: w0 (w1) [K blocks][tiles][1][64][8].  The opt-in bf16 mode; its own tolerance.
     5 (ABI 8): the same kernel on TWO fp16 planes (f16x2, see PDSE_F16_ACT_EXP): w0 (w1) [K blocks][tiles][2][64][8] hold hi / lo
     of W * 2^wexp, the gathered activations are scaled by 2^PDSE_F16_ACT_EXP and split in registers, three f16 products per
     multiply-add, fp32 accumulation; the accumulators are scaled back before the epilogue.  fp32-equivalent (same tolerances
     as korder 3). */
  int32_t korder;
  /* Dual-phase stride-(1,2) ConvTranspose2d (BIGLU only, korder 1): one launch computes the even
     output bins f_o = 2j (weights w0/w1 over all ntaps taps) AND the odd bins 2j+1 (weights w2/w3
     over the taps selected by p1mask, ksteps1 k-steps), stores them side by side (out_sf = 2 bin
     strides).  w2 == NULL: single phase.  Fout1 = number of valid odd bins (j < Fout1). */
  int32_t ksteps1;
  const float* w2;
  const float* w3;
  int32_t p1mask;
  int32_t Fout1;
  /* Chained 1x1 convolutions (BIGLU with C2 == 64, korder 1 only): the block output y (64 channels, after the
     folded BatchNorm and the activation) feeds, register to register, nx_n <= 3 further 1x1 convolutions of 32
     output channels each - the NEXT stage's conv1 (model/diff3.py:146-149, :343-345) and, in the encoder, the
     skip-connection halves of the two decoders' conv1 (the 128-channel concat of :343 split by linearity):
         z_i = Wn_i y + nx_bias[i][b] (+ nx_add[i] at the same address),  stored to nx_out[i].
     nx_w: [nx_n][2][16][64] chain-packed like wc2.  Element address of channel c, frame t, bin q:
         nx_out[i] + b*nx_sb[i] + c*nx_sc[i] + t*nx_st[i] + q*nx_sf[i] + nx_off[i]
     (dual-phase launches: q = 2j, 2j+1; nx_n == 1 there).  nx_row0 >= 0: that tile also gets its bias written to
     "frame -1" (address without nx_off) by the lanes of frame 0 - the encoder's explicit pad frame.
     nx_keep == 0: y itself is not stored (nobody else reads it). */
  const float* nx_w;
  const float* nx_bias[3];
  const float* nx_add[3];
  float* nx_out[3];
  int64_t nx_bias_sb[3];
  int64_t nx_sb[3], nx_sc[3], nx_st[3], nx_sf[3], nx_off[3];
  int32_t nx_n, nx_keep, nx_row0;
  int32_t wexp;   /* korder 5: power-of-two exponent the host scaled w0 / w1 by (packing.f16_wexp); otherwise unused */
  /* BIGLU, chained form only: biases of the two gather convolutions for OUTPUT FRAME 0 (batch stride bias0_sb /
     bias1_sb like bias0 / bias1); NULL: frame 0 uses bias0 / bias1 like every other frame.  Needed when conv1 is
     composed into the gather weights (encoder stage 1): frame 0 sees the zero pad frame, whose conv1 value is the
     pad bias, through the kt = 0 taps. */
  const float* bias0_t0;
  const float* bias1_t0;
  /* host copy of the first 12 entries of `taps` (korder 1 / 2 kernels, ntaps <= 10): read from the kernel arguments,
     i.e. from scalar registers - the device table costs every workgroup one dependent L2 round trip per tap */
  int32_t tap_dt[12], tap_df[12];
} pdse_gconv_desc;

/* Diffusion-step embedding + every per-stage time bias folded through the following 1x1
 * convolution (model/diff3.py:62-95, :146-147, :343-344):
 *   x = lerp(table, t); temb = silu(P2 silu(P1 x)); out[b][:] = WF temb + bf,
 * WF = stacked (W1_k · Wtp_k), bf = stacked (b1_k + W1_k · btp_k) for the 15 stages. */
typedef struct pdse_time_desc {
  const float* t; /* [B] fractional (or integral) step */
  const float* table; /* [max_steps][128] */
  const float* p1T;   /* [128][512] */
  const float* b1;
  const float* p2T; /* [512][512] */
  const float* b2;
  const float* wfT; /* [512][NF] */
  const float* bf;
  float* out;  /* [B][NF] */
  float* temb; /* [B][512] or NULL */
  int32_t B, NF, max_steps, pad_;
} pdse_time_desc;

enum pdse_ew_op {
  PDSE_EW_DIV = 0,          /* out = a / s0                       :942 init /= c            */
  PDSE_EW_UPDATE = 1,       /* out = s0 * (a - s1 * b)            :977                      */
  PDSE_EW_UPDATE_FINAL = 2, /* out = ((s0*(a - s1*b)) + c) * s2   :977 + :995-996           */
  PDSE_EW_COPY = 3,
  PDSE_EW_ADD_MUL = 4 /* out = (a + b) * s0 */
};
typedef struct pdse_ew_desc {
  const float* a;
  const float* b;
  const float* c;
  float* out;
  int64_t n;
  float s0, s1, s2;
  int32_t op;
} pdse_ew_desc;

/* sqrt-compression / square-decompression of a [B,2,T,F] spectrogram
 * (trainer/complex_ddpm_trainer.py:931-937 and :1004-1008):  power = 0.5 or 2. */
typedef struct pdse_compand_desc {
  const float* in;
  float* out;
  int64_t plane; /* T*F */
  int32_t B;
  int32_t mode; /* 0: mag**0.5 (compress), 1: mag**2 (decompress) */
  /* F > 0: the output is written as out[b*out_sb + ri*out_sc + t*out_st + f] instead of the input's own layout (the
   * ISTFT GEMM reads frame-major rows [B][T][2*168] whose 336 entries are its K dimension; the pad entries are never
   * written).  F == 0: out has the layout of in. */
  int64_t out_sb, out_sc, out_st;
  int32_t F, pad_;
} pdse_compand_desc;

/* waveform front-end: c[b] = sqrt(sum x^2 / len_b); xpad = reflect_pad(x / c, 160)   (:922-923, stft center=True).
 * lens (optional, device int32 [B]): true lengths of zero-padded utterances — the batched twin normalises each
 * utterance over its own samples before padding (utils/dataset.py:45-58); NULL: len_b = L. */
typedef struct pdse_wavprep_desc {
  const float* wav; /* [B][L] */
  float* xpad;      /* [B][L + 2*pad] */
  float* c;         /* [B] */
  const int32_t* lens;
  int32_t B, L, pad, normalize; /* normalize 0: c = 1 */
} pdse_wavprep_desc;

/* overlap-add of windowed inverse-DFT frames, window-envelope normalisation, trim, rescale
 * (torch.istft semantics, :1010-1016): frames [B][n_fft][T] (T innermost). */
typedef struct pdse_ola_desc {
  const float* frames;
  const float* win2; /* [n_fft] squared window */
  const float* c;    /* [B] or NULL */
  float* out;        /* [B][L] */
  int32_t B, T, L, n_fft, hop, pad_;
} pdse_ola_desc;

/* forward noising of the training step (trainer/complex_ddpm_trainer.py:707-729), a = sqrt(alpha_bar_t),
 * s = sqrt(1 - alpha_bar_t) per batch item; every product/sum rounded separately like the reference's tensor ops:
 *   mode 0 (pirorgrad, :718):  out[b] = a[b] * (label[b] - init[b]) + s[b] * noise[b]
 *   mode 1 (deltamu,   :721):  out[b] = a[b] * label[b] + s[b] * (noise[b] + init[b])
 *   mode 2 (neither,   :724):  out[b] = a[b] * label[b] + s[b] * noise[b]            (init unused, may be NULL)
 * (--sigma, :709-715: the caller passes noise already multiplied by sqrt(mask), pdse_sigma_mask_f32). */
typedef struct pdse_qsample_desc {
  const float* label;
  const float* init;
  const float* noise;
  const float* a; /* [B] */
  const float* s; /* [B] */
  float* out;
  int64_t plane; /* elements per batch item */
  int32_t B, mode;
} pdse_qsample_desc;

/* Masked complex MSE of the validation loop (utils/loss.py:34-44, used at trainer/complex_ddpm_trainer.py:490):
 *   loss = sum_{b, c, t < frames[b], f} (esti - label)^2 / (C * F * sum_b frames[b])
 * over [B][C][T][F] tensors; frames beyond an utterance's own (the zero padding of the batch) do not count.
 * Two launches with a fixed summation order (per-workgroup partial sums in double, then one workgroup): the value
 * does not depend on scheduling.  partial: scratch of B * PDSE_MASKLOSS_BLOCKS doubles. */
#define PDSE_MASKLOSS_BLOCKS 32
typedef struct pdse_maskloss_desc {
  const float* esti;
  const float* label;
  const int32_t* frames; /* device, [B] */
  double* partial;
  float* out; /* [1] */
  int32_t B, C, T, F;
} pdse_maskloss_desc;

/* per-(b,ch) abs-max mask of --sigma (:951-956): out = a * sqrt(|init|/max|init| / 2 + 0.5) */
typedef struct pdse_sigma_desc {
  const float* init;
  const float* a;
  float* out;
  float* maxbuf; /* [B*2] scratch */
  int64_t plane;
  int32_t nplanes, pad_;
} pdse_sigma_desc;

/* LayerNorm over the last dim of [B][T][N] rows with a strided/transposed store
 * (gcrn.py:31, :35): out[b*osb + (j / r)*os_hi + (j % r)*os_lo + t*os_t];
 * blk 8 (ABI 5): c = j / r is a channel of a tensor kept in blocks of 8 channels (pdse_src.blk): it is stored at
 * (c >> 3)*os_hi + (c & 7) instead of c*os_hi. */
typedef struct pdse_ln_desc {
  const float* in;
  const float* gamma;
  const float* beta;
  float* out;
  int64_t osb, os_hi, os_lo, os_t;
  int32_t B, T, N, r;
  float eps;
  int32_t blk;
} pdse_ln_desc;

/* One LSTM layer of the grouped LSTM (gcrn.py:6-40), all T steps, G independent groups.
 *   gx   [G][T][4H][Bp]   input projections + both biases (Bp = batch padded to 32)
 *   whh  [G][H/8][H/2][64] recurrent weights in MFMA A-fragment order (8 hidden units x 4 gates per slice)
 *   hT   [2][G][H][Bp], cst [G][H][Bp]   state, zeroed by the call
 *   y    [B][T][ysz]; y[b][t][u*y_su + g*y_sg] = h_t                   */
typedef struct pdse_lstm_desc {
  const float* gx;
  const float* whh;
  float* hT;
  float* cst;
  float* y;
  int64_t y_sb, y_st, y_su, y_sg;
  int32_t B, Bp, T, H, G, pad_;
} pdse_lstm_desc;

/* Both layers of the grouped LSTM and the LayerNorm between them (gcrn.py:22-35) as a layer wavefront: T + 2
 * launches, each carrying layer 1 at frame s, the layer-2 input projection (LayerNorm 1 folded) at frame s-1 and
 * layer 2 at frame s-2 (csrc/lstm.hip).  Replaces two pdse_lstm_f32 calls, the LayerNorm(1024) between them and
 * the two layer-2 input-projection GEMMs.
 *   gx1   [G][T][Bp][4H]        layer-1 input projections + both biases, gate rows innermost: the projection GEMM's
 *                                lanes are frames and its registers gate rows, so it stores 16 bytes per lane and
 *                                instruction (batch innermost, as pdse_lstm_desc.gx: every 4-byte store its own line)
 *   whh1, whh2 [G][H/8][H/8][64][4]   recurrent weights: slice s = gate rows q*H + 8s + u (tile row q*8 + u), K in natural
 *                                order, four k-steps per 16-byte lane entry (packing.pack_a4)
 *   wih2  same shape             W_ih of layer 2 times diag(gamma_ln1); K order (g', kq, i, hh): k-group gq = 32 g' + kq'
 *                                holds units u = 8 (32 g + kq') + 2 i + hh of source group g', i.e. feature 2 u + g'
 *   r2, c2 [G][4H]               row sums of the folded weights; W_ih beta_ln1 + b_ih + b_hh
 *   hT1, hT2 [2][G][H/8][2][Bp][4]   states, ping-pong by frame parity, unit 8 kq + 2 i + hh at [kq][hh][b][i]
 *   cst1, cst2 [G][H][Bp]; gx2 [2][G][4H][Bp]; part [2][G][H/8][Bp][2] (LayerNorm partial sums); all scratch
 *   y     layer-2 output, y[b*y_sb + t*y_st + u*y_su + g*y_sg] */
typedef struct pdse_glstm_desc {
  const float* gx1;
  const float* whh1;
  const float* wih2;
  const float* r2;
  const float* c2;
  const float* whh2;
  float* hT1;
  float* cst1;
  float* hT2;
  float* cst2;
  float* gx2;
  float* part;
  float* y;
  int64_t y_sb, y_st, y_su, y_sg;
  int32_t B, Bp, T, H, G;
  float eps;
  /* ABI 8: slices of 8 hidden units per workgroup - 0 / 1: one (3 x 128 workgroups per step at B <= 32; what plans that share the
     GPU with other batches take), 2: two slices from ONE fetch of the group's state (3 x 64 workgroups, a CU each: 10.8 instead of
     11.7 us per step when the plan owns the GPU; measured slower by 1 % with three batches in flight).  Same summation orders:
     bit-identical results. */
  int32_t slices;
  int32_t pad_;
} pdse_glstm_desc;

/* The same grouped LSTM (both layers + LayerNorm 1, gcrn.py:22-35) as ONE persistent launch for small batches
 * (1 <= B <= 8; ABI 6, csrc/lstmp.hip): 256 co-resident workgroups keep their gate rows in registers for all frames and
 * exchange the state of a frame through 8-byte {tag, value} granules.  The caller must own the device while it runs
 * (one batch in flight): workgroups wait for each other.  Every wait is bounded; a launch that could not complete
 * leaves the step it gave up at (+1) in *status, which the caller zeroes once and reads after synchronising.
 *   gx1   [G][T][Bp][4H]     as pdse_glstm_desc.gx1
 *   w1    [G][H][4][H]       W_hh of layer 1: w1[g][u][q][k] = weight_hh_l0[q*H + u][k] (gate order i,f,g,o)
 *   w2i   [G][H][4][H]       W_ih of layer 2 times diag(gamma_ln1), chunk g of the interleaved layer-1 output: k = feature
 *                            2 u' + g' - H g of the LayerNorm input (natural column order of weight_ih_l0)
 *   w2h   [G][H][4][H]       W_hh of layer 2
 *   r2, c2 [G][4H]           as pdse_glstm_desc
 *   gran  [4][2H][Bq] 8-byte granules (Bq = B rounded up to 1, 2, 4, 8), 16-byte aligned; zeroed by every launch
 *   y     layer-2 output, y[b*y_sb + t*y_st + u*y_su + g*y_sg] */
typedef struct pdse_glstmp_desc {
  const float* gx1;
  const float* w1;
  const float* w2i;
  const float* w2h;
  const float* r2;
  const float* c2;
  unsigned long long* gran;
  int32_t* status;
  float* y;
  int64_t y_sb, y_st, y_su, y_sg;
  int32_t B, Bp, T, H, G;
  float eps;
} pdse_glstmp_desc;

/* ---- DB-AIAT prior (model/dbaiat.py), channel-major [B,C,T,F] tensors ------------------- */

/* LayerNorm over the bins of every (b,c,t) row + per-channel PReLU
 * (dbaiat.py:498, :627-628, :545): out = prelu_c(ln_F(in)).  out rows may sit inside a larger
 * channel-concatenated buffer (out_sb); in is dense [B,C,T,F]. */
typedef struct pdse_rowln_desc {
  const float* in;
  const float* gamma; /* [F] */
  const float* beta;
  const float* slope; /* [C] */
  float* out;
  int64_t out_sb; /* batch stride of out (channel stride is T*F on both sides) */
  int32_t B, C, T, F;
  float eps;
  int32_t pad_;
} pdse_rowln_desc;

/* One layer of the dilated dense block (dbaiat.py:605-631) on the channel-blocked concatenation buffer (ABI 7, csrc/dense.hip):
 *   D[:, g_out .. g_out+8) = prelu(ln_F(conv_{(2,3), dilation (dil,1)}(pad(D[:, g_in .. g_in + cin/8))) + bias))
 * D: [B][G][T + tpad][F + 2][8] fp32 - entry (b, g, t, f) holds channels 8g .. 8g+7 of bin f of frame t at
 * (((b G + g)(T + tpad) + tpad + t)(F + 2) + f + 1) 8; the tpad leading rows and the two outer bins of every row are zero and stay
 * zero (the convolution's causal / same padding).  w: bf16 fragments of the 64 x 6 cin weights, K steps in the order (16-channel
 * block, time tap, bin tap): [cin/16][2][3][2 channel tiles][np planes][64 lanes][8] (packing.pack_dense); np = 3: exact
 * three-way split of both operands (six products, fp32-equivalent), np = 2: f16x2 (PDSE_F16_ACT_EXP; three f16 products,
 * fp32-equivalent), np = 1: plain bf16 (the opt-in bf16 mode).
 * Replaces pdse_gconv_f32 + pdse_rowln_prelu_f32 of one layer; input and output groups must not overlap. */
typedef struct pdse_dense_desc {
  float* D;
  const void* w;
  const float* bias;  /* [64] */
  const float* gamma; /* [F] */
  const float* beta;  /* [F] */
  const float* slope; /* [64] */
  int32_t B, T, F, G, tpad;
  int32_t g_in, cin; /* first input group, input channels (a multiple of 16) */
  int32_t g_out;
  int32_t dil, np;
  float eps;
  int32_t wexp;       /* np == 2 (ABI 8, f16x2): w holds fp16 hi / lo planes of W * 2^wexp (packing.f16_wexp); otherwise unused */
} pdse_dense_desc;

/* Row LayerNorm + PReLU like pdse_rowln_desc - or, with gamma == NULL, a plain re-layout - from a channel-major tensor (rows of F
 * contiguous floats at in + b in_sb + c in_sc + t in_st) into channel-blocked entries of 8 channels:
 * out + b out_sb + (c / 8) out_sg + t out_st + 8 f + c % 8 (the layout of pdse_dense_desc.D; `out` points at frame 0, bin 0). */
typedef struct pdse_rowlnb_desc {
  const float* in;
  const float* gamma; /* [F] or NULL */
  const float* beta;
  const float* slope; /* [C] */
  float* out;
  int64_t in_sb, in_sc, in_st;
  int64_t out_sb, out_sg, out_st;
  int32_t B, C, T, F;
  float eps;
  int32_t pad_;
} pdse_rowlnb_desc;

/* LayerNorm over the C channels at every (b,t,f) (nn.LayerNorm(d_model) on [S,N,32], dbaiat.py:76,81,87) */
typedef struct pdse_chln_desc {
  const float* in;
  const float* gamma; /* [C] */
  const float* beta;
  float* out;
  int64_t plane; /* T*F */
  int32_t B, C;
  float eps;
  int32_t pad_;
} pdse_chln_desc;

/* Multi-head self-attention core (nn.MultiheadAttention, dbaiat.py:77-79) on projected
 * qkv [B,3*E,T,F] (q already scaled by head_dim^-0.5): softmax(q k^T) v per (b, line, head).
 * axis 0: sequence runs over the bins F (one line per frame); axis 1: over the frames T. */
typedef struct pdse_attn_desc {
  const float* qkv;
  float* out; /* [B,E,T,F] */
  int32_t B, T, F, E, heads, axis, pad0_, pad1_;
} pdse_attn_desc;

/* Bidirectional single-layer GRU (dbaiat.py:45,83) along one axis, hidden H = 64 or 128, persistent per
 * line: gx [B,2*3*H,T,F] = W_ih x + b_ih of both directions ([fw r,z,n | bw r,z,n]);
 * whh [2][3H/32][H/2][64] MFMA A fragments, bhh [2][3H]; y [B,2H,T,F] = [fw | bw].
 * Fused input projection (H == 64 only): x != NULL replaces gx by the layer input x [B,H/2,T,F] itself;
 * wih [2][3H/32][H/4][64] A fragments of W_ih, bih [2][3H] - the projection then runs inside the recurrence
 * kernel and the 12x wider gx tensor never exists. */
typedef struct pdse_gru_desc {
  const float* gx;
  const float* whh;
  const float* bhh;
  float* y;
  int32_t B, T, F, H, axis;
  /* split 1 (ABI 5; fused H == 64 form only, csrc/gru3.hip): whh / wih hold the exact three-way bf16 splits of the
     weights as v_mfma_f32_32x32x16_bf16 A fragments (packing.pack_s3_gather per 32-row gate tile),
       whh [2 dirs][6 tiles][4 K blocks][3 planes][64 lanes][8 bf16],  wih [2][6][2][3][64][8];
     six bf16 products per multiply-add, fp32 accumulation - fp32-level accuracy at a third of the matrix-pipe time. */
  int32_t split;
  const float* x;
  const float* wih;
  const float* bih;
} pdse_gru_desc;

/* Swap the two inner axes of [N][R][Cc] -> [N][Cc][R] (fp32) through 32x32 LDS tiles: converts between the
 * model's [B,C,T,F] layout and the frames-innermost [B,C,F,T] layout the row GRU / column attention run on. */
typedef struct pdse_transpose_desc {
  const float* in;
  float* out;
  int32_t N, R, Cc, pad_;
} pdse_transpose_desc;

/* Magnitude front end and complex-ratio-mask back end of the dual-branch DB-AIAT prior (model/dbaiat.py:389-411).
 * mode 0: out[b][q] = sqrt(re^2 + im^2)                                  (torch.norm(x, dim=1), :389)
 * mode 1: mask = sigmoid(a3 * (sigmoid(a1*o + b1) * tanh(a2*o + b2)) + b3)  (dense_decoder_masking :579-582)
 *         out[b][0|1][q] = mask * |x| * cos|sin(atan2(im, re)) + ri[b][0|1][q]   (:407-409; cos/sin taken as re/|x|,
 *         im/|x|, and 1, 0 where |x| = 0 as atan2(0,0) = 0) */
typedef struct pdse_crm_desc {
  const float* x;    /* [B][2][plane] */
  const float* o;    /* [B][plane]      mode 1 */
  const float* ri;   /* [B][2][plane]   mode 1 */
  float* out;
  float a1, b1, a2, b2, a3, b3;
  int32_t plane, B, mode, pad_;
} pdse_crm_desc;

/* Last decoder stage of the GCRN prior + its Linear over the bins (model/gcrn.py:158-163), one decoder per launch:
 *   u = cat(in0, ELU(in1))  [32 ch, 80 bins]          (in0 = previous stage, already BN + ELU; in1 = encoder-1 skip)
 *   g[f] = (convT1(u)[f] + b1) * sigmoid(convT2(u)[f] + b2),  ConvTranspose2d(32 -> 1, (1,3), stride (1,2)): 161 bins
 *   y = ELU(bn_scale * g + bn_shift);   out[b, t, :] = fcT^T y + fcb        (Linear(161, 161))
 * A workgroup owns 32 (b, t) rows: the gated transposed convolution on the vector units (the single output channel wastes 31/32
 * of an MFMA tile), the Linear on the fp32 matrix cores with the rows as M (ABI 6; csrc/misc.hip). */
typedef struct pdse_gcrnlast_desc {
  const float* in0;   /* [B][16][T][80] */
  const float* in1;   /* [B][16][T][80] */
  const float* w1;    /* [32][3] main branch, ConvTranspose weight[c][0][0][k] */
  const float* w2;    /* [32][3] gate branch */
  const float* fcT;   /* [161][161] = fc.weight^T (kept for readers of the descriptor; the kernel reads fcp) */
  const float* fcb;   /* [161] */
  float* out;         /* element (b, t, o) at out + b*out_sb + t*161 + o */
  int64_t out_sb;
  float b1, b2, bn_scale, bn_shift;
  int32_t B, T;
  const float* fcp;   /* ABI 6: fc.weight^T as the MFMA B operand, [6 bin tiles][21 k groups][64 lanes][4]: entry i of lane (col, h)
                         of group g = fcT[2 (4 g + i) + h][32 tile + col], zero beyond 161 (packing.pack_a4) */
} pdse_gcrnlast_desc;

/* One dilated residual block of the eps-net's TCMs (model/diff3.py:215-257) over [B][256][T], fused with the
 * NEXT block's 1x1 input convolution (csrc/tcm.hip):
 *   g = main(BN(PReLU(h))) * sigmoid(mask(BN(PReLU(h))));  x_out = conv2(BN(PReLU(g))) + x;  h_out = conv1_next(x_out)
 * h = conv1(x) comes from the previous launch (a 1x1 gconv for the first block).  h_out NULL: last block.
 * Packed operands (prior-diffuse_amd/packing.py: pack_tcm_*):
 *   wbr [2 mi][2 kh][20 g][2 main|mask][64 lanes][4]   K row 2*ks+h = tap*64 + channel, ks = 80*kh + 4*g + e
 *   wc2 [8 mt][8 g][64 lanes][4]                        K row 2*(4*g+e)+h = gate channel
 *   wn1 [4 w][2 q][2 mo][4 g][64 lanes][4]              K row = x channel 64*w + 32*q + rho(4*g+e, h)
 *   xf  [2 main|mask][64][2 scale,shift]; xf2 [64][2]   (BatchNorm1d folded; PReLU slopes are scalars) */
typedef struct pdse_tcm_desc {
  const float* x;
  const float* h;
  float* x_out;      /* may alias x */
  float* h_out;      /* must not alias h */
  const float* wbr;
  const float* bmain;
  const float* bmask;
  const float* xf;
  const float* wc2;
  const float* bc2;
  const float* xf2;
  const float* wn1;
  const float* bn1;
  float slope_main, slope_mask, slope2;
  int32_t dil, B, T;
} pdse_tcm_desc;

/* The same residual block in split-bf16 arithmetic (csrc/tcm2.hip; exact three-way bf16 splits, six products, fp32
 * accumulation).  The bottleneck tensor travels between launches transformed and split:
 *   hs [B][2 main|mask][4 kb][2 kg][3 planes][T + 128][8] bf16 = the three bf16 planes of BN(PReLU(conv1(x))) for
 *      each branch, channel 16*kb + 8*kg + j, frame t at index t + 64; the 64 frames on either side must stay zero
 *      (the padding of the dilated convolution, dilation <= 32);
 *   mode 0: x_out = conv2(BN(PReLU(gate(hs)))) + x;  hs_out = split(next block's transforms of conv1_next(x_out));
 *           hs_out NULL: last block;     mode 1: hs_out from conv1_next(x) alone (the first block's conv1).
 * Packed operands (prior-diffuse_amd/packing.py):
 *   wbr [2 main|mask][2 mi][20 K blocks = tap*4 + kb][3][64 lanes][8]   pack_s3_gather per (branch, output tile)
 *   wc2 [8 mt][4 kb][3][64][8]                                          pack_s3_gather per output tile
 *   wn1 [2 mo][16 blocks][3][64][8]                                     pack_s3_chain (K order rho_bf16)
 *   par [64][4 main bias, mask bias, BN scale, BN shift of the gate] | [256] conv2 bias | [64] next conv1 bias |
 *       [64][4 next block's main scale, shift, mask scale, shift]       (832 floats, all present in either mode) */
typedef struct pdse_tcm2_desc {
  const float* x;
  float* x_out;          /* may alias x */
  const uint16_t* hs;
  uint16_t* hs_out;      /* must not alias hs */
  const uint16_t* wbr;
  const uint16_t* wc2;
  const uint16_t* wn1;
  const float* par;
  float slope2, slope_main_next, slope_mask_next;
  int32_t dil, B, T, mode;
  int32_t np;            /* planes of hs and of every packed weight: 3 (exact three-way split; 0 means 3), 1 (plain bf16,
                            round to nearest even: the opt-in bf16 mode) or 2 (ABI 8, f16x2: fp16 hi + lo of the operand scaled by
                            a power of two - hs by 2^PDSE_F16_ACT_EXP, the weights by 2^qexp) - hs is [B][2][4][2][np][T + 128][8] */
  int32_t qexp[3];       /* np == 2: exponents of wbr (both branches), wc2, wn1 (packing.f16_wexp); par stays in true scale */
} pdse_tcm2_desc;

/* The residual blocks of the TCM stack as ONE launch (ABI 6, csrc/tcm2.hip: tcm2s_kernel): blk[0..n-1] are the mode-0
 * descriptors in order, each reading the hs its predecessor wrote (two hs buffers alternate; blk[0].hs comes from the mode-1
 * launch in front).  Every
 * workgroup (utterance, 32-frame tile) walks the blocks itself and waits for the tiles within +-2 of its utterance through the
 * progress counters in flags; results are bit-identical to n launches of pdse_tcm2_bf16x3.
 *   flags  [B][ceil(T / 32)] int32, zeroed by every launch
 *   status [1] int32: 0, or the block (+1) at which a workgroup gave up waiting (bounded waits; the caller zeroes it once) */
#define PDSE_TCM2S_MAX 20
typedef struct pdse_tcm2s_desc {
  pdse_tcm2_desc blk[PDSE_TCM2S_MAX];
  int32_t* flags;
  int32_t* status;
  int32_t n;
  int32_t pad_;
} pdse_tcm2s_desc;

/* GroupNorm(1,C) statistics + the AIA layer update (dbaiat.py:142,147-148):
 *   out = base + k1 * gn(row) + k2 * gn(col);  stats scratch [B][4] (sum,sumsq of row | col). */
typedef struct pdse_gncomb_desc {
  const float* base;
  const float* row;
  const float* col;
  const float* g_row; /* [C] */
  const float* b_row;
  const float* g_col;
  const float* b_col;
  float* stats; /* [B][4] scratch */
  float* out;
  int64_t plane; /* T*F */
  int32_t B, C;
  float k1, k2, eps;
  int32_t pad_;
} pdse_gncomb_desc;

/* AHAM merge of the 4 layer outputs (dbaiat.py:266-288): w = softmax_i(conv1(avgpool(x_i)));
 * out = x_3 + sum_i w_i x_i.  means scratch [4][B][C]. */
typedef struct pdse_aham_desc {
  const float* x[4];
  const float* w; /* conv1 weight [C] */
  float* means;
  float* out;
  int64_t plane;
  int32_t B, C;
  float bias;
  int32_t pad_;
} pdse_aham_desc;


/* ---------------------------------------------------------------------------------------------------------------
 * BiConvGLU / BiConvTransGLU block of the eps-net on PLANE tensors (csrc/bglu.hip; round 3).
 * Replaces, per launch, one encoder / decoder stage of model/diff3.py (BiConvGLU :307-326, BiConvTransGLU :329-351,
 * the BatchNorm2d + PReLU that follow it :122-141 / :172-203) together with the NEXT stage's 1x1 conv1 (and, in the
 * encoder, the skip halves of both decoders' conv1) - the same mathematics as pdse_gconv_desc with epi BIGLU, korder 2.
 *
 * What changed against korder 2 (csrc/gconv3.hip):
 *  - the 32-channel conv1 output H that the gather convolutions read travels between launches as its exact bf16
 *    split planes in MFMA B-fragment order ("hp" tensors), written once by the producer's tail, so the K loop of the
 *    consumer has no VALU work and gathers with 16-byte loads (round 2 split every activation again in each of the
 *    4-6 taps that gather it);
 *  - the workgroup is 4 waves, one per SIMD with the whole 512-register file, and its loop is software-pipelined: the
 *    K loop of tile i+1 (matrix instructions only) is issued interleaved with the tail of tile i (vector
 *    instructions mostly) in ONE basic block - on gfx950 a vector instruction only hides behind a matrix instruction
 *    of the same wave's stream;
 *  - BatchNorm is folded into conv2 (W' = diag(s) Wc2, b' = s bc2 + t), -log2(e) into l_conv / r_conv
 *    (sigmoid(m) = rcp(1 + exp2(m'))), the gather biases seed the accumulators: all on the host, in float64.
 *
 * hp tensor (uint16 = bf16 bit patterns; np == 2: fp16 bit patterns), np planes (3: exact bf16 split, fp32-equivalent; 2: f16x2 - fp16
 * hi / lo of value * 2^PDSE_F16_ACT_EXP, fp32-equivalent, see below; 1: plain bf16):
 *     hp[b][tp][g][plane][fp][e],   tp in [0, Tp), g in [0, 4), fp in [0, Fp), e in [0, 8)
 *   holds channel c = 16 (g >> 1) + 8 (e >> 2) + 4 (g & 1) + (e & 3) of frame t = tp - t0, bin f = fp - f0: group
 *   g = 2 q + h is what lane half h feeds to K block q, and (q, h, e) is the accumulator register order of the
 *   producer (register 8 q + e of lane half h), so a producer lane stores its 16 channels as 2 x np 16-byte pieces.
 *   Margins (tp < t0, fp < f0, fp >= f0 + F) stay zero (decoders: the transposed convolution's out-of-range taps are
 *   addresses, not branches); encoder tensors hold the explicit pad frame (t = -1) at tp = 0.  OUTPUT tensors of a launch
 *   (nx_hp, nx_out[]) must be allocated with B + 1 items: item B takes the stores of lanes without a position.
 *
 * Packed weights (uint16, [blocks][np][64 lanes][8]): block = 16 consecutive k, lane (row = lane & 31, h = lane >> 5),
 *   element e.  Gather weights w0..w3: block tap*2 + q, k -> channel as above (packing.pack_bglu_gather);
 *   wlc / wrc / wc2 / nx_w: packing.pack_s3_chain (accumulator-register k order), pre-scaled as described.
 *   Encoder stage 1 (x0.ptr != NULL): K = 10 taps x 4 channels of the fp32 inputs (x, x_init), split in the kernel,
 *   three blocks (packing.pack_s3_gather(.., 3, 16)).
 * ------------------------------------------------------------------------------------------------------------- */
/* f16x2 planes (np == 2) hold (value * 2^PDSE_F16_ACT_EXP) as hi = RN16(.), lo = RN16(. - hi): exact to half an fp32 ulp for
 * 2^-6 <= |value| < 4094 (lo a normal fp16); below, the absolute error is <= 2^-29 (under the fp32 ulp of any value >= 2^-5 it is
 * added to); a value beyond +-4094 becomes an INFINITY in its planes (IEEE conversion, no saturation), so it surfaces as a
 * non-finite result (SamplerPipeline.check() raises; the drop-in trainer then re-runs that geometry on the three-plane bf16 split)
 * instead of a silently clipped one.  Nominal activations of the path (unit-RMS spectrograms): |value| <= ~12 (tools/act_range.py). */
#define PDSE_F16_ACT_EXP 4

typedef struct pdse_bglu_desc {
  const uint16_t* hp;      /* input planes, or NULL for encoder stage 1 */
  int64_t hp_sb;           /* batch stride, uint16 units */
  int32_t hp_Tp, hp_Fp, hp_t0, hp_f0;
  pdse_src x0, x1;         /* encoder stage 1: two fp32 sources of two channels each (x, x_init) */
  int32_t Tin, Fin;        /* logical input extent (stage 1 bounds checks) */
  int32_t ntaps, sf_in;
  int32_t tap_dt[10], tap_df[10];
  int32_t p1mask, Fout1;   /* dual phase (transposed conv): taps of the odd output bins, number of valid odd bins */
  int32_t B, Tout, Fout, np;
  const uint16_t* w0;      /* gather weights L, R (even phase); w2, w3: odd phase or NULL */
  const uint16_t* w1;
  const uint16_t* w2;
  const uint16_t* w3;
  const uint16_t* wlc;     /* [2][np][64][8], scaled by -log2 e */
  const uint16_t* wrc;
  const uint16_t* wc2;     /* [2 tiles][2][np][64][8], BatchNorm folded (C2 == 64) */
  const float* wc2v;       /* [32] (C2 == 1) */
  const uint16_t* nx_w;    /* [nx_n][4][np][64][8] */
  const float* bias0;      /* [B or 1][32] gather biases */
  const float* bias1;
  const float* bias0_t0;   /* output frame 0 (NULL: as bias0) */
  const float* bias1_t0;
  int64_t bias_sb;
  const float* blc;        /* [32], scaled by -log2 e */
  const float* brc;
  const float* bc2;        /* [C2], BatchNorm folded */
  float slope;             /* PReLU slope of the block output (1: none); must be <= 1: PReLU(v) = max(v, slope v) */
  int32_t C2;              /* 64 or 1 */
  float* out;              /* C2 == 1: one channel; C2 == 64 and nx_n == 0: the 64-channel block output */
  int64_t out_sb, out_sc, out_st, out_sf, out_off;   /* dual phase: out_sf = stride of 2 bins */
  int32_t hp_par, nx_n; /* hp_par: see below */
  /* chained tile 0 (next stage's conv1): written as planes */
  uint16_t* nx_hp;
  int64_t nx_hp_sb;
  int32_t nx_Tp, nx_Fp, nx_t0, nx_f0;
  int32_t nx_row0;         /* 1: lanes of output frame 0 also write the tile's bias to frame -1 (encoder pad frame) */
  /* hp_par / nx_par 1 (ABI 5): the bins of every (frame, group, plane) row of hp / nx_hp are stored split by parity - bin
     index i (margin included) at (i & 1) * ((Fp + 1) >> 1) + (i >> 1) - so the stride-2 taps of the encoders (sf_in = 2)
     read 512 contiguous bytes per 32 lanes like the stride-1 taps of the decoders do.  hp_par needs sf_in == 2. */
  int32_t nx_par;
  /* The fp32 skip halves (nx_add read by the decoders, nx_out written by the encoder) are kept in groups of four channels,
     [B (+1)][8 groups][T][F][4]: channel c of (b, t, bin) lives at b*sb + (c >> 2)*sc + (c & 3) + t*st + bin*sf (every
     stride a multiple of 4 floats, the base 16-byte aligned), so a lane moves its accumulator rows as four 16-byte
     accesses (ABI 5; until then [B][32][T][F] with sixteen 4-byte accesses). */
  const float* nx_add;     /* fp32 addend at (b, c, t, bin) or NULL (decoders: the encoder's skip half) */
  int64_t add_sb, add_sc, add_st, add_sf;
  /* chained tiles 1, 2 (encoder: the decoders' skip halves): fp32 */
  float* nx_out[2];
  int64_t nx_sb[2], nx_sc[2], nx_st[2], nx_sf[2];
  const float* nx_bias[3]; /* [B or 1][32] per tile */
  int64_t nx_bias_sb[3];
  /* skip_Fh > 0 (ABI 5): the bins of the skip halves (nx_out written by the encoder at bin j; nx_add read by the dual-phase
     decoders at bins 2j and 2j+1) are stored split by parity, bin i at (i & 1) * skip_Fh + (i >> 1) (skip_Fh = ceil(F / 2) of the
     tensor): each decoder phase then reads whole lines instead of every other 16 bytes of them.  0: bin i at i. */
  int32_t skip_Fh;
  /* ABI 6: items (of nx_hp_sb / nx_sb[i] elements each) the caller ALLOCATED for nx_hp and for every nx_out tensor.  Lanes beyond
     the last position store unconditionally into a dump item at index B, so B + 1 are required; the launcher refuses fewer (a
     B-item tensor handed to this public entry point would otherwise be written past its end, silently). */
  int32_t nx_items;
  /* ABI 8, np == 2 (the f16x2 form: every operand as hi + lo fp16 planes, three f16 MFMA products per multiply-add, fp32
     accumulation; see PDSE_F16_ACT_EXP): the power-of-two exponents the host scaled the four weight groups by before it split
     them (packing.f16_wexp) - [0] gather weights w0..w3, [1] wlc / wrc, [2] wc2, [3] nx_w.  Biases and every fp32 tensor in
     HBM stay in true scale; the kernel moves them to the accumulators' exponent itself.  Ignored for np 1 / 3. */
  int32_t qexp[4];
} pdse_bglu_desc;

/* fp32 [B, 32, T, F] -> hp planes (the standalone conv1 of the first decoder stage).
 * ABI 8, w[0] != NULL: the 32 channels are COMPUTED here - the 1x1 convolution over `in` (cin0 channels) and `in1` (cin1 channels, the
 * same strides; cin0 + cin1 <= 128) with per-item biases, y[c] = bias[b][c] + sum_k w[k][c] x[k] as one fp32 fmaf chain in k order -
 * for nd = 1 or 2 weight sets at once (the two decoders' stage-5 conv1 over the same two sources: model/diff3.py:343-345), whose
 * planes go to hp (set 0) and hp1 (set 1).  Replaces pdse_gconv_f32 + pdse_split_planes of each decoder (4 launches per forward). */
typedef struct pdse_planes_desc {
  const float* in;
  int64_t in_sb, in_sc, in_st, in_sf;
  uint16_t* hp;
  int64_t hp_sb;
  int32_t hp_Tp, hp_Fp, hp_t0, hp_f0;
  int32_t B, T, F, np;
  const float* in1;        /* second source or NULL (cin1 == 0) */
  const float* w[2];       /* [cin0 + cin1][32] fp32, k-major; w[0] == NULL: plain split of `in` */
  const float* bias[2];    /* [B][bias_sb] (+ 32 floats) or NULL */
  int64_t bias_sb;
  uint16_t* hp1;           /* planes of weight set 1 (nd == 2) */
  int32_t cin0, cin1, nd, pad_;
} pdse_planes_desc;

enum pdse_op_kind {
  PDSE_OP_GCONV = 0,
  PDSE_OP_TIME = 1,
  PDSE_OP_EW = 2,
  PDSE_OP_COMPAND = 3,
  PDSE_OP_WAVPREP = 4,
  PDSE_OP_OLA = 5,
  PDSE_OP_SIGMA = 6,
  PDSE_OP_LN = 7,
  PDSE_OP_LSTM = 8,
  PDSE_OP_ROWLN = 9,
  PDSE_OP_CHLN = 10,
  PDSE_OP_ATTN = 11,
  PDSE_OP_GRU = 12,
  PDSE_OP_GNCOMB = 13,
  PDSE_OP_AHAM = 14,
  PDSE_OP_QSAMPLE = 15,
  PDSE_OP_TRANSPOSE = 16,
  PDSE_OP_TCM = 17,
  PDSE_OP_CRM = 18,
  PDSE_OP_GCRNLAST = 19,
  PDSE_OP_MASKLOSS = 20,
  PDSE_OP_GLSTM = 21,
  PDSE_OP_TCM2 = 22,
  PDSE_OP_BGLU = 23,
  PDSE_OP_PLANES = 24,
  PDSE_OP_GLSTMP = 25,
  PDSE_OP_TCM2S = 26,
  PDSE_OP_DENSE = 27,
  PDSE_OP_ROWLNB = 28
};

int pdse_abi_version(void);
const char* pdse_last_error(void);
/* sizeof() of a descriptor as the library was compiled, for binding self-checks */
int pdse_desc_size(int op_kind);

/* direct launches (one operator) */
int pdse_gconv_f32(const pdse_gconv_desc* d, pdse_stream_t s);
int pdse_time_embed_f32(const pdse_time_desc* d, pdse_stream_t s);
int pdse_ew_f32(const pdse_ew_desc* d, pdse_stream_t s);
int pdse_compand_f32(const pdse_compand_desc* d, pdse_stream_t s);
int pdse_wavprep_f32(const pdse_wavprep_desc* d, pdse_stream_t s);
int pdse_ola_f32(const pdse_ola_desc* d, pdse_stream_t s);
int pdse_sigma_mask_f32(const pdse_sigma_desc* d, pdse_stream_t s);
int pdse_layernorm_f32(const pdse_ln_desc* d, pdse_stream_t s);
int pdse_lstm_f32(const pdse_lstm_desc* d, pdse_stream_t s);
int pdse_rowln_prelu_f32(const pdse_rowln_desc* d, pdse_stream_t s);
int pdse_chln_f32(const pdse_chln_desc* d, pdse_stream_t s);
int pdse_attention_f32(const pdse_attn_desc* d, pdse_stream_t s);
int pdse_bigru_f32(const pdse_gru_desc* d, pdse_stream_t s);
int pdse_gn_combine_f32(const pdse_gncomb_desc* d, pdse_stream_t s);
int pdse_aham_f32(const pdse_aham_desc* d, pdse_stream_t s);
int pdse_qsample_f32(const pdse_qsample_desc* d, pdse_stream_t s);
int pdse_transpose_f32(const pdse_transpose_desc* d, pdse_stream_t s);
int pdse_tcm_f32(const pdse_tcm_desc* d, pdse_stream_t s);
int pdse_crm_f32(const pdse_crm_desc* d, pdse_stream_t s);
int pdse_gcrnlast_f32(const pdse_gcrnlast_desc* d, pdse_stream_t s);
int pdse_masked_mse_f32(const pdse_maskloss_desc* d, pdse_stream_t s);
int pdse_glstm_f32(const pdse_glstm_desc* d, pdse_stream_t s);
int pdse_glstm_persistent_f32(const pdse_glstmp_desc* d, pdse_stream_t s);
int pdse_tcm2_bf16x3(const pdse_tcm2_desc* d, pdse_stream_t s);
int pdse_tcm2_stack_bf16x3(const pdse_tcm2s_desc* d, pdse_stream_t s);
int pdse_bglu_planes(const pdse_bglu_desc* d, pdse_stream_t s);
int pdse_split_planes(const pdse_planes_desc* d, pdse_stream_t s);
int pdse_dense_layer_bf16x3(const pdse_dense_desc* d, pdse_stream_t s);
int pdse_rowln_blocked_f32(const pdse_rowlnb_desc* d, pdse_stream_t s);
/* Kernel form of pdse_bglu_planes (ABI 6; process-wide, tuning only): -1 / 0 = 8 waves with the generated slot schedule
 * (the product kernel).  A library built with -DBGLU_FORMS also holds the forms that were measured and not kept
 * (profiles/r03_bglu_forms.txt, r04_bglu_forms.txt): 1 = 4 waves software-pipelined, 2 / 3 = 16 / 12 waves with strictly
 * sequential tiles, 4 = 8 waves with the vector-memory instructions spread over the slots.  Every form reads the same
 * descriptor and computes the same block.  Returns the previous setting, or -2 for a form this library does not hold. */
int pdse_bglu_set_form(int form);

/* plans: a recorded operator sequence replayed by one call (and capturable in a hipGraph) */
typedef struct pdse_plan pdse_plan;
int pdse_plan_create(pdse_plan** out);
int pdse_plan_add(pdse_plan* p, int op_kind, const void* desc, int tag);
int pdse_plan_size(const pdse_plan* p);
/* Bind the plan to a device ordinal: every run / capture / replay / timing call makes that device current for its
 * duration and restores the caller's current device (one process per GPU: LOCAL_RANK != 0 while torch still has
 * device 0 current).  -1 (default): launch on whatever device is current.  Direct pdse_*_f32 launches always use the
 * current device. */
int pdse_plan_set_device(pdse_plan* p, int device);
/* drop every recorded operator (and a captured graph): the plan object can be re-recorded for another geometry */
int pdse_plan_clear(pdse_plan* p);
int pdse_plan_run(pdse_plan* p, pdse_stream_t s);
int pdse_plan_run_range(pdse_plan* p, int begin, int end, pdse_stream_t s);
/* capture the whole plan into a hipGraph on stream s, then replay it with launch_graph */
int pdse_plan_build_graph(pdse_plan* p, pdse_stream_t s);
int pdse_plan_launch_graph(pdse_plan* p, pdse_stream_t s);
/* time ops [begin,end) with hipEvents on stream s: ms_out[i] = elapsed ms of op begin+i */
int pdse_plan_time_ops(pdse_plan* p, int begin, int end, pdse_stream_t s, float* ms_out);
/* sum of elapsed ms of the ops carrying `tag`, hipEvents around each (bench roofline leg) */
int pdse_plan_time_tag(pdse_plan* p, int tag, pdse_stream_t s, float* ms_out, int* count_out);
void pdse_plan_destroy(pdse_plan* p);

/* ---- network-level entry points (ABI 6; SURVEY 8b: prior_forward, eps_forward, the whole path) --------------------------------
 * A recorded plan can be written to a file together with everything it points at (packed weights, tables, buffer sizes; the
 * Python builders write it: prior-diffuse_amd/planfile.py) and loaded by a host that has neither Python nor the builders:
 * pdse_plan_load allocates the buffers on the current device, uploads the data and rebases every descriptor.  The loaded plan
 * owns its buffers (pdse_plan_destroy frees them); inputs and outputs are found by name.  The three calls below copy their
 * device-resident arguments into / out of those buffers on stream s (device to device, asynchronous) around one pdse_plan_run:
 * they replace self.model(feat) (trainer/complex_ddpm_trainer.py:941), self.model_ddpm(audio, init, t) (:968) and the per-batch
 * body of generate_wav (:921-1016) for the (B, T) the plan was recorded for.  A null argument skips that copy. */
int pdse_plan_load(const char* path, pdse_plan** out);
int pdse_plan_region(const pdse_plan* p, const char* name, void** dev_ptr, uint64_t* nbytes);
/* feat, x_init: [B,2,T,161] fp32 */
int pdse_prior_forward(pdse_plan* p, const float* feat, float* x_init, pdse_stream_t s);
/* x, x_init, eps: [B,2,T,161] fp32; t: [B] fp32 diffusion steps (fractional steps interpolate the embedding table) */
int pdse_eps_forward(pdse_plan* p, const float* x, const float* x_init, const float* t, float* eps, pdse_stream_t s);
/* wav, wav_out: [B,L] fp32; x_T, spec_out: [B,2,T,161] fp32 (spec_out may be null) */
int pdse_enhance(pdse_plan* p, const float* wav, const float* x_T, float* wav_out, float* spec_out, pdse_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* PDSE_H */
