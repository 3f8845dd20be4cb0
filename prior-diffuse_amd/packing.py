"""Host-side weight preparation: BatchNorm folding, time-bias folding and packing of
every weight matrix into the MFMA A-fragment order the kernels stream
(include/pdse.h, ``pdse_gconv_desc``).  Pure numpy, float64 folding, float32 result.

Fragment order (v_mfma_f32_32x32x2_f32, weights = A operand):
    w[mtile][kstep][lane] = W[k = 2*kstep + (lane >> 5)][co = 32*mtile + (lane & 31)]
and for a chained 1x1 product that consumes an accumulator tile as its B operand the
k order follows the accumulator rows:  k = rho(r, lane >> 5),
rho(r, h) = (r & 3) + 8*(r >> 2) + 4*h.
"""
import numpy as np

BN_EPS = 1e-5

_LANE = np.arange(64)
_COL = _LANE & 31
_H = _LANE >> 5
RHO = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for h in (0, 1)] for r in range(16)])  # [16][2]


def _np(x):
    return x.detach().cpu().numpy().astype(np.float64) if hasattr(x, "detach") else np.asarray(x, np.float64)


def pack_a(wk):
    """wk [K, M] (k-major) -> float32 [mtiles, ksteps, 64]."""
    wk = np.asarray(wk, np.float64)
    K, M = wk.shape
    ksteps, mtiles = (K + 1) // 2, (M + 31) // 32
    pad = np.zeros((ksteps * 2, mtiles * 32))
    pad[:K, :M] = wk
    k_idx = 2 * np.arange(ksteps)[:, None] + _H[None, :]            # [ksteps, 64]
    out = np.empty((mtiles, ksteps, 64), np.float32)
    for mt in range(mtiles):
        out[mt] = pad[k_idx, 32 * mt + _COL[None, :]]
    return out


# (epilogue, taps, two sources, load transform) -> channel pairs per chunk (CP) of the pipelined
# (korder 1) instantiations in csrc/gconv2.hip::pdse_gconv2_launch — keep in sync.
V2_CP = {
    (0, 1, False, 0): 8, (0, 1, False, 1): 8, (0, 1, True, 0): 8, (0, 4, False, 0): 2,
    (0, 3, False, 0): 4, (0, 6, False, 0): 2,
    (1, 1, True, 0): 4, (1, 2, True, 0): 2, (1, 3, False, 0): 4, (1, 5, False, 2): 4,
    (2, 2, False, 0): 4, (2, 4, False, 0): 2, (2, 6, False, 0): 2, (2, 10, False, 0): 2, (2, 10, True, 0): 2,
}


def v2_supported(epi, ntaps, two_src, xf_mode, cin1):
    return (not cin1) and (epi, ntaps, bool(two_src), xf_mode) in V2_CP


def korder1_rows(ntaps, c0, c1, cp):
    """Row order of the pipelined kernel: for each source, channel pairs in chunks of ``cp``
    (the last chunk of a source zero-padded), taps innermost; each k-step holds the two
    channels of a pair.  Returns tap-major row indices (k = tap*Cin + ci), -1 = zero row."""
    cin = c0 + c1
    rows = []
    for cbase, c in ((0, c0), (c0, c1)):
        cps = c // 2
        for q in range((cps + cp - 1) // cp):
            for cc in range(cp):
                pair = q * cp + cc
                for tap in range(ntaps):
                    for hh in (0, 1):
                        rows.append(tap * cin + cbase + 2 * pair + hh if pair < cps else -1)
    rows = np.asarray(rows)
    assert (len(rows) // 2) % 4 == 0
    return rows


def pack_a4(wk, rows):
    """Pipelined-kernel weights: rows of wk [K, M] re-ordered by ``rows`` (korder1_rows) and
    packed 4 k-steps deep: float32 [mtiles, ksteps/4, 64 lanes, 4]."""
    wk = np.asarray(wk, np.float64)
    M = wk.shape[1]
    ordered = np.where((rows >= 0)[:, None], wk[np.maximum(rows, 0)], 0.0)   # [2*ksteps, M]
    ksteps, mtiles = len(rows) // 2, (M + 31) // 32
    pad = np.zeros((2 * ksteps, mtiles * 32))
    pad[:, :M] = ordered
    g = np.arange(ksteps // 4)[:, None, None]
    lane_h = _H[None, :, None]
    i = np.arange(4)[None, None, :]
    k_idx = 2 * (4 * g + i) + lane_h                                  # [groups, 64, 4]
    out = np.empty((mtiles, ksteps // 4, 64, 4), np.float32)
    for mt in range(mtiles):
        out[mt] = pad[k_idx, (32 * mt + _COL)[None, :, None]]
    return out


def unpack_a4(w, mtiles, ksteps):
    """Inverse of pack_a4's fragment step: -> [2*ksteps, mtiles*32] in kernel row order."""
    w = np.asarray(w, np.float32).reshape(mtiles, ksteps // 4, 2, 32, 4)     # [mt, g, h, col, i]
    return w.transpose(1, 4, 2, 0, 3).reshape(2 * ksteps, mtiles * 32)        # k = 2(4g+i)+h


def pack_chain(w):
    """w [Mout, 32] (out x in) for a 1x1 product fed from an accumulator tile
    -> float32 [mtiles, 16, 64] with k ordered by rho."""
    w = np.asarray(w, np.float64)
    M, K = w.shape
    assert K == 32
    mtiles = (M + 31) // 32
    pad = np.zeros((mtiles * 32, 32))
    pad[:M] = w
    k_idx = RHO[:, _H]                                              # [16, 64]
    out = np.empty((mtiles, 16, 64), np.float32)
    for mt in range(mtiles):
        out[mt] = pad[32 * mt + _COL[None, :], k_idx]
    return out


def bn_fold(sd, prefix):
    """Eval-mode BatchNorm -> (scale, shift) with y*scale + shift."""
    w, b = _np(sd[prefix + ".weight"]), _np(sd[prefix + ".bias"])
    m, v = _np(sd[prefix + ".running_mean"]), _np(sd[prefix + ".running_var"])
    scale = w / np.sqrt(v + BN_EPS)
    return scale.astype(np.float32), (b - m * scale).astype(np.float32)


def conv_kmat(w, taps_kk):
    """Conv weight [Cout, Cin, kh, kw] + list of (kt, kf) -> [ntaps*Cin, Cout], k = tap*Cin + ci."""
    w = _np(w)
    return np.concatenate([w[:, :, kt, kf].T for kt, kf in taps_kk], axis=0)


def convT_kmat(w, taps_kk):
    """ConvTranspose weight [Cin, Cout, kh, kw] + (kt, kf) list -> [ntaps*Cin, Cout]."""
    w = _np(w)
    return np.concatenate([w[:, :, kt, kf] for kt, kf in taps_kk], axis=0)


def conv_taps(kh, kw, top_pad):
    """Strided Conv2d k=(kh,kw) on an input padded by ``top_pad`` frames on top:
    weight index (kt,kf) reads frame t + kt - top_pad, bin j*stride + kf."""
    kk = [(kt, kf) for kt in range(kh) for kf in range(kw)]
    taps = [(kt - top_pad, kf) for kt, kf in kk]
    return kk, taps


def convT_phase_taps(kh, kw, phase):
    """Stride-(1,2) ConvTranspose2d, output bins f_o = 2j + phase: weight (kt,kf) with
    kf ≡ phase (mod 2) reads frame t - kt, bin j - (kf - phase)/2."""
    kk = [(kt, kf) for kt in range(kh) for kf in range(phase, kw, 2)]
    taps = [(-kt, -(kf - phase) // 2) for kt, kf in kk]
    return kk, taps


def taps_array(taps):
    return np.asarray(taps, np.int32).reshape(-1, 2)


# --------------------------------------------------------------------------
# STFT / inverse STFT bases (torch.stft / torch.istft conventions of
# trainer/complex_ddpm_trainer.py:926-930, :1010-1015): n_fft = win = 320, hop 160,
# periodic hann, onesided, no normalisation.
# --------------------------------------------------------------------------
def hann_periodic(n):
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def stft_kmat(n_fft=320):
    """[n_fft taps, 2*F] : co = ri*F + f."""
    F = n_fft // 2 + 1
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    f = np.arange(F, dtype=np.float64)[None, :]
    w = hann_periodic(n_fft)[:, None]
    ang = 2.0 * np.pi * n * f / n_fft
    return np.concatenate([w * np.cos(ang), -w * np.sin(ang)], axis=1)


def istft_kmat(n_fft=320):
    """[(f, ri) = F*2 rows, n_fft] windowed inverse real DFT, k = f*2 + ri."""
    F = n_fft // 2 + 1
    n = np.arange(n_fft, dtype=np.float64)[None, :]
    f = np.arange(F, dtype=np.float64)[:, None]
    a = np.full((F, 1), 2.0)
    a[0, 0] = a[F - 1, 0] = 1.0
    w = hann_periodic(n_fft)[None, :]
    ang = 2.0 * np.pi * f * n / n_fft
    re = a * np.cos(ang) * w / n_fft
    im = -a * np.sin(ang) * w / n_fft
    out = np.empty((F * 2, n_fft))
    out[0::2] = re
    out[1::2] = im
    return out


ISTFT_ROW = 168   # F = 161 bins padded to a multiple of 8: K = 2 * 168 = 21 blocks of 16


def istft_kmat_rows(n_fft=320):
    """The same matrix with K ordered (ri, f) and each half padded to ISTFT_ROW rows of zeros: [2 * 168, n_fft] -
    the K dimension of the ISTFT as ONE channel axis of the decompressed spectrogram stored [B][T][2 * 168]."""
    F = n_fft // 2 + 1
    k = istft_kmat(n_fft)
    out = np.zeros((2 * ISTFT_ROW, n_fft))
    out[:F] = k[0::2]
    out[ISTFT_ROW:ISTFT_ROW + F] = k[1::2]
    return out


# ------------------------------------------------------------------------------------------
# grouped LSTM, layer wavefront (csrc/lstm.hip: glstm_wave_kernel; include/pdse.h: pdse_glstm_desc)
# ------------------------------------------------------------------------------------------
def pack_lstm_slices(W, korder=None):
    """W [4H, K] (gate-major rows i,f,g,o) -> float32 [H/8 slices][K/8 k-groups][64 lanes][4]: slice s holds gate rows
    q*H + 8s + u as tile row q*8 + u; lane (col, h) of k-group q, entry i = W[row(col), korder[2*(4q+i) + h]]."""
    W = np.asarray(W, np.float64)
    H4, K = W.shape
    H = H4 // 4
    kk = np.arange(K) if korder is None else np.asarray(korder)
    col = np.arange(32)
    rows = (col // 8)[None, :] * H + 8 * np.arange(H // 8)[:, None] + (col % 8)[None, :]          # [S, 32]
    q, h, i = np.arange(K // 8)[:, None, None], np.arange(2)[None, :, None], np.arange(4)[None, None, :]
    kidx = kk[2 * (4 * q + i) + h]                                                                  # [Q, 2, 4]
    out = W[rows[:, None, None, :, None], kidx[None, :, :, None, :]]                                # [S, Q, 2, 32, 4]
    return np.ascontiguousarray(out.reshape(H // 8, K // 8, 64, 4), np.float32)


def glstm_ih2_korder(g, H=512):
    """K order of stage B (layer-2 input projection) for chunk g of the interleaved layer-1 output: position
    k' = 2*(4*gq + i) + hh reads source group g' = gq >> 5, unit u = 8*(32g + (gq & 31)) + 2i + hh, which is feature
    2u + g' of the LayerNorm input; returns the column of W_ih (chunk-local feature index) for every k'."""
    order = np.empty(H, np.int64)
    for gq in range(H // 8):
        gs, kq = gq >> 5, (H // 16) * g + (gq & 31)
        for i in range(4):
            for hh in range(2):
                u = 8 * kq + 2 * i + hh
                order[2 * (4 * gq + i) + hh] = 2 * u + gs - H * g
    assert sorted(order.tolist()) == list(range(H))
    return order


# ------------------------------------------------------------------------------------------
# fused TCM residual block (csrc/tcm.hip, include/pdse.h: pdse_tcm_desc)
# ------------------------------------------------------------------------------------------
_RHO = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for h in (0, 1)] for r in range(16)])


def pack_tcm_branch(k_main, k_mask):
    """k_* [320, 64] (row = tap*64 + channel) -> [2 mi][2 kh][20 g][2][64 lanes][4]."""
    out = np.zeros((2, 2, 20, 2, 64, 4), np.float32)
    for which, km in enumerate((k_main, k_mask)):
        km = np.asarray(km, np.float32).reshape(160, 2, 2, 32)            # [ks, h, mi, col]
        km = km.reshape(2, 20, 4, 2, 2, 32)                               # [kh, g, e, h, mi, col]
        out[:, :, :, which] = km.transpose(4, 0, 1, 3, 5, 2).reshape(2, 2, 20, 64, 4)
    return out


def pack_tcm_conv2(k2):
    """k2 [64, 256] (row = gate channel) -> [8 mt][8 g][64 lanes][4]."""
    k = np.asarray(k2, np.float32).reshape(8, 4, 2, 8, 32)                # [g, e, h, mt, col]
    return np.ascontiguousarray(k.transpose(3, 0, 2, 4, 1).reshape(8, 8, 64, 4))


def pack_tcm_next(w1):
    """w1 [64 out, 256 in] -> [4 w][2 q][2 mo][4 g][64 lanes][4]: K row = 64*w + 32*q + rho(4*g+e, h)."""
    w1 = np.asarray(w1, np.float32)
    out = np.zeros((4, 2, 2, 4, 2, 32, 4), np.float32)                    # [w, q, mo, g, h, col, e]
    for g in range(4):
        for e in range(4):
            for h in (0, 1):
                rows = _RHO[4 * g + e, h]
                for w in range(4):
                    for q in range(2):
                        cin = 64 * w + 32 * q + rows
                        out[w, q, :, g, h, :, e] = w1[:, cin].reshape(2, 32)
    return out.reshape(4, 2, 2, 4, 64, 4)


# ------------------------------------------------------------------------------------------
# split-bf16 operands (csrc/gconv3.hip): every fp32 value is the EXACT sum of three bf16 numbers
#   x = b1 + b2 + b3,   b1 = trunc16(x), b2 = trunc16(x - b1), b3 = x - b1 - b2
# (24 significand bits = 3 x 8), so a product a*b = sum_ij a_i b_j; the kernels evaluate the six leading terms
# (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2) on the bf16 matrix cores with fp32 accumulation - what is dropped is below
# 2^-23 of the product, i.e. fp32-level accuracy at 16/6 of the fp32 MFMA rate.
# ------------------------------------------------------------------------------------------
def split_bf16x3(w):
    """float array -> (b1, b2, b3) uint16 arrays (the bf16 bit patterns), w == b1 + b2 + b3 exactly in fp32."""
    w = np.ascontiguousarray(w, np.float32)
    m = np.uint32(0xFFFF0000)
    u1 = w.view(np.uint32) & m
    r1 = w - u1.view(np.float32)
    u2 = r1.view(np.uint32) & m
    r2 = r1 - u2.view(np.float32)
    u3 = r2.view(np.uint32) & m
    assert np.array_equal(u1.view(np.float32) + u2.view(np.float32) + u3.view(np.float32), w) or not np.all(np.isfinite(w))
    return tuple((u >> np.uint32(16)).astype(np.uint16) for u in (u1, u2, u3))


def join_bf16x3(planes):
    """Inverse of split_bf16x3 on uint16 planes [..., 3 stacked first]."""
    f = [(np.asarray(p, np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32) for p in planes]
    return f[0] + f[1] + f[2]


def rho_bf16(s, j, h):
    """Accumulator row held by element j of k-block s on lane half h when a 32x32 f32 accumulator tile is used as the
    B operand of v_mfma_f32_32x32x16_bf16 (registers 8s..8s+7): row of register r = (r & 3) + 8 (r >> 2) + 4 h."""
    r = 8 * s + j
    return (r & 3) + 8 * (r >> 2) + 4 * h


def pack_s3_gather(wk, ntaps, cin=32, npl=3, qe=0):
    """wk [ntaps*cin, 32] (k = tap*cin + c, k-major) -> uint16 [ntaps*cin/16 blocks][npl planes][64 lanes][8]:
    block tap*(cin/16) + q, lane (row = lane & 31, h = lane >> 5), element j = wk[tap*cin + 16q + 8h + j][row].
    npl 3: exact three-way bf16 split; 1: the RNE bf16 value (bf16 mode)."""
    wk = np.asarray(wk, np.float64).astype(np.float32)
    assert wk.shape == (ntaps * cin, 32) and cin % 16 == 0
    nb = ntaps * cin // 16
    k = (16 * np.arange(nb)[:, None, None] + 8 * np.arange(2)[None, :, None] + np.arange(8)[None, None, :])   # [nb, h, j]
    frag = wk[k][:, :, :, :].transpose(0, 1, 3, 2)                      # [nb, h, row, j]
    frag = frag.reshape(nb, 64, 8)
    if npl == 1:
        return np.ascontiguousarray(bf16_rne(frag)[:, None])             # [nb, 1, 64, 8]
    if npl == 2:                                                         # f16x2: fp16 hi + lo of frag * 2^qe
        return np.ascontiguousarray(np.stack(split_f16x2(np.ldexp(frag, qe)), 1))
    p = split_bf16x3(frag)
    return np.ascontiguousarray(np.stack(p, 1))                          # [nb, 3, 64, 8]


def unpack_s3_gather(packed, ntaps, cin=32, qe=0):
    packed = np.asarray(packed, np.uint16)
    npl = packed.size // (ntaps * cin // 16 * 512)
    packed = packed.reshape(ntaps * cin // 16, npl, 2, 32, 8)                          # [nb, plane, h, row, j]
    f = from_planes(packed.transpose(1, 0, 2, 3, 4), qe)                                # [nb, h, row, j]
    return f.transpose(0, 1, 3, 2).reshape(ntaps * cin, 32)                             # k = 16 nb + 8h + j


def pack_s3_chain(w):
    """w [Mout, Kin] (out x in; Kin = 32 or 64 accumulator channels) -> uint16 [mtiles][Kin/16 blocks][3][64][8]:
    block s, lane (row, h), element j = w[32 mt + row][32 (s >> 1) + rho_bf16(s & 1, j, h)]."""
    w = np.asarray(w, np.float64).astype(np.float32)
    M, K = w.shape
    assert K % 32 == 0
    mt = (M + 31) // 32
    pad = np.zeros((mt * 32, K), np.float32)
    pad[:M] = w
    s_, h_, j_ = np.meshgrid(np.arange(K // 16), np.arange(2), np.arange(8), indexing="ij")
    kk = 32 * (s_ >> 1) + ((8 * (s_ & 1) + j_) & 3) + 8 * ((8 * (s_ & 1) + j_) >> 2) + 4 * h_    # [nb, h, j]
    frag = pad.reshape(mt, 32, K)[:, :, kk]                               # [mt, row, nb, h, j]
    frag = frag.transpose(0, 2, 3, 1, 4).reshape(mt, K // 16, 64, 8)      # lane = h*32 + row
    p = split_bf16x3(frag)
    return np.ascontiguousarray(np.stack(p, 2))                           # [mt, nb, 3, 64, 8]


def unpack_s3_chain(packed, mt, K):
    packed = np.asarray(packed, np.uint16).reshape(mt, K // 16, 3, 2, 32, 8)   # [mt, nb, plane, h, row, j]
    f = join_bf16x3([packed[:, :, i] for i in range(3)])                      # [mt, nb, h, row, j]
    out = np.zeros((mt * 32, K), np.float32)
    for s in range(K // 16):
        for h in range(2):
            for j in range(8):
                out[:, 32 * (s >> 1) + rho_bf16(s & 1, j, h)] = f[:, s, h, :, j].reshape(-1)
    return out


# split-bf16 TCM residual block (csrc/tcm2.hip, include/pdse.h: pdse_tcm2_desc)
TCM2_HS_PAD = 64           # zero frames on either side of hs (2 x the largest dilation)


def tcm2_hs_shape(B, T, npl=3):
    """hs (uint16): [B][2 branch][4 kb][2 kg][npl planes][T + 2 pad][8], channel 16 kb + 8 kg + j, frame t at t + pad."""
    return (B, 2, 4, 2, npl, T + 2 * TCM2_HS_PAD, 8)


def pack_tcm2_branch(k_main, k_mask, npl=3, qe=0):
    """k_* [320, 64] (row = tap*64 + channel) -> uint16 [2 main|mask][2 mi][20 blocks][npl][64][8]."""
    return np.ascontiguousarray(np.stack([np.stack([pack_s3_gather(np.asarray(k)[:, 32 * mi:32 * mi + 32], 5, 64, npl, qe)
                                                    for mi in range(2)]) for k in (k_main, k_mask)]))


def unpack_tcm2_branch(packed, qe=0):
    packed = np.asarray(packed, np.uint16)
    packed = packed.reshape(2, 2, 20, packed.size // (2 * 2 * 20 * 512), 64, 8)
    return [np.concatenate([unpack_s3_gather(packed[br, mi], 5, 64, qe) for mi in range(2)], axis=1) for br in range(2)]


def pack_tcm2_conv2(k2, npl=3, qe=0):
    """k2 [64, 256] (row = gate channel) -> uint16 [8 mt][4 kb][npl][64][8]."""
    return np.ascontiguousarray(np.stack([pack_s3_gather(np.asarray(k2)[:, 32 * mt:32 * mt + 32], 1, 64, npl, qe) for mt in range(8)]))


def unpack_tcm2_conv2(packed, qe=0):
    packed = np.asarray(packed, np.uint16)
    packed = packed.reshape(8, 4, packed.size // (8 * 4 * 512), 64, 8)
    return np.concatenate([unpack_s3_gather(packed[mt], 1, 64, qe) for mt in range(8)], axis=1)


def tcm2_split_h(v_main, v_mask, npl=3):
    """Transformed bottleneck tensors [B, 64, T] of the two branches -> hs uint16 (tcm2_hs_shape; margins zero)."""
    B, C, T = v_main.shape
    hs = np.zeros(tcm2_hs_shape(B, T, npl), np.uint16)
    for br, v in enumerate((v_main, v_mask)):
        p = to_planes(np.asarray(v, np.float32).reshape(B, 4, 2, 8, T).transpose(0, 1, 2, 4, 3), npl, F16_ACT_EXP)   # [npl][B, kb, kg, T, j]
        for i in range(npl):
            hs[:, br, :, :, i, TCM2_HS_PAD:TCM2_HS_PAD + T, :] = p[i]
    return hs


def tcm2_join_h(hs, B, T):
    """Inverse of tcm2_split_h: (v_main, v_mask) float32 [B, 64, T]; asserts the zero margins."""
    hs = np.asarray(hs, np.uint16)
    npl = hs.size // int(np.prod(tcm2_hs_shape(B, T, 1)))
    hs = hs.reshape(tcm2_hs_shape(B, T, npl))
    assert not hs[..., :TCM2_HS_PAD, :].any() and not hs[..., TCM2_HS_PAD + T:, :].any(), "hs: the margins must stay zero"
    out = []
    for br in range(2):
        v = from_planes(np.stack([hs[:, br, :, :, i, TCM2_HS_PAD:TCM2_HS_PAD + T, :] for i in range(npl)], 0), F16_ACT_EXP)    # [B, kb, kg, T, j]
        out.append(np.ascontiguousarray(v.transpose(0, 1, 2, 4, 3).reshape(B, 64, T)))
    return out


def s3_gemm_krows(ntaps, c0, c1):
    """Row order (k = tap*Cin + ci of the k-major weight matrix) of the K blocks of csrc/gconv4.hip: 16 consecutive
    channels per block, blocks in the order (source, tap, channel block)."""
    cin = c0 + c1
    rows = []
    for cbase, c in ((0, c0), (c0, c1)):
        for tap in range(ntaps):
            for cb in range(c // 16):
                rows.extend(tap * cin + cbase + 16 * cb + np.arange(16))
    return np.asarray(rows, np.int64)


def _pack_gemm_rows(ordered, npl, qe=0):
    """ordered [nkb*16, mt*32] (K rows already in K-step order) -> uint16 [nkb][mt][npl][64 lanes][8] (npl 2: fp16 hi / lo of
    the value * 2^qe)."""
    nkb, mt = ordered.shape[0] // 16, ordered.shape[1] // 32
    frag = ordered.reshape(nkb, 2, 8, mt, 32).transpose(0, 3, 1, 4, 2)    # [kb, mt, h, row, j]
    frag = frag.reshape(nkb, mt, 64, 8)
    if npl == 1:
        return np.ascontiguousarray(bf16_rne(frag)[:, :, None])           # [kb, mt, 1, 64, 8]
    if npl == 2:
        return np.ascontiguousarray(np.stack(split_f16x2(np.ldexp(frag.astype(np.float32), qe)), 2))
    p = split_bf16x3(frag)
    return np.ascontiguousarray(np.stack(p, 2))                           # [kb, mt, 3, 64, 8]


def pack_s3_gemm(wk, ntaps, c0, c1, npl=3, qe=0):
    """wk [ntaps*(c0+c1), Cout] (k-major) -> uint16 [K blocks][ceil(Cout/32)][npl planes][64 lanes][8]
    (npl 3: exact three-way bf16 split, korder 3; 1: the RNE bf16 value, korder 4 - the opt-in bf16 mode)."""
    wk = np.asarray(wk, np.float64).astype(np.float32)
    K, M = wk.shape
    assert K == ntaps * (c0 + c1) and c0 % 16 == 0 and c1 % 16 == 0
    mt = (M + 31) // 32
    pad = np.zeros((K, mt * 32), np.float32)
    pad[:, :M] = wk
    return _pack_gemm_rows(pad[s3_gemm_krows(ntaps, c0, c1)], npl, qe)


def dense_krows(cin):
    """K-step order of csrc/dense.hip for the (2,3) kernel of a dense-block layer: (16-channel block, time tap kt, bin tap kf);
    rows index the k-major matrix of conv_kmat with taps [(kt, kf) for kt in 0..1 for kf in 0..2]."""
    rows = []
    for kb in range(cin // 16):
        for kt in range(2):
            for kf in range(3):
                rows.extend((kt * 3 + kf) * cin + 16 * kb + np.arange(16))
    return np.asarray(rows, np.int64)


def pack_dense(wk, cin, npl=3, qe=0):
    """wk [6*cin, 64] (k-major, conv_kmat order) -> uint16 [cin/16 * 6 K steps][2][npl][64][8] (pdse_dense_desc.w)."""
    wk = np.asarray(wk, np.float64).astype(np.float32)
    assert wk.shape == (6 * cin, 64) and cin % 16 == 0
    return _pack_gemm_rows(wk[dense_krows(cin)], npl, qe)


def unpack_s3_gemm(packed, ntaps, c0, c1, M, npl=3, qe=0):
    K = ntaps * (c0 + c1)
    mt = (M + 31) // 32
    packed = np.asarray(packed, np.uint16).reshape(K // 16, mt, npl, 2, 32, 8)    # [kb, mt, plane, h, row, j]
    f = (join_bf16x3([packed[:, :, i] for i in range(3)]) if npl == 3 else
         from_planes(np.stack([packed[:, :, 0], packed[:, :, 1]], 0), qe) if npl == 2 else bf16_to_f32(packed[:, :, 0]))   # [kb, mt, h, row, j]
    ordered = f.transpose(0, 2, 4, 1, 3).reshape(K, mt * 32)                       # k within block = 8h + j
    out = np.zeros((K, mt * 32), np.float32)
    out[s3_gemm_krows(ntaps, c0, c1)] = ordered
    return out[:, :M]


# --------------------------------------------------------------------------
# BiConv(Trans)GLU blocks on plane tensors (csrc/bglu.hip, include/pdse.h: pdse_bglu_desc)
# --------------------------------------------------------------------------
LOG2E = 1.4426950408889634
HP_T0, HP_F0 = 1, 2          # margins of an hp tensor: one frame on top (pad frame / zeros), two bins either side


def bf16_rne(x):
    """float array -> uint16 bf16 bit patterns, round to nearest even (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def bf16_to_f32(u):
    return (np.asarray(u, np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


# f16x2 operands (np == 2; include/pdse.h: PDSE_F16_ACT_EXP, csrc/gconv_common.h: split8h): a value, scaled by a power of two
# into the fp16 range, is hi = RN16(x), lo = RN16(x - hi) - 22 significand bits and the sign of lo, |x - hi - lo| <= 2^-23 |x|
# while lo is a normal fp16 - and a product is a1 b1 + a1 b2 + a2 b1 on the f16 matrix cores (fp32 accumulation).
F16_ACT_EXP = 4            # activations: planes hold value * 2^4 (exact for 2^-6 <= |value| < 4094; include/pdse.h)
F16_MAX = 65504.0


def f16_wexp(*mats):
    """Power-of-two exponent q for a group of weight matrices that share one accumulator: max |w| * 2^q in [2^13, 2^14), so
    every weight down to 2^-15 of the largest keeps a normal lo part (the kernel undoes 2^q where it re-scales the
    accumulator; exact)."""
    m = max((float(np.max(np.abs(np.asarray(w, np.float64)))) if np.asarray(w).size else 0.0) for w in mats)
    if not np.isfinite(m) or m <= 0.0:
        return 0
    return int(np.clip(13 - int(np.floor(np.log2(m))), -40, 40))


def split_f16x2(x):
    """float32 array (already scaled) -> (hi, lo) uint16 fp16 bit patterns: hi = RN16(x), lo = RN16(x - hi), IEEE conversions as
    the kernels' v_cvt_pk_f16_f32: a value beyond the fp16 range gives hi = +-inf, lo = -+inf (their sum is not a number: an
    overflow never passes for a value)."""
    x = np.ascontiguousarray(x, np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


def f16_to_f32(u):
    return np.asarray(u, np.uint16).view(np.float16).astype(np.float32)


def to_planes(frag, npl, q=0):
    """float array [...] -> uint16 [npl, ...]: the exact three-way bf16 split (npl 3), the RNE bf16 value (npl 1), or the
    fp16 hi / lo pair of frag * 2^q (npl 2)."""
    if npl == 3:
        return np.stack(split_bf16x3(frag), 0)
    if npl == 2:
        return np.stack(split_f16x2(np.ldexp(np.asarray(frag, np.float32), q)), 0)
    return bf16_rne(frag)[None]


def from_planes(planes, q=0):
    """Inverse of to_planes on a leading plane axis (two planes: fp16 hi + lo, scaled back by 2^-q)."""
    planes = np.asarray(planes, np.uint16)
    if planes.shape[0] == 2:
        return np.ldexp(f16_to_f32(planes[0]) + f16_to_f32(planes[1]), -q).astype(np.float32)
    return sum(bf16_to_f32(planes[i]) for i in range(planes.shape[0]))


def bglu_chan(q, h, e):
    """Channel held by element e of group g = 2q + h of an hp tensor = accumulator register 8q + e of lane half h."""
    return 16 * q + 8 * (e >> 2) + 4 * h + (e & 3)


def hp_shape(B, T, F, npl):
    """hp[b][tp][g][plane][fp][e]: frame t at tp = t + HP_T0 (tp 0: pad frame / zeros), bin f at fp = f + HP_F0."""
    return (B, T + HP_T0, 4, npl, F + 2 * HP_F0, 8)


def hp_par_pos(Fp):
    """Parity-split row order of an hp tensor (pdse_bglu_desc.hp_par / nx_par): bin index i (margin included) is stored at
    pos[i] = (i & 1) * ((Fp + 1) >> 1) + (i >> 1)."""
    i = np.arange(Fp)
    return (i & 1) * ((Fp + 1) >> 1) + (i >> 1)


def hp_split(x, npl):
    """[B, 32, T, F] float -> hp uint16 (margins zero)."""
    x = np.asarray(x, np.float32)
    B, C, T, F = x.shape
    assert C == 32
    hp = np.zeros(hp_shape(B, T, F, npl), np.uint16)
    q, h, e = np.meshgrid(np.arange(2), np.arange(2), np.arange(8), indexing="ij")
    ch = bglu_chan(q, h, e).reshape(4, 8)                                   # [g, e]
    v = x[:, ch]                                                            # [B, g, e, T, F]
    p = to_planes(v.transpose(0, 3, 1, 4, 2), npl, F16_ACT_EXP)             # [npl, B, T, g, F, e]
    hp[:, HP_T0:, :, :, HP_F0:HP_F0 + F, :] = p.transpose(1, 2, 3, 0, 4, 5)
    return hp


def hp_join(hp, with_margins=False):
    """hp uint16 -> float [B, 32, Tp, Fp] (with_margins) or [B, 32, T, F]."""
    hp = np.asarray(hp, np.uint16)
    B, Tp, _, npl, Fp, _ = hp.shape
    v = from_planes(hp.transpose(3, 0, 1, 2, 4, 5), F16_ACT_EXP)            # [B, Tp, g, Fp, e]
    out = np.zeros((B, 32, Tp, Fp), np.float32)
    for g in range(4):
        for e in range(8):
            out[:, bglu_chan(g >> 1, g & 1, e)] = v[:, :, g, :, e]
    return out if with_margins else out[:, :, HP_T0:, HP_F0:Fp - HP_F0]


def pack_bglu_gather(wk, ntaps, npl, qe=0):
    """wk [ntaps*32, 32] (k = tap*32 + channel, k-major) -> uint16 [ntaps*2 blocks][npl][64 lanes][8]: block tap*2 + q,
    lane (row, h), element e = wk[tap*32 + bglu_chan(q, h, e)][row]."""
    wk = np.asarray(wk, np.float64).astype(np.float32)
    assert wk.shape == (ntaps * 32, 32)
    tap, q, h, e = np.meshgrid(np.arange(ntaps), np.arange(2), np.arange(2), np.arange(8), indexing="ij")
    k = tap * 32 + bglu_chan(q, h, e)                                       # [tap, q, h, e]
    frag = wk[k].transpose(0, 1, 2, 4, 3).reshape(ntaps * 2, 64, 8)         # [.., h, row, e] -> lane = h*32 + row
    return np.ascontiguousarray(to_planes(frag, npl, qe).transpose(1, 0, 2, 3))  # [nb, npl, 64, 8]


def unpack_bglu_gather(packed, ntaps, qe=0):
    packed = np.asarray(packed, np.uint16)
    npl = packed.shape[1]
    f = from_planes(packed.transpose(1, 0, 2, 3), qe).reshape(ntaps, 2, 2, 32, 8)   # [tap, q, h, row, e]
    out = np.zeros((ntaps * 32, 32), np.float32)
    for q in range(2):
        for h in range(2):
            for e in range(8):
                out[np.arange(ntaps) * 32 + bglu_chan(q, h, e)] = f[:, q, h, :, e]
    return out


def pack_bglu_in4(wk, npl, q=0):
    """Encoder stage 1: wk [40, 32] (k = tap*4 + channel over (x 0, x 1, x_init 0, x_init 1)) -> [3 blocks][npl][64][8],
    block q, lane (row, h), element e = wk[16q + 8h + e][row] (rows >= 40 zero)."""
    wk = np.concatenate([np.asarray(wk, np.float64), np.zeros((8, 32))], 0).astype(np.float32)
    k = 16 * np.arange(3)[:, None, None] + 8 * np.arange(2)[None, :, None] + np.arange(8)[None, None, :]
    frag = wk[k].transpose(0, 1, 3, 2).reshape(3, 64, 8)
    return np.ascontiguousarray(to_planes(frag, npl, q).transpose(1, 0, 2, 3))


def unpack_bglu_in4(packed, q=0):
    packed = np.asarray(packed, np.uint16)
    f = from_planes(packed.transpose(1, 0, 2, 3), q).reshape(3, 2, 32, 8)       # [q, h, row, e]
    return f.transpose(0, 1, 3, 2).reshape(48, 32)[:40]


def pack_bglu_chain(w, npl, q=0):
    """w [Mout, Kin] -> uint16 [mtiles][Kin/16 blocks][npl][64][8] in the k order of an accumulator tile used as B operand
    (pack_s3_chain with a plane count)."""
    w = np.asarray(w, np.float64).astype(np.float32)
    M, K = w.shape
    mt = (M + 31) // 32
    pad = np.zeros((mt * 32, K), np.float32)
    pad[:M] = w
    s_, h_, j_ = np.meshgrid(np.arange(K // 16), np.arange(2), np.arange(8), indexing="ij")
    kk = 32 * (s_ >> 1) + ((8 * (s_ & 1) + j_) & 3) + 8 * ((8 * (s_ & 1) + j_) >> 2) + 4 * h_
    frag = pad.reshape(mt, 32, K)[:, :, kk].transpose(0, 2, 3, 1, 4).reshape(mt, K // 16, 64, 8)
    return np.ascontiguousarray(to_planes(frag, npl, q).transpose(1, 2, 0, 3, 4))   # [mt, nb, npl, 64, 8]


def unpack_bglu_chain(packed, M, K, q=0):
    packed = np.asarray(packed, np.uint16)
    mt = packed.shape[0]
    f = from_planes(packed.transpose(2, 0, 1, 3, 4), q).reshape(mt, K // 16, 2, 32, 8)   # [mt, nb, h, row, j]
    out = np.zeros((mt * 32, K), np.float32)
    for s in range(K // 16):
        for h in range(2):
            for j in range(8):
                out[:, 32 * (s >> 1) + rho_bf16(s & 1, j, h)] = f[:, s, h, :, j].reshape(-1)
    return out[:M]
