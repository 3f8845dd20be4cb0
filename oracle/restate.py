"""ORACLE — CPU restatement of the Prior-DiffuSE sampling path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  It is the *checker* of the HIP path, never the thing
measured or shipped; the product package ``prior-diffuse_amd`` does not import it
and fails loudly when ``libpdse.so`` is missing.

What it is: a plain PyTorch-CPU fp32 functional restatement (no ``nn.Module``)
of every reference function on the path, keyed on the reference's own
``state_dict`` names, each function citing the reference file:line it follows.

Pinning: ``oracle/make_golden.py`` imports the reference's real modules from
``/root/reference`` (this container only), loads the same seeded weights and
records their outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors.  The ~80 harness lines of
``generate_wav`` that cannot execute under torch 2.10 (legacy ``torch.stft``
signature, unconditional ``.cuda()``, librosa/soundfile) are restated from the
source text and pinned by an independent float64 DFT in the tests.

Parity mode: both networks in eval mode (BatchNorm uses running statistics),
as in the reference's validation loop (trainer/complex_ddpm_trainer.py:400-401).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
C_SCALE = 11.0  # reference: trainer/complex_ddpm_trainer.py:30


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _bn(sd, p, x):
    """Eval-mode BatchNorm{1,2}d (running stats, eps 1e-5)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _prelu(sd, p, x):
    return F.prelu(x, sd[p + ".weight"])


# --------------------------------------------------------------------------
# A1  schedule  (reference: trainer/complex_ddpm_trainer.py:105-156)
# --------------------------------------------------------------------------
def inference_schedule(noise_schedule, inference_noise_schedule, fast_sampling):
    training = np.array(noise_schedule)
    inference = np.array(inference_noise_schedule) if fast_sampling else training
    talpha_cum = np.cumprod(1 - training)
    beta = inference
    alpha = 1 - beta
    alpha_cum = np.cumprod(alpha)
    sigmas = [0 for _ in alpha]
    for n in range(len(alpha) - 1, -1, -1):
        sigmas[n] = ((1.0 - alpha_cum[n - 1]) / (1.0 - alpha_cum[n]) * beta[n]) ** 0.5
    T = []
    for s in range(len(inference)):
        for t in range(len(training) - 1):
            if talpha_cum[t + 1] <= alpha_cum[s] <= talpha_cum[t]:
                tw = (talpha_cum[t] ** 0.5 - alpha_cum[s] ** 0.5) / (
                    talpha_cum[t] ** 0.5 - talpha_cum[t + 1] ** 0.5)
                T.append(t + tw)
                break
    return alpha, beta, alpha_cum, sigmas, np.array(T, dtype=np.float32)


# --------------------------------------------------------------------------
# A2  front-end: RMS normalise, STFT, sqrt-compression
#     (reference: trainer/complex_ddpm_trainer.py:921-937; batched twin
#      utils/dataset.py:45-74)
# --------------------------------------------------------------------------
def rms_scale(wav):
    """c = sqrt(sum(x^2)/L) per utterance (reference :922)."""
    return torch.sqrt(torch.sum(wav * wav, dim=-1) / wav.shape[-1])


def stft_ri(wav):
    """[B,L] -> [B,2,T,F]: torch.stft(n_fft=320, hop=160, win=320, periodic hann,
    center=True, reflect) viewed as real, permuted like the reference
    (:926-930 ``.permute(2,1,0)`` on the legacy [F,T,2] layout)."""
    spec = torch.stft(wav, n_fft=320, hop_length=160, win_length=320,
                      window=torch.hann_window(320), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)  # [B,F,T]
    ri = torch.view_as_real(spec)  # [B,F,T,2]
    return ri.permute(0, 3, 2, 1).contiguous()


def compress_sqrt(ri):
    """phase = atan2(im, re); mag = |X|**0.5; [mag cos, mag sin] (reference :931-937)."""
    phase = torch.atan2(ri[:, -1], ri[:, 0])
    mag = torch.norm(ri, dim=1) ** 0.5
    return torch.stack((mag * torch.cos(phase), mag * torch.sin(phase)), dim=1)


# --------------------------------------------------------------------------
# A7  back-end: square-decompression, ISTFT
#     (reference: trainer/complex_ddpm_trainer.py:1004-1016; batched twin
#      utils/metrics.py:528-566)
# --------------------------------------------------------------------------
def decompress_square(x):
    mag, phase = torch.norm(x, dim=1), torch.atan2(x[:, -1], x[:, 0])
    mag = mag ** 2
    return torch.stack((mag * torch.cos(phase), mag * torch.sin(phase)), dim=1)


def istft_ri(ri, length):
    """[B,2,T,F] -> [B,length] (reference :1009-1015)."""
    spec = torch.complex(ri[:, 0], ri[:, 1]).permute(0, 2, 1)  # [B,F,T]
    return torch.istft(spec, n_fft=320, hop_length=160, win_length=320,
                       window=torch.hann_window(320), length=length)


# --------------------------------------------------------------------------
# A6  ε-network DiffUNet1  (reference: model/diff3.py)
# --------------------------------------------------------------------------
def build_time_table(max_steps=50):
    """reference: model/diff3.py:89-95 (fp32 torch arithmetic, sin‖cos)."""
    steps = torch.arange(max_steps).unsqueeze(1)
    dims = torch.arange(64).unsqueeze(0)
    table = steps * 10.0 ** (dims * 4.0 / 63.0)
    return torch.cat([torch.sin(table), torch.cos(table)], dim=1)


def time_embedding(sd, t, table):
    """reference: model/diff3.py:68-87 — int t indexes, float t lerps."""
    if t.dtype in (torch.int32, torch.int64):
        x = table[t]
    else:
        low_idx = torch.floor(t).long()
        high_idx = torch.ceil(t).long()
        low, high = table[low_idx], table[high_idx]
        x = low + (high - low) * (t - low_idx).unsqueeze(1)
    x = x.to(sd["time_embedding.projection1.weight"].dtype)     # float64 weights: the "exact arithmetic" runs of the tests
    x = F.linear(x, sd["time_embedding.projection1.weight"], sd["time_embedding.projection1.bias"])
    x = x * torch.sigmoid(x)
    x = F.linear(x, sd["time_embedding.projection2.weight"], sd["time_embedding.projection2.bias"])
    return x * torch.sigmoid(x)


def biconvglu(sd, p, x):
    """reference: model/diff3.py:307-326 (stride (1,2))."""
    x = F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"])
    left = F.conv2d(x, sd[p + ".l.weight"], sd[p + ".l.bias"], stride=(1, 2))
    right = F.conv2d(x, sd[p + ".r.weight"], sd[p + ".r.bias"], stride=(1, 2))
    left_mask = torch.sigmoid(F.conv2d(left, sd[p + ".l_conv.weight"], sd[p + ".l_conv.bias"]))
    right_mask = torch.sigmoid(F.conv2d(right, sd[p + ".r_conv.weight"], sd[p + ".r_conv.bias"]))
    left, right = left * right_mask, right * left_mask
    return F.conv2d(left + right, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"])


def biconvtransglu(sd, p, x, temb):
    """reference: model/diff3.py:329-351 (stride (1,2)); temb None = prior DiffUNet."""
    if temb is not None:
        tp = F.linear(temb, sd[p + ".tp.weight"], sd[p + ".tp.bias"])
        x = x + tp.unsqueeze(-1).unsqueeze(-1)
    x = F.conv_transpose2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"])
    left = F.conv_transpose2d(x, sd[p + ".l.weight"], sd[p + ".l.bias"], stride=(1, 2))
    right = F.conv_transpose2d(x, sd[p + ".r.weight"], sd[p + ".r.bias"], stride=(1, 2))
    left_mask = torch.sigmoid(F.conv_transpose2d(left, sd[p + ".l_conv.weight"], sd[p + ".l_conv.bias"]))
    right_mask = torch.sigmoid(F.conv_transpose2d(right, sd[p + ".r_conv.weight"], sd[p + ".r_conv.bias"]))
    left, right = left * right_mask, right * left_mask
    return F.conv_transpose2d(left + right, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"])


def unet_encoder(sd, x, temb):
    """reference: model/diff3.py:144-166 — pad one frame on top, add the time
    bias to the *padded* tensor (so the pad row equals the bias), BiConvGLU, BN, PReLU."""
    en_list = []
    for k in range(1, 6):
        x = F.pad(x, (0, 0, 1, 0))
        if temb is not None:
            tp = F.linear(temb, sd["en.tp%d.weight" % k], sd["en.tp%d.bias" % k])
            x = x + tp.unsqueeze(-1).unsqueeze(-1)
        x = biconvglu(sd, "en.conv%d" % k, x)
        x = _prelu(sd, "en.en%d.1" % k, _bn(sd, "en.en%d.0" % k, x))
        en_list.append(x)
    return x, en_list


def tcm_residual(sd, p, x, dilation):
    """reference: model/diff3.py:215-257."""
    t = x
    x = F.conv1d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"])

    def branch(b):
        y = _bn(sd, p + "." + b + ".1", _prelu(sd, p + "." + b + ".0", x))
        return F.conv1d(y, sd[p + "." + b + ".2.weight"], sd[p + "." + b + ".2.bias"],
                        padding=2 * dilation, dilation=dilation)

    x = branch("mainbranch") * torch.sigmoid(branch("maskbranch"))
    x = _bn(sd, p + ".conv2.1", _prelu(sd, p + ".conv2.0", x))
    x = F.conv1d(x, sd[p + ".conv2.2.weight"], sd[p + ".conv2.2.bias"])
    return x + t


def tcms(sd, x):
    """reference: model/diff3.py:260-277 — 3 × residual(d = 1,2,4,8,16,32)."""
    for i in range(3):
        for j, d in enumerate((1, 2, 4, 8, 16, 32)):
            x = tcm_residual(sd, "TCMs.%d.residual%d" % (i, j + 1), x, d)
    return x


def unet_decoder(sd, de, x, en_list, temb):
    """reference: model/diff3.py:205-212 + Chomp_T (:298-304), BN, PReLU."""
    for k in (5, 4, 3, 2, 1):
        x = torch.cat((x, en_list[k - 1]), dim=1)
        x = biconvtransglu(sd, "%s.de%d.0" % (de, k), x, temb)
        x = x[:, :, :-1, :]
        if k > 1:
            x = _prelu(sd, "%s.de%d.3" % (de, k), _bn(sd, "%s.de%d.2" % (de, k), x))
    return x


def _unet_trunk(sd, x, temb, taps=None):
    x, en_list = unet_encoder(sd, x, temb)
    b, _, t, _ = x.shape
    x = x.permute(0, 2, 1, 3).reshape(b, t, -1).permute(0, 2, 1)  # [B, c*4+f, T]
    x = tcms(sd, x).permute(0, 2, 1)
    x = x.reshape(b, t, 64, 4).permute(0, 2, 1, 3)
    if taps is not None:
        taps["en_list"] = en_list
        taps["tcm_out"] = x
    x_real = unet_decoder(sd, "de_real", x, en_list, temb)
    x_imag = unet_decoder(sd, "de_imag", x, en_list, temb)
    return torch.cat((x_real, x_imag), dim=1)


def diffunet1_forward(sd, x, x_init, t, table=None, taps=None):
    """reference: model/diff3.py:37-57."""
    if table is None:
        table = build_time_table(50)
    x = F.conv2d(torch.cat((x, x_init), dim=1), sd["preprocess.conv.weight"], sd["preprocess.conv.bias"])
    temb = time_embedding(sd, t, table)
    if taps is not None:
        taps["pre"] = x
        taps["temb"] = temb
    return _unet_trunk(sd, x, temb, taps)


def nocon_forward(sd, x, t, table=None):
    """Alt eps-net of the ``deltamu`` parameterisation (reference: model/piror_grad.py:28-40):
    DiffUNet1 without Preprocess, forward(x, t)."""
    if table is None:
        table = build_time_table(50)
    return _unet_trunk(sd, x, time_embedding(sd, t, table))


def diffunet_forward(sd, x):
    """Prior DiffUNet (reference: model/diff.py:23-33): same trunk, no time input."""
    return _unet_trunk(sd, x, None)


# --------------------------------------------------------------------------
# A3  prior GCRN  (reference: model/gcrn.py)
# --------------------------------------------------------------------------
def lstm_layer(x, w_ih, w_hh, b_ih, b_hh):
    """Single-layer batch_first nn.LSTM restated as an explicit loop
    (gate order i,f,g,o; zero initial state)."""
    b, t, _ = x.shape
    hdim = w_hh.shape[1]
    h = x.new_zeros(b, hdim)
    c = x.new_zeros(b, hdim)
    gx = F.linear(x, w_ih, b_ih)  # [B,T,4H]
    out = []
    for s in range(t):
        g = gx[:, s] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return torch.stack(out, dim=1)


def glstm(sd, x):
    """reference: model/gcrn.py:22-40 — layer 1 outputs are *interleaved*
    (stack dim=-1 + flatten), layer 2 outputs are concatenated."""
    out = x.transpose(1, 2).contiguous()
    out = out.view(out.size(0), out.size(1), -1)

    def run(layer, g, inp):
        p = "glstm.%s.%d." % (layer, g)
        return lstm_layer(inp, sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"],
                          sd[p + "bias_ih_l0"], sd[p + "bias_hh_l0"])

    ch = torch.chunk(out, 2, dim=-1)
    out = torch.stack([run("lstm_list1", g, ch[g]) for g in range(2)], dim=-1)
    out = torch.flatten(out, start_dim=-2, end_dim=-1)
    out = F.layer_norm(out, (1024,), sd["glstm.ln1.weight"], sd["glstm.ln1.bias"], 1e-5)
    ch = torch.chunk(out, 2, dim=-1)
    out = torch.cat([run("lstm_list2", g, ch[g]) for g in range(2)], dim=-1)
    out = F.layer_norm(out, (1024,), sd["glstm.ln2.weight"], sd["glstm.ln2.bias"], 1e-5)
    out = out.view(out.size(0), out.size(1), x.size(1), -1)
    return out.transpose(1, 2).contiguous()


def _glu_conv(sd, p, x):
    """reference: model/gcrn.py:43-61 (kernel (1,3), stride (1,2))."""
    a = F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=(1, 2))
    g = torch.sigmoid(F.conv2d(x, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], stride=(1, 2)))
    return a * g


def _glu_convT(sd, p, x, output_padding=(0, 0)):
    """reference: model/gcrn.py:64-84."""
    a = F.conv_transpose2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=(1, 2),
                           output_padding=output_padding)
    g = torch.sigmoid(F.conv_transpose2d(x, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"],
                                         stride=(1, 2), output_padding=output_padding))
    return a * g


def gcrn_forward(sd, x, taps=None):
    """reference: model/gcrn.py:136-166.  ELU is re-applied to the skip tensors
    inside every ``elu(cat(...))`` exactly as the reference does."""
    e = []
    out = x
    for k in range(1, 6):
        out = F.elu(_bn(sd, "bn%d" % k, _glu_conv(sd, "conv%d" % k, out)))
        e.append(out)
    e1, e2, e3, e4, e5 = e
    out = glstm(sd, e5)
    if taps is not None:
        taps["e5"] = e5
        taps["glstm"] = out
    out = torch.cat((out, e5), dim=1)
    res = []
    for br in (1, 2):
        d = out
        for k, skip in ((5, e4), (4, e3), (3, e2), (2, e1)):
            op = (0, 1) if k == 2 else (0, 0)
            d = _bn(sd, "bn%d_t_%d" % (k, br), _glu_convT(sd, "conv%d_t_%d" % (k, br), d, op))
            d = F.elu(torch.cat((d, skip), dim=1))
        d = F.elu(_bn(sd, "bn1_t_%d" % br, _glu_convT(sd, "conv1_t_%d" % br, d)))
        res.append(F.linear(d, sd["fc%d.weight" % br], sd["fc%d.bias" % br]))
    return torch.cat(res, dim=1)


PRIORS = {"GCRN": gcrn_forward, "DiffUNet": diffunet_forward}


# --------------------------------------------------------------------------
# A4 + A5  x_T scaling and the reverse loop
#     (reference: trainer/complex_ddpm_trainer.py:939-998; batched twin :441-494)
# --------------------------------------------------------------------------
def sigma_mask(init):
    """reference :951-956 — per-(b,ch) |init|/max/2 + 0.5."""
    tmp = torch.flatten(torch.abs(init), start_dim=2)
    tmp = tmp / torch.max(tmp, dim=2, keepdim=True).values
    tmp = tmp / 2 + 0.5
    return tmp.view(init.shape)


def reverse_loop(ddpm_sd, init_scaled, x_T, alpha, beta, alpha_cum, sigmas, T, table=None,
                 use_sigma=False, trace=None, deltamu=False, feat_scaled=None, xT_plus_init=None):
    """Runs n = S-1 … 0 on ``audio = x_T`` with ``init_scaled = X_init / 11``.
    feat_scaled (noisy feature / 11): the branch with neither ``pirorgrad`` nor ``deltamu`` set — DiffUNet1 is
    conditioned on it instead of X_init (reference :972-974).

    The n>0 noise term is kept for fidelity: ``newsigma = max(0, σ - c1σ)`` is
    identically 0 (reference :986-992), so no RNG draw changes the result.
    """
    if table is None:
        table = build_time_table(50)
    if xT_plus_init is None:                                   # :946-949 tests ``self.deltamu`` alone: with pirorgrad AND
        xT_plus_init = deltamu                                 # deltamu set, DiffUNet1 starts from noise + X_init/11
    audio = x_T + init_scaled if xT_plus_init else x_T.clone()  # reference :947-950
    if use_sigma:
        mask = sigma_mask(init_scaled)
        audio = audio * (mask ** 0.5)
    N = audio.shape[0]
    gamma = [sigmas[n] for n in range(len(alpha))]
    gamma[0] = 0.2
    for n in range(len(alpha) - 1, -1, -1):
        c1 = 1 / alpha[n] ** 0.5
        c2 = beta[n] / (1 - alpha_cum[n]) ** 0.5
        tn = torch.tensor([T[n]]).repeat(N)
        if deltamu:
            eps = nocon_forward(ddpm_sd, audio, tn, table)
        else:
            eps = diffunet1_forward(ddpm_sd, audio, init_scaled if feat_scaled is None else feat_scaled, tn, table)
        audio = float(c1) * (audio - float(c2) * eps)
        if n > 0:
            newsigma = max(0, gamma[n] - c1 * gamma[n])
            assert newsigma == 0
        if trace is not None:
            trace.append(audio.clone())
    return audio


def sample(prior_name, prior_sd, ddpm_sd, feat, x_T, noise_schedule, inference_noise_schedule,
           fast_sampling=True, use_sigma=False, trace=None, deltamu=False, cond="init", xT_plus_init=None):
    """feat [B,2,T,F] (compressed spectrogram) -> enhanced compressed spectrogram.

    reference: trainer/complex_ddpm_trainer.py:941-998.  cond "feat": neither pirorgrad nor deltamu (:972-974),
    DiffUNet1 conditioned on feat / 11 and no final ``+ X_init`` (:994-995).
    """
    alpha, beta, alpha_cum, sigmas, T = inference_schedule(noise_schedule, inference_noise_schedule,
                                                           fast_sampling)
    init = PRIORS[prior_name](prior_sd, feat)
    init = init / C_SCALE
    feat_cond = cond == "feat" and not deltamu
    audio = reverse_loop(ddpm_sd, init, x_T, alpha, beta, alpha_cum, sigmas, T, use_sigma=use_sigma,
                         trace=trace, deltamu=deltamu, feat_scaled=feat / C_SCALE if feat_cond else None,
                         xT_plus_init=xT_plus_init)
    if not (deltamu or feat_cond):   # ``if self.pirorgrad: audio += init_audio`` (reference :995-996)
        audio = audio + init
    audio = audio * C_SCALE
    return audio, init * C_SCALE


def enhance(prior_name, prior_sd, ddpm_sd, wav, x_T, noise_schedule, inference_noise_schedule,
            fast_sampling=True, use_sigma=False):
    """wav [B,L] -> enhanced wav [B,L]: the whole of generate_wav's per-file body,
    batched (reference: trainer/complex_ddpm_trainer.py:921-1016)."""
    c = rms_scale(wav)
    feat = compress_sqrt(stft_ri(wav / c[:, None]))
    spec, _ = sample(prior_name, prior_sd, ddpm_sd, feat, x_T, noise_schedule,
                     inference_noise_schedule, fast_sampling, use_sigma)
    out = istft_ri(decompress_square(spec), wav.shape[-1])
    return out * c[:, None], spec


# --------------------------------------------------------------------------
# A3''  prior aia_complex_trans_ri  (reference: model/dbaiat.py)
# --------------------------------------------------------------------------
def _ln_last(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def dense_block(sd, p, x):
    """reference: model/dbaiat.py:605-631 — 4 dilated (2,3) convs, LayerNorm over F, per-channel
    PReLU, dense concatenation [newest, ..., input]; causal zero padding of dil frames on top,
    one bin left/right."""
    skip = x
    out = x
    for i in range(4):
        dil = 2 ** i
        out = F.pad(skip, (1, 1, dil, 0))
        out = F.conv2d(out, sd["%s.conv%d.weight" % (p, i + 1)], sd["%s.conv%d.bias" % (p, i + 1)], dilation=(dil, 1))
        out = _ln_last(sd, "%s.norm%d" % (p, i + 1), out)
        out = F.prelu(out, sd["%s.prelu%d.weight" % (p, i + 1)])
        skip = torch.cat([out, skip], dim=1)
    return out


def dense_encoder(sd, x, p="en_ri"):
    """reference: model/dbaiat.py:481-501 (dense_encoder) and :504-524 (dense_encoder_mag: one input channel)."""
    out = F.conv2d(x, sd[p + ".inp_conv.weight"], sd[p + ".inp_conv.bias"])
    out = F.prelu(_ln_last(sd, p + ".inp_norm", out), sd[p + ".inp_prelu.weight"])
    out = dense_block(sd, p + ".enc_dense1", out)
    out = F.conv2d(out, sd[p + ".enc_conv1.weight"], sd[p + ".enc_conv1.bias"], stride=(1, 2))
    return F.prelu(_ln_last(sd, p + ".enc_norm1", out), sd[p + ".enc_prelu1.weight"])


def gru_layer(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Single-direction nn.GRU layer, seq-first [S,N,I], zero initial state, gate order r,z,n:
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1-z) n + z h."""
    S, N, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(N, H)
    gx = F.linear(x, w_ih, b_ih)
    outs = [None] * S
    order = range(S - 1, -1, -1) if reverse else range(S)
    for s in order:
        gh = F.linear(h, w_hh, b_hh)
        xr, xz, xn = gx[s].chunk(3, dim=1)
        hr, hz, hn = gh.chunk(3, dim=1)
        r = torch.sigmoid(xr + hr)
        z = torch.sigmoid(xz + hz)
        n = torch.tanh(xn + r * hn)
        h = (1 - z) * n + z * h
        outs[s] = h
    return torch.stack(outs, dim=0)


def mha(sd, p, x, heads=4):
    """nn.MultiheadAttention self-attention, seq-first [S,N,E], no mask, no dropout."""
    S, N, E = x.shape
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    hd = E // heads

    def split(t):
        return t.reshape(S, N * heads, hd).transpose(0, 1)          # [N*heads, S, hd]

    q, k, v = split(q) * (hd ** -0.5), split(k), split(v)
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    out = torch.bmm(att, v).transpose(0, 1).reshape(S, N, E)
    return F.linear(out, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def aia_encoder_layer(sd, p, src):
    """reference: model/dbaiat.py:66-88 (TransformerEncoderLayer.forward)."""
    src_norm = _ln_last(sd, p + ".norm3", src)
    src = src + mha(sd, p + ".self_attn", src_norm)
    src = _ln_last(sd, p + ".norm1", src)
    g = p + ".gru."
    fw = gru_layer(src, sd[g + "weight_ih_l0"], sd[g + "weight_hh_l0"], sd[g + "bias_ih_l0"], sd[g + "bias_hh_l0"])
    bw = gru_layer(src, sd[g + "weight_ih_l0_reverse"], sd[g + "weight_hh_l0_reverse"],
                   sd[g + "bias_ih_l0_reverse"], sd[g + "bias_hh_l0_reverse"], reverse=True)
    out = torch.cat([fw, bw], dim=-1)
    src2 = F.linear(F.relu(out), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return _ln_last(sd, p + ".norm2", src + src2)


def aia_transformer(sd, x, taps=None):
    """reference: model/dbaiat.py:133-154 — 4 x (row over F, col over T), shared output head."""
    p = "dual_trans"
    b, c, dim2, dim1 = x.shape
    output = F.prelu(F.conv2d(x, sd[p + ".input.0.weight"], sd[p + ".input.0.bias"]), sd[p + ".input.1.weight"])
    outs = []
    for i in range(4):
        row_in = output.permute(3, 0, 2, 1).contiguous().view(dim1, b * dim2, -1)
        row = aia_encoder_layer(sd, "%s.row_trans.%d" % (p, i), row_in)
        row = row.view(dim1, b, dim2, -1).permute(1, 3, 2, 0).contiguous()
        row = F.group_norm(row, 1, sd["%s.row_norm.%d.weight" % (p, i)], sd["%s.row_norm.%d.bias" % (p, i)], 1e-8)
        col_in = output.permute(2, 0, 3, 1).contiguous().view(dim2, b * dim1, -1)
        col = aia_encoder_layer(sd, "%s.col_trans.%d" % (p, i), col_in)
        col = col.view(dim2, b, dim1, -1).permute(1, 3, 0, 2).contiguous()
        col = F.group_norm(col, 1, sd["%s.col_norm.%d.weight" % (p, i)], sd["%s.col_norm.%d.bias" % (p, i)], 1e-8)
        if taps is not None and i == 0:
            taps["row0"], taps["col0"] = row, col
        output = output + sd[p + ".k1"] * row + sd[p + ".k2"] * col
        outs.append(F.conv2d(F.prelu(output, sd[p + ".output.0.weight"]), sd[p + ".output.1.weight"],
                             sd[p + ".output.1.bias"]))
    return outs


def aham(sd, outs, p="aham"):
    """reference: model/dbaiat.py:266-288 (AHAM) / :308-330 (AHAM_ori, same arithmetic) — softmax over the 4 layer
    outputs of conv1(avgpool(x_i))."""
    ys = [F.conv2d(o.mean(dim=(2, 3), keepdim=True), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]) for o in outs]
    w = torch.softmax(torch.cat(ys, dim=1), dim=1)                  # [B,4,1,1]
    merged = sum(w[:, i:i + 1] * outs[i] for i in range(4))
    return outs[-1] + merged


def dense_decoder(sd, p, x):
    """reference: model/dbaiat.py:527-548 + SPConvTranspose2d :587-602 (sub-pixel, r = 2)."""
    out = dense_block(sd, p + ".dec_dense1", x)
    out = F.conv2d(F.pad(out, (1, 1, 0, 0)), sd[p + ".dec_conv1.conv.weight"], sd[p + ".dec_conv1.conv.bias"])
    bsz, nch, H, W = out.shape
    out = out.view(bsz, 2, nch // 2, H, W).permute(0, 2, 3, 4, 1).contiguous().view(bsz, nch // 2, H, -1)
    out = F.pad(out, (1, 0, 0, 0))
    out = F.prelu(_ln_last(sd, p + ".dec_norm1", out), sd[p + ".dec_prelu1.weight"])
    return F.conv2d(out, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])


def aia_complex_trans_ri_forward(sd, x, taps=None):
    """reference: model/dbaiat.py:461-478."""
    x_ri = dense_encoder(sd, x)
    outs = aia_transformer(sd, x_ri, taps)
    merged = aham(sd, outs)
    if taps is not None:
        taps["en_ri"], taps["trans_last"], taps["aham"] = x_ri, outs[-1], merged
    real = dense_decoder(sd, "de1", merged)
    imag = dense_decoder(sd, "de2", merged)
    return torch.cat((real, imag), dim=1)


PRIORS["aia_complex_trans_ri"] = aia_complex_trans_ri_forward


# --------------------------------------------------------------------------
# A3''  dual-branch prior dual_aia_trans_merge_crm  (reference: model/dbaiat.py:373-413)
# --------------------------------------------------------------------------
def aia_transformer_merge(sd, x_mag, x_ri, p="aia_trans_merge"):
    """reference: model/dbaiat.py:200-246 (AIA_Transformer_merge.forward), both branches as written there:
    layer i of either branch reads, for i >= 1, the SUM of the two branches' previous outputs, and adds its
    normalised row/column results to the shared input projection (not to a running state)."""
    b, c, dim2, dim1 = x_mag.shape
    merge = torch.cat((x_mag, x_ri), dim=1)
    inp = F.prelu(F.conv2d(merge, sd[p + ".input.0.weight"], sd[p + ".input.0.bias"]), sd[p + ".input.1.weight"])
    input_mag, input_ri = inp, inp                                   # the same module applied to the same tensor (:206-207)

    def layer(i, u, base):
        row_in = u.permute(3, 0, 2, 1).contiguous().view(dim1, b * dim2, -1)
        row = aia_encoder_layer(sd, "%s.row_trans.%d" % (p, i), row_in)
        row = row.view(dim1, b, dim2, -1).permute(1, 3, 2, 0).contiguous()
        row = F.group_norm(row, 1, sd["%s.row_norm.%d.weight" % (p, i)], sd["%s.row_norm.%d.bias" % (p, i)], 1e-8)
        col_in = u.permute(2, 0, 3, 1).contiguous().view(dim2, b * dim1, -1)
        col = aia_encoder_layer(sd, "%s.col_trans.%d" % (p, i), col_in)
        col = col.view(dim2, b, dim1, -1).permute(1, 3, 0, 2).contiguous()
        col = F.group_norm(col, 1, sd["%s.col_norm.%d.weight" % (p, i)], sd["%s.col_norm.%d.bias" % (p, i)], 1e-8)
        o = base + sd[p + ".k1"] * row + sd[p + ".k2"] * col
        return F.conv2d(F.prelu(o, sd[p + ".output.0.weight"]), sd[p + ".output.1.weight"], sd[p + ".output.1.bias"])

    list_mag, list_ri = [], []
    for i in range(4):
        u_mag = input_mag if i == 0 else list_mag[-1] + list_ri[-1]
        list_mag.append(layer(i, u_mag, input_mag))
        u_ri = input_ri if i == 0 else list_ri[-1] + list_mag[-2]
        list_ri.append(layer(i, u_ri, input_ri))
    return list_mag, list_ri


def dense_decoder_masking(sd, p, x):
    """reference: model/dbaiat.py:551-584 — dense block, sub-pixel up-convolution, LayerNorm/PReLU, 64->1, then the
    scalar gate sigmoid(maskconv(sigmoid(mask1(o)) * tanh(mask2(o))))."""
    out = dense_block(sd, p + ".dec_dense1", x)
    out = F.conv2d(F.pad(out, (1, 1, 0, 0)), sd[p + ".dec_conv1.conv.weight"], sd[p + ".dec_conv1.conv.bias"])
    bsz, nch, H, W = out.shape
    out = out.view(bsz, 2, nch // 2, H, W).permute(0, 2, 3, 4, 1).contiguous().view(bsz, nch // 2, H, -1)
    out = F.pad(out, (1, 0, 0, 0))
    out = F.prelu(_ln_last(sd, p + ".dec_norm1", out), sd[p + ".dec_prelu1.weight"])
    out = F.conv2d(out, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])
    m1 = torch.sigmoid(F.conv2d(out, sd[p + ".mask1.0.weight"], sd[p + ".mask1.0.bias"]))
    m2 = torch.tanh(F.conv2d(out, sd[p + ".mask2.0.weight"], sd[p + ".mask2.0.bias"]))
    return torch.sigmoid(F.conv2d(m1 * m2, sd[p + ".maskconv.weight"], sd[p + ".maskconv.bias"]))


def dual_aia_trans_merge_crm_forward(sd, x, taps=None):
    """reference: model/dbaiat.py:386-413."""
    x_mag_ori = torch.norm(x, dim=1)
    x_phase_ori = torch.atan2(x[:, -1], x[:, 0])
    x_ri = dense_encoder(sd, x, "en_ri")
    x_mag_en = dense_encoder(sd, x_mag_ori.unsqueeze(1), "en_mag")
    list_mag, list_ri = aia_transformer_merge(sd, x_mag_en, x_ri)
    m_ri = aham(sd, list_ri, "aham")
    m_mag = aham(sd, list_mag, "aham_mag")
    mask = dense_decoder_masking(sd, "de_mag_mask", m_mag).squeeze(1)
    real = dense_decoder(sd, "de1", m_ri).squeeze(1)
    imag = dense_decoder(sd, "de2", m_ri).squeeze(1)
    if taps is not None:
        taps.update(en_ri=x_ri, en_mag=x_mag_en, trans_last=list_ri[-1], aham=m_ri, aham_mag=m_mag, mask=mask)
    x_mag_out = mask * x_mag_ori
    return torch.stack((x_mag_out * torch.cos(x_phase_ori) + real, x_mag_out * torch.sin(x_phase_ori) + imag), dim=1)


PRIORS["dual_aia_trans_merge_crm"] = dual_aia_trans_merge_crm_forward


def enhance_ragged(prior_name, prior_sd, ddpm_sd, wavs, x_T, noise_schedule, inference_noise_schedule,
                   fast_sampling=True, use_sigma=False):
    """The batched twin of generate_wav used by the validation loop: per-utterance c = sqrt(len / sum x^2)
    BEFORE zero-padding (utils/dataset.py:45-58), batched STFT (:59-74), sampling (:441-494), per-utterance
    ISTFT cut to (frame_num - 1) * 160 samples with frame_num = len // 160 + 1 (utils/metrics.py:553-563,
    utils/dataset.py:101).  Returns the list of trimmed waveforms, rescaled by 1/c like generate_wav."""
    lens = [int(w.numel()) for w in wavs]
    cs = [float(np.sqrt(n / float((w.double() ** 2).sum()))) for w, n in zip(wavs, lens)]
    batch = torch.nn.utils.rnn.pad_sequence([w * c for w, c in zip(wavs, cs)], batch_first=True)
    feat = compress_sqrt(stft_ri(batch))
    spec, _ = sample(prior_name, prior_sd, ddpm_sd, feat, x_T, noise_schedule, inference_noise_schedule,
                     fast_sampling, use_sigma)
    com = decompress_square(spec)
    outs = []
    for i, n in enumerate(lens):
        z = torch.complex(com[i, 0], com[i, 1]).permute(1, 0)[None]
        y = torch.istft(z, n_fft=320, hop_length=160, win_length=320, window=torch.hann_window(320))[0]
        outs.append(y[: (n // 160) * 160] / cs[i])
    return outs


def com_mse_loss(esti, label, frame_list):
    """reference: utils/loss.py:34-44 — complex MSE over the frames each utterance really has (the zero padding
    of a ragged batch does not count)."""
    mask = torch.zeros_like(esti)
    for i, n in enumerate(frame_list):
        mask[i, :, :n] = 1.0
    return (((esti - label) * mask) ** 2).sum() / mask.sum()


def q_sample(label, init, t, noise, noise_schedule, mode="pirorgrad", sigma=False):
    """reference: trainer/complex_ddpm_trainer.py:42-44, :704-729 (label / init already divided by 11)."""
    noise_level = torch.tensor(np.cumprod(1 - np.array(noise_schedule)).astype(np.float32))
    ns = noise_level[t].unsqueeze(1).unsqueeze(2).unsqueeze(3)
    if sigma:
        noise = noise * sigma_mask(init) ** 0.5
    if mode == "pirorgrad":
        return ns ** 0.5 * (label - init) + (1.0 - ns) ** 0.5 * noise
    if mode == "deltamu":
        return ns ** 0.5 * label + (1.0 - ns) ** 0.5 * (noise + init)
    return ns ** 0.5 * label + (1.0 - ns) ** 0.5 * noise
