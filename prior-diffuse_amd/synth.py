"""Seeded synthetic weights and inputs for the sampling path.

No trained checkpoint ships with the reference, so every parity test, the
bench and the smoke run use *seeded, gain-rescaled* random weights keyed on the
reference's own ``state_dict`` names and shapes:

  * ε-network ``DiffUNet1``   (reference: model/diff3.py:14-57, 762 tensors)
  * prior ``GCRN``            (reference: model/gcrn.py:87-166, 159 tensors)
  * prior ``DiffUNet``        (reference: model/diff.py:13-33)

Default ``nn.Module`` init attenuates the deep path (encoder 2-5 → TCM →
decoder 5-2) below fp32 epsilon at the output, so goldens from it would not
exercise most kernels.  Here each tensor is drawn with a fan-in gain that keeps
activations O(1) through every block, BatchNorm running statistics, affine
parameters and PReLU slopes are randomised, and everything comes from numpy's
PCG64 (stable across platforms) so both sides regenerate identical weights
from ``(arch, seed)`` alone — the fixtures never store weights.

Names/shapes are written out here by hand; ``oracle/make_golden.py`` checks
them by a *strict* ``load_state_dict`` into the reference modules.
"""
from collections import OrderedDict

import numpy as np

F_BINS = 161


# --------------------------------------------------------------------------
# architecture tables: ordered (key, shape, kind[, fan_in])
# --------------------------------------------------------------------------
def _bn(prefix, c):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_mean"),
        (prefix + ".running_var", (c,), "bn_var"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def _conv(prefix, shape, fan_in, gain=1.0):
    return [
        (prefix + ".weight", shape, "w", fan_in, gain),
        (prefix + ".bias", (shape[0],), "bias"),
    ]


def _convT(prefix, shape, fan_in, gain=1.0):
    # ConvTranspose weights are [Cin, Cout, kh, kw]; bias has Cout entries
    return [
        (prefix + ".weight", shape, "w", fan_in, gain),
        (prefix + ".bias", (shape[1],), "bias"),
    ]


def _biconvglu(prefix, cin, cout, kw):
    # reference: model/diff3.py:307-326
    out = []
    out += _conv(prefix + ".conv1", (32, cin, 1, 1), cin, 1.0)
    out += _conv(prefix + ".l", (32, 32, 2, kw), 32 * 2 * kw, 1.2)
    out += _conv(prefix + ".l_conv", (32, 32, 1, 1), 32, 1.5)
    out += _conv(prefix + ".r", (32, 32, 2, kw), 32 * 2 * kw, 1.2)
    out += _conv(prefix + ".r_conv", (32, 32, 1, 1), 32, 1.5)
    out += _conv(prefix + ".conv2", (cout, 32, 1, 1), 32, 1.4)
    return out


def _biconvtransglu(prefix, cin, cout, kw, with_tp=True, out_gain=1.4):
    # reference: model/diff3.py:329-351 (key order: tp, conv1, l, l_conv, r_conv, r, conv2)
    out = []
    if with_tp:
        out += _conv(prefix + ".tp", (cin, 512), 512, 0.5)
    out += _convT(prefix + ".conv1", (cin, 32, 1, 1), cin, 1.0)
    # stride-2 transposed conv: each output sees ~kh*kw/2 taps per input channel
    out += _convT(prefix + ".l", (32, 32, 2, kw), 32 * 2 * kw / 2.0, 1.2)
    out += _convT(prefix + ".l_conv", (32, 32, 1, 1), 32, 1.5)
    out += _convT(prefix + ".r_conv", (32, 32, 1, 1), 32, 1.5)
    out += _convT(prefix + ".r", (32, 32, 2, kw), 32 * 2 * kw / 2.0, 1.2)
    out += _convT(prefix + ".conv2", (32, cout, 1, 1), 32, out_gain)
    return out


def _residual(prefix):
    # reference: model/diff3.py:215-257
    out = []
    out += _conv(prefix + ".conv1", (64, 256, 1), 256, 1.0)
    for br in ("mainbranch", "maskbranch"):
        out += [(prefix + "." + br + ".0.weight", (1,), "prelu")]
        out += _bn(prefix + "." + br + ".1", 64)
        out += _conv(prefix + "." + br + ".2", (64, 64, 5), 64 * 5, 1.4)
    out += [(prefix + ".conv2.0.weight", (1,), "prelu")]
    out += _bn(prefix + ".conv2.1", 64)
    out += _conv(prefix + ".conv2.2", (256, 64, 1), 64, 0.6)
    return out


def _unet_body(time_cond):
    spec = []
    # Encoder (reference: model/diff3.py:105-166)
    cins = [2, 64, 64, 64, 64]
    kws = [5, 3, 3, 3, 3]
    for i in range(5):
        spec += _biconvglu("en.conv%d" % (i + 1), cins[i], 64, kws[i])
    if time_cond:
        for i in range(5):
            spec += _conv("en.tp%d" % (i + 1), (cins[i], 512), 512, 0.5)
    for i in range(5):
        spec += _bn("en.en%d.0" % (i + 1), 64)
        spec += [("en.en%d.1.weight" % (i + 1), (1,), "prelu")]
    # Decoders (reference: model/diff3.py:169-212)
    for de in ("de_real", "de_imag"):
        for k in (5, 4, 3, 2, 1):
            cout = 64 if k > 1 else 1
            kw = 3 if k > 1 else 5
            # the last stage is scaled down so the network is a mild (not chaotic) map:
            # with unit gain a 1e-4 input perturbation grows to 40 % over 6 reverse steps
            # and no fp32 implementation could be compared with another
            spec += _biconvtransglu("%s.de%d.0" % (de, k), 128, cout, kw, with_tp=time_cond,
                                    out_gain=1.4 if k > 1 else 0.21)
            if k > 1:
                spec += _bn("%s.de%d.2" % (de, k), 64)
                spec += [("%s.de%d.3.weight" % (de, k), (1,), "prelu")]
    # TCMs (reference: model/diff3.py:260-277)
    for i in range(3):
        for j in range(1, 7):
            spec += _residual("TCMs.%d.residual%d" % (i, j))
    return spec


def diffunet1_spec():
    """Ordered (key, shape, kind, ...) of ``DiffUNet1.state_dict()``."""
    spec = []
    spec += _conv("preprocess.conv", (2, 4, 1, 1), 4, 1.0)
    spec += _conv("time_embedding.projection1", (512, 128), 128, 1.4)
    spec += _conv("time_embedding.projection2", (512, 512), 512, 1.4)
    spec += _unet_body(time_cond=True)
    return spec


def nocon_spec():
    """Ordered spec of ``Nocon`` (reference: model/piror_grad.py:15-40): DiffUNet1 without Preprocess."""
    return [e for e in diffunet1_spec() if not e[0].startswith("preprocess.")]


def diffunet_spec():
    """Ordered spec of the prior ``DiffUNet`` (reference: model/diff.py:13-33)."""
    return _unet_body(time_cond=False)


def gcrn_spec():
    """Ordered spec of ``GCRN.state_dict()`` (reference: model/gcrn.py:87-134)."""
    spec = []
    enc = [(2, 16), (16, 32), (32, 64), (64, 128), (128, 256)]
    for i, (ci, co) in enumerate(enc):
        for c in ("conv1", "conv2"):
            spec += _conv("conv%d.%s" % (i + 1, c), (co, ci, 1, 3), ci * 3, 1.6)
    for layer in ("lstm_list1", "lstm_list2"):
        for g in range(2):
            p = "glstm.%s.%d" % (layer, g)
            spec += [
                (p + ".weight_ih_l0", (2048, 512), "w", 512, 1.0),
                (p + ".weight_hh_l0", (2048, 512), "w", 512, 1.0),
                (p + ".bias_ih_l0", (2048,), "bias"),
                (p + ".bias_hh_l0", (2048,), "bias"),
            ]
    for ln in ("glstm.ln1", "glstm.ln2"):
        spec += [(ln + ".weight", (1024,), "bn_w"), (ln + ".bias", (1024,), "bn_b")]
    dec = [(5, 512, 128), (4, 256, 64), (3, 128, 32), (2, 64, 16), (1, 32, 1)]
    for br in (1, 2):
        for k, ci, co in dec:
            for c in ("conv1", "conv2"):
                spec += _convT("conv%d_t_%d.%s" % (k, br, c), (ci, co, 1, 3), ci * 3 / 2.0, 1.6)
    for i, (_, co) in enumerate(enc):
        spec += _bn("bn%d" % (i + 1), co)
    for br in (1, 2):
        for k, _, co in dec:
            spec += _bn("bn%d_t_%d" % (k, br), co)
    for fc in ("fc1", "fc2"):
        spec += _conv(fc, (161, 161), 161, 1.0)
    return spec


def _dense_block(prefix, width_f):
    # reference: model/dbaiat.py:605-631 (depth 4, dilations 1,2,4,8 along time)
    out = []
    for i in range(1, 5):
        out += _conv("%s.conv%d" % (prefix, i), (64, 64 * i, 2, 3), 64 * i * 6, 1.5)
        out += [("%s.norm%d.weight" % (prefix, i), (width_f,), "bn_w"), ("%s.norm%d.bias" % (prefix, i), (width_f,), "bn_b")]
        out += [("%s.prelu%d.weight" % (prefix, i), (64,), "prelu")]
    return out


def _aia_layer(prefix, d=32):
    # reference: model/dbaiat.py:41-88 (TransformerEncoderLayer, d_model d, 4 heads, biGRU d -> 2d)
    out = [
        (prefix + ".self_attn.in_proj_weight", (3 * d, d), "w", d, 1.5),
        (prefix + ".self_attn.in_proj_bias", (3 * d,), "bias"),
    ]
    out += _conv(prefix + ".self_attn.out_proj", (d, d), d, 1.0)
    for suf in ("", "_reverse"):
        out += [
            (prefix + ".gru.weight_ih_l0" + suf, (6 * d, d), "w", d, 1.5),
            (prefix + ".gru.weight_hh_l0" + suf, (6 * d, 2 * d), "w", 2 * d, 1.5),
            (prefix + ".gru.bias_ih_l0" + suf, (6 * d,), "bias"),
            (prefix + ".gru.bias_hh_l0" + suf, (6 * d,), "bias"),
        ]
    out += _conv(prefix + ".linear2", (d, 4 * d), 4 * d, 1.5)
    for n in (1, 2, 3):
        out += [(prefix + ".norm%d.weight" % n, (d,), "bn_w"), (prefix + ".norm%d.bias" % n, (d,), "bn_b")]
    return out


def _dense_encoder(p, cin):
    # reference: model/dbaiat.py:481-501 (cin 2) / :504-524 (cin 1)
    spec = _conv(p + ".inp_conv", (64, cin, 1, 1), cin, 1.0)
    spec += [(p + ".inp_norm.weight", (161,), "bn_w"), (p + ".inp_norm.bias", (161,), "bn_b"),
             (p + ".inp_prelu.weight", (64,), "prelu")]
    spec += _dense_block(p + ".enc_dense1", 161)
    spec += _conv(p + ".enc_conv1", (64, 64, 1, 3), 64 * 3, 1.5)
    spec += [(p + ".enc_norm1.weight", (80,), "bn_w"), (p + ".enc_norm1.bias", (80,), "bn_b"),
             (p + ".enc_prelu1.weight", (64,), "prelu")]
    return spec


def _dense_decoder(de):
    spec = _dense_block(de + ".dec_dense1", 80)
    spec += _conv(de + ".dec_conv1.conv", (128, 64, 1, 3), 64 * 3, 1.5)
    spec += [(de + ".dec_norm1.weight", (161,), "bn_w"), (de + ".dec_norm1.bias", (161,), "bn_b"),
             (de + ".dec_prelu1.weight", (64,), "prelu")]
    return spec


def dual_aia_trans_merge_crm_spec():
    """Ordered spec of the dual-branch DB-AIAT prior (reference: model/dbaiat.py:373-413): ri + magnitude encoders,
    AIA_Transformer_merge (d_model 64, :157-246), two AHAM_ori, two dense decoders and the masking decoder (:551-584)."""
    spec = _dense_encoder("en_ri", 2) + _dense_encoder("en_mag", 1)
    p = "aia_trans_merge"
    spec += [(p + ".k1", (1,), "gain1"), (p + ".k2", (1,), "gain1")]
    spec += _conv(p + ".input.0", (64, 128, 1, 1), 128, 1.4)
    spec += [(p + ".input.1.weight", (1,), "prelu")]
    for kind in ("row_trans", "col_trans"):
        for i in range(4):
            spec += _aia_layer("%s.%s.%d" % (p, kind, i), 64)
    for kind in ("row_norm", "col_norm"):
        for i in range(4):
            spec += [("%s.%s.%d.weight" % (p, kind, i), (64,), "bn_w"), ("%s.%s.%d.bias" % (p, kind, i), (64,), "bn_b")]
    spec += [(p + ".output.0.weight", (1,), "prelu")]
    spec += _conv(p + ".output.1", (64, 64, 1, 1), 64, 0.7)
    for ah in ("aham", "aham_mag"):
        spec += [(ah + ".k3", (1,), "gain1")]
        spec += _conv(ah + ".conv1", (1, 64, 1, 1), 64, 4.0)
    for de in ("de1", "de2"):
        spec += _dense_decoder(de)
        spec += _conv(de + ".out_conv", (1, 64, 1, 1), 64, 1.0)
    de = "de_mag_mask"
    spec += _dense_decoder(de)
    for m in ("mask1.0", "mask2.0", "maskconv"):
        spec += _conv("%s.%s" % (de, m), (1, 1, 1, 1), 1, 1.0)
    spec += _conv(de + ".out_conv", (1, 64, 1, 1), 64, 1.0)
    return spec


def aia_complex_trans_ri_spec():
    """Ordered spec of the DB-AIAT prior selected by conf/dbaiat.yml:13
    (reference: model/dbaiat.py:450-478 and the blocks it names)."""
    spec = []
    spec += _conv("en_ri.inp_conv", (64, 2, 1, 1), 2, 1.0)
    spec += [("en_ri.inp_norm.weight", (161,), "bn_w"), ("en_ri.inp_norm.bias", (161,), "bn_b"),
             ("en_ri.inp_prelu.weight", (64,), "prelu")]
    spec += _dense_block("en_ri.enc_dense1", 161)
    spec += _conv("en_ri.enc_conv1", (64, 64, 1, 3), 64 * 3, 1.5)
    spec += [("en_ri.enc_norm1.weight", (80,), "bn_w"), ("en_ri.enc_norm1.bias", (80,), "bn_b"),
             ("en_ri.enc_prelu1.weight", (64,), "prelu")]
    spec += [("dual_trans.k1", (1,), "gain1"), ("dual_trans.k2", (1,), "gain1")]
    spec += _conv("dual_trans.input.0", (32, 64, 1, 1), 64, 1.4)
    spec += [("dual_trans.input.1.weight", (1,), "prelu")]
    for kind in ("row_trans", "col_trans"):
        for i in range(4):
            spec += _aia_layer("dual_trans.%s.%d" % (kind, i))
    for kind in ("row_norm", "col_norm"):
        for i in range(4):
            spec += [("dual_trans.%s.%d.weight" % (kind, i), (32,), "bn_w"),
                     ("dual_trans.%s.%d.bias" % (kind, i), (32,), "bn_b")]
    spec += [("dual_trans.output.0.weight", (1,), "prelu")]
    spec += _conv("dual_trans.output.1", (64, 32, 1, 1), 32, 1.4)
    spec += [("aham.k3", (1,), "gain1")]
    spec += _conv("aham.conv1", (1, 64, 1, 1), 64, 4.0)
    for de in ("de1", "de2"):
        spec += _dense_block(de + ".dec_dense1", 80)
        spec += _conv(de + ".dec_conv1.conv", (128, 64, 1, 3), 64 * 3, 1.5)
        spec += [(de + ".dec_norm1.weight", (161,), "bn_w"), (de + ".dec_norm1.bias", (161,), "bn_b"),
                 (de + ".dec_prelu1.weight", (64,), "prelu")]
        spec += _conv(de + ".out_conv", (1, 64, 1, 1), 64, 1.0)
    return spec


ARCH_SPECS = {
    "Nocon": nocon_spec,
    "dual_aia_trans_merge_crm": dual_aia_trans_merge_crm_spec,
    "aia_complex_trans_ri": aia_complex_trans_ri_spec,
    "DiffUNet1": diffunet1_spec,
    "DiffUNet": diffunet_spec,
    "GCRN": gcrn_spec,
}


# --------------------------------------------------------------------------
# seeded tensors
# --------------------------------------------------------------------------
def make_state_dict(arch, seed=1234, as_torch=True):
    """Deterministic state_dict for ``arch`` ∈ {'DiffUNet1','DiffUNet','GCRN'}.

    Every tensor is a pure function of (arch, seed, its position in the spec).
    """
    spec = ARCH_SPECS[arch]()
    rng = np.random.Generator(np.random.PCG64([seed, sum(map(ord, arch))]))
    sd = OrderedDict()
    for entry in spec:
        key, shape, kind = entry[0], entry[1], entry[2]
        if kind == "w":
            fan_in, gain = entry[3], entry[4]
            v = rng.standard_normal(shape) * (gain / np.sqrt(fan_in))
        elif kind == "bias":
            v = rng.standard_normal(shape) * 0.1
        elif kind == "bn_w":
            v = rng.uniform(0.8, 1.2, shape)
        elif kind == "bn_b":
            v = rng.standard_normal(shape) * 0.1
        elif kind == "bn_mean":
            v = rng.standard_normal(shape) * 0.2
        elif kind == "bn_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif kind == "prelu":
            v = rng.uniform(0.1, 0.4, shape)
        elif kind == "gain1":
            v = rng.uniform(0.6, 1.0, shape)
        elif kind == "bn_nbt":
            sd[key] = np.array(100, dtype=np.int64)
            continue
        else:
            raise ValueError(kind)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    if as_torch:
        import torch

        return OrderedDict((k, torch.from_numpy(v.copy()) if v.shape else torch.tensor(int(v)))
                           for k, v in sd.items())
    return sd


def synthetic_waveforms(batch, length, seed=1234):
    """``randn(B, L)`` RMS-normalised per utterance, then
    ``x_T = randn(B, 2, T, 161)`` drawn from the *same* generator afterwards.

    Returns (wav[B,L] float32, x_T[B,2,T,161] float32) as torch CPU tensors.
    """
    import torch

    g = torch.Generator().manual_seed(seed)
    wav = torch.randn(batch, length, generator=g, dtype=torch.float32)
    wav = wav / torch.sqrt(torch.mean(wav * wav, dim=1, keepdim=True))
    frames = 1 + length // 160
    x_T = torch.randn(batch, 2, frames, F_BINS, generator=g, dtype=torch.float32)
    return wav, x_T


def synthetic_spectrogram(batch, frames, seed=1234):
    """Spectrogram-level inputs: feat, x_T ~ randn(B,2,T,161)."""
    import torch

    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(batch, 2, frames, F_BINS, generator=g, dtype=torch.float32)
    x_T = torch.randn(batch, 2, frames, F_BINS, generator=g, dtype=torch.float32)
    return feat, x_T
