"""CPU: the host side of the product — weight folding/packing, stride algebra and plan
construction — replayed descriptor by descriptor on the numpy interpreter of the ABI
semantics (tests/emu.py) and compared with the oracle.  No HIP kernel runs here; the
kernels themselves are checked on the GPU (test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

import emu
from conftest import pkg, rel_l2, seeded, tcm2_blocks
from oracle import restate as R

TOL = 5e-6


def test_pack_roundtrip():
    P = pkg("packing")
    rng = np.random.default_rng(0)
    wk = rng.standard_normal((37, 70))
    packed = P.pack_a(wk)
    assert packed.shape == (3, 19, 64)
    back = emu._unpack_a(packed.reshape(-1), 3, 19)
    assert np.allclose(back[:37, :70], wk.astype(np.float32))
    assert np.all(back[37:] == 0) and np.all(back[:, 70:] == 0)
    w = rng.standard_normal((64, 32))
    assert np.allclose(emu._unpack_chain(P.pack_chain(w).reshape(-1), 2), w.astype(np.float32))
    # rho enumerates every accumulator row exactly once per lane half pair
    assert sorted(P.RHO.reshape(-1).tolist()) == list(range(32))


@pytest.mark.parametrize("chained,split,plane_h,parity", [(True, False, False, True), (False, False, False, True), (True, True, False, True),
                                                         (True, True, True, True), (True, True, True, False)])
def test_eps_net_plan_vs_oracle(weights, chained, split, plane_h, parity, monkeypatch):
    """chained: every stage's conv1 rides on the previous stage's tail and the encoder/decoder block outputs are never
    stored (only en[4], the TCM input, is); unchained: the per-stage launches with all intermediates in memory.
    split: the BIGLU blocks' weights as exact three-way bf16 splits in bf16 MFMA fragment order (korder 2)."""
    nets = pkg("nets")
    monkeypatch.setattr(nets.EpsNetPlan, "chain_conv1", chained)
    monkeypatch.setattr(nets.EpsNetPlan, "split_bf16", split)
    monkeypatch.setattr(nets.EpsNetPlan, "plane_h", plane_h)      # csrc/bglu.hip: conv1 outputs as bf16 split planes
    monkeypatch.setattr(nets.EpsNetPlan, "parity_planes", parity)  # encoder planes / skip halves with their bins split by parity
    lib = pkg("_lib")
    B, T = 2, 12
    sd = weights("DiffUNet1")
    ctx = nets.Ctx("cpu")
    net = nets.EpsNetPlan(ctx, sd, B, T, time_cond=True, nsteps=1, exclusive=True)      # exclusive: the TCM stack as one persistent launch
    net.build_time()
    net.build_step(0)
    n_conv1 = sum(1 for _, tag in net.descs if tag == nets.TAG_EPS_CONV1)
    n_planes = {False: 0, True: 15}[plane_h]   # stages on plane tensors: the encoder / every stage
    # chained: only decoder stage 5 x 2; on plane tensors those two conv1 and their split into planes are ONE launch (pdse_planes_desc.w)
    assert n_conv1 == (1 if plane_h is True else (2 if chained else 16))
    n_split = sum(1 for d, _ in net.descs if isinstance(d, pkg("_lib").GconvDesc) and d.korder == 2)
    assert n_split == (15 - n_planes if split else 0)     # encoder stages 1-5 + 2 x 5 decoder stages
    assert sum(1 for d, _ in net.descs if isinstance(d, pkg("_lib").BgluDesc)) == n_planes
    if plane_h:   # encoders 2-5 read parity-split planes, encoders 1-4 write them and the skip halves, decoders 5-2 read those
        bg = [d for d, _ in net.descs if isinstance(d, lib.BgluDesc)]
        assert sum(d.hp_par for d in bg) == (4 if parity else 0) and sum(d.nx_par for d in bg) == (4 if parity else 0)
        assert sum(1 for d in bg if d.skip_Fh) == (12 if parity else 0)
    assert len(tcm2_blocks(net.descs)) == (19 if split else 0)      # the first block's conv1 + 18 residual blocks (csrc/tcm2.hip)
    stacks = [d for d, _ in net.descs if isinstance(d, pkg("_lib").Tcm2sDesc)]
    assert len(stacks) == (1 if split else 0) and all(d.n == 18 for d in stacks)   # ... the 18 as ONE persistent launch (round 4)
    x, xi = seeded((B, 2, T, 161), 3), seeded((B, 2, T, 161), 4) * 0.3
    t = torch.tensor([4.086654, 22.992493])
    net.x.copy_(x)
    net.x_init.copy_(xi)
    net.tsteps.copy_(t.view(1, B))
    emu.run(net.descs, ctx.all_tensors())
    taps = {}
    with torch.no_grad():
        ref = R.diffunet1_forward(sd, x, xi, t, taps=taps)
    assert rel_l2(net.temb[0], taps["temb"]) < TOL
    if not chained:
        assert rel_l2(net.en[0], taps["en_list"][0]) < TOL
    assert rel_l2(net.en[4].permute(0, 1, 3, 2), taps["en_list"][4]) < TOL
    assert rel_l2(net.out, ref) < TOL
    if split:                                          # the split bottleneck tensors keep their zero margins
        Pk = pkg("packing")
        for hs in net.tcm_hs:
            Pk.tcm2_join_h(hs.numpy().view(np.uint16), B, T)


@pytest.mark.parametrize("fused_glstm,split,block8,persist", [(True, True, True, True), (True, True, True, False), (False, True, True, False),
                                                              (True, False, True, False), (True, True, False, False)])
def test_gcrn_and_diffunet_prior_plans_vs_oracle(weights, fused_glstm, split, block8, persist, monkeypatch):
    """fused_glstm: both LSTM layers + LayerNorm 1 as one layer-wavefront operator (LayerNorm folded into the layer-2
    input projection, permuted K order); False: two per-frame LSTM operators with the LayerNorm and projections between.
    split: the gated convolutions / input projections packed for the split-bf16 GEMM kernel (korder 3).
    persist: the small-batch form of the wavefront - one persistent launch, its own operand layout (pdse_glstmp_desc)."""
    nets = pkg("nets")
    monkeypatch.setattr(nets.GcrnPlan, "persist_lstm", persist)
    monkeypatch.setattr(nets.GcrnPlan, "fused_glstm", fused_glstm)
    monkeypatch.setattr(nets.GcrnPlan, "split_bf16", split)
    monkeypatch.setattr(nets.GcrnPlan, "block8", block8)          # tensors between the GEMM convolutions in blocks of 8 channels
    B, T = 2, 10
    x = seeded((B, 2, T, 161), 5)
    ctx = nets.Ctx("cpu")
    net = nets.GcrnPlan(ctx, weights("GCRN"), B, T, exclusive=True)
    net.build()
    assert sum(1 for d, _ in net.descs if isinstance(d, pkg("_lib").GlstmpDesc)) == (1 if persist else 0)
    n3 = sum(1 for d, _ in net.descs if isinstance(d, pkg("_lib").GconvDesc) and d.korder == 5)       # the default split: f16x2 (korder 5)
    assert n3 == ((4 + 16 + (2 if fused_glstm else 4)) if split else 0)   # encoder 2-5, 2 x 4 x 2 decoder phases, projections
    nblk = sum(1 for d, _ in net.descs if isinstance(d, pkg("_lib").GconvDesc) and d.in0.blk)
    assert nblk == (3 + 2 + 16 if (split and block8) else 0)               # encoders 3-5, the layer-1 projections, every decoder launch
    net.x.copy_(x)
    emu.run(net.descs, ctx.all_tensors())
    taps = {}
    with torch.no_grad():
        ref = R.gcrn_forward(weights("GCRN"), x, taps=taps)
    assert rel_l2(net.enc_out(5), taps["e5"]) < TOL
    assert rel_l2(net.glstm_out(), taps["glstm"]) < TOL
    assert rel_l2(net.out, ref) < TOL
    ctx2 = nets.Ctx("cpu")
    p = nets.EpsNetPlan(ctx2, weights("DiffUNet"), B, T, time_cond=False)
    p.build_step(0)
    p.x.copy_(x)
    emu.run(p.descs, ctx2.all_tensors())
    with torch.no_grad():
        refp = R.diffunet_forward(weights("DiffUNet"), x)
    assert rel_l2(p.out, refp) < TOL


@pytest.mark.parametrize("sigma", [False, True])
def test_whole_pipeline_plan_vs_oracle(weights, sigma):
    """wav -> STFT -> GCRN -> 6 reverse steps -> ISTFT as ONE recorded plan (614 operators),
    ragged length (L % hop != 0), three different input scales."""
    params = pkg("params").params
    B, L_ = 2, 1700
    P = pkg("pipeline").SamplerPipeline("cpu", "GCRN", weights("GCRN"), weights("DiffUNet1"), B, L_=L_,
                                        fast_sampling=True, use_sigma=sigma)
    assert list(P.ranges)[:3] == ["stft", "prior", "prologue"] and "step0" in P.ranges and "istft" in P.ranges
    wav, x_T = pkg("synth").synthetic_waveforms(B, L_, seed=7)
    wav = wav * torch.tensor([0.1, 3.0])[:, None]
    P.stft.wav.copy_(wav)
    P.xT_in.copy_(x_T)
    emu.run(P.descs, P.ctx.all_tensors())
    with torch.no_grad():
        ref_wav, ref_spec = R.enhance("GCRN", weights("GCRN"), weights("DiffUNet1"), wav, x_T, params.noise_schedule,
                                      params.inference_noise_schedule, True, sigma)
    assert rel_l2(P.spec, ref_spec) < 2e-5
    assert rel_l2(P.istft.wav, ref_wav) < 2e-5
    with pytest.raises(pkg("_lib").PdseError):
        P.run()  # a CPU-built pipeline must refuse to run: there is no CPU product path


def test_step_descriptors_are_cloned_not_repacked(weights):
    """Later diffusion steps re-use the packed weights of the first and only re-point the
    per-step time biases."""
    nets = pkg("nets")
    ctx = nets.Ctx("cpu")
    net = nets.EpsNetPlan(ctx, weights("DiffUNet1"), 1, 8, time_cond=True, nsteps=3)
    net.build_step(0)
    n_keep = len(ctx.all_tensors())
    first = [d for d, _ in net.descs]
    net.build_step(2)
    assert len(ctx.all_tensors()) == n_keep
    second = [d for d, _ in net.descs[len(first):]]
    assert len(second) == len(first)
    delta = 2 * 1 * net.NSLOT * 32 * 4
    moved = 0
    L = pkg("_lib")
    for a, b in zip(first, second):
        assert type(a) is type(b)
        if isinstance(a, (L.TcmDesc, L.Tcm2Desc, L.Tcm2sDesc)):         # fused TCM blocks carry no time bias: identical clones
            assert bytes(a) == bytes(b)
            continue
        if isinstance(a, L.PlanesDesc):                      # the decoders' stage-5 conv1 computed into planes: two time biases move
            assert a.w[0] == b.w[0] and a.w[1] == b.w[1] and a.hp == b.hp and a.hp1 == b.hp1
            for i in range(2):
                assert b.bias[i] - a.bias[i] == delta
                moved += 1
            continue
        assert a.w0 == b.w0 and a.out == b.out
        for f in ("bias0", "bias1", "bias0_t0", "bias1_t0"):
            if getattr(a, f) != getattr(b, f):
                assert getattr(b, f) - getattr(a, f) == delta
                moved += 1
        if isinstance(a, L.GconvDesc) and a.padrow != b.padrow:
            assert b.padrow - a.padrow == delta
        assert a.nx_n == b.nx_n and a.nx_w == b.nx_w
        for i in range(a.nx_n):                            # chained conv1 tiles carry the per-step time bias
            if a.nx_bias[i] != b.nx_bias[i]:
                assert b.nx_bias[i] - a.nx_bias[i] == delta
                moved += 1
    assert moved == 18      # 14 stage biases (launches or chained tiles) + l, r biases of the composed stage 1 x (frame >= 1, frame 0)


def test_stft_bases_against_numpy_fft():
    P = pkg("packing")
    rng = np.random.default_rng(1)
    x = rng.standard_normal(320)
    spec = np.fft.rfft(x * P.hann_periodic(320))
    got = x @ P.stft_kmat(320)
    assert np.allclose(got[:161], spec.real) and np.allclose(got[161:], spec.imag)
    z = rng.standard_normal(161) + 1j * rng.standard_normal(161)
    frame = np.fft.irfft(z, 320) * P.hann_periodic(320)
    k = np.empty(322)
    k[0::2], k[1::2] = z.real, z.imag
    assert np.allclose(k @ P.istft_kmat(320), frame)


def test_eps_net_plan_bf16_mode_vs_oracle(weights):
    """The opt-in bf16 mode (one plane): every BiConv(Trans)GLU stage and every TCM block with plain bf16 matrix operands and
    bf16 exchanged tensors, replayed on the interpreter (which rounds where the kernels round): the result stays within the
    mode's stated tolerance of the fp32 oracle and is NOT fp32-equivalent."""
    nets, Lb = pkg("nets"), pkg("_lib")
    B, T = 2, 12
    sd = weights("DiffUNet1")
    ctx = nets.Ctx("cpu")
    net = nets.EpsNetPlan(ctx, sd, B, T, time_cond=True, nsteps=1, planes=1)
    net.build_time()
    net.build_step(0)
    assert all(d.np == 1 for d, _ in net.descs if isinstance(d, (Lb.BgluDesc, Lb.PlanesDesc))) and all(d.np == 1 for d in tcm2_blocks(net.descs))
    assert sum(1 for d, _ in net.descs if isinstance(d, Lb.BgluDesc)) == 15 and len(tcm2_blocks(net.descs)) == 19
    x, xi = seeded((B, 2, T, 161), 3), seeded((B, 2, T, 161), 4) * 0.3
    t = torch.tensor([4.086654, 22.992493])
    net.x.copy_(x)
    net.x_init.copy_(xi)
    net.tsteps.copy_(t.view(1, B))
    emu.run(net.descs, ctx.all_tensors())
    with torch.no_grad():
        ref = R.diffunet1_forward(sd, x, xi, t)
    e = rel_l2(net.out, ref)
    assert 1e-4 < e < 3e-2, e
    with pytest.raises(ValueError):
        nets.EpsNetPlan(nets.Ctx("cpu"), weights("Nocon"), B, T, time_cond=True, nsteps=1, with_pre=False, planes=1)


def test_eps_net_plan_f16x2_mode_vs_oracle(weights):
    """The f16x2 form of the block kernels (two planes: fp16 hi + lo of the power-of-two scaled operand, three f16 products
    per multiply-add): weights packed with their per-group exponents (pdse_bglu_desc.qexp), plane tensors at
    2^PDSE_F16_ACT_EXP, replayed on the interpreter - fp32-equivalent: the SAME tolerance as the three-plane split."""
    nets, Lb, Pk = pkg("nets"), pkg("_lib"), pkg("packing")
    B, T = 2, 12
    sd = weights("DiffUNet1")
    ctx = nets.Ctx("cpu")
    net = nets.EpsNetPlan(ctx, sd, B, T, time_cond=True, nsteps=1, planes=2)
    net.build_time()
    net.build_step(0)
    bg = [d for d, _ in net.descs if isinstance(d, Lb.BgluDesc)]
    assert len(bg) == 15 and all(d.np == 2 for d in bg) and all(d.np == 2 for d, _ in net.descs if isinstance(d, Lb.PlanesDesc))
    assert all(d.np == 2 for d in tcm2_blocks(net.descs)) and all(5 <= d.qexp[2] <= 30 for d in tcm2_blocks(net.descs) if d.hs_out)
    assert all(5 <= q <= 30 for d in bg for q in (d.qexp[0], d.qexp[1])), [tuple(d.qexp) for d in bg]    # weights below 1: scaled up
    x, xi = seeded((B, 2, T, 161), 3), seeded((B, 2, T, 161), 4) * 0.3
    t = torch.tensor([4.086654, 22.992493])
    net.x.copy_(x)
    net.x_init.copy_(xi)
    net.tsteps.copy_(t.view(1, B))
    emu.run(net.descs, ctx.all_tensors())
    with torch.no_grad():
        ref = R.diffunet1_forward(sd, x, xi, t)
    e = rel_l2(net.out, ref)
    assert e < 2e-5, e
    # the split itself: hi + lo reproduces a scaled weight to half an fp32 ulp, an activation inside the fp16 window likewise
    w = np.asarray(seeded((96, 32), 9)) * 0.07
    q = Pk.f16_wexp(w)
    back = Pk.unpack_bglu_gather(Pk.pack_bglu_gather(w, 3, 2, q), 3, q)
    assert 2.0 ** 13 <= np.abs(w).max() * 2.0 ** q < 2.0 ** 14 and np.max(np.abs(back - w) / np.abs(w)) <= 2.0 ** -23
    h = np.asarray(seeded((1, 32, 3, 9), 10))
    h[np.abs(h) < 2.0 ** (-2 - Pk.F16_ACT_EXP)] = 0.25
    assert np.max(np.abs(Pk.hp_join(Pk.hp_split(h, 2)) - h) / np.abs(h)) <= 2.0 ** -23
    big = np.full((1, 32, 1, 1), 1e6, np.float32)                                # beyond the window: infinities in the planes, never a clipped value
    with np.errstate(invalid="ignore"):
        assert not np.isfinite(Pk.hp_join(Pk.hp_split(big, 2))).any()


def test_priors_one_plane_gemm_mode_vs_oracle(weights):
    """The bf16 mode of the priors (round 4): the GEMM-shaped convolutions packed as ONE bf16 plane (korder 4, packing.pack_s3_gemm
    with npl = 1), replayed on the interpreter with both operands rounded to bf16: within the mode's prior tolerance of the fp32
    oracle, and not fp32-equivalent."""
    nets = pkg("nets")
    B, T = 2, 10
    x = seeded((B, 2, T, 161), 5)
    ctx = nets.Ctx("cpu")
    net = nets.GcrnPlan(ctx, weights("GCRN"), B, T, planes=1)
    net.build()
    assert sum(1 for d, _ in net.descs if isinstance(d, pkg("_lib").GconvDesc) and d.korder == 4) == 22      # every korder-3 launch of the exact form
    net.x.copy_(x)
    emu.run(net.descs, ctx.all_tensors())
    with torch.no_grad():
        ref = R.gcrn_forward(weights("GCRN"), x)
    e = rel_l2(net.out, ref)
    assert 1e-4 < e < 1e-2, e
    with pytest.raises(ValueError):
        nets.GcrnPlan(nets.Ctx("cpu"), weights("GCRN"), B, T, planes=4)
    # the f16x2 form of the same launches (korder 5: fp16 hi + lo planes of W * 2^wexp): fp32-equivalent on the interpreter
    ctx2 = nets.Ctx("cpu")
    net2 = nets.GcrnPlan(ctx2, weights("GCRN"), B, T, planes=2)
    net2.build()
    k5 = [d for d, _ in net2.descs if isinstance(d, pkg("_lib").GconvDesc) and d.korder == 5]
    assert len(k5) == 22 and all(5 <= d.wexp <= 30 for d in k5)
    net2.x.copy_(x)
    emu.run(net2.descs, ctx2.all_tensors())
    assert rel_l2(net2.out, ref) < 2e-5
