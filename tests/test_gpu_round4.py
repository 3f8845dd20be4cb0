"""GPU parity, round 4: the persistent small-batch form of the GCRN prior's grouped LSTM (csrc/lstmp.hip,
pdse_glstmp_desc; reference model/gcrn.py:6-40) against the reference's goldens and against the layer wavefront it
replaces for B <= 8; everything through the C-ABI."""
import numpy as np
import pytest
import torch

from conftest import golden, pkg, rel_l2, seeded, tcm2_blocks

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge

    ge.build()
    lib = pkg("_lib")
    lib.load()
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return lib


def _gcrn(weights, persist, monkeypatch):
    nets = pkg("nets")
    monkeypatch.setattr(nets.GcrnPlan, "persist_lstm", persist)
    return pkg("ops").GCRNOp(weights("GCRN"), DEV, exclusive=True)


@pytest.mark.parametrize("persist", [True, False])
def test_gcrn_golden_both_lstm_forms(L, weights, persist, monkeypatch):
    """The reference's own GCRN outputs (tests/golden/gcrn_small.npz: e5, the LSTM block's output, the network output;
    gcrn_t401: rows of a 4 s utterance) with the LSTM as one persistent launch and as the layer wavefront."""
    g = golden("gcrn_small")
    op = _gcrn(weights, persist, monkeypatch)
    out = op(seeded((2, 2, 20, 161), g["seed_x"]).to(DEV))
    net = op._plans[(2, 20)]
    assert net.persist == persist
    assert sum(1 for d, _ in net.descs if isinstance(d, L.GlstmpDesc)) == (1 if persist else 0)
    torch.cuda.synchronize()
    assert rel_l2(net.glstm_out().cpu(), g["glstm"]) < 2e-5
    assert rel_l2(out.cpu(), g["out"]) < 2e-5
    g = golden("gcrn_t401")
    out = op(seeded((1, 2, 401, 161), g["seed_x"]).to(DEV)).cpu()
    assert rel_l2(out[0, :, ::16, :], g["rows"]) < 5e-5
    if persist:
        assert int(op._plans[(1, 401)].status[0].item()) == 0


@pytest.mark.parametrize("B,T", [(1, 50), (2, 33), (3, 50), (4, 17), (5, 50), (8, 50), (1, 1001), (8, 401)])   # 5, 8: above the default limit
def test_persistent_lstm_equals_wavefront(L, weights, B, T, monkeypatch):
    """Every batch size the persistent form takes (granule widths 1, 2, 4, 8 and their padded items), short and long
    utterances: the LSTM block's output and the prior's output against the wavefront kernels (fp32 both, different
    summation order: 1e-5), a second run of the same plan bit for bit (the granule ring is re-initialised by every launch),
    and utterance b of a batch bit for bit equal to the same utterance run alone (batch invariance)."""
    x = seeded((B, 2, T, 161), 40 + B).to(DEV)
    monkeypatch.setattr(pkg("nets").GcrnPlan, "PERSIST_MAX_B", 8)      # the kernel takes B <= 8; the plan builder picks it up to the measured break-even
    op_w = _gcrn(weights, False, monkeypatch)
    ref = op_w(x).clone()
    ref_l = op_w._plans[(B, T)].glstm_out().clone()
    op_p = _gcrn(weights, True, monkeypatch)
    out = op_p(x).clone()
    net = op_p._plans[(B, T)]
    assert net.persist
    got_l = net.glstm_out().clone()
    torch.cuda.synchronize()
    assert int(net.status[0].item()) == 0
    assert rel_l2(got_l.cpu(), ref_l.cpu()) < 1e-5
    assert rel_l2(out.cpu(), ref.cpu()) < 1e-5
    out2 = op_p(x)
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and int(net.status[0].item()) == 0
    if B > 1 and T <= 50:
        b = B - 1
        alone = op_p(x[b:b + 1].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(alone[0], out[b])


def test_persistent_lstm_rejects_what_it_cannot_run(L):
    d = L.GlstmpDesc()
    with pytest.raises(L.PdseError, match="glstmp: null"):
        L.launch(d)
    buf = torch.zeros(64, device=DEV)
    for f in ("gx1", "w1", "w2i", "w2h", "r2", "c2", "gran", "status", "y"):
        setattr(d, f, buf.data_ptr())
    d.B, d.Bp, d.T, d.H, d.G = 9, 32, 4, 512, 2
    with pytest.raises(L.PdseError, match="glstmp: bad sizes"):
        L.launch(d)


# ------------------------------------------------------------------ network-level C entry points on a saved plan
def _client(tmp_path, kind, plan, golden_path, tol):
    import os
    import subprocess
    import sys

    from conftest import ROOT

    lib = os.path.join(ROOT, "prior-diffuse_amd", "libpdse.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "plan_client.py"), lib, kind, plan, golden_path, repr(tol)],
                       capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    return float(r.stdout.strip().split()[-1])


def test_saved_plans_through_the_bare_c_abi(L, weights, tmp_path):
    """pdse_plan_load + pdse_eps_forward / pdse_prior_forward / pdse_enhance from a process that imports nothing of this package:
    the reference's goldens for DiffUNet1 and GCRN (tests/golden/diffunet1_small.npz, gcrn_small.npz) at their own tolerances,
    and the whole path wav -> wav equal to the pipeline that wrote the plan."""
    import os

    from conftest import GOLDEN

    pf = pkg("planfile")
    g = golden("diffunet1_small")
    eps_plan = pf.save_eps_net(str(tmp_path / "eps.plan"), weights("DiffUNet1"), int(g["B"]), int(g["T"]), DEV)
    assert _client(tmp_path, "eps", eps_plan, os.path.join(GOLDEN, "diffunet1_small.npz"), 2e-5) < 2e-5
    prior_plan = pf.save_prior(str(tmp_path / "gcrn.plan"), "GCRN", weights("GCRN"), 2, 20, DEV)
    assert _client(tmp_path, "prior", prior_plan, os.path.join(GOLDEN, "gcrn_small.npz"), 2e-5) < 2e-5
    B, L_ = 2, 2400
    pipe = pkg("pipeline").SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, L_=L_)
    wav, x_T = pkg("synth").synthetic_waveforms(B, L_, seed=31)
    plan = pf.save_pipeline(str(tmp_path / "path.plan"), pipe)          # before the first run: buffers as the builders left them
    out, spec = pipe.enhance(wav.to(DEV), x_T.to(DEV))
    torch.cuda.synchronize()
    ref = str(tmp_path / "ref.npz")
    np.savez(ref, wav=wav.numpy(), x_T=x_T.numpy(), wav_out=out.cpu().numpy(), spec=spec.cpu().numpy())
    assert _client(tmp_path, "enhance", plan, ref, 1e-12) == 0.0        # the same kernels on the same operands: bit for bit
    # ... and a plan that owns the GPU: the persistent LSTM (granule ring, status word) and the TCM stack descriptor (18 block
    # descriptors by value, every pointer in them re-based) survive the file as well
    pipe = pkg("pipeline").SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, L_=L_, exclusive=True)
    assert any(isinstance(d, L.GlstmpDesc) for d, _ in pipe.descs) and any(isinstance(d, L.Tcm2sDesc) for d, _ in pipe.descs)
    plan = pf.save_pipeline(str(tmp_path / "path_x.plan"), pipe)
    out, spec = pipe.enhance(wav.to(DEV), x_T.to(DEV))
    pipe.check()
    np.savez(ref, wav=wav.numpy(), x_T=x_T.numpy(), wav_out=out.cpu().numpy(), spec=spec.cpu().numpy())
    assert _client(tmp_path, "enhance", plan, ref, 1e-12) == 0.0


def test_block_kernel_refuses_a_tensor_without_its_dump_item(L, weights):
    """pdse_bglu_planes stores unconditionally into item B of nx_hp / nx_out (lanes beyond the last position): the descriptor says
    how many items were allocated and the launcher refuses fewer than B + 1 (advisor, round 3)."""
    nets = pkg("nets")
    net = nets.EpsNetPlan(nets.Ctx(DEV), weights("DiffUNet1"), 1, 8, time_cond=True, nsteps=1)
    net.build_time()
    net.build_step(0)
    d = [d for d, _ in net.descs if isinstance(d, L.BgluDesc) and d.nx_n > 0][0]
    assert d.nx_items == d.B + 1
    bad = type(d).from_buffer_copy(d)
    bad.nx_items = bad.B
    with pytest.raises(L.PdseError, match="dump target"):
        L.launch(bad)


# ------------------------------------------------------------------ the bf16 rows of BASELINE configs 4 and 5 (config 2: test_gpu_round2.py)
@pytest.mark.parametrize("config", [4, 5])
def test_bf16_mode_tolerance_configs_4_and_5(L, weights, config, tmp_path):
    """The opt-in bf16 mode at the shapes BASELINE names it for, against the reference's own fp32 results: config 4 (B=32, T=401,
    aia_complex_trans_ri prior, 6 steps; tests/golden/full_aia_complex_trans_ri_seed404_t401_6step.npz) and config 5 (B=16,
    L=160,000, T=1001, GCRN, STFT..ISTFT; full_generate_wav_seed505_l160000.npz).  Stated tolerance: 3e-2 rel-L2 on the
    enhanced spectrogram and waveform (the mode's own, DESIGN.md 4.1g: 8 significand bits per operand over 15 blocks x 6 steps),
    2e-2 on the prior's output (its GEMM-shaped convolutions multiply plain bf16 operands too: csrc/gconv4.hip, korder 4; tensors
    and recurrences stay fp32).  The mode must also be selectable from the drop-in class."""
    from conftest import assert_rows2_and_checksums

    P = pkg("pipeline").SamplerPipeline
    if config == 4:
        B, T = 32, 401
        feat, x_T = pkg("synth").synthetic_spectrogram(B, T, seed=404)
        pipe = P(DEV, "aia_complex_trans_ri", weights("aia_complex_trans_ri"), weights("DiffUNet1"), B, T=T, dtype="bf16")
        spec, init = pipe.sample(feat.to(DEV), x_T.to(DEV))
        g = golden("full_aia_complex_trans_ri_seed404_t401_6step")
        e_init, e_spec = rel_l2(init[0].cpu(), g["init"][0]), rel_l2(spec[0].cpu(), g["out"][0])
        print("bf16 mode, config 4: prior %.2e | spectrogram %.2e" % (e_init, e_spec))
        in_mode = sum(1 for d, _ in pipe.descs if (isinstance(d, L.GconvDesc) and d.korder == 4) or (isinstance(d, L.DenseDesc) and d.np == 1))
        assert in_mode >= 12      # the dense-block layers (csrc/dense.hip since ABI 7) and the strided convolutions are in the mode
        assert 1e-4 < e_init < 2e-2 and 1e-4 < e_spec < 3e-2      # DB-AIAT: four dense blocks of bf16 products deep: measured 1.0e-2
    else:
        B, L_ = 16, 160000
        wav, x_T = pkg("synth").synthetic_waveforms(B, L_, seed=505)
        wav = wav * torch.linspace(0.05, 2.0, B)[:, None]
        pipe = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, L_=L_, dtype="bf16")
        assert pipe.T == 1001 and all(d.np == 1 for d, _ in pipe.descs if isinstance(d, L.BgluDesc)) and all(d.np == 1 for d in tcm2_blocks(pipe.descs))
        out, spec = pipe.enhance(wav.to(DEV), x_T.to(DEV))
        g = golden("full_generate_wav_seed505_l160000")
        e_wav = rel_l2(out[0].cpu(), g["wav"])
        e_spec = rel_l2(np.asarray(spec[:1].cpu(), np.float64)[:, :, ::2], g["spec_rows2"])
        print("bf16 mode, config 5: spectrogram %.2e | waveform %.2e" % (e_spec, e_wav))
        assert 1e-4 < e_spec < 3e-2 and e_wav < 3e-2
        import argparse

        ns = argparse.Namespace
        tr = pkg("trainer").ComplexDDPMTrainer(
            ns(retrain=False, joint=True, draw=False, sigma=False, checkpoint="x", generated_wav=str(tmp_path), bf16=True),
            ns(model=ns(name="GCRN"), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt")),
            device=DEV, prior_state_dict=weights("GCRN"), ddpm_state_dict=weights("DiffUNet1"))
        assert tr.dtype == "bf16"
        y = tr.enhance(wav[:1, :4000].to(DEV), x_T=x_T[:1, :, :26].to(DEV))
        p = tr._pipes[next(reversed(tr._pipes))]
        assert p.dtype == "bf16" and all(d.np == 1 for d, _ in p.descs if isinstance(d, L.BgluDesc)) and torch.isfinite(y).all()


# ------------------------------------------------------------------ the TCM stack as one persistent launch (csrc/tcm2.hip: tcm2s_kernel)
@pytest.mark.parametrize("B,T,planes", [(2, 20, 3), (32, 401, 3), (3, 1001, 3), (32, 401, 1), (2, 20, 2), (32, 401, 2)])
def test_tcm_stack_launch_is_bit_identical_to_one_launch_per_block(L, weights, B, T, planes, monkeypatch):
    """The 18 residual blocks as ONE launch whose workgroups wait for their neighbours' progress counters (pdse_tcm2s_desc)
    against 18 launches of the same block kernel: the residual stream after the stack and the network output bit for bit, at a
    small shape, at the bench shape (416 co-resident workgroups), at T = 1001 (32 tiles per utterance), in the one-plane bf16
    mode and in the two-plane f16x2 form (the default); run three times (the counters are re-initialised by every launch) and with two plans interleaved on two streams (two
    stack launches competing for the CUs)."""
    nets = pkg("nets")
    x, xi = seeded((B, 2, T, 161), 71).to(DEV), (seeded((B, 2, T, 161), 72) * 0.3).to(DEV)
    res = {}
    for stack in (False, True):
        monkeypatch.setattr(nets.EpsNetPlan, "tcm_stack", stack)
        net = nets.EpsNetPlan(nets.Ctx(DEV), weights("DiffUNet1"), B, T, time_cond=True, nsteps=1, planes=planes, exclusive=True)
        net.build_time()
        net.build_step(0)
        net.finish()
        assert sum(1 for d, _ in net.descs if isinstance(d, L.Tcm2sDesc)) == (1 if stack else 0) and all(d.np == planes for d in tcm2_blocks(net.descs))
        net.x.copy_(x)
        net.x_init.copy_(xi)
        net.tsteps.fill_(7.25)
        for _ in range(3 if stack else 1):
            net.plan.run()
        torch.cuda.synchronize()
        last = tcm2_blocks(net.descs)[-1]
        res[stack] = (net.out.clone(), (net.tcm_a if last.x_out == net.tcm_a.data_ptr() else net.tcm_b).clone(), net)
    assert int(res[True][2].tcm_status[0].item()) == 0
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][0], res[False][0])
    # two stack launches in flight at once
    a = res[True][2]
    monkeypatch.setattr(nets.EpsNetPlan, "tcm_stack", True)
    b = nets.EpsNetPlan(nets.Ctx(DEV), weights("DiffUNet1"), B, T, time_cond=True, nsteps=1, planes=planes, exclusive=True)
    b.build_time()
    b.build_step(0)
    b.finish()
    b.x.copy_(x)
    b.x_init.copy_(xi)
    b.tsteps.fill_(7.25)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        a.plan.run(s1.cuda_stream)
        b.plan.run(s2.cuda_stream)
    torch.cuda.synchronize()
    assert int(a.tcm_status[0].item()) == 0 and int(b.tcm_status[0].item()) == 0
    assert torch.equal(a.out, res[False][0]) and torch.equal(b.out, res[False][0])


@pytest.mark.parametrize("prior", ["GCRN", "aia_complex_trans_ri", "dual_aia_trans_merge_crm"])
def test_bf16_mode_of_the_priors_one_plane_gemm_convolutions(L, weights, prior):
    """korder 4 of csrc/gconv4.hip - the GEMM-shaped convolutions on ONE bf16 plane (weights and gathered activations rounded to
    nearest even, one MFMA product, fp32 accumulate) - in all three priors: against the reference's golden forward within the
    mode's prior tolerance (2e-2; GCRN measures 2e-3, DB-AIAT 1e-2), clearly apart from the exact form (so the mode is really on), and the exact form untouched."""
    nets = pkg("nets")
    cls = {"GCRN": nets.GcrnPlan, "aia_complex_trans_ri": nets.AiaPlan, "dual_aia_trans_merge_crm": nets.DualAiaPlan}[prior]
    g = golden({"GCRN": "gcrn_small", "aia_complex_trans_ri": "aia_small", "dual_aia_trans_merge_crm": "dual_aia_small"}[prior])
    x = seeded(tuple(g["out"].shape), g["seed_x"]).to(DEV)
    err = {}
    for planes in (3, 1):
        net = cls(nets.Ctx(DEV), weights(prior), x.shape[0], x.shape[2], planes=planes)
        net.build()
        net.finish()
        net.x.copy_(x)
        net.plan.run()
        torch.cuda.synchronize()
        n4 = sum(1 for d, _ in net.descs if isinstance(d, L.GconvDesc) and d.korder == 4)
        assert (n4 > 0) == (planes == 1)
        err[planes] = rel_l2(net.out.cpu(), g["out"])
    print("%s: exact split %.2e | one bf16 plane %.2e" % (prior, err[3], err[1]))
    assert err[3] < 5e-5 and 1e-4 < err[1] < 2e-2


# ---- DB-AIAT dense-block layers as one launch each (csrc/dense.hip; reference model/dbaiat.py:605-631) ------------------------
def _dense_block_reference(x, ws, bs, gs, bes, sls):
    """fp64 restatement of DenseBlock.forward for depth 4 (pad (1,1,dil,0) -> conv (2,3) dilation (dil,1) -> LayerNorm(F) -> PReLU)."""
    F_ = x.shape[-1]
    skip, outs = x, []
    for i in range(4):
        dil = 2 ** i
        out = torch.nn.functional.pad(skip, (1, 1, dil, 0))
        out = torch.nn.functional.conv2d(out, ws[i], bs[i], dilation=(dil, 1))
        out = torch.nn.functional.layer_norm(out, (F_,), gs[i], bes[i], 1e-5)
        out = torch.where(out > 0, out, sls[i].view(1, -1, 1, 1) * out)
        outs.append(out)
        skip = torch.cat([out, skip], 1)
    return outs


@pytest.mark.parametrize("B,T,F_,npl", [(2, 21, 161, 3), (3, 50, 80, 3), (1, 7, 161, 3), (2, 37, 80, 1), (1, 30, 36, 3), (2, 21, 161, 2), (3, 50, 80, 2)])
def test_dense_layers_and_relayout_vs_fp64(L, B, T, F_, npl):
    """pdse_rowln_blocked_f32 (copy and LayerNorm + PReLU forms) and four pdse_dense_layer_bf16x3 launches - every dilation, frame
    counts that are no multiple of the row block, rows cut by tile boundaries, both bin widths of the model and the smallest
    the kernel takes - against an fp64 restatement of the reference's DenseBlock.  Pads of the buffer must still be zero."""
    P = pkg("packing")
    gen = torch.Generator().manual_seed(B * 1000 + T * 10 + F_)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)   # noqa: E731
    x = rnd(B, 64, T, F_)
    ws = [rnd(64, 64 * (i + 1), 2, 3) / np.sqrt(6 * 64 * (i + 1)) for i in range(4)]
    bs = [0.1 * rnd(64) for _ in range(4)]
    gs = [1 + 0.1 * rnd(F_) for _ in range(4)]
    bes = [0.1 * rnd(F_) for _ in range(4)]
    sls = [0.25 + 0.05 * rnd(64) for _ in range(4)]
    # the block's input: LayerNorm + PReLU of a raw tensor (the encoder's inp_norm / inp_prelu), fp64
    g0, b0, s0 = 1 + 0.1 * rnd(F_), 0.1 * rnd(F_), 0.25 + 0.05 * rnd(64)
    xin = torch.nn.functional.layer_norm(x, (F_,), g0, b0, 1e-5)
    xin = torch.where(xin > 0, xin, s0.view(1, -1, 1, 1) * xin)
    want = _dense_block_reference(xin, ws, bs, gs, bes, sls)

    G, TP = 40, 8
    Tp, Fp = T + TP, F_ + 2
    dev = lambda a: a.to(torch.float32).contiguous().to(DEV)   # noqa: E731
    D = torch.zeros(B, G, Tp, Fp, 8, device=DEV)
    off = lambda g: ((g * Tp + TP) * Fp + 1) * 8               # noqa: E731
    xd = dev(x)
    keep = [xd]
    r = L.RowlnbDesc()
    r.in_, r.out = xd.data_ptr(), D.data_ptr() + 4 * off(32)
    for name, t in (("gamma", g0), ("beta", b0), ("slope", s0)):
        keep.append(dev(t))
        setattr(r, name, keep[-1].data_ptr())
    r.in_sb, r.in_sc, r.in_st = 64 * T * F_, T * F_, F_
    r.out_sb, r.out_sg, r.out_st = G * Tp * Fp * 8, Tp * Fp * 8, Fp * 8
    r.B, r.C, r.T, r.F, r.eps = B, 64, T, F_, 1e-5
    st = torch.cuda.current_stream().cuda_stream
    L.launch(r, st)

    def groups(g, n):   # [B, 8n, T, F] view of groups g..g+n
        return D[:, g:g + n, TP:, 1:F_ + 1].permute(0, 1, 4, 2, 3).reshape(B, 8 * n, T, F_)

    torch.cuda.synchronize()
    assert rel_l2(groups(32, 8).cpu(), xin) < 1e-6
    kk = [(kt, kf) for kt in range(2) for kf in range(3)]
    for i in range(1, 5):
        d = L.DenseDesc()
        d.D = D.data_ptr()
        wk = P.conv_kmat(ws[i - 1], kk)
        d.wexp = P.f16_wexp(wk) if npl == 2 else 0
        w = torch.from_numpy(P.pack_dense(wk, 64 * i, npl, d.wexp).view(np.int16)).to(DEV)
        keep.append(w)
        d.w = w.data_ptr()
        for name, t in (("bias", bs[i - 1]), ("gamma", gs[i - 1]), ("beta", bes[i - 1]), ("slope", sls[i - 1])):
            keep.append(dev(t))
            setattr(d, name, keep[-1].data_ptr())
        d.B, d.T, d.F, d.G, d.tpad = B, T, F_, G, TP
        d.g_in, d.cin, d.g_out, d.dil, d.np, d.eps = (5 - i) * 8, 64 * i, (4 - i) * 8, 2 ** (i - 1), npl, 1e-5
        L.launch(d, st)
    torch.cuda.synchronize()
    tol = 3e-2 if npl == 1 else 1e-5                     # three bf16 planes and f16x2: fp32-equivalent
    errs = [rel_l2(groups((4 - i) * 8, 8).cpu(), want[i - 1]) for i in range(1, 5)]
    print("dense layers B %d T %d F %d planes %d: %s" % (B, T, F_, npl, " ".join("%.2e" % e for e in errs)))
    assert max(errs) < tol, errs
    assert float(D[:, :, :TP].abs().max()) == 0.0 and float(D[:, :, :, 0].abs().max()) == 0.0 and float(D[:, :, :, -1].abs().max()) == 0.0
    # plain re-layout (the decoders' input) and the refusals
    r.gamma = r.beta = r.slope = None
    r.out = D.data_ptr() + 4 * off(0)
    L.launch(r, st)
    torch.cuda.synchronize()
    assert torch.equal(groups(0, 8), xd)
    d.g_out = d.g_in
    with pytest.raises(L.PdseError, match="overlap"):
        L.launch(d, st)
    d.g_out, d.dil = 0, 16
    with pytest.raises(L.PdseError, match="dilation"):
        L.launch(d, st)


@pytest.mark.parametrize("B,T", [(32, 60), (5, 33), (16, 401)])
def test_glstm_two_slices_per_workgroup_is_bit_identical(L, weights, B, T):
    """pdse_glstm_f32 with two slices of 8 hidden units per workgroup (pdse_glstm_desc.slices = 2: one fetch of the group's state for
    both - what a plan that owns the GPU takes since round 4) against one slice per workgroup: every summation keeps its order, so the
    whole GCRN prior is bit-identical - at the bench's batch, at a padded batch and over 401 frames."""
    nets = pkg("nets")
    x = seeded((B, 2, T, 161), 91).to(DEV)
    outs = []
    for exclusive in (False, True):
        net = nets.GcrnPlan(nets.Ctx(DEV), weights("GCRN"), B, T, exclusive=exclusive)
        net.build()
        net.finish()
        gl = [d for d, _ in net.descs if isinstance(d, L.GlstmDesc)]
        assert not net.persist and len(gl) == 1 and gl[0].slices == (2 if exclusive else 1)
        net.x.copy_(x)
        net.plan.run()
        torch.cuda.synchronize()
        outs.append((net.out.clone(), net.glstm_out().clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])
    d = L.GlstmDesc.from_buffer_copy(gl[0])
    d.slices = 3
    with pytest.raises(L.PdseError, match="slices"):
        L.launch(d, torch.cuda.current_stream().cuda_stream)
