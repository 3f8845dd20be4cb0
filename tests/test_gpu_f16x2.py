"""GPU parity of the f16x2 form of the matrix-core kernels (round 4): every operand as fp16 hi + lo of its power-of-two scaled
value, three f16 MFMA products per multiply-add (csrc/gconv_common.h: split8h; include/pdse.h: PDSE_F16_ACT_EXP,
pdse_bglu_desc.qexp).  It replaces arithmetic of the reference's fp32 blocks (model/diff3.py:215-351) and is held to the SAME
goldens and tolerances as the exact three-way bf16 split and the fp32 MFMA kernels; everything through the C-ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import full_pair, golden, pkg, rel_l2, seeded

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


def _eps_plan(weights, B, T, planes):
    nets = pkg("nets")
    net = nets.EpsNetPlan(nets.Ctx(DEV), weights("DiffUNet1"), B, T, time_cond=True, nsteps=1, planes=planes)
    net.build_time()
    net.build_step(0)
    net.finish()
    return net


def test_f16x2_blocks_vs_goldens(L, weights):
    """DiffUNet1 forward with all 15 BiConv(Trans)GLU stages in the f16x2 form against the reference's golden vectors (small T
    with the TCM input as an intermediate; T = 401), tolerance 2e-5 as for every fp32-equivalent arithmetic of the block; the
    distance to the three-plane bf16 split is printed beside it."""
    g = golden("diffunet1_small")
    B, T = int(g["B"]), int(g["T"])
    x = seeded((B, 2, T, 161), g["seed_x"])
    xi = seeded((B, 2, T, 161), g["seed_init"]) * float(g["init_scale"])
    outs = {}
    for planes in (2, 3):
        net = _eps_plan(weights, B, T, planes)
        net.x.copy_(x)
        net.x_init.copy_(xi)
        net.tsteps.copy_(torch.from_numpy(g["t"]).view(1, B))
        net.plan.run()
        torch.cuda.synchronize()
        bg = [d for d, _ in net.descs if isinstance(d, L.BgluDesc)]
        assert len(bg) == 15 and all(d.np == planes for d in bg)
        e5 = rel_l2(net.en[4].cpu().permute(0, 1, 3, 2), g["en5"])
        e = rel_l2(net.out.cpu(), g["out"])
        print("planes %d: encoder output vs golden %.2e | eps-net vs golden %.2e" % (planes, e5, e))
        assert e5 < 2e-5 and e < 2e-5
        outs[planes] = net.out.cpu().clone()
        if planes == 2:
            Pk = pkg("packing")
            for hp, enc in [(hp, True) for hp in net.hp_en.values()] + [(hp, False) for hp in net.hp_de.values()]:   # margins stay zero
                raw = hp.cpu().numpy().view(np.uint16)[:B]
                if enc and net.parity_planes:
                    raw = raw[:, :, :, :, Pk.hp_par_pos(raw.shape[4]), :]
                full = Pk.hp_join(raw, with_margins=True)
                assert not full[:, :, :, :2].any() and not full[:, :, :, -2:].any() and (enc or not full[:, :, 0].any())
    print("f16x2 vs bf16x3: %.2e" % rel_l2(outs[2], outs[3]))
    assert rel_l2(outs[2], outs[3]) < 1e-5
    g4 = golden("diffunet1_t401")
    net = _eps_plan(weights, 1, 401, 2)
    net.x.copy_(seeded((1, 2, 401, 161), g4["seed_x"]))
    net.x_init.copy_(seeded((1, 2, 401, 161), g4["seed_init"]) * 0.3)
    net.tsteps.fill_(float(g4["t"]))
    net.plan.run()
    torch.cuda.synchronize()
    out4 = net.out.cpu()
    assert rel_l2(out4[0, :, ::16, :], g4["rows"]) < 2e-5
    assert abs(float(out4.double().pow(2).sum()) - float(g4["sumsq"])) < 4e-5 * float(g4["sumsq"])


def test_f16x2_gcrn_prior_vs_goldens(L, weights):
    """GCRN's gated (transposed) convolutions and LSTM input projection as f16x2 GEMMs (csrc/gconv4.hip, korder 5: two fp16 planes of
    W * 2^wexp streamed through the ring, activations scaled and split in registers) against the reference's golden (2e-5, as the
    three-plane form), with the intermediates the fixture holds, and at T = 401."""
    nets = pkg("nets")
    g = golden("gcrn_small")
    x = seeded((2, 2, 20, 161), g["seed_x"])
    outs = {}
    for planes in (2, 3):
        net = nets.GcrnPlan(nets.Ctx(DEV), weights("GCRN"), 2, 20, planes=planes)
        net.build()
        net.finish()
        net.x.copy_(x)
        net.plan.run()
        torch.cuda.synchronize()
        k = [d.korder for d, _ in net.descs if isinstance(d, L.GconvDesc) and d.korder >= 3]
        assert len(k) == 22 and all(v == (5 if planes == 2 else 3) for v in k)
        outs[planes] = net.out.cpu().clone()
        e, e5, el = rel_l2(outs[planes], g["out"]), rel_l2(net.enc_out(5).cpu(), g["e5"]), rel_l2(net.glstm_out().cpu(), g["glstm"])
        print("GCRN planes %d: encoder %.2e | LSTM block %.2e | output vs golden %.2e" % (planes, e5, el, e))
        assert e < 2e-5 and e5 < 2e-5 and el < 2e-5
    print("GCRN f16x2 vs bf16x3: %.2e" % rel_l2(outs[2], outs[3]))
    assert rel_l2(outs[2], outs[3]) < 5e-6
    g4 = golden("gcrn_t401")
    net = nets.GcrnPlan(nets.Ctx(DEV), weights("GCRN"), 1, 401, planes=2)
    net.build()
    net.finish()
    net.x.copy_(seeded((1, 2, 401, 161), g4["seed_x"]))
    net.plan.run()
    torch.cuda.synchronize()
    assert rel_l2(net.out.cpu()[0, :, ::16, :], g4["rows"]) < 5e-5          # 401 recurrent frames: the bound of test_gcrn_golden


@pytest.mark.parametrize("name,fixture", [("aia_complex_trans_ri", "aia_small"), ("dual_aia_trans_merge_crm", "dual_aia_small")])
def test_f16x2_aia_priors_vs_goldens(L, weights, name, fixture):
    """Both DB-AIAT priors with their dense-block layers (csrc/dense.hip, np 2) and GEMM-shaped convolutions (korder 5) in the f16x2
    form against the reference's goldens (5e-5, the bound of the three-plane form's test), beside the three-plane form."""
    nets = pkg("nets")
    g = golden(fixture)
    x = seeded((2, 2, 12, 161), g["seed_x"])
    cls = nets.AiaPlan if name == "aia_complex_trans_ri" else nets.DualAiaPlan
    outs = {}
    for planes in (2, 3):
        net = cls(nets.Ctx(DEV), weights(name), 2, 12, planes=planes)
        net.build()
        net.finish()
        net.x.copy_(x)
        net.plan.run()
        torch.cuda.synchronize()
        dn = [d for d, _ in net.descs if isinstance(d, L.DenseDesc)]
        assert dn and all(d.np == planes for d in dn) and all(d.korder != (3 if planes == 2 else 5) for d, _ in net.descs if isinstance(d, L.GconvDesc))
        outs[planes] = net.out.cpu().clone()
        e = rel_l2(outs[planes], g["out"])
        print("%s planes %d: vs golden %.2e" % (name, planes, e))
        assert e < 5e-5
    print("%s f16x2 vs bf16x3: %.2e" % (name, rel_l2(outs[2], outs[3])))
    assert rel_l2(outs[2], outs[3]) < 2e-5


def test_f16x2_sampling_full_size_b32(L, weights):
    """BASELINE config 2 (B = 32, T = 401, GCRN + 6 reverse steps) with split="f16x2": utterances 0 / 17 / 31 bit-identical to their
    B = 1 runs, utterance 0 against the reference's own fp32 loop and its float64 evaluation (<= 1e-4, as the bf16x3 pass);
    the small sampling goldens (6 and 50 steps)."""
    B, T = 32, 401
    feat, x_T = pkg("synth").synthetic_spectrogram(B, T, seed=1234)
    P = pkg("pipeline").SamplerPipeline
    big = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, T=T, split="f16x2")
    assert big.split == "f16x2" and all(d.np == 2 for d, _ in big.descs if isinstance(d, L.BgluDesc))
    assert sum(1 for d, _ in big.descs if isinstance(d, L.GconvDesc) and d.korder == 5) == 22 and not any(isinstance(d, L.GconvDesc) and d.korder == 3 and d.epi == L.EPI_GLU for d, _ in big.descs)
    spec, init = big.sample(feat.to(DEV), x_T.to(DEV))
    bank = big.bank
    del big
    one = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, T=T, split="f16x2", bank=bank)
    for b in (0, 17, 31):
        s1, _ = one.sample(feat[b:b + 1].to(DEV), x_T[b:b + 1].to(DEV))
        assert torch.equal(s1[0], spec[b]), b
    ref, exact, _ = full_pair("full_gcrn_seed1234_t401_6step")
    e_ref, e_exact, e_ref_exact = rel_l2(spec[:1].cpu(), ref), rel_l2(spec[:1].cpu(), exact), rel_l2(ref, exact)
    print("f16x2, 6 steps, B=32: vs the reference's fp32 run %.2e | vs its float64 evaluation %.2e | fp32 run vs float64 %.2e" % (e_ref, e_exact, e_ref_exact))
    assert e_ref < 1e-4 and e_exact < 1e-4
    for tag, fast in (("gcrn_fast", True), ("gcrn_full", False)):        # the reference's small sampling fixtures: 6 and 50 steps
        g = golden("sample_" + tag)
        f2, x2 = seeded((2, 2, 16, 161), g["seed_feat"]), seeded((2, 2, 16, 161), g["seed_xT"])
        pipe = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 2, T=16, fast_sampling=fast, split="f16x2", split_bf16=True)
        assert all(d.np == 2 for d, _ in pipe.descs if isinstance(d, L.BgluDesc))
        out, _ = pipe.sample(f2.to(DEV), x2.to(DEV))
        e = rel_l2(out.cpu(), g["out"])
        print("f16x2 sampling golden %s: %.2e" % (tag, e))
        assert e < 1e-4


@pytest.mark.parametrize("scale_x,scale_w", [(2.0 ** -6, 1.0), (0.25, 1.0), (4.0, 1.0), (1.0, 2.0 ** -5), (1.0, 0.5)])
def test_f16x2_is_fp32_equivalent_across_input_and_weight_scales(L, weights, scale_x, scale_w):
    """What "fp32-equivalent" means away from the unit-scale fixtures.  The eps-net is evaluated on inputs 64 times smaller and 4 times
    larger than the fixtures' spectrograms, and with the gather weights of every stage scaled (the host picks a power-of-two exponent
    per weight group, so weight magnitude must not matter), by: the oracle in float64 (the exact answer), the oracle in fp32, the
    f16x2 kernels and the exact-fp32 MFMA kernels.  The random network amplifies rounding noise with the input scale (at 16 x
    every fp32 implementation, the reference's own arithmetic included, is 0.2 from the float64 answer; with the gather weights doubled
    the activations grow by 2 per stage, the fp32 oracle is 0.35 from it and the f16x2 pass ends non-finite - loudly, see the next test),
    so the bound is relative: the
    f16x2 kernels must sit no further from the exact answer than 1.5 x the fp32 oracle does (+ 1e-6)."""
    import importlib

    R = importlib.import_module("oracle.restate")
    nets = pkg("nets")
    sd = dict(weights("DiffUNet1"))
    if scale_w != 1.0:
        for k in list(sd):
            if k.endswith((".l.weight", ".r.weight")):
                sd[k] = sd[k] * scale_w
    B, T = 2, 40
    x, xi = seeded((B, 2, T, 161), 31) * scale_x, seeded((B, 2, T, 161), 32) * (0.3 * scale_x)
    t = torch.full((B,), 10.451817)
    with torch.no_grad():
        ref32 = R.diffunet1_forward(sd, x, xi, t)
        exact = R.diffunet1_forward({k: v.double() for k, v in sd.items()}, x.double(), xi.double(), t.double())
    outs = {}
    for tag, kw in (("f16x2", dict(planes=2)), ("fp32", dict(split_bf16=False))):
        net = nets.EpsNetPlan(nets.Ctx(DEV), sd, B, T, time_cond=True, nsteps=1, **kw)
        net.build_time()
        net.build_step(0)
        net.finish()
        net.x.copy_(x)
        net.x_init.copy_(xi)
        net.tsteps.copy_(t.view(1, B))
        net.plan.run()
        torch.cuda.synchronize()
        assert (tag == "fp32") == (not any(isinstance(d, L.BgluDesc) for d, _ in net.descs))
        outs[tag] = net.out.cpu().clone()
    e16, e32k, e32o = rel_l2(outs["f16x2"], exact), rel_l2(outs["fp32"], exact), rel_l2(ref32, exact)
    print("input scale %g, gather-weight scale %g: distance to the float64 answer - f16x2 %.2e | fp32 MFMA kernels %.2e | fp32 oracle %.2e" % (
        scale_x, scale_w, e16, e32k, e32o))
    assert torch.isfinite(outs["f16x2"]).all() and e16 < 1.5 * e32o + 1e-6


def test_f16x2_window_overflow_is_loud_and_the_trainer_falls_back(L, weights, tmp_path):
    """An activation beyond the fp16 window (|value| 2^4 > 65504) must not pass for a value: its planes are infinities (IEEE
    conversion), the pass ends non-finite, SamplerPipeline.check() raises PdseRangeError - and the drop-in trainer repeats the
    geometry on the three-plane bf16 split, whose result it returns (bit-identical to a pipeline built with split="bf16x3")."""
    import argparse

    sd = dict(weights("DiffUNet1"))
    sd["en.conv3.conv1.bias"] = sd["en.conv3.conv1.bias"] + 3.0e4          # conv1 output of encoder stage 3: far outside +-4094
    wav, x_T = pkg("synth").synthetic_waveforms(2, 4000, seed=77)
    P = pkg("pipeline").SamplerPipeline
    pipe = P(DEV, "GCRN", weights("GCRN"), sd, 2, L_=4000)
    assert pipe.split == "f16x2"
    pipe.enhance(wav.to(DEV), x_T.to(DEV))
    with pytest.raises(L.PdseRangeError, match="fp16"):
        pipe.check()
    ref = P(DEV, "GCRN", weights("GCRN"), sd, 2, L_=4000, split="bf16x3")
    want, _ = ref.enhance(wav.to(DEV), x_T.to(DEV))
    ref.check()
    assert torch.isfinite(want).all()
    ns = argparse.Namespace
    tr = pkg("trainer").ComplexDDPMTrainer(
        ns(retrain=False, joint=True, draw=False, sigma=False, checkpoint="x", generated_wav=str(tmp_path)),
        ns(model=ns(name="GCRN"), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt")),
        device=DEV, prior_state_dict=weights("GCRN"), ddpm_state_dict=sd, exclusive=False)
    got = tr.enhance_batch([wav[0], wav[1]], x_T=x_T)
    assert len(tr._range_fallback) == 1 and torch.equal(torch.stack(got), want)
    got2 = tr.enhance_batch([wav[0], wav[1]], x_T=x_T)                      # the geometry stays on the fallback: no second warning, same bits
    assert torch.equal(torch.stack(got2), want)


def test_f16x2_planes_of_values_beyond_the_window_are_infinities(L):
    """The split kernel against its host restatement bit for bit; values inside the window come back within half an fp32 ulp, a value
    beyond it as hi = +-inf, lo = -+inf (never a clipped number)."""
    Pk = pkg("packing")
    B, T, F = 1, 3, 5
    top = 65504.0 / 2 ** Pk.F16_ACT_EXP                                       # the largest plane value in true scale
    x = seeded((B, 32, T, F), 21)
    x[0, 3, 1, 2] = 1.0e6
    x[0, 4, 1, 2] = -2.0 * top
    x[torch.abs(x) < 2.0 ** (-2 - Pk.F16_ACT_EXP)] = 0.5
    shp = Pk.hp_shape(B, T, F, 2)
    hp = torch.zeros(*shp, dtype=torch.int16, device=DEV)
    d = L.PlanesDesc()
    xd = x.to(DEV)
    d.in_, (d.in_sb, d.in_sc, d.in_st, d.in_sf) = xd.data_ptr(), (32 * T * F, T * F, F, 1)
    d.hp, d.hp_sb, d.hp_Tp, d.hp_Fp, d.hp_t0, d.hp_f0 = hp.data_ptr(), int(np.prod(shp[1:])), shp[1], shp[4], Pk.HP_T0, Pk.HP_F0
    d.B, d.T, d.F, d.np = B, T, F, 2
    plan = L.Plan(torch.device(DEV))
    plan.add(d)
    plan.run()
    torch.cuda.synchronize()
    raw = hp.cpu().numpy().view(np.uint16)
    assert np.array_equal(raw, Pk.hp_split(x.numpy(), 2))                     # the host restatement of the split, bit for bit
    with np.errstate(invalid="ignore"):
        back = Pk.hp_join(raw)
    xs = x.numpy().copy()
    inside = np.abs(xs) < 0.97 * top
    assert np.max(np.abs(back - xs)[inside] / np.abs(xs)[inside]) <= 2.0 ** -23
    assert not np.isfinite(back[0, 3, 1, 2]) and not np.isfinite(back[0, 4, 1, 2]) and np.isfinite(back[inside]).all()


def test_rccl_leg_of_the_sharded_path_single_rank(L, weights, tmp_path):
    """the branch of shard.gather_shards that hands DEVICE buffers to RCCL (backend "nccl"), and the barrier / max-reduce
    of bench.py's timed region, executed on this box's one GPU in a process group of one rank (two ranks per device are refused by
    RCCL; the gloo tests cover world size 2): the gathered tensor equals the rank's own result bit for bit."""
    from conftest import ROOT

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 1000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_worker.py"), str(tmp_path)], env=env, timeout=600)
    assert p.returncode == 0
    got = torch.load(os.path.join(str(tmp_path), "rccl.pt"))
    assert got["full"].shape == (5, 4000) and torch.equal(got["full"], got["local"])
