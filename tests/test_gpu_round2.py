"""GPU parity, round 2: the BASELINE configurations at their full per-GPU sizes (config 3: B=32, 50 steps; config 4:
B=32 with the DB-AIAT priors; config 5: B=16 x 10 s through STFT..ISTFT), the variable-length ``generate_wav`` path on
the shared weight bank, the real sharded path on two ranks, and the rows widened this round (third conditioning
branch, --sigma on deltamu, q_sample branches, masked validation loss).  Everything goes through the C-ABI.

Tolerances as in test_gpu_parity.py: network forwards 2e-5 / 5e-5, sampling 1e-4 rel-L2 (north star), elementwise
arithmetic bit-exact, batch invariance bit-exact."""
import argparse
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_rows2_and_checksums, full_pair, golden, pkg, rel_l2, seeded, tcm2_blocks

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def R():
    from oracle import restate

    return restate


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge

    ge.build()
    lib = pkg("_lib")
    lib.load()
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return lib


def _trainer(weights, prior="GCRN", ddpm="DiffUNet1", sigma=False, params=None, out="y", **kw):
    ns = argparse.Namespace
    return pkg("trainer").ComplexDDPMTrainer(
        ns(retrain=False, joint=True, draw=False, sigma=sigma, checkpoint="x", generated_wav=out),
        ns(model=ns(name=prior), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt")),
        device=DEV, prior_state_dict=weights(prior), ddpm_state_dict=weights(ddpm), params=params, **kw)


def _errors_vs_fp32_and_exact(fixture, got):
    """rel-L2 of a HIP result against (a) the reference's fp32 CPU run and (b) the same statements evaluated in float64
    (tests/golden/full_*.npz, oracle/make_golden_full.py), plus (c) the fp32 run's own distance to (b).  Over 50 reverse steps the random eps-net amplifies rounding noise
    (DESIGN.md §2): two correct fp32 implementations differ by about the sum of their distances to the exact answer, so
    the 1e-4 bound is asserted against the exact evaluation and, against the fp32 path, with that path's own noise
    added - both numbers are printed."""
    ref, exact, _ = full_pair(fixture)
    return rel_l2(got, ref), rel_l2(got, exact), rel_l2(ref, exact)


# ------------------------------------------------------------------ BASELINE configs at full per-GPU size
@pytest.mark.parametrize("prior", ["aia_complex_trans_ri", "dual_aia_trans_merge_crm"])
def test_config4_aia_prior_b32_t401(L, weights, R, prior):
    """BASELINE config 4 per GPU: B=32 x 4 s, DB-AIAT prior + 6-step sampling.  Attention / GRU grids at B=32
    (2,560-12,832 lines): utterances 0 / 17 / 31 equal their own B=1 runs bit for bit, utterance 0 meets the oracle."""
    params = pkg("params").params
    B, T = 32, 401
    feat, x_T = pkg("synth").synthetic_spectrogram(B, T, seed=404)
    P = pkg("pipeline").SamplerPipeline
    big = P(DEV, prior, weights(prior), weights("DiffUNet1"), B, T=T)
    spec, init = big.sample(feat.to(DEV), x_T.to(DEV))
    bank = big.bank
    del big
    one = P(DEV, prior, weights(prior), weights("DiffUNet1"), 1, T=T, bank=bank)
    for b in (0, 17, 31):
        s1, i1 = one.sample(feat[b:b + 1].to(DEV), x_T[b:b + 1].to(DEV))
        assert torch.equal(i1[0], init[b]), b
        assert torch.equal(s1[0], spec[b]), b
    assert torch.isfinite(spec).all()
    g = golden("full_%s_seed404_t401_6step" % prior)        # the reference's own loop on model/dbaiat.py + model/diff3.py
    assert rel_l2(init[0].cpu(), g["init"][0]) < 1e-4
    assert rel_l2(spec[0].cpu(), g["out"][0]) < 1e-4


def test_config5_long_utterances_b16_t1001(L, weights, R):
    """BASELINE config 5 per GPU: B=16 x 10 s (L = 160,000, T = 1001) through the whole path incl. STFT / ISTFT:
    two utterances equal their B=1 runs bit for bit, utterance 0 meets the oracle (spectrogram and waveform)."""
    params = pkg("params").params
    B, L_ = 16, 160000
    wav, x_T = pkg("synth").synthetic_waveforms(B, L_, seed=505)
    wav = wav * torch.linspace(0.05, 2.0, B)[:, None]              # per-utterance RMS must cancel
    P = pkg("pipeline").SamplerPipeline
    big = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, L_=L_)
    assert big.T == 1001
    out, spec = big.enhance(wav.to(DEV), x_T.to(DEV))
    bank = big.bank
    del big
    one = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, L_=L_, bank=bank)
    for b in (0, 11):
        o1, s1 = one.enhance(wav[b:b + 1].to(DEV), x_T[b:b + 1].to(DEV))
        assert torch.equal(s1[0], spec[b]) and torch.equal(o1[0], out[b]), b
    g = golden("full_generate_wav_seed505_l160000")        # generate_wav's per-file statements executed from the reference's text
    assert_rows2_and_checksums(spec[:1].cpu(), g, "spec_", 1e-4)
    assert rel_l2(out[0].cpu(), g["wav"]) < 1e-4


def test_config3_full_schedule_b32(L, weights, R):
    """BASELINE config 3: B=32, full 50-step schedule, fp32: two utterances equal the B=1 run bit for bit (whose
    agreement with the oracle over 50 steps is test_full_50_step_schedule_at_t401), schedule indices bit-exact."""
    B, T = 32, 401
    feat, x_T = pkg("synth").synthetic_spectrogram(B, T, seed=78)
    f1, x1 = pkg("synth").synthetic_spectrogram(1, T, seed=77)     # utterance 0: the input whose 50-step oracle and float64
    feat[0], x_T[0] = f1[0], x1[0]                                   # evaluations are shared with the B=1 tests (memoised)
    P = pkg("pipeline").SamplerPipeline
    big = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, T=T, fast_sampling=False)
    assert big.nsteps == 50 and np.array_equal(big.schedule[4], np.arange(50, dtype=np.float32))
    spec, init = big.sample(feat.to(DEV), x_T.to(DEV))
    bank = big.bank
    del big
    one = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, T=T, fast_sampling=False, bank=bank)
    for b in (0, 23):
        s1, _ = one.sample(feat[b:b + 1].to(DEV), x_T[b:b + 1].to(DEV))
        assert torch.equal(s1[0], spec[b]), b
    e_ref, e_exact, e_ref_exact = _errors_vs_fp32_and_exact("full_gcrn_seed77_t401_50step", spec[:1].cpu())
    print("config 3 (50 steps): HIP vs fp32 CPU oracle %.2e | HIP vs float64 evaluation %.2e | fp32 CPU oracle vs float64 "
          "evaluation %.2e" % (e_ref, e_exact, e_ref_exact))
    assert e_exact < 1e-4                                   # distance to the exact-arithmetic answer
    assert e_ref < 1e-4 + e_ref_exact                       # the fp32 CPU path carries its own rounding noise of that size


# ------------------------------------------------------------------ the reference's entry point: B=1, a new length per file
def test_generate_wav_many_lengths_shared_weights(L, weights, tmp_path):
    """HOT LOOP 1 (trainer/complex_ddpm_trainer.py:917-1018): 20 files of 20 different lengths.  Every file equals a
    stand-alone pipeline run on the same generator stream (incl. the reference's discarded per-step draws), a new
    length costs milliseconds of host time (descriptors only - weights are packed once), and HBM does not grow with
    the number of lengths seen."""
    wavio, P = pkg("wavio"), pkg("pipeline")
    rng = np.random.default_rng(5)
    data = tmp_path / "noisy"
    data.mkdir()
    lens = [3200 + 317 * i for i in range(20)]
    for i, n in enumerate(lens):
        wavio.write_wav(str(data / ("f%02d.wav" % i)), 0.1 * rng.standard_normal(n))
    t = _trainer(weights, out=str(tmp_path / "out"))
    torch.manual_seed(99)
    torch.cuda.manual_seed_all(99)
    t0 = time.perf_counter()
    written = t.generate_wav(load_pre_train=False, data_path=str(data))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert len(written) == 20 and len(t._pipes) <= t.MAX_PLANS
    bank_bytes = t.bank.nbytes()
    mem_after = torch.cuda.memory_allocated()
    # replay on the same stream of draws with stand-alone pipelines (own weight bank each)
    torch.manual_seed(99)
    torch.cuda.manual_seed_all(99)
    for i, n in enumerate(lens[:6]):
        wav = torch.from_numpy(wavio.read_wav(str(data / ("f%02d.wav" % i))))[None].to(DEV)
        T = 1 + n // 160
        x_T = torch.randn(1, 2, T, 161, device=DEV)
        pipe = P.SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, L_=n, exclusive=True)   # as the trainer builds it
        out, _ = pipe.enhance(wav, x_T)
        for _ in range(5):
            torch.randn(1, 2, T, 161, device=DEV)
        ref_path = str(tmp_path / "ref.wav")
        wavio.write_wav(ref_path, out[0].cpu().numpy())
        assert np.array_equal(wavio.read_wav(ref_path), wavio.read_wav(sorted(written)[i])), i
        del pipe
    # a new length: host time of recording the plan, weights already resident
    torch.cuda.synchronize()
    times = []
    for n in (4001, 5003, 6007, 7013, 8017):
        t1 = time.perf_counter()
        t._pipe(1, L_=n)
        times.append(time.perf_counter() - t1)
    assert t.bank.nbytes() == bank_bytes                     # nothing was re-packed or re-uploaded
    assert sorted(times)[len(times) // 2] < 0.020, times    # median plan build < 20 ms
    # more new lengths, none longer than the ones alive when mem_after was taken (the three longest of the first 20):
    # the LRU keeps MAX_PLANS buffer sets, so HBM in use cannot exceed that mark however many lengths go by
    for n in range(3301, 9000, 431):
        t.enhance(torch.zeros(1, n, device=DEV) + 0.01)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= mem_after + (1 << 20), (n, torch.cuda.memory_allocated(), mem_after)
    print("generate_wav: 20 files (%.1f s audio) in %.2f s wall; new-length plan build %s ms; bank %.1f MB" % (
        sum(lens) / 16000.0, wall, [round(1e3 * x, 2) for x in times], bank_bytes / 1e6))


def test_real_sharded_path_two_ranks_equals_single_rank(L, weights, tmp_path):
    """SURVEY 8e with the REAL pipeline: two ranks (gloo rendezvous, both on this GPU) each enhance their contiguous
    shard of a 5-utterance global batch; the gathered waveforms equal the single-rank run bit for bit."""
    port = 29700 + os.getpid() % 1000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "shard_worker.py"),
                                       str(tmp_path)], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    wav, x_T = pkg("synth").synthetic_waveforms(5, 4000, seed=21)
    t = _trainer(weights, exclusive=False)      # as the ranks of a distributed job build it: the kernels that do not depend on the batch size
    ref = t.enhance(wav, x_T=x_T).cpu()
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), "out%d.pt" % r))
        assert torch.equal(got, ref), r


def test_plan_is_bound_to_its_device(L):
    """a plan launches on the device its buffers live on, whatever device is current on the thread."""
    plan = L.Plan(DEV)
    x = torch.arange(8, dtype=torch.float32, device=DEV)
    out = torch.empty_like(x)
    d = L.EwDesc()
    d.a, d.out, d.n, d.s0, d.op = x.data_ptr(), out.data_ptr(), 8, 2.0, L.EW_DIV
    plan.add(d)
    plan.run(torch.cuda.current_stream(DEV).cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, x / 2) and torch.cuda.current_device() == 0
    import ctypes as C

    rc = L.load().pdse_plan_set_device(plan._h, torch.cuda.device_count() + 3)
    assert rc != 0 and b"no such device" in L.load().pdse_last_error()
    if torch.cuda.device_count() > 1:                       # multi-GPU box: run on device 1 while device 0 is current
        dev1 = "cuda:1"
        p1 = L.Plan(dev1)
        y = torch.arange(8, dtype=torch.float32, device=dev1)
        o1 = torch.empty_like(y)
        d1 = L.EwDesc()
        d1.a, d1.out, d1.n, d1.s0, d1.op = y.data_ptr(), o1.data_ptr(), 8, 4.0, L.EW_DIV
        p1.add(d1)
        p1.run(torch.cuda.current_stream(dev1).cuda_stream)
        torch.cuda.synchronize(dev1)
        assert torch.equal(o1, y / 4) and torch.cuda.current_device() == 0
    del C


# ------------------------------------------------------------------ rows widened this round
@pytest.mark.parametrize("tag,ddpm,kw", [
    ("gcrn_fast_featcond", "DiffUNet1", dict(cond="feat")),
    ("gcrn_fast_featcond_sigma", "DiffUNet1", dict(cond="feat", use_sigma=True)),
    ("gcrn_fast_deltamu_sigma", "Nocon", dict(deltamu=True, use_sigma=True)),
])
def test_sampling_branches_vs_reference_fixture(L, weights, tag, ddpm, kw):
    """Third conditioning branch (neither pirorgrad nor deltamu, :74-75, :972-974) and --sigma on deltamu
    (:947-956); fixtures from the reference's own statements (oracle/make_golden.py::ref_generate_body)."""
    g = golden("sample_" + tag)
    feat, x_T = seeded((2, 2, 16, 161), g["seed_feat"]), seeded((2, 2, 16, 161), g["seed_xT"])
    pipe = pkg("pipeline").SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights(ddpm), 2, T=16, **kw)
    spec, init = pipe.sample(feat.to(DEV), x_T.to(DEV))
    assert rel_l2(init.cpu(), g["init"]) < 2e-5
    assert rel_l2(spec.cpu(), g["out"]) < 1e-4


def test_trainer_selects_the_conditioning_branch(L, weights):
    P = pkg("params")
    g = golden("sample_gcrn_fast_featcond_sigma")
    feat, x_T = seeded((2, 2, 16, 161), g["seed_feat"]), seeded((2, 2, 16, 161), g["seed_xT"])
    prm = P.AttrDict(dict(P.params))
    prm.pirorgrad = False                                   # neither flag: DiffUNet1(audio, batch_feat, t)
    t = _trainer(weights, sigma=True, params=prm)
    assert t.cond == "feat" and not t.deltamu
    assert rel_l2(t.sample(feat, x_T).cpu(), g["out"]) < 1e-4
    prm2 = P.AttrDict(dict(P.params))
    prm2.deltamu = True                                     # both flags: the reference's if/elif takes pirorgrad (:70-73)
    t2 = _trainer(weights, params=prm2)                     # ... for the model and the eps call; x_T still follows deltamu
    assert t2.cond == "init" and not t2.deltamu and t2.xT_plus_init            # (:946-949): noise + X_init/11, final + X_init
    gb = golden("sample_gcrn_fast_bothflags")               # the reference's own statements with both flags set
    assert rel_l2(t2.sample(feat, x_T).cpu(), gb["out"]) < 1e-4
    assert rel_l2(gb["out"], golden("sample_gcrn_fast")["out"]) > 1e-2         # a different result from pirorgrad alone


def test_q_sample_branches_bit_exact(L, weights, R):
    """Training-step forward noising, all three parameterisations with and without the --sigma mask (:704-729)."""
    params = pkg("params").params
    g = torch.Generator().manual_seed(6)
    label, init, noise = (torch.randn(3, 2, 20, 161, generator=g) for _ in range(3))
    t = torch.tensor([0, 17, 49])
    for mode in ("pirorgrad", "deltamu", "plain"):
        for sigma in (False, True):
            # --sigma: noise * mask ** 0.5.  torch's vectorised CPU sqrt is not correctly rounded (it differs from the
            # IEEE result in ~0.7 % of elements); the HIP kernel - like the reference's CUDA path - uses IEEE sqrt, so
            # the expected masked noise is formed with numpy's (correctly rounded) sqrt
            nz = noise * torch.from_numpy(np.sqrt(R.sigma_mask(init).numpy())) if sigma else noise
            ref = R.q_sample(label, init, t, nz, params.noise_schedule, mode, False)
            got = pkg("ops").q_sample(label.to(DEV), init.to(DEV), t.to(DEV), noise.to(DEV), mode=mode, sigma=sigma)
            torch.cuda.synchronize()
            diff = (got.cpu() - ref).abs()
            assert torch.equal(got.cpu(), ref), (mode, sigma, int((diff > 0).sum()), float(diff.max()))


def test_masked_validation_loss_and_ragged_batch_vs_reference_fixture(L, weights):
    """SURVEY 8f rank 2 against the batch the reference's own validation-loop text produced
    (oracle/make_golden.py::ref_validation_batch): the masked complex MSE (utils/loss.py:34-44) on the fixture's
    tensors, and the drop-in's ragged batch (per-utterance c, zero padding, trim to (frame_num - 1) * 160)."""
    g = golden("ragged_validation")
    ops = pkg("ops")
    frames = [int(n) for n in g["frame_list"]]
    loss = ops.com_mse_loss(torch.from_numpy(g["audio"]).to(DEV), torch.from_numpy(g["label"]).to(DEV), frames)
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * float(g["loss"])
    with pytest.raises(ValueError):
        ops.com_mse_loss(torch.zeros(2, 2, 4, 161, device=DEV), torch.zeros(2, 2, 4, 161, device=DEV), [5, 1])
    gen = torch.Generator().manual_seed(int(g["seed"]))
    lens = [int(n) for n in g["lens"]]
    wavs = [0.2 * torch.randn(n, generator=gen) for n in lens]
    x_T = torch.randn(3, 2, 1 + max(lens) // 160, 161, generator=gen)
    t = _trainer(weights)
    got = t.enhance_batch(wavs, x_T=x_T, trim_to_frames=True)
    for i, (w, o) in enumerate(zip(wavs, got)):
        c = float(np.sqrt(w.numel() / float((w.double() ** 2).sum())))      # the reference's metric stays normalised
        ref = g["utt%d" % i]
        assert o.numel() == ref.shape[0]
        assert rel_l2(o.cpu() * c, ref) < 1e-4
    # the batch's spectrogram and the loss on it, device to device
    B, L_ = 3, max(lens)
    pipe = t._pipe(B, L_=L_)
    spec = pipe.spec
    assert rel_l2(spec.cpu(), g["audio"]) < 1e-4
    loss2 = ops.com_mse_loss(spec.contiguous(), torch.from_numpy(g["label"]).to(DEV), frames)
    assert abs(float(loss2) - float(g["loss"])) <= 1e-4 * float(g["loss"])


def test_aia_prior_long_sequence_t1001(L, weights, R):
    """DB-AIAT priors on 10 s utterances: the column attention walks its 1001 keys in LDS-sized chunks
    (model/dbaiat.py:91-154 takes any T).  Oracle at T = 1001, and the attention core alone against a plain softmax
    at a length that is neither a multiple of the chunk nor of four."""
    x = seeded((1, 2, 1001, 161), 93)
    got = pkg("ops").AiaOp(weights("aia_complex_trans_ri"), DEV)(x.to(DEV)).cpu()
    assert_rows2_and_checksums(got, golden("full_aia_seed93_t1001"), "", 1e-4)     # the reference module at T = 1001
    # attention core: qkv [B, 3E, T, F], sequence over frames (axis 1), E = 32, 4 heads of 8
    B, E, T, F_ = 2, 32, 1103, 3
    g = torch.Generator().manual_seed(8)
    qkv = torch.randn(B, 3 * E, T, F_, generator=g).to(DEV)
    out = torch.empty(B, E, T, F_, device=DEV)
    d = L.AttnDesc()
    d.qkv, d.out, d.B, d.T, d.F, d.E, d.heads, d.axis = qkv.data_ptr(), out.data_ptr(), B, T, F_, E, 4, 1
    L.launch(d)
    torch.cuda.synchronize()
    q, k, v = (t_.view(B, 4, 8, T, F_).permute(0, 4, 1, 3, 2).double() for t_ in qkv.split(E, dim=1))   # [B,F,h,T,8]
    ref_att = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v                                        # q is pre-scaled
    ref_att = ref_att.permute(0, 2, 4, 3, 1).reshape(B, E, T, F_)
    assert rel_l2(out.cpu(), ref_att.cpu()) < 2e-6


# ------------------------------------------------------------------ split-bf16 BIGLU blocks (csrc/gconv3.hip)
def test_split_bf16_networks_vs_goldens(L, weights, monkeypatch):
    """The eps-net with its BIGLU blocks on the bf16 matrix cores (exact three-way operand splits, six products, fp32
    accumulate) against the SAME golden vectors and tolerances as the fp32 path: small T with the TCM input as an
    intermediate, integer steps, T = 401 rows, the DiffUNet prior; and its distance to the fp32 kernels."""
    nets, ops = pkg("nets"), pkg("ops")
    g = golden("diffunet1_small")
    B, T = int(g["B"]), int(g["T"])
    x = seeded((B, 2, T, 161), g["seed_x"])
    xi = seeded((B, 2, T, 161), g["seed_init"]) * float(g["init_scale"])
    t = torch.from_numpy(g["t"]).to(DEV)
    monkeypatch.setattr(nets.EpsNetPlan, "split_bf16", False)
    op32 = ops.DiffUNet1Op(weights("DiffUNet1"), DEV)
    out32 = op32(x.to(DEV), xi.to(DEV), t)
    assert all(d.korder == 1 for d, _ in op32._plans[(B, T)].descs if isinstance(d, L.GconvDesc))
    monkeypatch.setattr(nets.EpsNetPlan, "split_bf16", True)
    op = ops.DiffUNet1Op(weights("DiffUNet1"), DEV)
    out = op(x.to(DEV), xi.to(DEV), t)
    net = op._plans[(B, T)]
    torch.cuda.synchronize()
    # the 15 BIGLU stages: on plane tensors by default (csrc/bglu.hip), on csrc/gconv3.hip with plane_h False
    assert sum(1 for d, _ in net.descs if isinstance(d, L.BgluDesc) or (isinstance(d, L.GconvDesc) and d.korder == 2)) == 15
    assert rel_l2(net.en[4].cpu().permute(0, 1, 3, 2), g["en5"]) < 2e-5
    print("split-bf16 vs golden %.2e | fp32 kernels vs golden %.2e | split vs fp32 kernels %.2e" % (
        rel_l2(out.cpu(), g["out"]), rel_l2(out32.cpu(), g["out"]), rel_l2(out.cpu(), out32.cpu())))
    assert rel_l2(out.cpu(), g["out"]) < 2e-5
    assert rel_l2(out.cpu(), out32.cpu()) < 5e-6
    gi = golden("diffunet1_int_t")
    assert rel_l2(op(x.to(DEV), xi.to(DEV), torch.from_numpy(gi["t"]).to(DEV)).cpu(), gi["out"]) < 2e-5
    g4 = golden("diffunet1_t401")
    x4 = seeded((1, 2, 401, 161), g4["seed_x"])
    xi4 = seeded((1, 2, 401, 161), g4["seed_init"]) * 0.3
    out4 = op(x4.to(DEV), xi4.to(DEV), torch.tensor([float(g4["t"])], device=DEV)).cpu()
    assert rel_l2(out4[0, :, ::16, :], g4["rows"]) < 2e-5
    gp = golden("diffunet_prior_small")
    outp = ops.DiffUNetOp(weights("DiffUNet"), DEV)(seeded(tuple(gp["out"].shape), gp["seed_x"]).to(DEV))
    assert rel_l2(outp.cpu(), gp["out"]) < 2e-5


@pytest.mark.parametrize("tag,fast", [("gcrn_fast", True), ("gcrn_full", False)])
def test_split_bf16_sampling_vs_goldens(L, weights, tag, fast):
    """6-step trace and 50-step result of the reference's loop (goldens) with the split-bf16 eps-net: <= 1e-4."""
    g = golden("sample_" + tag)
    feat, x_T = seeded((2, 2, 16, 161), g["seed_feat"]), seeded((2, 2, 16, 161), g["seed_xT"])
    pipe = pkg("pipeline").SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 2, T=16, fast_sampling=fast,
                                           split_bf16=True)
    assert pipe.split_bf16
    spec, init = pipe.sample(feat.to(DEV), x_T.to(DEV))
    e = rel_l2(spec.cpu(), g["out"])
    print("split-bf16 sampling %s: rel-L2 vs golden %.2e" % (tag, e))
    assert e < 1e-4


def test_split_bf16_full_size_b32(L, weights, R):
    """B=32, T=401, 6 steps with the split-bf16 eps-net: batch invariance bit for bit, utterance 0 vs the fp32 CPU
    oracle and vs the float64 evaluation <= 1e-4; and the 50-step schedule at T=401 (B=1) against the float64
    evaluation, printed beside the fp32 kernels' distance."""
    params = pkg("params").params
    B, T = 32, 401
    feat, x_T = pkg("synth").synthetic_spectrogram(B, T, seed=1234)
    P = pkg("pipeline").SamplerPipeline
    big = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, T=T, split_bf16=True)
    spec, init = big.sample(feat.to(DEV), x_T.to(DEV))
    bank = big.bank
    del big
    one = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, T=T, split_bf16=True, bank=bank)
    for b in (0, 17, 31):
        s1, _ = one.sample(feat[b:b + 1].to(DEV), x_T[b:b + 1].to(DEV))
        assert torch.equal(s1[0], spec[b]), b
    e_ref, e_exact, e_ref_exact = _errors_vs_fp32_and_exact("full_gcrn_seed1234_t401_6step", spec[:1].cpu())
    print("split-bf16, 6 steps, B=32: vs fp32 CPU oracle %.2e | vs float64 evaluation %.2e | fp32 CPU oracle vs float64 %.2e"
          % (e_ref, e_exact, e_ref_exact))
    assert e_ref < 1e-4 and e_exact < 1e-4
    feat1, x_T1 = pkg("synth").synthetic_spectrogram(1, T, seed=77)
    full = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, T=T, fast_sampling=False, split_bf16=True, bank=bank)
    s50, _ = full.sample(feat1.to(DEV), x_T1.to(DEV))
    e_ref, e_exact, e_ref_exact = _errors_vs_fp32_and_exact("full_gcrn_seed77_t401_50step", s50.cpu())
    print("split-bf16, 50 steps, T=401: vs fp32 CPU oracle %.2e | vs float64 evaluation %.2e | fp32 CPU oracle vs float64 %.2e"
          % (e_ref, e_exact, e_ref_exact))
    assert e_exact < 1e-4 and e_ref < 1e-4 + e_ref_exact


def test_split_bf16_gcrn_prior_vs_fp32_kernels_and_goldens(L, weights, monkeypatch):
    """GCRN's gated convolutions and LSTM input projection as split-bf16 GEMMs (csrc/gconv4.hip, the default) against the
    goldens and against the exact-fp32 kernels; the pipeline flag reaches the prior (bench.py's fp32_exact pass)."""
    nets, ops = pkg("nets"), pkg("ops")
    g = golden("gcrn_small")
    x = seeded((2, 2, 20, 161), g["seed_x"]).to(DEV)
    op = ops.GCRNOp(weights("GCRN"), DEV)
    out = op(x)
    assert sum(1 for d, _ in op._plans[(2, 20)].descs if isinstance(d, L.GconvDesc) and d.korder == 5) == 22      # the default split: f16x2
    monkeypatch.setattr(nets.GcrnPlan, "split_bf16", False)
    op32 = ops.GCRNOp(weights("GCRN"), DEV)
    out32 = op32(x)
    assert all(d.korder not in (3, 5) for d, _ in op32._plans[(2, 20)].descs if isinstance(d, L.GconvDesc))
    print("GCRN split-bf16 vs golden %.2e | fp32 kernels vs golden %.2e | split vs fp32 kernels %.2e" % (
        rel_l2(out.cpu(), g["out"]), rel_l2(out32.cpu(), g["out"]), rel_l2(out.cpu(), out32.cpu())))
    assert rel_l2(out.cpu(), g["out"]) < 2e-5 and rel_l2(out.cpu(), out32.cpu()) < 5e-6
    monkeypatch.undo()
    P = pkg("pipeline").SamplerPipeline
    p32 = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, T=20, split_bf16=False)
    assert all(d.korder in (0, 1) for d, _ in p32.descs if isinstance(d, L.GconvDesc))
    pfull = P(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, T=20, fast_sampling=False)
    assert not pfull.split_bf16 and all(d.korder in (0, 1) for d, _ in pfull.descs if isinstance(d, L.GconvDesc))


def test_split_tcm_blocks_match_fp32_kernel_and_validate_descriptors(L, weights):
    """csrc/tcm2.hip against csrc/tcm.hip on the same eps-net: the TCM stack's output (the decoders' input) from the
    split-bf16 blocks - pre-split bottleneck tensor, rotated K order - must equal the exact-fp32 blocks' to fp32 level,
    the bottleneck tensor's zero margins must survive all 19 launches, and bad descriptors are refused."""
    nets, P = pkg("nets"), pkg("packing")
    sd = weights("DiffUNet1")
    B, T = 3, 77                                      # three frame tiles, the last one ragged
    x, xi = seeded((B, 2, T, 161), 41).to(DEV), (seeded((B, 2, T, 161), 42) * 0.3).to(DEV)
    outs, tcm = {}, {}
    for split in (False, True):
        net = nets.EpsNetPlan(nets.Ctx(DEV), sd, B, T, time_cond=True, nsteps=1, split_bf16=split)
        net.build_time()
        net.build_step(0)
        net.finish()
        net.x.copy_(x)
        net.x_init.copy_(xi)
        net.tsteps.fill_(7.25)
        net.plan.run()
        torch.cuda.synchronize()
        outs[split] = net.out.clone()
        assert len(tcm2_blocks(net.descs)) == (19 if split else 0)
        last = (tcm2_blocks(net.descs) or [d for d, _ in net.descs if isinstance(d, L.TcmDesc)])[-1]
        tcm[split] = (net.tcm_a if last.x_out == net.tcm_a.data_ptr() else net.tcm_b).clone()
        if split:
            for hs in net.tcm_hs:                     # margins untouched (join asserts it), planes finite
                vm, vk = P.tcm2_join_h(hs.cpu().numpy().view(np.uint16), B, T)
                assert np.isfinite(vm).all() and np.isfinite(vk).all()
            split_net = net
    assert rel_l2(tcm[True].cpu(), tcm[False].cpu()) < 5e-6
    assert rel_l2(outs[True].cpu(), outs[False].cpu()) < 2e-5
    # descriptor validation
    d = [d for d in tcm2_blocks(split_net.descs) if d.mode == 0][0]
    bad = type(d).from_buffer_copy(d)
    bad.dil = 33
    with pytest.raises(L.PdseError, match="dilation"):
        L.launch(bad)
    bad = type(d).from_buffer_copy(d)
    bad.hs_out = bad.hs
    with pytest.raises(L.PdseError, match="alias"):
        L.launch(bad)
    bad = type(d).from_buffer_copy(d)
    bad.mode = 2
    with pytest.raises(L.PdseError, match="mode"):
        L.launch(bad)


# ------------------------------------------------------------------ BIGLU blocks on plane tensors (csrc/bglu.hip), round 3
@pytest.mark.parametrize("plane_h", [False, True])
def test_plane_blocks_vs_goldens_and_gconv3(L, weights, plane_h, monkeypatch):
    """The eps-net with none / the encoder / every BiConv(Trans)GLU stage on plane tensors (conv1 outputs exchanged as exact
    bf16 split planes, software-pipelined 4-wave workgroups) against the reference's golden vectors - same tolerances as
    every other arithmetic of the block - at small T with the TCM input as an intermediate, and at T = 401."""
    nets, ops = pkg("nets"), pkg("ops")
    monkeypatch.setattr(nets.EpsNetPlan, "plane_h", plane_h)
    g = golden("diffunet1_small")
    B, T = int(g["B"]), int(g["T"])
    x = seeded((B, 2, T, 161), g["seed_x"])
    xi = seeded((B, 2, T, 161), g["seed_init"]) * float(g["init_scale"])
    op = ops.DiffUNet1Op(weights("DiffUNet1"), DEV)
    out = op(x.to(DEV), xi.to(DEV), torch.from_numpy(g["t"]).to(DEV))
    net = op._plans[(B, T)]
    torch.cuda.synchronize()
    assert sum(1 for d, _ in net.descs if isinstance(d, L.BgluDesc)) == {False: 0, True: 15}[plane_h]
    assert rel_l2(net.en[4].cpu().permute(0, 1, 3, 2), g["en5"]) < 2e-5
    e = rel_l2(out.cpu(), g["out"])
    print("plane_h %s: eps-net vs golden %.2e" % (plane_h, e))
    assert e < 2e-5
    if plane_h:
        Pk = pkg("packing")
        tensors = [(hp, True) for hp in net.hp_en.values()] + ([(hp, False) for hp in net.hp_de.values()] if plane_h is True else [])
        for hp, enc in tensors:                                           # margins untouched by every launch
            raw = hp.cpu().numpy().view(np.uint16)[:B]
            if enc and net.parity_planes:                                 # rows stored split by parity: back to bin order
                raw = raw[:, :, :, :, Pk.hp_par_pos(raw.shape[4]), :]
            full = Pk.hp_join(raw, with_margins=True)
            assert not full[:, :, :, :2].any() and not full[:, :, :, -2:].any() and (enc or not full[:, :, 0].any())
    g4 = golden("diffunet1_t401")
    x4 = seeded((1, 2, 401, 161), g4["seed_x"])
    xi4 = seeded((1, 2, 401, 161), g4["seed_init"]) * 0.3
    out4 = op(x4.to(DEV), xi4.to(DEV), torch.tensor([float(g4["t"])], device=DEV)).cpu()
    assert rel_l2(out4[0, :, ::16, :], g4["rows"]) < 2e-5
    assert abs(float(out4.double().pow(2).sum()) - float(g4["sumsq"])) < 4e-5 * float(g4["sumsq"])


def test_bf16_mode_tolerance(L, weights):
    """The opt-in bf16 mode (BASELINE configs 2/4/5 name bf16; SamplerPipeline(dtype="bf16"), bench.py --bf16): plain bf16
    operands in the eps-net's BiConv(Trans)GLU and TCM blocks and bf16 conv1 / bottleneck tensors between them.  bf16 keeps 8 significand bits
    (2^-9 = 2e-3 per rounding); over the 15 blocks of a forward and 6 reverse steps the stated tolerance is 3e-2 rel-L2 on
    the enhanced spectrogram against the reference's own fp32 loop at B=32, T=401 - 300x the fp32 tolerance, which is why
    this mode is never the default (measured: 1.6e-2).  The eps-net alone: 3e-2 against the golden forward (measured 1.8e-2)."""
    nets, ops = pkg("nets"), pkg("ops")
    g = golden("diffunet1_small")
    B, T = int(g["B"]), int(g["T"])
    x = seeded((B, 2, T, 161), g["seed_x"])
    xi = seeded((B, 2, T, 161), g["seed_init"]) * float(g["init_scale"])
    net = nets.EpsNetPlan(nets.Ctx(DEV), weights("DiffUNet1"), B, T, time_cond=True, nsteps=1, planes=1)
    net.build_time()
    net.build_step(0)
    net.finish()
    net.x.copy_(x)
    net.x_init.copy_(xi)
    net.tsteps.copy_(torch.from_numpy(g["t"]).view(1, B))
    net.plan.run()
    torch.cuda.synchronize()
    assert all(d.np == 1 for d, _ in net.descs if isinstance(d, L.BgluDesc)) and all(d.np == 1 for d in tcm2_blocks(net.descs)) and sum(1 for d, _ in net.descs if isinstance(d, L.BgluDesc)) == 15
    e_net = rel_l2(net.out.cpu(), g["out"])
    B, T = 32, 401
    feat, x_T = pkg("synth").synthetic_spectrogram(B, T, seed=1234)
    pipe = pkg("pipeline").SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights("DiffUNet1"), B, T=T, dtype="bf16")
    spec, init = pipe.sample(feat.to(DEV), x_T.to(DEV))
    ref, exact, ref_init = full_pair("full_gcrn_seed1234_t401_6step")
    e_init, e_spec = rel_l2(init[0].cpu(), ref_init[0]), rel_l2(spec[0].cpu(), ref[0])
    print("bf16 mode: eps-net forward vs golden %.2e | 6-step spectrogram vs the reference %.2e | prior (one-plane GEMM convolutions) %.2e" % (e_net, e_spec, e_init))
    assert 2e-5 < e_init < 1e-2            # round 4: the prior's GEMM-shaped convolutions are in the mode as well (korder 4): its own bound
    assert e_net < 3e-2 and e_spec < 3e-2
    assert e_spec > 1e-4                   # and it is NOT fp32-equivalent: the mode must stay opt-in
    with pytest.raises(ValueError):
        pkg("pipeline").SamplerPipeline(DEV, "GCRN", weights("GCRN"), weights("Nocon"), 1, T=16, dtype="bf16", deltamu=True)
