"""CPU: host logic added in round 2 — the shared weight bank (plans for new utterance lengths re-record
descriptors only), the sampling branches the reference's text defines (third conditioning branch, --sigma on deltamu)
replayed on the descriptor emulator against the reference-generated fixtures, bench.py's self-launch path, wav I/O."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import emu
from conftest import ROOT, golden, pkg, rel_l2, seeded
from oracle import restate as R


def test_weight_bank_is_shared_across_geometries(weights):
    """A second plan for another (B, T) uploads nothing: its descriptors point at the weights the first plan packed,
    and replaying them (emulator) still reproduces the oracle."""
    nets, P = pkg("nets"), pkg("pipeline")
    params = pkg("params").params
    bank = nets.WeightBank()
    a = P.SamplerPipeline("cpu", "GCRN", weights("GCRN"), weights("DiffUNet1"), 1, L_=1600, bank=bank)
    n_items, n_bytes, n_tensors = len(bank.items), bank.nbytes(), len(bank.keep)
    assert n_bytes > 40e6                                   # GCRN 39 MB + eps-net 11 MB of packed weights
    b = P.SamplerPipeline("cpu", "GCRN", weights("GCRN"), weights("DiffUNet1"), 2, L_=1130, bank=bank)
    assert (len(bank.items), bank.nbytes(), len(bank.keep)) == (n_items, n_bytes, n_tensors)
    wa = {d.w0 for d, _ in a.descs if isinstance(d, pkg("_lib").GconvDesc)}
    wb = {d.w0 for d, _ in b.descs if isinstance(d, pkg("_lib").GconvDesc)}
    assert wa == wb and len(a.descs) == len(b.descs)
    per_plan = sum(t.numel() * t.element_size() for t in b.ctx.keep)
    assert per_plan < 0.5 * n_bytes                         # what a new length costs: activations only
    wav, x_T = pkg("synth").synthetic_waveforms(2, 1130, seed=8)
    b.stft.wav.copy_(wav)
    b.xT_in.copy_(x_T)
    emu.run(b.descs, b.ctx.all_tensors())
    with torch.no_grad():
        ref_wav, ref_spec = R.enhance("GCRN", weights("GCRN"), weights("DiffUNet1"), wav, x_T, params.noise_schedule,
                                      params.inference_noise_schedule, True, False)
    assert rel_l2(b.spec, ref_spec) < 2e-5 and rel_l2(b.istft.wav, ref_wav) < 2e-5


def test_weight_bank_detects_a_diverging_builder(weights):
    nets = pkg("nets")
    bank = nets.WeightBank()
    p = nets.GcrnPlan(nets.Ctx("cpu", bank), weights("GCRN"), 1, 6)
    p.build()
    q = nets.GcrnPlan(nets.Ctx("cpu", bank), weights("GCRN"), 1, 6)
    q._mi = 3                                               # pretend the builder skipped three weight sites
    with pytest.raises(RuntimeError, match="diverged"):
        q.build()


@pytest.mark.parametrize("tag,ddpm,kw", [
    ("gcrn_fast_featcond", "DiffUNet1", dict(cond="feat")),
    ("gcrn_fast_featcond_sigma", "DiffUNet1", dict(cond="feat", use_sigma=True)),
    ("gcrn_fast_deltamu_sigma", "Nocon", dict(deltamu=True, use_sigma=True)),
])
def test_sampling_branches_plan_vs_reference_fixture(weights, tag, ddpm, kw):
    """Plans of the branches added this round (DiffUNet1 conditioned on the noisy feature, :972-974; --sigma with
    deltamu, :947-956) replayed on the emulator against fixtures made by the reference's own statements."""
    g = golden("sample_" + tag)
    feat, x_T = seeded((2, 2, 16, 161), g["seed_feat"]), seeded((2, 2, 16, 161), g["seed_xT"])
    pipe = pkg("pipeline").SamplerPipeline("cpu", "GCRN", weights("GCRN"), weights(ddpm), 2, T=16, **kw)
    pipe.feat.copy_(feat)
    pipe.xT_in.copy_(x_T)
    emu.run(pipe.descs, pipe.ctx.all_tensors())
    assert rel_l2(pipe.prior.out, g["init"]) < 5e-6
    assert rel_l2(pipe.spec, g["out"]) < 5e-5


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_flag_spawns_its_own_ranks():
    """``bench.py --gpus 2`` without a launcher starts two rank processes itself (gloo rehearsal: --dry-run) and
    rank 0 prints ONE JSON line with n_gpus = 2; the timing is the maximum over ranks."""
    r = _bench("--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["config"]["global_batch"] == 64 and out["steps"] == 4
    assert out["max_elapsed_s"] >= out["rank0_elapsed_s"] - 1e-3 and out["max_elapsed_s"] >= 4 * 0.002 * 0.9


def test_bench_gpus_flag_errors():
    r = _bench("--gpus", "2", "--steps", "1")               # no GPUs in the build container: clean refusal, no traceback
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "exposes" in r.stderr and "Traceback" not in r.stderr
    r = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_wav_io_and_resampling(tmp_path):
    import wave

    wavio = pkg("wavio")
    t = np.arange(48000) / 48000.0
    x = 0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t)
    path = str(tmp_path / "a48.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(48000)
        f.writeframes(np.round(x * 32767).astype("<i2").tobytes())
    y = wavio.read_wav(path, 16000)                         # librosa.load(sr=16000) resamples; so does the drop-in
    n = np.arange(16000) / 16000.0
    ref = 0.4 * np.sin(2 * np.pi * 440 * n) + 0.2 * np.sin(2 * np.pi * 3000 * n)
    assert y.shape == (16000,) and np.abs(y[300:-300] - ref[300:-300]).max() < 2e-4
    alias = wavio.resample(np.sin(2 * np.pi * 10000 * t), 48000, 16000)     # above the new Nyquist: must vanish
    assert np.abs(alias[300:-300]).max() < 1e-3
    assert wavio.resample(x, 44100, 16000).shape == (int(np.ceil(48000 * 160 / 441)),)
    out = str(tmp_path / "b.wav")
    wavio.write_wav(out, ref)
    back = wavio.read_wav(out)
    assert np.abs(back - ref).max() <= 1.0 / 32768 + 1e-7


def test_attrdict_surface():
    P = pkg("params")
    prm = P.AttrDict(dict(P.params))
    prm.deltamu, prm.pirorgrad = True, False
    assert prm["deltamu"] is True and P.params.deltamu is False and prm.fast_sampling is True
    with pytest.raises(AttributeError):
        prm.missing
    del prm.deltamu
    assert "deltamu" not in prm and not hasattr(P.AttrDict, "override")


def test_masked_loss_and_q_sample_restatements():
    """The oracle's restatements of the reference's tensor expressions (utils/loss.py:34-44; trainer :704-729)."""
    g = torch.Generator().manual_seed(2)
    esti, label = torch.randn(3, 2, 9, 5, generator=g), torch.randn(3, 2, 9, 5, generator=g)
    fl = [9, 4, 1]
    want = sum(((esti[i, :, :n] - label[i, :, :n]) ** 2).sum() for i, n in enumerate(fl)) / (2 * 5 * sum(fl))
    assert abs(float(R.com_mse_loss(esti, label, fl)) - float(want)) < 1e-6
    params = pkg("params").params
    lab, ini, noi = (torch.randn(2, 2, 4, 7, generator=g) for _ in range(3))
    t = torch.tensor([3, 40])
    ab = torch.tensor(np.cumprod(1 - np.array(params.noise_schedule)).astype(np.float32))[t].view(2, 1, 1, 1)
    assert torch.equal(R.q_sample(lab, ini, t, noi, params.noise_schedule, "deltamu"),
                       ab ** 0.5 * lab + (1.0 - ab) ** 0.5 * (noi + ini))


def test_split_tcm_operands_round_trip():
    """csrc/tcm2.hip operands: weights as bf16 fragment planes, the bottleneck tensor as frames-innermost planes."""
    P = pkg("packing")
    rng = np.random.default_rng(5)
    km, kk = (rng.standard_normal((320, 64)).astype(np.float32) for _ in range(2))
    packed = P.pack_tcm2_branch(km, kk)
    assert packed.shape == (2, 2, 20, 3, 64, 8) and packed.dtype == np.uint16
    back = P.unpack_tcm2_branch(packed)
    assert np.array_equal(back[0], km) and np.array_equal(back[1], kk)       # the three planes sum to the fp32 weight
    k2 = rng.standard_normal((64, 256)).astype(np.float32)
    assert np.array_equal(P.unpack_tcm2_conv2(P.pack_tcm2_conv2(k2)), k2)
    a, b = (rng.standard_normal((3, 64, 9)).astype(np.float32) for _ in range(2))
    hs = P.tcm2_split_h(a, b)
    assert hs.shape == P.tcm2_hs_shape(3, 9) == (3, 2, 4, 2, 3, 9 + 128, 8)
    x, y = P.tcm2_join_h(hs, 3, 9)
    assert np.array_equal(x, a) and np.array_equal(y, b)
    hs[0, 0, 0, 0, 0, 3, 0] = 1                                               # a write into the margin is caught
    with pytest.raises(AssertionError):
        P.tcm2_join_h(hs, 3, 9)


def test_f16x2_split_and_weight_exponent_edge_cases():
    """packing.split_f16x2 / f16_wexp (the host side of the f16x2 operand form, include/pdse.h PDSE_F16_ACT_EXP): zeros, values whose
    lo part is subnormal or zero, the largest finite fp16, an overflow (infinities, never a clipped value), all-zero and tiny weight
    groups, and the exponent that puts the largest weight in [2^13, 2^14)."""
    P = pkg("packing")
    x = np.array([0.0, -0.0, 1.0, 1.0 + 2.0 ** -11, 3.0e-5, 6.0e-8, 65504.0, -65504.0, 65519.9, 70000.0, -1.0e9], np.float32)
    hi, lo = P.split_f16x2(x)
    h, l = P.f16_to_f32(hi), P.f16_to_f32(lo)
    fin = np.abs(x) < 65520.0                                       # RN16 keeps values below 65520 finite
    assert np.array_equal(np.isfinite(h), fin) and np.all(np.isinf(h[~fin]) & np.isinf(l[~fin]) & (np.sign(h[~fin]) == -np.sign(l[~fin])))
    assert np.array_equal((h + l)[:3], x[:3]) and (h + l)[3] == x[3]                 # 1 + 2^-11: hi = 1, lo = 2^-11 exactly
    assert np.all(np.abs((h + l)[fin] - x[fin]) <= np.maximum(2.0 ** -23 * np.abs(x[fin]), 2.0 ** -25))
    for w, q in ((np.zeros((4, 4)), 0), (np.full((2, 2), 0.3), 15), (np.array([[1e-12, -3e-12]]), 40), (np.array([[3.0e5]]), -5)):
        assert P.f16_wexp(w) == q, (w.ravel()[:2], P.f16_wexp(w))
    w = np.array([[0.3, -0.01, 2.0e-6]])
    q = P.f16_wexp(w)
    assert 2.0 ** 13 <= 0.3 * 2.0 ** q < 2.0 ** 14
    back = P.from_planes(P.to_planes(w.astype(np.float32), 2, q), q)
    err = np.abs(back - w.astype(np.float32)) / np.abs(w)
    assert err[0, :2].max() <= 2.0 ** -23 and err[0, 2] <= 2.0 ** -21      # down to 2^-15 of the largest: half an fp32 ulp; 2e-6 (2^-17 of it): 21 bits
