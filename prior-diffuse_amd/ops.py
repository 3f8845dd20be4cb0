"""Operator-level boundary: callables with the reference's ``nn.Module.__call__``
signatures for the two networks on the path,

    self.model(feat) -> X_init                   trainer/complex_ddpm_trainer.py:941
    self.model_ddpm(audio, init, t) -> eps       trainer/complex_ddpm_trainer.py:968

on contiguous fp32 ``[B,2,T,161]`` tensors living on an MI355X.  Each (B, T) builds its
plan once (weights packed into HBM once per operator) and replays it afterwards.
"""
from collections import OrderedDict

import torch

from . import _lib as L
from . import nets


class _PlannedOp:
    MAX_PLANS = 3   # recorded (B, T) geometries kept (least recently used first out); packed weights are shared

    def __init__(self, state_dict, device="cuda:0", bank=None, exclusive=False):
        """exclusive: the caller runs nothing else on the GPU beside this operator (one batch in flight), which allows
        launches whose workgroups wait for each other - the persistent LSTM of the GCRN prior at B <= 4 (csrc/lstmp.hip).
        Off by default: with it a small batch takes another kernel than a large one, so an utterance's result is no longer
        bit-identical across batch compositions (it stays within 1e-5)."""
        self.exclusive = bool(exclusive)
        self.sd = state_dict
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PdseError("operators run on the GPU only (no CPU fallback); got device %s" % device)
        L.load()
        self.bank = bank if bank is not None else nets.WeightBank()
        self._plans = OrderedDict()

    def _plan(self, key, build):
        """LRU of recorded plans; ``build(ctx)`` records a new one against the shared weight bank."""
        net = self._plans.get(key)
        if net is None:
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.popitem(last=False)
            net = self._plans[key] = build(nets.Ctx(self.device, self.bank))
            net.finish()
        else:
            self._plans.move_to_end(key)
        return net

    def _check(self, x):
        if x.dim() != 4 or x.shape[1] != 2 or x.shape[3] != nets.F0:
            raise ValueError("expected [B,2,T,161], got %s" % (tuple(x.shape),))
        if x.dtype != torch.float32:
            raise TypeError("fp32 only (the reference is fp32-only)")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def eval(self):  # parity contract: eval-mode statistics always
        return self

    @staticmethod
    def _steps(t):
        """Diffusion steps as float32 without a device->host round trip: a tensor that already lives on the GPU is
        range-checked by the kernel's clamp-free table walk only when ``PDSE_CHECK_STEPS`` asks for it (the check
        costs two synchronisations per call, i.e. per reverse step when the operator is driven from a Python loop);
        host tensors and python numbers are checked for free."""
        import os

        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        tf = t.to(torch.float32)
        if (not tf.is_cuda) or os.environ.get("PDSE_CHECK_STEPS", "0") == "1":
            if tf.numel() and (float(tf.min()) < 0 or float(tf.max()) > 49):
                raise IndexError("diffusion step outside the 50-entry embedding table")  # reference: IndexError
        return tf


class DiffUNet1Op(_PlannedOp):
    """ε-network ``DiffUNet1.forward(x, x_init, t)`` (model/diff3.py:37-57).
    ``t`` float32 -> lerp of the step embedding, int32/int64 -> table lookup."""

    def __call__(self, x, x_init, t):
        self._check(x)
        self._check(x_init)
        B, _, T, _ = x.shape
        if t.shape != (B,):
            raise ValueError("t must have shape [B]")
        tf = self._steps(t)

        def build(ctx):
            net = nets.EpsNetPlan(ctx, self.sd, B, T, time_cond=True, nsteps=1, exclusive=self.exclusive)
            net.build_time()
            net.build_step(0)
            return net

        net = self._plan((B, T), build)
        net.x.copy_(x)
        net.x_init.copy_(x_init)
        net.tsteps.copy_(tf.view(1, B))
        net.plan.run(self._stream())
        return net.out.clone()


class NoconOp(_PlannedOp):
    """Alt eps-net of the ``deltamu`` parameterisation: ``Nocon.forward(x, t)`` (model/piror_grad.py:28-40)."""

    def __call__(self, x, t):
        self._check(x)
        B, _, T, _ = x.shape
        if t.shape != (B,):
            raise ValueError("t must have shape [B]")
        tf = self._steps(t)

        def build(ctx):
            net = nets.EpsNetPlan(ctx, self.sd, B, T, time_cond=True, nsteps=1, with_pre=False, exclusive=self.exclusive)
            net.build_time()
            net.build_step(0)
            return net

        net = self._plan((B, T), build)
        net.x.copy_(x)
        net.tsteps.copy_(tf.view(1, B))
        net.plan.run(self._stream())
        return net.out.clone()


class DiffUNetOp(_PlannedOp):
    """Prior ``DiffUNet.forward(x)`` (model/diff.py:23-33)."""

    def __call__(self, x):
        self._check(x)
        B, _, T, _ = x.shape

        def build(ctx):
            net = nets.EpsNetPlan(ctx, self.sd, B, T, time_cond=False, exclusive=self.exclusive)
            net.build_step(0)
            return net

        net = self._plan((B, T), build)
        net.x.copy_(x)
        net.plan.run(self._stream())
        return net.out.clone()


class GCRNOp(_PlannedOp):
    """Prior ``GCRN.forward(x)`` (model/gcrn.py:136-166)."""

    def __call__(self, x):
        self._check(x)
        B, _, T, _ = x.shape

        def build(ctx):
            net = nets.GcrnPlan(ctx, self.sd, B, T, exclusive=self.exclusive)
            net.build()
            return net

        net = self._plan((B, T), build)
        net.x.copy_(x)
        net.plan.run(self._stream())
        return net.out.clone()


class AiaOp(_PlannedOp):
    """Prior ``aia_complex_trans_ri.forward(x)`` (model/dbaiat.py:461-478, conf/dbaiat.yml:13)."""

    PLAN = "AiaPlan"

    def __call__(self, x):
        self._check(x)
        B, _, T, _ = x.shape

        def build(ctx):
            net = getattr(nets, self.PLAN)(ctx, self.sd, B, T)
            net.build()
            return net

        net = self._plan((B, T), build)
        net.x.copy_(x)
        net.plan.run(self._stream())
        return net.out.clone()


class DualAiaOp(AiaOp):
    """Prior ``dual_aia_trans_merge_crm.forward(x)`` (model/dbaiat.py:386-413), the dual-branch DB-AIAT model."""

    PLAN = "DualAiaPlan"


PRIOR_OPS = {"GCRN": GCRNOp, "DiffUNet": DiffUNetOp, "aia_complex_trans_ri": AiaOp, "dual_aia_trans_merge_crm": DualAiaOp}


def q_sample(label, init, t, noise, noise_schedule=None, mode="pirorgrad", sigma=False):
    """Forward noising of the training step (trainer/complex_ddpm_trainer.py:42-44, :704-729),
    t int64 [B], alpha_bar the float32 cumprod the trainer keeps in ``noise_level``:

        mode "pirorgrad": noisy = sqrt(ab_t) * (label - init) + sqrt(1 - ab_t) * noise          (:718)
        mode "deltamu":   noisy = sqrt(ab_t) * label + sqrt(1 - ab_t) * (noise + init)          (:721)
        mode "plain":     noisy = sqrt(ab_t) * label + sqrt(1 - ab_t) * noise                   (:724)
        sigma: noise is first multiplied by sqrt(|init| / max|init| / 2 + 0.5) per (b, channel) (:709-715)

    ``label`` / ``init`` are already divided by 11 by the caller, like the reference.  Bit-exact with the
    reference's fp32 tensor arithmetic."""
    import numpy as np

    from .params import params as _p

    if label.device.type != "cuda":
        raise L.PdseError("q_sample runs on the GPU only (no CPU fallback)")
    modes = {"pirorgrad": 0, "deltamu": 1, "plain": 2}
    if mode not in modes:
        raise ValueError("mode must be one of %s" % sorted(modes))
    for x in (label, init, noise):
        if not x.is_contiguous() or x.dtype != torch.float32 or x.shape != label.shape:
            raise ValueError("label, init, noise must be contiguous fp32 tensors of one shape")
    stream = torch.cuda.current_stream(label.device).cuda_stream
    if sigma:
        if label.dim() != 4:
            raise ValueError("sigma mask needs [B,C,T,F] tensors")
        masked = torch.empty_like(noise)
        sd_ = L.SigmaDesc()
        sd_.init, sd_.a, sd_.out = init.data_ptr(), noise.data_ptr(), masked.data_ptr()
        maxbuf = torch.empty(label.shape[0] * label.shape[1], device=label.device, dtype=torch.float32)
        sd_.maxbuf, sd_.plane, sd_.nplanes = maxbuf.data_ptr(), label.shape[2] * label.shape[3], label.shape[0] * label.shape[1]
        L.launch(sd_, stream, device=label.device)
        noise = masked
    beta = np.array(_p.noise_schedule if noise_schedule is None else noise_schedule)
    noise_level = torch.tensor(np.cumprod(1 - beta).astype(np.float32), device=label.device)     # :42-44
    ns = noise_level[t.to(label.device).long()]
    a, s = ns ** 0.5, (1.0 - ns) ** 0.5
    out = torch.empty_like(label)
    d = L.QsampleDesc()
    d.label, d.init, d.noise, d.out = label.data_ptr(), init.data_ptr(), noise.data_ptr(), out.data_ptr()
    d.a, d.s, d.plane, d.B, d.mode = a.data_ptr(), s.data_ptr(), label[0].numel(), label.shape[0], modes[mode]
    L.launch(d, stream, device=label.device)
    return out


def com_mse_loss(esti, label, frame_list):
    """Masked complex MSE of the validation loop (utils/loss.py:34-44; trainer/complex_ddpm_trainer.py:490):
    ``((esti - label) * mask) ** 2).sum() / mask.sum()`` over [B,2,T,F] with mask = 1 on the first ``frame_list[b]``
    frames of utterance b.  Returns a 0-dim fp32 tensor on the device (no host synchronisation)."""
    if esti.device.type != "cuda":
        raise L.PdseError("com_mse_loss runs on the GPU only (no CPU fallback)")
    if esti.dim() != 4 or esti.shape != label.shape or esti.dtype != torch.float32 or label.dtype != torch.float32:
        raise ValueError("esti, label: fp32 tensors of one [B,C,T,F] shape")
    if not (esti.is_contiguous() and label.is_contiguous()):
        raise ValueError("esti, label must be contiguous")
    B, C_, T, F_ = esti.shape
    frames = torch.as_tensor(list(frame_list), dtype=torch.int32)
    if frames.numel() != B or int(frames.min()) < 0 or int(frames.max()) > T:
        raise ValueError("frame_list: one frame count in [0, T] per utterance")
    frames = frames.to(esti.device)
    partial = torch.empty(B * L.MASKLOSS_BLOCKS, dtype=torch.float64, device=esti.device)
    out = torch.empty(1, dtype=torch.float32, device=esti.device)
    d = L.MasklossDesc()
    d.esti, d.label, d.frames, d.partial, d.out = (esti.data_ptr(), label.data_ptr(), frames.data_ptr(), partial.data_ptr(),
                                                    out.data_ptr())
    d.B, d.C, d.T, d.F = B, C_, T, F_
    L.launch(d, torch.cuda.current_stream(esti.device).cuda_stream, device=esti.device)
    return out[0]
