"""Operator-level boundary: callables with the reference's ``nn.Module.__call__``
signatures for the two networks on the path,

    self.model(feat) -> X_init                   trainer/complex_ddpm_trainer.py:941
    self.model_ddpm(audio, init, t) -> eps       trainer/complex_ddpm_trainer.py:968

on contiguous fp32 ``[B,2,T,161]`` tensors living on an MI355X.  Each (B, T) builds its
plan once (weights packed into HBM once per operator) and replays it afterwards.
"""
import torch

from . import _lib as L
from . import nets


class _PlannedOp:
    def __init__(self, state_dict, device="cuda:0"):
        self.sd = state_dict
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PdseError("operators run on the GPU only (no CPU fallback); got device %s" % device)
        L.load()
        self._plans = {}

    def _check(self, x):
        if x.dim() != 4 or x.shape[1] != 2 or x.shape[3] != nets.F0:
            raise ValueError("expected [B,2,T,161], got %s" % (tuple(x.shape),))
        if x.dtype != torch.float32:
            raise TypeError("fp32 only (the reference is fp32-only)")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def eval(self):  # parity contract: eval-mode statistics always
        return self


class DiffUNet1Op(_PlannedOp):
    """ε-network ``DiffUNet1.forward(x, x_init, t)`` (model/diff3.py:37-57).
    ``t`` float32 -> lerp of the step embedding, int32/int64 -> table lookup."""

    def __call__(self, x, x_init, t):
        self._check(x)
        self._check(x_init)
        B, _, T, _ = x.shape
        if t.shape != (B,):
            raise ValueError("t must have shape [B]")
        tf = t.to(torch.float32)
        if float(tf.min()) < 0 or float(tf.max()) > 49:
            raise IndexError("diffusion step outside the 50-entry embedding table")  # reference: IndexError
        key = (B, T)
        if key not in self._plans:
            net = nets.EpsNetPlan(nets.Ctx(self.device), self.sd, B, T, time_cond=True, nsteps=1)
            net.build_time()
            net.build_step(0)
            net.finish()
            self._plans[key] = net
        net = self._plans[key]
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
        tf = t.to(torch.float32)
        if t.shape != (B,) or float(tf.min()) < 0 or float(tf.max()) > 49:
            raise IndexError("t must be [B] diffusion steps inside the 50-entry embedding table")
        key = (B, T)
        if key not in self._plans:
            net = nets.EpsNetPlan(nets.Ctx(self.device), self.sd, B, T, time_cond=True, nsteps=1, with_pre=False)
            net.build_time()
            net.build_step(0)
            net.finish()
            self._plans[key] = net
        net = self._plans[key]
        net.x.copy_(x)
        net.tsteps.copy_(tf.view(1, B))
        net.plan.run(self._stream())
        return net.out.clone()


class DiffUNetOp(_PlannedOp):
    """Prior ``DiffUNet.forward(x)`` (model/diff.py:23-33)."""

    def __call__(self, x):
        self._check(x)
        B, _, T, _ = x.shape
        key = (B, T)
        if key not in self._plans:
            net = nets.EpsNetPlan(nets.Ctx(self.device), self.sd, B, T, time_cond=False)
            net.build_step(0)
            net.finish()
            self._plans[key] = net
        net = self._plans[key]
        net.x.copy_(x)
        net.plan.run(self._stream())
        return net.out.clone()


class GCRNOp(_PlannedOp):
    """Prior ``GCRN.forward(x)`` (model/gcrn.py:136-166)."""

    def __call__(self, x):
        self._check(x)
        B, _, T, _ = x.shape
        key = (B, T)
        if key not in self._plans:
            net = nets.GcrnPlan(nets.Ctx(self.device), self.sd, B, T)
            net.build()
            net.finish()
            self._plans[key] = net
        net = self._plans[key]
        net.x.copy_(x)
        net.plan.run(self._stream())
        return net.out.clone()


class AiaOp(_PlannedOp):
    """Prior ``aia_complex_trans_ri.forward(x)`` (model/dbaiat.py:461-478, conf/dbaiat.yml:13)."""

    PLAN = "AiaPlan"

    def __call__(self, x):
        self._check(x)
        B, _, T, _ = x.shape
        key = (B, T)
        if key not in self._plans:
            net = getattr(nets, self.PLAN)(nets.Ctx(self.device), self.sd, B, T)
            net.build()
            net.finish()
            self._plans[key] = net
        net = self._plans[key]
        net.x.copy_(x)
        net.plan.run(self._stream())
        return net.out.clone()


class DualAiaOp(AiaOp):
    """Prior ``dual_aia_trans_merge_crm.forward(x)`` (model/dbaiat.py:386-413), the dual-branch DB-AIAT model."""

    PLAN = "DualAiaPlan"


PRIOR_OPS = {"GCRN": GCRNOp, "DiffUNet": DiffUNetOp, "aia_complex_trans_ri": AiaOp, "dual_aia_trans_merge_crm": DualAiaOp}


def q_sample(label, init, t, noise, noise_schedule=None):
    """Forward noising of the training step, prior-grad parameterisation (SURVEY §8f rank 4;
    trainer/complex_ddpm_trainer.py:42-44, :704-727):

        noisy = sqrt(alpha_bar_t) * (label - init) + sqrt(1 - alpha_bar_t) * noise,   t int64 [B]

    ``label`` / ``init`` are already divided by 11 by the caller, like the reference.  Bit-exact with the
    reference's fp32 tensor arithmetic (alpha_bar is the float32 cumprod the trainer keeps in ``noise_level``)."""
    import numpy as np

    from .params import params as _p

    if label.device.type != "cuda":
        raise L.PdseError("q_sample runs on the GPU only (no CPU fallback)")
    beta = np.array(_p.noise_schedule if noise_schedule is None else noise_schedule)
    noise_level = torch.tensor(np.cumprod(1 - beta).astype(np.float32), device=label.device)     # :42-44
    ns = noise_level[t.to(label.device).long()]
    a, s = ns ** 0.5, (1.0 - ns) ** 0.5
    out = torch.empty_like(label)
    d = L.QsampleDesc()
    d.label, d.init, d.noise, d.out = label.data_ptr(), init.data_ptr(), noise.data_ptr(), out.data_ptr()
    d.a, d.s, d.plane, d.B = a.data_ptr(), s.data_ptr(), label[0].numel(), label.shape[0]
    for x in (label, init, noise):
        if not x.is_contiguous() or x.dtype != torch.float32 or x.shape != label.shape:
            raise ValueError("label, init, noise must be contiguous fp32 tensors of one shape")
    L.launch(d, torch.cuda.current_stream(label.device).cuda_stream)
    return out
