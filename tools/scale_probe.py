"""Which arithmetic leaves the oracle at large input scales?  python tools/scale_probe.py  (debug helper)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
R = importlib.import_module("oracle.restate")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


sd = synth.make_state_dict("DiffUNet1")
B, T = 2, 40
g = torch.Generator().manual_seed(31)
x0, xi0 = torch.randn(B, 2, T, 161, generator=g), torch.randn(B, 2, T, 161, generator=g) * 0.3
t = torch.full((B,), 10.451817)
for sc in (1.0, 4.0, 16.0, 64.0):
    x, xi = x0 * sc, xi0 * sc
    with torch.no_grad():
        ref = R.diffunet1_forward(sd, x, xi, t)
        ref64 = R.diffunet1_forward({k: v.double() for k, v in sd.items()}, x.double(), xi.double(), t.double()) if hasattr(R, "diffunet1_forward") else ref
    outs = {}
    for tag, kw in (("f16x2", dict(planes=2)), ("bf16x3", dict(planes=3)), ("fp32", dict(split_bf16=False))):
        net = nets.EpsNetPlan(nets.Ctx("cuda:0"), sd, B, T, time_cond=True, nsteps=1, **kw)
        net.build_time()
        net.build_step(0)
        net.finish()
        net.x.copy_(x)
        net.x_init.copy_(xi)
        net.tsteps.copy_(t.view(1, B))
        net.plan.run()
        torch.cuda.synchronize()
        outs[tag] = net.out.cpu().clone()
    print("scale %5g | out rms %.3g | vs fp64 oracle: fp32-oracle %.2e f16x2 %.2e bf16x3 %.2e fp32-kernels %.2e | f16x2 vs bf16x3 %.2e" % (
        sc, float(ref.pow(2).mean().sqrt()), rel(ref, ref64), rel(outs["f16x2"], ref64), rel(outs["bf16x3"], ref64), rel(outs["fp32"], ref64),
        rel(outs["f16x2"], outs["bf16x3"])))
