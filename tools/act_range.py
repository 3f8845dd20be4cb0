"""Largest magnitudes of the plane tensors the f16x2 kernels exchange (hp of every stage, the TCM bottleneck hs) for the nominal
inputs of the bench: python tools/act_range.py  (how far the fp16 window of include/pdse.h: PDSE_F16_ACT_EXP is from them)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
Pk = importlib.import_module("prior-diffuse_amd.packing")

B, T = 2, 200
net = nets.EpsNetPlan(nets.Ctx("cuda:0"), synth.make_state_dict("DiffUNet1"), B, T, time_cond=True, nsteps=1, planes=2)
net.build_time()
net.build_step(0)
net.finish()
g = torch.Generator().manual_seed(5)
net.x.copy_(torch.randn(B, 2, T, 161, generator=g))
net.x_init.copy_(torch.randn(B, 2, T, 161, generator=g) * 0.3)
net.tsteps.fill_(10.45)
net.plan.run()
torch.cuda.synchronize()
print("window: full precision for %.4g <= |x|, saturation at %.5g" % (2.0 ** (-2 - Pk.F16_ACT_EXP), 65504.0 / 2 ** Pk.F16_ACT_EXP))
for name, hp in [("hp_en%d" % k, v) for k, v in net.hp_en.items()] + [("hp_de%d" % k, v) for k, v in net.hp_de.items()]:
    v = Pk.hp_join(hp.cpu().numpy().view(np.uint16)[:B], with_margins=True)
    a = np.abs(v[v != 0])
    print("%-8s max %.3g  rms %.3g  1%% quantile %.3g" % (name, a.max(), np.sqrt((a ** 2).mean()), np.quantile(a, 0.01)))
for i, hs in enumerate(net.tcm_hs):
    vm, vk = Pk.tcm2_join_h(hs.cpu().numpy().view(np.uint16), B, T)
    for nm, v in (("main", vm), ("mask", vk)):
        a = np.abs(v[v != 0])
        print("tcm hs%d %s max %.3g rms %.3g" % (i, nm, a.max(), np.sqrt((a ** 2).mean())))
print("eps-net output rms %.3g" % float(net.out.pow(2).mean().sqrt()))
