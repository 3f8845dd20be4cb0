"""Compare every intermediate of an eps-net forward between two plane forms (debug helper): python tools/cmp_planes.py"""
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


def run(planes, B=2, T=12):
    net = nets.EpsNetPlan(nets.Ctx("cuda:0"), synth.make_state_dict("DiffUNet1"), B, T, time_cond=True, nsteps=1, planes=planes)
    net.build_time()
    net.build_step(0)
    net.finish()
    g = torch.Generator().manual_seed(5)
    net.x.copy_(torch.randn(B, 2, T, 161, generator=g))
    net.x_init.copy_(torch.randn(B, 2, T, 161, generator=g) * 0.3)
    net.tsteps.fill_(10.45)
    net.plan.run()
    torch.cuda.synchronize()
    out = {"out": net.out.cpu().numpy(), "en5": net.en[4].cpu().numpy(), "H5": net.H5.cpu().numpy()}
    for di in range(2):
        for k in range(1, 5):
            out["Pskip%d_%d" % (di, k)] = net.Pskip[di][k].cpu().numpy()[:B]
    for k, hp in net.hp_en.items():
        raw = hp.cpu().numpy().view(np.uint16)[:B]
        if net.parity_planes:
            raw = raw[:, :, :, :, Pk.hp_par_pos(raw.shape[4]), :]
        out["hp_en%d" % k] = Pk.hp_join(raw, with_margins=True)
    for k, hp in net.hp_de.items():
        out["hp_de%d" % k] = Pk.hp_join(hp.cpu().numpy().view(np.uint16)[:B], with_margins=True)
    return out


a, b = run(3), run(2)
for k in a:
    den = np.sqrt((a[k].astype(np.float64) ** 2).sum()) + 1e-30
    print("%-10s rel-L2 %.3e  (norm %.3e, finite %s)" % (k, np.sqrt(((a[k].astype(np.float64) - b[k]) ** 2).sum()) / den, den, np.isfinite(b[k]).all()))

for name in ("Pskip0_4", "Pskip1_1"):
    A, Bb = a[name].astype(np.float64), b[name].astype(np.float64)      # [B, 8 groups, T, F, 4]
    A = A.reshape(A.shape[0], 8, -1, A.shape[-1] // 4 if False else A.shape[3], 4) if A.ndim == 5 else A
    print(name, A.shape)
    # per channel (group g, lane c): mean of a, mean of b - a, rms of b - a, slope of b on a
    for g in range(8):
        for c in range(4):
            if A.ndim == 4:      # [B, 32, T, F] storage re-read as groups of 4: channel = 4 g + c at [:, g*4 + c]?
                pa, pb = A.reshape(A.shape[0], 8, -1, 4)[:, g, :, c].ravel(), Bb.reshape(Bb.shape[0], 8, -1, 4)[:, g, :, c].ravel()
            else:
                pa, pb = A[:, g, ..., c].ravel(), Bb[:, g, ..., c].ravel()
            sl = float((pa * pb).sum() / (pa * pa).sum())
            print("  ch %2d: mean a %+.3e  mean(b-a) %+.3e  rms(b-a) %.3e  rms a %.3e  slope %.4f" % (4 * g + c, pa.mean(), (pb - pa).mean(), np.sqrt(((pb - pa) ** 2).mean()), np.sqrt((pa ** 2).mean()), sl))
