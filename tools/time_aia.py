"""Diagnostic: per-operator hipEvent timings of the DB-AIAT prior at B=32, T=401 (DUAL=1: the dual-branch model)."""
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")

B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 401))
if os.environ.get("DUAL"):
    net = nets.DualAiaPlan(nets.Ctx("cuda:0"), synth.make_state_dict("dual_aia_trans_merge_crm"), B, T)
else:
    net = nets.AiaPlan(nets.Ctx("cuda:0"), synth.make_state_dict("aia_complex_trans_ri"), B, T)
net.build()
net.finish()
net.x.copy_(torch.randn(B, 2, T, 161))
n = len(net.descs)
runs = [net.plan.time_ops(0, n) for _ in range(4)][1:]
med = [statistics.median(r[i] for r in runs) * 1e3 for i in range(n)]
tot = {}
for i, (d, tag) in enumerate(net.descs):
    k = type(d).__name__
    if k == "GconvDesc":
        k += ":nt%d:cin%d:cout%d" % (d.ntaps, d.in0.C, d.Cout)
    if k == "AttnDesc" or k == "GruDesc":
        k += ":axis%d:T%d:F%d" % (d.axis, d.T, d.F)
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += med[i]
print("total %.1f ms" % (sum(med) / 1e3))
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-40s n=%3d  %9.1f us total  %8.1f us each" % (k, c, t, t / c))
