"""Diagnostic: per-launch hipEvent timings + algorithmic TFLOP/s of the GCRN prior at B=32, T=401."""
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
L = importlib.import_module("prior-diffuse_amd._lib")

B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 401))
net = nets.GcrnPlan(nets.Ctx("cuda:0"), synth.make_state_dict("GCRN"), B, T)
net.build()
net.finish()
net.x.copy_(torch.randn(B, 2, T, 161))
n = len(net.descs)
runs = [net.plan.time_ops(0, n) for _ in range(5)][1:]
med = [statistics.median(r[i] for r in runs) * 1e3 for i in range(n)]
print("total %.1f us over %d ops" % (sum(med), n))
print("%4s %-5s %4s %5s %5s %5s %6s %6s %9s %8s %7s" % ("op", "kind", "epi", "taps", "cin", "cout", "Tout", "Fout", "GFLOP", "us", "TF/s"))
for i, (d, tag) in enumerate(net.descs):
    if isinstance(d, L.GconvDesc):
        accs = 1 if d.epi == L.EPI_LINEAR else 2
        cin = d.in0.C + d.in1.C
        g = 2.0 * d.B * d.Tout * d.Fout * accs * d.ntaps * max(cin, 1) * d.Cout / 1e9
        print("%4d %-5s %4d %5d %5d %5d %6d %6d %9.2f %8.1f %7.1f" % (i, "gconv", d.epi, d.ntaps, cin, d.Cout, d.Tout, d.Fout, g, med[i],
                                                                   g / med[i] * 1e3))
    else:
        print("%4d %-5s %61.1f" % (i, type(d).__name__[:5], med[i]))
