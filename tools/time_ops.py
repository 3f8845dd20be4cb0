"""Diagnostic: per-operator hipEvent timings of one eps-net forward at the bench shape
(B=32, T=401).  PDSE_LIB=<path to a diagnostic libpdse build> selects an ablation build
(csrc/gconv2.hip, PDSE_ABLATE).  Not part of the product or the tests."""
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
net = nets.EpsNetPlan(nets.Ctx("cuda:0"), synth.make_state_dict("DiffUNet1"), B, T, time_cond=True, nsteps=1)
net.build_time()
net.build_step(0)
net.finish()
g = torch.Generator().manual_seed(0)
net.x.copy_(torch.randn(B, 2, T, 161, generator=g))
net.x_init.copy_(torch.randn(B, 2, T, 161, generator=g) * 0.3)
net.tsteps.fill_(10.45)
n = len(net.descs)
runs = [net.plan.time_ops(0, n) for _ in range(6)][1:]
med = [statistics.median(r[i] for r in runs) * 1e3 for i in range(n)]
names = {nets.TAG_EPS_BLOCK: "block", nets.TAG_EPS_CONV1: "conv1", nets.TAG_TCM: "tcm", nets.TAG_EW: "ew"}
tot = {}
for i, (d, tag) in enumerate(net.descs):
    tot[names.get(tag, "?")] = tot.get(names.get(tag, "?"), 0) + med[i]
sel = [i for i, (d, tag) in enumerate(net.descs) if tag == nets.TAG_EPS_BLOCK][:5] + \
      [i for i, (d, tag) in enumerate(net.descs) if tag == nets.TAG_EPS_BLOCK][-2:] + \
      [i for i, (d, tag) in enumerate(net.descs) if tag == nets.TAG_TCM][:3] + \
      [i for i, (d, tag) in enumerate(net.descs) if tag == nets.TAG_EPS_CONV1][:3]
print(os.environ.get("PDSE_LIB", "default"), " total %.0f us" % sum(med), {k: round(v) for k, v in tot.items()})
print("   ", " ".join("%s%d:nt%d=%.0f" % (names.get(net.descs[i][1], "?")[0], i, getattr(net.descs[i][0], "ntaps", 0), med[i]) for i in sel))

# hot-cache repeat of single launches (same weights every time): separates cold-operand latency from the kernel itself
if os.environ.get("HOT"):
    st = torch.cuda.current_stream()
    for i in [i for i, (d, tag) in enumerate(net.descs) if tag == nets.TAG_TCM][:4]:
        d = net.descs[i][0]
        for _ in range(5):
            L.launch(d, st.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            L.launch(d, st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        print("hot  op %d (%s): %.1f us per launch" % (i, type(d).__name__, e0.elapsed_time(e1) * 5))

# per-launch efficiency table of the step: algorithmic GFLOP (from the descriptor), time, TFLOP/s
if os.environ.get("TABLE"):
    def gflop(d):
        if isinstance(d, L.TcmDesc):
            return 2.0 * d.B * d.T * (2 * 320 * 64 + 64 * 256 + (256 * 64 if d.h_out else 0)) / 1e9
        if not isinstance(d, L.GconvDesc):
            return 0.0
        pos0 = d.B * d.Tout * d.Fout
        accs = 1 if d.epi == L.EPI_LINEAR else 2
        cin = d.in0.C + d.in1.C
        fl = 2.0 * pos0 * accs * (d.ntaps * cin) * d.Cout
        if d.epi == L.EPI_BIGLU:
            tail = 2.0 * (2 * 32 * 32 + 32 * d.C2)
            fl += pos0 * tail
            if d.w2:
                pos1 = d.B * d.Tout * d.Fout1
                fl += 2.0 * pos1 * 2 * (bin(d.p1mask).count("1") * cin) * 32 + pos1 * tail
            fl += 2.0 * 64 * 32 * d.nx_n * (pos0 + (d.B * d.Tout * d.Fout1 if d.w2 else 0))
        return fl / 1e9
    print("%4s %-6s %-4s %5s %5s %5s %6s %9s %8s %7s" % ("op", "tag", "epi", "taps", "cin", "cout", "Fout", "GFLOP", "us", "TF/s"))
    for i, (d, tag) in enumerate(net.descs):
        g = gflop(d)
        if med[i] < 15:
            continue
        if isinstance(d, L.GconvDesc):
            print("%4d %-6s %-4d %5d %5d %5d %6d %9.2f %8.1f %7.1f" % (i, names.get(tag, "?"), d.epi, d.ntaps, d.in0.C + d.in1.C, d.Cout,
                                                                  d.Fout, g, med[i], g / med[i] * 1e3))
        else:
            print("%4d %-6s %-4s %5s %5s %5s %6s %9.2f %8.1f %7.1f" % (i, names.get(tag, "?"), type(d).__name__[:4], "", "", "", "", g, med[i],
                                                                  g / med[i] * 1e3))
