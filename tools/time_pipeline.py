"""Diagnostic: per-operator hipEvent timings of the whole default pipeline (B=32, 4 s) outside the eps-net blocks:
signal path, prior, elementwise.  python tools/time_pipeline.py"""
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
pipeline = importlib.import_module("prior-diffuse_amd.pipeline")
L = importlib.import_module("prior-diffuse_amd._lib")

B, Ls = 32, 64000
p = pipeline.SamplerPipeline("cuda:0", "GCRN", synth.make_state_dict("GCRN"), synth.make_state_dict("DiffUNet1"), B, L_=Ls)
n = len(p.descs)
st = torch.cuda.current_stream().cuda_stream
runs = [p.plan.time_ops(0, n, st) for _ in range(5)][1:]
med = [statistics.median(r[i] for r in runs) * 1e3 for i in range(n)]
names = {nets.TAG_SIGNAL: "signal", nets.TAG_EW: "ew", nets.TAG_NONE: "none"}
print("ops %d total %.2f ms" % (n, sum(med) / 1e3))
seen = set()
for i, (d, tag) in enumerate(p.descs):
    other = isinstance(d, L.GconvDesc) and d.korder in (0, 1) and tag not in names
    if other:
        key = (d.ntaps, d.in0.C + d.in1.C, d.Cout, d.Tout, d.Fout, d.korder, d.epi)
        if key in seen:
            continue
        seen.add(key)
    if tag in names or other or not isinstance(d, (L.GconvDesc, L.Tcm2Desc)):
        extra = ""
        if isinstance(d, L.GconvDesc):
            extra = "taps %d cin %d cout %d %dx%d korder %d" % (d.ntaps, d.in0.C + d.in1.C, d.Cout, d.Tout, d.Fout, d.korder)
        print("%4d %-7s %-14s %8.1f us  %s" % (i, names.get(tag, "tag%d" % tag), type(d).__name__, med[i], extra))
