"""Timing of the grouped-LSTM operators alone (B=32, T=401): per-frame launches (pdse_lstm_f32 x 2 + LayerNorm +
projections) against the layer wavefront (pdse_glstm_f32).  python tools/time_glstm.py"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
L = importlib.import_module("prior-diffuse_amd._lib")


def run(fused, B=32, T=401, reps=5, persist=False):
    nets.GcrnPlan.fused_glstm = fused
    nets.GcrnPlan.persist_lstm = persist
    p = nets.GcrnPlan(nets.Ctx("cuda:0"), synth.make_state_dict("GCRN"), B, T, exclusive=True)
    p.build()
    p.finish()
    p.x.normal_()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for _ in range(reps):
        ms = p.plan.time_ops(0, len(p.descs), st)
        for (d, tag), m in zip(p.descs, ms):
            key = type(d).__name__ + (":lstm" if tag == nets.TAG_LSTM else "")
            res.setdefault(key, []).append(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.plan.run(st)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / reps * 1e3
    per = {}
    for (d, tag), i in zip(p.descs, range(len(p.descs))):
        pass
    ms = p.plan.time_ops(0, len(p.descs), st)
    lstm = sum(m for (d, tag), m in zip(p.descs, ms) if tag == nets.TAG_LSTM)
    proj = sum(m for (d, tag), m in zip(p.descs, ms) if isinstance(d, L.GconvDesc) and d.Cout == 2048)
    ln = sum(m for (d, tag), m in zip(p.descs, ms) if isinstance(d, L.LnDesc))
    print("fused=%s persistent=%s  B=%d T=%d: prior %.2f ms | recurrent ops %.3f ms (%.2f us/step) | input projections %.3f ms | LayerNorm %.3f ms"
          % (fused, bool(getattr(p, "persist", False)), B, T, tot, lstm, lstm * 1e3 / ((T + 1) if getattr(p, "persist", False) else (T + 2) if fused else 2 * T), proj, ln), flush=True)


if __name__ == "__main__":
    if "--persist" in sys.argv:      # the small-batch persistent form (csrc/lstmp.hip) beside the wavefront
        for B, T in ((1, 401), (2, 401), (4, 401), (8, 401), (1, 1001)):
            run(True, B, T)
            run(True, B, T, persist=True)
        sys.exit(0)
    for B, T in ((32, 401), (1, 401), (16, 1001)):
        run(False, B, T)
        run(True, B, T)
