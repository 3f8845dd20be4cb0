"""Per-launch timing of one eps-net forward (B=32, T=401) in fp32 and split-bf16 mode: python tools/time_eps.py"""
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
L = importlib.import_module("prior-diffuse_amd._lib")


def run(split, B=32, T=401, plane_h=None, planes=None):
    net = nets.EpsNetPlan(nets.Ctx("cuda:0"), synth.make_state_dict("DiffUNet1"), B, T, time_cond=True, nsteps=1, split_bf16=split,
                          plane_h=plane_h, planes=planes, exclusive=True)
    net.build_time()
    net.build_step(0)
    net.finish()
    net.x.normal_()
    net.x_init.normal_()
    net.tsteps.fill_(10.45)
    st = torch.cuda.current_stream().cuda_stream
    best = None
    for _ in range(4):
        ms = net.plan.time_ops(0, len(net.descs), st)
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    tot = 0.0
    print("---- split_bf16 =", split, "plane_h =", net.plane_h, "planes =", net.planes)
    for (d, tag), m in zip(net.descs, best):
        tot += m
        if isinstance(d, L.GconvDesc) and d.epi == L.EPI_BIGLU:
            print("  BIGLU korder %d taps %2d dual %d nx %d C2 %2d  %3d x %3d positions: %7.1f us" % (
                d.korder, d.ntaps, 1 if d.w2 else 0, d.nx_n, d.C2, d.Tout, d.Fout, m * 1e3))
        if isinstance(d, L.BgluDesc):
            print("  BGLU planes %d taps %2d dual %d nx %d C2 %2d  %3d x %3d positions: %7.1f us" % (
                d.np, d.ntaps, 1 if d.p1mask else 0, d.nx_n, d.C2, d.Tout, d.Fout, m * 1e3))
        if isinstance(d, L.Tcm2Desc):
            print("  TCM split mode %d dil %2d: %6.1f us" % (d.mode, d.dil, m * 1e3))
        if isinstance(d, L.Tcm2sDesc):
            print("  TCM stack, %d residual blocks in one launch: %6.1f us (%.1f us per block)" % (d.n, m * 1e3, m * 1e3 / d.n))
    tcm = sum(m for (d, tag), m in zip(net.descs, best) if tag == nets.TAG_TCM)
    print("  TCM %.1f us; whole forward %.3f ms" % (tcm * 1e3, tot))


if __name__ == "__main__":
    if os.environ.get("BSWEEP"):          # batch-size sweep of the default form (does a smaller working set run faster per item?)
        for B_ in (32, 16, 8, 4):
            run(True, B=B_)
        sys.exit(0)
    if os.environ.get("PLANES"):          # PLANES=3,2: the block kernels' plane forms side by side (bf16x3 / f16x2 / bf16)
        for p_ in os.environ["PLANES"].split(","):
            run(True, planes=int(p_))
        sys.exit(0)
    run(False)
    run(True, plane_h=False)
    run(True)
    nets.EpsNetPlan.tcm_stack = False      # one launch per residual block (round 3) beside the stack launch (round 4)
    run(True, plane_h=True)
    nets.EpsNetPlan.tcm_stack = True
    run(True, planes=1)
