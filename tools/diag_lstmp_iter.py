"""Diagnostic: per-run device time (hipEvents) and wall time of the GCRN prior plan, persistent LSTM against the wavefront, 300 runs
each - looks for sporadic long runs."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
sd = synth.make_state_dict("GCRN")
for B, excl in ((1, True), (1, False), (4, True)):
    p = nets.GcrnPlan(nets.Ctx("cuda:0"), sd, B, 401, exclusive=excl)
    p.build(); p.finish(); p.x.normal_()
    st = torch.cuda.current_stream().cuda_stream
    dev, wall = [], []
    for i in range(300):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.record(); p.plan.run(st); e1.record()
        torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3); dev.append(e0.elapsed_time(e1))
    srt = sorted(dev)
    print("B %d persistent %s: device ms median %.2f p99 %.2f max %.2f | wall median %.2f max %.2f | runs > 2x median: device %d wall %d" % (
        B, p.persist, srt[150], srt[297], srt[-1], sorted(wall)[150], max(wall), sum(d > 2 * srt[150] for d in dev), sum(w > 2 * sorted(wall)[150] for w in wall)), flush=True)
