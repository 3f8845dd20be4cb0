"""bench.py — the reference's headline workload on N MI355X GPUs of one node.

Metric (BASELINE.json): enhanced-audio seconds per wall second (real-time factor) of
6-step fast sampling; workload = configs[1] restated at the shapes the reference can
actually run: B=32 utterances of 4 s at 16 kHz per GPU, spectrograms
[32,2,401,161], GCRN prior + DiffUNet1 x 6.  Arithmetic of the graded line ("dtype": "f16x2", round 4): every fp32 operand of
the contractions is scaled by a power of two into the fp16 window and carried as hi = RN16(x), lo = RN16(x - hi) - within
half an fp32 ulp of x - and the three leading cross products run on the f16 matrix cores with fp32 accumulation:
fp32-equivalent (what is dropped is <= 2^-22 |ab|), parity-tested against the same fixtures and tolerances as exact fp32
(tests/test_gpu_f16x2.py).  Measured beside it in the same run: "bf16x3" (the exact three-way bf16 split with six products,
the default until round 4; ``--split bf16x3`` makes it the main line) and "fp32_exact" (v_mfma_f32_32x32x2_f32 everywhere;
``--fp32``), each sequential AND in flight.  ``--bf16`` is the opt-in single-product bf16 mode with bf16 block-boundary
storage (BASELINE configs 2/4/5 name bf16; its own tolerance, never the graded line).

A "step" is one pass of the whole hot path over one batch: waveforms already resident in
HBM -> STFT -> prior -> 6 reverse steps -> ISTFT -> waveforms in HBM, replayed from one
hipGraph.  Multi-GPU: utterances are sharded over ranks (weak scaling, B=32 per rank), no
data-path collective; the only collectives are the timing barrier and a max-reduce.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]

``--gpus N`` with N > 1 and no launcher environment (WORLD_SIZE unset): this process only spawns N fresh rank
processes (it never touches a GPU itself) and waits for them; under ``torch.distributed.run`` the ranks already exist
and ``--gpus`` must equal WORLD_SIZE.  Either way: one process per GPU, RCCL (backend "nccl"), rendezvous on 127.0.0.1.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per utterance: flops of one eps-net forward at T=401
EPS_GFLOP_PER_UTT_STEP = 10.29
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md, fp32 matrix = fp32 vector peak
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md, dense bf16 MFMA
EPS_MB_PER_UTT_STEP = 139.3            # block-boundary bytes of one eps-net forward at T=401 (86,858 elements x T x 4 B)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md, HBM3E


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, dry_run):
    """Parent of a self-launched multi-GPU run: start n children (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
    environment), wait, return the worst exit code.  The parent makes no HIP call (device_count() does not initialise
    the runtime on this image), and the children are fresh interpreters, never an exec of a process that touched the GPU."""
    import subprocess

    if not dry_run:
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but this node exposes %d GPU(s)" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30,
                    help="timed passes (default 30: with three batches in flight the first and last passes of the timed region "
                         "overlap less than the rest - 10 passes read 17.6-17.9 ms, 24-30 read 17.0-17.2 ms with the same library)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--prior", default="GCRN", choices=["GCRN", "DiffUNet", "aia_complex_trans_ri", "dual_aia_trans_merge_crm"],
                    help="discriminative prior (BASELINE configs 1-3: GCRN; config 4: aia_complex_trans_ri)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight, each on its own HIP stream (1: strictly sequential, hipGraph replay). "
                         "The latency-bound LSTM/TCM chains of one batch run beside the MFMA-bound blocks of the others; "
                         "every batch's result is bit-identical to the sequential run")
    ap.add_argument("--stage-streams", action="store_true",
                    help="with --inflight 2: prior stream | loop stream instead of one stream per batch")
    ap.add_argument("--streams", type=int, default=1, help="concurrent sub-batch pipelines per GPU")
    ap.add_argument("--full-schedule", action="store_true",
                    help="BASELINE config 3: the full 50-step reverse schedule instead of 6-step fast sampling")
    ap.add_argument("--fp32", action="store_true",
                    help="exact fp32 MFMA arithmetic everywhere (v_mfma_f32_32x32x2_f32).  Default for fast sampling: the "
                         "eps-net's BIGLU blocks run on the bf16 matrix cores with exact three-way bf16 operand splits "
                         "(six products, fp32 accumulate) - fp32-level accuracy, parity-tested against the same goldens "
                         "and tolerances; the full 50-step schedule always runs exact fp32")
    ap.add_argument("--split", choices=["bf16x3", "f16x2"], default=None,
                    help="the fp32-equivalent operand split of the matrix-core kernels (SamplerPipeline.default_split when omitted): "
                         "bf16x3 = exact three-way bf16 split, six bf16 products; f16x2 = fp16 hi + lo of the power-of-two scaled "
                         "operand, three f16 products (same goldens and tolerances)")
    ap.add_argument("--bf16", action="store_true",
                    help="the opt-in bf16 mode (BASELINE configs 2/4/5): plain bf16 operands and bf16 conv1 tensors in the "
                         "eps-net's BiConv(Trans)GLU blocks (SamplerPipeline(dtype='bf16')); own tolerance (3e-2), dtype 'bf16' "
                         "in the line - never the graded default")
    ap.add_argument("--tcm-launches", action="store_true",
                    help="A/B: one launch per TCM residual block (round 3) instead of the persistent stack launch (csrc/tcm2.hip: tcm2s_kernel)")
    ap.add_argument("--inflight-stack", action="store_true",
                    help="measurement: the in-flight pipelines take the TCM stack as ONE launch per forward (default: one launch per block)")
    ap.add_argument("--no-file-loop", action="store_true",
                    help="skip the B = 1 generate_wav file loop reported as file_loop_b1 (profiling runs: keeps its B = 1 launches "
                         "out of the kernel trace, so that per-kernel averages of the trace are B = 32 launches only)")
    ap.add_argument("--no-fp32-compare", action="store_true",
                    help="skip the extra exact-fp32 pass reported as fp32_exact (profiling runs: keeps its kernels out of the trace)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launch / rendezvous / timing-reduction / reporting path on CPU (gloo, no GPU, no "
                         "kernels): the JSON line carries \"dry_run\": true and value null")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.dry_run))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE=%s of the launcher" % (args.gpus, os.environ["WORLD_SIZE"]))
    fast = not args.full_schedule
    if args.bf16 and (args.fp32 or args.full_schedule or args.streams > 1):
        ap.error("--bf16 is the 6-step fast-sampling mode; not with --fp32 / --full-schedule / --streams")
    args.split_bf16 = fast and not args.fp32
    args.overlap = args.inflight > 1
    args.depth, args.by_batch = max(2, args.inflight), not args.stage_streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PDSE_BENCH_BACKEND=gloo + PDSE_BENCH_DEVICE=0: rehearse the N>1 code path with several ranks on ONE GPU
        # (RCCL refuses two ranks per device); the driver's real runs use nccl, one rank per GPU
        backend = "gloo" if args.dry_run else os.environ.get("PDSE_BENCH_BACKEND", "nccl")
        if "PDSE_BENCH_DEVICE" in os.environ:
            local = int(os.environ["PDSE_BENCH_DEVICE"])
        if backend == "nccl":
            if local >= torch.cuda.device_count():
                sys.exit("bench.py: rank %d needs GPU %d, node exposes %d" % (rank, local, torch.cuda.device_count()))
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.dry_run:
        return dry_run(args, dist, world, rank)
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local

    def note(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    import __graft_entry__ as ge

    ge.build()
    synth = importlib.import_module("prior-diffuse_amd.synth")
    pipeline = importlib.import_module("prior-diffuse_amd.pipeline")
    nets = importlib.import_module("prior-diffuse_amd.nets")
    shard = importlib.import_module("prior-diffuse_amd.shard")

    if args.tcm_launches:
        nets.EpsNetPlan.tcm_stack = False
    B, L_ = args.batch, int(args.seconds * 16000)
    T = 1 + L_ // 160
    gs, ds = synth.make_state_dict(args.prior), synth.make_state_dict("DiffUNet1")
    # the global batch is generated once from the seed and sliced, so results do not depend on N
    lo, hi = shard.shard_range(B * world, world, rank)
    wav, x_T = synth.synthetic_waveforms(B * world, L_, seed=1234)
    wav, x_T = wav[lo:hi].to(dev), x_T[lo:hi].to(dev)

    use_graph = not args.no_graph
    dtype = "bf16" if args.bf16 else "f32"
    if args.inflight_stack:
        pipeline.PipelinedSampler.stack_in_flight = True
    if args.split is not None:
        pipeline.SamplerPipeline.default_split = args.split
    args.split = pipeline.SamplerPipeline.default_split
    f16 = args.split == "f16x2" and args.split_bf16 and not args.bf16      # three f16 products per fp32-equivalent multiply-add
    products = 3 if f16 else 6
    if args.overlap:
        runner = pipeline.PipelinedSampler(dev, args.prior, gs, ds, B, L_, depth=args.depth, by_batch=args.by_batch,
                                           graph=use_graph, fast_sampling=fast, split_bf16=args.split_bf16, dtype=dtype)
        pipe = runner.pipes[0]
    elif args.streams > 1:
        runner = pipeline.ConcurrentSampler(dev, args.prior, gs, ds, B, L_=L_, nsplit=args.streams, fast_sampling=fast, split_bf16=args.split_bf16)
        pipe = runner.pipes[0]
    else:
        runner = pipe = pipeline.SamplerPipeline(dev, args.prior, gs, ds, B, L_=L_, fast_sampling=fast, split_bf16=args.split_bf16, dtype=dtype)

    def step():
        if args.overlap:
            runner.submit(wav, x_T)          # every submitted batch completes inside the timed region (drain below)
        else:
            runner.enhance(wav, x_T, graph=use_graph)

    note("plan built: %d operators; warm-up (graph=%s)" % (len(pipe.descs), use_graph))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    note("warm-up done; timing %d steps" % args.steps)

    def barrier():
        if args.overlap:
            runner.drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # persistent launches (TCM stack; the small-batch LSTM) wait for their own workgroups with bounded polls: a launch that gave
    # up has left its code in a status word - a timed region with one of those set is not a measurement
    for p_ in (runner.pipes if hasattr(runner, "pipes") else [runner]):
        for st_ in (getattr(p_.eps, "tcm_status", None), getattr(p_.prior, "status", None), getattr(p_.prior, "tcm_status", None)):
            if st_ is not None and int(st_[0].item()) != 0:
                sys.exit("bench.py: a persistent launch gave up waiting (status %d): result invalid" % int(st_[0].item()))
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    audio_seconds = args.steps * B * world * args.seconds
    rtf = audio_seconds / elapsed

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    note("timed region: %.3f ms/step; per-stage hipEvent timing" % ms_per_step)
    # ---- roofline of the dominant kernel family: the fused BiConv(Trans)GLU blocks, live hipEvent timing
    # kernel durations are measured on the launch stream of ONE full-batch pipeline, launches back to back
    # (in the timed region above the sub-batch pipelines overlap each other, so wall time < sum of durations)
    stream = torch.cuda.current_stream().cuda_stream
    if args.streams > 1 or args.overlap:
        bank = pipe.bank
        del runner
        torch.cuda.empty_cache()
        pipe = pipeline.SamplerPipeline(dev, args.prior, gs, ds, B, L_=L_, fast_sampling=fast, bank=bank, split_bf16=args.split_bf16, dtype=dtype)
    pipe.stft.wav.copy_(wav)
    pipe.xT_in.copy_(x_T)
    torch.cuda.synchronize()
    # the strictly sequential pass (one batch at a time, one stream, hipGraph replay) beside the in-flight number.  A pipeline that
    # runs alone owns the GPU (exclusive=True): the TCM stack is then one persistent launch (csrc/tcm2.hip: tcm2s_kernel, bit-
    # identical results); the in-flight runners and the per-stage timing below keep one launch per block
    seq_steps = max(3, min(args.steps, 10))
    pipe.enhance(wav, x_T, graph=use_graph)
    torch.cuda.synchronize()
    pseq = pipeline.SamplerPipeline(dev, args.prior, gs, ds, B, L_=L_, fast_sampling=fast, bank=pipe.bank, split_bf16=args.split_bf16,
                                    dtype=dtype, exclusive=not args.tcm_launches)
    pseq.enhance(wav, x_T, graph=use_graph)
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for _ in range(seq_steps):
        pseq.run(graph=use_graph)
    torch.cuda.synchronize()
    ms_sequential = (time.perf_counter() - ts) / seq_steps * 1e3
    pseq.check()
    tcm_stack_ms = pseq.plan.time_tag(nets.TAG_TCM, stream)[0]
    del pseq
    torch.cuda.empty_cache()
    def compare_pass(note_, **kw):
        """The same workload under another arithmetic (sequential, and in flight when the main line is), measured in the same run."""
        pc = pipeline.SamplerPipeline(dev, args.prior, gs, ds, B, L_=L_, fast_sampling=fast, **kw)
        pc.enhance(wav, x_T, graph=use_graph)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(seq_steps):
            pc.run(graph=use_graph)
        torch.cuda.synchronize()
        msq = (time.perf_counter() - t0_) / seq_steps * 1e3
        res = {"ms_per_step_sequential": round(msq, 3), "value_sequential": round(B * args.seconds / (msq * 1e-3), 2), "note": note_}
        del pc
        torch.cuda.empty_cache()
        if args.overlap:                                # ... and with the same number of batches in flight as the main line
            rc = pipeline.PipelinedSampler(dev, args.prior, gs, ds, B, L_, depth=args.depth, by_batch=args.by_batch,
                                           graph=use_graph, fast_sampling=fast, **kw)
            for _ in range(max(2, args.warmup)):
                rc.submit(wav, x_T)
            rc.drain()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                rc.submit(wav, x_T)
            rc.drain()
            torch.cuda.synchronize()
            msf = (time.perf_counter() - t0_) / args.steps * 1e3
            res.update({"ms_per_step": round(msf, 3), "value": round(B * args.seconds / (msf * 1e-3), 2), "batches_in_flight": args.inflight})
            del rc
        torch.cuda.empty_cache()
        return res

    fp32_exact = bf16x3 = None
    if args.split_bf16 and not args.no_fp32_compare:
        fp32_exact = compare_pass("same workload, v_mfma_f32_32x32x2_f32 everywhere (bench.py --fp32)", split_bf16=False)
        if f16:
            bf16x3 = compare_pass("same workload with the exact three-way bf16 operand split, six bf16 products per multiply-add "
                                  "(bench.py --split bf16x3; the default until round 4)", split_bf16=True, split="bf16x3")
    per_tag = {}
    for name, tag in (("eps_block", nets.TAG_EPS_BLOCK), ("eps_conv1", nets.TAG_EPS_CONV1), ("tcm", nets.TAG_TCM),
                      ("prior_conv", nets.TAG_PRIOR), ("lstm", nets.TAG_LSTM), ("signal", nets.TAG_SIGNAL),
                      ("elementwise", nets.TAG_EW)):
        ms, cnt = pipe.plan.time_tag(tag, stream)
        per_tag[name] = {"ms": round(ms, 4), "launches": cnt}
    eps_ms = per_tag["eps_block"]["ms"] + per_tag["eps_conv1"]["ms"] + per_tag["tcm"]["ms"]
    eps_flop = EPS_GFLOP_PER_UTT_STEP * 1e9 * (T / 401.0) * B * pipe.nsteps
    achieved = eps_flop / (eps_ms * 1e-3) / 1e12
    # HBM traffic of the same launches from the committed rocprofv3 PMC passes (FETCH_SIZE | WRITE_SIZE collected
    # separately, profiles/r01_pmc_traffic_final.json); read-side doubled as MI355X_MICROARCH.md prescribes for gfx950
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r04_pmc_traffic_f16x2.json" if f16 else "r04_pmc_traffic.json")   # PMC passes of THIS arithmetic
    if not os.path.exists(tpath) and not f16:
        tpath = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    n_eps_launch = max(1, sum(per_tag[k]["launches"] for k in ("eps_block", "eps_conv1", "tcm")))
    if os.path.exists(tpath) and B == 32 and T == 401 and fast and args.prior == "GCRN":
        pj = json.load(open(tpath))["eps_net_one_pass"]
        traffic = round((pj["fetch_size_bytes_x2"] + pj["write_size_bytes"]) / pj["launches"])
    # split-bf16 mode: an fp32-equivalent FMA costs six bf16 MFMA products, so the roof of the algorithmic FLOP rate is the
    # dense bf16 peak / 6 (2.5 PF / 6); the TCM blocks inside the family still run on the fp32 matrix cores
    peak = BF16_MFMA_PEAK_TFLOPS / products if args.split_bf16 else FP32_MFMA_PEAK_TFLOPS
    if args.bf16:
        peak = BF16_MFMA_PEAK_TFLOPS
    roofline = {"bound": "mfma", "kernel": (("bglu_kernel (BiConv(Trans)GLU blocks on %s plane tensors) + tcm2_kernel" % ("f16x2" if f16 else "split-bf16")) if args.split_bf16 else
                                            "gconv2_kernel + tcm_block_kernel") + " (eps-net: BiConvGLU/BiConvTransGLU/TCM launches)",
                "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                "peak_note": (("dense f16 MFMA 2500 TFLOP/s / 3 products per fp32-equivalent multiply-add (f16x2 split)" if f16 else
                               "dense bf16 MFMA 2500 TFLOP/s / 6 products per fp32-equivalent multiply-add") if args.split_bf16
                              else "fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
                "frac": round(achieved / peak, 4),
                "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/%s)" % os.path.basename(tpath),
                "algorithmic_flop_per_launch": round(eps_flop / n_eps_launch),
                "avg_launch_ms": round(eps_ms / max(1, sum(per_tag[k]["launches"] for k in ("eps_block", "eps_conv1", "tcm"))), 5),
                "per_stage_ms": per_tag}

    # ---- which roof binds: the eps-net's arithmetic intensity at block-boundary traffic is 74 FLOP/B (10.29 GFLOP
    # over 139.3 MB per utterance-step).  The ridge of the six-product bf16 split is 417 TF / 8 TB/s = 52 FLOP/B - matrix-bound; with
    # three f16 products it is 833 / 8 = 104 FLOP/B: the same work is now HBM-bound, and the line reports that roof (algorithmic
    # bytes per launch over the same hipEvent time) with the matrix-side fraction beside it.
    eps_bytes = EPS_MB_PER_UTT_STEP * 1e6 * (T / 401.0) * B * pipe.nsteps
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    if args.split_bf16 and not args.bf16 and (eps_flop / eps_bytes) < ridge:
        gbs = eps_bytes / (eps_ms * 1e-3) / 1e9
        mf = {k: roofline[k] for k in ("achieved", "peak", "unit", "peak_note", "frac", "frac_of_fp32_mfma_peak")}
        roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "peak_note": "HBM3E 8 TB/s (MI355X_MICROARCH.md); arithmetic intensity %.0f FLOP/B at block-boundary traffic < ridge %.0f "
                                      "FLOP/B of the f16 dense peak / 3" % (eps_flop / eps_bytes, ridge),
                         "algorithmic_bytes_per_launch": round(eps_bytes / n_eps_launch), "mfma": mf})
        del roofline["frac_of_fp32_mfma_peak"]

    # ---- the single largest launch of the eps-net: algorithmic FLOPs from its descriptor / its own hipEvent time
    lib = importlib.import_module("prior-diffuse_amd._lib")

    def bglu_flops(d):
        pos0, pos1 = d.B * d.Tout * d.Fout, (d.B * d.Tout * d.Fout1 if d.p1mask else 0)
        cin = 4 if d.x0.ptr else 32
        tail = 2.0 * (2 * 32 * 32 + 32 * d.C2) + 2.0 * 64 * 32 * d.nx_n
        return (2.0 * pos0 * 2 * d.ntaps * cin * 32 + pos0 * tail +
                2.0 * pos1 * 2 * bin(d.p1mask).count("1") * cin * 32 + pos1 * tail)

    def gconv_flops(d):
        if isinstance(d, lib.BgluDesc):
            return bglu_flops(d)
        pos0 = d.B * d.Tout * d.Fout
        accs = 1 if d.epi == lib.EPI_LINEAR else 2
        cin = d.in0.C + d.in1.C
        fl = 2.0 * pos0 * accs * (d.ntaps * cin) * d.Cout
        if d.epi == lib.EPI_BIGLU:
            tail = 2.0 * (2 * 32 * 32 + 32 * d.C2)
            fl += pos0 * tail
            if d.w2:
                pos1 = d.B * d.Tout * d.Fout1
                fl += 2.0 * pos1 * 2 * (bin(d.p1mask).count("1") * cin) * 32 + pos1 * tail
            fl += 2.0 * 64 * 32 * d.nx_n * (pos0 + (d.B * d.Tout * d.Fout1 if d.w2 else 0))   # chained next-stage 1x1 tiles
        return fl

    b0, e0 = pipe.ranges["step%d" % (pipe.nsteps - 1)]
    ms_ops = [min(col) for col in zip(*[pipe.plan.time_ops(b0, e0, stream) for _ in range(3)])]
    cand = [(ms_ops[i - b0], i) for i in range(b0, e0) if isinstance(pipe.descs[i][0], (lib.GconvDesc, lib.BgluDesc))]
    top_ms, top_i = max(cand)
    top_d = pipe.descs[top_i][0]
    if isinstance(top_d, lib.BgluDesc):
        top_name, top_dual = "bglu_kernel (%d plane%s)" % (top_d.np, "s" if top_d.np > 1 else ""), bool(top_d.p1mask)
    else:
        top_name, top_dual = ("gconv3_kernel" if top_d.korder == 2 else "gconv2_kernel"), bool(top_d.w2)
    top_tf = gconv_flops(top_d) / (top_ms * 1e-3) / 1e12
    roofline["largest_launch"] = {
        "kernel": top_name + " BIGLU%s, %d taps, 32 -> 32 -> %d channels, %d x %d x %d positions" % (
            " dual-phase" if top_dual else "", top_d.ntaps, top_d.C2, top_d.B, top_d.Tout, top_d.Fout),
        "ms": round(top_ms, 4), "algorithmic_gflop": round(gconv_flops(top_d) / 1e9, 2),
        "achieved": round(top_tf, 2), "unit": "TFLOP/s", "frac": round(top_tf / peak, 4), "frac_of": "the matrix roof (%.0f TFLOP/s)" % peak}

    if args.bf16:
        # the bf16 row: the BiConv(Trans)GLU launches against the HBM roof (north_star frames configs 2/4/5 as HBM-bound) -
        # algorithmic bytes = what a launch must read and write at its block boundary (descriptor arithmetic), live hipEvents
        def bglu_bytes(d):
            pos = d.B * d.Tout
            bins_in = d.Fin if d.x0.ptr else d.hp_Fp - 4
            nbytes = pos * bins_in * (4 * 4 if d.x0.ptr else 32 * 2 * d.np)
            bins_out = (d.Fout + d.Fout1) if d.p1mask else d.Fout
            if d.nx_n:
                nbytes += pos * bins_out * 32 * 2 * d.np + (pos * bins_out * 32 * 4 if d.nx_add else 0)
                nbytes += (d.nx_n - 1) * pos * d.Fout * 32 * 4
            else:
                nbytes += pos * bins_out * d.C2 * 4
            return nbytes

        idx = [i for i in range(b0, e0) if isinstance(pipe.descs[i][0], lib.BgluDesc)]
        bsum = sum(bglu_bytes(pipe.descs[i][0]) for i in idx)
        tsum = sum(ms_ops[i - b0] for i in idx) * 1e-3
        roofline["bf16_blocks_hbm"] = {"bound": "hbm", "kernel": "bglu_kernel<.., 1> (15 launches of one eps-net forward)",
                                       "achieved": round(bsum / tsum / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(bsum / tsum / 8e12, 4), "algorithmic_bytes": bsum, "ms": round(tsum * 1e3, 4)}
        roofline["peak_note"] = "bf16 mode: blocks issue ONE bf16 MFMA product per multiply-add: roof of their algorithmic FLOP rate = dense bf16 peak"
    # ---- the reference's own entry point: generate_wav, B = 1, one 4 s file after the other (wav read -> enhance -> wav write)
    file_loop = None
    if fast and args.prior == "GCRN" and not args.bf16 and not args.no_file_loop:
        import argparse as _ap
        import tempfile

        import numpy as np

        wavio = importlib.import_module("prior-diffuse_amd.wavio")
        trainer_mod = importlib.import_module("prior-diffuse_amd.trainer")
        nfiles = 8
        with tempfile.TemporaryDirectory() as td:
            src, dst = os.path.join(td, "noisy"), os.path.join(td, "out")
            os.makedirs(src)
            rng = np.random.default_rng(7)
            for i in range(nfiles):
                wavio.write_wav(os.path.join(src, "f%02d.wav" % i), 0.1 * rng.standard_normal(L_))
            ns = _ap.Namespace
            tr = trainer_mod.ComplexDDPMTrainer(
                ns(retrain=False, joint=True, draw=False, sigma=False, checkpoint="x", generated_wav=dst),
                ns(model=ns(name="GCRN"), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt")),
                device=dev, prior_state_dict=gs, ddpm_state_dict=ds)
            tr.generate_wav(load_pre_train=False, data_path=src)          # warm-up: packs the weights, records the plan
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            tr.generate_wav(load_pre_train=False, data_path=src)
            torch.cuda.synchronize()
            tfl = time.perf_counter() - tf0
            del tr
        file_loop = {"files": nfiles, "seconds_per_file": args.seconds, "wall_s": round(tfl, 4),
                     "value": round(nfiles * args.seconds / tfl, 1), "unit": "audio_s/s",
                     "ms_per_file": round(tfl / nfiles * 1e3, 2),
                     "note": "ComplexDDPMTrainer.generate_wav (trainer/complex_ddpm_trainer.py:903-1018), B = 1, wav read + "
                             "enhance + wav write per file, plan and weights resident, hipGraph replay once a length comes back; the trainer owns "
                             "the GPU (exclusive): the prior's LSTM is one persistent launch (csrc/lstmp.hip), the TCM stack one launch per forward"}
    cpu = None
    if not args.no_cpu_baseline and world == 1:      # the CPU leg belongs to the N = 1 line only (rank 0 of a larger job skips it)
        from oracle import restate as R

        t_cpu0 = time.perf_counter()
        params = importlib.import_module("prior-diffuse_amd.params").params
        # the GPU box gives one GPU a 16-core share of the host; never oversubscribe it
        nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
        torch.set_num_threads(nthreads)
        cb = 8 if "aia" not in args.prior else 2    # about 3 x 3-5 s of host work
        if not fast:
            cb = 2                                   # 50 steps: ~8x the work per utterance
        cb = max(1, min(cb, int(cb * 4.0 / args.seconds)))
        note("cpu baseline: oracle on %d host threads, %d utterances" % (nthreads, cb))
        w_cpu, x_cpu = synth.synthetic_waveforms(B * world, L_, seed=1234)
        w_cpu, x_cpu = w_cpu[:cb], x_cpu[:cb]
        runs = []
        with torch.no_grad():       # one warm-up, then the median of three timed runs
            R.enhance(args.prior, gs, ds, w_cpu[:1], x_cpu[:1], params.noise_schedule, params.inference_noise_schedule, True)
            for _ in range(3):
                tc = time.perf_counter()
                R.enhance(args.prior, gs, ds, w_cpu, x_cpu, params.noise_schedule, params.inference_noise_schedule, fast)
                runs.append(time.perf_counter() - tc)
        tc = sorted(runs)[1]
        cpu_leg_s = time.perf_counter() - t_cpu0
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
        except OSError:
            pass
        cpu = {"value": round(cb * args.seconds / tc, 3), "unit": "audio_s/s", "cores": nthreads, "kind": "port",
               "cpu_model": model, "leg_wall_s": round(cpu_leg_s, 1),
               "sample": "oracle (torch-CPU fp32 restatement), %d of the %d utterances in one batch, full path incl. "
                         "STFT/ISTFT, 1-utterance warm-up + median of 3 timed runs (%s s)" % (
                             cb, B, " / ".join("%.1f" % r for r in runs))}

    out = {
        "metric": "enhanced-audio sec/sec (RTF), %s, B=%d" % ("6-step fast sampling" if fast else "50-step full schedule", B),
        "value": round(rtf, 2), "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "ms_per_step_sequential": round(ms_sequential, 3),
        # box-independent: summed hipEvent durations of the eps-net launches (the roofline's denominator) and of everything
        "eps_net_kernel_ms": round(eps_ms, 4), "all_kernel_ms": round(sum(v["ms"] for v in per_tag.values()), 4),
        "sequential_note": "one batch at a time owns the GPU: TCM stack as one persistent launch (%.3f ms per pass; %.3f as %d launches in "
                           "the in-flight pass)" % (tcm_stack_ms, per_tag["tcm"]["ms"], per_tag["tcm"]["launches"]),
        "value_sequential": round(B * args.seconds / (ms_sequential * 1e-3), 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.bf16 else (args.split if args.split_bf16 else "f32"), "data": "synthetic",
        "dtype_note": ("OPT-IN bf16 mode: plain bf16 operands (one MFMA product, fp32 accumulate) and bf16 conv1 / bottleneck tensors in the eps-net's "
                       "BiConv(Trans)GLU and TCM blocks; prior as in the default; tolerance 3e-2 rel-L2 (tests/test_gpu_round2.py::"
                       "test_bf16_mode_tolerance) - not the graded line") if args.bf16 else (("matrix-core kernels (eps-net blocks, TCM stack, GCRN's gated convolutions): every fp32 operand scaled by a power of two and carried "
                        "as fp16 hi + lo (within half an fp32 ulp inside the fp16 window), three f16 MFMA products per multiply-add, fp32 "
                        "accumulate - fp32-equivalent, same goldens and tolerances as the exact-fp32 kernels (tests/test_gpu_f16x2.py); "
                        "everything else fp32") if f16 else
                       ("eps-net blocks: fp32 operands split exactly into three bf16 terms, six-product bf16 MFMA, fp32 accumulate "
                        "(fp32-level accuracy, same parity tolerances); everything else fp32")) if args.split_bf16 else "fp32 throughout",
        "frames_per_s_per_gpu": round(args.steps * B * T / elapsed, 1),
        "config": {"workload": "B=%d x %.0f s 16 kHz utterances per GPU, [B,2,%d,161] spectrograms, %s prior + "
                               "DiffUNet1 %d-step sampling, STFT..ISTFT, seeded random weights" % (B, args.seconds, T, args.prior, pipe.nsteps),
                   "global_batch": B * world, "frames": T, "parallelism": "batch-shard x%d" % world,
                   "graph": use_graph and (not args.overlap or args.by_batch), "streams_per_gpu": args.streams,
                   "batches_in_flight": args.inflight},
        "roofline": roofline, "cpu_baseline": cpu, "fp32_exact": fp32_exact, "bf16x3": bf16x3, "file_loop_b1": file_loop,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def dry_run(args, dist, world, rank):
    """CPU rehearsal of everything around the kernels: rank set-up, sharding of the global batch, barrier, max-reduce of
    the elapsed time and the single JSON line from rank 0.  No GPU, no kernels, no throughput claim."""
    shard = importlib.import_module("prior-diffuse_amd.shard")
    B = args.batch
    lo, hi = shard.shard_range(B * world, world, rank)
    assert hi - lo == B

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))          # uneven ranks: the reduction must return the slowest
    barrier()
    elapsed = time.perf_counter() - t0
    mine = elapsed
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "enhanced-audio sec/sec (RTF), 6-step fast sampling, B=%d" % B, "value": None,
                          "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "rank0_elapsed_s": round(mine, 4), "max_elapsed_s": round(elapsed, 4),
                          "config": {"workload": "dry run (no kernels)", "global_batch": B * world,
                                     "parallelism": "batch-shard x%d" % world}}))
    return 0


if __name__ == "__main__":
    main()
