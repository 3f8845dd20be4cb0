"""The whole sampling path as ONE recorded plan:

    [wav -> RMS normalise -> STFT -> sqrt-compress]            :921-937
    prior(feat) -> X_init; init = X_init / 11                   :939-942
    [--sigma: x_T *= sqrt(mask(init))]                          :951-956
    for n = S-1 .. 0: eps = DiffUNet1(audio, init, T[n]);
                      audio = c1 (audio - c2 eps)               :964-992
    spec = (audio + init) * 11                                  :995-996
    [square-decompress -> ISTFT -> * c]                         :1004-1016

(line numbers: reference trainer/complex_ddpm_trainer.py).  Once built for a (B, T) the
plan owns every buffer in HBM; ``run()`` is a single C call that issues the launches, or
a single hipGraph replay.
"""
import numpy as np
import torch

from . import _lib as L
from . import nets, packing
from .params import PRIOR_SCALE_C, params as default_params
from .schedule import inference_schedule, step_coefficients

F0 = 161


def _expect(t, like, name):
    """Inputs must have exactly the geometry the plan was recorded for (``copy_`` would silently broadcast)."""
    if tuple(t.shape) != tuple(like.shape) or t.dtype != like.dtype:
        raise ValueError(f"{name}: expected {tuple(like.shape)} {like.dtype}, got {tuple(t.shape)} {t.dtype}")


class SamplerPipeline:
    # the fp32-equivalent operand split of the matrix-core kernels (see ``split``)
    default_split = "f16x2"

    def __init__(self, device, prior_name, prior_sd, ddpm_sd, B, T=None, L_=None, fast_sampling=True,
                 use_sigma=False, params=default_params, with_signal=None, deltamu=False, cond="init", bank=None,
                 split_bf16=None, xT_plus_init=None, dtype="f32", exclusive=False, split=None):
        """deltamu: the alternative parameterisation of utils/params.py:36 — ddpm_sd is a ``Nocon`` state_dict,
        x_T = noise + X_init/11 (:947-948), eps = Nocon(x, t) (:970-971), no final ``+ X_init`` (:995).
        cond (deltamu False): what conditions DiffUNet1 — "init": X_init/11 (pirorgrad, :967-969, + X_init at the end,
        :994-995); "feat": the noisy feature / 11 (the branch with neither flag set, :74-75, :972-974; no final add).
        xT_plus_init: x_T = noise + X_init/11 (:946-949 tests ``self.deltamu`` on its own, while model selection :70-73
        and the eps call :967-971 let ``pirorgrad`` win): default = deltamu; True with deltamu False is the reference's
        behaviour when BOTH flags are set (DiffUNet1 conditioned on X_init, start from noise + X_init/11, final + X_init).
        dtype: "f32" (default: fp32 or fp32-equivalent split-bf16 arithmetic, see split_bf16) or "bf16" - the OPT-IN reduced
        precision mode of BASELINE configs 2/4/5: the eps-net's BiConv(Trans)GLU and TCM blocks multiply plain bf16 operands
        (one MFMA product, fp32 accumulate) and exchange their conv1 / bottleneck tensors as bf16 (csrc/bglu.hip, csrc/tcm2.hip,
        one plane); round 4: the priors' GEMM-shaped convolutions (GCRN's gated convolutions and LSTM projection, DB-AIAT's
        dense blocks) multiply plain bf16 operands as well (csrc/gconv4.hip, korder 4); the diffusion state, the skip halves,
        the residual stream of the TCM stack, the recurrences and every tensor of the priors stay fp32.  Its tolerance is its own (stated in
        tests/test_gpu_round2.py::test_bf16_mode_tolerance), it is never the default and never the graded bench line.
        bank: a ``nets.WeightBank`` shared with other pipelines built from the same state_dicts (packed weights are
        uploaded once, every further (B, T) only records descriptors).
        exclusive: this pipeline is the only work on the GPU while it runs (one batch in flight - ``ComplexDDPMTrainer``
        passes True): launches whose workgroups wait for each other may then be used - the TCM stack as one launch
        (csrc/tcm2.hip: tcm2s_kernel, bit-identical) and, for B <= 4, the GCRN prior's LSTM (csrc/lstmp.hip; 1e-5 from, not
        bit-identical to, the kernels large batches take).
        ``ConcurrentSampler`` / ``PipelinedSampler`` and the sharded path keep False.
        split_bf16: the eps-net's BIGLU blocks and the priors' GEMM-shaped convolutions on the bf16 matrix cores with exact three-way
        operand splits - fp32-level accuracy at 16/6 of the fp32 MFMA rate (csrc/gconv3.hip); None: on for fast sampling,
        off (exact fp32 MFMA) for the full 50-step schedule.
        split (with split_bf16, dtype "f32"): which fp32-equivalent operand split the matrix-core kernels use - "bf16x3": the exact
        three-way bf16 split, six bf16 products per multiply-add; "f16x2": fp16 hi + lo of the power-of-two scaled operand (within
        half an fp32 ulp inside the fp16 window, see include/pdse.h PDSE_F16_ACT_EXP), three f16 products, two thirds of the plane
        bytes.  Both are held to the same goldens and tolerances.  None: ``SamplerPipeline.default_split``."""
        if L_ is not None:
            T = 1 + L_ // 160
        if with_signal is None:
            with_signal = L_ is not None
        if cond not in ("init", "feat"):
            raise ValueError("cond must be 'init' or 'feat'")
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        if dtype == "bf16" and (deltamu or split_bf16 is False):
            raise ValueError("the bf16 mode runs the DiffUNet1 blocks on the bf16 matrix cores (split_bf16 False / Nocon: no)")
        self.dtype = dtype
        split = self.default_split if split is None else split
        if split not in ("bf16x3", "f16x2"):
            raise ValueError("split must be 'bf16x3' or 'f16x2'")
        self.split = split
        if split_bf16 is None:
            # 6-step fast sampling: split-bf16 blocks (2e-6 from the fp32 CPU path, tolerance 1e-4).  The full 50-step
            # schedule amplifies rounding noise ~500x (the fp32 CPU path itself sits 4-7e-5 from the exact answer), so
            # it keeps exact fp32 MFMA arithmetic - BASELINE config 3 is an fp32 configuration anyway.
            split_bf16 = bool(fast_sampling)
        self.B, self.T, self.L = B, T, L_
        self.device = torch.device(device)
        self.ctx = ctx = nets.Ctx(device, bank)
        self.bank = ctx.bank
        self.plan = L.Plan(self.device) if self.device.type == "cuda" else None
        alpha, beta, alpha_cum, sigmas, Tarr = inference_schedule(params, fast_sampling)
        self.schedule = (alpha, beta, alpha_cum, sigmas, Tarr)
        c1, c2 = step_coefficients(alpha, beta, alpha_cum)
        S = len(alpha)
        self.nsteps = S
        self.ranges = {}
        self.descs = []

        # ---- builders share one plan object and one descriptor list
        def adopt(pb):
            pb.descs = self.descs
            return pb

        self.stft = adopt(nets.StftPlan(ctx, B, L_, plan=self.plan, split_bf16=split_bf16)) if with_signal else None
        pplanes = 1 if dtype == "bf16" else None     # bf16 mode: the priors' GEMM-shaped convolutions on plain bf16 operands too (korder 4)
        aplanes = pplanes if pplanes is not None else ((2 if split == "f16x2" else 3) if split_bf16 else None)   # DB-AIAT: dense blocks (np) and GEMM convolutions
        if prior_name == "GCRN":
            gplanes = pplanes if pplanes is not None else ((2 if split == "f16x2" else 3) if split_bf16 else None)   # korder 5 / 3 (csrc/gconv4.hip)
            self.prior = adopt(nets.GcrnPlan(ctx, prior_sd, B, T, plan=self.plan, split_bf16=split_bf16 or dtype == "bf16", exclusive=exclusive,
                                             planes=gplanes))
        elif prior_name == "DiffUNet":
            self.prior = adopt(nets.EpsNetPlan(ctx, prior_sd, B, T, time_cond=False, plan=self.plan, split_bf16=split_bf16, exclusive=exclusive))
        elif prior_name == "aia_complex_trans_ri":
            self.prior = adopt(nets.AiaPlan(ctx, prior_sd, B, T, plan=self.plan, split_bf16=split_bf16 or dtype == "bf16", planes=aplanes))
        elif prior_name == "dual_aia_trans_merge_crm":
            self.prior = adopt(nets.DualAiaPlan(ctx, prior_sd, B, T, plan=self.plan, split_bf16=split_bf16 or dtype == "bf16", planes=aplanes))
        else:
            raise ValueError("prior %r not built (GCRN, DiffUNet, aia_complex_trans_ri, dual_aia_trans_merge_crm)" % prior_name)
        if dtype == "bf16":
            split_bf16 = True
        self.eps = adopt(nets.EpsNetPlan(ctx, ddpm_sd, B, T, time_cond=True, nsteps=S, plan=self.plan,
                                         with_pre=not deltamu, split_bf16=split_bf16,
                                         planes=1 if dtype == "bf16" else (2 if split == "f16x2" else 3),
                                         exclusive=exclusive))
        self.split_bf16 = self.eps.split_bf16
        self.deltamu = deltamu
        self.xT_plus_init = xT_plus_init = bool(deltamu if xT_plus_init is None else xT_plus_init)
        self.cond_feat = cond_feat = (cond == "feat") and not deltamu
        self.istft = adopt(nets.IstftPlan(ctx, B, T, L_, plan=self.plan, split_bf16=split_bf16)) if with_signal else None

        self.feat = self.prior.x                     # prior input = compressed spectrogram
        # X_init / 11: the eps-net's conditioning input only in the pirorgrad parameterisation
        self.init = self.eps.x_init if not (deltamu or cond_feat) else ctx.alloc(B, 2, T, F0)
        self.zero = ctx.alloc(B, 2, T, F0, zero=True) if (deltamu or cond_feat) else None
        self.audio = self.eps.x                      # x_t, updated in place
        self.spec = self.istft.spec if with_signal else ctx.alloc(B, 2, T, F0)
        self.xT_in = ctx.alloc(B, 2, T, F0)          # injected x_T (kept so a replay starts from it)
        n = B * 2 * T * F0

        def mark(name, fn):
            b = len(self.descs)
            fn()
            self.ranges[name] = (b, len(self.descs))

        if with_signal:
            mark("stft", lambda: self.stft.build(feat=self.feat))
        if prior_name in ("GCRN", "aia_complex_trans_ri", "dual_aia_trans_merge_crm"):
            mark("prior", lambda: self.prior.build(x=self.feat, out=self.prior.out))
        else:
            mark("prior", lambda: self.prior.build_step(0, x=self.feat, out=self.prior.out))

        def ew(op, a, b=None, c=None, out=None, s0=0.0, s1=0.0, s2=0.0):
            d = L.EwDesc()
            d.a, d.b, d.c, d.out = a.data_ptr(), nets.Ctx.ptr(b), nets.Ctx.ptr(c), out.data_ptr()
            d.n, d.s0, d.s1, d.s2, d.op = n, s0, s1, s2, op
            self.eps.add(d, nets.TAG_EW)

        def prologue():
            ew(L.EW_DIV, self.prior.out, out=self.init, s0=PRIOR_SCALE_C)
            if cond_feat:
                ew(L.EW_DIV, self.feat, out=self.eps.x_init, s0=PRIOR_SCALE_C)        # batch_feat /= c  (:943)
            src = self.xT_in
            if xT_plus_init:
                ew(L.EW_ADD_MUL, self.xT_in, b=self.init, out=self.audio, s0=1.0)     # randn_like(init) + init  (:947-948)
                src = self.audio
            if use_sigma:                                                             # audio * sqrt(mask(init))  (:951-956)
                d = L.SigmaDesc()
                d.init, d.a, d.out = self.init.data_ptr(), src.data_ptr(), self.audio.data_ptr()
                d.maxbuf = ctx.alloc(B * 2).data_ptr()
                d.plane, d.nplanes = T * F0, B * 2
                self.eps.add(d, nets.TAG_EW)
            elif not xT_plus_init:
                ew(L.EW_COPY, self.xT_in, out=self.audio)
            self.eps.build_time()

        mark("prologue", prologue)
        # diffusion-step rows are stored in loop order: row i is step n = S-1-i
        self.eps.tsteps.copy_(torch.from_numpy(np.ascontiguousarray(Tarr[::-1]))[:, None].expand(S, B))
        for i in range(S):
            nstep = S - 1 - i

            def one(i=i, nstep=nstep):
                self.eps.build_step(i, x=self.audio, x_init=None if deltamu else (self.eps.x_init if cond_feat else self.init),
                                    out=self.eps.out)
                if nstep > 0:
                    ew(L.EW_UPDATE, self.audio, b=self.eps.out, out=self.audio, s0=float(c1[nstep]), s1=float(c2[nstep]))
                else:
                    ew(L.EW_UPDATE_FINAL, self.audio, b=self.eps.out, c=self.zero if self.zero is not None else self.init,
                       out=self.spec,
                       s0=float(c1[0]), s1=float(c2[0]), s2=PRIOR_SCALE_C)

            mark("step%d" % nstep, one)
        if with_signal:
            mark("istft", lambda: self.istft.build(spec=self.spec, c=self.stft.c))
        if self.plan is not None:
            self.plan.keep(ctx.keep, ctx.bank)
        self._graph_stream = None

    # ------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def run(self, first=None, last=None, graph=False):
        """Run ranges first..last (names from ``self.ranges``), whole plan by default."""
        if self.plan is None:
            raise L.PdseError("SamplerPipeline built on a CPU context cannot run: there is no CPU fallback")
        if graph:
            if first is not None or last is not None:
                raise ValueError("graph replay always runs the whole plan")
            if not self.plan.has_graph:
                self._graph_stream = torch.cuda.Stream(self.device)
                with torch.cuda.stream(self._graph_stream):
                    self.plan.build_graph(self._graph_stream.cuda_stream)
                self._graph_stream.synchronize()
            self.plan.launch_graph(self._stream())
            return
        b = self.ranges[first][0] if first else 0
        e = self.ranges[last][1] if last else len(self.descs)
        self.plan.run_range(b, e, self._stream())

    def check(self):
        """Synchronises and raises if a persistent launch of the last run gave up (its workgroups wait for each other
        with bounded polls: another workload on the GPU can keep them from all being resident)."""
        for net, field, what in ((self.prior, "status", "persistent LSTM gave up at step %d"), (self.eps, "tcm_status", "TCM stack launch gave up at block %d"),
                                 (self.prior, "tcm_status", "TCM stack launch of the prior gave up at block %d")):
            st = getattr(net, field, None)
            if st is not None:
                code = int(st[0].item())
                if code:
                    st.zero_()
                    raise L.PdseError((what % (code - 1)) + ": its workgroups wait for each other and were not all resident in time (is "
                                      "another workload sharing the GPU?); build the pipeline with exclusive=False")
        if self.split == "f16x2" and self.split_bf16 and self.dtype == "f32" and not bool(torch.isfinite(self.spec).all()):
            raise L.PdseRangeError("non-finite values in the enhanced spectrogram of an f16x2 pass: an activation beyond +-%g left the fp16 "
                                   "window (include/pdse.h: PDSE_F16_ACT_EXP), or the input was not finite; build the pipeline with "
                                   "split='bf16x3' (ComplexDDPMTrainer does so for this geometry by itself)" % (65504.0 / 2 ** packing.F16_ACT_EXP))

    def sample(self, feat, x_T, graph=False):
        """feat, x_T [B,2,T,161] -> (enhanced compressed spectrogram, X_init); the
        spectrogram-level body of generate_wav (:939-998)."""
        _expect(feat, self.feat, "feat")
        _expect(x_T, self.xT_in, "x_T")
        self.feat.copy_(feat)
        self.xT_in.copy_(x_T)
        if graph and self.stft is None:
            self.run(graph=True)
        else:
            self.run("prior", "step0")
        return self.spec.clone(), self.prior.out.clone()

    def enhance(self, wav, x_T, graph=False, lens=None):
        """wav [B,L], x_T [B,2,T,161] -> (enhanced wav [B,L], spectrogram [B,2,T,161]).
        lens: true lengths of zero-padded utterances (RMS normalisation over each utterance's own samples,
        utils/dataset.py:45-58); default: every utterance fills L."""
        if self.stft is None:
            raise ValueError("pipeline was built without the signal front/back end (pass L_)")
        _expect(wav, self.stft.wav, "wav")
        _expect(x_T, self.xT_in, "x_T")
        self.stft.wav.copy_(wav)
        self.xT_in.copy_(x_T)
        if lens is None:
            self.stft.lens.fill_(self.L)
        else:
            self.stft.lens.copy_(torch.as_tensor(lens, dtype=torch.int32))
        self.run(graph=graph)
        return self.istft.wav.clone(), self.spec.clone()


class ConcurrentSampler:
    """The same path with the batch cut into ``nsplit`` contiguous sub-batches, each a complete
    ``SamplerPipeline`` replayed on its own HIP stream.

    Why: the path alternates MFMA-bound phases (BiConvGLU blocks) with strictly sequential,
    latency-bound ones (401 LSTM frames x 2 layers, 18 dilated TCM residuals x 6 steps) that
    leave most of the 256 CUs idle.  Utterances are independent, so while one sub-batch walks
    its LSTM/TCM chain the other's convolutions fill the chip.  Results are bit-identical to
    the single-pipeline run (tiling never crosses a batch item)."""

    def __init__(self, device, prior_name, prior_sd, ddpm_sd, B, T=None, L_=None, nsplit=2, **kw):
        from .shard import shard_range

        self.device = torch.device(device)
        self.B, self.nsplit = B, max(1, min(nsplit, B))
        self.spans = [shard_range(B, self.nsplit, r) for r in range(self.nsplit)]
        kw.setdefault("bank", nets.WeightBank())    # one packed copy of the weights for all sub-batch pipelines
        kw["exclusive"] = False                     # the sub-batches share the GPU: no launch may wait for its own workgroups
        self.pipes = [SamplerPipeline(device, prior_name, prior_sd, ddpm_sd, hi - lo, T=T, L_=L_, **kw)
                      for lo, hi in self.spans]
        self.T, self.L = self.pipes[0].T, self.pipes[0].L
        self.nsteps = self.pipes[0].nsteps
        self.streams = [torch.cuda.Stream(self.device) for _ in self.pipes]
        self._graphs = False

    def _run_all(self, graph, first=None, last=None):
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        for p, s in zip(self.pipes, self.streams):
            s.wait_event(start)
            with torch.cuda.stream(s):
                if graph:
                    if not p.plan.has_graph:
                        p.plan.build_graph(s.cuda_stream)
                    p.plan.launch_graph(s.cuda_stream)
                else:
                    b = p.ranges[first][0] if first else 0
                    e = p.ranges[last][1] if last else len(p.descs)
                    p.plan.run_range(b, e, s.cuda_stream)
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)

    def enhance(self, wav, x_T, graph=False):
        for p, (lo, hi) in zip(self.pipes, self.spans):
            p.stft.wav.copy_(wav[lo:hi])
            p.xT_in.copy_(x_T[lo:hi])
        self._run_all(graph)
        return (torch.cat([p.istft.wav for p in self.pipes], 0), torch.cat([p.spec for p in self.pipes], 0))

    def sample(self, feat, x_T, graph=False):
        for p, (lo, hi) in zip(self.pipes, self.spans):
            p.feat.copy_(feat[lo:hi])
            p.xT_in.copy_(x_T[lo:hi])
        if graph and self.pipes[0].stft is None:
            self._run_all(True)
        else:
            self._run_all(False, "prior", "step0")
        return (torch.cat([p.spec for p in self.pipes], 0), torch.cat([p.prior.out for p in self.pipes], 0))


class PipelinedSampler:
    """Throughput mode for a stream of batches: two ``SamplerPipeline`` instances (ping-pong
    buffers) and two HIP streams.  Stage 1 of batch n+1 (STFT + prior — dominated by the
    latency-bound LSTM frame chain, which leaves most CUs idle) runs on the *prior* stream
    while stage 2 of batch n (reverse loop + ISTFT — MFMA-bound) runs on the *loop* stream.
    Per-batch results are bit-identical to ``SamplerPipeline.enhance``; only the schedule
    differs.  ``submit`` returns immediately; ``result`` waits for that batch."""

    # measurement switch (bench.py --inflight-stack): let the in-flight pipelines use the launches that wait for their own workgroups
    # (the TCM stack as one launch; at B = 32 nothing else).  Measured slower in rounds 4 (17.8 vs 17.4 ms) - see DESIGN.md 4.1b.
    stack_in_flight = False

    def __init__(self, device, prior_name, prior_sd, ddpm_sd, B, L_, depth=2, by_batch=False, graph=True, **kw):
        """depth: batches in flight (= buffer sets).  by_batch False: two stage streams (prior | loop);
        True: every batch runs start to end on its own stream, ``depth`` streams round-robin."""
        self.device = torch.device(device)
        self.depth, self.by_batch, self.graph = depth, by_batch, graph
        kw.setdefault("bank", nets.WeightBank())    # the in-flight buffer sets share one packed copy of the weights
        kw["exclusive"] = bool(self.stack_in_flight)   # batches in flight share the GPU: no launch may wait for its own workgroups
        self.pipes = [SamplerPipeline(device, prior_name, prior_sd, ddpm_sd, B, L_=L_, **kw) for _ in range(depth)]
        self.s_prior = torch.cuda.Stream(self.device, priority=-1)   # tiny dependent launches: schedule them first
        self.s_loop = torch.cuda.Stream(self.device)
        self.s_batch = [torch.cuda.Stream(self.device) for _ in range(depth)] if by_batch else []
        self.n = 0
        self.done = [None] * depth      # event: batch in slot finished
        self.nsteps = self.pipes[0].nsteps
        if by_batch and graph:          # capture every buffer set's graph now, not inside somebody's timed region
            for p, st in zip(self.pipes, self.s_batch):
                with torch.cuda.stream(st):
                    p.plan.build_graph(st.cuda_stream)
            torch.cuda.synchronize(self.device)

    def submit(self, wav, x_T):
        slot = self.n % self.depth
        p = self.pipes[slot]
        if self.by_batch:
            cur = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            st = self.s_batch[slot]
            with torch.cuda.stream(st):
                st.wait_event(ready)                              # same stream as the slot's previous batch: ordered
                for src in (wav, x_T):                            # inputs were allocated on the caller's stream: tell
                    if src.is_cuda:                               # the caching allocator this stream still reads them
                        src.record_stream(st)
                p.stft.wav.copy_(wav, non_blocking=True)
                p.xT_in.copy_(x_T, non_blocking=True)
                if self.graph:        # one hipGraph per in-flight buffer set: ~8000 launches become one host call
                    if not p.plan.has_graph:
                        p.plan.build_graph(st.cuda_stream)
                    p.plan.launch_graph(st.cuda_stream)
                else:
                    p.plan.run_range(0, len(p.descs), st.cuda_stream)
                self.done[slot] = torch.cuda.Event()
                self.done[slot].record(st)
            self.n += 1
            return slot
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.s_prior):
            self.s_prior.wait_event(ready)
            if self.done[slot] is not None:
                self.s_prior.wait_event(self.done[slot])      # the slot's buffers are free again
            for src in (wav, x_T):
                if src.is_cuda:
                    src.record_stream(self.s_prior)
            p.stft.wav.copy_(wav, non_blocking=True)
            p.xT_in.copy_(x_T, non_blocking=True)
            p.plan.run_range(p.ranges["stft"][0], p.ranges["prior"][1], self.s_prior.cuda_stream)
            staged = torch.cuda.Event()
            staged.record(self.s_prior)
        with torch.cuda.stream(self.s_loop):
            self.s_loop.wait_event(staged)
            p.plan.run_range(p.ranges["prologue"][0], len(p.descs), self.s_loop.cuda_stream)
            self.done[slot] = torch.cuda.Event()
            self.done[slot].record(self.s_loop)
        self.n += 1
        return slot

    def result(self, slot):
        torch.cuda.current_stream(self.device).wait_event(self.done[slot])
        p = self.pipes[slot]
        return p.istft.wav, p.spec

    def drain(self):
        cur = torch.cuda.current_stream(self.device)
        for e in self.done:
            if e is not None:
                cur.wait_event(e)
