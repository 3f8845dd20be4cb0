"""Plan builders: turn reference ``state_dict``s into recorded operator sequences
(include/pdse.h plans) for the networks and signal-processing stages on the path.

  EpsNetPlan     DiffUNet1.forward(x, x_init, t)        model/diff3.py:37-57
  (same builder, ``time_cond=False``)  prior DiffUNet   model/diff.py:23-33
  GcrnPlan       GCRN.forward                           model/gcrn.py:136-166
  StftPlan / IstftPlan   front/back end                 trainer/complex_ddpm_trainer.py:921-937, :1004-1016

Every tensor lives in HBM for the lifetime of the plan (weights packed once, workspaces
allocated once per (B, T)); a plan run is a pure sequence of kernel launches with no
allocation and no host synchronisation, so it can be captured into a hipGraph.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import packing as P

F0 = 161
BIG = 1 << 30

# tags for bench/roofline attribution
TAG_NONE, TAG_EPS_BLOCK, TAG_EPS_CONV1, TAG_TCM, TAG_PRIOR, TAG_LSTM, TAG_SIGNAL, TAG_EW = range(8)


class Ctx:
    """Device-memory helper: torch tensors for storage, raw pointers for the ABI."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.keep = []

    def alloc(self, *shape, zero=False):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def up(self, arr, dtype=np.float32):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(self.device)
        self.keep.append(t)
        return t

    @staticmethod
    def ptr(t, off=0):
        return 0 if t is None else t.data_ptr() + 4 * int(off)


class PlanBase:
    force_generic = False   # tests: route every convolution through the un-pipelined kernel

    def __init__(self, ctx, plan=None):
        self.ctx = ctx
        self.descs = []  # python-side copy of every descriptor (tests replay these on the CPU emulator)
        self.plan = plan if plan is not None else (L.Plan() if ctx.device.type == "cuda" else None)

    def add(self, desc, tag=TAG_NONE):
        self.descs.append((desc, tag))
        if self.plan is not None:
            self.plan.add(desc, tag)

    def finish(self):
        if self.plan is not None:
            self.plan.keep(self.ctx.keep)

    # ---- descriptor helpers -------------------------------------------------
    def src(self, t, C_, sb, sc, st, sf, off=0, act=L.ACT_NONE):
        return L.Src(Ctx.ptr(t, off), sb, sc, st, sf, C_, act)

    def gconv(self, *, in0, in1=None, Tin, Fin, taps, sf_in, wk0, wk1=None, Cout, bias0=None, bias0_sb=0,
              bias1=None, bias1_sb=0, epi=L.EPI_LINEAR, act=L.ACT_NONE, act_slope=0.0, post=None,
              padrow=None, padrow_sb=0, padrow_off=0, xf=None, cin1=False, chain=None, resid=None, out,
              out_strides, out_off=0, out_cr=1, B, Tout, Fout, tag=TAG_NONE, bias0_off=0, phase1=None, nx=None,
              bias1_off=0, bias_t0=None):
        """wk0/wk1: [K, Cout] float64 k-major matrices (packed here); biases/post: numpy or device tensors.
        phase1 (dual-phase transposed conv, BIGLU): dict(wk2, wk3, mask, ntaps1, Fout1).
        nx (BIGLU, C2 == 64): dict(keep, row0, tiles=[dict(w [32,64], bias, bias_off, bias_sb, out, strides (sb, sc, st,
        sf), off, add)]) - 1x1 convolutions chained onto the block output (include/pdse.h: nx_*)."""
        ctx = self.ctx
        d = L.GconvDesc()
        d.in0 = in0
        d.in1 = in1 if in1 is not None else L.Src(0, 0, 0, 0, 0, 0, 0)
        d.Tin, d.Fin = Tin, Fin
        d.padrow, d.padrow_sb = Ctx.ptr(padrow, padrow_off), padrow_sb
        tp = ctx.up(P.taps_array(taps), np.int32)
        d.taps, d.ntaps, d.sf_in = tp.data_ptr(), len(taps), sf_in
        if xf is not None:
            d.xf_mode = xf["mode"]
            d.xf_scale0, d.xf_shift0 = ctx.up(xf["scale0"]).data_ptr(), ctx.up(xf["shift0"]).data_ptr()
            d.xf_slope0 = float(xf["slope0"])
            if xf["mode"] == 2:
                d.xf_scale1, d.xf_shift1 = ctx.up(xf["scale1"]).data_ptr(), ctx.up(xf["shift1"]).data_ptr()
                d.xf_slope1 = float(xf["slope1"])
        d.cin1 = 1 if cin1 else 0
        wk0 = np.asarray(wk0)
        key = (epi, len(taps), d.in1.C > 0, d.xf_mode)
        if P.v2_supported(*key, cin1) and padrow is None and not self.force_generic:
            rows = P.korder1_rows(len(taps), d.in0.C, d.in1.C, P.V2_CP[key])    # pipelined kernel order
            w0 = P.pack_a4(wk0, rows)
            d.w0, d.ksteps, d.Cout = ctx.up(w0).data_ptr(), len(rows) // 2, Cout
            if wk1 is not None:
                d.w1 = ctx.up(P.pack_a4(wk1, rows)).data_ptr()
            d.korder = 1
            if phase1 is not None:
                rows1 = P.korder1_rows(phase1["ntaps1"], d.in0.C, 0, P.V2_CP[key])
                d.w2 = ctx.up(P.pack_a4(phase1["wk2"], rows1)).data_ptr()
                d.w3 = ctx.up(P.pack_a4(phase1["wk3"], rows1)).data_ptr()
                d.ksteps1, d.p1mask, d.Fout1 = len(rows1) // 2, phase1["mask"], phase1["Fout1"]
        else:
            w0 = P.pack_a(wk0)
            d.w0, d.ksteps, d.Cout = ctx.up(w0).data_ptr(), w0.shape[1], Cout
            if wk1 is not None:
                d.w1 = ctx.up(P.pack_a(wk1)).data_ptr()

        def dev(x):
            if x is None:
                return None
            return x if torch.is_tensor(x) else ctx.up(x)

        d.bias0, d.bias0_sb = Ctx.ptr(dev(bias0), bias0_off), bias0_sb
        d.bias1, d.bias1_sb = Ctx.ptr(dev(bias1), bias1_off), bias1_sb
        if bias_t0 is not None:                               # (tensor, offset of bias0_t0, offset of bias1_t0)
            d.bias0_t0, d.bias1_t0 = Ctx.ptr(bias_t0[0], bias_t0[1]), Ctx.ptr(bias_t0[0], bias_t0[2])
        d.epi, d.act, d.act_slope = epi, act, float(act_slope)
        if post is not None:
            d.post_scale, d.post_shift = ctx.up(post[0]).data_ptr(), ctx.up(post[1]).data_ptr()
        if chain is not None:
            d.C2 = chain["C2"]
            d.wlc, d.wrc = ctx.up(P.pack_chain(chain["wlc"])).data_ptr(), ctx.up(P.pack_chain(chain["wrc"])).data_ptr()
            d.blc, d.brc = ctx.up(chain["blc"]).data_ptr(), ctx.up(chain["brc"]).data_ptr()
            if chain["C2"] == 1:
                d.wc2 = ctx.up(np.asarray(chain["wc2"], np.float64).reshape(32)).data_ptr()
            else:
                d.wc2 = ctx.up(P.pack_chain(chain["wc2"])).data_ptr()
            d.bc2 = ctx.up(chain["bc2"]).data_ptr()
        if nx is not None:
            tiles = nx["tiles"]
            d.nx_n, d.nx_keep, d.nx_row0 = len(tiles), 1 if nx.get("keep") else 0, nx.get("row0", -1)
            packs = []
            for i, tl in enumerate(tiles):
                w = np.asarray(tl["w"], np.float64)                       # [32 out, 64 in]
                packs.append(np.concatenate([P.pack_chain(w[:, :32]), P.pack_chain(w[:, 32:])], 0))   # [2,16,64]
                d.nx_bias[i] = Ctx.ptr(dev(tl["bias"]), tl.get("bias_off", 0))
                d.nx_bias_sb[i] = tl.get("bias_sb", 0)
                d.nx_out[i] = Ctx.ptr(tl["out"])
                d.nx_sb[i], d.nx_sc[i], d.nx_st[i], d.nx_sf[i] = tl["strides"]
                d.nx_off[i] = tl.get("off", 0)
                d.nx_add[i] = Ctx.ptr(tl.get("add"))
            d.nx_w = ctx.up(np.stack(packs, 0)).data_ptr()
        d.resid = Ctx.ptr(resid)
        d.out = Ctx.ptr(out)
        d.out_sb, d.out_sc_hi, d.out_sc_lo, d.out_st, d.out_sf = out_strides
        d.out_off, d.out_cr = out_off, out_cr
        d.B, d.Tout, d.Fout = B, Tout, Fout
        self.add(d, tag)
        return d


def nchw(C_, T, F):
    """(sb, sc, st, sf) of a contiguous [B, C, T, F] tensor."""
    return C_ * T * F, T * F, F, 1


def nchw_out(C_, T, F):
    """(out_sb, out_sc_hi, out_sc_lo, out_st, out_sf) of a contiguous [B, C, T, F] output."""
    return C_ * T * F, T * F, 0, F, 1


# ==========================================================================
# DiffUNet1 / DiffUNet
# ==========================================================================
class EpsNetPlan(PlanBase):
    """ε-network (``time_cond=True``) or prior DiffUNet (``time_cond=False``).

    Buffers: x, x_init (inputs, [B,2,T,161]); t ([nsteps][B] steps); eps / out ([B,2,T,161]).
    ``build_step(i)`` appends the operators of one forward at diffusion step row i.
    """

    ENC_F = [161, 79, 39, 19, 9, 4]
    fused_tcm = True        # one launch per TCM residual block (csrc/tcm.hip); False: three gather-GEMM launches
    chain_conv1 = True      # every stage's 1x1 input convolution rides on the previous stage's tail (pdse.h: nx_*)
    compose_stage1 = True   # (with chain_conv1) encoder stage 1: conv1 composed into the gather weights
    NSLOT = 20  # 15 stages + en1 real-row bias + 4 composed encoder-stage-1 biases (l/r x frame >= 1 / frame 0)

    def __init__(self, ctx, sd, B, T, time_cond=True, nsteps=1, plan=None, table=None, with_pre=None):
        """with_pre False + time_cond True: ``Nocon`` (model/piror_grad.py), DiffUNet1 without Preprocess."""
        super().__init__(ctx, plan)
        self.sd, self.B, self.T, self.time_cond, self.nsteps = sd, B, T, time_cond, nsteps
        self.with_pre = time_cond if with_pre is None else with_pre
        a = ctx.alloc
        self.x = a(B, 2, T, F0)
        self.x_init = a(B, 2, T, F0) if self.with_pre else None
        self.out = a(B, 2, T, F0)
        self.H = a(B, 32, T + 1, F0)                 # conv1 output of the current block (+ explicit pad frame)
        self.H2 = a(B, 32, T + 1, F0)                # ... of the next block, written by the current block's tail (chain_conv1)
        # skip halves of the decoders' conv1 (+ time bias), produced by the encoder tails: [real|imag][stage 1..4]
        self.Pskip = [[None] + [a(B, 32, T, self.ENC_F[k]) for k in range(1, 5)] for _ in range(2)]
        self.zero32 = a(32, zero=True)
        self.en = [a(B, 64, T, f) for f in self.ENC_F[1:5]] + [a(B, 64, 4, T)]  # en5 stored [B,64,4,T]
        self.tcm_a, self.tcm_b = a(B, 256, T), a(B, 256, T)
        self.tcm_h, self.tcm_g = a(B, 64, T), a(B, 64, T)
        self.dec = [a(B, 64, T, 79), a(B, 64, T, 79)]  # ping-pong decoder activations (largest F=79)
        if time_cond:
            self.tsteps = a(nsteps, B, zero=True)
            self.tbias = a(nsteps, B, self.NSLOT * 32)
            self.temb = a(nsteps, B, 512)
            self._prep_time(table)

    # ---- weights helpers ------------------------------------------------
    def w(self, k):
        return P._np(self.sd[k])

    def _stage_fold(self):
        """Per stage k: (W1 [32,Cin], b1 [32], Wtp [Cin,512], btp [Cin])."""
        st = []
        for k in range(1, 6):
            W1 = self.w("en.conv%d.conv1.weight" % k)[:, :, 0, 0]
            st.append((W1, self.w("en.conv%d.conv1.bias" % k), "en.tp%d" % k))
        for de in ("de_real", "de_imag"):
            for k in (5, 4, 3, 2, 1):
                p = "%s.de%d.0" % (de, k)
                W1 = self.w(p + ".conv1.weight")[:, :, 0, 0].T
                st.append((W1, self.w(p + ".conv1.bias"), p + ".tp"))
        return st

    def _prep_time(self, table):
        ctx = self.ctx
        if table is None:
            # non-persistent buffer of the reference, rebuilt exactly as model/diff3.py:89-93 does (fp32 torch ops)
            steps = torch.arange(50).unsqueeze(1)
            dims = torch.arange(64).unsqueeze(0)
            tb = steps * 10.0 ** (dims * 4.0 / 63.0)
            table = torch.cat([torch.sin(tb), torch.cos(tb)], dim=1)
        self.table = ctx.up(table.numpy())
        wf, bf = [], []
        for W1, b1, tp in self._stage_fold():
            Wtp, btp = self.w(tp + ".weight"), self.w(tp + ".bias")
            wf.append(W1 @ Wtp)
            bf.append(b1 + W1 @ btp)
        # slot 15: en1 real rows also carry W1 · b_preprocess (the pad row does not, diff3.py:145-147)
        W1 = self.w("en.conv1.conv1.weight")[:, :, 0, 0]
        wf.append(wf[0])
        bf.append(bf[0] + (W1 @ self.w("preprocess.conv.bias") if self.with_pre else 0.0))
        # slots 16..19: encoder stage 1 with conv1 composed into the two gather convolutions (l, r): their biases become
        #   frame >= 1:  (S0 + S1) b_real + b        frame 0:  S0 b_pad + S1 b_real + b,    S_kt = sum_kf W[:, :, kt, kf]
        # (kt = 0 reads frame t-1; for t = 0 that is the zero pad frame, whose conv1 value is the pad bias, slot 0)
        for br in ("l", "r"):
            Wg = self.w("en.conv1.%s.weight" % br)                                  # [32, 32, 2, 5]
            S0, S1 = Wg[:, :, 0].sum(-1), Wg[:, :, 1].sum(-1)
            bg = self.w("en.conv1.%s.bias" % br)
            wf += [(S0 + S1) @ wf[0], (S0 + S1) @ wf[0]]
            bf += [(S0 + S1) @ bf[15] + bg, S0 @ bf[0] + S1 @ bf[15] + bg]
        WF, BF = np.concatenate(wf, 0), np.concatenate(bf, 0)      # [NSLOT*32, 512], [NSLOT*32]
        self.t_p1T = ctx.up(self.w("time_embedding.projection1.weight").T)
        self.t_b1 = ctx.up(self.w("time_embedding.projection1.bias"))
        self.t_p2T = ctx.up(self.w("time_embedding.projection2.weight").T)
        self.t_b2 = ctx.up(self.w("time_embedding.projection2.bias"))
        self.t_wfT, self.t_bf = ctx.up(WF.T), ctx.up(BF)

    def build_time(self):
        """One launch computes the folded biases of all recorded diffusion steps."""
        d = L.TimeDesc()
        d.t, d.table = self.tsteps.data_ptr(), self.table.data_ptr()
        d.p1T, d.b1, d.p2T, d.b2 = (self.t_p1T.data_ptr(), self.t_b1.data_ptr(), self.t_p2T.data_ptr(),
                                    self.t_b2.data_ptr())
        d.wfT, d.bf = self.t_wfT.data_ptr(), self.t_bf.data_ptr()
        d.out, d.temb = self.tbias.data_ptr(), self.temb.data_ptr()
        d.B, d.NF, d.max_steps = self.nsteps * self.B, self.NSLOT * 32, 50
        self.add(d, TAG_EW)

    # ---- blocks -----------------------------------------------------------
    def _bias_for(self, step, slot):
        """(tensor-or-array, element offset, batch stride) of a stage's conv1 bias."""
        return self.tbias, (step * self.B) * self.NSLOT * 32 + slot * 32, self.NSLOT * 32

    def _biconvglu_composed(self, step, src_x, src_init, out_t, out_strides, nx):
        """Encoder stage 1 with its 1x1 conv1 (and the folded Preprocess) composed into the two (2,5) gather
        convolutions: conv1 -> l / r is linear (model/diff3.py:316-319), so l(conv1(u)) is ONE (2,5) convolution over
        the 4 input channels (x, x_init) with W'[o,i] = sum_c Wl[o,c] W1[c,i] - 40 instead of 320 k-rows, no conv1
        launch, no 32-channel H tensor.  conv1's bias (time-conditioned, different for the zero pad frame) moves into
        per-batch-item biases of l / r with a separate value for output frame 0 (slots 16..19 of the folded table)."""
        B, T = self.B, self.T
        p = "en.conv1"
        Fin, Fout = F0, (F0 - 5) // 2 + 1
        W1 = self.w(p + ".conv1.weight")[:, :, 0, 0]                       # [32, Cin]
        if self.with_pre:
            W1 = W1 @ self.w("preprocess.conv.weight")[:, :, 0, 0]          # [32, 4] over (x, x_init)
        kk, taps = P.conv_taps(2, 5, 1)                                     # weight row kt reads frame t + kt - 1
        comp = {br: np.einsum("ockf,ci->oikf", self.w("%s.%s.weight" % (p, br)).astype(np.float64), W1.astype(np.float64))
                for br in ("l", "r")}
        if self.time_cond:
            tb, o_l, sbb = self._bias_for(step, 16)
            _, o_l0, _ = self._bias_for(step, 17)
            _, o_r, _ = self._bias_for(step, 18)
            _, o_r0, _ = self._bias_for(step, 19)
            bias = dict(bias0=tb, bias0_off=o_l, bias0_sb=sbb, bias1=tb, bias1_off=o_r, bias1_sb=sbb, bias_t0=(tb, o_l0, o_r0))
        else:
            b1 = self.w(p + ".conv1.bias")                                  # pad frame = conv1(0) = b1 as well
            bias = {}
            for key, br in (("bias0", "l"), ("bias1", "r")):
                Wg = self.w("%s.%s.weight" % (p, br))
                bias[key] = Wg.sum((2, 3)) @ b1 + self.w("%s.%s.bias" % (p, br))
        post = P.bn_fold(self.sd, "en.en1.0")
        chain = dict(C2=64, wlc=self.w(p + ".l_conv.weight")[:, :, 0, 0], blc=self.w(p + ".l_conv.bias"),
                     wrc=self.w(p + ".r_conv.weight")[:, :, 0, 0], brc=self.w(p + ".r_conv.bias"),
                     wc2=self.w(p + ".conv2.weight")[:, :, 0, 0], bc2=self.w(p + ".conv2.bias"))
        self.gconv(in0=src_x, in1=src_init, Tin=T, Fin=Fin, taps=taps, sf_in=2, wk0=P.conv_kmat(comp["l"], kk),
                   wk1=P.conv_kmat(comp["r"], kk), Cout=32, epi=L.EPI_BIGLU, act=L.ACT_PRELU,
                   act_slope=float(self.w("en.en1.1.weight")[0]), post=post, chain=chain, out=out_t, out_strides=out_strides,
                   B=B, Tout=T, Fout=Fout, tag=TAG_EPS_BLOCK, nx=nx, **bias)
        return Fout

    def _biconvglu(self, step, k, src_x, src_init, Fin, out_t, out_strides, H=None, do_conv1=True, nx=None):
        """Encoder stage k (model/diff3.py:144-166 + :307-326 + BN + PReLU).  do_conv1 False: H already holds this
        stage's conv1 output (chained onto the previous stage's tail); nx: 1x1 tiles chained onto this stage's tail."""
        B, T = self.B, self.T
        H = self.H if H is None else H
        p = "en.conv%d" % k
        kw = 5 if k == 1 else 3
        Fout = (Fin - kw) // 2 + 1
        W1 = self.w(p + ".conv1.weight")[:, :, 0, 0]                       # [32, Cin]
        if k == 1 and self.with_pre:
            Wp = self.w("preprocess.conv.weight")[:, :, 0, 0]               # [2, 4]: fold Preprocess into conv1
            W1 = W1 @ Wp
        if self.time_cond:
            tb, off_real, sbb = self._bias_for(step, 15 if k == 1 else k - 1)
            _, off_pad, _ = self._bias_for(step, k - 1)
            bias0, pad = tb, tb
        else:
            b1 = self.ctx.up(self.w(p + ".conv1.bias"))
            bias0, pad, off_real, off_pad, sbb = b1, b1, 0, 0, 0
        # H = conv1 output with an explicit frame -1 in row 0: the reference pads the input with one
        # zero frame on top and THEN adds the time bias (diff3.py:146-147), so that frame is
        # conv1(0 + tp) = the folded bias.  Rows 1..T hold the real frames.
        HT = T + 1
        h_out = (32 * HT * Fin, HT * Fin, 0, Fin, 1)
        if not do_conv1:
            pass
        elif k == 1 and self.time_cond:
            # real frames carry W1*b_preprocess in their bias (slot 15), the pad frame does not (slot 0)
            self.gconv(in0=src_x, in1=src_init, Tin=T, Fin=Fin, taps=[(0, 0)], sf_in=1, wk0=W1.T, Cout=32,
                       bias0=bias0, bias0_off=off_real, bias0_sb=sbb, out=H, out_strides=h_out, out_off=Fin,
                       B=B, Tout=T, Fout=Fin, tag=TAG_EPS_CONV1)
            self.gconv(in0=src_x, in1=src_init, Tin=T, Fin=Fin, taps=[(-(T + 1), 0)], sf_in=1, wk0=W1.T, Cout=32,
                       bias0=pad, bias0_off=off_pad, bias0_sb=sbb, out=H, out_strides=h_out,
                       B=B, Tout=1, Fout=Fin, tag=TAG_EPS_CONV1)
        else:
            # same bias for every frame: one launch over T+1 output frames reading input frame r-1
            self.gconv(in0=src_x, in1=src_init, Tin=T, Fin=Fin, taps=[(-1, 0)], sf_in=1, wk0=W1.T, Cout=32,
                       bias0=bias0, bias0_off=off_real, bias0_sb=sbb, out=H, out_strides=h_out,
                       B=B, Tout=HT, Fout=Fin, tag=TAG_EPS_CONV1)
        kk, taps = P.conv_taps(2, kw, 0)            # H row r = frame r-1: weight row kt reads H row t + kt
        post = P.bn_fold(self.sd, "en.en%d.0" % k)
        chain = dict(C2=64, wlc=self.w(p + ".l_conv.weight")[:, :, 0, 0], blc=self.w(p + ".l_conv.bias"),
                     wrc=self.w(p + ".r_conv.weight")[:, :, 0, 0], brc=self.w(p + ".r_conv.bias"),
                     wc2=self.w(p + ".conv2.weight")[:, :, 0, 0], bc2=self.w(p + ".conv2.bias"))
        self.gconv(in0=self.src(H, 32, *nchw(32, HT, Fin)), Tin=HT, Fin=Fin, taps=taps, sf_in=2,
                   wk0=P.conv_kmat(self.sd[p + ".l.weight"], kk), wk1=P.conv_kmat(self.sd[p + ".r.weight"], kk),
                   Cout=32, bias0=self.w(p + ".l.bias"), bias1=self.w(p + ".r.bias"), epi=L.EPI_BIGLU,
                   act=L.ACT_PRELU, act_slope=float(self.w("en.en%d.1.weight" % k)[0]), post=post,
                   chain=chain, out=out_t, out_strides=out_strides, B=B, Tout=T, Fout=Fout, tag=TAG_EPS_BLOCK, nx=nx)
        return Fout

    def _biconvtransglu(self, step, slot, p, k, in0, in1, Fin, out_t, out_strides_fn, bn_prefix, prelu_key, out_off=0,
                        H=None, do_conv1=True, nx=None):
        """Decoder stage (model/diff3.py:205-212 + :329-351, last frame chomped, BN, PReLU).  do_conv1 / nx as in
        ``_biconvglu``."""
        B, T = self.B, self.T
        H = self.H if H is None else H
        kw = 5 if k == 1 else 3
        Fout = 2 * (Fin - 1) + kw
        W1 = self.w(p + ".conv1.weight")[:, :, 0, 0].T                      # [32, 128]
        if self.time_cond:
            tb, off, sbb = self._bias_for(step, slot)
            bias0 = tb
        else:
            bias0, off, sbb = self.w(p + ".conv1.bias"), 0, 0
        if do_conv1:
            self.gconv(in0=in0, in1=in1, Tin=T, Fin=Fin, taps=[(0, 0)], sf_in=1, wk0=W1.T, Cout=32, bias0=bias0,
                       bias0_off=off, bias0_sb=sbb, out=H, out_strides=nchw_out(32, T, Fin), B=B, Tout=T,
                       Fout=Fin, tag=TAG_EPS_CONV1)
        C2 = 64 if k > 1 else 1
        post = P.bn_fold(self.sd, bn_prefix) if bn_prefix else None
        wc2 = self.w(p + ".conv2.weight")[:, :, 0, 0].T                     # [C2, 32]
        chain = dict(C2=C2, wlc=self.w(p + ".l_conv.weight")[:, :, 0, 0].T, blc=self.w(p + ".l_conv.bias"),
                     wrc=self.w(p + ".r_conv.weight")[:, :, 0, 0].T, brc=self.w(p + ".r_conv.bias"),
                     wc2=wc2, bc2=self.w(p + ".conv2.bias"))
        osb, osc, _, ost, osf = out_strides_fn(Fout)
        common = dict(Cout=32, bias0=self.w(p + ".l.bias"), bias1=self.w(p + ".r.bias"), epi=L.EPI_BIGLU,
                      act=L.ACT_PRELU if prelu_key else L.ACT_NONE,
                      act_slope=float(self.w(prelu_key)[0]) if prelu_key else 0.0, post=post, chain=chain, out=out_t,
                      out_strides=(osb, osc, 0, ost, 2 * osf), B=B, Tout=T, tag=TAG_EPS_BLOCK)
        src_h = self.src(H, 32, *nchw(32, T, Fin))
        kk0, taps0 = P.convT_phase_taps(2, kw, 0)
        kk1, taps1 = P.convT_phase_taps(2, kw, 1)
        if not self.force_generic:
            # both output phases in one launch: the odd bins read a subset of the even bins' taps, so the
            # activation loads are shared and every wave stores neighbouring (2j, 2j+1) bins together
            mask = sum(1 << taps0.index(tp) for tp in taps1)
            ph1 = dict(wk2=P.convT_kmat(self.sd[p + ".l.weight"], kk1), wk3=P.convT_kmat(self.sd[p + ".r.weight"], kk1),
                       mask=mask, ntaps1=len(taps1), Fout1=Fout // 2)
            self.gconv(in0=src_h, Tin=T, Fin=Fin, taps=taps0, sf_in=1, wk0=P.convT_kmat(self.sd[p + ".l.weight"], kk0),
                       wk1=P.convT_kmat(self.sd[p + ".r.weight"], kk0), out_off=out_off, Fout=(Fout + 1) // 2,
                       phase1=ph1, nx=nx, **common)
        else:
            for phase, (kk, taps) in enumerate(((kk0, taps0), (kk1, taps1))):
                self.gconv(in0=src_h, Tin=T, Fin=Fin, taps=taps, sf_in=1, wk0=P.convT_kmat(self.sd[p + ".l.weight"], kk),
                           wk1=P.convT_kmat(self.sd[p + ".r.weight"], kk), out_off=phase * osf + out_off,
                           Fout=(Fout - phase + 1) // 2, **common)
        return Fout

    def _tcm_conv1(self, p, xin, hout):
        """The 1x1 input convolution of a TCM residual block (diff3.py:223) as its own launch."""
        B, T = self.B, self.T
        self.gconv(in0=self.src(xin, 256, 256 * T, T, 0, 1), Tin=1, Fin=T, taps=[(0, 0)], sf_in=1,
                   wk0=self.w(p + ".conv1.weight")[:, :, 0].T, Cout=64, bias0=self.w(p + ".conv1.bias"),
                   out=hout, out_strides=(64 * T, T, 0, 0, 1), B=B, Tout=1, Fout=T, tag=TAG_TCM)

    def _tcm_branch_mats(self, p):
        def km(br):
            w = self.w(p + "." + br + ".2.weight")                          # [64, 64, 5]
            return np.concatenate([w[:, :, k].T for k in range(5)], axis=0)
        return km("mainbranch"), km("maskbranch")

    def _residual(self, p, dil, xin, xout):
        """TCM residual block (model/diff3.py:215-257) over [B,256,T] (T on the lanes): three launches."""
        B, T = self.B, self.T
        s64 = (64 * T, T, 0, 1)
        o256 = (256 * T, T, 0, 0, 1)    # out_sb, sc_hi, sc_lo, st, sf
        o64 = (64 * T, T, 0, 0, 1)
        self._tcm_conv1(p, xin, self.tcm_h)
        taps = [(0, (k - 2) * dil) for k in range(5)]
        kmain, kmask = self._tcm_branch_mats(p)
        sm, hm = P.bn_fold(self.sd, p + ".mainbranch.1")
        sk, hk = P.bn_fold(self.sd, p + ".maskbranch.1")
        xf = dict(mode=2, scale0=sm, shift0=hm, slope0=self.w(p + ".mainbranch.0.weight")[0],
                  scale1=sk, shift1=hk, slope1=self.w(p + ".maskbranch.0.weight")[0])
        self.gconv(in0=self.src(self.tcm_h, 64, *s64), Tin=1, Fin=T, taps=taps, sf_in=1, wk0=kmain,
                   wk1=kmask, Cout=64, bias0=self.w(p + ".mainbranch.2.bias"),
                   bias1=self.w(p + ".maskbranch.2.bias"), epi=L.EPI_GLU, xf=xf, out=self.tcm_g,
                   out_strides=o64, B=B, Tout=1, Fout=T, tag=TAG_TCM)
        s2, h2 = P.bn_fold(self.sd, p + ".conv2.1")
        xf2 = dict(mode=1, scale0=s2, shift0=h2, slope0=self.w(p + ".conv2.0.weight")[0])
        self.gconv(in0=self.src(self.tcm_g, 64, *s64), Tin=1, Fin=T, taps=[(0, 0)], sf_in=1,
                   wk0=self.w(p + ".conv2.2.weight")[:, :, 0].T, Cout=256, bias0=self.w(p + ".conv2.2.bias"),
                   xf=xf2, resid=xin, out=xout, out_strides=o256, B=B, Tout=1, Fout=T, tag=TAG_TCM)

    def _residual_fused(self, p, dil, xin, xout, hin, hout, p_next):
        """The same block as ONE launch (csrc/tcm.hip): dilated branches + gate + conv2 + residual, and the next
        block's conv1 (``p_next``; None for the last block) chained onto the fresh x in registers."""
        up = self.ctx.up
        kmain, kmask = self._tcm_branch_mats(p)
        sm, hm = P.bn_fold(self.sd, p + ".mainbranch.1")
        sk, hk = P.bn_fold(self.sd, p + ".maskbranch.1")
        s2, h2 = P.bn_fold(self.sd, p + ".conv2.1")
        d = L.TcmDesc()
        d.x, d.h, d.x_out = xin.data_ptr(), hin.data_ptr(), xout.data_ptr()
        d.wbr = up(P.pack_tcm_branch(kmain, kmask)).data_ptr()
        d.bmain = up(self.w(p + ".mainbranch.2.bias")).data_ptr()
        d.bmask = up(self.w(p + ".maskbranch.2.bias")).data_ptr()
        d.xf = up(np.stack([np.stack([sm, hm], 1), np.stack([sk, hk], 1)], 0).astype(np.float32)).data_ptr()
        d.wc2 = up(P.pack_tcm_conv2(self.w(p + ".conv2.2.weight")[:, :, 0].T)).data_ptr()
        d.bc2 = up(self.w(p + ".conv2.2.bias")).data_ptr()
        d.xf2 = up(np.stack([s2, h2], 1).astype(np.float32)).data_ptr()
        if p_next is not None:
            d.h_out = hout.data_ptr()
            d.wn1 = up(P.pack_tcm_next(self.w(p_next + ".conv1.weight")[:, :, 0])).data_ptr()
            d.bn1 = up(self.w(p_next + ".conv1.bias")).data_ptr()
        d.slope_main = float(self.w(p + ".mainbranch.0.weight")[0])
        d.slope_mask = float(self.w(p + ".maskbranch.0.weight")[0])
        d.slope2 = float(self.w(p + ".conv2.0.weight")[0])
        d.dil, d.B, d.T = dil, self.B, self.T
        self.add(d, TAG_TCM)

    def build_step(self, step=0, x=None, x_init=None, out=None):
        """Append one forward.  x / x_init / out default to the plan's own buffers.

        The first call packs and uploads the weights; later calls with the same buffers
        re-use those descriptors and only re-point the per-step time biases."""
        B, T = self.B, self.T
        x = self.x if x is None else x
        out = self.out if out is None else out
        x_init = self.x_init if x_init is None else x_init
        key = (x.data_ptr(), 0 if x_init is None else x_init.data_ptr(), out.data_ptr())
        tmpl = getattr(self, "_tmpl", None)
        if tmpl is not None and tmpl[0] == key and self.time_cond:
            _, step0, items = tmpl
            lo = self.tbias.data_ptr()
            hi = lo + self.tbias.numel() * 4
            delta = (step - step0) * B * self.NSLOT * 32 * 4
            for d, tag in items:
                d2 = type(d).from_buffer_copy(d)
                if isinstance(d2, L.GconvDesc):
                    for f in ("bias0", "bias1", "bias0_t0", "bias1_t0", "padrow"):
                        v = getattr(d2, f)
                        if v and lo <= v < hi:
                            setattr(d2, f, v + delta)
                    for i in range(d2.nx_n):
                        v = d2.nx_bias[i]
                        if v and lo <= v < hi:
                            d2.nx_bias[i] = v + delta
                self.add(d2, tag)
            return out
        begin = len(self.descs)
        self._build_step(step, x, x_init, out)
        self._tmpl = (key, step, list(self.descs[begin:]))
        return out

    def _build_step(self, step, x, x_init, out):
        B, T = self.B, self.T
        s2 = nchw(2, T, F0)
        chained = self.chain_conv1 and not self.force_generic
        # encoder: stage 1 reads (x, x_init) through the folded Preprocess 1x1
        src_x = self.src(x, 2, *s2)
        src_i = self.src(x_init, 2, *s2) if self.with_pre else None
        Fin = F0
        Hc, Hn = self.H, self.H2
        for k in range(1, 6):
            if k < 5:
                o, ostr = self.en[k - 1], nchw_out(64, T, self.ENC_F[k])
            else:
                o, ostr = self.en[4], (64 * 4 * T, 4 * T, 0, 1, T)           # [B,64,4,T]: channel c*4+f of [B,256,T]
            nx = None
            if chained and k < 5:
                # this stage's 64-channel output is only ever read by 1x1 convolutions - the next stage's conv1 and the
                # skip half of the two decoders' conv1 (diff3.py:343: conv1(cat(x, skip) + tp), split by linearity) -
                # so its tail evaluates those in registers and the output itself is never written
                Fo, HT = self.ENC_F[k], T + 1
                bias_of = (lambda slot: dict(zip(("bias", "bias_off", "bias_sb"), self._bias_for(step, slot)))) \
                    if self.time_cond else None
                pn = "en.conv%d" % (k + 1)
                t0 = dict(w=self.w(pn + ".conv1.weight")[:, :, 0, 0], out=Hn, strides=(32 * HT * Fo, HT * Fo, Fo, 1), off=Fo)
                t0.update(bias_of(k) if self.time_cond else dict(bias=self.w(pn + ".conv1.bias")))
                tiles = [t0]
                for di, de in enumerate(("de_real", "de_imag")):
                    pd = "%s.de%d.0" % (de, k)
                    tl = dict(w=self.w(pd + ".conv1.weight")[:, :, 0, 0].T[:, 64:], out=self.Pskip[di][k],   # ConvTranspose: [in, out]
                              strides=(32 * T * Fo, T * Fo, Fo, 1))
                    tl.update(bias_of(5 + 5 * di + (5 - k)) if self.time_cond else dict(bias=self.w(pd + ".conv1.bias")))
                    tiles.append(tl)
                nx = dict(keep=False, row0=0, tiles=tiles)
            if chained and k == 1 and self.compose_stage1:
                Fin = self._biconvglu_composed(step, src_x, src_i, o, ostr, nx)
            else:
                Fin = self._biconvglu(step, k, src_x, src_i, Fin, o, ostr, H=Hc, do_conv1=(k == 1 or not chained), nx=nx)
            if chained:
                Hc, Hn = Hn, Hc
            src_x, src_i = self.src(o, 64, *nchw(64, T, Fin)), None
        # TCMs over [B,256,T]
        cur, nxt = self.en[4], self.tcm_a
        names = [("TCMs.%d.residual%d" % (i, j + 1), dil) for i in range(3) for j, dil in enumerate((1, 2, 4, 8, 16, 32))]
        if self.force_generic or not self.fused_tcm:
            for p, dil in names:
                self._residual(p, dil, cur, nxt)
                cur, nxt = nxt, (self.tcm_b if nxt is self.tcm_a else self.tcm_a)
        else:
            # one launch per block; each also produces the next block's conv1 output (h ping-pongs: halo reads)
            hcur, hnxt = self.tcm_h, self.tcm_g
            self._tcm_conv1(names[0][0], cur, hcur)
            for n, (p, dil) in enumerate(names):
                p_next = names[n + 1][0] if n + 1 < len(names) else None
                self._residual_fused(p, dil, cur, nxt, hcur, hnxt, p_next)
                cur, nxt = nxt, (self.tcm_b if nxt is self.tcm_a else self.tcm_a)
                hcur, hnxt = hnxt, hcur
        tcm_out = cur
        # decoders
        for di, de in enumerate(("de_real", "de_imag")):
            in0 = self.src(tcm_out, 64, 256 * T, 4 * T, 1, T)               # [B,64,4,T] viewed as [B,64,T,4]
            Fin = 4
            Hc, Hn = self.H, self.H2
            for n, k in enumerate((5, 4, 3, 2, 1)):
                skip = self.en[k - 1]
                if k == 5:
                    in1 = self.src(skip, 64, 256 * T, 4 * T, 1, T)
                else:
                    in1 = self.src(skip, 64, *nchw(64, T, self.ENC_F[k]))
                p = "%s.de%d.0" % (de, k)
                if k > 1:
                    o, ooff = self.dec[n & 1], 0
                    fn = lambda Fo: nchw_out(64, T, Fo)  # noqa: E731
                    bn, pr = "%s.de%d.2" % (de, k), "%s.de%d.3.weight" % (de, k)
                else:
                    o, ooff = out, di * T * F0
                    fn = lambda Fo: nchw_out(2, T, Fo)   # noqa: E731
                    bn, pr = None, None
                nx = None
                if chained and k > 1:
                    # stage k-1's conv1 = W[:, :64] * (this stage's output) + the skip half the encoder left in Pskip
                    Fo = 2 * (Fin - 1) + 3
                    wn = self.w("%s.de%d.0.conv1.weight" % (de, k - 1))[:, :, 0, 0].T[:, :64]   # ConvTranspose: [in, out]
                    nx = dict(keep=False, row0=-1, tiles=[dict(w=wn, bias=self.zero32, out=Hn, add=self.Pskip[di][k - 1],
                                                                strides=(32 * T * Fo, T * Fo, Fo, 2))])
                Fin = self._biconvtransglu(step, 5 + 5 * di + n, p, k, in0, in1, Fin, o, fn, bn, pr, out_off=ooff, H=Hc,
                                           do_conv1=(k == 5 or not chained), nx=nx)
                if chained:
                    Hc, Hn = Hn, Hc
                in0 = self.src(o, 64, *nchw(64, T, Fin))
        return out


# ==========================================================================
# GCRN prior
# ==========================================================================
class GcrnPlan(PlanBase):
    fused_last = True       # last decoder stage + Linear(161,161) as one persistent launch (pdse_gcrnlast_desc)
    ENC_C = [2, 16, 32, 64, 128, 256]
    ENC_F = [161, 80, 39, 19, 9, 4]

    def __init__(self, ctx, sd, B, T, plan=None):
        super().__init__(ctx, plan)
        self.sd, self.B, self.T = sd, B, T
        a = ctx.alloc
        self.Bp = (B + 31) // 32 * 32
        self.x = a(B, 2, T, F0)
        self.out = a(B, 2, T, F0)
        self.e = [a(B, self.ENC_C[i + 1], T, self.ENC_F[i + 1]) for i in range(5)]
        self.gx = a(2, T, 2048, self.Bp, zero=True)
        self.hT = a(2, 2, 512, self.Bp, zero=True)
        self.cst = a(2, 512, self.Bp, zero=True)
        self.y = a(B, T, 1024)
        self.yn = a(B, 1024, T)
        self.glstm = a(B, 256, T, 4)
        self.d = [a(B, 128, T, 9), a(B, 64, T, 19), a(B, 32, T, 39), a(B, 16, T, 80), a(B, 1, T, 161)]

    def w(self, k):
        return P._np(self.sd[k])

    def _lstm_layer(self, layer, xproj_src_fn, y_su, y_sg):
        B, T, Bp = self.B, self.T, self.Bp
        whh = np.empty((2, 64, 256, 64), np.float32)
        for g in range(2):
            p = "glstm.%s.%d." % (layer, g)
            in0, Tin, Fin, taps, wk, Tout, Fout, ost, osf = xproj_src_fn(g, self.w(p + "weight_ih_l0"))
            self.gconv(in0=in0, Tin=Tin, Fin=Fin, taps=taps, sf_in=1, wk0=wk, Cout=2048,
                       bias0=self.w(p + "bias_ih_l0") + self.w(p + "bias_hh_l0"), out=self.gx,
                       out_strides=(1, Bp, 0, ost, osf), out_off=g * T * 2048 * Bp, B=B, Tout=Tout, Fout=Fout,
                       tag=TAG_PRIOR)
            W = self.w(p + "weight_hh_l0")                                   # [2048, 512], gate order i,f,g,o
            # slice s owns hidden units 8s..8s+7: tile row i = q*8 + u  <->  W row q*512 + 8s + u
            rows = (np.arange(4)[:, None] * 512 + np.arange(8)[None, :]).reshape(-1)   # [32]
            for s in range(64):
                whh[g, s] = P.pack_a(W[rows + 8 * s, :].T)[0]
        d = L.LstmDesc()
        d.gx, d.whh = self.gx.data_ptr(), self.ctx.up(whh).data_ptr()
        d.hT, d.cst, d.y = self.hT.data_ptr(), self.cst.data_ptr(), self.y.data_ptr()
        d.y_sb, d.y_st, d.y_su, d.y_sg = T * 1024, 1024, y_su, y_sg
        d.B, d.Bp, d.T, d.H, d.G = B, Bp, T, 512, 2
        self.add(d, TAG_LSTM)

    def _ln(self, name, out_t, osb, os_hi, os_lo, os_t, r):
        d = L.LnDesc()
        d.in_, d.out = self.y.data_ptr(), out_t.data_ptr()
        d.gamma, d.beta = self.ctx.up(self.w(name + ".weight")).data_ptr(), self.ctx.up(self.w(name + ".bias")).data_ptr()
        d.osb, d.os_hi, d.os_lo, d.os_t = osb, os_hi, os_lo, os_t
        d.B, d.T, d.N, d.r, d.eps = self.B, self.T, 1024, r, 1e-5
        self.add(d, TAG_PRIOR)

    def build(self, x=None, out=None):
        B, T, Bp = self.B, self.T, self.Bp
        x = self.x if x is None else x
        out = self.out if out is None else out
        # encoder: GluConv2d k(1,3) s(1,2) + BN + ELU (gcrn.py:138-142)
        src = self.src(x, 2, *nchw(2, T, F0))
        for k in range(1, 6):
            ci, co, Fin, Fout = self.ENC_C[k - 1], self.ENC_C[k], self.ENC_F[k - 1], self.ENC_F[k]
            kk, taps = P.conv_taps(1, 3, 0)
            p = "conv%d" % k
            self.gconv(in0=src, Tin=T, Fin=Fin, taps=taps, sf_in=2, wk0=P.conv_kmat(self.sd[p + ".conv1.weight"], kk),
                       wk1=P.conv_kmat(self.sd[p + ".conv2.weight"], kk), Cout=co, bias0=self.w(p + ".conv1.bias"),
                       bias1=self.w(p + ".conv2.bias"), epi=L.EPI_GLU, act=L.ACT_ELU, post=P.bn_fold(self.sd, "bn%d" % k),
                       out=self.e[k - 1], out_strides=nchw_out(co, T, Fout), B=B, Tout=T, Fout=Fout, tag=TAG_PRIOR)
            src = self.src(self.e[k - 1], co, *nchw(co, T, Fout))

        # grouped LSTM (gcrn.py:22-40)
        def proj1(g, Wih):
            # group g = channels 128g..128g+127 of e5 [B,256,T,4]; k = f*128 + c'  <->  W_ih column c'*4 + f
            wk = np.concatenate([Wih[:, f::4].T for f in range(4)], axis=0)
            s = self.src(self.e[4], 128, *nchw(256, T, 4), off=128 * g * T * 4)
            return s, T, 4, [(0, f) for f in range(4)], wk, T, 1, 2048 * Bp, 0

        self._lstm_layer("lstm_list1", proj1, y_su=2, y_sg=1)               # stack(dim=-1)+flatten: index u*2+g
        self._ln("glstm.ln1", self.yn, 1024 * T, T, 0, 1, 1)                 # -> [B,1024,T]

        def proj2(g, Wih):
            s = self.src(self.yn, 512, 1024 * T, T, 0, 1, off=512 * g * T)
            return s, 1, T, [(0, 0)], Wih.T, 1, T, 0, 2048 * Bp

        self._lstm_layer("lstm_list2", proj2, y_su=1, y_sg=512)              # cat: index g*512+u
        self._ln("glstm.ln2", self.glstm, 256 * T * 4, T * 4, 1, 4, 4)       # j = c*4+f -> [B,256,T,4]

        # two decoders (gcrn.py:150-164)
        dec = [(5, 512, 128), (4, 256, 64), (3, 128, 32), (2, 64, 16), (1, 32, 1)]
        for br in (1, 2):
            in0 = self.src(self.glstm, 256, *nchw(256, T, 4))
            in1 = self.src(self.e[4], 256, *nchw(256, T, 4))
            Fin = 4
            for n, (k, ci, co) in enumerate(dec):
                p = "conv%d_t_%d" % (k, br)
                Fout = 2 * (Fin - 1) + 3 + (1 if k == 2 else 0)
                if k == 1 and self.fused_last and not self.force_generic:
                    # last stage (one output channel) + Linear(161,161) in one persistent launch (csrc/misc.hip)
                    sc, sh = P.bn_fold(self.sd, "bn1_t_%d" % br)
                    g = L.GcrnLastDesc()
                    g.in0, g.in1 = self.d[3].data_ptr(), self.e[0].data_ptr()
                    up = self.ctx.up
                    g.w1 = up(self.w(p + ".conv1.weight")[:, 0, 0, :]).data_ptr()          # [32, 3]
                    g.w2 = up(self.w(p + ".conv2.weight")[:, 0, 0, :]).data_ptr()
                    g.fcT, g.fcb = up(self.w("fc%d.weight" % br).T).data_ptr(), up(self.w("fc%d.bias" % br)).data_ptr()
                    g.out, g.out_sb = Ctx.ptr(out, (br - 1) * T * F0), 2 * T * F0
                    g.b1, g.b2 = float(self.w(p + ".conv1.bias")[0]), float(self.w(p + ".conv2.bias")[0])
                    g.bn_scale, g.bn_shift, g.B, g.T = float(sc[0]), float(sh[0]), B, T
                    self.add(g, TAG_PRIOR)
                    break
                osb, osc, _, ost, osf = nchw_out(co, T, Fout)
                for phase in (0, 1):
                    kk, taps = P.convT_phase_taps(1, 3, phase)
                    self.gconv(in0=in0, in1=in1, Tin=T, Fin=Fin, taps=taps, sf_in=1,
                               wk0=P.convT_kmat(self.sd[p + ".conv1.weight"], kk),
                               wk1=P.convT_kmat(self.sd[p + ".conv2.weight"], kk), Cout=co,
                               bias0=self.w(p + ".conv1.bias"), bias1=self.w(p + ".conv2.bias"), epi=L.EPI_GLU,
                               act=L.ACT_ELU, post=P.bn_fold(self.sd, "bn%d_t_%d" % (k, br)), out=self.d[n],
                               out_strides=(osb, osc, 0, ost, 2 * osf), out_off=phase, B=B, Tout=T,
                               Fout=(Fout - phase + 1) // 2, tag=TAG_PRIOR)
                Fin = Fout
                if k > 1:
                    in0 = self.src(self.d[n], co, *nchw(co, T, Fout))
                    skip = self.e[k - 2]
                    in1 = self.src(skip, co, *nchw(co, T, Fout), act=L.ACT_ELU)   # elu(cat(.., skip)) re-applies ELU
            else:   # (no break: the unfused form) Linear(161,161) over the bins (gcrn.py:162-163): taps enumerate the input bin
                self.gconv(in0=self.src(self.d[4], 1, *nchw(1, T, F0)), Tin=T, Fin=F0, taps=[(0, f) for f in range(F0)],
                           sf_in=1, wk0=self.w("fc%d.weight" % br).T, Cout=F0, bias0=self.w("fc%d.bias" % br), cin1=True,
                           out=out, out_strides=(2 * T * F0, 1, 0, F0, 0), out_off=(br - 1) * T * F0, B=B, Tout=T, Fout=1,
                           tag=TAG_PRIOR)
        return out


# ==========================================================================
# signal front / back end
# ==========================================================================
class StftPlan(PlanBase):
    """wav [B,L] -> c [B], compressed spectrogram [B,2,T,161]
    (trainer/complex_ddpm_trainer.py:921-937)."""

    def __init__(self, ctx, B, L_, plan=None, normalize=True):
        super().__init__(ctx, plan)
        if L_ <= 160:
            raise ValueError(f"utterance of {L_} samples: the centred STFT reflects 160 samples on each side (torch.stft "
                             "raises for the same input)")
        self.B, self.L = B, L_
        self.T = 1 + L_ // 160
        self.wav = ctx.alloc(B, L_)
        self.xpad = ctx.alloc(B, L_ + 320)
        self.c = ctx.alloc(B)
        self.lens = torch.full((B,), L_, dtype=torch.int32, device=ctx.device)   # true lengths of zero-padded utterances
        ctx.keep.append(self.lens)
        self.feat = ctx.alloc(B, 2, self.T, F0)
        self.normalize = normalize

    def build(self, feat=None):
        B, L_, T = self.B, self.L, self.T
        feat = self.feat if feat is None else feat
        d = L.WavprepDesc()
        d.wav, d.xpad, d.c, d.lens = self.wav.data_ptr(), self.xpad.data_ptr(), self.c.data_ptr(), self.lens.data_ptr()
        d.B, d.L, d.pad, d.normalize = B, L_, 160, 1 if self.normalize else 0
        self.add(d, TAG_SIGNAL)
        Lp = L_ + 320
        self.gconv(in0=self.src(self.xpad, 1, Lp, 0, 0, 1), Tin=1, Fin=Lp, taps=[(0, n) for n in range(320)],
                   sf_in=160, wk0=P.stft_kmat(320), Cout=2 * F0, cin1=True, out=feat,
                   out_strides=(2 * T * F0, T * F0, 1, 0, F0), out_cr=F0, B=B, Tout=1, Fout=T, tag=TAG_SIGNAL)
        c = L.CompandDesc()
        c.in_, c.out, c.plane, c.B, c.mode = feat.data_ptr(), feat.data_ptr(), T * F0, B, 0
        self.add(c, TAG_SIGNAL)
        return feat


class IstftPlan(PlanBase):
    """compressed spectrogram [B,2,T,161] -> wav [B,L] * c
    (trainer/complex_ddpm_trainer.py:1004-1016)."""

    def __init__(self, ctx, B, T, L_, plan=None):
        super().__init__(ctx, plan)
        self.B, self.T, self.L = B, T, L_
        self.spec = ctx.alloc(B, 2, T, F0)
        self.dec = ctx.alloc(B, 2, T, F0)
        self.frames = ctx.alloc(B, 320, T)
        self.wav = ctx.alloc(B, L_)
        self.win2 = ctx.up(P.hann_periodic(320) ** 2)

    def build(self, spec=None, c=None):
        B, T = self.B, self.T
        spec = self.spec if spec is None else spec
        cd = L.CompandDesc()
        cd.in_, cd.out, cd.plane, cd.B, cd.mode = spec.data_ptr(), self.dec.data_ptr(), T * F0, B, 1
        self.add(cd, TAG_SIGNAL)
        self.gconv(in0=self.src(self.dec, 2, *nchw(2, T, F0)), Tin=T, Fin=F0, taps=[(0, f) for f in range(F0)],
                   sf_in=1, wk0=P.istft_kmat(320), Cout=320, out=self.frames, out_strides=(320 * T, T, 0, 1, 0),
                   B=B, Tout=T, Fout=1, tag=TAG_SIGNAL)
        o = L.OlaDesc()
        o.frames, o.win2, o.c, o.out = (self.frames.data_ptr(), self.win2.data_ptr(), Ctx.ptr(c), self.wav.data_ptr())
        o.B, o.T, o.L, o.n_fft, o.hop = B, T, self.L, 320, 160
        self.add(o, TAG_SIGNAL)
        return self.wav


# ==========================================================================
# DB-AIAT prior  aia_complex_trans_ri  (model/dbaiat.py:450-478)
# ==========================================================================
class AiaPlan(PlanBase):
    """dense_encoder -> AIA_Transformer (4 x {row over bins, col over frames}) -> AHAM ->
    two dense_decoders.  Convolutions / Linear layers run on the gather-GEMM kernels, the rest
    on the operators of csrc/aia.hip."""

    FH = 80  # bins after the stride-2 encoder conv
    fused_gru_input = True   # d_model 32: W_ih x inside the GRU kernel (csrc/aia.hip, gru_kernel<64, true>)

    def __init__(self, ctx, sd, B, T, plan=None, d=32):
        """d: d_model of the transformer layers (32: AIA_Transformer(64, 64); 64: AIA_Transformer_merge(128, 64))."""
        super().__init__(ctx, plan)
        self.sd, self.B, self.T, self.d = sd, B, T, d
        a = ctx.alloc
        FH = self.FH
        self.x = a(B, 2, T, F0)
        self.out = a(B, 2, T, F0)
        self.tmp161 = a(B, 64, T, F0)
        self.tmp80 = a(B, 64, T, FH)
        self.D161 = a(B, 320, T, F0)            # dense buffer [out4,out3,out2,out1,x]
        self.D80 = a(B, 320, T, FH)
        self.x_ri = a(B, 64, T, FH)
        self.cur = a(B, d, T, FH)                # AIA state ("output" in dbaiat.py:138)
        self.nxt = a(B, d, T, FH)
        self.n_a, self.n_b = a(B, d, T, FH), a(B, d, T, FH)
        self.qkv = a(B, 3 * d, T, FH)
        self.att = a(B, d, T, FH)
        self.gx = a(B, 12 * d, T, FH)            # both directions x (r, z, n) x hidden 2d
        self.gy = a(B, 4 * d, T, FH)
        self.s1, self.s2 = a(B, d, T, FH), a(B, d, T, FH)
        self.br = [a(B, d, T, FH), a(B, d, T, FH)]        # row / col branch outputs
        self.outs = [a(B, 64, T, FH) for _ in range(4)]
        self.merged = a(B, 64, T, FH)
        self.gn_stats = a(B, 64, 4)
        self.means = a(4, B, 64)
        self.dec_up = a(B, 64, T, F0, zero=True)  # sub-pixel output; bin 0 is the left zero pad, never written
        self.t_a, self.t_b, self.t_c = a(B, d, FH, T), a(B, d, FH, T), a(B, d, FH, T)   # "ft" staging (frames innermost)

    def w(self, k):
        return P._np(self.sd[k])

    # ---- small operators ---------------------------------------------------
    def _rowln(self, src, dst, dst_sb, C_, F_, norm, prelu, dst_off=0):
        d = L.RowlnDesc()
        d.in_, d.out = src.data_ptr(), Ctx.ptr(dst, dst_off)
        d.gamma, d.beta = self.ctx.up(self.w(norm + ".weight")).data_ptr(), self.ctx.up(self.w(norm + ".bias")).data_ptr()
        d.slope = self.ctx.up(self.w(prelu + ".weight")).data_ptr()
        d.out_sb, d.B, d.C, d.T, d.F, d.eps = dst_sb, self.B, C_, self.T, F_, 1e-5
        self.add(d, TAG_PRIOR)

    def _swap(self, src, dst, to_ft):
        """[B,d,T,F] <-> [B,d,F,T] (tile transpose, one read + one write of a d-channel tensor)."""
        d = L.TransposeDesc()
        d.in_, d.out, d.N = src.data_ptr(), dst.data_ptr(), self.B * self.d
        d.R, d.Cc = (self.T, self.FH) if to_ft else (self.FH, self.T)
        self.add(d, TAG_PRIOR)

    def _chln(self, src, dst, norm):
        d = L.ChlnDesc()
        d.in_, d.out = src.data_ptr(), dst.data_ptr()
        d.gamma, d.beta = self.ctx.up(self.w(norm + ".weight")).data_ptr(), self.ctx.up(self.w(norm + ".bias")).data_ptr()
        d.plane, d.B, d.C, d.eps = self.T * self.FH, self.B, self.d, 1e-5
        self.add(d, TAG_PRIOR)

    def _pw(self, src_t, Cin, wk, bias, out_t, Cout, F_, resid=None, act=L.ACT_NONE, act_slope=0.0, xf=None,
            in_layout="tf", out_layout="tf"):
        """1x1 convolution / Linear over the channel axis of [B,Cin,T,F_].

        Layouts: "tf" = [B,C,T,F] (bins innermost, the model's own), "ft" = [B,C,F,T] (frames
        innermost).  The row GRU and the column attention walk the axis that is strided in "tf";
        their operands are produced in "ft" so that the 32 lines of a workgroup sit next to each
        other in memory at every step.  A launch whose two sides differ iterates its lanes along t
        (coalesced on the "ft" side — the wide tensors — and scattered on the 32-channel "tf" side)."""
        T, B = self.T, self.B
        plane = T * F_
        if in_layout == "tf" and out_layout == "tf":
            self.gconv(in0=self.src(src_t, Cin, *nchw(Cin, T, F_)), Tin=T, Fin=F_, taps=[(0, 0)], sf_in=1, cin1=Cin == 1,
                       wk0=wk, Cout=Cout, bias0=bias, act=act, act_slope=act_slope, xf=xf, resid=resid, out=out_t,
                       out_strides=nchw_out(Cout, T, F_), B=B, Tout=T, Fout=F_, tag=TAG_PRIOR)
            return
        # lanes along t: the kernel's "frame" index is the bin f, its "bin" index is the frame t
        i_st, i_sf = (1, F_) if in_layout == "tf" else (T, 1)
        o_st, o_sf = (1, F_) if out_layout == "tf" else (T, 1)     # the residual is addressed like the output
        self.gconv(in0=self.src(src_t, Cin, Cin * plane, plane, i_st, i_sf), Tin=F_, Fin=T, taps=[(0, 0)], sf_in=1,
                   wk0=wk, Cout=Cout, bias0=bias, act=act, act_slope=act_slope, xf=xf, resid=resid, out=out_t,
                   out_strides=(Cout * plane, plane, 0, o_st, o_sf), B=B, Tout=F_, Fout=T, tag=TAG_PRIOR)

    def _dense_block(self, p, D, F_, tmp):
        """dbaiat.py:605-631.  D [B,320,T,F_] holds [out4,out3,out2,out1,x]; x is already in block 4."""
        B, T = self.B, self.T
        for i in range(1, 5):
            dil = 2 ** (i - 1)
            cin = 64 * i
            cstart = (5 - i) * 64                                    # skip = channels cstart..319
            kk = [(kt, kf) for kt in range(2) for kf in range(3)]
            taps = [((kt - 1) * dil, kf - 1) for kt, kf in kk]       # pad (1,1,dil,0): causal in time, same in bins
            self.gconv(in0=self.src(D, cin, *nchw(320, T, F_), off=cstart * T * F_), Tin=T, Fin=F_, taps=taps, sf_in=1,
                       wk0=P.conv_kmat(self.sd["%s.conv%d.weight" % (p, i)], kk), Cout=64,
                       bias0=self.w("%s.conv%d.bias" % (p, i)), out=tmp, out_strides=nchw_out(64, T, F_), B=B, Tout=T,
                       Fout=F_, tag=TAG_PRIOR)
            self._rowln(tmp, D, 320 * T * F_, 64, F_, "%s.norm%d" % (p, i), "%s.prelu%d" % (p, i),
                        dst_off=(4 - i) * 64 * T * F_)
        return 0  # out4 sits in channel block 0

    def _encoder_layer(self, p, axis, src_t, dst_t):
        """TransformerEncoderLayer (dbaiat.py:66-88) along bins (axis 0) or frames (axis 1).

        Attention over frames and the GRU over bins run on "ft" operands with the kernels'
        coalesced code path (attention axis 0 / GRU axis 1 on the swapped tensor)."""
        B, T, FH, dm = self.B, self.T, self.FH, self.d
        self._chln(src_t, self.n_a, p + ".norm3")
        Wi, bi = self.w(p + ".self_attn.in_proj_weight").copy(), self.w(p + ".self_attn.in_proj_bias").copy()
        Wi[:dm] *= (dm // 4) ** -0.5                                  # q scaled by head_dim^-0.5 inside the projection
        bi[:dm] *= (dm // 4) ** -0.5
        att_ft = axis == 1                                            # sequence over frames: operands in "ft"
        lay = "ft" if att_ft else "tf"
        if att_ft:
            # the whole attention sub-block runs frames-innermost: two tile transposes of 32-channel tensors
            # (in: normed input and the residual; out: the sum) instead of lane-scattered 1x1 launches
            self._swap(self.n_a, self.t_a, True)
            self._swap(src_t, self.t_b, True)
            qkv_in, res_in, s1_out = self.t_a, self.t_b, self.t_c
        else:
            qkv_in, res_in, s1_out = self.n_a, src_t, self.s1
        self._pw(qkv_in, dm, Wi.T, bi, self.qkv, 3 * dm, FH, in_layout=lay, out_layout=lay)
        d = L.AttnDesc()
        d.qkv, d.out, d.B, d.E, d.heads, d.axis = self.qkv.data_ptr(), self.att.data_ptr(), B, dm, 4, 0
        d.T, d.F = (FH, T) if att_ft else (T, FH)                     # "ft": the innermost axis is the sequence either way
        self.add(d, TAG_PRIOR)
        self._pw(self.att, dm, self.w(p + ".self_attn.out_proj.weight").T, self.w(p + ".self_attn.out_proj.bias"),
                 s1_out, dm, FH, resid=res_in, in_layout=lay, out_layout=lay)         # src + attention
        if att_ft:
            self._swap(self.t_c, self.s1, False)
        self._chln(self.s1, self.n_b, p + ".norm1")
        g = p + ".gru."
        Wih = np.concatenate([self.w(g + "weight_ih_l0"), self.w(g + "weight_ih_l0_reverse")], 0)   # [12d, d]
        bih = np.concatenate([self.w(g + "bias_ih_l0"), self.w(g + "bias_ih_l0_reverse")], 0)
        gru_ft = axis == 0                                            # sequence over bins: lines = frames -> "ft"
        lay = "ft" if gru_ft else "tf"
        if gru_ft:
            self._swap(self.n_b, self.t_a, True)
            gx_in, s2_out = self.t_a, self.t_b
        else:
            gx_in, s2_out = self.n_b, self.s2
        whh = np.stack([P.pack_a(self.w(g + "weight_hh_l0" + suf).T) for suf in ("", "_reverse")], 0)   # [2, 3H/32, H/2, 64]
        bhh = np.stack([self.w(g + "bias_hh_l0"), self.w(g + "bias_hh_l0_reverse")], 0)
        gd = L.GruDesc()
        if self.fused_gru_input and dm == 32:
            # the input projection runs inside the recurrence kernel: the 12x wider gx tensor never exists
            wih = np.stack([P.pack_a(self.w(g + "weight_ih_l0" + suf).T) for suf in ("", "_reverse")], 0)   # [2, 6, 16, 64]
            gd.x, gd.wih, gd.bih = gx_in.data_ptr(), self.ctx.up(wih).data_ptr(), self.ctx.up(bih.reshape(2, -1)).data_ptr()
        else:
            self._pw(gx_in, dm, Wih.T, bih, self.gx, 12 * dm, FH, in_layout=lay, out_layout=lay)
        gd.gx, gd.y = self.gx.data_ptr(), self.gy.data_ptr()
        gd.whh, gd.bhh = self.ctx.up(whh).data_ptr(), self.ctx.up(bhh).data_ptr()
        gd.B, gd.H, gd.axis = B, 2 * dm, 1                            # lines on the innermost axis, sequence on the outer
        gd.T, gd.F = (FH, T) if gru_ft else (T, FH)
        self.add(gd, TAG_LSTM)
        # relu -> linear2 -> + residual (the normed tensor); ReLU = the load transform with slope 0, identity affine
        relu = dict(mode=1, scale0=np.ones(4 * dm), shift0=np.zeros(4 * dm), slope0=0.0)
        self._pw(self.gy, 4 * dm, self.w(p + ".linear2.weight").T, self.w(p + ".linear2.bias"), s2_out, dm, FH,
                 resid=gx_in, xf=relu, in_layout=lay, out_layout=lay)
        if gru_ft:
            self._swap(self.t_b, self.s2, False)
        self._chln(self.s2, dst_t, p + ".norm2")

    def _dense_encoder(self, p, src_t, cin, dst):
        """dense_encoder / dense_encoder_mag (dbaiat.py:481-524): src [B,cin,T,161] -> dst [B,64,T,80]."""
        B, T, FH, sd = self.B, self.T, self.FH, self.sd
        self._pw(src_t, cin, self.w(p + ".inp_conv.weight")[:, :, 0, 0].T, self.w(p + ".inp_conv.bias"), self.tmp161, 64, F0)
        self._rowln(self.tmp161, self.D161, 320 * T * F0, 64, F0, p + ".inp_norm", p + ".inp_prelu", dst_off=256 * T * F0)
        self._dense_block(p + ".enc_dense1", self.D161, F0, self.tmp161)
        kk, taps = P.conv_taps(1, 3, 0)
        self.gconv(in0=self.src(self.D161, 64, *nchw(320, T, F0)), Tin=T, Fin=F0, taps=taps, sf_in=2,
                   wk0=P.conv_kmat(sd[p + ".enc_conv1.weight"], kk), Cout=64, bias0=self.w(p + ".enc_conv1.bias"),
                   out=self.tmp80, out_strides=nchw_out(64, T, FH), B=B, Tout=T, Fout=FH, tag=TAG_PRIOR)
        self._rowln(self.tmp80, dst, 64 * T * FH, 64, FH, p + ".enc_norm1", p + ".enc_prelu1")

    def _aham(self, p, outs, dst):
        """AHAM / AHAM_ori (dbaiat.py:266-288, :308-330)."""
        ah = L.AhamDesc()
        for i in range(4):
            ah.x[i] = outs[i].data_ptr()
        ah.w, ah.bias = self.ctx.up(self.w(p + ".conv1.weight").reshape(64)).data_ptr(), float(self.w(p + ".conv1.bias")[0])
        ah.means, ah.out, ah.plane, ah.B, ah.C = self.means.data_ptr(), dst.data_ptr(), self.T * self.FH, self.B, 64
        self.add(ah, TAG_PRIOR)

    def _dense_decoder(self, de, merged, out, out_C, out_off):
        """dense_decoder (dbaiat.py:527-548) with the sub-pixel up-convolution (:587-602); its 64 -> 1 output goes to
        channel ``out_off // (T*161)`` of ``out`` [B,out_C,T,161].  (The masking decoder :551-584 is the same up to
        its scalar gate, which the CRM operator applies.)"""
        B, T, FH, sd = self.B, self.T, self.FH, self.sd
        for b in range(B):                # merged -> channel block 4 of every batch item (one strided copy per item)
            c = L.EwDesc()
            c.a, c.out = Ctx.ptr(merged, b * 64 * T * FH), Ctx.ptr(self.D80, (b * 320 + 256) * T * FH)
            c.n, c.op = 64 * T * FH, L.EW_COPY
            self.add(c, TAG_EW)
        self._dense_block(de + ".dec_dense1", self.D80, FH, self.tmp80)
        taps = [(0, kf - 1) for kf in range(3)]                   # pad (1,1) in bins
        kk = [(0, kf) for kf in range(3)]
        # co = r*64 + c  ->  channel c, bin 1 + 2w + r   (sub-pixel r = 2, then one zero bin on the left)
        self.gconv(in0=self.src(self.D80, 64, *nchw(320, T, FH)), Tin=T, Fin=FH, taps=taps, sf_in=1,
                   wk0=P.conv_kmat(sd[de + ".dec_conv1.conv.weight"], kk), Cout=128,
                   bias0=self.w(de + ".dec_conv1.conv.bias"), out=self.dec_up,
                   out_strides=(64 * T * F0, 1, T * F0, F0, 2), out_cr=64, out_off=1, B=B, Tout=T, Fout=FH,
                   tag=TAG_PRIOR)
        self._rowln(self.dec_up, self.tmp161, 64 * T * F0, 64, F0, de + ".dec_norm1", de + ".dec_prelu1")
        self.gconv(in0=self.src(self.tmp161, 64, *nchw(64, T, F0)), Tin=T, Fin=F0, taps=[(0, 0)], sf_in=1,
                   wk0=self.w(de + ".out_conv.weight")[:, :, 0, 0].T, Cout=1, bias0=self.w(de + ".out_conv.bias"),
                   out=out, out_strides=nchw_out(out_C, T, F0), out_off=out_off, B=B, Tout=T, Fout=F0, tag=TAG_PRIOR)

    def _gncomb(self, p, i, base, out):
        g = L.GncombDesc()
        up = self.ctx.up
        g.base, g.row, g.col, g.out = base.data_ptr(), self.br[0].data_ptr(), self.br[1].data_ptr(), out.data_ptr()
        g.g_row, g.b_row = up(self.w("%s.row_norm.%d.weight" % (p, i))).data_ptr(), up(self.w("%s.row_norm.%d.bias" % (p, i))).data_ptr()
        g.g_col, g.b_col = up(self.w("%s.col_norm.%d.weight" % (p, i))).data_ptr(), up(self.w("%s.col_norm.%d.bias" % (p, i))).data_ptr()
        g.stats, g.plane, g.B, g.C = self.gn_stats.data_ptr(), self.T * self.FH, self.B, self.d
        g.k1, g.k2, g.eps = float(self.w(p + ".k1")[0]), float(self.w(p + ".k2")[0]), 1e-8
        self.add(g, TAG_PRIOR)

    def build(self, x=None, out=None):
        B, T, FH = self.B, self.T, self.FH
        x = self.x if x is None else x
        out = self.out if out is None else out
        self._dense_encoder("en_ri", x, 2, self.x_ri)
        # ---- AIA_Transformer (dbaiat.py:133-154)
        p = "dual_trans"
        self._pw(self.x_ri, 64, self.w(p + ".input.0.weight")[:, :, 0, 0].T, self.w(p + ".input.0.bias"), self.cur, 32, FH,
                 act=L.ACT_PRELU, act_slope=float(self.w(p + ".input.1.weight")[0]))
        slope_o = float(self.w(p + ".output.0.weight")[0])
        ident = dict(mode=1, scale0=np.ones(32), shift0=np.zeros(32), slope0=slope_o)   # PReLU on load
        cur, nxt = self.cur, self.nxt
        for i in range(4):
            self._encoder_layer("%s.row_trans.%d" % (p, i), 0, cur, self.br[0])
            self._encoder_layer("%s.col_trans.%d" % (p, i), 1, cur, self.br[1])
            self._gncomb(p, i, cur, nxt)
            cur, nxt = nxt, cur
            self._pw(cur, 32, self.w(p + ".output.1.weight")[:, :, 0, 0].T, self.w(p + ".output.1.bias"), self.outs[i], 64,
                     FH, xf=ident)
        self._aham("aham", self.outs, self.merged)
        for ch, de in enumerate(("de1", "de2")):
            self._dense_decoder(de, self.merged, out, 2, ch * T * F0)
        return out


class DualAiaPlan(AiaPlan):
    """dual_aia_trans_merge_crm (model/dbaiat.py:373-413): ri and magnitude encoders -> AIA_Transformer_merge
    (d_model 64, :157-246) -> AHAM_ori x2 -> two dense decoders + the masking decoder -> magnitude/phase recombination.

    The reference runs the merge transformer twice per layer, once per branch; the two runs are the same
    computation: layer 0 of both reads the shared input projection (:206-207), layer i >= 1 of the magnitude
    branch reads mag[i-1] + ri[i-1] (:211) and of the ri branch ri[i-1] + mag[i-1] (:229, ``[-2]`` after the
    append), and IEEE addition commutes.  The golden vectors of the reference module confirm that the two output
    lists are bit-identical (tests/test_oracle_golden.py), so each layer is evaluated once."""

    def __init__(self, ctx, sd, B, T, plan=None):
        super().__init__(ctx, sd, B, T, plan, d=64)
        a = ctx.alloc
        self.mag = a(B, 1, T, F0)
        self.x_mag_en = a(B, 64, T, self.FH)
        self.inp = a(B, 64, T, self.FH)          # shared input projection ("input_mag" == "input_ri")
        self.merged_mag = a(B, 64, T, self.FH)
        self.ri_dec = a(B, 2, T, F0)
        self.o_mask = a(B, 1, T, F0)

    def build(self, x=None, out=None):
        B, T, FH = self.B, self.T, self.FH
        x = self.x if x is None else x
        out = self.out if out is None else out
        c = L.CrmDesc()
        c.x, c.out, c.plane, c.B, c.mode = x.data_ptr(), self.mag.data_ptr(), T * F0, B, 0
        self.add(c, TAG_PRIOR)
        self._dense_encoder("en_ri", x, 2, self.x_ri)
        self._dense_encoder("en_mag", self.mag, 1, self.x_mag_en)
        p = "aia_trans_merge"
        # input projection over cat(mag, ri) (dbaiat.py:205-207): two-source 1x1 + PReLU
        self.gconv(in0=self.src(self.x_mag_en, 64, *nchw(64, T, FH)), in1=self.src(self.x_ri, 64, *nchw(64, T, FH)), Tin=T,
                   Fin=FH, taps=[(0, 0)], sf_in=1, wk0=self.w(p + ".input.0.weight")[:, :, 0, 0].T, Cout=64,
                   bias0=self.w(p + ".input.0.bias"), act=L.ACT_PRELU, act_slope=float(self.w(p + ".input.1.weight")[0]),
                   out=self.inp, out_strides=nchw_out(64, T, FH), B=B, Tout=T, Fout=FH, tag=TAG_PRIOR)
        slope_o = float(self.w(p + ".output.0.weight")[0])
        ident = dict(mode=1, scale0=np.ones(64), shift0=np.zeros(64), slope0=slope_o)   # PReLU on load
        u = self.inp
        for i in range(4):
            self._encoder_layer("%s.row_trans.%d" % (p, i), 0, u, self.br[0])
            self._encoder_layer("%s.col_trans.%d" % (p, i), 1, u, self.br[1])
            self._gncomb(p, i, self.inp, self.nxt)                    # input + k1 * row + k2 * col  (:225, :243)
            self._pw(self.nxt, 64, self.w(p + ".output.1.weight")[:, :, 0, 0].T, self.w(p + ".output.1.bias"), self.outs[i],
                     64, FH, xf=ident)
            if i < 3:                                                 # next layer reads mag[i] + ri[i] = out + out
                e = L.EwDesc()
                e.a, e.b, e.out = self.outs[i].data_ptr(), self.outs[i].data_ptr(), self.cur.data_ptr()
                e.n, e.op, e.s0 = self.outs[i].numel(), L.EW_ADD_MUL, 1.0
                self.add(e, TAG_EW)
                u = self.cur
        self._aham("aham", self.outs, self.merged)
        self._aham("aham_mag", self.outs, self.merged_mag)
        self._dense_decoder("de1", self.merged, self.ri_dec, 2, 0)
        self._dense_decoder("de2", self.merged, self.ri_dec, 2, T * F0)
        self._dense_decoder("de_mag_mask", self.merged_mag, self.o_mask, 1, 0)
        c = L.CrmDesc()
        c.x, c.o, c.ri, c.out = x.data_ptr(), self.o_mask.data_ptr(), self.ri_dec.data_ptr(), out.data_ptr()
        de = "de_mag_mask."
        c.a1, c.b1 = float(self.w(de + "mask1.0.weight").reshape(-1)[0]), float(self.w(de + "mask1.0.bias")[0])
        c.a2, c.b2 = float(self.w(de + "mask2.0.weight").reshape(-1)[0]), float(self.w(de + "mask2.0.bias")[0])
        c.a3, c.b3 = float(self.w(de + "maskconv.weight").reshape(-1)[0]), float(self.w(de + "maskconv.bias")[0])
        c.plane, c.B, c.mode = T * F0, B, 1
        self.add(c, TAG_PRIOR)
        return out
