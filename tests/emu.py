"""TEST INFRASTRUCTURE: a numpy interpreter of the libpdse.so operator descriptors
(include/pdse.h) over CPU tensors.

It executes the *same descriptors* the plan builders record — same packed weights, tap
tables, strides, folded biases — with the semantics the header documents, so the host
logic (weight folding/packing, stride algebra, plan construction) is checked against
the oracle on the CPU-only build container.  It is never used by the product path and
says nothing about the HIP kernels themselves; those are checked on the GPU (-m gpu).
"""
import importlib

import numpy as np
import torch

L = importlib.import_module("prior-diffuse_amd._lib")
P = importlib.import_module("prior-diffuse_amd.packing")

RHO = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for h in (0, 1)] for r in range(16)])


class Mem:
    """Resolve raw pointers to flat float32/int32 numpy views of the tensors a Ctx keeps."""

    def __init__(self, tensors):
        self.regions = []
        for t in tensors:
            if torch.is_tensor(t) and t.device.type == "cpu" and t.numel() > 0:
                self.regions.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t))
        self.regions.sort(key=lambda r: r[0])

    def view(self, ptr, dtype=np.float32):
        """(flat array of the owning tensor, element offset of ptr)."""
        if not ptr:
            return None, 0
        for lo, hi, t in self.regions:
            if lo <= ptr < hi:
                flat = t.view(-1).numpy()
                assert flat.dtype == dtype, (flat.dtype, dtype)
                return flat, (ptr - lo) // t.element_size()
        raise KeyError("pointer %x not inside any known tensor" % ptr)

    def arr(self, ptr, n, dtype=np.float32):
        flat, off = self.view(ptr, dtype)
        return flat[off:off + n]


def _act(v, act, slope=0.0):
    if act == L.ACT_PRELU:
        return np.where(v > 0, v, np.float32(slope) * v)
    if act == L.ACT_ELU:
        return np.where(v > 0, v, np.expm1(np.minimum(v, 0)))
    if act == L.ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-v))
    return v


def _sig(v):
    return 1.0 / (1.0 + np.exp(-v))


def _unpack_a(w, mtiles, ksteps):
    w = w.reshape(mtiles, ksteps, 2, 32)            # [mt, ks, h, col]
    return w.transpose(1, 2, 0, 3).reshape(2 * ksteps, mtiles * 32)   # [k, co]


def _unpack_chain(w, mtiles):
    w = w.reshape(mtiles, 16, 2, 32)                # [mt, r, h, col(out)]
    out = np.zeros((mtiles * 32, 32), np.float32)   # [out, in]
    for r in range(16):
        for h in (0, 1):
            out[:, RHO[r, h]] = w[:, r, h, :].reshape(-1)
    return out


def run_gconv(d, mem):
    B, To, Fo = d.B, d.Tout, d.Fout
    Cin = d.in0.C + d.in1.C
    mtiles = (d.Cout + 31) // 32
    dual = d.epi != L.EPI_LINEAR
    taps = mem.arr(d.taps, 2 * d.ntaps, np.int32).reshape(-1, 2)
    if d.korder == 1:   # pipelined layout: 4-k-step groups in (source, pair chunk, tap) order
        key = (d.epi, d.ntaps, d.in1.C > 0, d.xf_mode)
        rows = P.korder1_rows(d.ntaps, d.in0.C, d.in1.C, P.V2_CP[key])
        assert len(rows) == 2 * d.ksteps

        def unpack(ptr):
            wk = P.unpack_a4(mem.arr(ptr, mtiles * d.ksteps * 64), mtiles, d.ksteps)
            out = np.zeros((d.ntaps * Cin, mtiles * 32), np.float32)
            out[rows[rows >= 0]] = wk[rows >= 0]
            assert np.all(wk[rows < 0] == 0)
            return out
    elif d.korder in (3, 4, 5):   # split-bf16 (3) / plain bf16 (4) / f16x2 (5) GEMM-shaped convolution (csrc/gconv4.hip): weights streamed through LDS
        assert d.epi in (L.EPI_LINEAR, L.EPI_GLU) and not d.cin1 and d.xf_mode == 0

        def unpack(ptr):
            npl = {3: 3, 4: 1, 5: 2}[d.korder]
            n = d.ntaps * Cin // 16 * mtiles * npl * 64 * 8
            wk = P.unpack_s3_gemm(mem.arr(ptr, n, np.int16).view(np.uint16), d.ntaps, d.in0.C, d.in1.C, mtiles * 32, npl, d.wexp)
            return wk
    elif d.korder == 2:   # split-bf16 BIGLU block (csrc/gconv3.hip): exact 3-way bf16 splits in bf16 MFMA fragment order
        assert d.epi == L.EPI_BIGLU and Cin in (32, 4) and d.Cout == 32

        def unpack(ptr):
            if Cin == 4:    # composed encoder stage 1: K = 40 padded to three 16-deep blocks
                return P.unpack_s3_gather(mem.arr(ptr, 3 * 3 * 64 * 8, np.int16).view(np.uint16), 3, 16)[:d.ntaps * 4]
            return P.unpack_s3_gather(mem.arr(ptr, d.ntaps * 2 * 3 * 64 * 8, np.int16).view(np.uint16), d.ntaps)
    else:
        def unpack(ptr):
            return _unpack_a(mem.arr(ptr, mtiles * d.ksteps * 64), mtiles, d.ksteps)
    W0 = unpack(d.w0)
    W1 = unpack(d.w1) if dual else None
    ph1 = None
    if d.w2:    # dual-phase transposed conv: odd output bins from the taps in p1mask
        sel = [i for i in range(d.ntaps) if (d.p1mask >> i) & 1]
        if d.korder == 2:
            def unpack1(ptr):
                return P.unpack_s3_gather(mem.arr(ptr, len(sel) * 2 * 3 * 64 * 8, np.int16).view(np.uint16), len(sel))
        else:
            rows1 = P.korder1_rows(len(sel), d.in0.C, 0, P.V2_CP[(d.epi, d.ntaps, False, d.xf_mode)])
            assert len(rows1) == 2 * d.ksteps1

            def unpack1(ptr):
                wk = P.unpack_a4(mem.arr(ptr, d.ksteps1 * 64), 1, d.ksteps1)
                out = np.zeros((len(sel) * Cin, 32), np.float32)
                out[rows1[rows1 >= 0]] = wk[rows1 >= 0]
                return out
        ph1 = (sel, unpack1(d.w2), unpack1(d.w3), np.zeros((B, 32, To, Fo), np.float32), np.zeros((B, 32, To, Fo), np.float32))
    bI = np.arange(B)[:, None, None, None]
    tI = np.arange(To)[None, None, :, None]
    jI = np.arange(Fo)[None, None, None, :]
    acc0 = np.zeros((B, mtiles * 32, To, Fo), np.float32)
    acc1 = np.zeros_like(acc0) if dual else None
    srcs = [(d.in0, 0)] + ([(d.in1, d.in0.C)] if d.in1.C else [])
    pad_flat, pad_off = mem.view(d.padrow)
    xs0 = mem.arr(d.xf_scale0, Cin) if d.xf_mode else None
    xh0 = mem.arr(d.xf_shift0, Cin) if d.xf_mode else None
    xs1 = mem.arr(d.xf_scale1, Cin) if d.xf_mode == 2 else None
    xh1 = mem.arr(d.xf_shift1, Cin) if d.xf_mode == 2 else None
    for ti, (dt, df) in enumerate(taps):
        tin, fin = tI + dt, jI * d.sf_in + df
        fok = (fin >= 0) & (fin < d.Fin)
        inb = fok & (tin >= 0) & (tin < d.Tin)
        isp = fok & (tin == -1) & bool(d.padrow)
        for S, cbase in srcs:
            flat, off = mem.view(S.ptr)
            cI = np.arange(S.C)[None, :, None, None]
            if S.blk:   # channel-blocked source (pdse_src.blk = 8, korder 3 only)
                assert S.blk == 8 and d.korder in (3, 4, 5), "blocked sources are read by the korder 3 / 4 / 5 kernel only"
                idx = off + bI * S.sb + (cI >> 3) * S.sc + (cI & 7) + tin * S.st + fin * S.sf
            else:
                idx = off + bI * S.sb + cI * S.sc + tin * S.st + fin * S.sf
            idx = np.where(inb, idx, 0)
            v = flat[np.broadcast_to(idx, (B, S.C, To, Fo))]
            v = _act(v, S.act)
            v = np.where(inb, v, 0.0).astype(np.float32)
            if d.korder == 4:            # one plane: the gathered activation is rounded to bf16 (after the load-side ELU)
                v = _bf16_round(v).astype(np.float32)
            if d.padrow:
                pv = pad_flat[pad_off + bI * d.padrow_sb + cbase + cI]
                v = np.where(np.broadcast_to(isp, v.shape), np.broadcast_to(pv, v.shape), v)
            v0 = v1 = v
            if d.xf_mode:
                sl = slice(cbase, cbase + S.C)
                u = np.where(v > 0, v, np.float32(d.xf_slope0) * v)
                v0 = np.where(inb, u * xs0[sl][None, :, None, None] + xh0[sl][None, :, None, None], v)
                if d.xf_mode == 2:
                    u1 = np.where(v > 0, v, np.float32(d.xf_slope1) * v)
                    v1 = np.where(inb, u1 * xs1[sl][None, :, None, None] + xh1[sl][None, :, None, None], v)
                else:
                    v1 = v0
            if d.cin1:
                k0 = ti
            else:
                k0 = ti * Cin + cbase
            if ph1 is not None and ti in ph1[0]:
                k1 = ph1[0].index(ti) * Cin + cbase
                ph1[3][...] += np.einsum("km,bktf->bmtf", ph1[1][k1:k1 + S.C], v0.astype(np.float32), optimize=True)
                ph1[4][...] += np.einsum("km,bktf->bmtf", ph1[2][k1:k1 + S.C], v0.astype(np.float32), optimize=True)
            acc0 += np.einsum("km,bktf->bmtf", W0[k0:k0 + S.C], v0.astype(np.float32), optimize=True)
            if dual:
                acc1 += np.einsum("km,bktf->bmtf", W1[k0:k0 + S.C], v1.astype(np.float32), optimize=True)

    out_flat, out_off = mem.view(d.out)

    def bias(ptr, sb, n):
        if not ptr:
            return np.zeros((1, n, 1, 1), np.float32)
        flat, off = mem.view(ptr)
        return flat[off + np.arange(B)[:, None] * sb + np.arange(n)[None, :]][:, :, None, None]

    def store(y, C_):
        co = np.arange(C_)[None, :, None, None]
        idx = (out_off + bI * d.out_sb + (co // d.out_cr) * d.out_sc_hi + (co % d.out_cr) * d.out_sc_lo
               + tI * d.out_st + jI * d.out_sf + d.out_off)
        idx = np.broadcast_to(idx, y.shape)
        if d.resid:
            rflat, roff = mem.view(d.resid)
            y = y + rflat[idx - out_off + roff]
        out_flat[idx] = y

    def post(y, C_):
        if d.post_scale:
            y = y * mem.arr(d.post_scale, C_)[None, :, None, None] + mem.arr(d.post_shift, C_)[None, :, None, None]
        return _act(y, d.act, d.act_slope).astype(np.float32)

    if d.epi in (L.EPI_LINEAR, L.EPI_GLU):
        y = acc0[:, :d.Cout] + bias(d.bias0, d.bias0_sb, d.Cout)
        if d.epi == L.EPI_GLU:
            y = y * _sig(acc1[:, :d.Cout] + bias(d.bias1, d.bias1_sb, d.Cout))
        store(post(y, d.Cout), d.Cout)
    else:
        def tail(aL, aR, extra, jmax):
            bl, br = bias(d.bias0, d.bias0_sb, 32), bias(d.bias1, d.bias1_sb, 32)          # [B or 1, 32, 1, 1]
            Lh, Rh = aL[:, :32] + bl, aR[:, :32] + br
            if d.bias0_t0:                                                                   # output frame 0 has its own biases
                Lh[:, :, 0] = (aL[:, :32] + bias(d.bias0_t0, d.bias0_sb, 32))[:, :, 0]
                Rh[:, :, 0] = (aR[:, :32] + bias(d.bias1_t0, d.bias1_sb, 32))[:, :, 0]
            def chain_w(ptr, mt, K=32):
                if d.korder == 2:
                    return P.unpack_s3_chain(mem.arr(ptr, mt * (K // 16) * 3 * 64 * 8, np.int16).view(np.uint16), mt, K)
                assert K == 32
                return _unpack_chain(mem.arr(ptr, mt * 16 * 64), mt)

            Wlc, Wrc = chain_w(d.wlc, 1), chain_w(d.wrc, 1)
            mL = _sig(np.einsum("oc,bctf->botf", Wlc, Lh) + mem.arr(d.blc, 32)[None, :, None, None])
            mR = _sig(np.einsum("oc,bctf->botf", Wrc, Rh) + mem.arr(d.brc, 32)[None, :, None, None])
            G = Lh * mR + Rh * mL
            if d.C2 == 1:
                y = np.einsum("c,bctf->btf", mem.arr(d.wc2, 32), G)[:, None] + mem.arr(d.bc2, 1)[0]
            else:
                t2 = (d.C2 + 31) // 32
                Wc2 = chain_w(d.wc2, t2)[:d.C2]
                y = np.einsum("oc,bctf->botf", Wc2, G) + mem.arr(d.bc2, d.C2)[None, :, None, None]
            y = post(y.astype(np.float32), d.C2)
            if d.nx_n == 0 or d.nx_keep:
                co = np.arange(d.C2)[None, :, None, None]
                idx = (out_off + bI * d.out_sb + (co // d.out_cr) * d.out_sc_hi + (co % d.out_cr) * d.out_sc_lo
                       + tI * d.out_st + jI * d.out_sf + d.out_off + extra)
                idx = np.broadcast_to(idx, y.shape)
                out_flat[idx[..., :jmax]] = y[..., :jmax]
            # chained next-stage 1x1 tiles (pdse.h: nx_*)
            for i in range(d.nx_n):
                if d.korder == 2:
                    n16 = 4 * 3 * 64 * 8
                    Wn = P.unpack_s3_chain(mem.arr(d.nx_w, d.nx_n * n16, np.int16).view(np.uint16)[i * n16:(i + 1) * n16], 1, 64)
                else:
                    Wn = np.concatenate([_unpack_chain(mem.arr(d.nx_w, 6 * 1024)[(2 * i + m2) * 1024:(2 * i + m2 + 1) * 1024], 1)
                                         for m2 in (0, 1)], axis=1)                    # [32 out, 64 in]
                bflat, boff = mem.view(d.nx_bias[i])
                bz = bflat[boff + np.arange(B)[:, None] * d.nx_bias_sb[i] + np.arange(32)[None, :]][:, :, None, None]
                z = (np.einsum("oc,bctf->botf", Wn, y) + bz).astype(np.float32)
                zflat, zoff = mem.view(d.nx_out[i])
                co = np.arange(32)[None, :, None, None]
                rel = (bI * d.nx_sb[i] + co * d.nx_sc[i] + tI * d.nx_st[i] + jI * d.nx_sf[i] + d.nx_off[i]
                       + (d.nx_sf[i] // 2 if extra else 0))
                rel = np.broadcast_to(rel, z.shape)[..., :jmax]
                z = z[..., :jmax]
                if d.nx_add[i]:
                    aflat, aoff = mem.view(d.nx_add[i])
                    z = z + aflat[aoff + rel]
                zflat[zoff + rel] = z
                if i == d.nx_row0:
                    r0 = np.broadcast_to(bI * d.nx_sb[i] + co * d.nx_sc[i] + jI * d.nx_sf[i] + 0 * tI, (B, 32) + z.shape[2:])[:, :, :1]
                    zflat[zoff + r0[..., :jmax]] = np.broadcast_to(bz, r0.shape)[..., :jmax]

        tail(acc0, acc1, 0, Fo)
        if ph1 is not None:
            tail(ph1[3], ph1[4], d.out_sf // 2, d.Fout1)


def run_time(d, mem):
    B, NF = d.B, d.NF
    t = mem.arr(d.t, B)
    table = mem.arr(d.table, d.max_steps * 128).reshape(d.max_steps, 128)
    lo = np.clip(np.floor(t).astype(int), 0, d.max_steps - 1)
    hi = np.clip(np.ceil(t).astype(int), 0, d.max_steps - 1)
    x = table[lo] + (table[hi] - table[lo]) * (t - np.floor(t))[:, None]
    p1T = mem.arr(d.p1T, 128 * 512).reshape(128, 512)
    p2T = mem.arr(d.p2T, 512 * 512).reshape(512, 512)
    y = x @ p1T + mem.arr(d.b1, 512)
    y = y * _sig(y)
    y = y @ p2T + mem.arr(d.b2, 512)
    y = (y * _sig(y)).astype(np.float32)
    if d.temb:
        mem.arr(d.temb, B * 512)[:] = y.reshape(-1)
    wfT = mem.arr(d.wfT, 512 * NF).reshape(512, NF)
    mem.arr(d.out, B * NF)[:] = (y @ wfT + mem.arr(d.bf, NF)).astype(np.float32).reshape(-1)


def run_ew(d, mem):
    a = mem.arr(d.a, d.n)
    b = mem.arr(d.b, d.n) if d.b else None
    c = mem.arr(d.c, d.n) if d.c else None
    s0, s1, s2 = np.float32(d.s0), np.float32(d.s1), np.float32(d.s2)
    if d.op == L.EW_DIV:
        y = a / s0
    elif d.op == L.EW_UPDATE:
        y = s0 * (a - s1 * b)
    elif d.op == L.EW_UPDATE_FINAL:
        y = ((s0 * (a - s1 * b)) + c) * s2
    elif d.op == L.EW_ADD_MUL:
        y = (a + b) * s0
    else:
        y = a.copy()
    mem.arr(d.out, d.n)[:] = y.astype(np.float32)


def run_compand(d, mem):
    n = d.B * 2 * d.plane
    x = mem.arr(d.in_, n).reshape(d.B, 2, d.plane).copy()
    mag = np.sqrt(x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1])
    safe = np.where(mag > 0, mag, 1)
    cr, sr = np.where(mag > 0, x[:, 0] / safe, 1.0), np.where(mag > 0, x[:, 1] / safe, 0.0)
    m2 = np.sqrt(mag) if d.mode == 0 else mag * mag
    out = np.stack([m2 * cr, m2 * sr], axis=1).astype(np.float32)
    if d.F > 0:                                    # strided output: out[b*out_sb + ri*out_sc + t*out_st + f]
        T = d.plane // d.F
        flat, off = mem.view(d.out)
        idx = (off + np.arange(d.B)[:, None, None, None] * d.out_sb + np.arange(2)[None, :, None, None] * d.out_sc
               + np.arange(T)[None, None, :, None] * d.out_st + np.arange(d.F)[None, None, None, :])
        flat[idx] = out.reshape(d.B, 2, T, d.F)
        return
    mem.arr(d.out, n)[:] = out.reshape(-1)


def run_wavprep(d, mem):
    x = mem.arr(d.wav, d.B * d.L).reshape(d.B, d.L)
    lens = mem.arr(d.lens, d.B, np.int32).astype(np.float32) if d.lens else np.full(d.B, d.L, np.float32)
    c = np.sqrt((x.astype(np.float32) ** 2).sum(1) / lens) if d.normalize else np.ones(d.B, np.float32)
    if d.c:
        mem.arr(d.c, d.B)[:] = c
    xp = np.pad(x / c[:, None], ((0, 0), (d.pad, d.pad)), mode="reflect")
    mem.arr(d.xpad, d.B * (d.L + 2 * d.pad))[:] = xp.astype(np.float32).reshape(-1)


def run_ola(d, mem):
    fr = mem.arr(d.frames, d.B * d.n_fft * d.T).reshape(d.B, d.n_fft, d.T)
    w2 = mem.arr(d.win2, d.n_fft)
    full = d.n_fft + d.hop * (d.T - 1)
    y = np.zeros((d.B, max(full, d.n_fft // 2 + d.L)), np.float32)
    env = np.zeros(y.shape[1], np.float32)
    for t in range(d.T):
        y[:, t * d.hop:t * d.hop + d.n_fft] += fr[:, :, t]
        env[t * d.hop:t * d.hop + d.n_fft] += w2
    h = d.n_fft // 2
    out = np.where(env[h:h + d.L] > 1e-11, y[:, h:h + d.L] / np.maximum(env[h:h + d.L], 1e-30), 0)
    if d.c:
        out = out * mem.arr(d.c, d.B)[:, None]
    mem.arr(d.out, d.B * d.L)[:] = out.astype(np.float32).reshape(-1)


def run_sigma(d, mem):
    n = d.nplanes * d.plane
    init = mem.arr(d.init, n).reshape(d.nplanes, d.plane)
    a = mem.arr(d.a, n).reshape(d.nplanes, d.plane)
    m = np.abs(init) / np.abs(init).max(1, keepdims=True)
    m = m / np.float32(2) + np.float32(0.5)
    mem.arr(d.out, n)[:] = (a * np.sqrt(m)).astype(np.float32).reshape(-1)


def run_ln(d, mem):
    x = mem.arr(d.in_, d.B * d.T * d.N).reshape(d.B, d.T, d.N)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    y = ((x - mu) / np.sqrt(var + np.float32(d.eps)) * mem.arr(d.gamma, d.N) + mem.arr(d.beta, d.N)).astype(np.float32)
    flat, off = mem.view(d.out)
    b = np.arange(d.B)[:, None, None]
    t = np.arange(d.T)[None, :, None]
    j = np.arange(d.N)[None, None, :]
    c = j // d.r
    cpos = (c >> 3) * d.os_hi + (c & 7) if d.blk else c * d.os_hi          # blk 8: channels in blocks of 8 (pdse_src.blk)
    flat[off + b * d.osb + cpos + (j % d.r) * d.os_lo + t * d.os_t] = y


def run_lstm(d, mem):
    H, G, T, Bp, B = d.H, d.G, d.T, d.Bp, d.B
    gx = mem.arr(d.gx, G * T * 4 * H * Bp).reshape(G, T, 4 * H, Bp)
    whh = mem.arr(d.whh, G * (H // 8) * (H // 2) * 64).reshape(G, H // 8, H // 2, 2, 32)
    yflat, yoff = mem.view(d.y)
    for g in range(G):
        # rebuild W_hh [4H, H] from the slice fragments: tile row i = q*8+u <-> row q*H + 8s + u
        W = np.zeros((4 * H, H), np.float32)
        for s in range(H // 8):
            frag = whh[g, s]                                  # [ks, h, i]
            wk = frag.reshape(H // 2 * 2, 32)                 # k = 2ks + h
            for i in range(32):
                W[(i // 8) * H + 8 * s + (i % 8)] = wk[:, i]
        h = np.zeros((B, H), np.float32)
        c = np.zeros((B, H), np.float32)
        for t in range(T):
            gate = gx[g, t, :, :B].T + h @ W.T
            i_, f_, g_, o_ = np.split(gate, 4, axis=1)
            c = _sig(f_) * c + _sig(i_) * np.tanh(g_)
            h = (_sig(o_) * np.tanh(c)).astype(np.float32)
            idx = (yoff + np.arange(B)[:, None] * d.y_sb + t * d.y_st + np.arange(H)[None, :] * d.y_su + g * d.y_sg)
            yflat[idx] = h


def _unpack_lstm_slices(w, H, K):
    """[H/8][K/8][64][4] -> [4H, K] in the packed K order (inverse of packing.pack_lstm_slices with korder = identity)."""
    w = w.reshape(H // 8, K // 8, 2, 32, 4)                  # [s, q, h, col, i]
    W = np.zeros((4 * H, K), np.float32)
    col = np.arange(32)
    for s_ in range(H // 8):
        rows = (col // 8) * H + 8 * s_ + (col % 8)
        for q in range(K // 8):
            for h in range(2):
                for i in range(4):
                    W[rows, 2 * (4 * q + i) + h] = w[s_, q, h, :, i]
    return W


def run_glstm(d, mem):
    """pdse_glstm_desc: both LSTM layers + the folded LayerNorm, from the packed operands, frame by frame."""
    H, G, T, Bp, B = d.H, d.G, d.T, d.Bp, d.B
    gx1 = mem.arr(d.gx1, G * T * 4 * H * Bp).reshape(G, T, Bp, 4 * H)
    n = G * (H // 8) * (H // 8) * 256
    W1 = [_unpack_lstm_slices(w, H, H) for w in mem.arr(d.whh1, n).reshape(G, -1)]
    W2 = [_unpack_lstm_slices(w, H, H) for w in mem.arr(d.whh2, n).reshape(G, -1)]
    Wi = [_unpack_lstm_slices(w, H, H) for w in mem.arr(d.wih2, n).reshape(G, -1)]
    r2, c2 = mem.arr(d.r2, G * 4 * H).reshape(G, 4 * H), mem.arr(d.c2, G * 4 * H).reshape(G, 4 * H)
    yflat, yoff = mem.view(d.y)

    def cell(gate, c):
        i_, f_, g_, o_ = np.split(gate, 4, axis=1)
        c = _sig(f_) * c + _sig(i_) * np.tanh(g_)
        return (_sig(o_) * np.tanh(c)).astype(np.float32), c

    h1 = np.zeros((G, B, H), np.float32)
    c1 = np.zeros((G, B, H), np.float32)
    h2 = np.zeros((G, B, H), np.float32)
    c2s = np.zeros((G, B, H), np.float32)
    for t in range(T):
        for g in range(G):
            h1[g], c1[g] = cell(gx1[g, t, :B, :] + h1[g] @ W1[g].T, c1[g])
        allh = np.stack([h1[0], h1[1]], -1).reshape(B, -1).astype(np.float64)       # feature 2u + g'
        mu, var = allh.mean(1), allh.var(1)
        rs = 1.0 / np.sqrt(var + d.eps)
        for g in range(G):
            xk = np.zeros((B, H), np.float32)                                       # B operand in stage B's K order
            for gq in range(H // 8):
                gs, kq = gq >> 5, (H // 16) * g + (gq & 31)
                for i in range(4):
                    for hh in range(2):
                        xk[:, 2 * (4 * gq + i) + hh] = h1[gs][:, 8 * kq + 2 * i + hh]
            gx2 = rs[:, None] * (xk @ Wi[g].T - mu[:, None] * r2[g][None, :]) + c2[g][None, :]
            h2[g], c2s[g] = cell(gx2.astype(np.float32) + h2[g] @ W2[g].T, c2s[g])
            idx = (yoff + np.arange(B)[:, None] * d.y_sb + t * d.y_st + np.arange(H)[None, :] * d.y_su + g * d.y_sg)
            yflat[idx] = h2[g]


def run_glstmp(d, mem):
    """pdse_glstmp_desc (csrc/lstmp.hip): the persistent form - the same recurrence from its own operand layout
    (w1 / w2i / w2h [G][H][4][H], natural K order), frame by frame."""
    H, G, T, Bp, B = d.H, d.G, d.T, d.Bp, d.B
    gx1 = mem.arr(d.gx1, G * T * 4 * H * Bp).reshape(G, T, Bp, 4 * H)
    n = G * H * 4 * H
    unp = lambda p_: mem.arr(p_, n).reshape(G, H, 4, H).transpose(0, 2, 1, 3).reshape(G, 4 * H, H)   # noqa: E731  rows q*H + u
    W1, Wi, W2 = unp(d.w1), unp(d.w2i), unp(d.w2h)
    r2, c2 = mem.arr(d.r2, G * 4 * H).reshape(G, 4 * H), mem.arr(d.c2, G * 4 * H).reshape(G, 4 * H)
    yflat, yoff = mem.view(d.y)

    def cell(gate, c):
        i_, f_, g_, o_ = np.split(gate, 4, axis=1)
        c = _sig(f_) * c + _sig(i_) * np.tanh(g_)
        return (_sig(o_) * np.tanh(c)).astype(np.float32), c

    h1 = np.zeros((G, B, H), np.float32)
    c1 = np.zeros((G, B, H), np.float32)
    h2 = np.zeros((G, B, H), np.float32)
    c2s = np.zeros((G, B, H), np.float32)
    for t in range(T):
        for g in range(G):
            h1[g], c1[g] = cell(gx1[g, t, :B, :] + h1[g] @ W1[g].T, c1[g])
        allh = np.stack([h1[0], h1[1]], -1).reshape(B, -1)                           # feature 2u + g'
        mu, var = allh.astype(np.float64).mean(1), allh.astype(np.float64).var(1)
        rs = 1.0 / np.sqrt(var + d.eps)
        for g in range(G):
            xk = allh[:, H * g:H * g + H]
            gx2 = rs[:, None] * (xk @ Wi[g].T - mu[:, None] * r2[g][None, :]) + c2[g][None, :]
            h2[g], c2s[g] = cell(gx2.astype(np.float32) + h2[g] @ W2[g].T, c2s[g])
            idx = (yoff + np.arange(B)[:, None] * d.y_sb + t * d.y_st + np.arange(H)[None, :] * d.y_su + g * d.y_sg)
            yflat[idx] = h2[g]


def run_tcm(d, mem):
    """pdse_tcm_desc: fused TCM residual block + the next block's conv1, from the packed operands."""
    B, T, dil = d.B, d.T, d.dil
    x = mem.arr(d.x, B * 256 * T).reshape(B, 256, T).astype(np.float64)
    h = mem.arr(d.h, B * 64 * T).reshape(B, 64, T).astype(np.float64)
    wbr = mem.arr(d.wbr, 2 * 2 * 20 * 2 * 64 * 4).reshape(2, 2, 20, 2, 2, 32, 4)      # [mi, kh, g, which, hh, col, e]
    km = wbr.transpose(3, 1, 2, 6, 4, 0, 5).reshape(2, 320, 64)                       # [which][(kh,g,e,hh)][(mi,col)]
    xf = mem.arr(d.xf, 256).reshape(2, 64, 2)
    xf2 = mem.arr(d.xf2, 128).reshape(64, 2)
    pre = []
    for which, slope in ((0, d.slope_main), (1, d.slope_mask)):
        v = np.where(h > 0, h, np.float32(slope) * h) * xf[which, :, 0][None, :, None] + xf[which, :, 1][None, :, None]
        vp = np.zeros((B, 64, T + 4 * dil))
        vp[:, :, 2 * dil:2 * dil + T] = v
        cols = np.concatenate([vp[:, :, k * dil:k * dil + T] for k in range(5)], axis=1)   # [B, 320, T], row = tap*64 + c
        pre.append(np.einsum("bkt,ko->bot", cols, km[which]))
    main = pre[0] + mem.arr(d.bmain, 64)[None, :, None]
    mask = pre[1] + mem.arr(d.bmask, 64)[None, :, None]
    g = main * _sig(mask)
    g = np.where(g > 0, g, np.float32(d.slope2) * g) * xf2[:, 0][None, :, None] + xf2[:, 1][None, :, None]
    wc2 = mem.arr(d.wc2, 8 * 8 * 64 * 4).reshape(8, 8, 2, 32, 4)                      # [mt, g, hh, col, e]
    k2 = wc2.transpose(1, 4, 2, 0, 3).reshape(64, 256)                                # [(g,e,hh)][(mt,col)]
    xo = np.einsum("bkt,ko->bot", g, k2) + mem.arr(d.bc2, 256)[None, :, None] + x
    if d.h_out:
        wn = mem.arr(d.wn1, 4 * 2 * 2 * 4 * 64 * 4).reshape(4, 2, 2, 4, 2, 32, 4)     # [w, q, mo, g, hh, col, e]
        w1 = np.zeros((64, 256))
        for gi in range(4):
            for e in range(4):
                for hh in (0, 1):
                    for w in range(4):
                        for q in range(2):
                            w1[:, 64 * w + 32 * q + RHO[4 * gi + e, hh]] = wn[w, q, :, gi, hh, :, e].reshape(64)
        ho = np.einsum("bkt,ok->bot", xo, w1) + mem.arr(d.bn1, 64)[None, :, None]
        mem.arr(d.h_out, B * 64 * T)[:] = ho.astype(np.float32).reshape(-1)
    mem.arr(d.x_out, B * 256 * T)[:] = xo.astype(np.float32).reshape(-1)


def run_tcm2(d, mem):
    """pdse_tcm2_desc: the same block on split operands; h travels as the bf16 planes of both branches' transforms."""
    B, T, dil = d.B, d.T, d.dil
    npl = d.np or 3                                                  # 1: plain bf16 operands (the opt-in bf16 mode); 2: f16x2
    qA, q2, qN = (tuple(d.qexp) if npl == 2 else (0, 0, 0))
    rnd = _bf16_round if npl == 1 else (lambda v_: v_)
    par = mem.arr(d.par, 832)
    u16 = lambda ptr, n: mem.arr(ptr, n, np.int16).view(np.uint16)   # noqa: E731
    x = mem.arr(d.x, B * 256 * T).reshape(B, 256, T).astype(np.float64)
    if d.mode == 0:
        km = P.unpack_tcm2_branch(u16(d.wbr, 2 * 2 * 20 * npl * 64 * 8), qA)
        v = P.tcm2_join_h(u16(d.hs, int(np.prod(P.tcm2_hs_shape(B, T, npl)))), B, T)
        pre = []
        for which in range(2):
            vp = np.zeros((B, 64, T + 4 * dil))
            vp[:, :, 2 * dil:2 * dil + T] = v[which]
            cols = np.concatenate([vp[:, :, k * dil:k * dil + T] for k in range(5)], axis=1)
            pre.append(np.einsum("bkt,ko->bot", cols, km[which].astype(np.float64)))
        gp = par[:256].reshape(64, 4)
        g = (pre[0] + gp[:, 0][None, :, None]) * _sig(pre[1] + gp[:, 1][None, :, None])
        g = np.where(g > 0, g, np.float32(d.slope2) * g) * gp[:, 2][None, :, None] + gp[:, 3][None, :, None]
        g = rnd(g.astype(np.float32)).astype(np.float64)        # the kernel splits (or rounds) the fp32 value
        k2 = P.unpack_tcm2_conv2(u16(d.wc2, 8 * 4 * npl * 64 * 8), q2).astype(np.float64)
        xo = np.einsum("bkt,ko->bot", g, k2) + par[256:512][None, :, None] + x
        mem.arr(d.x_out, B * 256 * T)[:] = xo.astype(np.float32).reshape(-1)
    else:
        xo = x
    if d.hs_out:
        w1 = P.unpack_bglu_chain(np.asarray(u16(d.wn1, 2 * 16 * npl * 64 * 8)).reshape(2, 16, npl, 64, 8), 64, 256, qN).astype(np.float64)
        ho = (np.einsum("bkt,ok->bot", rnd(xo.astype(np.float32)).astype(np.float64), w1) + par[512:576][None, :, None]).astype(np.float32)
        xn = par[576:].reshape(64, 4)
        vm = xn[:, 0][None, :, None] * np.where(ho > 0, ho, np.float32(d.slope_main_next) * ho) + xn[:, 1][None, :, None]
        vk = xn[:, 2][None, :, None] * np.where(ho > 0, ho, np.float32(d.slope_mask_next) * ho) + xn[:, 3][None, :, None]
        mem.arr(d.hs_out, int(np.prod(P.tcm2_hs_shape(B, T, npl))), np.int16)[:] = P.tcm2_split_h(vm.astype(np.float32), vk.astype(np.float32), npl).view(np.int16).reshape(-1)


def run_gcrnlast(d, mem):
    """pdse_gcrnlast_desc: gated ConvTranspose 32 -> 1 (1,3)/stride 2 + BN + ELU + Linear(161,161)."""
    B, T = d.B, d.T
    x0 = mem.arr(d.in0, B * 16 * T * 80).reshape(B, 16, T, 80).astype(np.float64)
    x1 = mem.arr(d.in1, B * 16 * T * 80).reshape(B, 16, T, 80).astype(np.float64)
    u = np.concatenate([x0, np.where(x1 > 0, x1, np.expm1(np.minimum(x1, 0)))], axis=1)       # [B,32,T,80]
    w = [mem.arr(d.w1, 96).reshape(32, 3).astype(np.float64), mem.arr(d.w2, 96).reshape(32, 3).astype(np.float64)]
    pre = []
    for wk in w:
        o = np.zeros((B, T, 161))
        for k in range(3):                                    # output bin 2*j + k <- input bin j
            o[:, :, k:k + 160:2] += np.einsum("bctf,c->btf", u, wk[:, k])
        pre.append(o)
    y = (pre[0] + d.b1) * _sig(pre[1] + d.b2)
    y = y * d.bn_scale + d.bn_shift
    y = np.where(y > 0, y, np.expm1(np.minimum(y, 0)))
    fcT = mem.arr(d.fcT, 161 * 161).reshape(161, 161).astype(np.float64)
    res = (y @ fcT + mem.arr(d.fcb, 161)).astype(np.float32)                                     # [B,T,161]
    flat, off = mem.view(d.out)
    idx = off + np.arange(B)[:, None, None] * d.out_sb + np.arange(T)[None, :, None] * 161 + np.arange(161)[None, None, :]
    flat[idx] = res



def _bf16_round(v):
    """What a one-plane (plain bf16) operand keeps of a float array."""
    return P.bf16_to_f32(P.bf16_rne(v)).reshape(np.shape(v))


def run_planes(d, mem):
    """pdse_planes_desc: fp32 [B, 32, T, F] -> hp planes; with w: the planes of the 1x1 convolution over one or two sources, for
    one or two weight sets (the two decoders' stage-5 conv1)."""
    assert d.hp_t0 == P.HP_T0 and d.hp_f0 == P.HP_F0 and d.hp_Tp == d.T + P.HP_T0 and d.hp_Fp == d.F + 2 * P.HP_F0

    def gather(ptr, C):
        flat, off = mem.view(ptr)
        b, c, t, f = np.meshgrid(np.arange(d.B), np.arange(C), np.arange(d.T), np.arange(d.F), indexing="ij")
        return flat[off + b * d.in_sb + c * d.in_sc + t * d.in_st + f * d.in_sf]

    def put(ptr, x):
        hpf, hoff = mem.view(ptr, np.int16)
        hp = hpf.view(np.uint16)[hoff:hoff + d.B * d.hp_sb].reshape(d.B, d.hp_Tp, 4, d.np, d.hp_Fp, 8)
        hp[...] = P.hp_split(x.astype(np.float32), d.np)

    if not d.w[0]:
        put(d.hp, gather(d.in_, 32))
        return
    x = gather(d.in_, d.cin0).astype(np.float64)
    if d.cin1:
        x = np.concatenate([x, gather(d.in1, d.cin1).astype(np.float64)], 1)
    for i in range(d.nd):
        w = mem.arr(d.w[i], (d.cin0 + d.cin1) * 32).reshape(-1, 32).astype(np.float64)             # [k][c]
        y = np.einsum("kc,bktf->bctf", w, x)
        if d.bias[i]:
            flat, off = mem.view(d.bias[i])
            y = y + flat[off + np.arange(d.B)[:, None] * d.bias_sb + np.arange(32)[None, :]][:, :, None, None]
        put(d.hp1 if i else d.hp, y)


def run_bglu(d, mem):
    """pdse_bglu_desc (include/pdse.h, csrc/bglu.hip): gather convolutions on a plane tensor (or, encoder stage 1, on the two
    fp32 sources), the BIGLU tail with the host-side foldings the descriptor documents (sigmoid from pre-activations
    scaled by -log2 e, BatchNorm inside conv2, PReLU as max(v, slope v)), chained tiles written as planes / fp32.
    One-plane (bf16) launches round every matrix operand to bf16 where the kernel does."""
    B, T, Fo, npl = d.B, d.Tout, d.Fout, d.np
    rnd = _bf16_round if npl == 1 else (lambda v: v)
    u16 = lambda ptr, n: mem.arr(ptr, n, np.int16).view(np.uint16)   # noqa: E731
    blk = npl * 64 * 8
    nt = d.ntaps
    qG, qLC, qC2, qNX = (tuple(d.qexp) if npl == 2 else (0, 0, 0, 0))    # f16x2: the power-of-two scale of each weight group
    taps = [(d.tap_dt[i], d.tap_df[i]) for i in range(nt)]
    bI, tI, jI = np.meshgrid(np.arange(B), np.arange(T), np.arange(Fo), indexing="ij")

    def unp(ptr, nb):
        return np.asarray(u16(ptr, nb * blk)).reshape(nb, npl, 64, 8)

    if d.x0.ptr:
        W0, W1 = P.unpack_bglu_in4(unp(d.w0, 3), qG), P.unpack_bglu_in4(unp(d.w1, 3), qG)     # [40, 32], k = tap*4 + channel
        accL = np.zeros((B, 32, T, Fo), np.float32)
        accR = np.zeros_like(accL)
        for ti, (dt, df) in enumerate(taps):
            tin, fin = tI + dt, jI * d.sf_in + df
            inb = (tin >= 0) & (tin < d.Tin) & (fin >= 0) & (fin < d.Fin)
            tcl, fcl = np.clip(tin, 0, d.Tin - 1), np.clip(fin, 0, d.Fin - 1)
            for si, S in enumerate((d.x0, d.x1)):
                flat, off = mem.view(S.ptr)
                for c in range(2):
                    v = rnd(np.where(inb, flat[off + bI * S.sb + c * S.sc + tcl * S.st + fcl * S.sf], 0.0).astype(np.float32))
                    k = ti * 4 + si * 2 + c
                    accL += W0[k][None, :, None, None] * v[:, None]
                    accR += W1[k][None, :, None, None] * v[:, None]
        accs = [(accL, accR)]
    else:
        hpf, hoff = mem.view(d.hp, np.int16)
        hp = hpf.view(np.uint16)[hoff:hoff + B * d.hp_sb].reshape(B, d.hp_Tp, 4, npl, d.hp_Fp, 8)
        if d.hp_par:                                                                         # rows stored split by parity
            assert d.sf_in == 2
            hp = hp[:, :, :, :, P.hp_par_pos(d.hp_Fp), :]                                     # natural[i] = stored[pos[i]]
        H = P.hp_join(hp, with_margins=True)                                                 # [B, 32, Tp, Fp]
        W = [P.unpack_bglu_gather(unp(p_, 2 * nt), nt, qG) for p_ in (d.w0, d.w1)]
        ph_taps = [list(range(nt))]
        if d.p1mask:
            t1 = [i for i in range(nt) if (d.p1mask >> i) & 1]
            W += [P.unpack_bglu_gather(unp(p_, 2 * len(t1)), len(t1), qG) for p_ in (d.w2, d.w3)]
            ph_taps.append(t1)
        accs = []
        for ph, tl in enumerate(ph_taps):
            aL = np.zeros((B, 32, T, Fo), np.float32)
            aR = np.zeros_like(aL)
            for r, ti in enumerate(tl):
                dt, df = taps[ti]
                tp, fp = tI + dt + d.hp_t0, jI * d.sf_in + df + d.hp_f0
                assert tp.min() >= 0 and tp.max() < d.hp_Tp and fp.min() >= 0 and fp.max() < d.hp_Fp
                v = H[bI, :, tp, fp].transpose(0, 3, 1, 2)                                  # [B, 32, T, Fo]
                aL += np.einsum("km,bktf->bmtf", W[2 * ph][32 * r:32 * r + 32], v, optimize=True)
                aR += np.einsum("km,bktf->bmtf", W[2 * ph + 1][32 * r:32 * r + 32], v, optimize=True)
            accs.append((aL, aR))

    def vec(ptr, sb, n):
        flat, off = mem.view(ptr)
        return flat[off + np.arange(B)[:, None] * sb + np.arange(n)[None, :]][:, :, None, None]

    bl, br = vec(d.bias0, d.bias_sb, 32), vec(d.bias1, d.bias_sb, 32)
    Wlc = P.unpack_bglu_chain(unp(d.wlc, 2)[None], 32, 32, qLC)
    Wrc = P.unpack_bglu_chain(unp(d.wrc, 2)[None], 32, 32, qLC)
    blc, brc = mem.arr(d.blc, 32)[None, :, None, None], mem.arr(d.brc, 32)[None, :, None, None]
    if d.C2 == 64:
        Wc2 = P.unpack_bglu_chain(np.asarray(u16(d.wc2, 4 * blk)).reshape(2, 2, npl, 64, 8), 64, 32, qC2)
        bc2 = mem.arr(d.bc2, 64)[None, :, None, None]
    Wn = []
    if d.nx_n:
        nxw = np.asarray(u16(d.nx_w, d.nx_n * 4 * blk)).reshape(d.nx_n, 1, 4, npl, 64, 8)
        Wn = [P.unpack_bglu_chain(nxw[i], 32, 64, qNX) for i in range(d.nx_n)]
    slope = np.float32(d.slope)
    assert slope <= 1.0
    for ph, (aL, aR) in enumerate(accs):
        Lh, Rh = aL + bl, aR + br
        if d.bias0_t0:
            Lh[:, :, 0] = (aL + vec(d.bias0_t0, d.bias_sb, 32))[:, :, 0]
            Rh[:, :, 0] = (aR + vec(d.bias1_t0, d.bias_sb, 32))[:, :, 0]
        mL = np.einsum("oc,bctf->botf", Wlc, rnd(Lh)) + blc          # pre-activations x (-log2 e)
        mR = np.einsum("oc,bctf->botf", Wrc, rnd(Rh)) + brc
        G = (Lh / (1.0 + np.exp2(mR)) + Rh / (1.0 + np.exp2(mL))).astype(np.float32)
        jmax = Fo if ph == 0 else d.Fout1
        if d.C2 == 1:
            v = np.einsum("c,bctf->btf", mem.arr(d.wc2v, 32), G) + mem.arr(d.bc2, 1)[0]
            y = np.maximum(v, slope * v).astype(np.float32)
            flat, off = mem.view(d.out)
            bin_ = d.out_sf // 2 if d.p1mask else 0
            idx = off + bI[..., 0:1] * 0 + bI * d.out_sb + tI * d.out_st + jI * d.out_sf + d.out_off + ph * bin_
            flat[idx[..., :jmax]] = y[..., :jmax]
            continue
        O = np.einsum("oc,bctf->botf", Wc2, rnd(G)) + bc2
        Y = np.maximum(O, slope * O).astype(np.float32)
        if d.nx_n == 0:
            flat, off = mem.view(d.out)
            co = np.arange(64)[None, :, None, None]
            idx = off + bI[:, None] * d.out_sb + co * d.out_sc + tI[:, None] * d.out_st + jI[:, None] * d.out_sf + d.out_off
            flat[np.broadcast_to(idx, Y.shape)] = Y
            continue
        for i in range(d.nx_n):
            Z = np.einsum("oc,bctf->botf", Wn[i], rnd(Y)) + (vec(d.nx_bias[i], d.nx_bias_sb[i], 32) if d.nx_bias[i] else 0.0)
            co = np.arange(32)[None, :, None, None]
            if i == 0:
                bins = (2 * jI + ph) if d.p1mask else jI
                if d.nx_add:
                    flat, off = mem.view(d.nx_add)
                    bpos = ((bins & 1) * d.skip_Fh + (bins >> 1)) if d.skip_Fh else bins                     # bins split by parity
                    idx = off + bI[:, None] * d.add_sb + (co >> 2) * d.add_sc + (co & 3) + tI[:, None] * d.add_st + bpos[:, None] * d.add_sf   # groups of 4 channels
                    idx = np.where(np.broadcast_to((jI < jmax)[:, None], Z.shape), np.broadcast_to(idx, Z.shape), off)
                    Z = Z + flat[idx]
                Z = Z.astype(np.float32)
                hpf, hoff = mem.view(d.nx_hp, np.int16)
                nhp = hpf.view(np.uint16)[hoff:hoff + B * d.nx_hp_sb].reshape(B, d.nx_Tp, 4, npl, d.nx_Fp, 8)
                new = P.hp_split(np.zeros((B, 32, d.nx_Tp - d.nx_t0, d.nx_Fp - 2 * d.nx_f0), np.float32), npl)   # shape helper
                assert new.shape == nhp.shape and d.nx_t0 == P.HP_T0 and d.nx_f0 == P.HP_F0
                pos = P.hp_par_pos(d.nx_Fp) if d.nx_par else np.arange(d.nx_Fp)          # stored index of bin index i
                for bb in range(B):
                    for tt in range(T):
                        cols = np.arange(jmax)
                        fb = (2 * cols + ph) if d.p1mask else cols
                        piece = P.hp_split(Z[bb:bb + 1, :, tt:tt + 1, :jmax], npl)[0, P.HP_T0, :, :, P.HP_F0:P.HP_F0 + jmax, :]
                        nhp[bb, tt + d.nx_t0, :, :, pos[fb + d.nx_f0], :] = piece.transpose(2, 0, 1, 3)
                if d.nx_row0:
                    bz = np.broadcast_to(vec(d.nx_bias[0], d.nx_bias_sb[0], 32), (B, 32, 1, Fo)).astype(np.float32)
                    row = nhp[:, d.nx_t0 - 1]                                           # view [B, 4, npl, Fp, 8]
                    row[:, :, :, pos[d.nx_f0:d.nx_f0 + Fo], :] = P.hp_split(bz, npl)[:, P.HP_T0, :, :, P.HP_F0:P.HP_F0 + Fo, :]
            else:
                flat, off = mem.view(d.nx_out[i - 1])
                k_ = i - 1
                jp = ((jI & 1) * d.skip_Fh + (jI >> 1)) if d.skip_Fh else jI
                idx = off + bI[:, None] * d.nx_sb[k_] + (co >> 2) * d.nx_sc[k_] + (co & 3) + tI[:, None] * d.nx_st[k_] + jp[:, None] * d.nx_sf[k_]
                flat[np.broadcast_to(idx, Z.shape)] = Z.astype(np.float32)


def run_tcm2s(d, mem):
    """pdse_tcm2s_desc: the residual blocks of the stack, one after the other (what the persistent launch computes)."""
    for i in range(d.n):
        run_tcm2(d.blk[i], mem)


RUNNERS = {L.Tcm2sDesc: run_tcm2s, L.BgluDesc: run_bglu, L.PlanesDesc: run_planes, L.GcrnLastDesc: run_gcrnlast, L.TcmDesc: run_tcm, L.Tcm2Desc: run_tcm2, L.GconvDesc: run_gconv, L.TimeDesc: run_time, L.EwDesc: run_ew, L.CompandDesc: run_compand,
           L.WavprepDesc: run_wavprep, L.OlaDesc: run_ola, L.SigmaDesc: run_sigma, L.LnDesc: run_ln,
           L.LstmDesc: run_lstm, L.GlstmDesc: run_glstm, L.GlstmpDesc: run_glstmp}


def run(descs, keep, begin=0, end=None):
    """Execute descriptors [begin, end) (as recorded by a PlanBase on a CPU Ctx)."""
    mem = Mem(_flatten(keep))
    for d, _tag in descs[begin:end]:
        RUNNERS[type(d)](d, mem)


def _flatten(x):
    out = []
    for it in x:
        if isinstance(it, (list, tuple)):
            out.extend(_flatten(it))
        else:
            out.append(it)
    return out
