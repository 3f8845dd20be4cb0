"""Performance-only harness of csrc/bglu.hip on synthetic operands (random planes / weights; results are not checked):
python tools/time_bglu.py   -> microseconds per launch of each decoder / encoder geometry at B=32, T=401."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
L = importlib.import_module("prior-diffuse_amd._lib")
DEV = "cuda:0"


ZERO = bool(os.environ.get("BGLU_ZERO"))     # all-zero operands: same instructions, no toggling - how much of a launch is the clock the chip holds under load


def rnd16(*shape):
    if ZERO:
        return torch.zeros(*shape, device=DEV, dtype=torch.int16)
    return (torch.randn(*shape, device=DEV) * 0.1).to(torch.bfloat16).view(torch.int16)


def bench(name, NT, p1mask, C2, nx_n, Fin, Fout, Fout1, sf_in, taps, NP=3, B=32, T=401):
    d = L.BgluDesc()
    Tp, Fp = T + 1, Fin + 4
    hp = rnd16(B, Tp, 4, NP, Fp, 8)
    d.hp, d.hp_sb, d.hp_Tp, d.hp_Fp, d.hp_t0, d.hp_f0 = hp.data_ptr(), hp[0].numel(), Tp, Fp, 1, 2
    d.ntaps, d.sf_in = NT, sf_in
    d.hp_par = 1 if (sf_in == 2 and not os.environ.get("NOPAR")) else 0   # encoders: rows split by parity (synthetic data: timing only)
    for i, (dt, df) in enumerate(taps):
        d.tap_dt[i], d.tap_df[i] = dt, df
    d.p1mask, d.Fout1 = p1mask, Fout1
    d.B, d.Tout, d.Fout, d.np = B, T, Fout, NP
    nt1 = bin(p1mask).count("1")
    keep = [hp]

    def W(nb):
        w = rnd16(nb, NP, 64, 8)
        keep.append(w)
        return w.data_ptr()

    def Fv(n):
        v = torch.randn(n, device=DEV) * (0.0 if ZERO else 0.1)
        keep.append(v)
        return v.data_ptr()

    d.w0, d.w1 = W(2 * NT), W(2 * NT)
    if p1mask:
        d.w2, d.w3 = W(2 * nt1), W(2 * nt1)
    d.wlc, d.wrc = W(2), W(2)
    if C2 == 64:
        d.wc2 = W(4)
    else:
        d.wc2v = Fv(32)
    if nx_n:
        d.nx_w = W(4 * nx_n)
    d.bias0, d.bias1, d.bias_sb = Fv(32 * B), Fv(32 * B), 32
    d.blc, d.brc, d.bc2 = Fv(32), Fv(32), Fv(64)
    d.slope, d.C2, d.nx_n, d.nx_items = 0.25, C2, nx_n, B + 1
    Fo = 2 * Fout if p1mask else Fout
    if C2 == 1 or nx_n == 0:
        out = torch.empty(B, 64 if C2 == 64 else 1, T, Fo + 2, device=DEV)
        keep.append(out)
        d.out, d.out_sb, d.out_sc, d.out_st = out.data_ptr(), out[0].numel(), T * (Fo + 2), Fo + 2
        d.out_sf = 2 if p1mask else 1
    if nx_n:
        nFp = Fo + 4 + (1 if p1mask else 0)
        nhp = torch.zeros(B + 1, T + 1, 4, NP, nFp, 8, dtype=torch.int16, device=DEV)   # item B: dump target
        keep.append(nhp)
        d.nx_hp, d.nx_hp_sb, d.nx_Tp, d.nx_Fp, d.nx_t0, d.nx_f0 = nhp.data_ptr(), nhp[0].numel(), T + 1, nFp, 1, 2
        if p1mask:
            add = torch.randn(B, 32, T, Fo + 1, device=DEV) * (0.0 if ZERO else 1.0)
            keep.append(add)
            d.nx_add, d.add_sb, d.add_sc, d.add_st, d.add_sf = add.data_ptr(), add[0].numel(), 4 * T * (Fo + 1), 4 * (Fo + 1), 4   # [B][8][T][F][4]
        if not os.environ.get("NOPAR"):   # skip halves with their bins split by parity (both the encoder's stores and the decoder's loads)
            d.skip_Fh = (Fo + 2) // 2 if p1mask else (Fout + 1) // 2
        for i in range(nx_n):
            d.nx_bias[i], d.nx_bias_sb[i] = Fv(32 * B), 32
        for i in range(nx_n - 1):
            sk = torch.empty(B + 1, 32, T, Fout, device=DEV)
            keep.append(sk)
            d.nx_out[i], d.nx_sb[i], d.nx_sc[i], d.nx_st[i], d.nx_sf[i] = sk.data_ptr(), sk[0].numel(), 4 * T * Fout, 4 * Fout, 4
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.launch(d, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        L.launch(d, st)
    e1.record()
    torch.cuda.synchronize()
    print("%-34s np %d  %4d x %3d positions: %7.1f us" % (name, NP, T, Fout, e0.elapsed_time(e1) / n * 1e3), flush=True)


DEC = [(0, 0), (0, -1), (-1, 0), (-1, -1)]
DEC1 = [(0, 0), (0, -1), (0, -2), (-1, 0), (-1, -1), (-1, -2)]
ENC = [(-1, 0), (-1, 1), (-1, 2), (0, 0), (0, 1), (0, 2)]
if __name__ == "__main__":
    form = int(os.environ.get("BGLU_FORM", "-1"))       # pdse_bglu_set_form: 0 = 8 waves, 1 = 4 waves pipelined, 2 = 16 waves, 3 = 12 waves
    if L.load().pdse_bglu_set_form(form) == -2:
        sys.exit("this libpdse.so holds the product form only: build with -DBGLU_FORMS (HIPCC_FLAGS) to time the others")
    print("kernel form %d" % form, flush=True)
    for NP in (3, 1):
        for Fin in (4, 9, 19, 39):
            bench("decoder (4 taps, dual, nx 1)", 4, 5, 64, 1, Fin, Fin + 1, Fin, 1, DEC, NP)
        bench("last decoder (6 taps, dual, C2 1)", 6, 27, 1, 0, 79, 81, 80, 1, DEC1, NP)
        for Fin in (79, 39, 19):
            bench("encoder (6 taps, nx 3)", 6, 0, 64, 3, Fin, (Fin - 3) // 2 + 1, 0, 2, ENC, NP)
        bench("encoder 5 (6 taps, kept)", 6, 0, 64, 0, 9, 4, 0, 2, ENC, NP)
