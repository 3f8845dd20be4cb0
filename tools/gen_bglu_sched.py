"""Generates prior-diffuse_amd/csrc/bglu_sched.inc: the hand-pipelined instruction order of one iteration of
csrc/bglu.hip's loop, per kernel variant.

Why a generator: on gfx950 a vector instruction only hides behind a matrix instruction of the SAME wave's stream (about
five of them per v_mfma_f32_32x32x16_bf16), and hipcc keeps source order - all MFMAs of the K loop, then the vector work
of the tail - so the interleaving has to be spelled out.  An iteration is cut into SLOTS: one slot = one "mm" (the six
MFMAs of one 32x32x16 fp32-equivalent product, 192 matrix-pipe cycles; one MFMA in bf16 mode) beside one chunk of vector
work (a split of eight values = 44 instructions, a third of the gate, ...) and the LDS reads of the NEXT slot's weight
fragments.  Slots are separated by sched_barrier(0); inside a slot a short sched_group_barrier pattern alternates one
MFMA with a few vector instructions.  (One sched_group_barrier pattern over the whole 4000-instruction iteration did not
finish compiling in 30 minutes.)

The order is a list schedule: mm items of the tail (they wait for their vector producers) have priority over the K loop's
(always ready), a vector chunk may follow its producing mm no earlier than two slots later (MFMA result latency) and feed
an mm one slot later.

    python tools/gen_bglu_sched.py          # rewrites csrc/bglu_sched.inc
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "prior-diffuse_amd", "csrc", "bglu_sched.inc")              # the product kernel's schedules (8 waves)
OUT_FORMS = os.path.join(ROOT, "prior-diffuse_amd", "csrc", "bglu_sched_forms.inc")  # the measured-and-not-kept forms (-DBGLU_FORMS builds)


class Item:
    def __init__(self, name, kind, code, frag=None, deps=(), prio=0, after=None):
        self.name, self.kind, self.code, self.frag, self.deps, self.prio = name, kind, code, frag, list(deps), prio
        self.slot = None
        self.after = after      # code emitted right after this item (requests of the tile after next)


def popc(m):
    return bin(m).count("1")


def build(variant):
    NT, P1, C2, NXN, IN4 = variant["NT"], variant["P1"], variant["C2"], variant["NXN"], variant["IN4"]
    pipe = variant.get("pipe", True)      # False: K loop and tail of the SAME tile, one after the other (8-wave workgroups)
    spread = variant.get("spread", False) # vector-memory instructions one or two at a time, spread over the slots, instead of bursts of 4-6
    dual = P1 != 0
    accn = "accn" if pipe else "acc"
    kprio = -1 if pipe else 1000
    items = []
    M, V = [], []

    def m(name, code, frag, deps=(), prio=0, after=None):
        it = Item(name, "M", code, frag, deps, prio, after)
        M.append(it)
        items.append(it)
        return it

    def v(name, code, deps=(), prio=0):
        it = Item(name, "V", code, None, deps, prio)
        V.append(it)
        items.append(it)
        return it

    # ---- K loop of the NEXT tile (accn), always ready
    if IN4:
        for q in range(3):
            vk = v("K.split%d" % q, "V_KSPLIT(%d);" % q, prio=2000 - q)
            m("K.L%d" % q, "MM(%s.L, kb[%d]);" % (accn, q), "GLI(%d)" % q, [vk], prio=kprio)
            last = m("K.R%d" % q, "MM(%s.R, kb[%d]);" % (accn, q), "GRI(%d)" % q, [vk], prio=kprio,
                     after="REQ_IN4();" if q == 2 else None)
    else:
        for tap in range(NT):
            mine = []
            for q in range(2):
                last = None
                mine.append(m("K.L%d%d" % (tap, q), "MM(%s.L, in.pl[%d][%d]);" % (accn, tap, q), "GL(%d,%d)" % (tap, q), prio=kprio))
                last = m("K.R%d%d" % (tap, q), "MM(%s.R, in.pl[%d][%d]);" % (accn, tap, q), "GR(%d,%d)" % (tap, q), prio=kprio)
                mine.append(last)
                if (P1 >> tap) & 1:
                    rk = popc(P1 & ((1 << tap) - 1))
                    mine.append(m("K.L1%d%d" % (tap, q), "MM(%s.L1, in.pl[%d][%d]);" % (accn, tap, q), "GL1(%d,%d)" % (rk, q), prio=kprio))
                    last = m("K.R1%d%d" % (tap, q), "MM(%s.R1, in.pl[%d][%d]);" % (accn, tap, q), "GR1(%d,%d)" % (rk, q), prio=kprio)
                    mine.append(last)
                if q == 1 and not spread:
                    if pipe:
                        last.after = "REQ(%d);" % tap
                    else:   # two taps ahead: the rest of this tile, then the first two taps of the next one
                        last.after = "REQ_CUR(%d);" % (tap + 2) if tap + 2 < NT else "REQ(%d);" % (tap + 2 - NT)
            if spread:
                # the six loads of the tap two ahead go out one (or two) at a time behind this tap's mm items: no wave ever
                # queues six 1 KB requests at once, and the memory pipeline sees a steady trickle instead of eight waves' bursts
                tgt = ("REQ_CUR1(%d" % (tap + 2)) if tap + 2 < NT else ("REQ1(%d" % (tap + 2 - NT))
                loads = ["%s,%d,%d);" % (tgt, q, pl) for q in range(2) for pl in range(3)]
                for k, code in enumerate(loads):
                    it = mine[k % len(mine)]
                    it.after = (it.after + " " if it.after else "") + code
    klast = last

    # ---- tails (current tile), phase A (and B)
    for ph, S in enumerate(("SA", "SB") if dual else ("SA",)):
        base = 50 - 5 * ph      # phase A slightly ahead of phase B
        if not pipe and ph == 0:
            take = v("take", "V_TAKE();", [klast], prio=3000)
        kdep = [] if pipe else [take]
        sl = [v("%s.sL%d" % (S, s), "V_SL(%s,%d);" % (S, s), kdep, prio=base + 40) for s in range(2)]
        sr = [v("%s.sR%d" % (S, s), "V_SR(%s,%d);" % (S, s), kdep, prio=base + 39) for s in range(2)]
        lc = [m("%s.lc%d" % (S, s), "MM(%s.mL, %s.lp[%d]);" % (S, S, s), "LCW(%d)" % s, [sl[s]], prio=base + 40) for s in range(2)]
        rc = [m("%s.rc%d" % (S, s), "MM(%s.mR, %s.rp[%d]);" % (S, S, s), "RCW(%d)" % s, [sr[s]], prio=base + 39) for s in range(2)]
        sg = [v("%s.SG%d" % (S, i), "V_SG(%s,%d,%d);" % (S, lo, hi), [lc[1], rc[1]], prio=base + 35)
              for i, (lo, hi) in enumerate(((0, 6), (6, 11), (11, 16)))]
        if C2 == 1:
            v("%s.dot" % S, "V_DOT(%s,%d);" % (S, ph), sg, prio=base + 30)
            continue
        spg = [v("%s.sG%d" % (S, s), "V_SPG(%s,%d);" % (S, s), [sg[1] if s == 0 else sg[2]], prio=base + 30) for s in range(2)]
        c2 = [[m("%s.c2%d%d" % (S, m2, s), "MM(%s.O%d, %s.gp[%d]);" % (S, m2, S, s), "C2W(%d,%d)" % (m2, s), [spg[s]],
                 prio=base + 30 - m2) for s in range(2)] for m2 in range(2)]
        sy = []
        for m2 in range(2):
            pr = v("%s.P%d" % (S, m2), "V_PR(%s,%d);" % (S, m2), [c2[m2][1]], prio=base + 25 - m2)
            sy.append([v("%s.sY%d%d" % (S, m2, s), "V_SY(%s,%d,%d);" % (S, m2, s), [pr], prio=base + 24 - m2) for s in range(2)])
        if NXN == 0:
            v("%s.keep" % S, "V_KEEP(%s);" % S, [sy[1][1]], prio=base + 10)    # the split chunks are dropped for NXN == 0 (see emit)
            continue
        zs = []
        if dual and spread:   # the addend's four 16-byte loads behind the four conv2 mm items, one each
            c2[0][0].after = "V_ZOFF(%s,%d); V_ZLD(%s,0);" % (S, ph, S)
            c2[0][1].after = "V_ZLD(%s,1);" % S
            c2[1][0].after = "V_ZLD(%s,2);" % S
            c2[1][1].after = "V_ZLD(%s,3);" % S
            zs = [c2[1][1]]
        elif dual:    # the fp32 addend of chained tile 0 (16 loads): requested when conv2 starts, four to five slots before its use
            zs = [v("%s.zseed" % S, "V_ZSEED(%s,%d);" % (S, ph), [c2[0][0]], prio=base + 28)]
        for i in range(NXN):
            nx = [m("%s.nx%d%d%d" % (S, i, m2, s), "MM(%s.Z%d, %s.yp[%d][%d]);" % (S, i, S, m2, s), "NXW(%d,%d,%d)" % (i, m2, s),
                    [sy[m2][s]] + (zs if i == 0 else []), prio=base + 20 - i) for m2 in range(2) for s in range(2)]
            if i == 0 and spread:   # each half's three plane stores right behind its split, and ahead of the other phase's remaining splits
                for s in range(2):
                    v("%s.sZ%d" % (S, s), "V_SZ(%s,%d); V_ST0Q(%s,%d,%d);" % (S, s, S, ph, s), [nx[3]], prio=base + 60)
            elif i == 0:
                sz = [v("%s.sZ%d" % (S, s), "V_SZ(%s,%d);" % (S, s), [nx[3]], prio=base + 10) for s in range(2)]
                v("%s.st0" % S, "V_ST0(%s,%d);" % (S, ph), sz, prio=base + 9)
            else:
                v("%s.st%d" % (S, i), "V_STSKIP(%s,%d);" % (S, i), [nx[3]], prio=base + 8)
    if NXN == 0 and C2 == 64:
        # no chained tile: the block output is not split
        for it in list(V):
            if ".sY" in it.name:
                V.remove(it)
                items.remove(it)
        for it in V:
            it.deps = [d for d in it.deps if d in items] or ([x for x in V if x.name.endswith(".P1") and x.name[:2] == it.name[:2]] if it.name.endswith(".keep") else [])
    return M, V


def schedule(M, V):
    """One mm and at most one vector chunk per slot."""
    slots = []
    done_m, done_v = {}, {}
    pend_m, pend_v = list(M), list(V)
    s = 0
    while pend_m or pend_v:
        def ready_m(it):
            for d in it.deps:
                if d.kind == "V" and (d not in done_v or done_v[d] >= s):
                    return False
                if d.kind == "M" and (d not in done_m or done_m[d] >= s):
                    return False
            # accumulate chains keep program order: an earlier mm into the same accumulator must be placed
            idx = M.index(it)
            acc = it.code.split(",")[0]
            for e in M[:idx]:
                if e.code.split(",")[0] == acc and e not in done_m:
                    return False
            return True

        def ready_v(it):
            for d in it.deps:
                if d.kind == "M" and (d not in done_m or done_m[d] > s - 2):
                    return False
                if d.kind == "V" and (d not in done_v or done_v[d] >= s):
                    return False
            return True

        cm = [it for it in pend_m if ready_m(it)]
        cv = [it for it in pend_v if ready_v(it)]
        pm = max(cm, key=lambda it: (it.prio, -M.index(it))) if cm else None
        pv = max(cv, key=lambda it: (it.prio, -V.index(it))) if cv else None
        if pm is not None:
            pend_m.remove(pm)
            done_m[pm] = s
        if pv is not None:
            pend_v.remove(pv)
            done_v[pv] = s
        slots.append((pm, pv))
        s += 1
        if s > 400:
            raise RuntimeError("schedule does not converge")
    return slots


def emit(name, variant, f):
    M, V = build(variant)
    slots = schedule(M, V)
    f.write("#if BGLU_SCHED == %d   // %s: %d mm, %d vector chunks, %d slots\n" % (variant["id"], name, len(M), len(V), len(slots)))
    # fragment double buffer: slot s uses buffer s & 1 (only slots that hold an mm count), the next mm's fragments are
    # read one slot ahead
    mm_slots = [i for i, (pm, _) in enumerate(slots) if pm is not None]
    first = slots[mm_slots[0]][0]
    f.write("  FRAG(0, %s);\n" % first.frag)
    buf = 0
    for i, (pm, pv) in enumerate(slots):
        f.write("  SLOT_BEGIN  /* %d: %s | %s */\n" % (i, pm.name if pm else "-", pv.name if pv else "-"))
        if pm is not None:
            # the NEXT mm's fragments first, fenced: they are in flight during this slot's six MFMAs
            nxt = [j for j in mm_slots if j > i]
            if nxt:
                f.write("    FRAG(%d, %s);\n    FRAG_FENCE\n" % (1 - buf, slots[nxt[0]][0].frag))
        if pv is not None:
            f.write("    %s\n" % pv.code)
        if pm is not None:
            f.write("    %s\n" % pm.code.replace("MM(", "MM(%d, " % buf))
            if pm.after:
                f.write("    %s\n" % pm.after)
            buf = 1 - buf
        f.write("  SLOT_END(%d, %d)\n" % (1 if pm is not None else 0, 1 if pv is not None else 0))
    f.write("#endif\n\n")
    return slots


VARIANTS0 = [
    ("decoder stages 5..2: dual phase, 4 taps, C2 = 64, one chained tile", dict(id=1, NT=4, P1=5, C2=64, NXN=1, IN4=False)),
    ("last decoder stage: dual phase, 6 taps, C2 = 1", dict(id=2, NT=6, P1=27, C2=1, NXN=0, IN4=False)),
    ("encoder stages 2..4: 6 taps, C2 = 64, three chained tiles", dict(id=3, NT=6, P1=0, C2=64, NXN=3, IN4=False)),
    ("encoder stage 5: 6 taps, C2 = 64, block output kept", dict(id=4, NT=6, P1=0, C2=64, NXN=0, IN4=False)),
    ("encoder stage 1: fp32 inputs, K = 40, three chained tiles", dict(id=5, NT=10, P1=0, C2=64, NXN=3, IN4=True)),
]


VARIANTS = (VARIANTS0 + [(name + " - K loop, then tail (8 waves)", dict(var, id=var["id"] + 10, pipe=False)) for name, var in VARIANTS0] +
            [(name + " - K loop, then tail (8 waves), vector-memory instructions spread over the slots",
              dict(var, id=var["id"] + 20, pipe=False, spread=True)) for name, var in VARIANTS0 if not var["IN4"]])


def main():
    head = ("// GENERATED by tools/gen_bglu_sched.py - do not edit.  One iteration of bglu_kernel's loop as a sequence of slots\n"
            "// (one mm beside one chunk of vector work); vocabulary: csrc/bglu.hip.\n\n")
    with open(OUT, "w") as f, open(OUT_FORMS, "w") as g:
        f.write(head)
        g.write(head.replace("do not edit.", "do not edit.  Forms that were measured and are not the product kernel (4 waves software-\n"
                                             "// pipelined, round 3; vector-memory instructions spread over the slots, round 4): compiled with -DBGLU_FORMS only."))
        for name, var in VARIANTS:
            product = not var.get("pipe", True) and not var.get("spread", False)
            slots = emit(name, var, f if product else g)
            print("%-70s %3d slots" % (name, len(slots)))
            for i, (pm, pv) in enumerate(slots):
                print("   %3d  %-12s %-12s" % (i, pm.name if pm else "-", pv.name if pv else "-"))


if __name__ == "__main__":
    main()
