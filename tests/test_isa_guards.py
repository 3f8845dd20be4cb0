"""CPU: guards on the generated gfx950 ISA that hipcc cannot give (cdna_hip_programming.md 5.7: the result of an inline-asm load
counts as written when the statement ends, so the compiler may copy, spill or re-use its destination registers while the data is
still in flight).  csrc/gconv4.hip issues its gathers (global_load_dword[x4]) and its ring reads (ds_read_b128) as inline asm and
waits for them later; this test disassembles the kernel and checks that no instruction names such a register between the request
and the wait that covers it.  A toolchain or flag change that breaks the register coalescing the kernel relies on fails HERE, not
as a memory fault on the GPU box (round 3 met one).

Round 4 met a second kind: a 16-byte buffer store WITH a scalar offset whose data registers the very next instruction (a
v_pk_mul_f32) overwrote - dword 1 of the stored value was the new one.  LLVM's hazard recogniser pads that case only for stores
without a scalar offset register.  scan_store_overwrite() reads every block-kernel translation unit for an instruction that
writes a wide store's data registers within two issue slots of it."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "prior-diffuse_amd", "csrc")


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


def scan(asm_text):
    """-> list of (kernel, line number, instruction, registers still in flight that it names)."""
    bad, in_asm, kernel = [], False, None
    flying = {}                                     # vgpr -> "lgkm" | "vm"
    for n, line in enumerate(asm_text.splitlines(), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, flying = m.group(1), {}
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or kernel is None:
            continue
        op = s.split()[0]
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in s:
                flying = {r: k for r, k in flying.items() if k != "lgkm"}
            if "vmcnt(0)" in s:
                flying = {r: k for r, k in flying.items() if k != "vm"}
            continue
        if op == "s_endpgm":
            kernel = None
            continue
        if in_asm and (op.startswith("ds_read") or op.startswith("global_load") or op.startswith("buffer_load")):
            dst = s.split(",")[0]
            for r in _regs(dst):
                flying[r] = "lgkm" if op.startswith("ds_read") else "vm"
            continue
        if in_asm:
            continue
        hit = _regs(s.split(";")[0]) & set(flying)
        if hit:
            bad.append((kernel, n, s, sorted(hit)))
    return bad


def test_scanner_sees_a_copy_of_an_inflight_register():
    text = "\n".join(["_Zk:", ";;#ASMSTART", "ds_read_b128 v[4:7], v1", ";;#ASMEND", "v_mov_b32_e32 v9, v5", ";;#ASMSTART",
                      "s_waitcnt lgkmcnt(0)", ";;#ASMEND", "v_mov_b32_e32 v9, v5", "s_endpgm"])
    bad = scan(text)
    assert len(bad) == 1 and bad[0][1] == 5 and bad[0][3] == [5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_gconv4_asm_loads_are_not_touched_in_flight(tmp_path):
    out = str(tmp_path / "gconv4.s")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-command-line-argument",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
                    os.path.join(CSRC, "gconv4.hip"), "-o", out], check=True)
    text = open(out).read()
    assert text.count("ds_read_b128") > 100 and text.count(";;#ASMSTART") > 500      # the kernels are in there
    bad = scan(text)
    assert not bad, "instructions that name a register an asm load is still writing:\n" + "\n".join(
        "%s:%d: %s  (in flight: v%s)" % (k[:60], n, s, r) for k, n, s, r in bad[:20])


def scan_store_overwrite(asm_text, window=2):
    """-> list of (kernel, store line, store, line, instruction, registers): instructions that write data registers of a 12/16-byte
    store within ``window`` issue slots behind it (s_nop n counts n + 1)."""
    bad, kernel, pend = [], None, []
    for n, line in enumerate(asm_text.splitlines(), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, pend = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith(".") or kernel is None or s.endswith(":"):
            continue
        code = s.split(";")[0].strip()
        op = code.split()[0]
        if op == "s_endpgm":
            kernel = None
            continue
        if pend:
            dest = set()
            if op.startswith(("v_", "ds_read", "buffer_load", "global_load", "scratch_load")) and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                rest = code.split(None, 1)
                dest = _regs(rest[1].split(",")[0]) if len(rest) > 1 else set()
            k = int(code.split()[1]) + 1 if op == "s_nop" else 1
            keep = []
            for r, rem, ln, ins in pend:
                if dest & r:
                    bad.append((kernel, ln, ins, n, code, sorted(dest & r)))
                if rem - k > 0:
                    keep.append((r, rem - k, ln, ins))
            pend = keep
        if re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]", op):
            ops = code.split(None, 1)[1].split(",")
            pend.append((_regs(ops[0] if op.startswith("buffer") else ops[1]), window, n, code))
    return bad


def test_store_scanner_sees_an_overwritten_store():
    text = "\n".join(["_Zk:", "buffer_store_dwordx4 v[16:19], v42, s[36:39], s15 offen", "v_pk_mul_f32 v[16:17], v[28:29], s[60:61]",
                      "buffer_store_dwordx4 v[0:3], v42, s[36:39], s15 offen", "s_nop 1", "v_mov_b32_e32 v1, v5",
                      "buffer_store_dword v7, v42, s[36:39], 0 offen", "v_mov_b32_e32 v7, v5", "s_endpgm"])
    bad = scan_store_overwrite(text)
    assert len(bad) == 1 and bad[0][1] == 2 and bad[0][5] == [16, 17]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_wide_stores_are_not_overwritten_behind_their_issue(tmp_path):
    """Every translation unit whose kernels store 16 bytes per lane from temporaries (csrc/gconv_common.h: bstore16 and friends)."""
    from concurrent.futures import ThreadPoolExecutor

    units = ["bglu.hip", "tcm2.hip", "dense.hip", "gconv4.hip"]

    def cc(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-command-line-argument",
                        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        return open(out).read()

    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
        texts = list(ex.map(cc, units))
    assert sum(t.count("buffer_store_dwordx4") for t in texts) > 100
    bad = [b for t in texts for b in scan_store_overwrite(t)]
    assert not bad, "instructions that overwrite the data of a wide store right behind it:\n" + "\n".join(
        "%s:%d: %s -> %d: %s (v%s)" % (k[:60], ln, ins, n, c, r) for k, ln, ins, n, c, r in bad[:20])
