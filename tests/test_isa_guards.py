"""CPU: guards on the generated gfx950 ISA that hipcc cannot give (cdna_hip_programming.md 5.7: the result of an inline-asm load
counts as written when the statement ends, so the compiler may copy, spill or re-use its destination registers while the data is
still in flight).  csrc/gconv4.hip issues its gathers (global_load_dword[x4]) and its ring reads (ds_read_b128) as inline asm and
waits for them later; this test disassembles the kernel and checks that no instruction names such a register between the request
and the wait that covers it.  A toolchain or flag change that breaks the register coalescing the kernel relies on fails HERE, not
as a memory fault on the GPU box (round 3 met one)."""
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
