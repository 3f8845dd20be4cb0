"""Builds a DIAGNOSTIC libpdse (never the product library) with extra compile-time switches, e.g. the block-kernel forms that were
measured and not kept:

    python tools/build_diag.py -o /tmp/libpdse_forms.so -DBGLU_FORMS
    PDSE_LIB=/tmp/libpdse_forms.so BGLU_FORM=3 python tools/time_bglu.py

(-DPDSE_DIAG: trace / mask environment hooks; -DBGLU_DIAG -DBGLU_SLOTSTAMP, -DBGLU_NO_MM ...: README.md.)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    args = sys.argv[1:]
    out = args[args.index("-o") + 1]
    extra = [a for a in args if a.startswith("-D")]
    bdir = out + ".build"
    os.makedirs(bdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include"), "-I" + ge.CSRC] + extra

    def cc(src):
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        subprocess.run([hipcc] + flags + ["-c", os.path.join(ge.CSRC, src), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(ge.SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, ge.SOURCES))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print("built", out, "with", " ".join(extra))


if __name__ == "__main__":
    main()
