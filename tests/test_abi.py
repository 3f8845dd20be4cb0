"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/pdse.h
declares; descriptor layouts of the ctypes binding match the compiled structs; argument
errors are reported through the status/last_error convention without touching a device."""
import os
import re

import pytest

from conftest import ROOT, pkg


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    return pkg("_lib")


def test_header_symbols_all_exported(lib):
    text = open(os.path.join(ROOT, "include", "pdse.h")).read()
    declared = set(re.findall(r"\b(pdse_[a-z0-9_]+)\s*\(", text))
    declared -= {"pdse_plan"}
    assert len(declared) >= 20
    handle = lib.load()
    for name in sorted(declared):
        assert hasattr(handle, name), name
    assert set(lib.EXPORTS) == declared


def test_descriptor_sizes_match(lib):
    import ctypes as C

    handle = lib.load()
    for kind, typ in lib.DESC_TYPES.items():
        assert handle.pdse_desc_size(kind) == C.sizeof(typ), typ.__name__
    assert handle.pdse_desc_size(99) == -1


def test_argument_errors_do_not_need_a_device(lib):
    with pytest.raises(lib.PdseError, match="null"):
        lib.launch(lib.GconvDesc())
    d = lib.LstmDesc()
    with pytest.raises(lib.PdseError, match="lstm"):
        lib.launch(d)
    with pytest.raises(lib.PdseError, match="bglu: null"):
        lib.launch(lib.BgluDesc())
    with pytest.raises(lib.PdseError, match="planes"):
        lib.launch(lib.PlanesDesc())
    with pytest.raises(lib.PdseError, match="glstmp"):
        lib.launch(lib.GlstmpDesc())
    with pytest.raises(lib.PdseError, match="tcm2s"):
        lib.launch(lib.Tcm2sDesc())
    assert lib.load().pdse_bglu_set_form(7) == -2 and lib.load().pdse_bglu_set_form(0) in (-1, 0)      # the product library holds form 0 only
    # round 3 (ABI 5): channel-blocked sources are a feature of the korder-3 kernel; the split GRU is the fused H = 64 form
    import ctypes as C

    buf = (C.c_float * 64)()
    ptr = C.addressof(buf)
    g = lib.GconvDesc()
    g.in0 = lib.Src(ptr, 64, 8, 8, 8, 16, 0, 8, 0)
    g.out, g.w0, g.taps = ptr, ptr, ptr
    g.B, g.Tout, g.Fout, g.Cout, g.ntaps, g.out_cr, g.korder, g.ksteps = 1, 1, 1, 32, 1, 1, 1, 8
    with pytest.raises(lib.PdseError, match="channel-blocked"):
        lib.launch(g)
    r = lib.GruDesc()
    r.whh, r.bhh, r.y, r.gx = ptr, ptr, ptr, ptr
    r.B, r.T, r.F, r.H, r.axis, r.split = 1, 1, 1, 128, 1, 1
    with pytest.raises(lib.PdseError, match="split-bf16"):
        lib.launch(r)
    n = lib.LnDesc()
    n.in_, n.gamma, n.beta, n.out = ptr, ptr, ptr, ptr
    n.B, n.T, n.N, n.r, n.blk = 1, 1, 8, 4, 3
    with pytest.raises(lib.PdseError, match="blk"):
        lib.launch(n)
    p = lib.Plan()
    assert len(p) == 0
    p.add(lib.EwDesc())
    assert len(p) == 1
    with pytest.raises(lib.PdseError, match="ew"):
        p.run()            # validation fails before any launch


def test_product_refuses_cpu(lib):
    import argparse

    ns = argparse.Namespace
    args = ns(retrain=False, joint=True, draw=False, sigma=False, checkpoint="x", generated_wav="y")
    config = ns(model=ns(name="GCRN"), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt"))
    with pytest.raises(lib.PdseError):
        pkg("trainer").ComplexDDPMTrainer(args, config, device="cpu", prior_state_dict={}, ddpm_state_dict={})
    with pytest.raises(lib.PdseError):
        pkg("ops").DiffUNet1Op({}, "cpu")


def test_generated_block_schedule_is_current(tmp_path, monkeypatch):
    """csrc/bglu_sched.inc is generated (tools/gen_bglu_sched.py): the committed file must be what the generator writes, and
    every variant's schedule must place each mm of an accumulate chain in program order and each vector chunk behind its
    producers."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_bglu_sched", os.path.join(ROOT, "tools", "gen_bglu_sched.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed, committed_forms = open(gen.OUT).read(), open(gen.OUT_FORMS).read()
    monkeypatch.setattr(gen, "OUT", str(tmp_path / "sched.inc"))
    monkeypatch.setattr(gen, "OUT_FORMS", str(tmp_path / "sched_forms.inc"))
    gen.main()
    assert open(str(tmp_path / "sched.inc")).read() == committed
    assert open(str(tmp_path / "sched_forms.inc")).read() == committed_forms      # the -DBGLU_FORMS schedules (diagnostic builds)
    for name, var in gen.VARIANTS:
        M, V = gen.build(var)
        slots = gen.schedule(M, V)
        at = {}
        for i, (pm, pv) in enumerate(slots):
            for it in (pm, pv):
                if it is not None:
                    at[it] = i
        assert set(at) == set(M) | set(V), name
        for it, i in at.items():
            for dep in it.deps:
                gap = i - at[dep]
                assert gap >= (2 if (it.kind, dep.kind) == ("V", "M") else 1 if it.kind != dep.kind else 0), (name, it.name, dep.name)
