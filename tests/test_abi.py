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
