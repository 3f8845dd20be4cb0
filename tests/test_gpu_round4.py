"""GPU parity, round 4: the persistent small-batch form of the GCRN prior's grouped LSTM (csrc/lstmp.hip,
pdse_glstmp_desc; reference model/gcrn.py:6-40) against the reference's goldens and against the layer wavefront it
replaces for B <= 8; everything through the C-ABI."""
import numpy as np
import pytest
import torch

from conftest import golden, pkg, rel_l2, seeded

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge

    ge.build()
    lib = pkg("_lib")
    lib.load()
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return lib


def _gcrn(weights, persist, monkeypatch):
    nets = pkg("nets")
    monkeypatch.setattr(nets.GcrnPlan, "persist_lstm", persist)
    return pkg("ops").GCRNOp(weights("GCRN"), DEV, exclusive=True)


@pytest.mark.parametrize("persist", [True, False])
def test_gcrn_golden_both_lstm_forms(L, weights, persist, monkeypatch):
    """The reference's own GCRN outputs (tests/golden/gcrn_small.npz: e5, the LSTM block's output, the network output;
    gcrn_t401: rows of a 4 s utterance) with the LSTM as one persistent launch and as the layer wavefront."""
    g = golden("gcrn_small")
    op = _gcrn(weights, persist, monkeypatch)
    out = op(seeded((2, 2, 20, 161), g["seed_x"]).to(DEV))
    net = op._plans[(2, 20)]
    assert net.persist == persist
    assert sum(1 for d, _ in net.descs if isinstance(d, L.GlstmpDesc)) == (1 if persist else 0)
    torch.cuda.synchronize()
    assert rel_l2(net.glstm_out().cpu(), g["glstm"]) < 2e-5
    assert rel_l2(out.cpu(), g["out"]) < 2e-5
    g = golden("gcrn_t401")
    out = op(seeded((1, 2, 401, 161), g["seed_x"]).to(DEV)).cpu()
    assert rel_l2(out[0, :, ::16, :], g["rows"]) < 5e-5
    if persist:
        assert int(op._plans[(1, 401)].status[0].item()) == 0


@pytest.mark.parametrize("B,T", [(1, 50), (2, 33), (3, 50), (4, 17), (5, 50), (8, 50), (1, 1001), (8, 401)])   # 5, 8: above the default limit
def test_persistent_lstm_equals_wavefront(L, weights, B, T, monkeypatch):
    """Every batch size the persistent form takes (granule widths 1, 2, 4, 8 and their padded items), short and long
    utterances: the LSTM block's output and the prior's output against the wavefront kernels (fp32 both, different
    summation order: 1e-5), a second run of the same plan bit for bit (the granule ring is re-initialised by every launch),
    and utterance b of a batch bit for bit equal to the same utterance run alone (batch invariance)."""
    x = seeded((B, 2, T, 161), 40 + B).to(DEV)
    monkeypatch.setattr(pkg("nets").GcrnPlan, "PERSIST_MAX_B", 8)      # the kernel takes B <= 8; the plan builder picks it up to the measured break-even
    op_w = _gcrn(weights, False, monkeypatch)
    ref = op_w(x).clone()
    ref_l = op_w._plans[(B, T)].glstm_out().clone()
    op_p = _gcrn(weights, True, monkeypatch)
    out = op_p(x).clone()
    net = op_p._plans[(B, T)]
    assert net.persist
    got_l = net.glstm_out().clone()
    torch.cuda.synchronize()
    assert int(net.status[0].item()) == 0
    assert rel_l2(got_l.cpu(), ref_l.cpu()) < 1e-5
    assert rel_l2(out.cpu(), ref.cpu()) < 1e-5
    out2 = op_p(x)
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and int(net.status[0].item()) == 0
    if B > 1 and T <= 50:
        b = B - 1
        alone = op_p(x[b:b + 1].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(alone[0], out[b])


def test_persistent_lstm_rejects_what_it_cannot_run(L):
    d = L.GlstmpDesc()
    with pytest.raises(L.PdseError, match="glstmp: null"):
        L.launch(d)
    buf = torch.zeros(64, device=DEV)
    for f in ("gx1", "w1", "w2i", "w2h", "r2", "c2", "gran", "status", "y"):
        setattr(d, f, buf.data_ptr())
    d.B, d.Bp, d.T, d.H, d.G = 9, 32, 4, 512, 2
    with pytest.raises(L.PdseError, match="glstmp: bad sizes"):
        L.launch(d)
