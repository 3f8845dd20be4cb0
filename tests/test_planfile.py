"""CPU: the plan-file writer (prior-diffuse_amd/planfile.py) against a Python reading of the format csrc/capi.hip loads: every
pointer field of every descriptor is listed, the file carries every buffer a descriptor points into, and a plan rebased into
fresh buffers - exactly what pdse_plan_load does - computes the oracle's DiffUNet1 output on the descriptor interpreter."""
import ctypes as C
import struct

import numpy as np
import torch

import emu
from conftest import pkg, rel_l2, seeded
from oracle import restate as R


def _load(path, L):
    """-> ([(descriptor, tag)], [tensors], {name: tensor}): the file rebased into new CPU buffers."""
    raw = open(path, "rb").read()
    assert raw[:8] == b"PDSEPLN1"
    abi, nreg, nops, _ = struct.unpack_from("<IIII", raw, 8)
    assert abi == L.ABI_VERSION
    pos, regions = 24, []
    for _ in range(nreg):
        base, nbytes, kind, nlen = struct.unpack_from("<QQII", raw, pos)
        pos += 24
        name = raw[pos:pos + nlen].decode()
        pos += nlen + ((8 - (nlen & 7)) & 7)
        dt = [torch.float32, torch.int16, torch.int32, torch.int64, torch.uint8][kind >> 8]
        buf = torch.zeros(nbytes // dt.itemsize, dtype=dt)
        if kind & 0xff == 2:
            buf.view(torch.uint8)[:nbytes] = torch.frombuffer(bytearray(raw[pos:pos + nbytes]), dtype=torch.uint8)
            pos += nbytes + ((8 - (nbytes & 7)) & 7)
        regions.append((base, nbytes, name, buf))
    descs = []
    for _ in range(nops):
        kind, tag, dsz, nptr = struct.unpack_from("<iiII", raw, pos)
        pos += 16
        offs = struct.unpack_from("<%dI" % nptr, raw, pos)
        pos += 4 * nptr + ((8 - ((4 * nptr) & 7)) & 7)
        typ = L.DESC_TYPES[kind]
        assert C.sizeof(typ) == dsz
        body = bytearray(raw[pos:pos + dsz])
        pos += dsz + ((8 - (dsz & 7)) & 7)
        for o in offs:
            v = struct.unpack_from("<Q", body, o)[0]
            if v:
                hit = [(b, t) for b, n, _, t in regions if b <= v < b + n]
                assert len(hit) == 1
                struct.pack_into("<Q", body, o, hit[0][1].data_ptr() + (v - hit[0][0]))
        descs.append((typ.from_buffer_copy(bytes(body)), tag))
    assert pos == len(raw)
    return descs, [t for _, _, _, t in regions], {n: t for _, _, n, t in regions if n}


def test_pointer_offsets_cover_every_pointer_field():
    L, pf = pkg("_lib"), pkg("planfile")
    assert pf.pointer_offsets(L.EwDesc) == [0, 8, 16, 24]
    for typ in L.DESC_TYPES.values():
        offs = pf.pointer_offsets(typ)
        assert len(set(offs)) == len(offs) and all(o % 8 == 0 and o + 8 <= C.sizeof(typ) for o in offs), typ.__name__
    # nested Src structs and pointer arrays
    assert len(pf.pointer_offsets(L.GconvDesc)) == 2 + 25 + 9
    assert L.BgluDesc.nx_out.offset in pf.pointer_offsets(L.BgluDesc) and L.BgluDesc.nx_out.offset + 8 in pf.pointer_offsets(L.BgluDesc)


def test_saved_eps_net_plan_rebased_into_new_buffers_vs_oracle(weights, tmp_path):
    L, pf = pkg("_lib"), pkg("planfile")
    B, T = 1, 6
    path = pf.save_eps_net(str(tmp_path / "eps.plan"), weights("DiffUNet1"), B, T, device="cpu")
    descs, tensors, named = _load(path, L)
    assert set(named) == {"x", "x_init", "t", "out"}
    x, xi = seeded((B, 2, T, 161), 3), seeded((B, 2, T, 161), 4) * 0.3
    t = torch.tensor([7.25])
    named["x"][:x.numel()] = x.flatten()
    named["x_init"][:xi.numel()] = xi.flatten()
    named["t"][:B] = t
    emu.run(descs, tensors)
    with torch.no_grad():
        ref = R.diffunet1_forward(weights("DiffUNet1"), x, xi, t)
    assert rel_l2(named["out"][:ref.numel()].view_as(ref), ref) < 2e-5
    # a file the loader must refuse: a pointer outside every region
    d = L.EwDesc()
    d.a = 12345
    try:
        pf.save(str(tmp_path / "bad.plan"), [(d, 0)], pkg("nets").Ctx("cpu"), {})
        raise AssertionError("saved a plan with a dangling pointer")
    except ValueError:
        pass
