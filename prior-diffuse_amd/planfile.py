"""Plan files: a recorded plan together with everything it points at, for hosts that have neither Python nor the plan
builders (include/pdse.h: pdse_plan_load, pdse_prior_forward, pdse_eps_forward, pdse_enhance).

``save(path, descs, ctx, named)`` writes the descriptors (with the byte offsets of their pointer fields, taken from the ctypes
layouts of ``_lib``), and one region per device allocation the plan uses: packed weights and tables with their bytes, zero-
initialised buffers (margins, dump items, states) as a size, scratch buffers as a size.  ``named`` gives the regions a C caller
finds inputs and outputs by.  The loader allocates, uploads and rebases (csrc/capi.hip).

    python -m prior-diffuse_amd.planfile ...   is not provided: the weights come from a state_dict, so exporting is a Python call:
    planfile.save_eps_net(path, ddpm_state_dict, B, T)      -> regions x, x_init, t, out      (pdse_eps_forward)
    planfile.save_prior(path, "GCRN", prior_state_dict, B, T) -> regions x, out               (pdse_prior_forward)
    planfile.save_pipeline(path, SamplerPipeline)            -> regions wav, x_T, wav_out, spec (pdse_enhance)
"""
import ctypes as C
import struct

import torch

from . import _lib as L
from . import nets

MAGIC = b"PDSEPLN1"
SCRATCH, ZERO, DATA = 0, 1, 2
# element type of a region, carried in bits 8.. of the kind word for readers that want typed views (the C loader ignores it)
DTYPES = {torch.float32: 0, torch.int16: 1, torch.int32: 2, torch.int64: 3, torch.uint8: 4}


def pointer_offsets(typ, base=0):
    """Byte offsets of every device-pointer field (c_void_p, also inside nested structs and arrays) of a descriptor type."""
    out = []
    for name, ft in typ._fields_:
        off = base + getattr(typ, name).offset
        if ft is C.c_void_p:
            out.append(off)
        elif isinstance(ft, type) and issubclass(ft, C.Structure):
            out += pointer_offsets(ft, off)
        elif isinstance(ft, type) and issubclass(ft, C.Array):
            if ft._type_ is C.c_void_p:
                out += [off + 8 * i for i in range(ft._length_)]
            elif isinstance(ft._type_, type) and issubclass(ft._type_, C.Structure):
                for i in range(ft._length_):
                    out += pointer_offsets(ft._type_, off + i * C.sizeof(ft._type_))
    return out


def _pad8(n):
    return (8 - (n & 7)) & 7


def save(path, descs, ctx, named):
    """descs: [(descriptor, tag)] as recorded by a PlanBase; ctx: its nets.Ctx; named: {region name: tensor}."""
    tensors = [t for t in ctx.all_tensors() if torch.is_tensor(t) and t.numel() > 0]
    regions, seen = [], set()
    names = {t.untyped_storage().data_ptr(): n for n, t in named.items()}
    if len(names) != len(named):
        raise ValueError("two names for one buffer")
    for t in tensors:
        st = t.untyped_storage()
        base = st.data_ptr()
        if base in seen:
            continue
        seen.add(base)
        nbytes = st.nbytes()
        if t.data_ptr() in ctx.scratch and t.data_ptr() == base:
            kind, data = SCRATCH, None
        else:
            raw = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (nbytes,)).cpu().numpy()
            kind, data = (DATA, raw.tobytes()) if raw.any() else (ZERO, None)
        regions.append((base, nbytes, kind | (DTYPES[t.dtype] << 8), names.pop(base, ""), data))
    if names:
        raise ValueError("named tensors that the plan does not keep: %s" % sorted(names.values()))
    spans = sorted((b, b + n) for b, n, _, _, _ in regions)
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IIII", L.ABI_VERSION, len(regions), len(descs), 0))
        for base, nbytes, kind, name, data in regions:
            nb = name.encode()
            f.write(struct.pack("<QQII", base, nbytes, kind, len(nb)) + nb + b"\0" * _pad8(len(nb)))
            if kind & 0xff == DATA:
                f.write(data + b"\0" * _pad8(nbytes))
        for d, tag in descs:
            kind = L.KIND_OF[type(d)]
            offs = pointer_offsets(type(d))
            raw = bytes(d)
            for o in offs:                  # every non-null pointer must lie in a region the file carries
                v = struct.unpack_from("<Q", raw, o)[0]
                if v and not any(lo <= v < hi for lo, hi in spans):
                    raise ValueError("%s: pointer field at offset %d (0x%x) is outside every buffer the context keeps" % (
                        type(d).__name__, o, v))
            f.write(struct.pack("<iiII", kind, int(tag), len(raw), len(offs)))
            f.write(struct.pack("<%dI" % len(offs), *offs) + b"\0" * _pad8(4 * len(offs)))
            f.write(raw + b"\0" * _pad8(len(raw)))
    return path


def save_eps_net(path, ddpm_sd, B, T, device="cuda:0", nocon=False):
    """DiffUNet1.forward(x, x_init, t) (model/diff3.py:37-57) for one (B, T): regions x, x_init, t [B] fp32, out."""
    ctx = nets.Ctx(device)
    net = nets.EpsNetPlan(ctx, ddpm_sd, B, T, time_cond=True, nsteps=1, with_pre=not nocon)
    net.build_time()
    net.build_step(0)
    net.finish()
    named = {"x": net.x, "t": net.tsteps, "out": net.out}
    if not nocon:
        named["x_init"] = net.x_init
    return save(path, net.descs, ctx, named)


def save_prior(path, prior_name, prior_sd, B, T, device="cuda:0"):
    """self.model(feat) -> X_init (trainer/complex_ddpm_trainer.py:941) for one (B, T): regions x, out."""
    ctx = nets.Ctx(device)
    if prior_name == "GCRN":
        net = nets.GcrnPlan(ctx, prior_sd, B, T)
        net.build()
    elif prior_name == "DiffUNet":
        net = nets.EpsNetPlan(ctx, prior_sd, B, T, time_cond=False)
        net.build_step(0)
    else:
        net = {"aia_complex_trans_ri": nets.AiaPlan, "dual_aia_trans_merge_crm": nets.DualAiaPlan}[prior_name](ctx, prior_sd, B, T)
        net.build()
    net.finish()
    return save(path, net.descs, ctx, {"x": net.x, "out": net.out})


def save_pipeline(path, pipe):
    """The whole path of one SamplerPipeline built with the signal front / back end (wav -> ... -> wav): regions wav [B,L],
    x_T [B,2,T,161], wav_out [B,L], spec [B,2,T,161].  The true lengths are stored as the pipeline holds them (full length)."""
    if pipe.stft is None:
        raise ValueError("the pipeline was built without the signal front / back end (pass L_)")
    pipe.stft.lens.fill_(pipe.L)
    return save(path, pipe.descs, pipe.ctx, {"wav": pipe.stft.wav, "x_T": pipe.xT_in, "wav_out": pipe.istft.wav, "spec": pipe.spec})
