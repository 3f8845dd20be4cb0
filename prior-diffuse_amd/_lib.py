"""ctypes binding of libpdse.so (the C-ABI declared in include/pdse.h).

The product path has no CPU fallback: if the shared library is missing or does not
match the header this module raises, and so does everything built on it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDSE_LIB") or os.path.join(_HERE, "libpdse.so")   # PDSE_LIB: diagnostic builds only

ABI_VERSION = 8

ACT_NONE, ACT_PRELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3
EPI_LINEAR, EPI_GLU, EPI_BIGLU = 0, 1, 2
EW_DIV, EW_UPDATE, EW_UPDATE_FINAL, EW_COPY, EW_ADD_MUL = 0, 1, 2, 3, 4
(OP_GCONV, OP_TIME, OP_EW, OP_COMPAND, OP_WAVPREP, OP_OLA, OP_SIGMA, OP_LN, OP_LSTM,
 OP_ROWLN, OP_CHLN, OP_ATTN, OP_GRU, OP_GNCOMB, OP_AHAM, OP_QSAMPLE, OP_TRANSPOSE, OP_TCM, OP_CRM, OP_GCRNLAST,
 OP_MASKLOSS, OP_GLSTM, OP_TCM2, OP_BGLU, OP_PLANES, OP_GLSTMP, OP_TCM2S, OP_DENSE, OP_ROWLNB) = range(29)
MASKLOSS_BLOCKS = 32

_fp = C.c_void_p  # device pointers travel as integers
_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float


class Src(C.Structure):
    _fields_ = [("ptr", _fp), ("sb", _i64), ("sc", _i64), ("st", _i64), ("sf", _i64),
                ("C", _i32), ("act", _i32), ("blk", _i32), ("pad_", _i32)]


class GconvDesc(C.Structure):
    _fields_ = [
        ("in0", Src), ("in1", Src),
        ("Tin", _i32), ("Fin", _i32),
        ("padrow", _fp), ("padrow_sb", _i64),
        ("taps", _fp), ("ntaps", _i32), ("sf_in", _i32),
        ("xf_scale0", _fp), ("xf_shift0", _fp), ("xf_scale1", _fp), ("xf_shift1", _fp),
        ("xf_slope0", _f32), ("xf_slope1", _f32), ("xf_mode", _i32), ("cin1", _i32),
        ("w0", _fp), ("w1", _fp), ("ksteps", _i32), ("Cout", _i32),
        ("bias0", _fp), ("bias1", _fp), ("bias0_sb", _i64), ("bias1_sb", _i64),
        ("epi", _i32), ("act", _i32), ("act_slope", _f32), ("C2", _i32),
        ("post_scale", _fp), ("post_shift", _fp),
        ("wlc", _fp), ("wrc", _fp), ("blc", _fp), ("brc", _fp), ("wc2", _fp), ("bc2", _fp),
        ("resid", _fp), ("out", _fp),
        ("out_sb", _i64), ("out_sc_hi", _i64), ("out_sc_lo", _i64), ("out_st", _i64),
        ("out_sf", _i64), ("out_off", _i64),
        ("out_cr", _i32), ("B", _i32), ("Tout", _i32), ("Fout", _i32),
        ("korder", _i32), ("ksteps1", _i32), ("w2", _fp), ("w3", _fp), ("p1mask", _i32), ("Fout1", _i32),
        ("nx_w", _fp), ("nx_bias", _fp * 3), ("nx_add", _fp * 3), ("nx_out", _fp * 3), ("nx_bias_sb", _i64 * 3),
        ("nx_sb", _i64 * 3), ("nx_sc", _i64 * 3), ("nx_st", _i64 * 3), ("nx_sf", _i64 * 3), ("nx_off", _i64 * 3),
        ("nx_n", _i32), ("nx_keep", _i32), ("nx_row0", _i32), ("wexp", _i32),
        ("bias0_t0", _fp), ("bias1_t0", _fp),
        ("tap_dt", _i32 * 12), ("tap_df", _i32 * 12),
    ]


class TimeDesc(C.Structure):
    _fields_ = [("t", _fp), ("table", _fp), ("p1T", _fp), ("b1", _fp), ("p2T", _fp), ("b2", _fp),
                ("wfT", _fp), ("bf", _fp), ("out", _fp), ("temb", _fp),
                ("B", _i32), ("NF", _i32), ("max_steps", _i32), ("pad_", _i32)]


class EwDesc(C.Structure):
    _fields_ = [("a", _fp), ("b", _fp), ("c", _fp), ("out", _fp), ("n", _i64),
                ("s0", _f32), ("s1", _f32), ("s2", _f32), ("op", _i32)]


class CompandDesc(C.Structure):
    _fields_ = [("in_", _fp), ("out", _fp), ("plane", _i64), ("B", _i32), ("mode", _i32),
                ("out_sb", _i64), ("out_sc", _i64), ("out_st", _i64), ("F", _i32), ("pad_", _i32)]


class WavprepDesc(C.Structure):
    _fields_ = [("wav", _fp), ("xpad", _fp), ("c", _fp), ("lens", _fp),
                ("B", _i32), ("L", _i32), ("pad", _i32), ("normalize", _i32)]


class OlaDesc(C.Structure):
    _fields_ = [("frames", _fp), ("win2", _fp), ("c", _fp), ("out", _fp),
                ("B", _i32), ("T", _i32), ("L", _i32), ("n_fft", _i32), ("hop", _i32), ("pad_", _i32)]


class SigmaDesc(C.Structure):
    _fields_ = [("init", _fp), ("a", _fp), ("out", _fp), ("maxbuf", _fp), ("plane", _i64),
                ("nplanes", _i32), ("pad_", _i32)]


class LnDesc(C.Structure):
    _fields_ = [("in_", _fp), ("gamma", _fp), ("beta", _fp), ("out", _fp),
                ("osb", _i64), ("os_hi", _i64), ("os_lo", _i64), ("os_t", _i64),
                ("B", _i32), ("T", _i32), ("N", _i32), ("r", _i32), ("eps", _f32), ("blk", _i32)]


class LstmDesc(C.Structure):
    _fields_ = [("gx", _fp), ("whh", _fp), ("hT", _fp), ("cst", _fp), ("y", _fp),
                ("y_sb", _i64), ("y_st", _i64), ("y_su", _i64), ("y_sg", _i64),
                ("B", _i32), ("Bp", _i32), ("T", _i32), ("H", _i32), ("G", _i32), ("pad_", _i32)]


class GlstmDesc(C.Structure):
    _fields_ = [("gx1", _fp), ("whh1", _fp), ("wih2", _fp), ("r2", _fp), ("c2", _fp), ("whh2", _fp),
                ("hT1", _fp), ("cst1", _fp), ("hT2", _fp), ("cst2", _fp), ("gx2", _fp), ("part", _fp), ("y", _fp),
                ("y_sb", _i64), ("y_st", _i64), ("y_su", _i64), ("y_sg", _i64),
                ("B", _i32), ("Bp", _i32), ("T", _i32), ("H", _i32), ("G", _i32), ("eps", _f32), ("slices", _i32), ("pad_", _i32)]


class GlstmpDesc(C.Structure):
    _fields_ = [("gx1", _fp), ("w1", _fp), ("w2i", _fp), ("w2h", _fp), ("r2", _fp), ("c2", _fp), ("gran", _fp), ("status", _fp),
                ("y", _fp), ("y_sb", _i64), ("y_st", _i64), ("y_su", _i64), ("y_sg", _i64),
                ("B", _i32), ("Bp", _i32), ("T", _i32), ("H", _i32), ("G", _i32), ("eps", _f32)]


class RowlnDesc(C.Structure):
    _fields_ = [("in_", _fp), ("gamma", _fp), ("beta", _fp), ("slope", _fp), ("out", _fp), ("out_sb", _i64),
                ("B", _i32), ("C", _i32), ("T", _i32), ("F", _i32), ("eps", _f32), ("pad_", _i32)]


class DenseDesc(C.Structure):
    _fields_ = [("D", _fp), ("w", _fp), ("bias", _fp), ("gamma", _fp), ("beta", _fp), ("slope", _fp),
                ("B", _i32), ("T", _i32), ("F", _i32), ("G", _i32), ("tpad", _i32), ("g_in", _i32), ("cin", _i32), ("g_out", _i32),
                ("dil", _i32), ("np", _i32), ("eps", _f32), ("wexp", _i32)]


class RowlnbDesc(C.Structure):
    _fields_ = [("in_", _fp), ("gamma", _fp), ("beta", _fp), ("slope", _fp), ("out", _fp),
                ("in_sb", _i64), ("in_sc", _i64), ("in_st", _i64), ("out_sb", _i64), ("out_sg", _i64), ("out_st", _i64),
                ("B", _i32), ("C", _i32), ("T", _i32), ("F", _i32), ("eps", _f32), ("pad_", _i32)]


class ChlnDesc(C.Structure):
    _fields_ = [("in_", _fp), ("gamma", _fp), ("beta", _fp), ("out", _fp), ("plane", _i64),
                ("B", _i32), ("C", _i32), ("eps", _f32), ("pad_", _i32)]


class AttnDesc(C.Structure):
    _fields_ = [("qkv", _fp), ("out", _fp), ("B", _i32), ("T", _i32), ("F", _i32), ("E", _i32),
                ("heads", _i32), ("axis", _i32), ("pad0_", _i32), ("pad1_", _i32)]


class GruDesc(C.Structure):
    _fields_ = [("gx", _fp), ("whh", _fp), ("bhh", _fp), ("y", _fp),
                ("B", _i32), ("T", _i32), ("F", _i32), ("H", _i32), ("axis", _i32), ("split", _i32),
                ("x", _fp), ("wih", _fp), ("bih", _fp)]


class GncombDesc(C.Structure):
    _fields_ = [("base", _fp), ("row", _fp), ("col", _fp), ("g_row", _fp), ("b_row", _fp), ("g_col", _fp),
                ("b_col", _fp), ("stats", _fp), ("out", _fp), ("plane", _i64), ("B", _i32), ("C", _i32),
                ("k1", _f32), ("k2", _f32), ("eps", _f32), ("pad_", _i32)]


class AhamDesc(C.Structure):
    _fields_ = [("x", _fp * 4), ("w", _fp), ("means", _fp), ("out", _fp), ("plane", _i64),
                ("B", _i32), ("C", _i32), ("bias", _f32), ("pad_", _i32)]


class QsampleDesc(C.Structure):
    _fields_ = [("label", _fp), ("init", _fp), ("noise", _fp), ("a", _fp), ("s", _fp), ("out", _fp),
                ("plane", _i64), ("B", _i32), ("mode", _i32)]


class MasklossDesc(C.Structure):
    _fields_ = [("esti", _fp), ("label", _fp), ("frames", _fp), ("partial", _fp), ("out", _fp),
                ("B", _i32), ("C", _i32), ("T", _i32), ("F", _i32)]


class TransposeDesc(C.Structure):
    _fields_ = [("in_", _fp), ("out", _fp), ("N", _i32), ("R", _i32), ("Cc", _i32), ("pad_", _i32)]


class TcmDesc(C.Structure):
    _fields_ = [("x", _fp), ("h", _fp), ("x_out", _fp), ("h_out", _fp), ("wbr", _fp), ("bmain", _fp), ("bmask", _fp),
                ("xf", _fp), ("wc2", _fp), ("bc2", _fp), ("xf2", _fp), ("wn1", _fp), ("bn1", _fp),
                ("slope_main", _f32), ("slope_mask", _f32), ("slope2", _f32), ("dil", _i32), ("B", _i32), ("T", _i32)]


class Tcm2Desc(C.Structure):
    _fields_ = [("x", _fp), ("x_out", _fp), ("hs", _fp), ("hs_out", _fp), ("wbr", _fp), ("wc2", _fp), ("wn1", _fp),
                ("par", _fp), ("slope2", _f32), ("slope_main_next", _f32), ("slope_mask_next", _f32),
                ("dil", _i32), ("B", _i32), ("T", _i32), ("mode", _i32), ("np", _i32), ("qexp", _i32 * 3)]


TCM2S_MAX = 20


class Tcm2sDesc(C.Structure):
    """The whole TCM stack as one launch (include/pdse.h: pdse_tcm2s_desc, csrc/tcm2.hip: tcm2s_kernel)."""
    _fields_ = [("blk", Tcm2Desc * TCM2S_MAX), ("flags", _fp), ("status", _fp), ("n", _i32), ("pad_", _i32)]


class BgluDesc(C.Structure):
    """BiConv(Trans)GLU block on plane tensors (include/pdse.h: pdse_bglu_desc, csrc/bglu.hip)."""
    _fields_ = [("hp", _fp), ("hp_sb", _i64), ("hp_Tp", _i32), ("hp_Fp", _i32), ("hp_t0", _i32), ("hp_f0", _i32),
                ("x0", Src), ("x1", Src), ("Tin", _i32), ("Fin", _i32), ("ntaps", _i32), ("sf_in", _i32),
                ("tap_dt", _i32 * 10), ("tap_df", _i32 * 10), ("p1mask", _i32), ("Fout1", _i32),
                ("B", _i32), ("Tout", _i32), ("Fout", _i32), ("np", _i32),
                ("w0", _fp), ("w1", _fp), ("w2", _fp), ("w3", _fp), ("wlc", _fp), ("wrc", _fp), ("wc2", _fp), ("wc2v", _fp),
                ("nx_w", _fp), ("bias0", _fp), ("bias1", _fp), ("bias0_t0", _fp), ("bias1_t0", _fp), ("bias_sb", _i64),
                ("blc", _fp), ("brc", _fp), ("bc2", _fp), ("slope", _f32), ("C2", _i32),
                ("out", _fp), ("out_sb", _i64), ("out_sc", _i64), ("out_st", _i64), ("out_sf", _i64), ("out_off", _i64),
                ("hp_par", _i32), ("nx_n", _i32),
                ("nx_hp", _fp), ("nx_hp_sb", _i64), ("nx_Tp", _i32), ("nx_Fp", _i32), ("nx_t0", _i32), ("nx_f0", _i32),
                ("nx_row0", _i32), ("nx_par", _i32),
                ("nx_add", _fp), ("add_sb", _i64), ("add_sc", _i64), ("add_st", _i64), ("add_sf", _i64),
                ("nx_out", _fp * 2), ("nx_sb", _i64 * 2), ("nx_sc", _i64 * 2), ("nx_st", _i64 * 2), ("nx_sf", _i64 * 2),
                ("nx_bias", _fp * 3), ("nx_bias_sb", _i64 * 3), ("skip_Fh", _i32), ("nx_items", _i32), ("qexp", _i32 * 4)]


class PlanesDesc(C.Structure):
    _fields_ = [("in_", _fp), ("in_sb", _i64), ("in_sc", _i64), ("in_st", _i64), ("in_sf", _i64),
                ("hp", _fp), ("hp_sb", _i64), ("hp_Tp", _i32), ("hp_Fp", _i32), ("hp_t0", _i32), ("hp_f0", _i32),
                ("B", _i32), ("T", _i32), ("F", _i32), ("np", _i32),
                ("in1", _fp), ("w", _fp * 2), ("bias", _fp * 2), ("bias_sb", _i64), ("hp1", _fp),
                ("cin0", _i32), ("cin1", _i32), ("nd", _i32), ("pad_", _i32)]


class CrmDesc(C.Structure):
    _fields_ = [("x", _fp), ("o", _fp), ("ri", _fp), ("out", _fp), ("a1", _f32), ("b1", _f32), ("a2", _f32), ("b2", _f32),
                ("a3", _f32), ("b3", _f32), ("plane", _i32), ("B", _i32), ("mode", _i32), ("pad_", _i32)]


class GcrnLastDesc(C.Structure):
    _fields_ = [("in0", _fp), ("in1", _fp), ("w1", _fp), ("w2", _fp), ("fcT", _fp), ("fcb", _fp), ("out", _fp),
                ("out_sb", _i64), ("b1", _f32), ("b2", _f32), ("bn_scale", _f32), ("bn_shift", _f32), ("B", _i32), ("T", _i32),
                ("fcp", _fp)]


DESC_TYPES = {OP_DENSE: DenseDesc, OP_ROWLNB: RowlnbDesc, OP_TCM2S: Tcm2sDesc, OP_GLSTMP: GlstmpDesc, OP_BGLU: BgluDesc, OP_PLANES: PlanesDesc, OP_TCM2: Tcm2Desc, OP_GLSTM: GlstmDesc, OP_MASKLOSS: MasklossDesc, OP_GCRNLAST: GcrnLastDesc, OP_CRM: CrmDesc, OP_TCM: TcmDesc, OP_TRANSPOSE: TransposeDesc, OP_QSAMPLE: QsampleDesc, OP_ROWLN: RowlnDesc, OP_CHLN: ChlnDesc, OP_ATTN: AttnDesc, OP_GRU: GruDesc, OP_GNCOMB: GncombDesc,
              OP_AHAM: AhamDesc, OP_GCONV: GconvDesc, OP_TIME: TimeDesc, OP_EW: EwDesc, OP_COMPAND: CompandDesc,
              OP_WAVPREP: WavprepDesc, OP_OLA: OlaDesc, OP_SIGMA: SigmaDesc, OP_LN: LnDesc,
              OP_LSTM: LstmDesc}
KIND_OF = {v: k for k, v in DESC_TYPES.items()}

EXPORTS = [
    "pdse_abi_version", "pdse_last_error", "pdse_desc_size",
    "pdse_gconv_f32", "pdse_time_embed_f32", "pdse_ew_f32", "pdse_compand_f32", "pdse_wavprep_f32",
    "pdse_ola_f32", "pdse_sigma_mask_f32", "pdse_layernorm_f32", "pdse_lstm_f32",
    "pdse_rowln_prelu_f32", "pdse_chln_f32", "pdse_attention_f32", "pdse_bigru_f32", "pdse_gn_combine_f32",
    "pdse_aham_f32", "pdse_qsample_f32", "pdse_transpose_f32", "pdse_tcm_f32", "pdse_crm_f32", "pdse_gcrnlast_f32",
    "pdse_masked_mse_f32", "pdse_glstm_f32", "pdse_glstm_persistent_f32", "pdse_tcm2_bf16x3", "pdse_tcm2_stack_bf16x3", "pdse_bglu_planes", "pdse_split_planes", "pdse_dense_layer_bf16x3", "pdse_rowln_blocked_f32", "pdse_bglu_set_form",
    "pdse_plan_create", "pdse_plan_add", "pdse_plan_size", "pdse_plan_set_device", "pdse_plan_clear", "pdse_plan_run",
    "pdse_plan_run_range",
    "pdse_plan_build_graph", "pdse_plan_launch_graph", "pdse_plan_time_ops", "pdse_plan_time_tag",
    "pdse_plan_destroy", "pdse_plan_load", "pdse_plan_region", "pdse_prior_forward", "pdse_eps_forward", "pdse_enhance",
]

_DIRECT = {OP_GCONV: "pdse_gconv_f32", OP_TIME: "pdse_time_embed_f32", OP_EW: "pdse_ew_f32",
           OP_COMPAND: "pdse_compand_f32", OP_WAVPREP: "pdse_wavprep_f32", OP_OLA: "pdse_ola_f32",
           OP_SIGMA: "pdse_sigma_mask_f32", OP_LN: "pdse_layernorm_f32", OP_LSTM: "pdse_lstm_f32",
           OP_ROWLN: "pdse_rowln_prelu_f32", OP_CHLN: "pdse_chln_f32", OP_ATTN: "pdse_attention_f32",
           OP_GRU: "pdse_bigru_f32", OP_GNCOMB: "pdse_gn_combine_f32", OP_AHAM: "pdse_aham_f32",
           OP_QSAMPLE: "pdse_qsample_f32", OP_TRANSPOSE: "pdse_transpose_f32", OP_TCM: "pdse_tcm_f32", OP_CRM: "pdse_crm_f32", OP_GCRNLAST: "pdse_gcrnlast_f32",
           OP_MASKLOSS: "pdse_masked_mse_f32", OP_GLSTM: "pdse_glstm_f32", OP_GLSTMP: "pdse_glstm_persistent_f32",
           OP_TCM2: "pdse_tcm2_bf16x3", OP_TCM2S: "pdse_tcm2_stack_bf16x3", OP_BGLU: "pdse_bglu_planes", OP_PLANES: "pdse_split_planes",
           OP_DENSE: "pdse_dense_layer_bf16x3", OP_ROWLNB: "pdse_rowln_blocked_f32"}


class PdseError(RuntimeError):
    """Non-zero status from libpdse.so (the reference raises Python exceptions only)."""


class PdseRangeError(PdseError):
    """A pass on f16x2 operands produced non-finite values: an activation left the fp16 window of include/pdse.h
    (PDSE_F16_ACT_EXP; its planes became infinities).  The same pass on the three-plane bf16 split has no such window."""


_lib = None


def load():
    """Load libpdse.so once; verify ABI version and descriptor sizes."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (soname libamdhip64.so.7, the soname libpdse.so needs):
    # it must be in the process first so that both share ONE HIP/HSA runtime — two runtimes
    # in one process lose the device ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise PdseError(
            "libpdse.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise PdseError("libpdse.so lacks symbol %s declared in include/pdse.h" % name)
    lib.pdse_last_error.restype = C.c_char_p
    lib.pdse_abi_version.restype = C.c_int
    lib.pdse_desc_size.argtypes = [C.c_int]
    for kind, name in _DIRECT.items():
        getattr(lib, name).argtypes = [C.POINTER(DESC_TYPES[kind]), C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.pdse_plan_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.pdse_plan_add.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.pdse_plan_size.argtypes = [C.c_void_p]
    lib.pdse_plan_set_device.argtypes = [C.c_void_p, C.c_int]
    lib.pdse_plan_clear.argtypes = [C.c_void_p]
    lib.pdse_plan_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdse_plan_run_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pdse_plan_build_graph.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdse_plan_launch_graph.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdse_plan_time_ops.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    lib.pdse_plan_time_tag.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float),
                                       C.POINTER(C.c_int)]
    lib.pdse_plan_destroy.argtypes = [C.c_void_p]
    lib.pdse_plan_destroy.restype = None
    if lib.pdse_abi_version() != ABI_VERSION:
        raise PdseError("libpdse.so ABI %d != binding ABI %d" % (lib.pdse_abi_version(), ABI_VERSION))
    for kind, typ in DESC_TYPES.items():
        if lib.pdse_desc_size(kind) != C.sizeof(typ):
            raise PdseError("descriptor %s: library sizeof %d != binding sizeof %d"
                            % (typ.__name__, lib.pdse_desc_size(kind), C.sizeof(typ)))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise PdseError("%s failed: %s" % (what or "libpdse call", load().pdse_last_error().decode()))


def launch(desc, stream=0, device=None):
    """Launch one operator directly (on ``device`` when given, else on the current device)."""
    lib = load()
    kind = KIND_OF[type(desc)]
    if device is not None:
        import torch

        with torch.cuda.device(device):
            check(getattr(lib, _DIRECT[kind])(C.byref(desc), C.c_void_p(stream)), _DIRECT[kind])
        return
    check(getattr(lib, _DIRECT[kind])(C.byref(desc), C.c_void_p(stream)), _DIRECT[kind])


class Plan:
    """Recorded operator sequence replayed by one C call (include/pdse.h, plans)."""

    def __init__(self, device=None):
        """device: torch device (or ordinal) the plan's buffers live on; every run makes it current for the call."""
        self._lib = load()
        h = C.c_void_p()
        check(self._lib.pdse_plan_create(C.byref(h)), "pdse_plan_create")
        self._h = h
        if device is not None:
            import torch

            dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            check(self._lib.pdse_plan_set_device(self._h, int(idx)), "pdse_plan_set_device")
        self._keep = []  # python-side owners of every buffer named by a descriptor
        self.has_graph = False

    def add(self, desc, tag=0, keep=()):
        check(self._lib.pdse_plan_add(self._h, KIND_OF[type(desc)], C.byref(desc), int(tag)), "pdse_plan_add")
        self._keep.extend(keep)
        return len(self) - 1

    def keep(self, *objs):
        self._keep.extend(objs)

    def __len__(self):
        return self._lib.pdse_plan_size(self._h)

    def run(self, stream=0):
        check(self._lib.pdse_plan_run(self._h, C.c_void_p(stream)), "pdse_plan_run")

    def run_range(self, begin, end, stream=0):
        check(self._lib.pdse_plan_run_range(self._h, begin, end, C.c_void_p(stream)), "pdse_plan_run_range")

    def build_graph(self, stream):
        check(self._lib.pdse_plan_build_graph(self._h, C.c_void_p(stream)), "pdse_plan_build_graph")
        self.has_graph = True

    def launch_graph(self, stream):
        check(self._lib.pdse_plan_launch_graph(self._h, C.c_void_p(stream)), "pdse_plan_launch_graph")

    def time_ops(self, begin, end, stream=0):
        buf = (C.c_float * (end - begin))()
        check(self._lib.pdse_plan_time_ops(self._h, begin, end, C.c_void_p(stream), buf), "pdse_plan_time_ops")
        return list(buf)

    def time_tag(self, tag, stream=0):
        ms, cnt = C.c_float(), C.c_int()
        check(self._lib.pdse_plan_time_tag(self._h, tag, C.c_void_p(stream), C.byref(ms), C.byref(cnt)),
              "pdse_plan_time_tag")
        return ms.value, cnt.value

    def __del__(self):
        try:
            if self._h:
                self._lib.pdse_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass
