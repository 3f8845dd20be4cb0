"""A host WITHOUT the plan builders: nothing of prior-diffuse_amd is imported - libpdse.so is bound with bare ctypes, a plan file
is loaded (pdse_plan_load), the network-level entry points are called on device buffers and the results are compared with the
reference's golden outputs.  torch is used for device memory and the seeded inputs only.  argv: libpdse.so kind plan golden.npz
kind = eps | prior | enhance"""
import ctypes as C
import sys

import numpy as np
import torch


def seeded(shape, seed):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    libpath, kind, plan_path, golden = sys.argv[1:5]
    assert not any(m.startswith("prior-diffuse_amd") or m.startswith("prior_diffuse_amd") for m in sys.modules)
    torch.zeros(1, device="cuda:0")                      # torch's HIP runtime first: libpdse.so must share it
    lib = C.CDLL(libpath)
    lib.pdse_last_error.restype = C.c_char_p

    def ok(rc, what):
        if rc:
            raise RuntimeError("%s: %s" % (what, lib.pdse_last_error().decode()))

    plan = C.c_void_p()
    ok(lib.pdse_plan_load(plan_path.encode(), C.byref(plan)), "pdse_plan_load")
    g = np.load(golden)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())              # noqa: E731
    if kind == "eps":
        B, T = int(g["B"]), int(g["T"])
        x = seeded((B, 2, T, 161), g["seed_x"]).cuda()
        xi = (seeded((B, 2, T, 161), g["seed_init"]) * float(g["init_scale"])).cuda()
        t = torch.from_numpy(g["t"]).float().cuda()
        out = torch.empty_like(x)
        for _ in range(2):                                # the second call: the plan is re-runnable
            ok(lib.pdse_eps_forward(plan, ptr(x), ptr(xi), ptr(t), ptr(out), st), "pdse_eps_forward")
        torch.cuda.synchronize()
        err = rel_l2(out.cpu().numpy(), g["out"])
    elif kind == "prior":
        x = seeded((2, 2, 20, 161), g["seed_x"]).cuda()
        out = torch.empty_like(x)
        ok(lib.pdse_prior_forward(plan, ptr(x), ptr(out), st), "pdse_prior_forward")
        torch.cuda.synchronize()
        err = rel_l2(out.cpu().numpy(), g["out"])
    else:
        wav, x_T = torch.from_numpy(g["wav"]).cuda(), torch.from_numpy(g["x_T"]).cuda()
        out, spec = torch.empty_like(wav), torch.empty_like(x_T)
        ok(lib.pdse_enhance(plan, ptr(wav), ptr(x_T), ptr(out), ptr(spec), st), "pdse_enhance")
        torch.cuda.synchronize()
        err = max(rel_l2(out.cpu().numpy(), g["wav_out"]), rel_l2(spec.cpu().numpy(), g["spec"]))
    # errors of the C API surface: an unknown region, a plan that is not a file
    p, n = C.c_void_p(), C.c_uint64()
    assert lib.pdse_plan_region(plan, b"no-such-region", C.byref(p), C.byref(n)) != 0
    assert lib.pdse_plan_load(b"/nonexistent/plan", C.byref(p)) != 0
    lib.pdse_plan_destroy(plan)
    print("plan_client %s rel-L2 %.3e" % (kind, err))
    return 0 if err < float(sys.argv[5]) else 1


if __name__ == "__main__":
    sys.exit(main())
