"""The RCCL leg of the sharded path on the one GPU of the test box (tests/test_gpu_f16x2.py): a process group of ONE rank with
backend "nccl" (= RCCL on ROCm) - the barrier and the max-reduce bench.py times its region with, and the device-buffer all_gather
of prior-diffuse_amd/shard.py::gather_shards.  (Two ranks cannot share a device under RCCL; the N > 1 collective runs on the
driver's 8-GPU node only.)  argv: out_dir."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg  # noqa: E402


def main():
    out_dir = sys.argv[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    synth, shard = pkg("synth"), pkg("shard")
    ns = argparse.Namespace
    t = pkg("trainer").ComplexDDPMTrainer(
        ns(retrain=False, joint=True, draw=False, sigma=False, checkpoint="x", generated_wav="y"),
        ns(model=ns(name="GCRN"), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt")),
        device="cuda:0", prior_state_dict=synth.make_state_dict("GCRN"), ddpm_state_dict=synth.make_state_dict("DiffUNet1"))
    wav, x_T = synth.synthetic_waveforms(5, 4000, seed=21)
    dist.barrier()
    local = shard.enhance_sharded(lambda w, x: t.enhance(w, x_T=x), wav.cuda(), x_T.cuda(), gather=False)
    full = shard.gather_shards(local, 5)                       # ncclAllGather on device buffers
    tt = torch.tensor([1.25], device="cuda:0", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)                  # what bench.py does with its elapsed time
    torch.cuda.synchronize()
    assert full.is_cuda and float(tt.item()) == 1.25
    torch.save({"full": full.cpu(), "local": local.cpu()}, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
