"""One rank of the real sharded path for tests/test_gpu_round2.py: every rank builds the drop-in trainer on the SAME
GPU (gloo rendezvous - RCCL refuses two ranks per device), enhances its contiguous shard of the global batch with
the real HIP pipeline and all-gathers the waveforms.  argv: out_dir."""
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
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth = pkg("synth")
    ns = argparse.Namespace
    t = pkg("trainer").ComplexDDPMTrainer(
        ns(retrain=False, joint=True, draw=False, sigma=False, checkpoint="x", generated_wav="y"),
        ns(model=ns(name="GCRN"), train=ns(fft_num=320, win_size=320, win_shift=160, feat_type="sqrt")),
        device="cuda:0", prior_state_dict=synth.make_state_dict("GCRN"), ddpm_state_dict=synth.make_state_dict("DiffUNet1"))
    wav, x_T = synth.synthetic_waveforms(5, 4000, seed=21)            # ragged split: 3 + 2 utterances
    out = pkg("shard").enhance_sharded(lambda w, x: t.enhance(w, x_T=x), wav.cuda(), x_T.cuda())
    torch.cuda.synchronize()
    torch.save(out.cpu(), os.path.join(out_dir, "out%d.pt" % rank))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
