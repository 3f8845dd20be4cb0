"""TEST INFRASTRUCTURE (not a product path): how far the reference's own ``generate_wav`` is from its eval-mode twin.

``ComplexDDPMTrainer.generate_wav`` only calls ``self.model.eval()`` (trainer/complex_ddpm_trainer.py:914): DiffUNet1
stays in train mode, so its BatchNorm layers normalise with the statistics of the single utterance being enhanced
(and keep updating their running averages).  The drop-in folds the running statistics (eval mode, as the reference's
validation loop and every fixture do).  This script runs the reference's own sampling statements (make_golden.py:
ref_generate_body, cut from the AST) both ways on the seeded weights and prints the gap, which INTEGRATION.md quotes.
    python oracle/bn_train_mode_gap.py
"""
import copy
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as G  # noqa: E402

synth = importlib.import_module("prior-diffuse_amd.synth")


def main():
    torch.manual_seed(0)
    ref = G.load_reference()
    eps_net = ref.diff3.DiffUNet1(ref.params)
    eps_net.load_state_dict(synth.make_state_dict("DiffUNet1", 1234), strict=True)
    gcrn = ref.gcrn.GCRN()
    gcrn.load_state_dict(synth.make_state_dict("GCRN", 1234), strict=True)
    gcrn.eval()
    sched = G.ref_inference_schedule(ref, True)
    for T in (101, 401):
        feat = G.seeded((1, 2, T, 161), 31) * 0.5
        x_T = G.seeded((1, 2, T, 161), 32)
        outs = {}
        with torch.no_grad():
            for mode in ("eval", "train"):
                net = copy.deepcopy(eps_net)
                net.train(mode == "train")
                outs[mode] = G.ref_generate_body(gcrn, net, feat, x_T, sched, True, False, False)[0]
        gap = (outs["train"] - outs["eval"]).norm() / outs["eval"].norm()
        print("T=%d, B=1, GCRN prior + DiffUNet1, 6 steps, seeded weights: |train-mode - eval-mode| / |eval-mode| = %.3e" % (T, gap))


if __name__ == "__main__":
    main()
