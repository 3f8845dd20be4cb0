"""CLI mirror of the reference's ``main.py`` for the ``--generate`` path
(main.py:20-101): same flags, same derived directories, YAML config -> Namespace.

    python -m prior_diffuse_amd.main --retrain --assets A --sigma --joint --generate
"""
import argparse
import logging
import os

import numpy as np
import torch
import yaml


def dict2namespace(config):  # main.py:9-17
    ns = argparse.Namespace()
    for key, value in config.items():
        setattr(ns, key, dict2namespace(value) if isinstance(value, dict) else value)
    return ns


def parse_args_and_config(argv=None):
    p = argparse.ArgumentParser(description=__doc__)
    p.add_argument("--seed", type=int, default=1234, help="Random seed")
    p.add_argument("--trainer", type=str, default="ComplexDDPMTrainer", help="The trainer to execute")
    p.add_argument("--config", type=str, default="diff.yml", help="Path to the config file")
    p.add_argument("--verbose", type=str, default="info", help="Verbose level: info | debug | warning | critical")
    p.add_argument("--doc", type=str, default="diff", help="A string for documentation purpose")
    p.add_argument("--comment", type=str, default="", help="A string for experiment comment")
    p.add_argument("--assets", type=str, default="assets_dpm", help="Path for saving running related data.")
    for flag in ("generate", "retrain", "joint", "eval", "sigma", "noisy", "draw"):
        p.add_argument("--" + flag, action="store_true")
    p.add_argument("--data", type=str, default="data/noisy_testset_wav", help="directory of noisy wavs")
    p.add_argument("--bf16", action="store_true",
                   help="(not a flag of the reference) the opt-in reduced-precision mode: plain bf16 operands and bf16 block-boundary "
                        "tensors in the eps-net's blocks and the priors' GEMM-shaped convolutions, tolerance 3e-2 rel-L2 against the fp32 path; default: fp32-equivalent arithmetic")
    args = p.parse_args(argv)
    args.log = os.path.join(args.assets, "log", args.doc)
    args.checkpoint = os.path.join(args.assets, "checkpoint", args.doc)
    args.generated_wav = os.path.join(args.assets, "wav", args.doc)
    with open(os.path.join("conf", args.config), "r") as f:
        config = dict2namespace(yaml.safe_load(f))
    level = getattr(logging, args.verbose.upper(), None)
    if not isinstance(level, int):
        raise ValueError("level {} not supported".format(args.verbose))  # main.py:50-51
    for d in (args.log, args.checkpoint, args.generated_wav):
        os.makedirs(d, exist_ok=True)
    logging.basicConfig(level=level, format="%(levelname)s - %(filename)s - %(asctime)s - %(message)s",
                        handlers=[logging.StreamHandler(), logging.FileHandler(os.path.join(args.log, "stdout.txt"))])
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)
    return args, config


def main(argv=None):
    from .trainer import ComplexDDPMTrainer

    args, config = parse_args_and_config(argv)
    if args.trainer != "ComplexDDPMTrainer":
        raise NotImplementedError("only ComplexDDPMTrainer's sampling path is built")
    trainer = ComplexDDPMTrainer(args, config)
    if args.generate:
        return trainer.generate_wav(load_pre_train=True, data_path=args.data)
    trainer.train_ddpm()


if __name__ == "__main__":
    main()
