"""Drop-in for the sampling half of the reference's ``ComplexDDPMTrainer``
(trainer/complex_ddpm_trainer.py): same constructor arguments, same
``inference_schedule()`` return tuple, same ``generate_wav()`` behaviour, plus the batched
entry points the reference lacks (it can only read wav files from a directory):

    enhance(wav[B,L], x_T=None)      -> wav[B,L]         whole path, batched
    sample(feat[B,2,T,161], x_T)     -> spectrogram      :939-998 on a given spectrogram

Training (train_ddpm/train_step/train/draw_audio) is out of scope.
Deliberate differences from the reference, all recorded in SURVEY.md §0/§2:
  * both networks run with eval-mode BatchNorm (the reference's generate_wav forgets
    ``model_ddpm.eval()``, :914 — a bug, not a target);
  * no ``exit()`` after generation (:1021), no wandb, no CUDA_VISIBLE_DEVICES override.
"""
import glob
import logging
import os
from copy import deepcopy

import numpy as np
import torch

from . import _lib as L
from . import ops, wavio
from .params import PRIOR_SCALE_C, params as default_params
from .pipeline import SamplerPipeline
from .schedule import inference_schedule as _inference_schedule


class ComplexDDPMTrainer(object):
    def __init__(self, args, config, device=None, prior_state_dict=None, ddpm_state_dict=None, params=None):
        """args: .retrain .joint .draw .sigma .checkpoint .generated_wav
        config: .model.name, .train.{fft_num, win_size, win_shift, feat_type}
        Weights come from ``<args.checkpoint>/best_checkpoint.pth`` under the reference's
        rules (:91-97) or from the two state_dict arguments (synthetic runs)."""
        self.c = PRIOR_SCALE_C                                        # :30
        self.args = deepcopy(args)
        self.config = deepcopy(config)
        self.params = default_params if params is None else params    # :34 (override: synthetic runs / deltamu)
        self.pirorgrad = self.params.pirorgrad
        self.deltamu = self.params.deltamu
        if self.pirorgrad == self.deltamu:
            # :70-75 also has a third branch (neither flag: DiffUNet1 conditioned on the noisy feature); not built
            raise NotImplementedError("exactly one of pirorgrad (DiffUNet1) / deltamu (Nocon) must be set")
        tr = self.config.train
        if (tr.fft_num, tr.win_size, tr.win_shift) != (320, 320, 160) or tr.feat_type != "sqrt":
            raise NotImplementedError("STFT 320/320/160 with feat_type 'sqrt' is baked into every model of the path")
        if device is None:
            rank = int(os.environ.get("LOCAL_RANK", "0"))
            device = "cuda:%d" % rank
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise L.PdseError("ComplexDDPMTrainer needs an MI355X (device %s unavailable); no CPU fallback" % device)
        L.load()
        self.prior_name = self.config.model.name
        if self.prior_name not in ops.PRIOR_OPS:
            raise NotImplementedError("prior %r: built priors are %s" % (self.prior_name, sorted(ops.PRIOR_OPS)))
        self.prior_sd, self.ddpm_sd = prior_state_dict, ddpm_state_dict
        if getattr(self.args, "retrain", False):                      # :91-97
            self._load_checkpoint()
        if self.prior_sd is None or self.ddpm_sd is None:
            raise ValueError("no weights: pass state_dicts or use --retrain with a best_checkpoint.pth")
        self.model = ops.PRIOR_OPS[self.prior_name](self.prior_sd, self.device)       # :69
        self.model_ddpm = (ops.NoconOp if self.deltamu else ops.DiffUNet1Op)(self.ddpm_sd, self.device)   # :70-73
        self._pipes = {}

    # ---- A8 checkpoint rules (:91-97, :906-913) ---------------------------
    def _load_checkpoint(self):
        path = os.path.join(self.args.checkpoint, "best_checkpoint.pth")
        data = torch.load(path, map_location="cpu")
        if isinstance(data, (list, tuple)):
            self.prior_sd = data[0]
            if getattr(self.args, "draw", False) or getattr(self.args, "joint", False):
                self.ddpm_sd = data[2]
        else:
            self.prior_sd = data
        logging.info("loaded %s", path)

    # ---- A1 ---------------------------------------------------------------
    def inference_schedule(self, fast_sampling=False):
        return _inference_schedule(self.params, fast_sampling)

    # ---- batched entry points ----------------------------------------------
    def _pipe(self, B, T=None, L_=None):
        key = (B, T, L_, bool(getattr(self.args, "sigma", False)), bool(self.params.fast_sampling))
        if key not in self._pipes:
            self._pipes[key] = SamplerPipeline(self.device, self.prior_name, self.prior_sd, self.ddpm_sd, B, T=T,
                                               L_=L_, fast_sampling=self.params.fast_sampling,
                                               use_sigma=key[3], params=self.params, deltamu=self.deltamu)
        return self._pipes[key]

    def _x_T(self, shape, x_T):
        if x_T is None:                                               # :947-950 randn_like(init)
            return torch.randn(*shape, device=self.device, dtype=torch.float32)
        return x_T.to(self.device)

    def sample(self, feat, x_T=None):
        """feat [B,2,T,161] compressed spectrogram -> enhanced compressed spectrogram."""
        feat = feat.to(self.device)
        B, _, T, _ = feat.shape
        spec, _ = self._pipe(B, T=T).sample(feat, self._x_T(feat.shape, x_T))
        return spec

    def enhance(self, wav, x_T=None):
        """wav [B,L] (any scale; RMS-normalised internally like :922-923) -> enhanced [B,L]."""
        wav = wav.to(self.device, torch.float32)
        B, L_ = wav.shape
        T = 1 + L_ // 160
        out, _ = self._pipe(B, L_=L_).enhance(wav, self._x_T((B, 2, T, 161), x_T))
        return out

    def enhance_batch(self, wavs, x_T=None, trim_to_frames=False):
        """Ragged batch, the validation loop's convention: every utterance is RMS-normalised
        over its own samples, zero-padded to the longest (utils/dataset.py:45-58), enhanced in one batch
        (:408-494) and cut back — to its own length, or with ``trim_to_frames`` to ``(frame_num - 1) * 160``
        samples as utils/metrics.py:562-563 does.  Returns a list of 1-D tensors (rescaled by c)."""
        wavs = [torch.as_tensor(w, dtype=torch.float32).flatten() for w in wavs]
        lens = [int(w.numel()) for w in wavs]
        if min(lens) < 161:
            raise ValueError("utterances must be longer than the reflect padding (160 samples)")
        batch = torch.nn.utils.rnn.pad_sequence(wavs, batch_first=True).to(self.device)
        B, L_ = batch.shape
        T = 1 + L_ // 160
        out, _ = self._pipe(B, L_=L_).enhance(batch, self._x_T((B, 2, T, 161), x_T), lens=lens)
        cut = [(n // 160) * 160 if trim_to_frames else n for n in lens]
        return [out[i, :cut[i]].clone() for i in range(B)]

    # ---- A2..A7: the reference's entry point --------------------------------
    def generate_wav(self, load_pre_train=True, data_path="data/noisy_testset_wav"):
        """Per-file B=1 enhancement of ``data_path/*.wav`` into ``args.generated_wav``
        (:903-1018).  Returns the list of written paths instead of calling exit()."""
        if load_pre_train and getattr(self.args, "retrain", False):
            self._load_checkpoint()
        os.makedirs(self.args.generated_wav, exist_ok=True)
        written = []
        with torch.no_grad():
            for path in sorted(glob.glob(data_path + "/*.wav")):
                wav = torch.from_numpy(wavio.read_wav(path, 16000))[None]
                out = self.enhance(wav)[0].cpu().numpy()
                dst = os.path.join(self.args.generated_wav, path.split("/")[-1])
                wavio.write_wav(dst, out, 16000)
                written.append(dst)
        print("success!")
        return written

    def train_ddpm(self):
        raise NotImplementedError("training is outside the sampling path this package implements (SURVEY.md §8)")

    train = train_step = draw_audio = train_ddpm
