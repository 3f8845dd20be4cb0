"""Schedule constants of the sampling path.

Mirrors the reference's module-level ``params`` AttrDict
(reference: utils/params.py:19-41): same attribute names, same values, so that
``trainer.params.fast_sampling`` etc. read identically on both sides.
"""
import numpy as np


class AttrDict(dict):
    """Mapping whose keys read and write as attributes (``params.fast_sampling``), which is how the reference's
    trainer consumes its module-level ``params`` (utils/params.py:35-41)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name) from None

    def copy(self):
        return AttrDict(self)


# reference: utils/params.py:35-41 (the active, un-commented "diffwave" schedule)
params = AttrDict(
    deltamu=False,
    pirorgrad=True,
    ours=False,
    fast_sampling=True,
    noise_schedule=np.linspace(1e-4, 0.05, 50).tolist(),
    inference_noise_schedule=[0.0001, 0.001, 0.01, 0.05, 0.2, 0.5],
)

# hard-coded in the reference trainer (trainer/complex_ddpm_trainer.py:30)
PRIOR_SCALE_C = 11.0

# STFT geometry (reference: conf/diff.yml:6-9, conf/gcrn.yml:6-8)
SAMPLE_RATE = 16000
FFT_NUM = 320
WIN_SIZE = 320
WIN_SHIFT = 160
NUM_BINS = FFT_NUM // 2 + 1  # 161, baked into every model of the path
