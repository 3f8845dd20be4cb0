"""Schedule constants of the sampling path.

Mirrors the reference's module-level ``params`` AttrDict
(reference: utils/params.py:19-41): same attribute names, same values, so that
``trainer.params.fast_sampling`` etc. read identically on both sides.
"""
import numpy as np


class AttrDict(dict):
    """dict whose keys are also attributes (reference: utils/params.py:19-32)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self

    def override(self, attrs):
        if isinstance(attrs, dict):
            self.__dict__.update(**attrs)
        elif isinstance(attrs, (list, tuple, set)):
            for attr in attrs:
                self.override(attr)
        elif attrs is not None:
            raise NotImplementedError
        return self


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
