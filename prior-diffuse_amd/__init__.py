"""prior-diffuse_amd — MI355X-native reverse-diffusion sampling path of Prior-DiffuSE.

One hot path only: STFT → discriminative prior → X_init →
N-step complex-spectrogram DDPM reverse loop over DiffUNet1 → ISTFT, behind the
reference's ``ComplexDDPMTrainer.inference_schedule()/generate_wav()`` surface.
Python host code calls hand-written HIP kernels (gfx950) through the C-ABI
declared in ``include/pdse.h``; PyTorch is used for device memory, streams and
``torch.distributed`` only.

The directory name contains a hyphen, so import it with
``importlib.import_module("prior-diffuse_amd")`` or through the
``prior_diffuse_amd`` alias package at the repo root.
"""
from .params import params, AttrDict  # noqa: F401
from .schedule import inference_schedule  # noqa: F401

__all__ = ["params", "AttrDict", "inference_schedule"]
