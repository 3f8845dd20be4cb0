"""Noise-schedule arithmetic of the reverse loop (host side, float64).

Follows ``ComplexDDPMTrainer.inference_schedule``
(reference: trainer/complex_ddpm_trainer.py:105-156) operation by operation so
the returned arrays are bit-identical: float64 numpy throughout, ``T`` cast to
float32 at the end (reference :139).
"""
import numpy as np


def inference_schedule(params, fast_sampling=False):
    """Return ``(alpha, beta, alpha_cum, sigmas, T)`` exactly like the reference.

    alpha, beta, alpha_cum : float64 ndarray [S]   (S = 6 fast / 50 full)
    sigmas                 : python list of S floats; ``sigmas[0]`` wraps to
                             ``alpha_cum[-1]`` like the reference (:127-128)
    T                      : float32 ndarray, fractional training-step index
                             aligned to the training schedule (:131-139)
    """
    training_noise_schedule = np.array(params.noise_schedule)
    inference_noise_schedule = (
        np.array(params.inference_noise_schedule) if fast_sampling else training_noise_schedule
    )

    talpha = 1 - training_noise_schedule
    talpha_cum = np.cumprod(talpha)

    beta = inference_noise_schedule
    alpha = 1 - beta
    alpha_cum = np.cumprod(alpha)
    sigmas = [0 for _ in alpha]
    for n in range(len(alpha) - 1, -1, -1):
        sigmas[n] = ((1.0 - alpha_cum[n - 1]) / (1.0 - alpha_cum[n]) * beta[n]) ** 0.5

    T = []
    for s in range(len(inference_noise_schedule)):
        for t in range(len(training_noise_schedule) - 1):
            if talpha_cum[t + 1] <= alpha_cum[s] <= talpha_cum[t]:
                twiddle = (talpha_cum[t] ** 0.5 - alpha_cum[s] ** 0.5) / (
                    talpha_cum[t] ** 0.5 - talpha_cum[t + 1] ** 0.5
                )
                T.append(t + twiddle)
                break
    T = np.array(T, dtype=np.float32)
    return alpha, beta, alpha_cum, sigmas, T


def step_coefficients(alpha, beta, alpha_cum):
    """Per-step (c1, c2) of the posterior mean, as the float32 values the
    reference's tensor arithmetic actually multiplies by.

    reference: trainer/complex_ddpm_trainer.py:965-966 computes them in python
    float64; ``c2 * eps`` / ``c1 * (...)`` then run as fp32 tensor-scalar ops
    (scalar rounded to float32).  Returned arrays are indexed by n.
    """
    c1 = 1.0 / alpha ** 0.5
    c2 = beta / (1.0 - alpha_cum) ** 0.5
    return c1.astype(np.float32), c2.astype(np.float32)
