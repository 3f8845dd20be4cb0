"""16-bit / float PCM wav read+write with the standard library only (the reference uses
librosa.load(sr=16000) and soundfile.write, trainer/complex_ddpm_trainer.py:921, :1018;
neither is a dependency of this package).  Precondition: RIFF/WAVE PCM (8/16/32-bit integer); files the ``wave``
module rejects (IEEE float, WAVE_FORMAT_EXTENSIBLE) raise and ``generate_wav`` skips them with a warning.  Other
sample rates are converted with a Kaiser-windowed-sinc polyphase filter; librosa's default (soxr_hq) is a
different filter of similar quality, so resampled inputs agree with the reference's to filter-design accuracy
(about -80 dB), not bit for bit."""
import wave
from math import gcd

import numpy as np


def resample(x, sr_in, sr_out, half_width=16, beta=8.6):
    """Rational-rate conversion: out[n] = sum_k h(n * down - k * up) x[k] with a Kaiser-windowed sinc low-pass at
    the narrower Nyquist band, evaluated polyphase (only the taps that meet a non-zero input sample)."""
    x = np.asarray(x, dtype=np.float64)
    if sr_in == sr_out or x.size == 0:
        return x.astype(np.float32)
    g = gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    m = max(up, down)
    half = half_width * m                                     # filter half length on the up-sampled grid
    n_h = np.arange(-half, half + 1)
    h = np.sinc(n_h / m) / m * np.kaiser(2 * half + 1, beta) * up
    n_out = int(np.ceil(x.size * up / down))
    out = np.zeros(n_out)
    n = np.arange(n_out)
    q = n * down                                              # position of output n on the up-sampled grid
    k_c, r = q // up, q % up                                  # nearest input sample at or before it, offset in between
    jmax = half // up + 1
    for j in range(-jmax, jmax + 1):                          # input sample k_c - j sits at grid offset r + j * up
        off = r + j * up
        k = k_c - j
        ok = (np.abs(off) <= half) & (k >= 0) & (k < x.size)
        out[ok] += h[off[ok] + half] * x[k[ok]]
    return out.astype(np.float32)


def read_wav(path, sr=16000):
    with wave.open(path, "rb") as f:
        n, ch, width, rate = f.getnframes(), f.getnchannels(), f.getsampwidth(), f.getframerate()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError("unsupported sample width %d in %s" % (width, path))
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)  # librosa.load(mono=True)
    if rate != sr:
        x = resample(x, rate, sr)        # librosa.load(sr=16000) resamples (VoiceBank-DEMAND ships at 48 kHz)
    return x


def write_wav(path, x, sr=16000):
    """soundfile.write default subtype for .wav: PCM_16."""
    y = np.clip(np.asarray(x, dtype=np.float64), -1.0, 1.0 - 1.0 / 32768.0)
    pcm = np.round(y * 32768.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())
