"""16-bit / float PCM wav read+write with the standard library only (the reference uses
librosa.load(sr=16000) and soundfile.write, trainer/complex_ddpm_trainer.py:921, :1018;
neither is a dependency of this package)."""
import wave

import numpy as np


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
        raise ValueError("%s is %d Hz; resampling to %d Hz is not built (the VoiceBank test set is 16 kHz after "
                         "the reference's preprocessing)" % (path, rate, sr))
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
