"""Host-side front door of the extraction path: audio decode, mel filter bank,
integer frame arithmetic, feature-file writer.

Mirrors, in this order, the reference's ``librosa.load(path, sr=16000)``
(preprocess_speech.py:47), the construction of ``WhisperFeatureExtractor.mel_filters``
(HF feature_extraction_whisper.py:95-103), ``n_frames = ceil(len/320)``
(preprocess_whisper.py:49-50,75-76) and ``torch.save(feats, <basename>.pt)``
(preprocess_speech.py:69-71).
"""
from __future__ import annotations

import math
import os
import wave as _wave

import numpy as np
import torch

TARGET_SR = 16000


class UnsupportedAudio(ValueError):
    pass


def _native_wav(path: str, pinned: bool = False):
    """RIFF/WAVE through libserhip's ser_wav_read_f32 (plain C, runs without the GIL): (samples, rate), or None when the file is
    a WAVE whose sample format that reader does not decode (the Python decoder below then gives the verdict).  A file that is not
    RIFF/WAVE, is truncated, or whose header is inconsistent raises ``UnsupportedAudio`` with the reader's message: the driver
    logs it as "Failed to process ..." like the reference does for any per-file error (preprocess_speech.py:46,72-73).
    ``pinned``: decode straight into page-locked memory from torch's caching host allocator (the array keeps its block alive; the
    allocator re-issues a block only after the copies enqueued from it have completed), so the launching thread can enqueue the
    H2D copy of every utterance without first packing the batch into a staging buffer (engine.upload)."""
    import ctypes
    from ._lib import lib
    sr, ch = ctypes.c_int32(0), ctypes.c_int32(0)
    bpath = os.fsencode(path)
    n = lib.ser_wav_read_f32(bpath, None, 0, ctypes.byref(sr), ctypes.byref(ch))
    if n == -5:                                           # a format tag / sample width the native reader does not decode
        return None
    if n < 0:                                             # malformed, truncated or unreadable: this file's verdict ("Failed to process")
        raise UnsupportedAudio(lib.ser_last_error().decode("utf-8", "replace"))
    x = None
    if pinned and n > 0:
        try:
            x = torch.empty(int(n), dtype=torch.float32, pin_memory=True).numpy()
        except RuntimeError:                              # page-locked memory exhausted / not available: decode into pageable memory
            x = None
    if x is None:
        x = np.empty(int(n), dtype=np.float32)
    got = lib.ser_wav_read_f32(bpath, x.ctypes.data, int(n), None, None)
    if got != n:                                          # the file changed between the two passes
        raise UnsupportedAudio(lib.ser_last_error().decode("utf-8", "replace") if got < 0 else f"{path}: frame count changed while reading")
    return x, int(sr.value)


def load_wav_16k(path: str, resample: bool = False, pinned: bool = False) -> np.ndarray:
    """Decode a RIFF/WAVE file to mono float32 in [-1, 1] the way soundfile (behind
    ``librosa.load``) does: integer PCM / 2^(bits-1), channels averaged.
    16 kHz input is bit-faithful to the reference.  Other rates: the reference resamples with soxr_hq
    (librosa 0.10.1), which is not available offline and not restated -- by default such
    files raise and the driver logs "Failed to process" like any other per-file error; with
    ``resample=True`` (driver flag ``--resample``) they go through a Kaiser-windowed polyphase filter
    (``scipy.signal.resample_poly``): usable features, but PARITY UNPINNED against the reference's resampler.
    16 kHz files take the native reader (same arithmetic, no GIL); everything else the Python path below."""
    native = _native_wav(path, pinned)
    if native is not None and native[1] == TARGET_SR:
        return native[0]
    try:
        with _wave.open(path, "rb") as wf:
            sr, ch, width, n = wf.getframerate(), wf.getnchannels(), wf.getsampwidth(), wf.getnframes()
            raw = wf.readframes(n)
        if width == 2:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif width == 4:
            x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif width == 1:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif width == 3:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        else:
            raise UnsupportedAudio(f"{width * 8}-bit PCM not supported")
    except _wave.Error:
        # IEEE-float WAVE files: scipy can read them
        from scipy.io import wavfile
        sr, data = wavfile.read(path)
        if data.dtype.kind != "f":
            raise
        x = data.astype(np.float32)
        ch = 1 if x.ndim == 1 else x.shape[1]
        x = x.reshape(-1)
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1).astype(np.float32)
    if sr != TARGET_SR:
        if not resample:
            raise UnsupportedAudio(f"sample rate {sr} Hz: only {TARGET_SR} Hz input is supported (pass --resample for a "
                                   f"polyphase resampler whose parity with librosa's soxr_hq is unpinned)")
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), TARGET_SR)
        x = resample_poly(x.astype(np.float64), TARGET_SR // g, int(sr) // g, window=("kaiser", 14.0)).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)


def whisper_mel_filters(n_mels: int = 128, n_fft: int = 400, sr: int = TARGET_SR) -> np.ndarray:
    """[201, n_mels] slaney-scale, slaney-normalised triangular filters over 0-8 kHz,
    float64 then cast to fp32 (HF audio_utils.mel_filter_bank as called from
    feature_extraction_whisper.py:95-103)."""
    bins = 1 + n_fft // 2
    freqs = np.linspace(0, sr // 2, bins)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(8000.0), n_mels + 2))
    width = np.diff(pts)
    slope = pts[None, :] - freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slope[:, :-2] / width[:-1], slope[:, 2:] / width[1:]))
    fb *= (2.0 / (pts[2:] - pts[:-2]))[None, :]
    return fb.astype(np.float32)


def whisper_saved_rows(num_samples: int, feat_dim: int) -> int:
    """Rows the reference keeps: min(ceil(len/320), feats.shape[1]) where feats is already
    [1500, D], so the cap is the hidden size (preprocess_whisper.py:49-50,75-76)."""
    return min(int(math.ceil(num_samples / 320)), int(feat_dim))


def feature_path(save_path: str, wav_path: str) -> str:
    """<save_path>/<splitext(basename(wav))[0]>.pt (preprocess_speech.py:69-70)."""
    return os.path.join(save_path, os.path.splitext(os.path.basename(wav_path))[0] + ".pt")


def tmp_name(path: str) -> str:
    """Name a feature file is written under before it is renamed into place: ``<path>.<pid>.tmp``.  The pid keeps two independent jobs that
    share a --save_path (sharding by --wav_dir with --use_n_layer is a common way to run the reference script on several GPUs,
    README.md:41-43) from deleting each other's files in progress (round 4 used ``<path>.tmp`` and removed every such file at start-up)."""
    return f"{path}.{os.getpid()}.tmp"


def is_partial(name: str) -> bool:
    return name.endswith(".tmp")


def stale_partial(save_path: str, name: str, max_age_s: float = 6 * 3600.0) -> bool:
    """A partial output nobody is writing any more: its writer's pid is gone (same host; never within a minute of its last write, so a
    writer on another host is not taken for a dead one) or the file has not been touched for hours (a recycled pid).  ``<path>.tmp``
    names of earlier versions count as stale at once."""
    import time
    parts = name.split(".")
    pid = parts[-2] if len(parts) >= 3 and parts[-2].isdigit() else None
    if pid is None:
        return True
    try:
        age = time.time() - os.path.getmtime(os.path.join(save_path, name))
    except OSError:
        return False
    if age > max_age_s:
        return True
    if age < 60.0:                                   # just written: a live writer, possibly on another host where its pid means nothing here
        return False
    if int(pid) == os.getpid():
        return False
    try:
        os.kill(int(pid), 0)
    except ProcessLookupError:
        return True
    except OSError:
        return False
    return False


def save_feature(feats: torch.Tensor, path: str) -> None:
    """``torch.save`` of a bare [T, D] float32 CPU tensor: what the downstream heads
    ``torch.load`` (bin/train_cat_bimodal_lazy_1head.py:220-228)."""
    t = feats.detach()
    if t.device.type != "cpu":
        t = t.cpu()
    t = t.to(torch.float32).contiguous()
    # written under a temporary name and renamed when complete: a killed process or a full disk leaves no truncated <name>.pt
    # for --skip_existing (which only asks whether the file exists) to mistake for a finished one
    tmp = tmp_name(path)
    try:
        if t.dim() == 2 and t.shape[1] > 0:
            # same archive torch.save writes, produced by libserhip's ser_pt_write_f32 without the GIL: the writer threads
            # of the driver no longer serialise on the interpreter (pickler + Python zip bookkeeping of torch.save)
            from ._lib import check, lib
            check(lib.ser_pt_write_f32(os.fsencode(tmp), t.data_ptr(), t.shape[0], t.shape[1]), "ser_pt_write_f32")
        else:
            torch.save(t.clone(), tmp)
        os.replace(tmp, path)
    except BaseException:
        try:
            os.remove(tmp)
        except OSError:
            pass
        raise
