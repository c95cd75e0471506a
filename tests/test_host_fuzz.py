"""Untrusted-input hardening of the host-side file I/O (rows a5 / a21), CPU only.

The reference's contract for a bad file is one log line and the next file (``Failed to process ...``,
preprocessing/preprocess_speech.py:46,72-73) -- never undefined behaviour.  ``ser_wav_read_f32`` parses RIFF headers that come
from disk, so it is built here with AddressSanitizer + UBSan (``make -C interspeech_ser_amd/csrc asan``: plain g++ on hostio.hip +
hosterr.hip, no GPU) and fed a corpus of malformed files plus seeded random mutations of a valid one; the same corpus then goes
through the product library (ctypes) and must give the same verdicts, and through ``driver._run`` around a stubbed model.
Round 3's reader sized its buffer by the header's blockAlign and read channels x width bytes per frame: the first corpus entry
(blockAlign 2, 60 000 channels, 64 data bytes) read 120 KB of heap past a 64-byte vector.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from interspeech_ser_amd import config as C
from interspeech_ser_amd import driver, frontend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "interspeech_ser_amd", "csrc")
PCM_GUID = b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"


def fmt_chunk(tag=1, ch=1, sr=16000, bits=16, align=None, rate=None, ext=None):
    width = max(bits // 8, 1)
    align = ch * width if align is None else align
    rate = sr * ch * width if rate is None else rate
    body = struct.pack("<HHIIHH", tag & 0xFFFF, ch & 0xFFFF, sr & 0xFFFFFFFF, rate & 0xFFFFFFFF, align & 0xFFFF, bits & 0xFFFF)
    if ext is not None:                                  # WAVE_FORMAT_EXTENSIBLE tail: cbSize, valid bits, channel mask, sub-format
        body += struct.pack("<HHIH", 22, bits, 4, ext) + PCM_GUID
    return b"fmt " + struct.pack("<I", len(body)) + body


def riff(*chunks, riff_len=None):
    body = b"WAVE" + b"".join(chunks)
    return b"RIFF" + struct.pack("<I", len(body) if riff_len is None else riff_len) + body


def data_chunk(payload, declared=None):
    return b"data" + struct.pack("<I", len(payload) if declared is None else declared) + payload + (b"\x00" if len(payload) & 1 else b"")


def corpus():
    """name -> (file bytes, expected frames or None for "must be refused", expected samples or None)."""
    rng = np.random.default_rng(5)
    pcm16 = rng.integers(-30000, 30000, size=400).astype("<i2")
    f16 = pcm16.astype(np.float32) / 32768.0
    out = {}
    out["blockalign2_60000ch"] = (riff(fmt_chunk(ch=60000, align=2), data_chunk(b"\x01\x02" * 32)), None, None)
    # blockAlign smaller / larger than channels x width: libsndfile decodes with channels x width whatever the field says
    st = rng.integers(-30000, 30000, size=(16, 8)).astype("<i2")
    out["blockalign2_8ch"] = (riff(fmt_chunk(ch=8, align=2), data_chunk(st.tobytes())), 16, (st.astype(np.float32) / 32768.0).mean(axis=1, dtype=np.float32))
    out["blockalign0"] = (riff(fmt_chunk(align=0), data_chunk(pcm16.tobytes())), 400, f16)
    out["blockalign65535"] = (riff(fmt_chunk(align=65535), data_chunk(pcm16.tobytes())), 400, f16)
    out["valid_pcm16"] = (riff(fmt_chunk(), data_chunk(pcm16.tobytes())), 400, f16)
    out["data_len_ffffffff"] = (riff(fmt_chunk(), data_chunk(pcm16.tobytes(), declared=0xFFFFFFFF)), 400, f16)     # streaming writers
    out["data_len_beyond_file"] = (riff(fmt_chunk(), data_chunk(pcm16.tobytes(), declared=10 ** 9)), 400, f16)
    out["data_len_short_odd"] = (riff(fmt_chunk(), data_chunk(pcm16.tobytes(), declared=301)), 150, f16[:150])         # a trailing half sample is dropped
    out["riff_len_wrong"] = (riff(fmt_chunk(), data_chunk(pcm16.tobytes()), riff_len=7), 400, f16)
    out["fmt_after_data"] = (riff(data_chunk(pcm16.tobytes()), fmt_chunk()), None, None)
    out["no_data_chunk"] = (riff(fmt_chunk()), None, None)
    out["fmt_too_short"] = (riff(b"fmt " + struct.pack("<I", 8) + b"\x01\x00\x01\x00\x80\x3e\x00\x00", data_chunk(pcm16.tobytes())), None, None)
    out["fmt_len_ffffffff"] = (riff(b"fmt " + struct.pack("<I", 0xFFFFFFFF) + fmt_chunk()[8:], data_chunk(pcm16.tobytes())), None, None)
    out["junk_len_ffffffff"] = (riff(b"junk" + struct.pack("<I", 0xFFFFFFFF) + b"abcd", fmt_chunk(), data_chunk(pcm16.tobytes())), None, None)
    out["odd_chunks"] = (riff(b"LIST" + struct.pack("<I", 5) + b"INFOx\x00", fmt_chunk(), b"junk" + struct.pack("<I", 3) + b"abc\x00",
                              data_chunk(pcm16.tobytes())), 400, f16)
    out["channels_0"] = (riff(fmt_chunk(ch=0, align=2), data_chunk(pcm16.tobytes())), None, None)
    out["channels_1025"] = (riff(fmt_chunk(ch=1025), data_chunk(b"\x00" * 4100)), None, None)
    out["bits_0"] = (riff(fmt_chunk(bits=0, align=2), data_chunk(pcm16.tobytes())), None, None)
    # formats the native reader hands to the Python decoder (rc -5), which reads them: 12 valid bits in a 16-bit container, float64
    out["bits_12"] = (riff(fmt_chunk(bits=12, align=2), data_chunk(pcm16.tobytes())), "python", f16)
    out["bits_40"] = (riff(fmt_chunk(bits=40), data_chunk(b"\x00" * 400)), None, None)
    f64 = rng.standard_normal(100)
    out["float_64bit"] = (riff(fmt_chunk(tag=3, bits=64), data_chunk(f64.astype("<f8").tobytes())), "python", f64.astype(np.float32))
    out["tag_adpcm"] = (riff(fmt_chunk(tag=2, bits=4, align=256), data_chunk(b"\x00" * 512)), None, None)
    out["extensible_pcm16"] = (riff(fmt_chunk(tag=0xFFFE, ext=1), data_chunk(pcm16.tobytes())), 400, f16)
    flt = rng.standard_normal(300).astype("<f4")
    out["extensible_float32"] = (riff(fmt_chunk(tag=0xFFFE, bits=32, ext=3), data_chunk(flt.tobytes())), 300, flt)
    out["extensible_cut_guid"] = (riff(b"fmt " + struct.pack("<I", 20) + fmt_chunk(tag=0xFFFE, ext=1)[8:28], data_chunk(pcm16.tobytes())), None, None)
    b24 = rng.integers(0, 256, size=3 * 100, dtype=np.uint8)
    v = b24.reshape(-1, 3).astype(np.int32)
    v = v[:, 0] | (v[:, 1] << 8) | (v[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v)
    out["pcm24"] = (riff(fmt_chunk(bits=24), data_chunk(b24.tobytes())), 100, (v.astype(np.float64) / 8388608.0).astype(np.float32))
    out["empty_data"] = (riff(fmt_chunk(), data_chunk(b"")), 0, np.zeros(0, np.float32))
    whole = riff(fmt_chunk(), data_chunk(pcm16.tobytes()))
    for cut in (0, 3, 4, 11, 12, 15, 19, 20, 27, 35, 36, 40, 43):          # truncated inside the RIFF header, a chunk header, the fmt body
        out[f"truncated_{cut:02d}"] = (whole[:cut], None, None)
    out["truncated_45"] = (whole[:45], 0, np.zeros(0, np.float32))         # header complete, half a sample of data
    out["truncated_244"] = (whole[:244], 100, f16[:100])
    return out


@pytest.fixture(scope="module")
def fuzz_binary():
    subprocess.check_call(["make", "-C", CSRC, "asan"], stdout=subprocess.DEVNULL)
    path = os.path.join(CSRC, "build", "asan", "hostio_fuzz")
    assert os.path.isfile(path)
    return path


def run_sanitized(binary, *args):
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:exitcode=99:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([binary, *args], capture_output=True, text=True, env=env, timeout=300)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return r.stdout


def test_wav_reader_corpus_under_address_and_ub_sanitizers(tmp_path, fuzz_binary, built_library):
    from interspeech_ser_amd._lib import lib
    cases = corpus()
    names = sorted(cases)
    for n in names:
        (tmp_path / (n + ".wav")).write_bytes(cases[n][0])
    paths = [str(tmp_path / (n + ".wav")) for n in names]
    lines = run_sanitized(fuzz_binary, "wav", *paths).strip().split("\n")
    assert len(lines) == len(names)
    for n, p, line in zip(names, paths, lines):
        want_frames, want = cases[n][1], cases[n][2]
        got = int(line.split()[0])
        # the sanitized build and the shipped library are the same source: same verdict, same samples
        sr, ch = ctypes.c_int32(-1), ctypes.c_int32(-1)
        prod = lib.ser_wav_read_f32(os.fsencode(p), None, 0, ctypes.byref(sr), ctypes.byref(ch))
        assert prod == got, (n, prod, got)
        if want_frames == "python":
            assert got == -5, (n, line)
            x = frontend.load_wav_16k(p)
            assert x.dtype == np.float32 and np.array_equal(x, want), n
        elif want_frames is None:
            assert got < 0, (n, line)
            assert lib.ser_last_error(), n
            with pytest.raises(Exception):
                frontend.load_wav_16k(p)
        else:
            assert got == want_frames, (n, line)
            x = frontend.load_wav_16k(p)
            assert x.dtype == np.float32 and x.shape == (want_frames,) and np.array_equal(x, want), n


def test_wav_reader_survives_seeded_mutations(tmp_path, fuzz_binary):
    """4 x 1500 random edits of valid files (byte flips and edge values in the header fields, truncations) decoded under the
    sanitizers: whatever the verdict, no out-of-bounds access, no overflow, no leak, and a short buffer is always refused."""
    rng = np.random.default_rng(9)
    seeds = {
        "m16": riff(fmt_chunk(), data_chunk(rng.integers(-3000, 3000, size=257).astype("<i2").tobytes())),
        "s24": riff(fmt_chunk(ch=2, bits=24), data_chunk(rng.integers(0, 256, size=6 * 40, dtype=np.uint8).tobytes())),
        "ext": riff(b"LIST" + struct.pack("<I", 4) + b"INFO", fmt_chunk(tag=0xFFFE, ext=1), data_chunk(rng.integers(-3000, 3000, size=64).astype("<i2").tobytes())),
        "f32": riff(fmt_chunk(tag=3, bits=32, ch=3), data_chunk(rng.standard_normal(3 * 50).astype("<f4").tobytes())),
    }
    for i, (name, blob) in enumerate(seeds.items()):
        p = tmp_path / (name + ".wav")
        p.write_bytes(blob)
        out = run_sanitized(fuzz_binary, "mutate", str(p), "1500", str(100 + i), str(tmp_path))
        assert out.strip() == "mutations 1500 rc 0"


@pytest.mark.parametrize("rows,cols", [(0, 8), (3, 5), (149, 1024)])
def test_pt_writer_under_sanitizers_is_a_torch_archive(tmp_path, fuzz_binary, rows, cols):
    p = tmp_path / "feat.pt"
    assert run_sanitized(fuzz_binary, "pt", str(p), str(rows), str(cols)).split()[0] == "0"
    t = torch.load(str(p))
    want = ((torch.arange(rows * cols) % 2001).to(torch.float32) - 1000.0) * 0.125
    assert t.dtype == torch.float32 and tuple(t.shape) == (rows, cols) and torch.equal(t.reshape(-1), want)


def test_pt_writer_refuses_bad_shapes_and_paths(tmp_path, fuzz_binary):
    for rows, cols in ((1 << 40, 1 << 40), (-1, 4), (4, 0), (1 << 28, 2), (3, 1 << 62)):
        out = run_sanitized(fuzz_binary, "pt", str(tmp_path / "x.pt"), str(rows), str(cols))
        assert int(out.split()[0]) < 0, (rows, cols, out)
        assert not (tmp_path / "x.pt").exists()
    out = run_sanitized(fuzz_binary, "pt", str(tmp_path / "no_such_dir" / "x.pt"), "2", "2")
    assert int(out.split()[0]) < 0 and "cannot open" in out


def test_malformed_files_are_logged_and_skipped_by_the_driver(tmp_path, capsys):
    """The reference's contract (preprocess_speech.py:46,72-73): a bad file is one ``Failed to process`` line, the others are written."""
    class Stub:
        pipelined = False

        def __init__(self, args, whisper, device):
            self.geo = C.TINY_WAVLM
            self.weight_source = "stub"

        def extract(self, waves, layer_index):
            return [torch.full((self.geo.frames_for(len(w)), 4), 1.0) for w in waves]

    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    cases = corpus()
    bad = ["blockalign2_60000ch", "fmt_after_data", "truncated_20", "channels_0", "junk_len_ffffffff"]
    for n in bad:
        (wav_dir / (n + ".wav")).write_bytes(cases[n][0])
    rng = np.random.default_rng(2)
    good = riff(fmt_chunk(), data_chunk((3000 * rng.standard_normal(4000)).astype("<i2").tobytes()))
    (wav_dir / "good_a.wav").write_bytes(good)
    (wav_dir / "good_b.wav").write_bytes(riff(fmt_chunk(align=0), data_chunk((3000 * rng.standard_normal(5000)).astype("<i2").tobytes())))
    out = tmp_path / "pt"
    assert driver._run(["--wav_dir", str(wav_dir), "--save_path", str(out), "--use_n_layer", "--n_layer", "1"], whisper=False,
                       extractor_factory=Stub) == 0
    log = capsys.readouterr().out
    for n in bad:
        assert f"Failed to process" in log and n + ".wav" in log, n
    assert log.count("Failed to process") == len(bad)
    assert sorted(os.listdir(out)) == ["good_a.pt", "good_b.pt"]
