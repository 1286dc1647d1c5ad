"""WAV input contract of the one-shot CLI (reference src/input/audio.rs:348-434) and, on a GPU, the
`transcribe` plumbing of BASELINE.json config #1 with the reference's JSON fields (src/main.rs:1054-1066)."""
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

from conftest import ROOT
from openhush_amd import cli, synth


def _write_wav(path, data_i16, rate=16000, ch=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(np.asarray(data_i16, "<i2").tobytes())


def test_int16_scaling_channel_average_and_min_duration_pad(tmp_path):
    x = np.array([0, 16384, -32768, 32767] * 1000, np.int16)
    _write_wav(tmp_path / "m.wav", x)
    s = cli.load_wav_file(str(tmp_path / "m.wav"))
    assert s.dtype == np.float32 and len(s) == 17600            # 4000 samples padded to 1.1 s
    assert np.array_equal(s[:4], np.array([0.0, 0.5, -1.0, 32767 / 32768], np.float32)) and not s[4000:].any()
    st = np.stack([np.full(20000, 8192, np.int16), np.full(20000, -8192, np.int16)], axis=1).reshape(-1)
    _write_wav(tmp_path / "s.wav", st, ch=2)
    s2 = cli.load_wav_file(str(tmp_path / "s.wav"))
    assert len(s2) == 20000 and not s2.any()                     # L/R average


def test_other_rates_are_resampled_and_float_files_are_read(tmp_path):
    """reference src/input/audio.rs:369-407: any rate is resampled to 16 kHz (High = sinc by default, Low = linear), float
    files are taken as they are"""
    from openhush_amd import engine as E
    t = np.arange(44100 * 2) / 44100.0
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    _write_wav(tmp_path / "r.wav", np.round(tone * 32767).astype(np.int16), rate=44100)
    want = np.round(tone * 32767).astype(np.int16).astype(np.float32) / np.float32(32768)
    hi = cli.load_wav_file(str(tmp_path / "r.wav"))
    assert np.array_equal(hi, E.resample_sinc(want, 44100, 16000)) and abs(len(hi) - 32000) < 520
    lo = cli.load_wav_file(str(tmp_path / "r.wav"), "low")
    assert np.array_equal(lo, E.resample_linear(want, 44100, 16000)) and len(lo) == 32000
    seg = hi[2000:-2000].astype(np.float64)
    tt = np.arange(len(seg)) / 16000.0
    a = np.stack([np.sin(2 * np.pi * 1000.0 * tt), np.cos(2 * np.pi * 1000.0 * tt)], axis=1)
    coef = np.linalg.lstsq(a, seg, rcond=None)[0]
    assert abs(np.hypot(*coef) - 0.5) < 2e-3                     # still a 1 kHz tone of the same level
    # 32-bit float, stereo, 48 kHz (format tag 3)
    import struct
    st = np.stack([np.full(96000, 0.25, np.float32), np.full(96000, -0.75, np.float32)], axis=1).reshape(-1)
    body = st.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, 2, 48000, 48000 * 8, 8, 32)
    with open(tmp_path / "f.wav", "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt +
                b"data" + struct.pack("<I", len(body)) + body)
    fl = cli.load_wav_file(str(tmp_path / "f.wav"))
    assert abs(len(fl) - 32000) < 520 and np.abs(fl[500:-500] + 0.25).max() < 1e-3      # (0.25 - 0.75) / 2, DC gain 1
    # 24-bit and 8-bit integer files
    v = np.array([0, 1 << 22, -(1 << 23), (1 << 23) - 1] * 5000, np.int32)
    b24 = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 48000, 3, 24)
    with open(tmp_path / "i24.wav", "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(b24)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt +
                b"data" + struct.pack("<I", len(b24)) + b24)
    s24 = cli.load_wav_file(str(tmp_path / "i24.wav"))
    assert np.array_equal(s24[:4], np.array([0.0, 0.5, -1.0, ((1 << 23) - 1) / float(1 << 23)], np.float32))
    with open(tmp_path / "bad.wav", "wb") as f:
        f.write(b"not a wav file at all")
    with pytest.raises(ValueError):
        cli.load_wav_file(str(tmp_path / "bad.wav"))


def test_cli_surface_like_the_reference_cli_tests(tmp_path):
    """reference tests/cli_integration.rs:252-269: `transcribe --help` lists --model and --format; a missing file fails
    with 'not found' on stderr (no device is touched on either path)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    h = subprocess.run([sys.executable, "-m", "openhush_amd.cli", "transcribe", "--help"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert h.returncode == 0 and "--model" in h.stdout and "--format" in h.stdout
    m = subprocess.run([sys.executable, "-m", "openhush_amd.cli", "transcribe", "nonexistent.wav", "--model-path", str(tmp_path / "ggml-tiny.bin")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert m.returncode != 0 and ("not found" in m.stderr or "error" in m.stderr)


@pytest.mark.gpu
def test_transcribe_cli_json(tmp_path, tmp_models):
    pcm = synth.synth_audio(5, 160000)                            # the 10 s file of config #1
    _write_wav(tmp_path / "ten.wav", np.round(pcm * 32767).astype(np.int16))
    out = subprocess.run([sys.executable, "-m", "openhush_amd.cli", "transcribe", str(tmp_path / "ten.wav"), "--model-path",
                          tmp_models("micro"), "--format", "json", "--dtype", "f16"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    j = json.loads(out.stdout)
    assert set(j) == {"text", "language", "duration_ms", "audio_duration_secs", "transcription_time_ms", "real_time_factor", "model"}
    assert j["language"] == "en" and abs(j["audio_duration_secs"] - 10.0) < 1e-6 and j["model"] == "micro-s1234"
    assert j["real_time_factor"] > 0 and isinstance(j["text"], str)
    # a 44.1 kHz file of the same 10 s (reference :394-404: resampled, not rejected): same duration within the resampler's
    # chunk tail, and the text of the engine run on the loader's own 16 kHz samples
    from openhush_amd import cli, engine as E
    t441 = np.arange(441000) / 44100.0
    src = (0.4 * np.sin(2 * np.pi * 220.0 * t441) * (1.0 + 0.5 * np.sin(2 * np.pi * 3.0 * t441))).astype(np.float32)
    _write_wav(tmp_path / "ten441.wav", np.round(src * 32767).astype(np.int16), rate=44100)
    out2 = subprocess.run([sys.executable, "-m", "openhush_amd.cli", "transcribe", str(tmp_path / "ten441.wav"), "--model-path",
                           tmp_models("micro"), "--format", "json", "--dtype", "f16"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr
    j2 = json.loads(out2.stdout)
    assert abs(j2["audio_duration_secs"] - 10.0) < 0.04
    eng = E.WhisperEngine.new(tmp_models("micro"), "auto", False, True, 0, E.OHW_DTYPE_F16, 8)
    assert eng.transcribe(E.AudioBuffer(cli.load_wav_file(str(tmp_path / "ten441.wav")), 16000)).text == j2["text"]
    eng.close()
