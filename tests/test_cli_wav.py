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
    _write_wav(tmp_path / "r.wav", x, rate=44100)
    with pytest.raises(ValueError):
        cli.load_wav_file(str(tmp_path / "r.wav"))


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
