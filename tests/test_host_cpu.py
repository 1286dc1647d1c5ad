"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, and the host logic that needs no GPU behaves like the reference's own unit tests say."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from openhush_amd import engine as E


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ohw.h")).read()
    declared = set(re.findall(r"\b(ohw_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = E.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert missing == []
    assert declared == set(E.EXPORTS), declared ^ set(E.EXPORTS)
    assert lib.ohw_abi_version() == 2


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from openhush_amd import synth
    with pytest.raises(E.LoadFailed) as ei:
        E.Context.synthetic(synth.PRESETS["nano"].as_list())
    assert ei.value.code == E.OHW_E_NO_GPU


# reference src/engine/validation.rs:139-203
def test_validate_audio_reference_cases():
    info = E.validate_audio(np.zeros(16000, np.float32), 16000)
    assert abs(info.duration_secs - 1.0) < 0.01
    for samples, rate, kind in ((np.zeros(0, np.float32), 16000, "Empty"), (np.zeros(800, np.float32), 16000, "TooShort"),
                                (np.zeros(44100, np.float32), 44100, "InvalidSampleRate")):
        with pytest.raises(E.ValidationFailed) as ei:
            E.validate_audio(samples, rate)
        assert ei.value.kind == kind
    s = np.zeros(16000, np.float32); s[500] = np.nan; s[1000] = np.nan
    with pytest.raises(E.ValidationFailed) as ei:
        E.validate_audio(s, 16000)
    assert ei.value.kind == "ContainsNaN" and ei.value.info.nan_count == 2
    s = np.zeros(16000, np.float32); s[500] = np.inf
    with pytest.raises(E.ValidationFailed) as ei:
        E.validate_audio(s, 16000)
    assert ei.value.kind == "ContainsInfinite" and ei.value.info.inf_count == 1
    with pytest.raises(E.ValidationFailed) as ei:
        E.validate_audio(np.zeros(16000 * 7201, np.float32), 16000)
    assert ei.value.kind == "TooLong"


def test_validate_audio_duration_bound_is_tested_in_f32_like_the_reference():
    """`samples.len() as f32 / sample_rate as f32 > 7200.0` (src/engine/validation.rs:64-72): two hours + 4 samples still reads
    7 200.0 and is valid, + 8 samples (7 200.0005) is the first length that is too long"""
    n = 7200 * 16000
    z = np.zeros(n + 8, np.float32)
    z[::16000] = 0.25                                   # not silent: only the duration is under test
    info = E.validate_audio(z[:n + 4], 16000)
    assert info.sample_count == n + 4 and info.duration_secs == 7200.0
    with pytest.raises(E.ValidationFailed) as ei:
        E.validate_audio(z, 16000)
    assert ei.value.kind == "TooLong"


def test_validate_audio_matches_oracle_statistics():
    from oracle import oracle
    rng = np.random.default_rng(3)
    s = (rng.standard_normal(50000) * 0.2).astype(np.float32)
    info = E.validate_audio(s, 16000)
    code, ref = oracle.validate_audio(s, 16000)
    assert code == "Ok"
    assert (info.min_value, info.max_value, info.rms, info.sample_count) == (ref.min_value, ref.max_value, ref.rms, ref.sample_count)


# reference src/engine/whisper.rs:852-878
def test_lang_id_to_code_reference_samples():
    for i, c in ((0, "en"), (2, "de"), (6, "fr"), (7, "ja"), (15, "it"), (1, "zh"), (10, "pl"), (20, "he"), (30, "th"),
                 (50, "br"), (70, "ka"), (93, "haw"), (98, "su")):
        assert E.lang_id_to_code(i) == c
        assert E.lang_code_to_id(c) == i
    assert E.lang_id_to_code(999) == "unknown" and E.lang_id_to_code(-1) == "unknown"
    assert E.lang_id_to_code(99) == "unknown"   # large-v3's "yue": SURVEY.md A5
    assert E.lang_code_to_id("xx") == -1


# reference src/engine/whisper.rs:56-79, tests :741-783
def test_model_names():
    assert E.model_filename("tiny") == "ggml-tiny.bin"
    assert E.model_filename("LARGE") == "ggml-large-v3.bin"
    assert E.model_filename("large-v3") == "ggml-large-v3.bin" and E.model_filename("largev3") == "ggml-large-v3.bin"
    with pytest.raises(KeyError):
        E.model_filename("tiny.en")     # the reference has no .en models


# reference src/engine/whisper.rs tests :786-831 (known answers of the reference's own unit tests)
def test_model_sizes_and_format_size():
    assert [E.model_size_bytes(m) for m in ("tiny", "base", "small", "medium", "large-v3")] == \
        [75_000_000, 142_000_000, 466_000_000, 1_500_000_000, 3_000_000_000]
    assert E.format_size(500) == "500 B" and E.format_size(1023) == "1023 B"
    assert E.format_size(1024) == "1 KB" and E.format_size(5120) == "5 KB"
    assert E.format_size(1024 * 1024) == "1 MB" and E.format_size(75_000_000) == "72 MB" and E.format_size(500 * 1024 * 1024) == "500 MB"
    assert E.format_size(1024 ** 3) == "1.0 GB" and E.format_size(3_000_000_000) == "2.8 GB"


def test_from_config_resolves_model_path_and_device(tmp_path):
    """WhisperEngine::from_config (src/engine/whisper.rs:183-201) with the reference's own known answers for effective_model()
    (src/config.rs:1592-1622): preset -> model, custom -> the explicit model, an unknown name -> base, device "cpu" in any case
    -> use_gpu false; the engine call itself then reports the missing file before any device work"""
    T = E.TranscriptionConfig
    assert T(preset="instant").effective_model() == "small" and T(preset="balanced").effective_model() == "medium"
    assert T(preset="quality").effective_model() == "large-v3" and T(preset="custom", model="tiny").effective_model() == "tiny"
    assert T().effective_model() == "medium" and T().language == "auto" and T().device == "cuda"       # the defaults
    path, gpu = T(preset="quality").engine_arguments(str(tmp_path))
    assert path == str(tmp_path / "models" / "ggml-large-v3.bin") and gpu is True
    path, gpu = T(preset="custom", model="no-such-model", device="CPU").engine_arguments(str(tmp_path))
    assert path == str(tmp_path / "models" / "ggml-base.bin") and gpu is False
    with pytest.raises(E.ModelNotFound) as ei:
        E.WhisperEngine.from_config(T(preset="instant", language="de"), str(tmp_path))
    assert "ggml-small.bin" in str(ei.value) and "openhush model download small" in str(ei.value)


def test_missing_model_is_reported_before_any_device_work(tmp_path):
    # reference src/engine/whisper.rs:141-154 and test :984-997
    with pytest.raises(E.ModelNotFound) as ei:
        E.WhisperEngine.new(str(tmp_path / "ggml-medium.bin"), "auto", False, True)
    assert ei.value.code == E.OHW_E_MODEL_NOT_FOUND and "openhush model download medium" in str(ei.value)


def test_oracle_is_not_reachable_from_the_product_package():
    pkg = os.path.join(ROOT, "openhush_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                # comments may NAME the oracle (as the spec twin of a generator); nothing may import, include or load it
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libwhisper_ref" not in src and not re.search(r"#include\s+[\"<].*whisper_ref", src), f
