"""The CPU oracle (oracle/whisper_ref.c) against golden vectors.

Goldens: tests/golden/{micro,micro-v3,sampler}.npz written by tools/gen_golden.py from the
`transformers` Whisper implementation (fp32) — an independent implementation of the published
model.  validate_audio cases restate the reference's own unit tests
(reference src/engine/validation.rs:139-203).  Tolerances are absolute, stated per check.
"""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN
from openhush_amd import synth
from oracle import oracle


@pytest.fixture(scope="module", params=["micro", "micro-v3"])
def case(request, tmp_models):
    g = np.load(os.path.join(GOLDEN, f"{request.param}.npz"))
    m = oracle.Model.load(tmp_models(request.param, int(g["seed"])))
    return request.param, g, m


def _pcm(g, tag):
    if tag == "a":
        return g["pcm_a"]
    pcm = synth.synth_audio(7)
    # regenerated with libm calls: allow last-bit drift, but the statistics must agree
    st = g["pcm_b_stats"]
    assert abs(pcm.astype(np.float64).sum() - st[0]) < 1e-2
    return pcm


@pytest.mark.parametrize("tag", ["a", "b"])
def test_log_mel_matches_feature_extractor(case, tag):
    _, g, m = case
    mel = m.log_mel(_pcm(g, tag), mode=0)
    assert mel.shape == (m.n_mels, 3000)
    # log10 of fp64 power spectra, then (x+4)/4 in f32: 2e-4 abs covers the complex64 spectrum rounding
    assert np.abs(mel[:, ::10] - g[f"mel_{tag}_sub"]).max() < 2e-4
    assert np.abs(mel[:, -4:] - g[f"mel_{tag}_last"]).max() < 2e-4
    st = g[f"mel_{tag}_stats"]
    assert abs(mel.astype(np.float64).sum() - st[0]) / mel.size < 1e-5
    assert abs(mel.max() - st[3]) < 2e-4 and abs(mel.min() - st[2]) < 2e-4


def test_mel_mode1_differs_only_in_tail_frames(case):
    _, g, m = case
    pcm = _pcm(g, "b")
    a, b = m.log_mel(pcm, 0), m.log_mel(pcm, 1)
    # zero tail (whisper.cpp convention) vs reflect: only frames whose window crosses sample 480000
    assert np.array_equal(a[:, :2998], b[:, :2998])
    assert not np.array_equal(a[:, 2999], b[:, 2999])
    pcm3 = g["pcm_a"]  # 3 s: tail is zeros either way
    assert np.array_equal(m.log_mel(pcm3, 0), m.log_mel(pcm3, 1))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_encoder_and_decoder_match(case, tag):
    _, g, m = case
    mel = m.log_mel(_pcm(g, tag), mode=0)
    enc, conv1, stem, block0 = m.encode(mel, taps=True)
    # fp32 end to end; differences are summation order only
    assert np.abs(conv1[::50] - g[f"conv1_{tag}_sub"]).max() < 2e-4
    assert np.abs(stem[::25] - g[f"stem_{tag}_sub"]).max() < 3e-4
    assert np.abs(block0[::25] - g[f"block0_{tag}_sub"]).max() < 1e-3
    assert np.abs(enc[::10] - g[f"enc_{tag}_sub"]).max() < 1e-3
    st = oracle.State(m)
    st.set_encoder_output(enc)
    xk, xv = st.cross_kv()
    assert np.abs(xk[0][::25] - g[f"xk0_{tag}_sub"]).max() < 1e-3
    assert np.abs(xv[-1][::25] - g[f"xvl_{tag}_sub"]).max() < 1e-3
    forced = [int(t) for t in g["forced_tokens"]]
    logits = st.decode(forced, 0, all_pos=True)
    cols = g["logit_cols"]
    # logits have sigma ~ 4: 5e-3 abs is ~1e-3 relative
    assert np.abs(logits[:, cols] - g[f"logits_{tag}_cols"]).max() < 5e-3
    top = np.take_along_axis(logits, g[f"logits_{tag}_top_idx"], axis=1)
    assert np.abs(top - g[f"logits_{tag}_top_val"]).max() < 5e-3
    assert np.array_equal(logits.argmax(1), g[f"logits_{tag}_top_idx"][:, 0])
    # incremental decoding with the KV cache gives the same last-position logits as the full pass
    st2 = oracle.State(m)
    st2.set_encoder_output(enc)
    st2.decode(forced[:3], 0)
    for i in range(3, len(forced)):
        last = st2.decode([forced[i]], i)
    assert np.abs(last - logits[-1]).max() < 1e-4


def test_special_tokens_follow_vocab_size(case):
    name, _, m = case
    if name == "micro":
        assert (m.tok_eot, m.tok_sot, m.tok_translate, m.tok_transcribe, m.tok_not, m.tok_beg) == (50257, 50258, 50358, 50359, 50363, 50364)
        assert m.n_langs == 99
    else:
        assert (m.tok_eot, m.tok_sot, m.tok_translate, m.tok_transcribe, m.tok_not, m.tok_beg) == (50257, 50258, 50359, 50360, 50364, 50365)
        assert m.n_langs == 100
    assert m.tok_blank == 220
    p = m.default_params()
    assert m.build_prompt(p) == [m.tok_sot, m.tok_sot + 1, m.tok_transcribe]
    assert p.n_max == 220 and p.max_initial_ts == 50


def test_logits_filter_matches_timestamp_processor(tmp_models):
    g = np.load(os.path.join(GOLDEN, "sampler.npz"))
    m = oracle.Model.load(tmp_models("nano"))
    p = m.default_params()
    p.suppress_blank = 0
    rows = g["rows_f16"].astype(np.float32)
    for r in range(rows.shape[0]):
        h = [int(t) for t in g["hists"][g["hist_of_row"][r]] if t >= 0]
        tok, lp, filt, lps = m.process_logits(p, rows[r], h)
        assert tok == int(g["argmax"][r]), (r, h)
        fin = np.isfinite(filt)
        # Known, documented difference: after a closed segment the processor forbids
        # re-emitting the last timestamp (<= last), whisper.cpp's seek_delta rule forbids only < last.
        ts = [t for t in h if t >= m.tok_beg]
        opening = len(h) >= 1 and h[-1] >= m.tok_beg and (len(h) >= 2 and h[-2] < m.tok_beg)
        extra = 1 if (ts and not opening and np.isfinite(rows[r][ts[-1]]) and fin[ts[-1]]) else 0
        assert int(fin.sum()) - extra == int(g["finite_count"][r]), (r, h)
        if extra:
            fin = fin.copy(); fin[ts[-1]] = False
        assert zlib.crc32(np.packbits(fin).tobytes()) == int(g["finite_crc"][r])
        # logprobs are normalised BEFORE the timestamp-mass rule masks text tokens (whisper.cpp order),
        # so they sum to <= 1 afterwards
        keep = np.isfinite(lps)
        assert np.exp(lps[keep].astype(np.float64)).sum() < 1.0 + 1e-4


# ---- validate_audio: the reference's own cases (validation.rs:139-203) -------------------------
def test_validate_audio_reference_cases():
    code, info = oracle.validate_audio(np.zeros(16000, np.float32), 16000)
    assert code == "Ok" and abs(info.duration_secs - 1.0) < 0.01
    assert oracle.validate_audio(np.zeros(0, np.float32), 16000)[0] == "Empty"
    assert oracle.validate_audio(np.zeros(800, np.float32), 16000)[0] == "TooShort"
    s = np.zeros(16000, np.float32); s[500] = np.nan; s[1000] = np.nan
    code, info = oracle.validate_audio(s, 16000)
    assert code == "ContainsNaN" and info.nan_count == 2
    s = np.zeros(16000, np.float32); s[500] = np.inf
    code, info = oracle.validate_audio(s, 16000)
    assert code == "ContainsInfinite" and info.inf_count == 1
    assert oracle.validate_audio(np.zeros(44100, np.float32), 44100)[0] == "InvalidSampleRate"


def test_validate_audio_too_long_is_rejected_before_the_scan():
    # 7201 s (validation.rs:166-175); np.zeros is lazily mapped, like the reference's zeroed Vec
    s = np.zeros(16000 * 7201, np.float32)
    assert oracle.validate_audio(s, 16000)[0] == "TooLong"


def test_validate_audio_stats():
    s = np.array([0.5, -0.25, 0.125, 0.0] * 1000, np.float32)
    code, info = oracle.validate_audio(s, 16000)
    assert code == "Ok"
    assert info.min_value == -0.25 and info.max_value == 0.5 and info.sample_count == 4000
    assert abs(info.rms - np.sqrt((s.astype(np.float64) ** 2).mean())) < 1e-7
