"""Audio preprocessing right before the path: libohw's host DSP against (i) the known answers of the
reference's own unit tests (src/input/audio.rs:1165-1329, restated one by one) and (ii) the numpy restatement in
oracle/dsp.py on the same signals.  No GPU involved: these run in the CPU suite."""
import math

import numpy as np
import pytest

from openhush_amd import engine as E
from oracle import dsp as O


def _sine(amp, n=16000):
    i = np.arange(n, dtype=np.float32)
    return (np.float32(amp) * np.sin(np.float32(2.0) * np.float32(math.pi) * np.float32(440.0) * i / np.float32(16000.0))).astype(np.float32)


def test_resample_same_rate_and_downsample():                      # :1165-1177
    x = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    assert np.array_equal(E.resample_linear(x, 16000, 16000), x)
    r = E.resample_linear(np.arange(100, dtype=np.float32), 48000, 16000)
    assert len(r) < 100 and len(r) == 33
    assert np.array_equal(r, O.resample_linear(np.arange(100, dtype=np.float32), 48000, 16000))
    up = E.resample_linear(_sine(0.5, 1600), 16000, 48000)       # the 16 kHz -> 48 kHz leg of the RNNoise path (:996-1001)
    assert len(up) == 4800 and np.array_equal(up, O.resample_linear(_sine(0.5, 1600), 16000, 48000))


def test_rms_db_silence_and_full_scale():                           # :1189-1211
    assert math.isinf(E.AudioBuffer(np.zeros(16000, np.float32)).rms_db())
    assert math.isinf(E.AudioBuffer(np.zeros(0, np.float32)).rms_db())
    rms = E.AudioBuffer(_sine(1.0)).rms_db()
    assert abs(rms - (-3.01)) < 0.1
    assert abs(rms - O.rms_db(_sine(1.0))) < 1e-4


def test_normalize_rms():                                           # :1213-1229
    b = E.AudioBuffer(_sine(0.01))
    b.normalize_rms(-18.0)
    assert abs(b.rms_db() - (-18.0)) < 0.5
    assert np.allclose(b.samples, O.normalize_rms(_sine(0.01), -18.0), rtol=2e-6, atol=1e-9)
    z = E.AudioBuffer(np.zeros(100, np.float32))
    z.normalize_rms(-18.0)                                          # silent audio is skipped (:110-119)
    assert not z.samples.any()


def test_apply_gain():                                              # :1231-1243
    b = E.AudioBuffer(np.array([0.5, -0.5, 0.25, -0.25], np.float32))
    b.apply_gain(6.02)
    assert abs(b.samples[0] - 1.0) < 0.01 and abs(b.samples[1] + 1.0) < 0.01


def test_compress_reduces_dynamic_range():                          # :1245-1291
    x = np.concatenate([_sine(0.8), _sine(0.1)])
    b = E.AudioBuffer(x.copy())
    b.compress(-20.0, 4.0, 5.0, 50.0, 0.0)
    lb, qb = np.abs(x[:16000]).max(), np.abs(x[16000:]).max()
    la, qa = np.abs(b.samples[:16000]).max(), np.abs(b.samples[16000:]).max()
    assert la < lb and la / qa < lb / qb
    ref = O.compress(x, 16000, -20.0, 4.0, 5.0, 50.0, 0.0)
    assert np.allclose(b.samples, ref, rtol=1e-4, atol=1e-6)        # powf / log10f: last-place differences accumulate in the envelope
    r = E.AudioBuffer(x.copy())
    r.compress(-20.0, 1.0, 5.0, 50.0, 6.0)                          # ratio <= 1: untouched, not even the make-up gain (:147-149)
    assert np.array_equal(r.samples, x)


def test_limiter_prevents_clipping_and_preserves_quiet_audio():     # :1293-1329
    b = E.AudioBuffer(np.array([0.5, 1.5, -1.2, 0.8, 2.0, -0.3], np.float32))
    n = b.limit(-1.0, 50.0)
    ceiling = 10.0 ** (-1.0 / 20.0)
    assert n == 3 and np.all(np.abs(b.samples) <= ceiling + 0.01)
    assert np.allclose(b.samples, O.limit(np.array([0.5, 1.5, -1.2, 0.8, 2.0, -0.3], np.float32), 16000, -1.0, 50.0), rtol=1e-5, atol=1e-7)
    q = np.array([0.1, -0.2, 0.15, -0.05], np.float32)
    b = E.AudioBuffer(q.copy())
    assert b.limit(-1.0, 50.0) == 0 and np.all(np.abs(b.samples - q) < 0.001)


def test_preprocess_audio_chain_and_defaults():
    """TranscriptionWorker::preprocess_audio (reference src/queue/worker.rs:196-240): off by default; with the switch on,
    normalise -> compress -> limit with the reference's default settings (src/config.rs:1129-1160)."""
    cfg = E.default_preprocess_config()
    assert cfg.preprocessing == 0 and (cfg.normalization_target_db, cfg.compression_threshold_db, cfg.compression_ratio) == (-18.0, -24.0, 4.0)
    assert (cfg.compression_attack_ms, cfg.compression_release_ms, cfg.compression_makeup_gain_db) == (5.0, 50.0, 6.0)
    assert (cfg.limiter_ceiling_db, cfg.limiter_release_ms) == (-1.0, 50.0)
    x = np.concatenate([_sine(0.3, 8000), _sine(0.02, 8000)])
    b = E.AudioBuffer(x.copy())
    b.preprocess()                                                  # default config: untouched
    assert np.array_equal(b.samples, x)
    cfg.preprocessing = 1
    b.preprocess(cfg)
    ref = O.limit(O.compress(O.normalize_rms(x, -18.0), 16000, -24.0, 4.0, 5.0, 50.0, 6.0), 16000, -1.0, 50.0)
    assert np.allclose(b.samples, ref, rtol=2e-4, atol=2e-6)
    assert np.abs(b.samples).max() <= 10.0 ** (-1.0 / 20.0) + 1e-3
    cfg.compression_enabled = 0
    cfg.limiter_enabled = 0
    c = E.AudioBuffer(x.copy())
    c.preprocess(cfg)
    assert np.allclose(c.samples, O.normalize_rms(x, -18.0), rtol=2e-6, atol=1e-9)


# ---- the sinc resampler against an independent fp64 checker (oracle/dsp.py; rubato 0.16.2 is not in the reference tree:
#      PARITY UNPINNED against the crate itself) ----------------------------------------------------------------------------
@pytest.mark.parametrize("rates", [(48000, 16000), (44100, 16000), (8000, 16000), (22050, 16000), (96000, 16000), (11025, 16000)])
def test_host_sinc_resampler_matches_the_fp64_oracle(rates):
    fr, to = rates
    rng = np.random.default_rng(fr)
    for n in (1, 255, 1024, 1025, 3000, int(fr * 0.9)):               # below one chunk, exact chunks, ragged tails
        t = np.arange(n) / fr
        x = (0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3100.0 * t + 0.3) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        want = O.resample_sinc(x, fr, to)
        got = E.resample_sinc(x, fr, to)
        assert got.size == want.size == O.sinc_out_len(n, to / fr), (rates, n, got.size, want.size)
        # fp32 sums of 2 x 256 products against fp64: observed <= 1e-6 on O(1) signals
        assert np.abs(got - want).max() < 3e-6 * max(1.0, float(np.abs(want).max())), (rates, n, float(np.abs(got - want).max()))
    assert E.resample_sinc(np.zeros(0, np.float32), fr, to).size == 0 and O.resample_sinc(np.zeros(0), fr, to).size == 0


def test_sinc_table_is_a_unit_gain_low_pass():
    """what the checker itself must satisfy: every one of the 256 sub-filters sums to 1 within the table's own ripple (DC gain
    1 at every fractional position) and the table is symmetric about its centre tap"""
    for ratio in (1 / 3, 16000 / 44100, 2.0):
        h = O.sinc_table64(ratio).reshape(O.SINC_L, O.SINC_F)
        assert np.abs(h.sum(axis=0) - 1.0).max() < 2e-3
        flat = h.reshape(-1)
        assert np.abs(flat[1:] - flat[1:][::-1]).max() < 1e-12


# ---- the RNNoise stage without the network (reference src/input/audio.rs:249-341; worker.rs:197-207) ---------------------
def test_denoise_framing_matches_the_oracle_bit_for_bit():
    rng = np.random.default_rng(9)
    x = (0.2 * rng.standard_normal(16000 * 2 + 137)).astype(np.float32)        # 201 frames at 48 kHz, the last one short
    seen = []

    def net(frame):                                   # stands for DenoiseState::process_frame: sees 16-bit-range samples
        seen.append(float(np.abs(frame).max()))
        return (frame * np.float32(0.5)).astype(np.float32)
    for strength in (1.0, 0.4, 7.0):                  # 7.0 clamps to 1.0 (:254)
        a = E.AudioBuffer(x.copy(), 16000)
        d = E.Denoiser(net)
        a.denoise(strength, d)
        assert d.frames == 201
        assert np.array_equal(a.samples, O.denoise(x, 16000, strength, net)), strength
    assert max(seen) > 1000.0                         # frames really are scaled by 32767
    # pass-through engine: the up / down linear resampling alone, first 10 ms faded in
    a = E.AudioBuffer(x.copy(), 16000)
    a.denoise(1.0, E.Denoiser())
    assert np.array_equal(a.samples, O.denoise(x, 16000, 1.0, lambda f: f))
    assert np.abs(a.samples[400:-400] - x[400:-400]).max() < 1e-6 and abs(a.samples[0]) < 1e-6 and len(a.samples) == len(x)
    # strength <= 0 and an empty buffer leave the samples alone (:250-252); a 48 kHz buffer is not resampled
    b = E.AudioBuffer(x.copy(), 16000); b.denoise(0.0, E.Denoiser(net)); assert np.array_equal(b.samples, x)
    c = E.AudioBuffer(x[:4800].copy(), 48000); c.denoise(1.0, E.Denoiser(net))
    assert np.array_equal(c.samples, O.denoise(x[:4800], 48000, 1.0, net))


def test_preprocess_runs_noise_reduction_first_and_independently_of_the_switch():
    """worker.rs:197-211: denoise when noise_reduction.enabled, whatever `preprocessing` says; then the chain"""
    x = _sine(0.05, 8000)
    half = lambda f: (f * np.float32(0.5)).astype(np.float32)     # noqa: E731
    off = E.default_preprocess_config()                           # preprocessing = 0
    a = E.AudioBuffer(x.copy(), 16000)
    a.preprocess(off, noise_reduction=True, strength=1.0, denoiser=E.Denoiser(half))
    assert np.array_equal(a.samples, O.denoise(x, 16000, 1.0, half))
    on = E.default_preprocess_config(); on.preprocessing = 1
    b = E.AudioBuffer(x.copy(), 16000)
    b.preprocess(on, noise_reduction=True, strength=1.0, denoiser=E.Denoiser(half))
    c = E.AudioBuffer(O.denoise(x, 16000, 1.0, half), 16000)
    c.preprocess(on)
    assert np.array_equal(b.samples, c.samples)
    with pytest.raises(E.WhisperError):                           # enabled without an engine: an error, never a silent skip
        E.AudioBuffer(x.copy(), 16000).preprocess(on, noise_reduction=True)
    import ctypes as C
    buf = x.copy()
    assert E.lib().ohw_preprocess_audio_ex(E._fp(buf), buf.size, 16000, C.byref(on), 1, 1.0, None) == E.OHW_E_INVALID_ARG
    assert np.array_equal(buf, x)
