"""The front-end steps next to the path that are host code, no GPU needed:
the VadState segmenter with the reference's own unit tests restated (src/vad/mod.rs:252-314), the VadEngine hook and the
daemon's continuous-mode loop (src/daemon.rs:2062-2138), and the sinc resampler (src/input/audio.rs:1007-1095; rubato's
algorithm restated, parity unpinned: checked by what a band-limited resampler must do)."""
import numpy as np
import pytest

from openhush_amd import engine as E


def _cfg(**kw):
    c = E.default_vad_config()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_vad_config_defaults():
    c = E.default_vad_config()
    assert (c.enabled, c.min_silence_ms, c.min_speech_ms, c.speech_pad_ms) == (0, 700, 250, 30) and c.threshold == pytest.approx(0.5)


def test_vad_state_speech_detection():
    """reference src/vad/mod.rs:256-288"""
    st = E.VadState(_cfg(threshold=0.5, min_silence_ms=100, min_speech_ms=50), 16000)
    assert st.update(0.8, True, 512) is None and st.is_speech() and st.speech_start() == 0
    assert st.update(0.8, True, 512) is None
    assert st.update(0.1, False, 512) is None and st.is_speech()          # brief silence: not enough to end
    seg = st.update(0.1, False, 1600)                                      # more silence: the segment is returned
    assert seg is not None and not st.is_speech()
    start, end, avg = seg
    assert (start, end) == (0, 1536) and avg == pytest.approx((0.8 + 0.8 + 0.1 + 0.1) / 4)
    st.reset()
    assert st.speech_start() is None and st.update(0.1, False, 512) is None


def test_vad_state_too_short():
    """reference src/vad/mod.rs:290-313"""
    st = E.VadState(_cfg(threshold=0.5, min_silence_ms=100, min_speech_ms=500), 16000)
    st.update(0.8, True, 512)
    assert st.update(0.1, False, 3200) is None and not st.is_speech()


def test_vad_run_with_a_plugged_in_engine_and_with_the_energy_detector():
    rng = np.random.default_rng(3)
    sr = 16000
    x = (rng.standard_normal(sr * 6) * 0.001).astype(np.float32)
    x[sr:2 * sr] += (0.3 * np.sin(2 * np.pi * 220 * np.arange(sr) / sr)).astype(np.float32)            # 1.0 .. 2.0 s
    x[int(3.5 * sr):int(4.8 * sr)] += (0.3 * np.sin(2 * np.pi * 330 * np.arange(int(1.3 * sr)) / sr)).astype(np.float32)
    cfg = _cfg(min_silence_ms=500, min_speech_ms=250)
    # the hook: any callable samples -> probability (what a host's SileroVad would be)
    calls = []

    def engine(chunk):
        calls.append(len(chunk))
        return 0.9 if float(np.sqrt(np.mean(chunk * chunk))) > 0.05 else 0.05
    segs = E.vad_segments(x, cfg, poll_samples=1600, process=engine)
    assert len(segs) == 2 and set(calls) == {1600}
    (s0, e0, p0), (s1, e1, p1) = segs
    # a segment starts at the first speech poll and is cut at the poll that completed the min_silence (the reference cuts at
    # the current position: the trailing silence stays in)
    assert s0 == sr and e0 == 2 * sr + 8000 and s1 == int(3.5 * sr) and e1 == int(4.8 * sr) + 8000
    assert 0.05 < p0 < 0.9
    segs2 = E.vad_segments(x, cfg, poll_samples=1600, energy_threshold_db=-30.0)
    assert [(a, b) for a, b, _ in segs2] == [(s0, e0), (s1, e1)]
    # the reference measures a segment up to the START OF THE POLL that completed min_silence (src/vad/mod.rs:191), so 0.1 s
    # of speech followed by 0.5 s of silence polls counts as 0.5 s and passes min_speech_ms = 250; restated as it is
    y = np.zeros(sr * 2, np.float32)
    y[8000:9600] = 0.3
    assert [(a, b) for a, b, _ in E.vad_segments(y, cfg, poll_samples=1600, energy_threshold_db=-30.0)] == [(8000, 17600)]
    assert E.vad_segments(y, _cfg(min_silence_ms=100, min_speech_ms=500), poll_samples=1600, energy_threshold_db=-30.0) == []
    assert E.vad_segments(np.zeros(0, np.float32), cfg) == []


def test_sinc_resampler_is_a_band_limited_resampler():
    sr_in, sr_out = 44100, 16000
    n = sr_in * 2
    t = np.arange(n) / sr_in
    x = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    y = E.resample_sinc(x, sr_in, sr_out)
    # length: every full 1024-sample chunk gives its share, the last partial chunk ceil(len * ratio)
    assert abs(len(y) - n * sr_out / sr_in) < 520
    # a 1 kHz tone stays a 1 kHz tone of the same amplitude (after the filter's start-up, before the tail that the chunked
    # process leaves unfinished)
    seg = y[2000:-2000].astype(np.float64)
    tt = np.arange(len(seg)) / sr_out
    a = np.stack([np.sin(2 * np.pi * 1000.0 * tt), np.cos(2 * np.pi * 1000.0 * tt)], axis=1)
    coef, res, *_ = np.linalg.lstsq(a, seg, rcond=None)
    assert abs(np.hypot(*coef) - 0.5) < 2e-3 and np.sqrt(np.mean((seg - a @ coef) ** 2)) < 2e-3
    # above the new Nyquist frequency (8 kHz) nothing comes through: a 12 kHz tone is removed, not aliased to 4 kHz
    hi = (0.5 * np.sin(2 * np.pi * 12000.0 * t)).astype(np.float32)
    assert np.abs(E.resample_sinc(hi, sr_in, sr_out)[2000:-2000]).max() < 2e-3
    # DC gain 1, both directions; the same rate returns the input
    dc = np.full(20000, 0.25, np.float32)
    assert np.abs(E.resample_sinc(dc, 48000, 16000)[500:-500] - 0.25).max() < 1e-3
    assert np.abs(E.resample_sinc(dc, 8000, 16000)[1000:-1000] - 0.25).max() < 1e-3
    assert np.array_equal(E.resample_sinc(dc, 16000, 16000), dc)
    # against scipy's polyphase resampler (another windowed-sinc design) on noise band-limited well below the cut-off:
    # the two agree up to a constant delay
    import scipy.signal as ss
    rng = np.random.default_rng(5)
    noise = ss.sosfiltfilt(ss.butter(8, 3000, fs=sr_in, output="sos"), rng.standard_normal(n)).astype(np.float32)
    mine = E.resample_sinc(noise, sr_in, sr_out).astype(np.float64)
    ref = ss.resample_poly(noise.astype(np.float64), 160, 441)
    m = min(len(mine), len(ref)) - 4000
    # the chunked process has a constant sub-sample delay (the read position starts one output step late and the sub-filters
    # sit one input sample early: (1 / ratio - 1) * ratio = 0.64 output samples): compare after the best constant shift of a
    # cubic spline through scipy's result, and require that shift to be THE constant one over the whole signal
    from scipy.interpolate import CubicSpline
    cs = CubicSpline(np.arange(len(ref)), ref)
    rms = float(np.sqrt(np.mean(ref[2000:m] ** 2)))
    shifts = []
    for lo, hi in ((2000, 8000), (8000, 16000), (16000, m)):
        grid = np.arange(lo, hi)
        err, d = min((float(np.sqrt(np.mean((mine[grid] - cs(grid + d)) ** 2))), d) for d in np.linspace(-2, 2, 161))
        assert err < 0.02 * rms, (lo, err / rms)
        shifts.append(d)
    assert max(shifts) - min(shifts) < 0.06 and abs(shifts[0] - (441 / 160 - 1) * 160 / 441) < 0.06
