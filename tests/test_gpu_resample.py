"""The sinc resampler on the device (include/ohw.h, ohw_resampler_*) against the ORACLE's independent fp64
restatement of the reference's resample(.., ResamplingQuality::High) (oracle/dsp.py resample_sinc; reference
src/input/audio.rs:1007-1095; rubato 0.16.2 itself is not in the reference tree: parity unpinned against the crate) and
against the library's host version: the same number of samples and the same samples up to fp32 summation, for the capture
rates the reference meets; then the 48 kHz -> 16 kHz output feeds the log-mel without leaving the device."""
import numpy as np
import pytest

from openhush_amd import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from openhush_amd import engine
    engine.lib()
    return engine


def _tone(rate, secs, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(rate * secs)) / rate
    x = 0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3100.0 * t + 0.3) + 0.05 * rng.standard_normal(t.size)
    return x.astype(np.float32)


@pytest.mark.parametrize("rates", [(48000, 16000), (44100, 16000), (8000, 16000), (22050, 16000), (16000, 16000)])
def test_device_resampler_matches_host(E, rates):
    fr, to = rates
    rs = E.DeviceResampler(fr, to)
    for n in (1, 700, 1024, 1025, 5000, int(fr * 3.3)):            # below one chunk, exact chunks, a ragged tail
        x = _tone(fr, 4.0, n)[:n]
        from oracle import dsp as O
        ref = O.resample_sinc(x, fr, to)                 # the checker: fp64, written independently of both product versions
        want = E.resample_sinc(x, fr, to)
        got = rs.run(x)
        assert got.size == want.size == ref.size == rs.out_len(n) == (n if fr == to else O.sinc_out_len(n, to / fr)), (rates, n, got.size, want.size, ref.size)
        if want.size:
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.abs(got - ref).max() < 3e-6 * scale, (rates, n, float(np.abs(got - ref).max()))       # device vs oracle
            assert np.abs(want - ref).max() < 3e-6 * scale, (rates, n, float(np.abs(want - ref).max()))     # host vs oracle
            assert np.abs(got - want).max() < 2e-6 * scale, (rates, n, float(np.abs(got - want).max()))
    assert rs.out_len(0) == 0
    rs.close()


def test_resampled_audio_feeds_the_log_mel_on_the_device(E, tmp_models):
    """48 kHz PCM in HBM -> resampler -> ohw_mel with pcm_on_device: nothing goes back to the host in between"""
    ctx = E.Context.from_file(tmp_models("micro"), 0, E.OHW_DTYPE_F16)
    st = E.State(ctx, 1)
    x48 = _tone(48000, 12.0, 5)
    rs = E.DeviceResampler(48000, 16000)
    n16 = rs.out_len(x48.size)
    d_in = torch.from_numpy(x48).cuda()
    d_out = torch.zeros(synth.CHUNK_SAMPLES, dtype=torch.float32, device="cuda")
    assert rs.run_device(d_in.data_ptr(), x48.size, d_out.data_ptr(), d_out.numel(), torch.cuda.current_stream().cuda_stream) == n16
    torch.cuda.synchronize()
    host16 = E.resample_sinc(x48, 48000, 16000)
    assert np.abs(d_out[:n16].cpu().numpy() - host16).max() < 2e-6
    st.mel_device(d_out.data_ptr(), synth.CHUNK_SAMPLES, [n16], E.OHW_MEL_ZERO_TAIL)
    mel_dev = st.fetch("mel", 1)[0]
    mel_host = st.mel(host16[None, :], [n16], E.OHW_MEL_ZERO_TAIL)[0]
    assert np.abs(mel_dev - mel_host).max() < 1e-4
    with pytest.raises(E.WhisperError):
        E.DeviceResampler(16000, 1000000)            # ratio outside 1/16 .. 16
    rs.close(); st.close()
