"""Streaming glue: the chunk scheduler against the reference's chunk-timer arm
(src/daemon.rs:1958-2011) and AudioRecorder::extract_chunk (src/input/audio.rs:737-785) on the CPU; the whole session
(scheduler -> VAD hook -> mel -> encoder -> beam search / greedy -> tracker) on the GPU."""
import numpy as np
import pytest

from openhush_amd import streaming as S
from openhush_amd import synth
from openhush_amd.tracker import BackpressureStrategy, ChunkResult, TranscriptionTracker


@pytest.fixture(params=["python", "native"])
def impl(request):
    """(extract_chunk, tracker class, scheduler class): the Python mirror and the C++ code behind the C ABI (csrc/tracker.cpp)"""
    from openhush_amd.tracker import NativeTranscriptionTracker
    if request.param == "python":
        return S.extract_chunk, TranscriptionTracker, S.ChunkScheduler
    return S.native_extract_chunk, NativeTranscriptionTracker, S.NativeChunkScheduler


def test_extract_chunk_lengths(impl):
    extract_chunk = impl[0]
    rec = np.arange(16000 * 8, dtype=np.float32)
    assert extract_chunk(rec, 0, 0) is None                              # empty
    assert extract_chunk(rec, 100, 100 + 1599) is None                     # 0.0999 s < 0.1 s
    c = extract_chunk(rec, 100, 100 + 1600)                              # exactly 0.1 s: kept, padded to 1.1 s
    assert c.size == 17600 and np.array_equal(c[:1600], rec[100:1700]) and not c[1600:].any()
    c = extract_chunk(rec, 0, 17599)                                     # 1.0999 s: padded by one sample
    assert c.size == 17600 and c[-1] == 0.0
    c = extract_chunk(rec, 0, 17600)
    assert c.size == 17600 and c[-1] == rec[17599]
    c = extract_chunk(rec, 16000, 16000 * 6)                             # 5 s: as it is
    assert c.size == 80000 and np.array_equal(c, rec[16000:96000])


def test_scheduler_positions_and_ids(impl):
    _, Tracker, Scheduler = impl
    tr = Tracker()
    sch = Scheduler(tr, sequence_id=7)
    rec = np.zeros(16000 * 20, np.float32)
    assert sch.tick(rec, 800) is None                                      # too short: neither the position nor the id moves
    assert (sch.last_chunk_pos, sch.next_chunk_id) == (0, 0)
    j = sch.tick(rec, 80000)
    assert (j.sequence_id, j.chunk_id, j.samples.size, j.is_final) == (7, 0, 80000, False)
    assert (sch.last_chunk_pos, sch.next_chunk_id) == (80000, 1)
    j = sch.tick(rec, 88000, is_final=True)                                # 0.5 s: padded
    assert (j.chunk_id, j.samples.size, j.is_final) == (1, 17600, True)
    assert tr.pending_count() == 2


def test_scheduler_backpressure_drop_newest_still_advances(impl):
    _, Tracker, Scheduler = impl
    tr = Tracker()
    sch = Scheduler(tr, 1, max_pending=2, high_water_mark=1, strategy=BackpressureStrategy.DROP_NEWEST)
    rec = np.zeros(16000 * 30, np.float32)
    assert sch.tick(rec, 16000 * 5) is not None and sch.tick(rec, 16000 * 10) is not None
    assert sch.tick(rec, 16000 * 15) is None                               # refused: "skip submitting the job but still update state"
    assert (sch.last_chunk_pos, sch.next_chunk_id, sch.rejected) == (16000 * 15, 3, 1)
    tr.add_result(ChunkResult("a", 1, 0)); tr.take_ready()
    j = sch.tick(rec, 16000 * 20)
    assert j is not None and j.chunk_id == 3 and j.samples.size == 16000 * 5


@pytest.mark.gpu
@pytest.mark.parametrize("beam", [5, 0])
def test_streaming_session_matches_single_windows(tmp_models, beam):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from openhush_amd import engine as E
    path = tmp_models("micro")
    ctx = E.Context.from_file(path, 0, E.OHW_DTYPE_F16)
    p = ctx.default_params(); p.n_max = 16
    # a 20.3 s recording: speech, 6 s of silence, speech; chunk timer every 5 s, the release 0.3 s after the last tick
    rec = np.concatenate([synth.synth_audio(5)[:16000 * 9], np.zeros(16000 * 6, np.float32), synth.synth_audio(9)[:int(16000 * 5.3)]])
    calls = []

    def vad(s):                                   # the VadEngine hook: RMS level of the chunk
        calls.append(len(s))
        return 1.0 if float(np.sqrt(np.mean(s.astype(np.float64) ** 2))) > 1e-3 else 0.0

    ses = S.StreamingSession(ctx, beam_size=beam, vad=vad, sequence_id=3, params=p)
    out = []
    ticks = list(range(16000 * 5, len(rec), 16000 * 5)) + [len(rec)]
    for i, pos in enumerate(ticks):
        out += ses.tick(rec, pos, is_final=(i == len(ticks) - 1))
    assert [r.chunk_id for r in out] == list(range(len(ticks))) and all(r.sequence_id == 3 for r in out)
    assert out[-1].is_final and abs(out[-1].duration_secs - 1.1) < 1e-6        # the 0.3 s tail was padded to 1.1 s
    assert ses.skipped_silent == 1 and out[2].text == ""                        # chunk 2 = [10 s, 15 s) is all silence
    assert ses.tracker.is_empty()
    # every decoded chunk equals the same samples taken through a fresh state as one window
    st = E.State(ctx, max(1, beam))
    tr = TranscriptionTracker()
    last = 0
    for i, pos in enumerate(ticks):
        s = S.extract_chunk(rec, last, pos); last = pos
        if i == 2:
            continue
        st.mel(s[None, :], [len(s)], E.OHW_MEL_ZERO_TAIL, want=False)
        st.encode(1)
        toks = st.beam_search(1, beam, p)[0]["tokens"] if beam else st.greedy(1, p)[0][0]
        text = b"".join(ctx.token_text(t) for t in toks if t < ctx.tok.eot).decode("utf-8", "replace").strip()
        tr.add_result(ChunkResult(text, 3, i))
        want = tr.take_ready()[0].text                                          # the tracker's overlap de-duplication applied
        assert out[i].text == want, (i, out[i].text, want)
    assert any(r.text for r in out)


@pytest.mark.gpu
def test_streaming_job_longer_than_30s_and_the_noise_reduction_stage(tmp_models):
    """(i) A job longer than 30 s (a late timer tick, a long VAD segment) is transcribed whole, window after window - the
    reference's worker hands the whole buffer to engine.transcribe (src/queue/worker.rs:152); round 2 dropped everything after
    30 s.  (ii) BASELINE config #5's stage order with a plugged-in denoiser: noise reduction -> (normalise / compress / limit)
    -> VAD -> decode, i.e. what reaches the engine is the preprocessed buffer (reference worker.rs:147-152, 197-240)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from openhush_amd import engine as E
    ctx = E.Context.from_file(tmp_models("micro"), 0, E.OHW_DTYPE_F16)
    p = ctx.default_params(); p.n_max = 12
    rec = np.concatenate([synth.synth_audio(31), synth.synth_audio(32), synth.synth_audio(33)[:16000 * 7]])       # 67 s
    ses = S.StreamingSession(ctx, beam_size=0, sequence_id=5, params=p)
    out = ses.tick(rec, len(rec), is_final=True)                       # ONE tick after 67 s: one job of three windows
    assert len(out) == 1 and ses.windows_decoded == 3 and abs(out[0].duration_secs - 67.0) < 1e-6
    st = E.State(ctx, 1)
    parts = []
    for off in range(0, len(rec), E.CHUNK_SAMPLES):
        s = rec[off:off + E.CHUNK_SAMPLES]
        st.mel(s[None, :], [len(s)], E.OHW_MEL_ZERO_TAIL, want=False); st.encode(1)
        parts.append(b"".join(ctx.token_text(t) for t in st.greedy(1, p)[0][0] if t < ctx.tok.eot))
    assert out[0].text == b"".join(parts).decode("utf-8", "replace").strip() and len(parts[2]) > 0
    # (ii) the denoiser halves every frame, then the chain normalises to -18 dB: the engine must see exactly that buffer
    cfg = E.default_preprocess_config(); cfg.preprocessing = 1
    half = lambda f: (f * np.float32(0.5)).astype(np.float32)           # noqa: E731
    chunk = synth.synth_audio(41)[:16000 * 5]
    seen = []

    def vad(s):
        seen.append(np.array(s))
        return 1.0
    ses2 = S.StreamingSession(ctx, beam_size=5, vad=vad, sequence_id=6, params=p, audio_config=cfg, noise_reduction=True,
                              noise_reduction_strength=1.0, denoiser=E.Denoiser(half))
    got = ses2.tick(chunk, len(chunk), is_final=True)
    want_buf = E.AudioBuffer(chunk.copy(), 16000)
    want_buf.preprocess(cfg, True, 1.0, E.Denoiser(half))
    assert np.array_equal(seen[0], want_buf.samples) and not np.array_equal(seen[0], chunk)
    st5 = E.State(ctx, 5)
    st5.mel(want_buf.samples[None, :], [len(chunk)], E.OHW_MEL_ZERO_TAIL, want=False); st5.encode(1)
    toks = st5.beam_search(1, 5, p)[0]["tokens"]
    assert got[0].text == b"".join(ctx.token_text(t) for t in toks if t < ctx.tok.eot).decode("utf-8", "replace").strip()
    with pytest.raises(E.WhisperError):
        S.StreamingSession(ctx, noise_reduction=True)                   # enabled without an engine
    st.close(); st5.close()


def test_energy_vad_hook_gates_silence():
    from openhush_amd import engine as E
    v = E.EnergyVad(-40.0)
    tone = (0.1 * np.sin(np.arange(16000) * 0.1)).astype(np.float32)
    assert v(np.zeros(16000, np.float32)) < 0.01 < 0.9 < v(tone)
    assert v(np.zeros(0, np.float32)) == 0.0
    v.close()
