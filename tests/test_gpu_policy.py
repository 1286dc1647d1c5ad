"""whisper.cpp's per-window decode policy in the engine against the oracle's restatement: the loop exits,
result_len / seek_delta, the acceptance test, the no-speech rule and the temperature ladder sampled on the host with
std::mt19937 + std::discrete_distribution (the oracle draws with its own hand-written Mersenne twister).

How passes are compared: the engine reports every decode pass of a transcribe (ohw_engine_last_trace).  The oracle walks
each pass along the engine's own tokens (forced) with the SAME generator state - a pass consumes one draw per token on
both sides - and records what it would have picked at every step: at T = 0 a different pick is legitimate only inside
the logit tolerance (top-2 margin), at T > 0 only where the draw fell within 0.02 of an interval edge of the oracle's
own cumulative distribution (f16 logits are off by up to ~0.01: every probability moves by ~1 %, an edge of the cumulative
sum by up to that share of the mass; 0.0025 was observed at T = 0.8), and at least 98 % of all steps must agree outright.  The decisions (retry at the next temperature, kept
tokens, no-speech) must then be identical.
"""
import numpy as np
import pytest

from openhush_amd import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TOL = 0.03          # f16 logits (test_gpu_parity)


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from openhush_amd import engine
    engine.lib()
    return engine


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def _bias(om, ts_b, eot_b):
    b = np.zeros(om.n_vocab, np.float32)
    b[om.tok_beg:] = ts_b
    b[om.tok_eot] = eot_b
    return b


def test_host_temperature_sampler_matches_oracle_draw_for_draw(E, oracle):
    """ohw_sample_host (std::mt19937 + std::discrete_distribution) against the oracle's ref_sample_step (own MT19937, own
    cumulative search) on the sampler goldens' rows at three temperatures, one shared generator per side."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "sampler.npz"))
    rows = g["rows_f16"].astype(np.float32)
    hists = [[int(t) for t in g["hists"][g["hist_of_row"][r]] if t >= 0] for r in range(rows.shape[0])]
    hp = synth.PRESETS["nano"]
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_F16)
    om = oracle.Model.synth(hp.as_list(), 1234)
    p, op = ctx.default_params(), om.default_params()
    rng, orng = E.HostRng(0), oracle.MT19937(0)
    import ctypes as C
    L = oracle.lib()
    L.ref_sample_step.argtypes = [C.c_void_p, C.POINTER(oracle.SampleParams), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int,
                                  C.POINTER(C.c_float), C.c_float, C.POINTER(oracle.MT19937), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.POINTER(C.c_double), C.POINTER(C.c_float)]
    picks = set()
    for T in (0.2, 0.6, 1.0):
        for r in range(rows.shape[0]):
            tok, lp, _ = ctx.sample_host(p, rows[r], hists[r], T, rng)
            lg = rows[r].copy()
            cur = np.asarray(hists[r] or [0], np.int32)
            olp, gap = C.c_float(0), C.c_double(0)
            otok = L.ref_sample_step(om.h, C.byref(op), lg.ctypes.data_as(C.POINTER(C.c_float)), cur.ctypes.data_as(C.POINTER(C.c_int32)), len(hists[r]),
                                     None, T, C.byref(orng), C.byref(olp), None, C.byref(gap), None)
            assert tok == otok or gap.value < 1e-6, (T, r, tok, otok, gap.value)
            if tok == otok:
                assert abs(lp - olp.value) < 2e-4
            picks.add(tok)
    assert len(picks) > 10          # the draws really vary
    # temperature 0 is the arg-max sampler; the no-speech probability comes with a window's first step
    tok0, lp0, nsp = ctx.sample_host(p, rows[12], [], 0.0, None)
    rt, rlp, rns = om.process_logits_ex(op, rows[12], [], None, 0.0)
    assert tok0 == rt and abs(lp0 - rlp) < 2e-4 and abs(nsp - rns) < 1e-6


def _walk_and_compare(E, oracle, om, eng, windows_pcm, bias, pol, seeks=None, ends=None, mode=0, n_max=220, mels=None):
    """Replays every pass of the engine's trace on the oracle; returns (passes compared, steps, steps where the oracle's own
    pick equals the engine's token)."""
    trace = eng.last_trace()
    qual = eng.last_quality_ex()
    op = om.default_params(); op.n_max = n_max
    by_win = {}
    for w, T, toks in trace:
        by_win.setdefault(w, []).append((T, toks))
    assert sorted(by_win) == list(range(len(windows_pcm)))
    temps = [0.0] + ([round(pol.temperature_inc * k, 6) for k in range(1, 100) if pol.temperature_inc * k < 1.0 + 1e-6] if pol.temperature_inc > 0 else [])
    n_pass = n_steps = n_same = 0
    shared_rng = oracle.MT19937(0)
    kept_all = []
    for w in sorted(by_win):
        s = oracle.State(om)
        s.set_encoder_output(om.encode(mels[w] if mels is not None else om.log_mel(windows_pcm[w], 1)))
        rng = shared_rng if mode == 1 else oracle.MT19937(0)
        seek = seeks[w] if seeks else 0
        end = ends[w] if ends else oracle.mel_frames(len(windows_pcm[w]))
        passes = by_win[w]
        first_again = False
        for k, (T, toks) in enumerate(passes):
            assert abs(T - temps[k]) < 1e-3, (w, k, T)
            r = s.decode_pass(op, bias, T, rng, toks)
            assert len(r["choice"]) >= len(toks)
            for i, t in enumerate(toks):
                n_steps += 1
                if r["choice"][i] == t:
                    n_same += 1
                elif T == 0.0:
                    assert r["margins"][i] < 2 * TOL, (w, k, i, t, r["choice"][i], float(r["margins"][i]))
                else:
                    # a draw next to an interval edge, or a near-tie of the timestamp-mass rule (timestamps' total mass
                    # against the best text token: it switches the whole distribution; margins[] carries it at T > 0)
                    assert r["gaps"][i] < 0.02 or r["margins"][i] < 2 * TOL / T, (w, k, i, t, r["choice"][i], float(r["gaps"][i]), float(r["margins"][i]))
            ev = oracle.evaluate_sequence(om, toks, r["plogs"], seek, end, n_max, False, mode)
            assert ev.n_sampled == len(toks), (w, k, ev.n_sampled, len(toks))       # the engine stopped where whisper.cpp's loop exits
            again = oracle.pass_needs_fallback(ev, pol, r["no_speech_prob"], k == len(temps) - 1)
            if k == 0:
                first_again = oracle.pass_needs_fallback(ev, pol, r["no_speech_prob"], False)
            assert again == (k + 1 < len(passes)), (w, k, T, ev.as_dict(), r["no_speech_prob"])
            n_pass += 1
        q = qual[w]
        ns = oracle.window_is_no_speech(ev, pol, r["no_speech_prob"])
        assert q["no_speech"] == ns and q["failed"] == bool(ev.failed) and q["result_len"] == ev.result_len and q["seek_delta"] == ev.seek_delta
        assert abs(q["temperature"] - passes[-1][0]) < 1e-3 and q["would_fallback"] == first_again
        keep = 0 if ns else ev.n_keep
        assert q["n_tokens"] == keep
        kept_all += passes[-1][1][:keep]
        if ev.result_len > 0:
            assert abs(q["avg_logprob"] - ev.avg_logprob) < 2 * TOL and abs(q["entropy"] - ev.entropy) < 1e-4
    assert eng.last_tokens() == kept_all
    return n_pass, n_steps, n_same


def test_temperature_ladder_matches_oracle_pass_by_pass(E, oracle, tmp_models):
    """micro model, f16, three windows in one batch.  Without a bias the procedural weights repeat one token: no timestamp,
    so the repetition guard fails every pass and the ladder runs to T = 1.0 (6 passes per window, 220 tokens each).  With
    the timestamp / end-of-text bias of test_gpu_sampler some windows are accepted early and some fall back."""
    path = tmp_models("micro")
    om = oracle.Model.load(path)
    pcm = np.concatenate([synth.synth_audio(7), synth.synth_audio(3), synth.synth_audio(11, 200000)])
    wins = [pcm[0:480000], pcm[480000:960000], pcm[960000:]]
    pol = oracle.default_policy()
    total = [0, 0, 0]
    for bias in (None, _bias(om, 6.0, 27.0)):
        eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 3)
        if bias is not None:
            E.lib().ohw_state_set_logit_bias(E.lib().ohw_engine_state(eng.h), bias.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float)), bias.size)
        res = eng.transcribe(E.AudioBuffer(pcm, 16000))
        n_pass, n_steps, n_same = _walk_and_compare(E, oracle, om, eng, wins, bias, pol)
        print(f"ladder: bias={'yes' if bias is not None else 'no'}: {n_pass} passes, {n_same} / {n_steps} steps identical; "
              f"temperatures kept {[round(q['temperature'], 1) for q in eng.last_quality_ex()]}")
        total = [a + b for a, b in zip(total, (n_pass, n_steps, n_same))]
        if bias is None:
            # every window runs the whole ladder; whether its LAST pass is marked failed is the oracle's call on the tokens that
            # pass drew (checked window by window above): a draw at T = 1.0 may be end-of-text, which ends a pass legitimately
            assert n_pass == 18 and all(abs(q["temperature"] - 1.0) < 1e-3 for q in eng.last_quality_ex())
        again = eng.transcribe(E.AudioBuffer(pcm, 16000))
        assert again.text == res.text                    # a fresh generator per window: the ladder is reproducible
        eng.close()
    assert total[2] >= 0.98 * total[1]


def test_no_speech_rule_and_ladder_off(E, oracle, tmp_models):
    path = tmp_models("micro")
    om = oracle.Model.load(path)
    pcm = np.concatenate([synth.synth_audio(7), synth.synth_audio(3)])
    wins = [pcm[:480000], pcm[480000:]]
    import ctypes as C
    eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 2)
    eng.set_decode_policy(temperature_inc=0.0)
    base = eng.transcribe(E.AudioBuffer(pcm, 16000))
    assert len(eng.last_trace()) == 2 and all(q["temperature"] == 0.0 and q["would_fallback"] for q in eng.last_quality_ex())
    assert len(base.text) > 0
    # no-speech: a bias that makes the no-speech token dominate the first step; logprob_thold 0 makes every text "unlikely"
    bias = np.zeros(om.n_vocab, np.float32)
    bias[om.tok_nosp] = 60.0
    E.lib().ohw_state_set_logit_bias(E.lib().ohw_engine_state(eng.h), bias.ctypes.data_as(C.POINTER(C.c_float)), bias.size)
    eng.set_decode_policy(temperature_inc=0.0, logprob_thold=0.0)
    res = eng.transcribe(E.AudioBuffer(pcm, 16000))
    q = eng.last_quality_ex()
    assert res.text == "" and eng.last_tokens() == [] and all(x["no_speech"] and x["no_speech_prob"] > 0.9 and x["n_tokens"] == 0 for x in q)
    pol = oracle.default_policy(); pol.temperature_inc = 0.0; pol.logprob_thold = 0.0
    _walk_and_compare(E, oracle, om, eng, wins, bias, pol)
    eng.close()


def test_short_tail_and_sub_second_input_yield_nothing(E, tmp_models):
    """whisper.cpp returns nothing for less than 1 s of audio and never decodes a last window of at most 100 frames
    (`seek + 100 >= seek_end`); the reference pads one-shot files to 1.1 s for that reason (src/input/audio.rs:767-776)."""
    path = tmp_models("micro")
    eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 2)
    eng.set_decode_policy(temperature_inc=0.0)
    r = eng.transcribe(E.AudioBuffer(synth.synth_audio(5, 12000), 16000))              # 0.75 s: valid audio, too short to decode
    assert r.text == "" and eng.last_tokens() == []
    full = eng.transcribe(E.AudioBuffer(synth.synth_audio(5), 16000))
    n_full = len(eng.last_tokens())
    pcm = np.concatenate([synth.synth_audio(5), synth.synth_audio(6, 8000)])           # 30 s + a 0.5 s tail
    r2 = eng.transcribe(E.AudioBuffer(pcm, 16000))
    q = eng.last_quality_ex()
    assert len(q) == 2 and q[1]["n_tokens"] == 0 and q[0]["n_tokens"] == n_full and r2.text == full.text
    eng.set_window_mode(E.OHW_WINDOW_SEEK)
    r3 = eng.transcribe(E.AudioBuffer(synth.synth_audio(5, 12000), 16000))
    assert r3.text == ""
    eng.close()


def test_seek_loop_with_timestamps_matches_oracle(E, oracle, tmp_models):
    """OHW_WINDOW_SEEK with a bias that makes the model emit timestamps: windows advance by 2 * (last timestamp - begin)
    frames (not 3000), the tokens after the last timestamp are dropped and decoded again by the next window; one generator
    for the whole call; the windows are cut from the spectrogram of the WHOLE recording (ohw_recording_set / ohw_mel_seek).  Every pass of every window is replayed on the oracle."""
    import ctypes as C
    path = tmp_models("micro")
    om = oracle.Model.load(path)
    bias = _bias(om, 8.0, 26.0)
    # the tail 20 dB quieter: its own maximum would clamp 2.0 lower than the recording's, which the seek mode uses
    pcm = np.concatenate([synth.synth_audio(41), 0.1 * synth.synth_audio(42, 200000)]).astype(np.float32)
    eng = E.WhisperEngine.new(path, "en", False, True, 0, E.OHW_DTYPE_F16, 1)
    eng.set_window_mode(E.OHW_WINDOW_SEEK)
    E.lib().ohw_state_set_logit_bias(E.lib().ohw_engine_state(eng.h), bias.ctypes.data_as(C.POINTER(C.c_float)), bias.size)
    pol = oracle.default_policy()
    eng.transcribe(E.AudioBuffer(pcm, 16000))
    q = eng.last_quality_ex()
    seek_end = oracle.mel_frames(len(pcm))
    seeks, wins = [], []
    seek = 0
    for x in q:
        seeks.append(seek)
        wins.append(pcm[seek * 160: seek * 160 + 480000])
        seek += x["seek_delta"] if x["seek_delta"] > 0 else 3000
    assert seek + 100 >= seek_end and len(q) >= 2                   # the loop ran to the end of the audio
    assert any(x["seek_delta"] != 3000 for x in q)                  # timestamps really drove the seek
    # the oracle's windows: frames [seek, seek + 3000) of the recording-wide spectrogram (global clamp, real neighbours)
    rec_max = om.recording_max(pcm)
    mels = [om.log_mel_seek(pcm, sk, rec_max) for sk in seeks]
    assert any(np.abs(mels[i] - om.log_mel(wins[i], 1)).max() > 0.1 for i in range(1, len(seeks)))    # it matters here
    n_pass, n_steps, n_same = _walk_and_compare(E, oracle, om, eng, wins, bias, pol, seeks=seeks, ends=[seek_end] * len(q), mode=1, mels=mels)
    print(f"seek loop: {len(q)} windows, seek deltas {[x['seek_delta'] for x in q]}, {n_pass} passes, {n_same} / {n_steps} steps identical")
    assert n_same >= 0.98 * n_steps
    eng.close()


def test_pool_behind_the_c_abi(E, tmp_models):
    """ohw_pool_* (SURVEY.md 8e behind the boundary): one engine + host thread per listed device, the model read once and its
    arena copied to the other engines, windows dealt round-robin, results gathered in recording order.  On this one-GPU box:
    n = 1, and device 0 listed twice (two engines sharing the card; the peer-copy path) - both equal the single engine."""
    path = tmp_models("micro")
    pcm = np.concatenate([synth.synth_audio(70 + w) for w in range(4)] + [synth.synth_audio(75, 90000)])      # 5 windows, short tail
    eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_BF16, 2)
    eng.set_decode_policy(temperature_inc=0.0)
    ref = eng.transcribe(E.AudioBuffer(pcm, 16000))
    ref_tokens, ref_lens = eng.last_tokens(), [q[0] for q in eng.last_quality()]
    eng.close()
    for devices in ([0], [0, 0], [0, 0, 0]):
        pool = E.EnginePool(path, "auto", False, devices, E.OHW_DTYPE_BF16, 2)
        pool.set_decode_policy(temperature_inc=0.0)
        assert pool.n_devices == len(devices) and pool.broadcast_kind == ("none" if len(devices) == 1 else "peer")
        res = pool.transcribe(E.AudioBuffer(pcm, 16000))
        assert res.text == ref.text and res.language == "en"
        assert pool.last_tokens() == ref_tokens and pool.last_window_tokens() == ref_lens
        short = pool.transcribe(E.AudioBuffer(pcm[:500000], 16000))               # fewer windows than engines
        assert len(pool.last_window_tokens()) == 2 and short.text.startswith(ref.text[:40])
        with pytest.raises(E.ValidationFailed):
            pool.transcribe(E.AudioBuffer(np.zeros(100, np.float32), 16000))
        pool.close()
    with pytest.raises(E.ModelNotFound):
        E.EnginePool(path + ".missing", "auto", False, [0])
    with pytest.raises(E.WhisperError) as bad:
        E.EnginePool(path, "auto", False, [0, 7])           # no such device on this box
    assert "device 7" in str(bad.value) and "entry 1" in str(bad.value)      # the failing device names itself


def test_pool_deals_windows_by_index_in_every_window_mode(E, tmp_models):
    """ohw_pool_transcribe == ohw_engine_transcribe token for token in all three window modes, with device 0 listed 2x and 3x:
    every engine is handed the whole recording and cuts its own windows w, w + G, ... (round 2 concatenated a device's
    windows into one buffer: in FIXED_RECORDING_MEL the clamp maximum and the samples at the 30 s marks then came from the
    wrong neighbours).  Dealing cases: n_win < G, n_win = G + 1 with a short last window, an exact multiple, one window."""
    path = tmp_models("micro")
    # loud / quiet windows so that the recording-wide clamp matters; 4 windows + a 2 s tail
    scale = (1.0, 0.05, 0.6, 0.02, 0.8)
    pcm = np.concatenate([np.float32(scale[w]) * synth.synth_audio(90 + w) for w in range(4)] + [np.float32(scale[4]) * synth.synth_audio(95, 32000)])
    recs = {"5w_short_tail": pcm, "2w": pcm[:2 * 480000], "3w_exact": pcm[:3 * 480000], "1w": pcm[:300000], "4w_plus_1s": pcm[:4 * 480000 + 16000]}
    bias = np.zeros(51865, np.float32); bias[50364:] = 6.0; bias[50257] = 27.0      # timestamps and end-of-text occur: the seek loop moves
    for mode in (E.OHW_WINDOW_FIXED, E.OHW_WINDOW_FIXED_RECORDING_MEL, E.OHW_WINDOW_SEEK):
        eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_F16, 2)
        eng.set_decode_policy(temperature_inc=0.0)
        eng.set_window_mode(mode)
        E.State.set_logit_bias(_Borrowed(E, eng.state_h), bias)
        ref = {}
        for name, r in recs.items():
            res = eng.transcribe(E.AudioBuffer(r, 16000))
            ref[name] = (res.text, eng.last_tokens(), [q[0] for q in eng.last_quality()])
        eng.close()
        for devices in ([0, 0], [0, 0, 0]):
            pool = E.EnginePool(path, "auto", False, devices, E.OHW_DTYPE_F16, 2)
            pool.set_decode_policy(temperature_inc=0.0)
            pool.set_window_mode(mode)
            for i in range(len(devices)):
                E.State.set_logit_bias(_Borrowed(E, E.lib().ohw_engine_state(pool.engine_handle(i))), bias)
            for name, r in recs.items():
                res = pool.transcribe(E.AudioBuffer(r, 16000))
                got = (res.text, pool.last_tokens(), pool.last_window_tokens())
                assert got == ref[name], (mode, devices, name)
            pool.close()
    # engines that disagree on the mode are refused
    pool = E.EnginePool(path, "auto", False, [0, 0], E.OHW_DTYPE_F16, 2)
    E.lib().ohw_engine_set_window_mode(pool.engine_handle(1), E.OHW_WINDOW_FIXED_RECORDING_MEL)
    with pytest.raises(E.WhisperError):
        pool.transcribe(E.AudioBuffer(pcm, 16000))
    pool.close()


class _Borrowed:
    """a state handle owned by an engine, for State's methods"""
    def __init__(self, E, h):
        import ctypes as C
        self.h = h if isinstance(h, C.c_void_p) else C.c_void_p(h)


def test_fixed_cuts_on_the_recording_wide_spectrogram(E, oracle, tmp_models):
    """OHW_WINDOW_FIXED_RECORDING_MEL: the batched fixed-cut path (two batches here: the LANES schedule, states sharing one
    recording) on windows cut from the spectrogram of the whole recording - a quiet first window gets the loud file's clamp.
    Every pass of every window is replayed on the oracle's ref_log_mel_seek windows."""
    path = tmp_models("micro")
    om = oracle.Model.load(path)
    pcm = np.concatenate([0.05 * synth.synth_audio(51), synth.synth_audio(52), 0.3 * synth.synth_audio(53, 250000)]).astype(np.float32)
    wins = [pcm[w * 480000:(w + 1) * 480000] for w in range(3)]
    rec_max = om.recording_max(pcm)
    mels = [om.log_mel_seek(pcm, w * 3000, rec_max) for w in range(3)]
    assert np.abs(mels[0] - om.log_mel(wins[0], 1)).max() > 0.5           # the quiet window alone clamps 2.6 lower
    pol = oracle.default_policy()
    out = {}
    for mode in (E.OHW_WINDOW_FIXED_RECORDING_MEL, E.OHW_WINDOW_FIXED):
        eng = E.WhisperEngine.new(path, "en", False, True, 0, E.OHW_DTYPE_F16, 2)
        eng.set_window_mode(mode)
        eng.transcribe(E.AudioBuffer(pcm, 16000))
        out[mode] = eng.last_trace()
        if mode == E.OHW_WINDOW_FIXED_RECORDING_MEL:
            n_pass, n_steps, n_same = _walk_and_compare(E, oracle, om, eng, wins, None, pol, mels=mels)
            print(f"fixed cuts, recording-wide mel: {n_pass} passes, {n_same} / {n_steps} steps identical")
            assert n_same >= 0.98 * n_steps
            again = eng.transcribe(E.AudioBuffer(pcm[:480000 * 2], 16000))          # another recording through the same states
            assert len(eng.last_quality_ex()) == 2 and again is not None
        eng.close()
    assert out[E.OHW_WINDOW_FIXED_RECORDING_MEL] != out[E.OHW_WINDOW_FIXED]          # the clamp changes what window 0 decodes to
