"""whisper.cpp's per-window decode policy as the oracle restates it (recalled from upstream,
parity unpinned: no whisper.cpp source offline) - the pieces that need no GPU: the Mersenne twister against the C++
standard's known answers, the loop exits / result_len / seek_delta bookkeeping on crafted sequences, the acceptance
test, and the whole procedure on a tiny synthetic model."""
import numpy as np
import pytest

from openhush_amd import synth
from oracle import oracle


def test_mt19937_known_answers():
    """ISO C++ [rand.predef]: the 10000th consecutive invocation of a default-constructed std::mt19937 (seed 5489)
    produces 4123659995; its first output is 3499211612."""
    r = oracle.MT19937(5489)
    x = [r.next() for _ in range(10000)]
    assert x[0] == 3499211612 and x[-1] == 4123659995
    a, b = oracle.MT19937(0), oracle.MT19937(0)
    assert [a.next() for _ in range(5)] == [b.next() for _ in range(5)]      # whisper.cpp's seed


@pytest.fixture(scope="module")
def model():
    return oracle.Model.synth(synth.PRESETS["nano"].as_list(), 1234)


def test_mel_frames():
    assert oracle.mel_frames(480000) == 2999          # 1 + (480000 + 200 - 400) / 160
    assert oracle.mel_frames(16000) == 99             # exactly 1 s is "too short" for whisper.cpp (needs > 100 frames)
    assert oracle.mel_frames(17600) == 109            # the reference pads to 1.1 s for this reason (src/input/audio.rs:767-776)


def test_loop_exits_and_result_len(model):
    m = model
    tb, eot = m.tok_beg, m.tok_eot
    lp = lambda n: [-0.5] * n       # noqa: E731
    # end-of-text with no timestamp while audio is left: failed
    ev = oracle.evaluate_sequence(m, [10, 11, eot], lp(3), 0, 9000, 220)
    assert ev.failed and not ev.completed and ev.result_len == 0
    # the same in the last window of the audio (seek + 3000 + 100 >= seek_end): everything counts
    ev = oracle.evaluate_sequence(m, [10, 11, eot], lp(3), 0, 2999, 220)
    assert ev.completed and not ev.failed and ev.result_len == 3 and ev.n_keep == 2 and ev.seek_delta == 3000
    assert abs(ev.avg_logprob + 0.5) < 1e-6
    # timestamps: result_len runs to the last timestamp token, seek_delta = 2 * (ts - beg)
    seq = [tb, 10, 11, tb + 500, tb + 500, 12, 13, tb + 900, 14, eot]
    ev = oracle.evaluate_sequence(m, seq, lp(len(seq)), 0, 9000, 220, window_mode=1)
    assert ev.completed and ev.result_len == 8 and ev.seek_delta == 1800 and ev.n_keep == 8
    ev0 = oracle.evaluate_sequence(m, seq, lp(len(seq)), 0, 9000, 220, window_mode=0)
    assert ev0.n_keep == 9                              # fixed cuts keep the text after the last timestamp
    # a timestamp that leaves less than 1 s of audio ends the window there
    seq = [tb, 10, tb + 1460, 11, 12, eot]
    ev = oracle.evaluate_sequence(m, seq, lp(len(seq)), 0, 2999, 220)
    assert ev.completed and ev.n_sampled == 3 and ev.result_len == 3 and ev.seek_delta == 2920
    # timestamps going back in time: failed (cannot happen through the filter, the rule is restated anyway)
    seq = [tb + 100, 5, tb + 300, tb + 300, 6, tb + 200, 7]
    ev = oracle.evaluate_sequence(m, seq, lp(len(seq)), 0, 9000, 220)
    assert ev.failed and ev.n_sampled == 6
    # the repetition guard: n_max tokens without a timestamp past the middle of the window
    seq = [tb + 10] + [42] * 219
    ev = oracle.evaluate_sequence(m, seq, lp(220), 0, 9000, 220)
    assert ev.failed and ev.result_len == 1 and ev.n_keep == 220
    seq = [tb + 10, 1, tb + 800] + [42] * 217
    ev = oracle.evaluate_sequence(m, seq, lp(220), 0, 9000, 220)
    assert not ev.failed and not ev.completed and ev.result_len == 3 and ev.seek_delta == 1600
    # entropy over the last 32 of the first result_len tokens
    seq = list(range(100, 140)) + [tb + 700, eot]
    ev = oracle.evaluate_sequence(m, seq, lp(len(seq)), 0, 9000, 220)
    assert ev.result_len == 41 and abs(ev.entropy - np.log(32)) < 1e-5


def test_acceptance_and_no_speech_rules(model):
    m = model
    q = oracle.default_policy()
    assert (q.temperature_inc, q.entropy_thold, q.logprob_thold, q.no_speech_thold) == pytest.approx((0.2, 2.4, -1.0, 0.6))
    tb = m.tok_beg
    good = list(range(100, 140)) + [tb + 700, m.tok_eot]
    ev = oracle.evaluate_sequence(m, good, [-0.3] * len(good), 0, 9000, 220)
    assert not oracle.pass_needs_fallback(ev, q, 0.01, False)
    ev_lp = oracle.evaluate_sequence(m, good, [-1.5] * len(good), 0, 9000, 220)
    assert oracle.pass_needs_fallback(ev_lp, q, 0.01, False)            # unlikely text, speech present: retry
    assert not oracle.pass_needs_fallback(ev_lp, q, 0.9, False)         # ... but not when the window is no-speech
    assert oracle.window_is_no_speech(ev_lp, q, 0.9) and not oracle.window_is_no_speech(ev, q, 0.9)
    rep = [tb] + [42] * 40 + [tb + 700, m.tok_eot]
    ev_rep = oracle.evaluate_sequence(m, rep, [-0.1] * len(rep), 0, 9000, 220)
    assert ev_rep.entropy < 2.4 and oracle.pass_needs_fallback(ev_rep, q, 0.01, False)
    assert not oracle.pass_needs_fallback(ev_rep, q, 0.01, True)        # the last temperature is accepted whatever it gives


def test_whole_window_procedure_on_a_tiny_model(model):
    m = model
    enc = m.encode(m.log_mel(synth.synth_audio(3), 1))
    s = oracle.State(m)
    s.set_encoder_output(enc)
    p = m.default_params(); p.n_max = 24
    q = oracle.default_policy()
    kept, res, sampled = s.decode_window(p, q, None, 0, 2999, 0, None)
    # procedural weights repeat one token: no timestamp -> the repetition guard fails every pass up to the last temperature
    assert res.n_passes == 6 and abs(res.temperature - 1.0) < 1e-6 and res.ev.failed and kept == sampled[:len(kept)]
    again, res2, _ = s.decode_window(p, q, None, 0, 2999, 0, None)
    assert again == kept and res2.n_passes == res.n_passes            # a fresh generator (seed 0) per call: reproducible
    q0 = oracle.default_policy(); q0.temperature_inc = 0.0
    kept0, res0, sampled0 = s.decode_window(p, q0, None, 0, 2999, 0, None)
    g = s.greedy_ex(p)
    assert res0.n_passes == 1 and res0.temperature == 0.0 and sampled0 == g["tokens"] and kept0 == g["tokens"]
    # a pass at T > 0 draws one canonical double (two 32-bit words) per token from the shared generator
    rng = oracle.MT19937(0)
    r1 = s.decode_pass(p, None, 0.6, rng)
    probe = oracle.MT19937(0)
    for _ in range(2 * len(r1["tokens"])):
        probe.next()
    assert rng.next() == probe.next()
    # the no-speech rule: a bias that makes the no-speech token likely on the first step and the text unlikely
    bias = np.zeros(m.n_vocab, np.float32)
    bias[m.tok_nosp] = 60.0
    q1 = oracle.default_policy(); q1.temperature_inc = 0.0; q1.logprob_thold = 0.0
    keptn, resn, _ = s.decode_window(p, q1, bias, 0, 2999, 0, None)
    assert resn.no_speech_prob > 0.9 and resn.no_speech == 1 and keptn == []
