"""The persistent small-batch decoder step (openhush_amd/csrc/decode_persist.hip: the 32 layers of a single-token step of at
most 16 rows as ONE launch whose workgroups hand activations to each other) against the launch-per-kernel path it replaces
and against the oracle.  The two paths sum in different orders (key slices, K slices), so logits agree within the 16-bit
tolerance of tests/test_gpu_parity.py, not bit for bit; within the persistent path a row's bits do not depend on its batch."""
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


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def _pair(E, ctx, rows):
    a, b = E.State(ctx, rows), E.State(ctx, rows)
    a.set_persistent(True); b.set_persistent(False)
    return a, b


@pytest.mark.parametrize("preset,dt,tol", [("micro", 1, 0.03), ("micro", 0, 0.25), ("tiny", 0, 0.25), ("micro-v3", 1, 0.03)])
def test_single_token_steps_match_the_launch_path_and_the_oracle(E, oracle, tmp_models, preset, dt, tol):
    path = tmp_models(preset)
    ctx = E.Context.from_file(path, 0, dt)
    om = oracle.Model.load(path)
    tok = ctx.tok
    for rows in (1, 3, 5, 16):
        pcm = np.stack([synth.synth_audio(40 + r) for r in range(rows)])
        sp, sl = _pair(E, ctx, rows)
        for s_ in (sp, sl):
            s_.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); s_.encode(rows)
        prompt = np.tile(np.asarray([tok.sot, tok.sot + 1, tok.transcribe], np.int32), (rows, 1))
        l0p, l0l = sp.decode(prompt, [0] * rows), sl.decode(prompt, [0] * rows)
        assert np.array_equal(l0p, l0l)                                 # the prompt pass (n_new = 3) is the launch path in both
        launches0 = sp.counter("persist_launches")
        feed = l0l.argmax(axis=1).astype(np.int32)[:, None]
        n_past = [3] * rows
        worst = 0.0
        for step in range(6):
            lp, ll = sp.decode(feed, n_past), sl.decode(feed, n_past)
            worst = max(worst, float(np.abs(lp - ll).max()))
            assert np.abs(lp - ll).max() < tol, (preset, rows, step, float(np.abs(lp - ll).max()))
            feed = ll.argmax(axis=1).astype(np.int32)[:, None]
            n_past = [x + 1 for x in n_past]
        assert sp.counter("persist_launches") == launches0 + 6 and sl.counter("persist_launches") == 0
        if rows == 3:
            # ragged positions: rewind row 1 by two tokens (its cache rows 3, 4 are simply overwritten) and step again
            n2 = [n_past[0], 3, n_past[2]]
            lp, ll = sp.decode(feed, n2), sl.decode(feed, n2)
            assert np.abs(lp - ll).max() < tol
            # and the oracle on row 0's own token path
            mel = om.log_mel(pcm[0], 1)
            s = oracle.State(om); s.set_encoder_output(om.encode(mel))
        print(f"{preset} dtype {dt} rows {rows}: worst |persistent - launches| = {worst:.4f}")
        sp.close(); sl.close()


def test_greedy_and_beam_tokens_match_the_launch_path(E, tmp_models):
    ctx = E.Context.from_file(tmp_models("micro"), 0, E.OHW_DTYPE_F16)
    pcm = np.stack([synth.synth_audio(s) for s in (3, 11, 7)])
    bias = np.zeros(ctx.hp.n_vocab, np.float32); bias[ctx.tok.timestamp_begin:] = 6.0; bias[ctx.tok.eot] = 27.0
    out = {}
    for on in (True, False):
        st = E.State(ctx, 15)
        st.set_persistent(on)
        st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(3)
        p = ctx.default_params(); p.n_max = 24
        g_plain = st.greedy_ex(3, p)
        st.set_logit_bias(bias)
        g_bias = st.greedy_ex(3, p)
        b5 = st.beam_search(3, 5, p)
        b2 = st.beam_search(3, 2, p)
        assert st.greedy_ex(3, p)[0]["tokens"] == g_bias[0]["tokens"]              # replayed graph
        assert (st.counter("persist_launches") > 0) == on
        out[on] = (g_plain, g_bias, b5, b2)
        st.close()
    for a, b in zip(out[True], out[False]):
        for x, y in zip(a, b):
            assert x["tokens"] == y["tokens"], (x["tokens"], y["tokens"])


def test_a_row_does_not_depend_on_its_batch_and_runs_are_repeatable(E, tmp_models):
    ctx = E.Context.from_file(tmp_models("micro"), 0, E.OHW_DTYPE_BF16)
    tok = ctx.tok
    pcm = np.stack([synth.synth_audio(60 + r) for r in range(5)])
    prompt = np.asarray([tok.sot, tok.sot + 1, tok.transcribe], np.int32)
    res = {}
    for rows in ((0, 1, 2, 3, 4), (2,), (4, 2)):
        st = E.State(ctx, len(rows))
        st.mel(pcm[list(rows)], None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(len(rows))
        l0 = st.decode(np.tile(prompt, (len(rows), 1)), [0] * len(rows))
        feed = np.full((len(rows), 1), 1000, np.int32)
        l1 = st.decode(feed, [3] * len(rows))
        l1b = st.decode(feed, [3] * len(rows))                          # the same step again: identical bits
        assert np.array_equal(l1, l1b)
        l2 = st.decode(feed, [4] * len(rows))
        res[rows] = {r: (l1[i], l2[i]) for i, r in enumerate(rows)}
        st.close()
    for other in ((2,), (4, 2)):
        for r in other:
            assert np.array_equal(res[other][r][0], res[(0, 1, 2, 3, 4)][r][0]) and np.array_equal(res[other][r][1], res[(0, 1, 2, 3, 4)][r][1]), (other, r)


def test_large_v3_dims_one_row_and_a_beam_of_five(E):
    hp = synth.PRESETS["large-v3"]
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
    tok = ctx.tok
    prompt = np.asarray([tok.sot, tok.sot + 1, tok.transcribe], np.int32)
    for rows in (1, 5):
        pcm = np.stack([synth.synth_audio(80 + r) for r in range(rows)])
        sp, sl = _pair(E, ctx, rows)
        for s_ in (sp, sl):
            s_.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); s_.encode(rows)
        l0 = sl.decode(np.tile(prompt, (rows, 1)), [0] * rows); sp.decode(np.tile(prompt, (rows, 1)), [0] * rows)
        feed = l0.argmax(axis=1).astype(np.int32)[:, None]
        n_past = [3] * rows
        for step in range(4):
            lp, ll = sp.decode(feed, n_past), sl.decode(feed, n_past)
            sig = float(ll.std())
            worst = float(np.abs(lp - ll).max())
            print(f"large-v3 rows {rows} step {step}: worst |persistent - launches| = {worst:.4f} (sigma {sig:.2f})")
            assert worst < 0.07 * sig                                   # the tolerance both paths meet against the oracle
            feed = ll.argmax(axis=1).astype(np.int32)[:, None]
            n_past = [x + 1 for x in n_past]
        sp.close(); sl.close()
    # one window, beam = 5: the K rows share the window's cross K/V inside the persistent step
    out = {}
    for on in (True, False):
        st = E.State(ctx, 5)
        st.set_persistent(on)
        st.mel(synth.synth_audio(21)[None, :80000], [80000], E.OHW_MEL_ZERO_TAIL, want=False); st.encode(1)
        bias = np.zeros(hp.n_vocab, np.float32); bias[tok.timestamp_begin:] = 4.0; bias[tok.eot] = 9.0
        st.set_logit_bias(bias)
        p = ctx.default_params(); p.n_max = 16
        out[on] = st.beam_search(1, 5, p)[0]
        st.close()
    assert out[True]["tokens"] == out[False]["tokens"], (out[True], out[False])
