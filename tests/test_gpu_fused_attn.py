"""OHW_DEC_FUSE_ATTN=1 (read when a state is made; off by default - measured slower): single-token decoder steps of at most 16 rows
run their masked self-attention INSIDE the QKV launch (decode.hip: the workgroup that publishes the last of a head's q / k / v columns
does the head's attention, self_attn_row<COH>).  Same arithmetic in the same order as the separate self_attn_kernel: logits must be
bit-identical between the two forms."""
import os

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


def _pair(E, ctx, rows):
    old = os.environ.get("OHW_DEC_FUSE_ATTN")
    try:
        os.environ["OHW_DEC_FUSE_ATTN"] = "1"
        a = E.State(ctx, rows)
        os.environ["OHW_DEC_FUSE_ATTN"] = "0"
        b = E.State(ctx, rows)
    finally:
        if old is None:
            os.environ.pop("OHW_DEC_FUSE_ATTN", None)
        else:
            os.environ["OHW_DEC_FUSE_ATTN"] = old
    return a, b


@pytest.mark.parametrize("preset,dt", [("micro", 1), ("micro", 0), ("tiny", 0), ("large-v3", 0)])
def test_fused_self_attention_gives_the_separate_launch_bits(E, preset, dt):
    hp = synth.PRESETS[preset]
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, dt)
    tok = ctx.tok
    prompt = np.asarray([tok.sot, tok.sot + 1, tok.transcribe], np.int32)
    for rows in ((1, 3, 16) if preset != "large-v3" else (1, 5)):
        pcm = np.stack([synth.synth_audio(90 + r) for r in range(rows)])
        sf, ss = _pair(E, ctx, rows)
        for s_ in (sf, ss):
            s_.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); s_.encode(rows)
        l0f, l0s = sf.decode(np.tile(prompt, (rows, 1)), [0] * rows), ss.decode(np.tile(prompt, (rows, 1)), [0] * rows)
        assert np.array_equal(l0f, l0s)                       # the prompt pass (3 new tokens) is the separate launch in both
        feed = l0s.argmax(axis=1).astype(np.int32)[:, None]
        n_past = [3] * rows
        for step in range(70 if preset == "micro" else 6):    # micro: past 64 keys - a second chunk of the key loop
            lf, ls = sf.decode(feed, n_past), ss.decode(feed, n_past)
            assert np.array_equal(lf, ls), (preset, rows, step, float(np.abs(lf - ls).max()))
            feed = ls.argmax(axis=1).astype(np.int32)[:, None]
            n_past = [x + 1 for x in n_past]
        if rows == 3:
            n2 = [n_past[0], 3, n_past[2]]                    # ragged positions: row 1 rewound
            assert np.array_equal(sf.decode(feed, n2), ss.decode(feed, n2))
        sf.close(); ss.close()


def test_greedy_and_beam_results_equal_with_and_without_the_fusion(E, tmp_models):
    ctx = E.Context.from_file(tmp_models("micro"), 0, E.OHW_DTYPE_F16)
    pcm = np.stack([synth.synth_audio(s) for s in (3, 11, 7)])
    bias = np.zeros(ctx.hp.n_vocab, np.float32); bias[ctx.tok.timestamp_begin:] = 6.0; bias[ctx.tok.eot] = 27.0
    sf, ss = _pair(E, ctx, 15)
    out = []
    for st in (sf, ss):
        st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(3)
        p = ctx.default_params(); p.n_max = 24
        g_plain = st.greedy_ex(3, p)
        st.set_logit_bias(bias)
        out.append((g_plain, st.greedy_ex(3, p), st.beam_search(3, 5, p), st.beam_search(3, 2, p), st.greedy_ex(3, p)))
        st.close()
    for a, b in zip(*out):
        for x, y in zip(a, b):
            assert x["tokens"] == y["tokens"] and x.get("sum_logprob") == y.get("sum_logprob"), (x, y)
