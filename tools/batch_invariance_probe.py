"""Does a window's result depend on the decode batch it rides in?  large-v3 dimensions, synthetic weights: the same windows
through a 32-window state and as the first rows of a 120-window state; logits of the prompt pass and of single-token steps
compared bit for bit, then the greedy tokens."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from openhush_amd import engine as E, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="large-v3")
    ap.add_argument("--small", type=int, default=32)
    ap.add_argument("--big", type=int, default=120)
    ap.add_argument("--cus", type=int, default=0, help="run the big state on a CU-masked stream of this many compute units (a LANES lane: 64)")
    ap.add_argument("--invariant", type=int, default=1, help="ohw_state_set_batch_invariant on both states")
    a = ap.parse_args()
    hp = synth.PRESETS[a.preset]
    ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
    pcm = np.stack([synth.synth_audio(100 + w) for w in range(a.big)])
    tok = ctx.tok
    prompt = np.asarray([tok.sot, tok.sot + 1, tok.transcribe, tok.no_timestamps], np.int32)
    sa, sb = E.State(ctx, a.small), E.State(ctx, a.big)
    sa.set_batch_invariant(bool(a.invariant)); sb.set_batch_invariant(bool(a.invariant))
    lane = None
    if a.cus > 0:
        lane = E.Stream(0, 0, a.cus)
        sb.set_stream(lane.ptr)
    for f in range(0, a.big, a.small):
        n = min(a.small, a.big - f)
        sb.mel(pcm[f:f + n], None, E.OHW_MEL_ZERO_TAIL, want=False); sb.encode_slice(n, f, a.big)
    p = ctx.default_params(); p.force_len = 40
    lb0 = sb.decode(np.tile(prompt, (a.big, 1)), [0] * a.big)
    tb = lb0.argmax(axis=1).astype(np.int32)[:, None]
    lb1 = sb.decode(tb, [4] * a.big)
    gb = sb.greedy_ex(a.big, p)
    for f in range(0, a.big, a.small):
        n = min(a.small, a.big - f)
        sa.mel(pcm[f:f + n], None, E.OHW_MEL_ZERO_TAIL, want=False); sa.encode(n)
        la0 = sa.decode(np.tile(prompt, (n, 1)), [0] * n)
        la1 = sa.decode(la0.argmax(axis=1).astype(np.int32)[:, None], [4] * n)
        ga = sa.greedy_ex(n, p)
        d0, d1 = np.abs(la0 - lb0[f:f + n]), np.abs(la1 - lb1[f:f + n])
        same = sum(x["tokens"] == y["tokens"] for x, y in zip(ga, gb[f:f + n]))
        dl = max(abs(x["sum_logprob"] - y["sum_logprob"]) for x, y in zip(ga, gb[f:f + n]))
        print(f"windows {f}..{f + n - 1}: prompt-pass rows differing {int((d0.max(axis=1) > 0).sum())} (max {d0.max():.2e}), step rows differing "
              f"{int((d1.max(axis=1) > 0).sum())} (max {d1.max():.2e}), greedy identical {same} / {n}, max |sum_logprob diff| {dl:.2e}")


if __name__ == "__main__":
    main()
