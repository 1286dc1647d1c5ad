import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from openhush_amd import engine as E, synth
from tests.conftest import *  # noqa
import tempfile
from openhush_amd import modelfile
hp = synth.PRESETS["micro"]
ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_F16)
tok = ctx.tok
for W in (1, 3):
    for K in (2, 3, 5):
        for biased in (False, True):
            st = E.State(ctx, W * K)
            pcm = np.stack([synth.synth_audio(s) for s in (3, 11, 7)[:W]])
            if biased:
                bias = np.zeros(hp.n_vocab, np.float32); bias[tok.timestamp_begin:] = 6.0; bias[tok.eot] = 27.0
                st.set_logit_bias(bias)
            st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(W)
            p = ctx.default_params(); p.n_max = 24
            try:
                r = st.beam_search(W, K, p)
                print("W", W, "K", K, "biased", biased, "ok", [len(x["tokens"]) for x in r], flush=True)
            except Exception as e:
                print("W", W, "K", K, "biased", biased, "FAIL", str(e)[:120], flush=True)
            st.close()
