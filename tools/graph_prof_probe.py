import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from openhush_amd import engine as E, synth
model, tokens, calls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
hp = synth.PRESETS[model]
ctx = E.Context.synthetic(hp.as_list(), 1234, 0, E.OHW_DTYPE_BF16)
st = E.State(ctx, 1)
pcm = synth.synth_audio(5)[None]
st.mel(pcm, None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(1)
p = ctx.default_params(); p.force_len = tokens
for i in range(calls):
    st.greedy(1, p)
    print("greedy call", i, "done; step graphs", st.counter("step_graphs"), flush=True)
