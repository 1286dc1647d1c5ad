"""GPU diagnostic: run-to-run determinism and file-vs-synthetic equality, stage by stage."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openhush_amd import engine as E, synth, modelfile

pcm = np.stack([synth.synth_audio(7)])
ns = [synth.CHUNK_SAMPLES]

def run(ctx):
    st = E.State(ctx, 1)
    st.mel(pcm, ns, E.OHW_MEL_REFLECT, want=False)
    st.encode(1)
    out = {k: st.fetch(k, 1) for k in ("mel", "conv1", "stem", "block0", "enc", "xk0")}
    out["logits"] = st.decode(np.asarray([[ctx.tok.sot, ctx.tok.sot + 1, ctx.tok.transcribe]], np.int32), [0])
    return out

for preset in ("nano", "micro", "micro-v3"):
    hp = synth.PRESETS[preset]
    path = os.path.join(tempfile.gettempdir(), f"ggml-{preset}.bin")
    modelfile.write_synthetic_model(path, hp, 1234)
    cf = E.Context.from_file(path, 0, 0)
    cs = E.Context.synthetic(hp.as_list(), 1234, 0, 0)
    a, b, c = run(cf), run(cf), run(cs)
    for k in a:
        print(preset, k, "rerun-equal", np.array_equal(a[k], b[k]), "maxdiff", float(np.abs(a[k] - b[k]).max()),
              "| file-vs-synth equal", np.array_equal(a[k], c[k]), "maxdiff", float(np.abs(a[k] - c[k]).max()))
    df, ds = cf.weight_digests(), cs.weight_digests()
    print(preset, "weight buffers", len(df), "differing:", [k for k in df if df[k] != ds[k]])

def fnv64(b):
    h = 0xcbf29ce484222325
    for c in b:
        h = ((h ^ c) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h

hp = synth.PRESETS["nano"]
spec = [s for s in synth.tensor_specs(hp) if s.name == "encoder.blocks.0.attn_ln.weight"][0]
u = synth.uniform_pm1(synth.tensor_key(1234, spec.name), 0, hp.n_audio_state)
two_step = ((u * np.float32(0.1)).astype(np.float32) + np.float32(1.0)).astype(np.float32)
fma = (u.astype(np.float64) * np.float64(np.float32(0.1)) + 1.0).astype(np.float32)
path = os.path.join(tempfile.gettempdir(), "ggml-nano.bin")
cf = E.Context.from_file(path, 0, 0)
cs = E.Context.synthetic(hp.as_list(), 1234, 0, 0)
print("gamma digests: file", hex(cf.weight_digests()["enc0.ln1.g"]), "synth", hex(cs.weight_digests()["enc0.ln1.g"]),
      "numpy two-step", hex(fnv64(two_step.tobytes())), "numpy fma", hex(fnv64(fma.tobytes())), "n diff two-step vs fma", int((two_step != fma).sum()))
