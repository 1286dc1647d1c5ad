"""BASELINE.json config #4 on ONE GPU through the product API: one long synthetic recording (default 1 h = 120 windows),
`ohw_engine_transcribe` with max_batch 32 (host PCM in, text out; two batches in flight), large-v3 dimensions.
    gpurun -- python tools/long_audio_probe.py [--minutes 60] [--model large-v3]
Writes a synthetic ggml model file to /tmp first (procedural weights: natural end-of-text is arbitrary, so the token counts
- and with them the time - are not those of real speech; the run shows the path at scale and the effect of the pipeline)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openhush_amd import engine as E, modelfile, synth

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=60.0)
ap.add_argument("--model", default="large-v3")
ap.add_argument("--max-batch", type=int, default=32)
a = ap.parse_args()
path = f"/tmp/ggml-{a.model}-synth.bin"
t0 = time.perf_counter()
if not os.path.exists(path):
    modelfile.write_synthetic_model(path, synth.PRESETS[a.model], 1234)
print(f"model file {os.path.getsize(path) / 1e9:.2f} GB written in {time.perf_counter() - t0:.1f} s", flush=True)
n_win = int(np.ceil(a.minutes * 60 / 30))
pcm = np.concatenate([synth.synth_audio(1000 + w) for w in range(n_win)])[: int(a.minutes * 60 * 16000)]
for cus in ("96", "0"):
    os.environ["OHW_ENGINE_ENC_CUS"] = cus
    t0 = time.perf_counter()
    eng = E.WhisperEngine.new(path, "auto", False, True, 0, E.OHW_DTYPE_BF16, a.max_batch)
    t_load = time.perf_counter() - t0
    eng.transcribe(E.AudioBuffer(pcm[: 480000 * 2], 16000))           # warm-up (allocations, graph capture)
    t0 = time.perf_counter()
    res = eng.transcribe(E.AudioBuffer(pcm, 16000))
    dt = time.perf_counter() - t0
    q = eng.last_quality()
    ntok = sum(x[0] for x in q)
    print(f"enc_cus={cus:>2}: load {t_load:.1f} s; {len(pcm) / 16000:.0f} s of audio ({len(q)} windows, {ntok} tokens, "
          f"max {max(x[0] for x in q)} per window) in {dt:.2f} s = {len(pcm) / 16000 / dt:.0f} audio-s/s", flush=True)
    eng.close()
