"""Writes a synthetic 16 kHz WAV (3.2 windows) and a synthetic 'micro' ggml model file for trying the CLI:
    python tools/make_cli_demo.py OUTDIR"""
import os, sys, wave
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openhush_amd import modelfile, synth
out = sys.argv[1] if len(sys.argv) > 1 else "."
os.makedirs(out, exist_ok=True)
pcm = np.concatenate([synth.synth_audio(70 + w) for w in range(3)] + [synth.synth_audio(73, 100000)])
with wave.open(os.path.join(out, "long.wav"), "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
    w.writeframes(np.round(pcm * 32767).astype("<i2").tobytes())
modelfile.write_synthetic_model(os.path.join(out, "ggml-micro.bin"), synth.PRESETS["micro"], 1234)
