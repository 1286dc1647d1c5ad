"""openhush_amd — MI355X-native Whisper hot path for OpenHush.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI, built into libohw.so),
engine.py (the host-side mirror of the reference's WhisperEngine over ctypes), modelfile.py
(ggml model-file format) and synth.py (procedural weights / audio used by tests and bench).
"""
__all__ = ["synth", "modelfile"]
