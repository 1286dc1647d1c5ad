"""Procedural model weights and synthetic audio (numpy side).

No Whisper checkpoint can be fetched offline, so every model used by the tests and by bench.py
is generated from a counter-based hash.  The same generator exists three times and is checked
bit-for-bit between them in tests/: here (numpy; writes ggml model files), in oracle/whisper_ref.c
(CPU oracle) and in openhush_amd/csrc/weights.hip (device-side fill for bench-sized models).

Model dimensions: SURVEY.md section 8 (public Whisper model cards); the reference only names
the five sizes (reference src/engine/whisper.rs:45-51) and their files (:71-79).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Iterator, List, Tuple

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
CHUNK_SECONDS = 30
CHUNK_SAMPLES = CHUNK_SECONDS * SAMPLE_RATE  # 480000
CHUNK_FRAMES = CHUNK_SAMPLES // HOP          # 3000
N_FREQ = N_FFT // 2 + 1                      # 201


@dataclasses.dataclass(frozen=True)
class HParams:
    n_vocab: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int
    n_mels: int
    ftype: int = 1  # 1 = 2-D+ tensors stored f16 (the stock ggml files), 0 = all f32

    def as_list(self) -> List[int]:
        return [self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head,
                self.n_audio_layer, self.n_text_ctx, self.n_text_state, self.n_text_head,
                self.n_text_layer, self.n_mels, self.ftype]

    @property
    def is_multilingual(self) -> bool:
        return self.n_vocab >= 51865

    @property
    def n_langs(self) -> int:
        # whisper.cpp: num_languages = n_vocab - 51765 - (multilingual ? 1 : 0)
        return self.n_vocab - 51765 - (1 if self.is_multilingual else 0)


def _hp(n_vocab, d, heads, layers, n_mels):
    return HParams(n_vocab, 1500, d, heads, layers, 448, d, heads, layers, n_mels)


PRESETS: Dict[str, HParams] = {
    # name -> dims; "micro"/"nano" are test-only shapes (not Whisper sizes) that keep d_head = 64.
    "nano": _hp(51865, 128, 2, 2, 80),
    "micro": _hp(51865, 256, 4, 2, 80),
    "micro-v3": _hp(51866, 256, 4, 2, 128),
    "tiny": _hp(51865, 384, 6, 4, 80),
    "base": _hp(51865, 512, 8, 6, 80),
    "small": _hp(51865, 768, 12, 12, 80),
    "medium": _hp(51865, 1024, 16, 24, 80),
    "large-v3": _hp(51866, 1280, 20, 32, 128),
}

# ---------------------------------------------------------------------------------------------
# counter-based generator
# ---------------------------------------------------------------------------------------------
_M32 = np.uint32(0xFFFFFFFF)


def fnv1a32(name: str) -> int:
    h = 0x811C9DC5
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


def _fmix32_scalar(h: int) -> int:
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32, copy=True)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def tensor_key(seed: int, name: str) -> int:
    return _fmix32_scalar(fnv1a32(name) ^ ((seed * 0x9E3779B9) & 0xFFFFFFFF))


def uniform_pm1(key: int, start: int, count: int) -> np.ndarray:
    """u[i] in [-1, 1), exact multiples of 2^-23, for flat indices start .. start+count-1."""
    idx = np.arange(start, start + count, dtype=np.uint64).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = _fmix32((idx * np.uint32(0x9E3779B1)) ^ np.uint32(key))
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)


# kinds of tensors: value = offset + u * scale   (all in float32, one rounding per op)
KIND_LINEAR = 0   # scale = sqrt(3 / fan_in)      (variance 1 / fan_in)
KIND_BIAS = 1     # scale = 0.1
KIND_GAMMA = 2    # 1 + 0.1 u
KIND_EMBED = 3    # scale = sqrt(3 / d) * emb_gain  (token embedding; tied logits)
KIND_POS = 4      # scale = 0.1   (decoder learned positions); encoder positions are sinusoids


@dataclasses.dataclass(frozen=True)
class TensorSpec:
    name: str
    shape: Tuple[int, ...]   # row-major (numpy order); ggml writes dims reversed
    kind: int
    fan_in: int
    f16: bool                # stored as f16 when hparams.ftype == 1


def tensor_specs(hp: HParams) -> List[TensorSpec]:
    d, dt = hp.n_audio_state, hp.n_text_state
    out: List[TensorSpec] = []

    def lin(name, n_out, n_in, bias=True):
        out.append(TensorSpec(name + ".weight", (n_out, n_in), KIND_LINEAR, n_in, True))
        if bias:
            out.append(TensorSpec(name + ".bias", (n_out,), KIND_BIAS, 1, False))

    def ln(name, n):
        out.append(TensorSpec(name + ".weight", (n,), KIND_GAMMA, 1, False))
        out.append(TensorSpec(name + ".bias", (n,), KIND_BIAS, 1, False))

    out.append(TensorSpec("encoder.positional_embedding", (hp.n_audio_ctx, d), KIND_POS, 1, False))
    out.append(TensorSpec("encoder.conv1.weight", (d, hp.n_mels, 3), KIND_LINEAR, 3 * hp.n_mels, True))
    out.append(TensorSpec("encoder.conv1.bias", (d,), KIND_BIAS, 1, False))
    out.append(TensorSpec("encoder.conv2.weight", (d, d, 3), KIND_LINEAR, 3 * d, True))
    out.append(TensorSpec("encoder.conv2.bias", (d,), KIND_BIAS, 1, False))
    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}."
        ln(p + "attn_ln", d)
        lin(p + "attn.query", d, d)
        lin(p + "attn.key", d, d, bias=False)
        lin(p + "attn.value", d, d)
        lin(p + "attn.out", d, d)
        ln(p + "mlp_ln", d)
        lin(p + "mlp.0", 4 * d, d)
        lin(p + "mlp.2", d, 4 * d)
    ln("encoder.ln_post", d)
    out.append(TensorSpec("decoder.positional_embedding", (hp.n_text_ctx, dt), KIND_POS, 1, False))
    out.append(TensorSpec("decoder.token_embedding.weight", (hp.n_vocab, dt), KIND_EMBED, dt, True))
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}."
        ln(p + "attn_ln", dt)
        lin(p + "attn.query", dt, dt)
        lin(p + "attn.key", dt, dt, bias=False)
        lin(p + "attn.value", dt, dt)
        lin(p + "attn.out", dt, dt)
        ln(p + "cross_attn_ln", dt)
        lin(p + "cross_attn.query", dt, dt)
        lin(p + "cross_attn.key", dt, d, bias=False)
        lin(p + "cross_attn.value", dt, d)
        lin(p + "cross_attn.out", dt, dt)
        ln(p + "mlp_ln", dt)
        lin(p + "mlp.0", 4 * dt, dt)
        lin(p + "mlp.2", dt, 4 * dt)
    ln("decoder.ln", dt)
    return out


EMB_GAIN = np.float32(4.0)  # spreads the tied logits (sigma ~ 4) so greedy margins are not all tiny


def kind_scale_offset(kind: int, fan_in: int) -> Tuple[np.float32, np.float32]:
    if kind == KIND_LINEAR:
        return np.float32(math.sqrt(3.0 / fan_in)), np.float32(0.0)
    if kind == KIND_BIAS:
        return np.float32(0.1), np.float32(0.0)
    if kind == KIND_GAMMA:
        return np.float32(0.1), np.float32(1.0)
    if kind == KIND_EMBED:
        return np.float32(np.float32(math.sqrt(3.0 / fan_in)) * EMB_GAIN), np.float32(0.0)
    if kind == KIND_POS:
        return np.float32(0.1), np.float32(0.0)
    raise ValueError(kind)


def sinusoids(length: int, channels: int) -> np.ndarray:
    """Encoder positional embedding of the published model (float32 values as the files store)."""
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float64))
    t = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def gen_tensor(seed: int, spec: TensorSpec, hp: HParams) -> np.ndarray:
    """float32 array holding the stored value (already rounded through f16 when stored f16)."""
    n = int(np.prod(spec.shape))
    if spec.name == "encoder.positional_embedding":
        return sinusoids(*spec.shape)
    scale, offset = kind_scale_offset(spec.kind, spec.fan_in)
    u = uniform_pm1(tensor_key(seed, spec.name), 0, n)
    v = (u * scale).astype(np.float32)
    if offset != 0:
        v = (v + offset).astype(np.float32)
    if spec.f16 and hp.ftype == 1:
        v = v.astype(np.float16).astype(np.float32)
    return v.reshape(spec.shape)


def iter_tensors(seed: int, hp: HParams) -> Iterator[Tuple[TensorSpec, np.ndarray]]:
    for spec in tensor_specs(hp):
        yield spec, gen_tensor(seed, spec, hp)


# ---------------------------------------------------------------------------------------------
# slaney mel filterbank (what the ggml model files carry; OpenAI's mel_filters.npz = librosa)
# ---------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mel)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), 200.0 * m / 3.0)


def mel_filterbank(n_mels: int) -> np.ndarray:
    """[n_mels, 201] float32, slaney scale + slaney norm, 0..8000 Hz."""
    fft_freqs = np.linspace(0.0, SAMPLE_RATE / 2.0, N_FREQ)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(8000.0), n_mels + 2)
    hz_pts = _mel_to_hz(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# synthetic audio: harmonic "speech-like" mix + noise, peak 0.5
# ---------------------------------------------------------------------------------------------
def synth_audio(chunk_id: int, n_samples: int = CHUNK_SAMPLES, seed: int = 0x0A5A0000) -> np.ndarray:
    key = _fmix32_scalar((seed + chunk_id) & 0xFFFFFFFF)
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    pu = uniform_pm1(key, 0, 16).astype(np.float64)
    f0 = 200.0 + 100.0 * pu[0]
    drift = 1.0 + 0.1 * np.sin(2 * np.pi * (0.3 + 0.2 * pu[1]) * t)
    phase = 2 * np.pi * np.cumsum(f0 * drift) / SAMPLE_RATE
    sig = np.zeros(n_samples, dtype=np.float64)
    for h in range(1, 6):
        sig += (1.0 / h) * np.sin(h * phase + np.pi * pu[1 + h])
    env = 0.5 * (1.0 + np.sin(2 * np.pi * 4.0 * t + np.pi * pu[8]))
    sig *= env
    noise = uniform_pm1(_fmix32_scalar(key ^ 0x5BD1E995), 0, n_samples).astype(np.float64)
    sig += 10 ** (-30 / 20) * noise * max(1e-9, np.abs(sig).max())
    sig *= 0.5 / max(1e-9, np.abs(sig).max())
    return sig.astype(np.float32)
