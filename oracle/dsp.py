"""CPU restatement of the reference's audio preprocessing (TEST INFRASTRUCTURE: only tests/ may import this).

Follows reference src/input/audio.rs: rms_db :86-102, apply_gain :123-129, normalize_rms :108-120, compress :139-191,
limit :197-239, resample_linear :972-990 - float32 scalars, the reference's operation order (pure-Python loops: small
inputs only).  Pinned by the reference's own unit tests for these functions (src/input/audio.rs:1165-1329), which
tests/test_dsp.py restates; a bit-level pin against the Rust build is not possible here (no rustc): libm's powf / expf /
log10f may differ from Rust's in the last place.
"""
import math

import numpy as np

F = np.float32


def rms_db(x):
    if len(x) == 0:
        return -math.inf
    s = F(0)
    for v in np.asarray(x, F):
        s = F(s + F(v * v))
    rms = F(math.sqrt(F(s / F(len(x)))))
    return float(F(20) * F(math.log10(rms))) if rms > 0 else -math.inf


def apply_gain(x, gain_db):
    g = F(math.pow(10.0, float(F(gain_db) / F(20))))
    return (np.asarray(x, F) * g).astype(F)


def normalize_rms(x, target_db):
    cur = rms_db(x)
    return apply_gain(x, F(target_db) - F(cur)) if math.isfinite(cur) else np.asarray(x, F).copy()


def compress(x, rate, threshold_db, ratio, attack_ms, release_ms, makeup_gain_db):
    x = np.asarray(x, F).copy()
    if len(x) == 0 or ratio <= 1.0:
        return x
    thr = F(math.pow(10.0, float(F(threshold_db) / F(20))))
    att = F(math.exp(float(F(-1) / (F(attack_ms) * F(rate) / F(1000)))))
    rel = F(math.exp(float(F(-1) / (F(release_ms) * F(rate) / F(1000)))))
    env = F(0)
    for i in range(len(x)):
        a = F(abs(x[i]))
        if a > env:
            env = F(F(att * env) + F(F(F(1) - att) * a))
        else:
            env = F(F(rel * env) + F(F(F(1) - rel) * a))
        gain = F(1)
        if env > thr:
            over = F(F(20) * F(math.log10(float(F(env / thr)))))
            red = F(over - F(over / F(ratio)))
            gain = F(math.pow(10.0, float(F(-red) / F(20))))
        x[i] = F(x[i] * gain)
    return apply_gain(x, makeup_gain_db) if makeup_gain_db != 0.0 else x


def limit(x, rate, ceiling_db, release_ms):
    x = np.asarray(x, F).copy()
    ceil_ = F(math.pow(10.0, float(F(ceiling_db) / F(20))))
    rel = F(math.exp(float(F(-1) / (F(release_ms) * F(rate) / F(1000)))))
    gr = F(1)
    for i in range(len(x)):
        a = F(abs(x[i]))
        target = F(ceil_ / a) if a > ceil_ else F(1)
        gr = target if target < gr else F(F(rel * gr) + F(F(F(1) - rel) * target))
        x[i] = F(x[i] * gr)
    return x


def resample_linear(x, from_rate, to_rate):
    x = np.asarray(x, F)
    if from_rate == to_rate:
        return x.copy()
    ratio = to_rate / from_rate
    n = int(len(x) * ratio)
    out = np.empty(n, F)
    for i in range(n):
        src = i / ratio
        lo = int(math.floor(src))
        hi = min(lo + 1, len(x) - 1)
        frac = src - lo
        out[i] = F(F(x[lo] * F(F(1) - F(frac))) + F(x[hi] * F(frac)))
    return out


# ---- resample(.., ResamplingQuality::High): rubato SincFixedIn as the reference configures it ---------------------------
# reference src/input/audio.rs:1007-1095: sinc_len 256, f_cutoff 0.95, oversampling_factor 256, SincInterpolationType::Linear,
# WindowFunction::BlackmanHarris2, chunks of 1024 input samples (the last one zero-padded, ceil(len * ratio) samples of its
# output kept).  The crate (rubato 0.16.2, Cargo.lock:5571-5574) is NOT under /root/reference: PARITY UNPINNED - this is its published
# algorithm restated in fp64 and vectorised numpy, written independently of openhush_amd/csrc/dsp.cpp and resample.hip so that
# the product's host and device resamplers have a checker that is not themselves.
#
# Form used here (one closed expression per output instead of the crate's chunk state machine):
#   table   h[x] = bh2(x / T) * sinc((x - T/2) * fc / F) / (sum(h) / F),  T = L * F, L = 256 taps, F = 256 sub-filters,
#           bh2 = the squared 4-term Blackman-Harris window, fc = 0.95 (times the ratio when it is below 1)
#   output  k = 0, 1, ...: read position t_k = -L/2 + (k + 1) / ratio in input samples (the crate starts at -L/2 and steps
#           BEFORE every output); with i = floor(t_k), phase = (t_k - i) * F, s = floor(phase), a = phase - s:
#           y_k = (1 - a) * D(i, s) + a * D(i', s'),  (i', s') = (i, s + 1) or (i + 1, 0) when s + 1 == F,
#           D(i, s) = sum_n x[i + n] * h[F * n + (F - 1 - s)]   (x = the input, zero before its start and after its end)
#   count   the crate emits outputs while t < chunk - (L + 1) - ceil(1 / ratio) within each chunk: restated as a plain loop
SINC_L, SINC_F, SINC_CHUNK = 256, 256, 1024


def sinc_table64(ratio):
    """h [T] float64, T = L * F"""
    T = SINC_L * SINC_F
    fc = 0.95 if ratio >= 1.0 else 0.95 * ratio
    x = np.arange(T, dtype=np.float64)
    u = x / T
    bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * u) + 0.14128 * np.cos(4 * np.pi * u) - 0.01168 * np.cos(6 * np.pi * u)
    h = bh * bh * np.sinc((x - T // 2) * fc / SINC_F)          # np.sinc(z) = sin(pi z) / (pi z)
    return h / (h.sum() / SINC_F)


def sinc_out_len(n, ratio):
    """samples the reference's chunk loop appends for n input samples"""
    step = 1.0 / ratio
    end = float(SINC_CHUNK - (SINC_L + 1) - math.ceil(step))
    t, total = -float(SINC_L // 2), 0
    for pos in range(0, n, SINC_CHUNK):
        ln = min(SINC_CHUNK, n - pos)
        made = 0
        while t < end:
            t += step
            made += 1
        t -= SINC_CHUNK
        total += min(made, math.ceil(ln * ratio)) if ln < SINC_CHUNK else made
    return total


def resample_sinc(x, from_rate, to_rate):
    """float64 result of the reference's high-quality resampler for a float32 input (rounded to float32 by the caller if wanted)"""
    x = np.asarray(x, np.float64)
    if len(x) == 0:
        return np.zeros(0)
    if from_rate == to_rate:
        return x.copy()
    ratio = to_rate / from_rate
    n_out = sinc_out_len(len(x), ratio)
    h = sinc_table64(ratio).reshape(SINC_L, SINC_F)           # h[n, p] = table[F * n + p]
    k = np.arange(n_out, dtype=np.float64)
    t = -float(SINC_L // 2) + (k + 1.0) / ratio
    i0 = np.floor(t)
    phase = (t - i0) * SINC_F
    s0 = np.minimum(np.floor(phase), SINC_F - 1)
    a = phase - s0
    i0 = i0.astype(np.int64); s0 = s0.astype(np.int64)
    s1 = s0 + 1
    i1 = i0 + (s1 == SINC_F)
    s1 = np.where(s1 == SINC_F, 0, s1)
    # input with L zeros in front (positions down to -L/2 - ... are read) and enough behind (the zero-padded last chunk)
    pad_front = SINC_L
    xp = np.concatenate([np.zeros(pad_front), x, np.zeros(SINC_CHUNK + 2 * SINC_L)])
    out = np.empty(n_out)
    taps = np.arange(SINC_L)
    B = 4096
    for lo in range(0, n_out, B):
        hi = min(n_out, lo + B)
        w0 = xp[(i0[lo:hi, None] + pad_front) + taps[None, :]]
        w1 = xp[(i1[lo:hi, None] + pad_front) + taps[None, :]]
        d0 = np.einsum("kn,nk->k", w0, h[:, SINC_F - 1 - s0[lo:hi]])
        d1 = np.einsum("kn,nk->k", w1, h[:, SINC_F - 1 - s1[lo:hi]])
        out[lo:hi] = d0 + a[lo:hi] * (d1 - d0)
    return out


# ---- AudioBuffer::denoise (reference src/input/audio.rs:249-341) without the network -------------------------------------
# The reference resamples 16 kHz -> 48 kHz (linear, resample_for_rnnoise :996-1001), feeds nnnoiseless' DenoiseState 480-sample
# frames scaled to the 16-bit range, fades the first frame in, keeps only the real part of a short last frame, resamples back,
# truncates / zero-extends to the original length and mixes with `strength`.  The trained network is in a crate (nnnoiseless 0.5.2,
# Cargo.lock:3972-3975) that is not under /root/reference; `process_frame(frame_f32[480]) -> out_f32[480]` stands for it (the product takes the same hook).
def denoise(x, rate, strength, process_frame):
    x = np.asarray(x, F)
    if len(x) == 0 or strength <= 0.0:
        return x.copy()
    strength = F(min(max(float(strength), 0.0), 1.0))
    up = resample_linear(x, rate, 48000) if rate != 48000 else x.copy()
    den = []
    n_frames = (len(up) + 479) // 480
    for i in range(n_frames):
        chunk = up[i * 480:(i + 1) * 480]
        frame = np.zeros(480, F)
        frame[:len(chunk)] = chunk * F(32767.0)
        out = np.asarray(process_frame(frame), F)
        if i == 0:
            fade = (np.arange(480, dtype=F) / F(480)).astype(F)
            den.append((out * fade / F(32767.0)).astype(F))
        elif len(chunk) < 480:
            den.append((out[:len(chunk)] / F(32767.0)).astype(F))
        else:
            den.append((out / F(32767.0)).astype(F))
    den = np.concatenate(den) if den else np.zeros(0, F)
    down = resample_linear(den, 48000, rate) if rate != 48000 else den
    res = np.zeros(len(x), F)
    m = min(len(x), len(down))
    res[:m] = down[:m]
    if strength < 1.0:
        res = (x * F(F(1) - strength) + res * strength).astype(F)
    return res
