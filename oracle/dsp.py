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
