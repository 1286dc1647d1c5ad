"""ctypes binding of oracle/libwhisper_ref.so (the CPU restatement in whisper_ref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported from openhush_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwhisper_ref.so")

CHUNK_SAMPLES = 480000
CHUNK_FRAMES = 3000


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "whisper_ref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class AudioInfo(C.Structure):
    _fields_ = [("duration_secs", C.c_float), ("sample_count", C.c_int64), ("min_value", C.c_float),
                ("max_value", C.c_float), ("rms", C.c_float), ("nan_count", C.c_int64), ("inf_count", C.c_int64)]


class SampleParams(C.Structure):
    _fields_ = [("lang_id", C.c_int32), ("translate", C.c_int32), ("no_timestamps", C.c_int32),
                ("suppress_blank", C.c_int32), ("max_initial_ts", C.c_int32), ("n_max", C.c_int32),
                ("force_len", C.c_int32)]


class DecodePolicy(C.Structure):
    _fields_ = [("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float), ("no_speech_thold", C.c_float)]


class SeqEval(C.Structure):
    _fields_ = [("n_sampled", C.c_int32), ("result_len", C.c_int32), ("n_keep", C.c_int32), ("seek_delta", C.c_int32),
                ("failed", C.c_int32), ("completed", C.c_int32), ("avg_logprob", C.c_float), ("entropy", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class WindowResult(C.Structure):
    _fields_ = [("ev", SeqEval), ("temperature", C.c_float), ("no_speech_prob", C.c_float), ("n_passes", C.c_int32), ("no_speech", C.c_int32)]


class MT19937(C.Structure):
    """oracle's own Mersenne twister (std::mt19937 semantics)"""
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]

    def __init__(self, seed: int = 0):
        super().__init__()
        lib().ref_mt_seed(C.byref(self), seed)

    def next(self) -> int:
        return int(lib().ref_mt_next(C.byref(self)))


AUDIO_ERRORS = {0: "Ok", 1: "Empty", 2: "InvalidSampleRate", 3: "TooLong", 4: "TooShort", 5: "ContainsNaN", 6: "ContainsInfinite"}

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        L.ref_model_load.restype = C.c_void_p
        L.ref_model_load.argtypes = [C.c_char_p]
        L.ref_model_synth.restype = C.c_void_p
        L.ref_model_synth.argtypes = [ip, C.c_uint32]
        L.ref_model_free.argtypes = [C.c_void_p]
        L.ref_model_save.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_model_hparams.argtypes = [C.c_void_p, ip]
        L.ref_model_mel_filters.restype = fp
        L.ref_model_mel_filters.argtypes = [C.c_void_p]
        L.ref_model_tensor.restype = fp
        L.ref_model_tensor.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.ref_model_n_tensors.argtypes = [C.c_void_p]
        L.ref_model_tensor_name.restype = C.c_char_p
        L.ref_model_tensor_name.argtypes = [C.c_void_p, C.c_int]
        L.ref_model_special_tokens.argtypes = [C.c_void_p, ip]
        L.ref_model_n_langs.argtypes = [C.c_void_p]
        L.ref_validate_audio.argtypes = [fp, C.c_int64, C.c_uint32, C.POINTER(AudioInfo)]
        L.ref_log_mel.argtypes = [C.c_void_p, fp, C.c_int64, C.c_int, fp]
        L.ref_log_mel_recording_max.argtypes = [C.c_void_p, fp, C.c_int64]
        L.ref_log_mel_recording_max.restype = C.c_double
        L.ref_log_mel_seek.argtypes = [C.c_void_p, fp, C.c_int64, C.c_int64, C.c_double, fp]
        L.ref_encode.argtypes = [C.c_void_p, fp, fp, fp, fp, fp]
        L.ref_state_new.restype = C.c_void_p
        L.ref_state_new.argtypes = [C.c_void_p]
        L.ref_state_free.argtypes = [C.c_void_p]
        L.ref_set_encoder_output.argtypes = [C.c_void_p, fp]
        L.ref_state_cross_k.restype = fp
        L.ref_state_cross_k.argtypes = [C.c_void_p]
        L.ref_state_cross_v.restype = fp
        L.ref_state_cross_v.argtypes = [C.c_void_p]
        L.ref_decode.argtypes = [C.c_void_p, ip, C.c_int, C.c_int, C.c_int, fp, fp]
        L.ref_default_sample_params.argtypes = [C.c_void_p, C.POINTER(SampleParams)]
        L.ref_build_prompt.argtypes = [C.c_void_p, C.POINTER(SampleParams), ip]
        L.ref_process_logits.argtypes = [C.c_void_p, C.POINTER(SampleParams), fp, ip, C.c_int, fp, fp]
        L.ref_greedy.argtypes = [C.c_void_p, C.POINTER(SampleParams), ip, fp, fp, fp]
        L.ref_process_logits_ex.argtypes = [C.c_void_p, C.POINTER(SampleParams), fp, ip, C.c_int, fp, C.c_float, fp, fp, fp]
        L.ref_greedy_ex.argtypes = [C.c_void_p, C.POINTER(SampleParams), fp, ip, C.c_int, ip, fp, fp, ip, ip, fp]
        L.ref_transcribe_chunk.argtypes = [C.c_void_p, fp, C.c_int64, C.c_int, C.POINTER(SampleParams), ip,
                                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        dp = C.POINTER(C.c_double)
        L.ref_default_decode_policy.argtypes = [C.POINTER(DecodePolicy)]
        L.ref_mel_frames.argtypes = [C.c_int64]
        L.ref_mt_seed.argtypes = [C.POINTER(MT19937), C.c_uint32]
        L.ref_mt_next.argtypes = [C.POINTER(MT19937)]
        L.ref_mt_next.restype = C.c_uint32
        L.ref_evaluate_sequence.argtypes = [C.c_void_p, ip, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SeqEval)]
        L.ref_pass_needs_fallback.argtypes = [C.POINTER(SeqEval), C.POINTER(DecodePolicy), C.c_float, C.c_int]
        L.ref_window_is_no_speech.argtypes = [C.POINTER(SeqEval), C.POINTER(DecodePolicy), C.c_float]
        L.ref_decode_pass.argtypes = [C.c_void_p, C.POINTER(SampleParams), fp, C.c_float, C.POINTER(MT19937), ip, C.c_int, ip, fp, ip, fp, dp, fp]
        L.ref_decode_window.argtypes = [C.c_void_p, C.POINTER(SampleParams), C.POINTER(DecodePolicy), fp, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(MT19937), ip, fp, C.POINTER(WindowResult)]
        L.ref_beam_search.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(SampleParams), fp, ip, fp, ip, ip, fp, ip, ip]
        L.ref_score_sequence.argtypes = [C.c_void_p, C.POINTER(SampleParams), fp, ip, C.c_int, C.c_int]
        L.ref_score_sequence.restype = C.c_float
        L.ref_num_threads.restype = C.c_int
        L.ref_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def validate_audio(samples: np.ndarray, rate: int) -> Tuple[str, AudioInfo]:
    info = AudioInfo()
    s = np.ascontiguousarray(samples, dtype=np.float32)
    ptr = _fp(s) if s.size else C.cast(None, C.POINTER(C.c_float))
    code = lib().ref_validate_audio(ptr, s.size, rate, C.byref(info))
    return AUDIO_ERRORS[code], info


class Model:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: model load failed")
        self.h = C.c_void_p(handle)
        hp = np.zeros(11, dtype=np.int32)
        lib().ref_model_hparams(self.h, _ip(hp))
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer,
         self.n_text_ctx, self.n_text_state, self.n_text_head, self.n_text_layer, self.n_mels, self.ftype) = [int(x) for x in hp]
        st = np.zeros(10, dtype=np.int32)
        lib().ref_model_special_tokens(self.h, _ip(st))
        (self.tok_eot, self.tok_sot, self.tok_translate, self.tok_transcribe, self.tok_solm, self.tok_prev,
         self.tok_nosp, self.tok_not, self.tok_beg, self.tok_blank) = [int(x) for x in st]
        self.n_langs = int(lib().ref_model_n_langs(self.h))

    @classmethod
    def load(cls, path: str) -> "Model":
        return cls(lib().ref_model_load(path.encode()))

    @classmethod
    def synth(cls, hparams_list: List[int], seed: int = 1234) -> "Model":
        hp = np.asarray(hparams_list, dtype=np.int32)
        return cls(lib().ref_model_synth(_ip(hp), seed))

    def save(self, path: str) -> None:
        """write the model as a ggml file (the format Model.load and libohw read)"""
        if lib().ref_model_save(self.h, path.encode()) != 0:
            raise OSError(f"oracle: cannot write {path}")

    def close(self):
        if self.h:
            lib().ref_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor_names(self) -> List[str]:
        return [lib().ref_model_tensor_name(self.h, i).decode() for i in range(lib().ref_model_n_tensors(self.h))]

    def tensor(self, name: str) -> np.ndarray:
        n = C.c_int64(0)
        p = lib().ref_model_tensor(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def mel_filters(self) -> np.ndarray:
        p = lib().ref_model_mel_filters(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n_mels, 201)).copy()

    def log_mel(self, pcm: np.ndarray, mode: int = 0) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        out = np.empty((self.n_mels, CHUNK_FRAMES), dtype=np.float32)
        lib().ref_log_mel(self.h, _fp(pcm), pcm.size, mode, _fp(out))
        return out

    def recording_max(self, pcm: np.ndarray) -> float:
        """maximum of log10 mel power over every frame of the recording (whisper.cpp's whole-input clamp, as recalled)"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        return float(lib().ref_log_mel_recording_max(self.h, _fp(pcm), pcm.size))

    def log_mel_seek(self, pcm: np.ndarray, seek: int, rec_max: Optional[float] = None) -> np.ndarray:
        """frames [seek, seek + 3000) of the recording-wide spectrogram, clamped with the recording's maximum"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if rec_max is None:
            rec_max = self.recording_max(pcm)
        out = np.empty((self.n_mels, CHUNK_FRAMES), dtype=np.float32)
        lib().ref_log_mel_seek(self.h, _fp(pcm), pcm.size, int(seek), float(rec_max), _fp(out))
        return out

    def encode(self, mel: np.ndarray, taps: bool = False):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        d = self.n_audio_state
        out = np.empty((self.n_audio_ctx, d), dtype=np.float32)
        if taps:
            t1 = np.empty((CHUNK_FRAMES, d), dtype=np.float32)
            t2 = np.empty((self.n_audio_ctx, d), dtype=np.float32)
            t3 = np.empty((self.n_audio_ctx, d), dtype=np.float32)
            lib().ref_encode(self.h, _fp(mel), _fp(out), _fp(t1), _fp(t2), _fp(t3))
            return out, t1, t2, t3
        nul = C.cast(None, C.POINTER(C.c_float))
        lib().ref_encode(self.h, _fp(mel), _fp(out), nul, nul, nul)
        return out

    def default_params(self) -> SampleParams:
        p = SampleParams()
        lib().ref_default_sample_params(self.h, C.byref(p))
        return p

    def build_prompt(self, p: SampleParams) -> List[int]:
        out = np.zeros(8, dtype=np.int32)
        n = lib().ref_build_prompt(self.h, C.byref(p), _ip(out))
        return [int(x) for x in out[:n]]

    def process_logits(self, p: SampleParams, logits: np.ndarray, cur: List[int]):
        """returns (token, logprob_of_token, filtered_logits, logprobs)"""
        lg = np.ascontiguousarray(logits, dtype=np.float32).copy()
        lps = np.empty_like(lg)
        c = np.asarray(cur if len(cur) else [0], dtype=np.int32)
        lp = C.c_float(0)
        tok = lib().ref_process_logits(self.h, C.byref(p), _fp(lg), _ip(c), len(cur), _fp(lps), C.byref(lp))
        return int(tok), float(lp.value), lg, lps

    def process_logits_ex(self, p: SampleParams, logits: np.ndarray, cur: List[int], bias: Optional[np.ndarray] = None,
                          temperature: float = 0.0):
        """(token, logprob_of_token, no_speech_prob or None) with the additive bias / temperature applied first"""
        lg = np.ascontiguousarray(logits, dtype=np.float32).copy()
        c = np.asarray(cur if len(cur) else [0], dtype=np.int32)
        lp, ns = C.c_float(0), C.c_float(-1.0)
        nul = C.cast(None, C.POINTER(C.c_float))
        bp = _fp(np.ascontiguousarray(bias, dtype=np.float32)) if bias is not None else nul
        tok = lib().ref_process_logits_ex(self.h, C.byref(p), _fp(lg), _ip(c), len(cur), bp, temperature, nul, C.byref(lp), C.byref(ns))
        return int(tok), float(lp.value), (float(ns.value) if len(cur) == 0 else None)

    def transcribe_chunk(self, pcm: np.ndarray, p: Optional[SampleParams] = None, mel_mode: int = 0):
        """(tokens, (t_mel, t_enc, t_dec)) — the timed CPU baseline path"""
        p = p or self.default_params()
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        out = np.zeros(self.n_text_ctx, dtype=np.int32)
        t = [C.c_double(0), C.c_double(0), C.c_double(0)]
        n = lib().ref_transcribe_chunk(self.h, _fp(pcm), pcm.size, mel_mode, C.byref(p), _ip(out),
                                       C.byref(t[0]), C.byref(t[1]), C.byref(t[2]))
        return [int(x) for x in out[:n]], tuple(x.value for x in t)


class State:
    def __init__(self, model: Model):
        self.m = model
        self.h = C.c_void_p(lib().ref_state_new(model.h))

    def close(self):
        if self.h:
            lib().ref_state_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_encoder_output(self, enc: np.ndarray):
        enc = np.ascontiguousarray(enc, dtype=np.float32)
        lib().ref_set_encoder_output(self.h, _fp(enc))

    def cross_kv(self):
        m = self.m
        shape = (m.n_text_layer, m.n_audio_ctx, m.n_text_state)
        k = np.ctypeslib.as_array(lib().ref_state_cross_k(self.h), shape=shape).copy()
        v = np.ctypeslib.as_array(lib().ref_state_cross_v(self.h), shape=shape).copy()
        return k, v

    def decode(self, tokens: List[int], n_past: int, all_pos: bool = False, want_hidden: bool = False):
        t = np.asarray(tokens, dtype=np.int32)
        V = self.m.n_vocab
        logits = np.empty((len(tokens), V) if all_pos else (V,), dtype=np.float32)
        hid = np.empty(self.m.n_text_state, dtype=np.float32)
        lib().ref_decode(self.h, _ip(t), len(tokens), n_past, 1 if all_pos else 0, _fp(logits), _fp(hid))
        return (logits, hid) if want_hidden else logits

    def greedy(self, p: Optional[SampleParams] = None):
        """(tokens, logprobs, margins, avg_logprob)"""
        p = p or self.m.default_params()
        n_cap = self.m.n_text_ctx
        out = np.zeros(n_cap, dtype=np.int32)
        lps = np.zeros(n_cap, dtype=np.float32)
        mg = np.zeros(n_cap, dtype=np.float32)
        avg = C.c_float(0)
        n = lib().ref_greedy(self.h, C.byref(p), _ip(out), _fp(lps), _fp(mg), C.byref(avg))
        return [int(x) for x in out[:n]], lps[:n].copy(), mg[:n + 1].copy(), float(avg.value)


def _state_greedy_ex(self, p: Optional[SampleParams] = None, bias: Optional[np.ndarray] = None, forced: Optional[List[int]] = None):
    """dict(tokens, logprobs [n (+1 when ended_by_eot)], margins [steps], choice [steps], ended_by_eot, no_speech_prob).
    With `forced` the walk follows those tokens (an end-of-text entry ends it) while the oracle's own pick and top-2
    margin are recorded at every step."""
    p = p or self.m.default_params()
    cap = self.m.n_text_ctx + 1
    out = np.zeros(cap, dtype=np.int32)
    lps = np.zeros(cap, dtype=np.float32)
    mg = np.zeros(cap, dtype=np.float32)
    ch = np.full(cap, -1, dtype=np.int32)
    eot, ns = C.c_int32(0), C.c_float(0)
    nul = C.cast(None, C.POINTER(C.c_float))
    bp = _fp(np.ascontiguousarray(bias, dtype=np.float32)) if bias is not None else nul
    f = np.asarray(forced, dtype=np.int32) if forced is not None and len(forced) else None
    fptr = _ip(f) if f is not None else C.cast(None, C.POINTER(C.c_int32))
    nf = len(f) if f is not None else 0
    if forced is not None and f is None:
        f = np.zeros(1, np.int32); fptr = _ip(f)      # forced = []: one recorded step, nothing consumed
    n = lib().ref_greedy_ex(self.h, C.byref(p), bp, fptr, nf, _ip(out), _fp(lps), _fp(mg), _ip(ch), C.byref(eot), C.byref(ns))
    steps = int((ch >= 0).sum())
    return {"tokens": [int(x) for x in out[:n]], "logprobs": lps[:n + (1 if eot.value else 0)].copy(), "margins": mg[:steps].copy(),
            "choice": [int(x) for x in ch[:steps]], "ended_by_eot": bool(eot.value), "no_speech_prob": float(ns.value)}


State.greedy_ex = _state_greedy_ex


def default_policy() -> DecodePolicy:
    q = DecodePolicy()
    lib().ref_default_decode_policy(C.byref(q))
    return q


def mel_frames(n_samples: int) -> int:
    return int(lib().ref_mel_frames(n_samples))


def evaluate_sequence(model: Model, tokens, plogs, seek: int, seek_end: int, n_max: int, no_timestamps: bool = False, window_mode: int = 0) -> SeqEval:
    t = np.asarray(list(tokens) or [0], dtype=np.int32)
    l = np.asarray(list(plogs) or [0.0], dtype=np.float32)
    ev = SeqEval()
    lib().ref_evaluate_sequence(model.h, _ip(t), _fp(l), len(tokens), seek, seek_end, n_max, int(no_timestamps), window_mode, C.byref(ev))
    return ev


def pass_needs_fallback(ev: SeqEval, pol: DecodePolicy, no_speech_prob: float, is_last: bool) -> bool:
    return bool(lib().ref_pass_needs_fallback(C.byref(ev), C.byref(pol), no_speech_prob, int(is_last)))


def window_is_no_speech(ev: SeqEval, pol: DecodePolicy, no_speech_prob: float) -> bool:
    return bool(lib().ref_window_is_no_speech(C.byref(ev), C.byref(pol), no_speech_prob))


def _state_decode_pass(self, p: SampleParams, bias, temperature: float, rng: Optional[MT19937], forced=None):
    """One pass over the window at `temperature`: dict(tokens (end-of-text last when sampled), plogs, choice, margins, gaps,
    no_speech_prob).  With `forced` the walk follows those tokens; choice / margins (T = 0) / gaps (T > 0: distance of the
    draw to the nearer edge of the chosen interval) record what the oracle itself would have done at every step."""
    cap = self.m.n_text_ctx + 1
    out = np.zeros(cap, np.int32); lps = np.zeros(cap, np.float32); ch = np.full(cap, -1, np.int32)
    mg = np.zeros(cap, np.float32); gaps = np.ones(cap, np.float64)
    ns = C.c_float(0)
    nul = C.cast(None, C.POINTER(C.c_float))
    bp = _fp(np.ascontiguousarray(bias, dtype=np.float32)) if bias is not None else nul
    f = np.asarray(forced, dtype=np.int32) if forced is not None and len(forced) else None
    n = lib().ref_decode_pass(self.h, C.byref(p), bp, temperature, C.byref(rng) if rng is not None else None,
                              _ip(f) if f is not None else C.cast(None, C.POINTER(C.c_int32)), len(f) if f is not None else 0,
                              _ip(out), _fp(lps), _ip(ch), _fp(mg), gaps.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ns))
    steps = int((ch >= 0).sum())
    return {"tokens": [int(x) for x in out[:n]], "plogs": lps[:n].copy(), "choice": [int(x) for x in ch[:steps]],
            "margins": mg[:steps].copy(), "gaps": gaps[:steps].copy(), "no_speech_prob": float(ns.value)}


def _state_decode_window(self, p: SampleParams, pol: DecodePolicy, bias=None, seek: int = 0, seek_end: int = 2999, window_mode: int = 0,
                         rng: Optional[MT19937] = None):
    """whisper.cpp's whole per-window procedure (T = 0, then the ladder): (kept tokens, WindowResult, all sampled tokens)"""
    cap = self.m.n_text_ctx + 1
    out = np.zeros(cap, np.int32); lps = np.zeros(cap, np.float32)
    res = WindowResult()
    nul = C.cast(None, C.POINTER(C.c_float))
    bp = _fp(np.ascontiguousarray(bias, dtype=np.float32)) if bias is not None else nul
    n = lib().ref_decode_window(self.h, C.byref(p), C.byref(pol), bp, seek, seek_end, window_mode, C.byref(rng) if rng is not None else None,
                                _ip(out), _fp(lps), C.byref(res))
    return [int(x) for x in out[:n]], res, [int(x) for x in out[:res.ev.n_sampled]]


def beam_search(model: Model, enc: np.ndarray, p: SampleParams, beam_size: int, bias=None):
    """the published Whisper beam search on one encoded window: dict(tokens, sum_logprob, n_finished, candidates=[(tokens, sum)])"""
    states = [State(model) for _ in range(beam_size)]
    for s in states:
        s.set_encoder_output(enc)
    arr = (C.c_void_p * beam_size)(*[s.h for s in states])
    cap = model.n_text_ctx
    out = np.zeros(cap, np.int32); allt = np.zeros((beam_size, cap), np.int32); alll = np.zeros(beam_size, np.int32)
    alls = np.zeros(beam_size, np.float32)
    sm, na, nf = C.c_float(0), C.c_int32(0), C.c_int32(0)
    nul = C.cast(None, C.POINTER(C.c_float))
    bp = _fp(np.ascontiguousarray(bias, dtype=np.float32)) if bias is not None else nul
    n = lib().ref_beam_search(arr, beam_size, C.byref(p), bp, _ip(out), C.byref(sm), _ip(allt), _ip(alll), _fp(alls), C.byref(na), C.byref(nf))
    return {"tokens": [int(x) for x in out[:n]], "sum_logprob": float(sm.value), "n_finished": int(nf.value),
            "candidates": [([int(x) for x in allt[i, :alll[i]]], float(alls[i])) for i in range(na.value)]}


def _state_score_sequence(self, p: SampleParams, tokens, ended: bool, bias=None) -> float:
    t = np.asarray(list(tokens) or [0], dtype=np.int32)
    nul = C.cast(None, C.POINTER(C.c_float))
    bp = _fp(np.ascontiguousarray(bias, dtype=np.float32)) if bias is not None else nul
    return float(lib().ref_score_sequence(self.h, C.byref(p), bp, _ip(t), len(tokens), int(ended)))


State.score_sequence = _state_score_sequence
State.decode_pass = _state_decode_pass
State.decode_window = _state_decode_window


def num_threads() -> int:
    return int(lib().ref_num_threads())


def set_num_threads(n: int) -> None:
    lib().ref_set_num_threads(n)
