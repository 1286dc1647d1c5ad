"""ctypes binding of libohw.so and a Python mirror of the reference's `WhisperEngine`.

The product path is the HIP library: nothing here computes; there is no CPU fallback and no import
of oracle/.  If libohw.so is missing or cannot be loaded this module raises at import of the
library handle (`lib()`), loudly.

Mirror of the reference interface (reference src/engine/whisper.rs):
    WhisperEngine.new(model_path, language, translate, use_gpu)   :129-179
    WhisperEngine.transcribe(AudioBuffer) -> TranscriptionResult  :204-310
    WhisperEngine.benchmark(safety_margin) -> BenchmarkResult     :334-387
    WhisperError variants                                          :14-27
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OHW_LIB") or os.path.join(_HERE, "libohw.so")   # OHW_LIB: instrumented builds for tools/

CHUNK_SAMPLES = 480000
CHUNK_FRAMES = 3000

OHW_DTYPE_AUTO, OHW_DTYPE_BF16, OHW_DTYPE_F16 = -1, 0, 1
OHW_MEL_REFLECT, OHW_MEL_ZERO_TAIL = 0, 1
OHW_WINDOW_FIXED, OHW_WINDOW_SEEK, OHW_WINDOW_FIXED_RECORDING_MEL = 0, 1, 2
OHW_SCHEDULE_SEQUENTIAL, OHW_SCHEDULE_PIPELINE, OHW_SCHEDULE_LANES = 0, 1, 2
EPI_BIAS_T, EPI_BIAS_GELU_T, EPI_BIAS_RESID_F32, EPI_F32 = 0, 1, 2, 4

OHW_E_MODEL_NOT_FOUND, OHW_E_LOAD_FAILED, OHW_E_TRANSCRIBE = -3001, -3002, -3003
OHW_E_NO_GPU, OHW_E_OOM, OHW_E_INVALID_ARG, OHW_E_VALIDATION = -3004, -3005, -3006, -3007


class HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_vocab", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                         "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_mels", "ftype")]

    def as_list(self):
        return [int(getattr(self, n)) for n, _ in self._fields_]


class SpecialTokens(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("eot", "sot", "translate", "transcribe", "solm", "prev", "nosp", "no_timestamps",
                                         "timestamp_begin", "blank", "n_langs")]


class AudioInfo(C.Structure):
    _fields_ = [("error", C.c_int32), ("duration_secs", C.c_float), ("sample_count", C.c_int64), ("min_value", C.c_float),
                ("max_value", C.c_float), ("rms", C.c_float), ("nan_count", C.c_int64), ("inf_count", C.c_int64)]


class SampleParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("lang_id", "translate", "no_timestamps", "suppress_blank", "max_initial_ts", "n_max", "force_len")]


class PreprocessConfig(C.Structure):
    """ohw_preprocess_config: the reference's [audio] preprocessing settings (src/config.rs:1020-1070, defaults :1129-1160)"""
    _fields_ = [("preprocessing", C.c_int32), ("normalization_enabled", C.c_int32), ("normalization_target_db", C.c_float),
                ("compression_enabled", C.c_int32), ("compression_threshold_db", C.c_float), ("compression_ratio", C.c_float),
                ("compression_attack_ms", C.c_float), ("compression_release_ms", C.c_float), ("compression_makeup_gain_db", C.c_float),
                ("limiter_enabled", C.c_int32), ("limiter_ceiling_db", C.c_float), ("limiter_release_ms", C.c_float)]


class GreedyResult(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_int32)), ("n_tokens", C.POINTER(C.c_int32)), ("sum_logprob", C.POINTER(C.c_float)),
                ("token_logprobs", C.POINTER(C.c_float)), ("ended_by_eot", C.POINTER(C.c_int32)), ("no_speech_prob", C.POINTER(C.c_float))]


class VadConfig(C.Structure):
    """reference src/vad/mod.rs:57-100"""
    _fields_ = [("enabled", C.c_int32), ("threshold", C.c_float), ("min_silence_ms", C.c_uint32), ("min_speech_ms", C.c_uint32),
                ("speech_pad_ms", C.c_uint32)]


class SpeechSegment(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("avg_probability", C.c_float)]


DENOISE_FRAME_FN = C.CFUNCTYPE(C.c_float, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float))
DENOISE_RESET_FN = C.CFUNCTYPE(None, C.c_void_p)


class DenoiseEngineC(C.Structure):
    """ohw_denoise_engine: where a host plugs in nnnoiseless::DenoiseState (reference src/input/audio.rs:275-293)"""
    _fields_ = [("user", C.c_void_p), ("process_frame", DENOISE_FRAME_FN), ("reset", DENOISE_RESET_FN)]


VAD_PROCESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float))
VAD_RESET_FN = C.CFUNCTYPE(None, C.c_void_p)


class VadEngineC(C.Structure):
    _fields_ = [("user", C.c_void_p), ("process", VAD_PROCESS_FN), ("reset", VAD_RESET_FN), ("chunk_size", C.c_int32), ("sample_rate", C.c_uint32)]


class BeamResult(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_int32)), ("n_tokens", C.POINTER(C.c_int32)), ("sum_logprob", C.POINTER(C.c_float)),
                ("n_finished", C.POINTER(C.c_int32))]


class WindowQuality(C.Structure):
    _fields_ = [("n_tokens", C.c_int32), ("avg_logprob", C.c_float), ("entropy", C.c_float), ("would_fallback", C.c_int32),
                ("temperature", C.c_float), ("no_speech_prob", C.c_float), ("no_speech", C.c_int32), ("seek_delta", C.c_int32),
                ("result_len", C.c_int32), ("failed", C.c_int32)]


class DecodePolicy(C.Structure):
    """ohw_decode_policy: whisper.cpp's defaults (temperature_inc 0.2, entropy_thold 2.4, logprob_thold -1.0, no_speech_thold 0.6)"""
    _fields_ = [("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float), ("no_speech_thold", C.c_float)]


class Timings(C.Structure):
    _fields_ = [("mel_ms", C.c_float), ("encode_ms", C.c_float), ("decode_ms", C.c_float), ("total_ms", C.c_float), ("decode_steps", C.c_int32)]


AUDIO_ERRORS = {0: "Ok", 1: "Empty", 2: "InvalidSampleRate", 3: "TooLong", 4: "TooShort", 5: "ContainsNaN", 6: "ContainsInfinite"}

# every symbol include/ohw.h declares
EXPORTS = [
    "ohw_validate_audio", "ohw_ctx_create", "ohw_ctx_create_synthetic", "ohw_ctx_info", "ohw_token_text", "ohw_ctx_free",
    "ohw_state_create", "ohw_state_free", "ohw_state_set_stream", "ohw_state_max_batch", "ohw_mel", "ohw_recording_set", "ohw_mel_seek", "ohw_encode", "ohw_decode",
    "ohw_default_sample_params", "ohw_sample_greedy_host", "ohw_greedy", "ohw_state_timings", "ohw_engine_new",
    "ohw_engine_transcribe", "ohw_engine_last_tokens", "ohw_engine_benchmark", "ohw_engine_free", "ohw_engine_state",
    "ohw_engine_ctx", "ohw_lang_id_to_code", "ohw_lang_code_to_id", "ohw_last_error", "ohw_abi_version", "ohw_state_fetch",
    "ohw_dbg_gemm", "ohw_dbg_attention", "ohw_state_profile_begin", "ohw_state_profile_end", "ohw_ctx_weight_digest", "ohw_detect_language", "ohw_state_ctx", "ohw_engine_set_window_mode", "ohw_engine_last_text", "ohw_engine_last_quality",
    "ohw_stream_create", "ohw_stream_destroy", "ohw_stream_wait", "ohw_stream_sync",
    "ohw_ctx_create_shell", "ohw_ctx_blob_size", "ohw_ctx_blob_export", "ohw_ctx_blob_import",
    "ohw_default_preprocess_config", "ohw_preprocess_audio", "ohw_dsp_rms_db", "ohw_dsp_apply_gain", "ohw_dsp_normalize_rms",
    "ohw_dsp_compress", "ohw_dsp_limit", "ohw_dsp_resample_linear",
    "ohw_tracker_new", "ohw_tracker_free", "ohw_tracker_add_pending", "ohw_tracker_add_result", "ohw_tracker_take_ready", "ohw_tracker_ready_get",
    "ohw_tracker_reset_dedup", "ohw_tracker_is_empty", "ohw_tracker_is_pending", "ohw_tracker_pending_count", "ohw_tracker_waiting_count", "ohw_extract_chunk",
    "ohw_chunk_scheduler_new", "ohw_chunk_scheduler_free", "ohw_chunk_scheduler_tick", "ohw_chunk_scheduler_position", "ohw_chunk_scheduler_next_id",
    "ohw_chunk_scheduler_rejected", "ohw_resampler_create", "ohw_resampler_free", "ohw_resampler_out_len", "ohw_resampler_run", "ohw_greedy_ex", "ohw_state_set_logit_bias", "ohw_state_set_batch_invariant", "ohw_dbg_sample",
    "ohw_decode_active", "ohw_rng_new", "ohw_rng_free", "ohw_sample_host", "ohw_default_decode_policy", "ohw_engine_set_decode_policy",
    "ohw_engine_last_trace", "ohw_engine_set_schedule", "ohw_ctx_dtype",
    "ohw_beam_search", "ohw_encode_slice", "ohw_dsp_resample_sinc", "ohw_default_vad_config", "ohw_vad_state_new", "ohw_vad_state_free", "ohw_vad_state_update",
    "ohw_vad_state_is_speech", "ohw_vad_state_speech_start", "ohw_vad_state_reset", "ohw_vad_energy_engine", "ohw_vad_energy_engine_free",
    "ohw_vad_run",
    "ohw_pool_create", "ohw_pool_transcribe", "ohw_pool_last_text", "ohw_pool_last_tokens", "ohw_pool_last_quality",
    "ohw_dbg_counter", "ohw_dsp_denoise", "ohw_denoise_passthrough_engine", "ohw_preprocess_audio_ex", "ohw_pool_set_window_mode",
    "ohw_state_set_persistent", "ohw_pool_broadcast_note", "ohw_pool_create_synthetic", "ohw_pool_set_force_len", "ohw_pool_set_schedule", "ohw_engine_set_force_len",
    "ohw_pool_set_decode_policy", "ohw_pool_n_devices", "ohw_pool_broadcast_kind", "ohw_pool_engine", "ohw_pool_free",
]


class WhisperError(RuntimeError):
    """reference src/engine/whisper.rs:14-27"""
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


class ModelNotFound(WhisperError):
    pass


class LoadFailed(WhisperError):
    pass


class TranscriptionFailed(WhisperError):
    pass


class ValidationFailed(WhisperError):
    def __init__(self, code: int, msg: str, info: Optional[AudioInfo] = None):
        super().__init__(code, msg)
        self.info = info
        self.kind = AUDIO_ERRORS.get(info.error, "?") if info is not None else "?"


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m openhush_amd.build` (no fallback path exists)")
        L = C.CDLL(LIB_PATH)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.ohw_last_error.restype = C.c_char_p
        L.ohw_lang_id_to_code.restype = C.c_char_p
        L.ohw_lang_id_to_code.argtypes = [C.c_int32]
        L.ohw_lang_code_to_id.argtypes = [C.c_char_p]
        L.ohw_validate_audio.argtypes = [fp, C.c_int64, C.c_uint32, C.POINTER(AudioInfo)]
        L.ohw_ctx_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_ctx_create_synthetic.argtypes = [C.POINTER(HParams), C.c_uint32, C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_ctx_info.argtypes = [vp, C.POINTER(HParams), C.POINTER(SpecialTokens)]
        L.ohw_ctx_dtype.argtypes = [vp]
        L.ohw_ctx_create_shell.argtypes = [C.POINTER(HParams), C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_ctx_blob_size.argtypes = [vp]
        L.ohw_ctx_blob_size.restype = C.c_size_t
        L.ohw_ctx_blob_export.argtypes = [vp, vp, C.c_size_t]
        L.ohw_ctx_blob_import.argtypes = [vp, vp, C.c_size_t]
        L.ohw_token_text.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p)]
        L.ohw_ctx_free.argtypes = [vp]
        L.ohw_ctx_free.restype = None
        L.ohw_state_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.ohw_state_free.argtypes = [vp]
        L.ohw_state_free.restype = None
        L.ohw_state_set_stream.argtypes = [vp, vp]
        L.ohw_default_preprocess_config.argtypes = [C.POINTER(PreprocessConfig)]
        L.ohw_default_preprocess_config.restype = None
        L.ohw_preprocess_audio.argtypes = [fp, C.c_int64, C.c_uint32, C.POINTER(PreprocessConfig)]
        L.ohw_dsp_denoise.argtypes = [fp, C.c_int64, C.c_uint32, C.c_float, C.POINTER(DenoiseEngineC)]
        L.ohw_denoise_passthrough_engine.argtypes = [C.POINTER(DenoiseEngineC)]
        L.ohw_denoise_passthrough_engine.restype = None
        L.ohw_preprocess_audio_ex.argtypes = [fp, C.c_int64, C.c_uint32, C.POINTER(PreprocessConfig), C.c_int, C.c_float, C.POINTER(DenoiseEngineC)]
        L.ohw_pool_set_window_mode.argtypes = [vp, C.c_int]
        L.ohw_pool_create_synthetic.argtypes = [C.POINTER(HParams), C.c_uint32, C.c_char_p, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_pool_set_force_len.argtypes = [vp, C.c_int]
        L.ohw_pool_set_schedule.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.ohw_engine_set_force_len.argtypes = [vp, C.c_int]
        L.ohw_pool_broadcast_note.argtypes = [vp]
        L.ohw_pool_broadcast_note.restype = C.c_char_p
        L.ohw_dsp_rms_db.argtypes = [fp, C.c_int64]
        L.ohw_dsp_rms_db.restype = C.c_float
        L.ohw_dsp_apply_gain.argtypes = [fp, C.c_int64, C.c_float]
        L.ohw_dsp_apply_gain.restype = None
        L.ohw_dsp_normalize_rms.argtypes = [fp, C.c_int64, C.c_float]
        L.ohw_dsp_normalize_rms.restype = None
        L.ohw_dsp_compress.argtypes = [fp, C.c_int64, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
        L.ohw_dsp_compress.restype = None
        L.ohw_dsp_limit.argtypes = [fp, C.c_int64, C.c_uint32, C.c_float, C.c_float]
        L.ohw_dsp_limit.restype = C.c_int64
        L.ohw_dsp_resample_linear.argtypes = [fp, C.c_int64, C.c_uint32, C.c_uint32, fp, C.c_int64]
        L.ohw_dsp_resample_linear.restype = C.c_int64
        L.ohw_dsp_resample_sinc.argtypes = [fp, C.c_int64, C.c_uint32, C.c_uint32, fp, C.c_int64]
        L.ohw_dsp_resample_sinc.restype = C.c_int64
        L.ohw_resampler_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.ohw_tracker_new.argtypes = [C.c_int]
        L.ohw_tracker_new.restype = vp
        L.ohw_tracker_free.argtypes = [vp]
        L.ohw_tracker_free.restype = None
        L.ohw_tracker_add_pending.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.ohw_tracker_add_result.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float]
        L.ohw_tracker_take_ready.argtypes = [vp]
        L.ohw_tracker_ready_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                            C.POINTER(C.c_float)]
        L.ohw_tracker_reset_dedup.argtypes = [vp]
        L.ohw_tracker_reset_dedup.restype = None
        for fn in (L.ohw_tracker_is_empty, L.ohw_tracker_pending_count, L.ohw_tracker_waiting_count):
            fn.argtypes = [vp]
        L.ohw_tracker_is_pending.argtypes = [vp, C.c_uint64, C.c_uint32]
        L.ohw_extract_chunk.argtypes = [fp, C.c_int64, C.c_int64, C.c_int64, fp, C.c_int64]
        L.ohw_extract_chunk.restype = C.c_int64
        L.ohw_chunk_scheduler_new.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        L.ohw_chunk_scheduler_new.restype = vp
        L.ohw_chunk_scheduler_free.argtypes = [vp]
        L.ohw_chunk_scheduler_free.restype = None
        L.ohw_chunk_scheduler_tick.argtypes = [vp, fp, C.c_int64, C.c_int64, C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
        L.ohw_chunk_scheduler_tick.restype = C.c_int64
        L.ohw_chunk_scheduler_position.argtypes = [vp]
        L.ohw_chunk_scheduler_position.restype = C.c_int64
        L.ohw_chunk_scheduler_next_id.argtypes = [vp]
        L.ohw_chunk_scheduler_next_id.restype = C.c_uint32
        L.ohw_chunk_scheduler_rejected.argtypes = [vp]
        L.ohw_chunk_scheduler_rejected.restype = C.c_int64
        L.ohw_resampler_free.argtypes = [vp]
        L.ohw_resampler_free.restype = None
        L.ohw_resampler_out_len.argtypes = [vp, C.c_int64]
        L.ohw_resampler_out_len.restype = C.c_int64
        L.ohw_resampler_run.argtypes = [vp, vp, C.c_int64, C.c_int, vp, C.c_int64, C.c_int, vp]
        L.ohw_default_vad_config.argtypes = [C.POINTER(VadConfig)]
        L.ohw_default_vad_config.restype = None
        L.ohw_vad_state_new.argtypes = [C.POINTER(VadConfig), C.c_uint32]
        L.ohw_vad_state_new.restype = vp
        L.ohw_vad_state_free.argtypes = [vp]
        L.ohw_vad_state_free.restype = None
        L.ohw_vad_state_update.argtypes = [vp, C.c_float, C.c_int, C.c_int64, C.POINTER(SpeechSegment)]
        L.ohw_vad_state_is_speech.argtypes = [vp]
        L.ohw_vad_state_speech_start.argtypes = [vp]
        L.ohw_vad_state_speech_start.restype = C.c_int64
        L.ohw_vad_state_reset.argtypes = [vp]
        L.ohw_vad_state_reset.restype = None
        L.ohw_vad_energy_engine.argtypes = [C.c_float, C.POINTER(VadEngineC)]
        L.ohw_vad_energy_engine_free.argtypes = [C.POINTER(VadEngineC)]
        L.ohw_vad_energy_engine_free.restype = None
        L.ohw_vad_run.argtypes = [C.POINTER(VadEngineC), C.POINTER(VadConfig), fp, C.c_int64, C.c_int64, C.POINTER(SpeechSegment), C.c_int64]
        L.ohw_vad_run.restype = C.c_int64
        L.ohw_stream_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_stream_destroy.argtypes = [vp]
        L.ohw_stream_wait.argtypes = [vp, vp]
        L.ohw_stream_sync.argtypes = [vp]
        L.ohw_state_max_batch.argtypes = [vp]
        L.ohw_mel.argtypes = [vp, vp, C.c_int64, ip, C.c_int, C.c_int, C.c_int, fp]
        L.ohw_recording_set.argtypes = [vp, vp, C.c_int64, C.c_int, fp]
        L.ohw_mel_seek.argtypes = [vp, ip, C.c_int, fp]
        L.ohw_encode.argtypes = [vp, C.c_int]
        L.ohw_encode_slice.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.ohw_detect_language.argtypes = [vp, C.c_int, ip, fp]
        L.ohw_state_ctx.argtypes = [vp]
        L.ohw_state_ctx.restype = vp
        L.ohw_decode.argtypes = [vp, ip, C.c_int, ip, C.c_int, fp]
        L.ohw_default_sample_params.argtypes = [vp, C.POINTER(SampleParams)]
        L.ohw_default_sample_params.restype = None
        L.ohw_sample_greedy_host.argtypes = [vp, C.POINTER(SampleParams), fp, ip, C.c_int, fp]
        L.ohw_greedy.argtypes = [vp, C.POINTER(SampleParams), C.c_int, ip, ip, C.c_int, fp]
        L.ohw_greedy_ex.argtypes = [vp, C.POINTER(SampleParams), C.c_int, C.c_int, C.POINTER(GreedyResult)]
        L.ohw_state_set_logit_bias.argtypes = [vp, fp, C.c_int]
        L.ohw_state_set_batch_invariant.argtypes = [vp, C.c_int]
        L.ohw_state_set_persistent.argtypes = [vp, C.c_int]
        L.ohw_beam_search.argtypes = [vp, C.POINTER(SampleParams), C.c_int, C.c_int, C.c_int, C.POINTER(BeamResult)]
        L.ohw_dbg_sample.argtypes = [vp, C.POINTER(SampleParams), fp, ip, C.c_int, ip, C.c_int, ip, fp, fp]
        L.ohw_decode_active.argtypes = [vp, ip, C.c_int, ip, C.c_int, ip, fp]
        L.ohw_rng_new.argtypes = [C.c_uint32]
        L.ohw_rng_new.restype = vp
        L.ohw_rng_free.argtypes = [vp]
        L.ohw_rng_free.restype = None
        L.ohw_sample_host.argtypes = [vp, C.POINTER(SampleParams), fp, ip, C.c_int, C.c_float, vp, fp, fp]
        L.ohw_default_decode_policy.argtypes = [C.POINTER(DecodePolicy)]
        L.ohw_default_decode_policy.restype = None
        L.ohw_engine_set_decode_policy.argtypes = [vp, C.POINTER(DecodePolicy)]
        L.ohw_engine_last_trace.argtypes = [vp, C.POINTER(ip), C.POINTER(C.c_int)]
        L.ohw_state_timings.argtypes = [vp, C.POINTER(Timings)]
        L.ohw_engine_new.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_engine_transcribe.argtypes = [vp, fp, C.c_int64, C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p,
                                            C.POINTER(C.c_uint64), C.POINTER(AudioInfo)]
        L.ohw_engine_set_window_mode.argtypes = [vp, C.c_int]
        L.ohw_engine_set_schedule.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.ohw_engine_last_quality.argtypes = [vp, C.POINTER(C.POINTER(WindowQuality)), C.POINTER(C.c_int)]
        L.ohw_engine_last_text.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
        L.ohw_engine_last_tokens.argtypes = [vp, C.POINTER(ip), C.POINTER(C.c_int)]
        L.ohw_engine_benchmark.argtypes = [vp, C.c_float, fp, fp, fp]
        L.ohw_engine_free.argtypes = [vp]
        L.ohw_engine_free.restype = None
        L.ohw_engine_state.argtypes = [vp]
        L.ohw_engine_state.restype = vp
        L.ohw_engine_ctx.argtypes = [vp]
        L.ohw_engine_ctx.restype = vp
        L.ohw_pool_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.ohw_pool_transcribe.argtypes = [vp, fp, C.c_int64, C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(AudioInfo)]
        L.ohw_pool_last_text.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
        L.ohw_pool_last_tokens.argtypes = [vp, C.POINTER(ip), C.POINTER(C.c_int)]
        L.ohw_pool_last_quality.argtypes = [vp, C.POINTER(C.POINTER(WindowQuality)), C.POINTER(C.c_int)]
        L.ohw_pool_set_decode_policy.argtypes = [vp, C.POINTER(DecodePolicy)]
        L.ohw_pool_n_devices.argtypes = [vp]
        L.ohw_pool_broadcast_kind.argtypes = [vp]
        L.ohw_pool_broadcast_kind.restype = C.c_char_p
        L.ohw_pool_engine.argtypes = [vp, C.c_int]
        L.ohw_pool_engine.restype = vp
        L.ohw_pool_free.argtypes = [vp]
        L.ohw_pool_free.restype = None
        L.ohw_state_fetch.argtypes = [vp, C.c_char_p, C.c_int, fp, C.c_int64]
        L.ohw_state_profile_begin.argtypes = [vp, C.c_int]
        L.ohw_state_profile_end.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ohw_ctx_weight_digest.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(C.c_uint64)]
        L.ohw_dbg_gemm.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, vp]
        L.ohw_dbg_attention.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.ohw_dbg_counter.argtypes = [vp, C.c_char_p]
        _lib = L
    return _lib


def last_error() -> str:
    return lib().ohw_last_error().decode("utf-8", "replace")


def _raise(code: int, info: Optional[AudioInfo] = None):
    msg = last_error()
    if code == OHW_E_MODEL_NOT_FOUND:
        raise ModelNotFound(code, msg)
    if code == OHW_E_VALIDATION:
        raise ValidationFailed(code, msg, info)
    if code in (OHW_E_LOAD_FAILED, OHW_E_NO_GPU, OHW_E_OOM):
        raise LoadFailed(code, msg)
    raise TranscriptionFailed(code, msg)


def _check(code: int):
    if code != 0:
        _raise(code)


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def validate_audio(samples: np.ndarray, sample_rate: int) -> AudioInfo:
    """validation::validate_audio — raises ValidationFailed with .kind in AUDIO_ERRORS values"""
    s = np.ascontiguousarray(samples, dtype=np.float32)
    info = AudioInfo()
    ptr = _fp(s) if s.size else C.cast(None, C.POINTER(C.c_float))
    rc = lib().ohw_validate_audio(ptr, s.size, sample_rate, C.byref(info))
    if rc != 0:
        raise ValidationFailed(rc, "Audio validation failed: " + AUDIO_ERRORS.get(info.error, "?"), info)
    return info


def lang_id_to_code(i: int) -> str:
    return lib().ohw_lang_id_to_code(i).decode()


def lang_code_to_id(code: str) -> int:
    return int(lib().ohw_lang_code_to_id(code.encode()))


class Context:
    """ohw_ctx: the model resident in HBM"""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)
        self.hp = HParams()
        self.tok = SpecialTokens()
        _check(lib().ohw_ctx_info(self.h, C.byref(self.hp), C.byref(self.tok)))

    @classmethod
    def from_file(cls, path: str, device: int = 0, dtype: int = OHW_DTYPE_BF16) -> "Context":
        h = C.c_void_p()
        _check(lib().ohw_ctx_create(path.encode(), device, dtype, C.byref(h)))
        return cls(h.value)

    @classmethod
    def synthetic(cls, hparams: Sequence[int], seed: int = 1234, device: int = 0, dtype: int = OHW_DTYPE_BF16) -> "Context":
        hp = HParams(*[int(x) for x in hparams])
        h = C.c_void_p()
        _check(lib().ohw_ctx_create_synthetic(C.byref(hp), seed, device, dtype, C.byref(h)))
        return cls(h.value)

    @classmethod
    def shell(cls, hparams: Sequence[int], device: int = 0, dtype: int = OHW_DTYPE_BF16) -> "Context":
        """every buffer allocated, no weights: the receiving end of a weight broadcast (import_blob)"""
        hp = HParams(*[int(x) for x in hparams])
        h = C.c_void_p()
        _check(lib().ohw_ctx_create_shell(C.byref(hp), device, dtype, C.byref(h)))
        return cls(h.value)

    @property
    def dtype(self) -> int:
        return int(lib().ohw_ctx_dtype(self.h))

    def blob_size(self) -> int:
        return int(lib().ohw_ctx_blob_size(self.h))

    def export_blob(self, dst_device_ptr: int, capacity: int):
        _check(lib().ohw_ctx_blob_export(self.h, C.c_void_p(dst_device_ptr), capacity))

    def import_blob(self, src_device_ptr: int, nbytes: int):
        _check(lib().ohw_ctx_blob_import(self.h, C.c_void_p(src_device_ptr), nbytes))

    def default_params(self) -> SampleParams:
        p = SampleParams()
        lib().ohw_default_sample_params(self.h, C.byref(p))
        return p

    def weight_digests(self) -> dict:
        """name -> 64-bit digest of every resident weight buffer"""
        out, i = {}, 0
        name = C.create_string_buffer(64)
        d = C.c_uint64(0)
        while lib().ohw_ctx_weight_digest(self.h, i, name, C.byref(d)) == 0:
            out[name.value.decode()] = int(d.value)
            i += 1
        return out

    def token_text(self, i: int) -> bytes:
        s = C.c_char_p()
        n = lib().ohw_token_text(self.h, i, C.byref(s))
        return s.value[:n] if n else b""

    def sample_greedy_host(self, p: SampleParams, logits: np.ndarray, cur: List[int]) -> Tuple[int, float]:
        lg = np.ascontiguousarray(logits, dtype=np.float32).copy()
        c = np.asarray(cur if len(cur) else [0], dtype=np.int32)
        lp = C.c_float(0)
        tok = lib().ohw_sample_greedy_host(self.h, C.byref(p), _fp(lg), _ip(c), len(cur), C.byref(lp))
        return int(tok), float(lp.value)

    def sample_host(self, p: SampleParams, logits: np.ndarray, cur: List[int], temperature: float, rng: Optional["HostRng"]):
        """ohw_sample_host -> (token, logprob, no_speech_prob or None)"""
        lg = np.ascontiguousarray(logits, dtype=np.float32).copy()
        c = np.asarray(cur if len(cur) else [0], dtype=np.int32)
        lp, ns = C.c_float(0), C.c_float(-1)
        tok = lib().ohw_sample_host(self.h, C.byref(p), _fp(lg), _ip(c), len(cur), temperature, rng.h if rng else None, C.byref(lp), C.byref(ns))
        return int(tok), float(lp.value), (float(ns.value) if len(cur) == 0 else None)

    def close(self):
        if getattr(self, "h", None):
            lib().ohw_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostRng:
    """ohw_rng: the std::mt19937 whisper.cpp's decoders sample with (seed 0 per call)"""
    def __init__(self, seed: int = 0):
        self.h = C.c_void_p(lib().ohw_rng_new(seed))

    def __del__(self):
        try:
            if self.h:
                lib().ohw_rng_free(self.h)
                self.h = None
        except Exception:
            pass


class Stream:
    """HIP stream restricted to CU-mask bits [first_cu, first_cu + n_cu) (n_cu = 0: all CUs); see include/ohw.h"""
    def __init__(self, device: int = 0, first_cu: int = 0, n_cu: int = 0):
        self.h = C.c_void_p()
        _check(lib().ohw_stream_create(device, first_cu, n_cu, C.byref(self.h)))

    @property
    def ptr(self) -> int:
        return self.h.value or 0

    def wait(self, other: "Stream"):
        _check(lib().ohw_stream_wait(self.h, other.h))

    def sync(self):
        _check(lib().ohw_stream_sync(self.h))

    def close(self):
        if self.h:
            lib().ohw_stream_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class State:
    """ohw_state: activations + KV caches for up to max_batch 30 s windows"""

    def __init__(self, ctx: Context, max_batch: int = 1):
        self.ctx = ctx
        h = C.c_void_p()
        _check(lib().ohw_state_create(ctx.h, max_batch, C.byref(h)))
        self.h = h
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "h", None):
            lib().ohw_state_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: Optional[int]):
        _check(lib().ohw_state_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def counter(self, name: str) -> int:
        """ohw_dbg_counter: step_captures, beam_captures, step_graphs, beam_graphs"""
        v = int(lib().ohw_dbg_counter(self.h, name.encode()))
        if v < 0:
            raise ValueError(name)
        return v

    def mel(self, pcm: np.ndarray, n_samples: Optional[Sequence[int]] = None, mode: int = OHW_MEL_REFLECT, want: bool = True):
        """pcm: [B][stride] float32 host array"""
        pcm = np.ascontiguousarray(np.atleast_2d(pcm), dtype=np.float32)
        B, stride = pcm.shape
        ns = np.asarray(n_samples if n_samples is not None else [min(stride, CHUNK_SAMPLES)] * B, dtype=np.int32)
        out = np.empty((B, self.ctx.hp.n_mels, CHUNK_FRAMES), dtype=np.float32) if want else None
        _check(lib().ohw_mel(self.h, pcm.ctypes.data_as(C.c_void_p), stride, _ip(ns), B, 0, mode,
                             _fp(out) if want else C.cast(None, C.POINTER(C.c_float))))
        return out

    def mel_device(self, pcm_ptr: int, stride: int, n_samples: Sequence[int], mode: int = OHW_MEL_REFLECT):
        """pcm already resident in HBM (e.g. a torch tensor's data_ptr())"""
        ns = np.asarray(n_samples, dtype=np.int32)
        _check(lib().ohw_mel(self.h, C.c_void_p(pcm_ptr), stride, _ip(ns), len(ns), 1, mode, C.cast(None, C.POINTER(C.c_float))))

    def recording_set(self, pcm: np.ndarray) -> float:
        """ohw_recording_set: the whole recording into the state -> log10 of the largest mel power over all its frames"""
        x = np.ascontiguousarray(pcm, dtype=np.float32)
        mx = C.c_float(0.0)
        _check(lib().ohw_recording_set(self.h, x.ctypes.data_as(C.c_void_p), x.size, 0, C.byref(mx)))
        return float(mx.value)

    def mel_seek(self, seek_frames: Sequence[int], want: bool = True):
        """ohw_mel_seek: windows [seek, seek + 3000) of the recording-wide spectrogram -> [B][n_mels][3000] (or None)"""
        sk = np.asarray(seek_frames, dtype=np.int32)
        out = np.empty((len(sk), self.ctx.hp.n_mels, CHUNK_FRAMES), dtype=np.float32) if want else None
        _check(lib().ohw_mel_seek(self.h, _ip(sk), len(sk), _fp(out) if want else C.cast(None, C.POINTER(C.c_float))))
        return out

    def encode(self, batch: int):
        _check(lib().ohw_encode(self.h, batch))

    def encode_slice(self, batch: int, first: int, total: int):
        """encoder + cross K/V of the last mel's `batch` windows into windows [first, first + batch) of a decode batch of `total`"""
        _check(lib().ohw_encode_slice(self.h, batch, first, total))

    def detect_language(self, batch: int):
        """(lang_ids [B], probs [B][n_langs]) for the windows of the last encode"""
        ids = np.zeros(batch, dtype=np.int32)
        probs = np.zeros((batch, self.ctx.tok.n_langs), dtype=np.float32)
        _check(lib().ohw_detect_language(self.h, batch, _ip(ids), _fp(probs)))
        return ids, probs

    def decode(self, tokens: np.ndarray, n_past: Sequence[int]) -> np.ndarray:
        """tokens [B][n_new] -> logits [B][n_vocab] of the last fed position"""
        t = np.ascontiguousarray(np.atleast_2d(tokens), dtype=np.int32)
        B, n_new = t.shape
        npast = np.asarray(n_past, dtype=np.int32)
        out = np.empty((B, self.ctx.hp.n_vocab), dtype=np.float32)
        _check(lib().ohw_decode(self.h, _ip(t), n_new, _ip(npast), B, _fp(out)))
        return out

    def decode_active(self, tokens: np.ndarray, n_past: Sequence[int], active: Sequence[int]) -> np.ndarray:
        """ohw_decode_active: as decode(), only the windows with active[b] != 0 (the other rows of the result are zero)"""
        t = np.ascontiguousarray(np.atleast_2d(tokens), dtype=np.int32)
        B, n_new = t.shape
        npast = np.asarray(n_past, dtype=np.int32)
        act = np.asarray(active, dtype=np.int32)
        out = np.zeros((B, self.ctx.hp.n_vocab), dtype=np.float32)
        _check(lib().ohw_decode_active(self.h, _ip(t), n_new, _ip(npast), B, _ip(act), _fp(out)))
        return out

    def greedy(self, batch: int, p: Optional[SampleParams] = None):
        """device-resident greedy loop -> (list of token lists, sum_logprob[B])"""
        p = p or self.ctx.default_params()
        cap = self.ctx.hp.n_text_ctx
        toks = np.zeros((batch, cap), dtype=np.int32)
        nt = np.zeros(batch, dtype=np.int32)
        slp = np.zeros(batch, dtype=np.float32)
        _check(lib().ohw_greedy(self.h, C.byref(p), batch, _ip(toks), _ip(nt), cap, _fp(slp)))
        return [[int(x) for x in toks[b, :nt[b]]] for b in range(batch)], slp

    def greedy_ex(self, batch: int, p: Optional[SampleParams] = None):
        """ohw_greedy_ex -> list of dict(tokens, logprobs [n (+1 with the end-of-text token's)], ended_by_eot, no_speech_prob)"""
        p = p or self.ctx.default_params()
        cap = self.ctx.hp.n_text_ctx
        toks = np.zeros((batch, cap), dtype=np.int32)
        nt = np.zeros(batch, dtype=np.int32)
        slp = np.zeros(batch, dtype=np.float32)
        lps = np.zeros((batch, cap + 1), dtype=np.float32)
        eot = np.zeros(batch, dtype=np.int32)
        nsp = np.zeros(batch, dtype=np.float32)
        r = GreedyResult(_ip(toks), _ip(nt), _fp(slp), _fp(lps), _ip(eot), _fp(nsp))
        _check(lib().ohw_greedy_ex(self.h, C.byref(p), batch, cap, C.byref(r)))
        return [{"tokens": [int(x) for x in toks[b, :nt[b]]], "logprobs": lps[b, :nt[b] + (1 if eot[b] else 0)].copy(),
                 "ended_by_eot": bool(eot[b]), "no_speech_prob": float(nsp[b]), "sum_logprob": float(slp[b])} for b in range(batch)]

    def beam_search(self, n_windows: int, beam_size: int = 5, p: Optional[SampleParams] = None):
        """ohw_beam_search for the windows of the last encode -> [dict(tokens, sum_logprob, n_finished)]; the state needs
        max_batch >= n_windows * beam_size"""
        p = p or self.ctx.default_params()
        cap = self.ctx.hp.n_text_ctx
        toks = np.zeros((n_windows, cap), dtype=np.int32)
        nt = np.zeros(n_windows, dtype=np.int32)
        sm = np.zeros(n_windows, dtype=np.float32)
        nf = np.zeros(n_windows, dtype=np.int32)
        r = BeamResult(_ip(toks), _ip(nt), _fp(sm), _ip(nf))
        _check(lib().ohw_beam_search(self.h, C.byref(p), n_windows, beam_size, cap, C.byref(r)))
        return [{"tokens": [int(x) for x in toks[w, :nt[w]]], "sum_logprob": float(sm[w]), "n_finished": int(nf[w])} for w in range(n_windows)]

    def set_logit_bias(self, bias: Optional[np.ndarray]):
        """additive bias [n_vocab] on every logits row before the filter (None clears it)"""
        if bias is None:
            _check(lib().ohw_state_set_logit_bias(self.h, C.cast(None, C.POINTER(C.c_float)), 0))
        else:
            b = np.ascontiguousarray(bias, dtype=np.float32)
            _check(lib().ohw_state_set_logit_bias(self.h, _fp(b), b.size))

    def set_persistent(self, on: bool = True):
        """ohw_state_set_persistent: the one-launch decoder step for at most 16 single-token rows (default off: slower than the launches it replaces)"""
        _check(lib().ohw_state_set_persistent(self.h, int(on)))

    def set_batch_invariant(self, on: bool = True):
        """ohw_state_set_batch_invariant: kernel variants picked from n_new alone - a window's result no longer depends on its batch"""
        _check(lib().ohw_state_set_batch_invariant(self.h, int(bool(on))))

    def dbg_sample(self, p: SampleParams, logits: np.ndarray, histories: Sequence[Sequence[int]]):
        """the DEVICE sampler on caller-supplied rows -> (tokens [B], logprobs [B], no_speech_prob [B])"""
        lg = np.ascontiguousarray(np.atleast_2d(logits), dtype=np.float32)
        B = lg.shape[0]
        stride = max(1, max(len(h) for h in histories))
        hist = np.zeros((B, stride), dtype=np.int32)
        nh = np.zeros(B, dtype=np.int32)
        for b, h in enumerate(histories):
            hist[b, :len(h)] = h
            nh[b] = len(h)
        tok = np.zeros(B, dtype=np.int32)
        lp = np.zeros(B, dtype=np.float32)
        ns = np.zeros(B, dtype=np.float32)
        _check(lib().ohw_dbg_sample(self.h, C.byref(p), _fp(lg), _ip(hist), stride, _ip(nh), B, _ip(tok), _fp(lp), _fp(ns)))
        return tok, lp, ns

    def greedy_host_sampler(self, batch: int, p: Optional[SampleParams] = None):
        """the same loop with the HOST owning the sampler: logits cross PCIe every step"""
        p = p or self.ctx.default_params()
        ctx = self.ctx
        prompt = [ctx.tok.sot]
        if ctx.hp.n_vocab >= 51865:
            prompt += [ctx.tok.sot + 1 + p.lang_id, ctx.tok.translate if p.translate else ctx.tok.transcribe]
        if p.no_timestamps:
            prompt.append(ctx.tok.no_timestamps)
        n_max = p.force_len if p.force_len > 0 else p.n_max
        logits = self.decode(np.tile(np.asarray(prompt, np.int32), (batch, 1)), [0] * batch)
        out = [[] for _ in range(batch)]
        done = [False] * batch
        n_past = [len(prompt)] * batch
        feed = [0] * batch
        for _ in range(n_max):
            for b in range(batch):
                if done[b]:
                    continue
                tok, _ = ctx.sample_greedy_host(p, logits[b], out[b])
                if tok == ctx.tok.eot:
                    done[b] = True
                    continue
                out[b].append(tok)
                feed[b] = tok
                if len(out[b]) >= n_max or n_past[b] + 1 >= ctx.hp.n_text_ctx:
                    done[b] = True
            if all(done):
                break
            logits = self.decode(np.asarray(feed, np.int32).reshape(batch, 1), n_past)
            n_past = [n + (0 if done[b] else 1) for b, n in enumerate(n_past)]
        return out

    def profile_begin(self, kernel_class: int):
        _check(lib().ohw_state_profile_begin(self.h, kernel_class))

    def profile_end(self):
        """(launches, total_ms, work) for the class given to profile_begin"""
        n, ms, w = C.c_int64(0), C.c_double(0), C.c_double(0)
        _check(lib().ohw_state_profile_end(self.h, C.byref(n), C.byref(ms), C.byref(w)))
        return int(n.value), float(ms.value), float(w.value)

    def timings(self) -> Timings:
        t = Timings()
        _check(lib().ohw_state_timings(self.h, C.byref(t)))
        return t

    def fetch(self, what: str, batch: int) -> np.ndarray:
        hp = self.ctx.hp
        d, T = hp.n_audio_state, hp.n_audio_ctx
        shape = {"mel": (batch, hp.n_mels, CHUNK_FRAMES), "conv1": (batch, CHUNK_FRAMES, d)}.get(what, (batch, T, d))
        out = np.empty(shape, dtype=np.float32)
        _check(lib().ohw_state_fetch(self.h, what.encode(), batch, _fp(out), out.size))
        return out


# ---- mirror of the reference's engine types ------------------------------------------------------
@dataclasses.dataclass
class AudioBuffer:
    """reference src/input/audio.rs:55-61"""
    samples: np.ndarray
    sample_rate: int = 16000

    def duration_secs(self) -> float:
        return len(self.samples) / float(self.sample_rate)

    # ---- the reference's AudioBuffer DSP (src/input/audio.rs:86-239), in place, through libohw (host code) ----
    def _buf(self) -> np.ndarray:
        if not (isinstance(self.samples, np.ndarray) and self.samples.dtype == np.float32 and self.samples.flags.c_contiguous
                and self.samples.flags.writeable):
            self.samples = np.array(self.samples, dtype=np.float32, order="C")
        return self.samples

    def rms_db(self) -> float:
        b = self._buf()
        return float(lib().ohw_dsp_rms_db(_fp(b) if b.size else C.cast(None, C.POINTER(C.c_float)), b.size))

    def apply_gain(self, gain_db: float):
        b = self._buf()
        if b.size:
            lib().ohw_dsp_apply_gain(_fp(b), b.size, gain_db)

    def normalize_rms(self, target_db: float):
        b = self._buf()
        if b.size:
            lib().ohw_dsp_normalize_rms(_fp(b), b.size, target_db)

    def compress(self, threshold_db: float, ratio: float, attack_ms: float, release_ms: float, makeup_gain_db: float):
        b = self._buf()
        if b.size:
            lib().ohw_dsp_compress(_fp(b), b.size, self.sample_rate, threshold_db, ratio, attack_ms, release_ms, makeup_gain_db)

    def limit(self, ceiling_db: float, release_ms: float) -> int:
        b = self._buf()
        return int(lib().ohw_dsp_limit(_fp(b), b.size, self.sample_rate, ceiling_db, release_ms)) if b.size else 0

    def preprocess(self, config: Optional["PreprocessConfig"] = None, noise_reduction: bool = False, strength: float = 1.0,
                   denoiser: Optional["Denoiser"] = None):
        """TranscriptionWorker::preprocess_audio (reference src/queue/worker.rs:196-240): noise reduction first and
        independently of the preprocessing switch (needs a Denoiser: the network is the host's), then the chain"""
        b = self._buf()
        cfg = config or default_preprocess_config()
        if not b.size:
            return
        if noise_reduction:
            if denoiser is None:
                raise WhisperError(OHW_E_INVALID_ARG, "noise reduction is enabled but no denoise engine is plugged in")
            _check(lib().ohw_preprocess_audio_ex(_fp(b), b.size, self.sample_rate, C.byref(cfg), 1, strength, C.byref(denoiser.c)))
        else:
            _check(lib().ohw_preprocess_audio(_fp(b), b.size, self.sample_rate, C.byref(cfg)))

    def denoise(self, strength: float, denoiser: "Denoiser"):
        """AudioBuffer::denoise (reference src/input/audio.rs:249-341) with the plugged-in frame processor"""
        b = self._buf()
        if b.size:
            _check(lib().ohw_dsp_denoise(_fp(b), b.size, self.sample_rate, strength, C.byref(denoiser.c)))


class Denoiser:
    """ohw_denoise_engine around a Python callable frame(float32[480]) -> float32[480] (the stand-in for
    nnnoiseless::DenoiseState::process_frame); Denoiser() without a callable is the library's pass-through engine"""

    def __init__(self, process_frame=None, reset=None):
        self.c = DenoiseEngineC()
        self.frames = 0
        if process_frame is None:
            lib().ohw_denoise_passthrough_engine(C.byref(self.c))
            return

        def _frame(_user, out, inp):
            self.frames += 1
            res = np.asarray(process_frame(np.ctypeslib.as_array(inp, shape=(480,)).copy()), dtype=np.float32)
            np.ctypeslib.as_array(out, shape=(480,))[:] = res
            return 1.0

        def _reset(_user):
            if reset:
                reset()
        self._keep = (DENOISE_FRAME_FN(_frame), DENOISE_RESET_FN(_reset))
        self.c.user = None
        self.c.process_frame, self.c.reset = self._keep


def default_preprocess_config() -> PreprocessConfig:
    c = PreprocessConfig()
    lib().ohw_default_preprocess_config(C.byref(c))
    return c


def resample_linear(samples: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
    """resample(.., ResamplingQuality::Low) of the reference (src/input/audio.rs:960-990)"""
    x = np.ascontiguousarray(samples, dtype=np.float32)
    if x.size == 0:
        return x.copy()
    n = int(lib().ohw_dsp_resample_linear(_fp(x), x.size, from_rate, to_rate, C.cast(None, C.POINTER(C.c_float)), 0))
    out = np.empty(n, np.float32)
    if n:
        lib().ohw_dsp_resample_linear(_fp(x), x.size, from_rate, to_rate, _fp(out), n)
    return out


def resample_sinc(samples: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
    """resample(.., ResamplingQuality::High) of the reference (src/input/audio.rs:1007-1095): rubato's sinc resampler restated"""
    x = np.ascontiguousarray(samples, dtype=np.float32)
    if x.size == 0:
        return x.copy()
    n = int(lib().ohw_dsp_resample_sinc(_fp(x), x.size, from_rate, to_rate, C.cast(None, C.POINTER(C.c_float)), 0))
    out = np.empty(n, np.float32)
    if n:
        lib().ohw_dsp_resample_sinc(_fp(x), x.size, from_rate, to_rate, _fp(out), n)
    return out


class DeviceResampler:
    """ohw_resampler_*: the sinc resampler on the device (same output as resample_sinc up to fp32 summation order)"""
    def __init__(self, from_rate: int, to_rate: int, device: int = 0):
        self.h = C.c_void_p()
        _check(lib().ohw_resampler_create(device, from_rate, to_rate, C.byref(self.h)))

    def out_len(self, n: int) -> int:
        return int(lib().ohw_resampler_out_len(self.h, n))

    def run(self, samples: np.ndarray) -> np.ndarray:
        """host samples in, host samples out"""
        x = np.ascontiguousarray(samples, dtype=np.float32)
        out = np.empty(self.out_len(x.size), np.float32)
        if out.size:
            _check(lib().ohw_resampler_run(self.h, x.ctypes.data_as(C.c_void_p), x.size, 0, out.ctypes.data_as(C.c_void_p), out.size, 0, None))
        return out

    def run_device(self, in_ptr: int, n: int, out_ptr: int, out_cap: int, stream: int = 0) -> int:
        """device pointers (e.g. torch tensors' data_ptr()); asynchronous on `stream`; returns the number of output samples"""
        m = self.out_len(n)
        _check(lib().ohw_resampler_run(self.h, C.c_void_p(in_ptr), n, 1, C.c_void_p(out_ptr), out_cap, 1, C.c_void_p(stream) if stream else None))
        return m

    def close(self):
        if self.h:
            lib().ohw_resampler_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


def default_vad_config() -> VadConfig:
    c = VadConfig()
    lib().ohw_default_vad_config(C.byref(c))
    return c


class VadState:
    """reference src/vad/mod.rs:112-250"""
    def __init__(self, config: VadConfig, sample_rate: int = 16000):
        self.h = C.c_void_p(lib().ohw_vad_state_new(C.byref(config), sample_rate))

    def update(self, probability: float, is_speech: bool, chunk_samples: int):
        """the completed (start, end, avg_probability) when speech just ended, else None"""
        seg = SpeechSegment()
        rc = lib().ohw_vad_state_update(self.h, probability, int(is_speech), chunk_samples, C.byref(seg))
        return (int(seg.start), int(seg.end), float(seg.avg_probability)) if rc == 1 else None

    def is_speech(self) -> bool:
        return bool(lib().ohw_vad_state_is_speech(self.h))

    def speech_start(self):
        v = int(lib().ohw_vad_state_speech_start(self.h))
        return None if v < 0 else v

    def reset(self):
        lib().ohw_vad_state_reset(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().ohw_vad_state_free(self.h)
                self.h = None
        except Exception:
            pass


def vad_segments(samples: np.ndarray, config: VadConfig, poll_samples: int = 8000, process=None, energy_threshold_db: float = -40.0):
    """ohw_vad_run over a recording: `process(samples) -> probability` is the VadEngine hook (a Python callable here; the
    built-in energy detector when None).  Returns [(start, end, avg_probability)]."""
    x = np.ascontiguousarray(samples, dtype=np.float32)
    eng = VadEngineC()
    keep = None
    if process is None:
        _check(lib().ohw_vad_energy_engine(energy_threshold_db, C.byref(eng)))
    else:
        def _proc(user, ptr, n, out):
            out[0] = float(process(np.ctypeslib.as_array(ptr, shape=(n,)).copy()))
            return 0
        keep = VAD_PROCESS_FN(_proc)
        eng.process, eng.reset, eng.chunk_size, eng.sample_rate = keep, VAD_RESET_FN(0), 512, 16000
    cap = max(16, x.size // 1600)
    segs = (SpeechSegment * cap)()
    n = int(lib().ohw_vad_run(C.byref(eng), C.byref(config), _fp(x) if x.size else C.cast(None, C.POINTER(C.c_float)), x.size, poll_samples, segs, cap))
    if process is None:
        lib().ohw_vad_energy_engine_free(C.byref(eng))
    if n < 0:
        _check(n)
    return [(int(segs[i].start), int(segs[i].end), float(segs[i].avg_probability)) for i in range(min(n, cap))]


class EnergyVad:
    """the built-in energy detector as a VadEngine (process(samples) -> probability): ohw_vad_energy_engine.  Not Silero."""
    def __init__(self, threshold_db: float = -40.0):
        self.eng = VadEngineC()
        _check(lib().ohw_vad_energy_engine(threshold_db, C.byref(self.eng)))

    def __call__(self, samples: np.ndarray) -> float:
        x = np.ascontiguousarray(samples, dtype=np.float32)
        out = C.c_float(0.0)
        if self.eng.process(self.eng.user, _fp(x) if x.size else C.cast(None, C.POINTER(C.c_float)), x.size, C.byref(out)) != 0:
            raise WhisperError("vad process failed")
        return float(out.value)

    def close(self):
        if self.eng is not None:
            lib().ohw_vad_energy_engine_free(C.byref(self.eng))
            self.eng = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


@dataclasses.dataclass
class TranscriptionResult:
    """reference src/engine/whisper.rs:30-40"""
    text: str
    language: str
    duration_ms: int


@dataclasses.dataclass
class BenchmarkResult:
    """reference src/engine/whisper.rs:314-323"""
    overhead_secs: float
    recommended_chunk_interval: float
    test_audio_secs: float


class WhisperEngine:
    """Drop-in mirror of the reference's WhisperEngine over libohw.so."""

    def __init__(self, handle):
        self.h = handle
        self.ctx_h = C.c_void_p(lib().ohw_engine_ctx(self.h))
        self.state_h = C.c_void_p(lib().ohw_engine_state(self.h))

    @classmethod
    def new(cls, model_path: str, language: str, translate: bool, use_gpu: bool, device: int = 0,
            dtype: int = OHW_DTYPE_BF16, max_batch: int = 1) -> "WhisperEngine":
        h = C.c_void_p()
        rc = lib().ohw_engine_new(str(model_path).encode(), language.encode(), int(translate), int(use_gpu), device, dtype,
                                  max_batch, C.byref(h))
        if rc != 0:
            _raise(rc)
        return cls(h)

    @classmethod
    def from_config(cls, config: "TranscriptionConfig", data_dir: Optional[str] = None, device: int = 0, dtype: int = OHW_DTYPE_BF16,
                    max_batch: int = 1) -> "WhisperEngine":
        """reference src/engine/whisper.rs:183-201: model = effective_model() parsed as a WhisperModel (anything it does not know:
        Base), path = <data dir>/models/<filename>, use_gpu = device != "cpu" (any case), then new().  `data_dir` stands for
        Config::data_dir() (the platform's ProjectDirs path; default here: $XDG_DATA_HOME or ~/.local/share, then /openhush)."""
        path, use_gpu = config.engine_arguments(data_dir)
        return cls.new(path, config.language, config.translate, use_gpu, device, dtype, max_batch)

    def transcribe(self, audio: AudioBuffer) -> TranscriptionResult:
        s = np.ascontiguousarray(audio.samples, dtype=np.float32)
        buf = C.create_string_buffer(256)
        lang = C.create_string_buffer(8)
        ms = C.c_uint64(0)
        info = AudioInfo()
        ptr = _fp(s) if s.size else C.cast(None, C.POINTER(C.c_float))
        rc = lib().ohw_engine_transcribe(self.h, ptr, s.size, audio.sample_rate, buf, len(buf), lang, C.byref(ms), C.byref(info))
        if rc != 0:
            _raise(rc, info)
        full, n = C.c_char_p(), C.c_size_t(0)
        _check(lib().ohw_engine_last_text(self.h, C.byref(full), C.byref(n)))   # the fixed buffer may have truncated
        text = C.string_at(full, n.value).decode("utf-8", "replace") if n.value else ""
        return TranscriptionResult(text, lang.value.decode(), int(ms.value))

    def last_quality(self):
        """[(n_tokens, avg_logprob, entropy, would_fallback)] per window of the last transcribe"""
        return [(q["n_tokens"], q["avg_logprob"], q["entropy"], q["would_fallback"]) for q in self.last_quality_ex()]

    def last_quality_ex(self):
        """one dict per window of the last transcribe with every ohw_window_quality field"""
        q = C.POINTER(WindowQuality)()
        n = C.c_int(0)
        _check(lib().ohw_engine_last_quality(self.h, C.byref(q), C.byref(n)))
        out = []
        for i in range(n.value):
            d = {name: getattr(q[i], name) for name, _ in WindowQuality._fields_}
            d["would_fallback"], d["no_speech"], d["failed"] = bool(d["would_fallback"]), bool(d["no_speech"]), bool(d["failed"])
            out.append(d)
        return out

    def set_decode_policy(self, temperature_inc: Optional[float] = None, entropy_thold: Optional[float] = None,
                          logprob_thold: Optional[float] = None, no_speech_thold: Optional[float] = None):
        """ohw_engine_set_decode_policy: whisper.cpp's defaults unless overridden; temperature_inc = 0 keeps every window at T = 0"""
        pol = DecodePolicy()
        lib().ohw_default_decode_policy(C.byref(pol))
        for k, v in (("temperature_inc", temperature_inc), ("entropy_thold", entropy_thold), ("logprob_thold", logprob_thold),
                     ("no_speech_thold", no_speech_thold)):
            if v is not None:
                setattr(pol, k, v)
        _check(lib().ohw_engine_set_decode_policy(self.h, C.byref(pol)))

    def last_trace(self):
        """[(window, temperature, [sampled tokens, end-of-text included])] for every decode pass of the last transcribe"""
        p = C.POINTER(C.c_int32)()
        n = C.c_int(0)
        _check(lib().ohw_engine_last_trace(self.h, C.byref(p), C.byref(n)))
        out, i = [], 0
        while i + 3 <= n.value:
            w, t, k = p[i], p[i + 1], p[i + 2]
            out.append((int(w), t / 1000.0, [int(p[i + 3 + j]) for j in range(k)]))
            i += 3 + k
        return out

    def set_window_mode(self, mode: int):
        """OHW_WINDOW_FIXED (0, default), OHW_WINDOW_SEEK (1, whisper.cpp's timestamp-driven loop) or OHW_WINDOW_FIXED_RECORDING_MEL
        (2: fixed cuts taken from the spectrogram of the whole recording)"""
        _check(lib().ohw_engine_set_window_mode(self.h, mode))

    def set_schedule(self, schedule: int, lanes: int = 0, merge: int = 0):
        """OHW_SCHEDULE_SEQUENTIAL / _PIPELINE / _LANES (default) for audio longer than max_batch windows"""
        _check(lib().ohw_engine_set_schedule(self.h, schedule, lanes, merge))

    def last_tokens(self) -> List[int]:
        p = C.POINTER(C.c_int32)()
        n = C.c_int(0)
        lib().ohw_engine_last_tokens(self.h, C.byref(p), C.byref(n))
        return [int(p[i]) for i in range(n.value)]

    def benchmark(self, safety_margin: float) -> BenchmarkResult:
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        _check(lib().ohw_engine_benchmark(self.h, safety_margin, C.byref(a), C.byref(b), C.byref(c)))
        return BenchmarkResult(a.value, b.value, c.value)

    def close(self):
        if getattr(self, "h", None):
            lib().ohw_engine_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EnginePool:
    """ohw_pool: one engine per device of one node behind the C ABI (the reference's `[gpu] devices` intent,
    src/config.rs:921-929).  transcribe() has WhisperEngine.transcribe's contract."""

    def __init__(self, model_path: Optional[str], language: str = "auto", translate: bool = False, devices: Sequence[int] = (0,),
                 dtype: int = OHW_DTYPE_BF16, max_batch: int = 1, synthetic: Optional[Sequence[int]] = None, seed: int = 1234):
        """model_path: a ggml file; or synthetic = hparams list: procedural weights made on devices[0] (no file)"""
        ids = np.asarray(list(devices), dtype=np.int32)
        h = C.c_void_p()
        if synthetic is not None:
            hp = HParams(*[int(x) for x in synthetic])
            rc = lib().ohw_pool_create_synthetic(C.byref(hp), seed, language.encode(), int(translate), _ip(ids), len(ids), dtype, max_batch, C.byref(h))
        else:
            rc = lib().ohw_pool_create(str(model_path).encode(), language.encode(), int(translate), _ip(ids), len(ids), dtype, max_batch, C.byref(h))
        if rc != 0:
            _raise(rc)
        self.h = h

    def set_force_len(self, n_tokens: int):
        _check(lib().ohw_pool_set_force_len(self.h, n_tokens))

    def set_schedule(self, schedule: int, lanes: int = 0, merge: int = 0):
        _check(lib().ohw_pool_set_schedule(self.h, schedule, lanes, merge))

    @property
    def n_devices(self) -> int:
        return int(lib().ohw_pool_n_devices(self.h))

    @property
    def broadcast_kind(self) -> str:
        return lib().ohw_pool_broadcast_kind(self.h).decode()

    @property
    def broadcast_note(self) -> str:
        return lib().ohw_pool_broadcast_note(self.h).decode()

    def set_window_mode(self, mode: int):
        _check(lib().ohw_pool_set_window_mode(self.h, mode))

    def engine_handle(self, i: int):
        return lib().ohw_pool_engine(self.h, i)

    def set_decode_policy(self, **kw):
        pol = DecodePolicy()
        lib().ohw_default_decode_policy(C.byref(pol))
        for k, v in kw.items():
            setattr(pol, k, v)
        _check(lib().ohw_pool_set_decode_policy(self.h, C.byref(pol)))

    def transcribe(self, audio: AudioBuffer) -> TranscriptionResult:
        s = np.ascontiguousarray(audio.samples, dtype=np.float32)
        buf, lang, ms, info = C.create_string_buffer(256), C.create_string_buffer(8), C.c_uint64(0), AudioInfo()
        ptr = _fp(s) if s.size else C.cast(None, C.POINTER(C.c_float))
        rc = lib().ohw_pool_transcribe(self.h, ptr, s.size, audio.sample_rate, buf, len(buf), lang, C.byref(ms), C.byref(info))
        if rc != 0:
            _raise(rc, info)
        full, n = C.c_char_p(), C.c_size_t(0)
        _check(lib().ohw_pool_last_text(self.h, C.byref(full), C.byref(n)))
        text = C.string_at(full, n.value).decode("utf-8", "replace") if n.value else ""
        return TranscriptionResult(text, lang.value.decode(), int(ms.value))

    def last_tokens(self) -> List[int]:
        p, n = C.POINTER(C.c_int32)(), C.c_int(0)
        _check(lib().ohw_pool_last_tokens(self.h, C.byref(p), C.byref(n)))
        return [int(p[i]) for i in range(n.value)]

    def last_window_tokens(self) -> List[int]:
        q, n = C.POINTER(WindowQuality)(), C.c_int(0)
        _check(lib().ohw_pool_last_quality(self.h, C.byref(q), C.byref(n)))
        return [int(q[i].n_tokens) for i in range(n.value)]

    def close(self):
        if getattr(self, "h", None):
            lib().ohw_pool_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- model names: reference src/engine/whisper.rs:43-103 -----------------------------------------
_MODEL_FILES = {"tiny": "ggml-tiny.bin", "base": "ggml-base.bin", "small": "ggml-small.bin", "medium": "ggml-medium.bin",
                "large-v3": "ggml-large-v3.bin"}
_MODEL_ALIASES = {"large": "large-v3", "largev3": "large-v3"}


def model_filename(name: str) -> str:
    """WhisperModel::from_str + filename(): raises KeyError for names the reference rejects (e.g. 'tiny.en')"""
    key = name.lower()
    key = _MODEL_ALIASES.get(key, key)
    return _MODEL_FILES[key]


# approximate download sizes the reference shows to the user (src/engine/whisper.rs:82-92)
_MODEL_SIZES = {"tiny": 75_000_000, "base": 142_000_000, "small": 466_000_000, "medium": 1_500_000_000, "large-v3": 3_000_000_000}


def model_size_bytes(name: str) -> int:
    key = name.lower()
    return _MODEL_SIZES[_MODEL_ALIASES.get(key, key)]


_PRESET_MODEL = {"instant": "small", "balanced": "medium", "quality": "large-v3", "custom": "base"}


@dataclasses.dataclass
class TranscriptionConfig:
    """the fields of the reference's [transcription] table WhisperEngine::from_config reads (src/config.rs:662-696; defaults
    :641, :1080-1090)"""
    preset: str = "balanced"          # instant | balanced | quality | custom
    model: str = "large-v3"           # only used when preset == "custom"
    language: str = "auto"
    device: str = "cuda"
    translate: bool = False

    def effective_model(self) -> str:
        """src/config.rs:697-706 (known answers :1592-1622)"""
        return self.model if self.preset == "custom" else _PRESET_MODEL[self.preset]

    def engine_arguments(self, data_dir: Optional[str] = None):
        """(model_path, use_gpu) as from_config derives them"""
        import os
        try:
            fname = model_filename(self.effective_model())
        except KeyError:
            fname = model_filename("base")            # .parse().unwrap_or(WhisperModel::Base)
        if data_dir is None:
            data_dir = os.path.join(os.environ.get("XDG_DATA_HOME") or os.path.join(os.path.expanduser("~"), ".local", "share"), "openhush")
        return os.path.join(data_dir, "models", fname), self.device.lower() != "cpu"


def format_size(n: int) -> str:
    """reference src/engine/whisper.rs:444-458 (Rust {:.0} / {:.1} formatting: round half to even on the binary value)"""
    kb, mb, gb = 1024, 1024 ** 2, 1024 ** 3
    if n >= gb:
        return f"{n / gb:.1f} GB"
    if n >= mb:
        return f"{n / mb:.0f} MB"
    if n >= kb:
        return f"{n / kb:.0f} KB"
    return f"{n} B"
