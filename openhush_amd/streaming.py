"""Streaming glue around the path (BASELINE.json config #5): the daemon's chunk scheduler in front of the
engine and the TranscriptionTracker behind it, host side.

    ChunkScheduler     <- the chunk-timer arm of the daemon loop (reference src/daemon.rs:1912-2013) + AudioRecorder::
                          extract_chunk (src/input/audio.rs:741-776): on every tick the audio since the last tick becomes a job
                          (dropped below 0.1 s, padded to 1.1 s), registered with the tracker under its back-pressure settings
    StreamingSession   ties it to the MI355X path: an optional VAD gate (reference VadEngine hook: chunks without speech are
                          not decoded), one 30 s window per chunk through mel -> encoder -> beam search (beam = 5, the
                          hipGraph-captured decoder step) or the greedy loop, detokenise, tracker.add_result ->
                          take_ready (overlap de-duplication and ordering are the tracker's)
Nothing here computes: the arithmetic is in libohw.so.
"""
from __future__ import annotations

import dataclasses
from typing import Callable, List, Optional

import numpy as np

from . import engine as E
from .tracker import BackpressureStrategy, ChunkResult, TranscriptionTracker

SAMPLE_RATE = 16000
MIN_DURATION_SECS = 0.1            # reference src/input/audio.rs:29
WHISPER_MIN_DURATION_SECS = 1.1    # reference src/input/audio.rs:34


@dataclasses.dataclass
class ChunkJob:
    """reference src/queue/mod.rs TranscriptionJob"""
    samples: np.ndarray
    sequence_id: int
    chunk_id: int
    is_final: bool = False


def extract_chunk(recording: np.ndarray, from_pos: int, to_pos: int) -> Optional[np.ndarray]:
    """AudioRecorder::extract_chunk for a 16 kHz recorder: None below 0.1 s, zero-padded to 1.1 s (17 600 samples)"""
    s = np.asarray(recording[from_pos:to_pos], dtype=np.float32)
    if s.size == 0 or s.size / SAMPLE_RATE < MIN_DURATION_SECS:
        return None
    need = int(SAMPLE_RATE * WHISPER_MIN_DURATION_SECS)
    if s.size / SAMPLE_RATE < WHISPER_MIN_DURATION_SECS:
        s = np.concatenate([s, np.zeros(need - s.size, np.float32)])
    return s


class ChunkScheduler:
    """One recording (sequence): tick(current_pos) returns the job for the audio since the last tick, or None (too short, or
    refused by the tracker's back-pressure - the position still advances, as in the reference)."""

    def __init__(self, tracker: TranscriptionTracker, sequence_id: int, max_pending: int = 10, high_water_mark: int = 8,
                 strategy: BackpressureStrategy = BackpressureStrategy.WARN):
        self.tracker, self.sequence_id = tracker, sequence_id
        self.max_pending, self.high_water_mark, self.strategy = max_pending, high_water_mark, strategy
        self.last_chunk_pos = 0
        self.next_chunk_id = 0
        self.rejected = 0

    def tick(self, recording: np.ndarray, current_pos: int, is_final: bool = False) -> Optional[ChunkJob]:
        buf = extract_chunk(recording, self.last_chunk_pos, current_pos)
        if buf is None:
            return None                                   # "Chunk too short, skipping": the position is kept
        accepted = self.tracker.add_pending_with_config(self.sequence_id, self.next_chunk_id, self.max_pending, self.high_water_mark,
                                                        self.strategy)
        job = ChunkJob(buf, self.sequence_id, self.next_chunk_id, is_final) if accepted else None
        if not accepted:
            self.rejected += 1
        self.last_chunk_pos = current_pos
        self.next_chunk_id += 1
        return job


class NativeChunkScheduler:
    """The scheduler behind the C ABI (ohw_chunk_scheduler_*, csrc/tracker.cpp) on a NativeTranscriptionTracker; same
    interface and behaviour as ChunkScheduler."""
    def __init__(self, tracker, sequence_id: int, max_pending: int = 10, high_water_mark: int = 8,
                 strategy: BackpressureStrategy = BackpressureStrategy.WARN):
        import ctypes as C
        from .tracker import _STRATEGY_CODE
        self._C, self._L = C, E.lib()
        self.tracker, self.sequence_id = tracker, sequence_id
        self.h = C.c_void_p(self._L.ohw_chunk_scheduler_new(tracker.h, sequence_id, max_pending, high_water_mark, _STRATEGY_CODE[strategy]))
        if not self.h:
            raise ValueError("ohw_chunk_scheduler_new")

    last_chunk_pos = property(lambda self: int(self._L.ohw_chunk_scheduler_position(self.h)))
    next_chunk_id = property(lambda self: int(self._L.ohw_chunk_scheduler_next_id(self.h)))
    rejected = property(lambda self: int(self._L.ohw_chunk_scheduler_rejected(self.h)))

    def tick(self, recording: np.ndarray, current_pos: int, is_final: bool = False) -> Optional[ChunkJob]:
        C = self._C
        rec = np.ascontiguousarray(recording, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        cid, frm = C.c_uint32(), C.c_int64()
        n = int(self._L.ohw_chunk_scheduler_tick(self.h, rec.ctypes.data_as(fp), rec.size, current_pos, C.byref(cid), C.byref(frm)))
        if n < 0:
            raise ValueError("ohw_chunk_scheduler_tick")
        if n == 0:
            return None
        buf = np.empty(n, np.float32)
        self._L.ohw_extract_chunk(rec.ctypes.data_as(fp), rec.size, frm.value, current_pos, buf.ctypes.data_as(fp), n)
        return ChunkJob(buf, self.sequence_id, int(cid.value), is_final)

    def close(self):
        if self.h:
            self._L.ohw_chunk_scheduler_free(self.h)
            self.h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


def native_extract_chunk(recording: np.ndarray, from_pos: int, to_pos: int) -> Optional[np.ndarray]:
    """ohw_extract_chunk"""
    import ctypes as C
    rec = np.ascontiguousarray(recording, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    n = int(E.lib().ohw_extract_chunk(rec.ctypes.data_as(fp), rec.size, from_pos, to_pos, C.cast(None, fp), 0))
    if n <= 0:
        return None
    out = np.empty(n, np.float32)
    E.lib().ohw_extract_chunk(rec.ctypes.data_as(fp), rec.size, from_pos, to_pos, out.ctypes.data_as(fp), n)
    return out


class StreamingSession:
    """Chunks of one recording through the MI355X path, in the worker's stage order (reference src/queue/worker.rs:119-193):
    noise reduction (the RNNoise stage of BASELINE config #5: a plugged-in `denoiser`, reference src/input/audio.rs:249-341) ->
    normalise / compress / limit when `audio_config.preprocessing` is on (:196-240) -> VAD gate -> every 30 s window of the
    job through mel -> encoder -> beam search or greedy -> text.  beam_size 0 = greedy; vad: callable samples -> probability
    (the VadEngine hook) with `vad_threshold`, or None."""

    def __init__(self, ctx: E.Context, beam_size: int = 5, vad: Optional[Callable[[np.ndarray], float]] = None, vad_threshold: float = 0.5,
                 sequence_id: int = 1, tracker: Optional[TranscriptionTracker] = None, params: Optional[E.SampleParams] = None,
                 audio_config: Optional[E.PreprocessConfig] = None, noise_reduction: bool = False, noise_reduction_strength: float = 1.0,
                 denoiser: Optional["E.Denoiser"] = None):
        self.ctx = ctx
        self.beam_size = beam_size
        self.state = E.State(ctx, max(1, beam_size))          # one window at a time: beam_size decoder rows
        self.vad, self.vad_threshold = vad, vad_threshold
        self.tracker = tracker or TranscriptionTracker()
        self.scheduler = ChunkScheduler(self.tracker, sequence_id)
        self.params = params or ctx.default_params()
        self.audio_config = audio_config
        self.noise_reduction, self.noise_reduction_strength, self.denoiser = noise_reduction, noise_reduction_strength, denoiser
        if noise_reduction and denoiser is None:
            raise E.WhisperError(E.OHW_E_INVALID_ARG, "noise reduction is enabled but no denoise engine is plugged in")
        self.skipped_silent = 0
        self.windows_decoded = 0

    def _text(self, tokens: List[int]) -> str:
        return b"".join(self.ctx.token_text(t) for t in tokens if t < self.ctx.tok.eot).decode("utf-8", "replace").strip()

    def preprocess(self, samples: np.ndarray) -> np.ndarray:
        """TranscriptionWorker::preprocess_audio on a copy of the job's buffer"""
        if not self.noise_reduction and (self.audio_config is None or not self.audio_config.preprocessing):
            return samples
        buf = E.AudioBuffer(np.array(samples, dtype=np.float32, order="C"), SAMPLE_RATE)
        buf.preprocess(self.audio_config, self.noise_reduction, self.noise_reduction_strength, self.denoiser)
        return buf.samples

    def transcribe_job(self, job: ChunkJob) -> ChunkResult:
        s_all = self.preprocess(job.samples)
        if self.vad is not None and float(self.vad(s_all)) < self.vad_threshold:
            self.skipped_silent += 1
            text = ""
        else:
            # the worker hands the WHOLE buffer to engine.transcribe (worker.rs:152): a job longer than 30 s (a late timer
            # tick, a long VAD segment in continuous mode) is cut at the 30 s marks like any other input, window after window
            parts = []
            for off in range(0, len(s_all), E.CHUNK_SAMPLES):
                s = s_all[off:off + E.CHUNK_SAMPLES]
                self.state.mel(s[None, :], [len(s)], E.OHW_MEL_ZERO_TAIL, want=False)
                self.state.encode(1)
                if self.beam_size >= 2:
                    toks = self.state.beam_search(1, self.beam_size, self.params)[0]["tokens"]
                else:
                    toks = self.state.greedy(1, self.params)[0][0]
                parts.append(b"".join(self.ctx.token_text(t) for t in toks if t < self.ctx.tok.eot))
                self.windows_decoded += 1
            text = b"".join(parts).decode("utf-8", "replace").strip()
        return ChunkResult(text, job.sequence_id, job.chunk_id, job.is_final, len(job.samples) / SAMPLE_RATE)

    def tick(self, recording: np.ndarray, current_pos: int, is_final: bool = False) -> List[ChunkResult]:
        """the daemon's chunk-timer arm and the worker in one call: schedule, transcribe, hand to the tracker, return what the
        tracker releases"""
        job = self.scheduler.tick(recording, current_pos, is_final)
        if job is None:
            return []
        self.tracker.add_result(self.transcribe_job(job))
        return self.tracker.take_ready()
