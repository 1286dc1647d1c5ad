"""Streaming glue right after the path: the reference's TranscriptionTracker, host side.

Mirror of reference src/queue/mod.rs:59-297 - same method names, argument meaning and results: pending / completed chunk
keys (sequence_id, chunk_id), streaming mode (every completed chunk is released at once, sorted by key, with the words that
overlap the previous output removed) or ordered mode (released in sequence order), and the three back-pressure strategies.
Pinned by the reference's own unit tests (src/queue/mod.rs:318-469), restated in tests/test_tracker.py.
"""
from __future__ import annotations

import dataclasses
import enum
from typing import Dict, List, Set, Tuple


class BackpressureStrategy(enum.Enum):
    WARN = "warn"
    DROP_OLDEST = "drop_oldest"
    DROP_NEWEST = "drop_newest"


@dataclasses.dataclass
class ChunkResult:
    """reference src/queue/mod.rs:30-43 (its TranscriptionResult of the queue, not the engine's)"""
    text: str
    sequence_id: int
    chunk_id: int
    is_final: bool = False
    duration_secs: float = 1.0


@dataclasses.dataclass
class QueueStats:
    pending_count: int
    waiting_count: int


class TranscriptionTracker:
    def __init__(self, streaming: bool = True):          # new() = streaming, new_ordered() = not (:72-88)
        self.pending: Set[Tuple[int, int]] = set()
        self.completed: Dict[Tuple[int, int], ChunkResult] = {}
        self.next_output_id = 0
        self.streaming = streaming
        self.last_text_suffix = ""

    @classmethod
    def new_ordered(cls) -> "TranscriptionTracker":
        return cls(streaming=False)

    def add_pending(self, sequence_id: int, chunk_id: int) -> bool:                      # :94-96
        return self.add_pending_with_config(sequence_id, chunk_id, 10, 8, BackpressureStrategy.WARN)

    def add_pending_with_config(self, sequence_id: int, chunk_id: int, max_pending: int, high_water_mark: int,
                                strategy: BackpressureStrategy) -> bool:                 # :111-176
        n = len(self.pending)
        if max_pending > 0 and n >= max_pending:
            if strategy is BackpressureStrategy.DROP_OLDEST:
                if self.pending:
                    self.pending.remove(min(self.pending))
            elif strategy is BackpressureStrategy.DROP_NEWEST:
                return False
            # WARN: accept anyway
        self.pending.add((sequence_id, chunk_id))
        return True

    def stats(self) -> QueueStats:
        return QueueStats(len(self.pending), len(self.completed))

    def add_result(self, result: ChunkResult):                                           # :188-200
        key = (result.sequence_id, result.chunk_id)
        self.pending.discard(key)
        self.completed[key] = result

    def take_ready(self) -> List[ChunkResult]:                                           # :206-212
        return self._take_ready_streaming() if self.streaming else self._take_ready_ordered()

    def _take_ready_streaming(self) -> List[ChunkResult]:                                # :215-234
        ready = sorted(self.completed.values(), key=lambda r: (r.sequence_id, r.chunk_id))
        self.completed = {}
        for r in ready:
            if self.last_text_suffix and r.text:
                r.text = self._deduplicate_text(r.text)
            if len(r.text.encode("utf-8")) > 10:          # Rust String::len() counts bytes
                b = r.text.encode("utf-8")
                start = max(0, len(b) - 50)
                while start < len(b) and (b[start] & 0xC0) == 0x80:   # the Rust slice would panic inside a code point; stay on a boundary
                    start += 1
                self.last_text_suffix = b[start:].decode("utf-8")
        return ready

    def _take_ready_ordered(self) -> List[ChunkResult]:                                  # :237-248
        ready = []
        while (self.next_output_id, 0) in self.completed:
            ready.append(self.completed.pop((self.next_output_id, 0)))
            self.next_output_id += 1
        return ready

    def _deduplicate_text(self, text: str) -> str:                                       # :251-276
        words = text.split()
        if not words:
            return text
        skip = 0
        for i in range(1, min(len(words), 10) + 1):
            if " ".join(words[:i]) in self.last_text_suffix:
                skip = i
        return " ".join(words[skip:]) if skip > 0 else text

    def reset_dedup(self):
        self.last_text_suffix = ""

    def is_empty(self) -> bool:
        return not self.pending and not self.completed

    def pending_count(self) -> int:
        return len(self.pending)

    def waiting_count(self) -> int:
        return len(self.completed)


_STRATEGY_CODE = {BackpressureStrategy.WARN: 0, BackpressureStrategy.DROP_OLDEST: 1, BackpressureStrategy.DROP_NEWEST: 2}


class NativeTranscriptionTracker:
    """The same tracker behind the C ABI (ohw_tracker_*, openhush_amd/csrc/tracker.cpp): what a C or Rust host links.  Same
    methods and results as TranscriptionTracker above; tests/test_tracker.py runs the reference's cases through both."""
    def __init__(self, streaming: bool = True):
        import ctypes as C
        from . import engine as E
        self._C, self._L = C, E.lib()
        self.h = C.c_void_p(self._L.ohw_tracker_new(int(streaming)))
        if not self.h:
            raise MemoryError("ohw_tracker_new")

    @classmethod
    def new_ordered(cls) -> "NativeTranscriptionTracker":
        return cls(streaming=False)

    def add_pending(self, sequence_id: int, chunk_id: int) -> bool:
        return self.add_pending_with_config(sequence_id, chunk_id, 10, 8, BackpressureStrategy.WARN)

    def add_pending_with_config(self, sequence_id: int, chunk_id: int, max_pending: int, high_water_mark: int,
                                strategy: BackpressureStrategy) -> bool:
        rc = self._L.ohw_tracker_add_pending(self.h, sequence_id, chunk_id, max_pending, high_water_mark, _STRATEGY_CODE[strategy])
        if rc < 0:
            raise ValueError("ohw_tracker_add_pending")
        return rc == 1

    def add_result(self, result: ChunkResult):
        self._L.ohw_tracker_add_result(self.h, result.text.encode("utf-8"), result.sequence_id, result.chunk_id, int(result.is_final),
                                       float(result.duration_secs))

    def take_ready(self) -> List[ChunkResult]:
        C = self._C
        n = self._L.ohw_tracker_take_ready(self.h)
        out = []
        for i in range(n):
            text, seq, chunk, fin, dur = C.c_char_p(), C.c_uint64(), C.c_uint32(), C.c_int(), C.c_float()
            self._L.ohw_tracker_ready_get(self.h, i, C.byref(text), C.byref(seq), C.byref(chunk), C.byref(fin), C.byref(dur))
            out.append(ChunkResult(text.value.decode("utf-8"), int(seq.value), int(chunk.value), bool(fin.value), float(dur.value)))
        return out

    class _Pending:
        def __init__(self, owner):
            self.o = owner

        def __contains__(self, key) -> bool:
            return bool(self.o._L.ohw_tracker_is_pending(self.o.h, key[0], key[1]))

    @property
    def pending(self):
        """supports `(sequence_id, chunk_id) in tracker.pending`"""
        return NativeTranscriptionTracker._Pending(self)

    def reset_dedup(self):
        self._L.ohw_tracker_reset_dedup(self.h)

    def is_empty(self) -> bool:
        return bool(self._L.ohw_tracker_is_empty(self.h))

    def pending_count(self) -> int:
        return int(self._L.ohw_tracker_pending_count(self.h))

    def waiting_count(self) -> int:
        return int(self._L.ohw_tracker_waiting_count(self.h))

    def stats(self) -> QueueStats:
        return QueueStats(self.pending_count(), self.waiting_count())

    def close(self):
        if self.h:
            self._L.ohw_tracker_free(self.h)
            self.h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass
