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
