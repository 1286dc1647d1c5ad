"""The reference's TranscriptionTracker unit tests (src/queue/mod.rs:318-469), restated against openhush_amd/tracker.py."""
import random

import pytest

from openhush_amd.tracker import BackpressureStrategy as BP, ChunkResult, NativeTranscriptionTracker
from openhush_amd.tracker import TranscriptionTracker as PyTranscriptionTracker


@pytest.fixture(params=["python", "native"])
def TranscriptionTracker(request):
    """the Python mirror, and the C++ tracker behind the C ABI (ohw_tracker_*): the same cases through both"""
    return PyTranscriptionTracker if request.param == "python" else NativeTranscriptionTracker


def result(seq, chunk, text, is_final):
    return ChunkResult(text, seq, chunk, is_final, 1.0)


def test_streaming_mode_outputs_immediately(TranscriptionTracker):
    t = TranscriptionTracker()
    t.add_pending(0, 0)
    t.add_pending(0, 1)
    t.add_result(result(0, 1, "world", True))
    ready = t.take_ready()
    assert len(ready) == 1 and ready[0].text == "world"
    t.add_result(result(0, 0, "hello", False))
    ready = t.take_ready()
    assert len(ready) == 1 and ready[0].text == "hello"


def test_ordered_mode_waits(TranscriptionTracker):
    t = TranscriptionTracker.new_ordered()
    t.add_pending(0, 0)
    t.add_pending(1, 0)
    t.add_result(result(1, 0, "second", True))
    assert t.take_ready() == []
    t.add_result(result(0, 0, "first", True))
    ready = t.take_ready()
    assert [r.text for r in ready] == ["first", "second"]


def test_deduplication(TranscriptionTracker):
    t = TranscriptionTracker()
    t.add_pending(0, 0)
    t.add_result(result(0, 0, "hello world this is a test", False))
    assert t.take_ready()[0].text == "hello world this is a test"
    t.add_pending(0, 1)
    t.add_result(result(0, 1, "is a test and more words", True))
    assert t.take_ready()[0].text == "and more words"
    t.reset_dedup()
    t.add_result(result(0, 2, "is a test again", True))
    assert t.take_ready()[0].text == "is a test again"


def test_empty_tracker_and_counts(TranscriptionTracker):
    t = TranscriptionTracker()
    assert t.is_empty() and t.pending_count() == 0 and t.waiting_count() == 0
    t.add_pending(0, 0)
    t.add_pending(0, 1)
    assert t.pending_count() == 2
    t.add_result(result(0, 0, "test", False))
    assert t.pending_count() == 1 and t.waiting_count() == 1
    s = t.stats()
    assert (s.pending_count, s.waiting_count) == (1, 1)


def test_backpressure_strategies(TranscriptionTracker):
    t = TranscriptionTracker()
    assert all(t.add_pending_with_config(0, c, 3, 2, BP.DROP_NEWEST) for c in range(3))
    assert not t.add_pending_with_config(0, 3, 3, 2, BP.DROP_NEWEST) and t.pending_count() == 3
    t = TranscriptionTracker()
    assert all(t.add_pending_with_config(0, c, 3, 2, BP.DROP_OLDEST) for c in range(3))
    assert t.add_pending_with_config(0, 3, 3, 2, BP.DROP_OLDEST) and t.pending_count() == 3
    assert (0, 0) not in t.pending and (0, 3) in t.pending
    t = TranscriptionTracker()
    assert all(t.add_pending_with_config(0, c, 3, 2, BP.WARN) for c in range(3))
    assert t.add_pending_with_config(0, 3, 3, 2, BP.WARN) and t.pending_count() == 4


def test_native_tracker_equals_the_python_mirror_on_random_traffic():
    """differential test: random chunks, texts with overlapping words (and multi-byte characters), results in random order"""
    rng = random.Random(7)
    words = "the quick brown fox jumps over the lazy dog und über straße naïve 東京 は 晴れ".split()
    for streaming in (True, False):
        a, b = PyTranscriptionTracker(streaming), NativeTranscriptionTracker(streaming)
        outstanding = []
        for step in range(400):
            op = rng.random()
            if op < 0.45:
                seq, chunk = rng.randrange(0, 6), rng.randrange(0, 5)
                strat = rng.choice(list(BP))
                mp = rng.choice([0, 2, 3, 10])
                ra, rb = a.add_pending_with_config(seq, chunk, mp, 2, strat), b.add_pending_with_config(seq, chunk, mp, 2, strat)
                assert ra == rb
                if ra:
                    outstanding.append((seq, chunk))
            elif op < 0.8 and outstanding:
                seq, chunk = outstanding.pop(rng.randrange(len(outstanding)))
                text = " ".join(rng.choice(words) for _ in range(rng.randrange(0, 14)))
                if rng.random() < 0.2:
                    text = "  " + text + "\u00a0 "
                fin = rng.random() < 0.3
                for t in (a, b):
                    t.add_result(ChunkResult(text, seq, chunk, fin, 1.5))
            elif op < 0.97:
                ra, rb = a.take_ready(), b.take_ready()
                assert [(r.text, r.sequence_id, r.chunk_id, r.is_final) for r in ra] == [(r.text, r.sequence_id, r.chunk_id, r.is_final) for r in rb]
            else:
                a.reset_dedup(); b.reset_dedup()
            assert (a.pending_count(), a.waiting_count(), a.is_empty()) == (b.pending_count(), b.waiting_count(), b.is_empty())
        b.close()


def test_native_tracker_and_scheduler_reject_bad_arguments():
    """the C ABI's error paths (include/ohw.h): null handles, unknown strategies, positions outside the recording"""
    import ctypes as C

    import numpy as np

    from openhush_amd import engine as E
    L = E.lib()
    fp = C.POINTER(C.c_float)
    rec = np.zeros(16000, np.float32)
    assert L.ohw_tracker_add_pending(None, 0, 0, 10, 8, 0) < 0
    t = C.c_void_p(L.ohw_tracker_new(1))
    assert L.ohw_tracker_add_pending(t, 0, 0, 10, 8, 7) < 0                       # unknown back-pressure strategy
    assert L.ohw_tracker_add_result(t, None, 0, 0, 0, 1.0) < 0
    assert L.ohw_tracker_take_ready(t) == 0 and L.ohw_tracker_ready_get(t, 0, None, None, None, None, None) < 0
    assert L.ohw_extract_chunk(rec.ctypes.data_as(fp), rec.size, 0, rec.size + 1, C.cast(None, fp), 0) < 0     # past the recording
    assert L.ohw_extract_chunk(rec.ctypes.data_as(fp), rec.size, -1, 100, C.cast(None, fp), 0) < 0
    assert L.ohw_extract_chunk(rec.ctypes.data_as(fp), rec.size, 8000, 4000, C.cast(None, fp), 0) == 0         # empty range: too short
    assert not L.ohw_chunk_scheduler_new(None, 1, 10, 8, 0) and not L.ohw_chunk_scheduler_new(t, 1, 10, 8, 9)
    s = C.c_void_p(L.ohw_chunk_scheduler_new(t, 1, 10, 8, 0))
    assert L.ohw_chunk_scheduler_tick(s, rec.ctypes.data_as(fp), rec.size, rec.size + 5, None, None) < 0
    assert L.ohw_chunk_scheduler_tick(s, rec.ctypes.data_as(fp), rec.size, 8000, None, None) == 17600           # 0.5 s, padded to 1.1 s
    assert L.ohw_chunk_scheduler_position(s) == 8000 and L.ohw_chunk_scheduler_next_id(s) == 1
    L.ohw_chunk_scheduler_free(s); L.ohw_tracker_free(t)
    L.ohw_chunk_scheduler_free(None); L.ohw_tracker_free(None)                    # freeing nothing is fine
