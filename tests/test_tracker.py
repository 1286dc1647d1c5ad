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
