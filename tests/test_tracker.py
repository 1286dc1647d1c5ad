"""The reference's TranscriptionTracker unit tests (src/queue/mod.rs:318-469), restated against openhush_amd/tracker.py."""
from openhush_amd.tracker import BackpressureStrategy as BP, ChunkResult, TranscriptionTracker


def result(seq, chunk, text, is_final):
    return ChunkResult(text, seq, chunk, is_final, 1.0)


def test_streaming_mode_outputs_immediately():
    t = TranscriptionTracker()
    t.add_pending(0, 0)
    t.add_pending(0, 1)
    t.add_result(result(0, 1, "world", True))
    ready = t.take_ready()
    assert len(ready) == 1 and ready[0].text == "world"
    t.add_result(result(0, 0, "hello", False))
    ready = t.take_ready()
    assert len(ready) == 1 and ready[0].text == "hello"


def test_ordered_mode_waits():
    t = TranscriptionTracker.new_ordered()
    t.add_pending(0, 0)
    t.add_pending(1, 0)
    t.add_result(result(1, 0, "second", True))
    assert t.take_ready() == []
    t.add_result(result(0, 0, "first", True))
    ready = t.take_ready()
    assert [r.text for r in ready] == ["first", "second"]


def test_deduplication():
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


def test_empty_tracker_and_counts():
    t = TranscriptionTracker()
    assert t.is_empty() and t.pending_count() == 0 and t.waiting_count() == 0
    t.add_pending(0, 0)
    t.add_pending(0, 1)
    assert t.pending_count() == 2
    t.add_result(result(0, 0, "test", False))
    assert t.pending_count() == 1 and t.waiting_count() == 1
    s = t.stats()
    assert (s.pending_count, s.waiting_count) == (1, 1)


def test_backpressure_strategies():
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
