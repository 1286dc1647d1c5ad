// tracker.cpp — the streaming glue right after the path, host code behind the C ABI.
//
//   ohw_tracker_*          <- TranscriptionTracker (reference src/queue/mod.rs:59-297): pending / completed chunk keys
//                             (sequence_id, chunk_id); streaming mode releases every completed chunk at once, sorted by key,
//                             with the words that repeat the end of the previous output removed; ordered mode releases
//                             recordings in sequence order; three back-pressure strategies when too many chunks are pending
//   ohw_extract_chunk      <- AudioRecorder::extract_chunk for a 16 kHz recorder (reference src/input/audio.rs:737-785):
//                             nothing below 0.1 s, zero padding up to 1.1 s
//   ohw_chunk_scheduler_*  <- the chunk-timer arm of the daemon loop (reference src/daemon.rs:1958-2011): on a tick the
//                             audio since the last tick becomes a job registered with the tracker; a job the tracker refuses
//                             is skipped but the position still advances; a chunk that is too short moves nothing
// The reference's unit tests (src/queue/mod.rs:318-469) are restated in tests/test_tracker.py, which runs every case through the
// Python mirror (openhush_amd/tracker.py) AND through this file (NativeTranscriptionTracker), plus a randomised differential test.
#include <cstring>
#include <map>
#include <new>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

namespace {
using Key = std::pair<uint64_t, uint32_t>;
struct Result {
  std::string text;
  uint64_t sequence_id = 0;
  uint32_t chunk_id = 0;
  int is_final = 0;
  float duration_secs = 0.f;
};

// Unicode White_Space code points (what the reference's split_whitespace splits on); returns the byte length of the
// white-space character at s[i], or 0
size_t ws_len(const std::string& s, size_t i) {
  const unsigned char c = (unsigned char)s[i];
  if (c == 0x20 || (c >= 0x09 && c <= 0x0D)) return 1;
  if (c == 0xC2 && i + 1 < s.size()) {
    const unsigned char d = (unsigned char)s[i + 1];
    return (d == 0x85 || d == 0xA0) ? 2 : 0;
  }
  if (c == 0xE1 && i + 2 < s.size()) return ((unsigned char)s[i + 1] == 0x9A && (unsigned char)s[i + 2] == 0x80) ? 3 : 0;   // U+1680
  if (c == 0xE2 && i + 2 < s.size()) {
    const unsigned char d = (unsigned char)s[i + 1], e = (unsigned char)s[i + 2];
    if (d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) return 3;                            // U+2000-200A, 2028, 2029, 202F
    if (d == 0x81 && e == 0x9F) return 3;                                                                                       // U+205F
    return 0;
  }
  if (c == 0xE3 && i + 2 < s.size()) return ((unsigned char)s[i + 1] == 0x80 && (unsigned char)s[i + 2] == 0x80) ? 3 : 0;   // U+3000
  return 0;
}
std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  std::string cur;
  for (size_t i = 0; i < s.size();) {
    const size_t w = ws_len(s, i);
    if (w) {
      if (!cur.empty()) { out.push_back(cur); cur.clear(); }
      i += w;
    } else {
      cur.push_back(s[i]);
      ++i;
    }
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}
std::string join_from(const std::vector<std::string>& w, size_t first, size_t last) {
  std::string r;
  for (size_t i = first; i < last; ++i) { if (i > first) r.push_back(' '); r += w[i]; }
  return r;
}
}  // namespace

struct ohw_tracker {
  std::set<Key> pending;
  std::map<Key, Result> completed;
  uint64_t next_output_id = 0;
  bool streaming = true;
  std::string last_text_suffix;
  std::vector<Result> ready;           // what the last take_ready released

  // the words at the start of `text` that repeat the end of the previous output go (longest of the first <= 10 prefixes
  // that occurs in the kept suffix)
  std::string deduplicate(const std::string& text) const {
    const std::vector<std::string> words = split_ws(text);
    if (words.empty()) return text;
    size_t skip = 0;
    for (size_t i = 1; i <= words.size() && i <= 10; ++i)
      if (last_text_suffix.find(join_from(words, 0, i)) != std::string::npos) skip = i;
    return skip > 0 ? join_from(words, skip, words.size()) : text;
  }
};

struct ohw_chunk_scheduler {
  ohw_tracker* tracker;
  uint64_t sequence_id;
  int max_pending, high_water_mark, strategy;
  int64_t last_chunk_pos = 0;
  uint32_t next_chunk_id = 0;
  int64_t rejected = 0;
};

extern "C" {

ohw_tracker* ohw_tracker_new(int streaming) {
  ohw_tracker* t = new (std::nothrow) ohw_tracker();
  if (t) t->streaming = streaming != 0;
  return t;
}
void ohw_tracker_free(ohw_tracker* t) { delete t; }

int ohw_tracker_add_pending(ohw_tracker* t, uint64_t sequence_id, uint32_t chunk_id, uint32_t max_pending, uint32_t high_water_mark, int strategy) {
  if (!t || strategy < OHW_BACKPRESSURE_WARN || strategy > OHW_BACKPRESSURE_DROP_NEWEST) return OHW_E_INVALID_ARG;
  (void)high_water_mark;                     // the reference only logs a warning at the high-water mark
  if (max_pending > 0 && t->pending.size() >= (size_t)max_pending) {
    if (strategy == OHW_BACKPRESSURE_DROP_OLDEST) {
      if (!t->pending.empty()) t->pending.erase(t->pending.begin());      // the smallest (sequence, chunk) key
    } else if (strategy == OHW_BACKPRESSURE_DROP_NEWEST) {
      return 0;
    }                                                                      // WARN: accepted anyway
  }
  t->pending.insert(Key(sequence_id, chunk_id));
  return 1;
}

int ohw_tracker_add_result(ohw_tracker* t, const char* text, uint64_t sequence_id, uint32_t chunk_id, int is_final, float duration_secs) {
  if (!t || !text) return OHW_E_INVALID_ARG;
  const Key k(sequence_id, chunk_id);
  t->pending.erase(k);
  Result r;
  r.text = text; r.sequence_id = sequence_id; r.chunk_id = chunk_id; r.is_final = is_final != 0; r.duration_secs = duration_secs;
  t->completed[k] = std::move(r);
  return OHW_OK;
}

int ohw_tracker_take_ready(ohw_tracker* t) {
  if (!t) return OHW_E_INVALID_ARG;
  t->ready.clear();
  if (t->streaming) {
    for (auto& kv : t->completed) t->ready.push_back(std::move(kv.second));   // a map: already in key order
    t->completed.clear();
    for (Result& r : t->ready) {
      if (!t->last_text_suffix.empty() && !r.text.empty()) r.text = t->deduplicate(r.text);
      if (r.text.size() > 10) {
        size_t start = r.text.size() > 50 ? r.text.size() - 50 : 0;
        while (start < r.text.size() && ((unsigned char)r.text[start] & 0xC0) == 0x80) ++start;   // stay on a character boundary
        t->last_text_suffix = r.text.substr(start);
      }
    }
  } else {
    for (;;) {
      auto it = t->completed.find(Key(t->next_output_id, 0));
      if (it == t->completed.end()) break;
      t->ready.push_back(std::move(it->second));
      t->completed.erase(it);
      ++t->next_output_id;
    }
  }
  return (int)t->ready.size();
}

int ohw_tracker_ready_get(const ohw_tracker* t, int i, const char** text, uint64_t* sequence_id, uint32_t* chunk_id, int* is_final, float* duration_secs) {
  if (!t || i < 0 || (size_t)i >= t->ready.size()) return OHW_E_INVALID_ARG;
  const Result& r = t->ready[(size_t)i];
  if (text) *text = r.text.c_str();
  if (sequence_id) *sequence_id = r.sequence_id;
  if (chunk_id) *chunk_id = r.chunk_id;
  if (is_final) *is_final = r.is_final;
  if (duration_secs) *duration_secs = r.duration_secs;
  return OHW_OK;
}

void ohw_tracker_reset_dedup(ohw_tracker* t) { if (t) t->last_text_suffix.clear(); }
int ohw_tracker_is_empty(const ohw_tracker* t) { return !t || (t->pending.empty() && t->completed.empty()) ? 1 : 0; }
int ohw_tracker_is_pending(const ohw_tracker* t, uint64_t sequence_id, uint32_t chunk_id) { return t && t->pending.count(Key(sequence_id, chunk_id)) ? 1 : 0; }
int ohw_tracker_pending_count(const ohw_tracker* t) { return t ? (int)t->pending.size() : 0; }
int ohw_tracker_waiting_count(const ohw_tracker* t) { return t ? (int)t->completed.size() : 0; }

// reference src/input/audio.rs:737-785 with device rate = 16 kHz: returns the chunk's length (0: too short, nothing written;
// negative: error); with out == NULL or out_cap too small the length only
int64_t ohw_extract_chunk(const float* recording, int64_t n_recording, int64_t from_pos, int64_t to_pos, float* out, int64_t out_cap) {
  if (!recording || from_pos < 0 || to_pos > n_recording) return OHW_E_INVALID_ARG;
  const int64_t n = to_pos > from_pos ? to_pos - from_pos : 0;
  const float duration = (float)n / 16000.0f;
  if (n == 0 || duration < 0.1f) return 0;
  const int64_t need = (int64_t)(16000.0f * 1.1f);
  const int64_t len = duration < 1.1f ? need : n;
  if (!out || out_cap < len) return len;
  std::memcpy(out, recording + from_pos, (size_t)n * sizeof(float));
  if (len > n) std::memset(out + n, 0, (size_t)(len - n) * sizeof(float));
  return len;
}

ohw_chunk_scheduler* ohw_chunk_scheduler_new(ohw_tracker* tracker, uint64_t sequence_id, uint32_t max_pending, uint32_t high_water_mark, int strategy) {
  if (!tracker || strategy < OHW_BACKPRESSURE_WARN || strategy > OHW_BACKPRESSURE_DROP_NEWEST) return nullptr;
  ohw_chunk_scheduler* s = new (std::nothrow) ohw_chunk_scheduler();
  if (!s) return nullptr;
  s->tracker = tracker; s->sequence_id = sequence_id; s->max_pending = (int)max_pending; s->high_water_mark = (int)high_water_mark; s->strategy = strategy;
  return s;
}
void ohw_chunk_scheduler_free(ohw_chunk_scheduler* s) { delete s; }

// one tick of the chunk timer at recorder position current_pos.  Returns the job's length in samples and fills
// *chunk_id (the samples are [from, current_pos) of the recording, padded as ohw_extract_chunk pads: fetch them with it
// using *from_pos); 0: nothing to do (too short: nothing moved; refused by the tracker: position and id moved on).
int64_t ohw_chunk_scheduler_tick(ohw_chunk_scheduler* s, const float* recording, int64_t n_recording, int64_t current_pos, uint32_t* chunk_id, int64_t* from_pos) {
  if (!s || !recording || current_pos > n_recording) return OHW_E_INVALID_ARG;
  const int64_t len = ohw_extract_chunk(recording, n_recording, s->last_chunk_pos, current_pos, nullptr, 0);
  if (len <= 0) return len;
  const int accepted = ohw_tracker_add_pending(s->tracker, s->sequence_id, s->next_chunk_id, (uint32_t)s->max_pending, (uint32_t)s->high_water_mark, s->strategy);
  if (accepted < 0) return accepted;
  if (chunk_id) *chunk_id = s->next_chunk_id;
  if (from_pos) *from_pos = s->last_chunk_pos;
  s->last_chunk_pos = current_pos;
  s->next_chunk_id += 1;
  if (!accepted) { s->rejected += 1; return 0; }
  return len;
}
int64_t ohw_chunk_scheduler_position(const ohw_chunk_scheduler* s) { return s ? s->last_chunk_pos : 0; }
uint32_t ohw_chunk_scheduler_next_id(const ohw_chunk_scheduler* s) { return s ? s->next_chunk_id : 0; }
int64_t ohw_chunk_scheduler_rejected(const ohw_chunk_scheduler* s) { return s ? s->rejected : 0; }

}  // extern "C"
