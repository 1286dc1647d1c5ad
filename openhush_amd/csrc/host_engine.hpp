// host_engine.hpp — the engine object behind ohw_engine_* (host_engine.cpp) and the multi-device pool (pool.cpp).
#pragma once
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

struct ohw_engine {
  ohw_ctx* ctx = nullptr;
  ohw_state* state = nullptr;
  std::string language;
  bool translate = false;
  int max_batch = 1;
  int window_mode = OHW_WINDOW_FIXED;
  int force_len = 0;                     // measurement knob (ohw_engine_set_force_len): every window decodes exactly this many tokens
  std::vector<int32_t> last_tokens;
  std::string last_text;
  std::vector<ohw_window_quality> last_quality;
  // two batches in flight for audio longer than max_batch windows (include/ohw.h, ohw_stream_create): a second state
  // and three streams, made on first use; enc_cus = 0 keeps the batches strictly one after the other
  std::vector<ohw_state*> states;        // states[0] == state; the others are made on the first long input
  std::vector<void*> lane_streams;       // LANES schedule: one CU-masked stream per decode lane
  std::vector<ohw_state*> lane_states;   //   and one state of max_batch * merge windows per lane
  int lane_capacity = 0;
  void* s_full = nullptr; void* s_enc = nullptr; void* s_dec = nullptr;
  int schedule = OHW_SCHEDULE_LANES;     // how audio longer than max_batch windows is overlapped (include/ohw.h)
  int lanes = 4;                         // decodes side by side in the LANES schedule
  int merge = 3;                         // batches of max_batch windows a lane takes through ONE front-end pass and ONE decode
  int enc_cus = 96;
  int device = 0;
  ohw_decode_policy policy{0.2f, 2.4f, -1.0f, 0.6f};
  std::vector<int32_t> last_trace;   // every decode pass of the last transcribe: {window, temperature * 1000, n, tokens...}
};


namespace ohw {
// an engine around an already loaded context (takes ownership of ctx on success); throws Error
ohw_engine* engine_wrap_ctx(ohw_ctx* ctx, const std::string& language, bool translate, int max_batch, int device);
// the path after validation: windows, decode policy, text assembly (untrimmed text in *text); fills the engine's last_* records
// win_first / win_step (fixed-cut modes only): the engine takes windows win_first, win_first + win_step, ... of the recording
// (the pool's round-robin deal); the records it leaves (last_tokens / last_quality) list its own windows in that order
void engine_transcribe_core(ohw_engine* e, const float* samples, int64_t n, std::string* text, int64_t win_first = 0, int64_t win_step = 1);
}  // namespace ohw
