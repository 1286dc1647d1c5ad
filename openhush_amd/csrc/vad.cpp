// vad.cpp — the step in front of the path in continuous mode: voice-activity segmentation, host code.
//
//   ohw_vad_state_*    <- VadState (reference src/vad/mod.rs:112-250): the speech / silence state machine that turns per-chunk
//                         VAD results into speech segments (min_silence_ms to end a segment, min_speech_ms to keep it)
//   ohw_vad_engine     <- trait VadEngine (reference src/vad/mod.rs:34-55): process(samples) -> probability, reset,
//                         chunk_size, sample_rate - as a C struct of function pointers, so the reference's SileroVad (an ONNX
//                         model the reference loads through the silero-vad-rust crate; neither is available offline) or any
//                         other detector plugs in from the host side
//   ohw_vad_run        <- the daemon's continuous-mode loop (reference src/daemon.rs:2062-2138): every poll the audio since
//                         the last poll goes through the engine, the result through the state machine, and a completed
//                         segment is cut from the first speech poll to the current position
//   ohw_vad_energy_engine  a built-in detector for tests and for hosts without a model: short-time RMS against a threshold.
//                         It is NOT Silero and makes no claim to its accuracy.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "common.hpp"

struct ohw_vad_state {
  ohw_vad_config cfg;
  uint32_t sample_rate;
  std::vector<float> probabilities;
  bool in_speech = false;
  bool has_start = false;
  int64_t speech_start = 0;
  int64_t silence_samples = 0;
  int64_t total_samples = 0;
};

extern "C" {

void ohw_default_vad_config(ohw_vad_config* c) {        // reference src/vad/mod.rs:76-100
  if (!c) return;
  c->enabled = 0; c->threshold = 0.5f; c->min_silence_ms = 700; c->min_speech_ms = 250; c->speech_pad_ms = 30;
}

ohw_vad_state* ohw_vad_state_new(const ohw_vad_config* cfg, uint32_t sample_rate) {
  ohw_vad_state* s = new (std::nothrow) ohw_vad_state();
  if (!s) return nullptr;
  if (cfg) s->cfg = *cfg; else ohw_default_vad_config(&s->cfg);
  s->sample_rate = sample_rate;
  return s;
}

void ohw_vad_state_free(ohw_vad_state* s) { delete s; }

// VadState::update (reference src/vad/mod.rs:158-224)
int ohw_vad_state_update(ohw_vad_state* s, float probability, int is_speech, int64_t chunk_samples, ohw_speech_segment* seg) {
  if (!s || chunk_samples < 0) return OHW_E_INVALID_ARG;
  s->probabilities.push_back(probability);
  const int64_t prev_total = s->total_samples;
  s->total_samples += chunk_samples;
  const int64_t min_silence = (int64_t)((float)s->cfg.min_silence_ms / 1000.0f * (float)s->sample_rate);
  const int64_t min_speech = (int64_t)((float)s->cfg.min_speech_ms / 1000.0f * (float)s->sample_rate);
  if (is_speech) {
    s->silence_samples = 0;
    if (!s->in_speech) {                       // speech just started
      s->in_speech = true;
      s->has_start = true;
      s->speech_start = prev_total;
    }
    return 0;
  }
  s->silence_samples += chunk_samples;
  if (s->in_speech && s->silence_samples >= min_silence) {   // speech ended
    s->in_speech = false;
    const int64_t start = s->has_start ? s->speech_start : 0;
    s->has_start = false;
    const int64_t end = prev_total;            // the segment ends where this chunk of silence starts
    if (end - start >= min_speech) {
      float avg = 0.0f;
      if (!s->probabilities.empty()) {
        float sum = 0.0f;
        for (float p : s->probabilities) sum += p;
        avg = sum / (float)s->probabilities.size();
      }
      s->probabilities.clear();
      if (seg) { seg->start = start; seg->end = end; seg->avg_probability = avg; }
      return 1;
    }
    s->probabilities.clear();                  // too short: dropped
  }
  return 0;
}

int ohw_vad_state_is_speech(const ohw_vad_state* s) { return s && s->in_speech ? 1 : 0; }
int64_t ohw_vad_state_speech_start(const ohw_vad_state* s) { return s && s->has_start ? s->speech_start : -1; }
void ohw_vad_state_reset(ohw_vad_state* s) {
  if (!s) return;
  s->probabilities.clear();
  s->in_speech = false; s->has_start = false; s->speech_start = 0; s->silence_samples = 0; s->total_samples = 0;
}

// ---- built-in energy detector (a VadEngine implementation; not Silero) -------------------------------------------------
namespace {
struct EnergyVad { float threshold_db; };
int energy_process(void* user, const float* samples, int64_t n, float* probability) {
  const EnergyVad* v = (const EnergyVad*)user;
  if (n <= 0) { *probability = 0.0f; return 0; }
  // like SileroVad::process the result is the mean over 512-sample chunks (reference src/vad/silero.rs:39-94); the score
  // of a chunk is a logistic of its level in dB around the threshold (6 dB per unit)
  double total = 0.0;
  int64_t count = 0;
  for (int64_t off = 0; off < n; off += 512) {
    const int64_t len = n - off < 512 ? n - off : 512;
    double ss = 0.0;
    for (int64_t i = 0; i < len; ++i) ss += (double)samples[off + i] * samples[off + i];
    const double rms = std::sqrt(ss / 512.0);                       // a partial chunk is zero-padded to 512, as the reference pads
    const double db = rms > 1e-10 ? 20.0 * std::log10(rms) : -200.0;
    total += 1.0 / (1.0 + std::exp(-(db - (double)v->threshold_db) / 6.0));
    ++count;
  }
  *probability = (float)(total / (double)count);
  return 0;
}
void energy_reset(void*) {}
}  // namespace

int ohw_vad_energy_engine(float threshold_db, ohw_vad_engine* out) {
  if (!out) return OHW_E_INVALID_ARG;
  EnergyVad* v = new (std::nothrow) EnergyVad{threshold_db};
  if (!v) return OHW_E_OOM;
  out->user = v; out->process = energy_process; out->reset = energy_reset; out->chunk_size = 512; out->sample_rate = 16000;
  return OHW_OK;
}
void ohw_vad_energy_engine_free(ohw_vad_engine* e) {
  if (e && e->process == energy_process) { delete (EnergyVad*)e->user; e->user = nullptr; }
}

// The daemon's continuous-mode loop over a whole recording (reference src/daemon.rs:2062-2138): polls of poll_samples; the
// audio since the last poll is one engine.process call (is_speech = probability >= threshold); the first speech poll marks
// the segment start; when the state machine completes a segment it is cut at the CURRENT position (the reference extracts
// [speech_start_pos, current_pos), i.e. the trailing min_silence stays inside the cut).  Returns the number of segments
// (all of them; at most cap are written), or a negative error code.
int64_t ohw_vad_run(const ohw_vad_engine* engine, const ohw_vad_config* cfg, const float* samples, int64_t n, int64_t poll_samples,
                    ohw_speech_segment* out, int64_t cap) {
  if (!engine || !engine->process || !cfg || (!samples && n > 0) || poll_samples <= 0) return OHW_E_INVALID_ARG;
  ohw_vad_state* st = ohw_vad_state_new(cfg, engine->sample_rate);
  if (!st) return OHW_E_OOM;
  if (engine->reset) engine->reset(engine->user);
  int64_t n_seg = 0, last = 0, speech_start_pos = -1;
  while (last < n) {
    const int64_t cur = last + poll_samples < n ? last + poll_samples : n;
    float prob = 0.0f;
    if (engine->process(engine->user, samples + last, cur - last, &prob) != 0) { ohw_vad_state_free(st); return OHW_E_TRANSCRIBE; }
    const int is_speech = prob >= cfg->threshold ? 1 : 0;
    if (is_speech && speech_start_pos < 0) speech_start_pos = last;
    ohw_speech_segment seg{};
    if (ohw_vad_state_update(st, prob, is_speech, cur - last, &seg) == 1) {
      const int64_t start = speech_start_pos >= 0 ? speech_start_pos : seg.start;
      speech_start_pos = -1;
      if (n_seg < cap && out) { out[n_seg].start = start; out[n_seg].end = cur; out[n_seg].avg_probability = seg.avg_probability; }
      ++n_seg;
    }
    last = cur;
  }
  ohw_vad_state_free(st);
  return n_seg;
}

}  // extern "C"
