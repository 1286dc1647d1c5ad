// dsp.cpp — the audio preprocessing that runs right before the hot path, host side:
//   ohw_dsp_rms_db          <- AudioBuffer::rms_db          (reference src/input/audio.rs:86-102)
//   ohw_dsp_apply_gain      <- AudioBuffer::apply_gain      (:123-129)
//   ohw_dsp_normalize_rms   <- AudioBuffer::normalize_rms   (:108-120)
//   ohw_dsp_compress        <- AudioBuffer::compress        (:139-191)
//   ohw_dsp_limit           <- AudioBuffer::limit           (:197-239)
//   ohw_dsp_resample_linear <- resample_linear              (:972-990)
//   ohw_preprocess_audio    <- TranscriptionWorker::preprocess_audio (src/queue/worker.rs:196-240), without RNNoise
// The envelope follower and the limiter are first-order recurrences with a data-dependent branch per sample over the
// WHOLE recording: one sequential chain, nothing for a GPU to parallelise - they stay on the host (about 3 ns per sample).
// Arithmetic is in fp32 in the reference's operation order.  Not built: the rubato sinc resampler and RNNoise
// (third-party crates, sources and model not under /root/reference).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/ohw.h"

extern "C" {

float ohw_dsp_rms_db(const float* s, int64_t n) {
  if (!s || n <= 0) return -INFINITY;
  float sum_squares = 0.0f;                       // f32 accumulation in sample order, like the reference's iterator sum
  for (int64_t i = 0; i < n; ++i) sum_squares += s[i] * s[i];
  const float rms = std::sqrt(sum_squares / (float)n);
  return rms > 0.0f ? 20.0f * std::log10(rms) : -INFINITY;
}

void ohw_dsp_apply_gain(float* s, int64_t n, float gain_db) {
  if (!s) return;
  const float g = std::pow(10.0f, gain_db / 20.0f);
  for (int64_t i = 0; i < n; ++i) s[i] *= g;
}

void ohw_dsp_normalize_rms(float* s, int64_t n, float target_db) {
  const float cur = ohw_dsp_rms_db(s, n);
  if (std::isfinite(cur)) ohw_dsp_apply_gain(s, n, target_db - cur);   // silent audio is left alone
}

void ohw_dsp_compress(float* s, int64_t n, uint32_t sample_rate, float threshold_db, float ratio, float attack_ms, float release_ms,
                      float makeup_gain_db) {
  if (!s || n <= 0 || ratio <= 1.0f) return;
  const float threshold = std::pow(10.0f, threshold_db / 20.0f);
  const float attack = std::exp(-1.0f / (attack_ms * (float)sample_rate / 1000.0f));
  const float release = std::exp(-1.0f / (release_ms * (float)sample_rate / 1000.0f));
  float envelope = 0.0f;
  for (int64_t i = 0; i < n; ++i) {
    const float a = std::fabs(s[i]);
    if (a > envelope) envelope = attack * envelope + (1.0f - attack) * a;
    else envelope = release * envelope + (1.0f - release) * a;
    float gain = 1.0f;
    if (envelope > threshold) {
      const float over_db = 20.0f * std::log10(envelope / threshold);
      const float reduction_db = over_db - over_db / ratio;
      gain = std::pow(10.0f, -reduction_db / 20.0f);
    }
    s[i] *= gain;
  }
  if (makeup_gain_db != 0.0f) ohw_dsp_apply_gain(s, n, makeup_gain_db);
}

int64_t ohw_dsp_limit(float* s, int64_t n, uint32_t sample_rate, float ceiling_db, float release_ms) {
  if (!s || n <= 0) return 0;
  const float ceiling = std::pow(10.0f, ceiling_db / 20.0f);
  const float release = std::exp(-1.0f / (release_ms * (float)sample_rate / 1000.0f));
  float gr = 1.0f;
  int64_t limited = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float a = std::fabs(s[i]);
    float target = 1.0f;
    if (a > ceiling) { ++limited; target = ceiling / a; }
    if (target < gr) gr = target;                                  // instant attack
    else gr = release * gr + (1.0f - release) * target;            // smooth release
    s[i] *= gr;
  }
  return limited;
}

int64_t ohw_dsp_resample_linear(const float* in, int64_t n, uint32_t from_rate, uint32_t to_rate, float* out, int64_t out_cap) {
  if (!in || n <= 0 || from_rate == 0 || to_rate == 0) return 0;
  if (from_rate == to_rate) {                                      // resample(): same rate returns the input (:960-963)
    if (out && out_cap >= n) std::memcpy(out, in, (size_t)n * sizeof(float));
    return n;
  }
  const double ratio = (double)to_rate / (double)from_rate;
  const int64_t new_len = (int64_t)((double)n * ratio);
  if (!out || out_cap < new_len) return new_len;                   // size query
  for (int64_t i = 0; i < new_len; ++i) {
    const double src = (double)i / ratio;
    const int64_t lo = (int64_t)std::floor(src);
    const int64_t hi = lo + 1 < n - 1 ? lo + 1 : n - 1;
    const double frac = src - (double)lo;
    out[i] = in[lo] * (1.0f - (float)frac) + in[hi] * (float)frac;
  }
  return new_len;
}

void ohw_default_preprocess_config(ohw_preprocess_config* c) {
  if (!c) return;
  c->preprocessing = 0;                 // reference default: off (src/config.rs AudioConfig)
  c->normalization_enabled = 1; c->normalization_target_db = -18.0f;
  c->compression_enabled = 1; c->compression_threshold_db = -24.0f; c->compression_ratio = 4.0f;
  c->compression_attack_ms = 5.0f; c->compression_release_ms = 50.0f; c->compression_makeup_gain_db = 6.0f;
  c->limiter_enabled = 1; c->limiter_ceiling_db = -1.0f; c->limiter_release_ms = 50.0f;
}

int ohw_preprocess_audio(float* s, int64_t n, uint32_t sample_rate, const ohw_preprocess_config* c) {
  if (!s || !c || n < 0) return OHW_E_INVALID_ARG;
  if (!c->preprocessing) return OHW_OK;                                               // worker.rs:209-211
  if (c->normalization_enabled) ohw_dsp_normalize_rms(s, n, c->normalization_target_db);   // :217-219
  if (c->compression_enabled)
    ohw_dsp_compress(s, n, sample_rate, c->compression_threshold_db, c->compression_ratio, c->compression_attack_ms,
                     c->compression_release_ms, c->compression_makeup_gain_db);              // :222-230
  if (c->limiter_enabled) (void)ohw_dsp_limit(s, n, sample_rate, c->limiter_ceiling_db, c->limiter_release_ms);   // :233-235
  return OHW_OK;
}

}  // extern "C"
