// dsp.cpp — the audio preprocessing that runs right before the hot path, host side:
//   ohw_dsp_rms_db          <- AudioBuffer::rms_db          (reference src/input/audio.rs:86-102)
//   ohw_dsp_apply_gain      <- AudioBuffer::apply_gain      (:123-129)
//   ohw_dsp_normalize_rms   <- AudioBuffer::normalize_rms   (:108-120)
//   ohw_dsp_compress        <- AudioBuffer::compress        (:139-191)
//   ohw_dsp_limit           <- AudioBuffer::limit           (:197-239)
//   ohw_dsp_resample_linear <- resample_linear              (:972-990)
//   ohw_preprocess_audio    <- TranscriptionWorker::preprocess_audio (src/queue/worker.rs:196-240), without RNNoise
//   ohw_dsp_denoise         <- AudioBuffer::denoise         (:249-341): the framing around the network, the network a hook
//   ohw_preprocess_audio_ex <- preprocess_audio with its noise-reduction stage in the reference's position (worker.rs:199-207)
// The envelope follower and the limiter are first-order recurrences with a data-dependent branch per sample over the
//   ohw_dsp_resample_sinc   <- resample_sinc                (:1007-1095): rubato's SincFixedIn, restated from its published design
// WHOLE recording: one sequential chain, nothing for a GPU to parallelise - they stay on the host (about 3 ns per sample).
// Arithmetic is in fp32 in the reference's operation order.  Not built: the RNNoise NETWORK (the nnnoiseless 0.5.2 crate and
// its trained weights are not under /root/reference); everything the reference does around it is, behind ohw_denoise_engine.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

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

// ---- resample_sinc (reference src/input/audio.rs:1007-1095) -----------------------------------------------------------
// The reference hands 1024-sample chunks (the last one zero-padded) to rubato::SincFixedIn<f32>(ratio, 2.0, params, 1024, 1)
// with sinc_len 256, f_cutoff 0.95, oversampling_factor 256, SincInterpolationType::Linear, WindowFunction::BlackmanHarris2,
// appends every chunk's output and, for the last (partial) chunk, only ceil(len * ratio) samples of it.  rubato is a
// third-party crate that is not under /root/reference; what follows restates its algorithm as published (parity unpinned):
//   table    y[x] = w[x] * sinc((x - T/2) * fc / F), x = 0 .. T-1, T = sinc_len * F, w = squared Blackman-Harris (periodic over T),
//            fc = f_cutoff (times the ratio when downsampling); sub-filter F - 1 - p, tap n = y[F * n + p] / (sum(y) / F)
//   process  the resampler keeps the previous 2 * sinc_len inputs in front of the chunk; the read position idx starts at
//            -sinc_len / 2 and advances by 1 / ratio per output; an output is the linear blend (weight = fractional part of
//            idx * F) of the dot products of the sinc_len inputs starting at floor(idx) with the two nearest sub-filters;
//            a chunk's outputs stop at idx >= chunk - (sinc_len + 1) - ceil(1 / ratio); idx -= chunk for the next call.
}  // extern "C"

namespace ohw {
// the polyphase table [F = 256 sub-filters][L = 256 taps] of the sinc resampler for an output / input rate ratio (shared with
// the device resampler, resample.hip)
void sinc_table(double r, std::vector<float>& sincs) {
  constexpr int L = 256, F = 256;
  sincs.assign((size_t)F * L, 0.0f);
  const double fc = r >= 1.0 ? 0.95 : 0.95 * r;
  const int T = L * F;
  std::vector<double> y((size_t)T);
  double sum = 0.0;
  const double PI = 3.14159265358979323846;
  for (int x = 0; x < T; ++x) {
    const double xf = (double)x / (double)T;
    const double bh = 0.35875 - 0.48829 * std::cos(2.0 * PI * xf) + 0.14128 * std::cos(4.0 * PI * xf) - 0.01168 * std::cos(6.0 * PI * xf);
    const double arg = ((double)x - (double)(T / 2)) * fc / (double)F;
    const double sc = arg == 0.0 ? 1.0 : std::sin(PI * arg) / (PI * arg);
    y[(size_t)x] = bh * bh * sc;
    sum += y[(size_t)x];
  }
  sum /= (double)F;
  for (int p = 0; p < F; ++p)
    for (int n = 0; n < L; ++n) sincs[(size_t)(F - 1 - p) * L + n] = (float)(y[(size_t)(F * n + p)] / sum);   // sub-filter s = later fractional position
}
// how many output samples ohw_dsp_resample_sinc produces for n input samples: the chunk loop below without its dot products
// (output k reads the input around position -L/2 + (k + 1) / ratio whatever chunk emits it)
int64_t sinc_plan(int64_t n, double ratio) {
  constexpr int L = 256, CHUNK = 1024;
  const double t_ratio = 1.0 / ratio;
  const double end_idx = (double)(CHUNK - (L + 1) - (int64_t)std::ceil(t_ratio));
  double idx = -(double)(L / 2);
  int64_t total = 0;
  for (int64_t pos = 0; pos < n; pos += CHUNK) {
    const int64_t len = n - pos < CHUNK ? n - pos : CHUNK;
    int64_t made = 0;
    while (idx < end_idx) { idx += t_ratio; ++made; }
    idx -= (double)CHUNK;
    if (len < CHUNK) {
      const int64_t expected = (int64_t)std::ceil((double)len * ratio);
      if (expected < made) made = expected;
    }
    total += made;
  }
  return total;
}
}  // namespace ohw

extern "C" {

namespace {
struct SincResampler {
  static constexpr int L = 256, F = 256, CHUNK = 1024;
  double ratio, t_ratio, last_index;
  std::vector<float> sincs;     // [F][L]
  std::vector<float> buf;       // [2 L + CHUNK]
  explicit SincResampler(double r) : ratio(r), t_ratio(1.0 / r), last_index(-(double)(L / 2)), buf((size_t)2 * L + CHUNK, 0.0f) {
    ohw::sinc_table(r, sincs);
  }
  float dot(int64_t index, int sub) const {
    const float* w = &buf[(size_t)index];
    const float* h = &sincs[(size_t)sub * L];
    float acc = 0.0f;
    for (int n = 0; n < L; ++n) acc += w[n] * h[n];
    return acc;
  }
  // one chunk of exactly CHUNK input samples in, its outputs appended to `out`
  void process(const float* chunk, std::vector<float>& out) {
    std::memmove(buf.data(), buf.data() + CHUNK, sizeof(float) * (size_t)(2 * L));
    std::memcpy(buf.data() + 2 * L, chunk, sizeof(float) * (size_t)CHUNK);
    const double end_idx = (double)(CHUNK - (L + 1) - (int64_t)std::ceil(t_ratio));
    double idx = last_index;
    while (idx < end_idx) {
      idx += t_ratio;
      const double fl = std::floor(idx);
      int64_t i0 = (int64_t)fl;
      int s0 = (int)std::floor((idx - fl) * (double)F);
      int64_t i1 = i0;
      int s1 = s0 + 1;
      if (s1 >= F) { s1 -= F; i1 += 1; }
      const double scaled = idx * (double)F;
      const float frac = (float)(scaled - std::floor(scaled));
      const float p0 = dot(i0 + 2 * L, s0), p1 = dot(i1 + 2 * L, s1);
      out.push_back(p0 + frac * (p1 - p0));
    }
    last_index = idx - (double)CHUNK;
  }
};
}  // namespace

int64_t ohw_dsp_resample_sinc(const float* in, int64_t n, uint32_t from_rate, uint32_t to_rate, float* out, int64_t out_cap) {
  if (!in || n <= 0 || from_rate == 0 || to_rate == 0) return 0;
  if (from_rate == to_rate) {
    if (out && out_cap >= n) std::memcpy(out, in, (size_t)n * sizeof(float));
    return n;
  }
  const double ratio = (double)to_rate / (double)from_rate;
  if (ratio > 16.0 || ratio < 1.0 / 16.0) return 0;
  SincResampler rs(ratio);
  std::vector<float> res, piece, padded((size_t)SincResampler::CHUNK);
  res.reserve((size_t)((double)n * ratio) + 1024);
  for (int64_t pos = 0; pos < n; pos += SincResampler::CHUNK) {
    const int64_t len = n - pos < SincResampler::CHUNK ? n - pos : SincResampler::CHUNK;
    std::memcpy(padded.data(), in + pos, sizeof(float) * (size_t)len);
    if (len < SincResampler::CHUNK) std::memset(padded.data() + len, 0, sizeof(float) * (size_t)(SincResampler::CHUNK - len));
    piece.clear();
    rs.process(padded.data(), piece);
    size_t take = piece.size();
    if (len < SincResampler::CHUNK) {                              // the last, partial chunk: its proportional share only
      const size_t expected = (size_t)std::ceil((double)len * ratio);
      if (expected < take) take = expected;
    }
    res.insert(res.end(), piece.begin(), piece.begin() + (std::ptrdiff_t)take);
  }
  const int64_t total = (int64_t)res.size();
  if (!out || out_cap < total) return total;
  std::memcpy(out, res.data(), sizeof(float) * (size_t)total);
  return total;
}

void ohw_default_preprocess_config(ohw_preprocess_config* c) {
  if (!c) return;
  c->preprocessing = 0;                 // reference default: off (src/config.rs AudioConfig)
  c->normalization_enabled = 1; c->normalization_target_db = -18.0f;
  c->compression_enabled = 1; c->compression_threshold_db = -24.0f; c->compression_ratio = 4.0f;
  c->compression_attack_ms = 5.0f; c->compression_release_ms = 50.0f; c->compression_makeup_gain_db = 6.0f;
  c->limiter_enabled = 1; c->limiter_ceiling_db = -1.0f; c->limiter_release_ms = 50.0f;
}

// AudioBuffer::denoise (reference src/input/audio.rs:249-341) around a plugged-in frame processor
int ohw_dsp_denoise(float* s, int64_t n, uint32_t sample_rate, float strength, const ohw_denoise_engine* eng) {
  if (!s || n < 0 || !eng || !eng->process_frame || sample_rate == 0) return OHW_E_INVALID_ARG;
  if (n == 0 || !(strength > 0.0f)) return OHW_OK;                                   // :250-252
  strength = strength > 1.0f ? 1.0f : strength;                                       // :254
  constexpr uint32_t RATE = 48000;                                                    // :257-258
  constexpr int FRAME = 480;
  std::vector<float> up;
  if (sample_rate != RATE) {                                                          // :267-272, resample_for_rnnoise = linear
    up.resize((size_t)ohw_dsp_resample_linear(s, n, sample_rate, RATE, nullptr, 0));
    (void)ohw_dsp_resample_linear(s, n, sample_rate, RATE, up.data(), (int64_t)up.size());
  } else {
    up.assign(s, s + n);
  }
  if (eng->reset) eng->reset(eng->user);                                              // a fresh DenoiseState per call (:275)
  std::vector<float> den;
  den.reserve(up.size());
  float fin[FRAME], fout[FRAME];
  for (size_t pos = 0, i = 0; pos < up.size(); pos += FRAME, ++i) {                   // :283-314
    const size_t len = std::min<size_t>(FRAME, up.size() - pos);
    for (size_t j = 0; j < len; ++j) fin[j] = up[pos + j] * 32767.0f;
    for (size_t j = len; j < FRAME; ++j) fin[j] = 0.0f;
    (void)eng->process_frame(eng->user, fout, fin);
    if (i == 0) for (int j = 0; j < FRAME; ++j) den.push_back(fout[j] * ((float)j / (float)FRAME) / 32767.0f);   // fade-in
    else for (size_t j = 0; j < (len < FRAME ? len : (size_t)FRAME); ++j) den.push_back(fout[j] / 32767.0f);
  }
  std::vector<float> down;
  if (sample_rate != RATE) {                                                          // :317-321
    down.resize((size_t)ohw_dsp_resample_linear(den.data(), (int64_t)den.size(), RATE, sample_rate, nullptr, 0));
    (void)ohw_dsp_resample_linear(den.data(), (int64_t)den.size(), RATE, sample_rate, down.data(), (int64_t)down.size());
  } else {
    down.swap(den);
  }
  down.resize((size_t)n, 0.0f);                                                       // :325-326 truncate / zero-extend
  if (strength < 1.0f) for (int64_t i = 0; i < n; ++i) s[i] = s[i] * (1.0f - strength) + down[(size_t)i] * strength;   // :329-335
  else std::memcpy(s, down.data(), (size_t)n * sizeof(float));
  return OHW_OK;
}

static float passthrough_frame(void*, float* out, const float* in) {
  std::memcpy(out, in, 480 * sizeof(float));
  return 1.0f;
}
void ohw_denoise_passthrough_engine(ohw_denoise_engine* e) {
  if (!e) return;
  e->user = nullptr; e->process_frame = passthrough_frame; e->reset = nullptr;
}

int ohw_preprocess_audio_ex(float* s, int64_t n, uint32_t sample_rate, const ohw_preprocess_config* c, int noise_reduction_enabled,
                            float noise_reduction_strength, const ohw_denoise_engine* denoise) {
  if (!s || !c || n < 0) return OHW_E_INVALID_ARG;
  // noise reduction is independent of the preprocessing flag (worker.rs:197-207) and runs first
  if (noise_reduction_enabled) {
    if (!denoise) return OHW_E_INVALID_ARG;          // the reference would run RNNoise here: no engine plugged in is an error, not a skip
    const int rc = ohw_dsp_denoise(s, n, sample_rate, noise_reduction_strength, denoise);
    if (rc != OHW_OK) return rc;
  }
  return ohw_preprocess_audio(s, n, sample_rate, c);
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
