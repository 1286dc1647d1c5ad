// host_engine.cpp — the host side above the C ABI: a C++ mirror of the reference's WhisperEngine
// (reference src/engine/whisper.rs:110-387) written in C++ because the build image has no Rust
// toolchain.  Same method set, argument meaning and error behaviour:
//   ohw_engine_new         <- WhisperEngine::new          (:129-179)
//   ohw_engine_transcribe  <- WhisperEngine::transcribe   (:204-310)
//   ohw_engine_benchmark   <- WhisperEngine::benchmark    (:334-387)
//   ohw_validate_audio     <- validation::validate_audio  (src/engine/validation.rs:46-118)
//   ohw_lang_id_to_code    <- lang_id_to_code             (:627-731)
// The host owns 30 s windowing and (optionally) the token sampler, as BASELINE.json asks; the
// arithmetic is in the HIP library.  There is no CPU fallback: use_gpu = false is an error.
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "model.hpp"

namespace ohw {
extern thread_local std::string g_last_error;
}
extern "C" const ohw_ctx* ohw_state_ctx(const ohw_state* st);
using namespace ohw;

namespace {
const char* const kLangs[99] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi",
    "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk",
    "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw",
    "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo",
    "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su"};

template <typename F>
int guard(F&& f) {
  try {
    f();
    return OHW_OK;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return OHW_E_TRANSCRIBE;
  } catch (...) {
    g_last_error = "unknown error";
    return OHW_E_TRANSCRIBE;
  }
}
}  // namespace

struct ohw_engine {
  ohw_ctx* ctx = nullptr;
  ohw_state* state = nullptr;
  std::string language;
  bool translate = false;
  int max_batch = 1;
  int window_mode = OHW_WINDOW_FIXED;
  std::vector<int32_t> last_tokens;
  std::string last_text;
  std::vector<ohw_window_quality> last_quality;
  // two batches in flight for audio longer than max_batch windows (include/ohw.h, ohw_stream_create): a second state
  // and three streams, made on first use; enc_cus = 0 keeps the batches strictly one after the other
  ohw_state* state2 = nullptr;
  void* s_full = nullptr; void* s_enc = nullptr; void* s_dec = nullptr;
  int enc_cus = 96;
  int device = 0;
};

extern "C" {

const char* ohw_lang_id_to_code(int32_t id) { return (id >= 0 && id < 99) ? kLangs[id] : "unknown"; }

int32_t ohw_lang_code_to_id(const char* code) {
  if (!code) return -1;
  for (int i = 0; i < 99; ++i)
    if (std::strcmp(code, kLangs[i]) == 0) return i;
  return -1;
}

int ohw_validate_audio(const float* samples, int64_t n, uint32_t sample_rate, ohw_audio_info* info) {
  if (!info) return OHW_E_INVALID_ARG;
  std::memset(info, 0, sizeof *info);
  auto fail = [&](int code) { info->error = code; return OHW_E_VALIDATION; };
  if (n <= 0 || !samples) return fail(OHW_AUDIO_EMPTY);                      // validation.rs:51-53
  if (sample_rate != 16000u) return fail(OHW_AUDIO_BAD_RATE);               // :56-61
  const float duration = (float)n / (float)sample_rate;                     // :64
  info->duration_secs = duration;
  info->sample_count = n;
  if (duration > 7200.0f) return fail(OHW_AUDIO_TOO_LONG);                  // :67-72 (before the scan)
  if (duration < 0.1f) return fail(OHW_AUDIO_TOO_SHORT);                    // :74-79
  float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
  double ss = 0.0;
  int64_t n_nan = 0, n_inf = 0;
  for (int64_t i = 0; i < n; ++i) {                                         // :88-98
    const float v = samples[i];
    if (std::isnan(v)) ++n_nan;
    else if (std::isinf(v)) ++n_inf;
    else { mn = std::min(mn, v); mx = std::max(mx, v); ss += (double)v * (double)v; }
  }
  info->nan_count = n_nan;
  info->inf_count = n_inf;
  if (n_nan > 0) return fail(OHW_AUDIO_NAN);                                // :100-102
  if (n_inf > 0) return fail(OHW_AUDIO_INF);                                // :104-106
  info->min_value = mn;
  info->max_value = mx;
  info->rms = (float)std::sqrt(ss / (double)n);                             // :109
  info->error = OHW_AUDIO_OK;
  return OHW_OK;
}

// Host-side logits filter + arg-max: whisper.cpp's greedy path with the defaults the reference
// inherits.  Same rule order as the device sampler in decode.hip.
int32_t ohw_sample_greedy_host(const ohw_ctx* ctx, const ohw_sample_params* p, float* logits, const int32_t* cur, int n_cur,
                               float* logprob_out) {
  if (!ctx || !p || !logits) return -1;
  const ohw_special_tokens& t = ctx->tok;
  const int V = ctx->hp.n_vocab;
  const float NEG = -INFINITY;
  const bool is_initial = n_cur == 0;
  if (p->suppress_blank && is_initial) {
    logits[t.eot] = NEG;
    if (t.blank >= 0) logits[t.blank] = NEG;
  }
  logits[t.no_timestamps] = NEG;
  logits[t.sot] = NEG; logits[t.nosp] = NEG; logits[t.translate] = NEG; logits[t.transcribe] = NEG;
  logits[t.prev] = NEG; logits[t.solm] = NEG;
  for (int i = 0; i < t.n_langs; ++i) logits[t.sot + 1 + i] = NEG;
  if (p->force_len > 0 && n_cur < p->force_len) logits[t.eot] = NEG;
  if (p->no_timestamps) {
    for (int i = t.timestamp_begin; i < V; ++i) logits[i] = NEG;
  } else {
    const bool last_ts = n_cur > 0 && cur[n_cur - 1] >= t.timestamp_begin;
    const bool penult_ts = n_cur < 2 || cur[n_cur - 2] >= t.timestamp_begin;
    if (last_ts) {
      if (penult_ts) { for (int i = t.timestamp_begin; i < V; ++i) logits[i] = NEG; }
      else { for (int i = 0; i < t.eot; ++i) logits[i] = NEG; }
    }
    if (is_initial && p->max_initial_ts > 0)
      for (int i = t.timestamp_begin + p->max_initial_ts + 1; i < V; ++i) logits[i] = NEG;
    int last_seen = -1;
    for (int i = n_cur - 1; i >= 0; --i) if (cur[i] >= t.timestamp_begin) { last_seen = cur[i]; break; }
    if (last_seen >= 0) for (int i = t.timestamp_begin; i < last_seen; ++i) logits[i] = NEG;
  }
  float mx = NEG;
  for (int i = 0; i < V; ++i) mx = std::max(mx, logits[i]);
  double sum = 0.0, ts_sum = 0.0;
  float text_max = NEG;
  for (int i = 0; i < V; ++i) {
    if (!(logits[i] > NEG)) continue;
    const double e = std::exp((double)(logits[i] - mx));
    sum += e;
    if (i >= t.timestamp_begin) ts_sum += e; else text_max = std::max(text_max, logits[i]);
  }
  const float lse = mx + (float)std::log(sum);
  if (!p->no_timestamps && ts_sum > 0.0) {
    const float ts_lp = mx + (float)std::log(ts_sum) - lse;
    if (ts_lp > text_max - lse) for (int i = 0; i < t.timestamp_begin; ++i) logits[i] = NEG;
  }
  int best = 0;
  float bv = NEG;
  for (int i = 0; i < V; ++i) if (logits[i] > bv) { bv = logits[i]; best = i; }
  if (logprob_out) *logprob_out = bv - lse;
  return best;
}

int ohw_detect_language(ohw_state* st, int batch, int32_t* lang_ids_out, float* lang_probs_out) {
  return guard([&] {
    if (!st || !lang_ids_out || batch < 1) throw Error(OHW_E_INVALID_ARG, "bad argument");
    const ohw_ctx* ctx = ohw_state_ctx(st);
    const ohw_special_tokens& t = ctx->tok;
    if (ctx->hp.n_vocab < 51865) throw Error(OHW_E_INVALID_ARG, "language detection needs a multilingual model");
    const int V = ctx->hp.n_vocab, nl = t.n_langs;
    std::vector<int32_t> toks((size_t)batch, t.sot), past((size_t)batch, 0);
    std::vector<float> logits((size_t)batch * V);
    const int rc = ohw_decode(st, toks.data(), 1, past.data(), batch, logits.data());
    if (rc != OHW_OK) throw Error(rc, g_last_error);
    for (int b = 0; b < batch; ++b) {
      const float* lg = logits.data() + (size_t)b * V + t.sot + 1;
      float mx = -INFINITY;
      int best = 0;
      for (int i = 0; i < nl; ++i) if (lg[i] > mx) { mx = lg[i]; best = i; }
      lang_ids_out[b] = best;
      if (lang_probs_out) {
        double sum = 0.0;
        for (int i = 0; i < nl; ++i) sum += std::exp((double)(lg[i] - mx));
        for (int i = 0; i < nl; ++i) lang_probs_out[(size_t)b * nl + i] = (float)(std::exp((double)(lg[i] - mx)) / sum);
      }
    }
  });
}

int ohw_engine_new(const char* model_path, const char* language, int translate, int use_gpu, int device, int dtype, int max_batch,
                   ohw_engine** out) {
  return guard([&] {
    if (!out) throw Error(OHW_E_INVALID_ARG, "out is null");
    *out = nullptr;
    struct stat sb;
    if (!model_path || stat(model_path, &sb) != 0) {
      // reference :141-154: the model name is the file stem without "ggml-"
      std::string stem = model_path ? model_path : "";
      const size_t slash = stem.find_last_of('/');
      if (slash != std::string::npos) stem = stem.substr(slash + 1);
      const size_t dot = stem.find_last_of('.');
      if (dot != std::string::npos) stem = stem.substr(0, dot);
      std::string name = stem.rfind("ggml-", 0) == 0 ? stem.substr(5) : "unknown";
      throw Error(OHW_E_MODEL_NOT_FOUND, std::string("Model not found at ") + (model_path ? model_path : "(null)") +
                                             ". Run 'openhush model download " + name + "'");
    }
    if (!use_gpu)
      throw Error(OHW_E_NO_GPU, "device = \"cpu\": this engine has no CPU path (set [transcription] device to \"hip:N\")");
    const std::string lang = language ? language : "auto";
    if (lang != "auto" && ohw_lang_code_to_id(lang.c_str()) < 0)
      throw Error(OHW_E_LOAD_FAILED, "unknown language code '" + lang + "'");
    std::unique_ptr<ohw_engine> e(new ohw_engine());
    e->language = lang;
    e->translate = translate != 0;
    e->max_batch = std::max(1, max_batch);
    e->device = device;
    if (const char* ev = getenv("OHW_ENGINE_ENC_CUS")) e->enc_cus = std::max(0, atoi(ev));
    int rc = ohw_ctx_create(model_path, device, dtype, &e->ctx);
    if (rc != OHW_OK) throw Error(rc == OHW_E_MODEL_NOT_FOUND ? rc : (rc == OHW_E_NO_GPU || rc == OHW_E_OOM ? rc : OHW_E_LOAD_FAILED),
                                  "Failed to load model: " + g_last_error);
    rc = ohw_state_create(e->ctx, e->max_batch, &e->state);
    if (rc != OHW_OK) {
      const std::string msg = g_last_error;
      ohw_ctx_free(e->ctx);
      e->ctx = nullptr;
      throw Error(rc == OHW_E_OOM ? rc : OHW_E_LOAD_FAILED, "Failed to create state: " + msg);
    }
    *out = e.release();
  });
}

void ohw_engine_free(ohw_engine* e) {
  if (!e) return;
  ohw_state_free(e->state);
  if (e->state2) ohw_state_free(e->state2);
  for (void* st : {e->s_full, e->s_enc, e->s_dec}) if (st) (void)ohw_stream_destroy(st);
  ohw_ctx_free(e->ctx);
  delete e;
}

ohw_state* ohw_engine_state(ohw_engine* e) { return e ? e->state : nullptr; }
ohw_ctx* ohw_engine_ctx(ohw_engine* e) { return e ? e->ctx : nullptr; }

int ohw_engine_transcribe(ohw_engine* e, const float* samples, int64_t n, uint32_t sample_rate, char* text_buf, size_t text_cap,
                          char* language_out, uint64_t* duration_ms, ohw_audio_info* info_out) {
  return guard([&] {
    if (!e) throw Error(OHW_E_INVALID_ARG, "engine is null");
    ohw_audio_info info;
    const int vrc = ohw_validate_audio(samples, n, sample_rate, &info);   // reference :206
    if (info_out) *info_out = info;
    if (vrc != OHW_OK) {
      static const char* const names[] = {"ok", "Audio is empty (no samples)", "Unexpected sample rate", "Audio too long", "Audio too short",
                                          "Audio contains NaN values", "Audio contains infinite values"};
      throw Error(OHW_E_VALIDATION, std::string("Audio validation failed: ") + names[info.error]);
    }
    const auto t0 = std::chrono::steady_clock::now();                       // reference :231
    ohw_sample_params sp;
    ohw_default_sample_params(e->ctx, &sp);
    // reference :246-248: "auto" skips set_language and whisper.cpp keeps its default "en"
    sp.lang_id = e->language == "auto" ? 0 : ohw_lang_code_to_id(e->language.c_str());
    // reference :251-257 passes !translate to whisper-rs on the claim that the binding is inverted;
    // the resulting behaviour the reference documents is: config translate=true -> translate task
    sp.translate = e->translate ? 1 : 0;
    const ohw_special_tokens& tk = e->ctx->tok;
    if (sp.lang_id >= tk.n_langs) throw Error(OHW_E_TRANSCRIBE, "language is not supported by this model");

    e->last_tokens.clear();
    e->last_quality.clear();
    std::string text;
    const int max_tok = e->ctx->hp.n_text_ctx;
    // whisper.cpp's per-window acceptance test (as recalled): token-frequency entropy of the last 32 tokens
    // and the average log-probability
    auto quality = [&](const int32_t* t, int n, float sum_lp) {
      ohw_window_quality q{};
      q.n_tokens = n;
      q.avg_logprob = n > 0 ? sum_lp / (float)n : 0.0f;
      const int n32 = std::min(32, n);
      double ent = 0.0;
      for (int i = n - n32; i < n; ++i) {
        bool first = true;
        int cnt = 0;
        for (int j = n - n32; j < n; ++j) {
          if (t[j] == t[i]) { if (j < i) first = false; ++cnt; }
        }
        if (first && n32 > 0) { const double pr = (double)cnt / n32; ent -= pr * std::log(pr); }
      }
      q.entropy = (float)ent;
      q.would_fallback = (n > 0 && (q.entropy < 2.4f || q.avg_logprob < -1.0f)) ? 1 : 0;
      e->last_quality.push_back(q);
    };
    auto append_text = [&](const int32_t* t, int n) {
      for (int i = 0; i < n; ++i) {
        e->last_tokens.push_back(t[i]);
        if (t[i] < tk.eot) {                                                // segment text = text tokens only (:271-279)
          const char* sp_ = nullptr;
          const int len = ohw_token_text(e->ctx, t[i], &sp_);
          text.append(sp_, (size_t)len);
        }
      }
    };
    if (e->window_mode == OHW_WINDOW_SEEK) {
      // whisper.cpp's seek loop as recalled: sequential windows, advanced by the last timestamp
      std::vector<int32_t> toks((size_t)max_tok);
      int32_t ntok = 0;
      float slp = 0.f;
      const int64_t seek_end = n / HOP;               // 10 ms frames
      int64_t seek = 0;
      while (seek + 100 < seek_end) {
        const int64_t off = seek * HOP;
        const int32_t ns1 = (int32_t)std::min<int64_t>(CHUNK_SAMPLES, n - off);
        int rc = ohw_mel(e->state, samples + off, CHUNK_SAMPLES, &ns1, 1, 0, OHW_MEL_ZERO_TAIL, nullptr);
        if (rc == OHW_OK) rc = ohw_encode(e->state, 1);
        if (rc == OHW_OK) rc = ohw_greedy(e->state, &sp, 1, toks.data(), &ntok, max_tok, &slp);
        if (rc != OHW_OK) throw Error(OHW_E_TRANSCRIBE, "Transcription failed: " + g_last_error);
        int64_t seek_delta = 100 * 30;                // a full window when no timestamp was produced
        int result_len = ntok;
        for (int i = 0; i < ntok; ++i)
          if (toks[(size_t)i] > tk.timestamp_begin) { seek_delta = 2 * (int64_t)(toks[(size_t)i] - tk.timestamp_begin); result_len = i + 1; }
        if (seek_delta <= 0) seek_delta = 100 * 30;
        quality(toks.data(), ntok, slp);
        append_text(toks.data(), result_len);
        seek += seek_delta;
      }
    } else {
      // host-side windowing: fixed 30 s cuts (BASELINE.json north_star)
      const int64_t n_win = (n + CHUNK_SAMPLES - 1) / CHUNK_SAMPLES;
      std::vector<int32_t> toks((size_t)e->max_batch * max_tok), ntok((size_t)e->max_batch), ns((size_t)e->max_batch);
      std::vector<float> slp((size_t)e->max_batch);
      const int64_t n_batches = (n_win + e->max_batch - 1) / e->max_batch;
      auto batch_of = [&](int64_t bi) { return (int)std::min<int64_t>(e->max_batch, n_win - bi * e->max_batch); };
      auto collect = [&](int B) {
        for (int b = 0; b < B; ++b) {
          quality(&toks[(size_t)b * max_tok], ntok[(size_t)b], slp[(size_t)b]);
          append_text(&toks[(size_t)b * max_tok], ntok[(size_t)b]);
        }
      };
      bool pipelined = n_batches > 1 && e->enc_cus > 0;
      if (pipelined && !e->state2) {
        // first long input: the second state and the three streams (all CUs / encoder's share / decoder's share)
        int rc = ohw_state_create(e->ctx, e->max_batch, &e->state2);
        if (rc == OHW_OK) rc = ohw_stream_create(e->device, 0, 0, &e->s_full);
        int total = 0;
        if (rc == OHW_OK && hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) rc = OHW_E_TRANSCRIBE;
        if (rc == OHW_OK && e->enc_cus >= total) e->enc_cus = std::max(1, total * 3 / 8);   // a smaller device: the same 3 : 5 split
        if (rc == OHW_OK) rc = ohw_stream_create(e->device, 0, e->enc_cus, &e->s_enc);
        if (rc == OHW_OK) rc = ohw_stream_create(e->device, e->enc_cus, total - e->enc_cus, &e->s_dec);
        if (rc != OHW_OK) {
          // no CU-masked queues (or no memory for the second state) here: one batch after the other from now on
          if (e->state2) { ohw_state_free(e->state2); e->state2 = nullptr; }
          for (void** st : {&e->s_full, &e->s_enc, &e->s_dec}) if (*st) { (void)ohw_stream_destroy(*st); *st = nullptr; }
          e->enc_cus = 0;
          pipelined = false;
        }
      }
      if (!pipelined) {
        for (int64_t bi = 0; bi < n_batches; ++bi) {
          const int64_t w0 = bi * e->max_batch;
          const int B = batch_of(bi);
          for (int b = 0; b < B; ++b) ns[(size_t)b] = (int32_t)std::min<int64_t>(CHUNK_SAMPLES, n - (w0 + b) * CHUNK_SAMPLES);
          int rc = ohw_mel(e->state, samples + w0 * CHUNK_SAMPLES, CHUNK_SAMPLES, ns.data(), B, 0, OHW_MEL_ZERO_TAIL, nullptr);
          if (rc == OHW_OK) rc = ohw_encode(e->state, B);
          if (rc == OHW_OK) rc = ohw_greedy(e->state, &sp, B, toks.data(), ntok.data(), max_tok, slp.data());
          if (rc != OHW_OK) throw Error(OHW_E_TRANSCRIBE, "Transcription failed: " + g_last_error);   // reference :266-268
          collect(B);
        }
      } else {
        // mel + encoder + cross-K/V of batch i+1 (MFMA-bound) run beside the greedy decode of batch i (HBM- and
        // latency-bound) on disjoint CUs; the first front end and the last decode have the device to themselves
        ohw_state* sts[2] = {e->state, e->state2};
        std::vector<int32_t> ns2[2] = {std::vector<int32_t>((size_t)e->max_batch), std::vector<int32_t>((size_t)e->max_batch)};
        auto check = [&](int rc) { if (rc != OHW_OK) throw Error(OHW_E_TRANSCRIBE, "Transcription failed: " + g_last_error); };
        auto restore = [&] { (void)ohw_state_set_stream(e->state, nullptr); (void)ohw_state_set_stream(e->state2, nullptr); };
        auto front = [&](int64_t bi, void* stream) {
          ohw_state* st = sts[bi & 1];
          std::vector<int32_t>& nsv = ns2[bi & 1];
          const int64_t w0 = bi * e->max_batch;
          const int B = batch_of(bi);
          for (int b = 0; b < B; ++b) nsv[(size_t)b] = (int32_t)std::min<int64_t>(CHUNK_SAMPLES, n - (w0 + b) * CHUNK_SAMPLES);
          check(ohw_state_set_stream(st, stream));
          check(ohw_mel(st, samples + w0 * CHUNK_SAMPLES, CHUNK_SAMPLES, nsv.data(), B, 0, OHW_MEL_ZERO_TAIL, nullptr));
          check(ohw_encode(st, B));
        };
        try {
          front(0, e->s_full);
          void* last_front = e->s_full;
          for (int64_t bi = 0; bi < n_batches; ++bi) {
            const bool more = bi + 1 < n_batches;
            void* dstream = more ? e->s_dec : e->s_full;
            check(ohw_stream_wait(dstream, last_front));            // this batch's cross-K/V before its decode
            if (more) {
              check(ohw_stream_wait(e->s_enc, last_front));
              front(bi + 1, e->s_enc);
              last_front = e->s_enc;
            }
            check(ohw_state_set_stream(sts[bi & 1], dstream));
            const int B = batch_of(bi);
            check(ohw_greedy(sts[bi & 1], &sp, B, toks.data(), ntok.data(), max_tok, slp.data()));
            collect(B);
          }
        } catch (...) {
          (void)hipDeviceSynchronize();
          restore();
          throw;
        }
        restore();
      }
    }
    // reference :282-283: trim
    const size_t b0 = text.find_first_not_of(" \t\r\n");
    const size_t b1 = text.find_last_not_of(" \t\r\n");
    text = b0 == std::string::npos ? std::string() : text.substr(b0, b1 - b0 + 1);
    e->last_text = text;
    if (text_buf && text_cap > 0) {
      const size_t ncopy = std::min(text.size(), text_cap - 1);
      std::memcpy(text_buf, text.data(), ncopy);
      text_buf[ncopy] = 0;
    }
    if (language_out) {
      // reference :288-296: "auto" reports the state's language id (whisper.cpp default "en" -> 0)
      const std::string lang = e->language == "auto" ? ohw_lang_id_to_code(sp.lang_id) : e->language;
      std::strncpy(language_out, lang.c_str(), 7);
      language_out[7] = 0;
    }
    if (duration_ms)
      *duration_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  });
}

int ohw_engine_last_text(ohw_engine* e, const char** text, size_t* len) {
  if (!e || !text) return OHW_E_INVALID_ARG;
  *text = e->last_text.c_str();
  if (len) *len = e->last_text.size();
  return OHW_OK;
}

int ohw_engine_last_quality(ohw_engine* e, const ohw_window_quality** q, int* n_windows) {
  if (!e || !q || !n_windows) return OHW_E_INVALID_ARG;
  *q = e->last_quality.data();
  *n_windows = (int)e->last_quality.size();
  return OHW_OK;
}

int ohw_engine_set_window_mode(ohw_engine* e, int mode) {
  if (!e || (mode != OHW_WINDOW_FIXED && mode != OHW_WINDOW_SEEK)) return OHW_E_INVALID_ARG;
  e->window_mode = mode;
  return OHW_OK;
}

int ohw_engine_last_tokens(ohw_engine* e, const int32_t** tokens, int* n) {
  if (!e || !tokens || !n) return OHW_E_INVALID_ARG;
  *tokens = e->last_tokens.data();
  *n = (int)e->last_tokens.size();
  return OHW_OK;
}

int ohw_engine_benchmark(ohw_engine* e, float safety_margin, float* overhead_secs, float* recommended, float* test_audio_secs) {
  return guard([&] {
    if (!e) throw Error(OHW_E_INVALID_ARG, "engine is null");
    const float test_duration = 2.0f;                                         // reference :341
    std::vector<float> silence((size_t)(test_duration * 16000.0f), 0.0f);
    char buf[8];
    (void)ohw_engine_transcribe(e, silence.data(), (int64_t)silence.size(), 16000, buf, sizeof buf, nullptr, nullptr, nullptr);  // warm-up :353
    uint64_t total_ms = 0;
    for (int i = 0; i < 3; ++i) {                                            // :356-365
      const auto t0 = std::chrono::steady_clock::now();
      (void)ohw_engine_transcribe(e, silence.data(), (int64_t)silence.size(), 16000, buf, sizeof buf, nullptr, nullptr, nullptr);
      total_ms += (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    uint64_t avg_ms = total_ms / 3;                                          // integer average, :367
    // DEVIATION (documented in INTEGRATION.md): the reference has no lower bound, and an engine this
    // fast would round to 0 ms, which silently disables streaming chunks (src/daemon.rs:1189-1193).
    if (avg_ms < 1) avg_ms = 1;
    const float overhead = (float)avg_ms / 1000.0f;                          // :368
    if (overhead_secs) *overhead_secs = overhead;
    if (recommended) *recommended = overhead * (1.0f + safety_margin);       // :373
    if (test_audio_secs) *test_audio_secs = test_duration;
  });
}

}  // extern "C"
