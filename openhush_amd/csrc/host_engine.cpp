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
#include <random>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "model.hpp"

namespace ohw {
extern thread_local std::string g_last_error;
const float* state_bias_host(const ohw_state* st);
void state_share_recording(ohw_state* dst, ohw_state* src);
void state_drop_recording(ohw_state* st);
void state_set_graph_max_batch(ohw_state* st, int max_batch);
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
  ApiScope api;
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

#include "host_engine.hpp"

namespace {
// ---- whisper.cpp's per-window bookkeeping (whisper_full_with_state as recalled) ---------
struct SeqEval {
  int n_sampled = 0;     // tokens up to and including the one that ended the pass
  int result_len = 0;    // whisper.cpp result_len: up to and including the last timestamp token (> token_beg)
  int n_keep = 0;        // tokens that reach the text
  int seek_delta = 3000; // 10 ms frames the window advances by
  bool failed = false, completed = false;
  float avg_logprob = -INFINITY, entropy = 0.f;
};

// frames of 10 ms whisper.cpp attributes to n samples: mel.n_len_org = 1 + (n + 200 - 400) / 160
int mel_frames(int64_t n) { return (int)(1 + (n + 200 - 400) / 160); }

// One sampled sequence, judged after the fact (greedy decoding is causal: cutting at the first exit the walk finds equals
// having stopped there).  tok[0..n): the sampled tokens, the end-of-text token last when it was sampled.
//   exits      end-of-text; a timestamp that leaves less than 1 s of audio (seek + seek_delta + 100 >= seek_end);
//              failure when timestamps go back, when end-of-text arrives with no timestamp and audio left, or when the
//              loop reaches n_max without a timestamp past the middle of the window (the repetition-loop guard)
//   score      average log-probability and token-frequency entropy of the last 32 of the first result_len tokens
//   kept       fixed 30 s cuts: everything before the exit (each window is decoded once); seek loop: result_len tokens
//              (what follows the last timestamp is decoded again by the next window)
SeqEval evaluate_sequence(const ohw_special_tokens& tk, const int32_t* tok, const float* plog, int n, int seek, int seek_end, int n_max,
                          bool no_timestamps, int window_mode) {
  SeqEval r;
  bool has_ts = false;
  int stop = n;
  for (int i = 0; i < n; ++i) {
    const int t = tok[i];
    if (t > tk.timestamp_begin) {
      const int nd = 2 * (t - tk.timestamp_begin);
      if (has_ts && r.seek_delta > nd && r.result_len < i) { r.failed = true; stop = i + 1; break; }
      r.seek_delta = nd; r.result_len = i + 1; has_ts = true;
    }
    const bool audio_end = seek + r.seek_delta + 100 >= seek_end;
    if (t == tk.eot || (has_ts && audio_end)) {
      if (r.result_len == 0 && !no_timestamps) {
        if (audio_end) r.result_len = i + 1;
        else { r.failed = true; stop = i + 1; break; }
      }
      if (no_timestamps) { r.result_len = i + 1; r.seek_delta = 3000; }
      r.completed = true; stop = i + 1;
      break;
    }
    if (i == n_max - 1 && (r.result_len == 0 || r.seek_delta < 1500)) { r.failed = true; stop = i + 1; break; }
  }
  r.n_sampled = stop;
  if (r.result_len > 0) {
    double sum = 0.0;
    for (int i = 0; i < r.result_len; ++i) sum += plog[i];
    r.avg_logprob = (float)(sum / r.result_len);
    const int n32 = std::min(32, r.result_len), i0 = r.result_len - n32;
    double ent = 0.0;
    for (int i = i0; i < r.result_len; ++i) {
      bool first = true;
      int cnt = 0;
      for (int j = i0; j < r.result_len; ++j) if (tok[j] == tok[i]) { if (j < i) first = false; ++cnt; }
      if (first) { const double pr = (double)cnt / n32; ent -= pr * std::log(pr); }
    }
    r.entropy = (float)ent;
  }
  int keep = r.failed ? stop : (window_mode == OHW_WINDOW_SEEK ? r.result_len : stop);
  while (keep > 0 && tok[keep - 1] == tk.eot) --keep;
  r.n_keep = keep;
  return r;
}

bool needs_fallback(const SeqEval& r, const ohw_decode_policy& q, float no_speech_prob, bool is_last) {
  if (is_last) return false;
  const bool failed = r.failed || r.result_len == 0 || r.entropy < q.entropy_thold;
  return failed || (r.avg_logprob < q.logprob_thold && no_speech_prob < q.no_speech_thold);
}

struct WindowRun {
  std::vector<int32_t> tok;   // sampled tokens of the kept pass (end-of-text last when sampled)
  std::vector<float> plog;
  float nosp = 0.f, temp = 0.f;
  SeqEval ev;
  bool pending = false, t0_failed = false;
};

std::vector<float> ladder(const ohw_decode_policy& q) {
  std::vector<float> t;
  if (q.temperature_inc > 0.0f)
    for (float x = q.temperature_inc; x < 1.0f + 1e-6f && t.size() < 16; x += q.temperature_inc) t.push_back(x);
  return t;
}
}  // namespace

namespace ohw {

ohw_engine* engine_wrap_ctx(ohw_ctx* ctx, const std::string& language, bool translate, int max_batch, int device) {
  std::unique_ptr<ohw_engine> e(new ohw_engine());
  e->language = language;
  e->translate = translate;
  e->max_batch = std::max(1, max_batch);
  e->device = device;
  if (const char* ev = getenv("OHW_ENGINE_ENC_CUS")) e->enc_cus = std::max(0, atoi(ev));
  if (const char* ev = getenv("OHW_ENGINE_LANES")) e->lanes = std::min(16, std::max(1, atoi(ev)));
  if (const char* ev = getenv("OHW_ENGINE_MERGE")) e->merge = std::min(8, std::max(1, atoi(ev)));
  if (e->max_batch * e->merge > 256) e->merge = std::max(1, 256 / e->max_batch);     // a state holds at most 256 windows
  if (const char* ev = getenv("OHW_ENGINE_SCHEDULE")) {
    const std::string v = ev;
    e->schedule = v == "sequential" ? OHW_SCHEDULE_SEQUENTIAL : v == "pipeline" ? OHW_SCHEDULE_PIPELINE : OHW_SCHEDULE_LANES;
  }
  const int rc = ohw_state_create(ctx, e->max_batch, &e->state);
  if (rc != OHW_OK) throw Error(rc == OHW_E_OOM ? rc : OHW_E_LOAD_FAILED, "Failed to create state: " + g_last_error);
  e->ctx = ctx;
  return e.release();
}

void engine_transcribe_core(ohw_engine* e, const float* samples, int64_t n, std::string* text_out, int64_t win_first, int64_t win_step) {
  std::string& text = *text_out;
  if (win_first < 0 || win_step < 1) throw Error(OHW_E_INVALID_ARG, "transcribe: window dealing must be first >= 0, step >= 1");
    ohw_sample_params sp;
    ohw_default_sample_params(e->ctx, &sp);
    // reference :246-248: "auto" skips set_language and whisper.cpp keeps its default "en"
    sp.lang_id = e->language == "auto" ? 0 : ohw_lang_code_to_id(e->language.c_str());
    // reference :251-257 passes !translate to whisper-rs on the claim that the binding is inverted;
    // the resulting behaviour the reference documents is: config translate=true -> translate task
    sp.translate = e->translate ? 1 : 0;
    if (e->force_len > 0) { sp.force_len = std::min(e->force_len, sp.n_max); sp.n_max = sp.force_len; }      // bench: fixed decode length
    const ohw_special_tokens& tk = e->ctx->tok;
    if (sp.lang_id >= tk.n_langs) throw Error(OHW_E_TRANSCRIBE, "language is not supported by this model");

    e->last_tokens.clear();
    e->last_quality.clear();
    e->last_trace.clear();
    const int max_tok = e->ctx->hp.n_text_ctx;
    const int V = e->ctx->hp.n_vocab;
    const int n_max = sp.n_max;
    const ohw_decode_policy& pol = e->policy;
    const std::vector<float> temps = ladder(pol);
    int32_t prompt[8];
    int n_prompt = 0;
    prompt[n_prompt++] = tk.sot;
    if (V >= 51865) { prompt[n_prompt++] = tk.sot + 1 + sp.lang_id; prompt[n_prompt++] = sp.translate ? tk.translate : tk.transcribe; }
    if (sp.no_timestamps) prompt[n_prompt++] = tk.no_timestamps;
    auto check = [&](int rc) { if (rc != OHW_OK) throw Error(OHW_E_TRANSCRIBE, "Transcription failed: " + g_last_error); };   // reference :266-268
    // per-decode scratch: one per lane when several batches decode side by side
    struct Scratch {
      std::vector<int32_t> toks, ntok, eot, trace;
      std::vector<float> lps, nsp, logits;
      std::vector<WindowRun> runs;
      Scratch(int B, int max_tok) : toks((size_t)B * max_tok), ntok((size_t)B), eot((size_t)B), lps((size_t)B * (max_tok + 1)), nsp((size_t)B) {}
    };
    auto trace = [&](Scratch& sc, int64_t window, float temp, const std::vector<int32_t>& t) {
      sc.trace.push_back((int32_t)window); sc.trace.push_back((int32_t)std::lround(temp * 1000.f)); sc.trace.push_back((int32_t)t.size());
      sc.trace.insert(sc.trace.end(), t.begin(), t.end());
    };

    // T = 0: the device-resident greedy loop for B windows; then whisper.cpp's bookkeeping per window
    auto greedy_t0 = [&](Scratch& sc, ohw_state* st, int B, const int* seek, const int* seek_end, int64_t w0) {
      std::vector<int32_t>&toks = sc.toks, &ntok = sc.ntok, &eot = sc.eot;
      std::vector<float>&lps = sc.lps, &nsp = sc.nsp;
      std::vector<WindowRun>& runs = sc.runs;
      ohw_greedy_result gr{};
      gr.tokens = toks.data(); gr.n_tokens = ntok.data(); gr.token_logprobs = lps.data(); gr.ended_by_eot = eot.data(); gr.no_speech_prob = nsp.data();
      check(ohw_greedy_ex(st, &sp, B, max_tok, &gr));
      runs.assign((size_t)B, WindowRun());
      for (int b = 0; b < B; ++b) {
        WindowRun& r = runs[(size_t)b];
        const int nt = ntok[(size_t)b] + (eot[(size_t)b] ? 1 : 0);
        r.tok.assign(&toks[(size_t)b * max_tok], &toks[(size_t)b * max_tok] + ntok[(size_t)b]);
        if (eot[(size_t)b]) r.tok.push_back(tk.eot);
        r.plog.assign(&lps[(size_t)b * (max_tok + 1)], &lps[(size_t)b * (max_tok + 1)] + nt);
        r.nosp = nsp[(size_t)b];
        r.ev = evaluate_sequence(tk, r.tok.data(), r.plog.data(), nt, seek[b], seek_end[b], n_max, sp.no_timestamps != 0, e->window_mode);
        r.t0_failed = needs_fallback(r.ev, pol, r.nosp, false);
        r.pending = needs_fallback(r.ev, pol, r.nosp, temps.empty());
        trace(sc, w0 + b, 0.f, r.tok);
      }
    };
    // the temperature ladder for the windows of a batch whose pass failed the acceptance test: the HOST samples
    // (std::mt19937 + std::discrete_distribution, as whisper.cpp's decoders do), the device runs the decoder steps of the
    // pending windows only and re-uses their resident cross K/V; the logits of the pending rows cross PCIe every step
    auto run_ladder = [&](Scratch& sc, ohw_state* st, int B, const int* seek, const int* seek_end, int64_t w0, const std::vector<ohw_rng*>& rngs) {
      std::vector<float>& logits = sc.logits;
      std::vector<WindowRun>& runs = sc.runs;
      for (size_t ti = 0; ti < temps.size(); ++ti) {
        std::vector<int32_t> active((size_t)B, 0);
        bool any = false;
        for (int b = 0; b < B; ++b) if (runs[(size_t)b].pending) { active[(size_t)b] = 1; any = true; }
        if (!any) break;
        const float T = temps[ti];
        const bool is_last = ti + 1 == temps.size();
        logits.resize((size_t)B * V);
        std::vector<int32_t> ptoks((size_t)B * n_prompt), past((size_t)B, 0), feed((size_t)B, tk.eot), npast((size_t)B, n_prompt), live = active;
        for (int b = 0; b < B; ++b) std::memcpy(&ptoks[(size_t)b * n_prompt], prompt, (size_t)n_prompt * 4);
        check(ohw_decode_active(st, ptoks.data(), n_prompt, past.data(), B, active.data(), logits.data()));
        std::vector<WindowRun> pass((size_t)B);
        for (int i = 0; i < n_max; ++i) {
          bool any_live = false;
          for (int b = 0; b < B; ++b) {
            if (!live[(size_t)b]) continue;
            WindowRun& r = pass[(size_t)b];
            float lp = 0.f;
            if (const float* bias = state_bias_host(st))          // the state's logit bias: the device sampler adds it itself
              for (int i2 = 0; i2 < V; ++i2) logits[(size_t)b * V + i2] += bias[i2];
            const int32_t t = ohw_sample_host(e->ctx, &sp, &logits[(size_t)b * V], r.tok.data(), (int)r.tok.size(), T, rngs[(size_t)b], &lp, &r.nosp);
            if (t < 0) throw Error(OHW_E_TRANSCRIBE, "Transcription failed: host sampler");
            r.tok.push_back(t); r.plog.push_back(lp);
            const SeqEval ev = evaluate_sequence(tk, r.tok.data(), r.plog.data(), (int)r.tok.size(), seek[b], seek_end[b], n_max,
                                                 sp.no_timestamps != 0, e->window_mode);
            if (t == tk.eot || ev.completed || ev.failed || (int)r.tok.size() >= n_max || npast[(size_t)b] + 1 >= max_tok) live[(size_t)b] = 0;
            else { feed[(size_t)b] = t; any_live = true; }
          }
          if (!any_live) break;
          check(ohw_decode_active(st, feed.data(), 1, npast.data(), B, live.data(), logits.data()));
          for (int b = 0; b < B; ++b) if (live[(size_t)b]) ++npast[(size_t)b];
        }
        for (int b = 0; b < B; ++b) {
          if (!active[(size_t)b]) continue;
          WindowRun& r = pass[(size_t)b];
          r.temp = T;
          r.ev = evaluate_sequence(tk, r.tok.data(), r.plog.data(), (int)r.tok.size(), seek[b], seek_end[b], n_max, sp.no_timestamps != 0, e->window_mode);
          r.t0_failed = runs[(size_t)b].t0_failed;
          r.pending = needs_fallback(r.ev, pol, r.nosp, is_last);
          trace(sc, w0 + b, T, r.tok);
          runs[(size_t)b] = std::move(r);
        }
      }
    };
    auto emit = [&](const WindowRun& r) {
      // whisper.cpp (>= 1.7.3 as recalled): a window whose no-speech probability is high AND whose text is unlikely is dropped
      const bool no_speech = r.nosp > pol.no_speech_thold && r.ev.avg_logprob < pol.logprob_thold;
      const int keep = no_speech ? 0 : r.ev.n_keep;
      for (int i = 0; i < keep; ++i) {
        e->last_tokens.push_back(r.tok[(size_t)i]);
        if (r.tok[(size_t)i] < tk.eot) {                                    // segment text = text tokens only (:271-279)
          const char* sp_ = nullptr;
          const int len = ohw_token_text(e->ctx, r.tok[(size_t)i], &sp_);
          text.append(sp_, (size_t)len);
        }
      }
      ohw_window_quality q{};
      q.n_tokens = keep; q.avg_logprob = r.ev.avg_logprob; q.entropy = r.ev.entropy; q.would_fallback = r.t0_failed ? 1 : 0;
      q.temperature = r.temp; q.no_speech_prob = r.nosp; q.no_speech = no_speech ? 1 : 0; q.seek_delta = r.ev.seek_delta;
      q.result_len = r.ev.result_len; q.failed = r.ev.failed ? 1 : 0;
      e->last_quality.push_back(q);
    };
    struct Rngs {   // whisper.cpp seeds every decoder's generator with 0 once per whisper_full call
      std::vector<ohw_rng*> v;
      ~Rngs() { for (ohw_rng* r : v) ohw_rng_free(r); }
      void reset(size_t nr) { for (ohw_rng* r : v) ohw_rng_free(r); v.clear(); for (size_t i = 0; i < nr; ++i) v.push_back(ohw_rng_new(0)); }
    } rngs;

    // a batch records its passes in the order it ran them (every window's T = 0 pass, then the ladder's rungs); the trace lists
    // them window by window, each window's passes in order, so it reads the same however the windows were batched
    auto flush_trace = [&](Scratch& sc) {
      std::vector<std::pair<int32_t, size_t>> recs;            // (window, offset of the record)
      for (size_t i = 0; i + 3 <= sc.trace.size(); i += 3 + (size_t)sc.trace[i + 2]) recs.emplace_back(sc.trace[i], i);
      std::stable_sort(recs.begin(), recs.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      for (const auto& r : recs) e->last_trace.insert(e->last_trace.end(), sc.trace.begin() + (long)r.second, sc.trace.begin() + (long)(r.second + 3 + (size_t)sc.trace[r.second + 2]));
      sc.trace.clear();
    };
    if (e->window_mode == OHW_WINDOW_SEEK) {
      // whisper.cpp's seek loop as recalled: sequential windows, advanced by the last timestamp; one
      // generator for the whole call
      rngs.reset(1);
      Scratch sc(1, max_tok);
      const int seek_end = mel_frames(n);
      int seek = 0;
      int64_t w = 0;
      // whisper.cpp computes the log-mel spectrogram of the whole input once (the clamp uses its global maximum) and every
      // window reads 3000 frames of it at its seek offset
      if (seek_end >= 100) check(ohw_recording_set(e->state, samples, n, 0, nullptr));
      while (seek_end >= 100 && seek + 100 < seek_end) {
        const int32_t seek32 = seek;
        check(ohw_mel_seek(e->state, &seek32, 1, nullptr));      // 3000 frames of the recording-wide spectrogram
        check(ohw_encode(e->state, 1));
        greedy_t0(sc, e->state, 1, &seek, &seek_end, w);
        run_ladder(sc, e->state, 1, &seek, &seek_end, w, rngs.v);
        flush_trace(sc);
        emit(sc.runs[0]);
        seek += sc.runs[0].ev.seek_delta > 0 ? sc.runs[0].ev.seek_delta : 3000;
        ++w;
      }
    } else {
      // host-side windowing: fixed 30 s cuts (BASELINE.json north_star).  Each cut is its own `full` call in
      // whisper.cpp terms: seek 0, its own frame count as the end of the audio, a fresh generator - and, like a call with
      // less than 1 s of audio (`seek + 100 >= seek_end` before the first window), a cut of at most 100 frames yields nothing
      // the pool deals the recording's windows round-robin: this engine takes windows win_first, win_first + win_step, ... of
      // the WHOLE recording it is handed (no copy, and - FIXED_RECORDING_MEL - the real neighbouring samples and the
      // recording-wide clamp maximum); k counts this engine's windows, rec_win(k) is the window of the recording
      const int64_t n_win_rec = (n + CHUNK_SAMPLES - 1) / CHUNK_SAMPLES;
      const int64_t n_win = n_win_rec > win_first ? (n_win_rec - win_first + win_step - 1) / win_step : 0;
      auto rec_win = [&](int64_t k) { return win_first + k * win_step; };
      const int64_t n_batches = (n_win + e->max_batch - 1) / e->max_batch;
      auto batch_of = [&](int64_t bi) { return (int)std::min<int64_t>(e->max_batch, n_win - bi * e->max_batch); };
      // windows w0 .. w0 + B of the recording into st: each cut on its own (its own `full()` call), or cut from the spectrogram
      // of the whole recording (FIXED_RECORDING_MEL: st reads the recording e->state holds)
      const bool rec_mel = e->window_mode == OHW_WINDOW_FIXED_RECORDING_MEL;
      auto mel_windows = [&](ohw_state* st, int64_t w0, int B, const int32_t* nsv) {
        if (!rec_mel) { check(ohw_mel(st, samples + rec_win(w0) * CHUNK_SAMPLES, win_step * CHUNK_SAMPLES, nsv, B, 0, OHW_MEL_ZERO_TAIL, nullptr)); return; }
        std::vector<int32_t> seeks((size_t)B);
        for (int b = 0; b < B; ++b) seeks[(size_t)b] = (int32_t)(rec_win(w0 + b) * CHUNK_FRAMES);
        check(ohw_mel_seek(st, seeks.data(), B, nullptr));
      };
      if (rec_mel) check(ohw_recording_set(e->state, samples, n, 0, nullptr));
      auto fill_ns = [&](int64_t bi, std::vector<int32_t>& nsv) {
        const int64_t w0 = bi * e->max_batch;
        const int B = batch_of(bi);
        for (int b = 0; b < B; ++b) nsv[(size_t)b] = (int32_t)std::min<int64_t>(CHUNK_SAMPLES, n - rec_win(w0 + b) * CHUNK_SAMPLES);
      };
      auto front = [&](int64_t bi, ohw_state* st, void* stream, std::vector<int32_t>& nsv) {
        fill_ns(bi, nsv);
        check(ohw_state_set_stream(st, stream));
        mel_windows(st, bi * e->max_batch, batch_of(bi), nsv.data());
        check(ohw_encode(st, batch_of(bi)));
      };
      // the decode of one batch on whatever stream the state is set to; fills sc.runs (and sc.trace); re-entrant per Scratch
      auto decode_windows = [&](Scratch& sc, ohw_state* st, int64_t w0, int B, const int32_t* nsb) {
        std::vector<int> zero((size_t)B, 0), ends((size_t)B);
        for (int b = 0; b < B; ++b) ends[(size_t)b] = mel_frames(nsb[b]);
        greedy_t0(sc, st, B, zero.data(), ends.data(), w0);
        for (int b = 0; b < B; ++b) if (ends[(size_t)b] <= 100) { sc.runs[(size_t)b] = WindowRun(); sc.runs[(size_t)b].ev.result_len = 0; }
        bool any = false;
        for (int b = 0; b < B; ++b) any = any || sc.runs[(size_t)b].pending;
        if (any) {
          Rngs lr;
          lr.reset((size_t)B);
          run_ladder(sc, st, B, zero.data(), ends.data(), w0, lr.v);
        }
      };
      auto decode_batch = [&](Scratch& sc, ohw_state* st, int64_t bi, const int32_t* nsb) { decode_windows(sc, st, bi * e->max_batch, batch_of(bi), nsb); };
      auto collect = [&](Scratch& sc, int B) {
        flush_trace(sc);
        for (int b = 0; b < B; ++b) emit(sc.runs[(size_t)b]);
      };
      int schedule = n_batches > 1 ? e->schedule : OHW_SCHEDULE_SEQUENTIAL;
      if (schedule == OHW_SCHEDULE_LANES && e->lanes < 2) schedule = OHW_SCHEDULE_SEQUENTIAL;
      if (schedule == OHW_SCHEDULE_PIPELINE && e->enc_cus <= 0) schedule = OHW_SCHEDULE_SEQUENTIAL;
      // ---- resources of the overlapped schedules, made when a long input first needs them: more states, CU-masked streams ----
      if (schedule != OHW_SCHEDULE_SEQUENTIAL) {
        int rc = OHW_OK, total = 0;
        if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || total < 2) rc = OHW_E_TRANSCRIBE;
        if (rc == OHW_OK && !e->s_full) rc = ohw_stream_create(e->device, 0, 0, &e->s_full);
        if (e->states.empty()) e->states.assign(1, e->state);
        // LANES: decode batches of up to merge front-end batches (ohw_encode_slice); lane states hold max_batch * merge windows
        const int merge = std::max(1, e->merge);
        const int want_lanes = schedule == OHW_SCHEDULE_LANES ? (int)std::min<int64_t>(e->lanes, n_batches) : 0;
        if (schedule == OHW_SCHEDULE_PIPELINE) {
          while (rc == OHW_OK && (int)e->states.size() < 2) {
            ohw_state* st = nullptr;
            rc = ohw_state_create(e->ctx, e->max_batch, &st);
            if (rc == OHW_OK) e->states.push_back(st);
          }
        }
        if (rc == OHW_OK && schedule == OHW_SCHEDULE_PIPELINE && !e->s_enc) {
          if (e->enc_cus >= total) e->enc_cus = std::max(1, total * 3 / 8);   // a smaller device: the same 3 : 5 split
          rc = ohw_stream_create(e->device, 0, e->enc_cus, &e->s_enc);
          if (rc == OHW_OK) rc = ohw_stream_create(e->device, e->enc_cus, total - e->enc_cus, &e->s_dec);
        }
        if (rc == OHW_OK && schedule == OHW_SCHEDULE_LANES) {
          // a lane's state holds what this input can put into one decode batch (cross K/V is 245.76 MB per window at large-v3)
          const int64_t per_lane = std::min<int64_t>(merge, (n_batches + want_lanes - 1) / std::max(1, want_lanes));
          const int need = (int)(e->max_batch * per_lane);
          if (!e->lane_states.empty() && e->lane_capacity < need) {            // a longer input (or another merge factor) than before
            for (ohw_state* st : e->lane_states) ohw_state_free(st);
            e->lane_states.clear();
          }
          if (e->lane_states.empty()) e->lane_capacity = need;
          while (rc == OHW_OK && (int)e->lane_states.size() < want_lanes) {
            ohw_state* st = nullptr;
            rc = ohw_state_create(e->ctx, e->lane_capacity, &st);
            if (rc == OHW_OK) {
              state_set_graph_max_batch(st, 16);      // see engine.hip: a capture on a lane waits for the other lanes' decodes
              e->lane_states.push_back(st);
            }
          }
          if (rc == OHW_OK && want_lanes >= 2 && (int)e->lane_streams.size() != want_lanes) {
            for (void* ls : e->lane_streams) (void)ohw_stream_destroy(ls);     // another lane count: other CU ranges
            e->lane_streams.clear();
            const int per = std::max(1, total / want_lanes);
            for (int i = 0; rc == OHW_OK && i < want_lanes; ++i) {
              void* ls = nullptr;
              rc = ohw_stream_create(e->device, i * per, per, &ls);
              if (rc == OHW_OK) e->lane_streams.push_back(ls);
            }
          }
        }
        if (rc != OHW_OK) {
          // no CU-masked queues (or no memory for more states) here: one batch after the other from now on
          for (size_t i = 1; i < e->states.size(); ++i) ohw_state_free(e->states[i]);
          e->states.clear();
          for (ohw_state* st : e->lane_states) ohw_state_free(st);
          e->lane_states.clear();
          for (void* ls : e->lane_streams) (void)ohw_stream_destroy(ls);
          e->lane_streams.clear();
          for (void** st : {&e->s_full, &e->s_enc, &e->s_dec}) if (*st) { (void)ohw_stream_destroy(*st); *st = nullptr; }
          e->schedule = schedule = OHW_SCHEDULE_SEQUENTIAL;
        }
      }
      if (schedule == OHW_SCHEDULE_LANES && e->lane_states.empty()) schedule = OHW_SCHEDULE_SEQUENTIAL;
      if (schedule == OHW_SCHEDULE_PIPELINE && (e->states.size() < 2 || !e->s_enc)) schedule = OHW_SCHEDULE_SEQUENTIAL;
      // audio longer than one batch: kernel variants no longer picked from a batch's row count, so a window decodes to the
      // same bits whichever batch (a short last one, a merged one) and schedule it lands in (include/ohw.h)
      struct Invariant {
        ohw_engine* e; bool on;
        void set(bool v) {
          (void)ohw_state_set_batch_invariant(e->state, v);
          for (ohw_state* st : e->states) (void)ohw_state_set_batch_invariant(st, v);
          for (ohw_state* st : e->lane_states) (void)ohw_state_set_batch_invariant(st, v);
        }
        Invariant(ohw_engine* e_, bool on_) : e(e_), on(on_) { if (on) set(true); }
        ~Invariant() { if (on) set(false); }
      } invariant(e, n_batches > 1);
      struct SharedRecording {      // every state of this transcribe reads the recording e->state holds
        ohw_engine* e; bool on;
        SharedRecording(ohw_engine* e_, bool on_) : e(e_), on(on_) {
          if (!on) return;
          for (ohw_state* st : e->states) state_share_recording(st, e->state);
          for (ohw_state* st : e->lane_states) state_share_recording(st, e->state);
        }
        ~SharedRecording() {
          if (!on) return;
          for (ohw_state* st : e->states) if (st != e->state) state_drop_recording(st);
          for (ohw_state* st : e->lane_states) state_drop_recording(st);
        }
      } shared_recording(e, rec_mel);
      // the logit bias of the engine's own state (ohw_state_set_logit_bias on ohw_engine_state(e): the engine's form of
      // whisper.cpp's logits_filter_callback) applies to every state a schedule decodes on, or to none
      {
        const float* bias = state_bias_host(e->state);
        const int V_ = e->ctx->hp.n_vocab;
        auto same = [&](ohw_state* st) {
          if (st == e->state) return;
          const float* b2 = state_bias_host(st);
          if (!bias && !b2) return;
          if (bias && b2 && std::memcmp(bias, b2, (size_t)V_ * sizeof(float)) == 0) return;
          check(ohw_state_set_logit_bias(st, bias, bias ? V_ : 0));
        };
        for (ohw_state* st : e->states) same(st);
        for (ohw_state* st : e->lane_states) same(st);
      }
      auto restore = [&] {
        for (ohw_state* st : e->states) (void)ohw_state_set_stream(st, nullptr);
        for (ohw_state* st : e->lane_states) (void)ohw_state_set_stream(st, nullptr);
        (void)ohw_state_set_stream(e->state, nullptr);
      };
      if (schedule == OHW_SCHEDULE_SEQUENTIAL) {
        Scratch sc(e->max_batch, max_tok);
        std::vector<int32_t> ns((size_t)e->max_batch);
        for (int64_t bi = 0; bi < n_batches; ++bi) {
          fill_ns(bi, ns);
          mel_windows(e->state, bi * e->max_batch, batch_of(bi), ns.data());
          check(ohw_encode(e->state, batch_of(bi)));
          decode_batch(sc, e->state, bi, ns.data());
          collect(sc, batch_of(bi));
        }
      } else if (schedule == OHW_SCHEDULE_PIPELINE) {
        // mel + encoder + cross-K/V of batch i+1 (MFMA-bound) run beside the greedy decode of batch i (HBM- and
        // latency-bound) on disjoint CUs; the first front end and the last decode have the device to themselves
        Scratch sc(e->max_batch, max_tok);
        std::vector<int32_t> ns2[2] = {std::vector<int32_t>((size_t)e->max_batch), std::vector<int32_t>((size_t)e->max_batch)};
        try {
          front(0, e->states[0], e->s_full, ns2[0]);
          void* last_front = e->s_full;
          for (int64_t bi = 0; bi < n_batches; ++bi) {
            const bool more = bi + 1 < n_batches;
            void* dstream = more ? e->s_dec : e->s_full;
            check(ohw_stream_wait(dstream, last_front));            // this batch's cross-K/V before its decode
            if (more) {
              check(ohw_stream_wait(e->s_enc, last_front));
              front(bi + 1, e->states[(bi + 1) & 1], e->s_enc, ns2[(bi + 1) & 1]);
              last_front = e->s_enc;
            }
            check(ohw_state_set_stream(e->states[bi & 1], dstream));
            decode_batch(sc, e->states[bi & 1], bi, ns2[bi & 1].data());
            collect(sc, batch_of(bi));
          }
        } catch (...) {
          (void)hipDeviceSynchronize();
          restore();
          throw;
        }
        restore();
      } else {
        // LANES: groups of up to L lanes, each lane up to `merge` batches of max_batch windows.  The lanes' front ends (one
        // pass over all of a lane's windows each) run one after the other on every CU (MFMA-bound: nothing to gain from
        // sharing the chip); a lane decodes its windows as ONE batch (the decoder streams its weights once per step whatever
        // its batch); the L decodes run side by side, each on its own CU-masked stream driven by its own host thread - a
        // decode alternates an HBM-bound kernel (cross-attention) with a chain of latency-bound ones, and several together
        // keep HBM busy (tools/decode_overlap_probe.py).  The kernels' arithmetic does not depend on the CU budget or on a
        // window's batch neighbours, so the results equal the sequential schedule's.
        const int L = std::max(1, (int)std::min(e->lane_states.size(), e->lane_streams.empty() ? (size_t)1 : e->lane_streams.size()));
        const int merge = std::max(1, std::min(e->merge, e->lane_capacity / std::max(1, e->max_batch)));
        const int64_t DBw = (int64_t)e->max_batch * merge;                  // capacity of a lane's decode batch, windows
        std::vector<std::unique_ptr<Scratch>> scs;
        std::vector<std::vector<int32_t>> nss((size_t)L, std::vector<int32_t>((size_t)DBw));
        for (int i = 0; i < L; ++i) scs.emplace_back(new Scratch((int)DBw, max_tok));
        try {
          // a group = up to L * merge front-end batches, dealt to the lanes as evenly as they go (4 batches on 4 lanes are 4
          // decode batches of one front end each, not one lane with all four)
          for (int64_t b0 = 0; b0 < n_batches;) {
            const int gb = (int)std::min<int64_t>((int64_t)L * merge, n_batches - b0);
            const int grp = std::min(L, gb);
            std::vector<int64_t> lane_w0((size_t)grp);
            std::vector<int> lane_w((size_t)grp);
            int64_t bnext = b0;
            for (int j = 0; j < grp; ++j) {
              const int mj = gb / grp + (j < gb % grp ? 1 : 0);
              lane_w0[(size_t)j] = bnext * e->max_batch;
              lane_w[(size_t)j] = (int)(std::min<int64_t>(n_win, (bnext + mj) * e->max_batch) - lane_w0[(size_t)j]);
              bnext += mj;
            }
            b0 = bnext;
            for (int j = 0; j < grp; ++j) {
              const int64_t w0 = lane_w0[(size_t)j];
              const int Wd = lane_w[(size_t)j];
              ohw_state* st = e->lane_states[(size_t)j];
              check(ohw_state_set_stream(st, e->s_full));
              // ONE front-end pass over all the lane's windows (its state holds them): the encoder's GEMMs run in rounds of
              // 256 tiles of 256 rows, and 96 windows (563 row tiles) fill their last round where 32 (188) leave it 2/3 empty
              for (int b = 0; b < Wd; ++b) nss[(size_t)j][(size_t)b] = (int32_t)std::min<int64_t>(CHUNK_SAMPLES, n - rec_win(w0 + b) * CHUNK_SAMPLES);
              mel_windows(st, w0, Wd, nss[(size_t)j].data());
              check(ohw_encode(st, Wd));
            }
            if (grp == 1) {
              decode_windows(*scs[0], e->lane_states[0], lane_w0[0], lane_w[0], nss[0].data());
              collect(*scs[0], lane_w[0]);
              continue;
            }
            std::vector<std::string> errs((size_t)grp);
            std::vector<std::thread> th;
            // the host waits for the group's front ends before the lane threads start: lanes that begin to enqueue their
            // decode while the front ends still run cost 6 % of a step in bench.py (283.9 against 262 - 268 ms)
            check(ohw_stream_sync(e->s_full));
            for (int j = 0; j < grp; ++j) {
              check(ohw_stream_wait(e->lane_streams[(size_t)j], e->s_full));
              check(ohw_state_set_stream(e->lane_states[(size_t)j], e->lane_streams[(size_t)j]));
            }
            for (int j = 0; j < grp; ++j)
              th.emplace_back([&, j] {
                try {
                  decode_windows(*scs[(size_t)j], e->lane_states[(size_t)j], lane_w0[(size_t)j], lane_w[(size_t)j], nss[(size_t)j].data());
                } catch (const std::exception& ex) {
                  errs[(size_t)j] = ex.what()[0] ? ex.what() : "unknown error";
                }
              });
            {
              ApiRelease wait_only;        // the lanes take the library's gate themselves (exclusively while one captures its graph)
              for (auto& t : th) t.join();
            }
            for (int j = 0; j < grp; ++j) if (!errs[(size_t)j].empty()) throw Error(OHW_E_TRANSCRIBE, errs[(size_t)j]);
            for (int j = 0; j < grp; ++j) {
              check(ohw_stream_wait(e->s_full, e->lane_streams[(size_t)j]));
              collect(*scs[(size_t)j], lane_w[(size_t)j]);
            }
          }
        } catch (...) {
          (void)hipDeviceSynchronize();
          restore();
          throw;
        }
        restore();
      }
    }
}

}  // namespace ohw

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

// Host-side logits filter: whisper.cpp's whisper_process_logits with the defaults the reference inherits
// (SURVEY.md A4.6, Appendix A).  Same rule order as the device sampler in decode.hip.  Masks `logits` in place
// (timestamp-mass rule included) and returns the log-sum-exp taken BEFORE that rule, as whisper.cpp's logprobs are.
static float filter_logits(const ohw_ctx* ctx, const ohw_sample_params* p, float* logits, const int32_t* cur, int n_cur) {
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
  return lse;
}

int32_t ohw_sample_greedy_host(const ohw_ctx* ctx, const ohw_sample_params* p, float* logits, const int32_t* cur, int n_cur,
                               float* logprob_out) {
  if (!ctx || !p || !logits) return -1;
  const float lse = filter_logits(ctx, p, logits, cur, n_cur);
  const int V = ctx->hp.n_vocab;
  int best = 0;
  float bv = -INFINITY;
  for (int i = 0; i < V; ++i) if (logits[i] > bv) { bv = logits[i]; best = i; }
  if (logprob_out) *logprob_out = bv - lse;
  return best;
}

// whisper.cpp's decoders draw with std::mt19937 (seeded 0 once per whisper_full call) through
// std::discrete_distribution over exp(logprobs): whisper_sample_token with best = false
struct ohw_rng { std::mt19937 gen; };
ohw_rng* ohw_rng_new(uint32_t seed) { return new (std::nothrow) ohw_rng{std::mt19937(seed)}; }
void ohw_rng_free(ohw_rng* r) { delete r; }

int32_t ohw_sample_host(const ohw_ctx* ctx, const ohw_sample_params* p, float* logits, const int32_t* cur, int n_cur, float temperature,
                        ohw_rng* rng, float* logprob_out, float* no_speech_prob_out) {
  if (!ctx || !p || !logits) return -1;
  const int V = ctx->hp.n_vocab;
  if (temperature > 0.0f) for (int i = 0; i < V; ++i) logits[i] /= temperature;
  if (no_speech_prob_out && n_cur == 0) {
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) mx = std::max(mx, logits[i]);
    double sum = 0.0;
    for (int i = 0; i < V; ++i) sum += std::exp((double)(logits[i] - mx));
    *no_speech_prob_out = (float)std::exp((double)(logits[ctx->tok.nosp] - mx) - std::log(sum));
  }
  if (!(temperature > 0.0f)) return ohw_sample_greedy_host(ctx, p, logits, cur, n_cur, logprob_out);
  if (!rng) return -1;
  const float lse = filter_logits(ctx, p, logits, cur, n_cur);
  std::vector<float> probs((size_t)V);
  for (int i = 0; i < V; ++i) probs[(size_t)i] = logits[i] > -INFINITY ? std::exp(logits[i] - lse) : 0.0f;
  std::discrete_distribution<> dist(probs.begin(), probs.end());
  const int id = dist(rng->gen);
  if (logprob_out) *logprob_out = logits[id] - lse;
  return id;
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
    ohw_ctx* ctx = nullptr;
    int rc = ohw_ctx_create(model_path, device, dtype, &ctx);
    if (rc != OHW_OK) throw Error(rc == OHW_E_MODEL_NOT_FOUND ? rc : (rc == OHW_E_NO_GPU || rc == OHW_E_OOM ? rc : OHW_E_LOAD_FAILED),
                                  "Failed to load model: " + g_last_error);
    std::unique_ptr<ohw_engine> e;
    try {
      e.reset(engine_wrap_ctx(ctx, lang, translate != 0, max_batch, device));
    } catch (...) {
      ohw_ctx_free(ctx);
      throw;
    }
    *out = e.release();
  });
}

void ohw_engine_free(ohw_engine* e) {
  if (!e) return;
  ohw_state_free(e->state);
  for (size_t i = 1; i < e->states.size(); ++i) ohw_state_free(e->states[i]);
  for (ohw_state* st : e->lane_states) ohw_state_free(st);
  for (void* ls : e->lane_streams) (void)ohw_stream_destroy(ls);
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
    std::string text;
    engine_transcribe_core(e, samples, n, &text);
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
      const std::string lang = e->language == "auto" ? ohw_lang_id_to_code(0) : e->language;
      std::strncpy(language_out, lang.c_str(), 7);
      language_out[7] = 0;
    }
    if (duration_ms)
      *duration_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  });
}

void ohw_default_decode_policy(ohw_decode_policy* q) {
  if (!q) return;
  q->temperature_inc = 0.2f; q->entropy_thold = 2.4f; q->logprob_thold = -1.0f; q->no_speech_thold = 0.6f;
}

int ohw_engine_set_decode_policy(ohw_engine* e, const ohw_decode_policy* q) {
  if (!e || !q) return OHW_E_INVALID_ARG;
  e->policy = *q;
  return OHW_OK;
}

int ohw_engine_last_trace(ohw_engine* e, const int32_t** data, int* n) {
  if (!e || !data || !n) return OHW_E_INVALID_ARG;
  *data = e->last_trace.data();
  *n = (int)e->last_trace.size();
  return OHW_OK;
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

int ohw_engine_set_schedule(ohw_engine* e, int schedule, int lanes, int merge) {
  if (!e || schedule < OHW_SCHEDULE_SEQUENTIAL || schedule > OHW_SCHEDULE_LANES || lanes < 0 || lanes > 16 || merge < 0 || merge > 8) return OHW_E_INVALID_ARG;
  e->schedule = schedule;
  if (lanes > 0) e->lanes = lanes;
  if (merge > 0) e->merge = std::min(merge, std::max(1, 256 / e->max_batch));
  return OHW_OK;
}

int ohw_engine_set_force_len(ohw_engine* e, int n_tokens) {
  if (!e || n_tokens < 0) return OHW_E_INVALID_ARG;
  e->force_len = n_tokens;
  return OHW_OK;
}

int ohw_engine_set_window_mode(ohw_engine* e, int mode) {
  if (!e || (mode != OHW_WINDOW_FIXED && mode != OHW_WINDOW_SEEK && mode != OHW_WINDOW_FIXED_RECORDING_MEL)) return OHW_E_INVALID_ARG;
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
