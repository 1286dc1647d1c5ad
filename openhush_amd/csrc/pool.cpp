// pool.cpp — the multi-GPU pool BEHIND the C ABI: one engine per device of one node, one host thread per
// device, the fixed 30 s windows of a recording dealt round-robin, results gathered on the host in recording order.
//
// The reference has no multi-GPU code; it has the intent: requirement F14 "distribute transcriptions across multiple GPUs
// on a single machine" (reference REQUIREMENTS.md:26), the sketch "GPU pool preloads models, scheduler assigns jobs
// round-robin ... dedicated threads for GPU compute" (CLAUDE.md:64-65) and the config keys `[gpu] auto_detect / devices`
// that today do nothing (src/config.rs:921-929).  A Rust host that links libohw.so gets that pool from here.
//
// Model load: the file is read and repacked ONCE, on device_ids[0]; its resident weight arena (3.1 GB at large-v3) reaches
// the other devices in ONE broadcast over xGMI - RCCL (ncclCommInitAll + ncclBroadcast, loaded at run time: libohw.so has no
// link-time dependency on librccl) or, when RCCL cannot be loaded or a device is listed twice, peer-to-peer copies.
// No collective in the data path: windows are independent.
#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "host_engine.hpp"
#include "model.hpp"

namespace ohw {
extern thread_local std::string g_last_error;
}
using namespace ohw;

struct ohw_pool {
  std::vector<ohw_engine*> engines;    // engines[0] owns the context that read the file (and the vocabulary)
  std::vector<int> devices;
  std::string language;
  std::string last_text, broadcast;    // broadcast: "none" | "rccl" | "peer"
  std::string broadcast_note;          // why RCCL was given up, if it was
  std::vector<int32_t> last_tokens;
  std::vector<ohw_window_quality> last_quality;
};

namespace {

// no ApiScope here: the pool only calls C-ABI entries (each takes the gate itself) and waits for its worker threads - a
// caller holding the gate while a worker needs it exclusively (graph capture) would never be released
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

// RCCL entry points, resolved at run time
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
    Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
    GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    return CommInitAll && Broadcast && GroupStart && GroupEnd && CommDestroy;
  }
};

// every resident weight buffer of ctxs[i] against ctxs[0] (64-bit digests, a few ms per device): a broadcast that returned
// success but left other bytes on a device would otherwise surface as garbage transcripts on that device only
void verify_replicas(const std::vector<ohw_ctx*>& ctxs, const std::vector<int>& devs, const char* kind) {
  char name[64];
  for (int idx = 0;; ++idx) {
    uint64_t d0 = 0;
    if (ohw_ctx_weight_digest(ctxs[0], idx, name, &d0) != OHW_OK) break;        // past the last buffer
    for (size_t i = 1; i < ctxs.size(); ++i) {
      uint64_t di = 0;
      char nm[64];
      if (ohw_ctx_weight_digest(ctxs[i], idx, nm, &di) != OHW_OK || di != d0)
        throw Error(OHW_E_LOAD_FAILED, std::string("pool: device ") + std::to_string(devs[i]) + " holds other bytes than device " + std::to_string(devs[0]) +
                                           " in weight buffer '" + name + "' after the " + kind + " broadcast");
    }
  }
}

// RCCL communicators of one ncclCommInitAll, destroyed on every path out
struct Comms {
  Rccl& r;
  std::vector<ncclComm_t> c;
  const std::vector<int>& devs;
  bool live = false;
  Comms(Rccl& r_, const std::vector<int>& d) : r(r_), c(d.size(), nullptr), devs(d) {}
  ~Comms() {
    if (!live) return;
    for (size_t i = 0; i < c.size(); ++i) {
      (void)hipSetDevice(devs[i]);
      (void)hipDeviceSynchronize();
      if (c[i]) (void)r.CommDestroy(c[i]);
    }
  }
};

// One grouped broadcast of `bytes` from ctxs[0]'s buffer into the same buffer of every other context; false on ANY error
// (RCCL's or HIP's) with the group closed again - the caller then falls back to peer copies.
// NOTE: this branch has never executed (no box with two distinct devices was available to any round): it is written to the
// RCCL documentation's single-process pattern and guarded by the fallback and by verify_replicas.
bool rccl_broadcast(Rccl& r, Comms& cm, const std::vector<ohw_ctx*>& ctxs, int part) {
  const int n = (int)ctxs.size();
  if (r.GroupStart() != ncclSuccess) return false;
  bool ok = true;
  for (int i = 0; i < n && ok; ++i) {
    const DevBuf& src = part == 0 ? ctxs[0]->arena : ctxs[0]->mel_filters;
    const DevBuf& dst = part == 0 ? ctxs[(size_t)i]->arena : ctxs[(size_t)i]->mel_filters;
    ok = hipSetDevice(cm.devs[(size_t)i]) == hipSuccess &&
         r.Broadcast(i == 0 ? src.p : dst.p, dst.p, src.bytes, ncclUint8, 0, cm.c[(size_t)i], nullptr) == ncclSuccess;
  }
  const bool closed = r.GroupEnd() == ncclSuccess;      // always: an open group would swallow every later RCCL call
  return ok && closed;
}

void peer_copies(const std::vector<ohw_ctx*>& ctxs, const std::vector<int>& devs) {
  const int n = (int)ctxs.size();
  for (int i = 1; i < n; ++i) {
    HIP_CHECK(hipSetDevice(devs[(size_t)i]));
    if (devs[(size_t)i] != devs[0]) {
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, devs[(size_t)i], devs[0]);
      if (can) { const hipError_t e = hipDeviceEnablePeerAccess(devs[0], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e); (void)hipGetLastError(); }
    }
    HIP_CHECK(hipMemcpyPeer(ctxs[(size_t)i]->arena.p, devs[(size_t)i], ctxs[0]->arena.p, devs[0], ctxs[0]->arena.bytes));
    HIP_CHECK(hipMemcpyPeer(ctxs[(size_t)i]->mel_filters.p, devs[(size_t)i], ctxs[0]->mel_filters.p, devs[0], ctxs[0]->mel_filters.bytes));
    HIP_CHECK(hipDeviceSynchronize());
  }
}

// the weight arena and the filterbank of ctxs[0] into every other context: RCCL when the devices are distinct and the library
// loads; on ANY RCCL failure - and for a device listed twice - peer-to-peer copies; either way every replica is verified
std::string broadcast_weights(const std::vector<ohw_ctx*>& ctxs, const std::vector<int>& devs, std::string* note) {
  ApiScope api;
  const int n = (int)ctxs.size();
  if (n == 1) return "none";
  const bool distinct = std::set<int>(devs.begin(), devs.end()).size() == devs.size();
  const char* mode = getenv("OHW_POOL_BCAST");
  Rccl r;
  if (distinct && !(mode && std::string(mode) == "peer") && r.load()) {
    bool ok = false;
    {
      Comms cm(r, devs);
      if (r.CommInitAll(cm.c.data(), n, devs.data()) == ncclSuccess) {
        cm.live = true;
        ok = rccl_broadcast(r, cm, ctxs, 0) && rccl_broadcast(r, cm, ctxs, 1);
      }
    }                                                  // communicators synchronised and destroyed here, whatever happened
    (void)hipGetLastError();
    if (ok) {
      try {
        verify_replicas(ctxs, devs, "rccl");
        return "rccl";
      } catch (const Error& e) {
        if (note) *note = e.what();                    // wrong bytes after a "successful" broadcast: try the copies
      }
    } else if (note) {
      *note = "RCCL broadcast failed, fell back to peer copies";
    }
  }
  peer_copies(ctxs, devs);
  verify_replicas(ctxs, devs, "peer");
  return "peer";
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// the pool around a first context made by `make_first` on device_ids[0] (a model file read, or procedural weights)
template <typename MakeFirst>
ohw_pool* pool_build(const char* language, int translate, const int* device_ids, int n_devices, int dtype, int max_batch, MakeFirst&& make_first) {
  if (!device_ids || n_devices < 1 || n_devices > 64) throw Error(OHW_E_INVALID_ARG, "pool: device_ids / n_devices");
  const std::string lang = language ? language : "auto";
  if (lang != "auto" && ohw_lang_code_to_id(lang.c_str()) < 0) throw Error(OHW_E_LOAD_FAILED, "unknown language code '" + lang + "'");
  std::unique_ptr<ohw_pool> p(new ohw_pool());
  p->language = lang;
  p->devices.assign(device_ids, device_ids + n_devices);
  std::vector<ohw_ctx*> ctxs((size_t)n_devices, nullptr);
  auto cleanup = [&] {
    for (ohw_engine* e : p->engines) ohw_engine_free(e);
    p->engines.clear();
    for (ohw_ctx* c : ctxs) if (c) ohw_ctx_free(c);
  };
  try {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;
    for (int i = 0; i < n_devices; ++i)            // before the 3 GB file read: a bad id names itself
      if (device_ids[i] < 0 || device_ids[i] >= n_dev)
        throw Error(OHW_E_NO_GPU, "pool: device " + std::to_string(device_ids[i]) + " (entry " + std::to_string(i) + " of the device list) does not exist: " +
                                      std::to_string(n_dev) + " device(s) visible");
    int rc = make_first(&ctxs[0]);                  // the one file read
    if (rc != OHW_OK) throw Error(rc == OHW_E_NO_GPU || rc == OHW_E_OOM ? rc : OHW_E_LOAD_FAILED, "pool: device " + std::to_string(device_ids[0]) + ": Failed to load model: " + g_last_error);
    ohw_hparams hp;
    (void)ohw_ctx_info(ctxs[0], &hp, nullptr);
    dtype = ohw_ctx_dtype(ctxs[0]);                       // what OHW_DTYPE_AUTO resolved to
    for (int i = 1; i < n_devices; ++i) {
      rc = ohw_ctx_create_shell(&hp, device_ids[i], dtype, &ctxs[(size_t)i]);
      if (rc != OHW_OK) throw Error(rc, "pool: device " + std::to_string(device_ids[i]) + " (entry " + std::to_string(i) + " of the device list): " + g_last_error);
    }
    try {
      p->broadcast = broadcast_weights(ctxs, p->devices, &p->broadcast_note);
    } catch (const Error& e) {
      throw Error(e.code, std::string("pool: weight broadcast: ") + e.what());
    }
    for (int i = 0; i < n_devices; ++i) {
      ohw_engine* e = engine_wrap_ctx(ctxs[(size_t)i], lang, translate != 0, max_batch, device_ids[i]);
      ctxs[(size_t)i] = nullptr;                 // owned by the engine now
      p->engines.push_back(e);
    }
  } catch (...) {
    cleanup();
    throw;
  }
  return p.release();
}
}  // namespace

extern "C" {

int ohw_pool_create(const char* model_path, const char* language, int translate, const int* device_ids, int n_devices, int dtype,
                    int max_batch, ohw_pool** out) {
  return guard([&] {
    if (!out) throw Error(OHW_E_INVALID_ARG, "out is null");
    *out = nullptr;
    struct stat sb;
    if (!model_path || stat(model_path, &sb) != 0)
      throw Error(OHW_E_MODEL_NOT_FOUND, std::string("Model not found at ") + (model_path ? model_path : "(null)"));
    *out = pool_build(language, translate, device_ids, n_devices, dtype, max_batch,
                      [&](ohw_ctx** c) { return ohw_ctx_create(model_path, device_ids[0], dtype, c); });
  });
}

int ohw_pool_create_synthetic(const ohw_hparams* hp, uint32_t seed, const char* language, int translate, const int* device_ids, int n_devices,
                              int dtype, int max_batch, ohw_pool** out) {
  return guard([&] {
    if (!out || !hp) throw Error(OHW_E_INVALID_ARG, "out / hp is null");
    *out = nullptr;
    *out = pool_build(language, translate, device_ids, n_devices, dtype == OHW_DTYPE_AUTO ? OHW_DTYPE_BF16 : dtype, max_batch,
                      [&](ohw_ctx** c) { return ohw_ctx_create_synthetic(hp, seed, device_ids[0], dtype == OHW_DTYPE_AUTO ? OHW_DTYPE_BF16 : dtype, c); });
  });
}

void ohw_pool_free(ohw_pool* p) {
  if (!p) return;
  for (ohw_engine* e : p->engines) ohw_engine_free(e);
  delete p;
}

int ohw_pool_n_devices(const ohw_pool* p) { return p ? (int)p->engines.size() : 0; }
const char* ohw_pool_broadcast_kind(const ohw_pool* p) { return p ? p->broadcast.c_str() : ""; }
ohw_engine* ohw_pool_engine(ohw_pool* p, int i) { return (p && i >= 0 && i < (int)p->engines.size()) ? p->engines[(size_t)i] : nullptr; }

int ohw_pool_set_window_mode(ohw_pool* p, int mode) {
  if (!p) return OHW_E_INVALID_ARG;
  for (ohw_engine* e : p->engines) {
    const int rc = ohw_engine_set_window_mode(e, mode);
    if (rc != OHW_OK) return rc;
  }
  return OHW_OK;
}
const char* ohw_pool_broadcast_note(const ohw_pool* p) { return p ? p->broadcast_note.c_str() : ""; }
int ohw_pool_set_force_len(ohw_pool* p, int n_tokens) {
  if (!p) return OHW_E_INVALID_ARG;
  for (ohw_engine* e : p->engines) {
    const int rc = ohw_engine_set_force_len(e, n_tokens);
    if (rc != OHW_OK) return rc;
  }
  return OHW_OK;
}
int ohw_pool_set_schedule(ohw_pool* p, int schedule, int lanes, int merge) {
  if (!p) return OHW_E_INVALID_ARG;
  for (ohw_engine* e : p->engines) {
    const int rc = ohw_engine_set_schedule(e, schedule, lanes, merge);
    if (rc != OHW_OK) return rc;
  }
  return OHW_OK;
}

int ohw_pool_set_decode_policy(ohw_pool* p, const ohw_decode_policy* q) {
  if (!p || !q) return OHW_E_INVALID_ARG;
  for (ohw_engine* e : p->engines) (void)ohw_engine_set_decode_policy(e, q);
  return OHW_OK;
}

int ohw_pool_transcribe(ohw_pool* p, const float* samples, int64_t n, uint32_t sample_rate, char* text_buf, size_t text_cap,
                        char* language_out, uint64_t* duration_ms, ohw_audio_info* info_out) {
  return guard([&] {
    if (!p) throw Error(OHW_E_INVALID_ARG, "pool is null");
    ohw_audio_info info;
    const int vrc = ohw_validate_audio(samples, n, sample_rate, &info);
    if (info_out) *info_out = info;
    if (vrc != OHW_OK) {
      static const char* const names[] = {"ok", "Audio is empty (no samples)", "Unexpected sample rate", "Audio too long", "Audio too short",
                                          "Audio contains NaN values", "Audio contains infinite values"};
      throw Error(OHW_E_VALIDATION, std::string("Audio validation failed: ") + names[info.error]);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int G = (int)p->engines.size();
    const int64_t n_win = (n + CHUNK_SAMPLES - 1) / CHUNK_SAMPLES;
    const int mode = p->engines[0]->window_mode;
    for (ohw_engine* e : p->engines)
      if (e->window_mode != mode) throw Error(OHW_E_INVALID_ARG, "pool: the engines disagree on the window mode (use ohw_pool_set_window_mode)");
    // window w -> device w % G.  Every engine is handed the WHOLE recording (borrowed, not copied) and the arithmetic
    // progression of windows it owns (engine_transcribe_core: first g, step G): its windows are cut at the recording's own
    // 30 s marks - and in FIXED_RECORDING_MEL from the spectrogram of the whole recording - exactly as a single engine cuts them
    std::vector<std::string> errs((size_t)G);
    std::vector<std::thread> th;
    const bool seek = mode == OHW_WINDOW_SEEK;     // the seek loop is sequential by nature: device 0 alone
    const int used = seek ? 1 : (int)std::min<int64_t>(G, n_win);
    for (int g = 0; g < used; ++g) {
      th.emplace_back([&, g] {
        try {
          std::string text;
          if (seek) engine_transcribe_core(p->engines[(size_t)g], samples, n, &text);
          else engine_transcribe_core(p->engines[(size_t)g], samples, n, &text, g, used);
        } catch (const std::exception& ex) {
          errs[(size_t)g] = ex.what()[0] ? ex.what() : "unknown error";
        }
      });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < used; ++g)
      if (!errs[(size_t)g].empty()) throw Error(OHW_E_TRANSCRIBE, "Transcription failed on device " + std::to_string(p->devices[(size_t)g]) + ": " + errs[(size_t)g]);
    // gather in recording order: device g's k-th window record is window g + k * G
    p->last_tokens.clear();
    p->last_quality.clear();
    std::string text;
    std::vector<size_t> tok_pos((size_t)G, 0), win_pos((size_t)G, 0);
    const int64_t n_rec = seek ? (int64_t)p->engines[0]->last_quality.size() : n_win;
    for (int64_t w = 0; w < n_rec; ++w) {
      const int g = seek ? 0 : (int)(w % used);
      ohw_engine* e = p->engines[(size_t)g];
      if (win_pos[(size_t)g] >= e->last_quality.size()) throw Error(OHW_E_TRANSCRIBE, "pool: a device returned fewer windows than it was dealt");
      const ohw_window_quality& q = e->last_quality[win_pos[(size_t)g]++];
      p->last_quality.push_back(q);
      for (int i = 0; i < q.n_tokens; ++i) {
        const int32_t t = e->last_tokens[tok_pos[(size_t)g]++];
        p->last_tokens.push_back(t);
        if (t < p->engines[0]->ctx->tok.eot) {
          const char* sp = nullptr;
          const int len = ohw_token_text(p->engines[0]->ctx, t, &sp);     // device 0 read the file: it has the vocabulary
          text.append(sp, (size_t)len);
        }
      }
    }
    const size_t b0 = text.find_first_not_of(" \t\r\n");
    const size_t b1 = text.find_last_not_of(" \t\r\n");
    text = b0 == std::string::npos ? std::string() : text.substr(b0, b1 - b0 + 1);
    p->last_text = text;
    if (text_buf && text_cap > 0) {
      const size_t nc = std::min(text.size(), text_cap - 1);
      std::memcpy(text_buf, text.data(), nc);
      text_buf[nc] = 0;
    }
    if (language_out) {
      const std::string lang = p->language == "auto" ? ohw_lang_id_to_code(0) : p->language;
      std::strncpy(language_out, lang.c_str(), 7);
      language_out[7] = 0;
    }
    if (duration_ms) *duration_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  });
}

int ohw_pool_last_text(ohw_pool* p, const char** text, size_t* len) {
  if (!p || !text) return OHW_E_INVALID_ARG;
  *text = p->last_text.c_str();
  if (len) *len = p->last_text.size();
  return OHW_OK;
}
int ohw_pool_last_tokens(ohw_pool* p, const int32_t** tokens, int* n) {
  if (!p || !tokens || !n) return OHW_E_INVALID_ARG;
  *tokens = p->last_tokens.data();
  *n = (int)p->last_tokens.size();
  return OHW_OK;
}
int ohw_pool_last_quality(ohw_pool* p, const ohw_window_quality** q, int* n_windows) {
  if (!p || !q || !n_windows) return OHW_E_INVALID_ARG;
  *q = p->last_quality.data();
  *n_windows = (int)p->last_quality.size();
  return OHW_OK;
}

}  // extern "C"
