// engine.hip — ohw_state: activation buffers, KV caches and the stage drivers behind the C ABI
// (mel -> encoder -> cross K/V -> decoder steps -> device-side greedy loop).
//
// Replaces ctx.create_state() and the arithmetic inside state.full()
// (reference src/engine/whisper.rs:167-169, 266-268).
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <type_traits>
#include <vector>
#include <algorithm>

#include "attention.hpp"
#include "gemm.hpp"
#include "kernels.hpp"
#include "model.hpp"

namespace ohw {
ohw_ctx* ctx_from_file(const char* path, int device, int dtype);
ohw_ctx* ctx_synthetic(const ohw_hparams* hp, uint32_t seed, int device, int dtype);
ohw_ctx* ctx_shell(const ohw_hparams* hp, int device, int dtype);

thread_local std::string g_last_error;

// OHW_SEGV_TRACE=1 (diagnostics): print the native frames of a SIGSEGV before the default action takes the process down -
// the GPU debugger is not available on the pool, and a fault under a profiler or inside the runtime otherwise leaves nothing
namespace {
void segv_trace(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  static const char msg[] = "\n[libohw] SIGSEGV, native frames:\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct SegvTraceInit {
  SegvTraceInit() {
    const char* e = getenv("OHW_SEGV_TRACE");
    if (e && *e == '1') signal(SIGSEGV, segv_trace);
  }
} g_segv_trace_init;
}  // namespace

static std::shared_mutex g_api_mu;
static thread_local int g_api_depth = 0;
ApiScope::ApiScope() { if (g_api_depth++ == 0) g_api_mu.lock_shared(); }
ApiScope::~ApiScope() { if (--g_api_depth == 0) g_api_mu.unlock_shared(); }
ApiRelease::ApiRelease() { if (g_api_depth > 0) g_api_mu.unlock_shared(); }
ApiRelease::~ApiRelease() { if (g_api_depth > 0) g_api_mu.lock_shared(); }
CaptureGate::CaptureGate() { g_api_mu.unlock_shared(); g_api_mu.lock(); }
CaptureGate::~CaptureGate() { g_api_mu.unlock(); g_api_mu.lock_shared(); }

template <typename F>
static int guard(F&& f) {
  ApiScope api;
  try {
    f();
    return OHW_OK;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "host allocation failed";
    return OHW_E_OOM;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return OHW_E_TRANSCRIBE;
  } catch (...) {
    g_last_error = "unknown error";
    return OHW_E_TRANSCRIBE;
  }
}
}  // namespace ohw

using namespace ohw;

constexpr int DEC_KSPLIT_MAX = 8;   // most K-slices per output tile of a decoder RESID GEMM
static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const int v = atoi(e);
  return v < lo ? lo : v > hi ? hi : v;
}
// K-slices per output tile of the decoder's RESID GEMMs, read when a state is created (1 = no split, the default:
// with activation tiles mlp.2 takes 7.8 us unsplit, 7.6 us over 2 slices and 9.4 us over 4 - the hand-off costs
// about 3.5 us - so the split path is kept as a tuning knob for other shapes, exercised by the GPU tests)
static int dec_ksplit_long() { return env_int("OHW_DEC_KSPLIT_LONG", 1, 1, DEC_KSPLIT_MAX); }
static int dec_ksplit_short() { return env_int("OHW_DEC_KSPLIT_SHORT", 1, 1, DEC_KSPLIT_MAX); }


struct ohw_state {
  ohw_ctx* ctx = nullptr;
  int max_batch = 0;
  int enc_batch = 0;  // windows of the decode batch the cross K/V holds (the last ohw_encode, or the total of its slices)
  int mel_batch = 0;  // windows of the last ohw_mel
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // front end
  DevBuf pcm, n_samples, logmel, max_bits, mel_t;
  DevBuf rec_pcm, rec_max, rec_off;   // a whole recording, the maximum of its log-mel spectrogram, window offsets (ohw_recording_set)
  int64_t rec_n = 0;
  // encoder activations
  DevBuf c1, h, y, qkv, att, ffn, enc;
  // taps kept for diagnostics (small models / tests only)
  DevBuf tap_stem, tap_block0;
  bool taps = false;
  // decoder
  DevBuf xkv;      // T [2L][B][H][1500][64]
  DevBuf self_kv;  // T [L][2][B][H][n_text_ctx][64]
  DevBuf dx, dy, dq, da, df, logits;
  DevBuf dx16, xstat;            // post-norm path: 16-bit tiled copy of the residual stream, per-16-column statistics
  bool postnorm = true;
  DevBuf ks_slab, ks_ticket;   // split-K partial tiles and arrival tickets of the decoder's RESID GEMMs
  int ksplit_long = 1, ksplit_short = 1;
  int stream_cus = 0;            // CUs of the current stream's mask (0 = unrestricted)
  bool skip_done = false;        // inside ohw_greedy: cross-attention skips windows whose done flag is set
  DevBuf samp_part, samp_ticket;   // sampler: per-slice partial states and arrival tickets
  DevBuf xa_part, xa_ticket;       // cross-attention over key slices (small batches)
  int xa_rows = 0;
  DevBuf step_tok, n_past, tokens, n_cur, next_tok, done, n_done, sum_lp;
  // beam search (made on first use): candidates, cumulative scores, the kv_slot / token-history double buffers, finished pool
  DevBuf bm_cand_lp, bm_cand_tok, bm_sum, bm_slot[2], bm_tok2, bm_ncur, bm_npast, bm_done, bm_fin_cnt, bm_fin_tok, bm_fin_len, bm_fin_sum, bm_part, bm_ticket;
  // one entry = the PAIR of graphs of a (windows, beam size, sampler parameters, CU budget, cross-attention variant) key: the
  // odd and the even iteration (the token-history and kv_slot double buffers alternate); made, looked up and evicted together,
  // so a call never holds an exec of an entry it then evicts
  struct BeamGraph { hipGraph_t graph[2] = {nullptr, nullptr}; hipGraphExec_t exec[2] = {nullptr, nullptr}; int windows = 0, K = 0, cus = 0; bool invariant = false, persist = false; SamplerParams spar; };
  std::vector<BeamGraph> beam_graphs;
  // the persistent small-batch decoder step (decode_persist.hip): per-layer pointer table, granule arena, epoch / abort words
  DevBuf ps_layers, ps_gran, ps_words;
  PersistParams ps_layout{};         // region offsets of the arena
  bool persist = false;              // OHW_DEC_PERSIST / ohw_state_set_persistent (off: measured slower than the launches)
  bool fuse_attn = false;            // OHW_DEC_FUSE_ATTN=1: small single-token steps run their self-attention inside the QKV launch (measured slower: off)
  DevBuf attn_ticket;                // u32 [n_text_head], zero between launches
  int persist_launches = 0;
  int n_cu = 0;
  int step_captures = 0;
  int beam_captures = 0;            // graph pairs captured so far (ohw_dbg_counter: a second call with the same key adds none)
  DevBuf tok_lp, nosp_prob;        // per-token log-probabilities [B][max_tokens + 1], no-speech probability [B]
  DevBuf logit_bias;               // optional f32 [n_vocab] (ohw_state_set_logit_bias)
  std::vector<float> bias_host;    // the same on the host: the temperature ladder samples there (host_engine.cpp)
  bool bias_on = false;
  int m_max = 0;
  int64_t logits_ld = 0;
  // per-kernel-class profiling (bench): event pairs around every launch of one class
  int prof_class = 0;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  double prof_work = 0.0;
  // hipGraph of one greedy iteration {feed sampled token, single-token decoder step, sampler}
  // captured greedy iterations, one per (batch, sampler parameters, CU budget of the stream) seen; a handful at most
  struct StepGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int batch = 0; int cus = 0; bool invariant = false, persist = false; SamplerParams spar{}; };
  bool batch_invariant = false;      // cross-attention variant picked from n_new alone (state_set_batch_invariant)
  std::vector<StepGraph> step_graphs;
  bool graphs_enabled = true;
  int graph_max_batch = 32;         // graphs for batches below this (OHW_GRAPH_MAX_BATCH)
  // timing
  hipEvent_t ev[6]{};
  ohw_timings last{};
  int max_tokens = 0;
};

namespace {

struct ProfScope {
  ohw_state* st;
  bool on;
  ProfScope(ohw_state* s, int cls, double work) : st(s), on(s->prof_class == cls) {
    if (!on) return;
    if (st->prof_used + 2 > st->prof_ev.size()) {
      const size_t old = st->prof_ev.size();
      st->prof_ev.resize(old + 1024);
      for (size_t i = old; i < st->prof_ev.size(); ++i) HIP_CHECK(hipEventCreateWithFlags(&st->prof_ev[i], hipEventDisableSystemFence));
    }
    HIP_CHECK(hipEventRecord(st->prof_ev[st->prof_used], st->stream));
    st->prof_work += work;
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(st->prof_ev[st->prof_used + 1], st->stream);
    st->prof_used += 2;
  }
};

struct Dispatch {
  template <typename F> static void run(int dtype, F&& f) {
    if (dtype == OHW_DTYPE_BF16) f((bf16_t*)nullptr);
    else f((f16_t*)nullptr);
  }
};

void persist_prepare(ohw_state* st);

void state_alloc(ohw_state* st) {
  const ohw_ctx* c = st->ctx;
  const ohw_hparams& hp = c->hp;
  const int64_t B = st->max_batch, d = hp.n_audio_state, dt = hp.n_text_state, T = hp.n_audio_ctx;
  const int64_t L = hp.n_text_layer, H = hp.n_text_head;
  st->pcm.alloc((size_t)B * CHUNK_SAMPLES * 4);
  st->n_samples.alloc((size_t)B * 4);
  st->logmel.alloc((size_t)B * hp.n_mels * CHUNK_FRAMES * 4);
  st->max_bits.alloc((size_t)B * 4);
  st->mel_t.alloc((size_t)B * MEL_ROWS * MEL_CPAD * 2, true);
  st->c1.alloc((size_t)B * MEL_ROWS * d * 2 + 4 * d * 2, true);  // + slack: the last conv2 row reads 1 row past
  st->h.alloc((size_t)B * T * d * 4);
  st->y.alloc((size_t)B * T * d * 2);
  st->qkv.alloc((size_t)B * T * 3 * d * 2);
  st->att.alloc((size_t)B * T * d * 2);
  st->ffn.alloc((size_t)B * T * 4 * d * 2);
  st->enc.alloc((size_t)B * T * d * 2);
  st->taps = d <= 512;
  if (st->taps) {
    st->tap_stem.alloc((size_t)B * T * d * 4);
    st->tap_block0.alloc((size_t)B * T * d * 4);
  }
  st->xkv.alloc((size_t)2 * L * B * H * T * 64 * 2);
  st->self_kv.alloc((size_t)L * 2 * B * H * hp.n_text_ctx * 64 * 2, true);
  st->m_max = (int)B * 8;
  st->dx.alloc((size_t)st->m_max * dt * 4);
  // dy / da / df are read as 16-row activation tiles by 32-row workgroups: whole tiles, zeroed once (rows past M are
  // multiplied but never stored)
  const size_t m_tiles = ((size_t)st->m_max + 31) / 32 * 32;
  st->dy.alloc(m_tiles * dt * 2, true);
  // post-norm decoder GEMMs (decode.hip): OHW_DEC_POSTNORM=1 (off by default: measured, no gain - the LayerNorm prologue
  // already hides under the weights' first-byte latency); never with a split-K knob (the split path
  // publishes no statistics); dt must be a multiple of 32
  st->graphs_enabled = env_int("OHW_GRAPHS", 1, 0, 1) != 0;      // 0: the decode iterations are launched kernel by kernel (diagnostics)
  // the greedy iteration is replayed as a hipGraph below this batch size only: a graph pays while the step is launch-bound
  // (one window: 260 launches of 3 - 6 us); from 32 windows on a step's kernels outlast their launches (measured equal,
  // 371.0 ms per 32-window batch either way) and in the LANES schedule the replayed graphs lose 2.4 % to sporadic 40-us
  // stalls inside a replay (tools/lane_gap_analysis.py) - and a lane's first call no longer waits, at its capture, for the
  // other lanes to leave the library
  st->graph_max_batch = env_int("OHW_GRAPH_MAX_BATCH", 32, 1, 1 << 20);
  st->persist = env_int("OHW_DEC_PERSIST", 0, 0, 1) != 0;
  st->fuse_attn = env_int("OHW_DEC_FUSE_ATTN", 0, 0, 1) != 0;
  st->attn_ticket.alloc((size_t)hp.n_text_head * 4, true);
  (void)hipDeviceGetAttribute(&st->n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
  st->postnorm = env_int("OHW_DEC_POSTNORM", 0, 0, 1) != 0 && dec_ksplit_long() == 1 && dec_ksplit_short() == 1 && dt % 32 == 0;
  st->dx16.alloc(m_tiles * dt * 2, true);
  st->xstat.alloc((size_t)st->m_max * (dt / 16) * 2 * 4, true);
  st->dq.alloc((size_t)st->m_max * dt * 2);
  st->da.alloc(m_tiles * dt * 2, true);
  st->df.alloc(m_tiles * 4 * dt * 2, true);
  {
    const size_t tiles = (size_t)((dt + 15) / 16) * ((st->m_max + 31) / 32);
    st->ks_slab.alloc(tiles * DEC_KSPLIT_MAX * 2048);
    st->ks_ticket.alloc(tiles * 4, true);
    st->samp_part.alloc((size_t)B * SAMPLER_SPLIT * SAMPLER_PART_WORDS * 4, true);
    st->samp_ticket.alloc((size_t)B * 4, true);
    st->xa_rows = std::min(st->m_max, 24);       // 24 rows x 20 heads is already two (row, head) pairs per CU
    st->xa_part.alloc((size_t)st->xa_rows * H * XA_MAX_SPLIT * 68 * 4, true);
    st->xa_ticket.alloc((size_t)st->xa_rows * H * 4, true);
    st->ksplit_long = dec_ksplit_long();
    st->ksplit_short = dec_ksplit_short();
  }
  st->logits_ld = c->v_pad;
  st->logits.alloc((size_t)B * st->logits_ld * 4);
  st->max_tokens = hp.n_text_ctx;
  st->step_tok.alloc((size_t)st->m_max * 4);
  st->n_past.alloc((size_t)B * 4, true);
  st->tokens.alloc((size_t)B * st->max_tokens * 4, true);
  st->n_cur.alloc((size_t)B * 4, true);
  st->next_tok.alloc((size_t)B * 4, true);
  st->done.alloc((size_t)B * 4, true);
  st->n_done.alloc(16, true);
  st->sum_lp.alloc((size_t)B * 4, true);
  st->tok_lp.alloc((size_t)B * (st->max_tokens + 1) * 4, true);
  st->nosp_prob.alloc((size_t)B * 4, true);
  for (auto& e : st->ev) HIP_CHECK(hipEventCreate(&e));
  persist_prepare(st);
}

// the persistent decoder step's granule arena, epoch / abort words and per-layer pointer table (decode_persist.hip): made with
// the state, never lazily - a first use inside a graph capture must not allocate or memset
bool persist_dims_ok(const ohw_hparams& hp) {
  return hp.n_text_state % 64 == 0 && hp.n_text_state <= 1280 && hp.n_text_head * 64 == hp.n_text_state;
}
void persist_prepare(ohw_state* st) {
  const ohw_ctx* c = st->ctx;
  const ohw_hparams& hp = c->hp;
  if (!persist_dims_ok(hp) || st->ps_gran.p) return;
  PersistParams q{};
  q.d = hp.n_text_state; q.H = hp.n_text_head;
  const int64_t n_gran = persist_layout(&q);
  st->ps_gran.alloc((size_t)n_gran * 8, true);
  st->ps_words.alloc(64, true);
  const unsigned one = 1;
  HIP_CHECK(hipMemcpy(st->ps_words.p, &one, 4, hipMemcpyHostToDevice));
  std::vector<PersistLayer> lw((size_t)hp.n_text_layer);
  for (int l = 0; l < hp.n_text_layer; ++l) {
    const DecLayerW& w = c->dec[l];
    lw[(size_t)l] = PersistLayer{w.wqkv.p, w.wo.p, w.wxq.p, w.wxo.p, w.w1.p, w.w2.p, w.bqkv.as<float>(), w.bo.as<float>(), w.bxq.as<float>(),
                                 w.bxo.as<float>(), w.b1.as<float>(), w.b2.as<float>()};
  }
  st->ps_layers.alloc(lw.size() * sizeof(PersistLayer));
  HIP_CHECK(hipMemcpy(st->ps_layers.p, lw.data(), lw.size() * sizeof(PersistLayer), hipMemcpyHostToDevice));
  st->ps_layout = q;
}

template <typename T>
void run_mel(ohw_state* st, const float* pcm_dev, int64_t stride, int batch, int mode) {
  const ohw_ctx* c = st->ctx;
  MelParams p{};
  p.pcm = pcm_dev; p.pcm_stride = stride; p.n_samples = st->n_samples.as<int32_t>();
  p.filters = c->mel_filters.as<float>(); p.twiddle = c->twiddle.as<float>(); p.window = c->window.as<float>();
  p.logmel = st->logmel.as<float>(); p.max_bits = st->max_bits.as<int32_t>(); p.mel_t = st->mel_t.p;
  p.n_mels = c->hp.n_mels; p.batch = batch; p.mode = mode;
  launch_mel<T>(p, st->stream);
}

// first / total: the cross K/V of these B windows go to windows [first, first + B) of a decode batch of `total` windows
template <typename T>
void run_encode(ohw_state* st, int B, int first, int total) {
  const ohw_ctx* c = st->ctx;
  const ohw_hparams& hp = c->hp;
  hipStream_t s = st->stream;
  const int64_t d = hp.n_audio_state, Tn = hp.n_audio_ctx, M = (int64_t)B * Tn;
  GemmParams g{};
  // conv1 (k=3, pad 1) as a GEMM over overlapping rows of the time-major mel image
  g = GemmParams{};
  g.A = st->mel_t.p; g.W = c->conv1_w.p; g.bias = c->conv1_b.as<float>();
  g.out = (T*)st->c1.p + d;  // output row t -> image row 1 + t
  g.M = (int64_t)B * CHUNK_FRAMES; g.N = d; g.K = 3 * MEL_CPAD;
  g.lda = MEL_CPAD; g.a_batch_stride = (int64_t)MEL_ROWS * MEL_CPAD; g.rows_per_batch = CHUNK_FRAMES;
  g.ldc = d; g.c_batch_stride = (int64_t)MEL_ROWS * d;
  { ProfScope ps(st, OHW_PROF_ENC_GEMM, 2.0 * g.M * g.N * g.K); launch_gemm<T>(g, EPI_BIAS_GELU_T, s); }
  // conv2 (k=3, stride 2, pad 1): row t reads image rows 2t .. 2t+2 of conv1's padded output
  g = GemmParams{};
  g.A = st->c1.p; g.W = c->conv2_w.p; g.bias = c->conv2_b.as<float>(); g.pos = c->enc_pos.as<float>();
  g.out = st->h.p;
  g.M = M; g.N = d; g.K = 3 * d;
  g.lda = 2 * d; g.a_batch_stride = (int64_t)MEL_ROWS * d; g.rows_per_batch = Tn;
  g.ldc = d; g.c_batch_stride = Tn * d;
  { ProfScope ps(st, OHW_PROF_ENC_GEMM, 2.0 * g.M * g.N * g.K); launch_gemm<T>(g, EPI_GELU_POS_F32, s); }
  if (st->taps) HIP_CHECK(hipMemcpyAsync(st->tap_stem.p, st->h.p, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));

  auto dense = [&](const void* A, int64_t K, const DevBuf& W, const DevBuf& bias, void* out, int64_t N, int epi) {
    GemmParams q{};
    q.A = A; q.W = W.p; q.bias = bias.as<float>(); q.out = out;
    q.M = M; q.N = N; q.K = K; q.lda = K; q.a_batch_stride = 0; q.rows_per_batch = M; q.ldc = N; q.c_batch_stride = 0;
    ProfScope ps(st, OHW_PROF_ENC_GEMM, 2.0 * q.M * q.N * q.K);
    launch_gemm<T>(q, epi, s);
  };
  for (int l = 0; l < hp.n_audio_layer; ++l) {
    const EncLayerW& w = c->enc[l];
    launch_layernorm<T>(st->h.as<float>(), w.ln1.g.as<float>(), w.ln1.b.as<float>(), st->y.p, M, (int)d, s);
    dense(st->y.p, d, w.wqkv, w.bqkv, st->qkv.p, 3 * d, EPI_BIAS_T);
    {
      ProfScope ps(st, OHW_PROF_ENC_ATTN, 4.0 * B * hp.n_audio_head * (double)Tn * (double)Tn * 64.0);
      launch_encoder_attention<T>(st->qkv.p, st->att.p, B, (int)Tn, hp.n_audio_head, s);
    }
    dense(st->att.p, d, w.wo, w.bo, st->h.p, d, EPI_BIAS_RESID_F32);
    launch_layernorm<T>(st->h.as<float>(), w.ln2.g.as<float>(), w.ln2.b.as<float>(), st->y.p, M, (int)d, s);
    dense(st->y.p, d, w.w1, w.b1, st->ffn.p, 4 * d, EPI_BIAS_GELU_T);
    dense(st->ffn.p, 4 * d, w.w2, w.b2, st->h.p, d, EPI_BIAS_RESID_F32);
    if (l == 0 && st->taps) HIP_CHECK(hipMemcpyAsync(st->tap_block0.p, st->h.p, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
  }
  launch_layernorm<T>(st->h.as<float>(), c->ln_post.g.as<float>(), c->ln_post.b.as<float>(), st->enc.p, M, (int)d, s);
  // cross-attention K/V of every decoder layer in one GEMM: N = 2 * L * d, head-major output
  g = GemmParams{};
  g.A = st->enc.p; g.W = c->xkv_w.p; g.bias = c->xkv_b.as<float>(); g.out = st->xkv.p;
  g.M = M; g.N = (int64_t)2 * hp.n_text_layer * hp.n_text_state; g.K = d;
  g.lda = d; g.a_batch_stride = Tn * d; g.rows_per_batch = Tn; g.ldc = 0; g.c_batch_stride = 0;
  g.d_model = hp.n_text_state; g.n_head = hp.n_text_head; g.t_len = (int)Tn; g.batch = total; g.batch_offset = first;
  { ProfScope ps(st, OHW_PROF_ENC_GEMM, 2.0 * g.M * g.N * g.K); launch_gemm<T>(g, EPI_CROSSKV_T, s); }
}

// one decoder pass over M = B * n_new rows; tokens in tok_src (default st->step_tok), positions from st->n_past
// kv_group > 1 (beam search): B rows are kv_group beams per window - cross K/V is per window, self K/V goes through kv_slot,
// win_done flags whole windows
template <typename T>
void run_decoder_step(ohw_state* st, int B, int n_new, const int32_t* tok_src = nullptr, int kv_group = 1, const int32_t* kv_slot = nullptr,
                      const int32_t* win_done = nullptr) {
  const ohw_ctx* c = st->ctx;
  const ohw_hparams& hp = c->hp;
  hipStream_t s = st->stream;
  const int d = hp.n_text_state, H = hp.n_text_head, C = hp.n_text_ctx, Tn = hp.n_audio_ctx, L = hp.n_text_layer;
  const int M = B * n_new;
  if (M > st->m_max) throw Error(OHW_E_INVALID_ARG, "decode: batch * n_new exceeds the state's capacity (8 tokens per window per call)");
  const int32_t* n_past = st->n_past.as<int32_t>();
  const bool pn = st->postnorm;
  // ---- at most 16 single-token rows: the 32 layers in ONE persistent launch (decode_persist.hip) instead of 8 launches per
  // layer.  Not under ohw_state_set_batch_invariant (a window's bits must then not depend on which path its batch takes), not
  // while a kernel class is being profiled, not on the experimental post-norm / split-K paths.
  if (st->persist && n_new == 1 && M <= 16 && kv_group <= 5 && M % kv_group == 0 && !st->batch_invariant && !pn && st->prof_class == 0 &&
      st->ksplit_long == 1 && st->ksplit_short == 1 && st->ps_gran.p) {
    const int grid = std::max(1, std::min(st->stream_cus > 0 ? st->stream_cus : st->n_cu, 256));
    PersistParams q = st->ps_layout;
    launch_embed<T>(c->emb.p, c->dec_pos.as<float>(), tok_src ? tok_src : st->step_tok.as<int32_t>(), n_past, st->dx.as<float>(), nullptr, nullptr, M, 1, d, s);
    const int Wn = M / kv_group;
    q.layers = st->ps_layers.as<PersistLayer>(); q.L = L; q.M = M; q.group = kv_group; q.d = d; q.H = H; q.n_ctx = C; q.t_len = Tn;
    q.S = std::max(1, std::min(16, grid / std::max(1, Wn * H)));
    const int kb_mlp = 4 * d / 32;
    int ns = std::max(1, std::min(4, (grid + d / 16 - 1) / (d / 16)));
    while (ns < 4 && (kb_mlp + ns - 1) / ns + 1 > 56) ++ns;
    while (ns > 1 && kb_mlp / ns < 1) --ns;
    q.nsplit = ns;
    q.n_past = n_past; q.kv_slot = kv_slot;
    q.done = kv_group > 1 ? win_done : (st->skip_done ? st->done.as<int32_t>() : nullptr);
    q.x_in = st->dx.as<float>(); q.x_out = st->dx.as<float>();
    q.self_kv = st->self_kv.p; q.kv_layer = (int64_t)st->max_batch * H * C * 64; q.kv_row = (int64_t)H * C * 64;
    if ((int64_t)st->max_batch * q.kv_row * 2 >= ((int64_t)1 << 32)) throw Error(OHW_E_INVALID_ARG, "persistent step: a layer's K cache exceeds 4 GiB");
    q.xkv = st->xkv.p; q.xkv_slab = (int64_t)Wn * H * Tn * 64;
    q.g = st->ps_gran.as<unsigned long long>();
    q.epoch = st->ps_words.as<unsigned>(); q.abort_word = st->ps_words.as<unsigned>() + 8;
    launch_persist_step<T>(q, grid, s);
    ++st->persist_launches;
    launch_layernorm<T>(st->dx.as<float>(), c->dec_ln.g.as<float>(), c->dec_ln.b.as<float>(), st->dy.p, M, d, s, true);
    DecGemmParams lp{};
    lp.x = st->dy.p; lp.w = c->emb.p; lp.bias = nullptr; lp.out = st->logits.p; lp.cu_budget = st->stream_cus;
    lp.M = M; lp.N = hp.n_vocab; lp.K = d; lp.n_new = 1; lp.ld_out = st->logits_ld; lp.n_past = n_past; lp.d_model = d; lp.n_head = H; lp.n_ctx = C;
    launch_dec_gemm<T>(lp, DEPI_LOGITS, s);
    return;
  }
  launch_embed<T>(c->emb.p, c->dec_pos.as<float>(), tok_src ? tok_src : st->step_tok.as<int32_t>(), n_past, st->dx.as<float>(),
                  pn ? st->dx16.p : nullptr, pn ? st->xstat.as<float>() : nullptr, M, n_new, d, s);
  const int64_t kv_layer = (int64_t)st->max_batch * H * C * 64;      // elements per K (or V) cache of one layer
  const int64_t xkv_slab = (int64_t)(B / kv_group) * H * Tn * 64;    // cross K/V slab (batch of the last encode)
  auto gemm = [&](const void* x, const LayerNormW* ln, const DevBuf& w, const DevBuf& bias, void* out, int N, int K, int epi, int64_t ld,
                  const DevBuf* wsum = nullptr) {
    DecGemmParams p{};
    p.x = x; p.w = w.p; p.bias = bias.p ? bias.as<float>() : nullptr; p.out = out;
    p.ln = ln ? 1 : 0;
    if (ln && pn && wsum) {       // post-norm: the 16-bit tiled residual copy in, LayerNorm applied in the epilogue
      p.x = st->dx16.p; p.ln = 0; p.pn = 1; p.n_stat = K / 16; p.stat_in = st->xstat.as<float>(); p.wsum = wsum->as<float>();
    }
    if (epi == DEPI_BIAS_RESID && pn && N == d) { p.x16_out = st->dx16.p; p.stat_out = st->xstat.as<float>(); }
    p.cu_budget = st->stream_cus;
    if (epi == DEPI_BIAS_RESID) {
      const int ks = K >= 2 * d ? st->ksplit_long : st->ksplit_short;
      if (ks > 1 && ks <= K / 32) {
        p.ksplit = ks; p.slab = st->ks_slab.as<float>(); p.slab_bytes = (int32_t)st->ks_slab.bytes; p.ticket = st->ks_ticket.as<unsigned>();
      }
    }
    p.M = M; p.N = N; p.K = K; p.n_new = n_new; p.ld_out = ld; p.n_past = n_past;
    p.d_model = d; p.n_head = H; p.n_ctx = C;
    const int cls = epi == DEPI_BIAS_T ? OHW_PROF_DEC_GEMM_XQ : epi == DEPI_BIAS_GELU_T ? OHW_PROF_DEC_GEMM_FC1
                  : epi == DEPI_LOGITS ? OHW_PROF_DEC_GEMM_LOGITS : OHW_PROF_DEC_GEMM;
    // algorithmic bytes: the weights once per launch (the m-blocks of a prompt pass share them through L2)
    ProfScope ps(st, cls, 2.0 * (double)N * K);
    launch_dec_gemm<T>(p, epi, s);
  };
  // OHW_DEC_FUSE_ATTN=1: single-token steps of at most 16 rows run their self-attention inside the QKV launch (decode.hip,
  // self_attn_row<COH>): one launch less per layer, the same bits - and 1.6 us per layer SLOWER (large-v3, one row: 77.3 against
  // 74.9 ms per 48-step chunk): the drain of the write-through stores, the ticket and the loads from beyond L2 are three dependent
  // round trips, a kernel boundary plus the separate launch's first bytes two.  Off.
  const bool fuse_attn = st->fuse_attn && n_new == 1 && M <= 16 && !pn && d % 64 == 0 && kv_layer * 2 < ((int64_t)1 << 31) && st->attn_ticket.p;
  for (int l = 0; l < L; ++l) {
    const DecLayerW& w = c->dec[l];
    T* kc = (T*)st->self_kv.p + (int64_t)(2 * l) * kv_layer;
    T* vc = kc + kv_layer;
    {  // LN1 + fused QKV projection; K/V go straight into the cache at each window's position
      DecGemmParams p{};
      p.x = st->dx.p; p.ln = 1; p.cu_budget = st->stream_cus;
      if (pn) { p.x = st->dx16.p; p.ln = 0; p.pn = 1; p.n_stat = d / 16; p.stat_in = st->xstat.as<float>(); p.wsum = w.sqkv.as<float>(); }
      p.w = w.wqkv.p; p.bias = w.bqkv.as<float>(); p.out = st->dq.p;
      p.M = M; p.N = 3 * d; p.K = d; p.n_new = n_new; p.ld_out = d;
      p.k_cache = kc; p.v_cache = vc; p.n_past = n_past; p.d_model = d; p.n_head = H; p.n_ctx = C;
      if (fuse_attn) { p.attn_ticket = st->attn_ticket.as<unsigned>(); p.attn_out = st->da.p; p.attn_slots = kv_slot; p.kv_bytes = kv_layer * 2; }
      ProfScope ps(st, OHW_PROF_DEC_GEMM_QKV, 2.0 * (3.0 * d * d));
      launch_dec_gemm<T>(p, DEPI_QKV, s);
    }
    if (!fuse_attn) launch_self_attn<T>(st->dq.p, kc, vc, n_past, st->da.p, M, n_new, H, C, s, kv_slot);
    gemm(st->da.p, nullptr, w.wo, w.bo, st->dx.p, d, d, DEPI_BIAS_RESID, d);
    gemm(st->dx.p, &w.lnx, w.wxq, w.bxq, st->dq.p, d, d, DEPI_BIAS_T, d, &w.sxq);
    {
      // algorithmic bytes: K and V of every (query row, head); the prompt pass streams them once per (window, head)
      // for all its rows (cross_attn_rows_kernel: same condition as launch_cross_attn)
      const bool rows_path = n_new >= 2 && n_new <= 4 && ((int64_t)B * H >= 256 || st->batch_invariant);
      ProfScope psx(st, OHW_PROF_DEC_XATTN, 2.0 * 2.0 * (double)(kv_group > 1 ? B / kv_group : rows_path ? B : M) * H * Tn * 64.0);
      launch_cross_attn<T>(st->dq.p, (const T*)st->xkv.p + (int64_t)(2 * l) * xkv_slab, (const T*)st->xkv.p + (int64_t)(2 * l + 1) * xkv_slab,
                           st->da.p, M, n_new, H, Tn, st->xa_part.as<float>(), st->xa_ticket.as<unsigned>(), st->xa_rows,
                           kv_group > 1 ? win_done : (st->skip_done ? st->done.as<int32_t>() : nullptr), s, kv_group, st->batch_invariant);
    }
    gemm(st->da.p, nullptr, w.wxo, w.bxo, st->dx.p, d, d, DEPI_BIAS_RESID, d);
    gemm(st->dx.p, &w.ln2, w.w1, w.b1, st->df.p, 4 * d, d, DEPI_BIAS_GELU_T, 4 * d, &w.s1);
    gemm(st->df.p, nullptr, w.w2, w.b2, st->dx.p, d, 4 * d, DEPI_BIAS_RESID, d);
  }
  launch_layernorm<T>(st->dx.as<float>(), c->dec_ln.g.as<float>(), c->dec_ln.b.as<float>(), st->dy.p, M, d, s, true);
  DevBuf none;
  gemm(st->dy.p, nullptr, c->emb, none, st->logits.p, hp.n_vocab, d, DEPI_LOGITS, st->logits_ld);
}

// after a stream synchronisation: did a workgroup of a persistent decoder step give up waiting (decode_persist.hip)?  Loud.
void persist_check(ohw_state* st) {
  if (!st->ps_words.p || st->persist_launches == 0) return;
  unsigned w = 0;
  HIP_CHECK(hipMemcpy(&w, st->ps_words.as<unsigned>() + 8, 4, hipMemcpyDeviceToHost));
  if (w != 0) {
    const unsigned zero = 0;
    (void)hipMemcpy(st->ps_words.as<unsigned>() + 8, &zero, 4, hipMemcpyHostToDevice);
    throw Error(OHW_E_TRANSCRIBE, "persistent decoder step: a workgroup gave up waiting in layer " + std::to_string((w - 1) / 10) + ", phase " +
                                      std::to_string((w - 1) % 10) + " (are all workgroups resident? OHW_DEC_PERSIST=0 selects the launch-per-kernel path)");
  }
}

void fill_sampler(const ohw_state* st, const ohw_sample_params* sp, int B, SamplerParams* p) {
  const ohw_ctx* c = st->ctx;
  std::memset(p, 0, sizeof *p);     // every byte, padding included: the graph caches compare these structs with memcmp
  p->logits = st->logits.as<float>(); p->ld = st->logits_ld;
  p->tokens = st->tokens.as<int32_t>(); p->n_cur = st->n_cur.as<int32_t>(); p->n_past = st->n_past.as<int32_t>();
  p->next_tok = st->next_tok.as<int32_t>(); p->done = st->done.as<int32_t>(); p->n_done = st->n_done.as<int32_t>();
  p->sum_logprob = st->sum_lp.as<float>();
  p->partials = st->samp_part.as<float>(); p->tickets = st->samp_ticket.as<unsigned>();
  p->bias = st->bias_on ? st->logit_bias.as<float>() : nullptr;
  p->tok_lp = st->tok_lp.as<float>(); p->nosp_prob = st->nosp_prob.as<float>();
  p->batch = B; p->max_tokens = st->max_tokens; p->n_vocab = c->hp.n_vocab;
  p->eot = c->tok.eot; p->sot = c->tok.sot; p->translate = c->tok.translate; p->transcribe = c->tok.transcribe;
  p->solm = c->tok.solm; p->prev = c->tok.prev; p->nosp = c->tok.nosp; p->no_ts = c->tok.no_timestamps;
  p->ts_begin = c->tok.timestamp_begin; p->blank = c->tok.blank; p->n_langs = c->tok.n_langs;
  p->suppress_blank = sp->suppress_blank; p->no_timestamps = sp->no_timestamps; p->max_initial_ts = sp->max_initial_ts;
  p->n_max = sp->n_max; p->force_len = sp->force_len; p->n_text_ctx = c->hp.n_text_ctx;
}

int build_prompt(const ohw_ctx* c, const ohw_sample_params* sp, int32_t* out) {
  int n = 0;
  out[n++] = c->tok.sot;
  if (c->hp.n_vocab >= 51865) {
    out[n++] = c->tok.sot + 1 + sp->lang_id;
    out[n++] = sp->translate ? c->tok.translate : c->tok.transcribe;
  }
  if (sp->no_timestamps) out[n++] = c->tok.no_timestamps;
  return n;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* ohw_last_error(void) { return g_last_error.c_str(); }
int ohw_abi_version(void) { return OHW_ABI_VERSION; }

int ohw_ctx_create(const char* model_path, int device, int dtype, ohw_ctx** out) {
  return guard([&] {
    if (!out) throw Error(OHW_E_INVALID_ARG, "out is null");
    *out = nullptr;
    *out = ctx_from_file(model_path, device, dtype);
  });
}

int ohw_ctx_create_synthetic(const ohw_hparams* hp, uint32_t seed, int device, int dtype, ohw_ctx** out) {
  return guard([&] {
    if (!out) throw Error(OHW_E_INVALID_ARG, "out is null");
    *out = nullptr;
    *out = ctx_synthetic(hp, seed, device, dtype);
  });
}

int ohw_ctx_info(const ohw_ctx* ctx, ohw_hparams* hp, ohw_special_tokens* tok) {
  return guard([&] {
    if (!ctx) throw Error(OHW_E_INVALID_ARG, "ctx is null");
    if (hp) *hp = ctx->hp;
    if (tok) *tok = ctx->tok;
  });
}

int ohw_ctx_dtype(const ohw_ctx* ctx) { return ctx ? ctx->dtype : OHW_E_INVALID_ARG; }

int ohw_token_text(const ohw_ctx* ctx, int32_t id, const char** text) {
  if (!ctx || id < 0 || (size_t)id >= ctx->vocab.size()) { if (text) *text = ""; return 0; }
  if (text) *text = ctx->vocab[(size_t)id].c_str();
  return (int)ctx->vocab[(size_t)id].size();
}

void ohw_ctx_free(ohw_ctx* ctx) {
  if (!ctx) return;
  ApiScope api;
  (void)hipSetDevice(ctx->device);
  delete ctx;
}

int ohw_state_create(ohw_ctx* ctx, int max_batch, ohw_state** out) {
  return guard([&] {
    if (!ctx || !out) throw Error(OHW_E_INVALID_ARG, "ctx/out is null");
    if (max_batch < 1 || max_batch > 256) throw Error(OHW_E_INVALID_ARG, "max_batch must be in 1..256");
    *out = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    std::unique_ptr<ohw_state> st(new ohw_state());
    st->ctx = ctx;
    st->max_batch = max_batch;
    HIP_CHECK(hipStreamCreateWithFlags(&st->own_stream, hipStreamNonBlocking));
    st->stream = st->own_stream;
    state_alloc(st.get());
    HIP_CHECK(hipDeviceSynchronize());
    *out = st.release();
  });
}

void ohw_state_free(ohw_state* st) {
  if (!st) return;
  ApiScope api;
  (void)hipSetDevice(st->ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& e : st->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : st->prof_ev) if (e) (void)hipEventDestroy(e);
  for (auto& g : st->step_graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  for (auto& g : st->beam_graphs)
    for (int q = 0; q < 2; ++q) {
      if (g.exec[q]) (void)hipGraphExecDestroy(g.exec[q]);
      if (g.graph[q]) (void)hipGraphDestroy(g.graph[q]);
    }
  if (st->own_stream) (void)hipStreamDestroy(st->own_stream);
  delete st;
}

int ohw_state_set_stream(ohw_state* st, void* hip_stream) {
  return guard([&] {
    if (!st) throw Error(OHW_E_INVALID_ARG, "state is null");
    st->stream = hip_stream ? (hipStream_t)hip_stream : st->own_stream;
    st->stream_cus = 0;
    if (hip_stream) {
      uint32_t mask[16] = {0};
      if (hipExtStreamGetCUMask(st->stream, 16, mask) == hipSuccess) {
        int n = 0, total = 0;
        for (uint32_t m : mask) n += __builtin_popcount(m);
        (void)hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, st->ctx->device);
        if (n > 0 && n < total) st->stream_cus = n;
      } else {
        (void)hipGetLastError();
      }
    }

  });
}

int ohw_stream_create(int device, int first_cu, int n_cu, void** stream_out) {
  return guard([&] {
    if (!stream_out) throw Error(OHW_E_INVALID_ARG, "stream_out is null");
    *stream_out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) throw Error(OHW_E_NO_GPU, "stream: bad device index");
    HIP_CHECK(hipSetDevice(device));
    hipStream_t s = nullptr;
    if (n_cu <= 0) {
      HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
      hipDeviceProp_t prop;
      HIP_CHECK(hipGetDeviceProperties(&prop, device));
      const int total = prop.multiProcessorCount;
      if (first_cu < 0 || first_cu + n_cu > total) throw Error(OHW_E_INVALID_ARG, "stream: CU range exceeds the device's compute units");
      std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
      for (int b = first_cu; b < first_cu + n_cu; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
      HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    }
    *stream_out = (void*)s;
  });
}
int ohw_stream_destroy(void* stream) {
  return guard([&] {
    if (stream) HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  });
}
int ohw_stream_wait(void* waiter, void* signal) {
  return guard([&] {
    if (!waiter || !signal) throw Error(OHW_E_INVALID_ARG, "stream is null");
    hipEvent_t e = nullptr;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, (hipStream_t)signal);
    if (r == hipSuccess) r = hipStreamWaitEvent((hipStream_t)waiter, e, 0);
    (void)hipEventDestroy(e);   // released once the recorded work has completed
    HIP_CHECK(r);
  });
}
int ohw_stream_sync(void* stream) {
  return guard([&] {
    if (!stream) throw Error(OHW_E_INVALID_ARG, "stream is null");
    HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  });
}

int ohw_state_max_batch(const ohw_state* st) { return st ? st->max_batch : 0; }
}  // extern "C"
namespace ohw {
// the state's logit bias as the host sampler needs it (null when none is set); library-internal
// dst reads src's recording (no copy: a view of its samples) and gets its maximum; same device; dst must not outlive src's
// recording (the engine shares among its own states for the length of one transcribe)
// lane states of the engine's LANES schedule: a graph capture waits until every other thread has left the library, i.e. until
// the other lanes' decodes are over - a lane only captures where the replay clearly pays (batches below `max_batch`)
void state_set_graph_max_batch(ohw_state* st, int max_batch) { if (st && max_batch >= 1) st->graph_max_batch = max_batch; }
void state_share_recording(ohw_state* dst, ohw_state* src) {
  if (!dst || !src || dst == src) return;
  if (src->rec_n < 1) throw Error(OHW_E_INVALID_ARG, "share_recording: the source state holds no recording");
  HIP_CHECK(hipSetDevice(src->ctx->device));
  HIP_CHECK(hipStreamSynchronize(src->stream));                 // the maximum is final
  dst->rec_pcm.view(src->rec_pcm.p, src->rec_pcm.bytes);
  dst->rec_n = src->rec_n;
  if (!dst->rec_max.p) dst->rec_max.alloc(4);
  if (dst->rec_off.bytes < (size_t)dst->max_batch * 8) dst->rec_off.alloc((size_t)dst->max_batch * 8);
  HIP_CHECK(hipMemcpy(dst->rec_max.p, src->rec_max.p, 4, hipMemcpyDeviceToDevice));
}
void state_drop_recording(ohw_state* st) {
  if (!st || st->rec_pcm.owned) return;
  st->rec_pcm.release();          // a view: nothing is freed
  st->rec_n = 0;
}
const float* state_bias_host(const ohw_state* st) { return st && st->bias_on && !st->bias_host.empty() ? st->bias_host.data() : nullptr; }
}
extern "C" {
const ohw_ctx* ohw_state_ctx(const ohw_state* st) { return st ? st->ctx : nullptr; }

int ohw_mel(ohw_state* st, const float* pcm, int64_t pcm_stride, const int32_t* n_samples, int batch, int pcm_on_device,
            int mel_mode, float* mel_out) {
  return guard([&] {
    if (!st || !pcm || !n_samples) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (batch < 1 || batch > st->max_batch) throw Error(OHW_E_INVALID_ARG, "batch exceeds the state's max_batch");
    if (mel_mode != OHW_MEL_REFLECT && mel_mode != OHW_MEL_ZERO_TAIL) throw Error(OHW_E_INVALID_ARG, "bad mel_mode");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    hipStream_t s = st->stream;
    for (int b = 0; b < batch; ++b)
      if (n_samples[b] < 0 || n_samples[b] > CHUNK_SAMPLES || n_samples[b] > pcm_stride)
        throw Error(OHW_E_INVALID_ARG, "n_samples must be in 0..480000 and <= pcm_stride");
    HIP_CHECK(hipMemcpyAsync(st->n_samples.p, n_samples, (size_t)batch * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipEventRecord(st->ev[0], s));
    const float* pcm_dev = pcm;
    int64_t stride = pcm_stride;
    if (!pcm_on_device) {
      for (int b = 0; b < batch; ++b)
        if (n_samples[b] > 0)
          HIP_CHECK(hipMemcpyAsync(st->pcm.as<float>() + (int64_t)b * CHUNK_SAMPLES, pcm + (int64_t)b * pcm_stride,
                                   (size_t)n_samples[b] * 4, hipMemcpyHostToDevice, s));
      pcm_dev = st->pcm.as<float>();
      stride = CHUNK_SAMPLES;
    }
    Dispatch::run(st->ctx->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      run_mel<T>(st, pcm_dev, stride, batch, mel_mode);
    });
    HIP_CHECK(hipEventRecord(st->ev[1], s));
    st->mel_batch = batch;
    st->enc_batch = batch;
    if (mel_out) {
      HIP_CHECK(hipMemcpyAsync(mel_out, st->logmel.p, (size_t)batch * st->ctx->hp.n_mels * CHUNK_FRAMES * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    }
  });
}

int ohw_recording_set(ohw_state* st, const float* pcm, int64_t n, int pcm_on_device, float* log_max_out) {
  return guard([&] {
    if (!st || !pcm || n < 1) throw Error(OHW_E_INVALID_ARG, "recording: null or empty");
    if (n > (int64_t)7200 * 16000) throw Error(OHW_E_INVALID_ARG, "recording: more than two hours");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    hipStream_t s = st->stream;
    if (st->rec_pcm.bytes < (size_t)n * 4) st->rec_pcm.alloc((size_t)n * 4);
    HIP_CHECK(hipMemcpyAsync(st->rec_pcm.p, pcm, (size_t)n * 4, pcm_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    st->rec_n = n;
    // frames that touch a sample, in pseudo-windows of 3000; the frames of the 30 s zero tail are log10(1e-10) = -10
    const int64_t n_audio = (n + N_FFT / 2) / HOP + 1;
    const int chunks = (int)((n_audio + CHUNK_FRAMES - 1) / CHUNK_FRAMES);
    const size_t need = (size_t)std::max(chunks, st->max_batch) * 8;
    if (st->rec_off.bytes < need) st->rec_off.alloc(need);
    if (!st->rec_max.p) st->rec_max.alloc(4);
    std::vector<int64_t> offs((size_t)chunks);
    for (int i = 0; i < chunks; ++i) offs[(size_t)i] = (int64_t)i * CHUNK_SAMPLES;
    const float floor_v = -10.0f;
    int32_t floor_bits;
    std::memcpy(&floor_bits, &floor_v, 4);
    floor_bits ^= 0x7fffffff;                      // the kernels' ordered-int form of a negative float
    HIP_CHECK(hipMemcpyAsync(st->rec_max.p, &floor_bits, 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(st->rec_off.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));            // offs and floor_bits are stack-lifetime sources
    const ohw_ctx* c = st->ctx;
    MelParams p{};
    p.pcm = st->rec_pcm.as<float>(); p.n_samples = st->n_samples.as<int32_t>();
    p.filters = c->mel_filters.as<float>(); p.twiddle = c->twiddle.as<float>(); p.window = c->window.as<float>();
    p.logmel = st->logmel.as<float>(); p.max_bits = st->rec_max.as<int32_t>(); p.mel_t = st->mel_t.p;
    p.n_mels = c->hp.n_mels; p.batch = chunks; p.mode = OHW_MEL_ZERO_TAIL;
    p.offsets = st->rec_off.as<int64_t>(); p.n_total = n; p.shared_max = 1; p.max_only = 1;
    Dispatch::run(c->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      launch_mel<T>(p, s);
    });
    if (log_max_out) {
      int32_t bits = 0;
      HIP_CHECK(hipMemcpyAsync(&bits, st->rec_max.p, 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      if (bits < 0) bits ^= 0x7fffffff;
      std::memcpy(log_max_out, &bits, 4);
    }
  });
}

int ohw_mel_seek(ohw_state* st, const int32_t* seek_frames, int batch, float* mel_out) {
  return guard([&] {
    if (!st || !seek_frames) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (st->rec_n < 1) throw Error(OHW_E_INVALID_ARG, "mel_seek: no recording (ohw_recording_set)");
    if (batch < 1 || batch > st->max_batch) throw Error(OHW_E_INVALID_ARG, "batch exceeds the state's max_batch");
    const int64_t n_len = (st->rec_n + CHUNK_SAMPLES) / HOP;
    std::vector<int64_t> offs((size_t)batch);
    for (int b = 0; b < batch; ++b) {
      if (seek_frames[b] < 0 || seek_frames[b] >= n_len) throw Error(OHW_E_INVALID_ARG, "mel_seek: seek outside the recording's frames");
      offs[(size_t)b] = (int64_t)seek_frames[b] * HOP;
    }
    HIP_CHECK(hipSetDevice(st->ctx->device));
    hipStream_t s = st->stream;
    HIP_CHECK(hipEventRecord(st->ev[0], s));
    HIP_CHECK(hipMemcpyAsync(st->rec_off.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    const ohw_ctx* c = st->ctx;
    MelParams p{};
    p.pcm = st->rec_pcm.as<float>(); p.n_samples = st->n_samples.as<int32_t>();
    p.filters = c->mel_filters.as<float>(); p.twiddle = c->twiddle.as<float>(); p.window = c->window.as<float>();
    p.logmel = st->logmel.as<float>(); p.max_bits = st->rec_max.as<int32_t>(); p.mel_t = st->mel_t.p;
    p.n_mels = c->hp.n_mels; p.batch = batch; p.mode = OHW_MEL_ZERO_TAIL;
    p.offsets = st->rec_off.as<int64_t>(); p.n_total = st->rec_n; p.shared_max = 1; p.max_only = 0;
    Dispatch::run(c->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      launch_mel<T>(p, s);
    });
    HIP_CHECK(hipEventRecord(st->ev[1], s));
    st->mel_batch = batch;
    st->enc_batch = batch;
    if (mel_out) {
      HIP_CHECK(hipMemcpyAsync(mel_out, st->logmel.p, (size_t)batch * c->hp.n_mels * CHUNK_FRAMES * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    }
  });
}

int ohw_encode_slice(ohw_state* st, int batch, int first, int total) {
  return guard([&] {
    if (!st) throw Error(OHW_E_INVALID_ARG, "state is null");
    if (batch < 1 || batch != st->mel_batch) throw Error(OHW_E_INVALID_ARG, "encode: batch must equal the batch of the last ohw_mel");
    if (first < 0 || total < first + batch || total > st->max_batch) throw Error(OHW_E_INVALID_ARG, "encode: slice exceeds the state's max_batch");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    HIP_CHECK(hipEventRecord(st->ev[2], st->stream));
    Dispatch::run(st->ctx->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      run_encode<T>(st, batch, first, total);
    });
    HIP_CHECK(hipEventRecord(st->ev[3], st->stream));
    st->enc_batch = total;
  });
}

int ohw_encode(ohw_state* st, int batch) { return ohw_encode_slice(st, batch, 0, batch); }

int ohw_decode_active(ohw_state* st, const int32_t* tokens, int n_new, const int32_t* n_past, int batch, const int32_t* active,
                      float* logits_out) {
  return guard([&] {
    if (!st || !tokens || !n_past) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (batch < 1 || batch != st->enc_batch) throw Error(OHW_E_INVALID_ARG, "decode: batch must equal the batch of the last ohw_encode");
    if (n_new < 1 || n_new > 8) throw Error(OHW_E_INVALID_ARG, "decode: n_new must be in 1..8");
    const ohw_hparams& hp = st->ctx->hp;
    for (int b = 0; b < batch; ++b) {
      if (n_past[b] < 0 || n_past[b] + n_new > hp.n_text_ctx) throw Error(OHW_E_INVALID_ARG, "decode: position exceeds n_text_ctx");
      for (int i = 0; i < n_new; ++i)
        if (tokens[b * n_new + i] < 0 || tokens[b * n_new + i] >= hp.n_vocab) throw Error(OHW_E_INVALID_ARG, "decode: token id out of range");
    }
    HIP_CHECK(hipSetDevice(st->ctx->device));
    hipStream_t s = st->stream;
    HIP_CHECK(hipMemcpyAsync(st->step_tok.p, tokens, (size_t)batch * n_new * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(st->n_past.p, n_past, (size_t)batch * 4, hipMemcpyHostToDevice, s));
    // inactive windows ride along (their GEMM rows are free) but their cross K/V is not streamed: the done flags the
    // greedy loop uses for finished windows
    std::vector<int32_t> skip;
    struct SkipDone { bool& f; bool on; SkipDone(bool& r, bool o) : f(r), on(o) { if (on) f = true; } ~SkipDone() { if (on) f = false; } } skip_done(st->skip_done, active != nullptr);
    if (active) {
      skip.resize((size_t)batch);
      for (int b = 0; b < batch; ++b) skip[(size_t)b] = active[b] ? 0 : 1;
      HIP_CHECK(hipMemcpyAsync(st->done.p, skip.data(), skip.size() * 4, hipMemcpyHostToDevice, s));
    }
    Dispatch::run(st->ctx->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      run_decoder_step<T>(st, batch, n_new);
    });
    if (logits_out) {
      if (!active) {
        HIP_CHECK(hipMemcpy2DAsync(logits_out, (size_t)hp.n_vocab * 4, st->logits.p, (size_t)st->logits_ld * 4, (size_t)hp.n_vocab * 4,
                                   (size_t)batch, hipMemcpyDeviceToHost, s));
      } else {
        for (int b = 0; b < batch; ++b)
          if (active[b])
            HIP_CHECK(hipMemcpyAsync(logits_out + (size_t)b * hp.n_vocab, st->logits.as<float>() + (size_t)b * st->logits_ld, (size_t)hp.n_vocab * 4,
                                     hipMemcpyDeviceToHost, s));
      }
    }
    HIP_CHECK(hipStreamSynchronize(s));   // also keeps `skip` alive until its copy has run
    persist_check(st);
  });
}

int ohw_decode(ohw_state* st, const int32_t* tokens, int n_new, const int32_t* n_past, int batch, float* logits_out) {
  return ohw_decode_active(st, tokens, n_new, n_past, batch, nullptr, logits_out);
}

void ohw_default_sample_params(const ohw_ctx* ctx, ohw_sample_params* p) {
  if (!p) return;
  p->lang_id = 0; p->translate = 0; p->no_timestamps = 0; p->suppress_blank = 1; p->max_initial_ts = 50;
  p->n_max = ctx ? ctx->hp.n_text_ctx / 2 - 4 : 220;
  p->force_len = 0;
}

int ohw_greedy_ex(ohw_state* st, const ohw_sample_params* sp, int batch, int max_tokens, const ohw_greedy_result* res) {
  return guard([&] {
    if (!st || !sp || !res || !res->tokens || !res->n_tokens) throw Error(OHW_E_INVALID_ARG, "null argument");
    int32_t* tokens_out = res->tokens;
    int32_t* n_tokens_out = res->n_tokens;
    float* sum_logprob_out = res->sum_logprob;
    if (batch < 1 || batch != st->enc_batch) throw Error(OHW_E_INVALID_ARG, "greedy: batch must equal the batch of the last ohw_encode");
    const ohw_ctx* c = st->ctx;
    if (sp->lang_id < 0 || sp->lang_id >= c->tok.n_langs) throw Error(OHW_E_INVALID_ARG, "greedy: lang_id out of range");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = st->stream;
    int32_t prompt[8];
    const int n_prompt = build_prompt(c, sp, prompt);
    const int n_max_raw = sp->force_len > 0 ? sp->force_len : sp->n_max;
    const int n_max = std::min(std::min(n_max_raw, st->max_tokens), c->hp.n_text_ctx - n_prompt);
    if (n_max < 1) throw Error(OHW_E_INVALID_ARG, "greedy: n_max < 1");
    std::vector<int32_t> ptoks((size_t)batch * n_prompt);
    for (int b = 0; b < batch; ++b) std::memcpy(&ptoks[(size_t)b * n_prompt], prompt, (size_t)n_prompt * 4);
    HIP_CHECK(hipEventRecord(st->ev[4], s));
    HIP_CHECK(hipMemcpyAsync(st->step_tok.p, ptoks.data(), ptoks.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(st->n_past.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->n_cur.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->done.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->n_done.p, 0, 16, s));
    HIP_CHECK(hipMemsetAsync(st->sum_lp.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->next_tok.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->nosp_prob.p, 0, (size_t)batch * 4, s));
    SamplerParams spar;
    ohw_sample_params eff = *sp;
    eff.n_max = n_max;
    if (eff.force_len > 0) eff.force_len = n_max;
    fill_sampler(st, &eff, batch, &spar);
    int steps = 0;
    struct SkipDone { bool& f; explicit SkipDone(bool& r) : f(r) { f = true; } ~SkipDone() { f = false; } } skip_done(st->skip_done);
    Dispatch::run(c->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      run_decoder_step<T>(st, batch, n_prompt);
      std::vector<int32_t> np((size_t)batch, n_prompt);
      HIP_CHECK(hipMemcpyAsync(st->n_past.p, np.data(), np.size() * 4, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));  // np is a stack-lifetime source
      ++steps;
      spar.advance = 0;
      launch_sampler(spar, s);   // first token of every window from the prompt's logits
      spar.advance = 1;          // every later sampler call follows a single-token step
      // One greedy iteration = {feed next_tok, decoder step, sampler}.  It is launch-bound (about 260
      // short kernels), so it is captured once into a hipGraph and replayed; positions, tokens and
      // the done flags live in device memory, so the same graph serves every iteration.
      const bool use_graph = st->graphs_enabled && st->prof_class == 0 && s != nullptr && batch < st->graph_max_batch;
      hipGraphExec_t step_exec = nullptr;
      if (use_graph) {
        for (auto& g : st->step_graphs)
          if (g.batch == batch && g.cus == st->stream_cus && g.invariant == st->batch_invariant && g.persist == st->persist && std::memcmp(&g.spar, &spar, sizeof spar) == 0) step_exec = g.exec;
      }
      if (use_graph && !step_exec) {
        if (st->step_graphs.size() >= 8) {     // bounded: drop the oldest capture
          auto& g = st->step_graphs.front();
          if (g.exec) (void)hipGraphExecDestroy(g.exec);
          if (g.graph) (void)hipGraphDestroy(g.graph);
          st->step_graphs.erase(st->step_graphs.begin());
        }
        ohw_state::StepGraph ng;
        // The iteration is captured on the state's OWN stream (non-blocking) and replayed on whatever stream the state
        // runs on: a CU-masked stream (hipExtStreamCreateWithCUMask) is a blocking stream, and while a blocking stream
        // captures, any use of the legacy stream by another thread - another engine loading its model, say - fails with
        // "would make the legacy stream depend on a capturing blocking stream" and kills the capture (two engines driven
        // by two threads: tests/test_gpu_configs.py).  The own stream is idle whenever the state runs on an external one.
        // Capture mode RELAXED: in the other modes HIP (ROCm 7.2) rejects every synchronous memory call of EVERY thread
        // while a capture is open - hipMemset in another engine's state allocation failed that way - and this thread
        // makes no call during the capture that the stricter modes would have to catch.
        hipStream_t cap = st->own_stream;
        CaptureGate gate;   // no other thread is inside the library while this stream captures (common.hpp)
        struct StreamSwap { ohw_state* st; hipStream_t keep; ~StreamSwap() { st->stream = keep; } } swap{st, st->stream};
        st->stream = cap;
        HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
        try {
          run_decoder_step<T>(st, batch, 1, st->next_tok.as<int32_t>());   // the token the sampler just wrote
          launch_sampler(spar, cap);
        } catch (...) {
          hipGraph_t g = nullptr;
          (void)hipStreamEndCapture(cap, &g);
          if (g) (void)hipGraphDestroy(g);
          throw;
        }
        HIP_CHECK(hipStreamEndCapture(cap, &ng.graph));
        hipError_t ie = hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) { (void)hipGraphDestroy(ng.graph); HIP_CHECK(ie); }
        ng.batch = batch; ng.cus = st->stream_cus; ng.invariant = st->batch_invariant; ng.persist = st->persist; ng.spar = spar;
        st->step_graphs.push_back(ng);
        ++st->step_captures;
        step_exec = ng.exec;
      }
      int32_t n_done_host = 0;
      for (int it = 1; it < n_max; ++it) {
        if (use_graph) {
          HIP_CHECK(hipGraphLaunch(step_exec, s));
        } else {
          run_decoder_step<T>(st, batch, 1, st->next_tok.as<int32_t>());
          launch_sampler(spar, s);
        }
        ++steps;
        if (sp->force_len <= 0 && ((it & 7) == 7)) {
          HIP_CHECK(hipMemcpyAsync(&n_done_host, st->n_done.p, 4, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipStreamSynchronize(s));
          if (n_done_host >= batch) break;
        }
      }
    });
    HIP_CHECK(hipEventRecord(st->ev[5], s));
    std::vector<int32_t> toks((size_t)batch * st->max_tokens), ncur((size_t)batch);
    HIP_CHECK(hipMemcpyAsync(toks.data(), st->tokens.p, toks.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(ncur.data(), st->n_cur.p, ncur.size() * 4, hipMemcpyDeviceToHost, s));
    if (sum_logprob_out) HIP_CHECK(hipMemcpyAsync(sum_logprob_out, st->sum_lp.p, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    std::vector<int32_t> last((size_t)batch);
    std::vector<float> lps;
    HIP_CHECK(hipMemcpyAsync(last.data(), st->next_tok.p, last.size() * 4, hipMemcpyDeviceToHost, s));
    if (res->no_speech_prob) HIP_CHECK(hipMemcpyAsync(res->no_speech_prob, st->nosp_prob.p, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    if (res->token_logprobs) {
      lps.resize((size_t)batch * (st->max_tokens + 1));
      HIP_CHECK(hipMemcpyAsync(lps.data(), st->tok_lp.p, lps.size() * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    persist_check(st);
    for (int b = 0; b < batch; ++b) {
      const int n = std::min(ncur[(size_t)b], max_tokens);
      n_tokens_out[b] = n;
      std::memcpy(tokens_out + (size_t)b * max_tokens, &toks[(size_t)b * st->max_tokens], (size_t)n * 4);
      const bool eot = last[(size_t)b] == c->tok.eot && ncur[(size_t)b] <= max_tokens;
      if (res->ended_by_eot) res->ended_by_eot[b] = eot ? 1 : 0;
      if (res->token_logprobs) {
        float* dst = res->token_logprobs + (size_t)b * (max_tokens + 1);
        const int m = std::min(n + (eot ? 1 : 0), max_tokens + 1);
        std::memcpy(dst, &lps[(size_t)b * (st->max_tokens + 1)], (size_t)m * 4);
      }
    }
    st->last.decode_steps = steps;
  });
}

int ohw_greedy(ohw_state* st, const ohw_sample_params* sp, int batch, int32_t* tokens_out, int32_t* n_tokens_out, int max_tokens,
               float* sum_logprob_out) {
  ohw_greedy_result r{};
  r.tokens = tokens_out; r.n_tokens = n_tokens_out; r.sum_logprob = sum_logprob_out;
  return ohw_greedy_ex(st, sp, batch, max_tokens, &r);
}

// OHW_DEBUG_MARKS=1: progress lines on stderr (which HIP call a tool died under)
#define BMARK(what, i) do { static const bool on_ = env_int("OHW_DEBUG_MARKS", 0, 0, 1) != 0; if (on_) { std::fprintf(stderr, "[ohw beam] %s %d\n", what, (int)(i)); std::fflush(stderr); } } while (0)

int ohw_beam_search(ohw_state* st, const ohw_sample_params* sp, int n_windows, int beam_size, int max_tokens, const ohw_beam_result* res) {
  return guard([&] {
    if (!st || !sp || !res || !res->tokens || !res->n_tokens) throw Error(OHW_E_INVALID_ARG, "null argument");
    const int W = n_windows, K = beam_size, R = W * K;
    if (W < 1 || W != st->enc_batch) throw Error(OHW_E_INVALID_ARG, "beam search: n_windows must equal the batch of the last ohw_encode");
    if (K < 2 || K > 5) throw Error(OHW_E_INVALID_ARG, "beam search: beam_size must be in 2..5");
    if (R > st->max_batch) throw Error(OHW_E_INVALID_ARG, "beam search: the state needs max_batch >= n_windows * beam_size decoder rows");
    const ohw_ctx* c = st->ctx;
    if (sp->lang_id < 0 || sp->lang_id >= c->tok.n_langs) throw Error(OHW_E_INVALID_ARG, "beam search: lang_id out of range");
    if (sp->force_len > 0) throw Error(OHW_E_INVALID_ARG, "beam search: force_len is a greedy-only knob");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = st->stream;
    const int MT = st->max_tokens, C = c->hp.n_text_ctx, MB = st->max_batch;
    if (!st->bm_sum.p) {
      st->bm_cand_lp.alloc((size_t)MB * 6 * 4); st->bm_cand_tok.alloc((size_t)MB * 6 * 4); st->bm_sum.alloc((size_t)MB * 4, true);
      st->bm_slot[0].alloc((size_t)MB * C * 4, true); st->bm_slot[1].alloc((size_t)MB * C * 4, true); st->bm_tok2.alloc((size_t)MB * MT * 4, true);
      st->bm_ncur.alloc((size_t)MB * 4, true); st->bm_npast.alloc((size_t)MB * 4, true); st->bm_done.alloc((size_t)MB * 4, true);
      st->bm_fin_cnt.alloc((size_t)MB * 4, true); st->bm_fin_tok.alloc((size_t)MB * MT * 4, true); st->bm_fin_len.alloc((size_t)MB * 4, true);
      st->bm_fin_sum.alloc((size_t)MB * 4, true);
      st->bm_part.alloc((size_t)MB * BEAM_SPLIT * BEAM_PART_WORDS * 4, true); st->bm_ticket.alloc((size_t)MB * 4, true);
    }
    int32_t prompt[8];
    const int n_prompt = build_prompt(c, sp, prompt);
    const int n_max = std::min(std::min(sp->n_max, MT), C - n_prompt);
    if (n_max < 1) throw Error(OHW_E_INVALID_ARG, "beam search: n_max < 1");
    std::vector<int32_t> ptoks((size_t)W * n_prompt), np0((size_t)W, n_prompt - 1);
    for (int w = 0; w < W; ++w) std::memcpy(&ptoks[(size_t)w * n_prompt], prompt, (size_t)n_prompt * 4);
    HIP_CHECK(hipEventRecord(st->ev[4], s));
    HIP_CHECK(hipMemcpyAsync(st->step_tok.p, ptoks.data(), ptoks.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(st->n_past.p, 0, (size_t)MB * 4, s));
    HIP_CHECK(hipMemsetAsync(st->n_done.p, 0, 16, s));
    for (DevBuf* b : {&st->bm_sum, &st->bm_ncur, &st->bm_done, &st->bm_fin_cnt, &st->bm_fin_len}) HIP_CHECK(hipMemsetAsync(b->p, 0, (size_t)MB * 4, s));
    HIP_CHECK(hipMemcpyAsync(st->bm_npast.p, np0.data(), np0.size() * 4, hipMemcpyHostToDevice, s));
    ohw_sample_params eff = *sp;
    eff.n_max = n_max;
    SamplerParams base;
    fill_sampler(st, &eff, R, &base);     // zeroes every byte first; advance stays 0: the beam step advances n_past itself
    int32_t* tokbuf[2] = {st->tokens.as<int32_t>(), st->bm_tok2.as<int32_t>()};
    auto params = [&](int q, SamplerParams* p, BeamParams* bp) {
      std::memcpy(p, &base, sizeof base);
      p->tokens = tokbuf[q];
      *bp = BeamParams{};
      bp->K = K; bp->cand_lp = st->bm_cand_lp.as<float>(); bp->cand_tok = st->bm_cand_tok.as<int32_t>(); bp->beam_sum = st->bm_sum.as<float>();
      bp->kv_slot = st->bm_slot[q].as<int32_t>(); bp->kv_slot_next = st->bm_slot[q ^ 1].as<int32_t>(); bp->tokens_next = tokbuf[q ^ 1];
      bp->n_cur = st->bm_ncur.as<int32_t>(); bp->n_past_w = st->bm_npast.as<int32_t>(); bp->win_done = st->bm_done.as<int32_t>();
      bp->fin_cnt = st->bm_fin_cnt.as<int32_t>(); bp->fin_tok = st->bm_fin_tok.as<int32_t>(); bp->fin_len = st->bm_fin_len.as<int32_t>();
      bp->fin_sum = st->bm_fin_sum.as<float>();
      bp->part = st->bm_part.as<unsigned>(); bp->tickets = st->bm_ticket.as<unsigned>();
    };
    int steps = 0, last_q = 1;
    Dispatch::run(c->dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      run_decoder_step<T>(st, W, n_prompt);                       // the prompt once per window; its K/V stays in cache rows 0 .. W-1
      HIP_CHECK(hipStreamSynchronize(s));                         // ptoks / np0 are stack-lifetime sources
      ++steps;
      SamplerParams p0; BeamParams b0;
      params(0, &p0, &b0);
      launch_beam_step(p0, b0, W, 1, s);                          // first candidates from the prompt's logits; writes side 1
      // one beam iteration = {decoder step of the W * K rows, top-k per row, update per window}; the token-history and
      // kv_slot double buffers alternate, so TWO graphs are captured (odd and even steps) and replayed in turn
      const bool use_graph = st->graphs_enabled && st->prof_class == 0 && s != nullptr;
      hipGraphExec_t exec[2] = {nullptr, nullptr};
      if (use_graph) {
        for (auto& g : st->beam_graphs)
          if (g.windows == W && g.K == K && g.cus == st->stream_cus && g.invariant == st->batch_invariant && g.persist == st->persist && std::memcmp(&g.spar, &base, sizeof base) == 0) {
            exec[0] = g.exec[0]; exec[1] = g.exec[1];
          }
      }
      if (use_graph && !exec[0]) {
        // room first: the oldest PAIR goes before anything of this call exists (round 2 evicted the front entry between the
        // two captures of a call - an entry whose exec the call might already hold)
        if (st->beam_graphs.size() >= 4) {
          auto& g = st->beam_graphs.front();
          for (int q = 0; q < 2; ++q) {
            if (g.exec[q]) (void)hipGraphExecDestroy(g.exec[q]);
            if (g.graph[q]) (void)hipGraphDestroy(g.graph[q]);
          }
          st->beam_graphs.erase(st->beam_graphs.begin());
        }
        ohw_state::BeamGraph ng;
        std::memset(&ng.spar, 0, sizeof ng.spar);
        auto drop = [&] {
          for (int q = 0; q < 2; ++q) {
            if (ng.exec[q]) (void)hipGraphExecDestroy(ng.exec[q]);
            if (ng.graph[q]) (void)hipGraphDestroy(ng.graph[q]);
          }
        };
        try {
          for (int q = 0; q < 2; ++q) {
            SamplerParams pq; BeamParams bq;
            params(q, &pq, &bq);
            hipStream_t cap = st->own_stream;
            CaptureGate gate;
            struct StreamSwap { ohw_state* st; hipStream_t keep; ~StreamSwap() { st->stream = keep; } } swap{st, st->stream};
            st->stream = cap;
            BMARK("capture begin", q);
            HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
            try {
              run_decoder_step<T>(st, R, 1, st->next_tok.as<int32_t>(), K, bq.kv_slot, bq.win_done);
              BMARK("decoder step captured", q);
              launch_beam_step(pq, bq, W, 0, cap);
              BMARK("beam step captured", q);
            } catch (...) {
              hipGraph_t g = nullptr;
              (void)hipStreamEndCapture(cap, &g);
              if (g) (void)hipGraphDestroy(g);
              throw;
            }
            HIP_CHECK(hipStreamEndCapture(cap, &ng.graph[q]));
            BMARK("capture ended", q);
            HIP_CHECK(hipGraphInstantiate(&ng.exec[q], ng.graph[q], nullptr, nullptr, 0));
            BMARK("instantiated", q);
          }
        } catch (...) {
          drop();
          throw;
        }
        ng.windows = W; ng.K = K; ng.cus = st->stream_cus; ng.invariant = st->batch_invariant; ng.persist = st->persist;
        std::memcpy(&ng.spar, &base, sizeof base);
        st->beam_graphs.push_back(ng);
        ++st->beam_captures;
        exec[0] = ng.exec[0]; exec[1] = ng.exec[1];
      }
      int32_t n_done_host = 0;
      for (int it = 1; it < n_max; ++it) {
        const int q = it & 1;
        if (use_graph) {
          BMARK("graph launch", it);
          HIP_CHECK(hipGraphLaunch(exec[q], s));
          BMARK("graph launched", it);
        } else {
          SamplerParams pq; BeamParams bq;
          params(q, &pq, &bq);
          run_decoder_step<T>(st, R, 1, st->next_tok.as<int32_t>(), K, bq.kv_slot, bq.win_done);
          launch_beam_step(pq, bq, W, 0, s);
        }
        ++steps;
        last_q = q ^ 1;
        if ((it & 7) == 7) {
          HIP_CHECK(hipMemcpyAsync(&n_done_host, st->n_done.p, 4, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipStreamSynchronize(s));
          if (n_done_host >= W) break;
        }
      }
    });
    HIP_CHECK(hipEventRecord(st->ev[5], s));
    // read the finished pools and the live beams back; rank on the host: cumulative log-probability / length
    std::vector<int32_t> fin_cnt((size_t)W), fin_len((size_t)R), fin_tok((size_t)R * MT), live_tok((size_t)R * MT), ncur((size_t)W);
    std::vector<float> fin_sum((size_t)R), live_sum((size_t)R);
    HIP_CHECK(hipMemcpyAsync(fin_cnt.data(), st->bm_fin_cnt.p, (size_t)W * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(fin_len.data(), st->bm_fin_len.p, (size_t)R * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(fin_sum.data(), st->bm_fin_sum.p, (size_t)R * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(fin_tok.data(), st->bm_fin_tok.p, (size_t)R * MT * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(live_tok.data(), tokbuf[last_q], (size_t)R * MT * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(live_sum.data(), st->bm_sum.p, (size_t)R * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(ncur.data(), st->bm_ncur.p, (size_t)W * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    persist_check(st);
    for (int w = 0; w < W; ++w) {
      struct Cand { const int32_t* t; int n; float sum; };
      std::vector<Cand> cands;
      for (int f = 0; f < fin_cnt[(size_t)w]; ++f) cands.push_back({&fin_tok[(size_t)(w * K + f) * MT], fin_len[(size_t)(w * K + f)], fin_sum[(size_t)(w * K + f)]});
      if ((int)cands.size() < K) {
        // not enough finished sequences: the live beams join, most likely first (the published decoder's finalize())
        std::vector<int> order((size_t)K);
        for (int j = 0; j < K; ++j) order[(size_t)j] = j;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return live_sum[(size_t)(w * K + a)] > live_sum[(size_t)(w * K + b)]; });
        for (int j : order) {
          if ((int)cands.size() >= K) break;
          if (!(live_sum[(size_t)(w * K + j)] > -INFINITY)) continue;
          cands.push_back({&live_tok[(size_t)(w * K + j) * MT], ncur[(size_t)w], live_sum[(size_t)(w * K + j)]});
        }
      }
      int best = -1;
      float best_score = -INFINITY;
      for (size_t i = 0; i < cands.size(); ++i) {
        const float score = cands[i].sum / (float)std::max(1, cands[i].n);
        if (best < 0 || score > best_score) { best = (int)i; best_score = score; }
      }
      const int n = best >= 0 ? std::min(cands[(size_t)best].n, max_tokens) : 0;
      res->n_tokens[w] = n;
      if (n) std::memcpy(res->tokens + (size_t)w * max_tokens, cands[(size_t)best].t, (size_t)n * 4);
      if (res->sum_logprob) res->sum_logprob[w] = best >= 0 ? cands[(size_t)best].sum : 0.f;
      if (res->n_finished) res->n_finished[w] = fin_cnt[(size_t)w];
    }
    st->last.decode_steps = steps;
  });
}

int ohw_dbg_counter(const ohw_state* st, const char* name) {
  if (!st || !name) return OHW_E_INVALID_ARG;
  const std::string n = name;
  if (n == "beam_captures") return st->beam_captures;
  if (n == "beam_graphs") return (int)st->beam_graphs.size();
  if (n == "step_captures") return st->step_captures;
  if (n == "step_graphs") return (int)st->step_graphs.size();
  if (n == "persist_launches") return st->persist_launches;
  return OHW_E_INVALID_ARG;
}

int ohw_state_set_persistent(ohw_state* st, int on) {
  if (!st) return OHW_E_INVALID_ARG;
  st->persist = on != 0;
  return OHW_OK;
}

int ohw_state_set_batch_invariant(ohw_state* st, int on) {
  if (!st) return OHW_E_INVALID_ARG;
  st->batch_invariant = on != 0;
  return OHW_OK;
}

int ohw_state_set_logit_bias(ohw_state* st, const float* bias, int n) {
  return guard([&] {
    if (!st) throw Error(OHW_E_INVALID_ARG, "state is null");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    HIP_CHECK(hipStreamSynchronize(st->stream));
    if (!bias) { st->bias_on = false; st->bias_host.clear(); return; }
    if (n != st->ctx->hp.n_vocab) throw Error(OHW_E_INVALID_ARG, "logit bias: n must equal n_vocab");
    if (!st->logit_bias.p) st->logit_bias.alloc((size_t)n * 4);
    HIP_CHECK(hipMemcpy(st->logit_bias.p, bias, (size_t)n * 4, hipMemcpyHostToDevice));
    st->bias_host.assign(bias, bias + n);
    st->bias_on = true;
  });
}

// test entry: the device sampler on caller-supplied rows.  logits [batch][n_vocab] (host), history [batch][hist_stride]
// with n_hist[b] tokens sampled so far.  Returns the token the sampler picks per row (end-of-text included), its
// log-probability, and the first-step no-speech probability (rows with n_hist == 0; 0 elsewhere).
int ohw_dbg_sample(ohw_state* st, const ohw_sample_params* sp, const float* logits, const int32_t* history, int hist_stride,
                   const int32_t* n_hist, int batch, int32_t* tokens_out, float* logprobs_out, float* no_speech_out) {
  return guard([&] {
    if (!st || !sp || !logits || !n_hist || !tokens_out) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (batch < 1 || batch > st->max_batch) throw Error(OHW_E_INVALID_ARG, "dbg_sample: batch exceeds the state's max_batch");
    const ohw_ctx* c = st->ctx;
    const int V = c->hp.n_vocab;
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = st->stream;
    std::vector<int32_t> hist((size_t)batch * st->max_tokens, 0), ncur((size_t)batch);
    for (int b = 0; b < batch; ++b) {
      if (n_hist[b] < 0 || n_hist[b] >= st->max_tokens || n_hist[b] > hist_stride || (n_hist[b] > 0 && !history))
        throw Error(OHW_E_INVALID_ARG, "dbg_sample: bad history length");
      ncur[(size_t)b] = n_hist[b];
      for (int i = 0; i < n_hist[b]; ++i) hist[(size_t)b * st->max_tokens + i] = history[(size_t)b * hist_stride + i];
    }
    HIP_CHECK(hipMemcpy2DAsync(st->logits.p, (size_t)st->logits_ld * 4, logits, (size_t)V * 4, (size_t)V * 4, (size_t)batch, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(st->tokens.p, hist.data(), hist.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(st->n_cur.p, ncur.data(), ncur.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(st->n_past.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->done.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->n_done.p, 0, 16, s));
    HIP_CHECK(hipMemsetAsync(st->sum_lp.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->next_tok.p, 0, (size_t)batch * 4, s));
    HIP_CHECK(hipMemsetAsync(st->nosp_prob.p, 0, (size_t)batch * 4, s));
    SamplerParams spar;
    fill_sampler(st, sp, batch, &spar);
    spar.advance = 0;
    launch_sampler(spar, s);
    std::vector<float> lps((size_t)batch * (st->max_tokens + 1));
    HIP_CHECK(hipMemcpyAsync(tokens_out, st->next_tok.p, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(lps.data(), st->tok_lp.p, lps.size() * 4, hipMemcpyDeviceToHost, s));
    if (no_speech_out) HIP_CHECK(hipMemcpyAsync(no_speech_out, st->nosp_prob.p, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (logprobs_out)
      for (int b = 0; b < batch; ++b) logprobs_out[b] = lps[(size_t)b * (st->max_tokens + 1) + n_hist[b]];
  });
}

int ohw_state_timings(ohw_state* st, ohw_timings* t) {
  return guard([&] {
    if (!st || !t) throw Error(OHW_E_INVALID_ARG, "null argument");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    HIP_CHECK(hipStreamSynchronize(st->stream));
    auto el = [&](int a, int b) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, st->ev[a], st->ev[b]) != hipSuccess) ms = 0.f;
      return ms;
    };
    st->last.mel_ms = el(0, 1);
    st->last.encode_ms = el(2, 3);
    st->last.decode_ms = el(4, 5);
    st->last.total_ms = st->last.mel_ms + st->last.encode_ms + st->last.decode_ms;
    *t = st->last;
  });
}

int ohw_state_profile_begin(ohw_state* st, int kernel_class) {
  return guard([&] {
    if (!st) throw Error(OHW_E_INVALID_ARG, "state is null");
    if (kernel_class < 0 || kernel_class > 8) throw Error(OHW_E_INVALID_ARG, "unknown kernel class");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    st->prof_class = kernel_class;
    st->prof_used = 0;
    st->prof_work = 0.0;
  });
}

int ohw_state_profile_end(ohw_state* st, int64_t* launches, double* total_ms, double* work) {
  return guard([&] {
    if (!st) throw Error(OHW_E_INVALID_ARG, "state is null");
    HIP_CHECK(hipSetDevice(st->ctx->device));
    HIP_CHECK(hipStreamSynchronize(st->stream));
    double ms = 0.0;
    for (size_t i = 0; i + 1 < st->prof_used; i += 2) {
      float t = 0.f;
      HIP_CHECK(hipEventElapsedTime(&t, st->prof_ev[i], st->prof_ev[i + 1]));
      ms += t;
    }
    if (launches) *launches = (int64_t)(st->prof_used / 2);
    if (total_ms) *total_ms = ms;
    if (work) *work = st->prof_work;
    st->prof_class = 0;
    st->prof_used = 0;
  });
}

int ohw_state_fetch(ohw_state* st, const char* what, int batch, float* out, int64_t out_elems) {
  return guard([&] {
    if (!st || !what || !out) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (batch < 1 || batch > st->enc_batch) throw Error(OHW_E_INVALID_ARG, "fetch: batch exceeds the last encode");
    const ohw_hparams& hp = st->ctx->hp;
    HIP_CHECK(hipSetDevice(st->ctx->device));
    hipStream_t s = st->stream;
    const int64_t d = hp.n_audio_state, Tn = hp.n_audio_ctx;
    const std::string w = what;
    DevBuf tmp;
    auto need = [&](int64_t n) { if (out_elems < n) throw Error(OHW_E_INVALID_ARG, "fetch: output buffer too small"); };
    auto from_t = [&](const void* src, int64_t n) {
      need(n);
      tmp.alloc((size_t)n * 4);
      Dispatch::run(st->ctx->dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        launch_to_f32<T>(src, tmp.as<float>(), n, s);
      });
      HIP_CHECK(hipMemcpyAsync(out, tmp.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    };
    auto from_f32 = [&](const void* src, int64_t n) {
      need(n);
      HIP_CHECK(hipMemcpyAsync(out, src, (size_t)n * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    };
    if (w == "mel") { from_f32(st->logmel.p, (int64_t)batch * hp.n_mels * CHUNK_FRAMES); return; }
    if (w == "enc") { from_t(st->enc.p, (int64_t)batch * Tn * d); return; }
    if (w == "stem" || w == "block0") {
      if (!st->taps) throw Error(OHW_E_INVALID_ARG, "fetch: taps are kept only for d_model <= 512");
      from_f32(w == "stem" ? st->tap_stem.p : st->tap_block0.p, (int64_t)batch * Tn * d);
      return;
    }
    if (w == "conv1") {
      // image rows 1..3000 of every window
      need((int64_t)batch * CHUNK_FRAMES * d);
      const int es = 2;
      tmp.alloc((size_t)batch * CHUNK_FRAMES * d * 4);
      DevBuf packed;
      packed.alloc((size_t)batch * CHUNK_FRAMES * d * es);
      for (int b = 0; b < batch; ++b)
        HIP_CHECK(hipMemcpyAsync((char*)packed.p + (size_t)b * CHUNK_FRAMES * d * es, (char*)st->c1.p + ((size_t)b * MEL_ROWS + 1) * d * es,
                                 (size_t)CHUNK_FRAMES * d * es, hipMemcpyDeviceToDevice, s));
      Dispatch::run(st->ctx->dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        launch_to_f32<T>(packed.p, tmp.as<float>(), (int64_t)batch * CHUNK_FRAMES * d, s);
      });
      HIP_CHECK(hipMemcpyAsync(out, tmp.p, (size_t)batch * CHUNK_FRAMES * d * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      return;
    }
    if ((w.rfind("xk", 0) == 0 || w.rfind("xv", 0) == 0) && w.size() > 2) {
      // head-major [B][H][T][64] of layer l -> [B][T][d] on the host
      const int l = std::atoi(w.c_str() + 2);
      if (l < 0 || l >= hp.n_text_layer) throw Error(OHW_E_INVALID_ARG, "fetch: layer out of range");
      const int H = hp.n_text_head;
      const int64_t slab = (int64_t)st->enc_batch * H * Tn * 64;
      const int64_t n = (int64_t)batch * H * Tn * 64;
      need(n);
      std::vector<float> hm((size_t)n);
      tmp.alloc((size_t)n * 4);
      const int which = w[1] == 'k' ? 0 : 1;
      Dispatch::run(st->ctx->dtype, [&](auto* tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        launch_to_f32<T>((const T*)st->xkv.p + (int64_t)(2 * l + which) * slab, tmp.as<float>(), n, s);
      });
      HIP_CHECK(hipMemcpyAsync(hm.data(), tmp.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      for (int b = 0; b < batch; ++b)
        for (int h = 0; h < H; ++h)
          for (int64_t t = 0; t < Tn; ++t)
            std::memcpy(out + ((int64_t)b * Tn + t) * (H * 64) + h * 64, &hm[(size_t)((((int64_t)b * H + h) * Tn + t) * 64)], 64 * 4);
      return;
    }
    throw Error(OHW_E_INVALID_ARG, std::string("fetch: unknown activation '") + what + "'");
  });
}

int ohw_ctx_create_shell(const ohw_hparams* hp, int device, int dtype, ohw_ctx** out) {
  return guard([&] {
    if (!out) throw Error(OHW_E_INVALID_ARG, "out is null");
    *out = nullptr;
    *out = ctx_shell(hp, device, dtype);
  });
}

// blob = [weight arena | mel filterbank], each part padded to 256 bytes
static size_t blob_part(size_t n) { return (n + 255) / 256 * 256; }
size_t ohw_ctx_blob_size(const ohw_ctx* ctx) { return ctx ? blob_part(ctx->arena.bytes) + blob_part(ctx->mel_filters.bytes) : 0; }

int ohw_ctx_blob_export(const ohw_ctx* ctx, void* dst_device, size_t capacity) {
  return guard([&] {
    if (!ctx || !dst_device) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (capacity < ohw_ctx_blob_size(ctx)) throw Error(OHW_E_INVALID_ARG, "blob_export: destination is smaller than ohw_ctx_blob_size");
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMemcpy(dst_device, ctx->arena.p, ctx->arena.bytes, hipMemcpyDeviceToDevice));
    HIP_CHECK(hipMemcpy((char*)dst_device + blob_part(ctx->arena.bytes), ctx->mel_filters.p, ctx->mel_filters.bytes, hipMemcpyDeviceToDevice));
    HIP_CHECK(hipDeviceSynchronize());
  });
}

int ohw_ctx_blob_import(ohw_ctx* ctx, const void* src_device, size_t bytes) {
  return guard([&] {
    if (!ctx || !src_device) throw Error(OHW_E_INVALID_ARG, "null argument");
    if (bytes != ohw_ctx_blob_size(ctx)) throw Error(OHW_E_LOAD_FAILED, "blob_import: size does not match this model's layout (hparams / dtype differ?)");
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMemcpy(ctx->arena.p, src_device, ctx->arena.bytes, hipMemcpyDeviceToDevice));
    HIP_CHECK(hipMemcpy(ctx->mel_filters.p, (const char*)src_device + blob_part(ctx->arena.bytes), ctx->mel_filters.bytes, hipMemcpyDeviceToDevice));
    HIP_CHECK(hipDeviceSynchronize());
  });
}

int ohw_ctx_weight_digest(const ohw_ctx* ctx, int index, char* name_out, uint64_t* digest) {
  return guard([&] {
    if (!ctx || !name_out || !digest || index < 0) throw Error(OHW_E_INVALID_ARG, "bad argument");
    std::vector<std::pair<std::string, const DevBuf*>> bufs;
    auto add = [&](const std::string& n, const DevBuf& b) { bufs.emplace_back(n, &b); };
    add("mel_filters", ctx->mel_filters); add("conv1_w", ctx->conv1_w); add("conv1_b", ctx->conv1_b);
    add("conv2_w", ctx->conv2_w); add("conv2_b", ctx->conv2_b); add("enc_pos", ctx->enc_pos);
    for (size_t i = 0; i < ctx->enc.size(); ++i) {
      const EncLayerW& l = ctx->enc[i];
      const std::string p = "enc" + std::to_string(i) + ".";
      add(p + "ln1.g", l.ln1.g); add(p + "ln1.b", l.ln1.b); add(p + "wqkv", l.wqkv); add(p + "bqkv", l.bqkv);
      add(p + "wo", l.wo); add(p + "bo", l.bo); add(p + "ln2.g", l.ln2.g); add(p + "ln2.b", l.ln2.b);
      add(p + "w1", l.w1); add(p + "b1", l.b1); add(p + "w2", l.w2); add(p + "b2", l.b2);
    }
    add("ln_post.g", ctx->ln_post.g); add("ln_post.b", ctx->ln_post.b); add("xkv_w", ctx->xkv_w); add("xkv_b", ctx->xkv_b);
    add("dec_pos", ctx->dec_pos); add("emb", ctx->emb);
    for (size_t i = 0; i < ctx->dec.size(); ++i) {
      const DecLayerW& l = ctx->dec[i];
      const std::string p = "dec" + std::to_string(i) + ".";
      add(p + "ln1.g", l.ln1.g); add(p + "ln1.b", l.ln1.b); add(p + "wqkv", l.wqkv); add(p + "bqkv", l.bqkv);
      add(p + "wo", l.wo); add(p + "bo", l.bo); add(p + "lnx.g", l.lnx.g); add(p + "lnx.b", l.lnx.b);
      add(p + "wxq", l.wxq); add(p + "bxq", l.bxq); add(p + "wxo", l.wxo); add(p + "bxo", l.bxo);
      add(p + "ln2.g", l.ln2.g); add(p + "ln2.b", l.ln2.b); add(p + "w1", l.w1); add(p + "b1", l.b1); add(p + "w2", l.w2); add(p + "b2", l.b2);
    }
    add("dec_ln.g", ctx->dec_ln.g); add("dec_ln.b", ctx->dec_ln.b);
    if ((size_t)index >= bufs.size()) throw Error(OHW_E_INVALID_ARG, "index past the last weight buffer");
    HIP_CHECK(hipSetDevice(ctx->device));
    const DevBuf& b = *bufs[(size_t)index].second;
    std::vector<unsigned char> host(b.bytes);
    HIP_CHECK(hipMemcpy(host.data(), b.p, b.bytes, hipMemcpyDeviceToHost));
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : host) { h ^= c; h *= 0x100000001b3ull; }
    *digest = h;
    std::snprintf(name_out, 64, "%s", bufs[(size_t)index].first.c_str());
  });
}

int ohw_dbg_gemm(int dtype, const void* A, const void* W, const float* bias, void* out, int64_t M, int64_t N, int64_t K, int epilogue,
                 void* stream) {
  return guard([&] {
    GemmParams g{};
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.a_batch_stride = 0; g.rows_per_batch = M; g.ldc = N; g.c_batch_stride = 0;
    if (epilogue != EPI_BIAS_T && epilogue != EPI_BIAS_GELU_T && epilogue != EPI_BIAS_RESID_F32 && epilogue != EPI_F32)
      throw Error(OHW_E_INVALID_ARG, "dbg_gemm: epilogue must be 0, 1, 2 or 4");
    Dispatch::run(dtype, [&](auto* tag) {
      using T = std::remove_pointer_t<decltype(tag)>;
      launch_gemm<T>(g, epilogue, (hipStream_t)stream);
    });
  });
}

int ohw_dbg_attention(int dtype, const void* qkv, void* out, int batch, int T, int n_head, void* stream) {
  return guard([&] {
    Dispatch::run(dtype, [&](auto* tag) {
      using TT = std::remove_pointer_t<decltype(tag)>;
      launch_encoder_attention<TT>(qkv, out, batch, T, n_head, (hipStream_t)stream);
    });
  });
}

}  // extern "C"
