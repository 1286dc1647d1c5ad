// model.hip — model creation: ggml model-file reader and the procedural generator, both feeding one
// "placer" that repacks each named tensor into its HBM layout.
//
// Replaces WhisperContext::new_with_params (reference src/engine/whisper.rs:156-160).  File format:
// SURVEY.md Appendix A (the stock ggml-*.bin files the reference downloads, src/engine/whisper.rs:71-102).
#include <sys/stat.h>

#include <cmath>
#include <cstring>
#include <memory>

#include "kernels.hpp"
#include "model.hpp"

namespace ohw {

static uint32_t fmix32_h(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
static uint32_t fnv1a32(const std::string& s) {
  uint32_t h = 0x811C9DC5u;
  for (unsigned char c : s) { h ^= c; h *= 0x01000193u; }
  return h;
}
static uint32_t tensor_key(uint32_t seed, const std::string& name) { return fmix32_h(fnv1a32(name) ^ (seed * 0x9E3779B9u)); }

enum { KIND_LINEAR = 0, KIND_BIAS = 1, KIND_GAMMA = 2, KIND_EMBED = 3, KIND_POS = 4 };
static void kind_scale_offset(int kind, int fan_in, float* scale, float* offset) {
  *offset = 0.0f;
  switch (kind) {
    case KIND_LINEAR: *scale = (float)std::sqrt(3.0 / fan_in); break;
    case KIND_BIAS: *scale = 0.1f; break;
    case KIND_GAMMA: *scale = 0.1f; *offset = 1.0f; break;
    case KIND_EMBED: *scale = (float)std::sqrt(3.0 / fan_in) * 4.0f; break;
    default: *scale = 0.1f; break;
  }
}

void set_special_tokens(ohw_ctx* c) {
  // whisper.cpp vocab layout
  const int n_vocab = c->hp.n_vocab;
  const bool multilingual = n_vocab >= 51865;
  ohw_special_tokens& t = c->tok;
  t.n_langs = n_vocab - 51765 - (multilingual ? 1 : 0);
  t.eot = 50256; t.sot = 50257; t.translate = 50357; t.transcribe = 50358; t.solm = 50359; t.prev = 50360;
  t.nosp = 50361; t.no_timestamps = 50362; t.timestamp_begin = 50363;
  if (multilingual) {
    t.eot++; t.sot++;
    const int dt = t.n_langs - 98;
    t.translate += dt; t.transcribe += dt; t.solm += dt; t.prev += dt; t.nosp += dt; t.no_timestamps += dt; t.timestamp_begin += dt;
  }
  t.blank = -1;
  for (size_t i = 0; i < c->vocab.size(); ++i)
    if (c->vocab[i] == " ") { t.blank = (int32_t)i; break; }
}

static void check_hparams(const ohw_hparams& hp) {
  auto bad = [](const char* what) { throw Error(OHW_E_LOAD_FAILED, std::string("unsupported model dimensions: ") + what); };
  if (hp.n_audio_state <= 0 || hp.n_audio_state % 128 != 0) bad("n_audio_state must be a multiple of 128");
  if (hp.n_text_state != hp.n_audio_state) bad("n_text_state != n_audio_state");
  if (hp.n_audio_head * 64 != hp.n_audio_state || hp.n_text_head * 64 != hp.n_text_state) bad("d_head must be 64");
  if (hp.n_audio_ctx != 1500) bad("n_audio_ctx must be 1500");
  if (hp.n_text_ctx <= 0 || hp.n_text_ctx > 448) bad("n_text_ctx must be <= 448");
  if (hp.n_mels <= 0 || hp.n_mels > MEL_CPAD) bad("n_mels must be <= 128");
  if (hp.n_vocab < 51864 || hp.n_vocab > 52000) bad("n_vocab");
  if (hp.n_audio_layer <= 0 || hp.n_text_layer <= 0 || hp.n_audio_layer > 64 || hp.n_text_layer > 64) bad("layer count");
}

// ------------------------------------------------------------------------------------------------
// placer: named tensor (f32 staging copy on the device) -> final layout
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Placer {
  ohw_ctx* c;
  hipStream_t s;
  int placed = 0;
  explicit Placer(ohw_ctx* ctx, hipStream_t st) : c(ctx), s(st) {}

  // Decoder pre-LayerNorms are folded into the linear layer they feed (launch_fold_ln): the decode GEMMs then
  // only normalise.  Tensors arrive in file order, so the f32 copies of the five affected matrices of a layer
  // wait here until that layer's LayerNorms and biases are in; then: fold, repack, free.
  enum { F_WQ, F_WK, F_WV, F_WXQ, F_W1, F_BQ, F_BV, F_BXQ, F_B1, F_LN1G, F_LN1B, F_LNXG, F_LNXB, F_LN2G, F_LN2B, F_COUNT };
  struct PendingLayer { DevBuf w[5]; unsigned seen = 0; };
  std::vector<PendingLayer> pending;

  void stash(int li, int slot, const float* src, int64_t n) {
    if (pending.empty()) pending.resize(c->dec.size());
    pending[li].w[slot].alloc((size_t)n * 4);
    HIP_CHECK(hipMemcpyAsync(pending[li].w[slot].p, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  }
  void arrived(int li, int what) {
    if (pending.empty()) pending.resize(c->dec.size());
    PendingLayer& pl = pending[li];
    pl.seen |= 1u << what;
    if (pl.seen != (1u << F_COUNT) - 1) return;
    DecLayerW& l = c->dec[li];
    const int64_t dt = c->hp.n_text_state;
    for (int q = 0; q < 3; ++q) {
      launch_fold_ln((float*)pl.w[q].p, l.bqkv.as<float>() + q * dt, l.ln1.g.as<float>(), l.ln1.b.as<float>(), dt, dt, s);
      launch_repack_tiled<T>((float*)pl.w[q].p, (T*)l.wqkv.p + q * dt * dt, dt, dt, dt, s);
    }
    launch_fold_ln((float*)pl.w[F_WXQ].p, l.bxq.as<float>(), l.lnx.g.as<float>(), l.lnx.b.as<float>(), dt, dt, s);
    launch_repack_tiled<T>((float*)pl.w[F_WXQ].p, l.wxq.p, dt, dt, dt, s);
    launch_fold_ln((float*)pl.w[F_W1].p, l.b1.as<float>(), l.ln2.g.as<float>(), l.ln2.b.as<float>(), 4 * dt, dt, s);
    launch_repack_tiled<T>((float*)pl.w[F_W1].p, l.w1.p, 4 * dt, 4 * dt, dt, s);
    launch_tiled_rowsum<T>(l.wqkv.p, l.sqkv.as<float>(), 3 * dt, dt, s);
    launch_tiled_rowsum<T>(l.wxq.p, l.sxq.as<float>(), dt, dt, s);
    launch_tiled_rowsum<T>(l.w1.p, l.s1.as<float>(), 4 * dt, dt, s);
    HIP_CHECK(hipStreamSynchronize(s));
    for (auto& b : pl.w) b.release();
  }

  static void f32_copy(DevBuf& dst, const float* src, int64_t n, hipStream_t s) {
    HIP_CHECK(hipMemcpyAsync(dst.p, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  }

  void allocate() {
    // two passes over the same list: size the arena, then hand out 256-byte aligned views
    size_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
      size_t off = 0;
      if (pass == 1) {
        c->arena.alloc(total, true);
        c->weight_bytes = total;
      }
      auto A = [&](DevBuf& b, size_t bytes, bool = false) {
        const size_t sz = (bytes + 255) / 256 * 256;
        if (pass == 1) b.view((char*)c->arena.p + off, bytes);
        off += sz;
      };
      layout(A);
      total = off;
    }
  }

  template <typename Alloc>
  void layout(Alloc&& A) {
    const ohw_hparams& hp = c->hp;
    const int64_t d = hp.n_audio_state, dt = hp.n_text_state, L = hp.n_text_layer;
    const size_t e = 2;
    A(c->conv1_w, (size_t)d * 3 * MEL_CPAD * e); A(c->conv1_b, (size_t)d * 4);
    A(c->conv2_w, (size_t)d * 3 * d * e); A(c->conv2_b, (size_t)d * 4);
    A(c->enc_pos, (size_t)hp.n_audio_ctx * d * 4);
    c->enc.resize(hp.n_audio_layer);
    for (auto& l : c->enc) {
      A(l.ln1.g, d * 4); A(l.ln1.b, d * 4); A(l.ln2.g, d * 4); A(l.ln2.b, d * 4);
      A(l.wqkv, (size_t)3 * d * d * e); A(l.bqkv, (size_t)3 * d * 4, true);
      A(l.wo, (size_t)d * d * e); A(l.bo, d * 4);
      A(l.w1, (size_t)4 * d * d * e); A(l.b1, (size_t)4 * d * 4);
      A(l.w2, (size_t)4 * d * d * e); A(l.b2, d * 4);
    }
    A(c->ln_post.g, d * 4); A(c->ln_post.b, d * 4);
    A(c->xkv_w, (size_t)2 * L * dt * d * e); A(c->xkv_b, (size_t)2 * L * dt * 4, true);
    A(c->dec_pos, (size_t)hp.n_text_ctx * dt * 4);
    c->v_pad = ((int64_t)hp.n_vocab + 15) / 16 * 16;
    A(c->emb, (size_t)c->v_pad * dt * e);
    c->dec.resize(hp.n_text_layer);
    for (auto& l : c->dec) {
      A(l.ln1.g, dt * 4); A(l.ln1.b, dt * 4); A(l.lnx.g, dt * 4); A(l.lnx.b, dt * 4); A(l.ln2.g, dt * 4); A(l.ln2.b, dt * 4);
      A(l.wqkv, (size_t)3 * dt * dt * e); A(l.bqkv, (size_t)3 * dt * 4, true);
      A(l.wo, (size_t)dt * dt * e); A(l.bo, dt * 4);
      A(l.wxq, (size_t)dt * dt * e); A(l.bxq, dt * 4);
      A(l.wxo, (size_t)dt * dt * e); A(l.bxo, dt * 4);
      A(l.w1, (size_t)4 * dt * dt * e); A(l.b1, (size_t)4 * dt * 4);
      A(l.w2, (size_t)4 * dt * dt * e); A(l.b2, dt * 4);
      A(l.sqkv, (size_t)3 * dt * 4); A(l.sxq, dt * 4); A(l.s1, (size_t)4 * dt * 4);
    }
    A(c->dec_ln.g, dt * 4); A(c->dec_ln.b, dt * 4);
  }

  // returns false for names the engine does not use
  bool place(const std::string& name, const std::vector<int64_t>& dims, const float* src) {
    const ohw_hparams& hp = c->hp;
    const int64_t d = hp.n_audio_state, dt = hp.n_text_state;
    int64_t n = 1;
    for (auto v : dims) n *= v;
    auto expect = [&](int64_t want) {
      if (n != want) throw Error(OHW_E_LOAD_FAILED, "tensor " + name + " has an unexpected size");
    };
    auto vec = [&](DevBuf& dst, int64_t want, int64_t off = 0) {
      expect(want);
      HIP_CHECK(hipMemcpyAsync((float*)dst.p + off, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    };
    auto plain = [&](DevBuf& dst, int64_t rows, int64_t cols, int64_t row_off = 0) {
      expect(rows * cols);
      launch_convert_rows<T>(src, (T*)dst.p + row_off * cols, rows, cols, cols, s);
    };
    auto tiled = [&](DevBuf& dst, int64_t rows, int64_t cols, int64_t row_off = 0, int64_t rows_pad = -1) {
      expect(rows * cols);
      if (rows_pad < 0) rows_pad = rows;
      launch_repack_tiled<T>(src, (T*)dst.p + row_off * cols, rows, rows_pad, cols, s);
    };
    ++placed;
    if (name == "encoder.positional_embedding") { vec(c->enc_pos, (int64_t)hp.n_audio_ctx * d); return true; }
    if (name == "encoder.conv1.weight") { expect(d * hp.n_mels * 3); launch_repack_conv<T>(src, c->conv1_w.p, d, hp.n_mels, MEL_CPAD, s); return true; }
    if (name == "encoder.conv1.bias") { vec(c->conv1_b, d); return true; }
    if (name == "encoder.conv2.weight") { expect(d * d * 3); launch_repack_conv<T>(src, c->conv2_w.p, d, d, d, s); return true; }
    if (name == "encoder.conv2.bias") { vec(c->conv2_b, d); return true; }
    if (name == "encoder.ln_post.weight") { vec(c->ln_post.g, d); return true; }
    if (name == "encoder.ln_post.bias") { vec(c->ln_post.b, d); return true; }
    if (name == "decoder.positional_embedding") { vec(c->dec_pos, (int64_t)hp.n_text_ctx * dt); return true; }
    if (name == "decoder.token_embedding.weight") { tiled(c->emb, hp.n_vocab, dt, 0, c->v_pad); return true; }
    if (name == "decoder.ln.weight") { vec(c->dec_ln.g, dt); return true; }
    if (name == "decoder.ln.bias") { vec(c->dec_ln.b, dt); return true; }
    int li = -1;
    char leaf[96] = {0};
    if (sscanf(name.c_str(), "encoder.blocks.%d.%95s", &li, leaf) == 2) {
      if (li < 0 || li >= hp.n_audio_layer) throw Error(OHW_E_LOAD_FAILED, "layer index out of range: " + name);
      EncLayerW& l = c->enc[li];
      const std::string lf = leaf;
      if (lf == "attn_ln.weight") { vec(l.ln1.g, d); return true; }
      if (lf == "attn_ln.bias") { vec(l.ln1.b, d); return true; }
      if (lf == "attn.query.weight") { plain(l.wqkv, d, d, 0); return true; }
      if (lf == "attn.query.bias") { vec(l.bqkv, d, 0); return true; }
      if (lf == "attn.key.weight") { plain(l.wqkv, d, d, d); return true; }
      if (lf == "attn.value.weight") { plain(l.wqkv, d, d, 2 * d); return true; }
      if (lf == "attn.value.bias") { vec(l.bqkv, d, 2 * d); return true; }
      if (lf == "attn.out.weight") { plain(l.wo, d, d); return true; }
      if (lf == "attn.out.bias") { vec(l.bo, d); return true; }
      if (lf == "mlp_ln.weight") { vec(l.ln2.g, d); return true; }
      if (lf == "mlp_ln.bias") { vec(l.ln2.b, d); return true; }
      if (lf == "mlp.0.weight") { plain(l.w1, 4 * d, d); return true; }
      if (lf == "mlp.0.bias") { vec(l.b1, 4 * d); return true; }
      if (lf == "mlp.2.weight") { plain(l.w2, d, 4 * d); return true; }
      if (lf == "mlp.2.bias") { vec(l.b2, d); return true; }
    } else if (sscanf(name.c_str(), "decoder.blocks.%d.%95s", &li, leaf) == 2) {
      if (li < 0 || li >= hp.n_text_layer) throw Error(OHW_E_LOAD_FAILED, "layer index out of range: " + name);
      DecLayerW& l = c->dec[li];
      const std::string lf = leaf;
      if (lf == "attn_ln.weight") { vec(l.ln1.g, dt); arrived(li, F_LN1G); return true; }
      if (lf == "attn_ln.bias") { vec(l.ln1.b, dt); arrived(li, F_LN1B); return true; }
      if (lf == "attn.query.weight") { expect(dt * dt); stash(li, F_WQ, src, n); arrived(li, F_WQ); return true; }
      if (lf == "attn.query.bias") { vec(l.bqkv, dt, 0); arrived(li, F_BQ); return true; }
      if (lf == "attn.key.weight") { expect(dt * dt); stash(li, F_WK, src, n); arrived(li, F_WK); return true; }
      if (lf == "attn.value.weight") { expect(dt * dt); stash(li, F_WV, src, n); arrived(li, F_WV); return true; }
      if (lf == "attn.value.bias") { vec(l.bqkv, dt, 2 * dt); arrived(li, F_BV); return true; }
      if (lf == "attn.out.weight") { tiled(l.wo, dt, dt); return true; }
      if (lf == "attn.out.bias") { vec(l.bo, dt); return true; }
      if (lf == "cross_attn_ln.weight") { vec(l.lnx.g, dt); arrived(li, F_LNXG); return true; }
      if (lf == "cross_attn_ln.bias") { vec(l.lnx.b, dt); arrived(li, F_LNXB); return true; }
      if (lf == "cross_attn.query.weight") { expect(dt * dt); stash(li, F_WXQ, src, n); arrived(li, F_WXQ); return true; }
      if (lf == "cross_attn.query.bias") { vec(l.bxq, dt); arrived(li, F_BXQ); return true; }
      if (lf == "cross_attn.key.weight") { plain(c->xkv_w, dt, d, (int64_t)(2 * li) * dt); return true; }
      if (lf == "cross_attn.value.weight") { plain(c->xkv_w, dt, d, (int64_t)(2 * li + 1) * dt); return true; }
      if (lf == "cross_attn.value.bias") { vec(c->xkv_b, dt, (int64_t)(2 * li + 1) * dt); return true; }
      if (lf == "cross_attn.out.weight") { tiled(l.wxo, dt, dt); return true; }
      if (lf == "cross_attn.out.bias") { vec(l.bxo, dt); return true; }
      if (lf == "mlp_ln.weight") { vec(l.ln2.g, dt); arrived(li, F_LN2G); return true; }
      if (lf == "mlp_ln.bias") { vec(l.ln2.b, dt); arrived(li, F_LN2B); return true; }
      if (lf == "mlp.0.weight") { expect(4 * dt * dt); stash(li, F_W1, src, n); arrived(li, F_W1); return true; }
      if (lf == "mlp.0.bias") { vec(l.b1, 4 * dt); arrived(li, F_B1); return true; }
      if (lf == "mlp.2.weight") { tiled(l.w2, dt, 4 * dt); return true; }
      if (lf == "mlp.2.bias") { vec(l.b2, dt); return true; }
    }
    --placed;
    return false;
  }
};

static int expected_tensor_count(const ohw_hparams& hp) { return 5 + 15 * hp.n_audio_layer + 2 + 2 + 24 * hp.n_text_layer + 2; }

static void upload_front_end(ohw_ctx* c, const std::vector<float>& filters) {
  c->mel_filters.alloc(filters.size() * 4);
  HIP_CHECK(hipMemcpy(c->mel_filters.p, filters.data(), filters.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> tw(2 * N_FFT), win(N_FFT);
  for (int i = 0; i < N_FFT; ++i) {
    tw[i] = (float)std::cos(2.0 * M_PI * i / N_FFT);
    tw[N_FFT + i] = (float)std::sin(2.0 * M_PI * i / N_FFT);
    win[i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / N_FFT));
  }
  c->twiddle.alloc(tw.size() * 4);
  c->window.alloc(win.size() * 4);
  HIP_CHECK(hipMemcpy(c->twiddle.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(c->window.p, win.data(), win.size() * 4, hipMemcpyHostToDevice));
}

static void select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) throw Error(OHW_E_NO_GPU, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n) throw Error(OHW_E_NO_GPU, "requested HIP device index is out of range");
  HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    throw Error(OHW_E_NO_GPU, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
}

// ------------------------------------------------------------------------------------------------
// ggml file
// ------------------------------------------------------------------------------------------------
struct FileReader {
  FILE* f = nullptr;
  explicit FileReader(const char* path) { f = fopen(path, "rb"); }
  ~FileReader() { if (f) fclose(f); }
  void read(void* dst, size_t n) {
    if (fread(dst, 1, n, f) != n) throw Error(OHW_E_LOAD_FAILED, "model file is truncated");
  }
  template <typename U> U get() { U v; read(&v, sizeof v); return v; }
};

template <typename T>
static void load_file_typed(ohw_ctx* c, FileReader& fr) {
  hipStream_t s = nullptr;
  Placer<T> placer(c, s);
  placer.allocate();
  DevBuf raw, staging;
  std::vector<unsigned char> host;
  for (;;) {
    int32_t hdr[3];
    if (fread(hdr, 4, 3, fr.f) != 3) break;
    if (hdr[0] < 1 || hdr[0] > 4 || hdr[1] <= 0 || hdr[1] > 255 || (hdr[2] != 0 && hdr[2] != 1))
      throw Error(OHW_E_LOAD_FAILED, "unsupported tensor header (only f32 / f16 tensors are supported)");
    int32_t dims_r[4] = {1, 1, 1, 1};
    fr.read(dims_r, 4 * (size_t)hdr[0]);
    std::string name((size_t)hdr[1], '\0');
    fr.read(&name[0], (size_t)hdr[1]);
    std::vector<int64_t> dims;
    int64_t n = 1;
    for (int i = hdr[0] - 1; i >= 0; --i) { dims.push_back(dims_r[i]); n *= dims_r[i]; }
    if (n <= 0 || n > ((int64_t)1 << 31)) throw Error(OHW_E_LOAD_FAILED, "tensor " + name + " has a bad size");
    const size_t esz = hdr[2] == 1 ? 2 : 4;
    host.resize((size_t)n * esz);
    fr.read(host.data(), host.size());
    if (staging.bytes < (size_t)n * 4) staging.alloc((size_t)n * 4);
    if (hdr[2] == 1) {
      if (raw.bytes < (size_t)n * 2) raw.alloc((size_t)n * 2);
      HIP_CHECK(hipMemcpy(raw.p, host.data(), host.size(), hipMemcpyHostToDevice));
      launch_f16_to_f32(raw.p, staging.as<float>(), n, s);
    } else {
      HIP_CHECK(hipMemcpy(staging.p, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    placer.place(name, dims, staging.as<float>());
    HIP_CHECK(hipStreamSynchronize(s));
  }
  if (placer.placed != expected_tensor_count(c->hp))
    throw Error(OHW_E_LOAD_FAILED, "model file is missing tensors: found " + std::to_string(placer.placed) + " of " +
                                       std::to_string(expected_tensor_count(c->hp)));
}

ohw_ctx* ctx_from_file(const char* path, int device, int dtype) {
  struct stat st;
  if (!path || stat(path, &st) != 0) throw Error(OHW_E_MODEL_NOT_FOUND, std::string("model not found at ") + (path ? path : "(null)"));
  if (dtype != OHW_DTYPE_BF16 && dtype != OHW_DTYPE_F16 && dtype != OHW_DTYPE_AUTO)
    throw Error(OHW_E_INVALID_ARG, "dtype must be OHW_DTYPE_AUTO, OHW_DTYPE_BF16 or OHW_DTYPE_F16");
  FileReader fr(path);
  if (!fr.f) throw Error(OHW_E_LOAD_FAILED, std::string("cannot open ") + path);
  if (fr.get<uint32_t>() != 0x67676d6cu) throw Error(OHW_E_LOAD_FAILED, "not a ggml model file (bad magic)");
  std::unique_ptr<ohw_ctx> c(new ohw_ctx());
  fr.read(&c->hp, sizeof c->hp);
  check_hparams(c->hp);
  // AUTO: the file's own weight precision.  The stock ggml-*.bin files store f16 (ftype 1): f16 keeps them EXACT (bf16 would
  // drop 3 mantissa bits of every weight for +1 % throughput); f32 files (ftype 0) get bf16's range
  if (dtype == OHW_DTYPE_AUTO) dtype = c->hp.ftype == 1 ? OHW_DTYPE_F16 : OHW_DTYPE_BF16;
  const int32_t n_mel = fr.get<int32_t>(), n_fft = fr.get<int32_t>();
  if (n_mel != c->hp.n_mels || n_fft != N_FREQ) throw Error(OHW_E_LOAD_FAILED, "mel filterbank shape mismatch");
  std::vector<float> filters((size_t)n_mel * n_fft);
  fr.read(filters.data(), filters.size() * 4);
  const int32_t n_tok = fr.get<int32_t>();
  if (n_tok < 0 || n_tok > c->hp.n_vocab) throw Error(OHW_E_LOAD_FAILED, "bad vocabulary size");
  c->vocab.resize((size_t)n_tok);
  for (auto& w : c->vocab) {
    const uint32_t len = fr.get<uint32_t>();
    if (len > 4096) throw Error(OHW_E_LOAD_FAILED, "bad vocabulary entry");
    w.resize(len);
    if (len) fr.read(&w[0], len);
  }
  select_device(device);
  c->device = device;
  c->dtype = dtype;
  set_special_tokens(c.get());
  upload_front_end(c.get(), filters);
  if (dtype == OHW_DTYPE_BF16) load_file_typed<bf16_t>(c.get(), fr);
  else load_file_typed<f16_t>(c.get(), fr);
  HIP_CHECK(hipDeviceSynchronize());
  return c.release();
}

// ------------------------------------------------------------------------------------------------
// procedural model (spec: openhush_amd/synth.py)
// ------------------------------------------------------------------------------------------------
static double hz_to_mel(double f) { return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4)) : 3.0 * f / 200.0; }
static double mel_to_hz(double m) { return m >= 15.0 ? 1000.0 * std::exp((std::log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; }
static std::vector<float> slaney_filters(int n_mels) {
  std::vector<double> hz((size_t)n_mels + 2);
  const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(8000.0);
  for (int i = 0; i < n_mels + 2; ++i) hz[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
  std::vector<float> out((size_t)n_mels * N_FREQ);
  for (int i = 0; i < n_mels; ++i) {
    const double enorm = 2.0 / (hz[i + 2] - hz[i]);
    for (int k = 0; k < N_FREQ; ++k) {
      const double fr = 8000.0 * k / (N_FREQ - 1);
      const double lo = (fr - hz[i]) / (hz[i + 1] - hz[i]), up = (hz[i + 2] - fr) / (hz[i + 2] - hz[i + 1]);
      const double w = lo < up ? lo : up;
      out[(size_t)i * N_FREQ + k] = (float)((w > 0 ? w : 0) * enorm);
    }
  }
  return out;
}

template <typename T>
static void synth_typed(ohw_ctx* c, uint32_t seed) {
  hipStream_t s = nullptr;
  Placer<T> placer(c, s);
  placer.allocate();
  const ohw_hparams& hp = c->hp;
  const int64_t d = hp.n_audio_state, dt = hp.n_text_state;
  const bool f16 = hp.ftype == 1;
  DevBuf staging;
  staging.alloc((size_t)hp.n_vocab * dt * 4);
  auto gen = [&](const std::string& name, std::vector<int64_t> dims, int kind, int64_t fan_in, bool as_f16) {
    int64_t n = 1;
    for (auto v : dims) n *= v;
    float scale, offset;
    kind_scale_offset(kind, (int)fan_in, &scale, &offset);
    launch_synth_fill(staging.as<float>(), n, tensor_key(seed, name), scale, offset, as_f16 ? 1 : 0, s);
    if (!placer.place(name, dims, staging.as<float>())) throw Error(OHW_E_LOAD_FAILED, "internal: unplaced tensor " + name);
    HIP_CHECK(hipStreamSynchronize(s));
  };
  auto lin = [&](const std::string& p, int64_t n_out, int64_t n_in, bool bias) {
    gen(p + ".weight", {n_out, n_in}, KIND_LINEAR, n_in, f16);
    if (bias) gen(p + ".bias", {n_out}, KIND_BIAS, 1, false);
  };
  auto ln = [&](const std::string& p, int64_t n) {
    gen(p + ".weight", {n}, KIND_GAMMA, 1, false);
    gen(p + ".bias", {n}, KIND_BIAS, 1, false);
  };
  {  // sinusoidal encoder positions, computed in double on the host like the published model
    std::vector<float> pos((size_t)hp.n_audio_ctx * d);
    const double inc = std::log(10000.0) / (double)(d / 2 - 1);
    for (int64_t p = 0; p < hp.n_audio_ctx; ++p)
      for (int64_t ch = 0; ch < d / 2; ++ch) {
        const double a = (double)p * std::exp(-inc * (double)ch);
        pos[(size_t)(p * d + ch)] = (float)std::sin(a);
        pos[(size_t)(p * d + d / 2 + ch)] = (float)std::cos(a);
      }
    HIP_CHECK(hipMemcpy(staging.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
    placer.place("encoder.positional_embedding", {hp.n_audio_ctx, d}, staging.as<float>());
    HIP_CHECK(hipStreamSynchronize(s));
  }
  gen("encoder.conv1.weight", {d, hp.n_mels, 3}, KIND_LINEAR, 3 * hp.n_mels, f16);
  gen("encoder.conv1.bias", {d}, KIND_BIAS, 1, false);
  gen("encoder.conv2.weight", {d, d, 3}, KIND_LINEAR, 3 * d, f16);
  gen("encoder.conv2.bias", {d}, KIND_BIAS, 1, false);
  for (int i = 0; i < hp.n_audio_layer; ++i) {
    const std::string p = "encoder.blocks." + std::to_string(i) + ".";
    ln(p + "attn_ln", d);
    lin(p + "attn.query", d, d, true); lin(p + "attn.key", d, d, false); lin(p + "attn.value", d, d, true); lin(p + "attn.out", d, d, true);
    ln(p + "mlp_ln", d);
    lin(p + "mlp.0", 4 * d, d, true); lin(p + "mlp.2", d, 4 * d, true);
  }
  ln("encoder.ln_post", d);
  gen("decoder.positional_embedding", {hp.n_text_ctx, dt}, KIND_POS, 1, false);
  gen("decoder.token_embedding.weight", {hp.n_vocab, dt}, KIND_EMBED, dt, f16);
  for (int i = 0; i < hp.n_text_layer; ++i) {
    const std::string p = "decoder.blocks." + std::to_string(i) + ".";
    ln(p + "attn_ln", dt);
    lin(p + "attn.query", dt, dt, true); lin(p + "attn.key", dt, dt, false); lin(p + "attn.value", dt, dt, true); lin(p + "attn.out", dt, dt, true);
    ln(p + "cross_attn_ln", dt);
    lin(p + "cross_attn.query", dt, dt, true); lin(p + "cross_attn.key", dt, d, false); lin(p + "cross_attn.value", dt, d, true);
    lin(p + "cross_attn.out", dt, dt, true);
    ln(p + "mlp_ln", dt);
    lin(p + "mlp.0", 4 * dt, dt, true); lin(p + "mlp.2", dt, 4 * dt, true);
  }
  ln("decoder.ln", dt);
  if (placer.placed != expected_tensor_count(hp)) throw Error(OHW_E_LOAD_FAILED, "internal: synthetic tensor count mismatch");
}

ohw_ctx* ctx_synthetic(const ohw_hparams* hp, uint32_t seed, int device, int dtype) {
  if (!hp) throw Error(OHW_E_INVALID_ARG, "hparams is null");
  if (dtype != OHW_DTYPE_BF16 && dtype != OHW_DTYPE_F16) throw Error(OHW_E_INVALID_ARG, "dtype must be OHW_DTYPE_BF16 or OHW_DTYPE_F16");
  check_hparams(*hp);
  std::unique_ptr<ohw_ctx> c(new ohw_ctx());
  c->hp = *hp;
  select_device(device);
  c->device = device;
  c->dtype = dtype;
  const int n_text = hp->n_vocab >= 51865 ? 50257 : 50256;
  c->vocab.resize((size_t)n_text);
  for (int i = 0; i < n_text; ++i) c->vocab[(size_t)i] = i == 220 ? std::string(" ") : " w" + std::to_string(i);
  set_special_tokens(c.get());
  upload_front_end(c.get(), slaney_filters(hp->n_mels));
  if (dtype == OHW_DTYPE_BF16) synth_typed<bf16_t>(c.get(), seed);
  else synth_typed<f16_t>(c.get(), seed);
  HIP_CHECK(hipDeviceSynchronize());
  return c.release();
}

// a context with every buffer allocated and NO weights in it: the receiving end of a weight broadcast
// (ohw_ctx_blob_import).  Vocabulary: placeholder strings (the rank that loaded the file detokenises).
ohw_ctx* ctx_shell(const ohw_hparams* hp, int device, int dtype) {
  if (!hp) throw Error(OHW_E_INVALID_ARG, "hparams is null");
  if (dtype != OHW_DTYPE_BF16 && dtype != OHW_DTYPE_F16) throw Error(OHW_E_INVALID_ARG, "dtype must be OHW_DTYPE_BF16 or OHW_DTYPE_F16");
  check_hparams(*hp);
  std::unique_ptr<ohw_ctx> c(new ohw_ctx());
  c->hp = *hp;
  select_device(device);
  c->device = device;
  c->dtype = dtype;
  const int n_text = hp->n_vocab >= 51865 ? 50257 : 50256;
  c->vocab.assign((size_t)n_text, std::string());
  set_special_tokens(c.get());
  upload_front_end(c.get(), slaney_filters(hp->n_mels));   // overwritten by the import
  hipStream_t s = nullptr;
  if (dtype == OHW_DTYPE_BF16) { Placer<bf16_t> pl(c.get(), s); pl.allocate(); }
  else { Placer<f16_t> pl(c.get(), s); pl.allocate(); }
  HIP_CHECK(hipDeviceSynchronize());
  return c.release();
}

}  // namespace ohw
