// model.hpp — weights resident in HBM (ohw_ctx) and per-state activation buffers (ohw_state).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

namespace ohw {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;   // false: a view into an arena (ohw_ctx::arena), never freed here
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; owned = o.owned; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  void view(void* ptr, size_t n) { release(); p = ptr; bytes = n; owned = false; }
  ~DevBuf() { release(); }
  void alloc(size_t n, bool zero = false) {
    release();
    if (n == 0) n = 16;
    HIP_CHECK(hipMalloc(&p, n));
    bytes = n;
    if (zero) HIP_CHECK(hipMemset(p, 0, n));
  }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr; bytes = 0; owned = true;
  }
  template <typename U> U* as() const { return (U*)p; }
};

struct LayerNormW { DevBuf g, b; };

struct EncLayerW {
  LayerNormW ln1, ln2;
  DevBuf wqkv, bqkv;  // T [3d][d], f32 [3d] (key part zero)
  DevBuf wo, bo, w1, b1, w2, b2;
};

struct DecLayerW {
  LayerNormW ln1, lnx, ln2;
  DevBuf wqkv, bqkv;  // tiled T [3d/16][d/32][64][8]
  DevBuf wo, bo;      // tiled
  DevBuf wxq, bxq, wxo, bxo;
  DevBuf w1, b1, w2, b2;
  DevBuf sqkv, sxq, s1;   // f32 row sums of the folded, rounded wqkv / wxq / w1 (post-norm GEMMs)
};

}  // namespace ohw

struct ohw_ctx {
  ohw_hparams hp{};
  ohw_special_tokens tok{};
  int dtype = OHW_DTYPE_BF16;
  int device = 0;
  std::vector<std::string> vocab;  // text tokens as stored in the model file
  // front end
  ohw::DevBuf mel_filters, twiddle, window;
  // encoder
  ohw::DevBuf conv1_w, conv1_b, conv2_w, conv2_b, enc_pos;
  std::vector<ohw::EncLayerW> enc;
  ohw::LayerNormW ln_post;
  ohw::DevBuf xkv_w, xkv_b;  // cross K/V projections of all decoder layers: T [2L*d][d_audio], f32 [2L*d]
  // decoder
  ohw::DevBuf dec_pos, emb;  // f32 [n_text_ctx][d]; tiled T [Vpad/16][d/32][64][8]
  std::vector<ohw::DecLayerW> dec;
  ohw::LayerNormW dec_ln;
  int64_t v_pad = 0;
  size_t weight_bytes = 0;
  // every weight buffer above is a view into this one allocation: one contiguous, large-fragment mapping
  // (the decoder touches all 1.8 GB once per step; per-tensor allocations cost a TLB miss train per kernel)
  ohw::DevBuf arena;
};
