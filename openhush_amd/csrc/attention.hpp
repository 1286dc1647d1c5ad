// attention.hpp — encoder self-attention (flash-style, non-causal, d_head = 64), gfx950.
#pragma once
#include "common.hpp"

namespace ohw {
// qkv: T [B*T][3*d] (q | k | v, head h at columns h*64), out: T [B*T][d]
template <typename T> void launch_encoder_attention(const void* qkv, void* out, int batch, int t_len, int n_head, hipStream_t stream);
}  // namespace ohw
