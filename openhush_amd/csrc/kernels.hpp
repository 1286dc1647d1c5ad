// kernels.hpp — launchers of the non-GEMM kernels of the hot path (gfx950).
#pragma once
#include "common.hpp"

namespace ohw {

// ---- weights (weights.hip) ---------------------------------------------------------------------
// dst[i] = value of the procedural generator (openhush_amd/synth.py) for flat index i
void launch_synth_fill(float* dst, int64_t n, uint32_t key, float scale, float offset, int round_f16, hipStream_t s);
void launch_f16_to_f32(const void* src_f16, float* dst, int64_t n, hipStream_t s);
// plain row-major convert: dst T [rows][cols] (dst row stride ld_dst) from src f32 [rows][cols]
template <typename T> void launch_convert_rows(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_dst, hipStream_t s);
// conv weight [d_out][c_in][3] f32 -> T [d_out][3][c_pad]
template <typename T> void launch_repack_conv(const float* src, void* dst, int64_t d_out, int64_t c_in, int64_t c_pad, hipStream_t s);
// decoder linear [N][K] f32 -> MFMA-fragment tiles T [Npad/16][K/32][64][8], rows >= N zero
template <typename T> void launch_repack_tiled(const float* src, void* dst, int64_t N, int64_t n_pad, int64_t K, hipStream_t s);
// wsum[n] = sum_k of the rounded 16-bit values of a fragment-tiled [N][K] matrix
template <typename T> void launch_tiled_rowsum(const void* w_tiled, float* wsum, int64_t N, int64_t K, hipStream_t s);
// fold a pre-LayerNorm's affine part into the linear layer that follows it (in place, f32 [N][K]):
//   bias[n] += sum_k beta[k] * W[n][k];  W[n][k] *= gamma[k]      so that  LN(x) W^T + b == ((x - mean) rstd) W'^T + b'
void launch_fold_ln(float* w, float* bias, const float* gamma, const float* beta, int64_t N, int64_t K, hipStream_t s);

// ---- front end (mel.hip) -----------------------------------------------------------------------
struct MelParams {
  const float* pcm;         // device [batch][pcm_stride]
  int64_t pcm_stride;
  const int32_t* n_samples; // device [batch]
  const float* filters;     // device [n_mels][201]
  const float* twiddle;     // device [2][400]: cos, sin of 2*pi*i/400
  const float* window;      // device [400] periodic Hann
  float* logmel;            // device [batch][n_mels][3000] (log10 power, then normalised in place)
  int32_t* max_bits;        // device [batch] ordered-int running max
  void* mel_t;              // device T [batch][3002][128] time-major image for conv1
  int32_t n_mels, batch, mode;
  // windows of a whole recording's spectrogram (ohw_recording_set / ohw_mel_seek): pcm is the recording (n_total samples),
  // window b starts at sample offsets[b] of it - real neighbouring samples at its edges, the reflection only at the
  // recording's start, zeros after its end; shared_max: ONE running maximum, max_bits[0], for all windows;
  // max_only: no spectrogram is stored (the pass that finds the recording's maximum)
  const int64_t* offsets = nullptr;
  int64_t n_total = 0;
  int32_t shared_max = 0, max_only = 0;
};
template <typename T> void launch_mel(const MelParams& p, hipStream_t s);

// ---- normalisation (misc.hip) --------------------------------------------------------------------
// tiled = true writes y in the decoder GEMMs' activation order (act_tiled_offset, common.hpp) instead of row-major
template <typename T> void launch_layernorm(const float* x, const float* gamma, const float* beta, void* y, int64_t rows, int d, hipStream_t s,
                                            bool tiled = false);
template <typename T> void launch_to_f32(const void* src, float* dst, int64_t n, hipStream_t s);

// ---- decoder step (decode.hip) ---------------------------------------------------------------------
enum DecEpilogue {
  DEPI_QKV = 0,        // n < d: q T [M][d]; d <= n < 2d: self-K cache; 2d <= n: self-V cache (+bias)
  DEPI_BIAS_T = 1,     // out T [M][N]
  DEPI_BIAS_GELU_T = 2, // out T in activation-tile order (it feeds mlp.2)
  DEPI_BIAS_RESID = 3, // x f32 [M][N] += v + bias
  DEPI_LOGITS = 4      // logits f32 [batch][ld_logits], only rows m with (m % n_new) == n_new - 1
};
struct DecGemmParams {
  const void* x;       // T activation tiles [ceil(M/16)][K/32][64][8] (act_tiled_offset); with ln != 0: f32 [M][K] residual stream, (x - mean) * rstd fused into the prologue
  int32_t ln;          //   (gamma / beta of that LayerNorm are folded into w / bias at load: launch_fold_ln)
  const void* w;       // tiled T [Npad/16][K/32][64][8]
  const float* bias;   // [N] or nullptr
  void* out;           // see DecEpilogue
  int32_t M, N, K, n_new;
  // DEPI_QKV
  void* k_cache; void* v_cache;     // T [B][H][n_text_ctx][64] of this layer
  const int32_t* n_past;            // device [B]
  int32_t d_model, n_head, n_ctx;
  int64_t ld_out;
  int32_t cu_budget;                // compute units the launch may use (0 = the whole device); picks n-tiles per workgroup
  // DEPI_BIAS_RESID with ksplit > 1: K is split over ksplit workgroups per output tile (see decode.hip)
  int32_t ksplit;                   // 0 / 1 = off
  int32_t slab_bytes;
  float* slab;                      // f32 [tiles][ksplit][512]
  unsigned* ticket;                 // [tiles], zero between launches (the kernel re-arms it)
  // post-norm path (ln == 0, pn != 0): x is the 16-bit tiled copy of the residual stream and the LayerNorm is applied AFTER
  // the product: out = rstd[m] * (x W^T - mean[m] * wsum[n]) + bias[n], mean / rstd from the per-16-column statistics the
  // producers of the residual stream publish (no full-row read, no LayerNorm prologue, no LDS image)
  int32_t pn;
  int32_t n_stat;                   // statistics tiles per row (d_model / 16)
  const float* stat_in;             // f32 [M][n_stat][2]: mean and sum of squared deviations of 16 columns
  const float* wsum;                // f32 [N]
  // DEPI_BIAS_RESID producers: besides x (f32) also its 16-bit tiled copy and this tile's statistics (all null: off)
  void* x16_out;                    // T tiles [ceil(M/16)][N/32][64][8]
  float* stat_out;                  // f32 [M][N/16][2]
  // DEPI_QKV, single-token steps of at most 16 rows (attn_ticket != null): the masked self-attention of a head runs INSIDE this
  // launch, in the workgroup that publishes the last of the head's q / k / v columns (decode.hip) - no self-attention launch
  unsigned* attn_ticket;            // [n_head], zero between launches (the kernel re-arms it)
  void* attn_out;                   // T activation tiles [1][d_model/32][64][8]: what launch_self_attn would have written
  const int32_t* attn_slots;        // beam search: i32 [rows][n_ctx] (launch_self_attn's kv_slot), else null
  int64_t kv_bytes;                 // bytes of one layer's K (or V) cache: bound of the kernel's buffer descriptors
};
template <typename T> void launch_dec_gemm(const DecGemmParams& p, int epilogue, hipStream_t s);

// x f32 [M][d] = token_embedding[tok[m]] + pos_emb[n_past[m / n_new] + m % n_new]
// x16 / stat (may be null): the 16-bit tiled copy and the per-16-column statistics of the post-norm path
template <typename T> void launch_embed(const void* emb_tiled, const float* pos, const int32_t* tok, const int32_t* n_past,
                                        float* x, void* x16, float* stat, int M, int n_new, int d, hipStream_t s);
// causal self-attention of the new tokens against the cache.  q T [M][d] -> out T, activation-tile order
// kv_slot (beam search, else null): i32 [rows][n_ctx], the cache row that holds position j of a row's sequence
template <typename T> void launch_self_attn(const void* q, const void* k_cache, const void* v_cache, const int32_t* n_past,
                                            void* out, int M, int n_new, int n_head, int n_ctx, hipStream_t s, const int32_t* kv_slot = nullptr);
// cross-attention: q T [M][d]; cross K/V head-major T [B][H][t_len][64] of this layer -> out T, activation-tile order
// partials / tickets (may be null): scratch for cutting the keys of a (row, head) over up to XA_MAX_SPLIT workgroups when
// M <= max_split_rows leaves most CUs idle: f32 [max_split_rows][n_head][XA_MAX_SPLIT][68], u32 [max_split_rows][n_head] (zero)
constexpr int XA_MAX_SPLIT = 8;
template <typename T> void launch_cross_attn(const void* q, const void* xk, const void* xv, void* out, int M, int n_new,
                                             int n_head, int t_len, float* partials, unsigned* tickets, int max_split_rows,
                                             const int32_t* done /* [B] or null: windows whose rows are skipped */, hipStream_t s,
                                             int kv_group = 1 /* beam search: consecutive rows that share one window's K/V */,
                                             bool batch_invariant = false /* pick the variant from n_new alone, never from M */);

// ---- the 32 decoder layers of a single-token step in ONE persistent launch, at most 16 rows (decode_persist.hip) ----------
struct PersistLayer {
  const void *wqkv, *wo, *wxq, *wxo, *w1, *w2;       // fragment tiles T [N/16][K/32][64][8]
  const float *bqkv, *bo, *bxq, *bxo, *b1, *b2;
};
struct PersistParams {
  const PersistLayer* layers;     // device [L]
  int32_t L, M, group, d, H, n_ctx, t_len, S, nsplit;   // rows, rows per window, d_model, heads, positions, encoder positions, cross key slices, mlp.2 K slices
  const int32_t* n_past;     // [M]
  const int32_t* kv_slot;    // [M][n_ctx] or null (beam search)
  const int32_t* done;       // [M / group] or null: windows whose cross-attention is skipped
  const float* x_in;         // f32 [M][d]: the embedding launch's output
  float* x_out;              // f32 [M][d]: what the final LayerNorm launch reads (may alias x_in)
  void* self_kv;             // T [L][2][rows][H][n_ctx][64]
  int64_t kv_layer, kv_row;  // elements per K (or V) cache of a layer; per cache row
  const void* xkv;           // T [2L][windows][H][t_len][64]
  int64_t xkv_slab;
  unsigned long long* g;     // ONE granule arena (8-byte {payload | tag}); the regions below are offsets into it
  int32_t o_x, o_q, o_kv, o_a, o_qx, o_h, o_xp, o_mp;
  unsigned* epoch;           // device word, 1 at first use; bumped by every completed launch
  unsigned* abort_word;      // device word, 0; phase + 1 when a workgroup gave up waiting
};
// granules a state's arena needs for up to 16 rows; fills the offsets of p
int64_t persist_layout(PersistParams* p);
template <typename T> void launch_persist_step(const PersistParams& p, int grid, hipStream_t s);

// device-side logits filter + arg-max (restates oracle ref_process_logits)
struct SamplerParams {
  const float* logits;   // [batch][ld]
  int64_t ld;
  int32_t* tokens;       // [batch][max_tokens] sampled so far
  int32_t* n_cur;        // [batch]
  int32_t* n_past;       // [batch]   (advanced by one for live windows)
  int32_t* next_tok;     // [batch]   token to feed next
  int32_t* done;         // [batch]
  int32_t* n_done;       // [1] number of finished windows
  float* sum_logprob;    // [batch]
  int32_t batch, max_tokens, n_vocab;
  int32_t eot, sot, translate, transcribe, solm, prev, nosp, no_ts, ts_begin, blank, n_langs;
  int32_t suppress_blank, no_timestamps, max_initial_ts, n_max, force_len, n_text_ctx;
  int32_t advance;       // 1: n_past[b] += 1 first (a single-token step ran since the last call)
  // the row is scanned by SAMPLER_SPLIT workgroups per window; the one that draws the last ticket merges the partials
  float* partials;       // [batch][SAMPLER_SPLIT][SAMPLER_PART_WORDS]
  unsigned* tickets;     // [batch], zero between launches (re-armed by the kernel)
  const float* bias;     // [n_vocab] added to every logit before the filter, or null (ohw_state_set_logit_bias)
  float* tok_lp;         // [batch][max_tokens + 1] log-probability of every sampled token; the end-of-text token's goes
                         //   to slot n_cur without being counted (whisper.cpp sums it into avg_logprobs), or null
  float* nosp_prob;      // [batch] softmax probability of the no-speech token in the window's first, unfiltered row, or null
};
// beam search state of a batch of windows (decode.hip): beam j of window w is decoder row w * K + j
struct BeamParams {
  int32_t K;              // beam size, 2..5
  float* cand_lp;         // [rows][K + 1] log-probabilities of every row's best next tokens
  int32_t* cand_tok;      // [rows][K + 1]
  float* beam_sum;        // [rows] cumulative log-probability
  int32_t* kv_slot;       // [rows][n_text_ctx] self-K/V row per position (this step's view)
  int32_t* kv_slot_next;  // the other half of the double buffer (written by the update, read by the next step)
  int32_t* tokens_next;   // [rows][max_tokens] the other half of the token-history double buffer
  int32_t* n_cur;         // [windows] tokens sampled so far
  int32_t* n_past_w;      // [windows] position written by the step that just ran
  int32_t* win_done;      // [windows]
  int32_t* fin_cnt;       // [windows] finished sequences in the pool (at most K)
  int32_t* fin_tok;       // [windows][K][max_tokens]
  int32_t* fin_len;       // [windows][K]
  float* fin_sum;         // [windows][K] cumulative log-probability, the end-of-text token's included
  unsigned* part;         // [rows][BEAM_SPLIT][BEAM_PART_WORDS] slice states of the split top-k (beam_topk_kernel)
  unsigned* tickets;      // [rows], zero between launches (re-armed by the kernel)
};
constexpr int BEAM_SPLIT = 8;         // workgroups per logits row
constexpr int BEAM_PART_WORDS = 32;   // max, sum, timestamp sum, best text logit, then (K + 1) text and (K + 1) timestamp candidates (value, index)
void launch_beam_step(const SamplerParams& p, const BeamParams& bp, int n_windows, int first, hipStream_t s);
constexpr int SAMPLER_SPLIT = 8;
constexpr int SAMPLER_PART_WORDS = 12;
void launch_sampler(const SamplerParams& p, hipStream_t s);

}  // namespace ohw
