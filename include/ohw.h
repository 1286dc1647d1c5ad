/*
 * ohw.h — C ABI of libohw.so, the MI355X-native Whisper hot path for OpenHush.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own for this path: it calls
 * whisper.cpp through the `whisper-rs` crate (reference src/engine/whisper.rs:10-12).  Each entry
 * point below names the reference call site it replaces; INTEGRATION.md shows the Rust `extern "C"`
 * block and the `WhisperEngine` impl a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C, opaque handles, caller-allocated outputs, no exceptions/aborts across the boundary;
 *   - every function returns OHW_OK (0) or a negative code; ohw_last_error() gives the text of the
 *     last failure on the calling thread;
 *   - handles carry no thread affinity (the reference builds the engine on a tokio thread and
 *     moves it to the `transcription-worker` thread: reference src/queue/worker.rs:22,100-103);
 *     every call selects its device itself.  Calls on ONE state must be serial (the reference
 *     serialises with RefCell::borrow_mut, src/engine/whisper.rs:240);
 *   - the library fails loudly: no CPU fallback exists.  Without a usable gfx950 device
 *     ohw_ctx_create* returns OHW_E_NO_GPU (reference error taxonomy OH-3004).
 */
#ifndef OHW_H
#define OHW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OHW_ABI_VERSION 2

/* error codes: the negated OH-30xx taxonomy the reference documents
 * (reference .claude/knowledge/error-codes.md:79-104; WhisperError variants src/engine/whisper.rs:14-27) */
enum {
  OHW_OK = 0,
  OHW_E_MODEL_NOT_FOUND = -3001, /* WhisperError::ModelNotFound   */
  OHW_E_LOAD_FAILED = -3002,     /* WhisperError::LoadFailed      */
  OHW_E_TRANSCRIBE = -3003,      /* WhisperError::TranscriptionFailed */
  OHW_E_NO_GPU = -3004,
  OHW_E_OOM = -3005,
  OHW_E_INVALID_ARG = -3006,
  OHW_E_VALIDATION = -3007       /* WhisperError::ValidationFailed (code in ohw_audio_info.error) */
};

/* compute dtype of weights/activations fed to MFMA (accumulation, LN, softmax, logits: fp32) */
/* OHW_DTYPE_AUTO (ohw_ctx_create / ohw_engine_new / ohw_pool_create only): f16 when the file stores f16 weights (ftype 1, the
 * stock ggml-*.bin files: the weights stay exact), bf16 for f32 files.  BASELINE.json's bench dtype is bf16, chosen explicitly. */
enum { OHW_DTYPE_AUTO = -1, OHW_DTYPE_BF16 = 0, OHW_DTYPE_F16 = 1 };

/* log-mel tail convention */
enum { OHW_MEL_REFLECT = 0 /* feature-extractor convention, pinned by goldens */,
       OHW_MEL_ZERO_TAIL = 1 /* whisper.cpp convention: zeros after the audio */ };

typedef struct ohw_ctx ohw_ctx;     /* model: weights resident in HBM  (whisper-rs WhisperContext) */
typedef struct ohw_state ohw_state; /* activations, KV caches, streams  (whisper-rs WhisperState)   */

typedef struct {
  int32_t n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype;
} ohw_hparams;

typedef struct {
  int32_t eot, sot, translate, transcribe, solm, prev, nosp, no_timestamps, timestamp_begin, blank;
  int32_t n_langs;
} ohw_special_tokens;

/* ---- audio validation: reference src/engine/validation.rs:46-118 (validate_audio) ------------- */
enum { OHW_AUDIO_OK = 0, OHW_AUDIO_EMPTY = 1, OHW_AUDIO_BAD_RATE = 2, OHW_AUDIO_TOO_LONG = 3,
       OHW_AUDIO_TOO_SHORT = 4, OHW_AUDIO_NAN = 5, OHW_AUDIO_INF = 6 };
typedef struct {
  int32_t error;          /* OHW_AUDIO_* */
  float duration_secs;
  int64_t sample_count;
  float min_value, max_value, rms;
  int64_t nan_count, inf_count;
} ohw_audio_info;
/* host-side scan, same order of checks and same arithmetic as the reference */
int ohw_validate_audio(const float* samples, int64_t n, uint32_t sample_rate, ohw_audio_info* info);

/* ---- model: replaces WhisperContext::new_with_params (reference src/engine/whisper.rs:156-160) - */
/* model_path: a ggml `ggml-*.bin` file (reference src/engine/whisper.rs:71-79).  A missing file   */
/* returns OHW_E_MODEL_NOT_FOUND before any device work (reference :141-154, test :984-997).       */
int ohw_ctx_create(const char* model_path, int device, int dtype, ohw_ctx** out);
/* procedural weights generated on the device (tests / bench; openhush_amd/synth.py is the spec)  */
int ohw_ctx_create_synthetic(const ohw_hparams* hp, uint32_t seed, int device, int dtype, ohw_ctx** out);
int ohw_ctx_info(const ohw_ctx* ctx, ohw_hparams* hp, ohw_special_tokens* tok);
int ohw_ctx_dtype(const ohw_ctx* ctx);   /* the 16-bit type the context computes in (what OHW_DTYPE_AUTO resolved to) */
/* ---- a loaded model as ONE device blob: the rank that read the file exports it, the caller broadcasts it (RCCL) and
 *      the other ranks import it into a shell context made from the same hparams and dtype - instead of every rank
 *      reading and repacking the file ("one ncclBroadcast of the packed weight blob at load").  A shell has
 *      no vocabulary strings: the exporting rank detokenises. ------------------------------------------------------ */
int ohw_ctx_create_shell(const ohw_hparams* hp, int device, int dtype, ohw_ctx** out);
size_t ohw_ctx_blob_size(const ohw_ctx* ctx);
int ohw_ctx_blob_export(const ohw_ctx* ctx, void* dst_device, size_t capacity);
int ohw_ctx_blob_import(ohw_ctx* ctx, const void* src_device, size_t bytes);
/* bytes of token `id` (text tokens only); returns length, 0 for specials                         */
int ohw_token_text(const ohw_ctx* ctx, int32_t id, const char** text);
void ohw_ctx_free(ohw_ctx* ctx); /* WhisperContext drop */

/* ---- audio preprocessing, the step right before the path (host code: the envelope follower and the
 *      limiter are sequential recurrences over the whole recording).  In place, fp32, the reference's operation order:
 *      AudioBuffer::{rms_db, apply_gain, normalize_rms, compress, limit} (reference src/input/audio.rs:86-239),
 *      resample_linear (:972-990), TranscriptionWorker::preprocess_audio (src/queue/worker.rs:196-240).  The rubato sinc
 *      resampler is restated below (ohw_dsp_resample_sinc, ohw_resampler_*).  RNNoise: everything AudioBuffer::denoise
 *      does around the network is built (ohw_dsp_denoise); the network itself (nnnoiseless 0.5.2, a third-party crate with
 *      trained weights that is not in the reference tree) plugs in through ohw_denoise_engine. ------------------------- */
typedef struct ohw_preprocess_config {
  int32_t preprocessing;              /* master switch, reference default 0 (src/config.rs:958,984) */
  int32_t normalization_enabled; float normalization_target_db;                       /* 1, -18 dB */
  int32_t compression_enabled; float compression_threshold_db, compression_ratio,     /* 1, -24 dB, 4:1, */
      compression_attack_ms, compression_release_ms, compression_makeup_gain_db;      /* 5 ms, 50 ms, +6 dB */
  int32_t limiter_enabled; float limiter_ceiling_db, limiter_release_ms;              /* 1, -1 dB, 50 ms */
} ohw_preprocess_config;
void ohw_default_preprocess_config(ohw_preprocess_config* c);
int ohw_preprocess_audio(float* samples, int64_t n, uint32_t sample_rate, const ohw_preprocess_config* c);
/* The noise-reduction stage of the chain (BASELINE config #5 names it; reference src/input/audio.rs:249-341, called first and
 * independently of the `preprocessing` switch: src/queue/worker.rs:197-207).  ohw_denoise_engine is what a host plugs its
 * nnnoiseless::DenoiseState into: process_frame takes ONE 480-sample frame at 48 kHz scaled to the 16-bit range (x 32767,
 * as DenoiseState::process_frame does), writes 480 samples and returns the frame's voice probability; reset (may be NULL)
 * stands for DenoiseState::new() and is called once per ohw_dsp_denoise.  ohw_dsp_denoise does the rest in the reference's
 * order: linear resampling to 48 kHz (resample_for_rnnoise :996-1001), zero-padded last frame, first frame faded in, only
 * the real part of a short last frame kept, linear resampling back, truncate / zero-extend to n, mix by `strength` in 0..1. */
typedef struct ohw_denoise_engine {
  void* user;
  float (*process_frame)(void* user, float* out480, const float* in480);
  void (*reset)(void* user);
} ohw_denoise_engine;
int ohw_dsp_denoise(float* samples, int64_t n, uint32_t sample_rate, float strength, const ohw_denoise_engine* engine);
void ohw_denoise_passthrough_engine(ohw_denoise_engine* e);   /* out = in: the chain's framing without a network (tests) */
/* preprocess_audio with the reference's noise_reduction settings (src/config.rs NoiseReductionConfig {enabled, strength}):
 * enabled != 0 needs an engine (OHW_E_INVALID_ARG without one - never a silent skip) */
int ohw_preprocess_audio_ex(float* samples, int64_t n, uint32_t sample_rate, const ohw_preprocess_config* c,
                            int noise_reduction_enabled, float noise_reduction_strength, const ohw_denoise_engine* denoise);
float ohw_dsp_rms_db(const float* samples, int64_t n);                 /* -inf for silence / empty */
void ohw_dsp_apply_gain(float* samples, int64_t n, float gain_db);
void ohw_dsp_normalize_rms(float* samples, int64_t n, float target_db);
void ohw_dsp_compress(float* samples, int64_t n, uint32_t sample_rate, float threshold_db, float ratio, float attack_ms,
                      float release_ms, float makeup_gain_db);
int64_t ohw_dsp_limit(float* samples, int64_t n, uint32_t sample_rate, float ceiling_db, float release_ms); /* samples over the ceiling */
/* returns the output length; with out == NULL or out_cap too small nothing is written (size query) */
int64_t ohw_dsp_resample_linear(const float* in, int64_t n, uint32_t from_rate, uint32_t to_rate, float* out, int64_t out_cap);

/* high-quality resampling: the reference's resample(.., ResamplingQuality::High) = rubato's SincFixedIn with sinc_len 256,
 * f_cutoff 0.95, oversampling 256, linear interpolation between sub-filters, BlackmanHarris2 window, fed in chunks of 1024
 * input samples (reference src/input/audio.rs:1007-1095).  The rubato crate is not in the reference tree: the algorithm is
 * restated from its published design (windowed-sinc polyphase table, two nearest sub-filters blended linearly), parity
 * unpinned.  Same size-query convention as ohw_dsp_resample_linear.                                                  */
int64_t ohw_dsp_resample_sinc(const float* in, int64_t n, uint32_t from_rate, uint32_t to_rate, float* out, int64_t out_cap);

/* the same resampler on the device (resample.hip): every output sample is independent, so a recording that lies in HBM - or
 * goes there for the log-mel anyway - is resampled there (one wave per output sample; 30 s at 48 kHz in a fraction of a
 * millisecond).  Same number of samples as ohw_dsp_resample_sinc and, up to the order of the fp32 sums, the same samples.
 *   ohw_resampler_create     builds the 256 x 256 polyphase table for from_rate -> to_rate on `device`;
 *   ohw_resampler_out_len    output samples for n input samples;
 *   ohw_resampler_run        in / out in host or device memory (in_on_device / out_on_device), on hip_stream (NULL = the
 *                            default stream); returns when host buffers may be reused, asynchronous when both are device.
 * One call at a time per handle (it stages host buffers in the handle's own device memory); handles are independent. */
typedef struct ohw_resampler ohw_resampler;
int ohw_resampler_create(int device, uint32_t from_rate, uint32_t to_rate, ohw_resampler** out);
void ohw_resampler_free(ohw_resampler* r);
int64_t ohw_resampler_out_len(const ohw_resampler* r, int64_t n);
int ohw_resampler_run(ohw_resampler* r, const float* in, int64_t n, int in_on_device, float* out, int64_t out_cap, int out_on_device,
                      void* hip_stream);

/* ---- voice-activity segmentation, the step in front of the path in continuous mode (host code) --------
 *      ohw_vad_state_*: the reference's VadState (src/vad/mod.rs:112-250) - per-chunk VAD results in, speech segments out;
 *      ohw_vad_engine: its VadEngine trait (src/vad/mod.rs:34-55) as a struct of function pointers, so the host plugs in the
 *      detector (the reference's SileroVad needs an ONNX model that is not available offline); ohw_vad_run: the daemon's
 *      continuous-mode loop over a recording (src/daemon.rs:2062-2138); ohw_vad_energy_engine: a built-in short-time-energy
 *      detector for tests and model-less hosts (not Silero). ------------------------------------------------------------ */
typedef struct { int32_t enabled; float threshold; uint32_t min_silence_ms, min_speech_ms, speech_pad_ms; } ohw_vad_config;
void ohw_default_vad_config(ohw_vad_config* c);     /* 0, 0.5, 700, 250, 30 (reference src/vad/mod.rs:76-100) */
typedef struct { int64_t start, end; float avg_probability; } ohw_speech_segment;   /* positions in samples */
typedef struct ohw_vad_state ohw_vad_state;
ohw_vad_state* ohw_vad_state_new(const ohw_vad_config* cfg, uint32_t sample_rate);
void ohw_vad_state_free(ohw_vad_state* s);
/* returns 1 and fills *seg when a speech segment just ended, 0 otherwise, negative on a bad argument */
int ohw_vad_state_update(ohw_vad_state* s, float probability, int is_speech, int64_t chunk_samples, ohw_speech_segment* seg);
int ohw_vad_state_is_speech(const ohw_vad_state* s);
int64_t ohw_vad_state_speech_start(const ohw_vad_state* s);   /* -1 when not in speech */
void ohw_vad_state_reset(ohw_vad_state* s);
typedef struct {
  void* user;
  int (*process)(void* user, const float* samples, int64_t n, float* probability);   /* 0 = ok; 16 kHz mono f32 */
  void (*reset)(void* user);                                                           /* may be NULL */
  int32_t chunk_size;     /* 512 for Silero */
  uint32_t sample_rate;   /* 16000 */
} ohw_vad_engine;
int ohw_vad_energy_engine(float threshold_db, ohw_vad_engine* out);
void ohw_vad_energy_engine_free(ohw_vad_engine* e);
/* number of speech segments of a recording (all of them; at most cap are written to out), or a negative error code */
int64_t ohw_vad_run(const ohw_vad_engine* engine, const ohw_vad_config* cfg, const float* samples, int64_t n, int64_t poll_samples,
                    ohw_speech_segment* out, int64_t cap);

/* ---- streaming glue right after the path, host code -----------------------------------------------
 *      ohw_tracker_*: TranscriptionTracker (reference src/queue/mod.rs:59-297) - chunks are registered as pending under
 *      back-pressure, results come back in any order, take_ready releases them (streaming mode: all completed chunks sorted
 *      by (sequence, chunk), the words that repeat the end of the previous output removed; ordered mode: recordings in
 *      sequence order).  ohw_extract_chunk: AudioRecorder::extract_chunk for a 16 kHz recorder (src/input/audio.rs:737-785:
 *      nothing below 0.1 s, zero padding to 1.1 s).  ohw_chunk_scheduler_*: the chunk-timer arm of the daemon loop
 *      (src/daemon.rs:1958-2011).  Strings are UTF-8; a text returned by ohw_tracker_ready_get lives until the next
 *      take_ready on that tracker.  A tracker or scheduler is used by one thread at a time (the reference holds its tracker
 *      inside the daemon's loop); they touch no device. */
enum { OHW_BACKPRESSURE_WARN = 0, OHW_BACKPRESSURE_DROP_OLDEST = 1, OHW_BACKPRESSURE_DROP_NEWEST = 2 };
typedef struct ohw_tracker ohw_tracker;
ohw_tracker* ohw_tracker_new(int streaming);
void ohw_tracker_free(ohw_tracker* t);
/* 1 = accepted, 0 = refused (DROP_NEWEST at max_pending), negative = error; max_pending 0 = unlimited */
int ohw_tracker_add_pending(ohw_tracker* t, uint64_t sequence_id, uint32_t chunk_id, uint32_t max_pending, uint32_t high_water_mark, int strategy);
int ohw_tracker_add_result(ohw_tracker* t, const char* text, uint64_t sequence_id, uint32_t chunk_id, int is_final, float duration_secs);
int ohw_tracker_take_ready(ohw_tracker* t);   /* number of released results, read with ohw_tracker_ready_get(t, 0 .. n-1, ..) */
int ohw_tracker_ready_get(const ohw_tracker* t, int i, const char** text, uint64_t* sequence_id, uint32_t* chunk_id, int* is_final,
                          float* duration_secs);
void ohw_tracker_reset_dedup(ohw_tracker* t);
int ohw_tracker_is_empty(const ohw_tracker* t);
int ohw_tracker_is_pending(const ohw_tracker* t, uint64_t sequence_id, uint32_t chunk_id);
int ohw_tracker_pending_count(const ohw_tracker* t);
int ohw_tracker_waiting_count(const ohw_tracker* t);
/* length of the chunk [from_pos, to_pos) after padding (0: shorter than 0.1 s); written to out when out_cap suffices */
int64_t ohw_extract_chunk(const float* recording, int64_t n_recording, int64_t from_pos, int64_t to_pos, float* out, int64_t out_cap);
typedef struct ohw_chunk_scheduler ohw_chunk_scheduler;
ohw_chunk_scheduler* ohw_chunk_scheduler_new(ohw_tracker* tracker, uint64_t sequence_id, uint32_t max_pending, uint32_t high_water_mark,
                                             int strategy);
void ohw_chunk_scheduler_free(ohw_chunk_scheduler* s);
/* one timer tick at recorder position current_pos: the job's length (its samples: ohw_extract_chunk(*from_pos, current_pos)) and
 * id, or 0 - too short (nothing moved) or refused by the tracker (position and id moved on, as in the reference) */
int64_t ohw_chunk_scheduler_tick(ohw_chunk_scheduler* s, const float* recording, int64_t n_recording, int64_t current_pos,
                                 uint32_t* chunk_id, int64_t* from_pos);
int64_t ohw_chunk_scheduler_position(const ohw_chunk_scheduler* s);
uint32_t ohw_chunk_scheduler_next_id(const ohw_chunk_scheduler* s);
int64_t ohw_chunk_scheduler_rejected(const ohw_chunk_scheduler* s);

/* ---- state: replaces ctx.create_state() (reference src/engine/whisper.rs:167-169) ------------- */
/* max_batch = number of independent 30 s windows processed together (the reference: 1)           */
int ohw_state_create(ohw_ctx* ctx, int max_batch, ohw_state** out);
void ohw_state_free(ohw_state* st); /* WhisperState drop: frees all device memory */
/* HIP stream (hipStream_t) every later call on this state enqueues on; NULL = the state's own    */
int ohw_state_set_stream(ohw_state* st, void* hip_stream);
int ohw_state_max_batch(const ohw_state* st);
const ohw_ctx* ohw_state_ctx(const ohw_state* st);

/* ---- streams for two batches in flight (the reference has one state and no overlap: src/queue/worker.rs:100-160
 *      transcribes one buffer at a time).  The encoder is MFMA-bound and the decoder HBM/latency-bound, so the engine
 *      runs the encoder of batch i+1 beside the decoder of batch i on DISJOINT sets of compute units: a stream made here
 *      is restricted to CU-mask bits [first_cu, first_cu + n_cu) (bits are dealt round-robin over the 8 XCDs, so any
 *      contiguous range is spread evenly; a mask that would leave an XCD without any CU is not honoured by the
 *      runtime - tools/probes/xcd_mask_probe.hip - so whole-XCD partitions cannot be made); n_cu = 0 makes an unrestricted stream.  ohw_stream_wait makes `waiter` wait
 *      for everything enqueued on `signal` so far (event record + wait, no host sync). ------------------------------- */
int ohw_stream_create(int device, int first_cu, int n_cu, void** stream_out);
int ohw_stream_destroy(void* stream);
int ohw_stream_wait(void* waiter, void* signal);
int ohw_stream_sync(void* stream);

/* ---- the stages of state.full() (reference src/engine/whisper.rs:266-268), split so that the   */
/*      host keeps windowing and sampling (BASELINE.json north_star) ------------------------------ */

/* log-mel of `batch` windows.  pcm: batch rows of `pcm_stride` floats (host memory, or device     */
/* memory when pcm_on_device != 0); n_samples[b] <= 480000 valid samples per row (rest = silence). */
/* mel_out (optional, host or NULL): [batch][n_mels][3000] f32 copy of the normalised log-mel.     */
int ohw_mel(ohw_state* st, const float* pcm, int64_t pcm_stride, const int32_t* n_samples, int batch,
            int pcm_on_device, int mel_mode, float* mel_out);

/* The spectrogram of a WHOLE recording, windows cut from it afterwards - what whisper.cpp does inside state.full() for
 * audio of any length (whisper_pcm_to_mel on all samples, then the encoder reads 3000 frames at the seek offset; reference
 * src/engine/whisper.rs:266-268 hands the whole buffer over).  Against ohw_mel on the samples from the seek offset on, a
 * window gets (a) the clamp `max - 8` from the maximum over ALL frames of the recording, (b) real neighbouring samples at
 * its edges (the 200-sample reflection only at the recording's start, zeros only after its end).
 *   ohw_recording_set: copies the recording (host, or device when pcm_on_device != 0; 1 .. 2 h of 16 kHz samples) into
 *     the state and finds that maximum in one pass (log_max_out, optional: log10 of the largest mel power);
 *   ohw_mel_seek: windows [seek_frames[b], +3000) (10 ms frames) of that spectrogram, for ohw_encode(batch) as after
 *     ohw_mel; mel_out as there.  The engine's OHW_WINDOW_SEEK mode runs on these two. */
int ohw_recording_set(ohw_state* st, const float* pcm, int64_t n, int pcm_on_device, float* log_max_out);
int ohw_mel_seek(ohw_state* st, const int32_t* seek_frames, int batch, float* mel_out);
/* encoder + cross-attention K/V of every decoder layer, for the windows of the last ohw_mel      */
int ohw_encode(ohw_state* st, int batch);
/* the same into windows [first, first + batch) of a decode batch of `total` windows (total <= max_batch): several front-end
 * passes can feed ONE decode - the decoder streams its weights once per step whatever its batch, so four 32-window front ends
 * decoded as one 128-row batch move 13 % fewer bytes per token than four 32-row decodes.  ohw_greedy / ohw_decode /
 * ohw_beam_search then take batch = total.                                                                          */
int ohw_encode_slice(ohw_state* st, int batch, int first, int total);
/* feed tokens[b][0..n_new) at positions n_past[b].. and return logits of the last fed position    */
/* per window: logits_out [batch][n_vocab] f32 (host).  tokens: [batch][n_new] row-major.          */
int ohw_decode(ohw_state* st, const int32_t* tokens, int n_new, const int32_t* n_past, int batch, float* logits_out);
/* the same for a SUBSET of the batch: windows with active[b] == 0 ride along (their rows of the weight-streaming GEMMs cost
 * nothing) but their cross K/V is not streamed and their logits row is not copied.  What the temperature fallback uses to
 * re-decode only the windows that failed, on their resident cross K/V.  active == NULL: every window.              */
int ohw_decode_active(ohw_state* st, const int32_t* tokens, int n_new, const int32_t* n_past, int batch, const int32_t* active,
                      float* logits_out);

/* language identification for the windows of the last ohw_encode (whisper.cpp whisper_lang_auto_detect: one
 * decoder step on [sot], soft-max over the language tokens only).  lang_ids_out [batch];
 * lang_probs_out [batch][n_langs] or NULL.  The reference never enables auto-detection ("auto" keeps
 * whisper.cpp's default "en": SURVEY.md section 0 item 6); this entry point exists for SURVEY.md row A4.8. */
int ohw_detect_language(ohw_state* st, int batch, int32_t* lang_ids_out, float* lang_probs_out);

/* sampling parameters: the whisper.cpp defaults the reference inherits because it sets none     */
/* (reference src/engine/whisper.rs:243-263)                                 */
typedef struct {
  int32_t lang_id;        /* 0 = "en": what language "auto" means in the reference */
  int32_t translate;      /* task token; the reference passes !config.translate (whisper.rs:251-257) */
  int32_t no_timestamps;  /* 0 */
  int32_t suppress_blank; /* 1 */
  int32_t max_initial_ts; /* 50 = 1.0 s / 0.02 s */
  int32_t n_max;          /* n_text_ctx/2 - 4 = 220 */
  int32_t force_len;      /* > 0: EOT suppressed, decoding stops after exactly this many tokens    */
} ohw_sample_params;
void ohw_default_sample_params(const ohw_ctx* ctx, ohw_sample_params* p);

/* host-side logits filter + arg-max for one window (the "token sampler" the host owns).          */
/* logits: [n_vocab], modified in place; cur: tokens sampled so far in this window.               */
int32_t ohw_sample_greedy_host(const ohw_ctx* ctx, const ohw_sample_params* p, float* logits,
                               const int32_t* cur, int n_cur, float* logprob_out);

/* the host sampler at a temperature: whisper.cpp's whisper_sample_token(best = false) - logits / temperature, the same
 * filter, then one draw of std::discrete_distribution over exp(logprobs) from a std::mt19937 (whisper.cpp seeds each
 * decoder's generator with 0 once per whisper_full call).  temperature <= 0: arg-max (rng may be NULL).
 * no_speech_prob_out (may be NULL) is written on a window's first step (n_cur == 0): soft-max probability of the
 * no-speech token in the scaled, unfiltered row.                                                                    */
typedef struct ohw_rng ohw_rng;
ohw_rng* ohw_rng_new(uint32_t seed);
void ohw_rng_free(ohw_rng* rng);
int32_t ohw_sample_host(const ohw_ctx* ctx, const ohw_sample_params* p, float* logits, const int32_t* cur, int n_cur,
                        float temperature, ohw_rng* rng, float* logprob_out, float* no_speech_prob_out);

/* device-resident greedy loop for the windows of the last ohw_encode: prompt, KV-cached steps,   */
/* logits filter and arg-max all stay on the GPU; only token ids come back.                        */
/* tokens_out [batch][max_tokens] i32, n_tokens_out [batch] (host); EOT is not stored.             */
int ohw_greedy(ohw_state* st, const ohw_sample_params* p, int batch, int32_t* tokens_out, int32_t* n_tokens_out,
               int max_tokens, float* sum_logprob_out /* [batch] or NULL */);

/* the same loop, with everything whisper.cpp's per-window bookkeeping needs: the log-probability of
 * every sampled token - whisper.cpp's avg_logprobs sums the end-of-text token's too: it is slot n_tokens[b] when
 * ended_by_eot[b] - and the no-speech probability of the window (soft-max of the first, unfiltered logits row at the
 * no-speech token).  Every pointer except tokens / n_tokens may be NULL. */
typedef struct {
  int32_t* tokens;        /* [batch][max_tokens] */
  int32_t* n_tokens;      /* [batch] */
  float* sum_logprob;     /* [batch]: sum over the stored tokens (end-of-text excluded) */
  float* token_logprobs;  /* [batch][max_tokens + 1] */
  int32_t* ended_by_eot;  /* [batch]: 1 = end-of-text was sampled, 0 = a length limit stopped the window */
  float* no_speech_prob;  /* [batch] */
} ohw_greedy_result;
int ohw_greedy_ex(ohw_state* st, const ohw_sample_params* p, int batch, int max_tokens, const ohw_greedy_result* out);

/* beam search for the windows of the last ohw_encode (BASELINE.json config #5: beam = 5, hipGraph-captured decoder step).
 * The reference never uses one (Greedy{best_of:1}, src/engine/whisper.rs:243); the rule is the published Whisper
 * BeamSearchDecoder: every live beam proposes its beam_size + 1 most likely next tokens after the same logits filter, a
 * window's candidates are ranked by cumulative log-probability, sequences ending in end-of-text go to its finished pool
 * (at most beam_size), the best beam_size others become the new beams; the result is the candidate with the best
 * cumulative log-probability per token.  Device-resident like ohw_greedy: beam j of window w is decoder row w * K + j, the
 * K rows of a window stream its cross K/V once and share their common past through a slot table (no K/V copies); one
 * iteration {decoder step, top-k, update} is captured as a hipGraph and replayed.  Needs max_batch >= n_windows * beam_size. */
typedef struct {
  int32_t* tokens;      /* [n_windows][max_tokens] the best sequence, end-of-text not stored */
  int32_t* n_tokens;    /* [n_windows] */
  float* sum_logprob;   /* [n_windows] or NULL: its cumulative log-probability (end-of-text's included when it ended so) */
  int32_t* n_finished;  /* [n_windows] or NULL: sequences that reached end-of-text */
} ohw_beam_result;
int ohw_beam_search(ohw_state* st, const ohw_sample_params* p, int n_windows, int beam_size, int max_tokens, const ohw_beam_result* out);

/* additive bias on every logits row before the filter, bias[n_vocab] (host; copied), NULL clears it.  This is the
 * engine's form of whisper.cpp's logits_filter_callback (whisper_full_params; the reference sets none,
 * src/engine/whisper.rs:243-263, so the default is no bias); tests use it to make end-of-text and timestamps win.
 * Set on ohw_engine_state(e) it holds for every state ohw_engine_transcribe decodes on (the schedules' lane states too). */
int ohw_state_set_logit_bias(ohw_state* st, const float* bias, int n);

/* the persistent small-batch decoder step (default OFF; OHW_DEC_PERSIST=1 turns it on for new states): single-token steps
 * of at most 16 rows - one utterance, the K rows of a beam search - run their 32 layers as ONE launch whose workgroups hand
 * activations to each other (openhush_amd/csrc/decode_persist.hip) instead of 8 launches per layer.  Correct (tests/
 * test_gpu_persist.py) and slower than the launches on MI355X: every all-to-all hand-off between the 256 workgroups costs
 * 3 - 6.5 us against 1.6 us of kernel boundary + 1.9 us of first-byte latency (profiles/r03_persist_trace.txt; large-v3,
 * one row: 2.2 ms per token against 1.44).  Never used under ohw_state_set_batch_invariant.  Results are deterministic and
 * a row's result does not depend on the other rows.                                                                        */
int ohw_state_set_persistent(ohw_state* st, int on);

/* batch-invariant decoding (default off): the decoder picks some kernel variants from the number of rows in flight - up to
 * 24 rows the keys of a cross-attention (row, head) are cut over several workgroups, and the prompt pass shares one K/V
 * stream among a window's rows only when there are enough windows - and a different variant sums the softmax in a
 * different order (last-bit differences in the logits).  With this on, the variant depends on n_new alone, so a window's
 * logits and tokens are bit-identical whatever batch it is decoded in; tiny batches lose a little latency.
 * ohw_engine_transcribe turns it on for audio longer than one batch, which makes its schedules agree token for token. */
int ohw_state_set_batch_invariant(ohw_state* st, int on);

/* per-stage device time of the last calls on this state, in milliseconds (reference logs the     */
/* same split per job: src/queue/worker.rs:170-180)                                               */
typedef struct { float mel_ms, encode_ms, decode_ms, total_ms; int32_t decode_steps; } ohw_timings;
int ohw_state_timings(ohw_state* st, ohw_timings* t);

/* per-kernel-class timing with HIP events on the state's stream (bench.py's roofline leg).        */
/* Classes = kernels: 1 encoder/cross-KV MFMA GEMM, 2 encoder attention, 3 decoder cross-attention, */
/* 4..8 the instantiations of the decoder weight-streaming GEMM (residual-accumulate out-projections */
/* and mlp.2; LN+QKV; LN+cross-query; LN+mlp.0+GELU; logits).  work = algorithmic flops (1, 2) or    */
/* bytes (3..8) summed over the launches between begin and end.                                     */
enum { OHW_PROF_NONE = 0, OHW_PROF_ENC_GEMM = 1, OHW_PROF_ENC_ATTN = 2, OHW_PROF_DEC_XATTN = 3, OHW_PROF_DEC_GEMM = 4,
       OHW_PROF_DEC_GEMM_QKV = 5, OHW_PROF_DEC_GEMM_XQ = 6, OHW_PROF_DEC_GEMM_FC1 = 7, OHW_PROF_DEC_GEMM_LOGITS = 8 };
int ohw_state_profile_begin(ohw_state* st, int kernel_class);
int ohw_state_profile_end(ohw_state* st, int64_t* launches, double* total_ms, double* work);

/* ---- WhisperEngine mirror (reference src/engine/whisper.rs:110-387): the host driver written   */
/*      in C++ because no Rust toolchain exists in the build image --------------------------------- */
typedef struct ohw_engine ohw_engine;
/* WhisperEngine::new(model_path, language, translate, use_gpu) — reference :129-179              */
int ohw_engine_new(const char* model_path, const char* language, int translate, int use_gpu, int device,
                   int dtype, int max_batch, ohw_engine** out);
/* WhisperEngine::transcribe(&AudioBuffer) — reference :204-310.  Text is copied into text_buf     */
/* (UTF-8, NUL-terminated, truncated to text_cap).  language_out: >= 8 bytes.                      */
int ohw_engine_transcribe(ohw_engine* e, const float* samples, int64_t n, uint32_t sample_rate,
                          char* text_buf, size_t text_cap, char* language_out, uint64_t* duration_ms,
                          ohw_audio_info* info);
/* full text of the last transcribe (owned by the engine until the next call): a 2 h file can exceed any   */
/* fixed text_buf; text_buf receives a truncated copy, this returns everything.                              */
int ohw_engine_last_text(ohw_engine* e, const char** text, size_t* len);
/* whisper.cpp's per-window decode policy, which the reference inherits because it sets none of these
 * (reference src/engine/whisper.rs:243-263 -> whisper_full_default_params; recalled from
 * upstream, unpinned):
 *   - exits of the decode loop: end-of-text; a timestamp that leaves less than 1 s of audio; failure when the loop
 *     reaches n_max without a timestamp past the middle of the window (the repetition guard) or when end-of-text comes
 *     with no timestamp while audio is left;
 *   - acceptance: not failed, token-frequency entropy of the last 32 tokens >= entropy_thold, and not
 *     (avg_logprob < logprob_thold while no_speech_prob < no_speech_thold); a pass that is not accepted is decoded again
 *     at temperature += temperature_inc (up to 1.0), sampled on the host with std::mt19937(0) + std::discrete_distribution
 *     on the window's resident cross K/V (ohw_decode_active); the last temperature is accepted whatever it gives;
 *   - no speech: a window with no_speech_prob > no_speech_thold and avg_logprob < logprob_thold yields no text.
 * temperature_inc <= 0 keeps every window at T = 0 (what bench.py times: SURVEY.md 8d). */
typedef struct { float temperature_inc, entropy_thold, logprob_thold, no_speech_thold; } ohw_decode_policy;
void ohw_default_decode_policy(ohw_decode_policy* q);   /* 0.2, 2.4, -1.0, 0.6 */
int ohw_engine_set_decode_policy(ohw_engine* e, const ohw_decode_policy* q);
/* per window of the last transcribe, for the pass that was kept */
typedef struct {
  int32_t n_tokens;        /* tokens that reached the text / ohw_engine_last_tokens (0 for a no-speech window) */
  float avg_logprob;       /* over the first result_len tokens (whisper.cpp avg_logprobs; -inf when result_len == 0) */
  float entropy;           /* of the last 32 of them */
  int32_t would_fallback;  /* the T = 0 pass failed the acceptance test */
  float temperature;       /* of the kept pass */
  float no_speech_prob;
  int32_t no_speech;       /* dropped by the no-speech rule */
  int32_t seek_delta;      /* 10 ms frames the window covers: 3000 unless a timestamp ended it (what the seek loop advances by) */
  int32_t result_len;      /* whisper.cpp result_len: tokens up to and including the last timestamp */
  int32_t failed;          /* the kept pass ended in one of the loop's failure exits */
} ohw_window_quality;
int ohw_engine_last_quality(ohw_engine* e, const ohw_window_quality** q, int* n_windows);
/* every decode pass of the last transcribe, flat: {window index, temperature * 1000, n, n sampled tokens
 * (end-of-text included when it was sampled)} repeated - what a parity check walks pass by pass */
int ohw_engine_last_trace(ohw_engine* e, const int32_t** data, int* n);
/* How audio longer than 30 s is windowed.  FIXED (default): host-side cuts every 30 s, windows batched
 * (BASELINE.json north_star).  SEEK: whisper.cpp's sequential loop as recalled (unpinned): the
 * next window starts at the last timestamp token of the previous one (seek += 2 * (ts - ts_begin) frames of
 * 10 ms, or 3000 when no timestamp was produced), tokens after that timestamp are dropped and re-decoded;
 * stops when less than 1 s is left.  One window at a time, so batch = 1.
 * FIXED_RECORDING_MEL: the cuts, batching and schedules of FIXED, but every window is cut from the spectrogram of the WHOLE
 * recording (ohw_recording_set / ohw_mel_seek): one clamp maximum for all windows and real samples across the 30 s marks -
 * what whisper.cpp's front end gives a recording handed over in one call - instead of treating each cut as its own call. */
enum { OHW_WINDOW_FIXED = 0, OHW_WINDOW_SEEK = 1, OHW_WINDOW_FIXED_RECORDING_MEL = 2 };
int ohw_engine_set_window_mode(ohw_engine* e, int mode);
/* measurement knob (bench.py --pool; SURVEY.md 8d "decode length is pinned"): n_tokens > 0 makes every window decode exactly
 * that many tokens with end-of-text suppressed (ohw_sample_params.force_len; use with temperature_inc = 0: a forced sequence
 * fails whisper.cpp's acceptance test by construction); 0 (default) = the reference's behaviour.                         */
int ohw_engine_set_force_len(ohw_engine* e, int n_tokens);
/* How audio of more than max_batch windows is overlapped on the device (the reference transcribes one buffer at a time,
 * src/queue/worker.rs:100-160; results are identical under every schedule):
 *   SEQUENTIAL  one batch after the other;
 *   PIPELINE    front end (mel, encoder, cross K/V) of batch i+1 on OHW_ENGINE_ENC_CUS compute units beside the decode of
 *               batch i on the rest (round 1's schedule);
 *   LANES       (default) groups of up to `lanes` lanes, each of up to `merge` batches of max_batch windows (dealt evenly):
 *               a lane takes its windows through ONE front-end pass and decodes them as ONE batch; the lanes' front ends run
 *               one after the other on every compute unit, then their decodes side by side, each on its own CU-masked
 *               stream and host thread - a decode alternates an HBM-bound kernel with a latency-bound chain, several of
 *               them together keep HBM busy, and a larger decode batch streams the decoder's weights once for all its rows.
 * The schedule's extra states and streams are made when a long input first needs them (a lane's state holds up to
 * merge x max_batch windows: 245.76 MB of cross K/V per window at large-v3); environment defaults:
 * OHW_ENGINE_SCHEDULE = sequential | pipeline | lanes, OHW_ENGINE_LANES (4), OHW_ENGINE_MERGE (3), OHW_ENGINE_ENC_CUS (96). */
enum { OHW_SCHEDULE_SEQUENTIAL = 0, OHW_SCHEDULE_PIPELINE = 1, OHW_SCHEDULE_LANES = 2 };
int ohw_engine_set_schedule(ohw_engine* e, int schedule, int lanes /* 0 = keep */, int merge /* 0 = keep */);
/* tokens of the last transcribe, per 30 s window concatenated (for parity tests)                  */
int ohw_engine_last_tokens(ohw_engine* e, const int32_t** tokens, int* n);
/* WhisperEngine::benchmark(safety_margin) — reference :334-387                                    */
int ohw_engine_benchmark(ohw_engine* e, float safety_margin, float* overhead_secs, float* recommended_chunk_interval,
                         float* test_audio_secs);
void ohw_engine_free(ohw_engine* e);
ohw_state* ohw_engine_state(ohw_engine* e);
ohw_ctx* ohw_engine_ctx(ohw_engine* e);
/* ---- multi-GPU pool behind the boundary.  The reference has no multi-GPU code, only the intent:
 *      requirement F14 "distribute transcriptions across multiple GPUs on a single machine" (reference REQUIREMENTS.md:26)
 *      and the config keys `[gpu] auto_detect / devices` that today do nothing (src/config.rs:921-929).  One engine and
 *      one host thread per listed device.  The model file is read ONCE (device_ids[0]); its resident weight arena reaches
 *      the other devices in one broadcast over xGMI: RCCL (ncclCommInitAll + ncclBroadcast, loaded with dlopen at run
 *      time) or peer-to-peer copies when RCCL cannot be loaded, OHW_POOL_BCAST=peer is set, or a device is listed twice.
 *      ohw_pool_transcribe has ohw_engine_transcribe's contract; the fixed 30 s windows of the recording are dealt
 *      round-robin (window w -> device w mod n) with no collective in the data path, results gathered on the host in
 *      recording order (the seek-loop window mode is sequential and runs on device_ids[0] alone). ----------------- */
typedef struct ohw_pool ohw_pool;
int ohw_pool_create(const char* model_path, const char* language, int translate, const int* device_ids, int n_devices,
                    int dtype, int max_batch, ohw_pool** out);
/* the same pool around procedural weights made on device_ids[0] (ohw_ctx_create_synthetic) instead of a file read: what
 * bench.py --pool and the tests use where no model file exists */
int ohw_pool_create_synthetic(const ohw_hparams* hp, uint32_t seed, const char* language, int translate, const int* device_ids,
                              int n_devices, int dtype, int max_batch, ohw_pool** out);
int ohw_pool_set_force_len(ohw_pool* p, int n_tokens);                     /* ohw_engine_set_force_len on every engine */
int ohw_pool_set_schedule(ohw_pool* p, int schedule, int lanes, int merge); /* ohw_engine_set_schedule on every engine */
int ohw_pool_transcribe(ohw_pool* p, const float* samples, int64_t n, uint32_t sample_rate, char* text_buf, size_t text_cap,
                        char* language_out, uint64_t* duration_ms, ohw_audio_info* info);
int ohw_pool_last_text(ohw_pool* p, const char** text, size_t* len);
int ohw_pool_last_tokens(ohw_pool* p, const int32_t** tokens, int* n);
int ohw_pool_last_quality(ohw_pool* p, const ohw_window_quality** q, int* n_windows);
int ohw_pool_set_decode_policy(ohw_pool* p, const ohw_decode_policy* q);
/* the window mode of EVERY engine of the pool (ohw_engine_set_window_mode): ohw_pool_transcribe rejects a pool whose engines
 * disagree.  All three modes give the single engine's tokens: each device is handed the whole recording and cuts its own
 * windows w, w + n, ... from it (in FIXED_RECORDING_MEL from the recording-wide spectrogram).                              */
int ohw_pool_set_window_mode(ohw_pool* p, int mode);
/* "" or why the RCCL broadcast was given up for peer copies.  After either kind every replica's weight buffers are compared
 * with device_ids[0]'s (64-bit digests); a mismatch fails ohw_pool_create with OHW_E_LOAD_FAILED naming the device.          */
const char* ohw_pool_broadcast_note(const ohw_pool* p);
int ohw_pool_n_devices(const ohw_pool* p);
const char* ohw_pool_broadcast_kind(const ohw_pool* p);   /* "none" (one device), "rccl" or "peer" */
ohw_engine* ohw_pool_engine(ohw_pool* p, int i);           /* the i-th device's engine (borrowed) */
void ohw_pool_free(ohw_pool* p);

/* lang id -> ISO code ("unknown" outside 0..98): reference lang_id_to_code :627-731               */
const char* ohw_lang_id_to_code(int32_t id);
/* ISO code -> lang id, -1 if unknown */
int32_t ohw_lang_code_to_id(const char* code);

/* ---- diagnostics (tests) ----------------------------------------------------------------------- */
const char* ohw_last_error(void);
int ohw_abi_version(void);
/* copy an internal activation to the host as f32: what = "mel" [B][n_mels][3000], "conv1"        */
/* [B][3000][d], "stem" / "block0" / "enc" [B][1500][d], "xk<l>" / "xv<l>" [B][1500][d]            */
int ohw_state_fetch(ohw_state* st, const char* what, int batch, float* out, int64_t out_elems);
/* 64-bit digest of the index-th resident weight buffer (engine layout); returns OHW_E_INVALID_ARG  */
/* past the last buffer.  Lets tests prove "synthetic ctx == ctx loaded from the synthetic file".  */
int ohw_ctx_weight_digest(const ohw_ctx* ctx, int index, char* name_out /* >= 64 bytes */, uint64_t* digest);
/* kernel-level entry points on raw device pointers (tests against a torch fp32 reference)         */
int ohw_dbg_gemm(int dtype, const void* A, const void* W, const float* bias, void* out, int64_t M, int64_t N,
                 int64_t K, int epilogue, void* stream);
int ohw_dbg_attention(int dtype, const void* qkv, void* out, int batch, int T, int n_head, void* stream);
/* the DEVICE sampler on caller-supplied rows: logits [batch][n_vocab] (host), history [batch][hist_stride] with
 * n_hist[b] tokens sampled so far in the window.  tokens_out [batch]: the pick (end-of-text included);
 * logprobs_out [batch] / no_speech_out [batch] may be NULL (no-speech is defined for rows with n_hist == 0). */
int ohw_dbg_sample(ohw_state* st, const ohw_sample_params* p, const float* logits, const int32_t* history, int hist_stride,
                   const int32_t* n_hist, int batch, int32_t* tokens_out, float* logprobs_out, float* no_speech_out);
/* counters of a state's graph caches: "step_captures" / "beam_captures" (graphs / graph pairs captured so far),
 * "step_graphs" / "beam_graphs" (entries held now), "persist_launches" (persistent decoder steps launched or captured);
 * OHW_E_INVALID_ARG for another name.  A second ohw_greedy /
 * ohw_beam_search with the same batch, parameters and stream must add no capture (tests/test_gpu_beam.py).        */
int ohw_dbg_counter(const ohw_state* st, const char* name);

#ifdef __cplusplus
}
#endif
#endif /* OHW_H */
