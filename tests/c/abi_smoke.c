/* The C ABI from C: include/ohw.h compiled as C99, libohw.so linked the way a C or Rust host links it.
 *   gcc -std=c99 -Wall -Werror -I include tests/c/abi_smoke.c -L openhush_amd -lohw -Wl,-rpath,$PWD/openhush_amd -lm -o abi_smoke
 * Without a GPU: `abi_smoke host` runs the host-only entry points.  With one: `abi_smoke gpu` builds a synthetic micro
 * model, runs three windows through the staged API (the two identical windows of the batch must give identical tokens) and,
 * given a model file (`abi_smoke gpu PATH`), a 70 s recording through the product API ohw_engine_new / _transcribe. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ohw.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != OHW_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, ohw_last_error()); return 1; } } while (0)

static int host_only(void) {
  /* validation, language table, tracker, chunk extraction, VAD state, DSP: no device involved */
  static float pcm[16000];
  ohw_audio_info info;
  for (int i = 0; i < 16000; ++i) pcm[i] = 0.1f * sinf(0.05f * (float)i);
  CHECK(ohw_validate_audio(pcm, 16000, 16000, &info));
  if (info.sample_count != 16000 || fabsf(info.duration_secs - 1.0f) > 1e-6f) return 2;
  if (ohw_validate_audio(pcm, 16000, 44100, &info) == OHW_OK) return 3;
  if (strcmp(ohw_lang_id_to_code(0), "en") != 0 || ohw_lang_code_to_id("de") < 0) return 4;
  ohw_tracker* t = ohw_tracker_new(1);
  if (!t || ohw_tracker_add_pending(t, 1, 0, 10, 8, OHW_BACKPRESSURE_WARN) != 1) return 5;
  CHECK(ohw_tracker_add_result(t, "hello world", 1, 0, 1, 1.0f));
  if (ohw_tracker_take_ready(t) != 1) return 6;
  const char* text = NULL;
  CHECK(ohw_tracker_ready_get(t, 0, &text, NULL, NULL, NULL, NULL));
  if (strcmp(text, "hello world") != 0) return 7;
  ohw_tracker_free(t);
  if (ohw_extract_chunk(pcm, 16000, 0, 8000, NULL, 0) != 17600) return 8;
  ohw_vad_config vc;
  ohw_default_vad_config(&vc);
  ohw_vad_state* vs = ohw_vad_state_new(&vc, 16000);
  if (!vs) return 9;
  ohw_vad_state_free(vs);
  if (ohw_dsp_resample_sinc(pcm, 16000, 16000, 16000, NULL, 0) != 16000) return 10;
  printf("host ok (abi %d)\n", ohw_abi_version());
  return 0;
}

static int with_gpu(const char* model_path) {
  ohw_hparams hp;
  memset(&hp, 0, sizeof hp);
  /* the test suite's 'micro' preset */
  hp.n_vocab = 51865; hp.n_audio_ctx = 1500; hp.n_audio_state = 256; hp.n_audio_head = 4; hp.n_audio_layer = 2;
  hp.n_text_ctx = 448; hp.n_text_state = 256; hp.n_text_head = 4; hp.n_text_layer = 2; hp.n_mels = 80; hp.ftype = 1;
  ohw_ctx* ctx = NULL;
  CHECK(ohw_ctx_create_synthetic(&hp, 1234, 0, OHW_DTYPE_F16, &ctx));
  ohw_state* st = NULL;
  CHECK(ohw_state_create(ctx, 3, &st));
  const int64_t n = 480000;
  float* pcm = (float*)calloc((size_t)(3 * n), sizeof(float));
  if (!pcm) return 20;
  for (int b = 0; b < 3; ++b)
    for (int64_t i = 0; i < n; ++i) pcm[b * n + i] = 0.2f * sinf((0.02f + 0.01f * (float)(b % 2)) * (float)i) * (float)((i / 8000) % 3);
  int32_t ns[3] = {480000, 480000, 480000};     /* windows 0 and 2 are identical */
  CHECK(ohw_mel(st, pcm, n, ns, 3, 0, OHW_MEL_ZERO_TAIL, NULL));
  CHECK(ohw_encode(st, 3));
  ohw_sample_params sp;
  ohw_default_sample_params(ctx, &sp);
  sp.n_max = 16;
  int32_t toks[3 * 448], nt[3];
  float slp[3];
  CHECK(ohw_greedy(st, &sp, 3, toks, nt, 448, slp));
  if (nt[0] < 1 || nt[0] != nt[2] || memcmp(toks, toks + 2 * 448, (size_t)nt[0] * 4) != 0) { fprintf(stderr, "identical windows differ\n"); return 21; }
  printf("gpu ok: %d tokens per window, first %d, sum logprob %.3f\n", nt[0], toks[0], (double)slp[0]);
  ohw_state_free(st);
  ohw_ctx_free(ctx);
  if (model_path) {
    /* the product API (reference WhisperEngine::{new, transcribe}): 70 s of audio = three fixed 30 s windows, two batches */
    ohw_engine* e = NULL;
    CHECK(ohw_engine_new(model_path, "en", 0, 1, 0, OHW_DTYPE_AUTO, 2, &e));
    char text[256], lang[8];
    uint64_t ms = 0;
    ohw_audio_info info;
    const int64_t n_rec = 70 * 16000;
    CHECK(ohw_engine_transcribe(e, pcm, n_rec, 16000, text, sizeof text, lang, &ms, &info));
    const char* full = NULL;
    size_t len = 0;
    CHECK(ohw_engine_last_text(e, &full, &len));
    const ohw_window_quality* q = NULL;
    int nq = 0;
    CHECK(ohw_engine_last_quality(e, &q, &nq));
    if (nq != 3 || strcmp(lang, "en") != 0 || len == 0 || strncmp(full, text, strlen(text)) != 0) { fprintf(stderr, "engine: %d windows, lang %s, %zu bytes\n", nq, lang, len); return 22; }
    printf("engine ok: %d windows, %zu bytes of text, %llu ms\n", nq, len, (unsigned long long)ms);
    if (ohw_engine_transcribe(e, pcm, n_rec, 44100, text, sizeof text, lang, &ms, &info) != OHW_E_VALIDATION) return 23;   /* wrong rate */
    ohw_engine_free(e);
  }
  free(pcm);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "gpu") == 0) return with_gpu(argc > 2 ? argv[2] : NULL);
  return host_only();
}
