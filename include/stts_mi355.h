/*
 * stts_mi355.h — C ABI of the MI355X-native StableTTS / Matcha ("multistream") inference path.
 *
 * Second model family behind the same boundary as include/vits_mi355.h: the graph that
 * vosk_tts/synth.py:64-87 feeds when config["model_type"] is multistream_v1/v2/v3
 *   input [1,5,T] int64, input_lengths [1], scales [3], sid [1], bert [1,768,T], phone_duration_extra [1,T]
 * (training/stabletts/matcha/onnx/export.py:64-98) and that the exporter builds from
 *   MatchaTTS.synthesise (matcha/models/matcha_tts.py:93-211)  +  vocoder.decode(mel).clamp(-1,1) (export.py:21-32).
 * SURVEY.md §8f rank 3.  Path relative file:line citations below are under training/stabletts/matcha/.
 *
 * Weights: an "STTSW001" blob (vosk_tts_amd/weights_stts.py; tensor names = MatchaTTS.state_dict() keys, the mel
 * `encoder.encoder.*` stack that only feeds `mel_enc` is not part of the path) plus a vocoder-only VITSW001 model
 * (vits_create on a blob with n_vocab = 0, e.g. the bundled HiFi-GAN V1) for vocoder.decode.
 *
 * The reference drives this graph with B = 1 and input_lengths = T (synth.py:69-70, export.py:69-70), and
 * synthesise() itself is written for one utterance (matcha_tts.py:150,174,182); stts_synthesize takes one utterance.
 * The stage entry points are batch-capable.  Same error codes and conventions as vits_mi355.h.
 */
#ifndef STTS_MI355_H
#define STTS_MI355_H

#include <stddef.h>
#include <stdint.h>

#include "vits_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STTS_ABI_VERSION 1

typedef struct stts_hparams {
  int32_t abi_version;
  int32_t n_vocab, n_spks, spk_emb_dim, n_feats;
  int32_t emb_dim, punc_dim, bert_dim, bert_proj_dim; /* text_encoder.py:98-108 */
  int32_t enc_hidden, enc_filter, enc_heads, enc_layers, enc_kernel, dp_out; /* dp_encoder, text_encoder.py:84-93 */
  int32_t dec_hidden, dec_filter, dec_heads, dec_layers, dec_kernel; /* estimator, flow_matching.py:300 */
  int32_t n_timesteps;  /* Euler steps baked into the exported graph (onnx/export.py:111), default 5 */
  float guidance_scale; /* classifier-free guidance, flow_matching.py:61 (0.5) */
  float mel_mean, mel_std; /* data_statistics, baselightningmodule.py:20-28 */
  int32_t hop_length, sampling_rate;
} stts_hparams;

typedef struct stts_model stts_model;

/* vocoder: a vits_model created from a vocoder-only blob on the same device (borrowed, must outlive the stts_model);
 * NULL = mel only (stts_synthesize then needs out_audio == NULL). */
int stts_create(const void* blob, size_t blob_bytes, vits_model* vocoder, int device, stts_model** out);
void stts_destroy(stts_model* m);
const char* stts_last_error(void);
int stts_get_hparams(const stts_model* m, stts_hparams* out);

typedef struct stts_synth_opts {
  const float* noise;    /* [n_feats, noise_stride] replaces torch.randn at flow_matching.py:52 (before * temperature) */
  int64_t noise_stride;  /* >= T_y rounded up to a multiple of 4 (fix_len_compatibility, utils/model.py:14-20) */
  uint64_t seed;         /* Philox seed when noise == NULL */
  int32_t n_timesteps;   /* 0 = hparams.n_timesteps */
  int32_t flags;         /* STTS_FLAG_* */
  const uint64_t* item_seeds; /* stts_synthesize_batch only, and only read when flags & STTS_FLAG_ITEM_SEEDS: [B] Philox seed of every
                               * item (otherwise seed + b), so that what a request gets does not depend on what it was batched with
                               * (MultiDeviceSynth).  The field was appended after the first release of this struct: a caller built
                               * against the shorter struct never sets the flag, so the library never reads past what it passed */
} stts_synth_opts;
#define STTS_FLAG_ITEM_SEEDS 1

/* The hot path for ONE utterance: MatchaTTS.synthesise (matcha_tts.py:93-211) -> denormalised mel ->
 * vocoder.decode(mel).clamp(-1,1) (onnx/export.py:28-31).
 *   ids [5,T_x] int64 (phoneme stream + 4 auxiliary streams, text_encoder.py:113-127), scales float[3] =
 *   [temperature, length_scale, dp_temperature] (export.py:46-49), sid (ignored when n_spks <= 1),
 *   bert [768,T_x] or NULL (= zeros, synth.py:79), phone_duration_extra [T_x] or NULL (matcha_tts.py:149-152).
 * Outputs are library-owned (vits_free_output): audio [*out_samples] (may be requested NULL), mel [n_feats, *out_frames]
 * (optional).  Re-entrant. */
int stts_synthesize(stts_model* m, const int64_t* ids, int32_t T_x, const float* scales, int64_t sid, const float* bert,
                    const float* phone_duration_extra, const stts_synth_opts* opts, float** out_audio, int64_t* out_samples,
                    float** out_mel, int64_t* out_frames);

/* Streaming form of stts_synthesize (the reference's transport is `stream AudioChunk`, server/tts_service.proto:46-54): the
 * acoustic model runs once over the utterance (the estimator's attention is global), the vocoder is streamed over the mel
 * in chunk_frames windows (vits_stream_open_latent with the clamp).  Chunks come from vits_stream_next / vits_stream_close;
 * their concatenation equals stts_synthesize's audio for the same arguments. */
int stts_stream_open(stts_model* m, const int64_t* ids, int32_t T_x, const float* scales, int64_t sid, const float* bert,
                     const float* phone_duration_extra, const stts_synth_opts* opts, int32_t chunk_frames, vits_stream** out,
                     int64_t* total_samples);

/* Batch of B independent utterances (throughput; not part of the reference, whose synthesise() takes one): item b gives
 * exactly what stts_synthesize returns for ids[b][:, :lengths[b]], sid[b], bert[b], phone_duration_extra[b] and seed
 * opts->item_seeds[b] (opts->seed + b without item seeds) -- every kernel masks or zero-pads per item, the unmasked convs of the estimator see zeros beyond each
 * item's own padded length, the vocoder decodes every item as if alone.  ids [B,5,T_x], lengths [B], sid [B],
 * bert [B,768,T_x] or NULL, phone_duration_extra [B,T_x] or NULL; opts->noise must be NULL.
 * out_audio: library-owned float [B, *out_samples], zero beyond out_lengths[b] samples. */
int stts_synthesize_batch(stts_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T_x, const float* scales,
                          const int64_t* sid, const float* bert, const float* phone_duration_extra, const stts_synth_opts* opts,
                          float** out_audio, int64_t* out_samples, int64_t* out_lengths);

/* ---- stage-level entry points (parity tests; host buffers) ---- */
/* TextEncoder.forward (text_encoder.py:111-139): x = cat(emb*sqrt(160), 4 x punc_emb*4, bert_proj(bert)) [B,256,T]
 * (returned unmasked, as the reference does) and mu_dp = dp_encoder(x, dur_spk_emb(sid)) [B,50,T]. */
int stts_stage_encoder(stts_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T, const int64_t* sid,
                       const float* bert, float* x, float* mu_dp);
/* matcha_tts.py:144-158: logw = sum_k sigmoid(mu_dp); phone_duration_extra override; w = clamp(round(logw*ls), 1).
 * durations int32 [B,T], y_lengths int64 [B]. */
int stts_stage_durations(stts_model* m, const float* mu_dp, int32_t B, int32_t T, float length_scale,
                         const float* phone_duration_extra, int32_t* durations, int64_t* y_lengths);
/* One estimator call Decoder.forward(x, mask, mu, t, c) (components/decoder.py:105-138): x [B,80,T], mu [B,256,T],
 * c [B,128] speaker vectors, t scalar, y_lengths [B] -> out [B,80,T]. */
int stts_stage_estimator(stts_model* m, const float* x, const float* mu, const int64_t* y_lengths, int32_t B, int32_t T,
                         float t, const float* c, float* out);
/* BASECFM.forward + solve_euler with classifier-free guidance (flow_matching.py:36-108,177-189) for B = 1:
 * mu_y [256,T] (T a multiple of 4), y_length valid frames, noise [80,T] unit normal -> x_1 [80,T]. */
int stts_stage_cfm(stts_model* m, const float* mu_y, int64_t y_length, int32_t T, int64_t sid, const float* noise,
                   float temperature, int32_t n_timesteps, float* out);

/* ---- word-embedding BERT encoder ----------------------------------------------------------------------------------
 * The `bert/model.onnx` of BERT-conditioned voices (vosk_tts/model.py:59-63; called at synth.py:27-34): a HuggingFace
 * BertModel exported by training/stabletts/matcha/onnx/bert-export.py:5-13 with output hidden_states[-3], i.e. the
 * activations after layer n_layers - 2, shape [tokens, hidden].  The model itself is the third-party `transformers`
 * BertModel (post-LN encoder: embeddings word + position + token_type -> LayerNorm(eps 1e-12); per layer self-attention,
 * dense + residual + LayerNorm, dense/GELU(erf)/dense + residual + LayerNorm); parity is pinned to that implementation
 * (tests/golden/bert_*.npz, oracle/gen_golden_stts.py).  One sentence per call, attention_mask all ones, as synth.py
 * drives it.  Blob "BERTW001" (vosk_tts_amd/weights_bert.py), tensor names = BertModel.state_dict() keys. */
typedef struct bert_hparams {
  int32_t abi_version;
  int32_t vocab_size, hidden, n_layers, out_layers, n_heads, intermediate, max_position, type_vocab;
  float ln_eps;
} bert_hparams;
#define BERT_ABI_VERSION 1
typedef struct bert_model bert_model;
int stts_bert_create(const void* blob, size_t blob_bytes, int device, bert_model** out);
void stts_bert_destroy(bert_model* m);
int stts_bert_get_hparams(const bert_model* m, bert_hparams* out);
/* input_ids / token_type_ids int64 [T] (token_type_ids may be NULL = zeros); out float [T, hidden] */
int stts_bert_encode(bert_model* m, const int64_t* input_ids, const int64_t* token_type_ids, int32_t T, float* out);

#ifdef __cplusplus
}
#endif
#endif
