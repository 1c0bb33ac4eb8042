/*
 * vits_mi355.h — C ABI of the MI355X-native VITS2 inference path.
 *
 * This is the drop-in boundary for ONE call in the reference:
 *
 *     audio = self.model.onnx.run(None, args)[0]          vosk_tts/synth.py:123-126
 *
 * where self.model.onnx is onnxruntime.InferenceSession(model.onnx)
 * (vosk_tts/model.py:43-46) executing the graph exported from
 * SynthesizerTrn.infer (training/vits2/models.py:1679-1704) by
 * training/vits2/onnx_export.py:61-104.  Feed/outputs of that graph:
 *   "input" int64 [B,T_x], "input_lengths" int64 [B], "scales" float32 [3] =
 *   [noise_scale, length_scale, noise_scale_w] (onnx_export.py:62-64),
 *   "sid" int64 [B]  ->  "output" float32 [B,1,1,S]   (onnx_export.py:65-72,97-98)
 *
 * All entry points are extern "C", take plain pointers and sizes, return an int
 * status (0 = ok) and never throw.  Two libraries export this same ABI:
 *   libvits_mi355.so  — the product: hand-written HIP kernels for gfx950
 *   oracle/libvits_oracle.so — TEST INFRASTRUCTURE: scalar CPU restatement of the
 *                              reference arithmetic (exports the vits_* stage symbols
 *                              with the prefix vitsref_ instead of vits_)
 *
 * Layout conventions: every activation tensor is channel-major contiguous
 * [B, C, T] float32 (the reference's layout), ids/lengths are int64 like the
 * ONNX feed, durations are int32.
 */
#ifndef VITS_MI355_H
#define VITS_MI355_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITS_ABI_VERSION 1
#define VITS_MAX_UPS 4
#define VITS_MAX_RESK 4
#define VITS_MAX_RESD 4

/* Status codes */
#define VITS_OK 0
#define VITS_ERR_ARG 1      /* bad argument (null pointer, bad size, id out of range) */
#define VITS_ERR_BLOB 2     /* weight blob malformed / tensor missing / shape mismatch */
#define VITS_ERR_DEVICE 3   /* HIP runtime error (message in vits_last_error) */
#define VITS_ERR_UNSUPPORTED 4
#define VITS_ERR_NOMEM 5

/*
 * Hyper-parameters of the graph; mirrors the "model"/"data" sections of
 * training/vits2/configs/mb_istft_vits2_multi.json (line numbers in comments)
 * plus the constants hard-coded in models.py.  All fields are 4 bytes; the struct
 * is stored verbatim in the weight blob header.
 */
typedef struct vits_hparams {
  int32_t abi_version;       /* VITS_ABI_VERSION */
  int32_t n_vocab;           /* len(symbols); 62 for text/symbols.py */
  int32_t hidden_channels;   /* json:57  192 */
  int32_t inter_channels;    /* json:56  192 */
  int32_t filter_channels;   /* json:58  768 */
  int32_t n_heads;           /* json:59  2 */
  int32_t n_layers;          /* json:60  6 */
  int32_t kernel_size;       /* json:61  3  (text-encoder FFN) */
  int32_t window_size;       /* attentions.py:15 default 4 */
  int32_t gin_channels;      /* json:72  256 */
  int32_t n_speakers;        /* json:37  200 */
  int32_t enc_cond_layer;    /* attentions.py:38  2 (speaker add before this layer); -1 = no speaker-conditioned encoder */
  int32_t dp_filter_channels;/* models.py:1625  256 */
  int32_t dp_kernel_size;    /* models.py:1625  3 */
  int32_t dp_n_flows;        /* models.py:1625  4 */
  int32_t dp_num_bins;       /* modules.py:347  10 */
  int32_t dp_dds_layers;     /* models.py:38  3 */
  int32_t flow_n_flows;      /* models.py:636  4 */
  int32_t flow_wn_layers;    /* models.py:1617  4 */
  int32_t flow_kernel_size;  /* models.py:1615  5 (WN conv and pre-transformer FFN kernel) */
  int32_t flow_dilation_rate;/* models.py:1616  1 */
  int32_t dec_type;          /* 0 = Multiband_iSTFT_Generator (models.py:974), 1 = Generator (models.py:845) */
  int32_t dec_initial_channel;/* json:67 512 */
  int32_t n_ups;             /* len(json:66) 2 */
  int32_t up_rates[VITS_MAX_UPS];   /* json:66 [4,4] */
  int32_t up_kernels[VITS_MAX_UPS]; /* json:68 [16,16] */
  int32_t n_resk;            /* len(json:64) 3 */
  int32_t res_kernels[VITS_MAX_RESK];  /* json:64 [3,7,11] */
  int32_t n_resd;            /* 3 */
  int32_t res_dilations[VITS_MAX_RESK][VITS_MAX_RESD]; /* json:65 */
  int32_t subbands;          /* json:53 4 */
  int32_t istft_n_fft;       /* json:54 16 */
  int32_t istft_hop;         /* json:55 4 */
  int32_t pqmf_taps;         /* pqmf.py:53 62 */
  float   pqmf_cutoff;       /* pqmf.py:53 0.15 */
  float   pqmf_beta;         /* pqmf.py:53 9.0 */
  float   dp_tail_bound;     /* modules.py:347 5.0 */
  int32_t sampling_rate;     /* json:29 22050 */
  int32_t hop_length;        /* json:31 256 */
  int32_t bert_dim;          /* 0 = plain VITS2 (the in-repo TextEncoder).  > 0: BERT-conditioned flavour (vosk_tts/synth.py:88-99 feeds
                                "bert" [B, bert_dim, T_x]): tensors enc_p.bert_proj.{weight [hidden, bert_dim, 1], bias} exist and the
                                text encoder input is (emb(ids)*sqrt(hidden) + bert_proj(bert)) * mask -- see vits_synth_opts.bert */
  int32_t conv_precision;    /* 0 = fp32 everywhere (default; every parity figure and the headline benchmark).  1 = "bf16x3": at
                                batch size the decoder's ResBlock convs, the encoders' / flow's plain convs and the WaveNet gate convs
                                run on the bf16 matrix core with every operand split into two bf16 pieces (hi*hi + hi*lo + lo*hi,
                                fp32 accumulation; csrc/conv_bf3.hip.h) -- BASELINE configs[2]'s reduced-precision variant in the
                                form that keeps fp32-class accuracy (~4e-6 per conv) */
  int32_t reserved[6];
} vits_hparams;

/*
 * Weight blob ("VITSW001"), little-endian, produced by vosk_tts_amd/weights.py:
 *   char     magic[8]                      "VITSW001"
 *   uint32   hparams_bytes                 sizeof(vits_hparams)
 *   vits_hparams hp
 *   uint32   n_tensors
 *   n_tensors x vits_blob_entry
 *   ... float32 data, each tensor 64-byte aligned, `offset` from start of blob
 * Tensor names are the reference's state_dict keys after remove_weight_norm
 * (onnx_export.py:77-80), e.g. "dec.resblocks.0.convs1.0.weight".
 */
typedef struct vits_blob_entry {
  char     name[96];
  uint32_t ndim;
  uint32_t dims[4];
  uint32_t pad_;
  uint64_t offset;
  uint64_t nelem;
} vits_blob_entry;

typedef struct vits_model vits_model; /* opaque */

/* ---- lifecycle ------------------------------------------------------------ */

/* Parses the blob, uploads and re-lays-out weights for the MFMA kernels on
 * HIP device `device`.  Replaces onnxruntime.InferenceSession(...) (model.py:46). */
int vits_create(const void* blob, size_t blob_bytes, int device, vits_model** out);
void vits_destroy(vits_model* m);
/* Thread-local message for the last non-zero status returned on this thread. */
const char* vits_last_error(void);
int vits_get_hparams(const vits_model* m, vits_hparams* out);
/* Returns 1 for the HIP library, 0 for the CPU oracle. */
int vits_is_device_backend(void);
/* Number of HIP devices visible to the process (replicas-only multi-GPU: one vits_model per device). */
int vits_device_count(void);

/* ---- the hot path: one .run() --------------------------------------------- */

/* Options beyond the ONNX feed.  All optional pointers may be NULL.
 * The reference draws exactly two noise tensors per call (models.py:96 and :1700);
 * parity runs inject them, production runs use the library's Philox stream. */
typedef struct vits_synth_opts {
  const float*   noise_dp;         /* [B,2,T_x]  replaces torch.randn at models.py:96 (before * noise_scale_w) */
  const float*   noise_prior;      /* [B,inter,T_y_max] replaces randn_like at models.py:1700; row stride = noise_prior_stride */
  int64_t        noise_prior_stride;/* T dimension stride of noise_prior (>= max T_y) */
  const int32_t* forced_durations; /* [B,T_x] replaces w_ceil (models.py:1690); skips nothing else */
  uint64_t       seed;             /* Philox seed when noise_* are NULL */
  int32_t        max_frames;       /* 0 = unlimited; capacity bound on T_y per item (error if exceeded) */
  int32_t        flags;            /* VITS_FLAG_* */
  const uint64_t* item_seeds;      /* [B] or NULL; with VITS_FLAG_SOLO_BATCH: item b uses item_seeds[b] instead of seed + b, so a
                                      request keeps its own noise draw however a server groups requests into batches */
  const float*   bert;             /* [B, bert_dim, T_x] or NULL: the "bert" feed of the BERT-conditioned flavours (synth.py:113-120);
                                      required when hparams.bert_dim > 0.  The shipped graphs' text encoder is not in the reference tree
                                      (SURVEY.md 8f rank 2); the wiring implemented is a 1x1 projection added to the scaled embedding. */
} vits_synth_opts;

#define VITS_FLAG_NONE 0
/* Batches whose items must not depend on their neighbours (a server batching unrelated requests): every item gives what
 * a single-utterance call with seed + b gives -- its own Philox noise streams, and the decoder sees zeros beyond the
 * item's own end instead of the reference's padded-batch continuation (in the reference a batched item
 * differs from its solo run over its last ~14 frames).  Default (flag clear) = the reference's padded-batch result. */
#define VITS_FLAG_SOLO_BATCH 1

/* Host-buffer entry point (what the Python Session.run adapter calls).
 *   ids      int64 [B,T_x] (padded with anything past lengths[b])
 *   lengths  int64 [B]
 *   scales   float [3] = [noise_scale, length_scale, noise_scale_w]
 *   sid      int64 [B] (ignored when n_speakers <= 1)
 * On success *out_audio points to a library-owned float [B, *out_samples] buffer
 * (row b is valid for out_lengths[b] samples — bit-identical to the reference's padded-batch
 * result there; samples beyond out_lengths[b] + 32 frames are zeros: ragged items are not decoded
 * past their own length plus a halo wider than the decoder's receptive field), to be released
 * with vits_free_output.  Re-entrant: each call
 * uses its own workspace and HIP stream. */
int vits_synthesize(vits_model* m, const int64_t* ids, const int64_t* lengths,
                    int32_t B, int32_t T_x, const float* scales, const int64_t* sid,
                    const vits_synth_opts* opts,
                    float** out_audio, int64_t* out_samples, int64_t* out_lengths);
void vits_free_output(float* p);

/* The same call with the tail of Synth.synth_audio (vosk_tts/synth.py:127-130) done on the device:
 * pcm = int16(clip(audio * pcm_scale * 32767, -32767, 32767))  (audio_float_to_int16, synth.py:16-23; numpy's astype
 * truncates toward zero like a C cast).  Half the bytes over PCIe and no host pass over the samples.
 * *out_pcm is a library-owned int16 [B, *out_samples] buffer, released with vits_free_pcm16. */
int vits_synthesize_pcm16(vits_model* m, const int64_t* ids, const int64_t* lengths,
                          int32_t B, int32_t T_x, const float* scales, const int64_t* sid,
                          const vits_synth_opts* opts, float pcm_scale,
                          int16_t** out_pcm, int64_t* out_samples, int64_t* out_lengths);
void vits_free_pcm16(int16_t* p);

/* ---- streaming synthesis of one utterance --------------------------------------------------
 * BASELINE.json configs[4] ("long-form streaming synthesis, chunked flow+vocoder") and the transport the reference
 * already declares: `rpc ... returns (stream AudioChunk)` (server/tts_service.proto:46-54,91-95; tts_server.py:54
 * currently sends the whole utterance as one chunk).  vits_stream_open runs SynthesizerTrn.infer up to the flow
 * (models.py:1680-1701) over the whole utterance -- the flow's attention is global -- and returns the total sample
 * count; each vits_stream_next returns the next chunk_frames*hop_length samples (fewer for the last chunk, 0 at the
 * end).  The decoder runs over frame windows with a halo of the model's receptive field (>= 32 frames) either side: the
 * first chunk alone (time to first audio), afterwards eight chunks per window, decoded one window ahead of the caller
 * while the chunks of the previous window are copied out.  The concatenated chunks equal the one-shot vits_synthesize
 * output (decoder receptive field < 25 frames).  B = 1; opts as for vits_synthesize.
 * vits_stream_open_latent streams the decoder over a latent the caller holds (host float [inter_channels, T_y]): the
 * vocoder half of a two-model voice (StableTTS mel -> vocoder, vosk_tts/synth.py:113-126); flags bit 0 clamps the audio
 * to [-1, 1] (onnx/export.py:28-31). */
typedef struct vits_stream vits_stream;
int vits_stream_open(vits_model* m, const int64_t* ids, int32_t T_x, const float* scales, int64_t sid,
                     const vits_synth_opts* opts, int32_t chunk_frames, vits_stream** out, int64_t* total_samples);
int vits_stream_open_latent(vits_model* m, const float* z, int32_t T_y, int32_t chunk_frames, uint32_t flags, vits_stream** out,
                            int64_t* total_samples);
int vits_stream_next(vits_stream* st, float* audio, int64_t capacity, int64_t* n_samples);
void vits_stream_close(vits_stream* st);

/* ---- monotonic alignment search (training-side reuse) ------------------
 * monotonic_align.maximum_path (training/vits2/monotonic_align/__init__.py:6-20 -> core.pyx:7-42):
 * for each item, the monotonic path through value[t_y, t_x] (rows = frames, columns = tokens) that maximises the
 * summed score: Q[y,x] = value[y,x] + max(Q[y-1,x-1], Q[y-1,x]) inside the band
 * max(0, t_x+y-t_y) <= x < min(t_x, y+1), then backtracking from (t_y-1, t_x-1) with the reference's strict `<` tie
 * rule.  values [B,T_y,T_x] float32 (NOT modified, unlike the Cython routine which accumulates in place),
 * t_ys/t_xs int32 [B] valid extents, paths int32 [B,T_y,T_x] (0/1, zero outside the valid extents).
 * Host buffers; bit-exact with the reference (one fp32 add per cell, same operands). */
int vits_mas_maximum_path(int device, const float* values, const int32_t* t_ys, const int32_t* t_xs,
                          int32_t B, int32_t T_y, int32_t T_x, int32_t* paths);

/* Device-resident variant used by bench.py: ids/lengths/sid already in HBM
 * (int64 device pointers), audio written to a caller-provided device buffer
 * [B, audio_capacity].  Durations must be forced (device int32 [B,T_x]) or
 * opts->max_frames set so no host round trip is needed to size buffers; the
 * call is asynchronous on `stream` (a hipStream_t) unless stream == NULL. */
typedef struct vits_session vits_session; /* per-thread workspace + stream + graph cache */
int vits_session_create(vits_model* m, int32_t max_B, int32_t max_Tx, int32_t max_Ty, vits_session** out);
void vits_session_destroy(vits_session* s);
int vits_session_synthesize_device(vits_session* s, const int64_t* d_ids, const int64_t* d_lengths,
                                   int32_t B, int32_t T_x, const float* scales, const int64_t* d_sid,
                                   const int32_t* d_forced_durations, int32_t T_y_max, uint64_t seed,
                                   float* d_audio, int64_t audio_capacity, void* stream);
/* Elapsed device time (ms) of the last vits_session_synthesize_device call,
 * measured with HIP events on the session stream (blocks until it completes). */
int vits_session_last_ms(vits_session* s, float* ms);

/* Blocks until the session stream is idle; returns any deferred device-side error
 * (bad token id, frame capacity exceeded). */
int vits_session_sync(vits_session* s);
/* use_graph: replay the forward as a cached hipGraph (default 1).  profile: run eagerly and
 * bracket every kernel launch with HIP events (for vits_session_profile_report). */
int vits_session_set_options(vits_session* s, int use_graph, int profile);
/* Nodes of the most recently captured forward graph of this session = kernel launches (+ copies) per forward; 0 before a capture. */
int vits_session_graph_nodes(vits_session* s);
/* on != 0: run the stochastic duration predictor even when durations are forced (its logw is then unused): lets a
 * fixed-work benchmark time the whole of SynthesizerTrn.infer instead of skipping rows a6-a9. */
int vits_session_set_sdp_always(vits_session* s, int on);
/* Per-kernel-family totals of the profiled forwards since the last report, one line per family:
 * "<op-name> <kernel-instantiation> <launches> <total_ms> <algorithmic_flops>". */
int vits_session_profile_report(vits_session* s, char* buf, size_t cap);

/* ---- stage-level entry points (parity tests; host buffers in/out) -------- */

/* a2: TextEncoder.forward (models.py:317-326).  out x,m_p,logs_p: [B,hidden|inter,T_x] */
int vits_stage_text_encoder(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T_x,
                            const int64_t* sid, float* x, float* m_p, float* logs_p);
/* a6: StochasticDurationPredictor.forward(reverse=True) (models.py:56-63,93-101).
 * x [B,hidden,T_x] (encoder output), noise [B,2,T_x] (unit normal), out logw [B,T_x] */
int vits_stage_duration(vits_model* m, const float* x, const int64_t* lengths, int32_t B, int32_t T_x,
                        const int64_t* sid, const float* noise, float noise_scale_w, float* logw);
/* a10+a11: length regulator + prior sample (models.py:1689-1700).
 * If forced_durations != NULL logw is ignored.  out durations int32 [B,T_x], y_lengths int64 [B];
 * z_p [B,inter,T_y_cap] with row stride T_y_cap (noise has the same stride). */
int vits_stage_regulate(vits_model* m, const float* logw, const int32_t* forced_durations,
                        const int64_t* lengths, int32_t B, int32_t T_x, float length_scale,
                        const float* m_p, const float* logs_p, const float* noise, float noise_scale,
                        int32_t T_y_cap, int32_t* durations, int64_t* y_lengths, float* z_p);
/* a12: ResidualCouplingTransformersBlock.forward(reverse=True) (models.py:750-757).
 * z_p, z: [B,inter,T_y] contiguous */
int vits_stage_flow(vits_model* m, const float* z_p, const int64_t* y_lengths, int32_t B, int32_t T_y,
                    const int64_t* sid, float* z);
/* a15-a21: dec((z*y_mask), g) (models.py:1016-1054 / 872-891, :1703).  z [B,inter,T_y] already masked.
 * audio [B, T_y*hop_length]; audio_mb [B,subbands,T_y*hop/subbands] may be NULL.  sid is used only by the
 * plain Generator variant (dec_type 1: x = conv_pre(x) + cond(g), models.py:873-875) and may be NULL. */
int vits_stage_decoder(vits_model* m, const float* z, int32_t B, int32_t T_y, const int64_t* sid, float* audio, float* audio_mb);

/* Single generic op for kernel-level parity: y = conv1d(act(x)) with the library's
 * main MFMA conv kernel.  x [B,C_in,T], w [C_out,C_in,K] (PyTorch layout), bias may be NULL.
 * lrelu_slope == 1.0f means no activation. */
int vits_op_conv1d(int device, const float* x, const float* w, const float* bias, int32_t B, int32_t C_in,
                   int32_t C_out, int32_t T, int32_t K, int32_t dilation, float lrelu_slope, float* y);

/* State of the persistent programs in this process.  A poll timeout (the program's workgroups were not all co-resident: another
 * process on the device, a transient) turns them off for VITS_PERSIST_REARM_MS (environment, default 1000) and they are re-armed
 * by the first call after that; the interval doubles (up to 64 x) while timeouts keep following the re-arms.  A server logs this
 * (vosk_tts_amd.session.VitsSession does, at WARNING, whenever `timeouts` grows). */
typedef struct vits_persist_info {
  int32_t configured_mask;     /* VITS_PERSIST (environment) / vits_debug_persist (vits_mi355_debug.h): 1 duration predictor, 2 text encoder, 4 flow */
  int32_t active_mask;         /* what a call would use now: 0 while switched off after a timeout */
  int32_t off_for_ms;          /* milliseconds until the re-arm (0 = armed) */
  int32_t timeouts;            /* poll timeouts since the library was loaded */
  int32_t rearms;              /* re-arms since the library was loaded */
  int32_t launches;            /* persistent launches of `m` that ran to completion (-1 without a model) */
  int32_t process_owns_device; /* outcome of the LAST call's lease of the cross-process lock of m's device (the lock is taken per call, not
                                * held): 1 got it, -1 was refused (another process had it), 0 not asked yet; -1 without a model */
  int32_t reserved;
} vits_persist_info;
int vits_persist_state(vits_model* m, vits_persist_info* out);

/* Algorithmic FLOPs of one forward (SURVEY.md §8a/§8d formula evaluated on the
 * model's own hparams): used by bench.py for the roofline line. */
double vits_algorithmic_flops(const vits_model* m, int32_t B, int32_t T_x, int32_t T_y);

#ifdef __cplusplus
}
#endif
#endif /* VITS_MI355_H */
