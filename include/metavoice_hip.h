/*
 * metavoice_hip.h -- C ABI of the MI355X-native MetaVoice-1B synthesis engine (libmetavoice_hip.so).
 *
 * The reference (metavoiceio/metavoice-src) has NO FFI / plugin interface: its boundary for this path is a
 * Python call surface.  This header is the C-ABI a binding for that surface binds to;
 * each entry point names the reference interface it replaces (paths relative to the reference repo).
 * Plain pointers and sizes only -- no torch types.  Every function returns 0 on success or a negative
 * mv_status; mv_last_error() gives the message.  All device work of one engine runs on the engine's
 * own HIP stream; an engine is NOT re-entrant (same as the reference: one global TTS, serving.py:46-51).
 *
 * Host-side mirror of the reference API built on this ABI: metavoice-src_amd/ (Python, ctypes).
 */
#ifndef METAVOICE_HIP_H
#define METAVOICE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mv_engine mv_engine;

enum mv_status {
    MV_OK = 0,
    MV_ERR_INVALID = -1,      /* bad argument / unsupported shape            (Python: ValueError)  */
    MV_ERR_HIP = -2,          /* a HIP runtime call failed                   (Python: RuntimeError) */
    MV_ERR_STATE = -3,        /* call order violated (e.g. run before begin) (Python: RuntimeError) */
    MV_ERR_MISSING_WEIGHT = -4,
    MV_ERR_PROMPT_TOO_LONG = -5 /* fam/llm/fast_inference_utils.py:203-204 -> ValueError           */
};

enum mv_dtype { MV_BF16 = 0, MV_F32 = 1, MV_I32 = 2, MV_I8 = 3 };
enum mv_weight_format { MV_W_BF16 = 0, MV_W_INT4_G128 = 1, MV_W_INT8 = 2 };

/* Mirrors ModelArgs (fam/llm/fast_model.py:52-84) plus engine sizing. */
typedef struct mv_config {
    int32_t n_layer;          /* 24   */
    int32_t n_head;           /* 16   (head_dim = dim / n_head must be 128) */
    int32_t dim;              /* 2048 */
    int32_t intermediate;     /* 5632 */
    int32_t vocab;            /* 2562 (<= 4096) */
    int32_t block_size;       /* 2048 */
    int32_t spk_dim;          /* 256  */
    float norm_eps;           /* 1e-5 */
    int32_t max_utterances;   /* utterances decoded together; each takes 2 CFG rows (fast_model.py:132-134) */
    int32_t weight_format;    /* mv_weight_format: quantisation_mode of TTS.__init__ (fast_inference.py:41-50) */
} mv_config;

/* Per-utterance sampling controls == **sampling_kwargs of generate() (fast_inference_utils.py:85-120).
 * The reference passes guidance/temperature/top_p as 0-d tensors in the model dtype, i.e. already rounded
 * to bf16 on the GPU path; the host shim does that rounding and passes plain floats here. */
typedef struct mv_sampling {
    float guidance_scale;     /* g          */
    float one_minus_guidance; /* (1 - g) as the reference computes it in the model dtype */
    float temperature;        /* clamped to >= 1e-5 by the engine (fast_inference_utils.py:92) */
    float top_p;              /* < 0 : disabled (None); 0 keeps only the largest token, as `cum <= 1 - top_p` does */
    int32_t top_k;            /* <= 0 : disabled (None) */
    uint64_t seed;            /* Philox key for the Exp(1) race noise (fast_inference_utils.py:61-65) */
} mv_sampling;

/* ---- lifecycle: replaces _load_model/build_model (fam/llm/fast_inference_utils.py:236-392) ------------- */
int mv_engine_create(const mv_config* cfg, mv_engine** out);
int mv_engine_destroy(mv_engine* e);
/* `name` is the gpt-fast tensor name AFTER the key remap of fast_inference_utils.py:250-278
 * ("tok_embeddings.weight", "layers.3.attention.wqkv.weight", ...).  `data` may be a host or a device
 * pointer (row-major [rows][cols], dtype MV_BF16 or MV_F32); the engine copies it into its own arena. */
int mv_load_weight(mv_engine* e, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols);
/* Checks every tensor is present, quantises if cfg.weight_format asks for it (numerics of
 * fam/llm/fast_quantize.py:70-132 / :38-67), allocates the static KV caches
 * (Transformer.setup_caches, fast_model.py:136-148) and captures the decode-step hipGraph
 * (the role torch.compile(mode="max-autotune") + CUDA graphs play at fast_inference_utils.py:357-361). */
int mv_finalize(mv_engine* e);
int64_t mv_model_bytes(const mv_engine* e);   /* == model_size of build_model (:348) for the stored format */

/* ---- generation: replaces generate()/prefill()/decode_n_tokens() (fast_inference_utils.py:123-228) ----- */
/* prompts: host int32 [n_utt][max_len] (row u valid for prompt_lens[u] entries); spk_emb: host bf16 bits
 * [n_utt][spk_dim]; max_new_tokens < 0 means "up to block_size" (generate(): :197-204).
 * prefill_mode 0 = teacher-forced decode steps, 1 = batched MFMA prefill. */
int mv_begin(mv_engine* e, int32_t n_utt, const int32_t* prompts, const int32_t* prompt_lens, int32_t max_len,
             const uint16_t* spk_emb, const mv_sampling* sampling, int32_t max_new_tokens, int32_t eoa_token,
             int32_t prefill_mode);
/* Runs decode steps (one hipGraph replay each) until every utterance has drawn its end-of-audio token or hit
 * its length cap, or max_steps (<0: no limit) have run.  `poll_every` steps the host reads the device-side
 * "all done" word -- the reference synchronises every token (fast_inference_utils.py:161). */
int mv_run(mv_engine* e, int32_t max_steps, int32_t poll_every, int32_t* steps_run);
float mv_last_run_ms(const mv_engine* e);      /* HIP-event time of the last mv_run on the engine stream */
/* Which decode path mv_run takes now: 2 = persistent kernel drawing the tokens itself (several steps per launch), 1 = persistent
 * kernel + sampler launch per step, 0 = per-layer launches (configuration not supported, mv_tune("mega", 0), or sticky after a
 * launch of the persistent kernel was abandoned).  *timeouts (optional) = persistent launches abandoned so far: a run that fell
 * back produces the same bits, so a parity test that wants to cover the persistent kernel asserts this stays 0. */
int32_t mv_decode_path(const mv_engine* e, int32_t* timeouts);
/* Full sequence prompt ++ generated (generate() return value, :226-228), EOA included when drawn. */
int mv_get_tokens(mv_engine* e, int32_t utt, int32_t* out, int32_t cap, int32_t* n_out);

/* ---- parity hooks (no reference equivalent; SURVEY.md 8b "mv_logits_debug") --------------------------- */
/* When enabled before mv_begin, every step stores the raw logits of all rows (fp32 [step][rows][vocab]). */
int mv_set_logits_capture(mv_engine* e, int32_t enable, int32_t max_steps);
int mv_get_captured_logits(mv_engine* e, float* out, int32_t first_step, int32_t n_steps);

/* ---- single-kernel entry points on caller-owned DEVICE buffers (used by tests and the roofline probe) -- */
/* out[r][n] = bf16( sum_k bf16_x[r][k] * W[n][k] ), optional fused RMSNorm prologue (norm_w != NULL) and
 * residual epilogue (residual != NULL): the building block of K2+K3, K6, K7, K8. */
int mv_op_gemv(void* stream, int32_t rows, int32_t N, int32_t K, const void* x, const void* w, const void* norm_w,
               float eps, const void* residual, void* out);
int mv_op_swiglu_gemv(void* stream, int32_t rows, int32_t N, int32_t K, const void* x, const void* w1, const void* w3,
                      const void* norm_w, float eps, void* out);
/* Split-KV decode attention over keys [0, n_keys) of a [rows][n_head][cache_len][128] bf16 cache. */
int mv_op_attention_decode(void* stream, int32_t rows, int32_t n_head, int32_t cache_len, const int32_t* n_keys_dev,
                           const void* q, const void* kcache, const void* vcache, void* scratch, void* out);
int64_t mv_op_attention_scratch_bytes(int32_t rows, int32_t n_head);
/* CFG mix + temperature + top-k/top-p + Exp(1) race on one (2, vocab) bf16 logits pair; writes token and the
 * kept-probabilities vector (fp32 [vocab], original token order). */
int mv_op_sample(void* stream, int32_t vocab, const void* logits_pair, const mv_sampling* s, uint32_t step,
                 int32_t* token_out, float* probs_out);

/* int4 group-128 weight-only path (fam/llm/fast_quantize.py:70-132,204-216).  w: [N][K] bf16 -> codes [N][K/2] bytes (column
 * 2j low nibble, 2j+1 high nibble of byte j), scales_zeros [N][K/128] uint32 (bf16 scale | bf16 zero << 16), optional
 * dequantised [N][K] bf16 (may alias w).  mv_op_gemv_int4: out = epilogue(x . dequant(W)^T) like mv_op_gemv /
 * mv_op_swiglu_gemv (codes2 != NULL selects the SwiGLU form). */
int mv_op_quantize_int4(void* stream, int32_t N, int32_t K, const void* w, void* codes, void* scales_zeros, void* dequantised);
int mv_op_gemv_int4(void* stream, int32_t rows, int32_t N, int32_t K, const void* x, const void* codes, const void* scales_zeros,
                    const void* codes2, const void* scales_zeros2, const void* norm_w, float eps, const void* residual, void* out);

/* Many-row blocks (first-stage prefill, second stage): out[M][N] = epilogue(a[M][K] . w[N][K]^T), bf16 in/out, fp32 MFMA
 * accumulate; w2 != NULL: SwiGLU gate bf16(bf16(silu(a.w^T + bias)) * bf16(a.w2^T + bias2)); gelu: exact-erf GELU;
 * residual [M][N] bf16 added after the bf16 rounding of the product (fam/llm/fast_model.py:179-180,237). */
int mv_op_gemm(void* stream, int32_t M, int32_t N, int32_t K, const void* a, const void* w, const void* w2, const void* bias,
               const void* bias2, const void* residual, int32_t gelu, void* out);
/* softmax(q k^T / sqrt(head_dim)) v for every query position of a packed c_attn output [batch][T][3][n_head][head_dim]
 * (fam/llm/layers/attn.py:132-159; causal = 0 is the second-stage case), head_dim 64 or 128. */
int mv_op_attention_rows(void* stream, int32_t batch, int32_t T, int32_t n_head, int32_t head_dim, int32_t causal, const void* qkv,
                         void* out);

/* Roofline probe: launches one kernel kind of the decode step `iters` times back to back on the engine stream, cycling
 * through the layers' weights (so the stream is HBM-cold: 24 x 46 MB >> the 256 MB Infinity Cache), bracketed by HIP
 * events on that stream.  Returns the average launch duration and the ALGORITHMIC bytes one launch must move
 * (SURVEY.md 8d).  kind: 0 norm+wqkv+kv-write, 1 attention (at the utterances' current positions), 2 wo+residual,
 * 3 norm+w1|w3+silu-gate, 4 w2+residual, 5 norm+LM head. */
int mv_probe_kernel(mv_engine* e, int32_t kind, int32_t iters, float* avg_ms, double* bytes_per_launch);

/* ---- second stage: replaces Model(...)/GPT for the non-causal model (fam/llm/inference.py:53-156, fam/llm/model.py:79-156,
 * 195-314) and _non_causal_sample (fam/llm/mixins/non_causal.py:15-67).  Sizes mirror GPTConfig (model.py:27-47) and come
 * from second_stage.pt["model_args"]; tensor names are the checkpoint's own ("transformer.h.0.attn.c_attn.weight", ...). */
typedef struct mv_stage2 mv_stage2;
typedef struct mv_stage2_config {
    int32_t n_layer, n_head, n_embd;
    int32_t hidden;            /* MLP width (layers.py:45-58) */
    int32_t block_size, spk_dim;
    int32_t n_in, n_out;       /* input hierarchies (2), target hierarchies (6) */
    int32_t vocab_in[4], vocab_out[8];
    int32_t bias;              /* GPTConfig.bias */
    int32_t norm_type;         /* 0 rmsnorm, 1 layernorm */
    int32_t act;               /* 0 swiglu, 1 gelu */
    int32_t causal;            /* 0 for the second stage */
    float norm_eps;            /* rmsnorm_eps */
    int32_t max_batch;
} mv_stage2_config;
int mv_stage2_create(const mv_stage2_config* cfg, mv_stage2** out);
int mv_stage2_destroy(mv_stage2* s);
int mv_stage2_load_weight(mv_stage2* s, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols);
int mv_stage2_finalize(mv_stage2* s);
/* idx: host int32 [batch][n_in][T]; spk_emb: host bf16 bits [batch][spk_dim] or NULL.  Logits stay on the device. */
int mv_stage2_forward(mv_stage2* s, int32_t batch, int32_t T, const int32_t* idx, const uint16_t* spk_emb);
int mv_stage2_get_logits(mv_stage2* s, int32_t head, float* out);  /* parity hook: fp32 [batch*T][vocab_out[head]] */
/* One draw per (position, target hierarchy): /temperature, top-k mask, softmax, multinomial; out: host [batch][n_out][T]. */
int mv_stage2_sample(mv_stage2* s, float temperature, int32_t top_k, uint64_t seed, int32_t* out);

/* ---- EnCodec 24 kHz token -> waveform decode (residual-VQ lookup + SEANet decoder, fp32).  Replaces the codec half of
 * `mbd.tokens_to_wav` / `codec_model.decode` reached through fam/llm/decoders.py:13,84-85 (audiocraft==1.2.0 / encodec==0.1.1,
 * third party).  Tensor names are those of transformers' EncodecModel state dict with weight-norm folded
 * ("decoder.layers.3.conv.weight", "decoder.layers.1.lstm.weight_hh_l0", "quantizer.layers.0.codebook.embed", ...). */
typedef struct mv_codec mv_codec;
typedef struct mv_codec_config {
    int32_t hidden;              /* 128: codebook / latent dimension */
    int32_t num_filters;         /* 32  */
    int32_t n_ratios;            /* 4   */
    int32_t ratios[8];           /* 8, 5, 4, 2 (decoder order) */
    int32_t kernel_size;         /* 7   */
    int32_t last_kernel_size;    /* 7   */
    int32_t residual_kernel_size;/* 3   */
    int32_t compress;            /* 2   */
    int32_t lstm_layers;         /* 2   */
    int32_t n_q;                 /* 8 codebooks at 6 kbps */
    int32_t bins;                /* 1024 */
    int32_t audio_channels;      /* 1   */
} mv_codec_config;
int mv_codec_create(const mv_codec_config* cfg, mv_codec** out);
int mv_codec_destroy(mv_codec* c);
int mv_codec_load_weight(mv_codec* c, const char* name, const float* data, int64_t numel);  /* host fp32 */
int mv_codec_finalize(mv_codec* c);
int64_t mv_codec_samples(const mv_codec* c, int32_t frames);   /* frames x prod(ratios) */
/* tokens: host int32 [n_q][frames]; wav_out: host fp32 [audio_channels][samples]; latent_out (optional): host fp32
 * [hidden][frames], the summed codebook vectors (`decode_latent`, the diffusion decoder's conditioning). */
int mv_codec_decode(mv_codec* c, const int32_t* tokens, int32_t n_q, int32_t frames, float* wav_out, float* latent_out);
/* The same decode with the results LEFT ON THE DEVICE for the next stage (mv_mbd_generate_device): *d_wav fp32 [samples] and
 * *d_latent fp32 [hidden][frames] point into the codec's own workspace and stay valid until the next call on this codec; the
 * codec's stream has been synchronised when the call returns.  (fam/llm/decoders.py:84-85: `mbd.tokens_to_wav` hands the codec's
 * outputs to the diffusion bands on the GPU; so does this pair of entry points -- no PCIe leg between the two stages.) */
int mv_codec_decode_device(mv_codec* c, const int32_t* tokens, int32_t n_q, int32_t frames, const float** d_wav, const float** d_latent);

/* ---- multi-band diffusion vocoder: replaces `mbd.tokens_to_wav` after the codec decode (fam/llm/decoders.py:13,84-85 ->
 * audiocraft==1.2.0 MultiBandDiffusion.generate / re_eq, third party): n_models DiffusionUnets (one per frequency band), n_eval
 * evaluations each under the sub-sampled DDPM sampler, MultiBandProcessor.return_sample, sum, 32-band re-equalisation against the
 * EnCodec decode.  Sizes come from the checkpoint's own cfg entries (host shim: metavoice-src_amd/mbd.py); tensor names are those of
 * audiocraft's DiffusionUnet state dict ("encoders.0.conv.weight", "decoders.3.res_blocks.0.conv1.bias", "embeddings.1.weight", ...). */
typedef struct mv_mbd mv_mbd;
typedef struct mv_mbd_config {
    int32_t n_models;        /* 4 bands                                               */
    int32_t chin;            /* 1 (mono)                                              */
    int32_t depth;           /* 4                                                     */
    int32_t channels[8];     /* hidden * growth^i: 48, 192, 768, 3072                 */
    int32_t kernel, stride;  /* 8, 4 (kernel must be 2 x stride)                      */
    int32_t norm_groups;     /* 4                                                     */
    int32_t res_blocks;      /* 1                                                     */
    int32_t act;             /* 0 relu, 1 gelu, 2 silu                                */
    int32_t codec_dim;       /* 128, or 0: unconditioned                              */
    int32_t emb_all_layers;  /* step embedding on every encoder level                 */
    int32_t bilstm;          /* BiLSTM bottleneck (without one the bottleneck is zeroed, as in audiocraft's forward) */
    int32_t num_steps;       /* 1000 diffusion steps in the embedding tables          */
    int32_t n_eval;          /* 20 evaluations (step_list 999, 949, ..., 49)          */
    int32_t proc_bands;      /* 8  mel bands of MultiBandProcessor, 1 = no processor  */
    int32_t eq_bands;        /* 32 mel bands of re_eq, 1 = no re-equalisation         */
    int32_t precision;       /* MFMA terms per product: 1 bf16, 3 bf16 hi/lo split (~fp32) */
} mv_mbd_config;
int mv_mbd_create(const mv_mbd_config* cfg, mv_mbd** out);
int mv_mbd_destroy(mv_mbd* m);
int mv_mbd_load_weight(mv_mbd* m, int32_t band, const char* name, const float* data, int64_t numel);   /* host fp32, torch layouts */
/* the scalars of NoiseSchedule.generate_subsampled for the n_eval evaluations: previous = (current - est_c * estimate) * inv_sqrt_alpha
 * + sigma * noise, clamped to +-clip (0: none), times `rescale` after step 0 */
int mv_mbd_set_schedule(mv_mbd* m, int32_t band, const int32_t* steps, const float* est_c, const float* inv_sqrt_alpha, const float* sigma,
                        float clip, float rescale, float noise_scale);
/* MultiBandProcessor.return_sample: band b of the model output times scale[b] plus mean[b] */
int mv_mbd_set_processor(mv_mbd* m, int32_t band, int32_t n_bands, const float* scale, const float* mean);
/* julius.LowPassFilters bank of SplitBands: which 0 = processor (proc_bands - 1 filters), 1 = re_eq (eq_bands - 1); taps odd */
int mv_mbd_set_filterbank(mv_mbd* m, int32_t which, int32_t n_filters, int32_t taps, const float* coeffs);
int mv_mbd_finalize(mv_mbd* m);
/* a second handle on the same (finalized) weights with its own stream and device workspace: the utterances of a batch are vocoded on several
 * lanes concurrently (they differ in length and GroupNorm spans a whole utterance, so there is no batch dimension).  Destroy clones first. */
int mv_mbd_clone(mv_mbd* src, mv_mbd** out);
/* latent: host fp32 [codec_dim][frames] (mv_codec_decode's latent_out); wav_ref: host fp32 [samples], the EnCodec decode (NULL: no
 * re_eq); noise: host fp32 [n_models][n_eval + 1][samples] injected Gaussian draws (parity tests) or NULL = Philox stream of `seed` */
int mv_mbd_generate(mv_mbd* m, const float* latent, int32_t frames, const float* wav_ref, int32_t samples, const float* noise, uint64_t seed,
                    float* wav_out);
/* the same with the codec's outputs read in place from DEVICE memory (mv_codec_decode_device): d_latent fp32 [codec_dim][frames]
 * (NULL when codec_dim == 0), d_wav_ref fp32 [samples] (NULL: no re_eq); the engine's own Philox noise stream */
int mv_mbd_generate_device(mv_mbd* m, const float* d_latent, int32_t frames, const float* d_wav_ref, int32_t samples, uint64_t seed,
                           float* wav_out);
/* parity hook: one DiffusionUnet evaluation of band `band` at diffusion step `step` on x host fp32 [samples] */
int mv_mbd_unet_forward(mv_mbd* m, int32_t band, const float* x, int32_t samples, int32_t step, const float* latent, int32_t frames, float* out);
/* parity hook at the launch shapes of mv_mbd_generate: one evaluation of EVERY band model in one pass (all bands per launch decides which
 * GEMM kernel a layer runs on).  x: host fp32 [n_models][samples]; out: host fp32 [samples] = sum over the bands of unet_b(x_b). */
int mv_mbd_unet_forward_sum(mv_mbd* m, const float* x, int32_t samples, int32_t step, const float* latent, int32_t frames, float* out);
/* launches per convolution-GEMM kernel since mv_mbd_create, out[8]: [0] mbd_gemm_t256_kernel, [1] mbd_gemm_glds3_kernel<64,192>,
 * [2] mbd_gemm_glds3_kernel<128,128>, [3] mbd_gemm_thin_kernel, [4] mbd_gemm_glds_kernel, [5] mbd_gemm_kernel<128>, [6] mbd_gemm_kernel<64>;
 * [7] (not a kernel) how many of those launches were transposed convolutions whose epilogue also wrote the decoder's `z + skip` */
int mv_mbd_kernel_counts(const mv_mbd* m, int64_t* out);

/* ---- speaker encoder: replaces SpeakerEncoder.forward (fam/quantiser/audio/speaker_encoder/model.py:21-59): 3-layer LSTM over
 * 40-bin mel frames of the partial utterances, relu(Linear(h_T)), L2-normalised per partial.  Tensor names are those of
 * `checkpoint["model_state"]`: lstm.weight_ih_l0 ... lstm.bias_hh_l2, linear.weight, linear.bias (host fp32).  The mel front
 * end (speaker_encoder/audio.py) and the mean over partials (model.py:98-103) stay on the host, as in the reference. */
typedef struct mv_spkenc mv_spkenc;
typedef struct mv_spkenc_config {
    int32_t n_mels;    /* 40  (mel_n_channels)      */
    int32_t hidden;    /* 256 (model_hidden_size)   */
    int32_t n_layers;  /* 3   (model_num_layers)    */
    int32_t emb;       /* 256 (model_embedding_size)*/
} mv_spkenc_config;
int mv_spkenc_create(const mv_spkenc_config* cfg, mv_spkenc** out);
int mv_spkenc_destroy(mv_spkenc* e);
int mv_spkenc_load_weight(mv_spkenc* e, const char* name, const float* data, int64_t numel);
int mv_spkenc_finalize(mv_spkenc* e);
/* mels: host fp32 [n_partials][frames][n_mels]; out: host fp32 [n_partials][emb]. */
int mv_spkenc_embed(mv_spkenc* e, const float* mels, int32_t n_partials, int32_t frames, float* out);

/* ---- speaker-reference front end on the GPU: what SpeakerEncoder.embed_utterance_from_file does to the audio through librosa before the
 * LSTM (fam/quantiser/audio/speaker_encoder/model.py:112-117, audio.py:10-22).  Host buffers in and out; filters are designed on the host
 * (metavoice-src_amd/speaker_encoder/audio.py).
 * mv_audio_resample : librosa.load(sr=16000)'s resampling as a polyphase FIR (scipy.signal.upfirdn's definition):
 *                     y[m] = sum_k h[(m + n_pre_remove) * down - k * up] * x[k], m < n_out
 * mv_audio_frame_rms: librosa.effects.trim's frame RMS: centred frames (zero padding frame_length / 2), rms_out[n_frames] in double,
 *                     n_frames = 1 + (n + 2 (frame_length / 2) - frame_length) / hop_length
 * mv_audio_mel      : librosa.feature.melspectrogram(power=2): centred frames of n_fft samples (zero padded), times `window`, |DFT|^2,
 *                     times filterbank [n_mels][n_fft / 2 + 1] -> mel_out [n_frames][n_mels], n_frames = 1 + n / hop_length */
/* mv_audio_set_device: the HIP device the three calls below work on (-1, the default: the calling thread's current device).  Process-wide; a
 *                     server whose request threads are new threads (current device 0) sets it once per rank. */
int mv_audio_set_device(int32_t device);
int mv_audio_resample(const float* x, int64_t n, const float* h, int32_t taps, int32_t up, int32_t down, int64_t n_pre_remove, int64_t n_out, float* y);
int mv_audio_frame_rms(const float* x, int64_t n, int32_t frame_length, int32_t hop_length, double* rms_out, int64_t n_frames);
int mv_audio_mel(const float* wav, int64_t n, const float* window, const float* filterbank, int32_t n_fft, int32_t hop_length, int32_t n_mels,
                 float* mel_out, int64_t n_frames);

/* Debug hook (MV_DEBUG_VECTORS=1 in the environment at mv_finalize, MetaVoice-1B widths only): bf16 bits of every layer's
 * q | attention output | h | SwiGLU activation | x of the LAST decode step, [n_layer][5][2 rows][5632]. */
int mv_debug_vectors(mv_engine* e, uint16_t* out, int64_t cap_elems);
/* Profiling hook (MV_MEGA_STAMPS=1): s_memtime stamps [2 workgroups: first, last][n_layer][12] of the persistent kernel's
 * phases in the LAST decode step (see tools/mega_stamps.py for the slot meanings). */
int mv_debug_stamps(mv_engine* e, uint64_t* out, int64_t cap_elems);

/* Tuning hook (no reference equivalent): process-wide launch knobs of the decode kernels and the side-stream prefetcher,
 * used by tools/sweep.py; keys are listed in DESIGN.md "Tuning".  A finalized engine passed as `e` (may be NULL)
 * re-captures its decode graph at the next mv_begin. */
int mv_tune(mv_engine* e, const char* key, int32_t value);

const char* mv_last_error(void);
const char* mv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* METAVOICE_HIP_H */
