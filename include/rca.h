/*
 * rca.h -- C ABI of the MI355X-native duplex codec-LM hot path.
 *
 * The reference (AbrahamSanders/realtime-codec-agent) has no FFI of its own: its
 * seams are two duck-typed Python objects held by RealtimeAgentResources
 * (realtime_agent_resources.py:19-39).  This header is the C boundary that sits
 * *behind* Python classes with those surfaces.  Each entry
 * point cites the reference call it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = RCA_OK, <0 = error (rca_last_error() has text);
 *     no exception crosses the boundary
 *   - plain pointers and sizes only; no torch / C++ types
 *   - the caller owns every I/O buffer; the library owns weights, KV cache, workspace
 *   - one handle per stream, handles are not re-entrant (matches the reference:
 *     AudioTokenizer and llama_cpp.Llama are single-threaded objects)
 *   - "_dev" variants take device (HBM) pointers and a hipStream_t passed as void*;
 *     the plain variants take host pointers and do the H2D/D2H copies themselves
 */
#ifndef RCA_H
#define RCA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCA_OK 0
#define RCA_ERR_ARG -1      /* bad argument / shape */
#define RCA_ERR_HIP -2      /* a HIP runtime call failed */
#define RCA_ERR_STATE -3    /* wrong call order / context overflow */
#define RCA_ERR_MISSING -4  /* a named tensor was not supplied */

#define RCA_MAX_STAGES 8

typedef struct rca_codec rca_codec_t;
typedef struct rca_lm rca_lm_t;

/* dtype tags for rca_tensor_t */
#define RCA_F32 0
#define RCA_BF16 1
#define RCA_Q8_0 2   /* GGUF block_q8_0 as stored in the file: per 32 values one fp16 scale then 32 int8 (34 bytes); numel = values */
#define RCA_F16 3    /* IEEE half (the reference's default model file is an F16 GGUF, realtime_agent_resources.py:12) */
#define RCA_Q6_K 5   /* GGUF block_q6_K as stored in the file (210 bytes per 256 values: low nibbles, high bit pairs, 16 int8 scales, fp16 d): the
                        format llama-quantize Q4_K_M gives output.weight and some attn_v / ffn_down tensors */
#define RCA_Q4_K 4   /* GGUF block_q4_K as stored in the file: per 256 values fp16 d, fp16 dmin, 12 bytes of 6-bit scales / minima, 128 bytes
                        of nibbles (144 bytes); numel = values.  What llama-quantize Q4_K_M writes for most tensors (prep_test_model.sh:31) */

/* A named host tensor handed to a create() call (weights). */
typedef struct {
    const char* name;
    const void* data;  /* host pointer */
    int64_t numel;
    int32_t dtype;     /* RCA_F32, RCA_BF16, RCA_F16, RCA_Q8_0, RCA_Q4_K or RCA_Q6_K (the last three: LM matrices and embedding table only) */
} rca_tensor_t;

const char* rca_last_error(void);
int rca_device_count(int* n);
/* hipDeviceSynchronize on `device`: the duplex profiler's fence before every timestamp (the reference's
 * realtime_agent_profiler.py:30-38 takes its timestamps without one) */
int rca_device_sync(int32_t device);
const char* rca_version(void);

/* ------------------------------------------------------------------ codec --
 * MagiCodec-style model object.  Architecture is described by
 * rca_codec_config_t; tensors are named as in oracle/codec_ref.py:
 *   enc.conv_in.{weight,bias}            [C0,1,k_in]
 *   enc.down.{i}.{weight,bias}           [C(i+1),C(i),2*s_i]
 *   enc.conv_out.{weight,bias}           [D,C_last,k_latent]
 *   quantizer.in_proj.{weight,bias}      [cd,D]
 *   quantizer.codebook.weight            [N,raw]
 *   quantizer.codebook_proj.{weight,bias}[cd,raw]
 *   dec.conv_in.{weight,bias}            [C_last,cd,k_latent]
 *   dec.up.{i}.{weight,bias}             [C(n-i),C(n-i-1),2*s]  (ConvTranspose1d layout [Cin,Cout,k])
 *   dec.conv_out.{weight,bias}           [1,C0,k_in]
 */
typedef struct {
    int32_t sample_rate;                  /* 16000 */
    int32_t n_stages;                     /* 4 */
    int32_t strides[RCA_MAX_STAGES];      /* 2,4,5,8 -> hop 320 (50 Hz) */
    int32_t channels[RCA_MAX_STAGES + 1]; /* 32,64,128,256,512 */
    int32_t k_in;                         /* 7 */
    int32_t k_latent;                     /* 3 */
    int32_t latent_dim;                   /* 256 */
    int32_t codebook_size;                /* 131072 */
    int32_t codebook_raw_dim;             /* 32 */
    int32_t codebook_dim;                 /* 16 */
    float leaky_slope;                    /* 0.1 */
} rca_codec_config_t;

/* replaces codec_bpe.tools.codec_utils.load_magicodec_model + .eval().to(device)
 * (audio_tokenizer.py:26-28).  The projected codebook and its half-norms are
 * computed once here instead of on every decode (audio_tokenizer.py:198). */
int rca_codec_create(const rca_codec_config_t* cfg, const rca_tensor_t* tensors, int32_t n_tensors,
                     int32_t device, rca_codec_t** out);
int rca_codec_destroy(rca_codec_t* h);
/* total stride of the encoder (samples per frame) */
int rca_codec_hop(const rca_codec_t* h, int32_t* hop);
/* frames produced for T samples: ceil(T / hop) (pad_audio right-pads to a hop multiple) */
int rca_codec_num_frames(const rca_codec_t* h, int32_t T, int32_t* F);

/* AudioTokenizer._magicodec_encode (audio_tokenizer.py:189-194):
 * pad_audio -> encoder -> quantizer.inference.  pcm [B,T] f32 -> codes [B,F] int64. */
int rca_codec_encode(rca_codec_t* h, const float* pcm_host, int32_t B, int32_t T, int64_t* codes_host);
int rca_codec_encode_dev(rca_codec_t* h, const float* pcm_dev, int32_t B, int32_t T, int64_t* codes_dev,
                         void* stream);

/* codec_bpe.audio_to_codes as driven by encode_audio_gpu_*.sh (--chunk_size_secs,
 * --context_secs, --batch_size): every `hop_samples` chunk of a long signal is
 * encoded with `ctx_samples` of left context and only the chunk's own codes are
 * kept (same rule as AudioTokenizer.tokenize_audio, audio_tokenizer.py:72-101).
 * audio [C,N] f32 resident in HBM -> codes [C, n_chunks*frames_per_chunk] int64.
 * Windows are cut on the device; `batch_windows` windows are encoded per pass. */
int rca_codec_encode_windows_dev(rca_codec_t* h, const float* audio_dev, int32_t C, int64_t N,
                                 int32_t chunk_samples, int32_t ctx_samples, int32_t batch_windows,
                                 int64_t* codes_dev, int64_t codes_per_channel, void* stream);

/* Same, restricted to chunks [chunk_begin, chunk_end) of the signal (left context is read from
 * the signal itself): the unit of work a batch shard / one bench step encodes.  Codes of chunk i
 * land at codes_dev[c * codes_per_channel + (i - chunk_begin) * frames_per_chunk ...]. */
int rca_codec_encode_chunk_range_dev(rca_codec_t* h, const float* audio_dev, int32_t C, int64_t N,
                                     int32_t chunk_samples, int32_t ctx_samples, int32_t batch_windows,
                                     int64_t chunk_begin, int64_t chunk_end, int64_t* codes_dev,
                                     int64_t codes_per_channel, void* stream);

/* The same per-window arithmetic for B windows of T samples taken ANYWHERE in one device buffer -- window b starts at
 * audio_dev + src_off_dev[b] (elements), its last n_keep codes go to codes_dev + dst_off_dev[b]; `span` = largest offset + T.
 * codec_bpe.audio_to_codes fills its --batch_size windows from one file at a time (encode_audio_gpu_1.sh:2-8 over corpora of
 * short utterances); with this call a pass holds windows of many files, including every file's short warm-up windows
 * (the rolling context is still shorter than context_secs, audio_tokenizer.py:72-74), grouped by length. */
int rca_codec_encode_rows_dev(rca_codec_t* h, const float* audio_dev, const int64_t* src_off_dev, int32_t B, int32_t T, int32_t n_keep,
                              int64_t* codes_dev, const int64_t* dst_off_dev, int64_t span, void* stream);

/* Streaming tail of the encoder.  AudioTokenizer.tokenize_audio re-encodes the whole rolling
 * window for every chunk and keeps only the last int(secs*framerate) codes (audio_tokenizer.py:72-74,98-101).
 * This returns exactly those codes -- bit-identical to the last n_keep columns of rca_codec_encode_dev(pcm, B, T) --
 * but runs the encoder only over the frames whose receptive field reaches them (n_keep + enc_left_frames whole
 * frames; the frame grid and the right edge of the window are unchanged).  Valid because this build's codec is a
 * finite-receptive-field conv stack; rca_codec_receptive_field reports the margins derived from the geometry.
 * pcm [B,T] f32 (row stride T) -> codes [B,n_keep] int64. */
int rca_codec_encode_tail_dev(rca_codec_t* h, const float* pcm_dev, int32_t B, int32_t T, int32_t n_keep,
                              int64_t* codes_dev, void* stream);
/* Streaming tail of the decoder: the last n_samples of rca_codec_decode_dev(codes, B, F), bit-identical
 * (AudioTokenizer.detokenize_audio keeps int(secs*sr)+preroll samples of a 100-code window,
 * audio_tokenizer.py:113,141-145).  codes [B,F] int64 (row stride F) -> pcm [B,n_samples] f32. */
int rca_codec_decode_tail_dev(rca_codec_t* h, const int64_t* codes_dev, int32_t B, int32_t F, int32_t n_samples,
                              float* pcm_dev, void* stream);
/* The same two calls with host buffers, as the streaming tokenizer makes them once per frame: H2D of the window,
 * the tail kernels, D2H of the result, one synchronisation.  Shapes repeat frame after frame, so from the second
 * call of a shape on the whole sequence replays as one hipGraph over pinned staging buffers
 * (rca_codec_set_stream_graphs(h, 0) switches the replay off).  Error behaviour as rca_codec_encode / _decode. */
int rca_codec_encode_tail(rca_codec_t* h, const float* pcm_host, int32_t B, int32_t T, int32_t n_keep, int64_t* codes_host);
int rca_codec_decode_tail(rca_codec_t* h, const int64_t* codes_host, int32_t B, int32_t F, int32_t n_samples, float* pcm_host);
int rca_codec_set_stream_graphs(rca_codec_t* h, int32_t enable);
/* Opt-in arithmetic of the encoder's MFMA conv layers (never the default; the default, 0, is the f32 matrix instruction whose
 * results equal the oracle's fma chains bit for bit).  3: bf16 matrix instruction on operands split into bf16 hi + lo (three
 * products per step, ~2^-16 relative); 1: operands rounded to bf16, one product -- the arithmetic class of the reference's own GPU
 * path, bf16 autocast (audio_tokenizer.py:24,78-82).  Neither is bit-exact: ids can differ from mode 0 near ties (bench.py reports
 * the measured fraction).  Decoder and streaming-tail kernels are not affected. */
int rca_codec_set_mfma_mode(rca_codec_t* h, int32_t mode);
/* whole frames a kept code / a kept sample can see to its left */
int rca_codec_receptive_field(const rca_codec_t* h, int32_t* enc_left_frames, int32_t* dec_left_frames);
/* Batch windows (rca_codec_encode_windows_dev / _chunk_range_dev): when enabled, each window is cut down to the
 * frames its kept codes can see -- same codes for any ctx_samples >= the receptive field, ~10x less work at
 * 2.0 s context.  Off by default: the default path computes every window in full, as the reference does. */
int rca_codec_set_window_trim(rca_codec_t* h, int32_t enable);

/* AudioTokenizer._magicodec_decode (audio_tokenizer.py:196-201):
 * embedding(codes, codebook_proj(codebook.weight)) -> decoder -> f32 PCM.
 * codes [B,F] int64 -> pcm [B,F*hop] f32. */
int rca_codec_decode(rca_codec_t* h, const int64_t* codes_host, int32_t B, int32_t F, float* pcm_host);
int rca_codec_decode_dev(rca_codec_t* h, const int64_t* codes_dev, int32_t B, int32_t F, float* pcm_dev,
                         void* stream);

/* The three sub-steps of the model object, for callers that drive them separately the way
 * the reference does (audio_tokenizer.py:190-192,198-200):
 *   codec_model.encoder(pad_audio(x))      pcm [B,T] -> z_e [B,F,D] f32
 *   codec_model.quantizer.inference(z_e)   z_e [R,D] -> idx [R] int64 (R = B*F rows)
 *   codec_model.decoder(z_q)               z_q [B,F,cd] -> pcm [B,F*hop] f32 */
int rca_codec_encoder_dev(rca_codec_t* h, const float* pcm_dev, int32_t B, int32_t T, float* ze_dev, void* stream);
int rca_codec_quantize_dev(rca_codec_t* h, const float* ze_dev, int64_t rows, int64_t* codes_dev, void* stream);
int rca_codec_decoder_dev(rca_codec_t* h, const float* zq_dev, int32_t B, int32_t F, float* pcm_dev, void* stream);
/* device pointer of the projected codebook [codebook_size, codebook_dim] f32 (owned by the handle) */
int rca_codec_codebook_dev(rca_codec_t* h, const float** out_dev);

/* AudioTokenizer.get_codec_embeddings (audio_tokenizer.py:151-159): projected
 * codebook [codebook_size, codebook_dim] f32, copied to the host. */
int rca_codec_codebook(rca_codec_t* h, float* out_host);

/* Debug/parity taps (tests only): run the encoder and copy the activation after
 * layer `layer` (0=conv_in, 1..n=down, n+1=conv_out, n+2=in_proj z) to the host. */
int rca_codec_encode_tap(rca_codec_t* h, const float* pcm_host, int32_t B, int32_t T, int32_t layer,
                         float* out_host, int64_t out_numel);

/* Kernel-variant switch (parity tests compare variants): 0 = scalar-chain kernels,
 * 1 = MFMA/LDS kernels (default when available). */
int rca_codec_set_variant(rca_codec_t* h, int32_t variant);

/* Per-kernel HIP-event timing for bench.py's roofline object: when enabled, every launch of a
 * profiled kernel class is bracketed by hipEventRecord on the stream it is launched on.
 * classes: 0 = implicit-GEMM conv (MFMA), 1 = codebook search, 2 = conv_in, 3 = everything else.
 * _read synchronises, returns the summed device time, the launch count and the summed ALGORITHMIC
 * FLOPs (2 per multiply-add over the real, unpadded K) and bytes since the last _read. */
int rca_codec_profile(rca_codec_t* h, int32_t enable);
int rca_codec_profile_read(rca_codec_t* h, int32_t kclass, double* total_ms, int64_t* launches, double* flops,
                           double* bytes);

/* --------------------------------------------------------------------- LM --
 * Llama-architecture decoder with a llama_cpp.Llama-like control surface
 * (SURVEY.md 8b-3): the deployed model is codec_llama.py after
 * persist_codec_embeddings (codec_llama.py:178-206), i.e. a vanilla Llama with
 * an untied lm_head.  Tensor names follow the HF state dict:
 *   model.embed_tokens.weight [V,H]; model.layers.{i}.self_attn.{q,k,v,o}_proj.weight;
 *   model.layers.{i}.mlp.{gate,up,down}_proj.weight; model.layers.{i}.input_layernorm.weight;
 *   model.layers.{i}.post_attention_layernorm.weight; model.norm.weight; lm_head.weight [V,H]
 */
typedef struct {
    int32_t vocab_size;
    int32_t hidden;
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads;
    int32_t head_dim;
    int32_t ffn;
    int32_t n_ctx;            /* KV slots allocated (llm_n_ctx, realtime_agent_resources.py:13) */
    float rms_eps;
    float rope_theta;
    int32_t rope_scaling;     /* 0 = none, 1 = llama3 */
    float rope_factor;        /* 32 */
    float rope_low_freq_factor;   /* 1 */
    float rope_high_freq_factor;  /* 4 */
    int32_t rope_orig_ctx;    /* 8192 */
    int32_t logits_all;       /* keep logits of every evaluated position (aux_llm) */
    int32_t decode_weights;   /* Every projection matrix and lm_head is kept ONCE, in the format it is streamed in by the decode step and
                                 de-quantised from by the prefill tiles.  0 = as supplied: RCA_BF16 / RCA_F16 / RCA_Q8_0 tensors keep their
                                 format (RCA_F32 is rounded to bf16); 1 = quantise to q8_0 at load the way llama-quantize writes the Q8_0 file
                                 the reference deploys (prep_test_model.sh:29; 8.5 bits per weight streamed); 2 = convert bf16 to fp16 (what
                                 convert_hf_to_gguf.py --outtype f16 writes, prep_test_model.sh:28); 3 = quantise to GGUF Q4_K blocks (4.6 bits per weight streamed;
                                 this build's own min / max rule picks the scales, the format and its de-quantisation are llama.cpp's).  Tensors that arrive quantised stay as
                                 they are.  The embedding table is gathered, not streamed, and keeps full precision: f32 rows for RCA_F32 /
                                 RCA_F16 / RCA_Q8_0 / RCA_Q4_K sources, bf16 rows for RCA_BF16. */
} rca_lm_config_t;

typedef struct {
    int32_t top_k;   /* llama.cpp semantics (llamacpp_utils.py:39-77 passes it straight through): 1..256 = that many ranked candidates on the
                        serial chain (float sums, inverse-CDF draw); <= 0 or >= vocabulary = the whole vocabulary; 257..vocabulary - 1 = a rank
                        cut found by a radix select, then the whole-vocabulary draw (Gumbel-max).  Never clamped.  temp <= 0 is greedy */
    float top_p;     /* 1.0 = off.  With top_k outside 1..256 the cut is a mass threshold over the sorted vocabulary in 2^-40 fixed point */
    float min_p;     /* 0.0 = off */
    float temp;      /* <=0: greedy */
    uint32_t seed;
    int32_t n_bias;          /* logit bias entries (llamacpp_utils.py:8-24) */
    const int32_t* bias_ids;
    const float* bias_vals;
    /* llama.cpp's penalties sampler, in front of top_k (realtime_agent_v2.py:172-185 forwards repeat_penalty / presence_penalty /
       frequency_penalty of realtime_agent_config.py:18-20): over the last `penalty_last_n` tokens this sampler accepted.
       repeat_penalty 1.0 (or 0 = unset) with 0 / 0 = off */
    float repeat_penalty;
    float freq_penalty;
    float presence_penalty;
    int32_t penalty_last_n;  /* 0 = llama-cpp-python's default window (last_n_tokens_size = 64), 1..64, < 0 = no window */
} rca_sampler_params_t;

/* llama_cpp.Llama(model_path=..., n_ctx=..., n_gpu_layers=-1) (realtime_agent_resources.py:19-33) */
int rca_lm_create(const rca_lm_config_t* cfg, const rca_tensor_t* tensors, int32_t n_tensors,
                  int32_t device, rca_lm_t** out);
/* random-init weights generated on the device from a counter hash (bench configs 3/4:
 * there are no checkpoints offline).  oracle/lm_ref.py regenerates the same values. */
int rca_lm_create_random(const rca_lm_config_t* cfg, uint64_t seed, float init_std, int32_t device,
                         rca_lm_t** out);
/* A second instance over the same weights: what the reference obtains by loading one GGUF twice (`llm` and its logits_all twin
 * `aux_llm`, realtime_agent_resources.py:19-33).  Own KV cache / workspace / sampler / stream; n_ctx may not exceed the
 * parent's.  Either handle may be destroyed first; weight-modifying calls act on both. */
int rca_lm_create_shared(rca_lm_t* parent, int32_t n_ctx, int32_t logits_all, rca_lm_t** out);
int rca_lm_destroy(rca_lm_t* h);

/* Llama.reset(): n_tokens = 0 (realtime_agent_v2.py:68) */
int rca_lm_reset(rca_lm_t* h);
/* Llama.eval(tokens): append n ids at position n_tokens, run the forward, keep the
 * last position's logits (every position's when logits_all) (llamacpp_utils.py:150) */
int rca_lm_eval(rca_lm_t* h, const int32_t* ids, int32_t n);
/* Llama.eval that returns once its LAST pass (a 128-token prefill tile, or a 1-2 token decode pass) is enqueued instead of
 * finished; any later call on the handle waits for it first.  The pieces are evaluated with the arithmetic of ONE long eval
 * (prefill tiles even for a handful of tokens), so a cache built piecewise equals the cache of a single rca_lm_eval.  Used to build the post-trim KV cache ahead of the sliding-window
 * trim (realtime_agent_v2.py:187-190,725-733) on a weight-sharing twin while the live handle keeps stepping. */
int rca_lm_eval_async(rca_lm_t* h, const int32_t* ids, int32_t n);
/* The reference recomputes the KV cache of the surviving context inside the frame that trims (realtime_agent_v2.py:725-733:
 * n_tokens = header, eval(suffix)).  Here a twin handle (rca_lm_create_shared, same n_ctx) can be given the header's KV
 * (rca_lm_copy_kv: positions [0, n_pos) of every layer, device to device), be fed the suffix over the preceding frames, and
 * trade caches with the live handle at the trim (rca_lm_swap_kv: O(1), both streams are drained first; captured step graphs
 * are kept per cache).  n_tokens is not exchanged: the caller sets it, as the reference does. */
int rca_lm_copy_kv(rca_lm_t* dst, rca_lm_t* src, int32_t n_pos);
int rca_lm_swap_kv(rca_lm_t* a, rca_lm_t* b);
/* run the handle's stream at the device's lowest (1) / highest (0) stream priority: background prefill next to a live session */
int rca_lm_set_low_priority(rca_lm_t* h, int32_t enable);
/* read / write Llama.n_tokens: the agent rolls the KV cache back by writing it
 * (realtime_agent_v2.py:208,219,261,465,730); stale slots are overwritten by the next eval */
int rca_lm_get_n_tokens(const rca_lm_t* h, int32_t* n);
int rca_lm_set_n_tokens(rca_lm_t* h, int32_t n);
/* llm._ctx.get_logits(): V floats of the last evaluated position (realtime_agent_v2.py:449,461) */
int rca_lm_get_logits(rca_lm_t* h, float* out_host);
/* llm._scores[row] for logits_all handles: row counts back from the last eval'd position */
int rca_lm_get_logits_row(rca_lm_t* h, int32_t pos, float* out_host);
/* device pointer of the last logits (V floats); valid until the next eval */
int rca_lm_logits_dev(rca_lm_t* h, const float** out_dev);

/* init_sampler_for_generate (llamacpp_utils.py:39-77) */
int rca_lm_sampler_init(rca_lm_t* h, const rca_sampler_params_t* p);
/* sample() (llamacpp_utils.py:79-95): draw from the last logits */
int rca_lm_sample(rca_lm_t* h, int32_t* token);
/* next(generate(tokens, reset=False)) (llamacpp_utils.py:145-161; realtime_agent_v2.py:355):
 * eval + sample with no host round trip in between; hipGraph-replayed for n<=2 */
int rca_lm_step(rca_lm_t* h, const int32_t* ids, int32_t n, int32_t* token);
/* One whole frame of process_audio_input_ids (realtime_agent_v2.py:332-372) as ONE hipGraph: n_steps (<= 8) S=2 steps, the
 * agent token sampled by step i fed back on the device together with user_ids[i] as step i+1's input pair; first_pair is the
 * pair step 0 evaluates (the last two ids of the sequence).  out_tokens[i] = token sampled by step i.  *n_done = number of steps
 * whose result stands: n_steps, or j + 1 when step j sampled a token <= audio_id_floor (the loop leaves audio mode there,
 * realtime_agent_v2.py:361-371); the KV position and the sampler's draw counter are then exactly what j + 1 single steps would
 * have left, and the caller continues step by step.  After a complete frame the logits are those of its last step; after a frame
 * cut short (n_done < n_steps) NO logits are available (the buffer holds a rolled-back step's): get_logits / token_probs / sample
 * fail with "no logits" until the next eval or step. */
int rca_lm_frame(rca_lm_t* h, const int32_t* first_pair, const int32_t* user_ids, int32_t n_steps, int32_t audio_id_floor,
                 int32_t* out_tokens, int32_t* n_done);
/* ONE duplex frame-step as ONE graph replay (north_star: "the per-frame encode -> LM-step -> decode loop is hipGraph-captured";
 * the loop is RealtimeAgent.process_audio, realtime_agent_v2.py:504-554): encode tail of the user's rolling PCM window
 * (audio_tokenizer.py:67-103) -> code -> token id (code_token_base + code: the codec tokens sit in the vocabulary in code order,
 * train_vanilla_latest.py:587-589) -> the chunk's n_steps LM steps exactly as rca_lm_frame runs them (process_audio_input_ids,
 * :332-372) -> token id -> code appended to the detokenizer's rolling code context (audio_tokenizer.py:113) -> decode tail
 * (:141-145) -> softmax(last logits)[probe_id] (measure_event_prob, :448-452).  One upload, one replay, one synchronisation.
 * The host owns both rolling windows and passes them whole; nothing rolls on the device, so a frame that takes the separate
 * calls instead (warm-up shapes, forced transcription / response, a trim between two steps) needs no repair.
 * The first frame of a shape runs the same launches eagerly (it sizes the codec workspace); the second captures. */
typedef struct rca_duplex_frame_args {
    const float* pcm_window;    /* host [T]: mono PCM window INCLUDING this frame's chunk */
    const int64_t* code_ctx;    /* host [F_ctx]: the code context BEFORE this frame's codes (may be NULL when F_ctx = 0) */
    int32_t T;
    int32_t F_ctx;
    int32_t n_steps;            /* codes per frame == LM steps (1..8) */
    int32_t n_samples;          /* PCM samples wanted from the end of decode(code_ctx + this frame's codes) */
    int32_t code_token_base;    /* token id of code 0 */
    int32_t audio_id_floor;     /* a sampled token <= this leaves audio mode (<|end_header|>) */
    int32_t probe_id;           /* token whose probability under the last step's logits is wanted, or -1 */
    int32_t first_pair[2];      /* [agent, user] pair of the previous frame (the first step's input) */
} rca_duplex_frame_args_t;
typedef struct rca_duplex_frame_out {
    int64_t user_codes[8];      /* the n_steps codes of the user's chunk (always valid) */
    int32_t tokens[8];          /* sampled agent tokens; entries >= n_done are -1 */
    int32_t n_done;             /* as rca_lm_frame: < n_steps when step n_done - 1 left audio mode (KV position / draws put back) */
    int32_t flags;              /* 0: pcm_out_host holds the decode tail.  bit 0: a sampled token is no codec token, bit 1: frame
                                   cut short -- in both cases pcm_out_host is untouched and the caller decodes on its own path */
    float probe_prob;           /* -1 when not asked for or the frame was cut short */
} rca_duplex_frame_out_t;
int rca_duplex_frame(rca_lm_t* lm, rca_codec_t* codec, const rca_duplex_frame_args_t* args, rca_duplex_frame_out_t* out,
                     float* pcm_out_host);
/* Optional: make rca_duplex_frame's one-time allocations for a call shape (pinned staging, device buffers, side stream) ahead of the
 * first frame -- a session calls it at reset() (realtime_agent_v2.py:127-161) so that no frame pays for a pinned allocation. */
int rca_duplex_prepare(rca_lm_t* lm, int32_t T, int32_t F_ctx, int32_t n_steps, int32_t n_samples);
/* Optional, after rca_duplex_prepare and rca_lm_sampler_init: capture every graph the session's frames can replay BEFORE the first
 * frame -- the one-replay frame of the call shape in `args` (T, F_ctx, n_steps, n_samples, code_token_base, probe_id >= 0 or not; the
 * data pointers and first_pair are not read) for every context bucket n_ctx reaches, on the KV cache of `lm` and, when the session
 * trims through a shadow cache, on `twin`'s too (may be NULL), and with n_probe > 0 the speculative one-token step + n_probe
 * probabilities (rca_lm_step_probe; realtime_agent_v2.py:455-466).  A session calls it at reset() (realtime_agent_v2.py:127-161): no
 * frame then pays a graph capture. */
int rca_duplex_precapture(rca_lm_t* lm, rca_lm_t* twin, rca_codec_t* codec, const rca_duplex_frame_args_t* args, int32_t n_probe);
/* what rca_duplex_frame needs from a codec handle whose tail calls it captures: a signature of every address a captured tail
 * call bakes in, a hand-over of the handle's stream ordering to the capturing stream, the codebook size */
int rca_codec_workspace_sig(rca_codec_t* h, uint64_t* sig);
int rca_codec_stream_handoff(rca_codec_t* h, void* stream);
int rca_codec_codebook_size(const rca_codec_t* h, int32_t* n);
/* rca_lm_step + rca_lm_token_probs of the position it evaluated, as ONE replay and one synchronisation: the agent's speculative
 * <|end_audio|> step (get_probable_event_speaker, realtime_agent_v2.py:455-466: eval, sample, softmax(logits)[agent speaker, user
 * speaker], then n_tokens -= 1).  Same token and the same probabilities as the two separate calls. */
int rca_lm_step_probe(rca_lm_t* h, const int32_t* ids, int32_t n, const int32_t* probe_ids, int32_t n_probe, int32_t* token, float* probs_out);
/* softmax(logits)[token] of the last position, reduced on the device
 * (measure_event_prob, realtime_agent_v2.py:448-452) */
int rca_lm_token_probs(rca_lm_t* h, const int32_t* token_ids, int32_t n, float* probs_out);
/* zero lm_head rows [row_begin, row_end) (random-init bench models: keeps the sampler on codec tokens,
 * as a trained model in audio mode does; the bytes streamed per step are unchanged) */
int rca_lm_mask_head_rows(rca_lm_t* h, int32_t row_begin, int32_t row_end);
/* The reference's deployment step CodecLlamaForCausalLM.persist_codec_embeddings (codec_llama.py:178-206) for a checkpoint
 * that still carries the frozen codec embedding and its projector (codec_llama.py:32-69): table row
 * codec_vocab_start + i  <-  linear_2(gelu(linear_1(codec_embed[i]))), fp32 arithmetic, stored as bf16 (nearest even).
 * Host pointers: codec_embed [n_codes, dim], w1 [hidden, dim], b1 [hidden], w2 [hidden, hidden], b2 [hidden];
 * out_f32 (optional, [n_codes, hidden]) receives the rows before the 16-bit rounding.  One call per codebook. */
int rca_lm_persist_codec_embeddings(rca_lm_t* h, const float* codec_embed, int32_t n_codes, int32_t dim,
                                    const float* w1, const float* b1, const float* w2, const float* b2,
                                    int32_t codec_vocab_start, float* out_f32);
/* llama_cpp's logits_all after creation: 1 = keep the logits of every evaluated position (rca_lm_get_logits_row), 0 = last
 * position only.  get_logprobs (llamacpp_utils.py:30-37) evaluates its long context with 0 and the scored tokens with 1. */
int rca_lm_set_logits_all(rca_lm_t* h, int32_t enable);
/* enable / disable hipGraph replay of the steady-state step (eager launches otherwise); tests and
 * bench compare the two */
int rca_lm_set_graphs(rca_lm_t* h, int32_t enable);
/* evals longer than 8 tokens (session prefill llamacpp_utils.py:145-161 via realtime_agent_v2.py:100, KV recompute
 * :725-733) run as 128-token tiles on bf16 MFMA with hi/lo-split activations (default; logits within ~1e-3 of the
 * decode path); 0 routes them through the 8-token GEMV chunks, which are bit-identical to decode */
int rca_lm_set_mfma_prefill(rca_lm_t* h, int32_t enable);
/* the format the projection matrices are kept and streamed in (0 bf16, 1 q8_0, 2 f16, 3 q4_k) and, optionally, the weight bytes one decode
 * step reads (llama.cpp prints the same two facts at load: file type and model size) */
int rca_lm_weight_format(const rca_lm_t* h, int32_t* fmt, int64_t* bytes_per_step);
/* decode steps merge the attention splits inside the attention launch (1, default: the workgroup that publishes its partial last
 * merges them; bit-identical to the separate merge launch) or in a launch of its own (0); tests compare the two */
int rca_lm_set_attn_fuse(rca_lm_t* h, int32_t enable);
/* synchronise the handle's stream (timing) */
int rca_lm_sync(rca_lm_t* h);
int rca_codec_sync(rca_codec_t* h);

#ifdef __cplusplus
}
#endif
#endif /* RCA_H */
