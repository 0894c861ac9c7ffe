/*
 * codec_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the
 * codec arithmetic behind AudioTokenizer._magicodec_encode / _magicodec_decode
 * (reference: realtime_codec_agent/audio_tokenizer.py:189-201).
 *
 * PARITY UNPINNED for the codec arithmetic itself: the reference delegates it to
 * the third-party MagiCodec package (github Ereboas/MagiCodec, unpinned HEAD,
 * magicodec_build.sh:2; loaded through codec-bpe[magicodec], requirements.txt:2),
 * which is absent from /root/reference and from this image, and the reference
 * holds no golden vectors for it.  The layer stack restated here
 * is therefore this build's own "MagiCodec-style" definition (strided conv1d
 * encoder, single 131072x16 codebook nearest-neighbour search, transposed-conv
 * decoder, hop 320).  What IS pinned by the reference -- the call order
 * pad_audio -> encoder -> quantizer.inference (audio_tokenizer.py:190-192) and
 * embedding(codes, codebook_proj(codebook.weight)) -> decoder (:198-200) -- is
 * followed exactly.
 *
 * Every multiply-accumulate is an explicit fmaf() in a fixed order, so the HIP
 * kernels (f32 VALU fma chains and f32 MFMA, both k-ordered fma chains) can be
 * compared bit for bit.  Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RCA_MAX_STAGES 8

typedef struct {
    int32_t sample_rate;
    int32_t n_stages;
    int32_t strides[RCA_MAX_STAGES];
    int32_t channels[RCA_MAX_STAGES + 1];
    int32_t k_in;
    int32_t k_latent;
    int32_t latent_dim;
    int32_t codebook_size;
    int32_t codebook_raw_dim;
    int32_t codebook_dim;
    float leaky_slope;
} oracle_codec_config_t;

static inline float lrelu(float x, float slope) { return x >= 0.0f ? x : x * slope; }

/* y[b][co][t] = bias[co] + sum_{ci asc} sum_{kk asc} w[co][ci][kk] * pre(x[b][ci][t*s + kk - padL])
 * with padL = (k - s + 1) / 2, zero outside [0, Lin); Lout = ceil-free: (Lin + s - 1) / s when Lin % s == 0
 * callers guarantee Lin % s == 0 so Lout = Lin / s.  pre() = LeakyReLU when pre_act. */
void oracle_conv1d(const float* x, int B, int Cin, int Lin, const float* w, const float* bias, int Cout,
                   int k, int s, int pre_act, float slope, float* y) {
    const int Lout = Lin / s;
    const int padL = (k - s + 1) / 2;
    float* xa = (float*)malloc((size_t)B * Cin * Lin * sizeof(float));
    const size_t nx = (size_t)B * Cin * Lin;
    if (pre_act) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)nx; ++i) xa[i] = lrelu(x[i], slope);
    } else {
        memcpy(xa, x, nx * sizeof(float));
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            float* acc = y + ((size_t)b * Cout + co) * Lout;
            for (int t = 0; t < Lout; ++t) acc[t] = bias[co];
            for (int ci = 0; ci < Cin; ++ci) {
                const float* xr = xa + ((size_t)b * Cin + ci) * Lin;
                const float* wr = w + ((size_t)co * Cin + ci) * k;
                for (int kk = 0; kk < k; ++kk) {
                    const float wv = wr[kk];
                    const int off = kk - padL;
                    /* valid t: 0 <= t*s + off < Lin */
                    int t_lo = off >= 0 ? 0 : (-off + s - 1) / s;
                    int t_hi = (Lin - 1 - off) / s; /* off <= Lin-1 always for our shapes */
                    if (Lin - 1 - off < 0) continue;
                    if (t_hi > Lout - 1) t_hi = Lout - 1;
                    const float* xs = xr + off;
#pragma omp simd
                    for (int t = t_lo; t <= t_hi; ++t) acc[t] = fmaf(wv, xs[(size_t)t * s], acc[t]);
                }
            }
        }
    }
    free(xa);
}

/* ConvTranspose1d, weight layout [Cin][Cout][k] (k = 2s), padL = (k - s + 1)/2, Lout = Lin * s:
 * y[b][co][u] = bias[co] + sum_{ci asc} sum_{kk asc, (u + padL - kk) % s == 0, 0 <= t < Lin}
 *                 w[ci][co][kk] * pre(x[b][ci][t]),  t = (u + padL - kk) / s */
void oracle_convtr1d(const float* x, int B, int Cin, int Lin, const float* w, const float* bias, int Cout,
                     int k, int s, int pre_act, float slope, float* y) {
    const int Lout = Lin * s;
    const int padL = (k - s + 1) / 2;
    float* xa = (float*)malloc((size_t)B * Cin * Lin * sizeof(float));
    const size_t nx = (size_t)B * Cin * Lin;
    for (size_t i = 0; i < nx; ++i) xa[i] = pre_act ? lrelu(x[i], slope) : x[i];
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            float* acc = y + ((size_t)b * Cout + co) * Lout;
            for (int u = 0; u < Lout; ++u) acc[u] = bias[co];
            for (int ci = 0; ci < Cin; ++ci) {
                const float* xr = xa + ((size_t)b * Cin + ci) * Lin;
                const float* wr = w + ((size_t)ci * Cout + co) * k;
                for (int u = 0; u < Lout; ++u) {
                    const int kk0 = (u + padL) % s;
                    float a = acc[u];
                    for (int kk = kk0; kk < k; kk += s) {
                        const int t = (u + padL - kk) / s;
                        if (t >= 0 && t < Lin) a = fmaf(wr[kk], xr[t], a);
                    }
                    acc[u] = a;
                }
            }
        }
    }
    free(xa);
}

/* out[r][j] = b[j] + sum_{d asc} w[j][d] * in[r][d]   (rows r, in-dim D, out-dim J) */
void oracle_linear(const float* in, long R, int D, const float* w, const float* b, int J, float* out) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        for (int j = 0; j < J; ++j) {
            float a = b[j];
            for (int d = 0; d < D; ++d) a = fmaf(w[(size_t)j * D + d], in[(size_t)r * D + d], a);
            out[(size_t)r * J + j] = a;
        }
    }
}

/* projected codebook cb[c][j] (codebook_proj(codebook.weight), audio_tokenizer.py:158,198)
 * and hc[c] = -0.5 * sum_j cb[c][j]^2 (chain from 0 in j order, then one multiply) */
void oracle_codebook(const oracle_codec_config_t* cfg, const float* raw, const float* pw, const float* pb,
                     float* cb, float* hc) {
    const int N = cfg->codebook_size, R = cfg->codebook_raw_dim, J = cfg->codebook_dim;
    oracle_linear(raw, N, R, pw, pb, J, cb);
    for (int c = 0; c < N; ++c) {
        float a = 0.0f;
        for (int j = 0; j < J; ++j) a = fmaf(cb[(size_t)c * J + j], cb[(size_t)c * J + j], a);
        hc[c] = -0.5f * a;
    }
}

/* nearest neighbour: score[c] = hc[c] + sum_{j asc} z[j]*cb[c][j]  (fma chain from hc[c]);
 * code = first c with the maximum score (== argmin ||z - cb[c]||^2 up to rounding). */
void oracle_vq_argmax(const float* z, long F, const float* cb, const float* hc, int N, int J, int64_t* codes) {
    /* transposed codebook so the inner loop vectorises over c; the per-c chain order is unchanged */
    float* cbT = (float*)malloc((size_t)N * J * sizeof(float));
    for (int c = 0; c < N; ++c)
        for (int j = 0; j < J; ++j) cbT[(size_t)j * N + c] = cb[(size_t)c * J + j];
#pragma omp parallel
    {
        float* sc = (float*)malloc((size_t)N * sizeof(float));
#pragma omp for schedule(static)
        for (long f = 0; f < F; ++f) {
            const float* zf = z + (size_t)f * J;
            for (int c = 0; c < N; ++c) sc[c] = hc[c];
            for (int j = 0; j < J; ++j) {
                const float zj = zf[j];
                const float* row = cbT + (size_t)j * N;
#pragma omp simd
                for (int c = 0; c < N; ++c) sc[c] = fmaf(zj, row[c], sc[c]);
            }
            float best = sc[0];
            int bi = 0;
            for (int c = 1; c < N; ++c)
                if (sc[c] > best) { best = sc[c]; bi = c; }
            codes[f] = bi;
        }
        free(sc);
    }
    free(cbT);
}

typedef struct {
    const float *enc_in_w, *enc_in_b;
    const float *enc_down_w[RCA_MAX_STAGES], *enc_down_b[RCA_MAX_STAGES];
    const float *enc_out_w, *enc_out_b;
    const float *q_in_w, *q_in_b;
    const float *q_codebook;
    const float *q_proj_w, *q_proj_b;
    const float *dec_in_w, *dec_in_b;
    const float *dec_up_w[RCA_MAX_STAGES], *dec_up_b[RCA_MAX_STAGES];
    const float *dec_out_w, *dec_out_b;
} oracle_codec_weights_t;

static int hop_of(const oracle_codec_config_t* c) {
    int h = 1;
    for (int i = 0; i < c->n_stages; ++i) h *= c->strides[i];
    return h;
}

/* AudioTokenizer._magicodec_encode (audio_tokenizer.py:189-194).
 * pcm [B][T] -> codes [B][F], F = ceil(T/hop).  tap_layer >= 0 also copies that
 * layer's activation into tap (0 conv_in, 1..n down, n+1 conv_out [B][D][F], n+2 z [B*F][cd]). */
int oracle_codec_encode(const oracle_codec_config_t* cfg, const oracle_codec_weights_t* W, const float* cb,
                        const float* hc, const float* pcm, int B, int T, int64_t* codes, int tap_layer,
                        float* tap) {
    const int hop = hop_of(cfg);
    const int F = (T + hop - 1) / hop;
    const int Tp = F * hop;
    const int n = cfg->n_stages;
    /* pad_audio: right-pad with zeros to a hop multiple (audio_tokenizer.py:190) */
    float* cur = (float*)calloc((size_t)B * Tp, sizeof(float));
    for (int b = 0; b < B; ++b) memcpy(cur + (size_t)b * Tp, pcm + (size_t)b * T, (size_t)T * sizeof(float));
    int C = 1, L = Tp;
    /* conv_in (no pre-activation on raw PCM) */
    {
        const int Co = cfg->channels[0];
        float* nxt = (float*)malloc((size_t)B * Co * L * sizeof(float));
        oracle_conv1d(cur, B, C, L, W->enc_in_w, W->enc_in_b, Co, cfg->k_in, 1, 0, cfg->leaky_slope, nxt);
        free(cur); cur = nxt; C = Co;
        if (tap_layer == 0) memcpy(tap, cur, (size_t)B * C * L * sizeof(float));
    }
    for (int i = 0; i < n; ++i) {
        const int s = cfg->strides[i], Co = cfg->channels[i + 1];
        float* nxt = (float*)malloc((size_t)B * Co * (L / s) * sizeof(float));
        oracle_conv1d(cur, B, C, L, W->enc_down_w[i], W->enc_down_b[i], Co, 2 * s, s, 1, cfg->leaky_slope, nxt);
        free(cur); cur = nxt; C = Co; L /= s;
        if (tap_layer == i + 1) memcpy(tap, cur, (size_t)B * C * L * sizeof(float));
    }
    {
        const int Co = cfg->latent_dim;
        float* nxt = (float*)malloc((size_t)B * Co * L * sizeof(float));
        oracle_conv1d(cur, B, C, L, W->enc_out_w, W->enc_out_b, Co, cfg->k_latent, 1, 1, cfg->leaky_slope, nxt);
        free(cur); cur = nxt; C = Co;
        if (tap_layer == n + 1) memcpy(tap, cur, (size_t)B * C * L * sizeof(float));
    }
    /* z_e [B][D][F] -> rows [B*F][D] -> in_proj -> z [B*F][cd] */
    const int D = cfg->latent_dim, J = cfg->codebook_dim;
    float* ze = (float*)malloc((size_t)B * F * D * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int f = 0; f < F; ++f) ze[((size_t)b * F + f) * D + d] = cur[((size_t)b * D + d) * F + f];
    float* z = (float*)malloc((size_t)B * F * J * sizeof(float));
    oracle_linear(ze, (long)B * F, D, W->q_in_w, W->q_in_b, J, z);
    if (tap_layer == n + 2) memcpy(tap, z, (size_t)B * F * J * sizeof(float));
    oracle_vq_argmax(z, (long)B * F, cb, hc, cfg->codebook_size, J, codes);
    free(ze); free(z); free(cur);
    return F;
}

/* codec_model.decoder(z_q) (audio_tokenizer.py:199-200): z_q [B][F][cd] -> pcm [B][F*hop], clamped to [-1, 1] */
int oracle_codec_decoder(const oracle_codec_config_t* cfg, const oracle_codec_weights_t* W, const float* zq, int B,
                         int F, float* pcm) {
    const int n = cfg->n_stages, J = cfg->codebook_dim;
    /* [B][F][cd] -> [B][cd][F] */
    float* cur = (float*)malloc((size_t)B * J * F * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            for (int j = 0; j < J; ++j) cur[((size_t)b * J + j) * F + f] = zq[((size_t)b * F + f) * J + j];
    int C = J, L = F;
    {
        const int Co = cfg->channels[n];
        float* nxt = (float*)malloc((size_t)B * Co * L * sizeof(float));
        oracle_conv1d(cur, B, C, L, W->dec_in_w, W->dec_in_b, Co, cfg->k_latent, 1, 0, cfg->leaky_slope, nxt);
        free(cur); cur = nxt; C = Co;
    }
    for (int i = 0; i < n; ++i) {
        const int s = cfg->strides[n - 1 - i], Co = cfg->channels[n - 1 - i];
        float* nxt = (float*)malloc((size_t)B * Co * L * s * sizeof(float));
        oracle_convtr1d(cur, B, C, L, W->dec_up_w[i], W->dec_up_b[i], Co, 2 * s, s, 1, cfg->leaky_slope, nxt);
        free(cur); cur = nxt; C = Co; L *= s;
    }
    {
        float* nxt = (float*)malloc((size_t)B * L * sizeof(float));
        oracle_conv1d(cur, B, C, L, W->dec_out_w, W->dec_out_b, 1, cfg->k_in, 1, 1, cfg->leaky_slope, nxt);
        free(cur); cur = nxt;
    }
    for (size_t i = 0; i < (size_t)B * L; ++i) {
        float v = cur[i];
        pcm[i] = v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v);
    }
    free(cur);
    return L;
}

/* AudioTokenizer._magicodec_decode (audio_tokenizer.py:196-201): codes [B][F] -> pcm [B][F*hop]:
 * embedding(codes, codebook_proj(codebook.weight)) -> decoder */
int oracle_codec_decode(const oracle_codec_config_t* cfg, const oracle_codec_weights_t* W, const float* cb,
                        const int64_t* codes, int B, int F, float* pcm) {
    const int J = cfg->codebook_dim;
    float* zq = (float*)malloc((size_t)B * F * J * sizeof(float));
    for (size_t i = 0; i < (size_t)B * F; ++i) {
        const int64_t c = codes[i];
        if (c < 0 || c >= cfg->codebook_size) { free(zq); return -1; }
        memcpy(zq + i * J, cb + (size_t)c * J, (size_t)J * sizeof(float));
    }
    const int L = oracle_codec_decoder(cfg, W, zq, B, F, pcm);
    free(zq);
    return L;
}
