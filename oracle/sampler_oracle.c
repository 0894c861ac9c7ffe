/*
 * sampler_oracle.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the device sampler behind
 * LlamaForAlternatingCodeChannels.sample (reference: llamacpp_utils.py:39-95 configures llama.cpp's
 * chain top_k -> top_p -> min_p -> temp -> dist; realtime_agent_config.py:11-20,29 gives top_k=100,
 * top_p=1, min_p=0, temp=1, seed=42).
 *
 * PARITY UNPINNED against llama.cpp's own sampler: llama-cpp-python is absent offline, unpinned
 * (only commit c37132b is mentioned, llamacpp_utils.py:10) and its RNG (std::mt19937 +
 * discrete_distribution) cannot be reproduced from the reference; the reference holds no sampled
 * golden tokens.  This file pins THIS build's sampler definition: exact top-k by (logit desc,
 * index asc), polynomial exp in fma, counter-based splitmix64 RNG, inverse-CDF draw.
 * top_k <= 0 is llama.cpp's "whole vocabulary" (its top-k sampler is then a no-op): with top_p >= 1 the draw from
 * softmax(logits / temp) over every token that passes min_p is made by the Gumbel-max rule -- argmax of
 * (v_i - max) / temp + g_i with g_i = -log(-log(u_i)), u_i from the counter RNG keyed by (seed, draw, token) -- which has
 * exactly that distribution and needs no sort of 259 k candidates; polynomial log in fma so that device and host agree
 * bit for bit.
 *
 * Round 4 -- the members the reference's config exposes and llama.cpp honours (realtime_agent_v2.py:172-185,
 * realtime_agent_config.py:18-20; llama-cpp-python's _init_sampler builds: custom logits processor (the logit bias) ->
 * penalties(last_n = 64, repeat, freq, present) -> top_k -> typical -> top_p -> min_p -> temp -> dist):
 *   * penalties: llama.cpp's llama_sampler_penalties restated -- over the last `n_prev` tokens THIS sampler accepted (sampled),
 *     a token seen `count` times has its logit multiplied (<= 0) or divided (> 0) by repeat_penalty, then
 *     count * frequency_penalty + presence_penalty subtracted; applied after the bias, before everything else.
 *   * top_k > 256 (up to the vocabulary) and top_k <= 0 together with top_p < 1: the "big" path.  Candidates = the top_k largest
 *     (value, lowest index first) keys (all of them for top_k <= 0); top_p keeps the smallest prefix of the sorted candidates whose
 *     mass reaches top_p of the candidates' total, with the masses in 2^-40 fixed point (w_i = trunc(exp(v_i - max) * 2^40), integer
 *     sums: independent of summation order, so the device's histogram passes and this loop agree exactly; need = max(1,
 *     trunc(top_p * W))); min_p as before; the draw by the Gumbel-max rule over what is left.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

float oracle_expf(float x) {
    if (x < -87.0f) return 0.0f;
    const float n = rintf(x * 1.44269504f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.3888889e-3f;
    p = fmaf(p, r, 8.3333333e-3f);
    p = fmaf(p, r, 4.1666668e-2f);
    p = fmaf(p, r, 1.6666667e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(p, (int)n);
}

/* natural log of a positive normal float: cephes' logf polynomial written as explicit fmas */
float oracle_logf(float x) {
    union { float f; uint32_t u; } b;
    b.f = x;
    int e = (int)(b.u >> 23) - 127;
    b.u = (b.u & 0x007FFFFFu) | 0x3F800000u;
    float m = b.f;
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float y = 7.0376836292e-2f;
    y = fmaf(y, f, -1.1514610310e-1f);
    y = fmaf(y, f, 1.1676998740e-1f);
    y = fmaf(y, f, -1.2420140846e-1f);
    y = fmaf(y, f, 1.4249322787e-1f);
    y = fmaf(y, f, -1.6668057665e-1f);
    y = fmaf(y, f, 2.0000714765e-1f);
    y = fmaf(y, f, -2.4999993993e-1f);
    y = fmaf(y, f, 3.3333331174e-1f);
    y = y * f * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

uint64_t oracle_splitmix(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + ctr * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

typedef struct { float v; int idx; } cand_t;
static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* llama.cpp's llama_sampler_penalties_apply on one logit: `count` occurrences in the window of accepted tokens */
static float penalise(float v, int count, float repeat, float freq, float present) {
    if (count <= 0) return v;
    if (v <= 0.0f) v = v * repeat; else v = v / repeat;
    const float t = (float)count * freq + present;
    return v - t;
}

/* returns the sampled token for draw number `counter` of the stream started by `seed`; prev[0..n_prev) = the window of tokens this
   sampler accepted before this draw (at most the last 64), in any order */
int oracle_sample_ex(const float* logits, int V, int top_k, float top_p, float min_p, float temp, uint32_t seed, uint64_t counter,
                     int n_bias, const int32_t* bias_ids, const float* bias_vals,
                     float repeat_penalty, float freq_penalty, float presence_penalty, int n_prev, const int32_t* prev) {
    cand_t* c = (cand_t*)malloc((size_t)V * sizeof(cand_t));
    const int pen = n_prev > 0 && (repeat_penalty != 1.0f || freq_penalty != 0.0f || presence_penalty != 0.0f);
    for (int i = 0; i < V; ++i) {
        float v = logits[i];
        for (int b = 0; b < n_bias; ++b)
            if (bias_ids[b] == i) v = v + bias_vals[b];
        c[i].v = v; c[i].idx = i;
    }
    if (pen) {
        for (int j = 0; j < n_prev; ++j) {
            int first = 1, count = 0;
            for (int q = 0; q < n_prev; ++q) { if (prev[q] == prev[j]) { if (q < j) first = 0; ++count; } }
            if (first && prev[j] >= 0 && prev[j] < V) c[prev[j]].v = penalise(c[prev[j]].v, count, repeat_penalty, freq_penalty, presence_penalty);
        }
    }
    const int greedy = temp <= 0.0f;
    if (!greedy && (top_k <= 0 || top_k > 256) && (top_p < 1.0f || (top_k > 256 && top_k < V))) {
        /* the "big" path: rank cut and / or fixed-point mass cut over the sorted vocabulary, then Gumbel-max */
        qsort(c, (size_t)V, sizeof(cand_t), cmp_cand);
        int cnt = (top_k <= 0 || top_k > V) ? V : top_k;
        const float mx = c[0].v;
        if (top_p < 1.0f) {
            uint64_t W = 0;
            for (int i = 0; i < cnt; ++i) W += (uint64_t)(oracle_expf(c[i].v - mx) * 1099511627776.0f);
            uint64_t need = (uint64_t)((double)top_p * (double)W);
            if (need < 1) need = 1;
            uint64_t cum = 0; int keep = cnt;
            for (int i = 0; i < cnt; ++i) { cum += (uint64_t)(oracle_expf(c[i].v - mx) * 1099511627776.0f); if (cum >= need) { keep = i + 1; break; } }
            cnt = keep;
        }
        const float inv_t = 1.0f / temp;
        const uint64_t draw = oracle_splitmix((uint64_t)seed, counter);
        int best = -1; float best_key = 0.0f;
        for (int i = 0; i < cnt; ++i) {
            const float d = c[i].v - mx;
            if (min_p > 0.0f && !(oracle_expf(d) >= min_p)) continue;
            const uint64_t z = oracle_splitmix(draw, (uint64_t)c[i].idx);
            const float u = (float)(uint32_t)(((z >> 41) << 1) | 1u) * 5.9604644775390625e-08f;
            const float g = -oracle_logf(-oracle_logf(u));
            const float key = fmaf(d, inv_t, g);
            if (best < 0 || key > best_key || (key == best_key && c[i].idx < best)) { best = c[i].idx; best_key = key; }   /* ties: lowest index */
        }
        free(c);
        return best;
    }
    if (!greedy && (top_k <= 0 || top_k > 256)) {   /* whole vocabulary: Gumbel-max over every token that passes min_p */
        float mx = c[0].v;
        for (int i = 1; i < V; ++i) if (c[i].v > mx) mx = c[i].v;
        const float inv_t = 1.0f / temp;
        const uint64_t draw = oracle_splitmix((uint64_t)seed, counter);
        int best = -1; float best_key = 0.0f;
        for (int i = 0; i < V; ++i) {
            const float d = c[i].v - mx;
            if (min_p > 0.0f && !(oracle_expf(d) >= min_p)) continue;
            const uint64_t z = oracle_splitmix(draw, (uint64_t)i);
            const float u = (float)(uint32_t)(((z >> 41) << 1) | 1u) * 5.9604644775390625e-08f;   /* odd / 2^24: strictly inside (0, 1) */
            const float g = -oracle_logf(-oracle_logf(u));
            const float key = fmaf(d, inv_t, g);
            if (best < 0 || key > best_key) { best = i; best_key = key; }   /* ties: lowest index */
        }
        free(c);
        return best;
    }
    qsort(c, (size_t)V, sizeof(cand_t), cmp_cand);
    int k = greedy ? 1 : top_k;
    if (k <= 0 || k > 256) k = 256;
    if (k > V) k = V;
    int cnt = k, pick = 0;
    if (!greedy && cnt > 1) {
        const float mx = c[0].v;
        if (top_p < 1.0f) {
            float tot = 0.0f;
            for (int i = 0; i < cnt; ++i) tot += oracle_expf(c[i].v - mx);
            float cum = 0.0f; int keep = cnt;
            for (int i = 0; i < cnt; ++i) { cum += oracle_expf(c[i].v - mx); if (cum >= top_p * tot) { keep = i + 1; break; } }
            cnt = keep;
        }
        if (min_p > 0.0f) {
            int keep = 1;
            for (int i = 1; i < cnt; ++i) { if (oracle_expf(c[i].v - mx) >= min_p) keep = i + 1; else break; }
            cnt = keep;
        }
        const float inv_t = 1.0f / temp;
        float tot = 0.0f;
        for (int i = 0; i < cnt; ++i) tot += oracle_expf((c[i].v - mx) * inv_t);
        const uint64_t z = oracle_splitmix((uint64_t)seed, counter);
        const float u = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
        const float target = u * tot;
        float cum = 0.0f;
        pick = cnt - 1;
        for (int i = 0; i < cnt; ++i) { cum += oracle_expf((c[i].v - mx) * inv_t); if (cum > target) { pick = i; break; } }
    }
    const int tok = c[pick].idx;
    free(c);
    return tok;
}

int oracle_sample(const float* logits, int V, int top_k, float top_p, float min_p, float temp, uint32_t seed, uint64_t counter,
                  int n_bias, const int32_t* bias_ids, const float* bias_vals) {
    return oracle_sample_ex(logits, V, top_k, top_p, min_p, temp, seed, counter, n_bias, bias_ids, bias_vals, 1.0f, 0.0f, 0.0f, 0, 0);
}
