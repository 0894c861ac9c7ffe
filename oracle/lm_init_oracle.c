/*
 * lm_init_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the device-side random
 * initialiser of the bench LM (realtime_codec_agent_amd/csrc/rca_lm.hip: lm_random_bf16_kernel), so that the
 * ~1B random-init model of BASELINE configs 3/4 can be rebuilt on the host for oracle/lm_ref.py.  The reference
 * has no counterpart (its weights come from a trained checkpoint, realtime_agent_resources.py:12).
 * value(tensor_id, i) = scale * (sum of the four 16-bit fields of splitmix64(seed ^ tensor_id * C, i) - 131070),
 * computed in float32 and rounded to bf16 (nearest even).  oracle/lm_ref.py keeps a numpy version of the same
 * formula; tests check the two against each other.
 */
#include <stdint.h>
#include <string.h>

static inline uint64_t splitmix(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + ctr * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

void oracle_random_bf16(uint64_t seed, uint64_t tensor_id, int64_t start, int64_t n, float scale, uint16_t* out) {
    const uint64_t s = seed ^ (tensor_id * 0xD6E8FEB86659FD93ull);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t z = splitmix(s, (uint64_t)(start + i));
        const int sum = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)((z >> 48) & 0xFFFF);
        const float f = (float)(sum - 131070) * scale;
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        out[i] = (uint16_t)(u >> 16);
    }
}

/* bf16 bits -> float32, in place widening of a whole tensor (n elements) */
void oracle_bf16_to_f32(const uint16_t* in, int64_t n, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t u = (uint32_t)in[i] << 16;
        memcpy(&out[i], &u, 4);
    }
}
