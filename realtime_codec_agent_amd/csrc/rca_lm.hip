// rca_lm.hip -- Llama-architecture autoregressive step on gfx950 (MI355X) with a
// llama_cpp.Llama-like control surface (include/rca.h, rca_lm_*).
//
// The deployed reference LM is codec_llama.py after persist_codec_embeddings
// (codec_llama.py:178-206): a vanilla Llama (RMSNorm, RoPE, GQA attention, SwiGLU, untied
// lm_head) evaluated by llama.cpp with 1-2 tokens per step (llamacpp_utils.py:145-161;
// realtime_agent_v2.py:355).  Here:
//   weights   bf16 in HBM, row-major [N][K]
//   decode    (1-8 tokens per pass) activations f32, every projection a wave-per-row-pair dot product with f32
//             accumulation, weights streamed once per step with non-temporal 16-B loads: HBM-bound, ~3 GB per step
//   prefill   (> 8 tokens) 128-token tiles on bf16 MFMA with hi/lo-split activations (lm_gemm128_kernel)
//   KV cache  fp16 [layer][pos][kv_head][64] (llama.cpp's default cache type)
//   attention decode: 256-key splits x blocks of 32 query rows on MFMA (K / V in whole rows through wave-private LDS images), the
//             splits merged inside the launch by data-tagged 8-byte granules; prefill: a flash-shaped kernel
//   sampler   logit bias + repeat / frequency / presence penalties patched in place -> histogram -> candidate gather ->
//             rank-by-counting top-k -> top-p/min-p/temperature -> inverse CDF (top_k 1..256), or the whole vocabulary by the
//             Gumbel-max rule behind radix-select rank / mass thresholds; counter-based RNG, polynomial exp / log, all on the device
//   the steady-state step (eval 1-2 tokens + sample) is captured once per context bucket into a hipGraph; the KV
//   position, input ids and RNG counter live in device memory so the graph replays unchanged.
#include <algorithm>
#include <type_traits>
#include <cmath>

#include "rca_common.h"

using namespace rca;

typedef unsigned short bf16_t;
typedef _Float16 f16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define LM_MAXM 1024       // tokens per prefill pass (activation buffers): up to 8 token blocks of 128 share one launch per GEMM
#define LM_STATE_DECODE_BYTES (8 + 4 * 16)   // n_tokens, m and the ids a decode step / frame graph needs (LmDevState prefix)
#define LM_TILE32 32       // token tile of the small-model prefill kernels (lm_gemm_mfma_kernel)
#define LM_GEMV_M 2        // tokens per decode pass (GEMV kernels); evals longer than LM_PREFILL_MIN go through the MFMA prefill path
#define LM_PREFILL_MIN 8    // evals of up to this many tokens are decode passes, longer ones prefill tiles
#define LM_KSLICE 2048     // widest hidden / attention-output dimension (one 16-byte chunk per lane and wave in the GEMV)
#define LM_MAXSPLIT 4      // ffn <= LM_KSLICE * LM_MAXSPLIT (up to four chunks per lane and wave)
#define ATT_KEYS 256       // keys per attention workgroup (8 waves x 32 keys)
#define LM_GRAPH_BUCKETS 8  // 4, 8, ..., 256 splits, the last bucket = all of them
#define ATT_EPOCH_INTS 64          // att_arrive: 64 launch counters (one per kv head), then the granules [kv head][split][8 rows][66]
#define ATT_TAG_MAXSP 128         // splits the in-launch merge can gather: nkv * splits <= 256 with at least 2 kv heads
#define ATT_TAG_SPINS (1 << 22)   // bound of the merger's re-polls (seconds): a NaN row tells
#define SAMP_MAXK 256
#define LM_FRAME_MAX 8        // steps of one frame graph (a chunk is 4-5 frames per channel, realtime_agent_config.py:21,56)
#define LM_FRAME_USER0 8     // LmDevState::ids[LM_FRAME_USER0 + i] = the user's token of frame i (ids[0..1] is the pair being evaluated)

struct LmDevState {
    int n_tokens;      // KV position of the first token of the current pass
    int m;             // tokens in the current pass
    int ids[LM_MAXM];
    unsigned long long rng_counter;
    int out_token;
    int pad;
    int frame_out[LM_FRAME_MAX];   // rca_lm_frame: the token sampled by each step of the frame
};

struct SamplerDev {
    int top_k;
    float top_p, min_p, temp;
    unsigned long long seed;
    int n_bias;
    int bias_ids[8];
    float bias_vals[8];
    // llama.cpp's penalties sampler (llama_sampler_penalties; llama-cpp-python adds it with penalty_last_n = last_n_tokens_size = 64
    // behind the logits processor and in front of top_k): over the last `last_n` tokens THIS sampler accepted
    float repeat_penalty, freq_penalty, presence_penalty;
    int last_n;
    int patch;      // 1: logit bias and / or penalties are applied IN PLACE by samp_prepare_kernel and undone by the sampler's last kernel
    int big_k;      // > 0: the "big" path cuts at this rank (top_k > SAMP_MAXK); 0: no rank cut
    int big_p;      // 1: the "big" path cuts at top_p of the candidates' mass (fixed point)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bf16_t f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// Cross-lane butterflies without the LDS crossbar: xor 32 / 16 through the gfx950 permlane swaps, xor 8..1 through
// DPP.  Pairing order is 32,16,8,4,2,1 exactly like a __shfl_xor loop, so sums keep the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_ROR4 0x124         // row_ror:4  (== xor 4 once lanes i and i^8 agree)
#define DPP_ROR8 0x128         // row_ror:8  (== xor 8 inside a row of 16)
#define DPP_HALF_MIRROR 0x141  // i -> 7-i inside 8 lanes (== xor 4 once the quads are uniform)
template <typename OP>
__device__ __forceinline__ float wave_butterfly(float v, OP op) {
    // v_permlane{32,16}_swap exchange halves / odd-even rows between TWO registers.  Written as asm: this
    // compiler's builtin mis-assigns the second result.  s_nop covers the VALU-write -> permlane-read hazard
    // the assembler does not see inside an asm block.
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // a = [lo, lo], b = [hi, hi]
    v = op(a, b);
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // a = [r0,r0,r2,r2], b = [r1,r1,r3,r3]
    v = op(a, b);
    v = op(v, dpp_mov<DPP_ROR8>(v));
    v = op(v, dpp_mov<DPP_ROR4>(v));
    v = op(v, dpp_mov<DPP_XOR2>(v));
    v = op(v, dpp_mov<DPP_XOR1>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return wave_butterfly(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
    return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); });
}

// ------------------------------------------------------------------------------------- embed
// The table is gathered, never streamed, so it keeps the precision it arrived in: bf16 rows for a bf16 checkpoint, f32 rows for
// an F32 / F16 tensor or a de-quantised Q8_0 / Q4_K one (llama.cpp de-quantises embedding rows exactly, get_rows) -- `f32tab`.
__global__ __launch_bounds__(256) void lm_embed_kernel(const LmDevState* __restrict__ stt, const void* __restrict__ table, int f32tab,
                                                       float* __restrict__ x, int H, int V) {
    const int m = blockIdx.x;
    if (m >= stt->m) return;
    int id = stt->ids[m];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    if (f32tab) {
        const float* row = reinterpret_cast<const float*>(table) + (long)id * H;
        for (int h = threadIdx.x; h < H; h += 256) x[(long)m * H + h] = row[h];
    } else {
        const bf16_t* row = reinterpret_cast<const bf16_t*>(table) + (long)id * H;
        for (int h = threadIdx.x; h < H; h += 256) x[(long)m * H + h] = __uint_as_float((unsigned)row[h] << 16);
    }
}

// --------------------------------------------------- residual add + RMSNorm shared arithmetic
// A wave owns one token row: lane l holds the 8-element chunks c = l + 64*it (it < 4, K <= 2048), adds the
// partial slices in order, accumulates sum(v^2) as an fma chain (it ascending, element ascending), reduces it
// with the xor-shuffle tree and scales (v * rstd) * w.  Used verbatim by the decode GEMV prologue (registers) and
// by the prefill norm kernel, so both paths produce identical bits.
template <typename F>
__device__ __forceinline__ float lm_row_norm(float (&v)[4][8], int nchunk, int K, float eps, F&& after_add) {
    const int lane = threadIdx.x & 63;
    float ss = 0.0f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            after_add(it, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(v[it][j], v[it][j], ss);
        }
    }
    ss = wave_sum(ss);
    return rsqrtf(ss / (float)K + eps);
}
__device__ __forceinline__ void lm_load_row(float (&v)[4][8], const float* __restrict__ row, int nchunk, bool valid) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane + 64 * it;
        if (valid && c < nchunk) {
            const float4 a = *reinterpret_cast<const float4*>(row + c * 8);
            const float4 b = *reinterpret_cast<const float4*>(row + c * 8 + 4);
            v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
            v[it][4] = b.x; v[it][5] = b.y; v[it][6] = b.z; v[it][7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[it][j] = 0.0f;
        }
    }
}
__device__ __forceinline__ void lm_add_row(float (&v)[4][8], const float* __restrict__ row, int nchunk) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            const float4 a = *reinterpret_cast<const float4*>(row + c * 8);
            const float4 b = *reinterpret_cast<const float4*>(row + c * 8 + 4);
            v[it][0] += a.x; v[it][1] += a.y; v[it][2] += a.z; v[it][3] += a.w;
            v[it][4] += b.x; v[it][5] += b.y; v[it][6] += b.z; v[it][7] += b.w;
        }
    }
}
__device__ __forceinline__ void lm_store_row(const float (&v)[4][8], float* __restrict__ row, int nchunk) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            *reinterpret_cast<float4*>(row + c * 8) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
            *reinterpret_cast<float4*>(row + c * 8 + 4) = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
        }
    }
}
__device__ __forceinline__ void lm_scale_row(float (&v)[4][8], float rstd, const float* __restrict__ w, int nchunk) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            const float4 a = *reinterpret_cast<const float4*>(w + c * 8);
            const float4 b = *reinterpret_cast<const float4*>(w + c * 8 + 4);
            const float wv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[it][j] = (v[it][j] * rstd) * wv[j];
        }
    }
}

// prefill chunks: one wave per token
__global__ __launch_bounds__(64) void lm_add_rmsnorm_kernel(const LmDevState* __restrict__ stt, const float* __restrict__ xin,
                                                            float* __restrict__ xout, const float* __restrict__ parts, int nparts,
                                                            long part_stride, const float* __restrict__ w, float* __restrict__ xn,
                                                            int K, float eps, bf16_t* __restrict__ hi = nullptr,
                                                            bf16_t* __restrict__ lo = nullptr) {
    const int m = blockIdx.x;
    if (m >= stt->m) return;
    const int nchunk = K >> 3;
    float v[4][8];
    lm_load_row(v, xin + (long)m * K, nchunk, true);
    for (int s2 = 0; s2 < nparts; ++s2) lm_add_row(v, parts + s2 * part_stride + (long)m * K, nchunk);
    if (xout) lm_store_row(v, xout + (long)m * K, nchunk);
    const float rstd = lm_row_norm(v, nchunk, K, eps, [](int, int) {});
    lm_scale_row(v, rstd, w, nchunk);
    if (hi) {   // prefill tiles: the normalised row goes straight out as the bf16 hi + lo pair the MFMA GEMM reads
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                unsigned ph[4], pl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16_t h0 = f32_to_bf16_rne(v[it][2 * j]), h1 = f32_to_bf16_rne(v[it][2 * j + 1]);
                    const bf16_t l0 = f32_to_bf16_rne(v[it][2 * j] - __uint_as_float((unsigned)h0 << 16));
                    const bf16_t l1 = f32_to_bf16_rne(v[it][2 * j + 1] - __uint_as_float((unsigned)h1 << 16));
                    ph[j] = (unsigned)h0 | ((unsigned)h1 << 16);
                    pl[j] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
                *reinterpret_cast<uint4*>(hi + (long)m * K + c * 8) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                *reinterpret_cast<uint4*>(lo + (long)m * K + c * 8) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            }
        }
    } else {
        lm_store_row(v, xn + (long)m * K, nchunk);
    }
}

// ------------------------------------------------------------------------------------ GEMV (decode passes, M <= 2)
// y[m][n] = sum_k W[n][k] * x[m][k] for the 1-2 tokens of a decode pass, weights streamed ONCE with non-temporal 16-byte
// loads.  Shape of the kernel:
//   * the 4 waves of a workgroup split K: wave w owns the chunk range [w * cpw, (w + 1) * cpw) (a chunk = 8 bf16 = one
//     16-byte lane load), so a lane keeps only ITS x values in registers (8 * NIT * M floats: 16 for the 2048-wide
//     projections) and the registers go to weights in flight instead: R rows x NIT chunks are requested per batch before
//     anything is consumed (R = 16: 16 KB per wave, ~200 KB per CU);
//   * per lane, per (row, token): one fma chain over its chunks (chunk ascending, element ascending);
//   * the R * M partial sums of a batch are reduced across the 64 lanes TOGETHER: two transposing steps on the gfx950
//     permlane swaps (xor 32, xor 16: every exchange halves the number of live registers) and four DPP steps inside the
//     rows of 16 -- R * M / 4 + ... instead of R * M butterflies; lane 0 of row q then holds value i + (V/4)(q&1) + (V/2)(q>>1);
//   * the 4 waves' sums meet in LDS (double-buffered by batch parity: one barrier per batch) and are added in wave order by
//     the first V threads, which run the fused epilogue.
// Any (R, batches per workgroup) gives the same bits: a value's reduction sequence does not depend on its slot.
//   PRO 1: prologue = RMSNorm of the residual row(s): each wave squares its own K range, the 4 partial sums are added in
//          wave order; xs = (v * rstd) * norm_w.  only_last: the single row handled is the pass's last token (head).
//   EPI 0: store (logits)          EPI 1: rows (2i, 2i+1) = (gate_i, up_i): h = silu(g) * u
//   EPI 2: QKV rows paired (d, d+32) inside each head: RoPE (HF rotate_half), q back to qkv[], k / v straight into the
//          fp16 KV cache at position n_tokens + m          EPI 3: add into the residual stream in place
struct GemvPro {
    const float* xin; const float* norm_w; float eps; int only_last;
};
struct GemvRope {
    const float* cos_t; const float* sin_t; f16_t* kc; f16_t* vc; int nh, nkv, n_ctx;
    int row_base;   // first row of this matrix inside the fused [q; k; v] layout (a multiple of 64): a Q4_K_M file keeps some attn_v tensors
                    // in another format than attn_q / attn_k, and such a layer's V projection is a matrix of its own
};
__device__ __forceinline__ void lane_swap32(float& a, float& b) {   // a = [a.lo, b.lo], b = [a.hi, b.hi]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap16(float& a, float& b) {   // a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, every lane gets it
    v = v + dpp_mov<DPP_ROR8>(v);
    v = v + dpp_mov<DPP_ROR4>(v);
    v = v + dpp_mov<DPP_XOR2>(v);
    v = v + dpp_mov<DPP_XOR1>(v);
    return v;
}
// V values per lane -> V / 4 registers; afterwards register i of row q (16 lanes) holds the full sum of value
// i + (V / 4) * (q & 1) + (V / 2) * (q >> 1).  V is a multiple of 4.
template <int V>
__device__ __forceinline__ void wave_reduce_transposed(float (&val)[V]) {
#pragma unroll
    for (int i = 0; i < V / 2; ++i) {
        lane_swap32(val[i], val[i + V / 2]);
        val[i] = val[i] + val[i + V / 2];
    }
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
        lane_swap16(val[i], val[i + V / 4]);
        val[i] = val[i] + val[i + V / 4];
    }
#pragma unroll
    for (int i = 0; i < V / 4; ++i) val[i] = row16_sum(val[i]);
}

// Q = 1: weights are q8_0 (GGUF block_q8_0: 32 int8 + one fp16 scale, what the reference deploys besides F16 and Q4_K_M,
// prep_test_model.sh:28-31) kept PACKED in HBM -- 8.5 bits per weight streamed instead of 16.  Device layout (built once at load
// from the 34-byte blocks): slot pairs interleaved, qs[pair][chunk] = 8 int8 of the pair's first row + 8 of its second row for
// the same 8 k (one 16-byte lane load feeds two rows with the x values the lane already holds), scales sc[k / 32][pair] =
// (fp16, fp16) so a lane fetches the scales of a whole batch with one or two vector loads.  Arithmetic per lane and row:
// p = fma chain of (float)q_j * x_j over its 8 elements, then d * p accumulated per chunk -- f32 activations throughout
// (llama.cpp quantises the activations to q8_1 for this product; here only the weights are quantised).
// (float)(int8) of byte `b` of a dword.  The dword must have passed through q8_opaque() first:
//   * if the optimiser can see that a loaded 16-byte value is only ever used byte by byte, it re-types the LOAD as sixteen byte
//     values and unpacks them the moment it lands -- 4x the registers and nothing left in flight;
//   * __builtin_amdgcn_sbfe + cast makes this compiler (ROCm 7.2) emit v_cvt_f32_u32_sdwa sext(...): the UNSIGNED conversion of
//     the sign-extended byte (4.29e9 for -1).
// Behind the opaque copy the plain shift / mask / cast form selects v_cvt_f32_i32_sdwa sext(BYTE_n): one instruction per element.
__device__ __forceinline__ unsigned q8_opaque(unsigned w) {
    asm volatile("" : "+v"(w));
    return w;
}
__device__ __forceinline__ float q8_byte_to_f32(unsigned w, int b) { return (float)(signed char)((w >> (8 * b)) & 0xffu); }
// (float) of UNSIGNED byte B of a dword as ONE instruction.  From the plain shift / mask / cast form this compiler selects
// v_cvt_f32_ubyte0 for byte 0 only and v_bfe_u32 + v_cvt_f32_ubyte0 for bytes 1-3: 96 extra instructions per 128 weights in the
// Q4_K body, whose VALU work (4.8 instructions per weight, not its bytes) is what bounds the Q4_K GEMVs: lm_head 70 -> 59.5 us,
// gate/up 9.6 -> 8.6, the step 0.686 -> 0.638 ms at 6.6 k context (profiles/r04/experiments/lm_step_micro.txt).
template <int B>
__device__ __forceinline__ float ubyte_to_f32(unsigned w) {
    float f;
    if constexpr (B == 0) asm("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f) : "v"(w));
    else if constexpr (B == 1) asm("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f) : "v"(w));
    else if constexpr (B == 2) asm("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f) : "v"(w));
    else asm("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(f) : "v"(w));
    return f;
}
struct GemvQ8 {          // the packed (quantised) forms: q8_0, and Q4_K (dd != nullptr)
    const u32x4* qs;      // q8_0: [N / 2][K / 8] 16-byte units.  Q4_K: [N / 4][K / 8] units = 8 nibbles x 4 slots (q4k_* below)
    const unsigned* sc;   // q8_0: [ceil(N / 16)][K / 32][8]: (fp16, fp16) per pair and 32-element block, pairs in groups of 8 -- the scales a
                          // wave needs for a batch (<= 8 pairs x its 16 blocks) are ONE contiguous run (was [K / 32][N / 2]: 16
                          // separate lines per wave and load).  Q4_K: ushort (sc | m << 8) per slot and 32-element sub-block,
                          // [ceil(N / 16)][K / 32][16]
    const unsigned* dd;   // Q4_K: (fp16 d | fp16 dmin << 16) per slot and 256-element super-block, [ceil(N / 16)][K / 256][16]
};
// Q4_K (GGUF block_q4_K, what llama-quantize Q4_K_M writes for most tensors, prep_test_model.sh:31): 4-bit values with a 6-bit scale
// and a 6-bit minimum per 32 and two fp16 factors per 256: value = (d * sc) * q - (dmin * m).  Device layout, built once at load:
// slots (rows in the order the GEMV's epilogues pair them) in quads, qs[quad][k / 8] = one dword per slot holding the 8 nibbles of
// 8 consecutive k (nibble j at bits 4 j): a 16-byte lane load feeds four rows with the x values the lane holds; 4.6 bits per weight
// streamed.  Arithmetic per lane, slot and 8-k chunk: p = fma chain of q_j x_j (j ascending), then val += (d * sc) p - (dmin * m) sum(x).
__host__ __device__ __forceinline__ long q4k_scm_index(long slot, long kblock, long nkb) { return ((slot >> 4) * nkb + kblock) * 16 + (slot & 15); }
__host__ __device__ __forceinline__ long q8_sc_index(long pair, long kblock, long nkb) { return ((pair >> 3) * nkb + kblock) * 8 + (pair & 7); }
// The format a projection matrix is kept in (ONE copy per matrix; the decode GEMV streams it, the prefill tiles de-quantise it while
// staging): the GEMV's template parameter Q.
#ifndef RCA_GEMV_VARIANT
#define RCA_GEMV_VARIANT 0
#endif
#define WF_BF16 0   // bf16 [N][K]
#define WF_Q8 1     // GGUF q8_0, pair-interleaved (GemvQ8)
#define WF_F16 2    // fp16 [N][K] (the reference's default GGUF is F16, realtime_agent_resources.py:12)
#define WF_Q4K 3    // GGUF Q4_K, quad-interleaved nibbles (GemvQ8 with dd)
#define WF_Q6K 4    // GGUF Q6_K re-encoded losslessly: int8 values (the 6-bit value - 32) in the q8_0 pair layout + one f32 scale d * sc per
                    // row and group of 16 ((s_a, s_b) per pair, pairs in groups of 8: [ceil(N / 16)][K / 16][8][2] floats); 10 bits per
                    // weight streamed (the file holds 6.6); d * sc * q is exact in f32, so the values are llama.cpp's dequantize_row_q6_K's
// minimum waves per SIMD asked of the register allocator.  The q8_0 bodies otherwise spread over 200+ registers (one wave per
// SIMD) although their live set is ~130: a streaming kernel wants the occupancy.
constexpr int gemv_min_waves(int Q, int R, int NIT) { return (Q != WF_Q8 && Q != WF_Q4K && Q != WF_Q6K) ? 1 : (R * NIT >= 16 ? 2 : 4); }
template <int M, int NIT, int R, int PRO, int EPI, int Q = 0>
__global__ __launch_bounds__(256, gemv_min_waves(Q, R, NIT)) void lm_gemv_kernel(const LmDevState* __restrict__ stt, const bf16_t* __restrict__ W, GemvQ8 q8,
                                                      const float* __restrict__ x, int N, int K, float* __restrict__ y,
                                                      int batches_per_wg, int ldy, GemvPro pro, GemvRope rope) {
    // (argument order: the weight pointers of either form, x, N, K -- what the first loads need -- are the 14 dwords the dispatcher
    //  preloads into SGPRs)
    constexpr int V = R * M;
    static_assert(V % 4 == 0 && R % 2 == 0, "R * M must be a multiple of 4");
    __shared__ float kred[2][4][V];
    __shared__ float nred[M][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int nchunk = K >> 3;
    const int cpw = (nchunk + 3) >> 2;                       // chunks per wave
    const int c0 = wave * cpw;
    const int cn = max(0, min(cpw, nchunk - c0));            // this wave's chunk count (<= 64 * NIT)
    const int n_batches = (N + R - 1) / R;
    const int b_beg = blockIdx.x * batches_per_wg;
    const int b_end = min(n_batches, b_beg + batches_per_wg);
    if (b_beg >= n_batches) return;   // never true for the grids launch_gemv_r builds; the peeled last batch below relies on it
    // slot r of batch b -> weight row
    auto row_of = [&](int b, int r) {
        if (EPI == 2) {   // slots (2s, 2s+1) = rows (d, d+32) of one head
            const int pair = b * (R / 2) + (r >> 1);
            return (pair >> 5) * 64 + (pair & 31) + 32 * (r & 1);
        }
        return b * R + r;
    };
    constexpr bool Q6 = Q == WF_Q6K;                 // the q8_0 body with another scale: f32 per group of 16 instead of fp16 per 32
    constexpr bool Q8 = Q == WF_Q8 || Q6, Q4 = Q == WF_Q4K;
    constexpr int NL = Q8 ? R / 2 : (Q4 ? R / 4 : R);   // 16-byte weight loads per chunk: one per row, per slot pair (q8_0 / Q6_K) or per slot quad (Q4_K)
    constexpr int H0 = NL / 2;                // the batch's registers refill in two halves: loads [0, H0) and [H0, NL)
    u32x4 wq[NL][NIT];
    uint2 wscm[Q4 ? NIT : 1][Q4 ? NL : 1];    // Q4_K: (sc | m << 8) of a quad's four slots for this lane's 32-element sub-block
    u32x4 wdd[Q4 ? NIT : 1][Q4 ? NL : 1];     //       (d | dmin << 16) of the four slots for this lane's 256-element super-block
    unsigned wsc[Q8 ? NIT : 1][Q8 ? R / 2 : 1];   // q8_0: (fp16, fp16) scales of the batch's pairs for this lane's 32-element block
    unsigned wsc2[Q6 ? NIT : 1][Q6 ? R / 2 : 1];  // Q6_K: wsc / wsc2 = the f32 scales (bits) of the pair's first / second row for this lane's group of 16
    const int npairs = N >> 1;
    // Every lane loads from a VALID address, whatever its chunk: a lane past the wave's range (narrow test models) re-reads chunk 0
    // and multiplies it by x = 0.  A select or an exec mask on the loaded value would be a vector instruction on the load's
    // result, i.e. a wait for it right behind its issue -- and with it for every load issued before.
    const int cbase = cn > 0 ? c0 : 0;
    // weight loads of ONE row (bf16 / fp16) or row pair (q8_0) of batch b into its registers; the q8_0 scales of a batch separately
    auto issue_row = [&](int b, int r) {
        if (Q4) {   // quad r of the batch: factors first (consumed with the quad), then the nibbles
            const long quad = min((long)b * (R / 4) + r, (long)((N + 3) >> 2) - 1);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = lane + 64 * it;
                const int cc = cbase + (c < cn ? c : 0);
                wdd[it][r] = *reinterpret_cast<const u32x4*>(q8.dd + q4k_scm_index(4 * quad, cc >> 5, K >> 8));
                wscm[it][r] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(q8.sc) + q4k_scm_index(4 * quad, cc >> 2, K >> 5));
                wq[r][it] = __builtin_nontemporal_load(q8.qs + quad * nchunk + cc);
            }
        } else if (Q8) {
            const long pp = min((long)b * (R / 2) + r, (long)npairs - 1);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = lane + 64 * it;
                wq[r][it] = __builtin_nontemporal_load(q8.qs + pp * nchunk + cbase + (c < cn ? c : 0));
            }
        } else {
            const int row = min(row_of(b, r), N - 1);
            const u32x4* wr = reinterpret_cast<const u32x4*>(W + (long)row * K) + cbase;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = lane + 64 * it;
                wq[r][it] = __builtin_nontemporal_load(wr + (c < cn ? c : 0));
            }
        }
    };
    // q8_0 scales of the pairs [s_beg, s_end) of batch b (a (fp16, fp16) dword per pair and 32-element block)
    auto issue_scales = [&](int b, int s_beg, int s_end) {
        if (!Q8) return;
        const long p0 = (long)b * (R / 2);
        if (Q6) {   // (s_a, s_b) floats per pair: two pairs per 16-byte load
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = lane + 64 * it;
                const int cc = cbase + (c < cn ? c : 0);
                const unsigned* sp = q8.sc + 2 * q8_sc_index(p0, cc >> 1, K >> 4);
#pragma unroll
                for (int s2 = 0; s2 < R / 2; s2 += 2) {
                    if (s2 < s_beg || s2 >= s_end) continue;
                    const u32x4 t = *reinterpret_cast<const u32x4*>(sp + 2 * s2);
                    wsc[it][s2] = t.x; wsc2[it][s2] = t.y; wsc[it][s2 + 1] = t.z; wsc2[it][s2 + 1] = t.w;
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = lane + 64 * it;
            const int cc = cbase + (c < cn ? c : 0);
            const unsigned* sp = q8.sc + q8_sc_index(p0, cc >> 2, K >> 5);   // a batch never crosses a group of 8 pairs (R <= 16)
            if (R / 2 >= 4) {   // four pairs per 16-byte load (groups are padded to 8 pairs: always in bounds and aligned)
#pragma unroll
                for (int s4 = 0; s4 < R / 8; ++s4) {
                    if (4 * s4 < s_beg || 4 * s4 >= s_end) continue;
                    const u32x4 t = *reinterpret_cast<const u32x4*>(sp + 4 * s4);
                    wsc[it][4 * s4 + 0] = t.x; wsc[it][4 * s4 + 1] = t.y; wsc[it][4 * s4 + 2] = t.z; wsc[it][4 * s4 + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < R / 2; ++s2)
                    if (s2 >= s_beg && s2 < s_end) wsc[it][s2] = sp[s2];
            }
        }
    };
    // half h (0 / 1) of batch b: its scales first (they are consumed with its first pair), then its rows
    auto load_half = [&](int b, int h) {
        issue_scales(b, h ? H0 : 0, h ? NL : H0);
#pragma unroll
        for (int r = 0; r < NL; ++r)
            if (h ? r >= H0 : r < H0) issue_row(b, r);
    };
    auto load_batch = [&](int b) { load_half(b, 0); load_half(b, 1); };
    // ---- this lane's x values.  Vector-memory results return in issue order, so the (short, L2-served) loads of x and of the
    // norm weights go out FIRST and the first batch of weight loads right behind them: the prologue below then waits for its own
    // operands only while the weights stay in flight under it.  (Issued the other way round, the first use of x would wait for
    // every weight load in front of it and the RMSNorm chain would run after the weights had landed instead of under them.)
    float xr[M][NIT][8];
    float nw[PRO == 1 ? NIT : 1][8];
    {
        int mbase = 0;
        if (PRO == 1 && pro.only_last) mbase = stt->m - 1;
        const float* xsrc = PRO == 1 ? pro.xin : x;
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = lane + 64 * it;
                if (c < cn) {
                    const float* p = xsrc + (long)(mbase + m) * K + (long)(c0 + c) * 8;
                    const float4 a = *reinterpret_cast<const float4*>(p);
                    const float4 b = *reinterpret_cast<const float4*>(p + 4);
                    xr[m][it][0] = a.x; xr[m][it][1] = a.y; xr[m][it][2] = a.z; xr[m][it][3] = a.w;
                    xr[m][it][4] = b.x; xr[m][it][5] = b.y; xr[m][it][6] = b.z; xr[m][it][7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xr[m][it][j] = 0.0f;
                }
            }
        if (PRO == 1) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = lane + 64 * it;
                if (c < cn) {
                    const float* p = pro.norm_w + (long)(c0 + c) * 8;
                    const float4 a = *reinterpret_cast<const float4*>(p);
                    const float4 b = *reinterpret_cast<const float4*>(p + 4);
                    nw[it][0] = a.x; nw[it][1] = a.y; nw[it][2] = a.z; nw[it][3] = a.w;
                    nw[it][4] = b.x; nw[it][5] = b.y; nw[it][6] = b.z; nw[it][7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) nw[it][j] = 0.0f;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the issue order: x / norm weights, then the weight stream
        // unconditional (the grid never holds a workgroup without a batch): a branch here would join a path with and one without
        // loads in flight, and the compiler would wait for ALL of them at the join
        load_batch(min(b_beg, n_batches - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (PRO == 1) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float ss = 0.0f;
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(xr[m][it][j], xr[m][it][j], ss);
                ss = wave_sum(ss);
                if (lane == 0) nred[m][wave] = ss;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float tot = ((nred[m][0] + nred[m][1]) + nred[m][2]) + nred[m][3];
                const float rstd = rsqrtf(tot / (float)K + pro.eps);
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xr[m][it][j] = (xr[m][it][j] * rstd) * nw[it][j];
            }
        }
    }
    int pos0 = 0;
    if (EPI == 2) pos0 = stt->n_tokens;
    float sx[Q4 ? M : 1][Q4 ? NIT : 1];   // Q4_K: sum of this lane's x values per chunk (the minimum term: - (dmin * m) * sum(x))
    if (Q4) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) t = t + xr[m][it][j];
                sx[m][it] = t;
            }
    }

    // The weight registers of a batch are two halves that refill separately: while the second half of batch b is being consumed the
    // first half of batch b + 1 is already in flight, and the other way round -- at least R / 2 loads per lane are outstanding from
    // the first instruction to the last half.  (Batch-synchronous -- wait for all R rows, compute them, then ask for the next R --
    // left the CU's memory pipe idle for the length of a batch's arithmetic between batches.)  Rows are consumed in issue order, so
    // the waits count down row by row.  Finer refills (row by row) make this compiler keep the next batch in a second register set
    // and copy it over behind a vmcnt(0) at the loop end, or spill (two-arm form): measured in profiles/r03/experiments.
    auto consume_row = [&](float (&val)[V], int r) {
        if (Q4) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const u32x4 a = wq[r][it];
                const u32x4 ddv = wdd[it][r];
                const uint2 sm = wscm[it][r];
                const unsigned aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {ddv.x, ddv.y, ddv.z, ddv.w}, sw[2] = {sm.x, sm.y};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = q8_opaque(aw[i]);
                    const unsigned lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;   // bytes of lo = nibbles 0, 2, 4, 6; of hi = 1, 3, 5, 7
                    float pq[M];
#pragma unroll
                    for (int m = 0; m < M; ++m) pq[m] = 0.0f;
                    float fq[8];   // nibble j = byte j / 2 of lo (j even) / hi (j odd)
                    fq[0] = ubyte_to_f32<0>(lo); fq[1] = ubyte_to_f32<0>(hi); fq[2] = ubyte_to_f32<1>(lo); fq[3] = ubyte_to_f32<1>(hi);
                    fq[4] = ubyte_to_f32<2>(lo); fq[5] = ubyte_to_f32<2>(hi); fq[6] = ubyte_to_f32<3>(lo); fq[7] = ubyte_to_f32<3>(hi);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                        for (int m = 0; m < M; ++m) pq[m] = __builtin_fmaf(fq[j], xr[m][it][j], pq[m]);
                    }
                    const f16x2 d2 = __builtin_bit_cast(f16x2, dw[i]);
                    const unsigned scm = (sw[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                    const float d1 = (float)d2[0] * (float)(scm & 0xffu);
                    const float m1 = (float)d2[1] * (float)(scm >> 8);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        float acc = val[m * R + 4 * r + i];
                        acc = __builtin_fmaf(d1, pq[m], acc);
                        acc = __builtin_fmaf(-m1, sx[m][it], acc);
                        val[m * R + 4 * r + i] = acc;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // one quad at a time (see the q8_0 branch)
            }
        } else if (Q8) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const u32x4 a = wq[r][it];
                const unsigned wa[2] = {q8_opaque(a.x), q8_opaque(a.y)}, wb[2] = {q8_opaque(a.z), q8_opaque(a.w)};
                float pa[M], pb[M];
#pragma unroll
                for (int m = 0; m < M; ++m) pa[m] = pb[m] = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float fa = q8_byte_to_f32(wa[j >> 2], j & 3);
                    const float fb = q8_byte_to_f32(wb[j >> 2], j & 3);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        pa[m] = __builtin_fmaf(fa, xr[m][it][j], pa[m]);
                        pb[m] = __builtin_fmaf(fb, xr[m][it][j], pb[m]);
                    }
                }
                const f16x2 d2 = __builtin_bit_cast(f16x2, wsc[it][r]);
                const float da = Q6 ? __uint_as_float(wsc[it][r]) : (float)d2[0], db = Q6 ? __uint_as_float(wsc2[it][r]) : (float)d2[1];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    val[m * R + 2 * r] = __builtin_fmaf(da, pa[m], val[m * R + 2 * r]);
                    val[m * R + 2 * r + 1] = __builtin_fmaf(db, pb[m], val[m * R + 2 * r + 1]);
                }
                // one pair at a time: without the fence the scheduler converts the bytes of EVERY load in flight up front
                // (16 floats each) and the kernel drops to one wave per SIMD
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const u32x4 a = wq[r][it];
                float f[8];
                if (Q == WF_F16) {   // fp16 weights: the widening is folded into v_fma_mix_f32 instead of a shift
                    const unsigned au[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f16x2 h2 = __builtin_bit_cast(f16x2, au[j]);
                        f[2 * j] = (float)h2[0];
                        f[2 * j + 1] = (float)h2[1];
                    }
                } else {
                    f[0] = bf16_lo(a.x); f[1] = bf16_hi(a.x); f[2] = bf16_lo(a.y); f[3] = bf16_hi(a.y);
                    f[4] = bf16_lo(a.z); f[5] = bf16_hi(a.z); f[6] = bf16_lo(a.w); f[7] = bf16_hi(a.w);
                }
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int j = 0; j < 8; ++j) val[m * R + r] = __builtin_fmaf(f[j], xr[m][it][j], val[m * R + r]);
            }
        }
    };
    // One batch: `more` (compile-time) = another batch of this workgroup follows, and the registers of each half refill as soon as
    // the half has been consumed.  The loop below runs the batches that have a successor; the last one is peeled, so the refill
    // is unconditional inside the loop (no merge of "old" and "new" register contents for the compiler to copy around).
    auto run_batch = [&](int b, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        float val[V];   // value index v = m * R + r
#pragma unroll
        for (int v = 0; v < V; ++v) val[v] = 0.0f;
        // epilogue operands that do not depend on the sums: requested here, ahead of the next batch's weights in the (in-order) return
        // queue, so that waiting for them later does not mean waiting for those
        float yv = 0.0f, cs = 0.0f, sn = 0.0f;
        if (EPI == 3 && tid < V) {
            const int m = tid / R, row = row_of(b, tid % R);
            if (row < N) yv = y[(long)m * ldy + row];
        }
        if (EPI == 2 && tid < V / 2) {
            const int m = tid / (R / 2), s = tid % (R / 2);
            const int d = (b * (R / 2) + s) & 31;
            const long pos = min(pos0 + m, rope.n_ctx - 1);
            cs = rope.cos_t[pos * 32 + d];
            sn = rope.sin_t[pos * 32 + d];
        }
#if RCA_GEMV_VARIANT == 2
        {   // batch-synchronous (the round-2 structure), for A/B runs
#pragma unroll
            for (int r = 0; r < NL; ++r) consume_row(val, r);
            if (more) load_batch(b + 1);
        }
#else
#pragma unroll
        for (int r = 0; r < H0; ++r) consume_row(val, r);
        if (more) {
            load_half(b + 1, 0);
            // the second half's arithmetic does not depend on the refill above and would be scheduled in front of it (the refill
            // would then be requested only after the WHOLE batch has landed): one register of every row passes through a volatile
            // statement that sits behind the refill
#pragma unroll
            for (int r = H0; r < NL; ++r) asm volatile("" : "+v"(wq[r][0].x));
        }
#pragma unroll
        for (int r = H0; r < NL; ++r) consume_row(val, r);
        if (more) load_half(b + 1, 1);
#endif
        wave_reduce_transposed<V>(val);
        const int buf = (b - b_beg) & 1;
        if ((lane & 15) == 0) {
            const int q = lane >> 4;
#pragma unroll
            for (int i = 0; i < V / 4; ++i) kred[buf][wave][i + (V / 4) * (q & 1) + (V / 2) * (q >> 1)] = val[i];
        }
        __syncthreads();
        auto total = [&](int v) { return ((kred[buf][0][v] + kred[buf][1][v]) + kred[buf][2][v]) + kred[buf][3][v]; };
        if (EPI == 0 || EPI == 3) {
            if (tid < V) {
                const int m = tid / R, row = row_of(b, tid % R);
                if (row < N) y[(long)m * ldy + row] = EPI == 3 ? yv + total(tid) : total(tid);
            }
        } else if (EPI == 1) {
            if (tid < V / 2) {
                const int m = tid / (R / 2), s = tid % (R / 2);
                const int row = b * R + 2 * s;
                if (row + 1 < N) {
                    const float g = total(m * R + 2 * s), u = total(m * R + 2 * s + 1);
                    y[(long)m * ldy + (row >> 1)] = (g / (1.0f + __expf(-g))) * u;
                }
            }
        } else {
            if (tid < V / 2) {
                const int m = tid / (R / 2), s = tid % (R / 2);
                const int rl = row_of(b, 2 * s);             // row inside this matrix
                const int r0 = rl + rope.row_base;           // row inside [q; k; v]
                const int pos = pos0 + m;
                if (rl + 32 < N && pos < rope.n_ctx) {
                    const float x1 = total(m * R + 2 * s), x2 = total(m * R + 2 * s + 1);
                    const int head = r0 >> 6, d = r0 & 63;   // d < 32
                    if (head < rope.nh + rope.nkv) {
                        const float o1 = x1 * cs + (-x2) * sn;
                        const float o2 = x2 * cs + x1 * sn;
                        if (head < rope.nh) {
                            y[(long)m * ldy + r0] = o1;
                            y[(long)m * ldy + r0 + 32] = o2;
                        } else {
                            f16_t* kp = rope.kc + ((long)pos * rope.nkv + (head - rope.nh)) * 64;
                            kp[d] = (f16_t)o1;
                            kp[d + 32] = (f16_t)o2;
                        }
                    } else {
                        f16_t* vp = rope.vc + ((long)pos * rope.nkv + (head - rope.nh - rope.nkv)) * 64;
                        vp[d] = (f16_t)x1;
                        vp[d + 32] = (f16_t)x2;
                    }
                }
            }
        }
    };
    for (int b = b_beg; b + 1 < b_end; ++b) run_batch(b, std::true_type{});
    run_batch(b_end - 1, std::false_type{});
}


// ------------------------------------------------------------------------- prefill: bf16 MFMA GEMM
// Prefill tiles (up to 32 tokens per pass) run the projections on v_mfma_f32_32x32x16_bf16: weights are bf16
// already; the f32 activations are split into bf16 hi + lo (x ~= hi + lo keeps ~16 mantissa bits), so every
// k step issues two MFMAs into one f32 accumulator.  A workgroup owns one 32-row x 32-token output tile, its 4
// waves split K and are summed in wave order through LDS (deterministic).  Fragments are loaded straight from
// global memory: A = 16 B of a weight row per lane, B = 16 B of a token row per lane.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void lm_split_bf16_kernel(const LmDevState* __restrict__ stt, const float* __restrict__ x,
                                                            bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int K) {
    const int m = blockIdx.y;
    if (m >= stt->m) return;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
        const float v = x[(long)m * K + k];
        const bf16_t h = f32_to_bf16_rne(v);
        const float r = v - __uint_as_float((unsigned)h << 16);
        hi[(long)m * K + k] = h;
        lo[(long)m * K + k] = f32_to_bf16_rne(r);
    }
}

#define GEMM_EPI_RESID 0
#define GEMM_EPI_ROPE 1
#define GEMM_EPI_SWIGLU 2
template <int EPI>
__global__ __launch_bounds__(256) void lm_gemm_mfma_kernel(const LmDevState* __restrict__ stt, const bf16_t* __restrict__ W,
                                                           const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl, int N, int K,
                                                           float* __restrict__ y, int ldy, bf16_t* __restrict__ oh, bf16_t* __restrict__ ol,
                                                           GemvRope rope) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int Mv = stt->m;
    auto n_of = [&](int rr) {
        if (EPI == GEMM_EPI_ROPE) return (tile >> 1) * 64 + (rr >> 4) * 32 + (tile & 1) * 16 + (rr & 15);
        return tile * 32 + rr;
    };
    const int Kw = K >> 2;                     // this wave's K slice
    const int k0 = wave * Kw;
    const int nsteps = Kw >> 4;
    const bf16_t* wrow = W + (long)n_of(lane & 31) * K + k0 + 8 * half;
    const bf16_t* hrow = xh + (long)(lane & 31) * K + k0 + 8 * half;
    const bf16_t* lrow = xl + (long)(lane & 31) * K + k0 + 8 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    constexpr int U = 4;
    uint4 a[U], bh[U], bl[U];
    auto load = [&](int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int st = min(s0 + u, nsteps - 1);
            a[u] = *reinterpret_cast<const uint4*>(wrow + st * 16);
            bh[u] = *reinterpret_cast<const uint4*>(hrow + st * 16);
            bl[u] = *reinterpret_cast<const uint4*>(lrow + st * 16);
        }
    };
    load(0);
    for (int s0 = 0; s0 < nsteps; s0 += U) {
        uint4 ca[U], cbh[U], cbl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { ca[u] = a[u]; cbh[u] = bh[u]; cbl[u] = bl[u]; }
        if (s0 + U < nsteps) load(s0 + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + u < nsteps) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, ca[u]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, cbh[u]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, cbl[u]), acc, 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] += red[0][r][lane]; acc[r] += red[1][r][lane]; acc[r] += red[2][r][lane]; }
    const int tok = lane & 31;
    if (tok >= Mv) return;
    if (EPI == GEMM_EPI_RESID) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n_of((r & 3) + 8 * (r >> 2) + 4 * half);
            if (n < N) y[(long)tok * ldy + n] = y[(long)tok * ldy + n] + acc[r];
        }
    } else if (EPI == GEMM_EPI_ROPE) {
        const int pos = stt->n_tokens + tok;
        if (pos >= rope.n_ctx) return;
#pragma unroll
        for (int r = 0; r < 8; ++r) {   // registers r and r+8 hold rows d and d+32 of one head
            const int n1 = n_of((r & 3) + 8 * (r >> 2) + 4 * half);
            const int head = n1 >> 6, d = n1 & 63;   // d < 32
            const float x1 = acc[r], x2 = acc[r + 8];
            if (head < rope.nh + rope.nkv) {
                const float c = rope.cos_t[(long)pos * 32 + d], sn = rope.sin_t[(long)pos * 32 + d];
                const float o1 = x1 * c + (-x2) * sn;
                const float o2 = x2 * c + x1 * sn;
                if (head < rope.nh) {
                    y[(long)tok * ldy + n1] = o1;
                    y[(long)tok * ldy + n1 + 32] = o2;
                } else {
                    f16_t* kp = rope.kc + ((long)pos * rope.nkv + (head - rope.nh)) * 64;
                    kp[d] = (f16_t)o1;
                    kp[d + 32] = (f16_t)o2;
                }
            } else {
                f16_t* vp = rope.vc + ((long)pos * rope.nkv + (head - rope.nh - rope.nkv)) * 64;
                vp[d] = (f16_t)x1;
                vp[d + 32] = (f16_t)x2;
            }
        }
    } else {  // SwiGLU: rows (2i, 2i+1) = (gate_i, up_i) sit in registers (2j, 2j+1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rr = ((2 * j) & 3) + 8 * ((2 * j) >> 2) + 4 * half;
            const int i = (tile * 32 + rr) >> 1;
            const float g = acc[2 * j], u = acc[2 * j + 1];
            const float hv = (g / (1.0f + __expf(-g))) * u;
            const bf16_t hb = f32_to_bf16_rne(hv);
            oh[(long)tok * ldy + i] = hb;
            ol[(long)tok * ldy + i] = f32_to_bf16_rne(hv - __uint_as_float((unsigned)hb << 16));
        }
    }
}

// advance the device-side KV position after a pass
__global__ void lm_advance_kernel(LmDevState* stt) { stt->n_tokens += stt->m; }
// steady-state step: next pass's first id is the token just sampled (realtime_agent_v2.py:355-363)
__global__ void lm_copy_logits_row_kernel(const float* __restrict__ src, float* __restrict__ dst, int V) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// --------------------------------------------------------------------------------- sampler
// Deterministic exp for x <= 0 (bit-identical to oracle/sampler_oracle.c): Cody-Waite reduction,
// degree-6 Horner in fma, exact ldexp.
__device__ __forceinline__ float rca_expf(float x) {
    if (x < -87.0f) return 0.0f;
    const float n = rintf(x * 1.44269504f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.3888889e-3f;
    p = __builtin_fmaf(p, r, 8.3333333e-3f);
    p = __builtin_fmaf(p, r, 4.1666668e-2f);
    p = __builtin_fmaf(p, r, 1.6666667e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return ldexpf(p, (int)n);
}
__device__ __forceinline__ unsigned long long splitmix(unsigned long long seed, unsigned long long ctr) {
    unsigned long long z = seed * 0x9E3779B97F4A7C15ull + ctr * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
// natural log of a positive normal float: cephes' logf polynomial as explicit fmas (the oracle's oracle_logf, bit for bit)
__device__ __forceinline__ float rca_logf(float x) {
    unsigned u = __float_as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, f, -1.1514610310e-1f);
    y = __builtin_fmaf(y, f, 1.1676998740e-1f);
    y = __builtin_fmaf(y, f, -1.2420140846e-1f);
    y = __builtin_fmaf(y, f, 1.4249322787e-1f);
    y = __builtin_fmaf(y, f, -1.6668057665e-1f);
    y = __builtin_fmaf(y, f, 2.0000714765e-1f);
    y = __builtin_fmaf(y, f, -2.4999993993e-1f);
    y = __builtin_fmaf(y, f, 3.3333331174e-1f);
    y = y * f * z;
    const float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}
__device__ __forceinline__ unsigned long long sample_key(float v, unsigned idx) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}
// the float whose monotone image is the key's upper half (sample_key's inverse: same bits, -0.0 and NaN payloads included)
__device__ __forceinline__ float sample_key_value(unsigned long long key) {
    const unsigned u = (unsigned)(key >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// Sampler = three launches (llama.cpp chain order top_k -> top_p -> min_p -> temp -> softmax -> draw;
// llamacpp_utils.py:39-95; realtime_agent_config.py:11-20,29):
//   1. samp_hist:   2048-bin histogram of the monotone keys' top 11 bits (128 workgroups, LDS then global)
//   2. samp_gather: every workgroup finds the bin holding the k-th largest value from the histogram and
//                   appends its slice's elements at or above that bin to a candidate list (a few hundred)
//   3. samp_final:  one workgroup: exact top-k of the candidates by 8-bit radix select over 64-bit
//                   (value, index) keys, bitonic sort, then the serial chain with the polynomial exp and
//                   the counter RNG.  Falls back to scanning the whole vocabulary if the list overflowed.
#define SAMP_BINS 2048
#define SAMP_CAND_CAP 4096
#define SAMP_FAST_CAP 1024   // candidate lists up to this size are ranked by counting in samp_final_kernel
#define SAMP_RING 128          // accepted tokens kept (window <= 64 + the <= 8 speculative steps of a frame graph: a cut frame only rewinds the counter)
#define SAMP_OV_CAP 80         // logits patched in place per draw: <= 64 penalised + <= 8 biased
#define SAMP_LEVELS 6          // radix levels of the 64-bit (value, ~index) key: 11 + 11 + 11 + 11 + 11 + 9 bits
struct SampWork {
    unsigned hist[SAMP_BINS];
    unsigned ncand;
    unsigned overflow;
    unsigned pad[2];
    unsigned long long cand[SAMP_CAND_CAP];
    // ---- round 4
    int ring[SAMP_RING];                   // token accepted by draw c sits in ring[c % SAMP_RING] (c = LmDevState::rng_counter before the draw)
    int ov_n;                              // logits currently patched in place
    int ov_ids[SAMP_OV_CAP];
    float ov_orig[SAMP_OV_CAP];
    unsigned long long rhist[2][SAMP_LEVELS][SAMP_BINS];   // [0] rank (counts) / [1] mass (2^-40 fixed point) histograms of the radix selects
    unsigned long long rstate[2][SAMP_LEVELS][2];          // after level l: {key prefix fixed so far, what is still needed inside it}
};

// The value the sampler sees for token i.  Logit bias and penalties are NOT applied here any more: samp_prepare_kernel writes the
// adjusted values of the (at most 72) affected tokens into the logits row before the sampler's passes and the sampler's last kernel
// puts the originals back, so every pass is a plain read (llama.cpp applies both to its candidate copy, never to the context's
// logits: _ctx.get_logits() / _scores keep the raw values here too).
__device__ __forceinline__ float samp_value(const float* __restrict__ logits, const SamplerDev* __restrict__ sp, int i) {
    (void)sp;
    return logits[i];
}

// llama_sampler_penalties_apply on one logit (llama.cpp llama-sampling.cpp): a token seen `count` times in the window
__device__ __forceinline__ float samp_penalise(float v, int count, float repeat, float freq, float present) {
    if (count <= 0) return v;
    v = v <= 0.0f ? v * repeat : v / repeat;
    const float t = (float)count * freq + present;
    return v - t;
}

// One workgroup of 128 threads, in front of the sampler's passes when SamplerDev::patch is set: thread j < window takes the token
// accepted j draws ago, thread 64 + b bias entry b.  The first holder of a token owns it: bias entries in order (as the logits
// processor adds them, llamacpp_utils.py:8-24), then the penalty; the original goes to the override list, the adjusted value into
// the row.
__global__ __launch_bounds__(128) void samp_prepare_kernel(float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                           const LmDevState* __restrict__ stt, SampWork* __restrict__ w) {
    __shared__ int tok[72];
    const int tid = threadIdx.x;
    const unsigned long long cnt = stt->rng_counter;
    const bool pen = sp->repeat_penalty != 1.0f || sp->freq_penalty != 0.0f || sp->presence_penalty != 0.0f;
    const int nwin = pen ? (int)min((unsigned long long)max(min(sp->last_n, 64), 0), cnt) : 0;
    const int nb = sp->n_bias;
    if (tid < 72) {
        int t = -1;
        if (tid < nwin) t = w->ring[(cnt - 1ull - (unsigned long long)tid) % SAMP_RING];
        else if (tid >= 64 && tid - 64 < nb) t = sp->bias_ids[tid - 64];
        tok[tid] = (t >= 0 && t < V) ? t : -1;
    }
    __syncthreads();
    if (tid < 72 && tok[tid] >= 0) {
        const int t = tok[tid];
        bool first = true;
        int count = 0;
        for (int q = 0; q < 72; ++q) {
            if (tok[q] != t) continue;
            if (q < tid) first = false;
            if (q < 64) ++count;
        }
        if (first) {
            const float orig = logits[t];
            float v = orig;
            for (int b = 0; b < nb; ++b)
                if (sp->bias_ids[b] == t) v = v + sp->bias_vals[b];
            v = samp_penalise(v, count, sp->repeat_penalty, sp->freq_penalty, sp->presence_penalty);
            const int slot = atomicAdd(&w->ov_n, 1);
            w->ov_ids[slot] = t;
            w->ov_orig[slot] = orig;
            logits[t] = v;
        }
    }
}
// the sampler's last kernel: the token just drawn joins the window, the patched logits get their values back
__device__ __forceinline__ void samp_accept_and_restore(float* __restrict__ logits, const SamplerDev* __restrict__ sp, SampWork* __restrict__ w,
                                                        unsigned long long draw, int tok, int tid) {
    if (tid == 0) w->ring[draw % SAMP_RING] = tok;
    if (sp->patch) {
        const int n = w->ov_n;
        if (tid < n) logits[w->ov_ids[tid]] = w->ov_orig[tid];
        __syncthreads();
        if (tid == 0) w->ov_n = 0;
    }
}
__device__ __forceinline__ int samp_k(const SamplerDev* sp, int V) {
    int k = sp->temp <= 0.0f ? 1 : sp->top_k;
    if (k <= 0 || k > SAMP_MAXK) k = SAMP_MAXK;
    return k > V ? V : k;
}

__global__ __launch_bounds__(256) void samp_hist_kernel(const float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                        SampWork* __restrict__ w) {
    __shared__ unsigned hl[SAMP_BINS];
    for (int b = threadIdx.x; b < SAMP_BINS; b += 256) hl[b] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
        const unsigned long long key = sample_key(samp_value(logits, sp, i), (unsigned)i);
        atomicAdd(&hl[(unsigned)(key >> 53)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SAMP_BINS; b += 256)
        if (hl[b]) atomicAdd(&w->hist[b], hl[b]);
}

__global__ __launch_bounds__(256) void samp_gather_kernel(const float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                          SampWork* __restrict__ w) {
    __shared__ unsigned wtot[4];
    __shared__ unsigned thr_bin;
    const int k = samp_k(sp, V);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The bin that holds the k-th largest value: the highest bin b with count(bins >= b) >= k, found as "group of 8 bins, then bin"
    // from the top.  Thread t owns bins 8 t .. 8 t + 7; S[t] = count of the bins of threads >= t is a suffix scan over the 256
    // threads (shuffles inside a wave + four wave totals), and the one thread whose group crosses k finishes inside its own
    // registers.  (One thread walking the 256 group sums and then 8 bins -- dependent LDS and global reads in a loop with an early
    // exit -- was ~7 of this kernel's 10 us.)  Unsigned integer sums: the same bin whatever the order.
    unsigned h8[8];
    {
        const uint4 a = *reinterpret_cast<const uint4*>(w->hist + tid * 8), b = *reinterpret_cast<const uint4*>(w->hist + tid * 8 + 4);
        h8[0] = a.x; h8[1] = a.y; h8[2] = a.z; h8[3] = a.w; h8[4] = b.x; h8[5] = b.y; h8[6] = b.z; h8[7] = b.w;
    }
    const unsigned s8 = ((h8[0] + h8[1]) + (h8[2] + h8[3])) + ((h8[4] + h8[5]) + (h8[6] + h8[7]));
    unsigned suf = s8;   // inclusive suffix sum over the lanes >= lane of this wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_down(suf, off);
        if (lane + off < 64) suf += o;
    }
    if (lane == 0) wtot[wave] = suf;
    __syncthreads();
    unsigned S = suf;
    for (int q = wave + 1; q < 4; ++q) S += wtot[q];
    // group g = the highest t >= 1 with S[t] >= k, or 0; the counts above it: S[t + 1] = S[t] - s8
    const unsigned above = S - s8;
    const bool mine = tid >= 1 ? (S >= (unsigned)k && above < (unsigned)k) : (above < (unsigned)k);
    if (mine) {
        unsigned acc = above;
        int j = 7;
        for (; j > 0; --j) {
            if (acc + h8[j] >= (unsigned)k) break;
            acc += h8[j];
        }
        thr_bin = (unsigned)(tid * 8 + j);
    }
    __syncthreads();
    const unsigned tb = thr_bin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
        const unsigned long long key = sample_key(samp_value(logits, sp, i), (unsigned)i);
        if ((unsigned)(key >> 53) >= tb) {
            const unsigned slot = atomicAdd(&w->ncand, 1u);
            if (slot < SAMP_CAND_CAP) w->cand[slot] = key;
            else w->overflow = 1u;
        }
    }
}

// Inside a frame graph samp_final_kernel's tail also does what a one-thread "next pair" launch and the next step's embedding launch
// did: record the token, advance the position, make [token, user id of this frame] the next pair and gather their embedding rows
// into x (two launches less per step of a frame).  (One launch for the WHOLE sampler -- histogram, last-arriver gather + selection --
// was built and measured: the single workgroup that then scans the 1 MB of logits for candidates is bound by one CU's memory pipe,
// 0.917 -> 0.996 ms per step: profiles/r03/experiments/one_launch_sampler.txt.)
struct SampTail {
    int frame_i;          // >= 0: step i of a frame graph (advance + next pair + embedding); -1: plain sample
    const void* table;    // embedding table
    int f32tab;
    float* x;
    int H;
};
__global__ __launch_bounds__(1024) void samp_final_kernel(float* logits, int V, const SamplerDev* __restrict__ sp,
                                                          LmDevState* __restrict__ stt, SampWork* __restrict__ w, SampTail tail) {
    __shared__ unsigned long long s_draw;
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel_prefix;
    __shared__ int sel_shift;      // bits already fixed (from the top)
    __shared__ unsigned need;      // how many more keys to take inside the current prefix bucket
    __shared__ unsigned long long cand[SAMP_MAXK];
    __shared__ unsigned ncand;
    __shared__ float cval[SAMP_MAXK];
    const int tid = threadIdx.x;
    const bool greedy = sp->temp <= 0.0f;
    const int k = samp_k(sp, V);
    // (requested together with the two counters it would otherwise wait for: slot tid exists whatever the count, SAMP_CAND_CAP >= 1024)
    const unsigned long long my_cand = w->cand[tid];
    const bool full = w->overflow != 0u;
    const int NN = full ? V : (int)min(w->ncand, (unsigned)SAMP_CAND_CAP);
    auto key_at = [&](int i) -> unsigned long long {
        return full ? sample_key(samp_value(logits, sp, i), (unsigned)i) : w->cand[i];
    };
    __shared__ unsigned long long ck[SAMP_FAST_CAP];
    __shared__ float e_plain[SAMP_MAXK], e_temp[SAMP_MAXK];
    __shared__ int s_tok;
    int n;
    if (!full && NN <= SAMP_FAST_CAP) {
        // the usual case, a few hundred candidates: every thread ranks its own key by counting the larger ones
        // (keys are unique), the k best land in sorted order -- two barriers instead of a radix select, a compaction
        // and a bitonic sort
        if (tid < NN) ck[tid] = my_cand;
        __syncthreads();
        if (tid < NN) {
            const unsigned long long key = ck[tid];
            int rank = 0;
#pragma unroll 8
            for (int j = 0; j < NN; ++j) rank += ck[j] > key ? 1 : 0;
            if (rank < SAMP_MAXK) cand[rank] = key;
        }
        __syncthreads();
        n = min(NN, SAMP_MAXK);
    } else {
        if (tid == 0) { sel_prefix = 0ull; sel_shift = 0; need = (unsigned)min(k, NN); ncand = 0; }
        __syncthreads();
        // radix select: afterwards the keys >= threshold are exactly the k largest
        for (int pass = 0; pass < 8; ++pass) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const int shift = 56 - 8 * pass;
            const unsigned long long prefix = sel_prefix;
            const int fixed = sel_shift;
            for (int i = tid; i < NN; i += 1024) {
                const unsigned long long key = key_at(i);
                const bool match = fixed == 0 ? true : ((key >> (64 - fixed)) == (prefix >> (64 - fixed)));
                if (match) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned acc = 0;
                int d = 255;
                for (; d > 0; --d) {
                    if (acc + hist[d] >= need) break;
                    acc += hist[d];
                }
                need -= acc;  // keys in higher digits are all taken
                sel_prefix = prefix | ((unsigned long long)d << shift);
                sel_shift = fixed + 8;
                if (hist[d] == need) need = 0xFFFFFFFFu;  // whole bucket taken: threshold fixed, stop refining
            }
            __syncthreads();
            if (need == 0xFFFFFFFFu) break;
        }
        const unsigned long long thr = sel_prefix;  // unfixed low bits are zero
        for (int i = tid; i < NN; i += 1024) {
            const unsigned long long key = key_at(i);
            if (key >= thr) {
                const unsigned slot = atomicAdd(&ncand, 1u);
                if (slot < SAMP_MAXK) cand[slot] = key;
            }
        }
        __syncthreads();
        n = min((int)ncand, SAMP_MAXK);
        if (tid < SAMP_MAXK && tid >= n) cand[tid] = 0ull;
        __syncthreads();
        for (int size = 2; size <= SAMP_MAXK; size <<= 1) {  // bitonic sort, descending; padding (0) sinks
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                if (tid < SAMP_MAXK) {
                    const int j = tid ^ stride;
                    if (j > tid) {
                        const unsigned long long a = cand[tid], b = cand[j];
                        const bool desc = (tid & size) == 0;
                        if (desc ? (a < b) : (a > b)) { cand[tid] = b; cand[j] = a; }
                    }
                }
                __syncthreads();
            }
        }
    }
    if (tid < n) cval[tid] = sample_key_value(cand[tid]);   // the key's upper half IS the value (the bits samp_value read at gather time)
    __syncthreads();
    if (tid < n && !greedy) {   // the exponentials in parallel; the sums below stay serial (their order is the definition)
        const float mx = cval[0];
        e_plain[tid] = rca_expf(cval[tid] - mx);
        e_temp[tid] = rca_expf((cval[tid] - mx) * (1.0f / sp->temp));
    }
    __syncthreads();
    if (tid == 0) {
        int cnt = min(n, k);
        int pick = 0;
        if (!greedy && cnt > 1) {
            if (sp->top_p < 1.0f) {  // smallest prefix whose probability mass reaches top_p
                float tot = 0.0f;
                for (int i = 0; i < cnt; ++i) tot += e_plain[i];
                float cum = 0.0f;
                int keep = cnt;
                for (int i = 0; i < cnt; ++i) {
                    cum += e_plain[i];
                    if (cum >= sp->top_p * tot) { keep = i + 1; break; }
                }
                cnt = keep;
            }
            if (sp->min_p > 0.0f) {  // keep p_i >= min_p * p_max
                int keep = 1;
                for (int i = 1; i < cnt; ++i)
                    if (e_plain[i] >= sp->min_p) keep = i + 1; else break;
                cnt = keep;
            }
            float tot = 0.0f;
            for (int i = 0; i < cnt; ++i) tot += e_temp[i];
            const unsigned long long z = splitmix(sp->seed, stt->rng_counter);
            const float u = (float)(unsigned)(z >> 40) * 5.9604644775390625e-08f;  // 2^-24
            const float target = u * tot;
            float cum = 0.0f;
            pick = cnt - 1;
            for (int i = 0; i < cnt; ++i) {
                cum += e_temp[i];
                if (cum > target) { pick = i; break; }
            }
        }
        const int tok = (int)(0xFFFFFFFFu - (unsigned)(cand[pick] & 0xFFFFFFFFull));
        s_draw = stt->rng_counter;
        stt->rng_counter += 1ull;
        stt->out_token = tok;
        if (tail.frame_i >= 0) {   // the pair just evaluated is in the cache; the next pair is [token just sampled, user's token of this frame]
            stt->frame_out[tail.frame_i] = tok;
            stt->n_tokens += 2;
            stt->ids[0] = tok;
            stt->ids[1] = stt->ids[LM_FRAME_USER0 + tail.frame_i];
        }
        s_tok = tok;
    }
    // re-arm the shared work area for the next call
    __syncthreads();
    samp_accept_and_restore(logits, sp, w, s_draw, s_tok, tid);
    for (int b = tid; b < SAMP_BINS; b += 1024) w->hist[b] = 0u;
    if (tid == 0) { w->ncand = 0u; w->overflow = 0u; }
    if (tail.frame_i >= 0) {   // lm_embed_kernel for the next pair (m = 2)
        const int id1 = stt->ids[LM_FRAME_USER0 + tail.frame_i];
        for (int e = tid; e < 2 * tail.H; e += 1024) {
            const int m = e / tail.H, hh = e - m * tail.H;
            int id = m == 0 ? s_tok : id1;
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            tail.x[e] = tail.f32tab ? reinterpret_cast<const float*>(tail.table)[(long)id * tail.H + hh]
                                    : __uint_as_float((unsigned)reinterpret_cast<const bf16_t*>(tail.table)[(long)id * tail.H + hh] << 16);
        }
    }
}

// ---- top_k <= 0: llama.cpp's "whole vocabulary" (its top-k sampler is then a no-op; llamacpp_utils.py:39-77 passes top_k straight
// through).  With top_p >= 1 the chain is min_p -> temp -> softmax -> draw over every token, and that draw is made by the Gumbel-max
// rule: token = argmax_i (v_i - max) / temp + g_i over the tokens that pass min_p, g_i = -log(-log(u_i)), u_i from the counter RNG
// keyed by (seed, draw, token).  Same distribution, no sort of 259 k candidates, one pass; ties go to the lowest index.  Three launches
// like the top-k sampler: slice maxima, slice winners, merge + the sampler tail.  (top_k <= 0 with top_p < 1 is refused at init.)
#define SAMPF_SLICES 64
__global__ __launch_bounds__(1024) void samp_full_max_kernel(const float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                             SampWork* __restrict__ w) {
    __shared__ float red[16];
    const int per = (V + SAMPF_SLICES - 1) / SAMPF_SLICES;
    const int i0 = blockIdx.x * per, i1 = min(V, i0 + per);
    float mx = -INFINITY;
    for (int i = i0 + threadIdx.x; i < i1; i += 1024) mx = fmaxf(mx, samp_value(logits, sp, i));
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k]);
        reinterpret_cast<float*>(w->hist)[blockIdx.x] = m;     // the histogram area is free in this mode (and re-zeroed by the merge)
    }
}
__global__ __launch_bounds__(1024) void samp_full_pick_kernel(const float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                              const LmDevState* __restrict__ stt, SampWork* __restrict__ w) {
    __shared__ unsigned long long red[16];
    const float* smax = reinterpret_cast<const float*>(w->hist);
    float mx = smax[0];
    for (int k = 1; k < SAMPF_SLICES; ++k) mx = fmaxf(mx, smax[k]);
    const float inv_t = 1.0f / sp->temp, min_p = sp->min_p;
    const unsigned long long draw = splitmix(sp->seed, stt->rng_counter);
    const int per = (V + SAMPF_SLICES - 1) / SAMPF_SLICES;
    const int i0 = blockIdx.x * per, i1 = min(V, i0 + per);
    unsigned long long best = 0ull;   // below every real key (a real key has bit 63 or bits of ~u set... see sample_key: never 0 with idx < 2^32 - 1)
    for (int i = i0 + threadIdx.x; i < i1; i += 1024) {
        const float d = samp_value(logits, sp, i) - mx;
        if (min_p > 0.0f && !(rca_expf(d) >= min_p)) continue;
        const unsigned long long z = splitmix(draw, (unsigned long long)i);
        const float u = (float)(unsigned)(((z >> 41) << 1) | 1ull) * 5.9604644775390625e-08f;
        const float g = -rca_logf(-rca_logf(u));
        const unsigned long long key = sample_key(__builtin_fmaf(d, inv_t, g), (unsigned)i);
        best = key > best ? key : best;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = red[0];
        for (int k = 1; k < 16; ++k) b = red[k] > b ? red[k] : b;
        w->cand[blockIdx.x] = b;
    }
}
__global__ __launch_bounds__(1024) void samp_full_final_kernel(float* logits, int V, const SamplerDev* __restrict__ sp, LmDevState* __restrict__ stt,
                                                               SampWork* __restrict__ w, SampTail tail) {
    __shared__ int s_tok;
    __shared__ unsigned long long s_draw;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned long long b = w->cand[0];
        for (int k = 1; k < SAMPF_SLICES; ++k) b = w->cand[k] > b ? w->cand[k] : b;
        const int tok = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
        s_draw = stt->rng_counter;
        stt->rng_counter += 1ull;
        stt->out_token = tok;
        if (tail.frame_i >= 0) {
            stt->frame_out[tail.frame_i] = tok;
            stt->n_tokens += 2;
            stt->ids[0] = tok;
            stt->ids[1] = stt->ids[LM_FRAME_USER0 + tail.frame_i];
        }
        s_tok = tok;
    }
    __syncthreads();
    samp_accept_and_restore(logits, sp, w, s_draw, s_tok, tid);
    for (int b = tid; b < SAMPF_SLICES; b += 1024) w->hist[b] = 0u;   // the top-k sampler expects a zeroed histogram
    if (tail.frame_i >= 0) {   // lm_embed_kernel for the next pair (m = 2), as in samp_final_kernel
        const int id1 = stt->ids[LM_FRAME_USER0 + tail.frame_i];
        for (int e = tid; e < 2 * tail.H; e += 1024) {
            const int m = e / tail.H, hh = e - m * tail.H;
            int id = m == 0 ? s_tok : id1;
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            tail.x[e] = tail.f32tab ? reinterpret_cast<const float*>(tail.table)[(long)id * tail.H + hh]
                                    : __uint_as_float((unsigned)reinterpret_cast<const bf16_t*>(tail.table)[(long)id * tail.H + hh] << 16);
        }
    }
}

// ---- The "big" path (round 4): top_k > SAMP_MAXK (a rank cut anywhere up to the vocabulary) and / or top_p < 1 without a small top_k
// (llamacpp_utils.py:39-77 hands both straight to llama.cpp; realtime_agent_config.py:11-13).  llama.cpp sorts the candidates and
// walks the sorted list; here the two cuts are THRESHOLD KEYS found by radix selects over the 64-bit (value, ~index) keys of the whole
// vocabulary -- 11 + 11 + 11 + 11 + 11 + 9 bits, one pass per level, 2048-bin histograms:
//   rank cut:  the top_k-th largest key T_k (histograms of counts);
//   mass cut:  the key T_p at which the mass of the keys above it first reaches top_p of the candidates' total, walking down from
//              the top (histograms of mass; the candidates are the keys >= T_k).  Masses are 2^-40 fixed point,
//              w_i = trunc(exp(v_i - max) * 2^40), so every sum is an integer sum -- independent of the order the atomics land in,
//              and the C restatement's sorted loop (oracle/sampler_oracle.c) gets the same threshold bit for bit;
//              need = max(1, trunc(top_p * W)) in double.
// Every workgroup of pass l first takes level l - 1's decision from that level's histogram (redundantly: a parallel suffix scan by
// one wave); workgroup 0 records it.  The draw itself is the whole-vocabulary sampler's Gumbel-max pick restricted to keys >= the
// threshold (and to min_p).
__device__ __constant__ int samp_rshift[SAMP_LEVELS] = {53, 42, 31, 20, 9, 0};
__device__ __forceinline__ int samp_rbins(int level) { return level == SAMP_LEVELS - 1 ? 512 : 2048; }

// decision of one level: walking the bins from the top, the bin in which the running total first reaches `need`, and what is still
// needed inside it.  Called by all 256 threads of a workgroup; wave 0 decides.  total_out (optional): the sum of all bins.
__device__ void samp_radix_decide(const unsigned long long* __restrict__ hist, int nbins, unsigned long long need, bool need_from_total, float top_p,
                                  int* bin_out, unsigned long long* need_out) {
    __shared__ unsigned long long grp[256];
    __shared__ int s_bin;
    __shared__ unsigned long long s_need;
    const int tid = threadIdx.x, per = nbins / 256;      // 8 or 2 bins per thread
    unsigned long long g = 0;
    for (int j = 0; j < per; ++j) g += hist[tid * per + j];
    grp[tid] = g;
    __syncthreads();
    if (tid < 64) {
        unsigned long long s4 = grp[4 * tid] + grp[4 * tid + 1] + grp[4 * tid + 2] + grp[4 * tid + 3];
        // inclusive suffix sums over the lanes (lane 63 = the top bins)
        unsigned long long incl = s4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_down(incl, off);
            if (tid + off < 64) incl += o;
        }
        const unsigned long long total = __shfl(incl, 0);
        if (need_from_total) {   // mass cut: the target is a share of the candidates' total mass
            need = (unsigned long long)((double)top_p * (double)total);
            if (need < 1ull) need = 1ull;
        }
        const unsigned long long above = incl - s4;                   // total of the lanes above this one
        const bool here = above < need && incl >= need;
        const unsigned long long vote = __builtin_amdgcn_ballot_w64(here);
        if (vote == 0ull) {                                           // cannot happen while need <= total; keep the state defined
            if (tid == 0) { s_bin = 0; s_need = need; }
        } else if (here) {
            unsigned long long acc = above;
            int gi = 4 * tid + 3;
            for (; gi > 4 * tid; --gi) {
                if (acc + grp[gi] >= need) break;
                acc += grp[gi];
            }
            int b = gi * per + per - 1;
            for (; b > gi * per; --b) {
                if (acc + hist[b] >= need) break;
                acc += hist[b];
            }
            s_bin = b;
            s_need = need - acc;
        }
    }
    __syncthreads();
    *bin_out = s_bin;
    *need_out = s_need;
    __syncthreads();
}

// state after level `upto` (prefix of the key fixed so far, what is still needed inside it), for select m (0 rank, 1 mass);
// upto = -1: the start.  Workgroup 0 records each decision it takes.
__device__ void samp_radix_state(const SamplerDev* __restrict__ sp, SampWork* __restrict__ w, int m, int upto, unsigned long long* prefix,
                                 unsigned long long* need) {
    if (upto < 0) { *prefix = 0ull; *need = m == 0 ? (unsigned long long)sp->big_k : 0ull; return; }
    unsigned long long pfx = 0ull, nd = m == 0 ? (unsigned long long)sp->big_k : 0ull;
    if (upto >= 1) { pfx = w->rstate[m][upto - 1][0]; nd = w->rstate[m][upto - 1][1]; }
    int bin;
    unsigned long long nd2;
    samp_radix_decide(w->rhist[m][upto], samp_rbins(upto), nd, m == 1 && upto == 0, sp->top_p, &bin, &nd2);
    pfx |= (unsigned long long)bin << samp_rshift[upto];
    if (blockIdx.x == 0 && threadIdx.x == 0) { w->rstate[m][upto][0] = pfx; w->rstate[m][upto][1] = nd2; }
    *prefix = pfx;
    *need = nd2;
}
__device__ __forceinline__ unsigned long long samp_mass_fx(float d) {   // exp(v - max) in 2^-40 fixed point (d <= 0)
    return (unsigned long long)(rca_expf(d) * 1099511627776.0f);
}

// pass `level` of select m over the whole vocabulary
__global__ __launch_bounds__(256) void samp_radix_pass_kernel(const float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                              SampWork* __restrict__ w, int m, int level) {
    __shared__ unsigned long long hl[SAMP_BINS];
    unsigned long long prefix, need;
    samp_radix_state(sp, w, m, level - 1, &prefix, &need);
    const int nb = samp_rbins(level), shift = samp_rshift[level];
    for (int b = threadIdx.x; b < nb; b += 256) hl[b] = 0ull;
    __syncthreads();
    const int fixed = level == 0 ? 0 : 64 - samp_rshift[level - 1];       // key bits already decided (from the top)
    unsigned long long tk = 0ull;
    float mx = 0.0f;
    if (m == 1) {
        if (sp->big_k > 0) tk = w->rstate[0][SAMP_LEVELS - 1][0];          // candidates = keys >= T_k (the rank select is finished)
        const float* smax = reinterpret_cast<const float*>(w->hist);
        mx = smax[0];
        for (int k = 1; k < SAMPF_SLICES; ++k) mx = fmaxf(mx, smax[k]);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
        const float v = logits[i];
        const unsigned long long key = sample_key(v, (unsigned)i);
        if (fixed && (key >> (64 - fixed)) != (prefix >> (64 - fixed))) continue;
        if (m == 1 && key < tk) continue;
        const unsigned long long add = m == 0 ? 1ull : samp_mass_fx(v - mx);
        if (add) atomicAdd(&hl[(unsigned)(key >> shift) & (unsigned)(nb - 1)], add);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256)
        if (hl[b]) atomicAdd(&w->rhist[m][level][b], hl[b]);
}
// closes select m: the last level's decision fixes the threshold key, recorded in rstate[m][last][0]
__global__ __launch_bounds__(256) void samp_radix_finish_kernel(const SamplerDev* __restrict__ sp, SampWork* __restrict__ w, int m) {
    unsigned long long prefix, need;
    samp_radix_state(sp, w, m, SAMP_LEVELS - 1, &prefix, &need);
    unsigned long long* hz = &w->rhist[m][0][0];          // the select is done: its histograms are zero again for the next draw
    for (int b = threadIdx.x; b < SAMP_LEVELS * SAMP_BINS; b += 256) hz[b] = 0ull;
}
// the whole-vocabulary pick restricted to keys >= the threshold of the last select that ran
__global__ __launch_bounds__(1024) void samp_big_pick_kernel(const float* __restrict__ logits, int V, const SamplerDev* __restrict__ sp,
                                                             const LmDevState* __restrict__ stt, SampWork* __restrict__ w) {
    __shared__ unsigned long long red[16];
    const float* smax = reinterpret_cast<const float*>(w->hist);
    float mx = smax[0];
    for (int k = 1; k < SAMPF_SLICES; ++k) mx = fmaxf(mx, smax[k]);
    const unsigned long long thr = w->rstate[sp->big_p ? 1 : 0][SAMP_LEVELS - 1][0];
    const float inv_t = 1.0f / sp->temp, min_p = sp->min_p;
    const unsigned long long draw = splitmix(sp->seed, stt->rng_counter);
    const int per = (V + SAMPF_SLICES - 1) / SAMPF_SLICES;
    const int i0 = blockIdx.x * per, i1 = min(V, i0 + per);
    unsigned long long best = 0ull;
    for (int i = i0 + threadIdx.x; i < i1; i += 1024) {
        const float v = logits[i];
        if (sample_key(v, (unsigned)i) < thr) continue;
        const float d = v - mx;
        if (min_p > 0.0f && !(rca_expf(d) >= min_p)) continue;
        const unsigned long long z = splitmix(draw, (unsigned long long)i);
        const float u = (float)(unsigned)(((z >> 41) << 1) | 1ull) * 5.9604644775390625e-08f;
        const float g = -rca_logf(-rca_logf(u));
        const unsigned long long key = sample_key(__builtin_fmaf(d, inv_t, g), (unsigned)i);
        best = key > best ? key : best;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = red[0];
        for (int k = 1; k < 16; ++k) b = red[k] > b ? red[k] : b;
        w->cand[blockIdx.x] = b;
    }
}

// probs[i] = softmax(logits)[ids[i]] : one workgroup, two sweeps (max, sum)
// softmax(logits)[ids] in two launches (the agent asks for P(<|end_audio|>) once per frame, realtime_agent_v2.py:448-452):
// 64 workgroups reduce a slice each to (max, sum of exp relative to it); one wave merges the slices in slice order.
#define PROBS_SLICES 64
__global__ __launch_bounds__(1024) void lm_softmax_slices_kernel(const float* __restrict__ logits, int V, float* __restrict__ part) {
    __shared__ float red[16];
    __shared__ float smax;
    const int per = (V + PROBS_SLICES - 1) / PROBS_SLICES;
    const int i0 = blockIdx.x * per, i1 = min(V, i0 + per);
    float mx = -INFINITY;
    for (int i = i0 + threadIdx.x; i < i1; i += 1024) mx = fmaxf(mx, logits[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { float m = red[0]; for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]); smax = m; }
    __syncthreads();
    mx = smax;
    float s = 0.0f;
    for (int i = i0 + threadIdx.x; i < i1; i += 1024) s += __expf(logits[i] - mx);
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0;
        for (int w = 0; w < 16; ++w) t += red[w];
        part[2 * blockIdx.x] = mx;
        part[2 * blockIdx.x + 1] = t;
    }
}
__global__ __launch_bounds__(64) void lm_token_probs_kernel(const float* __restrict__ logits, int V, const float* __restrict__ part,
                                                            const int* __restrict__ ids, int n, float* __restrict__ probs) {
    const float pm = part[2 * threadIdx.x], ps = part[2 * threadIdx.x + 1];   // PROBS_SLICES == 64 lanes
    const float mx = wave_max(pm);
    const float contrib = (pm == -INFINITY) ? 0.0f : ps * __expf(pm - mx);
    float tot = 0.0f;
    for (int b = 0; b < PROBS_SLICES; ++b) tot += __shfl(contrib, b);   // slice order
    if ((int)threadIdx.x < n) {
        const int id = ids[threadIdx.x];
        probs[threadIdx.x] = (id >= 0 && id < V) ? __expf(logits[id] - mx) / tot : 0.0f;
    }
}

// ------------------------------------------------------------------------- random init (bench)
// value(tensor_id, i) = std * 1.7320508 * (sum of four 16-bit uniforms - 131070) / 65535 ~ N(0, std^2)
// (Irwin-Hall, integer arithmetic only, reproduced by oracle/lm_ref.py), rounded to bf16 (RNE).
__global__ __launch_bounds__(256) void lm_random_bf16_kernel(bf16_t* __restrict__ out, long n, unsigned long long seed,
                                                             unsigned long long tensor_id, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long z = splitmix(seed ^ (tensor_id * 0xD6E8FEB86659FD93ull), (unsigned long long)i);
        const int sum = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)((z >> 48) & 0xFFFF);
        out[i] = f32_to_bf16_rne((float)(sum - 131070) * scale);
    }
}
__global__ __launch_bounds__(256) void lm_fill_f32_kernel(float* __restrict__ out, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = v;
}
__global__ __launch_bounds__(256) void lm_rope_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ cos_t,
                                                            float* __restrict__ sin_t, int n_ctx, int half) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_ctx * half) return;
    const int pos = (int)(i / half), d = (int)(i - (long)pos * half);
    const float ang = (float)pos * inv_freq[d];
    cos_t[i] = cosf(ang);
    sin_t[i] = sinf(ang);
}
// interleave gate/up rows: dst row 2i = gate_i, 2i+1 = up_i
__global__ __launch_bounds__(256) void lm_interleave_rows_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                                                 bf16_t* __restrict__ dst, int F, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)2 * F * H) return;
    const long row = i / H;
    const int h = (int)(i - row * H);
    const long src = (row >> 1) * H + h;
    dst[i] = (row & 1) ? up[src] : gate[src];
}
__global__ __launch_bounds__(256) void lm_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = f32_to_bf16_rne(in[i]);
}

// ------------------------------------------------------------------------- weight formats at load
// GGUF block_q8_0 = { fp16 d; int8 qs[32] } (34 bytes), value = d * q.  "Plain" form on the device: q [N][K] int8 and
// d [N][K / 32] fp16; from there the pair-interleaved layout the GEMV streams (GemvQ8).  There is no second copy: the prefill tiles
// de-quantise the packed blocks while staging (lm_gemm128_kernel).
__global__ __launch_bounds__(256) void lm_q8_unblock_kernel(const unsigned char* __restrict__ blocks, long nblocks, signed char* __restrict__ q,
                                                            f16_t* __restrict__ d) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks * 32; i += (long)gridDim.x * blockDim.x) {
        const long blk = i >> 5;
        const int j = (int)(i & 31);
        const unsigned char* bp = blocks + blk * 34;
        q[i] = (signed char)bp[2 + j];
        if (j == 0) {
            const unsigned short bits = (unsigned short)bp[0] | ((unsigned short)bp[1] << 8);
            d[blk] = __builtin_bit_cast(f16_t, bits);
        }
    }
}
__device__ __forceinline__ float w16_to_f32(bf16_t bits, int is_f16) {
    return is_f16 ? (float)__builtin_bit_cast(f16_t, bits) : __uint_as_float((unsigned)bits << 16);
}
// llama.cpp's quantize_row_q8_0: d = amax / 127, q = round(x / d) (ties away from zero), d stored as fp16
__global__ __launch_bounds__(256) void lm_q8_quantize_kernel(const bf16_t* __restrict__ w, int is_f16, long nblocks, signed char* __restrict__ q,
                                                             f16_t* __restrict__ d) {
    for (long blk = (long)blockIdx.x * blockDim.x + threadIdx.x; blk < nblocks; blk += (long)gridDim.x * blockDim.x) {
        float v[32];
        float amax = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            v[j] = w16_to_f32(w[blk * 32 + j], is_f16);
            amax = fmaxf(amax, fabsf(v[j]));
        }
        const float dd = amax / 127.0f;
        const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) q[blk * 32 + j] = (signed char)(int)roundf(v[j] * id);
        d[blk] = (f16_t)dd;
    }
}
__global__ __launch_bounds__(256) void lm_q8_dequant_f32_kernel(const signed char* __restrict__ q, const f16_t* __restrict__ d, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (float)d[i >> 5] * (float)q[i];
}
__global__ __launch_bounds__(256) void lm_bf16_to_f16_kernel(const bf16_t* __restrict__ in, f16_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (f16_t)__uint_as_float((unsigned)in[i] << 16);   // round to nearest even (values below the fp16 range go subnormal / to zero)
}
__global__ __launch_bounds__(256) void lm_f16_to_f32_kernel(const f16_t* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
// rows of two byte matrices interleaved: dst row 2i = a row i, 2i+1 = b row i (gate/up: 16-bit values, q8_0 q and d)
__global__ __launch_bounds__(256) void lm_interleave_rows_bytes_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                                       unsigned char* __restrict__ dst, long rows, long row_bytes) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * rows * row_bytes; i += (long)gridDim.x * blockDim.x) {
        const long row = i / row_bytes, o = i - row * row_bytes;
        dst[i] = (row & 1) ? b[(row >> 1) * row_bytes + o] : a[(row >> 1) * row_bytes + o];
    }
}
// ---- Q4_K.  Plain form on the device: q [N][K] one byte per 4-bit value, scm [N][K / 32] ushort (sc | m << 8), dd [N][K / 256]
// uint (d | dmin << 16).
__global__ __launch_bounds__(256) void lm_q4k_unblock_kernel(const unsigned char* __restrict__ blocks, long nblocks, unsigned char* __restrict__ q,
                                                             unsigned short* __restrict__ scm, unsigned* __restrict__ dd) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks * 256; i += (long)gridDim.x * blockDim.x) {
        const long blk = i >> 8;
        const int e = (int)(i & 255);
        const unsigned char* bp = blocks + blk * 144;
        const int t = e >> 6, l = e & 31, high = (e >> 5) & 1;   // weights 64 t + 32 high + l: low / high nibble of qs[32 t + l]
        const unsigned char byte = bp[16 + 32 * t + l];
        q[i] = high ? (byte >> 4) : (byte & 0xF);
        if ((e & 31) == 0) {   // sub-block j = e / 32: get_scale_min_k4
            const int j = e >> 5;
            const unsigned char* sc = bp + 4;
            unsigned scv, mv;
            if (j < 4) { scv = sc[j] & 63; mv = sc[j + 4] & 63; }
            else { scv = (sc[j + 4] & 0xF) | ((sc[j - 4] >> 6) << 4); mv = (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4); }
            scm[blk * 8 + j] = (unsigned short)(scv | (mv << 8));
        }
        if (e == 0) dd[blk] = (unsigned)bp[0] | ((unsigned)bp[1] << 8) | ((unsigned)bp[2] << 16) | ((unsigned)bp[3] << 24);
    }
}
// This build's Q4_K quantiser (oracle/q4k_ref.py::quantize_q4_k, operation for operation): per 32 values offset o = -min(0, min w)
// and step s = (max w + o) / 15; per 256 d = max s / 63, dmin = max o / 63 (fp16); sc = round(s / d), m = round(o / dmin);
// q = clamp(round((w + dmin m) / (d sc)), 0, 15).  llama-quantize searches for better scales; the FORMAT and its de-quantisation are GGUF's.
__global__ __launch_bounds__(256) void lm_q4k_quantize_kernel(const bf16_t* __restrict__ w, int is_f16, long nblocks, unsigned char* __restrict__ q,
                                                              unsigned short* __restrict__ scm, unsigned* __restrict__ dd) {
    for (long blk = (long)blockIdx.x * blockDim.x + threadIdx.x; blk < nblocks; blk += (long)gridDim.x * blockDim.x) {
        float s[8], o[8];
        float smax = 0.0f, omax = 0.0f;
        for (int j = 0; j < 8; ++j) {
            float mn = 0.0f, mx = -INFINITY;
            for (int l = 0; l < 32; ++l) {
                const float v = w16_to_f32(w[blk * 256 + j * 32 + l], is_f16);
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
            }
            s[j] = (mx - mn) / 15.0f;
            o[j] = -mn;
            smax = fmaxf(smax, s[j]);
            omax = fmaxf(omax, o[j]);
        }
        const f16_t dh = (f16_t)(smax / 63.0f), dminh = (f16_t)(omax / 63.0f);
        const float df = (float)dh, dminf = (float)dminh;
        for (int j = 0; j < 8; ++j) {
            float sc = df > 0.0f ? floorf(s[j] / df + 0.5f) : 0.0f;
            float mm = dminf > 0.0f ? floorf(o[j] / dminf + 0.5f) : 0.0f;
            sc = fminf(fmaxf(sc, 0.0f), 63.0f);
            mm = fminf(fmaxf(mm, 0.0f), 63.0f);
            const float d1 = df * sc, m1 = dminf * mm;
            for (int l = 0; l < 32; ++l) {
                const float v = w16_to_f32(w[blk * 256 + j * 32 + l], is_f16);
                float qq = d1 > 0.0f ? floorf((v + m1) / d1 + 0.5f) : 0.0f;
                qq = fminf(fmaxf(qq, 0.0f), 15.0f);
                q[blk * 256 + j * 32 + l] = (unsigned char)qq;
            }
            scm[blk * 8 + j] = (unsigned short)((unsigned)sc | ((unsigned)mm << 8));
        }
        dd[blk] = (unsigned)__builtin_bit_cast(unsigned short, dh) | ((unsigned)__builtin_bit_cast(unsigned short, dminh) << 16);
    }
}
// dequantize_row_q4_K's arithmetic: (d * sc) * q - (dmin * m), two products and a subtraction in f32
__device__ __forceinline__ float q4k_value(unsigned dd, unsigned scm, unsigned q) {
    const f16x2 d2 = __builtin_bit_cast(f16x2, dd);
    const float d1 = (float)d2[0] * (float)(scm & 0xffu), m1 = (float)d2[1] * (float)(scm >> 8);
    return d1 * (float)q - m1;
}
__global__ __launch_bounds__(256) void lm_q4k_dequant_f32_kernel(const unsigned char* __restrict__ q, const unsigned short* __restrict__ scm,
                                                                 const unsigned* __restrict__ dd, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = q4k_value(dd[i >> 8], scm[i >> 5], q[i]);
}
// slot -> row of a packed matrix: plain, or (qkv_pairs) slots (2 s, 2 s + 1) = rows (d, d + 32) of one head
__host__ __device__ __forceinline__ long packed_slot_row(long slot, int qkv_pairs) {
    if (!qkv_pairs) return slot;
    const long pp = slot >> 1;
    return (pp >> 5) * 64 + (pp & 31) + 32 * (slot & 1);
}
// plain -> the quad-interleaved GEMV layout (GemvQ8 with dd)
__global__ __launch_bounds__(256) void lm_q4k_pack_kernel(const unsigned char* __restrict__ q, const unsigned short* __restrict__ scm, const unsigned* __restrict__ dd,
                                                          int N, int K, int qkv_pairs, u32x4* __restrict__ qs, unsigned short* __restrict__ oscm, unsigned* __restrict__ odd) {
    const long nchunk = K >> 3, nquad = (N + 3) >> 2, nkb = K >> 5, nk256 = K >> 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nquad * nchunk; i += (long)gridDim.x * blockDim.x) {
        const long quad = i / nchunk;
        const int c = (int)(i - quad * nchunk);
        unsigned dw[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const long slot = 4 * quad + sl, row = packed_slot_row(slot, qkv_pairs);
            unsigned v = 0;
            if (slot < N) {
                const uint2 b8 = *reinterpret_cast<const uint2*>(q + row * K + 8 * c);
                const unsigned bytes[2] = {b8.x, b8.y};
#pragma unroll
                for (int j = 0; j < 8; ++j) v |= ((bytes[j >> 2] >> (8 * (j & 3))) & 0xFu) << (4 * j);
                if ((c & 3) == 0) oscm[q4k_scm_index(slot, c >> 2, nkb)] = scm[row * nkb + (c >> 2)];
                if ((c & 31) == 0) odd[q4k_scm_index(slot, c >> 5, nk256)] = dd[row * nk256 + (c >> 5)];
            }
            dw[sl] = v;
        }
        qs[i] = u32x4{dw[0], dw[1], dw[2], dw[3]};
    }
}
// ---- Q6_K.  Plain form on the device: q [N][K] int8 (the 6-bit value - 32), sc [N][K / 16] int8, d [N][K / 256] fp16.
__global__ __launch_bounds__(256) void lm_q6k_unblock_kernel(const unsigned char* __restrict__ blocks, long nblocks, signed char* __restrict__ q,
                                                             signed char* __restrict__ sc, f16_t* __restrict__ d) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks * 256; i += (long)gridDim.x * blockDim.x) {
        const long blk = i >> 8;
        const int e = (int)(i & 255);
        const unsigned char* bp = blocks + blk * 210;
        const int n = e >> 7, j = (e >> 5) & 3, l = e & 31;          // weight 128 n + 32 j + l (dequantize_row_q6_K's q1..q4 = j 0..3)
        const unsigned char lb = bp[64 * n + 32 * (j & 1) + l];
        const unsigned lo = (j >> 1) ? (lb >> 4) : (lb & 0xF);
        const unsigned hi = (bp[128 + 32 * n + l] >> (2 * j)) & 3u;
        q[i] = (signed char)((int)(lo | (hi << 4)) - 32);
        if ((e & 15) == 0) sc[blk * 16 + (e >> 4)] = (signed char)bp[192 + (e >> 4)];
        if (e == 0) {
            const unsigned short bits = (unsigned short)bp[208] | ((unsigned short)bp[209] << 8);
            d[blk] = __builtin_bit_cast(f16_t, bits);
        }
    }
}
__global__ __launch_bounds__(256) void lm_q6k_dequant_f32_kernel(const signed char* __restrict__ q, const signed char* __restrict__ sc, const f16_t* __restrict__ d,
                                                                 float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = ((float)d[i >> 8] * (float)sc[i >> 4]) * (float)q[i];
}
// plain -> the q8_0 pair layout for the values + (s_a, s_b) f32 scales per pair and group of 16
__global__ __launch_bounds__(256) void lm_q6k_pack_kernel(const signed char* __restrict__ q, const signed char* __restrict__ sc, const f16_t* __restrict__ d,
                                                          int N, int K, int qkv_pairs, u32x4* __restrict__ qs, float* __restrict__ osc) {
    const long nchunk = K >> 3, npairs = N >> 1, nk16 = K >> 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs * nchunk; i += (long)gridDim.x * blockDim.x) {
        const long pp = i / nchunk;
        const int c = (int)(i - pp * nchunk);
        const long ra = qkv_pairs ? (pp >> 5) * 64 + (pp & 31) : 2 * pp;
        const long rb = qkv_pairs ? ra + 32 : ra + 1;
        const uint2 a = *reinterpret_cast<const uint2*>(q + ra * K + 8 * c);
        const uint2 b = *reinterpret_cast<const uint2*>(q + rb * K + 8 * c);
        qs[i] = u32x4{a.x, a.y, b.x, b.y};
        if ((c & 1) == 0) {
            const long k16 = c >> 1;
            float* o = osc + 2 * q8_sc_index(pp, k16, nk16);
            o[0] = (float)d[ra * (K >> 8) + (k16 >> 4)] * (float)sc[ra * nk16 + k16];
            o[1] = (float)d[rb * (K >> 8) + (k16 >> 4)] * (float)sc[rb * nk16 + k16];
        }
    }
}
// plain -> GemvQ8.  pair pp = rows (2pp, 2pp+1), or for the fused QKV matrix (qkv_pairs) rows (d, d+32) of one head.
__global__ __launch_bounds__(256) void lm_q8_pack_kernel(const signed char* __restrict__ q, const f16_t* __restrict__ d, int N, int K, int qkv_pairs,
                                                         u32x4* __restrict__ qs, unsigned* __restrict__ sc) {
    const long nchunk = K >> 3, npairs = N >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs * nchunk; i += (long)gridDim.x * blockDim.x) {
        const long pp = i / nchunk;
        const int c = (int)(i - pp * nchunk);
        const long ra = qkv_pairs ? (pp >> 5) * 64 + (pp & 31) : 2 * pp;
        const long rb = qkv_pairs ? ra + 32 : ra + 1;
        const uint2 a = *reinterpret_cast<const uint2*>(q + ra * K + 8 * c);
        const uint2 b = *reinterpret_cast<const uint2*>(q + rb * K + 8 * c);
        qs[i] = u32x4{a.x, a.y, b.x, b.y};
        if ((c & 3) == 0) {
            const unsigned short da = __builtin_bit_cast(unsigned short, d[ra * (K >> 5) + (c >> 2)]);
            const unsigned short db = __builtin_bit_cast(unsigned short, d[rb * (K >> 5) + (c >> 2)]);
            sc[q8_sc_index(pp, c >> 2, K >> 5)] = (unsigned)da | ((unsigned)db << 16);
        }
    }
}

// =============================================================================================
// One projection matrix [N][K] as it is kept in HBM: ONE copy, in the format it arrived in (or was asked for through
// rca_lm_config_t::decode_weights).
struct WMat {
    int fmt = WF_BF16;
    int N = 0, K = 0;
    bf16_t* w = nullptr;        // WF_BF16 / WF_F16: row-major 16-bit values
    u32x4* qs = nullptr;        // WF_Q8 / WF_Q4K (GemvQ8)
    unsigned* sc = nullptr;
    unsigned* dd = nullptr;     // WF_Q4K
    void release() {
        for (void* p : {(void*)w, (void*)qs, (void*)sc, (void*)dd})
            if (p) (void)hipFree(p);
        w = nullptr; qs = nullptr; sc = nullptr; dd = nullptr;
    }
    long stream_bytes() const {   // bytes one decode pass reads of it
        const long n = (long)N * K;
        return fmt == WF_Q8 ? n + n / 16 : (fmt == WF_Q4K ? n / 2 + n / 16 + n / 64 : (fmt == WF_Q6K ? n + n / 4 : 2 * n));
    }
};
struct LmLayer {
    WMat qkv, o, gu, down;
    WMat vseg;              // split_v: qkv holds [q; k] only and the V projection is a matrix of its own (formats differ)
    bool split_v = false;
    float *attn_norm = nullptr, *ffn_norm = nullptr;
};

struct rca_lm {
    int n_cus = 256;            // compute units of the device (grid shapes that want one workgroup per CU)
    rca_lm_config_t cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    void* embed = nullptr;      // [V][H] bf16 or f32 rows (embed_f32): the table keeps the precision it arrived in
    int embed_f32 = 0;
    WMat head;
    float* final_norm = nullptr;
    std::vector<LmLayer> layers;
    float *cos_t = nullptr, *sin_t = nullptr;
    f16_t *kc = nullptr, *vc = nullptr;  // [L][n_ctx_pad][nkv][hd]
    long kv_layer_stride = 0;
    int n_ctx_pad = 0, n_splits = 0;
    // activations
    float *x = nullptr, *xn = nullptr, *qkv = nullptr, *attn = nullptr, *hbuf = nullptr,
          *att_part = nullptr, *logits = nullptr, *probs_dev = nullptr;
    int* probe_ids_dev = nullptr;
    int* att_arrive = nullptr;   // per kv head: workgroups of the current attention launch that have published their partial
    bf16_t *xh = nullptr, *xl = nullptr;   // prefill: bf16 hi / lo split of the current GEMM input [LM_MAXM][max K]
    float* gpart = nullptr;                // prefill: k-split partial sums of the narrow projections [8][LM_MAXM][hidden]
    long logits_rows_cap = 0;   // rows allocated in `logits` (1, or more when logits_all)
    int logits_rows = 0;        // rows valid from the last eval
    LmDevState* stt = nullptr;  // device
    SamplerDev* samp = nullptr; // device
    SampWork* swork = nullptr;  // device: sampler histogram + candidate list
    LmDevState* h_stt = nullptr;   // pinned host staging (ids, n_tokens, m in; out_token back)
    int* h_probe = nullptr;        // pinned: [0, 64) probe ids in, [64, 128) their probabilities back (rca_lm_step_probe)
    int n_tokens = 0;           // host mirror (llama_cpp.Llama.n_tokens)
    bool sampler_set = false;
    bool samp_full = false;     // top_k <= 0 or > SAMP_MAXK: the whole-vocabulary (Gumbel-max) sampler launches instead of the top-k ones
    bool samp_patch = false, samp_big_k = false, samp_big_p = false;   // which other sampler launches a captured step holds (rca_lm_sampler_init)
    // captured steady-state steps (n = 1, 2)
    // decode-step graphs per (tokens 1..2, context bucket): bucket b launches min(n_splits, 4 << b) attention splits
    // Two sets: the handle's KV cache can be exchanged with a twin's (rca_lm_swap_kv) and the cache address is baked into the
    // captured kernel nodes, so a set remembers the cache it was captured over (at most two caches ever rotate through a handle).
    struct GraphSet { const f16_t* kc = nullptr; hipGraphExec_t g[3][LM_GRAPH_BUCKETS] = {}; hipGraphExec_t fg[LM_FRAME_MAX + 1][LM_GRAPH_BUCKETS] = {};
                      hipGraphExec_t gp[3][LM_GRAPH_BUCKETS] = {}; int gp_nprobe[3][LM_GRAPH_BUCKETS] = {};   // step + token probabilities (rca_lm_step_probe)
                      // rca_lm_eval_async passes of ONE prefill tile (the shadow cache's background tiles): tokens of the pass -> graph.  The
                      // flash attention of the tile path takes the context from the device state, so one graph serves every position.
                      int tile_m[2] = {0, 0}; hipGraphExec_t tile_g[2] = {nullptr, nullptr}; int tile_seen[2] = {0, 0};
                      unsigned long long last_use = 0; };
    GraphSet gset[2];
    unsigned long long gset_clock = 0;
    bool async_pending = false;   // an rca_lm_eval_async pass may still be running on the stream
    unsigned long long rng_host = 0;   // host mirror of the device's draw counter (restored when a frame graph is cut short)
    bool graphs_enabled = true;
    bool mfma_prefill = true;   // evals longer than LM_PREFILL_MIN tokens use the bf16 MFMA tiles
    bool fuse_attn = true;      // decode steps merge the attention splits inside the attention launch (rca_lm_set_attn_fuse)
    // weight sharing (rca_lm_create_shared): a borrower points at the handle that owns the weights and the RoPE tables; an owner
    // destroyed while borrowers are alive keeps those allocations (and its struct) until the last borrower is gone
    rca_lm* weights_of = nullptr;
    int borrowers = 0;
    bool zombie = false;
    struct DuplexState* duplex = nullptr;   // rca_duplex_frame: buffers + graphs of the one-replay frame
};
struct DuplexState;
static void duplex_destroy(DuplexState* d);
static void duplex_drop_graphs(DuplexState* d);

static int lm_alloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(RCA_ERR_HIP, "hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    return RCA_OK;
}

static void lm_free_weights(rca_lm* h) {
    for (auto& L : h->layers) {
        for (WMat* m : {&L.qkv, &L.o, &L.gu, &L.down, &L.vseg}) m->release();
        for (void* p : {(void*)L.attn_norm, (void*)L.ffn_norm})
            if (p) (void)hipFree(p);
    }
    h->head.release();
    for (void* p : {(void*)h->embed, (void*)h->final_norm, (void*)h->cos_t, (void*)h->sin_t})
        if (p) (void)hipFree(p);
    h->layers.clear();
    h->embed = nullptr;
    h->final_norm = h->cos_t = h->sin_t = nullptr;
}

// The captured step graphs carry the addresses of the logits buffer, the KV cache and the workspace in their kernel nodes:
// whenever one of those is reallocated (or the handle goes away) every captured graph has to go with it.
static void lm_drop_graph_set(rca_lm::GraphSet& gs) {
    for (int i = 0; i < 3; ++i)
        for (int b = 0; b < LM_GRAPH_BUCKETS; ++b) {
            if (gs.g[i][b]) { (void)hipGraphExecDestroy(gs.g[i][b]); gs.g[i][b] = nullptr; }
            if (gs.gp[i][b]) { (void)hipGraphExecDestroy(gs.gp[i][b]); gs.gp[i][b] = nullptr; }
        }
    for (int i = 0; i <= LM_FRAME_MAX; ++i)
        for (int b = 0; b < LM_GRAPH_BUCKETS; ++b)
            if (gs.fg[i][b]) { (void)hipGraphExecDestroy(gs.fg[i][b]); gs.fg[i][b] = nullptr; }
    for (int i = 0; i < 2; ++i) {
        if (gs.tile_g[i]) { (void)hipGraphExecDestroy(gs.tile_g[i]); gs.tile_g[i] = nullptr; }
        gs.tile_m[i] = gs.tile_seen[i] = 0;
    }
    gs.kc = nullptr;
}
static void lm_drop_graphs(rca_lm* h) {
    for (auto& gs : h->gset) lm_drop_graph_set(gs);
    duplex_drop_graphs(h->duplex);
}
// the graph set captured over the KV cache that is installed now (evicting the older set if neither matches)
static rca_lm::GraphSet& lm_graph_set(rca_lm* h) {
    rca_lm::GraphSet* hit = nullptr;
    for (auto& gs : h->gset)
        if (gs.kc == h->kc) hit = &gs;
    if (!hit) {
        hit = h->gset[0].last_use <= h->gset[1].last_use ? &h->gset[0] : &h->gset[1];
        lm_drop_graph_set(*hit);
        hit->kc = h->kc;
    }
    hit->last_use = ++h->gset_clock;
    return *hit;
}

extern "C" int rca_lm_destroy(rca_lm_t* h) {
    if (!h) return RCA_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    lm_drop_graphs(h);
    duplex_destroy(h->duplex);
    h->duplex = nullptr;
    for (void* p : {(void*)h->kc, (void*)h->vc, (void*)h->x, (void*)h->xn, (void*)h->qkv, (void*)h->attn, (void*)h->hbuf,
                    (void*)h->att_part, (void*)h->logits, (void*)h->probs_dev, (void*)h->probe_ids_dev, (void*)h->att_arrive, (void*)h->xh, (void*)h->xl,
                    (void*)h->gpart, (void*)h->stt, (void*)h->samp, (void*)h->swork})
        if (p) (void)hipFree(p);
    h->kc = h->vc = nullptr;
    h->x = h->xn = h->qkv = h->attn = h->hbuf = h->att_part = h->logits = h->probs_dev = h->gpart = nullptr;
    h->probe_ids_dev = nullptr; h->att_arrive = nullptr; h->xh = h->xl = nullptr; h->stt = nullptr; h->samp = nullptr; h->swork = nullptr;
    if (h->h_stt) { (void)hipHostFree(h->h_stt); h->h_stt = nullptr; }
    if (h->h_probe) { (void)hipHostFree(h->h_probe); h->h_probe = nullptr; }
    if (h->stream) { (void)hipStreamDestroy(h->stream); h->stream = nullptr; }
    if (h->weights_of) {            // borrower: the weights belong to someone else
        rca_lm* owner = h->weights_of;
        h->layers.clear();
        if (--owner->borrowers == 0 && owner->zombie) { lm_free_weights(owner); delete owner; }
        delete h;
        return RCA_OK;
    }
    if (h->borrowers > 0) {         // owner with live borrowers: keep the weights until the last one goes
        h->zombie = true;
        return RCA_OK;
    }
    lm_free_weights(h);
    delete h;
    return RCA_OK;
}

static int lm_check_cfg(const rca_lm_config_t* c) {
    if (!c) return fail(RCA_ERR_ARG, "null config");
    if (c->head_dim != 64) return fail(RCA_ERR_ARG, "head_dim must be 64 (got %d)", c->head_dim);
    if (c->n_kv_heads < 1 || c->n_heads % c->n_kv_heads) return fail(RCA_ERR_ARG, "n_heads must be a multiple of n_kv_heads");
    const int G = c->n_heads / c->n_kv_heads;
    if (G != 1 && G != 2 && G != 4) return fail(RCA_ERR_ARG, "query group size %d unsupported (1, 2, 4)", G);
    if (c->hidden % 8 || c->ffn % 8 || (c->n_heads * c->head_dim) % 8) return fail(RCA_ERR_ARG, "hidden/ffn must be multiples of 8");
    if (c->hidden > LM_KSLICE || c->n_heads * c->head_dim > LM_KSLICE) return fail(RCA_ERR_ARG, "hidden > %d unsupported", LM_KSLICE);
    if (c->ffn > LM_KSLICE * LM_MAXSPLIT) return fail(RCA_ERR_ARG, "ffn > %d unsupported", LM_KSLICE * LM_MAXSPLIT);
    if (c->ffn > LM_KSLICE && c->ffn % LM_KSLICE) return fail(RCA_ERR_ARG, "ffn above %d must be a multiple of it", LM_KSLICE);
    if (c->vocab_size < 2 || c->n_layers < 1 || c->n_ctx < 2) return fail(RCA_ERR_ARG, "bad sizes");
    if (c->decode_weights < 0 || c->decode_weights > 3) return fail(RCA_ERR_ARG, "decode_weights %d (0 as supplied, 1 q8_0, 2 f16, 3 q4_k)", c->decode_weights);
    return RCA_OK;
}

// A source tensor on the device in its "plain" form (row-major, one array per component), before fusion (q;k;v rows, gate/up rows
// interleaved) and before the GEMV layout is built.
struct RawMat {
    int fmt = WF_BF16;
    long rows = 0, cols = 0;
    bf16_t* w16 = nullptr;      // WF_BF16 / WF_F16
    signed char* q = nullptr;   // WF_Q8: int8 [rows][cols]; WF_Q4K: one byte per 4-bit value
    f16_t* d = nullptr;         // WF_Q8: fp16 [rows][cols / 32]; WF_Q4K: ushort (sc | m << 8) [rows][cols / 32]
    unsigned* dd = nullptr;     // WF_Q4K: (d | dmin << 16) [rows][cols / 256]
    void release() {
        for (void* p : {(void*)w16, (void*)q, (void*)d, (void*)dd})
            if (p) (void)hipFree(p);
        w16 = nullptr; q = nullptr; d = nullptr; dd = nullptr;
    }
    // (array, bytes per row) of every component
    int parts(void** ptr, long* row_bytes) const {
        if (fmt == WF_Q4K) { ptr[0] = q; row_bytes[0] = cols; ptr[1] = d; row_bytes[1] = cols / 32 * 2; ptr[2] = dd; row_bytes[2] = cols / 256 * 4; return 3; }
        if (fmt == WF_Q6K) { ptr[0] = q; row_bytes[0] = cols; ptr[1] = d; row_bytes[1] = cols / 16; ptr[2] = dd; row_bytes[2] = cols / 256 * 2; return 3; }   // d: int8 scales, dd: fp16 d
        if (fmt == WF_Q8) { ptr[0] = q; row_bytes[0] = cols; ptr[1] = d; row_bytes[1] = cols / 32 * 2; return 2; }
        ptr[0] = w16; row_bytes[0] = cols * 2;
        return 1;
    }
    int alloc(int f, long r, long c) {
        fmt = f; rows = r; cols = c;
        int rc;
        if (f == WF_Q4K) {
            if (c % 256) return fail(RCA_ERR_ARG, "Q4_K needs rows of a multiple of 256 values (got %ld)", c);
            if ((rc = lm_alloc((void**)&q, (size_t)r * c)) != RCA_OK || (rc = lm_alloc((void**)&d, (size_t)r * (c / 32) * 2)) != RCA_OK ||
                (rc = lm_alloc((void**)&dd, (size_t)r * (c / 256) * 4)) != RCA_OK) { release(); return rc; }
            return RCA_OK;
        }
        if (f == WF_Q6K) {
            if (c % 256) return fail(RCA_ERR_ARG, "Q6_K needs rows of a multiple of 256 values (got %ld)", c);
            if ((rc = lm_alloc((void**)&q, (size_t)r * c)) != RCA_OK || (rc = lm_alloc((void**)&d, (size_t)r * (c / 16))) != RCA_OK ||
                (rc = lm_alloc((void**)&dd, (size_t)r * (c / 256) * 2)) != RCA_OK) { release(); return rc; }
            return RCA_OK;
        }
        if (f == WF_Q8) {
            if (c % 32) return fail(RCA_ERR_ARG, "q8_0 needs rows of a multiple of 32 values (got %ld)", c);
            if ((rc = lm_alloc((void**)&q, (size_t)r * c)) != RCA_OK || (rc = lm_alloc((void**)&d, (size_t)r * (c / 32) * 2)) != RCA_OK) { release(); return rc; }
            return RCA_OK;
        }
        return lm_alloc((void**)&w16, (size_t)r * c * 2);
    }
};

// Uploads a matrix.  RCA_BF16 / RCA_F16 stay what they are; RCA_F32 is rounded to bf16 (nearest even: this build's storage format
// for 32-bit checkpoints); RCA_Q8_0 (the 34-byte GGUF blocks as they sit in the file) is unblocked.
static int lm_upload_raw(rca_lm* h, const rca_tensor_t* ts, int nt, const std::string& name, long rows, long cols, RawMat* out) {
    const rca_tensor_t* t = find_tensor(ts, nt, name);
    const long numel = rows * cols;
    if (!t) return fail(RCA_ERR_MISSING, "tensor '%s' missing", name.c_str());
    if (t->numel != numel) return fail(RCA_ERR_ARG, "tensor '%s': numel %ld, expected %ld", name.c_str(), (long)t->numel, numel);
    int rc;
    if (t->dtype == RCA_Q8_0) {
        if ((rc = out->alloc(WF_Q8, rows, cols)) != RCA_OK) return rc;
        const long nblk = numel / 32;
        unsigned char* raw = nullptr;
        if ((rc = lm_alloc((void**)&raw, (size_t)nblk * 34)) != RCA_OK) { out->release(); return rc; }
        hipError_t e = hipMemcpy(raw, t->data, (size_t)nblk * 34, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            lm_q8_unblock_kernel<<<4096, 256, 0, h->stream>>>(raw, nblk, out->q, out->d);
            e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(raw);
        if (e != hipSuccess) { out->release(); return fail(RCA_ERR_HIP, "upload '%s': %s", name.c_str(), hipGetErrorString(e)); }
        return RCA_OK;
    }
    if (t->dtype == RCA_Q6_K) {
        if ((rc = out->alloc(WF_Q6K, rows, cols)) != RCA_OK) return rc;
        const long nblk = numel / 256;
        unsigned char* raw = nullptr;
        if ((rc = lm_alloc((void**)&raw, (size_t)nblk * 210)) != RCA_OK) { out->release(); return rc; }
        hipError_t e = hipMemcpy(raw, t->data, (size_t)nblk * 210, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            lm_q6k_unblock_kernel<<<4096, 256, 0, h->stream>>>(raw, nblk, out->q, (signed char*)out->d, (f16_t*)out->dd);
            e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(raw);
        if (e != hipSuccess) { out->release(); return fail(RCA_ERR_HIP, "upload '%s': %s", name.c_str(), hipGetErrorString(e)); }
        return RCA_OK;
    }
    if (t->dtype == RCA_Q4_K) {
        if ((rc = out->alloc(WF_Q4K, rows, cols)) != RCA_OK) return rc;
        const long nblk = numel / 256;
        unsigned char* raw = nullptr;
        if ((rc = lm_alloc((void**)&raw, (size_t)nblk * 144)) != RCA_OK) { out->release(); return rc; }
        hipError_t e = hipMemcpy(raw, t->data, (size_t)nblk * 144, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            lm_q4k_unblock_kernel<<<4096, 256, 0, h->stream>>>(raw, nblk, (unsigned char*)out->q, (unsigned short*)out->d, out->dd);
            e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(raw);
        if (e != hipSuccess) { out->release(); return fail(RCA_ERR_HIP, "upload '%s': %s", name.c_str(), hipGetErrorString(e)); }
        return RCA_OK;
    }
    if (t->dtype == RCA_BF16 || t->dtype == RCA_F16) {
        if ((rc = out->alloc(t->dtype == RCA_F16 ? WF_F16 : WF_BF16, rows, cols)) != RCA_OK) return rc;
        RCA_HIP(hipMemcpy(out->w16, t->data, (size_t)numel * 2, hipMemcpyHostToDevice));
        return RCA_OK;
    }
    if (t->dtype != RCA_F32) return fail(RCA_ERR_ARG, "tensor '%s': dtype %d is not supported for a projection matrix", name.c_str(), t->dtype);
    if ((rc = out->alloc(WF_BF16, rows, cols)) != RCA_OK) return rc;
    float* tmp = nullptr;
    if ((rc = lm_alloc((void**)&tmp, (size_t)numel * 4)) != RCA_OK) return rc;
    hipError_t e = hipMemcpy(tmp, t->data, (size_t)numel * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        lm_f32_to_bf16_kernel<<<2048, 256, 0, h->stream>>>(tmp, out->w16, numel);
        e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(RCA_ERR_HIP, "upload '%s': %s", name.c_str(), hipGetErrorString(e));
    return RCA_OK;
}
// The embedding table: never streamed, so it is kept exact -- f32 rows for RCA_F32 / RCA_F16 / RCA_Q8_0 sources, bf16 rows for RCA_BF16.
static int lm_upload_embed(rca_lm* h, const rca_tensor_t* ts, int nt, const std::string& name, long rows, long cols) {
    const rca_tensor_t* t = find_tensor(ts, nt, name);
    const long numel = rows * cols;
    if (!t) return fail(RCA_ERR_MISSING, "tensor '%s' missing", name.c_str());
    if (t->numel != numel) return fail(RCA_ERR_ARG, "tensor '%s': numel %ld, expected %ld", name.c_str(), (long)t->numel, numel);
    int rc;
    if (t->dtype == RCA_BF16) {
        h->embed_f32 = 0;
        if ((rc = lm_alloc(&h->embed, (size_t)numel * 2)) != RCA_OK) return rc;
        RCA_HIP(hipMemcpy(h->embed, t->data, (size_t)numel * 2, hipMemcpyHostToDevice));
        return RCA_OK;
    }
    h->embed_f32 = 1;
    if ((rc = lm_alloc(&h->embed, (size_t)numel * 4)) != RCA_OK) return rc;
    if (t->dtype == RCA_F32) {
        RCA_HIP(hipMemcpy(h->embed, t->data, (size_t)numel * 4, hipMemcpyHostToDevice));
        return RCA_OK;
    }
    if (t->dtype == RCA_F16) {
        f16_t* tmp = nullptr;
        if ((rc = lm_alloc((void**)&tmp, (size_t)numel * 2)) != RCA_OK) return rc;
        hipError_t e = hipMemcpy(tmp, t->data, (size_t)numel * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            lm_f16_to_f32_kernel<<<4096, 256, 0, h->stream>>>(tmp, (float*)h->embed, numel);
            e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "upload '%s': %s", name.c_str(), hipGetErrorString(e));
        return RCA_OK;
    }
    if (t->dtype == RCA_Q8_0 || t->dtype == RCA_Q4_K || t->dtype == RCA_Q6_K) {
        RawMat raw;
        if ((rc = lm_upload_raw(h, ts, nt, name, rows, cols, &raw)) != RCA_OK) return rc;
        if (raw.fmt == WF_Q6K) lm_q6k_dequant_f32_kernel<<<4096, 256, 0, h->stream>>>(raw.q, (const signed char*)raw.d, (const f16_t*)raw.dd, (float*)h->embed, numel);
        else if (raw.fmt == WF_Q4K) lm_q4k_dequant_f32_kernel<<<4096, 256, 0, h->stream>>>((const unsigned char*)raw.q, (const unsigned short*)raw.d, raw.dd, (float*)h->embed, numel);
        else lm_q8_dequant_f32_kernel<<<4096, 256, 0, h->stream>>>(raw.q, raw.d, (float*)h->embed, numel);
        hipError_t e = hipStreamSynchronize(h->stream);
        raw.release();
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "upload '%s': %s", name.c_str(), hipGetErrorString(e));
        return RCA_OK;
    }
    return fail(RCA_ERR_ARG, "tensor '%s': dtype %d is not supported for the embedding table", name.c_str(), t->dtype);
}
static int lm_upload_f32(const rca_tensor_t* ts, int nt, const std::string& name, long numel, float** out) {
    const rca_tensor_t* t = find_tensor(ts, nt, name);
    if (!t) return fail(RCA_ERR_MISSING, "tensor '%s' missing", name.c_str());
    if (t->numel != numel) return fail(RCA_ERR_ARG, "tensor '%s': numel %ld, expected %ld", name.c_str(), (long)t->numel, numel);
    int rc;
    if ((rc = lm_alloc((void**)out, (size_t)numel * 4)) != RCA_OK) return rc;
    if (t->dtype == RCA_BF16) {  // small tensors (norm weights): widen on the host
        std::vector<float> tmp((size_t)numel);
        const bf16_t* src = (const bf16_t*)t->data;
        for (long i = 0; i < numel; ++i) {
            const unsigned u = (unsigned)src[i] << 16;
            memcpy(&tmp[i], &u, 4);
        }
        RCA_HIP(hipMemcpy(*out, tmp.data(), (size_t)numel * 4, hipMemcpyHostToDevice));
    } else if (t->dtype == RCA_F32) {
        RCA_HIP(hipMemcpy(*out, t->data, (size_t)numel * 4, hipMemcpyHostToDevice));
    } else {
        return fail(RCA_ERR_ARG, "tensor '%s': norm weights must be f32 or bf16", name.c_str());
    }
    return RCA_OK;
}

// rca_lm_config_t::decode_weights applied to one plain matrix: 1 = quantise to q8_0 the way llama-quantize does, 2 = fp16.
// A matrix that ARRIVED quantised stays what it is.
static int lm_raw_convert(rca_lm* h, RawMat* m, int want) {
    if (want == 0 || m->fmt == WF_Q8 || m->fmt == WF_Q4K || m->fmt == WF_Q6K) return RCA_OK;
    int rc;
    if (want == 3) {
        RawMat qd;
        if ((rc = qd.alloc(WF_Q4K, m->rows, m->cols)) != RCA_OK) return rc;
        lm_q4k_quantize_kernel<<<4096, 256, 0, h->stream>>>(m->w16, m->fmt == WF_F16, m->rows * m->cols / 256, (unsigned char*)qd.q, (unsigned short*)qd.d, qd.dd);
        hipError_t e = hipStreamSynchronize(h->stream);
        m->release();
        if (e != hipSuccess) { qd.release(); return fail(RCA_ERR_HIP, "Q4_K quantise: %s", hipGetErrorString(e)); }
        *m = qd;
        return RCA_OK;
    }
    if (want == 1) {
        RawMat qd;
        if ((rc = qd.alloc(WF_Q8, m->rows, m->cols)) != RCA_OK) return rc;
        lm_q8_quantize_kernel<<<4096, 256, 0, h->stream>>>(m->w16, m->fmt == WF_F16, m->rows * m->cols / 32, qd.q, qd.d);
        hipError_t e = hipStreamSynchronize(h->stream);
        m->release();
        if (e != hipSuccess) { qd.release(); return fail(RCA_ERR_HIP, "q8_0 quantise: %s", hipGetErrorString(e)); }
        *m = qd;
        return RCA_OK;
    }
    if (want == 2 && m->fmt == WF_BF16) {
        f16_t* out = nullptr;
        if ((rc = lm_alloc((void**)&out, (size_t)m->rows * m->cols * 2)) != RCA_OK) return rc;
        lm_bf16_to_f16_kernel<<<4096, 256, 0, h->stream>>>(m->w16, out, m->rows * m->cols);
        hipError_t e = hipStreamSynchronize(h->stream);
        (void)hipFree(m->w16);
        m->w16 = reinterpret_cast<bf16_t*>(out);
        m->fmt = WF_F16;
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "fp16 conversion: %s", hipGetErrorString(e));
    }
    return RCA_OK;
}
// rows of `parts` stacked (q; k; v).  Parts of different formats cannot be fused.
static int lm_raw_concat(rca_lm* h, std::vector<RawMat*> parts, RawMat* out, const char* what) {
    long rows = 0;
    for (RawMat* p : parts) {
        if (p->fmt != parts[0]->fmt || p->cols != parts[0]->cols) return fail(RCA_ERR_ARG, "%s: its parts arrived in different formats", what);
        rows += p->rows;
    }
    int rc;
    if ((rc = out->alloc(parts[0]->fmt, rows, parts[0]->cols)) != RCA_OK) return rc;
    void* dp[3]; long drb[3];
    const int np = out->parts(dp, drb);
    long r0 = 0;
    for (RawMat* p : parts) {
        void* sp[3]; long srb[3];
        p->parts(sp, srb);
        for (int i = 0; i < np; ++i)
            RCA_HIP(hipMemcpyAsync((char*)dp[i] + r0 * drb[i], sp[i], (size_t)p->rows * srb[i], hipMemcpyDeviceToDevice, h->stream));
        r0 += p->rows;
    }
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}
// dst row 2i = a row i, 2i+1 = b row i (gate/up: SwiGLU becomes an epilogue)
static int lm_raw_interleave(rca_lm* h, RawMat* a, RawMat* b, RawMat* out, const char* what) {
    if (a->fmt != b->fmt || a->cols != b->cols || a->rows != b->rows) return fail(RCA_ERR_ARG, "%s: its parts arrived in different formats", what);
    int rc;
    if ((rc = out->alloc(a->fmt, 2 * a->rows, a->cols)) != RCA_OK) return rc;
    void *dp[3], *ap[3], *bp[3]; long rb[3];
    const int np = out->parts(dp, rb);
    a->parts(ap, rb);
    b->parts(bp, rb);
    for (int i = 0; i < np; ++i)
        lm_interleave_rows_bytes_kernel<<<4096, 256, 0, h->stream>>>((const unsigned char*)ap[i], (const unsigned char*)bp[i], (unsigned char*)dp[i], a->rows, rb[i]);
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}
// plain -> the layout the decode GEMV streams.  Takes ownership of `raw` (released or moved into `out`).
static int lm_finish_mat(rca_lm* h, RawMat* raw, int qkv_pairs, WMat* out, const char* what) {
    out->fmt = raw->fmt; out->N = (int)raw->rows; out->K = (int)raw->cols;
    if (raw->fmt == WF_Q4K) {
        const int N = out->N, K = out->K;
        if ((K % 256) || (N % 4) || (qkv_pairs && (N % 64))) { raw->release(); return fail(RCA_ERR_ARG, "Q4_K weights: %s is %d x %d (K must be a multiple of 256, N of 4)", what, N, K); }
        int rc;
        const long nquad = N / 4, nchunk = K / 8, g16 = (N + 15) / 16;
        const size_t scm_bytes = (size_t)g16 * (K / 32) * 16 * 2, dd_bytes = (size_t)g16 * (K / 256) * 16 * 4;
        if ((rc = lm_alloc((void**)&out->qs, (size_t)nquad * nchunk * 16)) != RCA_OK || (rc = lm_alloc((void**)&out->sc, scm_bytes)) != RCA_OK ||
            (rc = lm_alloc((void**)&out->dd, dd_bytes)) != RCA_OK) { raw->release(); return rc; }
        (void)hipMemsetAsync(out->sc, 0, scm_bytes, h->stream);
        (void)hipMemsetAsync(out->dd, 0, dd_bytes, h->stream);
        lm_q4k_pack_kernel<<<4096, 256, 0, h->stream>>>((const unsigned char*)raw->q, (const unsigned short*)raw->d, raw->dd, N, K, qkv_pairs, out->qs,
                                                        (unsigned short*)out->sc, out->dd);
        hipError_t e = hipStreamSynchronize(h->stream);
        raw->release();
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "Q4_K pack of %s: %s", what, hipGetErrorString(e));
        return RCA_OK;
    }
    if (raw->fmt == WF_Q6K) {
        const int N = out->N, K = out->K;
        if ((K % 256) || (N % 2) || (qkv_pairs && (N % 64))) { raw->release(); return fail(RCA_ERR_ARG, "Q6_K weights: %s is %d x %d (K must be a multiple of 256, N even)", what, N, K); }
        int rc;
        const long npairs = N / 2, nchunk = K / 8;
        const size_t sc_bytes = (size_t)((npairs + 7) / 8) * (K / 16) * 8 * 2 * 4;
        if ((rc = lm_alloc((void**)&out->qs, (size_t)npairs * nchunk * 16)) != RCA_OK || (rc = lm_alloc((void**)&out->sc, sc_bytes)) != RCA_OK) { raw->release(); return rc; }
        (void)hipMemsetAsync(out->sc, 0, sc_bytes, h->stream);
        lm_q6k_pack_kernel<<<4096, 256, 0, h->stream>>>(raw->q, (const signed char*)raw->d, (const f16_t*)raw->dd, N, K, qkv_pairs, out->qs, (float*)out->sc);
        hipError_t e = hipStreamSynchronize(h->stream);
        raw->release();
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "Q6_K pack of %s: %s", what, hipGetErrorString(e));
        return RCA_OK;
    }
    if (raw->fmt != WF_Q8) {
        out->w = raw->w16;
        raw->w16 = nullptr;
        return RCA_OK;
    }
    const int N = out->N, K = out->K;
    if ((K % 32) || (N % 2) || (qkv_pairs && (N % 64))) { raw->release(); return fail(RCA_ERR_ARG, "q8_0 weights: %s is %d x %d (K must be a multiple of 32, N even)", what, N, K); }
    int rc;
    const long npairs = N / 2, nchunk = K / 8;
    if ((rc = lm_alloc((void**)&out->qs, (size_t)npairs * nchunk * 16)) != RCA_OK ||
        (rc = lm_alloc((void**)&out->sc, (size_t)((npairs + 7) / 8) * (K / 32) * 8 * 4)) != RCA_OK) { raw->release(); return rc; }   // groups of 8 pairs, zero padded
    (void)hipMemsetAsync(out->sc, 0, (size_t)((npairs + 7) / 8) * (K / 32) * 8 * 4, h->stream);
    lm_q8_pack_kernel<<<4096, 256, 0, h->stream>>>(raw->q, raw->d, N, K, qkv_pairs, out->qs, out->sc);
    hipError_t e = hipStreamSynchronize(h->stream);
    raw->release();
    if (e != hipSuccess) return fail(RCA_ERR_HIP, "q8_0 pack of %s: %s", what, hipGetErrorString(e));
    return RCA_OK;
}

static int g128_splits(int N, int K);
static bool lm_can_gemm128(const rca_lm* h);
static int lm_common_init(rca_lm* h, const rca_tensor_t* ts, int nt, const rca_lm* rope_src = nullptr) {
    const rca_lm_config_t& c = h->cfg;
    int rc;
    const int half = c.head_dim / 2;
    h->n_ctx_pad = (c.n_ctx + ATT_KEYS - 1) / ATT_KEYS * ATT_KEYS;
    h->n_splits = h->n_ctx_pad / ATT_KEYS;
    if (rope_src) {   // borrower: the owner's tables cover at least this context (checked by the caller)
        h->cos_t = rope_src->cos_t;
        h->sin_t = rope_src->sin_t;
    } else {
    // RoPE tables from inv_freq (supplied by the host layer exactly as HF computes it, or derived here)
    std::vector<float> inv(half);
    const rca_tensor_t* tinv = find_tensor(ts, nt, "rope.inv_freq");
    if (tinv) {
        if (tinv->numel != half || tinv->dtype != RCA_F32) return fail(RCA_ERR_ARG, "rope.inv_freq must hold %d f32", half);
        memcpy(inv.data(), tinv->data, half * sizeof(float));
    } else {
        for (int i = 0; i < half; ++i) {
            double f = 1.0 / pow((double)c.rope_theta, (double)(2 * i) / (double)c.head_dim);
            if (c.rope_scaling == 1) {
                const double low_wl = (double)c.rope_orig_ctx / c.rope_low_freq_factor;
                const double high_wl = (double)c.rope_orig_ctx / c.rope_high_freq_factor;
                const double wl = 2.0 * M_PI / f;
                if (wl > low_wl) f = f / c.rope_factor;
                else if (wl >= high_wl) {
                    const double smooth = ((double)c.rope_orig_ctx / wl - c.rope_low_freq_factor) / (c.rope_high_freq_factor - c.rope_low_freq_factor);
                    f = (1.0 - smooth) * f / c.rope_factor + smooth * f;
                }
            }
            inv[i] = (float)f;
        }
    }
    float* inv_dev = nullptr;
    if ((rc = lm_alloc((void**)&inv_dev, half * 4)) != RCA_OK) return rc;
    RCA_HIP(hipMemcpy(inv_dev, inv.data(), half * 4, hipMemcpyHostToDevice));
    if ((rc = lm_alloc((void**)&h->cos_t, (size_t)h->n_ctx_pad * half * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->sin_t, (size_t)h->n_ctx_pad * half * 4)) != RCA_OK) return rc;
    lm_rope_table_kernel<<<cdiv((long)h->n_ctx_pad * half, 256), 256, 0, h->stream>>>(inv_dev, h->cos_t, h->sin_t, h->n_ctx_pad, half);
    RCA_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(inv_dev);
    }
    // KV cache
    h->kv_layer_stride = (long)h->n_ctx_pad * c.n_kv_heads * c.head_dim;
    const size_t kvb = (size_t)c.n_layers * h->kv_layer_stride * 2;
    if ((rc = lm_alloc((void**)&h->kc, kvb)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->vc, kvb)) != RCA_OK) return rc;
    RCA_HIP(hipMemsetAsync(h->kc, 0, kvb, h->stream));
    RCA_HIP(hipMemsetAsync(h->vc, 0, kvb, h->stream));
    // activations
    const int H = c.hidden, QKV = (c.n_heads + 2 * c.n_kv_heads) * c.head_dim, AO = c.n_heads * c.head_dim;
    if ((rc = lm_alloc((void**)&h->x, (size_t)LM_MAXM * H * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->xn, (size_t)LM_MAXM * H * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->qkv, (size_t)LM_MAXM * QKV * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->attn, (size_t)LM_MAXM * AO * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->hbuf, (size_t)LM_MAXM * c.ffn * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->att_part, (size_t)(LM_MAXM / 2) * c.n_kv_heads * h->n_splits * 8 * 66 * 4)) != RCA_OK) return rc;
    h->logits_rows_cap = c.logits_all ? 64 : 1;
    if ((rc = lm_alloc((void**)&h->logits, (size_t)h->logits_rows_cap * c.vocab_size * 4)) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->probs_dev, (64 + 2 * PROBS_SLICES) * 4)) != RCA_OK) return rc;   // [64 probs][slice (max, sum) pairs]
    if ((rc = lm_alloc((void**)&h->probe_ids_dev, 64 * 4)) != RCA_OK) return rc;
    {   // in-launch attention merge: 64 launch counters (tags start at 1: the zeroed granules match nothing), then the granules
        const size_t gb = ATT_EPOCH_INTS * 4 + (size_t)c.n_kv_heads * h->n_splits * 8 * 66 * 8;
        if ((rc = lm_alloc((void**)&h->att_arrive, gb)) != RCA_OK) return rc;
        RCA_HIP(hipMemsetAsync(h->att_arrive, 0, gb, h->stream));
        int ones[ATT_EPOCH_INTS];
        for (int i = 0; i < ATT_EPOCH_INTS; ++i) ones[i] = 1;
        RCA_HIP(hipMemcpyAsync(h->att_arrive, ones, sizeof(ones), hipMemcpyHostToDevice, h->stream));
        RCA_HIP(hipStreamSynchronize(h->stream));
    }
    {
        const size_t kmax = (size_t)std::max(std::max(H, AO), c.ffn);
        if ((rc = lm_alloc((void**)&h->xh, (size_t)LM_MAXM * kmax * 2)) != RCA_OK) return rc;
        if ((rc = lm_alloc((void**)&h->xl, (size_t)LM_MAXM * kmax * 2)) != RCA_OK) return rc;
        RCA_HIP(hipMemsetAsync(h->xh, 0, (size_t)LM_MAXM * kmax * 2, h->stream));
        RCA_HIP(hipMemsetAsync(h->xl, 0, (size_t)LM_MAXM * kmax * 2, h->stream));
        // k-split partial sums [split][N][LM_MAXM] of the 128-row GEMMs (lm_enqueue_prefill_tile128); other models never touch the buffer
        size_t part_rows = 64;   // largest split count x N over the four projections
        if (lm_can_gemm128(h)) {
            const int Fq = c.ffn;
            const int nk[4][2] = {{QKV, H}, {H, AO}, {2 * Fq, H}, {H, Fq}};
            for (int i = 0; i < 4; ++i) {
                const int ns = g128_splits(nk[i][0], nk[i][1]);
                if (ns > 1) part_rows = std::max(part_rows, (size_t)ns * nk[i][0]);
            }
        }
        if ((rc = lm_alloc((void**)&h->gpart, part_rows * LM_MAXM * 4)) != RCA_OK) return rc;
    }
    if ((rc = lm_alloc((void**)&h->stt, sizeof(LmDevState))) != RCA_OK) return rc;
    if ((rc = lm_alloc((void**)&h->samp, sizeof(SamplerDev))) != RCA_OK) return rc;
    RCA_HIP(hipMemsetAsync(h->stt, 0, sizeof(LmDevState), h->stream));
    RCA_HIP(hipMemsetAsync(h->samp, 0, sizeof(SamplerDev), h->stream));
    if ((rc = lm_alloc((void**)&h->swork, sizeof(SampWork))) != RCA_OK) return rc;
    RCA_HIP(hipMemsetAsync(h->swork, 0, sizeof(SampWork), h->stream));
    RCA_HIP(hipHostMalloc((void**)&h->h_stt, sizeof(LmDevState), hipHostMallocDefault));
    memset(h->h_stt, 0, sizeof(LmDevState));
    RCA_HIP(hipHostMalloc((void**)&h->h_probe, 128 * 4, hipHostMallocDefault));   // at creation: a pinned allocation inside a frame costs milliseconds
    memset(h->h_probe, 0, 128 * 4);
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}

static int lm_new(const rca_lm_config_t* cfg, int device, rca_lm** out) {
    int rc;
    if ((rc = lm_check_cfg(cfg)) != RCA_OK) return rc;
    RCA_HIP(hipSetDevice(device));
    rca_lm* h = new rca_lm();
    h->cfg = *cfg;
    h->device = device;
    if (hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->n_cus <= 0) h->n_cus = 256;
    h->layers.resize(cfg->n_layers);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(RCA_ERR_HIP, "stream create");
    }
    *out = h;
    return RCA_OK;
}

// Fusions + GEMV layouts of one layer from its seven plain matrices (which are consumed).
static int lm_build_layer(rca_lm* h, LmLayer& L, RawMat& q, RawMat& k, RawMat& v, RawMat& o, RawMat& g, RawMat& u, RawMat& dn) {
    const int want = h->cfg.decode_weights;
    int rc = RCA_OK;
    RawMat qkv, gu;
    auto done = [&](int code) { for (RawMat* m : {&q, &k, &v, &o, &g, &u, &dn, &qkv, &gu}) m->release(); return code; };
    for (RawMat* m : {&q, &k, &v, &o, &g, &u, &dn})
        if ((rc = lm_raw_convert(h, m, want)) != RCA_OK) return done(rc);
    if (q.fmt == k.fmt && k.fmt != v.fmt) {   // llama-quantize Q4_K_M: attn_v is Q6_K in the layers use_more_bits() picks, attn_q / attn_k Q4_K
        if ((rc = lm_raw_concat(h, {&q, &k}, &qkv, "the fused QK projection")) != RCA_OK) return done(rc);
        if ((rc = lm_finish_mat(h, &v, 1, &L.vseg, "v_proj")) != RCA_OK) return done(rc);
        L.split_v = true;
    } else if ((rc = lm_raw_concat(h, {&q, &k, &v}, &qkv, "the fused QKV projection")) != RCA_OK) return done(rc);
    q.release(); k.release(); v.release();
    if ((rc = lm_raw_interleave(h, &g, &u, &gu, "the fused gate/up projection")) != RCA_OK) return done(rc);
    g.release(); u.release();
    if ((rc = lm_finish_mat(h, &qkv, 1, &L.qkv, "the fused QKV projection")) != RCA_OK) return done(rc);
    if ((rc = lm_finish_mat(h, &o, 0, &L.o, "o_proj")) != RCA_OK) return done(rc);
    if ((rc = lm_finish_mat(h, &gu, 0, &L.gu, "the fused gate/up projection")) != RCA_OK) return done(rc);
    if ((rc = lm_finish_mat(h, &dn, 0, &L.down, "down_proj")) != RCA_OK) return done(rc);
    return done(RCA_OK);
}

extern "C" int rca_lm_create(const rca_lm_config_t* cfg, const rca_tensor_t* ts, int32_t nt, int32_t device, rca_lm_t** out) {
    if (!ts || !out) return fail(RCA_ERR_ARG, "null argument");
    rca_lm* h = nullptr;
    int rc;
    if ((rc = lm_new(cfg, device, &h)) != RCA_OK) return rc;
    auto bail = [&](int code) { rca_lm_destroy(h); return code; };
    const rca_lm_config_t& c = h->cfg;
    const long H = c.hidden, V = c.vocab_size, F = c.ffn, Q = (long)c.n_heads * c.head_dim, KVD = (long)c.n_kv_heads * c.head_dim;
    if ((rc = lm_upload_embed(h, ts, nt, "model.embed_tokens.weight", V, H)) != RCA_OK) return bail(rc);
    {
        RawMat ph;
        if ((rc = lm_upload_raw(h, ts, nt, "lm_head.weight", V, H, &ph)) != RCA_OK) return bail(rc);
        if ((rc = lm_raw_convert(h, &ph, c.decode_weights)) != RCA_OK) { ph.release(); return bail(rc); }
        if ((rc = lm_finish_mat(h, &ph, 0, &h->head, "lm_head")) != RCA_OK) return bail(rc);
    }
    if ((rc = lm_upload_f32(ts, nt, "model.norm.weight", H, &h->final_norm)) != RCA_OK) return bail(rc);
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = "model.layers." + std::to_string(l) + ".";
        LmLayer& L = h->layers[l];
        RawMat q, k, v, o, g, u, dn;
        if ((rc = lm_upload_raw(h, ts, nt, p + "self_attn.q_proj.weight", Q, H, &q)) != RCA_OK ||
            (rc = lm_upload_raw(h, ts, nt, p + "self_attn.k_proj.weight", KVD, H, &k)) != RCA_OK ||
            (rc = lm_upload_raw(h, ts, nt, p + "self_attn.v_proj.weight", KVD, H, &v)) != RCA_OK ||
            (rc = lm_upload_raw(h, ts, nt, p + "self_attn.o_proj.weight", H, Q, &o)) != RCA_OK ||
            (rc = lm_upload_raw(h, ts, nt, p + "mlp.gate_proj.weight", F, H, &g)) != RCA_OK ||
            (rc = lm_upload_raw(h, ts, nt, p + "mlp.up_proj.weight", F, H, &u)) != RCA_OK ||
            (rc = lm_upload_raw(h, ts, nt, p + "mlp.down_proj.weight", H, F, &dn)) != RCA_OK) {
            for (RawMat* m : {&q, &k, &v, &o, &g, &u, &dn}) m->release();
            return bail(rc);
        }
        if ((rc = lm_build_layer(h, L, q, k, v, o, g, u, dn)) != RCA_OK) return bail(rc);
        if ((rc = lm_upload_f32(ts, nt, p + "input_layernorm.weight", H, &L.attn_norm)) != RCA_OK) return bail(rc);
        if ((rc = lm_upload_f32(ts, nt, p + "post_attention_layernorm.weight", H, &L.ffn_norm)) != RCA_OK) return bail(rc);
    }
    if ((rc = lm_common_init(h, ts, nt)) != RCA_OK) return bail(rc);
    *out = h;
    return RCA_OK;
}

extern "C" int rca_lm_create_random(const rca_lm_config_t* cfg, uint64_t seed, float init_std, int32_t device, rca_lm_t** out) {
    if (!out) return fail(RCA_ERR_ARG, "null argument");
    rca_lm* h = nullptr;
    int rc;
    if ((rc = lm_new(cfg, device, &h)) != RCA_OK) return rc;
    auto bail = [&](int code) { rca_lm_destroy(h); return code; };
    const rca_lm_config_t& c = h->cfg;
    const long H = c.hidden, V = c.vocab_size, F = c.ffn, Q = (long)c.n_heads * c.head_dim, KVD = (long)c.n_kv_heads * c.head_dim;
    const float scale = init_std * 1.7320508f / 65535.0f;
    unsigned long long tid = 1;
    // every matrix is generated as bf16 values (hash of (seed, tensor id, element)), then brought into the requested decode format
    // the way a converter would from a bf16 checkpoint
    auto rnd = [&](long rows, long cols, int qkv_pairs, WMat* dst, const char* what) -> int {
        RawMat m;
        int r = m.alloc(WF_BF16, rows, cols);
        if (r != RCA_OK) return r;
        lm_random_bf16_kernel<<<4096, 256, 0, h->stream>>>(m.w16, rows * cols, seed, tid++, scale);
        if ((r = lm_raw_convert(h, &m, c.decode_weights)) != RCA_OK) { m.release(); return r; }
        return lm_finish_mat(h, &m, qkv_pairs, dst, what);
    };
    auto ones = [&](float** p, long n) -> int {
        int r = lm_alloc((void**)p, (size_t)n * 4);
        if (r != RCA_OK) return r;
        lm_fill_f32_kernel<<<cdiv(n, 256), 256, 0, h->stream>>>(*p, n, 1.0f);
        return RCA_OK;
    };
    // tensor ids: 1 embed, 2 head, then per layer 10*l + {3 qkv, 4 o, 5 gate/up(interleaved), 6 down}
    h->embed_f32 = 0;
    if ((rc = lm_alloc(&h->embed, (size_t)V * H * 2)) != RCA_OK) return bail(rc);
    lm_random_bf16_kernel<<<4096, 256, 0, h->stream>>>((bf16_t*)h->embed, V * H, seed, tid++, scale);
    if ((rc = rnd(V, H, 0, &h->head, "lm_head")) != RCA_OK || (rc = ones(&h->final_norm, H)) != RCA_OK) return bail(rc);
    for (int l = 0; l < c.n_layers; ++l) {
        LmLayer& L = h->layers[l];
        tid = 10ull * (l + 1) + 3;
        if ((rc = rnd(Q + 2 * KVD, H, 1, &L.qkv, "the fused QKV projection")) != RCA_OK || (rc = rnd(H, Q, 0, &L.o, "o_proj")) != RCA_OK ||
            (rc = rnd(2 * F, H, 0, &L.gu, "the fused gate/up projection")) != RCA_OK || (rc = rnd(H, F, 0, &L.down, "down_proj")) != RCA_OK ||
            (rc = ones(&L.attn_norm, H)) != RCA_OK || (rc = ones(&L.ffn_norm, H)) != RCA_OK)
            return bail(rc);
    }
    RCA_HIP(hipStreamSynchronize(h->stream));
    if ((rc = lm_common_init(h, nullptr, 0)) != RCA_OK) return bail(rc);
    *out = h;
    return RCA_OK;
}

// A second instance over the SAME weights (the reference keeps two llama.cpp models of one file: `llm` and the logits_all twin
// `aux_llm`, realtime_agent_resources.py:19-33): own KV cache, workspace, sampler, stream and graphs; weights and RoPE tables
// are the parent's.  Either handle may be destroyed first.  Calls that modify weights (rca_lm_mask_head_rows,
// rca_lm_persist_codec_embeddings) act on both.
extern "C" int rca_lm_create_shared(rca_lm_t* parent, int32_t n_ctx, int32_t logits_all, rca_lm_t** out) {
    if (!parent || !out) return fail(RCA_ERR_ARG, "null argument");
    rca_lm* owner = parent->weights_of ? parent->weights_of : parent;
    if (owner->zombie) return fail(RCA_ERR_STATE, "create_shared: the parent handle was destroyed");
    rca_lm_config_t cfg = parent->cfg;
    cfg.n_ctx = n_ctx;
    cfg.logits_all = logits_all;
    if (n_ctx < 2 || (n_ctx + ATT_KEYS - 1) / ATT_KEYS * ATT_KEYS > owner->n_ctx_pad)
        return fail(RCA_ERR_ARG, "create_shared: n_ctx %d exceeds the parent's RoPE tables (%d positions)", n_ctx, owner->n_ctx_pad);
    rca_lm* h = nullptr;
    int rc;
    if ((rc = lm_new(&cfg, parent->device, &h)) != RCA_OK) return rc;
    h->embed = owner->embed;
    h->embed_f32 = owner->embed_f32;
    h->head = owner->head;
    h->final_norm = owner->final_norm;
    h->layers = owner->layers;
    h->weights_of = owner;
    // the twin evaluates with the kernels its parent would use (the shadow cache must hold the bits recompute_kv_cache would leave)
    h->fuse_attn = parent->fuse_attn;
    h->mfma_prefill = parent->mfma_prefill;
    owner->borrowers++;
    if ((rc = lm_common_init(h, nullptr, 0, owner)) != RCA_OK) { rca_lm_destroy(h); return rc; }
    *out = h;
    return RCA_OK;
}

// ------------------------------------------------------------------------- forward pass (M tokens)
// Launch geometry of the decode GEMVs: R rows per batch (weights of a whole batch are in flight per wave before any is
// consumed) and batches per workgroup.  Defaults are sized for the 256 CUs (>= 2 workgroups each, all resident at once);
// RCA_GEMV_<QKV|O|GU|DOWN|HEAD>="R,batches" overrides one for tuning runs (scripts/lm_gemv_sweep.sh).  Results do not
// depend on the choice.
struct GemvGeom { int R, bpw; };
enum { GEMV_QKV = 0, GEMV_O, GEMV_GU, GEMV_DOWN, GEMV_HEAD, GEMV_KINDS };
static GemvGeom gemv_geom(int kind, int N, bool q8) {
    static GemvGeom tab[2][GEMV_KINDS];
    static bool init = false;
    if (!init) {
        // measured (scripts/lm_gemv_sweep.sh, profiles/r02): the same geometry serves both formats; a q8_0 batch of R rows is R / 2 loads
        const GemvGeom def[2][GEMV_KINDS] = {{{4, 1}, {4, 1}, {16, 2}, {4, 1}, {16, 8}}, {{4, 1}, {4, 1}, {16, 2}, {4, 1}, {16, 8}}};
        const char* names[2][GEMV_KINDS] = {{"RCA_GEMV_QKV", "RCA_GEMV_O", "RCA_GEMV_GU", "RCA_GEMV_DOWN", "RCA_GEMV_HEAD"},
                                           {"RCA_GEMVQ_QKV", "RCA_GEMVQ_O", "RCA_GEMVQ_GU", "RCA_GEMVQ_DOWN", "RCA_GEMVQ_HEAD"}};
        for (int f = 0; f < 2; ++f)
            for (int k = 0; k < GEMV_KINDS; ++k) {
                tab[f][k] = def[f][k];
                const char* e = getenv(names[f][k]);
                int r = 0, bw = 0;
                if (e && sscanf(e, "%d,%d", &r, &bw) == 2 && (r == 4 || r == 8 || r == 16) && bw >= 1) tab[f][k] = {r, bw};
            }
        init = true;
    }
    GemvGeom g = tab[q8 ? 1 : 0][kind];
    // small models: never fewer than ~64 workgroups while there are rows to hand out
    while (g.bpw > 1 && (N + g.R * g.bpw - 1) / (g.R * g.bpw) < 64) g.bpw >>= 1;
    return g;
}
template <int M, int NIT, int PRO, int EPI, int Q>
static void launch_gemv_r(const GemvGeom& g, rca_lm* h, const WMat& w, const float* x, float* y, int N, int K, int ldy, const GemvPro& pro,
                          const GemvRope& rope, hipStream_t st) {
    const int grid = cdiv(cdiv(N, g.R), g.bpw);
    const GemvQ8 qa{w.qs, w.sc, w.dd};
    switch (g.R) {
        case 4: lm_gemv_kernel<M, NIT, 4, PRO, EPI, Q><<<grid, 256, 0, st>>>(h->stt, w.w, qa, x, N, K, y, g.bpw, ldy, pro, rope); break;
        case 8: lm_gemv_kernel<M, NIT, 8, PRO, EPI, Q><<<grid, 256, 0, st>>>(h->stt, w.w, qa, x, N, K, y, g.bpw, ldy, pro, rope); break;
        default: lm_gemv_kernel<M, NIT, 16, PRO, EPI, Q><<<grid, 256, 0, st>>>(h->stt, w.w, qa, x, N, K, y, g.bpw, ldy, pro, rope); break;
    }
}
template <int PRO, int EPI, int Q>
static void launch_gemv_q(GemvGeom g, rca_lm* h, int M, const WMat& w, const float* x, float* y, int N, int K, int ldy, const GemvPro& pro,
                          const GemvRope& rope, hipStream_t st) {
    const int nit = cdiv(cdiv(K >> 3, 4), 64);
    // 16-byte weight loads in flight per lane: at most 16 (registers); q8_0 needs one load per row PAIR
    const int lpr = (Q == WF_Q8 || Q == WF_Q6K) ? 2 : (Q == WF_Q4K ? 4 : 1);
    const int max_loads = Q == WF_Q4K ? 8 : 16;   // Q4_K also holds the factors of every quad in registers
    while (g.R > 4 && (g.R / lpr) * (nit == 3 ? 4 : nit) > max_loads) g.R >>= 1;
    if ((Q == WF_Q4K || Q == WF_Q6K) && g.R < 8) g.R = 8;   // a Q4_K batch is at least two quads (one per register half); Q6_K scale loads cover two pairs
    if (PRO == 0 && EPI == 3 && nit > 1) {
        if (nit == 2) {
            if (M == 1) launch_gemv_r<1, 2, 0, 3, Q>(g, h, w, x, y, N, K, ldy, pro, rope, st);
            else launch_gemv_r<2, 2, 0, 3, Q>(g, h, w, x, y, N, K, ldy, pro, rope, st);
        } else {
            if (M == 1) launch_gemv_r<1, 4, 0, 3, Q>(g, h, w, x, y, N, K, ldy, pro, rope, st);
            else launch_gemv_r<2, 4, 0, 3, Q>(g, h, w, x, y, N, K, ldy, pro, rope, st);
        }
        return;
    }
    if (M == 1) launch_gemv_r<1, 1, PRO, EPI, Q>(g, h, w, x, y, N, K, ldy, pro, rope, st);
    else launch_gemv_r<2, 1, PRO, EPI, Q>(g, h, w, x, y, N, K, ldy, pro, rope, st);
}
// M = 1 or 2 tokens.  Only the down projection (K = ffn) needs more than one chunk per lane and wave.  The matrix is streamed in the
// format it is kept in.
template <int PRO, int EPI>
static void launch_gemv(int kind, rca_lm* h, int M, const WMat& w, const float* x, float* y, int N, int K, int ldy, const GemvPro& pro,
                        const GemvRope& rope, hipStream_t st) {
    const GemvGeom g = gemv_geom(kind, N, w.fmt == WF_Q8 || w.fmt == WF_Q4K || w.fmt == WF_Q6K);
    if (w.fmt == WF_Q6K) launch_gemv_q<PRO, EPI, WF_Q6K>(g, h, M, w, x, y, N, K, ldy, pro, rope, st);
    else if (w.fmt == WF_Q4K) launch_gemv_q<PRO, EPI, WF_Q4K>(g, h, M, w, x, y, N, K, ldy, pro, rope, st);
    else if (w.fmt == WF_Q8) launch_gemv_q<PRO, EPI, WF_Q8>(g, h, M, w, x, y, N, K, ldy, pro, rope, st);
    else if (w.fmt == WF_F16) launch_gemv_q<PRO, EPI, WF_F16>(g, h, M, w, x, y, N, K, ldy, pro, rope, st);
    else launch_gemv_q<PRO, EPI, WF_BF16>(g, h, M, w, x, y, N, K, ldy, pro, rope, st);
}

// Merge of the splits of one (token, head) row by ONE wave, lane <-> dim, in split order.  Latency code: every load it will ever
// need is issued in the first instructions -- the (m, l) pair of split `lane` and this lane's output element of up to CMB_PRE
// splits -- and none of them depends on the step state (the bound is the number of splits the attention kernel was LAUNCHED
// with; a split beyond the visible context carries m = -inf and weighs 0, its o is never used).  COHERENT: the partials were
// written by other workgroups of the SAME launch (fused path): every load is an agent-scope (sc1) load.
#define CMB_PRE 32
template <bool COHERENT>
__device__ __forceinline__ float attn_part_load(const float* p) {
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COHERENT>
__device__ __forceinline__ float attn_merge_row(const float* __restrict__ base, int nsp, int d) {
    const long sstride = 32L * 66;
    float pv[CMB_PRE];
#pragma unroll
    for (int j = 0; j < CMB_PRE; ++j) pv[j] = attn_part_load<COHERENT>(base + (long)min(j, nsp - 1) * sstride + 2 + d);
    float mx, ml0, ll0;
    {
        const int sp = min(d, nsp - 1);
        ml0 = attn_part_load<COHERENT>(base + sp * sstride);
        ll0 = attn_part_load<COHERENT>(base + sp * sstride + 1);
        if (d >= nsp) { ml0 = -INFINITY; ll0 = 0.0f; }
        mx = ml0;
    }
    for (int s0 = 64; s0 < nsp; s0 += 64) {   // more than 64 splits (contexts beyond 16 k): the rare, slower tail
        const int sp = s0 + d;
        mx = fmaxf(mx, sp < nsp ? attn_part_load<COHERENT>(base + sp * sstride) : -INFINITY);
    }
    mx = wave_max(mx);
    float L = 0.0f, O = 0.0f;
    {
        const float fl = (ml0 == -INFINITY) ? 0.0f : __expf(ml0 - mx);
        const int cnt = min(64, nsp);
#pragma unroll
        for (int j = 0; j < CMB_PRE; ++j) {
            if (j < cnt) {
                const float f = __shfl(fl, j), l = __shfl(ll0, j);
                L = __builtin_fmaf(l, f, L);
                O = __builtin_fmaf(f == 0.0f ? 0.0f : pv[j], f, O);
            }
        }
        for (int j = CMB_PRE; j < cnt; ++j) {
            const float f = __shfl(fl, j), l = __shfl(ll0, j);
            const float p = attn_part_load<COHERENT>(base + (long)j * sstride + 2 + d);
            L = __builtin_fmaf(l, f, L);
            O = __builtin_fmaf(f == 0.0f ? 0.0f : p, f, O);
        }
    }
    for (int s0 = 64; s0 < nsp; s0 += 64) {
        const int spl = min(s0 + d, nsp - 1);
        const float ml = attn_part_load<COHERENT>(base + spl * sstride), ll = attn_part_load<COHERENT>(base + spl * sstride + 1);
        const float fl = (ml == -INFINITY) ? 0.0f : __expf(ml - mx);
        const int cnt = min(64, nsp - s0);
        for (int j = 0; j < cnt; ++j) {
            const float f = __shfl(fl, j), l = __shfl(ll, j);
            const float p = attn_part_load<COHERENT>(base + (long)(s0 + j) * sstride + 2 + d);
            L = __builtin_fmaf(l, f, L);
            O = __builtin_fmaf(f == 0.0f ? 0.0f : p, f, O);
        }
    }
    return O / L;
}
// ------------------------------------------------------------------ attention on MFMA (decode steps and prefill tiles)
// grid (kv head, 256-key split, block of 32 query rows); a query row is (token, q head of this kv head), 32 / G tokens
// per block; 8 waves, each ONE block of 32 keys (so there is no online rescaling inside a wave):
//   S^T = K (32 keys x 64 dims, fp16 straight from the cache) x Q^T  on v_mfma_f32_32x32x16_f16, Q split into fp16
//         hi + lo (two MFMAs per 16 dims).  The result has lane <-> query row, registers <-> keys, so the softmax
//         statistics are 16 in-lane values plus one cross-half exchange.
//   O   = P x V: P goes in as the A operand exactly as it sits in registers (hi + lo fp16); the order of the
//         contraction index is free as long as both operands agree, so V is read from an LDS-transposed copy
//         vt[dim][key] in the key order P's registers have.
// Then the 8 waves are merged and the split partial is written like the pairwise kernel does:
// part[((qblock * nkv + g) * n_splits + split) * 32 + row][66] = {m, l, o[64]}, row = token_in_block * G + q_head.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short flash_s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
// Diagnostic build only (-DRCA_ATTN_TIMELINE, scripts/attn_timeline.py): thread 0 of every decode-attention workgroup stamps its phases
// with the 100 MHz wall clock into a buffer no other code reads.
#ifdef RCA_ATTN_TIMELINE
__device__ long* rca_attn_tl = nullptr;
#define ATL_STAMP(k) do { if (atl && threadIdx.x == 0) atl[k] = (long)wall_clock64(); } while (0)
#define ATL_STAMP_W7(k) do { if (atl && threadIdx.x == 448) atl[k] = (long)wall_clock64(); } while (0)   // the same phases seen by the last wave
#else
#define ATL_STAMP(k)
#define ATL_STAMP_W7(k)
#endif

#define ATTM_LDS (8 * 32 * 64 * 4 + 2 * 8 * 32 * 4)   // wo (aliases the K / V images) + wm + wl
template <int G>
// (argument order: what the first loads need -- the cache pointers, the head counts, n_ctx -- sits inside the 14 dwords the
//  dispatcher preloads into SGPRs; `part` / `arrive` / `attn_out` are only needed at the end)
__global__ __launch_bounds__(512) void lm_attn_mfma_kernel(const LmDevState* __restrict__ stt, const float* __restrict__ qkv,
                                                           const f16_t* __restrict__ kc, const f16_t* __restrict__ vc,
                                                           int nh, int nkv, int n_splits, float scale, int n_ctx,
                                                           float* __restrict__ part, int* __restrict__ arrive, float* __restrict__ attn_out) {
    // arrive != nullptr (decode steps: one query block, at most one workgroup per CU, at most 8 live query rows): the merge of the
    // splits happens HERE, by data-tagged granules (MI355X_MICROARCH.md, price list rows handoff-1to1 / allgather: "granule = one
    // naturally aligned 8-byte {data, tag} written by ONE sc1 store", polled with sc1 loads; R2: a granule needs no ordering).  Every
    // workgroup publishes its partial (m, l, o[64]) of the 8 live rows as 528 granules {f32 bits, tag} -- no drain, no barrier, no
    // ticket -- and leaves; the workgroups of the first splits then gather, one wave per (token, head) row and lane <-> dim, the
    // granules of every launched split, re-polling until each carries this launch's tag, and merge them in split order with the
    // arithmetic of the separate combine kernel (same bits).  The tag is the kv head's launch counter arrive[g]: read at entry by every workgroup of the
    // head, advanced by split 0's workgroup at its very end -- after it has seen every producer's granules, so no workgroup of this
    // launch can read the new value; the next launch (another layer, the same buffer) starts behind the kernel boundary.  Producers
    // never wait, so the merger's wait ends whatever the placement; its spin is bounded all the same (a NaN row would tell).
    // Round 3 did this with drained sc1 stores + an arrival ticket + a re-read by the last arriver: 4.2 us from the last partial's
    // stores to the merged row at 6.6 k context (profiles/r04/attn_decode_timeline.txt); the granules take the drain, the two
    // barriers and the ticket round trip out of the chain.
    const bool fused = arrive != nullptr;
#ifdef RCA_ATTN_TIMELINE
    long* const atl = (rca_attn_tl && fused) ? rca_attn_tl + ((long)(blockIdx.y * gridDim.x + blockIdx.x) & 1023) * 16 : nullptr;
#endif
    ATL_STAMP(0);
    ATL_STAMP_W7(11);
    unsigned long long* const gran = reinterpret_cast<unsigned long long*>(arrive + ATT_EPOCH_INTS) +
                                     ((long)blockIdx.x * n_splits + blockIdx.y) * (8 * 66);     // this workgroup's 8 x 66 granules
    unsigned tag = 0;
    auto part_store = [&](float* p, float v) { *p = v; };
    auto gran_store = [&](int row, int e, float v) {
        __hip_atomic_store(gran + row * 66 + e, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    };
    auto tag_merge = [&](int M_) {
        // The rows of a kv head are merged by DIFFERENT workgroups -- row r by the workgroup of split r % nmerge, nmerge = min(8,
        // launched splits) -- one wave per row, lane <-> dim: a row's sweep is splits x 528 bytes, and eight rows gathered by ONE
        // workgroup (139 KB of 8-byte sc1 loads through one CU at 6.6 k context) took 2 us per sweep.
        const int d = threadIdx.x & 63;
        const int nsp = (int)gridDim.y;
        const int nmerge = min(nsp, 8);
        const int w = (int)blockIdx.y + nmerge * (int)(threadIdx.x >> 6);       // row (token_in_block * G + q head) of this wave
        if (w < M_ * G) {
            const unsigned long long* base = reinterpret_cast<const unsigned long long*>(arrive + ATT_EPOCH_INTS) + ((long)blockIdx.x * n_splits) * (8 * 66) + w * 66;
            auto gload = [&](const unsigned long long* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            int spins = 0;
            // ONE sweep asks for everything the row needs -- (m, l) of split d + 64 c in lane d and this lane's output element of the
            // first CMB_PRE splits -- and is repeated as a whole until every granule carries the tag: in the common case the merge
            // costs one memory round trip behind the last producer's stores (two sweeps in sequence, (m, l) first, cost two).
            unsigned long long gm[ATT_TAG_MAXSP / 64], gl[ATT_TAG_MAXSP / 64], pg[CMB_PRE];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < ATT_TAG_MAXSP / 64; ++c) {
                    gm[c] = gl[c] = (unsigned long long)tag << 32;
                    if (64 * c < nsp) {
                        const int spj = min(64 * c + d, nsp - 1);
                        gm[c] = gload(base + (long)spj * (8 * 66));
                        gl[c] = gload(base + (long)spj * (8 * 66) + 1);
                    }
                }
#pragma unroll
                for (int j = 0; j < CMB_PRE; ++j) pg[j] = gload(base + (long)min(j, nsp - 1) * (8 * 66) + 2 + d);
#pragma unroll
                for (int c = 0; c < ATT_TAG_MAXSP / 64; ++c) ok = ok && (unsigned)(gm[c] >> 32) == tag && (unsigned)(gl[c] >> 32) == tag;
#pragma unroll
                for (int j = 0; j < CMB_PRE; ++j) ok = ok && (unsigned)(pg[j] >> 32) == tag;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull || ++spins > ATT_TAG_SPINS) break;
                __builtin_amdgcn_s_sleep(1);
            }
            float ml[ATT_TAG_MAXSP / 64], ll[ATT_TAG_MAXSP / 64];
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < ATT_TAG_MAXSP / 64; ++c) {
                const bool live = 64 * c + d < nsp;
                ml[c] = live ? __uint_as_float((unsigned)gm[c]) : -INFINITY;
                ll[c] = live ? __uint_as_float((unsigned)gl[c]) : 0.0f;
                mx = fmaxf(mx, ml[c]);
            }
            mx = wave_max(mx);
            float L = 0.0f, O = 0.0f;
#pragma unroll
            for (int c = 0; c < ATT_TAG_MAXSP / 64; ++c) {
                if (64 * c >= nsp) break;
                const float fl = (ml[c] == -INFINITY) ? 0.0f : __expf(ml[c] - mx);
                for (int j0 = 64 * c; j0 < min(nsp, 64 * c + 64); j0 += CMB_PRE) {
                    const int cnt = min(CMB_PRE, nsp - j0);
                    if (j0 > 0) {   // later blocks of CMB_PRE splits (models with fewer kv heads): their granules are long there by now
                        for (;;) {
                            bool ok = true;
#pragma unroll
                            for (int j = 0; j < CMB_PRE; ++j) pg[j] = gload(base + (long)min(j0 + j, nsp - 1) * (8 * 66) + 2 + d);
#pragma unroll
                            for (int j = 0; j < CMB_PRE; ++j) ok = ok && (unsigned)(pg[j] >> 32) == tag;
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull || ++spins > ATT_TAG_SPINS) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < CMB_PRE; ++j) {
                        if (j < cnt) {
                            const float f = __shfl(fl, (j0 + j) & 63), l = __shfl(ll[c], (j0 + j) & 63);
                            const float pv = __uint_as_float((unsigned)pg[j]);
                            L = __builtin_fmaf(l, f, L);
                            O = __builtin_fmaf(f == 0.0f ? 0.0f : pv, f, O);
                        }
                    }
                }
            }
            const int m = w / G, head = blockIdx.x * G + w % G;
            attn_out[(long)m * nh * 64 + head * 64 + d] = spins > ATT_TAG_SPINS ? __builtin_nanf("") : O / L;
        }
        // the next launch's tag (never 0: a zeroed buffer matches nothing).  Written by split 0's workgroup once its row is merged:
        // by then every producer has stored -- so has read -- this launch's tag; the other rows' mergers compare with the tag they
        // hold in a register, not with this word.
        if (blockIdx.y == 0 && threadIdx.x == 0) arrive[blockIdx.x] = (int)(tag + 1u == 0u ? 1u : tag + 1u);
        ATL_STAMP(6);
    };
    constexpr int HD = 64;
    constexpr int TPB = 32 / G;   // tokens per query block
    extern __shared__ __attribute__((aligned(16))) float attm_lds[];
    float (*wo)[32][HD] = reinterpret_cast<float (*)[32][HD]>(attm_lds);                        // [8][32][64]
    // (the waves' K / V images, 8 KB each, alias wo: they are dead before wo is written)
    float (*wm)[32] = reinterpret_cast<float (*)[32]>(attm_lds + 8 * 32 * HD);
    float (*wl)[32] = wm + 8;
    const int g = blockIdx.x, sp = blockIdx.y, qb = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kbase = sp * ATT_KEYS;
    const int kwb = kbase + wave * 32;
    const int t0 = qb * TPB;
    const int tl = col / G, hq = col % G;
    const int ld = (nh + 2 * nkv) * HD;
    // every load that does not depend on the step state goes out first (rows past the visible range: clamped, masked).
    // K / V of this wave's 32 keys are loaded in WHOLE ROWS -- lane l of load i takes 16-byte chunk l % 8 of key 8 i + l / 8: 8 cache
    // lines per instruction -- and reach the MFMA operand layouts through two wave-private 4 KB LDS images (the flash kernel's
    // swizzles).  Loading the A layout directly (lane <-> key: 16 bytes of each of 32 rows per instruction) made the CU's address
    // path the bottleneck of the whole launch: round-4 timeline, 1 k context, 32 workgroups: wave 0 had its keys 2.4 us after entry
    // and wave 7 1.55 us later (profiles/r04/attn_decode_timeline.txt).  K first, then q, then V: S^T starts while V is in flight.
    u32x4 kf[4], vf[4];
    f32x4 qf[4][2];
    char* const kbytes = reinterpret_cast<char*>(attm_lds) + wave * 8192;   // K image [key][128 bytes], chunk c of key k at c ^ ((k >> 1) & 7)
    char* const vbytes = kbytes + 4096;                                     // V image, chunk c of key k at c ^ (k1 k2 k0)
    const int ld_key = lane >> 3, ld_chunk = lane & 7;
    {
        long roff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) roff[i] = ((long)min(kwb + 8 * i + ld_key, n_ctx - 1) * nkv + g) * HD + 8 * ld_chunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) kf[i] = *reinterpret_cast<const u32x4*>(kc + roff[i]);
        // a decode step (fused merge) has at most two tokens: the lanes of the 24 dead query rows load nothing (their scores are
        // masked whatever q they hold) -- the q loads were as many bytes through the CU's address path as K and V together
        const float* qp = qkv + (long)min(t0 + tl, LM_MAXM - 1) * ld + (g * G + hq) * HD + 8 * half;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) { qf[sub][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; qf[sub][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        if (!fused || tl * G < 8) {   // the fused merge is only used while M * G <= 8
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                qf[sub][0] = *reinterpret_cast<const f32x4*>(qp + 16 * sub);
                qf[sub][1] = *reinterpret_cast<const f32x4*>(qp + 16 * sub + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = *reinterpret_cast<const u32x4*>(vc + roff[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
    auto pin_loads = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(kf[c]), "v"(vf[c]), "v"(qf[c][0]), "v"(qf[c][1]));
    };
    const int M = stt->m;
    if (t0 >= M) { pin_loads(); return; }
    const int pos0 = stt->n_tokens;
    const int ntok = min(TPB, M - t0);
    const int kmax = pos0 + t0 + ntok;   // keys [0, kmax) are visible to the last token of the block
    ATL_STAMP(12);
    float* pout = part + ((long)(qb * nkv + g) * n_splits + sp) * 32 * 66;
    if (fused) tag = (unsigned)arrive[blockIdx.x];
    if (kbase >= kmax) {   // nothing visible in this split
        if (fused) {   // all 8 x 66 granules, so that the merger has one rule: every granule of every launched split carries the tag
            for (int i = threadIdx.x; i < 8 * 66; i += 512) gran_store(i / 66, i % 66, (i % 66) == 0 ? -INFINITY : 0.0f);
        } else if (threadIdx.x < 32) { part_store(pout + threadIdx.x * 66, -INFINITY); part_store(pout + threadIdx.x * 66 + 1, 0.0f); }
        pin_loads();
        if (fused && sp < 8) tag_merge(M);   // a launched split beyond the context still merges the rows that fall to it
        return;
    }
    // ---- K image of this wave (written and read by this wave only: LDS operations of a wave complete in order)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 8 * i + ld_key;
        *reinterpret_cast<u32x4*>(kbytes + k * 128 + ((ld_chunk ^ ((k >> 1) & 7)) << 4)) = kf[i];
    }
    ATL_STAMP(13);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- S^T = K Q^T (hi + lo): lane (key = col, half) reads chunk half + 2 sub (dims 8 half + 16 sub .. + 8) of its key
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
        f16x8 qh, ql;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q = j < 4 ? qf[sub][0][j] : qf[sub][1][j - 4];
            const _Float16 h16 = (_Float16)q;
            qh[j] = h16;
            ql[j] = (_Float16)(q - (float)h16);
        }
        const u32x4 kraw = *reinterpret_cast<const u32x4*>(kbytes + col * 128 + (((half | (sub << 1)) ^ ((col >> 1) & 7)) << 4));
        const f16x8 kfr = __builtin_bit_cast(f16x8, kraw);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr, qh, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr, ql, sacc, 0, 0, 0);
    }
    // ---- V image (its loads were issued behind K and q)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 8 * i + ld_key;
        *reinterpret_cast<u32x4*>(vbytes + k * 128 + ((ld_chunk ^ ((((k >> 1) & 1) << 2) | (((k >> 2) & 1) << 1) | (k & 1))) << 4)) = vf[i];
    }
    ATL_STAMP(1);   // V of this wave is in registers (its image requested)
    ATL_STAMP_W7(9);
    // ---- softmax statistics of this wave's 32 keys for query row `col`: registers are keys
    const int qpos = pos0 + t0 + tl;
    const bool qvalid = tl < ntok;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = kwb + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float sv = (qvalid && key <= qpos) ? sacc[r] * scale : -INFINITY;
        sacc[r] = sv;
        mx = fmaxf(mx, sv);
    }
    {
        float a = mx, b = mx;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        mx = fmaxf(a, b);
    }
    float lsum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = (mx == -INFINITY) ? 0.0f : __expf(sacc[r] - mx);
        sacc[r] = e;
        lsum += e;
    }
    {
        float a = lsum, b = lsum;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        lsum = a + b;
    }
    // P as the A operand of O = P V: registers 8i..8i+7 feed MFMA i (hi + lo fp16)
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pv = sacc[8 * i + j];
            const _Float16 h16 = (_Float16)pv;
            ph[i][j] = h16;
            pl[i][j] = (_Float16)(pv - (float)h16);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- O = P V: dims 32 * nt + col; slot j of MFMA i, half h is key 16 i + 4 h + 8 (j >> 2) + (j & 3).  Transposed reads:
    // lane 4 q + p of a 16-lane group supplies row (key) q, dims 4 p .. 4 p + 3 of the group's 16 dims
    f32x16 oacc[2];
    {
        const int l16 = lane & 15, q4 = l16 >> 2, pp = l16 & 3;
        const int sw = ((q4 >> 1) << 2) | (half << 1) | (q4 & 1);   // rows 16 i + 4 half + q4 (+ 8)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int tr_off = (4 * half + q4) * 128 + (((4 * nt + 2 * (col >> 4) + (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[nt][r] = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* tp = vbytes + tr_off + 16 * i * 128;
                const flash_s16x4 t0v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) flash_s16x4*)tp);
                const flash_s16x4 t1v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) flash_s16x4*)(tp + 8 * 128));
                const f16x4 v0 = __builtin_bit_cast(f16x4, t0v), v1 = __builtin_bit_cast(f16x4, t1v);
                // keys past the visible range may hold anything (their p is 0): zero them
                f16x8 vb;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = kwb + 16 * i + 4 * half + 8 * (j >> 2) + (j & 3);
                    const _Float16 x = j < 4 ? v0[j] : v1[j - 4];
                    vb[j] = key < kmax ? x : (_Float16)0.0f;
                }
                oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[i], vb, oacc[nt], 0, 0, 0);
                oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[i], vb, oacc[nt], 0, 0, 0);
            }
        }
    }
    ATL_STAMP(2);      // S, softmax, P V of this wave
    ATL_STAMP_W7(10);
    __syncthreads();   // every wave is done with vt: wo may overwrite it
    ATL_STAMP(7);
    if (half == 0) { wm[wave][col] = mx; wl[wave][col] = lsum; }
    // (a decode step has at most 8 live query rows -- registers 0..3 of the accumulators: the other 24 rows are never merged)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r < 4 || !fused) wo[wave][(r & 3) + 8 * (r >> 2) + 4 * half][32 * nt + col] = oacc[nt][r];
    __syncthreads();
    ATL_STAMP(8);
    // ---- merge the 8 waves, write the split partial (a decode step has M * G <= 8 live rows of the 32: the others are skipped)
    const int live_rows = fused ? M * G : 32;
    for (int i = threadIdx.x; i < live_rows * HD; i += 512) {
        const int r = i / HD, d = i - r * HD;
        float m2 = wm[0][r];
#pragma unroll
        for (int w = 1; w < 8; ++w) m2 = fmaxf(m2, wm[w][r]);
        float L = 0.0f, O = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float f = (wm[w][r] == -INFINITY) ? 0.0f : __expf(wm[w][r] - m2);
            L = __builtin_fmaf(wl[w][r], f, L);
            O = __builtin_fmaf(wo[w][r][d], f, O);
        }
        if (fused) {
            gran_store(r, 2 + d, O);
            if (d == 0) { gran_store(r, 0, m2); gran_store(r, 1, L); }
        } else {
            part_store(pout + r * 66 + 2 + d, O);
            if (d == 0) { part_store(pout + r * 66, m2); part_store(pout + r * 66 + 1, L); }
        }
    }
    ATL_STAMP(3);      // 8 waves merged, partial stores issued
    if (fused && sp < 8) tag_merge(M);
}
#ifdef RCA_ATTN_TIMELINE
extern "C" int rca_debug_attn_timeline(long* out_host, int enable) {   // enable: allocate + arm; else copy the 1024 x 8 stamps out
    static long* buf = nullptr;
    if (enable) {
        if (!buf) { if (hipMalloc((void**)&buf, 1024 * 16 * sizeof(long)) != hipSuccess) return 1; }
        (void)hipMemset(buf, 0, 1024 * 16 * sizeof(long));
        return hipMemcpyToSymbol(HIP_SYMBOL(rca_attn_tl), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
    }
    (void)hipDeviceSynchronize();
    return buf && hipMemcpy(out_host, buf, 1024 * 16 * sizeof(long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
#ifdef RCA_ATTN_TIMELINE
// flash prefill attention: per wave [entry, loop start, loop end, exit] on the 100 MHz wall clock, then shader cycles summed over its
// blocks for [QK^T + mask/max, exp + sum, P split, PV], block count and the SIMD / CU it ran on (16 longs per wave)
__device__ long* rca_flash_tl = nullptr;
extern "C" int rca_debug_flash_timeline(long* out_host, int enable, int n_waves) {
    static long* buf = nullptr;
    static int cap = 0;
    if (enable) {
        if (!buf || cap < n_waves) { if (buf) (void)hipFree(buf); if (hipMalloc((void**)&buf, (size_t)n_waves * 16 * sizeof(long)) != hipSuccess) return 1; cap = n_waves; }
        (void)hipMemset(buf, 0, (size_t)cap * 16 * sizeof(long));
        return hipMemcpyToSymbol(HIP_SYMBOL(rca_flash_tl), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
    }
    (void)hipDeviceSynchronize();
    return buf && hipMemcpy(out_host, buf, (size_t)min(cap, n_waves) * 16 * sizeof(long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#define FTL_CLK() ((long)__builtin_readcyclecounter())
#define FTL_PHASE(k) do { const long t_now = FTL_CLK(); ftl_acc[k] += t_now - ftl_t; ftl_t = t_now; } while (0)
#else
#define FTL_PHASE(k)
#endif
// merges the splits of lm_attn_mfma_kernel as its own launch (prefill tiles, and decode steps whose grid exceeds one workgroup per
// CU): one wave per (token, head)
template <int G>
__global__ __launch_bounds__(64) void lm_attn_mfma_combine_kernel(const LmDevState* __restrict__ stt, const float* __restrict__ part,
                                                                  float* __restrict__ attn, int nh, int nkv, int n_splits, int nsp_launch,
                                                                  bf16_t* __restrict__ hi = nullptr, bf16_t* __restrict__ lo = nullptr) {
    constexpr int TPB = 32 / G;
    const int m = blockIdx.x / nh, head = blockIdx.x % nh;
    const int g = head / G, hq = head % G;
    const int qb = m / TPB, tl = m % TPB;
    const int r = tl * G + hq;
    const int d = threadIdx.x;
    const float* base = part + ((long)(qb * nkv + g) * n_splits) * 32 * 66 + r * 66;
    const float ov = attn_merge_row<false>(base, min(nsp_launch, n_splits), d);
    if (m >= stt->m) return;
    if (hi) {   // prefill tiles: the O-projection GEMM reads bf16 hi + lo
        const bf16_t hb = f32_to_bf16_rne(ov);
        hi[(long)m * nh * 64 + head * 64 + d] = hb;
        lo[(long)m * nh * 64 + head * 64 + d] = f32_to_bf16_rne(ov - __uint_as_float((unsigned)hb << 16));
    } else {
        attn[(long)m * nh * 64 + head * 64 + d] = ov;
    }
}

// ------------------------------------------------------------------ prefill attention, flash shape
// The decode-shaped kernel above gives every block of 32 query rows one workgroup PER 256 keys and merges the partial results in a
// second launch: at 6 k tokens of context a 1024-token pass wrote and re-read 0.2 GB of partials per layer and spent 245 us in
// attention for ~50 us of MFMA work.  Prefill passes run this kernel instead: a TEAM of three waves owns a tile of 32 query rows
// (32 / G tokens x the G heads of one kv head); wave w takes the key blocks of 32 at ABSOLUTE positions 32 (3 i + w) up to the tile's
// last visible key, each wave with its own online softmax in registers, and the three (m, l, O) are merged through LDS once at the
// end -- no partials in HBM, no second launch, three waves per SIMD whose MFMA and VALU phases overlap.
// Per key block:
//   K, V:         global -> registers (one block ahead) -> the wave's two 4 KB LDS images, loaded 8 whole rows per instruction
//   S^T = K Q^T   (v_mfma_f32_32x32x16_f16; K fragments by ds_read_b128; Q scaled by scale * log2(e) and split into fp16 hi + lo
//                 once per wave) -> lane <-> query row, registers <-> keys: max / sum / rescale are per-lane scalars, p = 2^(s - m);
//                 only the blocks that reach past the tile's first row are masked (a second copy of the loop body)
//   O^T += V^T P^T: V^T fragments by ds_read_b64_tr_b16, which hands a lane 4 consecutive KEYS of its dim; B = P exactly as it sits
//                 in the S^T registers (fp16 hi + lo, hi rounded toward zero so lo is the rest, one v_fma_mix_f32 each) -> O^T has
//                 dims on registers and the query row on the lane.
//   The O^T rescale by 2^(m - m') is skipped when no lane's maximum moved (multiplying by 1.0 changes nothing).
// A query row's result depends only on its own position (every row sees the same key blocks on the same waves in the same order;
// blocks past its position contribute exact zeros behind alpha = 1, and the three partial results are merged in wave order), so the
// bits do not depend on how a prompt is cut into evals or passes -- the property the shadow KV cache rests on
// (test_mfma_prefill_is_tiling_invariant).
// Measured at 6.6 k context, per layer and 1024-token pass (profiles/r03/flash_attention_steps.txt): one wave per tile 148 us ->
// key blocks split over the waves of a workgroup 101 -> coalesced loads through LDS 90 (the direct A-layout loads took 16 bytes from
// each of 32 cache lines per instruction: 4100 of 4900 cycles per block were spent waiting on the L1) -> one workgroup per CU ~78.
#define FLASH_WAVES 3        // waves that split the key blocks of one query tile
#define FLASH_OPITCH 68      // f32 per query row of a wave's O in the merge buffer
// TEAMS query tiles per workgroup (a team = FLASH_WAVES waves = one tile).  With 4 teams a workgroup takes more than half of a CU's
// LDS, so every CU gets exactly one and the dispatcher cannot pile five tiles on one CU and three on another (measured with one tile
// per workgroup: 414 .. 730 key blocks per CU, the last CU done at 135 us against a median of 101); the four tiles of workgroup j
// are j, 2 n - 1 - j, 2 n + j and 4 n - 1 - j of the head's 4 n tiles, whose causal lengths add up to the same total for every j.
template <int G, int TEAMS>
__global__ __launch_bounds__(64 * FLASH_WAVES * TEAMS, 3) void lm_attn_flash_kernel(const LmDevState* __restrict__ stt, const float* __restrict__ qkv,
                                                                        const f16_t* __restrict__ kc, const f16_t* __restrict__ vc,
                                                                        float* __restrict__ attn_out, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                                        int nh, int nkv, float scale, int n_ctx) {
    constexpr int HD = 64;
    constexpr int TPB = 32 / G;   // tokens per tile
    constexpr int NWAVES = FLASH_WAVES * TEAMS;
    // the K / V images of the key loop (8 KB per wave) and the merge buffer that follows it share one region (a barrier between the uses)
    __shared__ __attribute__((aligned(16))) float flash_lds[NWAVES * 32 * FLASH_OPITCH];
    __shared__ float mrg_m_all[NWAVES][32], mrg_l_all[NWAVES][32];
    static_assert(8192 <= sizeof(float) * 32 * FLASH_OPITCH, "K / V images fit the merge buffer");
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int team = wave_all / FLASH_WAVES, wave = wave_all % FLASH_WAVES;
    float (*mrg_o)[32][FLASH_OPITCH] = reinterpret_cast<float (*)[32][FLASH_OPITCH]>(flash_lds) + team * FLASH_WAVES;
    float (*mrg_m)[32] = mrg_m_all + team * FLASH_WAVES;
    float (*mrg_l)[32] = mrg_l_all + team * FLASH_WAVES;
    char* const kbytes = reinterpret_cast<char*>(flash_lds) + wave_all * 8192;   // K image of the current key block: [key][128 bytes]
    char* const vbytes = kbytes + 4096;                                          // V image
    const int M = stt->m;
    const int ntiles = (M + TPB - 1) / TPB;
    int tile = blockIdx.y;
    if (TEAMS == 4) {
        const int n4 = gridDim.y, j = blockIdx.y;
        tile = team == 0 ? j : team == 1 ? 2 * n4 - 1 - j : team == 2 ? 2 * n4 + j : 4 * n4 - 1 - j;
    } else if (TEAMS == 2) {
        tile = team == 0 ? (int)blockIdx.y : 2 * (int)gridDim.y - 1 - (int)blockIdx.y;
    }
    const bool active = tile < ntiles;   // team-uniform; an idle team still takes part in the workgroup's barriers
    if (TEAMS == 1 && !active) return;
    const int t0 = min(tile, ntiles - 1) * TPB;
#ifdef RCA_ATTN_TIMELINE
    long* const ftl = rca_flash_tl ? rca_flash_tl + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVES + wave_all) * 16 : nullptr;
    long ftl_acc[4] = {0, 0, 0, 0}, ftl_t = 0, ftl_n = 0;
    const long ftl_entry = (long)wall_clock64();
#endif
    const int pos0 = stt->n_tokens;
    const int ntok = min(TPB, M - t0);
    const int tl = min(col / G, ntok - 1), hq = col % G;   // rows past the pass repeat the last valid token's: they are never stored
    const int ld = (nh + 2 * nkv) * HD;
    const int qpos = pos0 + t0 + tl;
    const int kmax = min(pos0 + t0 + ntok, n_ctx);   // keys [0, kmax) are visible to the last token of this tile (and are all written)
    const int nblk = active ? (kmax + 31) >> 5 : 0;
    // ---- K / V of a key block, global -> registers -> LDS.  A load instruction covers 8 whole rows (lane l: key 8 i + l / 8, 16-byte
    // chunk l % 8 of its 128-byte row): 8 cache lines per instruction.  (Loading the MFMA A layout directly -- lane <-> key, 16 bytes of
    // each of 32 rows per instruction -- touched 32 lines for 1 KB and made the kernel wait on the L1: 4100 of 4900 cycles per block.)
    // Images: chunk c of key k sits at chunk c ^ sw(k); K: sw = (k >> 1) & 7, so the 16 lanes of a ds_read_b128 group (lane <-> key) tile
    // the 64 banks; V: sw = k1 k2 k0 (bits of k), so the 4 rows x 64 bytes of a transposed read do.  A store group is 8 lanes = one row.
    u32x4 kf[4], vf[4];
    const int ld_key = lane >> 3, ld_chunk = lane & 7;
    auto load_block = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long off = ((long)min(32 * max(kb, 0) + 8 * i + ld_key, kmax - 1) * nkv + g) * HD + 8 * ld_chunk;   // keys past kmax: a written row, p = 0
            kf[i] = *reinterpret_cast<const u32x4*>(kc + off);
            vf[i] = *reinterpret_cast<const u32x4*>(vc + off);
        }
    };
    int kw_off[4], vw_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 8 * i + ld_key;
        kw_off[i] = k * 128 + ((ld_chunk ^ ((k >> 1) & 7)) << 4);
        vw_off[i] = k * 128 + ((ld_chunk ^ ((((k >> 1) & 1) << 2) | (((k >> 2) & 1) << 1) | (k & 1))) << 4);
    }
    auto stage_block = [&]() {   // the registers' block becomes the current images
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(kbytes + kw_off[i]) = kf[i];
            *reinterpret_cast<u32x4*>(vbytes + vw_off[i]) = vf[i];
        }
    };
    load_block(min(wave, nblk - 1));
    // ---- Q of this lane's query row, times scale * log2(e), fp16 hi + lo
    f16x8 qh[4], ql[4];
    {
        const float qs = scale * 1.44269504088896340736f;
        const float* qp = qkv + (long)(t0 + tl) * ld + (g * G + hq) * HD + 8 * half;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * sub), b = *reinterpret_cast<const f32x4*>(qp + 16 * sub + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float q = (j < 4 ? a[j] : b[j - 4]) * qs;
                const _Float16 h16 = (_Float16)q;
                qh[sub][j] = h16;
                ql[sub][j] = (_Float16)(q - (float)h16);
            }
        }
    }
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 oacc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[nt][r] = 0.0f;
    // K fragments: lane (key = col, half) reads chunk half + 2 sub (dims 8 half + 16 sub .. + 8) of its key
    int kr_off[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) kr_off[sub] = col * 128 + (((half | (sub << 1)) ^ ((col >> 1) & 7)) << 4);
    // transposed reads: lane 4 q + p of a 16-lane group supplies row (key) q, dims 4 p .. 4 p + 3 of the group's 16 dims
    int tr_off[2];
    {
        const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
        const int sw = ((q >> 1) << 2) | (half << 1) | (q & 1);   // rows 16 i + 4 half + q (+ 8)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) tr_off[nt] = (4 * half + q) * 128 + (((4 * nt + 2 * (col >> 4) + (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1);
    }
    const int nfull = min((pos0 + t0 + 1) >> 5, nblk);   // blocks [0, nfull) are visible to every row of the tile: no mask
    auto block = [&](int kb, auto masked_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        u32x4 kcur[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) kcur[sub] = *reinterpret_cast<const u32x4*>(kbytes + kr_off[sub]);
        // ---- S^T = K Q^T (hi + lo), in log2 units
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const f16x8 kfr = __builtin_bit_cast(f16x8, kcur[sub]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr, qh[sub], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr, ql[sub], sacc, 0, 0, 0);
        }
        // ---- online softmax for query row `col`: register r is key 32 kb + 4 half + (r & 3) + 8 (r >> 2)
        float mx = -INFINITY;
        if (MASKED) {
            const int lim = qpos - 32 * kb - 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = ((r & 3) + 8 * (r >> 2) <= lim) ? sacc[r] : -INFINITY;
                sacc[r] = sv;
                mx = fmaxf(mx, sv);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
        }
        {
            float a = mx, b = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            mx = fmaxf(a, b);
        }
        FTL_PHASE(0);
        const float m_new = fmaxf(m_run, mx);
        const float m_ref = (m_new == -INFINITY) ? 0.0f : m_new;   // a row with nothing visible yet: 2^(-inf - 0) = 0 everywhere
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_ref);
        float lsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sacc[r] - m_ref);
            sacc[r] = e;
            lsum += e;
        }
        {
            float a = lsum, b = lsum;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            lsum = a + b;
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {   // wave-uniform
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[nt][r] *= alpha;
        }
        FTL_PHASE(1);
        // V^T fragments: all eight transposed reads go out before the P split below, which hides their latency.  The image was
        // written at the top of the block by this wave only (LDS operations of a wave complete in order).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        flash_s16x4 vfr[2][2][2];   // [nt][i][key group]: lane <-> dim 32 nt + col; slot j <-> key 16 i + 4 half + 8 (j >> 2) + (j & 3)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* tp = vbytes + tr_off[nt] + 16 * i * 128;
                vfr[nt][i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) flash_s16x4*)tp);
                vfr[nt][i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) flash_s16x4*)(tp + 8 * 128));
            }
        __builtin_amdgcn_sched_barrier(0);
        // P as the B operand of O^T += V^T P^T: registers 8 i .. 8 i + 7 feed MFMA i (fp16 hi toward zero + lo = the exact rest, rounded)
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = sacc[8 * i + 2 * j], p1 = sacc[8 * i + 2 * j + 1];
                const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0, p1));
                float r0, r1;   // p - (float)hi in one instruction each
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2), "v"(p0));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2), "v"(p1));
                ph[i][j] = h2;
                pl[i][j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
            }
        FTL_PHASE(2);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                typedef short s16x8_t __attribute__((__vector_size__(8 * sizeof(short))));
                const s16x8_t v01 = __builtin_shufflevector(vfr[nt][i][0], vfr[nt][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const f16x8 vb = __builtin_bit_cast(f16x8, v01);
                oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vb, __builtin_bit_cast(f16x8, ph[i]), oacc[nt], 0, 0, 0);
                oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vb, __builtin_bit_cast(f16x8, pl[i]), oacc[nt], 0, 0, 0);
            }
        }
        // the next block (in registers since the last iteration) replaces the images: this block's reads are done (one wave, LDS
        // operations complete in order); the block after it starts its way from the caches
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        stage_block();
        load_block(min(kb + 2 * FLASH_WAVES, nblk - 1));   // unconditional (the last iterations re-read a block): no merge of wait states
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    stage_block();                                          // block `wave` (waits for its loads)
    load_block(min(wave + FLASH_WAVES, nblk - 1));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int kb = wave;
#ifdef RCA_ATTN_TIMELINE
    const long ftl_loop0 = (long)wall_clock64();
    ftl_t = FTL_CLK();
    for (; kb < nfull; kb += FLASH_WAVES) { block(kb, std::false_type{}); FTL_PHASE(3); ++ftl_n; }
    for (; kb < nblk; kb += FLASH_WAVES) { block(kb, std::true_type{}); FTL_PHASE(3); ++ftl_n; }
    const long ftl_loop1 = (long)wall_clock64();
#else
    for (; kb < nfull; kb += FLASH_WAVES) block(kb, std::false_type{});
    for (; kb < nblk; kb += FLASH_WAVES) block(kb, std::true_type{});
#endif
    // ---- the four waves' (m, l, O^T): this lane holds, for query row col, dims 32 nt + 8 (r >> 2) + 4 half + (r & 3)
    __syncthreads();   // every wave is done with its V image
    if (half == 0) { mrg_m[wave][col] = m_run; mrg_l[wave][col] = l_run; }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4*>(&mrg_o[wave][col][32 * nt + 8 * rq + 4 * half]) =
                f32x4{oacc[nt][4 * rq], oacc[nt][4 * rq + 1], oacc[nt][4 * rq + 2], oacc[nt][4 * rq + 3]};
    __syncthreads();
#ifdef RCA_ATTN_TIMELINE
    if (ftl && lane == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        ftl[0] = ftl_entry; ftl[1] = ftl_loop0; ftl[2] = ftl_loop1; ftl[3] = (long)wall_clock64();
        for (int k = 0; k < 4; ++k) ftl[4 + k] = ftl_acc[k];
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xcc));   // XCC_ID
        ftl[8] = ftl_n; ftl[9] = hwid; ftl[10] = nblk; ftl[11] = xcc & 15;
    }
#endif
    // ---- merge in wave order: thread of the team <-> (query row, 8 dims)
    if (!active) return;
    for (int idx = threadIdx.x - 64 * FLASH_WAVES * team; idx < 256; idx += 64 * FLASH_WAVES) {
        const int row = idx >> 3, d0 = 8 * (idx & 7);
        if (row / G >= ntok) break;
        float mm = mrg_m[0][row];   // wave 0 holds key 0, which every row sees: finite
#pragma unroll
        for (int w = 1; w < FLASH_WAVES; ++w) mm = fmaxf(mm, mrg_m[w][row]);
        float lsum = 0.0f, o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.0f;
#pragma unroll
        for (int w = 0; w < FLASH_WAVES; ++w) {
            const float wgt = __builtin_amdgcn_exp2f(mrg_m[w][row] - mm);   // a wave that saw nothing: 2^-inf = 0
            lsum += mrg_l[w][row] * wgt;
            const f32x4 a = *reinterpret_cast<const f32x4*>(&mrg_o[w][row][d0]), b = *reinterpret_cast<const f32x4*>(&mrg_o[w][row][d0 + 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] += a[j] * wgt; o[4 + j] += b[j] * wgt; }
        }
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= inv;
        const long obase = (long)(t0 + row / G) * nh * HD + (long)(g * G + row % G) * HD + d0;
        if (hi) {   // prefill tiles: the O-projection GEMM reads bf16 hi + lo
            unsigned ph2[4], pl2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t h0 = f32_to_bf16_rne(o[2 * j]), h1 = f32_to_bf16_rne(o[2 * j + 1]);
                const bf16_t l0 = f32_to_bf16_rne(o[2 * j] - __uint_as_float((unsigned)h0 << 16));
                const bf16_t l1 = f32_to_bf16_rne(o[2 * j + 1] - __uint_as_float((unsigned)h1 << 16));
                ph2[j] = (unsigned)h0 | ((unsigned)h1 << 16);
                pl2[j] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
            *reinterpret_cast<uint4*>(hi + obase) = make_uint4(ph2[0], ph2[1], ph2[2], ph2[3]);
            *reinterpret_cast<uint4*>(lo + obase) = make_uint4(pl2[0], pl2[1], pl2[2], pl2[3]);
        } else {
            *reinterpret_cast<float4*>(attn_out + obase) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(attn_out + obase + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}
template <int G>
static void launch_attention_flash_g(rca_lm* h, int M, const f16_t* kc, const f16_t* vc, hipStream_t st, bf16_t* hi, bf16_t* lo) {
    const rca_lm_config_t& c = h->cfg;
    const float scale = 1.0f / sqrtf((float)c.head_dim);
    const int ntiles = cdiv(M * G, 32);
    // teams per workgroup: 4 (one workgroup per CU) once that still gives every CU of the device a workgroup, else 2, else 1
    if (c.n_kv_heads * cdiv(ntiles, 4) >= h->n_cus)
        lm_attn_flash_kernel<G, 4><<<dim3(c.n_kv_heads, cdiv(ntiles, 4)), 64 * FLASH_WAVES * 4, 0, st>>>(h->stt, h->qkv, kc, vc, h->attn, hi, lo, c.n_heads, c.n_kv_heads, scale, c.n_ctx);
    else if (c.n_kv_heads * cdiv(ntiles, 2) >= h->n_cus)
        lm_attn_flash_kernel<G, 2><<<dim3(c.n_kv_heads, cdiv(ntiles, 2)), 64 * FLASH_WAVES * 2, 0, st>>>(h->stt, h->qkv, kc, vc, h->attn, hi, lo, c.n_heads, c.n_kv_heads, scale, c.n_ctx);
    else
        lm_attn_flash_kernel<G, 1><<<dim3(c.n_kv_heads, ntiles), 64 * FLASH_WAVES, 0, st>>>(h->stt, h->qkv, kc, vc, h->attn, hi, lo, c.n_heads, c.n_kv_heads, scale, c.n_ctx);
}
static void launch_attention_flash(rca_lm* h, int M, const f16_t* kc, const f16_t* vc, hipStream_t st, bf16_t* hi, bf16_t* lo) {
    const int G = h->cfg.n_heads / h->cfg.n_kv_heads;
    if (G == 4) launch_attention_flash_g<4>(h, M, kc, vc, st, hi, lo);
    else if (G == 2) launch_attention_flash_g<2>(h, M, kc, vc, st, hi, lo);
    else launch_attention_flash_g<1>(h, M, kc, vc, st, hi, lo);
}

// split attention on MFMA + merge of the splits, for the M tokens of the current pass
static void launch_attention_mfma(rca_lm* h, int M, int nsp_launch, const f16_t* kc, const f16_t* vc, hipStream_t st,
                                  bf16_t* hi = nullptr, bf16_t* lo = nullptr, bool prefill = false) {
    const rca_lm_config_t& c = h->cfg;
    static const bool flash = !(getenv("RCA_LM_FLASH") && atoi(getenv("RCA_LM_FLASH")) == 0);   // 0: A/B runs against the split kernel
    if (prefill && flash) {   // every pass of the prefill-tile path, whatever its size: pieces of one long eval must use ONE arithmetic
        launch_attention_flash(h, M, kc, vc, st, hi, lo);
        return;
    }
    const int G = c.n_heads / c.n_kv_heads;
    const float scale = 1.0f / sqrtf((float)c.head_dim);
    dim3 agm(c.n_kv_heads, nsp_launch, cdiv(M * G, 32));
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)lm_attn_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTM_LDS);
        (void)hipFuncSetAttribute((const void*)lm_attn_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTM_LDS);
        (void)hipFuncSetAttribute((const void*)lm_attn_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTM_LDS);
        attr_done = true;
    }
    // decode steps: the merge of the splits is fused into the attention launch while the grid is at most one workgroup per CU (the
    // regime the sc1 hand-off is measured for); prefill tiles and longer contexts keep the separate combine launch
    const bool fuse = h->fuse_attn && !hi && agm.z == 1 && c.n_kv_heads * nsp_launch <= 256 && M * G <= 8 && nsp_launch <= ATT_TAG_MAXSP && c.n_kv_heads <= ATT_EPOCH_INTS;
    int* arrive = fuse ? h->att_arrive : nullptr;
#define RCA_ATTN_LAUNCH(GG)                                                                                                                        \
    lm_attn_mfma_kernel<GG><<<agm, 512, ATTM_LDS, st>>>(h->stt, h->qkv, kc, vc, c.n_heads, c.n_kv_heads, h->n_splits, scale, c.n_ctx, h->att_part, \
                                                        arrive, h->attn);                                                                          \
    if (!fuse) lm_attn_mfma_combine_kernel<GG><<<M * c.n_heads, 64, 0, st>>>(h->stt, h->att_part, h->attn, c.n_heads, c.n_kv_heads, h->n_splits, nsp_launch, hi, lo);
    if (G == 4) { RCA_ATTN_LAUNCH(4) }
    else if (G == 2) { RCA_ATTN_LAUNCH(2) }
    else { RCA_ATTN_LAUNCH(1) }
#undef RCA_ATTN_LAUNCH
}

// attention split blocks needed by a pass of m tokens on top of the current context
static int lm_splits_needed(const rca_lm* h, int m) { return std::min(h->n_splits, (h->n_tokens + m + ATT_KEYS - 1) / ATT_KEYS); }

// Enqueue one decode pass over the M <= 2 tokens whose ids / position are already in h->stt (device).
// want_logits: 0 none, 1 last token only, 2 every token (logits_all).
// Per layer: [norm+QKV+RoPE/KV-write] -> [split attention] -> [combine] -> [O proj + residual] -> [norm+gate/up+SwiGLU] -> [down + residual]
// nsp_launch: attention split blocks to launch (>= ceil((n_tokens + M) / ATT_KEYS); later splits exit at once).
static int lm_enqueue_pass(rca_lm* h, int M, int want_logits, hipStream_t st, int nsp_launch, bool skip_embed = false) {
    const rca_lm_config_t& c = h->cfg;
    const int H = c.hidden, QKV = (c.n_heads + 2 * c.n_kv_heads) * c.head_dim, AO = c.n_heads * c.head_dim, F = c.ffn;
    if (M < 1 || M > LM_GEMV_M) return fail(RCA_ERR_ARG, "decode pass of %d tokens", M);
    const GemvPro nopro{nullptr, nullptr, 0.0f, 0};
    GemvRope rope{h->cos_t, h->sin_t, nullptr, nullptr, c.n_heads, c.n_kv_heads, c.n_ctx, 0};
    const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    float* x = h->x;
    if (!skip_embed) lm_embed_kernel<<<M, 256, 0, st>>>(h->stt, h->embed, h->embed_f32, x, H, c.vocab_size);   // (inside a frame graph the previous step's sampler has gathered the rows)
    for (int l = 0; l < c.n_layers; ++l) {
        const LmLayer& L = h->layers[l];
        f16_t* kc = h->kc + (long)l * h->kv_layer_stride;
        f16_t* vc = h->vc + (long)l * h->kv_layer_stride;
        rope.kc = kc; rope.vc = vc;
        launch_gemv<1, 2>(GEMV_QKV, h, M, L.qkv, nullptr, h->qkv, L.qkv.N, H, QKV, GemvPro{x, L.attn_norm, c.rms_eps, 0}, rope, st);
        if (L.split_v) {   // the V projection of this layer is kept in another format than Q / K (a Q4_K_M file): its own launch
            rope.row_base = L.qkv.N;
            launch_gemv<1, 2>(GEMV_QKV, h, M, L.vseg, nullptr, h->qkv, L.vseg.N, H, QKV, GemvPro{x, L.attn_norm, c.rms_eps, 0}, rope, st);
            rope.row_base = 0;
        }
        launch_attention_mfma(h, M, nsp_launch, kc, vc, st);
        launch_gemv<0, 3>(GEMV_O, h, M, L.o, h->attn, x, H, AO, H, nopro, norope, st);
        launch_gemv<1, 1>(GEMV_GU, h, M, L.gu, nullptr, h->hbuf, 2 * F, H, F, GemvPro{x, L.ffn_norm, c.rms_eps, 0}, norope, st);
        launch_gemv<0, 3>(GEMV_DOWN, h, M, L.down, h->hbuf, x, H, F, H, nopro, norope, st);
    }
    if (want_logits) {
        const int only_last = want_logits == 1 ? 1 : 0;
        launch_gemv<1, 0>(GEMV_HEAD, h, only_last ? 1 : M, h->head, nullptr, h->logits, c.vocab_size, H, c.vocab_size,
                          GemvPro{x, h->final_norm, c.rms_eps, only_last}, norope, st);
    }
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

// One prefill tile (M <= 32 tokens already described by h->stt): projections on bf16 MFMA with hi/lo-split
// activations, attention through the same split-KV kernels.  Leaves the residual stream in h->x.
// ------------------------------------------------------------------ prefill GEMM, 128 weight rows x 128 tokens
// Y[tok][n] = sum_k W[n][k] * (xh + xl)[tok][k] on v_mfma_f32_32x32x16_bf16, LDS-staged and double-buffered.
// Workgroup = 4 waves in a 2 x 2 grid, each wave 64 rows x 64 tokens (2 x 2 MFMA tiles, hi and lo passes share the
// weight fragment).  A stage is 32 k: three 128 x 32 bf16 tiles (W, xh, xl) fetched with 16-byte loads one stage
// ahead and kept in LDS rows of 40 bf16 (80 B: the 16-byte fragment reads of 8 consecutive rows hit 8 distinct
// 4-bank groups).  Weights are read from HBM once per 128 tokens instead of once per 32.
// grid (N / 128, k splits): with one split the epilogue is fused (RoPE + KV write, SwiGLU + hi/lo split, residual
// add); with several (the narrow N = hidden projections, to put more than 16 workgroups on the chip) each split
// stores its partial sums and lm_gemm128_epilogue_kernel adds them in split order and runs the epilogue -- deterministic.
#define G128_PITCH 40
#define G128_LDS_T(NT) (2 * (NT) * 128 * G128_PITCH * 2)
#define G128_LDS G128_LDS_T(3)
// Weight formats other than bf16 (WF): the matrix is kept ONCE, in the format the decode GEMV streams, and de-quantised while it is
// staged -- a thread turns its 16 weights of the stage into f32 (fp16: widened; q8_0: d * q, exact in f32) and splits each into
// bf16 hi + bf16 lo (hi = RNE(w), lo = RNE(w - hi): exact for fp16 values, within 2^-17 for d * q), written to a fourth LDS tile.
// Three MFMAs per k step instead of two: w_hi x_hi + w_hi x_lo + w_lo x_hi (the dropped w_lo x_lo term is below the 16 bits the
// activation split keeps anyway).  So prefill and decode see the same weights to 2^-17, whatever the format.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even
    const bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split_w8(const float (&w)[8], uint4& hi, uint4& lo) {
    unsigned ph[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ph[j] = pack_bf16x2(w[2 * j], w[2 * j + 1]);
        pl[j] = pack_bf16x2(w[2 * j] - bf16_lo(ph[j]), w[2 * j + 1] - bf16_hi(ph[j]));
    }
    hi = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    lo = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}
template <int EPI, int WF>
__global__ __launch_bounds__(256, 2) void lm_gemm128_kernel(const LmDevState* __restrict__ stt, const bf16_t* __restrict__ W, GemvQ8 q8,
                                                            const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl, int N, int K,
                                                            int kslice, float* __restrict__ y, int ldy, bf16_t* __restrict__ oh,
                                                            bf16_t* __restrict__ ol, float* __restrict__ part, GemvRope rope, int nseq) {
    // nseq > 1: this workgroup walks nseq consecutive k slices itself.  Each slice is summed into a fresh accumulator and that is
    // added to a running total -- exactly the additions, in exactly the order, of "every slice its own workgroup, then
    // lm_gemm128_epilogue_kernel adds the partial sums starting from 0" -- so the result does not depend on which of the two forms a
    // pass used, and the partial sums never travel through HBM.  Used where the token blocks alone fill the chip.
    constexpr int NT = WF == WF_BF16 ? 3 : 4;   // LDS tiles per buffer: W (hi), xh, xl [, W lo]
    extern __shared__ __attribute__((aligned(16))) bf16_t g128_lds[];   // [2 buffers][NT][128 rows][G128_PITCH]
    typedef bf16_t tile_t[NT][128 * G128_PITCH];
    tile_t* sm = reinterpret_cast<tile_t*>(g128_lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 128;
    const int ks = blockIdx.y * nseq * kslice;
    const int tb = blockIdx.z * 128;   // token block of this workgroup: a pass holds up to LM_MAXM / 128 of them, the weights are
                                       // fetched from HBM by the first and served from L2 / Infinity Cache to the others
    const int spf = kslice >> 5;          // stages per slice
    const int nstage = nseq * spf;
    // staging role: two 16-byte chunks per tile and thread: rows c >> 2, k offset (c & 3) * 8
    const int srow0 = tid >> 2, skc = (tid & 3) * 8;
    const bf16_t* gW = W + (long)(n0 + srow0) * K + ks + skc;
    const bf16_t* gH = xh + (long)(tb + srow0) * K + ks + skc;
    const bf16_t* gL = xl + (long)(tb + srow0) * K + ks + skc;
    const long rstep = 64L * K;     // second chunk: row + 64
    const int soff0 = srow0 * G128_PITCH + skc, soff1 = soff0 + 64 * G128_PITCH;
    // q8_0: a thread's stage is ONE 16-byte unit of the packed layout = 8 k of the two rows of pair (n0 / 2 + tid / 4); the pair's
    // rows are (2p, 2p + 1), or (d, d + 32) of one head in the fused QKV matrix (the only matrix that runs the RoPE epilogue)
    const int qp = tid >> 2;
    const long q_nkb = K >> 5;
    const u32x4* gQ = (WF == WF_Q8 || WF == WF_Q6K) ? q8.qs + ((long)(n0 >> 1) + qp) * (K >> 3) + (ks >> 3) + (tid & 3) : nullptr;
    const unsigned* gS = WF == WF_Q8 ? q8.sc + q8_sc_index((n0 >> 1) + qp, ks >> 5, q_nkb) : nullptr;   // next k block: + 8
    const float* gS6 = WF == WF_Q6K ? reinterpret_cast<const float*>(q8.sc) + 2 * q8_sc_index((n0 >> 1) + qp, (ks >> 4) + ((tid & 3) >> 1), K >> 4) : nullptr;   // next stage: + 2 groups of 16
    // Q4_K: a thread's stage is HALF a 16-byte unit = 8 k of two adjacent slots: quad tid / 8, chunk (tid / 2) % 4, slots 2 (tid % 2) + {0, 1}
    const int q4quad = tid >> 3, q4half = tid & 1, q4slot = 4 * ((n0 >> 2) + q4quad) + 2 * q4half;
    const uint2* gQ4 = WF == WF_Q4K ? reinterpret_cast<const uint2*>(q8.qs + ((long)(n0 >> 2) + q4quad) * (K >> 3) + (ks >> 3) + ((tid >> 1) & 3)) + q4half : nullptr;
    const unsigned short* gS4 = WF == WF_Q4K ? reinterpret_cast<const unsigned short*>(q8.sc) + q4k_scm_index(q4slot, ks >> 5, q_nkb) : nullptr;   // next sub-block: + 16
    const unsigned* gD4 = WF == WF_Q4K ? q8.dd + q4k_scm_index(q4slot, 0, K >> 8) : nullptr;                                                     // super-block k / 256: + 16 each
    int q4row[2];   // LDS rows of the two slots
#pragma unroll
    for (int i = 0; i < 2; ++i) q4row[i] = (int)packed_slot_row(4 * q4quad + 2 * q4half + i, EPI == GEMM_EPI_ROPE);
    const int q4soff0 = q4row[0] * G128_PITCH + ((tid >> 1) & 3) * 8, q4soff1 = q4row[1] * G128_PITCH + ((tid >> 1) & 3) * 8;
    const int qra = EPI == GEMM_EPI_ROPE ? (qp >> 5) * 64 + (qp & 31) : 2 * qp;
    const int qsoff0 = qra * G128_PITCH + skc, qsoff1 = qsoff0 + (EPI == GEMM_EPI_ROPE ? 32 : 1) * G128_PITCH;
    // Software pipeline.  A workgroup's stage needs 8 KB of weights straight from HBM (~2 us away) and 16 KB of
    // activations from L2 (~0.7 us): weight chunks are requested DW stages ahead, activation chunks DX stages ahead,
    // both held in registers until their LDS buffer is free.
    constexpr int DW = 1, DX = 1;   // measured: deeper register prefetch (4 / 2) is slower at 2 workgroups per CU
    uint4 rw[DW][2], rh[DX][2], rl[DX][2];
    unsigned rs[DW];
    uint2 rdd[DW];
    auto gload_w = [&](int slot, int s) {
        const int st = min(s, nstage - 1);
        if (WF == WF_Q4K) {
            const uint2 t = gQ4[(long)st * 8];                       // + 4 units of 16 bytes per stage
            rw[slot][0] = make_uint4(t.x, t.y, 0u, 0u);
            rs[slot] = *reinterpret_cast<const unsigned*>(gS4 + (long)st * 16);          // (sc | m << 8) of the two slots
            rdd[slot] = *reinterpret_cast<const uint2*>(gD4 + (long)(((ks >> 5) + st) >> 3) * 16);   // (d | dmin << 16) of the two slots
        } else if (WF == WF_Q6K) {
            const u32x4 t = gQ[st * 4];
            rw[slot][0] = make_uint4(t.x, t.y, t.z, t.w);
            rdd[slot] = *reinterpret_cast<const uint2*>(gS6 + (long)st * 2 * 16);   // (s_a, s_b): two groups of 16 per stage, 8 pairs x 2 floats per group
        } else if (WF == WF_Q8) {
            const u32x4 t = gQ[st * 4];
            rw[slot][0] = make_uint4(t.x, t.y, t.z, t.w);
            rs[slot] = gS[(long)st * 8];
        } else {
            const int k = st << 5;
            rw[slot][0] = *reinterpret_cast<const uint4*>(gW + k); rw[slot][1] = *reinterpret_cast<const uint4*>(gW + rstep + k);
        }
    };
    auto gload_x = [&](int slot, int s) {
        const int k = min(s, nstage - 1) << 5;
        rh[slot][0] = *reinterpret_cast<const uint4*>(gH + k); rh[slot][1] = *reinterpret_cast<const uint4*>(gH + rstep + k);
        rl[slot][0] = *reinterpret_cast<const uint4*>(gL + k); rl[slot][1] = *reinterpret_cast<const uint4*>(gL + rstep + k);
    };
    auto swrite = [&](int buf, int ws, int xs) {
        if (WF == WF_BF16) {
            *reinterpret_cast<uint4*>(&sm[buf][0][soff0]) = rw[ws][0]; *reinterpret_cast<uint4*>(&sm[buf][0][soff1]) = rw[ws][1];
        } else if (WF == WF_F16) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned u[4] = {rw[ws][c].x, rw[ws][c].y, rw[ws][c].z, rw[ws][c].w};
                float wv[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 h2 = __builtin_bit_cast(f16x2, u[j]);
                    wv[2 * j] = (float)h2[0];
                    wv[2 * j + 1] = (float)h2[1];
                }
                uint4 hi, lo;
                split_w8(wv, hi, lo);
                *reinterpret_cast<uint4*>(&sm[buf][0][c ? soff1 : soff0]) = hi;
                *reinterpret_cast<uint4*>(&sm[buf][3][c ? soff1 : soff0]) = lo;
            }
        } else if (WF == WF_Q4K) {   // .x / .y = 8 nibbles of the first / second slot; the value is dequantize_row_q4_K's (d sc) q - (dmin m)
            const unsigned qw[2] = {rw[ws][0].x, rw[ws][0].y}, ddw[2] = {rdd[ws].x, rdd[ws].y};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned scm = (rs[ws] >> (16 * c)) & 0xffffu;
                float wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = q4k_value(ddw[c], scm, (qw[c] >> (4 * j)) & 0xFu);
                uint4 hi, lo;
                split_w8(wv, hi, lo);
                *reinterpret_cast<uint4*>(&sm[buf][0][c ? q4soff1 : q4soff0]) = hi;
                *reinterpret_cast<uint4*>(&sm[buf][3][c ? q4soff1 : q4soff0]) = lo;
            }
        } else {   // q8_0: .x .y = 8 int8 of the pair's first row, .z .w = of its second row; rs = (fp16 d_a, fp16 d_b); Q6_K: rdd = (f32 s_a, f32 s_b)
            const f16x2 d2 = __builtin_bit_cast(f16x2, rs[ws]);
            const unsigned u[4] = {rw[ws][0].x, rw[ws][0].y, rw[ws][0].z, rw[ws][0].w};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float dd = WF == WF_Q6K ? __uint_as_float(c ? rdd[ws].y : rdd[ws].x) : (float)d2[c];
                float wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = dd * (float)(signed char)((u[2 * c + (j >> 2)] >> (8 * (j & 3))) & 0xffu);
                uint4 hi, lo;
                split_w8(wv, hi, lo);
                *reinterpret_cast<uint4*>(&sm[buf][0][c ? qsoff1 : qsoff0]) = hi;
                *reinterpret_cast<uint4*>(&sm[buf][3][c ? qsoff1 : qsoff0]) = lo;
            }
        }
        *reinterpret_cast<uint4*>(&sm[buf][1][soff0]) = rh[xs][0]; *reinterpret_cast<uint4*>(&sm[buf][1][soff1]) = rh[xs][1];
        *reinterpret_cast<uint4*>(&sm[buf][2][soff0]) = rl[xs][0]; *reinterpret_cast<uint4*>(&sm[buf][2][soff1]) = rl[xs][1];
    };
    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.0f; tot[i][j][r] = 0.0f; }
    // fragment read offsets: row (lane & 31) of the 32-row tile, k offset 8 * half inside a 16-k substep
    const int fa = (wr * 64 + (lane & 31)) * G128_PITCH + 8 * half;
    const int fb = (wc * 64 + (lane & 31)) * G128_PITCH + 8 * half;
#pragma unroll
    for (int d = 0; d < DW; ++d) gload_w(d, d);
#pragma unroll
    for (int d = 0; d < DX; ++d) gload_x(d, d);
    swrite(0, 0, 0);
    __syncthreads();
    // stage s: weights in slot s % DW, activations in slot s % DX; the loop is unrolled by DW (a multiple of DX)
    for (int s0 = 0; s0 < nstage; s0 += DW) {
#pragma unroll
        for (int d = 0; d < DW; ++d) {
            const int s = s0 + d;
            if (s >= nstage) break;
            const int buf = s & 1;
            // slots of stage s are in LDS: refill them with the stages DW / DX ahead
            gload_w(d, s + DW);
            gload_x(d % DX, s + DX);
#pragma unroll
            for (int ksub = 0; ksub < 2; ++ksub) {
                bf16x8 a[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const bf16x8*>(&sm[buf][0][fa + i * 32 * G128_PITCH + ksub * 16]);
                    if (WF != WF_BF16) al[i] = *reinterpret_cast<const bf16x8*>(&sm[buf][NT - 1][fa + i * 32 * G128_PITCH + ksub * 16]);
                    bh[i] = *reinterpret_cast<const bf16x8*>(&sm[buf][1][fb + i * 32 * G128_PITCH + ksub * 16]);
                    bl[i] = *reinterpret_cast<const bf16x8*>(&sm[buf][2][fb + i * 32 * G128_PITCH + ksub * 16]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
                        if (WF != WF_BF16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            if (nseq > 1 && (s + 1) % spf == 0) {   // end of a slice: fold it into the running total (0 + p0, then + p1, ...)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            tot[i][j][r] = tot[i][j][r] + acc[i][j][r];
                            acc[i][j][r] = 0.0f;
                        }
            }
            if (s + 1 < nstage) swrite(buf ^ 1, (d + 1) % DW, (d + 1) % DX);
            __syncthreads();
        }
    }
    if (nseq > 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = tot[i][j];
    }
    // epilogue.  C layout: token = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * half inside a 32 x 32 tile
    const int Mv = stt->m;
    const int nsplit = gridDim.y;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tok = tb + wc * 64 + j * 32 + (lane & 31);
        if (tok >= Mv) continue;
        if (nsplit > 1) {   // partial sums of this k slice, token-contiguous [split][n][LM_MAXM]: lanes are tokens -> 128-byte runs
            float* p = part + ((long)blockIdx.y * N + n0 + wr * 64) * LM_MAXM + tok;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) p[(long)(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LM_MAXM] = acc[i][j][r];
        } else if (EPI == GEMM_EPI_RESID) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    y[(long)tok * ldy + n] = y[(long)tok * ldy + n] + acc[i][j][r];
                }
        } else if (EPI == GEMM_EPI_ROPE) {
            // the wave's 64 rows are one head: rows d (tile 0) and d + 32 (tile 1) sit in the same register slot
            const int pos = stt->n_tokens + tok;
            if (pos >= rope.n_ctx) continue;
            const int head = (n0 + wr * 64 + rope.row_base) >> 6;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2) + 4 * half;   // < 32
                const float x1 = acc[0][j][r], x2 = acc[1][j][r];
                if (head < rope.nh + rope.nkv) {
                    const float c = rope.cos_t[(long)pos * 32 + d], sn = rope.sin_t[(long)pos * 32 + d];
                    const float o1 = x1 * c + (-x2) * sn;
                    const float o2 = x2 * c + x1 * sn;
                    if (head < rope.nh) {
                        y[(long)tok * ldy + head * 64 + d] = o1;
                        y[(long)tok * ldy + head * 64 + d + 32] = o2;
                    } else {
                        f16_t* kp = rope.kc + ((long)pos * rope.nkv + (head - rope.nh)) * 64;
                        kp[d] = (f16_t)o1;
                        kp[d + 32] = (f16_t)o2;
                    }
                } else {
                    f16_t* vp = rope.vc + ((long)pos * rope.nkv + (head - rope.nh - rope.nkv)) * 64;
                    vp[d] = (f16_t)x1;
                    vp[d + 32] = (f16_t)x2;
                }
            }
        } else {   // SwiGLU: rows (2i, 2i+1) = (gate_i, up_i) sit in registers (2q, 2q+1)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int rr = ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * half;
                    const int fi = (n0 + wr * 64 + i * 32 + rr) >> 1;
                    const float g = acc[i][j][2 * q], u = acc[i][j][2 * q + 1];
                    const float hv = (g / (1.0f + __expf(-g))) * u;
                    const bf16_t hb = f32_to_bf16_rne(hv);
                    oh[(long)tok * ldy + fi] = hb;
                    ol[(long)tok * ldy + fi] = f32_to_bf16_rne(hv - __uint_as_float((unsigned)hb << 16));
                }
        }
    }
}
// Epilogue over the k-split partial sums part[split][n][LM_MAXM] (added in split order, starting from 0).  A block
// takes 64 rows x 32 tokens: it sums the splits with token-contiguous reads, turns the tile through LDS and runs the
// fused epilogue with row-contiguous writes -- residual add, RoPE + KV write (the 64 rows are one head), or SwiGLU +
// hi/lo split (the 64 rows are 32 interleaved gate/up pairs).
template <int EPI>
__global__ __launch_bounds__(256) void lm_gemm128_epilogue_kernel(const LmDevState* __restrict__ stt, const float* __restrict__ part, int nsplit, int grp, int N,
                                                                  float* __restrict__ y, int ldy, bf16_t* __restrict__ oh, bf16_t* __restrict__ ol,
                                                                  GemvRope rope) {
    __shared__ float tile[64][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty 0..7
    const int n0 = blockIdx.x * 64, t0 = blockIdx.y * 32;
    const int Mv = stt->m;
    if (t0 >= Mv) return;
    {
        float v[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) v[rr] = 0.0f;
        const float* p0 = part + ((long)n0 + ty) * LM_MAXM + t0 + tx;
        // The sum over the k slices is DEFINED in two levels: groups of `grp` consecutive slices are folded first (each from 0), the
        // group sums are folded in group order (from 0).  A pass stores either every slice (grp > 1 here) or the group sums its
        // workgroups folded themselves while walking `grp` slices (grp == 1 here: 0 + G == G) -- the same additions either way.
        for (int s0 = 0; s0 < nsplit; s0 += grp) {
            float g[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) g[rr] = 0.0f;
#pragma unroll 4
            for (int s = s0; s < s0 + grp; ++s) {   // eight independent row sums per thread: eight loads in flight per slice
                const float* ps = p0 + (long)s * N * LM_MAXM;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) g[rr] += ps[(long)rr * 8 * LM_MAXM];
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) v[rr] += g[rr];
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) tile[rr * 8 + ty][tx] = v[rr];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int tl = tt * 8 + ty;
        const int tok = t0 + tl;
        if (tok >= Mv) continue;
        if (EPI == GEMM_EPI_RESID) {
            float* yr = y + (long)tok * ldy + n0;
            yr[tx] = yr[tx] + tile[tx][tl];
            yr[tx + 32] = yr[tx + 32] + tile[tx + 32][tl];
        } else if (EPI == GEMM_EPI_ROPE) {
            const int pos = stt->n_tokens + tok;
            if (pos >= rope.n_ctx) continue;
            const int head = (n0 + rope.row_base) >> 6, d = tx;
            const float x1 = tile[d][tl], x2 = tile[d + 32][tl];
            if (head < rope.nh + rope.nkv) {
                const float c = rope.cos_t[(long)pos * 32 + d], sn = rope.sin_t[(long)pos * 32 + d];
                const float o1 = x1 * c + (-x2) * sn;
                const float o2 = x2 * c + x1 * sn;
                if (head < rope.nh) {
                    y[(long)tok * ldy + head * 64 + d] = o1;
                    y[(long)tok * ldy + head * 64 + d + 32] = o2;
                } else {
                    f16_t* kp = rope.kc + ((long)pos * rope.nkv + (head - rope.nh)) * 64;
                    kp[d] = (f16_t)o1;
                    kp[d + 32] = (f16_t)o2;
                }
            } else {
                f16_t* vp = rope.vc + ((long)pos * rope.nkv + (head - rope.nh - rope.nkv)) * 64;
                vp[d] = (f16_t)x1;
                vp[d + 32] = (f16_t)x2;
            }
        } else {   // rows (2i, 2i+1) = (gate_i, up_i)
            const int F = N >> 1;
            const int fi = (n0 >> 1) + tx;
            const float g = tile[2 * tx][tl], u = tile[2 * tx + 1][tl];
            const float hv = (g / (1.0f + __expf(-g))) * u;
            const bf16_t hb = f32_to_bf16_rne(hv);
            oh[(long)tok * F + fi] = hb;
            ol[(long)tok * F + fi] = f32_to_bf16_rne(hv - __uint_as_float((unsigned)hb << 16));
        }
    }
}

static bool lm_all_bf16(const rca_lm* h) {
    for (const LmLayer& L : h->layers)
        for (const WMat* m : {&L.qkv, &L.o, &L.gu, &L.down})
            if (m->fmt != WF_BF16 || L.split_v) return false;
    return true;
}
static bool lm_can_gemm128(const rca_lm* h);
// the 32-token tiles read bf16 fragments straight from HBM: bf16 models only; the 128-token tiles de-quantise any format while staging
static bool lm_can_mfma_prefill(const rca_lm* h) {
    const rca_lm_config_t& c = h->cfg;
    const int AO = c.n_heads * c.head_dim;
    if (lm_can_gemm128(h)) return true;
    return lm_all_bf16(h) && c.hidden % 64 == 0 && AO % 64 == 0 && c.ffn % 64 == 0 && (2 * c.ffn) % 32 == 0;
}
static int lm_enqueue_prefill_tile(rca_lm* h, int M, hipStream_t st, int nsp_launch) {
    const rca_lm_config_t& c = h->cfg;
    const int H = c.hidden, QKV = (c.n_heads + 2 * c.n_kv_heads) * c.head_dim, AO = c.n_heads * c.head_dim, F = c.ffn;
    GemvRope rope{h->cos_t, h->sin_t, nullptr, nullptr, c.n_heads, c.n_kv_heads, c.n_ctx, 0};
    const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    float* x = h->x;
    lm_embed_kernel<<<M, 256, 0, st>>>(h->stt, h->embed, h->embed_f32, x, H, c.vocab_size);
    for (int l = 0; l < c.n_layers; ++l) {
        const LmLayer& L = h->layers[l];
        f16_t* kc = h->kc + (long)l * h->kv_layer_stride;
        f16_t* vc = h->vc + (long)l * h->kv_layer_stride;
        rope.kc = kc; rope.vc = vc;
        lm_add_rmsnorm_kernel<<<M, 64, 0, st>>>(h->stt, x, nullptr, nullptr, 0, 0, L.attn_norm, h->xn, H, c.rms_eps);
        lm_split_bf16_kernel<<<dim3(cdiv(H, 256), M), 256, 0, st>>>(h->stt, h->xn, h->xh, h->xl, H);
        lm_gemm_mfma_kernel<GEMM_EPI_ROPE><<<QKV / 32, 256, 0, st>>>(h->stt, L.qkv.w, h->xh, h->xl, QKV, H, h->qkv, QKV, nullptr, nullptr, rope);
        launch_attention_mfma(h, M, nsp_launch, kc, vc, st, nullptr, nullptr, true);
        lm_split_bf16_kernel<<<dim3(cdiv(AO, 256), M), 256, 0, st>>>(h->stt, h->attn, h->xh, h->xl, AO);
        lm_gemm_mfma_kernel<GEMM_EPI_RESID><<<H / 32, 256, 0, st>>>(h->stt, L.o.w, h->xh, h->xl, H, AO, x, H, nullptr, nullptr, norope);
        lm_add_rmsnorm_kernel<<<M, 64, 0, st>>>(h->stt, x, nullptr, nullptr, 0, 0, L.ffn_norm, h->xn, H, c.rms_eps);
        lm_split_bf16_kernel<<<dim3(cdiv(H, 256), M), 256, 0, st>>>(h->stt, h->xn, h->xh, h->xl, H);
        // SwiGLU epilogue writes the hi/lo split of h straight into the (ffn-wide) split buffers of the down projection:
        // it reads xh/xl [M][H] and writes [M][F] -- distinct regions are needed, so h goes to the second half of hbuf
        bf16_t* hh = reinterpret_cast<bf16_t*>(h->hbuf);
        bf16_t* hl = hh + (long)LM_MAXM * F;
        lm_gemm_mfma_kernel<GEMM_EPI_SWIGLU><<<2 * F / 32, 256, 0, st>>>(h->stt, L.gu.w, h->xh, h->xl, 2 * F, H, nullptr, F, hh, hl, norope);
        lm_gemm_mfma_kernel<GEMM_EPI_RESID><<<H / 32, 256, 0, st>>>(h->stt, L.down.w, hh, hl, H, F, x, H, nullptr, nullptr, norope);
    }
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}
// Every 128-row GEMM is cut along k until ~512 workgroups per token block are in flight (a workgroup's stage is one exposed
// HBM round trip: parallel slices are what hides it), slices of >= 256 k, powers of two so they divide K / 32.  The count
// depends on the matrix only -- never on the tokens of the pass -- so any tiling of a prompt adds the same partial sums.
static int g128_splits(int N, int K) {
    int ns = 1;
    while (ns * (N / 128) < 512 && K % (ns * 2 * 32) == 0 && K / (ns * 2) >= 256) ns *= 2;
    return ns;
}
// Two-level definition of the sum over the ns slices (lm_gemm128_epilogue_kernel): at most four groups, folded first.  A function of the
// matrix only, like ns.  It lets a LONG pass store four group sums per output instead of ns partial sums (the narrow projections wrote
// and re-read 0.9 GB of partials per layer and 1024-token pass: down had 32 slices) without changing a bit of what a short pass
// computes from all ns.
static int g128_group(int ns) { return ns / std::min(ns, 4); }
struct G128Form { int grid_y, nseq, ep_nsplit, ep_grp; };   // ep_nsplit == 0: the kernel's own (fused) epilogue
static G128Form g128_form(int N, int ns, int tbz, int seq_min) {
    const int grp = g128_group(ns), ng = ns / grp;
    if (ns == 1) return {1, 1, 0, 1};
    if (grp == 1 && seq_min > 0 && (N / 128) * tbz >= seq_min) return {1, ns, 0, 1};            // every workgroup walks all slices (gate/up)
    if (grp > 1 && seq_min > 0 && (N / 128) * tbz * ng >= seq_min) return {ng, grp, ng, 1};     // workgroups walk a group, group sums stored
    return {ns, 1, ns, grp};                                                                     // every slice stored
}
static bool lm_can_gemm128(const rca_lm* h) {
    const rca_lm_config_t& c = h->cfg;
    const int H = c.hidden, QKV = (c.n_heads + 2 * c.n_kv_heads) * c.head_dim, AO = c.n_heads * c.head_dim, F = c.ffn;
    const int G = c.n_heads / c.n_kv_heads;
    for (const LmLayer& L : h->layers)   // a V projection kept apart from [q; k] (formats differ): both parts are tiled by 128 rows
        if (L.split_v && ((L.qkv.N % 128) || (L.vseg.N % 128))) return false;
    return c.head_dim == 64 && (G == 1 || G == 2 || G == 4) && H % 128 == 0 && QKV % 128 == 0 && (2 * F) % 128 == 0 && AO % 32 == 0 && F % 32 == 0;
}
// one 128-row GEMM launch in the format the matrix is kept in
template <int EPI>
static void launch_gemm128(rca_lm* h, const WMat& w, dim3 grid, hipStream_t st, const bf16_t* xh, const bf16_t* xl, int N, int K, int kslice,
                           float* y, int ldy, bf16_t* oh, bf16_t* ol, GemvRope rope, int nseq) {
    static bool attr_done = false;
    if (!attr_done) {   // the four-tile variants need 80 KB of LDS
        (void)hipFuncSetAttribute((const void*)lm_gemm128_kernel<EPI, WF_Q8>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS_T(4));
        (void)hipFuncSetAttribute((const void*)lm_gemm128_kernel<EPI, WF_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS_T(4));
        (void)hipFuncSetAttribute((const void*)lm_gemm128_kernel<EPI, WF_Q4K>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS_T(4));
        (void)hipFuncSetAttribute((const void*)lm_gemm128_kernel<EPI, WF_Q6K>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS_T(4));
        attr_done = true;
    }
    const GemvQ8 qa{w.qs, w.sc, w.dd};
    if (w.fmt == WF_Q6K)
        lm_gemm128_kernel<EPI, WF_Q6K><<<grid, 256, G128_LDS_T(4), st>>>(h->stt, w.w, qa, xh, xl, N, K, kslice, y, ldy, oh, ol, h->gpart, rope, nseq);
    else if (w.fmt == WF_Q4K)
        lm_gemm128_kernel<EPI, WF_Q4K><<<grid, 256, G128_LDS_T(4), st>>>(h->stt, w.w, qa, xh, xl, N, K, kslice, y, ldy, oh, ol, h->gpart, rope, nseq);
    else if (w.fmt == WF_Q8)
        lm_gemm128_kernel<EPI, WF_Q8><<<grid, 256, G128_LDS_T(4), st>>>(h->stt, w.w, qa, xh, xl, N, K, kslice, y, ldy, oh, ol, h->gpart, rope, nseq);
    else if (w.fmt == WF_F16)
        lm_gemm128_kernel<EPI, WF_F16><<<grid, 256, G128_LDS_T(4), st>>>(h->stt, w.w, qa, xh, xl, N, K, kslice, y, ldy, oh, ol, h->gpart, rope, nseq);
    else
        lm_gemm128_kernel<EPI, WF_BF16><<<grid, 256, G128_LDS_T(3), st>>>(h->stt, w.w, qa, xh, xl, N, K, kslice, y, ldy, oh, ol, h->gpart, rope, nseq);
}
static int lm_enqueue_prefill_tile128(rca_lm* h, int M, hipStream_t st, int nsp_launch) {
    const rca_lm_config_t& c = h->cfg;
    const int H = c.hidden, QKV = (c.n_heads + 2 * c.n_kv_heads) * c.head_dim, AO = c.n_heads * c.head_dim, F = c.ffn;
    GemvRope rope{h->cos_t, h->sin_t, nullptr, nullptr, c.n_heads, c.n_kv_heads, c.n_ctx, 0};
    const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    const int so = g128_splits(H, AO), sg = g128_splits(2 * F, H), sd = g128_splits(H, F);
    const int tbz = cdiv(M, 128);   // token blocks of this pass
    // A projection whose token blocks alone put `seq_min` workgroups on the chip is run with every workgroup walking the k slices itself
    // (same sums in the same order, no partial sums through HBM, no epilogue launch); RCA_LM_SEQ_MIN_WGS overrides the threshold
    // (0 = never) for A/B runs.
    static const int seq_min = getenv("RCA_LM_SEQ_MIN_WGS") ? atoi(getenv("RCA_LM_SEQ_MIN_WGS")) : 512;
    const G128Form fo = g128_form(H, so, tbz, seq_min), fg = g128_form(2 * F, sg, tbz, seq_min), fd = g128_form(H, sd, tbz, seq_min);
    float* x = h->x;
    lm_embed_kernel<<<M, 256, 0, st>>>(h->stt, h->embed, h->embed_f32, x, H, c.vocab_size);
    for (int l = 0; l < c.n_layers; ++l) {
        const LmLayer& L = h->layers[l];
        f16_t* kc = h->kc + (long)l * h->kv_layer_stride;
        f16_t* vc = h->vc + (long)l * h->kv_layer_stride;
        rope.kc = kc; rope.vc = vc;
        lm_add_rmsnorm_kernel<<<M, 64, 0, st>>>(h->stt, x, nullptr, nullptr, 0, 0, L.attn_norm, h->xn, H, c.rms_eps, h->xh, h->xl);
        for (int seg = 0; seg < (L.split_v ? 2 : 1); ++seg) {   // [q; k; v] as one matrix, or [q; k] and v when their formats differ
            const WMat& w = seg ? L.vseg : L.qkv;
            const int Ns = w.N, ss = g128_splits(Ns, H);
            const G128Form fq = g128_form(Ns, ss, tbz, seq_min);
            rope.row_base = seg ? L.qkv.N : 0;
            launch_gemm128<GEMM_EPI_ROPE>(h, w, dim3(Ns / 128, fq.grid_y, tbz), st, h->xh, h->xl, Ns, H, H / ss, h->qkv, QKV, nullptr, nullptr, rope, fq.nseq);
            if (fq.ep_nsplit) lm_gemm128_epilogue_kernel<GEMM_EPI_ROPE><<<dim3(Ns / 64, cdiv(M, 32)), 256, 0, st>>>(h->stt, h->gpart, fq.ep_nsplit, fq.ep_grp, Ns, h->qkv, QKV, nullptr, nullptr, rope);
        }
        rope.row_base = 0;
        launch_attention_mfma(h, M, nsp_launch, kc, vc, st, h->xh, h->xl, true);
        launch_gemm128<GEMM_EPI_RESID>(h, L.o, dim3(H / 128, fo.grid_y, tbz), st, h->xh, h->xl, H, AO, AO / so, x, H, nullptr, nullptr, norope, fo.nseq);
        if (fo.ep_nsplit) lm_gemm128_epilogue_kernel<GEMM_EPI_RESID><<<dim3(H / 64, cdiv(M, 32)), 256, 0, st>>>(h->stt, h->gpart, fo.ep_nsplit, fo.ep_grp, H, x, H, nullptr, nullptr, norope);
        lm_add_rmsnorm_kernel<<<M, 64, 0, st>>>(h->stt, x, nullptr, nullptr, 0, 0, L.ffn_norm, h->xn, H, c.rms_eps, h->xh, h->xl);
        // SwiGLU epilogue writes the hi/lo split of h straight into the (ffn-wide) split buffers of the down projection:
        // it reads xh/xl [M][H] and writes [M][F] -- distinct regions are needed, so h goes to the second half of hbuf
        bf16_t* hh = reinterpret_cast<bf16_t*>(h->hbuf);
        bf16_t* hl = hh + (long)LM_MAXM * F;
        launch_gemm128<GEMM_EPI_SWIGLU>(h, L.gu, dim3(2 * F / 128, fg.grid_y, tbz), st, h->xh, h->xl, 2 * F, H, H / sg, nullptr, F, hh, hl, norope, fg.nseq);
        if (fg.ep_nsplit) lm_gemm128_epilogue_kernel<GEMM_EPI_SWIGLU><<<dim3(2 * F / 64, cdiv(M, 32)), 256, 0, st>>>(h->stt, h->gpart, fg.ep_nsplit, fg.ep_grp, 2 * F, nullptr, 0, hh, hl, norope);
        launch_gemm128<GEMM_EPI_RESID>(h, L.down, dim3(H / 128, fd.grid_y, tbz), st, hh, hl, H, F, F / sd, x, H, nullptr, nullptr, norope, fd.nseq);
        if (fd.ep_nsplit) lm_gemm128_epilogue_kernel<GEMM_EPI_RESID><<<dim3(H / 64, cdiv(M, 32)), 256, 0, st>>>(h->stt, h->gpart, fd.ep_nsplit, fd.ep_grp, H, x, H, nullptr, nullptr, norope);
    }
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

// an rca_lm_eval_async pass may still own the pinned staging block and the activations: wait for it before anything else runs
static int lm_settle(rca_lm* h) {
    if (h->async_pending) {
        RCA_HIP(hipSetDevice(h->device));
        RCA_HIP(hipStreamSynchronize(h->stream));
        h->async_pending = false;
    }
    return RCA_OK;
}
static int lm_push_state(rca_lm* h, const int32_t* ids, int m, hipStream_t st) {
    h->h_stt->n_tokens = h->n_tokens;
    h->h_stt->m = m;
    for (int i = 0; i < m; ++i) h->h_stt->ids[i] = ids[i];
    // n_tokens, m and the m ids only; rng counter / out_token stay device-owned
    RCA_HIP(hipMemcpyAsync(h->stt, h->h_stt, 8 + 4 * (size_t)std::max(m, 16), hipMemcpyHostToDevice, st));
    return RCA_OK;
}

extern "C" int rca_lm_reset(rca_lm_t* h) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    h->n_tokens = 0;
    h->logits_rows = 0;
    return RCA_OK;
}
extern "C" int rca_lm_get_n_tokens(const rca_lm_t* h, int32_t* n) {
    if (!h || !n) return fail(RCA_ERR_ARG, "null");
    *n = h->n_tokens;
    return RCA_OK;
}
extern "C" int rca_lm_set_n_tokens(rca_lm_t* h, int32_t n) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    if (n < 0 || n > h->cfg.n_ctx) return fail(RCA_ERR_ARG, "n_tokens %d outside [0, %d]", n, h->cfg.n_ctx);
    h->n_tokens = n;
    return RCA_OK;
}
extern "C" int rca_lm_sync(rca_lm_t* h) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipSetDevice(h->device));
    RCA_HIP(hipStreamSynchronize(h->stream));
    h->async_pending = false;
    return RCA_OK;
}

// as_prefill: use the prefill tiles even for n <= LM_PREFILL_MIN (pieces of one long eval).
// wait_last = false: return as soon as the LAST pass / tile is enqueued (everything before it has completed, because the
// pinned staging block is reused per pass); the next call on the handle waits for it first.
static int lm_eval_impl(rca_lm_t* h, const int32_t* ids, int32_t n, bool wait_last, bool as_prefill) {
    if (!h || (!ids && n > 0) || n < 0) return fail(RCA_ERR_ARG, "eval: bad argument");
    if (n == 0) return RCA_OK;
    if (h->n_tokens + n > h->cfg.n_ctx) return fail(RCA_ERR_STATE, "context overflow: %d + %d > n_ctx %d", h->n_tokens, n, h->cfg.n_ctx);
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "eval: token id %d at index %d is outside the vocabulary [0, %d)", ids[i], i, h->cfg.vocab_size);
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    int rc;
    if (h->async_pending) { RCA_HIP(hipStreamSynchronize(st)); h->async_pending = false; }
    const bool all = h->cfg.logits_all != 0;
    if (all && n > h->logits_rows_cap) {
        RCA_HIP(hipStreamSynchronize(st));
        lm_drop_graphs(h);   // their head-GEMV and sampler nodes point at the buffer that is about to be freed
        (void)hipFree(h->logits);
        h->logits = nullptr;
        h->logits_rows_cap = n;
        if ((rc = lm_alloc((void**)&h->logits, (size_t)n * h->cfg.vocab_size * 4)) != RCA_OK) return rc;
    }
    float* logits_base = h->logits;
    if (!all && (n > LM_PREFILL_MIN || as_prefill) && h->mfma_prefill && lm_can_mfma_prefill(h)) {
        // long evals (session prefill, recompute_kv_cache): 32-token tiles on the bf16 MFMA path
        const bool big = lm_can_gemm128(h);
        const int tile = big ? LM_MAXM : LM_TILE32;
        // A background tile (rca_lm_eval_async of one pass: kv_shadow.py feeds one per few frames) is ~230 launches; issued eagerly
        // they cost the calling frame ~2 ms of host time before its own replay is even launched (tile frames 9.5 ms against a median
        // of 4.4).  The third pass of a size on this cache is captured and replayed from then on: one launch.  (First: eager --
        // the launchers' one-time attribute calls; the geometry of a pass depends on its token count only, the context is read
        // from the device state.)
        static const bool tile_graphs = !(getenv("RCA_LM_TILE_GRAPHS") && atoi(getenv("RCA_LM_TILE_GRAPHS")) == 0) &&
                                        !(getenv("RCA_LM_FLASH") && atoi(getenv("RCA_LM_FLASH")) == 0);   // (the split-attention A/B path launches per context)
        if (!wait_last && big && n <= tile && h->graphs_enabled && tile_graphs) {
            rca_lm::GraphSet& gs = lm_graph_set(h);
            int slot = -1;
            for (int i = 0; i < 2; ++i)
                if (gs.tile_m[i] == n) slot = i;
            if (slot < 0) {
                slot = gs.tile_seen[0] <= gs.tile_seen[1] ? 0 : 1;
                if (gs.tile_g[slot]) { (void)hipGraphExecDestroy(gs.tile_g[slot]); gs.tile_g[slot] = nullptr; }
                gs.tile_m[slot] = n; gs.tile_seen[slot] = 0;
            }
            if (++gs.tile_seen[slot] >= 2) {
                h->h_stt->n_tokens = h->n_tokens;
                h->h_stt->m = n;
                for (int i = 0; i < n; ++i) h->h_stt->ids[i] = ids[i];
                if (!gs.tile_g[slot]) {
                    hipGraph_t g = nullptr;
                    RCA_HIP(hipStreamSynchronize(st));
                    RCA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                    hipError_t e = hipMemcpyAsync(h->stt, h->h_stt, 8 + 4 * (size_t)std::max(n, 16), hipMemcpyHostToDevice, st);
                    rc = e == hipSuccess ? lm_enqueue_prefill_tile128(h, n, st, 1) : fail(RCA_ERR_HIP, "tile capture memcpy: %s", hipGetErrorString(e));
                    if (rc == RCA_OK) {
                        const rca_lm_config_t& c = h->cfg;
                        const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
                        launch_gemv<1, 0>(GEMV_HEAD, h, 1, h->head, nullptr, h->logits, c.vocab_size, c.hidden, c.vocab_size,
                                          GemvPro{h->x, h->final_norm, c.rms_eps, 1}, norope, st);
                    }
                    hipError_t e2 = hipStreamEndCapture(st, &g);
                    if (rc != RCA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
                    if (e2 != hipSuccess) return fail(RCA_ERR_HIP, "tile capture: %s", hipGetErrorString(e2));
                    e2 = hipGraphInstantiate(&gs.tile_g[slot], g, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(g);
                    if (e2 != hipSuccess) { gs.tile_g[slot] = nullptr; return fail(RCA_ERR_HIP, "tile graph instantiate: %s", hipGetErrorString(e2)); }
                    (void)hipGraphUpload(gs.tile_g[slot], st);
                }
                RCA_HIP(hipGraphLaunch(gs.tile_g[slot], st));
                h->n_tokens += n;
                h->logits_rows = 1;
                h->async_pending = true;
                return RCA_OK;
            }
        }
        for (int off = 0; off < n; off += tile) {
            const int m = std::min(tile, n - off);
            const bool last = off + m >= n;
            if ((rc = lm_push_state(h, ids + off, m, st)) != RCA_OK) return rc;
            rc = big ? lm_enqueue_prefill_tile128(h, m, st, lm_splits_needed(h, m)) : lm_enqueue_prefill_tile(h, m, st, lm_splits_needed(h, m));
            if (rc != RCA_OK) return rc;
            if (last) {   // logits of the final token: final norm + head on the register GEMV path
                const rca_lm_config_t& c = h->cfg;
                const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
                launch_gemv<1, 0>(GEMV_HEAD, h, 1, h->head, nullptr, h->logits, c.vocab_size, c.hidden, c.vocab_size,
                                  GemvPro{h->x, h->final_norm, c.rms_eps, 1}, norope, st);
                RCA_LAUNCH_CHECK();
            }
            h->n_tokens += m;
            if (!last) RCA_HIP(hipStreamSynchronize(st));
        }
    } else {
        for (int off = 0; off < n; off += LM_GEMV_M) {
            const int m = std::min(LM_GEMV_M, n - off);
            const bool last = off + m >= n;
            if ((rc = lm_push_state(h, ids + off, m, st)) != RCA_OK) return rc;
            if (all) h->logits = logits_base + (long)off * h->cfg.vocab_size;
            rc = lm_enqueue_pass(h, m, all ? 2 : (last ? 1 : 0), st, lm_splits_needed(h, m));
            h->logits = logits_base;
            if (rc != RCA_OK) return rc;
            h->n_tokens += m;
            // the pinned staging block is reused by the next chunk: wait until this one's copy has been consumed
            if (!last) RCA_HIP(hipStreamSynchronize(st));
        }
    }
    h->logits_rows = all ? n : 1;
    if (wait_last) RCA_HIP(hipStreamSynchronize(st));
    else h->async_pending = true;
    return RCA_OK;
}
extern "C" int rca_lm_eval(rca_lm_t* h, const int32_t* ids, int32_t n) { return lm_eval_impl(h, ids, n, true, false); }
// Asynchronous, and with the arithmetic of a LONG eval whatever n is: a cache built piecewise with this call holds the same bits
// as one rca_lm_eval of the whole sequence (short evals otherwise run as decode passes, which round differently from the tiles).
extern "C" int rca_lm_eval_async(rca_lm_t* h, const int32_t* ids, int32_t n) { return lm_eval_impl(h, ids, n, false, true); }

// ---- KV-cache plumbing between a handle and its weight-sharing twin (the shadow cache of the sliding-window trim)
static int lm_same_cache_shape(const rca_lm* a, const rca_lm* b) {
    return a->device == b->device && a->cfg.n_layers == b->cfg.n_layers && a->cfg.n_kv_heads == b->cfg.n_kv_heads &&
           a->cfg.head_dim == b->cfg.head_dim && a->n_ctx_pad == b->n_ctx_pad && a->cfg.n_ctx == b->cfg.n_ctx;
}
extern "C" int rca_lm_copy_kv(rca_lm_t* dst, rca_lm_t* src, int32_t n_pos) {
    if (!dst || !src || dst == src) return fail(RCA_ERR_ARG, "copy_kv: two different handles needed");
    if (!lm_same_cache_shape(dst, src)) return fail(RCA_ERR_ARG, "copy_kv: the two KV caches differ in shape");
    if (n_pos < 0 || n_pos > src->cfg.n_ctx) return fail(RCA_ERR_ARG, "copy_kv: %d positions outside [0, %d]", n_pos, src->cfg.n_ctx);
    RCA_HIP(hipSetDevice(dst->device));
    RCA_HIP(hipStreamSynchronize(src->stream));   // everything src has evaluated so far is in its cache
    src->async_pending = false;
    if (dst->async_pending) { RCA_HIP(hipStreamSynchronize(dst->stream)); dst->async_pending = false; }
    const size_t bytes = (size_t)n_pos * src->cfg.n_kv_heads * src->cfg.head_dim * sizeof(f16_t);
    for (int l = 0; l < src->cfg.n_layers && bytes; ++l) {
        RCA_HIP(hipMemcpyAsync(dst->kc + (long)l * dst->kv_layer_stride, src->kc + (long)l * src->kv_layer_stride, bytes, hipMemcpyDeviceToDevice, dst->stream));
        RCA_HIP(hipMemcpyAsync(dst->vc + (long)l * dst->kv_layer_stride, src->vc + (long)l * src->kv_layer_stride, bytes, hipMemcpyDeviceToDevice, dst->stream));
    }
    dst->async_pending = true;
    return RCA_OK;
}
extern "C" int rca_lm_swap_kv(rca_lm_t* a, rca_lm_t* b) {
    if (!a || !b || a == b) return fail(RCA_ERR_ARG, "swap_kv: two different handles needed");
    if (!lm_same_cache_shape(a, b)) return fail(RCA_ERR_ARG, "swap_kv: the two KV caches differ in shape");
    RCA_HIP(hipSetDevice(a->device));
    RCA_HIP(hipStreamSynchronize(a->stream));
    RCA_HIP(hipStreamSynchronize(b->stream));
    a->async_pending = b->async_pending = false;
    std::swap(a->kc, b->kc);
    std::swap(a->vc, b->vc);
    return RCA_OK;
}
extern "C" int rca_lm_set_low_priority(rca_lm_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipSetDevice(h->device));
    int lo = 0, hi = 0;
    RCA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));   // numerically lower = higher priority
    hipStream_t ns = nullptr;
    RCA_HIP(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, enable ? lo : hi));
    RCA_HIP(hipStreamSynchronize(h->stream));
    h->async_pending = false;
    (void)hipStreamDestroy(h->stream);
    h->stream = ns;
    return RCA_OK;
}

extern "C" int rca_lm_logits_dev(rca_lm_t* h, const float** out) {
    if (!h || !out) return fail(RCA_ERR_ARG, "null");
    if (h->logits_rows < 1) return fail(RCA_ERR_STATE, "no logits: call eval first");
    *out = h->logits + (long)(h->logits_rows - 1) * h->cfg.vocab_size;
    return RCA_OK;
}
extern "C" int rca_lm_get_logits(rca_lm_t* h, float* out_host) {
    if (!h || !out_host) return fail(RCA_ERR_ARG, "null");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    const float* src;
    int rc;
    if ((rc = rca_lm_logits_dev(h, &src)) != RCA_OK) return rc;
    RCA_HIP(hipSetDevice(h->device));
    RCA_HIP(hipMemcpyAsync(out_host, src, (size_t)h->cfg.vocab_size * 4, hipMemcpyDeviceToHost, h->stream));
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}
extern "C" int rca_lm_get_logits_row(rca_lm_t* h, int32_t row, float* out_host) {
    if (!h || !out_host) return fail(RCA_ERR_ARG, "null");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    if (row < 0 || row >= h->logits_rows) return fail(RCA_ERR_ARG, "row %d outside the %d rows of the last eval", row, h->logits_rows);
    RCA_HIP(hipSetDevice(h->device));
    RCA_HIP(hipMemcpyAsync(out_host, h->logits + (long)row * h->cfg.vocab_size, (size_t)h->cfg.vocab_size * 4, hipMemcpyDeviceToHost, h->stream));
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}

extern "C" int rca_lm_sampler_init(rca_lm_t* h, const rca_sampler_params_t* p) {
    if (!h || !p) return fail(RCA_ERR_ARG, "null");
    if (p->n_bias < 0 || p->n_bias > 8) return fail(RCA_ERR_ARG, "at most 8 logit-bias entries");
    // llama.cpp reads top_k <= 0 as "whole vocabulary" and honours any top_k up to it; penalties default to off.  Nothing is clamped:
    // 1..SAMP_MAXK ranked candidates take the serial chain (float sums, inverse-CDF draw), everything else the whole-vocabulary
    // sampler -- plain (top_p >= 1, no rank cut) or with the radix-select thresholds of the "big" path.
    const int V = h->cfg.vocab_size;
    if (!(p->repeat_penalty > 0.0f) && p->repeat_penalty != 0.0f) return fail(RCA_ERR_ARG, "sampler: repeat_penalty %g must be positive", (double)p->repeat_penalty);
    const float rep = p->repeat_penalty == 0.0f ? 1.0f : p->repeat_penalty;    // a zero-initialised struct (older callers) means "off"
    const bool stoch = p->temp > 0.0f;
    const bool full = stoch && (p->top_k <= 0 || p->top_k > SAMP_MAXK);
    const bool big_k = full && p->top_k > SAMP_MAXK && p->top_k < V;
    const bool big_p = full && p->top_p < 1.0f;
    const bool pen = rep != 1.0f || p->freq_penalty != 0.0f || p->presence_penalty != 0.0f;
    const bool patch = p->n_bias > 0 || pen;
    RCA_HIP(hipSetDevice(h->device));
    SamplerDev s;
    memset(&s, 0, sizeof(s));
    s.top_k = p->top_k; s.top_p = p->top_p; s.min_p = p->min_p; s.temp = p->temp; s.seed = p->seed; s.n_bias = p->n_bias;
    for (int i = 0; i < p->n_bias; ++i) { s.bias_ids[i] = p->bias_ids[i]; s.bias_vals[i] = p->bias_vals[i]; }
    s.repeat_penalty = rep; s.freq_penalty = p->freq_penalty; s.presence_penalty = p->presence_penalty;
    s.last_n = p->penalty_last_n > 0 ? std::min(p->penalty_last_n, 64) : (p->penalty_last_n == 0 ? 64 : 0);   // 0 = llama-cpp-python's default window, < 0 = off
    s.patch = patch ? 1 : 0; s.big_k = big_k ? p->top_k : 0; s.big_p = big_p ? 1 : 0;
    RCA_HIP(hipStreamSynchronize(h->stream));
    RCA_HIP(hipMemcpy(h->samp, &s, sizeof(s), hipMemcpyHostToDevice));
    // set_seed restarts the stream of draws (llamacpp_utils.py:58)
    const unsigned long long zero = 0;
    RCA_HIP(hipMemcpy(&h->stt->rng_counter, &zero, 8, hipMemcpyHostToDevice));
    h->rng_host = 0;
    h->sampler_set = true;
    if (full != h->samp_full || patch != h->samp_patch || big_k != h->samp_big_k || big_p != h->samp_big_p)
        lm_drop_graphs(h);   // the captured steps hold another set of sampler launches
    h->samp_full = full; h->samp_patch = patch; h->samp_big_k = big_k; h->samp_big_p = big_p;
    return RCA_OK;
}

// frame_i >= 0: step i of a frame graph -- the sampler's tail also advances the device state and gathers the next pair's embeddings
static void lm_enqueue_sample(rca_lm* h, float* lg, hipStream_t st, int frame_i = -1) {
    const int V = h->cfg.vocab_size;
    const SampTail tail{frame_i, h->embed, h->embed_f32, h->x, h->cfg.hidden};
    if (h->samp_patch) samp_prepare_kernel<<<1, 128, 0, st>>>(lg, V, h->samp, h->stt, h->swork);   // logit bias / penalties, in place
    if (h->samp_full) {   // top_k <= 0 or > SAMP_MAXK: whole vocabulary (Gumbel-max), with a rank and / or mass threshold on the big path
        samp_full_max_kernel<<<SAMPF_SLICES, 1024, 0, st>>>(lg, V, h->samp, h->swork);
        if (h->samp_big_k || h->samp_big_p) {
            for (int m = 0; m < 2; ++m) {
                if (!(m == 0 ? h->samp_big_k : h->samp_big_p)) continue;
                for (int l = 0; l < SAMP_LEVELS; ++l) samp_radix_pass_kernel<<<128, 256, 0, st>>>(lg, V, h->samp, h->swork, m, l);
                samp_radix_finish_kernel<<<1, 256, 0, st>>>(h->samp, h->swork, m);
            }
            samp_big_pick_kernel<<<SAMPF_SLICES, 1024, 0, st>>>(lg, V, h->samp, h->stt, h->swork);
        } else {
            samp_full_pick_kernel<<<SAMPF_SLICES, 1024, 0, st>>>(lg, V, h->samp, h->stt, h->swork);
        }
        samp_full_final_kernel<<<1, 1024, 0, st>>>(lg, V, h->samp, h->stt, h->swork, tail);
        return;
    }
    samp_hist_kernel<<<128, 256, 0, st>>>(lg, V, h->samp, h->swork);
    samp_gather_kernel<<<128, 256, 0, st>>>(lg, V, h->samp, h->swork);
    samp_final_kernel<<<1, 1024, 0, st>>>(lg, V, h->samp, h->stt, h->swork, tail);
}

static int lm_fetch_token(rca_lm* h, int32_t* token, hipStream_t st) {
    RCA_HIP(hipMemcpyAsync(&h->h_stt->out_token, &h->stt->out_token, 4, hipMemcpyDeviceToHost, st));
    RCA_HIP(hipStreamSynchronize(st));
    *token = h->h_stt->out_token;
    return RCA_OK;
}

extern "C" int rca_lm_sample(rca_lm_t* h, int32_t* token) {
    if (!h || !token) return fail(RCA_ERR_ARG, "null");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    if (!h->sampler_set) return fail(RCA_ERR_STATE, "sampler not initialised");
    if (h->logits_rows < 1) return fail(RCA_ERR_STATE, "no logits: call eval first");
    RCA_HIP(hipSetDevice(h->device));
    float* lg = h->logits + (long)(h->logits_rows - 1) * h->cfg.vocab_size;
    lm_enqueue_sample(h, lg, h->stream);
    RCA_LAUNCH_CHECK();
    h->rng_host += 1;
    return lm_fetch_token(h, token, h->stream);
}

// eval(ids[0..n)) + sample with no host round trip in between.  For n <= 2 and a plain (not
// logits_all) handle the whole step is one hipGraph replay.
// eval + sample (+ optionally softmax(logits)[probe ids] of the evaluated position) as one replay and one synchronisation
static int lm_step_impl(rca_lm_t* h, const int32_t* ids, int32_t n, const int32_t* probe_ids, int32_t n_probe, int32_t* token, float* probs_out,
                        int cap_bucket = -1) {
    if (!h || !ids || !token || n < 1) return fail(RCA_ERR_ARG, "step: bad argument");
    if (n_probe < 0 || n_probe > 8 || (n_probe > 0 && (!probe_ids || !probs_out))) return fail(RCA_ERR_ARG, "step: 0..8 probe ids");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    if (!h->sampler_set) return fail(RCA_ERR_STATE, "sampler not initialised");
    if (h->n_tokens + n > h->cfg.n_ctx) return fail(RCA_ERR_STATE, "context overflow: %d + %d > n_ctx %d", h->n_tokens, n, h->cfg.n_ctx);
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "step: token id %d at index %d is outside the vocabulary [0, %d)", ids[i], i, h->cfg.vocab_size);
    for (int i = 0; i < n_probe; ++i)
        if (probe_ids[i] < 0 || probe_ids[i] >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "step: probe id %d is outside the vocabulary", probe_ids[i]);
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    int rc;
    if (n > 2 || h->cfg.logits_all || !h->graphs_enabled) {
        if ((rc = rca_lm_eval(h, ids, n)) != RCA_OK) return rc;
        if ((rc = rca_lm_sample(h, token)) != RCA_OK) return rc;
        return n_probe ? rca_lm_token_probs(h, probe_ids, n_probe, probs_out) : RCA_OK;
    }
    // stage inputs in pinned memory; the graph's first node copies them to the device
    h->h_stt->n_tokens = h->n_tokens;
    h->h_stt->m = n;
    for (int i = 0; i < n; ++i) h->h_stt->ids[i] = ids[i];
    if (n_probe) {
        for (int i = 0; i < n_probe; ++i) h->h_probe[i] = probe_ids[i];
    }
    int bucket = 0;
    const int need = lm_splits_needed(h, n);
    while (bucket + 1 < LM_GRAPH_BUCKETS && (4 << bucket) < need) ++bucket;
    if (cap_bucket >= 0) bucket = cap_bucket;
    const int nsp_launch = bucket + 1 == LM_GRAPH_BUCKETS ? h->n_splits : std::min(h->n_splits, 4 << bucket);
    rca_lm::GraphSet& gs = lm_graph_set(h);
    hipGraphExec_t& gexec = n_probe ? gs.gp[n][bucket] : gs.g[n][bucket];
    if (n_probe && gexec && gs.gp_nprobe[n][bucket] != n_probe) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
    if (!gexec) {
        hipGraph_t g = nullptr;
        RCA_HIP(hipStreamSynchronize(st));
        RCA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipMemcpyAsync(h->stt, h->h_stt, LM_STATE_DECODE_BYTES, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && n_probe) e = hipMemcpyAsync(h->probe_ids_dev, h->h_probe, n_probe * 4, hipMemcpyHostToDevice, st);
        rc = e == hipSuccess ? lm_enqueue_pass(h, n, 1, st, nsp_launch) : fail(RCA_ERR_HIP, "capture memcpy: %s", hipGetErrorString(e));
        if (rc == RCA_OK) {
            lm_enqueue_sample(h, h->logits, st);
            if (n_probe) {   // rca_lm_token_probs' two launches, over the logits this step just wrote
                lm_softmax_slices_kernel<<<PROBS_SLICES, 1024, 0, st>>>(h->logits, h->cfg.vocab_size, h->probs_dev + 64);
                lm_token_probs_kernel<<<1, 64, 0, st>>>(h->logits, h->cfg.vocab_size, h->probs_dev + 64, h->probe_ids_dev, n_probe, h->probs_dev);
            }
            e = hipMemcpyAsync(&h->h_stt->out_token, &h->stt->out_token, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && n_probe) e = hipMemcpyAsync(h->h_probe + 64, h->probs_dev, n_probe * 4, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) rc = fail(RCA_ERR_HIP, "capture d2h: %s", hipGetErrorString(e));
        }
        hipError_t e2 = hipStreamEndCapture(st, &g);
        if (rc != RCA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e2 != hipSuccess) return fail(RCA_ERR_HIP, "end capture: %s", hipGetErrorString(e2));
        e2 = hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) { gexec = nullptr; return fail(RCA_ERR_HIP, "graph instantiate: %s", hipGetErrorString(e2)); }
        (void)hipGraphUpload(gexec, st);   // the device-side copy now, not inside the first frame that replays it (pre-captured graphs: first trim frame, first frame of a bucket)
        if (n_probe) gs.gp_nprobe[n][bucket] = n_probe;
    }
    if (cap_bucket >= 0) return RCA_OK;      // rca_duplex_precapture: the graph exists now, nothing is launched
    RCA_HIP(hipGraphLaunch(gexec, st));
    RCA_HIP(hipStreamSynchronize(st));
    h->n_tokens += n;
    h->logits_rows = 1;
    h->rng_host += 1;
    *token = h->h_stt->out_token;
    for (int i = 0; i < n_probe; ++i) probs_out[i] = reinterpret_cast<const float*>(h->h_probe + 64)[i];
    return RCA_OK;
}
static int lm_step_capture_only(rca_lm_t* h, int n, int n_probe, int bucket) {
    if (h->cfg.logits_all || !h->graphs_enabled || n > 2) return RCA_OK;
    const int32_t ids[2] = {0, 0}, probes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t tok = 0;
    float pr[8];
    return lm_step_impl(h, ids, n, n_probe ? probes : nullptr, n_probe, &tok, n_probe ? pr : nullptr, bucket);
}
// eval(ids[0..n)) + sample with no host round trip in between.  For n <= 2 and a plain (not
// logits_all) handle the whole step is one hipGraph replay.
extern "C" int rca_lm_step(rca_lm_t* h, const int32_t* ids, int32_t n, int32_t* token) {
    return lm_step_impl(h, ids, n, nullptr, 0, token, nullptr);
}
// rca_lm_step + rca_lm_token_probs of the position just evaluated, as ONE replay and one synchronisation: the agent's speculative
// <|end_audio|> step (get_probable_event_speaker, realtime_agent_v2.py:455-466: eval, sample, softmax(logits)[agent / user speaker])
extern "C" int rca_lm_step_probe(rca_lm_t* h, const int32_t* ids, int32_t n, const int32_t* probe_ids, int32_t n_probe, int32_t* token, float* probs_out) {
    if (n_probe < 1) return fail(RCA_ERR_ARG, "step_probe: 1..8 probe ids");
    return lm_step_impl(h, ids, n, probe_ids, n_probe, token, probs_out);
}

// One frame of the duplex loop as ONE graph (process_audio_input_ids, realtime_agent_v2.py:332-372, while every sampled token is an
// audio token): step i evaluates [agent_{i-1}, user_{i-1}] -- for i = 0 the pair the caller passes -- and samples agent_i, which
// goes back into step i + 1 on the device together with the user's token of frame i.  One host synchronisation per frame instead
// of one per step.  If step j samples a token <= audio_id_floor (the loop leaves audio mode there), steps after j have run on a
// wrong guess: n_done = j + 1, the KV position and the draw counter are put back to what the step-by-step loop would have (their
// cache slots are stale and get overwritten, exactly like a rollback), and the caller carries on step by step.
static int lm_frame_core(rca_lm_t* h, const int32_t* first_pair, const int32_t* user_ids, int32_t n_steps, int32_t audio_id_floor,
                         int32_t* out_tokens, int32_t* n_done, int cap_bucket) {
    if (!h || !first_pair || !user_ids || !out_tokens || !n_done) return fail(RCA_ERR_ARG, "frame: null argument");
    if (n_steps < 1 || n_steps > LM_FRAME_MAX) return fail(RCA_ERR_ARG, "frame: %d steps (1..%d)", n_steps, LM_FRAME_MAX);
    if (!h->sampler_set) return fail(RCA_ERR_STATE, "sampler not initialised");
    if (h->cfg.logits_all) return fail(RCA_ERR_STATE, "frame: not on a logits_all handle");
    if (cap_bucket < 0 && h->n_tokens + 2 * n_steps > h->cfg.n_ctx) return fail(RCA_ERR_STATE, "context overflow: %d + %d > n_ctx %d", h->n_tokens, 2 * n_steps, h->cfg.n_ctx);
    for (int i = 0; i < 2; ++i)
        if (first_pair[i] < 0 || first_pair[i] >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "frame: token id %d outside the vocabulary", first_pair[i]);
    for (int i = 0; i < n_steps; ++i)
        if (user_ids[i] < 0 || user_ids[i] >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "frame: token id %d outside the vocabulary", user_ids[i]);
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    int rc = RCA_OK;
    h->h_stt->n_tokens = h->n_tokens;
    h->h_stt->m = 2;
    h->h_stt->ids[0] = first_pair[0];
    h->h_stt->ids[1] = first_pair[1];
    for (int i = 0; i < n_steps; ++i) h->h_stt->ids[LM_FRAME_USER0 + i] = user_ids[i];
    int bucket = 0;
    const int need = lm_splits_needed(h, 2 * n_steps);
    while (bucket + 1 < LM_GRAPH_BUCKETS && (4 << bucket) < need) ++bucket;
    if (cap_bucket >= 0) bucket = cap_bucket;
    const int nsp_launch = bucket + 1 == LM_GRAPH_BUCKETS ? h->n_splits : std::min(h->n_splits, 4 << bucket);
    hipGraphExec_t& gexec = lm_graph_set(h).fg[n_steps][bucket];
    if (cap_bucket >= 0 && (gexec || !h->graphs_enabled)) return RCA_OK;
    if (!gexec || !h->graphs_enabled) {
        // graphs disabled (tests): the same launches, eagerly
        const bool capture = h->graphs_enabled;
        hipGraph_t g = nullptr;
        RCA_HIP(hipStreamSynchronize(st));
        if (capture) RCA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipMemcpyAsync(h->stt, h->h_stt, LM_STATE_DECODE_BYTES, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) rc = fail(RCA_ERR_HIP, "frame memcpy: %s", hipGetErrorString(e));
        for (int i = 0; i < n_steps && rc == RCA_OK; ++i) {
            rc = lm_enqueue_pass(h, 2, 1, st, nsp_launch, i > 0);
            if (rc != RCA_OK) break;
            lm_enqueue_sample(h, h->logits, st, i);
        }
        if (rc == RCA_OK) {
            e = hipMemcpyAsync(h->h_stt->frame_out, h->stt->frame_out, sizeof(int) * LM_FRAME_MAX, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) rc = fail(RCA_ERR_HIP, "frame d2h: %s", hipGetErrorString(e));
        }
        if (capture) {
            hipError_t e2 = hipStreamEndCapture(st, &g);
            if (rc != RCA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e2 != hipSuccess) return fail(RCA_ERR_HIP, "end capture: %s", hipGetErrorString(e2));
            e2 = hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e2 != hipSuccess) { gexec = nullptr; return fail(RCA_ERR_HIP, "graph instantiate: %s", hipGetErrorString(e2)); }
            (void)hipGraphUpload(gexec, st);
        } else if (rc != RCA_OK) {
            return rc;
        }
    }
    if (cap_bucket >= 0) return RCA_OK;      // rca_duplex_precapture: the graph exists now, nothing is launched
    if (h->graphs_enabled) RCA_HIP(hipGraphLaunch(gexec, st));
    RCA_HIP(hipStreamSynchronize(st));
    int done = n_steps;
    for (int i = 0; i < n_steps; ++i) {
        out_tokens[i] = h->h_stt->frame_out[i];
        if (out_tokens[i] <= audio_id_floor) { done = i + 1; break; }
    }
    *n_done = done;
    h->n_tokens += 2 * done;
    // a frame cut short leaves the logits of a LATER step (evaluated on a wrong-guess pair, rolled back above) in the buffer: nothing
    // may read them -- get_logits / token_probs / sample fail with "no logits" until the next eval
    h->logits_rows = done < n_steps ? 0 : 1;
    h->rng_host += (unsigned long long)done;
    if (done < n_steps) {   // the device drew n_steps times: put its counter where the step-by-step loop would be
        RCA_HIP(hipMemcpy(&h->stt->rng_counter, &h->rng_host, 8, hipMemcpyHostToDevice));
    }
    return RCA_OK;
}

extern "C" int rca_lm_frame(rca_lm_t* h, const int32_t* first_pair, const int32_t* user_ids, int32_t n_steps, int32_t audio_id_floor,
                            int32_t* out_tokens, int32_t* n_done) {
    return lm_frame_core(h, first_pair, user_ids, n_steps, audio_id_floor, out_tokens, n_done, -1);
}

// ---- one duplex frame as ONE graph (process_audio, realtime_agent_v2.py:504-554): encode tail of the user's PCM window -> code ->
// token id (affine: the codec tokens were added to the vocabulary in code order, train_vanilla_latest.py:587-589) -> the chunk's
// n_steps LM steps with device-side feedback (rca_lm_frame) -> token id -> code, appended to the detokenizer's code context ->
// decode tail -> softmax(last logits)[probe] (measure_event_prob, :448-452).  One upload, one replay, one synchronisation.  The host
// stays the owner of both rolling windows (it passes them in whole), so a frame that cannot take this path -- or is cut short --
// simply goes through the separate calls with nothing to repair on the device.
struct DuplexDev {            // device-side scratch of the duplex frame (one allocation)
    long long user_codes[LM_FRAME_MAX];
    float probe_prob;
    int flags;
    int probe_id;
    int pad;
};
struct DuplexPin {            // pinned mirror: inputs first (uploaded), outputs behind
    int probe_id;
    int pad0[3];
    long long user_codes[LM_FRAME_MAX];   // user_codes, probe_prob, flags: one copy of DuplexDev's head
    float probe_prob;
    int flags;
    int frame_out[LM_FRAME_MAX];
};
__global__ void duplex_codes_to_ids_kernel(const long long* __restrict__ codes, int n, int base, LmDevState* stt, int* flags) {
    if (threadIdx.x == 0) *flags = 0;
    if ((int)threadIdx.x < n) stt->ids[LM_FRAME_USER0 + threadIdx.x] = base + (int)codes[threadIdx.x];
}
// frame_out -> codes behind the context codes; a token that is not a codec token (the frame left audio mode, or a padding row of the
// vocabulary was drawn) becomes code 0 and raises flag bit 0: the PCM of this replay is then not used
__global__ void duplex_tokens_to_codes_kernel(const LmDevState* __restrict__ stt, int n, int base, int n_codes, long long* __restrict__ dst,
                                              int* flags) {
    if ((int)threadIdx.x < n) {
        const int c = stt->frame_out[threadIdx.x] - base;
        const bool ok = c >= 0 && c < n_codes;
        dst[threadIdx.x] = ok ? c : 0;
        if (!ok) atomicOr(flags, 1);
    }
}
struct DuplexGraphKey {
    int T, F_ctx, n_steps, n_samples, probe, bucket, base;
    const void* kc;
    unsigned long long codec_sig;
    const void* codec;
    bool operator==(const DuplexGraphKey& o) const {
        return T == o.T && F_ctx == o.F_ctx && n_steps == o.n_steps && n_samples == o.n_samples && probe == o.probe && bucket == o.bucket &&
               base == o.base && kc == o.kc && codec_sig == o.codec_sig && codec == o.codec;
    }
};
struct DuplexState {
    float* pcm_in = nullptr; size_t pcm_in_cap = 0;       // device: the PCM window
    long long* code_win = nullptr; size_t code_win_cap = 0;   // device: context codes + this frame's
    float* pcm_out = nullptr; size_t pcm_out_cap = 0;     // device: decode tail
    DuplexDev* dev = nullptr;
    char* pin = nullptr; size_t pin_cap = 0;              // pinned: [DuplexPin | pcm window | code ctx | pcm out]
    struct Entry { DuplexGraphKey key; hipGraphExec_t exec = nullptr; };
    std::vector<Entry> graphs;
    // call shapes whose codec workspace an eager frame has already sized (a capture must not allocate): one eager frame per SHAPE,
    // not per context bucket
    struct Warm { int T, F_ctx, n_steps, n_samples; const void* codec; unsigned long long codec_sig; };
    std::vector<Warm> warm;
    const void* logits_at_capture = nullptr;
    hipStream_t side = nullptr;      // the encode tail runs beside the first LM step (which does not need this chunk's codes)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork = true;
};
static void duplex_drop_graphs(DuplexState* d) {
    if (!d) return;
    for (auto& e : d->graphs)
        if (e.exec) (void)hipGraphExecDestroy(e.exec);
    d->graphs.clear();
}
static void duplex_destroy(DuplexState* d) {
    if (!d) return;
    duplex_drop_graphs(d);
    for (void* p : {(void*)d->pcm_in, (void*)d->code_win, (void*)d->pcm_out, (void*)d->dev})
        if (p) (void)hipFree(p);
    if (d->pin) (void)hipHostFree(d->pin);
    if (d->side) (void)hipStreamDestroy(d->side);
    if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
    if (d->ev_join) (void)hipEventDestroy(d->ev_join);
    delete d;
}
template <class T>
static int duplex_grow(T** p, size_t* cap, size_t n) {
    if (n <= *cap) return RCA_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    int rc = lm_alloc((void**)p, n * sizeof(T));
    if (rc == RCA_OK) *cap = n;
    return rc;
}

// buffers, side stream and events of the duplex frame for a call shape (growing a buffer invalidates the graphs captured over it)
static int duplex_reserve(rca_lm* h, int T, int F_ctx, int n, int n_samples) {
    int rc;
    hipStream_t st = h->stream;
    if (!h->duplex) {
        h->duplex = new DuplexState();
        const char* nf = getenv("RCA_DUPLEX_FORK");
        h->duplex->fork = !(nf && nf[0] == '0');
        RCA_HIP(hipStreamCreateWithFlags(&h->duplex->side, hipStreamNonBlocking));
        RCA_HIP(hipEventCreateWithFlags(&h->duplex->ev_fork, hipEventDisableTiming));
        RCA_HIP(hipEventCreateWithFlags(&h->duplex->ev_join, hipEventDisableTiming));
    }
    DuplexState* d = h->duplex;
    const int F = F_ctx + n;
    const size_t pin_pcm = sizeof(DuplexPin), pin_codes = pin_pcm + (((size_t)T * 4 + 255) & ~(size_t)255),
                 pin_out = pin_codes + (((size_t)F_ctx * 8 + 255) & ~(size_t)255), pin_total = pin_out + (size_t)n_samples * 4;
    if ((size_t)T > d->pcm_in_cap || (size_t)F > d->code_win_cap || (size_t)n_samples > d->pcm_out_cap || pin_total > d->pin_cap || !d->dev) {
        RCA_HIP(hipStreamSynchronize(st));
        duplex_drop_graphs(d);
        if ((rc = duplex_grow(&d->pcm_in, &d->pcm_in_cap, (size_t)T)) != RCA_OK) return rc;
        if ((rc = duplex_grow(&d->code_win, &d->code_win_cap, (size_t)F + 8)) != RCA_OK) return rc;
        if ((rc = duplex_grow(&d->pcm_out, &d->pcm_out_cap, (size_t)n_samples)) != RCA_OK) return rc;
        if (!d->dev && (rc = lm_alloc((void**)&d->dev, sizeof(DuplexDev))) != RCA_OK) return rc;
        if (pin_total > d->pin_cap) {
            if (d->pin) (void)hipHostFree(d->pin);
            d->pin = nullptr; d->pin_cap = 0;
            RCA_HIP(hipHostMalloc((void**)&d->pin, pin_total + (pin_total >> 2), hipHostMallocDefault));
            d->pin_cap = pin_total + (pin_total >> 2);
        }
    }
    return RCA_OK;
}
// The one-time allocations of rca_duplex_frame for a call shape (pinned staging, device buffers, side stream), so that a session can make
// them at reset() instead of inside its first one-replay frame (a pinned allocation costs milliseconds).  Optional.
extern "C" int rca_duplex_prepare(rca_lm_t* h, int32_t T, int32_t F_ctx, int32_t n_steps, int32_t n_samples) {
    if (!h || T < 1 || F_ctx < 0 || n_steps < 1 || n_steps > LM_FRAME_MAX || n_samples < 1) return fail(RCA_ERR_ARG, "duplex_prepare: bad shape");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    RCA_HIP(hipSetDevice(h->device));
    return duplex_reserve(h, T, F_ctx, n_steps, n_samples);
}

// cap_bucket < 0: run the frame.  cap_bucket >= 0 (rca_duplex_precapture): make sure the graph of this call shape exists for that
// context bucket on the KV cache currently installed -- nothing is launched except, once per shape, the codec's two tail calls on
// scratch buffers (they size its workspace; a capture must not allocate) -- and return.
static int duplex_frame_core(rca_lm_t* h, rca_codec_t* codec, const rca_duplex_frame_args_t* a, rca_duplex_frame_out_t* out,
                             float* pcm_out_host, int cap_bucket) {
    const bool cap_only = cap_bucket >= 0;
    if (!h || !codec || !a || (!cap_only && (!out || !pcm_out_host || !a->pcm_window || (a->F_ctx > 0 && !a->code_ctx)))) return fail(RCA_ERR_ARG, "duplex_frame: null argument");
    const int n = a->n_steps;
    if (n < 1 || n > LM_FRAME_MAX) return fail(RCA_ERR_ARG, "duplex_frame: %d steps (1..%d)", n, LM_FRAME_MAX);
    if (a->T < 1 || a->F_ctx < 0 || a->n_samples < 1) return fail(RCA_ERR_ARG, "duplex_frame: bad shape (T=%d F_ctx=%d n_samples=%d)", a->T, a->F_ctx, a->n_samples);
    if (!h->sampler_set) return fail(RCA_ERR_STATE, "sampler not initialised");
    if (h->cfg.logits_all) return fail(RCA_ERR_STATE, "duplex_frame: not on a logits_all handle");
    if (!cap_only && h->n_tokens + 2 * n > h->cfg.n_ctx) return fail(RCA_ERR_STATE, "context overflow: %d + %d > n_ctx %d", h->n_tokens, 2 * n, h->cfg.n_ctx);
    int32_t n_codes = 0;
    int rc;
    if ((rc = rca_codec_codebook_size(codec, &n_codes)) != RCA_OK) return rc;
    if (a->code_token_base < 0 || (long)a->code_token_base + n_codes > h->cfg.vocab_size) return fail(RCA_ERR_ARG, "duplex_frame: codes [%d, %d + %d) fall outside the vocabulary", a->code_token_base, a->code_token_base, n_codes);
    for (int i = 0; i < 2; ++i)
        if (a->first_pair[i] < 0 || a->first_pair[i] >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "duplex_frame: token id %d outside the vocabulary", a->first_pair[i]);
    if (a->probe_id >= h->cfg.vocab_size) return fail(RCA_ERR_ARG, "duplex_frame: probe id %d outside the vocabulary", a->probe_id);
    for (int i = 0; !cap_only && i < a->F_ctx; ++i)
        if (a->code_ctx[i] < 0 || a->code_ctx[i] >= n_codes) return fail(RCA_ERR_ARG, "decode: code out of range [0, %d)", n_codes);
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    if ((rc = duplex_reserve(h, a->T, a->F_ctx, n, a->n_samples)) != RCA_OK) return rc;
    DuplexState* d = h->duplex;
    const int F = a->F_ctx + n;
    const size_t pin_pcm = sizeof(DuplexPin), pin_codes = pin_pcm + (((size_t)a->T * 4 + 255) & ~(size_t)255),
                 pin_out = pin_codes + (((size_t)a->F_ctx * 8 + 255) & ~(size_t)255);
    if (d->logits_at_capture != (const void*)h->logits) { duplex_drop_graphs(d); d->logits_at_capture = h->logits; }
    DuplexPin* pin = reinterpret_cast<DuplexPin*>(d->pin);
    // stage the inputs
    h->h_stt->n_tokens = h->n_tokens;
    h->h_stt->m = 2;
    h->h_stt->ids[0] = a->first_pair[0];
    h->h_stt->ids[1] = a->first_pair[1];
    for (int i = 0; i < n; ++i) h->h_stt->ids[LM_FRAME_USER0 + i] = 0;
    pin->probe_id = a->probe_id >= 0 ? a->probe_id : 0;
    if (!cap_only) {
        memcpy(d->pin + pin_pcm, a->pcm_window, (size_t)a->T * 4);
        if (a->F_ctx) memcpy(d->pin + pin_codes, a->code_ctx, (size_t)a->F_ctx * 8);
    }
    int bucket = 0;
    const int need = lm_splits_needed(h, 2 * n);
    while (bucket + 1 < LM_GRAPH_BUCKETS && (4 << bucket) < need) ++bucket;
    if (cap_only) bucket = cap_bucket;
    const int nsp_launch = bucket + 1 == LM_GRAPH_BUCKETS ? h->n_splits : std::min(h->n_splits, 4 << bucket);
    uint64_t csig = 0;
    if ((rc = rca_codec_workspace_sig(codec, &csig)) != RCA_OK) return rc;
    DuplexGraphKey key{a->T, a->F_ctx, n, a->n_samples, a->probe_id >= 0 ? 1 : 0, bucket, a->code_token_base, (const void*)h->kc, csig, (const void*)codec};
    DuplexState::Entry* ent = nullptr;
    for (auto& e : d->graphs)
        if (e.key == key) ent = &e;
    if (!ent) {
        if (d->graphs.size() >= 32) { RCA_HIP(hipStreamSynchronize(st)); duplex_drop_graphs(d); }   // stale shapes / caches: start over
        d->graphs.push_back(DuplexState::Entry{key, nullptr});
        ent = &d->graphs.back();
    }
    bool warmed = false;
    for (auto& w : d->warm)
        warmed = warmed || (w.T == a->T && w.F_ctx == a->F_ctx && w.n_steps == n && w.n_samples == a->n_samples && w.codec == (const void*)codec && w.codec_sig == csig);
    auto enqueue = [&]() -> int {
        hipError_t e = hipMemcpyAsync(h->stt, h->h_stt, LM_STATE_DECODE_BYTES, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d->pcm_in, d->pin + pin_pcm, (size_t)a->T * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && a->F_ctx) e = hipMemcpyAsync(d->code_win, d->pin + pin_codes, (size_t)a->F_ctx * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&d->dev->probe_id, &pin->probe_id, 4, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "duplex h2d: %s", hipGetErrorString(e));
        // Step 0 evaluates the PREVIOUS frame's pair: this chunk's codes are first needed by the sampler tail of step 0, which makes
        // [token just sampled, user's code 0] the next pair.  The encode tail therefore runs on a side branch beside step 0's layers
        // (a dozen latency-bound launches of a few workgroups each) and joins in front of step 0's sampler.
        hipStream_t se = d->fork ? d->side : st;
        if (d->fork) {
            e = hipEventRecord(d->ev_fork, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(se, d->ev_fork, 0);
            if (e != hipSuccess) return fail(RCA_ERR_HIP, "duplex fork: %s", hipGetErrorString(e));
        }
        int r = rca_codec_encode_tail_dev(codec, d->pcm_in, 1, a->T, n, (int64_t*)d->dev->user_codes, se);
        if (r != RCA_OK) return r;
        duplex_codes_to_ids_kernel<<<1, 64, 0, se>>>(d->dev->user_codes, n, a->code_token_base, h->stt, &d->dev->flags);
        if (d->fork) {
            e = hipEventRecord(d->ev_join, se);
            if (e != hipSuccess) return fail(RCA_ERR_HIP, "duplex join: %s", hipGetErrorString(e));
        }
        for (int i = 0; i < n; ++i) {
            if ((r = lm_enqueue_pass(h, 2, 1, st, nsp_launch, i > 0)) != RCA_OK) return r;
            if (i == 0 && d->fork) {
                e = hipStreamWaitEvent(st, d->ev_join, 0);
                if (e != hipSuccess) return fail(RCA_ERR_HIP, "duplex join: %s", hipGetErrorString(e));
            }
            lm_enqueue_sample(h, h->logits, st, i);
        }
        duplex_tokens_to_codes_kernel<<<1, 64, 0, st>>>(h->stt, n, a->code_token_base, n_codes, d->code_win + a->F_ctx, &d->dev->flags);
        if ((r = rca_codec_decode_tail_dev(codec, (const int64_t*)d->code_win, 1, F, a->n_samples, d->pcm_out, st)) != RCA_OK) return r;
        if (a->probe_id >= 0) {
            lm_softmax_slices_kernel<<<PROBS_SLICES, 1024, 0, st>>>(h->logits, h->cfg.vocab_size, h->probs_dev + 64);
            lm_token_probs_kernel<<<1, 64, 0, st>>>(h->logits, h->cfg.vocab_size, h->probs_dev + 64, &d->dev->probe_id, 1, &d->dev->probe_prob);
        }
        RCA_LAUNCH_CHECK();
        e = hipMemcpyAsync(pin->frame_out, h->stt->frame_out, sizeof(int) * LM_FRAME_MAX, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(pin->user_codes, d->dev->user_codes, sizeof(long long) * LM_FRAME_MAX + 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d->pin + pin_out, d->pcm_out, (size_t)a->n_samples * 4, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "duplex d2h: %s", hipGetErrorString(e));
        return RCA_OK;
    };
    // the codec's ordering moves to this stream (nothing of it is in flight elsewhere once this returns)
    if ((rc = rca_codec_stream_handoff(codec, d->fork ? d->side : st)) != RCA_OK) return rc;
    if (cap_only && !warmed) {
        // size the codec workspace of this shape without touching the LM: the two tail calls on zeroed scratch input
        hipStream_t se = d->fork ? d->side : st;
        RCA_HIP(hipMemsetAsync(d->pcm_in, 0, (size_t)a->T * 4, se));
        RCA_HIP(hipMemsetAsync(d->code_win, 0, (size_t)F * 8, se));
        if ((rc = rca_codec_encode_tail_dev(codec, d->pcm_in, 1, a->T, n, (int64_t*)d->dev->user_codes, se)) != RCA_OK) return rc;
        if ((rc = rca_codec_decode_tail_dev(codec, (const int64_t*)d->code_win, 1, F, a->n_samples, d->pcm_out, se)) != RCA_OK) return rc;
        RCA_HIP(hipStreamSynchronize(se));
        uint64_t csig2 = 0;
        if ((rc = rca_codec_workspace_sig(codec, &csig2)) != RCA_OK) return rc;
        if (d->warm.size() >= 16) d->warm.clear();
        d->warm.push_back(DuplexState::Warm{a->T, a->F_ctx, n, a->n_samples, (const void*)codec, csig2});
        ent->key.codec_sig = csig2;
        warmed = true;
    }
    const bool want_graph = h->graphs_enabled && warmed;
    if (want_graph && !ent->exec) {
        // the eager frame before this one sized every workspace buffer of this shape: nothing allocates under capture
        hipGraph_t g = nullptr;
        RCA_HIP(hipStreamSynchronize(st));
        RCA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = enqueue();
        hipError_t e2 = hipStreamEndCapture(st, &g);
        if (rc != RCA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e2 != hipSuccess) return fail(RCA_ERR_HIP, "duplex capture: %s", hipGetErrorString(e2));
        e2 = hipGraphInstantiate(&ent->exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) { ent->exec = nullptr; return fail(RCA_ERR_HIP, "duplex graph instantiate: %s", hipGetErrorString(e2)); }
        (void)hipGraphUpload(ent->exec, st);
        // a workspace that moved under the eager run (first call of a shape) would have changed the signature: checked by the key
    }
    if (cap_only) return RCA_OK;
    if (want_graph) {
        RCA_HIP(hipGraphLaunch(ent->exec, st));
    } else if ((rc = enqueue()) != RCA_OK) {
        return rc;
    }
    RCA_HIP(hipStreamSynchronize(st));
    // the eager run may have (re)allocated codec workspace: graphs are keyed by the signature after it
    if (!warmed) {
        uint64_t csig2 = 0;
        if ((rc = rca_codec_workspace_sig(codec, &csig2)) != RCA_OK) return rc;
        if (d->warm.size() >= 16) d->warm.clear();
        d->warm.push_back(DuplexState::Warm{a->T, a->F_ctx, n, a->n_samples, (const void*)codec, csig2});
        ent->key.codec_sig = csig2;
    }
    int done = n;
    for (int i = 0; i < n; ++i) {
        out->tokens[i] = pin->frame_out[i];
        out->user_codes[i] = pin->user_codes[i];
        if (out->tokens[i] <= a->audio_id_floor) { done = i + 1; break; }
    }
    for (int i = done; i < n; ++i) { out->tokens[i] = -1; out->user_codes[i] = pin->user_codes[i]; }
    out->n_done = done;
    out->flags = pin->flags | (done < n ? 2 : 0);
    out->probe_prob = (a->probe_id >= 0 && done == n) ? pin->probe_prob : -1.0f;
    if (out->flags == 0) memcpy(pcm_out_host, d->pin + pin_out, (size_t)a->n_samples * 4);
    h->n_tokens += 2 * done;
    h->logits_rows = done < n ? 0 : 1;
    h->rng_host += (unsigned long long)done;
    if (done < n) RCA_HIP(hipMemcpy(&h->stt->rng_counter, &h->rng_host, 8, hipMemcpyHostToDevice));
    return RCA_OK;
}

extern "C" int rca_duplex_frame(rca_lm_t* h, rca_codec_t* codec, const rca_duplex_frame_args_t* a, rca_duplex_frame_out_t* out,
                                float* pcm_out_host) {
    return duplex_frame_core(h, codec, a, out, pcm_out_host, -1);
}

static int lm_step_capture_only(rca_lm_t* h, int n, int n_probe, int bucket);

// Captures, ahead of the first frame, every graph the one-replay frame of this call shape can need: one per context bucket the
// handle's n_ctx can reach, on the KV cache installed in `h` and -- when the session trims through a shadow cache (kv_shadow.py:
// the two caches trade places at every trim) -- on `twin`'s as well; plus, with n_probe > 0, the one-token step + n_probe
// probabilities of the agent's speculative <|end_audio|> step (rca_lm_step_probe).  Called at session start (reset()): no frame of
// the session then pays a capture (in round 3 the first frame of every bucket and the first trim frame did: 10-14 ms against a
// median of 4.7).  a->pcm_window / code_ctx / first_pair are not read.
extern "C" int rca_duplex_precapture(rca_lm_t* h, rca_lm_t* twin, rca_codec_t* codec, const rca_duplex_frame_args_t* a, int32_t n_probe) {
    if (!h || !codec || !a) return fail(RCA_ERR_ARG, "duplex_precapture: null argument");
    if (!h->sampler_set) return fail(RCA_ERR_STATE, "sampler not initialised");
    if (!h->graphs_enabled) return RCA_OK;
    if (twin && (twin == h || twin->cfg.n_ctx != h->cfg.n_ctx || twin->n_splits != h->n_splits)) return fail(RCA_ERR_ARG, "duplex_precapture: the twin must be another handle of the same n_ctx");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    if (twin) { const int src = lm_settle(twin); if (src != RCA_OK) return src; }
    int rc = RCA_OK;
    for (int pass = 0; pass < (twin ? 2 : 1) && rc == RCA_OK; ++pass) {
        if (pass == 1) { std::swap(h->kc, twin->kc); std::swap(h->vc, twin->vc); }      // host pointers only: nothing runs under capture
        for (int b = 0; b < LM_GRAPH_BUCKETS && rc == RCA_OK; ++b) {
            if (b > 0 && (4 << (b - 1)) >= h->n_splits) break;                       // the previous bucket already launches every split
            rc = duplex_frame_core(h, codec, a, nullptr, nullptr, b);
            if (rc == RCA_OK && n_probe > 0) rc = lm_step_capture_only(h, 1, n_probe, b);
            // the frames that cannot take the one-replay path (a trim, a text branch) replay the LM chunk or single steps
            if (rc == RCA_OK) {
                const int32_t pair[2] = {0, 0}, users[LM_FRAME_MAX] = {0, 0, 0, 0, 0, 0, 0, 0};
                int32_t toks[LM_FRAME_MAX], done = 0;
                rc = lm_frame_core(h, pair, users, a->n_steps, -1, toks, &done, b);
            }
            if (rc == RCA_OK) rc = lm_step_capture_only(h, 2, 0, b);
        }
        if (pass == 1) { std::swap(h->kc, twin->kc); std::swap(h->vc, twin->vc); }
    }
    return rc;
}

extern "C" int rca_lm_token_probs(rca_lm_t* h, const int32_t* token_ids, int32_t n, float* probs_out) {
    if (!h || !token_ids || !probs_out || n < 1 || n > 64) return fail(RCA_ERR_ARG, "token_probs: 1..64 ids");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    if (h->logits_rows < 1) return fail(RCA_ERR_STATE, "no logits: call eval first");
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    RCA_HIP(hipMemcpyAsync(h->probe_ids_dev, token_ids, n * 4, hipMemcpyHostToDevice, st));
    const float* lg = h->logits + (long)(h->logits_rows - 1) * h->cfg.vocab_size;
    lm_softmax_slices_kernel<<<PROBS_SLICES, 1024, 0, st>>>(lg, h->cfg.vocab_size, h->probs_dev + 64);
    lm_token_probs_kernel<<<1, 64, 0, st>>>(lg, h->cfg.vocab_size, h->probs_dev + 64, h->probe_ids_dev, n, h->probs_dev);
    RCA_LAUNCH_CHECK();
    RCA_HIP(hipMemcpyAsync(probs_out, h->probs_dev, n * 4, hipMemcpyDeviceToHost, st));
    RCA_HIP(hipStreamSynchronize(st));
    return RCA_OK;
}

// switch between "logits of every evaluated position" (llama_cpp's logits_all) and "last position only"
extern "C" int rca_lm_set_logits_all(rca_lm_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    if ((h->cfg.logits_all != 0) != (enable != 0)) {
        RCA_HIP(hipSetDevice(h->device));
        RCA_HIP(hipStreamSynchronize(h->stream));
        lm_drop_graphs(h);   // a later eval may move the logits buffer; nothing captured before the switch is replayed after it
    }
    h->cfg.logits_all = enable != 0;
    return RCA_OK;
}

// test / bench knob: disable graph replay (eager launches) to compare
extern "C" int rca_lm_set_graphs(rca_lm_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    h->graphs_enabled = enable != 0;
    return RCA_OK;
}

__global__ __launch_bounds__(256) void lm_q8_zero_row_scales_kernel(unsigned* __restrict__ sc, int nblk, int row_begin, int row_end) {
    const long total = (long)(row_end - row_begin) * nblk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = row_begin + (int)(i / nblk), j = (int)(i % nblk);
        unsigned short* half = reinterpret_cast<unsigned short*>(sc + q8_sc_index(row >> 1, j, nblk)) + (row & 1);
        *half = 0;
    }
}
__global__ __launch_bounds__(256) void lm_q6k_zero_row_scales_kernel(float* __restrict__ sc, int nk16, int row_begin, int row_end) {
    const long total = (long)(row_end - row_begin) * nk16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = row_begin + (int)(i / nk16), j = (int)(i % nk16);
        sc[2 * q8_sc_index(row >> 1, j, nk16) + (row & 1)] = 0.0f;
    }
}
__global__ __launch_bounds__(256) void lm_q4k_zero_row_factors_kernel(unsigned* __restrict__ dd, int nk256, int row_begin, int row_end) {
    const long total = (long)(row_end - row_begin) * nk256;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = row_begin + (int)(i / nk256), j = (int)(i % nk256);
        dd[q4k_scm_index(row, j, nk256)] = 0u;   // lm_head is packed with plain slots: slot == row
    }
}
__global__ __launch_bounds__(256) void lm_zero_rows_kernel(bf16_t* __restrict__ w, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) w[i] = 0;
}
// zero lm_head rows [row_begin, row_end): used with random-init weights so that, like a trained codec LM in
// audio mode, the sampler's top-k only ever holds codec tokens (text rows get logit 0)
extern "C" int rca_lm_mask_head_rows(rca_lm_t* h, int32_t row_begin, int32_t row_end) {
    if (!h || row_begin < 0 || row_end > h->cfg.vocab_size || row_begin > row_end) return fail(RCA_ERR_ARG, "mask_head_rows: bad range");
    RCA_HIP(hipSetDevice(h->device));
    const long n = (long)(row_end - row_begin) * h->cfg.hidden;
    if (n > 0 && h->head.fmt == WF_Q6K)   // Q6_K (what a Q4_K_M file keeps output.weight in): the row's f32 scales
        lm_q6k_zero_row_scales_kernel<<<256, 256, 0, h->stream>>>((float*)h->head.sc, h->cfg.hidden / 16, row_begin, row_end);
    if (n > 0 && h->head.fmt == WF_Q4K)   // Q4_K: d = dmin = 0 makes every value of the row (0 * sc) * q - (0 * m) = 0
        lm_q4k_zero_row_factors_kernel<<<256, 256, 0, h->stream>>>(h->head.dd, h->cfg.hidden / 256, row_begin, row_end);
    if (n > 0 && (h->head.fmt == WF_BF16 || h->head.fmt == WF_F16)) lm_zero_rows_kernel<<<2048, 256, 0, h->stream>>>(h->head.w + (long)row_begin * h->cfg.hidden, n);   // zero bits
    if (n > 0 && h->head.fmt == WF_Q8)   // the packed q8_0 head: a row is zero when its block scales are
        lm_q8_zero_row_scales_kernel<<<256, 256, 0, h->stream>>>(h->head.sc, h->cfg.hidden / 32, row_begin, row_end);
    RCA_LAUNCH_CHECK();
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}

// ------------------------------------------------------------------------------------- persist_codec_embeddings
// The reference's deployment step (codec_llama.py:178-206): every codec token's input embedding is the projector output
// linear_2(gelu(linear_1(codec_embed[code]))) (codec_llama.py:32-44, exact erf GELU) and is baked into the plain embedding
// table so the deployed model is a vanilla Llama.  Done here on the device for a checkpoint that still carries the frozen
// codec embedding + projector: a small elementwise kernel for the 16 -> H layer, then an H x H f32 GEMM on
// v_mfma_f32_32x32x2_f32 (code rows x features; one 32 x 32 tile per wave, operands straight from L2 -- a load-time
// operation, 1.1 TFLOP for the 1B model), bias added and rounded to bf16 (nearest even) into the table rows.
__global__ __launch_bounds__(256) void lm_codec_proj1_kernel(const float* __restrict__ e, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             float* __restrict__ h1, int rows, int dim, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * H) return;
    const int r = (int)(i / H), j = (int)(i - (long)r * H);
    float acc = 0.0f;
    for (int d = 0; d < dim; ++d) acc = __builtin_fmaf(w1[(long)j * dim + d], e[(long)r * dim + d], acc);
    const float x = acc + b1[j];
    h1[i] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__global__ __launch_bounds__(256) void lm_codec_proj2_kernel(const float* __restrict__ h1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                             void* __restrict__ table_rows, int f32tab, float* __restrict__ out_f32, int rows, int H) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hs = lane >> 5;
    const int n0 = blockIdx.y * 32, f0 = (blockIdx.x * 4 + wave) * 32;
    if (f0 >= H) return;
    // A = h1 (code rows), B = w2 (feature rows), both contiguous along k; a lane takes 4 consecutive k of its row per 8-block
    // (k half `hs`), so MFMA step s of a block multiplies k0+s (half 0) and k0+4+s (half 1): same pairing on both operands.
    const float* ap = h1 + (long)min(n0 + r32, rows - 1) * H + hs * 4;
    const float* bp = w2 + (long)min(f0 + r32, H - 1) * H + hs * 4;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k0 = 0; k0 < H; k0 += 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + k0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + k0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    const int f = f0 + r32;   // accumulator i of lane: code row 8*(i/4) + 4*hs + i%4, feature r32
    if (f >= H) return;
    const float bias = b2[f];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + 8 * (i / 4) + 4 * hs + (i & 3);
        if (n < rows) {
            const float v = acc[i] + bias;
            if (f32tab) reinterpret_cast<float*>(table_rows)[(long)n * H + f] = v;   // an f32 table keeps the projector output as it is
            else reinterpret_cast<bf16_t*>(table_rows)[(long)n * H + f] = f32_to_bf16_rne(v);
            if (out_f32) out_f32[(long)n * H + f] = v;
        }
    }
}

extern "C" int rca_lm_persist_codec_embeddings(rca_lm_t* h, const float* codec_embed, int32_t n_codes, int32_t dim, const float* w1,
                                               const float* b1, const float* w2, const float* b2, int32_t codec_vocab_start, float* out_f32) {
    if (!h || !codec_embed || !w1 || !b1 || !w2 || !b2) return fail(RCA_ERR_ARG, "persist_codec_embeddings: null argument");
    const int H = h->cfg.hidden;
    if (n_codes <= 0 || dim <= 0 || codec_vocab_start < 0 || (long)codec_vocab_start + n_codes > h->cfg.vocab_size)
        return fail(RCA_ERR_ARG, "persist_codec_embeddings: rows [%d, %ld) are outside the vocabulary of %d", codec_vocab_start,
                    (long)codec_vocab_start + n_codes, h->cfg.vocab_size);
    if (H % 8) return fail(RCA_ERR_ARG, "persist_codec_embeddings: hidden size must be a multiple of 8");
    RCA_HIP(hipSetDevice(h->device));
    const int chunk = std::min<int>(n_codes, 8192);
    float *de = nullptr, *dw1 = nullptr, *db1 = nullptr, *dw2 = nullptr, *db2 = nullptr, *dh1 = nullptr, *dout = nullptr;
    auto free_all = [&]() { for (float* p : {de, dw1, db1, dw2, db2, dh1, dout}) if (p) (void)hipFree(p); };
    auto up = [&](float** d, const float* src, size_t n) {
        if (hipMalloc((void**)d, n * 4) != hipSuccess) return false;
        return hipMemcpy(*d, src, n * 4, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&de, codec_embed, (size_t)n_codes * dim) || !up(&dw1, w1, (size_t)H * dim) || !up(&db1, b1, H) || !up(&dw2, w2, (size_t)H * H) ||
        !up(&db2, b2, H) || hipMalloc((void**)&dh1, (size_t)chunk * H * 4) != hipSuccess ||
        (out_f32 && hipMalloc((void**)&dout, (size_t)chunk * H * 4) != hipSuccess)) {
        free_all();
        return fail(RCA_ERR_HIP, "persist_codec_embeddings: device allocation / upload failed");
    }
    for (int r0 = 0; r0 < n_codes; r0 += chunk) {
        const int rows = std::min(chunk, n_codes - r0);
        lm_codec_proj1_kernel<<<(unsigned)(((long)rows * H + 255) / 256), 256, 0, h->stream>>>(de + (long)r0 * dim, dw1, db1, dh1, rows, dim, H);
        lm_codec_proj2_kernel<<<dim3((H + 127) / 128, (rows + 31) / 32), 256, 0, h->stream>>>(
            dh1, dw2, db2, (char*)h->embed + ((long)codec_vocab_start + r0) * H * (h->embed_f32 ? 4 : 2), h->embed_f32, dout, rows, H);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && out_f32) e = hipMemcpyAsync(out_f32 + (long)r0 * H, dout, (size_t)rows * H * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { free_all(); return fail(RCA_ERR_HIP, "persist_codec_embeddings: %s", hipGetErrorString(e)); }
    }
    free_all();
    return RCA_OK;
}

// test / bench knob: merge the attention splits inside the attention launch (1, default) or in a launch of its own (0)
extern "C" int rca_lm_set_attn_fuse(rca_lm_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    { const int src = lm_settle(h); if (src != RCA_OK) return src; }
    RCA_HIP(hipSetDevice(h->device));
    RCA_HIP(hipStreamSynchronize(h->stream));
    if (h->fuse_attn != (enable != 0)) lm_drop_graphs(h);
    h->fuse_attn = enable != 0;
    return RCA_OK;
}
// the format the projection matrices are kept (and streamed) in: 0 bf16, 1 q8_0, 2 f16, 3 q4_k (4 = Q6_K, which only ever appears next to Q4_K
// tensors); bytes = weight bytes one decode step reads
extern "C" int rca_lm_weight_format(const rca_lm_t* h, int32_t* fmt, int64_t* bytes_per_step) {
    if (!h || !fmt) return fail(RCA_ERR_ARG, "null");
    *fmt = h->layers.empty() ? h->head.fmt : h->layers[0].gu.fmt;
    if (bytes_per_step) {
        long b = h->head.stream_bytes();
        for (const LmLayer& L : h->layers) b += L.qkv.stream_bytes() + L.o.stream_bytes() + L.gu.stream_bytes() + L.down.stream_bytes() + (L.split_v ? L.vseg.stream_bytes() : 0);
        *bytes_per_step = b;
    }
    return RCA_OK;
}

// test / bench knob: route long evals through the exact GEMV chunks (0) or the bf16 MFMA prefill tiles (1, default)
extern "C" int rca_lm_set_mfma_prefill(rca_lm_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    h->mfma_prefill = enable != 0;
    return RCA_OK;
}
