// rca_codec.hip -- MagiCodec-style codec on gfx950 (MI355X): strided conv1d encoder,
// 131072x16 codebook nearest-neighbour search, transposed-conv decoder.
//
// Replaces the third-party arithmetic behind AudioTokenizer._magicodec_encode/_decode
// (reference audio_tokenizer.py:189-201).  All arithmetic is f32 with every
// multiply-accumulate an fma in a fixed k order (ci-major, tap-minor), so results are
// bit-identical to oracle/codec_oracle.c:
//   * variant 0 ("chain")  : one thread per output element, scalar v_fma_f32 chain
//   * variant 1 ("mfma")   : implicit-GEMM conv on v_mfma_f32_32x32x2_f32 with an
//                            LDS-staged im2col sliding window, and an MFMA scoring
//                            kernel for the codebook search.  gfx950's f32 MFMA is a
//                            k-ordered fma chain (one rounding per product), so the
//                            same bits come out.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no implicit contraction).
#include <stdarg.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <type_traits>

#include "rca_common.h"

namespace rca {
thread_local char g_err[512] = {0};
}
using namespace rca;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Source addressing for the first layer: row r of the batch reads
//   src + (r % C) * chan_stride + (r / C) * win_stride,  T valid samples (zero beyond).
// Plain encode: C = B, chan_stride = T, win_stride = 0.  Window mode (batch encode): rows are
// (window, channel) pairs cut out of a long [C][N] signal on the fly -- no window copy in HBM.
#define RCA_LAT_MAX_FRAMES 64   // streaming tail calls of at most this many (row x frame) pairs run on the latency kernels
#define RCA_LDS_BUDGET (152 * 1024)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RowSrc {
    const float* base;
    int C;
    long chan_stride;
    long win_stride;
    int T;
    // optional per-row table (device): row b starts at base + row_off[b] instead of the (channel, window) lattice above --
    // windows of many files batched into one pass (rca_codec_encode_rows_dev); `span` = max offset + T (host-known)
    const long* row_off = nullptr;
    long span = 0;
    __device__ __forceinline__ long off(long b) const {
        return row_off ? row_off[b] : (b % C) * chan_stride + (b / C) * win_stride;
    }
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

// ---------------------------------------------------------------------------- conv_in (Cin = 1)
// HBM-bound (448 FLOP per 128 B written): one thread per (row, t) computes every output channel
// from a k-tap register window; stores are coalesced along t for each channel.
template <int KS>
__global__ __launch_bounds__(256) void conv_in_kernel(RowSrc src, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int B, int Cout, int Lout) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Lout) return;
    const int b = (int)(idx / Lout);
    const int t = (int)(idx - (long)b * Lout);
    const float* xr = src.base + src.off(b);
    constexpr int padL = KS / 2;
    float xv[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int i = t + kk - padL;
        xv[kk] = (i >= 0 && i < src.T) ? xr[i] : 0.0f;
    }
    float* yo = y + (long)b * Cout * Lout + t;
    for (int co = 0; co < Cout; ++co) {
        float acc = bias[co];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int i = t + kk - padL;
            // skip out-of-range taps exactly like the oracle (keeps -0.0 bookkeeping identical)
            if (i >= 0 && i < src.T) acc = __builtin_fmaf(w[co * KS + kk], xv[kk], acc);
        }
        yo[(long)co * Lout] = acc;
    }
}

// ------------------------------------------------------------------------- generic chain conv
// y[b][co][t] = bias[co] + sum_{ci} sum_{kk} w[co][ci][kk] * pre(x[b][ci][t*s + kk - padL])
__global__ __launch_bounds__(256) void conv1d_chain_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int B, int Cin, int Lin, int Cout, int Lout, int k, int s,
                                                           int pre, float slope, int clamp_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Cout * Lout) return;
    const int t = (int)(idx % Lout);
    const long r = idx / Lout;
    const int co = (int)(r % Cout);
    const int b = (int)(r / Cout);
    const int padL = (k - s + 1) / 2;
    const float* xb = x + (long)b * Cin * Lin;
    const float* wr = w + (long)co * Cin * k;
    float acc = bias[co];
    const int i0 = t * s - padL;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xr = xb + (long)ci * Lin;
        for (int kk = 0; kk < k; ++kk) {
            const int i = i0 + kk;
            if (i >= 0 && i < Lin) {
                float v = xr[i];
                if (pre) v = lrelu(v, slope);
                acc = __builtin_fmaf(wr[ci * k + kk], v, acc);
            }
        }
    }
    if (clamp_out) acc = acc > 1.0f ? 1.0f : (acc < -1.0f ? -1.0f : acc);
    y[idx] = acc;
}

// ConvTranspose1d, weights [Cin][Cout][k], k = 2s, padL = (k-s+1)/2, Lout = Lin*s
__global__ __launch_bounds__(256) void convtr1d_chain_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int B, int Cin, int Lin, int Cout, int k, int s, int pre,
                                                             float slope) {
    const int Lout = Lin * s;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Cout * Lout) return;
    const int u = (int)(idx % Lout);
    const long r = idx / Lout;
    const int co = (int)(r % Cout);
    const int b = (int)(r / Cout);
    const int padL = (k - s + 1) / 2;
    const int kk0 = (u + padL) % s;
    const float* xb = x + (long)b * Cin * Lin;
    float acc = bias[co];
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xr = xb + (long)ci * Lin;
        const float* wr = w + ((long)ci * Cout + co) * k;
        for (int kk = kk0; kk < k; kk += s) {
            const int t = (u + padL - kk) / s;
            if (t >= 0 && t < Lin) {
                float v = xr[t];
                if (pre) v = lrelu(v, slope);
                acc = __builtin_fmaf(wr[kk], v, acc);
            }
        }
    }
    y[idx] = acc;
}

// ----------------------------------------------------------------- implicit-GEMM conv on f32 MFMA
// GEMM view: M = Cout, N = B*Lout (columns flattened over windows, so Lout = 100 wastes nothing),
// K = Cin*KS walked ci-major / tap-minor in chunks of CIC input channels.
//
// Every WAVE is a self-contained pipeline over its own (WM*32 channels) x (WN*32 columns) tile: no
// workgroup barrier anywhere.  Per chunk the wave (a) issues the loads of the NEXT chunk's input span,
// (b) runs the MFMA block of the current chunk from its private LDS window, refilling its single set of
// weight fragments quad by quad for the next chunk as it goes, (c) writes the next window to LDS.  The
// other wave of the SIMD fills the matrix pipe while one stages.  Everything outside the MFMA blocks is
// kept short on purpose: a non-MFMA instruction costs the SIMD about a quarter of an MFMA slot while the
// neighbour streams MFMAs (measured with the RCA_CONV_TIMELINE stamps below).
//
// LDS sliding window (per wave): for every input channel of the chunk the contiguous input span of
// the wave's columns is staged ONCE (coalesced buffer loads; LeakyReLU at the LDS write unless the
// producing layer already stored activated values), de-interleaved by stride phase:
// xs[ci][p][slot] = act(x[b][ci][t*S + p]) for the column (b,t) in `slot` (one halo slot each side).
// Tap kk of column j is xs[ci][(kk-padL) mod S][j + floor((kk-padL)/S)]: consecutive lanes read
// consecutive LDS words (no bank conflicts, no im2col copy).  Reads that would cross a window edge are
// zeroed by a per-lane bit mask, in the few waves that hold a row edge (fma(w, 0, acc) == acc; the
// oracle skips those taps).  v_mfma_f32_32x32x2_f32 is issued in ascending k into WM x WN accumulator
// tiles; A operands (weights) come pre-packed in fragment order from L2 (wp[co_tile][kquad][lane][4]).
// Buffer loads through a V# descriptor (base, num_records): a lane whose offset is >= num_records reads 0 without touching
// memory.  The LLVM intrinsics are bound by name with float result types (this compiler folds the integer-vector forms,
// __builtin_amdgcn_raw_buffer_load_b64 / .v2i32, into a single dword load).
typedef int rca_rsrc_t __attribute__((ext_vector_type(4)));
typedef float rca_f32x2_t __attribute__((ext_vector_type(2)));
__device__ float rca_buffer_load_f32(rca_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ rca_f32x2_t rca_buffer_load_f32x2(rca_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
typedef float rca_f32x4_t __attribute__((ext_vector_type(4)));
__device__ rca_f32x4_t rca_buffer_load_f32x4(rca_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ rca_rsrc_t rca_make_rsrc(const void* base, int num_records) {   // base must be wave-uniform
    const unsigned long a = (unsigned long)base;
    rca_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xFFFF));   // stride 0: raw buffer
    r[2] = num_records;
    r[3] = 0x00020000;
    return r;
}

// Optional per-wave timeline of conv1d_mfma_kernel (build with -DRCA_CONV_TIMELINE, run scripts/conv_timeline.py):
// entry / first chunk staged / chunk loop done / stores issued, in wall_clock64 ticks (10 ns), plus the hardware slot.
#ifdef RCA_CONV_TIMELINE
__device__ long* rca_prof_buf = nullptr;
#define RCA_TL_STAMP(v) const long v = tl_buf ? (long)wall_clock64() : 0   // tl_buf: the buffer pointer, read once per wave
#define RCA_TL_ADD(acc, a, b) acc += (b) - (a)
#define RCA_TL_MIN(acc, a, b) acc = ((b) - (a)) < acc ? ((b) - (a)) : acc
#else
#define RCA_TL_STAMP(v)
#define RCA_TL_ADD(acc, a, b)
#define RCA_TL_MIN(acc, a, b)
#endif
template <int S>
struct ConvLds {
    // row stride U (in floats) for NT columns: NT + 2 halo slots, padded so that the S phase rows a
    // staging wave writes together fall on different banks
    static constexpr int want = (S == 8) ? 4 : (S == 4) ? 8 : (S == 2) ? 16 : (S == 5) ? 6 : 2;
    static constexpr int stride(int NT) {
        int u = NT + 2;
        while ((u % 32) != want % 32) ++u;
        return u;
    }
};

// FUSE = 1 (first strided layer only): the input channels are not read from HBM but recomputed on the fly as
// conv_in(PCM) -- 7-tap Cin=1 conv, bias first, taps ascending, exactly conv_in_kernel's chain -- from a
// per-lane register window of raw PCM that is loaded once per wave.  Removes conv_in's 1 GB write and this
// layer's 1 GB read per 256-window step.
struct FuseIn {
    RowSrc src;           // raw PCM rows
    const float* w_in;    // [Cin][7]
    const float* b_in;    // [Cin]
};
// TR = 1: one launch computes a whole ConvTranspose1d (k = 2s) as s phase GEMMs.  Output u = s*t0 - padL + r
// (phase r) = bias + sum_ci ( w[ci][co][r] * x[ci][t0] + w[ci][co][r+s] * x[ci][t0-1] ): a 2-tap stride-1 conv over
// Lin+1 columns per row whose k order (channel-major, tap r before tap r+s) is the oracle's.  The phase rides
// on the channel-tile index (weights packed per phase), so one column tile's phases share the XCD's L2.
struct TrInfo {
    int s, padL, Lout, n_co;   // upsampling stride, left pad, output length per row, real channel tiles
};
// LDS word offset of the B operand of k pair kp for the lanes of one half (without the lane's column): the lanes of half 1 hold the
// odd k of the pair.  The difference between the halves takes one or two values per layer shape, so a lane needs one or two base
// registers (column + half * difference) and every read adds a compile-time constant -- not one address register per k pair.
template <int KS, int S, int TR>
struct ConvBOff {
    static constexpr int padL = (KS - S + 1) / 2;
    static constexpr int q_of(int kp, int half) {
        const int kl = 2 * kp + half, kk = kl % KS;
        const int d = TR ? -kk : kk - padL;
        return (d >= 0) ? d / S : -((-d + S - 1) / S);
    }
    static constexpr int off(int kp, int half, int U) {
        const int kl = 2 * kp + half, ci = kl / KS, kk = kl % KS;
        const int d = TR ? -kk : kk - padL;
        const int q = q_of(kp, half);
        return (ci * S + (d - q * S)) * U + 1 + q;
    }
    static constexpr int delta(int kp, int U) { return off(kp, 1, U) - off(kp, 0, U); }
    // k-th distinct value of delta over the k pairs of a chunk (the last one repeated when there are fewer)
    static constexpr int distinct(int k, int KPC, int U) {
        int found[4] = {0, 0, 0, 0};
        int n = 0;
        for (int kp = 0; kp < KPC; ++kp) {
            const int d = delta(kp, U);
            bool seen = false;
            for (int i = 0; i < n; ++i) seen = seen || found[i] == d;
            if (!seen && n < 4) found[n++] = d;
        }
        return found[k < n ? k : n - 1];
    }
    static constexpr int count(int KPC, int U) {
        int found[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int kp = 0; kp < KPC; ++kp) {
            const int d = delta(kp, U);
            bool seen = false;
            for (int i = 0; i < n; ++i) seen = seen || found[i] == d;
            if (!seen && n < 8) found[n++] = d;
        }
        return n;
    }
};

// waves per workgroup of conv1d_mfma_kernel: the waves never synchronise with each other, so the workgroup is only a unit of
// dispatch -- with one wave per workgroup a finished wave's slot is refilled at once instead of when its three siblings are done
#ifndef RCA_CONV_WPB
#define RCA_CONV_WPB 1
#endif
#ifndef RCA_CONV_OCC
#define RCA_CONV_OCC 2
#endif
#ifndef RCA_CONV_PIPE
#define RCA_CONV_PIPE 1
#endif
#ifndef RCA_CONV_PRIO
#define RCA_CONV_PRIO 1
#endif
#ifndef RCA_CONV_ABUF
#define RCA_CONV_ABUF 1   // weight refills as buffer loads with scalar offsets
#endif
#ifndef RCA_CONV_EPRIO
#define RCA_CONV_EPRIO 1   // priority of the MFMA blocks (the prologue / epilogue of a wave run at 0)
#endif
#ifdef RCA_ABL_NOBREAD   // timing experiment: B fragments from a register instead of LDS
#define RCA_ABL_BREAD(x) (slope)
#else
#define RCA_ABL_BREAD(x) (x)
#endif
// BF (opt-in, rca_codec_set_mfma_mode; never the default): the same kernel with the products on v_mfma_f32_32x32x16_bf16 instead of
// v_mfma_f32_32x32x2_f32.  Staging, LDS window, weight fragments and B reads are unchanged -- a lane already holds the operand of
// k = 2 kp + half for every k pair kp, and an MFMA may pair its 16 k slots with any 16 k as long as A and B agree -- so eight k pairs
// become ONE bf16 step: slot (half, j) <-> k = 2 (8 g + j) + half.  BF = 3: both operands split into bf16 hi + bf16 lo in registers,
// three MFMAs per step (hi hi + hi lo + lo hi: ~2^-16 relative, the prefill GEMM's scheme); BF = 1: both operands rounded to bf16,
// one MFMA (what the reference's bf16 autocast does, audio_tokenizer.py:24,78-82).  The matrix pipe's internal summation order is
// not documented, so neither is bit-exact against the f32 fma chain: code ids near a tie can differ (measured and reported).
typedef __bf16 conv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 conv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned conv_pack_bf16x2(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even
    const conv_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
template <int BF>
__device__ __forceinline__ void conv_split8(const float (&w)[8], uint4& hi, uint4& lo) {
    unsigned ph[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ph[j] = conv_pack_bf16x2(w[2 * j], w[2 * j + 1]);
        pl[j] = BF == 3 ? conv_pack_bf16x2(w[2 * j] - __uint_as_float(ph[j] << 16), w[2 * j + 1] - __uint_as_float(ph[j] & 0xffff0000u)) : 0u;
    }
    hi = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    lo = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}
// FULLC = 1 (chosen by the launcher when Cin is a multiple of the chunk -- every layer of the encoder): no channel of any chunk lies
// past Cin, so the per-channel choice between the input descriptor and the empty one (4 s_cselect + a compare per channel and
// chunk: 40-60 scalar instructions of a ~150-instruction staging phase) is gone.  Same loads, same values.
template <int KS, int S, int CIC, int WM, int WN, int FUSE, int TR, int BF = 0, int FULLC = 0>
__global__ __launch_bounds__(64 * RCA_CONV_WPB, (((KS == 8 && CIC == 2) || (KS == 16 && CIC == 1)) && WM * WN < 8) ? 3 : RCA_CONV_OCC) void conv1d_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int Cin, int Lin, int Cout, int Lout, long Ncols, int nchunks,
                                                          int pre, float slope, FuseIn fin, TrInfo tr) {
    constexpr int MT = WM * 32;            // channels per wave (and per workgroup)
    constexpr int NW = WN * 32;            // columns per wave
    constexpr int U = ConvLds<S>::stride(NW);
    constexpr int KPC = CIC * KS / 2;      // k pairs per chunk
    constexpr int QPC = KPC / 4;           // float4 weight quads per chunk per co-tile
    constexpr int E = (NW + 2) * S;        // staged elements per input channel
    // a lane stages PW consecutive elements per group: with an even stride two neighbours share their slot (phases p, p+1),
    // never straddle a row or the signal edge, come from one 8-byte load and go to LDS in one ds_write2 (offsets 0, U)
    constexpr int PW = (S % 2 == 0 && !TR) ? 2 : 1;
    constexpr int RE = (E + 64 * PW - 1) / (64 * PW);   // groups per lane
    constexpr int padL = (KS - S + 1) / 2;
    constexpr int BUF = CIC * S * U + 4;   // + a spare word that absorbs the lanes past the window
    static_assert((CIC * KS) % 8 == 0, "chunk must hold whole weight quads");

    extern __shared__ __attribute__((aligned(16))) float xs_all[];  // [RCA_CONV_WPB waves][2][CIC][S][U]
#ifdef RCA_CONV_TIMELINE
    long* const tl_buf = rca_prof_buf;
#endif
    RCA_TL_STAMP(tl0);

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xs = xs_all + wave * 2 * BUF;
    // Workgroup id -> (column tile, channel tile).  Workgroups are dealt round-robin to the 8 XCDs, each with
    // its own L2: consecutive workgroups OF ONE XCD walk the channel tiles of the same column tile, so the
    // input span is fetched from HBM once and re-read from that XCD's L2 (speed only; any placement is correct).
    const int n_co_real = (Cout + MT - 1) / MT;
    const int n_co = TR ? n_co_real * tr.s : n_co_real;   // TR: (phase, channel tile) pairs
    const long wg = blockIdx.x;
    const int xcd = (int)(wg & 7);
    const long seq = wg >> 3;
    const int co_tile_x = __builtin_amdgcn_readfirstlane((int)(seq % n_co));   // uniform, but the 64-bit division runs on the VALU: pin it to an SGPR
    const int phase = TR ? co_tile_x / n_co_real : 0;
    const int co_tile = TR ? co_tile_x % n_co_real : co_tile_x;
    const long col_tile = (seq / n_co) * 8 + xcd;
    const long n0 = (col_tile * RCA_CONV_WPB + wave) * NW;   // first column of this wave
    const int co0 = co_tile * MT;
    if (n0 >= Ncols) return;  // whole wave out of range (no barriers: safe)
    // column -> (batch row, position): every column this wave touches is n_base + dn with 0 <= dn <= NW + 2, and the
    // wave-uniform (b_base, t_base) of n_base turns that into 32-bit arithmetic (at most one row crossing when a row is
    // longer than the window)
    const long n_base = n0 > 0 ? n0 - 1 : 0;
    const int b_base = __builtin_amdgcn_readfirstlane((int)((unsigned)n_base / (unsigned)Lout));   // Ncols < 2^31 (host check)
    const int t_base = (int)(n_base - (long)b_base * Lout);
    const int lead = n0 > 0 ? 0 : 1;                            // staged slot 0 is column n0 - 1: absent for the very first wave
    const long left = Ncols - n_base;
    const int ncol_left = left > NW + 2 ? NW + 2 : (int)left;   // column n_base + dn exists for dn < ncol_left
    auto rel_bt = [&](int dn, int& bb, int& tt) {
        int dt = dn + t_base;
        bb = 0;
        if (Lout >= NW + 3) {
            if (dt >= Lout) { dt -= Lout; bb = 1; }
        } else {
            bb = dt / Lout;
            dt -= bb * Lout;
        }
        tt = dt;
    };

    // ---- staging role: element e -> (slot, phase); fixed per lane, channel added per load.  Loads go through a buffer
    // descriptor over the wave's first batch row: a slot outside the signal carries an out-of-range offset and reads 0,
    // so nothing has to be masked between the load and the LDS write.
    constexpr unsigned OOB = 0x80000000u;
    const rca_rsrc_t rs_x = rca_make_rsrc(FUSE ? fin.src.base : x + (long)b_base * Cin * Lin, 0x7FFFFFFF);
    const rca_rsrc_t rs_none = rca_make_rsrc(FUSE ? fin.src.base : x, 0);
    unsigned s_boff[RE];   // byte offset of x[b][0][t*S + p] from row b_base (< 2^31 by the host check), or OOB
    int s_loff[RE];        // p*U + slot; lanes past the window: a padding slot, or -1 (write predicated) when the rows have none
    bool s_ok[FUSE ? RE : 1];
    // ---- fused conv_in: 7-sample PCM window per staged element (zero outside the row's valid samples).  The two batch
    // rows a wave can touch (b_base, b_base + 1: the host only fuses when a row is longer than the window) are addressed
    // from the lower of their two PCM rows.
    float pc[FUSE ? RE : 1][6 + PW];
    rca_rsrc_t rs_pcm = rs_none;
    unsigned prow[2] = {0u, 0u};
    if (FUSE) {
        long o0, o1;
        if (fin.src.row_off) {   // rows of several files: two wave-uniform table reads (the row after the last one is never staged)
            const long nrows = (Ncols + Lout - 1) / Lout;
            o0 = fin.src.row_off[b_base];
            o1 = fin.src.row_off[b_base + 1 < nrows ? b_base + 1 : b_base];
        } else {
            const int C = fin.src.C;
            const int c0 = b_base % C, w0 = b_base / C;
            const int c1 = c0 + 1 == C ? 0 : c0 + 1, w1 = c0 + 1 == C ? w0 + 1 : w0;
            o0 = (long)c0 * fin.src.chan_stride + (long)w0 * fin.src.win_stride;
            o1 = (long)c1 * fin.src.chan_stride + (long)w1 * fin.src.win_stride;
        }
        const long om = o0 < o1 ? o0 : o1;
        rs_pcm = rca_make_rsrc(fin.src.base + om, 0x7FFFFFFF);
        prow[0] = (unsigned)(o0 - om);
        prow[1] = (unsigned)(o1 - om);
    }
#pragma unroll
    for (int r = 0; r < RE; ++r) {
        const int e = PW * (lane + 64 * r);   // first element of the group
        const int slot = e / S, p = e - slot * S;
        const int dn = slot - lead;
        bool ok = e < E && dn >= 0 && dn < ncol_left;
        int bb, t;
        rel_bt(ok ? dn : 0, bb, t);
        if (TR && t >= Lin) ok = false;   // column t0 = Lin exists (its x[t0-1] tap is valid) but has no x[t0]
        // lanes past the window write into the padding slot at the end of a phase row (U > NW + 2) so the LDS write needs no predicate;
        // where the rows have no padding (stride-1 layers) they keep -1 and the write is predicated
        s_loff[r] = e < E ? p * U + slot : (U > NW + 2 ? U - 1 : -1);
        s_boff[r] = ok ? ((unsigned)bb * (unsigned)(Cin * Lin) + (unsigned)(t * S + p)) * 4u : OOB;
        if (FUSE) {
            s_ok[r] = ok;
            const int i = t * S + p;  // PCM sample index == conv_in output index
            const unsigned ro = bb ? prow[1] : prow[0];
#pragma unroll
            for (int kk = 0; kk < 6 + PW; ++kk) {
                const int j = i + kk - 3;
                const unsigned off = (ok && j >= 0 && j < fin.src.T) ? (ro + (unsigned)j) * 4u : OOB;
                pc[r][kk] = rca_buffer_load_f32(rs_pcm, (int)off, 0, 0);
            }
        }
    }
    float sreg[CIC][RE][PW];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int cl = 0; cl < CIC; ++cl) {
            const int ci = c * CIC + cl;
            if (FUSE) {
                const int cc = ci < Cin ? ci : 0;
                const float* wr = fin.w_in + cc * 7;   // wave-uniform: scalar loads
                const float bi = fin.b_in[cc];
#pragma unroll
                for (int r = 0; r < RE; ++r)
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        float a = bi;
                        // out-of-range taps hold 0: fma(w, 0, a) == a (the oracle skips them)
#pragma unroll
                        for (int kk = 0; kk < 7; ++kk) a = __builtin_fmaf(wr[kk], pc[r][kk + j], a);
                        // computed, not loaded: the value is final here (zero outside the signal, pre-activation applied), so
                        // the write phase after the MFMA block is LDS stores only
                        a = (s_ok[r] && ci < Cin) ? a : 0.0f;
                        if (pre & 1) a = fmaxf(a, a * slope);
                        sreg[cl][r][j] = a;
                    }
            } else {
                // raw value only: nothing consumes it before the MFMA block.  A channel past Cin (last chunk of a layer
                // whose Cin is not a multiple of CIC) reads through the empty descriptor: zeros.
#ifdef RCA_ABL_NOXLOAD   // timing experiment: every input load reads through the empty descriptor (returns 0 without touching memory)
                const rca_rsrc_t rs = rs_none;
#else
                const rca_rsrc_t rs = (FULLC || ci < Cin) ? rs_x : rs_none;
#endif
                const int soff = ((FULLC || ci < Cin) ? ci : 0) * Lin * 4;
#pragma unroll
                for (int r = 0; r < RE; ++r) {
                    if (PW == 2) {
                        const rca_f32x2_t v2 = rca_buffer_load_f32x2(rs, (int)s_boff[r], soff, 0);
                        sreg[cl][r][0] = v2[0];
                        sreg[cl][r][PW - 1] = v2[1];
                    } else {
                        sreg[cl][r][0] = rca_buffer_load_f32(rs, (int)s_boff[r], soff, 0);
                    }
                }
            }
        }
    };
    const long kquads = (long)nchunks * QPC;
    // A fragments (weights), pre-packed in fragment order: quad q of chunk c for 32-row tile wm
    // weight quads through a buffer descriptor over this wave's first 32-row tile: the lane part of the address is one constant
    // VGPR (lane * 16), everything else -- 32-row tile, chunk, quad -- is scalar (SALU soffset + immediate), so a refill costs no
    // vector address arithmetic.  (The pointer form below adds 64-bit offsets per load on the VALU.)
    const rca_rsrc_t rs_w = rca_make_rsrc(reinterpret_cast<const float4*>(wp) + ((long)((TR ? phase * tr.n_co : 0) + co_tile * WM) * kquads) * 64, 0x7FFFFFFF);
    const int w_lane = lane * 16;
    auto w_load = [&](int wm, int c, int q) __attribute__((always_inline)) {
        const rca_f32x4_t v = rca_buffer_load_f32x4(rs_w, w_lane, (int)(((long)wm * kquads + (long)c * QPC + q) * 1024), 0);
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto w_ptr = [&](int wm, int c) {
        const int cot = (TR ? phase * tr.n_co : 0) + co_tile * WM + wm;   // TR: tr.n_co 32-row tiles per phase
        return reinterpret_cast<const float4*>(wp) + ((long)cot * kquads + (long)c * QPC) * 64 + lane;
    };
    float4 a[WM][QPC];
    // The first chunk's input loads and weight quads go out before the rest of the prologue (B-read offsets, edge masks, bias): that
    // arithmetic then runs under their memory latency instead of in front of it.  (Fused first layer: its staged values are computed
    // from the PCM window requested above, so only the weights can be requested here.)
    if (!FUSE) stage_load(0);
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        const float4* p0 = w_ptr(wm, 0);
#pragma unroll
        for (int q = 0; q < QPC; ++q) a[wm][q] = RCA_CONV_ABUF ? w_load(wm, 0, q) : p0[q * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- B-read role: per k-pair LDS offset and edge masks for this lane's half
    typedef ConvBOff<KS, S, TR> BO;    // TR: tap 0 reads x[t0], tap 1 reads x[t0-1]
    static_assert(BO::count(KPC, U) <= 3, "B-operand base registers");
    constexpr int BD0 = BO::distinct(0, KPC, U), BD1 = BO::distinct(1, KPC, U), BD2 = BO::distinct(2, KPC, U);
    const int b_base0 = (lane & 31) + half * BD0, b_base1 = (lane & 31) + half * BD1, b_base2 = (lane & 31) + half * BD2;
    auto b_off = [&](int kp) __attribute__((always_inline)) {   // kp is a constant after unrolling: one of the bases + an immediate
        const int dl = BO::delta(kp, U);
        return (dl == BD0 ? b_base0 : dl == BD1 ? b_base1 : b_base2) + BO::off(kp, 0, U);
    };
    unsigned m_first = 0, m_last = 0;  // bit kp: the read reaches into the previous / next column
#pragma unroll
    for (int kp = 0; kp < KPC; ++kp) {
        const int q = half ? BO::q_of(kp, 1) : BO::q_of(kp, 0);
        if (q < 0) m_first |= 1u << kp;
        if (q > 0) m_last |= 1u << kp;
    }
    unsigned zmask[WN];
    unsigned zany = 0;
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        int bz, t;
        rel_bt(1 - lead + wn * 32 + (lane & 31), bz, t);
        zmask[wn] = (t == 0 ? m_first : 0u) | (t == Lout - 1 ? m_last : 0u);
        zany |= zmask[wn];
    }
    // most waves hold no row edge: they run the MFMA block without the per-fragment edge selects
    const bool edges = __builtin_amdgcn_ballot_w64(zany != 0) != 0;

    // accumulators start at the bias; `bias` is padded to whole channel tiles, rows past Cout are never stored
    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + co0 + wm * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) acc[wm][wn][4 * q + j] = bq[j];
        }
    }

    // pre-activation applied here (ACT) unless the layer that produced x already stored activated values
    auto stage_write_t = [&](int buf, auto act_tag) __attribute__((always_inline)) {
        constexpr bool ACT = decltype(act_tag)::value;
#ifdef RCA_ABL_NOWRITE   // timing experiment: the staged values never reach LDS (and nothing waits for their loads)
        return;
#endif
        float* dst = xs + buf * BUF;
#pragma unroll
        for (int cl = 0; cl < CIC; ++cl) {
#pragma unroll
            for (int r = 0; r < RE; ++r) {
                float v[PW];
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    v[j] = sreg[cl][r][j];
                    // LeakyReLU as max(v, slope*v): identical values for 0 < slope < 1 (incl. -0), two VALU ops
                    if (ACT && !FUSE) v[j] = fmaxf(v[j], v[j] * slope);
                }
                // only the last group can hold lanes past the window: one LDS address per group plus immediates
                if (U > NW + 2 || 64 * PW * (r + 1) <= E || s_loff[r] >= 0) {
#pragma unroll
                    for (int j = 0; j < PW; ++j) dst[cl * S * U + s_loff[r] + j * U] = v[j];
                }
            }
        }
    };
    auto stage_write = [&](int buf) __attribute__((always_inline)) {
        if (pre & 1) stage_write_t(buf, std::true_type{});
        else stage_write_t(buf, std::false_type{});
    };

    // One chunk of MFMAs.  ONE set of weight fragments: as soon as the last k pair of quad q has been issued, the same
    // registers are refilled with quad q of chunk `cn` (a full chunk of MFMAs, ~2 us, ahead of their next use), so the
    // weights cost QPC x WM x 4 registers instead of twice that.
    auto compute_t = [&](int buf, int cn, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const float* xb = xs + buf * BUF;
        const float4* pn[WM];
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) pn[wm] = w_ptr(wm, cn);
        // the B fragments of (half) a chunk are requested up front; the MFMAs then consume them in order
        // behind counted lgkmcnt waits, so the matrix pipe is not re-stalled on LDS latency every k step
        constexpr int NB = (KPC * WN > 32) ? ((KPC % 4 == 0) ? 4 : 2) : (KPC * WN == 32 && WN == 4) ? 2 : 1;  // register budget for the fragment prefetch
        constexpr int PB = KPC / NB;
        static_assert(KPC % NB == 0, "batching");
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float bv[PB][WN];
#pragma unroll
            for (int i = 0; i < PB; ++i)
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) bv[i][wn] = xb[b_off(nb * PB + i) + wn * 32];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int kp = nb * PB + i;
                float bfr[WN];
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) bfr[wn] = (EDGE && ((zmask[wn] >> kp) & 1u)) ? 0.0f : bv[i][wn];
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) {
                    const float4 q4 = a[wm][kp >> 2];
                    const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn)
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfr[wn], acc[wm][wn], 0, 0, 0);
                }
                if ((kp & 3) == 3) {
#pragma unroll
                    for (int wm = 0; wm < WM; ++wm) a[wm][kp >> 2] = RCA_CONV_ABUF ? w_load(wm, cn, kp >> 2) : pn[wm][(kp >> 2) * 64];
                }
            }
        }
    };

    // The same chunk as an explicit software pipeline (RCA_CONV_PIPE, the default): the B fragments run RD k pairs ahead of the
    // MFMAs that consume them -- the first RD of a block are requested as soon as its window is in LDS, a whole staging phase
    // earlier -- the weight quad is refilled right behind its last MFMA, and a scheduling barrier per k pair keeps that order
    // (left alone, the scheduler issues each batch of reads right in front of its first use and sinks the refills to the end of the block).
    // measured per layer (same box, pipelined vs batched block): k16s8 950 vs 962 us, k8s4 1150 vs 1130, k10s5 1102 vs 1087, fused k4s2
    // 799 vs 788, k3 225 vs 220 -> the pipeline is used for the k = 16 layer only (all within a few per cent: the block is not
    // what bounds these layers)
    constexpr bool PIPE = RCA_CONV_PIPE && !TR && KS == 16 && BF == 0;
    constexpr int RD = KPC >= 8 ? 4 : 2;
    float bhead[RD][WN];
    auto read_head = [&](int buf) __attribute__((always_inline)) {
        const float* xb = xs + buf * BUF;
#pragma unroll
        for (int i = 0; i < RD; ++i)
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) bhead[i][wn] = RCA_ABL_BREAD(xb[b_off(i) + wn * 32]);
    };
    auto compute_p = [&](int buf, int cn, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const float* xb = xs + buf * BUF;
        const float4* pn[WM];
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) pn[wm] = w_ptr(wm, cn);
        float ring[KPC][WN];   // statically indexed: RD + 1 entries live at a time
#pragma unroll
        for (int i = 0; i < RD; ++i)
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) ring[i][wn] = bhead[i][wn];
#pragma unroll
        for (int kp = 0; kp < KPC; ++kp) {
#if RCA_CONV_PRIO
            // Priority rises with progress through the block: of the two waves that share a SIMD the one further along wins the matrix
            // pipe, finishes, and does its staging while the other has the pipe to itself -- the pair settles into alternation instead of
            // entering and leaving their MFMA blocks together (round-robin issue keeps two waves that started together in lock step:
            // both then stage at the same time and the pipe idles)
            if (kp == KPC / 4 && !RCA_CONV_EPRIO) __builtin_amdgcn_s_setprio(1);
            if (kp == KPC / 2) __builtin_amdgcn_s_setprio(2);
            if (kp == 3 * KPC / 4) __builtin_amdgcn_s_setprio(3);
#endif
            if (kp + RD < KPC) {
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) ring[kp + RD][wn] = RCA_ABL_BREAD(xb[b_off(kp + RD) + wn * 32]);
            }
            float bfr[WN];
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) bfr[wn] = (EDGE && ((zmask[wn] >> kp) & 1u)) ? 0.0f : ring[kp][wn];
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) {
                const float4 q4 = a[wm][kp >> 2];
                const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                for (int wn = 0; wn < WN; ++wn)
                    acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfr[wn], acc[wm][wn], 0, 0, 0);
            }
#ifndef RCA_ABL_NOAREFILL   // (timing experiment when defined: the weights of chunk 0 are reused)
            if ((kp & 3) == 3) {
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) a[wm][kp >> 2] = RCA_CONV_ABUF ? w_load(wm, cn, kp >> 2) : pn[wm][(kp >> 2) * 64];
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // BF != 0: the chunk on the bf16 matrix instruction, eight k pairs per step (see the note above the kernel)
    auto compute_b = [&](int buf, int cn, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const float* xb = xs + buf * BUF;
        constexpr int NG = (KPC + 7) / 8;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            uint4 bh[WN], bl[WN];
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) {
                float bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kp = g * 8 + i;
                    bv[i] = kp < KPC ? xb[b_off(kp < KPC ? kp : 0) + wn * 32] : 0.0f;
                    if (EDGE && kp < KPC && ((zmask[wn] >> kp) & 1u)) bv[i] = 0.0f;
                }
                conv_split8<BF>(bv, bh[wn], bl[wn]);
            }
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) {
                float aw[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * g + qq;
                    const float4 q4 = q < QPC ? a[wm][q < QPC ? q : 0] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    aw[4 * qq + 0] = q4.x; aw[4 * qq + 1] = q4.y; aw[4 * qq + 2] = q4.z; aw[4 * qq + 3] = q4.w;
                }
                uint4 ah, al;
                conv_split8<BF>(aw, ah, al);
                const conv_bf16x8 ahv = __builtin_bit_cast(conv_bf16x8, ah), alv = __builtin_bit_cast(conv_bf16x8, al);
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) {
                    const conv_bf16x8 bhv = __builtin_bit_cast(conv_bf16x8, bh[wn]);
                    acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, bhv, acc[wm][wn], 0, 0, 0);
                    if (BF == 3) {
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, __builtin_bit_cast(conv_bf16x8, bl[wn]), acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alv, bhv, acc[wm][wn], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * g + qq;
                    if (q < QPC) a[wm][q] = w_load(wm, cn, q);
                }
            }
        }
    };

    if (FUSE) stage_load(0);
    stage_write(0);
    __builtin_amdgcn_wave_barrier();
    if constexpr (PIPE) read_head(0);
    RCA_TL_STAMP(tl1);
#ifdef RCA_CONV_TIMELINE
    long tl_load = 1 << 30, tl_mfma = 0, tl_write = 0;   // even chunks only: SHORTEST MFMA block / sum of the MFMA blocks / activation + LDS write
#endif
    const int lastc = nchunks - 1;
    // The whole chunk loop exists twice, with and without the edge selects, and the wave picks one ONCE: a branch per chunk joins
    // two register allocations of the 64 accumulators behind every MFMA block (the compiler then copies them back -- 32 moves behind
    // a drained matrix pipe per chunk -- and sinks the weight refills out of the block to the join).
    auto chunk_loop = [&](auto edge_tag, auto act_tag) __attribute__((always_inline)) {
    for (int c = 0; c < nchunks; c += 2) {
        // The next chunk's loads are issued UNCONDITIONALLY (past the end they re-read the last chunk and the
        // result is never used): a branch around them would merge two paths in front of the MFMA block and
        // force its s_waitcnt to the conservative vmcnt(~0) of the path without newer loads.
        // sched_barrier(0) pins the three phases in program order so the scheduler does not sink the loads
        // down to their first use.
        const int c1 = min(c + 1, lastc);
        RCA_TL_STAMP(ta);
        stage_load(c1);
        __builtin_amdgcn_sched_barrier(0);
        RCA_TL_STAMP(tb);
        __builtin_amdgcn_s_setprio(RCA_CONV_EPRIO);
        if constexpr (BF != 0) compute_b(0, c1, edge_tag);
        else if constexpr (PIPE) compute_p(0, c1, edge_tag);
        else compute_t(0, c1, edge_tag);
        __builtin_amdgcn_s_setprio(RCA_CONV_PRIO ? 3 : 2);  // staging phases run at raised priority (measured +2.4 %)
        __builtin_amdgcn_sched_barrier(0);
        RCA_TL_STAMP(tc);
        stage_write_t(1, act_tag);
        __builtin_amdgcn_wave_barrier();
        if constexpr (PIPE) {
            read_head(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        RCA_TL_STAMP(td);
        RCA_TL_MIN(tl_load, tb, tc); RCA_TL_ADD(tl_mfma, tb, tc); RCA_TL_ADD(tl_write, tc, td);
        if (c + 1 >= nchunks) break;
        const int c2 = min(c + 2, lastc);
        stage_load(c2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(RCA_CONV_EPRIO);
        if constexpr (BF != 0) compute_b(1, c2, edge_tag);
        else if constexpr (PIPE) compute_p(1, c2, edge_tag);
        else compute_t(1, c2, edge_tag);
        __builtin_amdgcn_s_setprio(RCA_CONV_PRIO ? 3 : 2);
        __builtin_amdgcn_sched_barrier(0);
        stage_write_t(0, act_tag);
        __builtin_amdgcn_wave_barrier();
        if constexpr (PIPE) {
            read_head(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    };
    if (pre & 1) {
        if (edges) chunk_loop(std::true_type{}, std::true_type{});
        else chunk_loop(std::false_type{}, std::true_type{});
    } else {
        if (edges) chunk_loop(std::true_type{}, std::false_type{});
        else chunk_loop(std::false_type{}, std::false_type{});
    }

    // prologue and epilogue run below the MFMA blocks: the partner wave's matrix instructions are never queued behind their address
    // arithmetic and stores
    if (RCA_CONV_EPRIO) __builtin_amdgcn_s_setprio(0);
    RCA_TL_STAMP(tl2);
    // epilogue: C/D layout col = lane&31 (column), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel).  A store is
    // (wave-uniform row pointer) + (per-lane 32-bit element offset): scalar address arithmetic only.
    const long orow = TR ? tr.Lout : Lout;
    const bool full_rows = co0 + MT <= Cout;
    float* const yw = y + ((long)b_base * Cout + co0) * orow;
    // (16-byte stores after a 4 x 4 transpose inside each quad of lanes -- two quad_perm butterflies, a quarter of the store
    //  instructions -- were built and measured in round 3: bit-identical, slower on every layer (fused k4s2 689 vs 648 us, k8s4 1040
    //  vs 1026): the 64 extra moves + selects per tile cost more next to the neighbour's MFMA stream than the stores they replace.)
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int dn = 1 - lead + wn * 32 + (lane & 31);
        if (dn >= ncol_left) continue;
        int bb, t;
        rel_bt(dn, bb, t);
        int ocol = t;
        if (TR) {   // phase r of column t0 lands on output sample s*t0 - padL + r
            ocol = tr.s * t - tr.padL + phase;
            if (ocol < 0 || ocol >= (int)orow) continue;
        }
        // byte offset of (row bb, channel co0 + 4*half, column ocol) from yw: below 2^32 by the host check
        const unsigned voff = ((unsigned)bb * (unsigned)(Cout * (int)orow) + (unsigned)ocol + (unsigned)(4 * half) * (unsigned)orow) * 4u;
        if (pre & 2) {   // the consumer of y is a pre-activated layer: store LeakyReLU(y) once here instead of re-applying it at every read
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[wm][wn][r] = fmaxf(acc[wm][wn][r], acc[wm][wn][r] * slope);
        }
        if (full_rows) {
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cu = wm * 32 + (r & 3) + 8 * (r >> 2);   // + 4*half, folded into voff
                    char* const yr = reinterpret_cast<char*>(yw + (long)cu * orow);
                    *reinterpret_cast<float*>(yr + voff) = acc[wm][wn][r];
                }
        } else {
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cu = wm * 32 + (r & 3) + 8 * (r >> 2);
                    char* const yr = reinterpret_cast<char*>(yw + (long)cu * orow);
                    if (co0 + cu + 4 * half < Cout) *reinterpret_cast<float*>(yr + voff) = acc[wm][wn][r];
                }
        }
    }
#ifdef RCA_CONV_TIMELINE
    if (tl_buf && lane == 0) {
        const long tl3 = (long)wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID: wave slot, SIMD, CU, SE
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID
        // one region of 65536 records per kernel size, indexed by wave: no atomics (a single counter serialises 64k waves)
        constexpr int region = KS == 4 ? 0 : KS == 8 ? 1 : KS == 10 ? 2 : KS == 16 ? 3 : KS == 3 ? 4 : 5;
        long* o = tl_buf + ((long)region * 65536 + ((blockIdx.x * RCA_CONV_WPB + wave) & 0xFFFF)) * 8;
        o[6] = tl_load | (tl_mfma << 32);
        o[7] = tl_write | ((long)nchunks << 32);
        o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = tl3;
        o[4] = (long)hw | ((long)xcc << 32);
        o[5] = (long)KS | ((long)S << 8) | ((long)WM << 16) | ((long)wave << 24) | ((long)blockIdx.x << 32);
    }
#endif
}

// Same GEMM view, LDS window layout, k order and results as conv1d_mfma_kernel, different division of
// labour: a workgroup is 8 waves = 4 CONSUMERS (waves 0-3, one per SIMD) + 4 PRODUCERS (waves 4-7).
// Consumer i owns a 64-channel x 64-column tile and does nothing but LDS fragment reads and
// v_mfma_f32_32x32x2_f32 (the matrix pipe of its SIMD sees a dense stream); producer i stages consumer i's
// input window (global loads one chunk ahead, LeakyReLU / fused conv_in, LDS writes) and a quarter of the
// chunk's weight fragments, which all four consumers share through LDS.  One workgroup barrier per chunk
// (32 k-pairs = 128 MFMAs per consumer) hands buffer (c+1)&1 over.
template <int KS, int S, int CIC, int FUSE>
__global__ __launch_bounds__(512) void conv1d_ws_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int Cin, int Lin, int Cout, int Lout, long Ncols, int nchunks,
                                                        int pre, float slope, FuseIn fin) {
    constexpr int WM = 2, WN = 2;
    constexpr int MT = WM * 32, NW = WN * 32;
    constexpr int U = ConvLds<S>::stride(NW);
    constexpr int KPC = CIC * KS / 2;
    constexpr int QPC = KPC / 4;
    constexpr int E = (NW + 2) * S;
    constexpr int RE = (E + 63) / 64;
    constexpr int padL = (KS - S + 1) / 2;
    constexpr int BUF = CIC * S * U + 4;
    constexpr int WBUF = WM * QPC * 64 * 4;   // floats of weight fragments per chunk
    static_assert(KPC % 4 == 0 && KPC <= 64, "chunk shape");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* aw_all = lds;                       // [2][WM][QPC][64] float4
    float* xs_all = lds + 2 * WBUF;            // [4 consumers][2][BUF]

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int cw = wave & 3;                   // consumer index this wave is / serves
    float* xs = xs_all + cw * 2 * BUF;

    const int n_co = (Cout + MT - 1) / MT;
    const long wg = blockIdx.x;
    const int xcd = (int)(wg & 7);
    const long seq = wg >> 3;
    const int co_tile = (int)(seq % n_co);
    const long col_tile = (seq / n_co) * 8 + xcd;
    const long n0 = (col_tile * 4 + cw) * NW;
    const int co0 = co_tile * MT;
    const bool active = n0 < Ncols;            // inactive waves still take part in every barrier
    const long kquads = (long)nchunks * QPC;
    const int lastc = nchunks - 1;

    if (producer) {
        // ------------------------------------------------------------------ producer
        unsigned s_goff[RE];
        int s_loff[RE];
        bool s_ok[RE];
#pragma unroll
        for (int r = 0; r < RE; ++r) {
            const int e = lane + 64 * r;
            const int slot = e / S, p = e - slot * S;
            const long n = n0 - 1 + slot;
            s_ok[r] = active && e < E && n >= 0 && n < Ncols;
            // lanes past the window write into the padding slot at the end of a phase row (U > NW + 2) so the LDS write needs no predicate;
        // where the rows have no padding (stride-1 layers) they keep -1 and the write is predicated
        s_loff[r] = e < E ? p * U + slot : -1;
            const long nn = s_ok[r] ? n : 0;
            const long b = nn / Lout;
            const int t = (int)(nn - b * Lout);
            s_goff[r] = s_ok[r] ? (unsigned)(b * Cin * (long)Lin + (long)t * S + p) : 0u;
        }
        float pc[FUSE ? RE : 1][7];
        if (FUSE) {
#pragma unroll
            for (int r = 0; r < RE; ++r) {
                const int e = lane + 64 * r;
                const int slot = e / S, p = e - slot * S;
                const long n = n0 - 1 + slot;
                const long nn = s_ok[r] ? n : 0;
                const long b = nn / Lout;
                const int i = (int)(nn - b * Lout) * S + p;
                const float* row = fin.src.base + fin.src.off(b);
#pragma unroll
                for (int kk = 0; kk < 7; ++kk) {
                    const int j = i + kk - 3;
                    pc[r][kk] = (s_ok[r] && j >= 0 && j < fin.src.T) ? row[j] : 0.0f;
                }
            }
        }
        const float act_slope = pre ? slope : 1.0f;
        // weight share of this producer: items t, t+256, ... of the WM*QPC*64 float4 of a chunk
        constexpr int WITEMS = WM * QPC * 64;
        constexpr int WPER = (WITEMS + 255) / 256;
        const int ptid = threadIdx.x - 256;
        float sreg[CIC][RE];
        float4 wreg[WPER];
        auto issue_loads = [&](int c) {
#pragma unroll
            for (int i = 0; i < WPER; ++i) {
                const int it = ptid + 256 * i;
                const int itc = it < WITEMS ? it : 0;
                const int wm = itc / (QPC * 64), rem = itc - wm * (QPC * 64);   // rem = q*64 + lane
                const int cot = co_tile * WM + wm;
                wreg[i] = reinterpret_cast<const float4*>(wp)[((long)cot * kquads + (long)c * QPC) * 64 + rem];
            }
#pragma unroll
            for (int cl = 0; cl < CIC; ++cl) {
                const int ci = c * CIC + cl;
                if (FUSE) {
                    const int cc = ci < Cin ? ci : 0;
                    const float* wr = fin.w_in + cc * 7;
                    const float bi = fin.b_in[cc];
#pragma unroll
                    for (int r = 0; r < RE; ++r) {
                        float a = bi;
#pragma unroll
                        for (int kk = 0; kk < 7; ++kk) a = __builtin_fmaf(wr[kk], pc[r][kk], a);
                        sreg[cl][r] = a;
                    }
                } else {
                    const float* xc = x + (long)(ci < Cin ? ci : 0) * Lin;
#pragma unroll
                    for (int r = 0; r < RE; ++r) sreg[cl][r] = xc[s_goff[r]];
                }
            }
        };
        auto write_lds = [&](int c, int buf) {
            float4* aw = reinterpret_cast<float4*>(aw_all + buf * WBUF);
#pragma unroll
            for (int i = 0; i < WPER; ++i) {
                const int it = ptid + 256 * i;
                if (it < WITEMS) aw[it] = wreg[i];
            }
            float* dst = xs + buf * BUF;
#pragma unroll
            for (int cl = 0; cl < CIC; ++cl) {
                const bool cok = (c * CIC + cl) < Cin;
#pragma unroll
                for (int r = 0; r < RE; ++r) {
                    float v = (s_ok[r] && cok) ? sreg[cl][r] : 0.0f;
                    v = fmaxf(v, v * act_slope);
                    dst[s_loff[r] >= 0 ? cl * S * U + s_loff[r] : CIC * S * U] = v;
                }
            }
        };
        issue_loads(0);
        write_lds(0, 0);
        issue_loads(min(1, lastc));
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int c1 = min(c + 1, lastc);
            write_lds(c1, (c + 1) & 1);          // loaded during the previous iteration
            issue_loads(min(c + 2, lastc));      // in flight across the barrier and the next iteration
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    __builtin_amdgcn_s_setprio(3);
    int b_off[KPC];
    unsigned long long m_first = 0, m_last = 0;
#pragma unroll
    for (int kp = 0; kp < KPC; ++kp) {
        const int kl = 2 * kp + half;
        const int ci = kl / KS, kk = kl - ci * KS;
        const int d = kk - padL;
        const int q = (d >= 0) ? d / S : -((-d + S - 1) / S);
        const int p = d - q * S;
        b_off[kp] = (ci * S + p) * U + 1 + q + (lane & 31);
        if (q < 0) m_first |= 1ull << kp;
        if (q > 0) m_last |= 1ull << kp;
    }
    unsigned long long zmask[WN];
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const long n = n0 + wn * 32 + (lane & 31);
        const int t = (int)(n % Lout);
        zmask[wn] = (t == 0 ? m_first : 0ull) | (t == Lout - 1 ? m_last : 0ull);
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        const int cot = co0 + wm * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cot + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float bv = co < Cout ? bias[co] : 0.0f;
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) acc[wm][wn][r] = bv;
        }
    }
    __syncthreads();   // chunk 0 staged
    for (int c = 0; c < nchunks; ++c) {
        if (active) {
            const float* xb = xs + (c & 1) * BUF;
            const float4* aw = reinterpret_cast<const float4*>(aw_all + (c & 1) * WBUF) + lane;
            constexpr int NB = (KPC > 16) ? ((KPC % 16 == 0) ? KPC / 16 : ((KPC % 10 == 0) ? KPC / 10 : KPC / 12)) : 1;
            constexpr int PB = KPC / NB;      // k pairs per fragment batch (<= 16, multiple of 2)
            static_assert(KPC % NB == 0 && PB % 2 == 0, "batching");
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float bv[PB][WN];
#pragma unroll
                for (int i = 0; i < PB; ++i)
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) bv[i][wn] = xb[b_off[nb * PB + i] + wn * 32];
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const int kp = nb * PB + i;
                    float bfr[WN];
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) bfr[wn] = ((zmask[wn] >> kp) & 1ull) ? 0.0f : bv[i][wn];
#pragma unroll
                    for (int wm = 0; wm < WM; ++wm) {
                        const float4 q4 = aw[(wm * QPC + (kp >> 2)) * 64];
                        const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int wn = 0; wn < WN; ++wn)
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfr[wn], acc[wm][wn], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const long n = n0 + wn * 32 + (lane & 31);
        if (n >= Ncols) continue;
        const long b = n / Lout;
        const int t = (int)(n - b * Lout);
        float* yb = y + b * Cout * (long)Lout + t;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
            const int cot = co0 + wm * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cot + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < Cout) yb[(long)co * Lout] = acc[wm][wn][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------- quantizer pieces
// z[row][j] = b[j] + sum_d w[j][d] * ze[b][d][f]   for rows (b, f in [f0, f0+fc)); row = b*fc + (f-f0)
// ------------------------------------------------------------------ latency-oriented kernels (streaming tail)
// A streaming tail call is a handful of columns with reduction chains up to 4096 long whose order is fixed (bias
// first, channel-major, tap-minor: the oracle's order).  On the MFMA path every pair of k-steps is one 64-cycle
// instruction and every chunk of channels a round trip to HBM; as a scalar chain a step is one dependent v_fma
// (~8 cycles).  These kernels take everything else off that chain.  The workgroup first pulls its whole input tile
// (zero-filled outside the signal, LeakyReLU already applied) and the weight rows of its NW output channels into LDS
// with every load in flight at once; then each wave (one output channel x up to 64 columns) walks its chain:
//   x  from a sample-major tile xs[j][CP] (CP = Cin + 4): one 16-byte read fetches 4 channels of one tap, whatever
//      the stride / alignment of the layer, double-buffered one channel group ahead;
//   w  wave-uniform: one coalesced read puts the 4*K weights of a channel group in 4*K lanes, each step takes its
//      weight with v_readlane into an SGPR operand -- no LDS traffic per step.
template <int K, int S, int NW>
__global__ __launch_bounds__(64 * NW) void conv1d_lds_kernel(const float* __restrict__ x, long x_batch_stride, int Lvalid,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int Cin, int Lin, int Cout, int Lout, int NC, int W,
                                                             float slope, int pre, int clamp_out) {
    extern __shared__ __attribute__((aligned(16))) float lds_sm[];
    const int CP = Cin + 4;
    float* xs = lds_sm;                   // [W][CP]
    float* ws = lds_sm + (long)W * CP;    // [NW][Cin*K], natural order
    constexpr int padL = (K - S + 1) / 2;
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col0 = blockIdx.x * NC;
    const int co0 = blockIdx.y * NW;
    const int b = blockIdx.z;
    // the tile starts on a multiple of 4 samples at or before the first tap, so whole 16-byte groups are in or out
    const int i_first = col0 * S - padL;
    const int i_base = i_first & ~3;
    const int d0 = i_first - i_base;       // 0..3: where column 0's first tap sits in the tile
    const float* xb = x + (long)b * x_batch_stride;
    if ((Lin & 3) == 0 && Lvalid == Lin && (x_batch_stride & 3) == 0 && (Cin & 3) == 0) {
        // one thread = 4 channels x 4 samples: four 16-byte loads, transposed in registers, four 16-byte LDS stores;
        // consecutive threads take consecutive channel groups, so a wave's stores are one contiguous run per row
        const int W4 = W >> 2, C4 = Cin >> 2;      // Cin % 4 == 0 on this path
        const int nblk = W4 * C4;
#pragma unroll 4
        for (int e = tid; e < nblk; e += NT) {
            const int jg = e / C4, cg = e - jg * C4;
            const int j = jg << 2, ci = cg << 2;
            const int i = i_base + j;
            f32x4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (i >= 0 && i < Lin) v[c] = *reinterpret_cast<const f32x4*>(xb + (long)(ci + c) * Lin + i);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o = {v[0][q], v[1][q], v[2][q], v[3][q]};
                if (pre) { o[0] = lrelu(o[0], slope); o[1] = lrelu(o[1], slope); o[2] = lrelu(o[2], slope); o[3] = lrelu(o[3], slope); }
                *reinterpret_cast<f32x4*>(xs + (long)(j + q) * CP + ci) = o;
            }
        }
    } else {
        const int nx = Cin * W;
        if (Cin >= 64) {
#pragma unroll 8
            for (int e = tid; e < nx; e += NT) {   // short rows (a few frames): consecutive threads -> consecutive channels
                const int j = e / Cin, ci = e - j * Cin;
                const int i = i_base + j;
                float v = (i >= 0 && i < Lvalid) ? xb[(long)ci * Lin + i] : 0.0f;
                if (pre) v = lrelu(v, slope);
                xs[j * CP + ci] = v;
            }
        } else {
#pragma unroll 8
            for (int e = tid; e < nx; e += NT) {   // consecutive threads -> consecutive samples of one channel (coalesced)
                const int ci = e / W, j = e - ci * W;
                const int i = i_base + j;
                float v = (i >= 0 && i < Lvalid) ? xb[(long)ci * Lin + i] : 0.0f;
                if (pre) v = lrelu(v, slope);
                xs[j * CP + ci] = v;
            }
        }
    }
    const int nwe = Cin * K;
    if ((nwe & 3) == 0) {
        const int nw4 = nwe >> 2;
#pragma unroll 16
        for (int e = tid; e < NW * nw4; e += NT) {
            const int wv = e / nw4, q = e - wv * nw4;
            const int co = min(co0 + wv, Cout - 1);
            reinterpret_cast<float4*>(ws)[e] = reinterpret_cast<const float4*>(w + (long)co * nwe)[q];
        }
    } else {
        for (int e = tid; e < NW * nwe; e += NT) {
            const int wv = e / nwe, r = e - wv * nwe;
            ws[e] = w[(long)min(co0 + wv, Cout - 1) * nwe + r];
        }
    }
    __syncthreads();
    const int co = co0 + wave;
    if (co >= Cout) return;
    const int col = col0 + lane;
    const bool valid = lane < NC && col < Lout;
    const float* xl = xs + (long)((valid ? lane * S : 0) + d0) * CP;
    const float* wl = ws + (long)wave * nwe;
    float acc = bias[co];
    if ((Cin & 7) != 0) {   // conv_in (Cin = 1): a chain of K steps
        for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
            for (int kk = 0; kk < K; ++kk) acc = __builtin_fmaf(wl[ci * K + kk], xl[kk * CP + ci], acc);
    } else {
        constexpr int G = 4;
        f32x4 xa[K], xb2[K];
        float wa, wb;
        auto ld = [&](f32x4(&xq)[K], float& wr, int ci) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk) xq[kk] = *reinterpret_cast<const f32x4*>(xl + kk * CP + ci);
            wr = wl[ci * K + min(lane, G * K - 1)];
        };
        // the weight of step n+2 is broadcast (v_readlane -> SGPR) in the shadow of the dependent fma of step n; the
        // scheduling barriers keep that interleave (left alone the compiler batches all broadcasts before the chain)
        auto chain = [&](const f32x4(&xq)[K], float wr) {
            const int wi = __builtin_bit_cast(int, wr);
            float wq[G * K];
            wq[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, 0));
            wq[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, 1));
#pragma unroll
            for (int n = 0; n < G * K; ++n) {
                if (n + 2 < G * K) wq[n + 2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(wi, n + 2));
                acc = __builtin_fmaf(wq[n], xq[n % K][n / K], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        ld(xa, wa, 0);
        for (int ci = 0; ci < Cin; ci += 2 * G) {
            ld(xb2, wb, ci + G);
            chain(xa, wa);
            if (ci + 2 * G < Cin) ld(xa, wa, ci + 2 * G);
            chain(xb2, wb);
        }
    }
    if (clamp_out) acc = acc > 1.0f ? 1.0f : (acc < -1.0f ? -1.0f : acc);
    if (valid) y[((long)b * Cout + co) * Lout + col] = acc;
}

// ConvTranspose1d (k = 2S): output u takes taps kk0 = (u + padL) % S (input t0) and kk0 + S (input t0 - 1), in that
// order.  Both the frames and the taps a lane needs depend on u, so weights are read from a tap-major tile
// ws[kk][CP] like the input: four 16-byte reads per four channels (eight steps).
template <int S, int NW>
__global__ __launch_bounds__(64 * NW) void convtr1d_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ wt,
                                                               const float* __restrict__ bias, float* __restrict__ y, int Cin,
                                                               int Lin, int Cout, int NC, int NTI, float slope, int pre) {
    extern __shared__ __attribute__((aligned(16))) float lds_sm[];
    constexpr int K = 2 * S;
    constexpr int padL = (K - S + 1) / 2;
    constexpr int NT = 64 * NW;
    const int CP = Cin + 4;
    float* xs = lds_sm;                      // [NTI][CP] input frames t_base .. t_base + NTI - 1
    float* ws = lds_sm + (long)NTI * CP;     // [NW][K][CP]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Lout = Lin * S;
    const int u0 = blockIdx.x * NC;
    const int co0 = blockIdx.y * NW;
    const int b = blockIdx.z;
    const int t_base = (u0 + padL) / S - 1;
    const float* xb = x + (long)b * Cin * Lin;
    const int nx = Cin * NTI;
#pragma unroll 8
    for (int e = tid; e < nx; e += NT) {   // consecutive threads -> consecutive channels: conflict-free LDS stores
        const int j = e / Cin, ci = e - j * Cin;
        const int t = t_base + j;
        float v = (t >= 0 && t < Lin) ? xb[(long)ci * Lin + t] : 0.0f;
        if (pre) v = lrelu(v, slope);
        xs[j * CP + ci] = v;
    }
    if (wt && (Cin & 3) == 0) {
        // tap-major copy w_lat[co][k][ci] (packed at load): the NW x K rows of this workgroup are one contiguous run in memory and
        // land in LDS as 16-byte stores (the [ci][co][k] original costs a 16-byte load per 64-byte line and four scalar LDS stores)
        const int c4 = Cin >> 2;
#pragma unroll 8
        for (int e = tid; e < NW * K * c4; e += NT) {
            const int row = e / c4, q = e - row * c4;      // row = wv * K + kk
            const int wv = row / K, kk = row - wv * K;
            const int co = min(co0 + wv, Cout - 1);
            *reinterpret_cast<f32x4*>(ws + (long)row * CP + 4 * q) = *reinterpret_cast<const f32x4*>(wt + ((long)co * K + kk) * Cin + 4 * q);
        }
    } else {
    constexpr int VW = (K % 4 == 0) ? 4 : 2;      // the K = 2S taps of one (ci, co) are contiguous: 16- or 8-byte loads
    constexpr int KV = K / VW;
    typedef float fvw __attribute__((ext_vector_type(VW)));
    const int nwv = Cin * KV;
#pragma unroll 8
    for (int e = tid; e < NW * nwv; e += NT) {   // consecutive threads -> consecutive channels of one tap group
        const int wv = e / nwv, r = e - wv * nwv;
        const int kq = (r / Cin) * VW, ci = r - (r / Cin) * Cin;
        const int co = min(co0 + wv, Cout - 1);
        const fvw v = *reinterpret_cast<const fvw*>(w + ((long)ci * Cout + co) * K + kq);
#pragma unroll
        for (int q = 0; q < VW; ++q) ws[((long)wv * K + kq + q) * CP + ci] = v[q];
    }
    }
    __syncthreads();
    const int co = co0 + wave;
    if (co >= Cout) return;
    const int u = u0 + lane;
    const bool valid = lane < NC && u < Lout;
    const int uc = valid ? u : u0;
    const int kk0 = (uc + padL) % S;
    const int j0 = (uc + padL - kk0) / S - t_base;   // >= 1
    const float* x0 = xs + (long)j0 * CP;
    const float* x1 = x0 - CP;
    const float* w0 = ws + ((long)wave * K + kk0) * CP;
    const float* w1 = w0 + (long)S * CP;
    float acc = bias[co];
    f32x4 a0 = *reinterpret_cast<const f32x4*>(x0), a1 = *reinterpret_cast<const f32x4*>(x1);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(w0), b1 = *reinterpret_cast<const f32x4*>(w1);
    for (int ci = 0; ci < Cin; ci += 4) {
        const f32x4 ca0 = a0, ca1 = a1, cb0 = b0, cb1 = b1;
        if (ci + 4 < Cin) {
            a0 = *reinterpret_cast<const f32x4*>(x0 + ci + 4); a1 = *reinterpret_cast<const f32x4*>(x1 + ci + 4);
            b0 = *reinterpret_cast<const f32x4*>(w0 + ci + 4); b1 = *reinterpret_cast<const f32x4*>(w1 + ci + 4);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc = __builtin_fmaf(cb0[g], ca0[g], acc);
            acc = __builtin_fmaf(cb1[g], ca1[g], acc);
        }
    }
    if (valid) y[((long)b * Cout + co) * Lout + u] = acc;
}

__global__ __launch_bounds__(256) void in_proj_kernel(const float* __restrict__ ze, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ z, int B,
                                                      int D, int F, int f0, int fc, int J, int ze_is_rows) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * fc * J) return;
    const int j = (int)(idx % J);
    const long row = idx / J;
    const int f = f0 + (int)(row % fc);
    const int b = (int)(row / fc);
    float acc = bias[j];
    if (ze_is_rows) {  // ze laid out [B*F][D] (public quantizer.inference entry)
        const float* zr = ze + ((long)b * F + f) * D;
        for (int d = 0; d < D; ++d) acc = __builtin_fmaf(w[j * D + d], zr[d], acc);
    } else {  // encoder-native [B][D][F]
        const float* zr = ze + (long)b * D * F + f;
        int d = 0;
        for (; d + 16 <= D; d += 16) {   // 16 strided loads in flight per step of the chain
            float zv[16], wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { zv[i] = zr[(long)(d + i) * F]; wv[i] = w[j * D + d + i]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_fmaf(wv[i], zv[i], acc);
        }
        for (; d < D; ++d) acc = __builtin_fmaf(w[j * D + d], zr[(long)d * F], acc);
    }
    z[idx] = acc;
}

// cb[c][j] = pb[j] + sum_r pw[j][r] * raw[c][r]
__global__ __launch_bounds__(256) void codebook_proj_kernel(const float* __restrict__ raw, const float* __restrict__ pw,
                                                            const float* __restrict__ pb, float* __restrict__ cb,
                                                            int N, int R, int J) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * J) return;
    const int j = (int)(idx % J);
    const long c = idx / J;
    float acc = pb[j];
    for (int r = 0; r < R; ++r) acc = __builtin_fmaf(pw[j * R + r], raw[c * R + r], acc);
    cb[idx] = acc;
}

// hc[c] = -0.5 * sum_j cb[c][j]^2 ; also the MFMA-packed codebook:
// cbp[tile][h][lane][4]: element e of half h = cb[tile*32 + (lane&31)][2*(4h+e) + (lane>>5)]
__global__ __launch_bounds__(256) void codebook_norm_pack_kernel(const float* __restrict__ cb, float* __restrict__ hc,
                                                                 float* __restrict__ cbp, int N, int J) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float a = 0.0f;
    for (int j = 0; j < J; ++j) a = __builtin_fmaf(cb[c * J + j], cb[c * J + j], a);
    hc[c] = -0.5f * a;
    if (cbp && J == 16) {
        const long tile = c >> 5;
        const int l31 = (int)(c & 31);
        for (int k = 0; k < 16; ++k) {
            const int kp = k >> 1, hi = k & 1;  // k = 2*kp + hi
            const int h = kp >> 2, e = kp & 3;
            const int lane = l31 + 32 * hi;
            cbp[((tile * 2 + h) * 64 + lane) * 4 + e] = cb[c * J + k];
        }
    }
}

__device__ __forceinline__ unsigned long long pack_key(float s, unsigned idx) {
    unsigned u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);  // ties -> lower idx wins
}

// chain variant: block = 256 threads, FB frames per block, thread strides over codes.
template <int FB>
__global__ __launch_bounds__(256) void vq_chain_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                       const float* __restrict__ hc, unsigned long long* __restrict__ keys,
                                                       long F, int N, int c_per_split) {
    const long f0 = (long)blockIdx.x * FB;
    const int cbeg = blockIdx.y * c_per_split;
    const int cend = min(N, cbeg + c_per_split);
    float zr[FB][16];
#pragma unroll
    for (int i = 0; i < FB; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) zr[i][j] = (f0 + i < F) ? z[(f0 + i) * 16 + j] : 0.0f;
    float best[FB];
    unsigned bidx[FB];
#pragma unroll
    for (int i = 0; i < FB; ++i) { best[i] = -INFINITY; bidx[i] = 0xFFFFFFFFu; }
    for (int c = cbeg + threadIdx.x; c < cend; c += 256) {
        const float4* cr = reinterpret_cast<const float4*>(cb + (long)c * 16);
        const float4 c0 = cr[0], c1 = cr[1], c2 = cr[2], c3 = cr[3];
        const float cv[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
        const float h = hc[c];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            float a = h;
#pragma unroll
            for (int j = 0; j < 16; ++j) a = __builtin_fmaf(zr[i][j], cv[j], a);
            if (a > best[i]) { best[i] = a; bidx[i] = (unsigned)c; }
        }
    }
    __shared__ unsigned long long red[FB][4];
#pragma unroll
    for (int i = 0; i < FB; ++i) {
        unsigned long long key = bidx[i] == 0xFFFFFFFFu ? 0ull : pack_key(best[i], bidx[i]);
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other > key ? other : key;
        }
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = key;
    }
    __syncthreads();
    if (threadIdx.x < FB && f0 + threadIdx.x < F) {
        unsigned long long key = red[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) key = red[threadIdx.x][w] > key ? red[threadIdx.x][w] : key;
        atomicMax(&keys[f0 + threadIdx.x], key);
    }
}

// MFMA variant: scores[c][f] = hc[c] + sum_k cb[c][k] * z[f][k] on v_mfma_f32_32x32x2_f32
// (A = codebook tile 32 codes x 2 k, B = z tile 2 k x 32 frames, C initialised to hc[c]).
// Workgroup = 4 waves sharing FN*32 frames; the block's code range is dealt to the waves in
// interleaved 32-code tiles.  Per lane the running best over its rows is kept in registers
// (strict '>' on ascending codes = lowest index on ties), packed into a monotone 64-bit key at the
// end and merged by wave shuffle -> LDS -> one global atomicMax per frame.
template <int FN>
__global__ __launch_bounds__(256) void vq_mfma_kernel(const float* __restrict__ z, const float* __restrict__ cbp,
                                                      const float* __restrict__ hc, unsigned long long* __restrict__ keys,
                                                      long F, int N, int tiles_per_split) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long f0 = (long)blockIdx.x * (FN * 32);
    const int tile_beg = blockIdx.y * tiles_per_split;
    const int tile_end = min(N / 32, tile_beg + tiles_per_split);
    const int half = lane >> 5;

    // B fragments: zb[ft][kp] = z[f0 + ft*32 + (lane&31)][2*kp + half]
    float zb[FN][8];
#pragma unroll
    for (int ft = 0; ft < FN; ++ft) {
        const long f = f0 + ft * 32 + (lane & 31);
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) zb[ft][kp] = f < F ? z[f * 16 + 2 * kp + half] : 0.0f;
    }
    float best[FN];
    unsigned bidx[FN];
#pragma unroll
    for (int ft = 0; ft < FN; ++ft) { best[ft] = -INFINITY; bidx[ft] = 0xFFFFFFFFu; }

    for (int tile = tile_beg + wave; tile < tile_end; tile += 4) {
        const float4* ap = reinterpret_cast<const float4*>(cbp) + ((long)tile * 2) * 64 + lane;
        const float4 a0 = ap[0], a1 = ap[64];
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const int c0 = tile * 32;
        f32x16 cinit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 h4 = *reinterpret_cast<const float4*>(hc + c0 + 8 * g + 4 * half);
            cinit[4 * g + 0] = h4.x; cinit[4 * g + 1] = h4.y; cinit[4 * g + 2] = h4.z; cinit[4 * g + 3] = h4.w;
        }
#pragma unroll
        for (int ft = 0; ft < FN; ++ft) {
            f32x16 acc = cinit;
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], zb[ft][kp], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned c = (unsigned)(c0 + (r & 3) + 8 * (r >> 2) + 4 * half);
                if (acc[r] > best[ft]) { best[ft] = acc[r]; bidx[ft] = c; }
            }
        }
    }
    __shared__ unsigned long long red[FN][4][32];
#pragma unroll
    for (int ft = 0; ft < FN; ++ft) {
        unsigned long long key = bidx[ft] == 0xFFFFFFFFu ? 0ull : pack_key(best[ft], bidx[ft]);
        const unsigned long long other = __shfl_xor(key, 32);
        key = other > key ? other : key;
        if (lane < 32) red[ft][wave][lane] = key;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FN * 32; i += 256) {
        const int ft = i >> 5, l = i & 31;
        const long f = f0 + i;
        if (f < F) {
            unsigned long long key = red[ft][0][l];
            for (int w = 1; w < 4; ++w) key = red[ft][w][l] > key ? red[ft][w][l] : key;
            atomicMax(&keys[f], key);
        }
    }
}

// keys -> int64 codes with the output mapping of RowDst; also re-arms keys for the next call
struct RowDst {
    int64_t* base;
    int C;            // rows are (win, chan): chan = row_b % C, win = row_b / C
    long chan_stride;
    long win_stride;
    int fc;           // frames kept per row
    const long* row_off = nullptr;   // optional per-row table (device): row b's codes go to base + row_off[b]
};
__global__ __launch_bounds__(256) void vq_finalize_kernel(unsigned long long* __restrict__ keys, RowDst dst, long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const unsigned long long key = keys[i];
    keys[i] = 0ull;
    const long b = i / dst.fc;
    const int j = (int)(i - b * dst.fc);
    dst.base[(dst.row_off ? dst.row_off[b] : (b % dst.C) * dst.chan_stride + (b / dst.C) * dst.win_stride) + j] =
        (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
}

// zq[b][j][f] = cb[codes[b][f]][j]
__global__ __launch_bounds__(256) void embed_codes_kernel(const int64_t* __restrict__ codes, long ldc, const float* __restrict__ cb,
                                                          float* __restrict__ zq, int B, int F, int J, int N,
                                                          int* __restrict__ err) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * J * F) return;
    const int f = (int)(idx % F);
    const long r = idx / F;
    const int j = (int)(r % J);
    const int b = (int)(r / J);
    int64_t c = codes[(long)b * ldc + f];
    if (c < 0 || c >= N) { atomicExch(err, 1); c = 0; }
    zq[idx] = cb[c * J + j];
}

// [B][F][J] -> [B][J][F]
__global__ __launch_bounds__(256) void transpose_fj_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                           int F, int J) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * J * F) return;
    const int f = (int)(idx % F);
    const long r = idx / F;
    const int j = (int)(r % J);
    const int b = (int)(r / J);
    out[idx] = in[((long)b * F + f) * J + j];
}
// [B][D][F] -> [B][F][D]
__global__ __launch_bounds__(256) void transpose_df_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                           int D, int F) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * D * F) return;
    const int d = (int)(idx % D);
    const long r = idx / D;
    const int f = (int)(r % F);
    const int b = (int)(r / F);
    out[idx] = in[((long)b * D + d) * F + f];
}

// =============================================================================================
struct ConvLayer {
    int cin, cout, k, s, pre, tr;
    float* w = nullptr;     // original layout
    float* b = nullptr;
    float* bp = nullptr;    // bias padded with zeros to whole channel tiles (16-byte loads in the MFMA kernel)
    float* wp = nullptr;    // MFMA-packed (encoder non-transposed layers with cin*k >= 32)
    int K = 0, Kpad = 0, cout_pad = 0, nchunks = 0;
    float* wp2 = nullptr;   // packing for the warp-specialised kernel (larger K chunks)
    int nchunks2 = 0;
    float* wp_tr = nullptr; // transposed conv: per-phase packing [phase][co_tile][kquad][lane][4]
    float* w_lat = nullptr; // transposed conv, latency kernel: tap-major rows [co][k][ci] (a workgroup's weights are one contiguous run)
    int nchunks_tr = 0;
};

// input channels per K chunk of the MFMA conv, by (kernel size, stride)
#ifdef RCA_CONV_SMALLCHUNK   // experiment: 16-k chunks (half the weight / staging registers) so that 64 x 64 tiles fit three waves per SIMD
#define RCA_CIC_K8 2
#define RCA_CIC_K16 1
#else
#define RCA_CIC_K8 4
#define RCA_CIC_K16 2
#endif
#ifndef RCA_CIC_K3
#define RCA_CIC_K3 8
#endif
static int conv_cic(int k, int s) {
    if (k == 4 && s == 2) return 4;
    if (k == 8 && s == 4) return RCA_CIC_K8;
    if (k == 10 && s == 5) return 4;
    if (k == 16 && s == 8) return RCA_CIC_K16;
    if (k == 3 && s == 1) return RCA_CIC_K3;
    return 0;
}
// ... and of the warp-specialised kernel (32 k-pairs = 128 MFMAs per consumer between barriers)
static int conv_cic_ws(int k, int s) {
    if (k == 4 && s == 2) return 16;
    if (k == 8 && s == 4) return 8;
    if (k == 10 && s == 5) return 8;
    if (k == 16 && s == 8) return 4;
    if (k == 3 && s == 1) return 16;
    return 0;
}

// =====================================================================================================================
// bf16 "blocked" encoder pipeline (opt-in: rca_codec_set_mfma_mode 1 / 3, never the default and never the bench's `value`).
//
// Round 3's bf16 modes ran conv1d_mfma_kernel with its f32 data path: f32 activations in HBM, 4-byte LDS reads, one conversion per
// operand per use -- 1.84 ms per 256-window step against an HBM floor of ~0.3 ms.  This pipeline is built for the bf16 matrix
// instruction instead (the reference's own GPU arithmetic class is bf16 autocast, audio_tokenizer.py:24,78-82):
//   * activations live in HBM as bf16 in CHANNEL-BLOCKED layout  X[b][c / 16][t][c % 16]  (32 bytes per time step and block): the 16
//     input channels of one MFMA k step are contiguous, so a B operand (lane <-> output column, 8 channels of one tap) is ONE
//     16-byte LDS read and staging is a straight 16-byte copy;
//   * weights are rounded (mode 1) or split into hi + lo (mode 3) ONCE at pack time, in A-fragment order
//     wp[co / 32][ci / 16][tap][lane][8]: one 16-byte load per lane and MFMA;
//   * the LDS window is phase-de-interleaved like the f32 kernel's:  xs[plane][half][t % S][t / S]  of 16-byte cells, so the 16 lanes
//     of a ds_read_b128 group read 16 consecutive cells for every tap (conflict-free);
//   * a workgroup of 4 waves owns 128 channels x 128 columns (64 x 256 for the 64-channel layer) of ONE batch row, each wave 32 x 128
//     (one weight fragment feeds four MFMAs, no two waves load the same fragment);
//     per stage (CPS blocks of 16 input channels) the next stage's window is loaded into registers before the MFMA block and written
//     to the other LDS buffer behind it: one barrier per stage;
//   * mode 3 keeps a second (lo) plane of every activation and weight and issues hi hi + hi lo + lo hi, product-major (2.43 -> 2.18 ms
//     per step against conv1d_mfma_kernel<BF = 3>, which RCA_BF16_BLK_SPLIT=0 brings back for A/B runs).
// k order inside an MFMA differs from the f32 definition (tap-major over a 16-channel block), like every bf16 mode: not bit-exact,
// the fraction of equal code ids is measured by the bench and the test.
#ifndef RCA_BLK_ABL
#define RCA_BLK_ABL 0
#endif
typedef unsigned short conv_bf16raw;
typedef __attribute__((ext_vector_type(4))) unsigned conv_u32x4;

template <int KS, int SPLIT>
__global__ __launch_bounds__(256) void conv_in_blk_kernel(RowSrc src, const float* __restrict__ w, const float* __restrict__ bias,
                                                          conv_bf16raw* __restrict__ y_hi, conv_bf16raw* __restrict__ y_lo, int B, int Cout, int L,
                                                          int act, float slope) {
    // conv_in (Cin = 1) -> blocked bf16; the f32 chain is conv_in_kernel's (bias first, taps ascending, out-of-range taps skipped).
    // `w` is the PADDED table [co][8] = 7 taps + the bias (built once at pack time): wave-uniform 32-byte rows, one s_load_dwordx8
    // per channel through the scalar cache.  (224 separate scalar loads with a test per tap: 410 us; the table in LDS read by
    // broadcast: 127 us, bound by the LDS instruction rate; the memory floor of this launch is ~50 us.)
    static_assert(KS <= 7, "taps + bias fit 8 floats");
    (void)bias;
    const int b = blockIdx.y;                                  // grid (L / 256, B): no 64-bit division per thread
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    const float* xr = src.base + src.off(b);
    constexpr int padL = KS / 2;
    float xv[KS];
    bool interior = true;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int i = t + kk - padL;
        const bool ok = i >= 0 && i < src.T;
        interior = interior && ok;
        xv[kk] = ok ? xr[i] : 0.0f;
    }
    const int Cb = Cout / 16;
#pragma unroll 2
    for (int cb = 0; cb < Cb; ++cb) {
        unsigned hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int co = cb * 16 + 2 * j + e;
                const float* wr = w + co * 8;
                const float wk[8] = {wr[0], wr[1], wr[2], wr[3], wr[4], wr[5], wr[6], wr[7]};
                float acc = wk[7];
                if (interior) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) acc = __builtin_fmaf(wk[kk], xv[kk], acc);
                } else {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        const int i = t + kk - padL;
                        if (i >= 0 && i < src.T) acc = __builtin_fmaf(wk[kk], xv[kk], acc);
                    }
                }
                v[e] = act ? fmaxf(acc, acc * slope) : acc;
            }
            hi[j] = conv_pack_bf16x2(v[0], v[1]);
            if (SPLIT) lo[j] = conv_pack_bf16x2(v[0] - __uint_as_float(hi[j] << 16), v[1] - __uint_as_float(hi[j] & 0xffff0000u));
        }
        const long o = (((long)b * Cb + cb) * L + t) * 16;
        *reinterpret_cast<conv_u32x4*>(y_hi + o) = conv_u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<conv_u32x4*>(y_hi + o + 8) = conv_u32x4{hi[4], hi[5], hi[6], hi[7]};
        if (SPLIT) {
            *reinterpret_cast<conv_u32x4*>(y_lo + o) = conv_u32x4{lo[0], lo[1], lo[2], lo[3]};
            *reinterpret_cast<conv_u32x4*>(y_lo + o + 8) = conv_u32x4{lo[4], lo[5], lo[6], lo[7]};
        }
    }
}

// OUT = 0: y blocked bf16 (hi [+ lo]), LeakyReLU applied when `act` (the consumer is pre-activated); OUT = 1: y f32 [b][co][t].
// A wave owns 32 channels x 128 columns (1 x 4 MFMA tiles): one weight fragment (16 bytes per lane, straight from L2 into registers,
// PD taps ahead) feeds four MFMAs and no two waves of a workgroup load the same fragment; the workgroup is WGM waves along the channels
// x 4 / WGM along the columns.
// FUSE = 1 (the first strided layer): the input is not read from HBM but computed while staging as conv_in(PCM) -- conv_in_blk_kernel's
// f32 chain per (sample, channel), LeakyReLU, bf16 -- so conv_in's 524 MB write and this layer's 524 MB read per 256-window step
// disappear (a third of all bytes the blocked pipeline moved).
// WMT = 2: a wave owns 64 channels x 128 columns (2 x 4 MFMA tiles): every B fragment read from LDS feeds two MFMAs (with 1 x 4 tiles
// the MFMA phase of the wide layers ran against the LDS read rate: 1 KB per MFMA and wave).
template <int KS, int S, int WGM, int CPS, int SPLIT, int OUT, int FUSE = 0, int WMT = 1>
__global__ __launch_bounds__(256, (SPLIT || KS == 3 || WMT == 2) ? 2 : 3) void conv_bf16_blk_kernel(const conv_bf16raw* __restrict__ x_hi, const conv_bf16raw* __restrict__ x_lo,
                                                               const conv_bf16raw* __restrict__ w_hi, const conv_bf16raw* __restrict__ w_lo,
                                                               const float* __restrict__ bias, conv_bf16raw* __restrict__ y_hi,
                                                               conv_bf16raw* __restrict__ y_lo, float* __restrict__ y_f32, int Cin, int Lin, int Cout,
                                                               int Lout, int act, float slope, int n_rows, int n_ct, int n_cot, RowSrc fsrc,
                                                               const float* __restrict__ w_in8, int in_act) {
    constexpr int WGN = 4 / WGM;
    constexpr int NT = 128 * WGN;                         // columns per workgroup
    // hi + lo planes of the wide-stride layers: ONE LDS buffer (two barriers per stage) so that two workgroups fit a CU -- double
    // buffered they need 82 / 132 KB and ran one workgroup = one wave per SIMD (k16s8 764 us against 148 us rounded)
    constexpr bool SB = SPLIT && S >= 5;
    constexpr int padL = (KS - S + 1) / 2;
    constexpr int WIN = (NT - 1) * S + KS;                // input samples a column tile sees
    constexpr int SLOTS = NT + (KS - 1) / S;              // 16-byte cells per (plane, half, phase) row
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int CELLS = 2 * S * SLOTS;                  // cells of one block's window (one plane)
    constexpr int PIECES = 2 * WIN;                       // 16-byte pieces of one block's window (one plane)
    constexpr int NP = (PIECES + 255) / 256;              // per thread
    // weight fragments in flight (taps ahead).  The k = 3 layer keeps a whole stage's fragments in flight (every request of a stage is
    // then for the NEXT stage and the window loads go out at the top: 44 -> 34 us); on the k = 10 / k = 16 layers the same ring
    // costs the third wave per SIMD and measured slower (169 vs 161 us, 154 vs 147 us)
    constexpr int PD = WMT == 2 ? ((CPS * KS) % 4 == 0 ? 4 : 5)          // two channel tiles: twice the registers per tap
                                : (KS == 3 && !SPLIT) ? CPS * KS : (KS < 8 ? KS : ((CPS * KS) % 8 == 0 ? (SPLIT ? 4 : 8) : 5));
    static_assert((CPS * KS) % PD == 0, "the fragment ring keeps static slots across stages");
    extern __shared__ __attribute__((aligned(16))) conv_u32x4 blk_lds[];   // [2 buffers][CPS][NPL][CELLS]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm_w = wave / WGN, wn_w = wave % WGN;       // this wave's 32 x 128 tile inside the workgroup tile
    // Workgroup id -> (batch row, column tile, channel tile).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2:
    // the channel tiles of one (row, column tile) run back to back on ONE XCD, so the input window is fetched once (speed only; any
    // placement is correct).  (With the channel tile on blockIdx.y the k16s8 layer fetched its input 4 x, k10s5 2 x.)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int by = seq % n_cot;
    const int ct = (seq / n_cot) * 8 + xcd;               // (row, column tile) index; the grid is padded to a multiple of 8 of them
    if (ct >= n_ct * n_rows) return;
    const int b = ct / n_ct;
    const int n0 = (ct - b * n_ct) * NT;                  // first output column of the workgroup
    const int co0 = (by * WGM + wm_w) * 32 * WMT;         // first output channel of the wave
    const int Cbi = Cin / 16, nstages = Cbi / CPS;
    const int t_start = n0 * S - padL;
    // ---- staging role (fixed per thread): piece -> (sample, half) -> LDS cell; global offset in 16-byte units from the block's row.
    // Loads are issued unconditionally from a clamped offset and zeroed at the LDS write: a select or a branch in front of a load makes
    // the compiler wait for each one separately.
    int g_off[NP], l_off[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = tid + 256 * i;
        const int tr = piece >> 1, h = piece & 1;
        const int t = t_start + tr;
        const bool ok = piece < PIECES && t >= 0 && t < Lin;
        g_off[i] = ok ? t * 2 + h : -1;                    // -1: outside the signal (zeros)
        l_off[i] = piece < PIECES ? (h * S + tr % S) * SLOTS + tr / S : -1;
    }
    conv_u32x4 sreg[CPS][NPL][NP];
    auto stage_load = [&](int st) {
#pragma unroll
        for (int cb = 0; cb < CPS; ++cb) {
            const long row = ((long)b * Cbi + (st * CPS + cb)) * Lin * 2;    // 16-byte units
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const conv_u32x4* src = reinterpret_cast<const conv_u32x4*>(pl ? x_lo : x_hi) + row;
#pragma unroll
                for (int i = 0; i < NP; ++i) sreg[cb][pl][i] = src[g_off[i] >= 0 ? g_off[i] : 0];
            }
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int cb = 0; cb < CPS; ++cb)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                conv_u32x4* dst = blk_lds + ((buf * CPS + cb) * NPL + pl) * CELLS;
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if (l_off[i] >= 0) dst[l_off[i]] = g_off[i] >= 0 ? sreg[cb][pl][i] : conv_u32x4{0u, 0u, 0u, 0u};
            }
    };
    // ---- accumulators start at the bias
    f32x16 acc[WMT][4];
#pragma unroll
    for (int wm = 0; wm < WMT; ++wm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bias[co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
            for (int wn = 0; wn < 4; ++wn) acc[wm][wn][r] = bv;
        }
    // weights: fragment (co tile, block, tap) = 64 lanes x 16 bytes; a channel tile's fragments are consecutive over (block, tap)
    const long wtile = (long)Cbi * KS * 64;                  // 16-byte units per channel tile
    const conv_u32x4* wq_hi = reinterpret_cast<const conv_u32x4*>(w_hi) + (long)(co0 / 32) * wtile + lane;
    const conv_u32x4* wq_lo = reinterpret_cast<const conv_u32x4*>(w_lo) + (long)(co0 / 32) * wtile + lane;
    const int bcell0 = half * S * SLOTS + wn_w * 128 + n;   // + wn * 32 + (kk % S) * SLOTS + kk / S
    const int nfrag = Cbi * KS;                              // fragments of this wave, in the order they are consumed
    conv_u32x4 ah[PD][WMT], al[SPLIT ? PD : 1][WMT];
#pragma unroll
    for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int wm = 0; wm < WMT; ++wm) {
            ah[i][wm] = wq_hi[wm * wtile + (long)i * 64];
            if (SPLIT) al[i][wm] = wq_lo[wm * wtile + (long)i * 64];
        }
    if constexpr (FUSE) {
        // conv_in for the workgroup's window: a thread takes samples tid, tid + 256, ... and every channel of them.  All PCM taps are
        // requested first; the channel loop is outermost, so a channel's 8 weights (wave-uniform: one s_load_dwordx8) serve every
        // sample of the thread and the samples' fma chains are independent.  Taps outside the signal hold 0 (fma(w, 0, a) = a; the
        // f32 kernels skip them to keep the sign of a zero sum -- this arithmetic is not bit-exact anyway).
        static_assert(!FUSE || (CPS * 16 <= 64), "the fused first layer stages all its input channels at once");
        constexpr int NS = (WIN + 255) / 256;
        const float* xr = fsrc.base + fsrc.off(b);
        float xv[NS][7];
        bool exists[NS];
        int cell[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int tr = tid + 256 * q;
            const int t = t_start + tr;
            exists[q] = tr < WIN && t >= 0 && t < Lin;       // outside the layer's input: zero padding
            cell[q] = tr < WIN ? (tr % S) * SLOTS + tr / S : -1;
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const int i = t + kk - 3;
                xv[q][kk] = (tr < WIN && i >= 0 && i < fsrc.T) ? xr[i] : 0.0f;
            }
        }
#pragma unroll
        for (int cb = 0; cb < CPS; ++cb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned hi[NS][4], lo[NS][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[NS][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float* wr = w_in8 + (cb * 16 + 8 * h + 2 * j + e) * 8;      // wave-uniform: scalar loads
                        const float w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3], w4 = wr[4], w5 = wr[5], w6 = wr[6], bi = wr[7];
#pragma unroll
                        for (int q = 0; q < NS; ++q) {
                            float a = bi;
#if RCA_BLK_ABL == 1   // timing experiment: no conv_in arithmetic
                            v[q][e] = xv[q][e]; continue;
#endif
                            a = __builtin_fmaf(w0, xv[q][0], a); a = __builtin_fmaf(w1, xv[q][1], a); a = __builtin_fmaf(w2, xv[q][2], a);
                            a = __builtin_fmaf(w3, xv[q][3], a); a = __builtin_fmaf(w4, xv[q][4], a); a = __builtin_fmaf(w5, xv[q][5], a);
                            a = __builtin_fmaf(w6, xv[q][6], a);
                            a = in_act ? fmaxf(a, a * slope) : a;
                            v[q][e] = exists[q] ? a : 0.0f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        hi[q][j] = conv_pack_bf16x2(v[q][0], v[q][1]);
                        if (SPLIT) lo[q][j] = conv_pack_bf16x2(v[q][0] - __uint_as_float(hi[q][j] << 16), v[q][1] - __uint_as_float(hi[q][j] & 0xffff0000u));
                    }
                }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (cell[q] >= 0) {
                        blk_lds[(cb * NPL) * CELLS + h * S * SLOTS + cell[q]] = conv_u32x4{hi[q][0], hi[q][1], hi[q][2], hi[q][3]};
                        if (SPLIT) blk_lds[(cb * NPL + 1) * CELLS + h * S * SLOTS + cell[q]] = conv_u32x4{lo[q][0], lo[q][1], lo[q][2], lo[q][3]};
                    }
            }
    } else {
        stage_load(0);
        stage_write(0);
    }
    __syncthreads();
    int f = 0;                                               // next fragment to consume
    for (int st = 0; st < nstages; ++st) {
        const int buf = SB ? 0 : (st & 1);
        // Vector-memory results return in order: a weight fragment requested BEHIND the next stage's window loads (HBM latency) cannot
        // be used before they have landed.  The window loads are therefore issued in the middle of the stage, right behind the last
        // fragment request this stage itself consumes (fragment i asks for i + PD): everything queued behind them belongs to the
        // next stage, whose LDS write waits for them anyway.
        constexpr int LOAD_AT = CPS * KS - PD;
#pragma unroll
        for (int cb = 0; cb < CPS; ++cb) {
            const conv_u32x4* xb = blk_lds + ((buf * CPS + cb) * NPL) * CELLS;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (cb * KS + kk == LOAD_AT && st + 1 < nstages) stage_load(st + 1);
                const int slot = (cb * KS + kk) % PD;        // static after unrolling (PD divides the stage's fragments)
                const int cell = bcell0 + (kk % S) * SLOTS + kk / S;
                conv_u32x4 bh[4], bl[4];
#pragma unroll
                for (int wn = 0; wn < 4; ++wn) {
                    bh[wn] = xb[cell + wn * 32];
                    if (SPLIT) bl[wn] = xb[CELLS + cell + wn * 32];
                }
                conv_u32x4 a_h[WMT], a_l[WMT];
#pragma unroll
                for (int wm = 0; wm < WMT; ++wm) { a_h[wm] = ah[slot][wm]; if (SPLIT) a_l[wm] = al[slot][wm]; }
                {   // refill the slot with the fragment PD taps ahead (past the end: the last fragment again, never used)
                    const long fn = (long)min(f + PD, nfrag - 1) * 64;
#pragma unroll
                    for (int wm = 0; wm < WMT; ++wm) {
                        ah[slot][wm] = wq_hi[wm * wtile + fn];
                        if (SPLIT) al[slot][wm] = wq_lo[wm * wtile + fn];
                    }
                }
                ++f;
#if RCA_BLK_ABL == 2   // timing experiment: no MFMAs (operands kept alive)
                asm volatile("" ::"v"(a_h[0]), "v"(bh[0]), "v"(bh[1]), "v"(bh[2]), "v"(bh[3]));
                continue;
#endif
                // product-major: the column / channel tiles' accumulators are independent, the three products of one accumulator are not
#pragma unroll
                for (int wm = 0; wm < WMT; ++wm)
#pragma unroll
                    for (int wn = 0; wn < 4; ++wn)
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(conv_bf16x8, a_h[wm]), __builtin_bit_cast(conv_bf16x8, bh[wn]), acc[wm][wn], 0, 0, 0);
                if (SPLIT) {
#pragma unroll
                    for (int wm = 0; wm < WMT; ++wm)
#pragma unroll
                        for (int wn = 0; wn < 4; ++wn)
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(conv_bf16x8, a_h[wm]), __builtin_bit_cast(conv_bf16x8, bl[wn]), acc[wm][wn], 0, 0, 0);
#pragma unroll
                    for (int wm = 0; wm < WMT; ++wm)
#pragma unroll
                        for (int wn = 0; wn < 4; ++wn)
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(conv_bf16x8, a_l[wm]), __builtin_bit_cast(conv_bf16x8, bh[wn]), acc[wm][wn], 0, 0, 0);
                }
            }
        }
        if (SB) __syncthreads();                           // every wave is done reading the only buffer
        if (st + 1 < nstages) stage_write(SB ? 0 : (buf ^ 1));
        __syncthreads();
    }
    // ---- epilogue: lane <-> column, register r <-> channel (r & 3) + 8 (r >> 2) + 4 half of the 32-row tile
#if RCA_BLK_ABL == 3   // timing experiment: one store per wave instead of the tile
    if (lane == 0 && acc[0][0][0] + acc[0][1][1] + acc[WMT - 1][2][2] + acc[WMT - 1][3][3] == 12345.0f) y_hi[0] = 1;
    return;
#endif
#pragma unroll
    for (int wn = 0; wn < 4; ++wn)
#pragma unroll
    for (int wm = 0; wm < WMT; ++wm) {
        const int t = n0 + wn_w * 128 + wn * 32 + n;
        if (t >= Lout) continue;
        if (OUT == 1) {
            float* yr = y_f32 + ((long)b * Cout + co0 + wm * 32 + 4 * half) * Lout + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) yr[(long)((r & 3) + 8 * (r >> 2)) * Lout] = acc[wm][wn][r];
        } else {
            const int Cbo = Cout / 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {          // registers 4 g .. 4 g + 3: channels 8 g + 4 half .. + 3 of the tile
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[wm][wn][4 * g + j];
                    v[j] = act ? fmaxf(a, a * slope) : a;
                }
                const int ch = co0 + wm * 32 + 8 * g + 4 * half;
                const long o = (((long)b * Cbo + ch / 16) * Lout + t) * 16 + (ch & 15);
                const unsigned h0 = conv_pack_bf16x2(v[0], v[1]), h1 = conv_pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(y_hi + o) = make_uint2(h0, h1);
                if (SPLIT) {
                    const unsigned l0 = conv_pack_bf16x2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                    const unsigned l1 = conv_pack_bf16x2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                    *reinterpret_cast<uint2*>(y_lo + o) = make_uint2(l0, l1);
                }
            }
        }
    }
}

struct Bf16Pack { conv_bf16raw* hi = nullptr; conv_bf16raw* lo = nullptr; float* w_in8 = nullptr; };   // w_in8: conv_in's [co][7 taps + bias]

struct rca_codec {
    rca_codec_config_t cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    int hop = 1;
    int variant = 1;
    int mfma_mode = 0;          // 0: f32 matrix instruction (bit-exact, default); 1 / 3: bf16 matrix instruction, rounded / hi + lo split (opt-in)
    bool lat_mode = false;      // streaming tail: LDS-staged scalar-chain kernels (set per call)
    // receptive-field margins (whole frames) left of a kept frame / sample, derived from the layer geometry
    int enc_left_frames = 0, dec_left_frames = 0;
    bool window_trim = false;   // batch windows: encode only what the kept frames can see (same codes)
    std::vector<ConvLayer> enc, dec;
    std::vector<Bf16Pack> bf16_packs;   // per encoder layer: weights of the blocked bf16 pipeline (built on first use)
    bool bf16_blk_split = true;         // mode 3 on the blocked pipeline too (RCA_BF16_BLK_SPLIT=0: the round-3 kernel, for A/B runs)
    float *q_in_w = nullptr, *q_in_b = nullptr;
    float *cb = nullptr, *hc = nullptr, *cbp = nullptr;
    DevBuf act[2], zbuf, keys, io_a, io_b, tail;
    // streaming host calls: pinned staging + one captured graph per call shape (see rca_codec_encode_tail)
    void* pin = nullptr; size_t pin_cap = 0;
    struct StreamGraph { hipGraphExec_t exec = nullptr; int seen = 0; unsigned long long sig = 0; };
    std::map<std::array<int, 5>, StreamGraph> sgraphs;
    bool stream_graphs = true;
    int* err_flag = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t xev = nullptr;
    // bench profiling (rca_codec_profile): event pairs around profiled launches
    struct Prof { hipEvent_t a, b; int kclass; double flops, bytes; };
    bool profile = false;
    std::vector<Prof> prof;       // recorded this interval
    std::vector<Prof> prof_pool;  // reusable events
};

struct ProfScope {
    rca_codec* h; hipStream_t st; rca_codec::Prof p; bool on;
    ProfScope(rca_codec* h_, hipStream_t st_, int kclass, double flops, double bytes) : h(h_), st(st_), on(h_->profile) {
        if (!on) return;
        if (!h->prof_pool.empty()) { p = h->prof_pool.back(); h->prof_pool.pop_back(); }
        else { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
        p.kclass = kclass; p.flops = flops; p.bytes = bytes;
        (void)hipEventRecord(p.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(p.b, st);
        h->prof.push_back(p);
    }
};

static int upload(const rca_tensor_t* ts, int n, const std::string& name, long numel, float** out) {
    const rca_tensor_t* t = find_tensor(ts, n, name);
    if (!t) return fail(RCA_ERR_MISSING, "tensor '%s' missing", name.c_str());
    if (t->numel != numel) return fail(RCA_ERR_ARG, "tensor '%s': numel %ld, expected %ld", name.c_str(), (long)t->numel, numel);
    if (t->dtype != RCA_F32) return fail(RCA_ERR_ARG, "tensor '%s' must be f32", name.c_str());
    RCA_HIP(hipMalloc((void**)out, (size_t)numel * sizeof(float)));
    RCA_HIP(hipMemcpy(*out, t->data, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
    return RCA_OK;
}

// pack W[cout][K] into MFMA A-fragment order, zero padded to cout_pad x (nchunks*cic*k)
static int pack_weights_cic(const float* w_host, const ConvLayer& L, int cic, float** out, int* nchunks_out) {
    const int K = L.cin * L.k;
    const int nchunks = (L.cin + cic - 1) / cic;
    const int Kpad = nchunks * cic * L.k;
    const int cout_pad = (L.cout + 127) / 128 * 128;
    const long n = (long)(cout_pad / 32) * (Kpad / 8) * 64 * 4;
    std::vector<float> p((size_t)n, 0.0f);
    for (int cot = 0; cot < cout_pad / 32; ++cot)
        for (int q = 0; q < Kpad / 8; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int co = cot * 32 + (lane & 31);
                    const int k = 2 * (4 * q + e) + (lane >> 5);
                    float v = 0.0f;
                    if (co < L.cout && k < K) v = w_host[(long)co * K + k];
                    p[(((long)cot * (Kpad / 8) + q) * 64 + lane) * 4 + e] = v;
                }
    RCA_HIP(hipMalloc((void**)out, (size_t)n * sizeof(float)));
    RCA_HIP(hipMemcpy(*out, p.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    *nchunks_out = nchunks;
    return RCA_OK;
}
static int pack_weights(const float* w_host, ConvLayer& L) {
    L.K = L.cin * L.k;
    L.cout_pad = (L.cout + 127) / 128 * 128;
    int rc = pack_weights_cic(w_host, L, conv_cic(L.k, L.s), &L.wp, &L.nchunks);
    if (rc != RCA_OK) return rc;
    L.Kpad = L.nchunks * conv_cic(L.k, L.s) * L.k;
    return pack_weights_cic(w_host, L, conv_cic_ws(L.k, L.s), &L.wp2, &L.nchunks2);
}

// ConvTranspose1d weights w[Cin][Cout][2s] -> per phase r the GEMM matrix Wr[co][2*ci + j] = w[ci][co][r + j*s]
static int pack_weights_tr(const float* w_host, ConvLayer& L) {
    const int cic = 16, K = 2 * L.cin;
    const int nchunks = (L.cin + cic - 1) / cic;
    const int Kpad = nchunks * cic * 2;
    L.cout_pad = (L.cout + 127) / 128 * 128;
    const long per_phase = (long)(L.cout_pad / 32) * (Kpad / 8) * 64 * 4;
    std::vector<float> p((size_t)per_phase * L.s, 0.0f);
    for (int r = 0; r < L.s; ++r)
        for (int cot = 0; cot < L.cout_pad / 32; ++cot)
            for (int q = 0; q < Kpad / 8; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int co = cot * 32 + (lane & 31);
                        const int k = 2 * (4 * q + e) + (lane >> 5);
                        float v = 0.0f;
                        if (co < L.cout && k < K) {
                            const int ci = k >> 1, j = k & 1;
                            v = w_host[((long)ci * L.cout + co) * L.k + r + j * L.s];
                        }
                        p[r * per_phase + (((long)cot * (Kpad / 8) + q) * 64 + lane) * 4 + e] = v;
                    }
    RCA_HIP(hipMalloc((void**)&L.wp_tr, p.size() * sizeof(float)));
    RCA_HIP(hipMemcpy(L.wp_tr, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
    {   // w[ci][co][k] -> w_lat[co][k][ci]: convtr1d_lds_kernel reads the rows of its output channels with contiguous 16-byte loads
        std::vector<float> t((size_t)L.cin * L.cout * L.k);
        for (int ci = 0; ci < L.cin; ++ci)
            for (int co = 0; co < L.cout; ++co)
                for (int kk = 0; kk < L.k; ++kk) t[((size_t)co * L.k + kk) * L.cin + ci] = w_host[((size_t)ci * L.cout + co) * L.k + kk];
        RCA_HIP(hipMalloc((void**)&L.w_lat, t.size() * sizeof(float)));
        RCA_HIP(hipMemcpy(L.w_lat, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    L.nchunks_tr = nchunks;
    return RCA_OK;
}

static bool mfma_supported(int k, int s) {
    return (k == 4 && s == 2) || (k == 8 && s == 4) || (k == 10 && s == 5) || (k == 16 && s == 8) || (k == 3 && s == 1);
}

extern "C" const char* rca_last_error(void) { return g_err; }
extern "C" const char* rca_version(void) { return "rca-hip 0.1 (gfx950)"; }
extern "C" int rca_device_sync(int32_t device) {
    RCA_HIP(hipSetDevice(device));
    RCA_HIP(hipDeviceSynchronize());
    return RCA_OK;
}

extern "C" int rca_device_count(int* n) {
    if (!n) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipGetDeviceCount(n));
    return RCA_OK;
}

extern "C" int rca_codec_destroy(rca_codec_t* h) {
#ifdef RCA_CONV_TIMELINE
    if (h && getenv("RCA_CONV_TIMELINE_OUT")) {   // dump the records for scripts/conv_timeline.py
        long* b = nullptr;
        unsigned n = 6 << 16;
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(&b, HIP_SYMBOL(rca_prof_buf), 8);
        if (b) {
            std::vector<long> host((size_t)(6 << 16) * 8);
            (void)hipMemcpy(host.data(), b, host.size() * 8, hipMemcpyDeviceToHost);
            if (FILE* f = fopen(getenv("RCA_CONV_TIMELINE_OUT"), "wb")) { fwrite(&n, 4, 1, f); fwrite(host.data(), 8, host.size(), f); fclose(f); }
        }
    }
#endif

    if (!h) return RCA_OK;
    (void)hipSetDevice(h->device);
    for (auto& pk : h->bf16_packs) {
        if (pk.hi) (void)hipFree(pk.hi);
        if (pk.lo) (void)hipFree(pk.lo);
        if (pk.w_in8) (void)hipFree(pk.w_in8);
    }
    for (auto* v : {&h->enc, &h->dec})
        for (auto& L : *v) {
            if (L.w) (void)hipFree(L.w);
            if (L.b) (void)hipFree(L.b);
            if (L.bp) (void)hipFree(L.bp);
            if (L.wp) (void)hipFree(L.wp);
            if (L.wp2) (void)hipFree(L.wp2);
            if (L.wp_tr) (void)hipFree(L.wp_tr);
            if (L.w_lat) (void)hipFree(L.w_lat);
        }
    for (float* p : {h->q_in_w, h->q_in_b, h->cb, h->hc, h->cbp})
        if (p) (void)hipFree(p);
    if (h->err_flag) (void)hipFree(h->err_flag);
    for (auto& kv : h->sgraphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (h->pin) (void)hipHostFree(h->pin);
    h->act[0].release(); h->act[1].release(); h->zbuf.release(); h->keys.release(); h->io_a.release(); h->io_b.release(); h->tail.release();
    for (auto* v : {&h->prof, &h->prof_pool})
        for (auto& p : *v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (h->xev) (void)hipEventDestroy(h->xev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RCA_OK;
}

// How far to the left a kept output can see, in whole frames.
//  encoder: output t of a conv (k, s, padL = (k-s+1)/2) reads inputs [t*s - padL, t*s + k-1 - padL]; chaining the
//           lower bound from the last layer down to the PCM gives frame f -> sample f*hop - reach.
//  decoder: output u of a transposed conv reads inputs t >= ceil((u + padL - (k-1)) / s); a plain conv reads
//           t >= u - padL; chaining from the PCM up to the code sequence gives sample f*hop -> frame f - reach
//           (the bound is hop-periodic, so the frame boundary is the worst case).
static void codec_receptive_field(const rca_codec_config_t& c, int hop, int* enc_left, int* dec_left) {
    const int n = c.n_stages;
    long lo = 0;
    lo = lo - (c.k_latent - 1 + 1) / 2;                                        // conv_out (s = 1)
    for (int i = n - 1; i >= 0; --i) { const int s = c.strides[i], k = 2 * s; lo = lo * s - (k - s + 1) / 2; }
    lo = lo - (c.k_in - 1 + 1) / 2;                                            // conv_in (s = 1)
    *enc_left = (int)((-lo + hop - 1) / hop);
    auto ceil_div = [](long a, long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); };
    long u = 0;
    u = u - (c.k_in - 1 + 1) / 2;                                              // dec.conv_out (s = 1)
    for (int i = 0; i < n; ++i) { const int s = c.strides[i], k = 2 * s; u = ceil_div(u + (k - s + 1) / 2 - (k - 1), s); }   // up stages, last first
    u = u - (c.k_latent - 1 + 1) / 2;                                          // dec.conv_in (s = 1)
    *dec_left = (int)(-u);
}

extern "C" int rca_codec_create(const rca_codec_config_t* cfg, const rca_tensor_t* ts, int32_t nt, int32_t device,
                                rca_codec_t** out) {
    if (!cfg || !ts || !out) return fail(RCA_ERR_ARG, "null argument");
    if (cfg->n_stages < 1 || cfg->n_stages > RCA_MAX_STAGES) return fail(RCA_ERR_ARG, "n_stages out of range");
    if (cfg->codebook_dim != 16) return fail(RCA_ERR_ARG, "codebook_dim must be 16 (got %d)", cfg->codebook_dim);
    if (cfg->codebook_size % 128 != 0) return fail(RCA_ERR_ARG, "codebook_size must be a multiple of 128");
    if (cfg->k_in != 7 && cfg->k_in != 3 && cfg->k_in != 5) return fail(RCA_ERR_ARG, "k_in must be 3, 5 or 7");
    if (!(cfg->leaky_slope > 0.0f && cfg->leaky_slope < 1.0f)) return fail(RCA_ERR_ARG, "leaky_slope must be in (0, 1)");
    RCA_HIP(hipSetDevice(device));
    rca_codec* h = new rca_codec();
    h->cfg = *cfg;
    h->device = device;
    int rc = RCA_OK;
    auto bail = [&](int code) { rca_codec_destroy(h); return code; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(RCA_ERR_HIP, "stream create"));
    const int n = cfg->n_stages;
    h->hop = 1;
    for (int i = 0; i < n; ++i) h->hop *= cfg->strides[i];
    codec_receptive_field(*cfg, h->hop, &h->enc_left_frames, &h->dec_left_frames);

    auto add = [&](std::vector<ConvLayer>& v, const std::string& name, int cin, int cout, int k, int s, int pre, int tr,
                   bool pack) -> int {
        ConvLayer L;
        L.cin = cin; L.cout = cout; L.k = k; L.s = s; L.pre = pre; L.tr = tr;
        if ((rc = upload(ts, nt, name + ".weight", (long)cin * cout * k, &L.w)) != RCA_OK) return rc;
        if ((rc = upload(ts, nt, name + ".bias", cout, &L.b)) != RCA_OK) { v.push_back(L); return rc; }
        {
            const size_t nb = ((size_t)cout + 127) / 128 * 128 + 64;
            if (hipMalloc((void**)&L.bp, nb * 4) != hipSuccess || hipMemset(L.bp, 0, nb * 4) != hipSuccess ||
                hipMemcpy(L.bp, L.b, (size_t)cout * 4, hipMemcpyDeviceToDevice) != hipSuccess) {
                v.push_back(L);
                return fail(RCA_ERR_HIP, "bias pad alloc");
            }
        }
        if (pack && !tr && mfma_supported(k, s)) {
            const rca_tensor_t* t = find_tensor(ts, nt, name + ".weight");
            if ((rc = pack_weights((const float*)t->data, L)) != RCA_OK) { v.push_back(L); return rc; }
        }
        if (tr && k == 2 * s) {
            const rca_tensor_t* t = find_tensor(ts, nt, name + ".weight");
            if ((rc = pack_weights_tr((const float*)t->data, L)) != RCA_OK) { v.push_back(L); return rc; }
        }
        v.push_back(L);
        return RCA_OK;
    };
    if ((rc = add(h->enc, "enc.conv_in", 1, cfg->channels[0], cfg->k_in, 1, 0, 0, false)) != RCA_OK) return bail(rc);
    for (int i = 0; i < n; ++i) {
        const int s = cfg->strides[i];
        if ((rc = add(h->enc, "enc.down." + std::to_string(i), cfg->channels[i], cfg->channels[i + 1], 2 * s, s, 1, 0, true)) != RCA_OK)
            return bail(rc);
    }
    if ((rc = add(h->enc, "enc.conv_out", cfg->channels[n], cfg->latent_dim, cfg->k_latent, 1, 1, 0, true)) != RCA_OK) return bail(rc);
    if ((rc = add(h->dec, "dec.conv_in", cfg->codebook_dim, cfg->channels[n], cfg->k_latent, 1, 0, 0, true)) != RCA_OK) return bail(rc);
    for (int i = 0; i < n; ++i) {
        const int s = cfg->strides[n - 1 - i];
        if ((rc = add(h->dec, "dec.up." + std::to_string(i), cfg->channels[n - i], cfg->channels[n - 1 - i], 2 * s, s, 1, 1, false)) != RCA_OK)
            return bail(rc);
    }
    if ((rc = add(h->dec, "dec.conv_out", cfg->channels[0], 1, cfg->k_in, 1, 1, 0, false)) != RCA_OK) return bail(rc);

    const int J = cfg->codebook_dim, N = cfg->codebook_size, R = cfg->codebook_raw_dim, D = cfg->latent_dim;
    if ((rc = upload(ts, nt, "quantizer.in_proj.weight", (long)J * D, &h->q_in_w)) != RCA_OK) return bail(rc);
    if ((rc = upload(ts, nt, "quantizer.in_proj.bias", J, &h->q_in_b)) != RCA_OK) return bail(rc);
    float *raw = nullptr, *pw = nullptr, *pb = nullptr;
    auto free3 = [&]() { for (float* p : {raw, pw, pb}) if (p) (void)hipFree(p); };
    if ((rc = upload(ts, nt, "quantizer.codebook.weight", (long)N * R, &raw)) != RCA_OK) { free3(); return bail(rc); }
    if ((rc = upload(ts, nt, "quantizer.codebook_proj.weight", (long)J * R, &pw)) != RCA_OK) { free3(); return bail(rc); }
    if ((rc = upload(ts, nt, "quantizer.codebook_proj.bias", J, &pb)) != RCA_OK) { free3(); return bail(rc); }
    if (hipMalloc((void**)&h->cb, (size_t)N * J * 4) != hipSuccess || hipMalloc((void**)&h->hc, (size_t)N * 4) != hipSuccess ||
        hipMalloc((void**)&h->cbp, (size_t)N * J * 4) != hipSuccess || hipMalloc((void**)&h->err_flag, 4) != hipSuccess) {
        free3();
        return bail(fail(RCA_ERR_HIP, "codebook alloc"));
    }
    (void)hipMemsetAsync(h->err_flag, 0, 4, h->stream);
#ifdef RCA_CONV_TIMELINE
    if (getenv("RCA_CONV_TIMELINE_OUT")) {
        long* b = nullptr;
        (void)hipMalloc((void**)&b, (size_t)(6 << 16) * 8 * 8);
        (void)hipMemset(b, 0, (size_t)(6 << 16) * 8 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(rca_prof_buf), &b, 8);
    }
#endif
    // projected codebook + half norms are constants of the model: computed once (the reference
    // recomputes the projection on every decode call, audio_tokenizer.py:198)
    codebook_proj_kernel<<<cdiv((long)N * J, 256), 256, 0, h->stream>>>(raw, pw, pb, h->cb, N, R, J);
    codebook_norm_pack_kernel<<<cdiv(N, 256), 256, 0, h->stream>>>(h->cb, h->hc, h->cbp, N, J);
    hipError_t e = hipStreamSynchronize(h->stream);
    free3();
    if (e != hipSuccess) return bail(fail(RCA_ERR_HIP, "codebook init: %s", hipGetErrorString(e)));
    *out = h;
    return RCA_OK;
}

extern "C" int rca_codec_hop(const rca_codec_t* h, int32_t* hop) {
    if (!h || !hop) return fail(RCA_ERR_ARG, "null");
    *hop = h->hop;
    return RCA_OK;
}
extern "C" int rca_codec_num_frames(const rca_codec_t* h, int32_t T, int32_t* F) {
    if (!h || !F || T < 0) return fail(RCA_ERR_ARG, "bad argument");
    *F = (T + h->hop - 1) / h->hop;
    return RCA_OK;
}
extern "C" int rca_codec_set_variant(rca_codec_t* h, int32_t v) {
    if (!h || v < 0 || v > 2) return fail(RCA_ERR_ARG, "variant must be 0, 1 or 2");
    h->variant = v;
    return RCA_OK;
}
extern "C" int rca_codec_sync(rca_codec_t* h) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipSetDevice(h->device));
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}

template <int KS, int S, int CIC, int WM, int WN, int FUSE, int TR, int BF = 0>
static void launch_conv_cfg(const ConvLayer& L, const float* wp, int nchunks, const float* x, float* y, int Lin, int Lc, long Ncols,
                            float slope, const FuseIn& fin, const TrInfo& tr, hipStream_t st, int act) {
    constexpr int NT = RCA_CONV_WPB * WN * 32, MT = WM * 32;
    constexpr int lds = RCA_CONV_WPB * 2 * (CIC * S * ConvLds<S>::stride(WN * 32) + 4) * 4;
    static_assert(lds <= 65536, "LDS budget");
    // 1-D grid: column tiles padded to a multiple of 8 (one per XCD), times the channel tiles (times the phases)
    const long col_tiles = (cdiv(Ncols, NT) + 7) / 8 * 8;
    dim3 grid((unsigned)(col_tiles * cdiv(L.cout, MT) * (TR ? tr.s : 1)));
    if constexpr (!FUSE && !TR && BF == 0) {
        if (L.cin % CIC == 0 && nchunks * CIC == L.cin) {   // whole chunks only: the kernel without the per-channel descriptor selects
            conv1d_mfma_kernel<KS, S, CIC, WM, WN, 0, 0, 0, 1><<<grid, 64 * RCA_CONV_WPB, lds, st>>>(x, wp, L.bp, y, L.cin, Lin, L.cout, Lc, Ncols, nchunks, act, slope, fin, tr);
            return;
        }
    }
    conv1d_mfma_kernel<KS, S, CIC, WM, WN, FUSE, TR, BF><<<grid, 64 * RCA_CONV_WPB, lds, st>>>(x, wp, L.bp, y, L.cin, Lin, L.cout, Lc, Ncols, nchunks, act, slope, fin, tr);
}

template <int KS, int S, int CIC>
// act: bit 0 = LeakyReLU on the input while staging (the layer's pre-activation), bit 1 = LeakyReLU on the output before the
// store (the NEXT layer's pre-activation, applied once by the producer: same values, see run_encoder)
static int launch_conv_mfma(const ConvLayer& L, const float* x, float* y, int B, int Lin, int Lout, float slope, hipStream_t st,
                            const FuseIn* fuse = nullptr, int act = -1, int bf = 0) {
    if (act < 0) act = L.pre;
    const long Ncols = (long)B * Lout;
    if (bf) {   // opt-in bf16 matrix path (rca_codec_set_mfma_mode): 64 x 64 tiles for large launches, 32 x 32 below
        const FuseIn none_b{};
        const TrInfo notr_b{};
        const bool big = (long)cdiv(Ncols, 64) * cdiv(L.cout, 64) >= 2048;
#define RCA_BF_LAUNCH(FU, BFM) do { \
            if (big) launch_conv_cfg<KS, S, CIC, 2, 2, FU, 0, BFM>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, FU ? *fuse : none_b, notr_b, st, act); \
            else launch_conv_cfg<KS, S, CIC, 1, 1, FU, 0, BFM>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, FU ? *fuse : none_b, notr_b, st, act); } while (0)
        if constexpr (KS == 4 || KS == 8 || KS == 16) {
            if (fuse) { if (bf == 3) RCA_BF_LAUNCH(1, 3); else RCA_BF_LAUNCH(1, 1); RCA_LAUNCH_CHECK(); return RCA_OK; }
        }
        if (fuse) return fail(RCA_ERR_ARG, "internal: fused first layer of this shape has no bf16 instantiation");
        if (bf == 3) RCA_BF_LAUNCH(0, 3); else RCA_BF_LAUNCH(0, 1);
#undef RCA_BF_LAUNCH
        RCA_LAUNCH_CHECK();
        return RCA_OK;
    }
    // wave tile 64 channels x 64 columns (two waves per SIMD, most reuse per staged element) while that yields at least ~4
    // rounds of workgroups on the chip's 512 slots; below that the last, partly filled round costs more than the reuse
    // gains and smaller wave tiles win (more workgroups, 3-4 waves per SIMD).  Measured on the k16s8 layer of the 256-window
    // step (896 workgroups of 64 x 64 tiles): 64x64 1081 us / 0.54 GB fetched, 32x64 991 us / 1.3 GB, 64x32 1071 us / 2.6 GB,
    // 32x32 992 us / 3.3 GB -- every column tile streams the layer's 8.4 MB of weights through a 4 MB L2, so narrow
    // column tiles multiply the fetches: 32 x 64 in the middle range, 32 x 32 only for small launches.  64x64 stays ahead
    // at 3360 workgroups (k10s5: 1137 vs 1169 us).
    const long waves_big = (long)cdiv(Ncols, 64) * cdiv(L.cout, 64);
#ifdef RCA_CONV_FORCE_MID   // timing experiment: 32 x 64 wave tiles (3 waves per SIMD) also for the largest launches
    constexpr long BIG_MIN = 1L << 40, MID_MIN = 2048;
#else
    constexpr long BIG_MIN = 8192, MID_MIN = 2048;
#endif
    const FuseIn none{};
    const TrInfo notr{};
    if (fuse) {
        // The fused first layer (K = 128: an 8-chunk loop of ~20 us between ~9 us of prologue + epilogue) runs 64 x 128 wave tiles on
        // 2-channel chunks: the prologue is paid once per 128 columns and the loop is 16 chunks of the same 32 MFMAs (249 VGPRs, no
        // spills; the packed weights are the same array, Kpad does not change).  719 -> 664 us per launch of the 256-window step,
        // bit-identical.  RCA_FUSE_WIDE=0 brings the 64 x 64 tiles back (A/B).
        static const bool wide = []() { const char* e = getenv("RCA_FUSE_WIDE"); return !(e && e[0] == '0'); }();
        if constexpr (KS == 4 && S == 2) {
            if (wide && waves_big >= BIG_MIN && Lout >= 131 && L.cin % 2 == 0) {
                launch_conv_cfg<4, 2, 2, 2, 4, 1, 0>(L, L.wp, L.cin / 2, x, y, Lin, Lout, Ncols, slope, *fuse, notr, st, act);
                RCA_LAUNCH_CHECK();
                return RCA_OK;
            }
        }
        if (waves_big >= BIG_MIN) launch_conv_cfg<KS, S, CIC, 2, 2, 1, 0>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, *fuse, notr, st, act);
        else launch_conv_cfg<KS, S, CIC, 1, 1, 1, 0>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, *fuse, notr, st, act);
    } else {
        // (64 x 128 tiles on the k8s4 layer, <8, 4, 2, 2, 4>: 252 VGPRs, no spills, 1048 vs 1041 us -- its prologue is 7 % of a wave's
        //  life, not 30 %: not kept)
        if (waves_big >= BIG_MIN) launch_conv_cfg<KS, S, CIC, 2, 2, 0, 0>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, none, notr, st, act);
        else if (waves_big >= MID_MIN) launch_conv_cfg<KS, S, CIC, 1, 2, 0, 0>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, none, notr, st, act);
        // the stride-1 k = 3 layers have a short reduction per column tile: 64 x 32 wave tiles (half the staging per MFMA of 32 x 32)
        // measured 213 vs 222 us on conv_out of the 256-window step; weights (0.8 MB) stay in L2, so no extra fetches
        else if (KS == 3 && L.cout >= 64 && waves_big >= 1024) launch_conv_cfg<KS, S, CIC, (KS == 3 ? 2 : 1), 1, 0, 0>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, none, notr, st, act);
        else launch_conv_cfg<KS, S, CIC, 1, 1, 0, 0>(L, L.wp, L.nchunks, x, y, Lin, Lout, Ncols, slope, none, notr, st, act);
    }
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

// ConvTranspose1d (k = 2s) as s phase GEMMs in one launch of the same kernel (KS=2, S=1 view)
static int launch_convtr_mfma(const ConvLayer& L, const float* x, float* y, int B, int Lin, float slope, hipStream_t st) {
    const int Lc = Lin + 1;
    const long Ncols = (long)B * Lc;
    const FuseIn none{};
    const TrInfo tr{L.s, (L.k - L.s + 1) / 2, Lin * L.s, L.cout_pad / 32};
    const long waves_big = (long)cdiv(Ncols, 64) * cdiv(L.cout, 64) * L.s;
    if (L.cout > 32 && waves_big >= 2048) launch_conv_cfg<2, 1, 16, 2, 2, 0, 1>(L, L.wp_tr, L.nchunks_tr, x, y, Lin, Lc, Ncols, slope, none, tr, st, L.pre);
    else launch_conv_cfg<2, 1, 16, 1, 1, 0, 1>(L, L.wp_tr, L.nchunks_tr, x, y, Lin, Lc, Ncols, slope, none, tr, st, L.pre);
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

template <int KS, int S, int CIC, int FUSE>
static void launch_conv_ws(const ConvLayer& L, const float* x, float* y, int Lin, int Lout, long Ncols, float slope, const FuseIn& fin,
                           hipStream_t st) {
    constexpr int KPC = CIC * KS / 2, QPC = KPC / 4;
    constexpr int lds = (2 * (2 * QPC * 64 * 4) + 4 * 2 * (CIC * S * ConvLds<S>::stride(64) + 4)) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv1d_ws_kernel<KS, S, CIC, FUSE>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    const long col_tiles = (cdiv(Ncols, 256) + 7) / 8 * 8;
    dim3 grid((unsigned)(col_tiles * cdiv(L.cout, 64)));
    kern<<<grid, 512, lds, st>>>(x, L.wp2, L.b, y, L.cin, Lin, L.cout, Lout, Ncols, L.nchunks2, L.pre, slope, fin);
}
// variant 2 (experimental, slower than variant 1 as measured in round 1): true when the warp-specialised kernel was launched
static bool try_conv_ws(const ConvLayer& L, const float* x, float* y, int B, int Lin, int Lout, float slope, hipStream_t st, const FuseIn* fuse) {
    const long Ncols = (long)B * Lout;
    if (!L.wp2 || (long)cdiv(Ncols, 64) * cdiv(L.cout, 64) < 1024) return false;
    const FuseIn none{};
    if (L.k == 4 && L.s == 2) { if (fuse) launch_conv_ws<4, 2, 16, 1>(L, x, y, Lin, Lout, Ncols, slope, *fuse, st); else launch_conv_ws<4, 2, 16, 0>(L, x, y, Lin, Lout, Ncols, slope, none, st); }
    else if (L.k == 8 && L.s == 4 && !fuse) launch_conv_ws<8, 4, 8, 0>(L, x, y, Lin, Lout, Ncols, slope, none, st);
    else if (L.k == 16 && L.s == 8 && !fuse) launch_conv_ws<16, 8, 4, 0>(L, x, y, Lin, Lout, Ncols, slope, none, st);
    else if (L.k == 10 && L.s == 5 && !fuse) launch_conv_ws<10, 5, 8, 0>(L, x, y, Lin, Lout, Ncols, slope, none, st);
    else if (L.k == 3 && L.s == 1 && !fuse) launch_conv_ws<3, 1, 16, 0>(L, x, y, Lin, Lout, Ncols, slope, none, st);
    else return false;
    return true;
}

// Streaming tail: launch the LDS-staged chain kernel for one layer.  Returns false when the shape is not covered.
template <int K, int S, int NW>
static void launch_conv_lds(const ConvLayer& L, const float* x, long xbs, int Lvalid, float* y, int B, int Lin, int Lout, int NC, int W, int pre,
                            float slope, int clamp_out, hipStream_t st) {
    const size_t lds = ((size_t)W * (L.cin + 4) + (size_t)NW * L.cin * K) * 4;
    auto kern = conv1d_lds_kernel<K, S, NW>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RCA_LDS_BUDGET); attr_done = true; }
    dim3 grid(cdiv(Lout, NC), cdiv(L.cout, NW), B);
    kern<<<grid, 64 * NW, lds, st>>>(x, xbs, Lvalid, L.w, L.b, y, L.cin, Lin, L.cout, Lout, NC, W, slope, pre, clamp_out);
}
template <int S, int NW>
static void launch_convtr_lds(const ConvLayer& L, const float* x, float* y, int B, int Lin, int NC, int NTI, float slope, hipStream_t st) {
    const size_t lds = ((size_t)NTI * (L.cin + 4) + (size_t)NW * 2 * S * (L.cin + 4)) * 4;
    auto kern = convtr1d_lds_kernel<S, NW>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RCA_LDS_BUDGET); attr_done = true; }
    dim3 grid(cdiv((long)Lin * S, NC), cdiv(L.cout, NW), B);
    kern<<<grid, 64 * NW, lds, st>>>(x, L.w, L.w_lat, L.b, y, L.cin, Lin, L.cout, NC, NTI, slope, L.pre);
}
static bool try_conv_lds(rca_codec* h, const ConvLayer& L, const float* x, long xbs, int Lvalid, float* y, int B, int Lin, int clamp_out, hipStream_t st) {
    const float slope = h->cfg.leaky_slope;
    if (L.tr) {
        if (L.k != 2 * L.s) return false;
        const int Lout = Lin * L.s;
        const int NC = std::min(64, Lout);
        const int NTI = NC / L.s + 3;
        if (L.cin % 4 != 0) return false;
        auto fits = [&](int nw) { return ((size_t)NTI * (L.cin + 4) + (size_t)nw * L.k * (L.cin + 4)) * 4 <= RCA_LDS_BUDGET; };
#define RCA_TR(SS) if (L.s == SS) { if (fits(4)) launch_convtr_lds<SS, 4>(L, x, y, B, Lin, NC, NTI, slope, st); \
                                   else if (fits(2)) launch_convtr_lds<SS, 2>(L, x, y, B, Lin, NC, NTI, slope, st); \
                                   else if (fits(1)) launch_convtr_lds<SS, 1>(L, x, y, B, Lin, NC, NTI, slope, st); else return false; return true; }
        RCA_TR(2) RCA_TR(4) RCA_TR(5) RCA_TR(8)
#undef RCA_TR
        return false;
    }
    const int Lout = Lin / L.s;
    // columns per workgroup: as many as fit beside the weight rows (NW waves = NW output channels share the tile)
    const int nw = L.cout >= 4 ? 4 : 1;
    int NC = std::min(64, Lout);
    auto W_of = [&](int nc) { return (nc * L.s + L.k - L.s + 3 + 3) & ~3; };   // + up to 3 samples of alignment shift, multiple of 4
    auto lds_of = [&](int nc) { return ((size_t)W_of(nc) * (L.cin + 4) + (size_t)nw * L.cin * L.k) * 4; };
    while (NC > 1 && lds_of(NC) > RCA_LDS_BUDGET) NC = (NC + 1) / 2;
    if (lds_of(NC) > RCA_LDS_BUDGET || (L.cin >= 8 && 4 * L.k > 64)) return false;
    const int W = W_of(NC);
#define RCA_CV(KK, SS) if (L.k == KK && L.s == SS) { if (nw == 4) launch_conv_lds<KK, SS, 4>(L, x, xbs, Lvalid, y, B, Lin, Lout, NC, W, L.pre, slope, clamp_out, st); \
                                   else launch_conv_lds<KK, SS, 1>(L, x, xbs, Lvalid, y, B, Lin, Lout, NC, W, L.pre, slope, clamp_out, st); return true; }
    RCA_CV(4, 2) RCA_CV(8, 4) RCA_CV(10, 5) RCA_CV(16, 8) RCA_CV(3, 1) RCA_CV(7, 1)
#undef RCA_CV
    return false;
}

// true when run_conv sends this (non-transposed) layer to conv1d_mfma_kernel: the kernel indexes the input with 32-bit
// element offsets, counts columns in 31 bits and addresses a wave's rows with 32-bit byte offsets from its first batch row
static bool mfma_conv_ok(const rca_codec* h, const ConvLayer& L, int B, int Lin, int clamp_out) {
    if (h->variant < 1 || L.tr || !L.wp || clamp_out || !mfma_supported(L.k, L.s)) return false;
    const int Lout = Lin / L.s;
    return (double)L.cin * Lin < 2.6e8 && (double)L.cout * Lin < 2.6e8 && (double)B * L.cin * Lin < 4.0e9 && (double)B * Lout < 2.0e9;
}

// in_activated: x already holds LeakyReLU(x) (the producer applied this layer's pre-activation); want_post: store
// LeakyReLU(y) if the kernel that runs can (then *post_done = true and the consumer must be told its input is activated)
static int run_conv(rca_codec* h, const ConvLayer& L, const float* x, float* y, int B, int Lin, int clamp_out, hipStream_t st,
                    bool in_activated = false, bool want_post = false, bool* post_done = nullptr) {
    const float slope = h->cfg.leaky_slope;
    if (post_done) *post_done = false;
    if (in_activated && !(h->variant == 1 && !h->lat_mode && mfma_conv_ok(h, L, B, Lin, clamp_out)))
        return fail(RCA_ERR_ARG, "internal: activated input handed to a kernel without that mode");
    if (h->lat_mode && try_conv_lds(h, L, x, (long)L.cin * Lin, Lin, y, B, Lin, clamp_out, st)) {
        RCA_LAUNCH_CHECK();
        return RCA_OK;
    }
    // the MFMA kernels index the input with 32-bit element offsets and count columns in 31 bits
    // ... and addresses a wave's rows with 32-bit byte offsets from its first batch row (two rows of input / output at most)
    const bool rows32 = (double)L.cin * Lin < 2.6e8 && (double)L.cout * Lin * (L.tr ? L.s : 1) < 2.6e8;
    if (L.tr && h->variant >= 1 && L.wp_tr && rows32 && (double)B * L.cin * Lin < 4.0e9 && (double)B * (Lin + 1) < 2.0e9) return launch_convtr_mfma(L, x, y, B, Lin, slope, st);
    if (L.tr) {
        const long total = (long)B * L.cout * Lin * L.s;
        convtr1d_chain_kernel<<<cdiv(total, 256), 256, 0, st>>>(x, L.w, L.b, y, B, L.cin, Lin, L.cout, L.k, L.s, L.pre, slope);
        RCA_LAUNCH_CHECK();
        return RCA_OK;
    }
    const int Lout = Lin / L.s;
    const double cflops = 2.0 * L.cin * L.k * L.cout * (double)B * Lout;
    const double cbytes = 4.0 * ((double)B * L.cin * Lin + (double)B * L.cout * Lout + (double)L.cin * L.k * L.cout);
    // the MFMA kernel indexes the input with 32-bit element offsets
    if (mfma_conv_ok(h, L, B, Lin, clamp_out)) {
        ProfScope ps(h, st, 0, cflops, cbytes);
        if (h->variant == 2 && try_conv_ws(L, x, y, B, Lin, Lout, slope, st, nullptr)) { RCA_LAUNCH_CHECK(); return RCA_OK; }
        const int act = ((L.pre && !in_activated) ? 1 : 0) | (want_post ? 2 : 0);
        if (post_done) *post_done = want_post;
        if (L.k == 4 && L.s == 2) return launch_conv_mfma<4, 2, 4>(L, x, y, B, Lin, Lout, slope, st, nullptr, act, h->mfma_mode);
        if (L.k == 8 && L.s == 4) return launch_conv_mfma<8, 4, RCA_CIC_K8>(L, x, y, B, Lin, Lout, slope, st, nullptr, act, h->mfma_mode);
        if (L.k == 10 && L.s == 5) return launch_conv_mfma<10, 5, 4>(L, x, y, B, Lin, Lout, slope, st, nullptr, act, h->mfma_mode);
        if (L.k == 16 && L.s == 8) return launch_conv_mfma<16, 8, RCA_CIC_K16>(L, x, y, B, Lin, Lout, slope, st, nullptr, act, h->mfma_mode);
        if (L.k == 3 && L.s == 1) return launch_conv_mfma<3, 1, RCA_CIC_K3>(L, x, y, B, Lin, Lout, slope, st, nullptr, act, h->mfma_mode);
    }
    if (in_activated) return fail(RCA_ERR_ARG, "internal: activated input handed to a kernel without that mode");
    const long total = (long)B * L.cout * Lout;
    ProfScope ps(h, st, 3, cflops, cbytes);
    conv1d_chain_kernel<<<cdiv(total, 256), 256, 0, st>>>(x, L.w, L.b, y, B, L.cin, Lin, L.cout, Lout, L.k, L.s, L.pre, slope, clamp_out);
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

// weights of encoder layer li (>= 1) in A-fragment order, rounded to bf16 (hi) + the rounded remainder (lo)
static int pack_weights_bf16(const ConvLayer& L, Bf16Pack* out) {
    const long nw = (long)L.cout * L.cin * L.k;
    std::vector<float> w(nw);
    RCA_HIP(hipMemcpy(w.data(), L.w, nw * 4, hipMemcpyDeviceToHost));
    const int ncot = L.cout / 32, ncb = L.cin / 16;
    std::vector<conv_bf16raw> hi((size_t)ncot * ncb * L.k * 64 * 8), lo(hi.size());
    auto rne = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (conv_bf16raw)(u >> 16); };
    auto up = [](conv_bf16raw h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int cot = 0; cot < ncot; ++cot)
        for (int cb = 0; cb < ncb; ++cb)
            for (int kk = 0; kk < L.k; ++kk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = cot * 32 + (lane & 31), ci = cb * 16 + 8 * (lane >> 5) + j;
                        const float v = w[((long)co * L.cin + ci) * L.k + kk];
                        const size_t o = ((((size_t)cot * ncb + cb) * L.k + kk) * 64 + lane) * 8 + j;
                        hi[o] = rne(v);
                        lo[o] = rne(v - up(hi[o]));
                    }
    RCA_HIP(hipMalloc((void**)&out->hi, hi.size() * 2));
    RCA_HIP(hipMalloc((void**)&out->lo, lo.size() * 2));
    RCA_HIP(hipMemcpy(out->hi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    RCA_HIP(hipMemcpy(out->lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    return RCA_OK;
}

template <int KS, int S, int WGM, int CPS, int OUT, int FUSE = 0, int WMT = 1>
static int launch_conv_bf16(bool split, const ConvLayer& L, const Bf16Pack& wp, const conv_bf16raw* xh, const conv_bf16raw* xl, conv_bf16raw* yh,
                            conv_bf16raw* yl, float* yf, int B, int Lin, int act, float slope, hipStream_t st, RowSrc fsrc = RowSrc{},
                            const float* w_in8 = nullptr, int in_act = 0) {
    constexpr int WGN = 4 / WGM, NT = 128 * WGN;
    constexpr int SLOTS = NT + (KS - 1) / S;
    const int Lout = Lin / S;
    const int n_ct = cdiv(Lout, NT), n_cot = L.cout / (32 * WGM * (split ? 1 : WMT));
    const dim3 grid((unsigned)(((long)n_ct * B + 7) / 8 * 8 * n_cot));
    const int nstages = L.cin / 16 / CPS;
    const size_t lds = (size_t)((nstages > 1 && !(split && S >= 5)) ? 2 : 1) * CPS * (split ? 2 : 1) * 2 * S * SLOTS * 16;   // one-stage layers and the single-buffered split layers never touch a second buffer
    if (split) {
        auto k = conv_bf16_blk_kernel<KS, S, WGM, CPS, 1, OUT, FUSE, 1>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        k<<<grid, 256, lds, st>>>(xh, xl, wp.hi, wp.lo, L.b, yh, yl, yf, L.cin, Lin, L.cout, Lout, act, slope, B, n_ct, n_cot, fsrc, w_in8, in_act);
    } else {
        auto k = conv_bf16_blk_kernel<KS, S, WGM, CPS, 0, OUT, FUSE, WMT>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        k<<<grid, 256, lds, st>>>(xh, xl, wp.hi, wp.lo, L.b, yh, yl, yf, L.cin, Lin, L.cout, Lout, act, slope, B, n_ct, n_cot, fsrc, w_in8, in_act);
    }
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

// dispatch by layer geometry; false: no instantiation for this layer (the caller falls back to conv1d_mfma_kernel's bf16 modes)
static bool bf16_blk_layer_ok(const ConvLayer& L, bool last) {
    if (L.tr || L.cin % 16 || L.cout % 64) return false;
    if (last) return L.k == 3 && L.s == 1 && L.cin % 64 == 0 && L.cout % 128 == 0;
    if (L.k == 4 && L.s == 2) return L.cin % 32 == 0;
    if ((L.k == 8 && L.s == 4) || (L.k == 10 && L.s == 5) || (L.k == 16 && L.s == 8)) return L.cout % 128 == 0;
    return false;
}
static bool bf16_blk_ok(const rca_codec* h, int tap_layer) {
    const size_t n = h->enc.size();
    if (h->mfma_mode == 0 || h->lat_mode || h->variant != 1 || n < 3) return false;
    if (h->mfma_mode == 3 && !h->bf16_blk_split) return false;   // (RCA_BF16_BLK_SPLIT=0: the round-3 kernel, 2.43 ms per step against 2.18 here)
    if (tap_layer >= 0 && tap_layer != (int)n - 1) return false;
    if (h->enc[0].k != 7 || h->enc[0].cin != 1 || h->enc[0].cout % 16 || h->enc[0].cout > 64) return false;
    for (size_t li = 1; li < n; ++li)
        if (!bf16_blk_layer_ok(h->enc[li], li + 1 == n)) return false;
    return true;
}
static int run_encoder_bf16(rca_codec* h, RowSrc src, int B, int Tp, size_t max_elems, hipStream_t st, float** ze_out, int tap_layer, float* tap_dev) {
    const bool split = h->mfma_mode == 3;
    const float slope = h->cfg.leaky_slope;
    const size_t n = h->enc.size();
    int rc;
    if (h->bf16_packs.size() != n) {
        h->bf16_packs.assign(n, Bf16Pack{});
        for (size_t li = 1; li < n; ++li)
            if ((rc = pack_weights_bf16(h->enc[li], &h->bf16_packs[li])) != RCA_OK) return rc;
        const ConvLayer& L0 = h->enc[0];
        std::vector<float> w0((size_t)L0.cout * L0.k), b0(L0.cout), t8((size_t)L0.cout * 8, 0.0f);
        RCA_HIP(hipMemcpy(w0.data(), L0.w, w0.size() * 4, hipMemcpyDeviceToHost));
        RCA_HIP(hipMemcpy(b0.data(), L0.b, b0.size() * 4, hipMemcpyDeviceToHost));
        for (int co = 0; co < L0.cout; ++co) {
            for (int kk = 0; kk < L0.k; ++kk) t8[(size_t)co * 8 + kk] = w0[(size_t)co * L0.k + kk];
            t8[(size_t)co * 8 + 7] = b0[co];
        }
        RCA_HIP(hipMalloc((void**)&h->bf16_packs[0].w_in8, t8.size() * 4));
        RCA_HIP(hipMemcpy(h->bf16_packs[0].w_in8, t8.data(), t8.size() * 4, hipMemcpyHostToDevice));
    }
    auto hi_of = [&](int buf) { return h->act[buf].as<conv_bf16raw>(); };
    auto lo_of = [&](int buf) { return h->act[buf].as<conv_bf16raw>() + max_elems; };   // second half of the 4-byte-per-element buffer
    const double esz = split ? 4.0 : 2.0;
    int cur = 0, L = Tp;
    const bool fuse_in = h->enc[1].k == 4 && h->enc[1].s == 2 && h->enc[0].cout == 32 && !getenv("RCA_BF16_NO_FUSE_IN");
    if (!fuse_in) {
        const ConvLayer& L0 = h->enc[0];
        const long total = (long)B * L;
        ProfScope ps(h, st, 2, 2.0 * L0.k * L0.cout * (double)total, 4.0 * (double)total + esz * (double)total * L0.cout);
        const int act = h->enc[1].pre ? 1 : 0;
        if (split) conv_in_blk_kernel<7, 1><<<dim3(cdiv(L, 256), B), 256, 0, st>>>(src, h->bf16_packs[0].w_in8, L0.b, hi_of(cur), lo_of(cur), B, L0.cout, L, act, slope);
        else conv_in_blk_kernel<7, 0><<<dim3(cdiv(L, 256), B), 256, 0, st>>>(src, h->bf16_packs[0].w_in8, L0.b, hi_of(cur), lo_of(cur), B, L0.cout, L, act, slope);
        RCA_LAUNCH_CHECK();
    }
    for (size_t li = 1; li < n; ++li) {
        const ConvLayer& Ly = h->enc[li];
        const bool last = li + 1 == n;
        const int Lout = L / Ly.s;
        const int act = (!last && h->enc[li + 1].pre) ? 1 : 0;
        const bool fused_li = li == 1 && fuse_in;   // reads PCM (f32), computes conv_in on the way
        ProfScope ps(h, st, 0, 2.0 * Ly.cin * Ly.k * Ly.cout * (double)B * Lout + (fused_li ? 2.0 * h->enc[0].k * h->enc[0].cout * (double)B * L : 0.0),
                     (fused_li ? 4.0 * (double)B * L : esz * (double)B * Ly.cin * L) + (last ? 4.0 : esz) * (double)B * Ly.cout * Lout +
                         esz * (double)Ly.cin * Ly.k * Ly.cout);
        const Bf16Pack& wp = h->bf16_packs[li];
        const conv_bf16raw *xh = hi_of(cur), *xl = lo_of(cur);
        conv_bf16raw *yh = hi_of(cur ^ 1), *yl = lo_of(cur ^ 1);
        float* yf = h->act[cur ^ 1].as<float>();
        // rounded mode: 64 x 128 wave tiles (WMT = 2) on the k10 / s5 layer, where one B fragment feeding two MFMAs pays
        // (162 -> 147 us); measured equal on k8 and slower on k16 / k3 (profiles/r04/experiments/bf16_blk_ablations.txt).
        // hi + lo keeps 32 x 128 everywhere (its registers hold two planes)
        if (!split && Ly.k == 10 && Ly.cout % 256 == 0) rc = launch_conv_bf16<10, 5, 4, 1, 0, 0, 2>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st);
        else if (last) rc = launch_conv_bf16<3, 1, 4, 4, 1>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st);
        else if (Ly.k == 4 && li == 1 && fuse_in)
            rc = launch_conv_bf16<4, 2, 2, 2, 0, 1>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st, src, h->bf16_packs[0].w_in8, Ly.pre ? 1 : 0);
        else if (Ly.k == 4) rc = launch_conv_bf16<4, 2, 2, 2, 0>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st);
        else if (Ly.k == 8) rc = launch_conv_bf16<8, 4, 4, 1, 0>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st);
        else if (Ly.k == 10) rc = launch_conv_bf16<10, 5, 4, 1, 0>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st);
        else rc = launch_conv_bf16<16, 8, 4, 1, 0>(split, Ly, wp, xh, xl, yh, yl, yf, B, L, act, slope, st);
        if (rc != RCA_OK) return rc;
        L = Lout;
        cur ^= 1;
        if (last && tap_layer == (int)li) RCA_HIP(hipMemcpyAsync(tap_dev, yf, (size_t)B * Ly.cout * L * 4, hipMemcpyDeviceToDevice, st));
    }
    *ze_out = h->act[cur].as<float>();
    return RCA_OK;
}

// encoder stack: rows described by src -> ze [B][D][F] left in *ze_out (a workspace buffer).
// tap_layer >= 0 copies that layer's output (device->device) into tap_dev.
static int run_encoder(rca_codec* h, RowSrc src, int B, hipStream_t st, float** ze_out, int* F_out, int tap_layer, float* tap_dev) {
    const rca_codec_config_t& c = h->cfg;
    const int F = (src.T + h->hop - 1) / h->hop;
    const int Tp = F * h->hop;
    if (F < 1) return fail(RCA_ERR_ARG, "empty audio (T=%d)", src.T);
    size_t max_elems = 0;
    {
        long L = Tp;
        max_elems = std::max(max_elems, (size_t)B * c.channels[0] * L);
        for (int i = 0; i < c.n_stages; ++i) { L /= c.strides[i]; max_elems = std::max(max_elems, (size_t)B * c.channels[i + 1] * L); }
        max_elems = std::max(max_elems, (size_t)B * c.latent_dim * L);
    }
    int rc;
    if ((rc = h->act[0].ensure(max_elems * 4)) != RCA_OK) return rc;
    if ((rc = h->act[1].ensure(max_elems * 4)) != RCA_OK) return rc;
    if (bf16_blk_ok(h, tap_layer) && (double)B * max_elems < 4.0e18 && Tp / h->enc[1].s >= 64) {   // opt-in bf16 modes: the blocked pipeline
        *F_out = F;
        return run_encoder_bf16(h, src, B, Tp, max_elems, st, ze_out, tap_layer, tap_dev);
    }
    int cur = 0;
    int L = Tp;
    size_t first = 1;
    const ConvLayer& E0 = h->enc[0];
    const ConvLayer& E1 = h->enc[1];
    // conv_in fused into the first strided layer (MFMA variant, 7-tap conv_in, layer 0 not tapped)
    // (the fused kernel addresses the PCM of the two batch rows a wave can touch with 32-bit offsets from the lower one,
    //  and assumes a row is longer than a wave's window of columns)
    const bool fuse01 = !h->lat_mode && h->variant >= 1 && tap_layer != 0 && E0.k == 7 && E1.wp && E1.k == 2 * E1.s && (double)B * E1.cin * L < 4.0e9 &&
                        ((E1.k == 4 && E1.s == 2) || (E1.k == 8 && E1.s == 4) || (E1.k == 16 && E1.s == 8)) && L / E1.s >= 67 &&
                        (double)E1.cout * (L / E1.s) < 2.6e8 &&
                        (src.row_off ? (double)src.span : (double)src.C * (double)std::labs(src.chan_stride) + (double)std::labs(src.win_stride) + src.T) < 5.0e8;
    // LeakyReLU hand-off: when layer li+1 is pre-activated and runs on conv1d_mfma_kernel, layer li stores LeakyReLU(y) and
    // li+1 skips the activation while staging (max(v, slope*v) of the same value either way: bit-identical, but applied once
    // per element instead of once per element per consuming workgroup per chunk).  Off when a layer output is tapped.
    bool prev_post = false;
    auto handoff = [&](size_t next, int Lnext) {
        return tap_layer < 0 && h->variant == 1 && !h->lat_mode && next < h->enc.size() && h->enc[next].pre && mfma_conv_ok(h, h->enc[next], B, Lnext, 0);
    };
    if (fuse01) {
        float* y = h->act[cur].as<float>();
        FuseIn fin{src, E0.w, E0.b};
        const float slope = c.leaky_slope;
        const int Lout = L / E1.s;
        {
            ProfScope ps(h, st, 0, 2.0 * E1.cin * E1.k * E1.cout * (double)B * Lout + 2.0 * E0.k * E0.cout * (double)B * L,
                         4.0 * ((double)B * src.T + (double)B * E1.cout * Lout + (double)E1.cin * E1.k * E1.cout));
            prev_post = handoff(2, Lout);
            const int act = (E1.pre ? 1 : 0) | (prev_post ? 2 : 0);
            if (h->variant == 2 && try_conv_ws(E1, nullptr, y, B, L, Lout, slope, st, &fin)) { rc = RCA_OK; RCA_LAUNCH_CHECK(); }
            else if (E1.k == 4) rc = launch_conv_mfma<4, 2, 4>(E1, nullptr, y, B, L, Lout, slope, st, &fin, act, h->mfma_mode);
            else if (E1.k == 8) rc = launch_conv_mfma<8, 4, RCA_CIC_K8>(E1, nullptr, y, B, L, Lout, slope, st, &fin, act, h->mfma_mode);
            else rc = launch_conv_mfma<16, 8, RCA_CIC_K16>(E1, nullptr, y, B, L, Lout, slope, st, &fin, act, h->mfma_mode);
        }
        if (rc != RCA_OK) return rc;
        L = Lout;
        first = 2;
        if (tap_layer == 1) RCA_HIP(hipMemcpyAsync(tap_dev, y, (size_t)B * E1.cout * L * 4, hipMemcpyDeviceToDevice, st));
    } else if (h->lat_mode && !src.row_off && src.win_stride == 0 && src.C == B &&
               try_conv_lds(h, h->enc[0], src.base, src.chan_stride, src.T, h->act[cur].as<float>(), B, L, 0, st)) {
        RCA_LAUNCH_CHECK();   // streaming tail: conv_in through the same latency kernel (rows straight from the caller's window)
    } else {
        const ConvLayer& L0 = h->enc[0];
        const long total = (long)B * L;
        float* y = h->act[cur].as<float>();
        ProfScope ps(h, st, 2, 2.0 * L0.k * L0.cout * (double)total, 4.0 * ((double)total + (double)total * L0.cout));
        if (L0.k == 7) conv_in_kernel<7><<<cdiv(total, 256), 256, 0, st>>>(src, L0.w, L0.b, y, B, L0.cout, L);
        else if (L0.k == 5) conv_in_kernel<5><<<cdiv(total, 256), 256, 0, st>>>(src, L0.w, L0.b, y, B, L0.cout, L);
        else conv_in_kernel<3><<<cdiv(total, 256), 256, 0, st>>>(src, L0.w, L0.b, y, B, L0.cout, L);
        RCA_LAUNCH_CHECK();
        if (tap_layer == 0) RCA_HIP(hipMemcpyAsync(tap_dev, y, (size_t)B * L0.cout * L * 4, hipMemcpyDeviceToDevice, st));
    }
    for (size_t li = first; li < h->enc.size(); ++li) {
        const ConvLayer& Ly = h->enc[li];
        float* x = h->act[cur].as<float>();
        float* y = h->act[cur ^ 1].as<float>();
        bool done = false;
        if ((rc = run_conv(h, Ly, x, y, B, L, 0, st, prev_post, handoff(li + 1, L / Ly.s), &done)) != RCA_OK) return rc;
        prev_post = done;
        L /= Ly.s;
        cur ^= 1;
        if (tap_layer == (int)li) RCA_HIP(hipMemcpyAsync(tap_dev, y, (size_t)B * Ly.cout * L * 4, hipMemcpyDeviceToDevice, st));
    }
    *ze_out = h->act[cur].as<float>();
    *F_out = F;
    return RCA_OK;
}

// in_proj + nearest-neighbour search over rows (b, f in [f0, f0+fc)); writes int64 codes via dst
static int run_quantize(rca_codec* h, const float* ze, int ze_is_rows, int B, int F, int f0, int fc, RowDst dst, hipStream_t st,
                        float* ztap_dev) {
    const rca_codec_config_t& c = h->cfg;
    const long rows = (long)B * fc;
    const int J = c.codebook_dim, N = c.codebook_size;
    int rc;
    if ((rc = h->zbuf.ensure((size_t)rows * J * 4)) != RCA_OK) return rc;
    const size_t need_keys = (size_t)rows;
    if (h->keys.cap < need_keys * 8) {
        if ((rc = h->keys.ensure(need_keys * 8)) != RCA_OK) return rc;
        RCA_HIP(hipMemsetAsync(h->keys.p, 0, h->keys.cap, st));
    }
    float* z = h->zbuf.as<float>();
    in_proj_kernel<<<cdiv(rows * J, 256), 256, 0, st>>>(ze, h->q_in_w, h->q_in_b, z, B, c.latent_dim, F, f0, fc, J, ze_is_rows);
    RCA_LAUNCH_CHECK();
    if (ztap_dev) RCA_HIP(hipMemcpyAsync(ztap_dev, z, (size_t)rows * J * 4, hipMemcpyDeviceToDevice, st));
    unsigned long long* keys = h->keys.as<unsigned long long>();
    ProfScope ps(h, st, 1, 2.0 * (double)rows * N * J, 4.0 * ((double)N * (J + 1) + (double)rows * J) + 8.0 * rows);
    if (h->variant >= 1) {
        constexpr int FN = 2;
        const int ftiles = (int)cdiv(rows, FN * 32);
        const int total_tiles = N / 32;
        // enough code splits to put >= ~1024 workgroups on the chip, each wave keeping >= 4 tiles
        int splits = (int)std::max(1L, std::min((long)total_tiles / 16, (1024 + ftiles - 1) / (long)ftiles));
        int tps = (total_tiles + splits - 1) / splits;
        tps = (tps + 3) / 4 * 4;
        splits = (total_tiles + tps - 1) / tps;
        dim3 grid(ftiles, splits);
        vq_mfma_kernel<FN><<<grid, 256, 0, st>>>(z, h->cbp, h->hc, keys, rows, N, tps);
    } else {
        constexpr int FB = 4;
        const int fblocks = (int)cdiv(rows, FB);
        int splits = (int)std::max(1L, std::min((long)N / 1024, (1024 + fblocks - 1) / (long)fblocks));
        int cps = (N + splits - 1) / splits;
        cps = (cps + 255) / 256 * 256;
        splits = (N + cps - 1) / cps;
        dim3 grid(fblocks, splits);
        vq_chain_kernel<FB><<<grid, 256, 0, st>>>(z, h->cb, h->hc, keys, rows, N, cps);
    }
    RCA_LAUNCH_CHECK();
    dst.fc = fc;
    vq_finalize_kernel<<<cdiv(rows, 256), 256, 0, st>>>(keys, dst, rows);
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

// `_dev` entry points run on the caller's stream (NULL = the legacy default stream, as everywhere in
// HIP).  The handle's workspace is shared by all calls, so when the stream changes between two calls
// the new stream first waits for the work already queued on the previous one.
static hipStream_t pick_stream(rca_codec* h, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (h->last_stream_valid && h->last_stream != st) {
        if (!h->xev) (void)hipEventCreateWithFlags(&h->xev, hipEventDisableTiming);
        (void)hipEventRecord(h->xev, h->last_stream);
        (void)hipStreamWaitEvent(st, h->xev, 0);
    }
    h->last_stream = st;
    h->last_stream_valid = true;
    return st;
}

extern "C" int rca_codec_encode_dev(rca_codec_t* h, const float* pcm, int32_t B, int32_t T, int64_t* codes, void* stream) {
    if (!h || !pcm || !codes || B < 1 || T < 1) return fail(RCA_ERR_ARG, "encode: bad argument (B=%d T=%d)", B, T);
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    RowSrc src{pcm, B, (long)T, 0, T};
    float* ze; int F, rc;
    if ((rc = run_encoder(h, src, B, st, &ze, &F, -1, nullptr)) != RCA_OK) return rc;
    RowDst dst{codes, B, (long)F, 0, F};
    return run_quantize(h, ze, 0, B, F, 0, F, dst, st, nullptr);
}

// Frames that can be dropped from the left of a window of F frames when only the last `keep` are wanted.
static int trimmable_frames(const rca_codec* h, int F, int keep) { return std::max(0, F - keep - h->enc_left_frames); }

extern "C" int rca_codec_encode_tail_dev(rca_codec_t* h, const float* pcm, int32_t B, int32_t T, int32_t n_keep, int64_t* codes,
                                         void* stream) {
    if (!h || !pcm || !codes || B < 1 || T < 1 || n_keep < 1) return fail(RCA_ERR_ARG, "encode_tail: bad argument (B=%d T=%d keep=%d)", B, T, n_keep);
    const int F = (T + h->hop - 1) / h->hop;
    if (n_keep > F) return fail(RCA_ERR_ARG, "encode_tail: %d frames wanted, the window holds %d", n_keep, F);
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    // frames are counted from the window start: drop whole frames so the grid (and the right edge) stay put
    const int j = trimmable_frames(h, F, n_keep);
    RowSrc src{pcm + (long)j * h->hop, B, (long)T, 0, T - j * h->hop};
    float* ze; int Ft, rc;
    // a handful of frames: the chains, not the FLOPs, set the time -> LDS-staged scalar-chain kernels (same bits)
    h->lat_mode = h->variant >= 1 && (long)B * (F - j) <= RCA_LAT_MAX_FRAMES;
    rc = run_encoder(h, src, B, st, &ze, &Ft, -1, nullptr);
    h->lat_mode = false;
    if (rc != RCA_OK) return rc;
    RowDst dst{codes, B, (long)n_keep, 0, n_keep};
    return run_quantize(h, ze, 0, B, Ft, Ft - n_keep, n_keep, dst, st, nullptr);
}

extern "C" int rca_codec_receptive_field(const rca_codec_t* h, int32_t* enc_left_frames, int32_t* dec_left_frames) {
    if (!h || !enc_left_frames || !dec_left_frames) return fail(RCA_ERR_ARG, "null");
    *enc_left_frames = h->enc_left_frames;
    *dec_left_frames = h->dec_left_frames;
    return RCA_OK;
}

extern "C" int rca_codec_set_window_trim(rca_codec_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    h->window_trim = enable != 0;
    return RCA_OK;
}

extern "C" int rca_codec_encode(rca_codec_t* h, const float* pcm_host, int32_t B, int32_t T, int64_t* codes_host) {
    if (!h || !pcm_host || !codes_host || B < 1 || T < 1) return fail(RCA_ERR_ARG, "encode: bad argument (B=%d T=%d)", B, T);
    RCA_HIP(hipSetDevice(h->device));
    const int F = (T + h->hop - 1) / h->hop;
    int rc;
    if ((rc = h->io_a.ensure((size_t)B * T * 4)) != RCA_OK) return rc;
    if ((rc = h->io_b.ensure((size_t)B * F * 8)) != RCA_OK) return rc;
    RCA_HIP(hipMemcpyAsync(h->io_a.p, pcm_host, (size_t)B * T * 4, hipMemcpyHostToDevice, h->stream));
    if ((rc = rca_codec_encode_dev(h, h->io_a.as<float>(), B, T, h->io_b.as<int64_t>(), h->stream)) != RCA_OK) return rc;
    RCA_HIP(hipMemcpyAsync(codes_host, h->io_b.p, (size_t)B * F * 8, hipMemcpyDeviceToHost, h->stream));
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}

extern "C" int rca_codec_encode_tap(rca_codec_t* h, const float* pcm_host, int32_t B, int32_t T, int32_t layer, float* out_host,
                                    int64_t out_numel) {
    if (!h || !pcm_host || !out_host || B < 1 || T < 1) return fail(RCA_ERR_ARG, "tap: bad argument");
    RCA_HIP(hipSetDevice(h->device));
    const int n = h->cfg.n_stages;
    if (layer < 0 || layer > n + 2) return fail(RCA_ERR_ARG, "tap layer out of range");
    int rc;
    if ((rc = h->io_a.ensure((size_t)B * T * 4)) != RCA_OK) return rc;
    if ((rc = h->io_b.ensure((size_t)out_numel * 4)) != RCA_OK) return rc;
    RCA_HIP(hipMemcpyAsync(h->io_a.p, pcm_host, (size_t)B * T * 4, hipMemcpyHostToDevice, h->stream));
    RowSrc src{h->io_a.as<float>(), B, (long)T, 0, T};
    float* ze; int F;
    (void)pick_stream(h, h->stream);
    // element-count check
    {
        const int Fx = (T + h->hop - 1) / h->hop;
        long L = (long)Fx * h->hop, want;
        if (layer == 0) want = (long)B * h->cfg.channels[0] * L;
        else if (layer <= n) { for (int i = 0; i < layer; ++i) L /= h->cfg.strides[i]; want = (long)B * h->cfg.channels[layer] * L; }
        else if (layer == n + 1) want = (long)B * h->cfg.latent_dim * Fx;
        else want = (long)B * Fx * h->cfg.codebook_dim;
        if (want != out_numel) return fail(RCA_ERR_ARG, "tap: out_numel %ld, expected %ld", (long)out_numel, want);
    }
    if ((rc = run_encoder(h, src, B, h->stream, &ze, &F, layer <= n + 1 ? layer : -1, h->io_b.as<float>())) != RCA_OK) return rc;
    if (layer == n + 2) {
        DevBuf tmp;
        if ((rc = tmp.ensure((size_t)B * F * 8)) != RCA_OK) return rc;
        RowDst dst{tmp.as<int64_t>(), B, (long)F, 0, F};
        rc = run_quantize(h, ze, 0, B, F, 0, F, dst, h->stream, h->io_b.as<float>());
        hipError_t e = hipStreamSynchronize(h->stream);
        tmp.release();
        if (rc != RCA_OK) return rc;
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "tap sync: %s", hipGetErrorString(e));
    }
    RCA_HIP(hipMemcpyAsync(out_host, h->io_b.p, (size_t)out_numel * 4, hipMemcpyDeviceToHost, h->stream));
    RCA_HIP(hipStreamSynchronize(h->stream));
    return RCA_OK;
}

extern "C" int rca_codec_encode_chunk_range_dev(rca_codec_t* h, const float* audio, int32_t C, int64_t N, int32_t chunk, int32_t ctx,
                                                int32_t batch_windows, int64_t chunk_begin, int64_t chunk_end, int64_t* codes,
                                                int64_t codes_per_channel, void* stream) {
    if (!h || !audio || !codes || C < 1 || N < 1 || chunk < 1 || ctx < 0 || batch_windows < 1)
        return fail(RCA_ERR_ARG, "encode_windows: bad argument");
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    const int sr = h->cfg.sample_rate;
    // int(audio_secs * framerate) exactly as audio_tokenizer.py:99-100 computes it
    const double framerate = (double)sr / (double)h->hop;
    const int fpc = (int)(((double)chunk / (double)sr) * framerate);
    const long n_chunks = N / chunk;
    if (chunk_begin < 0 || chunk_end > n_chunks || chunk_begin > chunk_end)
        return fail(RCA_ERR_ARG, "chunk range [%ld,%ld) outside [0,%ld]", (long)chunk_begin, (long)chunk_end, n_chunks);
    if (fpc < 1) return fail(RCA_ERR_ARG, "chunk of %d samples holds no whole frame", chunk);
    if (codes_per_channel < (chunk_end - chunk_begin) * fpc)
        return fail(RCA_ERR_ARG, "codes buffer too small: %ld < %ld", (long)codes_per_channel, (long)(chunk_end - chunk_begin) * fpc);
    const int W = std::max(chunk, ctx);
    int rc;
    long i = chunk_begin;
    // warm-up: the rolling context is still shorter than ctx (audio_tokenizer.py:72-74)
    for (; i < chunk_end && (i + 1) * (long)chunk < W; ++i) {
        const int Tfull = (int)((i + 1) * chunk);
        const int jw = h->window_trim ? trimmable_frames(h, (Tfull + h->hop - 1) / h->hop, fpc) : 0;
        const int T = Tfull - jw * h->hop;
        RowSrc src{audio + (long)jw * h->hop, C, (long)N, 0, T};
        float* ze; int F;
        if ((rc = run_encoder(h, src, C, st, &ze, &F, -1, nullptr)) != RCA_OK) return rc;
        if (F < fpc) return fail(RCA_ERR_STATE, "window of %d samples has %d frames < %d kept", T, F, fpc);
        RowDst dst{codes + (i - chunk_begin) * fpc, C, (long)codes_per_channel, 0, fpc};
        if ((rc = run_quantize(h, ze, 0, C, F, F - fpc, fpc, dst, st, nullptr)) != RCA_OK) return rc;
    }
    // steady state: full windows [end - W, end), batch_windows rows (window x channel) per pass
    const int wins_per_pass = std::max(1, batch_windows / C);
    while (i < chunk_end) {
        const int nw = (int)std::min<long>(wins_per_pass, chunk_end - i);
        // window_trim: the kept frames see at most enc_left_frames to their left -- drop the whole frames before that
        const int jw = h->window_trim ? trimmable_frames(h, (W + h->hop - 1) / h->hop, fpc) : 0;
        const long start0 = (i + 1) * (long)chunk - W + (long)jw * h->hop;
        RowSrc src{audio + start0, C, (long)N, (long)chunk, W - jw * h->hop};
        float* ze; int F;
        if ((rc = run_encoder(h, src, nw * C, st, &ze, &F, -1, nullptr)) != RCA_OK) return rc;
        if (F < fpc) return fail(RCA_ERR_STATE, "window has %d frames < %d kept", F, fpc);
        RowDst dst{codes + (i - chunk_begin) * fpc, C, (long)codes_per_channel, (long)fpc, fpc};
        if ((rc = run_quantize(h, ze, 0, nw * C, F, F - fpc, fpc, dst, st, nullptr)) != RCA_OK) return rc;
        i += nw;
    }
    return RCA_OK;
}

extern "C" int rca_codec_encode_windows_dev(rca_codec_t* h, const float* audio, int32_t C, int64_t N, int32_t chunk, int32_t ctx,
                                            int32_t batch_windows, int64_t* codes, int64_t codes_per_channel, void* stream) {
    if (chunk < 1) return fail(RCA_ERR_ARG, "encode_windows: bad argument");
    return rca_codec_encode_chunk_range_dev(h, audio, C, N, chunk, ctx, batch_windows, 0, N / chunk, codes, codes_per_channel, stream);
}

// B windows of T samples each, anywhere in one device buffer (any file, channel, position): window b starts at
// audio + src_off[b]; the last n_keep codes of window b are stored at codes + dst_off[b].  Per window the same arithmetic as
// rca_codec_encode_dev (window_trim applies as in the chunk-range call).  This is what lets the batch CLI fill its 256-window
// passes from MANY files at once, including the short warm-up windows at the start of every file.
extern "C" int rca_codec_encode_rows_dev(rca_codec_t* h, const float* audio, const int64_t* src_off, int32_t B, int32_t T, int32_t n_keep,
                                         int64_t* codes, const int64_t* dst_off, int64_t span, void* stream) {
    if (!h || !audio || !src_off || !codes || !dst_off || B < 1 || T < 1 || n_keep < 1 || span < T)
        return fail(RCA_ERR_ARG, "encode_rows: bad argument");
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    static_assert(sizeof(long) == sizeof(int64_t), "offset tables are 64-bit");
    const int Ffull = (T + h->hop - 1) / h->hop;
    if (Ffull < n_keep) return fail(RCA_ERR_ARG, "encode_rows: a window of %d samples has %d frames < %d kept", T, Ffull, n_keep);
    const int jw = h->window_trim ? trimmable_frames(h, Ffull, n_keep) : 0;
    RowSrc src{audio + (long)jw * h->hop, B, 0, 0, T - jw * h->hop, reinterpret_cast<const long*>(src_off), (long)span};
    float* ze; int F;
    int rc;
    if ((rc = run_encoder(h, src, B, st, &ze, &F, -1, nullptr)) != RCA_OK) return rc;
    RowDst dst{codes, B, 0, 0, n_keep, reinterpret_cast<const long*>(dst_off)};
    return run_quantize(h, ze, 0, B, F, F - n_keep, n_keep, dst, st, nullptr);
}

extern "C" int rca_codec_encoder_dev(rca_codec_t* h, const float* pcm, int32_t B, int32_t T, float* ze_out, void* stream) {
    if (!h || !pcm || !ze_out || B < 1 || T < 1) return fail(RCA_ERR_ARG, "encoder: bad argument");
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    RowSrc src{pcm, B, (long)T, 0, T};
    float* ze; int F, rc;
    if ((rc = run_encoder(h, src, B, st, &ze, &F, -1, nullptr)) != RCA_OK) return rc;
    const long total = (long)B * h->cfg.latent_dim * F;
    transpose_df_kernel<<<cdiv(total, 256), 256, 0, st>>>(ze, ze_out, B, h->cfg.latent_dim, F);
    RCA_LAUNCH_CHECK();
    return RCA_OK;
}

extern "C" int rca_codec_quantize_dev(rca_codec_t* h, const float* ze_rows, int64_t rows, int64_t* codes, void* stream) {
    if (!h || !ze_rows || !codes || rows < 1) return fail(RCA_ERR_ARG, "quantize: bad argument");
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    RowDst dst{codes, 1, 0, (long)rows, (int)rows};
    // one "batch row" holding `rows` frames: B=1, F=rows
    return run_quantize(h, ze_rows, 1, 1, (int)rows, 0, (int)rows, dst, st, nullptr);
}

static int run_decoder(rca_codec* h, float* zq /*[B][J][F] in act[0]*/, int B, int F, float* pcm_out, hipStream_t st) {
    const rca_codec_config_t& c = h->cfg;
    int rc;
    int cur = 0;
    int L = F;
    for (size_t li = 0; li < h->dec.size(); ++li) {
        const ConvLayer& Ly = h->dec[li];
        const bool last = li + 1 == h->dec.size();
        float* x = h->act[cur].as<float>();
        float* y = last ? pcm_out : h->act[cur ^ 1].as<float>();
        if ((rc = run_conv(h, Ly, x, y, B, L, last ? 1 : 0, st)) != RCA_OK) return rc;
        if (Ly.tr) L *= Ly.s;
        cur ^= 1;
    }
    (void)c; (void)zq;
    return RCA_OK;
}

static int decoder_workspace(rca_codec* h, int B, int F) {
    const rca_codec_config_t& c = h->cfg;
    const int n = c.n_stages;
    size_t max_elems = (size_t)B * c.codebook_dim * F;
    long L = F;
    max_elems = std::max(max_elems, (size_t)B * c.channels[n] * L);
    for (int i = 0; i < n; ++i) { L *= c.strides[n - 1 - i]; max_elems = std::max(max_elems, (size_t)B * c.channels[n - 1 - i] * L); }
    int rc;
    if ((rc = h->act[0].ensure(max_elems * 4)) != RCA_OK) return rc;
    return h->act[1].ensure(max_elems * 4);
}

extern "C" int rca_codec_decode_dev(rca_codec_t* h, const int64_t* codes, int32_t B, int32_t F, float* pcm, void* stream) {
    if (!h || !codes || !pcm || B < 1 || F < 1) return fail(RCA_ERR_ARG, "decode: bad argument (B=%d F=%d)", B, F);
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    int rc;
    if ((rc = decoder_workspace(h, B, F)) != RCA_OK) return rc;
    const int J = h->cfg.codebook_dim;
    float* zq = h->act[0].as<float>();
    embed_codes_kernel<<<cdiv((long)B * J * F, 256), 256, 0, st>>>(codes, (long)F, h->cb, zq, B, F, J, h->cfg.codebook_size, h->err_flag);
    RCA_LAUNCH_CHECK();
    return run_decoder(h, zq, B, F, pcm, st);
}

extern "C" int rca_codec_decode_tail_dev(rca_codec_t* h, const int64_t* codes, int32_t B, int32_t F, int32_t n_samples, float* pcm,
                                         void* stream) {
    if (!h || !codes || !pcm || B < 1 || F < 1 || n_samples < 1) return fail(RCA_ERR_ARG, "decode_tail: bad argument (B=%d F=%d n=%d)", B, F, n_samples);
    if ((long)n_samples > (long)F * h->hop) return fail(RCA_ERR_ARG, "decode_tail: %d samples wanted, %d codes give %ld", n_samples, F, (long)F * h->hop);
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    // first frame any kept sample belongs to, minus what the decoder sees to its left
    const int f0 = (int)(((long)F * h->hop - n_samples) / h->hop);
    const int j = std::max(0, f0 - h->dec_left_frames);
    const int Ft = F - j;
    int rc;
    if ((rc = decoder_workspace(h, B, Ft)) != RCA_OK) return rc;
    const long Tt = (long)Ft * h->hop;
    if ((rc = h->tail.ensure((size_t)B * Tt * 4)) != RCA_OK) return rc;
    const int J = h->cfg.codebook_dim;
    float* zq = h->act[0].as<float>();
    embed_codes_kernel<<<cdiv((long)B * J * Ft, 256), 256, 0, st>>>(codes + j, (long)F, h->cb, zq, B, Ft, J, h->cfg.codebook_size, h->err_flag);
    RCA_LAUNCH_CHECK();
    h->lat_mode = h->variant >= 1 && (long)B * Ft <= RCA_LAT_MAX_FRAMES;
    rc = run_decoder(h, zq, B, Ft, h->tail.as<float>(), st);
    h->lat_mode = false;
    if (rc != RCA_OK) return rc;
    RCA_HIP(hipMemcpy2DAsync(pcm, (size_t)n_samples * 4, h->tail.as<float>() + (Tt - n_samples), (size_t)Tt * 4, (size_t)n_samples * 4, B,
                             hipMemcpyDeviceToDevice, st));
    return RCA_OK;
}

// ---- streaming host calls.  One duplex frame makes one encode-tail and one decode-tail call of the same shape as
// the frame before: H2D of the window, ~10 short kernels, D2H of the tail, one sync.  From the second call of a
// shape on, the whole sequence (copies included) is a hipGraph over pinned staging buffers.  A captured graph holds
// workspace pointers, so it is keyed by a signature of them and re-captured when any buffer has been reallocated.
static unsigned long long workspace_signature(const rca_codec* h) {
    unsigned long long sig = 1469598103934665603ull;
    for (const void* p : {h->act[0].p, h->act[1].p, h->zbuf.p, h->keys.p, h->io_a.p, h->io_b.p, h->tail.p, (void*)h->pin}) {
        sig ^= (unsigned long long)(uintptr_t)p;
        sig *= 1099511628211ull;
    }
    return sig;
}
static int ensure_pinned(rca_codec* h, size_t bytes) {
    if (bytes <= h->pin_cap) return RCA_OK;
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr; h->pin_cap = 0;
    RCA_HIP(hipHostMalloc(&h->pin, bytes + (bytes >> 2), hipHostMallocDefault));
    h->pin_cap = bytes + (bytes >> 2);
    return RCA_OK;
}
// kind 0: encode tail (a=T, b=n_keep), kind 1: decode tail (a=F, b=n_samples).  Pinned layout: [input | output | err].
static int stream_call(rca_codec* h, int kind, int B, int a, int b, const void* in_host, size_t in_bytes, void* out_host, size_t out_bytes) {
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    int rc;
    const size_t in_pad = (in_bytes + 255) & ~(size_t)255, out_pad = (out_bytes + 255) & ~(size_t)255;
    if ((rc = ensure_pinned(h, in_pad + out_pad + 256)) != RCA_OK) return rc;
    if ((rc = h->io_a.ensure(in_pad)) != RCA_OK) return rc;
    if ((rc = h->io_b.ensure(out_pad)) != RCA_OK) return rc;
    char* pin = (char*)h->pin;
    int* perr = (int*)(pin + in_pad + out_pad);
    memcpy(pin, in_host, in_bytes);
    auto enqueue = [&]() -> int {
        hipError_t e = hipMemcpyAsync(h->io_a.p, pin, in_bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "stream h2d: %s", hipGetErrorString(e));
        int r = kind == 0 ? rca_codec_encode_tail_dev(h, h->io_a.as<float>(), B, a, b, h->io_b.as<int64_t>(), st)
                          : rca_codec_decode_tail_dev(h, h->io_a.as<int64_t>(), B, a, b, h->io_b.as<float>(), st);
        if (r != RCA_OK) return r;
        e = hipMemcpyAsync(pin + in_pad, h->io_b.p, out_bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(perr, h->err_flag, 4, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "stream d2h: %s", hipGetErrorString(e));
        return RCA_OK;
    };
    (void)pick_stream(h, st);
    rca_codec::StreamGraph& g = h->sgraphs[{kind, B, a, b, h->variant}];
    const bool want_graph = h->stream_graphs && !h->profile && g.seen >= 1;
    if (want_graph && g.exec && g.sig != workspace_signature(h)) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    if (want_graph && !g.exec) {
        // the eager call before this one sized every workspace buffer for this shape: nothing allocates under capture
        hipGraph_t graph = nullptr;
        RCA_HIP(hipStreamSynchronize(st));
        RCA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = enqueue();
        hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc != RCA_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "stream capture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { g.exec = nullptr; return fail(RCA_ERR_HIP, "stream graph instantiate: %s", hipGetErrorString(e)); }
        g.sig = workspace_signature(h);
    }
    if (want_graph) {
        RCA_HIP(hipGraphLaunch(g.exec, st));
    } else if ((rc = enqueue()) != RCA_OK) {
        return rc;
    }
    ++g.seen;
    RCA_HIP(hipStreamSynchronize(st));
    if (*perr) {
        RCA_HIP(hipMemsetAsync(h->err_flag, 0, 4, st));
        return fail(RCA_ERR_ARG, "decode: code out of range [0, %d)", h->cfg.codebook_size);
    }
    memcpy(out_host, pin + in_pad, out_bytes);
    return RCA_OK;
}

extern "C" int rca_codec_encode_tail(rca_codec_t* h, const float* pcm_host, int32_t B, int32_t T, int32_t n_keep, int64_t* codes_host) {
    if (!h || !pcm_host || !codes_host || B < 1 || T < 1 || n_keep < 1) return fail(RCA_ERR_ARG, "encode_tail: bad argument (B=%d T=%d keep=%d)", B, T, n_keep);
    if (n_keep > (T + h->hop - 1) / h->hop) return fail(RCA_ERR_ARG, "encode_tail: %d frames wanted, the window holds %d", n_keep, (T + h->hop - 1) / h->hop);
    return stream_call(h, 0, B, T, n_keep, pcm_host, (size_t)B * T * 4, codes_host, (size_t)B * n_keep * 8);
}

extern "C" int rca_codec_decode_tail(rca_codec_t* h, const int64_t* codes_host, int32_t B, int32_t F, int32_t n_samples, float* pcm_host) {
    if (!h || !codes_host || !pcm_host || B < 1 || F < 1 || n_samples < 1) return fail(RCA_ERR_ARG, "decode_tail: bad argument (B=%d F=%d n=%d)", B, F, n_samples);
    if ((long)n_samples > (long)F * h->hop) return fail(RCA_ERR_ARG, "decode_tail: %d samples wanted, %d codes give %ld", n_samples, F, (long)F * h->hop);
    return stream_call(h, 1, B, F, n_samples, codes_host, (size_t)B * F * 8, pcm_host, (size_t)B * n_samples * 4);
}

// Support for a caller that captures the tail calls into a graph of its own (rca_duplex_frame): a signature of everything a captured
// tail call bakes in (workspace addresses, variant), and a hand-over of the handle's stream ordering to the caller's stream so that
// the capture records no wait on an event from outside it.
extern "C" int rca_codec_workspace_sig(rca_codec_t* h, uint64_t* sig) {
    if (!h || !sig) return fail(RCA_ERR_ARG, "null");
    *sig = (workspace_signature(h) * 1099511628211ull + (unsigned long long)h->variant) * 1099511628211ull + (unsigned long long)h->mfma_mode;
    return RCA_OK;
}
extern "C" int rca_codec_stream_handoff(rca_codec_t* h, void* stream) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipSetDevice(h->device));
    if (h->last_stream_valid && h->last_stream != (hipStream_t)stream) RCA_HIP(hipStreamSynchronize(h->last_stream));
    h->last_stream = (hipStream_t)stream;
    h->last_stream_valid = true;
    return RCA_OK;
}
extern "C" int rca_codec_codebook_size(const rca_codec_t* h, int32_t* n) {
    if (!h || !n) return fail(RCA_ERR_ARG, "null");
    *n = h->cfg.codebook_size;
    return RCA_OK;
}

extern "C" int rca_codec_set_mfma_mode(rca_codec_t* h, int32_t mode) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    if (mode != 0 && mode != 1 && mode != 3) return fail(RCA_ERR_ARG, "mfma mode %d (0 = f32 exact, 1 = bf16, 3 = bf16 hi + lo split)", mode);
    if (mode != h->mfma_mode) {
        RCA_HIP(hipSetDevice(h->device));
        RCA_HIP(hipStreamSynchronize(h->stream));
        for (auto& kv : h->sgraphs)      // captured tail calls hold the kernels of the old mode
            if (kv.second.exec) { (void)hipGraphExecDestroy(kv.second.exec); kv.second.exec = nullptr; }
    }
    h->mfma_mode = mode;
    { const char* e = getenv("RCA_BF16_BLK_SPLIT"); h->bf16_blk_split = !(e && atoi(e) == 0); }
    return RCA_OK;
}

extern "C" int rca_codec_set_stream_graphs(rca_codec_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    h->stream_graphs = enable != 0;
    return RCA_OK;
}

extern "C" int rca_codec_decoder_dev(rca_codec_t* h, const float* zq_bfj, int32_t B, int32_t F, float* pcm, void* stream) {
    if (!h || !zq_bfj || !pcm || B < 1 || F < 1) return fail(RCA_ERR_ARG, "decoder: bad argument");
    RCA_HIP(hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    int rc;
    if ((rc = decoder_workspace(h, B, F)) != RCA_OK) return rc;
    const int J = h->cfg.codebook_dim;
    float* zq = h->act[0].as<float>();
    transpose_fj_kernel<<<cdiv((long)B * J * F, 256), 256, 0, st>>>(zq_bfj, zq, B, F, J);
    RCA_LAUNCH_CHECK();
    return run_decoder(h, zq, B, F, pcm, st);
}

extern "C" int rca_codec_decode(rca_codec_t* h, const int64_t* codes_host, int32_t B, int32_t F, float* pcm_host) {
    if (!h || !codes_host || !pcm_host || B < 1 || F < 1) return fail(RCA_ERR_ARG, "decode: bad argument (B=%d F=%d)", B, F);
    RCA_HIP(hipSetDevice(h->device));
    int rc;
    const size_t T = (size_t)F * h->hop;
    if ((rc = h->io_a.ensure((size_t)B * F * 8)) != RCA_OK) return rc;
    if ((rc = h->io_b.ensure((size_t)B * T * 4)) != RCA_OK) return rc;
    RCA_HIP(hipMemcpyAsync(h->io_a.p, codes_host, (size_t)B * F * 8, hipMemcpyHostToDevice, h->stream));
    if ((rc = rca_codec_decode_dev(h, h->io_a.as<int64_t>(), B, F, h->io_b.as<float>(), h->stream)) != RCA_OK) return rc;
    int err = 0;
    RCA_HIP(hipMemcpyAsync(pcm_host, h->io_b.p, (size_t)B * T * 4, hipMemcpyDeviceToHost, h->stream));
    RCA_HIP(hipMemcpyAsync(&err, h->err_flag, 4, hipMemcpyDeviceToHost, h->stream));
    RCA_HIP(hipStreamSynchronize(h->stream));
    if (err) {
        RCA_HIP(hipMemsetAsync(h->err_flag, 0, 4, h->stream));
        return fail(RCA_ERR_ARG, "decode: code out of range [0, %d)", h->cfg.codebook_size);
    }
    return RCA_OK;
}

extern "C" int rca_codec_codebook_dev(rca_codec_t* h, const float** out) {
    if (!h || !out) return fail(RCA_ERR_ARG, "null");
    *out = h->cb;
    return RCA_OK;
}
extern "C" int rca_codec_codebook(rca_codec_t* h, float* out_host) {
    if (!h || !out_host) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipSetDevice(h->device));
    RCA_HIP(hipMemcpy(out_host, h->cb, (size_t)h->cfg.codebook_size * h->cfg.codebook_dim * 4, hipMemcpyDeviceToHost));
    return RCA_OK;
}

extern "C" int rca_codec_profile(rca_codec_t* h, int32_t enable) {
    if (!h) return fail(RCA_ERR_ARG, "null");
    h->profile = enable != 0;
    return RCA_OK;
}
extern "C" int rca_codec_profile_read(rca_codec_t* h, int32_t kclass, double* total_ms, int64_t* launches, double* flops,
                                      double* bytes) {
    if (!h || !total_ms || !launches || !flops || !bytes) return fail(RCA_ERR_ARG, "null");
    RCA_HIP(hipSetDevice(h->device));
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    std::vector<rca_codec::Prof> keep;
    for (auto& p : h->prof) {
        if (p.kclass != kclass) { keep.push_back(p); continue; }
        RCA_HIP(hipEventSynchronize(p.b));
        float t = 0;
        RCA_HIP(hipEventElapsedTime(&t, p.a, p.b));
        ms += t; fl += p.flops; by += p.bytes; ++n;
        h->prof_pool.push_back(p);
    }
    h->prof.swap(keep);
    *total_ms = ms; *launches = n; *flops = fl; *bytes = by;
    return RCA_OK;
}
