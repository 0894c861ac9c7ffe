// Micro-benchmark: which shape of the conv chunk loop keeps the f32 matrix pipe busy?
// Models conv1d_mfma_kernel<8,4,4,2,2> (64 x 64 wave tile, 16 k pairs = 64 MFMAs per chunk, 12 staged 8-byte elements per lane,
// 32 B-fragment LDS reads, 8 weight quads per chunk) with real dependencies (loads -> LDS writes -> LDS reads -> MFMA):
//   mode 0  phased     : [issue loads] [LDS reads + MFMAs] [LDS writes]            (the shipped structure)
//   mode 1  interleaved: every k pair = 4 MFMAs + 2 LDS reads (4 k pairs ahead) + one LDS write of the next chunk and the
//                        load of the chunk after it into the same register + weight refills
//   mode 2  MFMAs only (same accumulators, no memory)
// at 1 / 2 waves per SIMD (workgroups per CU), reporting TFLOP/s against the 157.3 peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define U 72
#define BUF (4 * 4 * U + 4)
#define KPC 16
#define RE 3
#define CIC 4

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ gx, const float4* __restrict__ gw, float* __restrict__ out,
                                            int nchunks, int ntiles, long xmask) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xs = lds_all + wave * 2 * BUF;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int rd_base = half * U + (lane & 31) + 1;
    int wr_off[RE];
    for (int r = 0; r < RE; ++r) {
        const int e = 2 * (lane + 64 * r);           // element pair of a 264-element channel window
        const int slot = e / 4, p = e & 3;
        wr_off[r] = e < 264 ? p * U + slot : 4 * 4 * U;   // spare word for the lanes past the window
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long xb = (((long)tile * 4 + wave) * 65536) & xmask;   // 64 KB of input per wave tile, streamed
        const float4* wq = gw + (long)(tile & 3) * nchunks * 8 * 64 + lane;
        f32x2 sreg[CIC][RE];
        float4 a[2][4];
        auto load_chunk = [&](int c) {
#pragma unroll
            for (int cl = 0; cl < CIC; ++cl)
#pragma unroll
                for (int r = 0; r < RE; ++r)
                    sreg[cl][r] = *reinterpret_cast<const f32x2*>(gx + ((xb + ((long)(c * CIC + cl) * 1024) + 2 * (lane + 64 * r)) & xmask));
        };
        auto write_chunk = [&](int buf) {
#pragma unroll
            for (int cl = 0; cl < CIC; ++cl)
#pragma unroll
                for (int r = 0; r < RE; ++r) {
                    float* d = xs + buf * BUF + cl * 4 * U + wr_off[r];
                    d[0] = sreg[cl][r][0];
                    d[U] = sreg[cl][r][1];
                }
        };
        if (MODE != 2) {
            load_chunk(0);
#pragma unroll
            for (int wm = 0; wm < 2; ++wm)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[wm][q] = wq[(wm * 4 + q) * 64];
            write_chunk(0);
            if (MODE == 1) load_chunk(1);
        } else {
            for (int wm = 0; wm < 2; ++wm) for (int q = 0; q < 4; ++q) a[wm][q] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
        }
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            const float* xr = xs + buf * BUF + rd_base;
            const float4* wn = wq + (long)((c + 1) % nchunks) * 8 * 64;
            if (MODE == 0) {
                load_chunk((c + 1) % nchunks);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    float bv[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kp = nb * 4 + i;
                        const int off = (kp >> 2) * 4 * U + ((kp & 3) >> 1) * 2 * U + (kp & 1);   // (ci, phase pair, q): immediates
                        bv[i][0] = xr[off];
                        bv[i][1] = xr[off + 32];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kp = nb * 4 + i;
#pragma unroll
                        for (int wm = 0; wm < 2; ++wm) {
                            const float4 q4 = a[wm][kp >> 2];
                            const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                            for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i][wn_], acc[wm][wn_], 0, 0, 0);
                        }
                        if ((kp & 3) == 3) {
#pragma unroll
                            for (int wm = 0; wm < 2; ++wm) a[wm][kp >> 2] = wn[(wm * 4 + (kp >> 2)) * 64];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                write_chunk(buf ^ 1);
                __builtin_amdgcn_wave_barrier();
            } else if (MODE == 1) {
                // ring of B fragments, 4 k pairs deep
                float bv[KPC][2];
                auto rd = [&](int kp) {
                    const int off = (kp >> 2) * 4 * U + ((kp & 3) >> 1) * 2 * U + (kp & 1);
                    bv[kp][0] = xr[off];
                    bv[kp][1] = xr[off + 32];
                };
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) rd(kp);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kp = 0; kp < KPC; ++kp) {
                    if (kp + 4 < KPC) rd(kp + 4);
#pragma unroll
                    for (int wm = 0; wm < 2; ++wm) {
                        const float4 q4 = a[wm][kp >> 2];
                        const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kp][wn_], acc[wm][wn_], 0, 0, 0);
                    }
                    if (kp < CIC * RE) {   // one staged element pair per k pair: write chunk c+1, then fetch chunk c+2 into the same register
                        const int cl = kp / RE, r = kp % RE;
                        float* d = xs + (buf ^ 1) * BUF + cl * 4 * U + wr_off[r];
                        d[0] = sreg[cl][r][0];
                        d[U] = sreg[cl][r][1];
                        sreg[cl][r] = *reinterpret_cast<const f32x2*>(gx + ((xb + ((long)(((c + 2) % nchunks) * CIC + cl) * 1024) + 2 * (lane + 64 * r)) & xmask));
                    }
                    if ((kp & 3) == 3) {
#pragma unroll
                        for (int wm = 0; wm < 2; ++wm) a[wm][kp >> 2] = wn[(wm * 4 + (kp >> 2)) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int kp = 0; kp < KPC; ++kp)
#pragma unroll
                    for (int wm = 0; wm < 2; ++wm) {
                        const float4 q4 = a[wm][kp >> 2];
                        const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, (float)kp, acc[wm][wn_], 0, 0, 0);
                    }
            }
        }
        // epilogue stand-in: one store per lane per tile, depending on every accumulator
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        out[((long)tile * 256 + threadIdx.x) & 0xFFFFF] = s;
    }
}

template <int MODE>
static void run(int wg_per_cu, int nchunks, const char* tag, const float* gx, const float4* gw, float* out, long xmask) {
    const int grid = 256 * wg_per_cu;
    const int ntiles = grid * 24;            // 24 wave tiles per wave: prologue-free steady state
    const size_t lds = 4 * 2 * BUF * 4;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256, lds>>>(gx, gw, out, nchunks, grid, xmask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256, lds>>>(gx, gw, out, nchunks, ntiles, xmask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)ntiles * 4 * nchunks * 64 * 4096.0;
    printf("%-12s waves/SIMD=%d chunks=%d : %6.1f TFLOP/s = %.3f of 157.3 (%.3f ms)\n", tag, wg_per_cu, nchunks, flops / ms / 1e9, flops / ms / 1e9 / 157.3, ms);
}
int main() {
    const long xfloats = 1L << 27;   // 512 MB of input: streamed
    float* gx; float4* gw; float* out;
    hipMalloc(&gx, xfloats * 4 + 65536); hipMemset(gx, 0, xfloats * 4 + 65536);
    hipMalloc(&gw, 4 * 64 * 8 * 64 * 16); hipMemset(gw, 0, 4 * 64 * 8 * 64 * 16);
    hipMalloc(&out, (1 << 20) * 4 + 4096);
    const long xmask = xfloats - 1;
    for (int w = 1; w <= 2; ++w) {
        run<2>(w, 16, "mfma-only", gx, gw, out, xmask);
        run<0>(w, 16, "phased", gx, gw, out, xmask);
        run<1>(w, 16, "interleaved", gx, gw, out, xmask);
    }
    run<0>(2, 64, "phased", gx, gw, out, xmask);
    run<1>(2, 64, "interleaved", gx, gw, out, xmask);
    return 0;
}
