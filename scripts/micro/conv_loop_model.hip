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
typedef float f32x4 __attribute__((ext_vector_type(4)));


// loads the compiler does not see (so it cannot place its own, conservative, s_waitcnt): the counted waits are written by hand
__device__ __forceinline__ f32x2 asm_ld2(const float* p) { f32x2 v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ f32x4 asm_ld4(const f32x4* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
#define WAIT_VM(N, ...) asm volatile("s_waitcnt vmcnt(" #N ")" : __VA_ARGS__ : : "memory")

#define U 72
#define BUF (4 * 4 * U + 4)
#define KPC 16
#define RE 3
#define CIC 4

template <int MODE, int ABL = 0>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ gx, const f32x4* __restrict__ gw, float* __restrict__ out,
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
    auto lane_el = [&](int r) { const int e = lane + 64 * r; return e < 132 ? e : 128; };   // 264 elements per channel window
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long xb = (((long)(tile >> 1) * 4 + wave) * 65536) & xmask;   // 64 KB of input per wave tile, streamed; channel-tile siblings share it
        const f32x4* wq = gw + (long)(tile & 3) * nchunks * 8 * 64 + lane;
        f32x2 sreg[CIC][RE];
        f32x2 sreg2[CIC][RE];
        f32x4 a[2][4];
        auto load_chunk = [&](int c) {
#pragma unroll
            for (int cl = 0; cl < CIC; ++cl)
#pragma unroll
                for (int r = 0; r < RE; ++r)
                    sreg[cl][r] = *reinterpret_cast<const f32x2*>(gx + ((xb + ((long)(c * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
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
            if (MODE == 5) {
#pragma unroll
                for (int cl = 0; cl < CIC; ++cl)
#pragma unroll
                    for (int r = 0; r < RE; ++r) {
                        sreg[cl][r] = asm_ld2(gx + ((xb + ((long)(1 * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
                        sreg2[cl][r] = asm_ld2(gx + ((xb + ((long)(2 * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
                    }
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            if (MODE == 4) {
#pragma unroll
                for (int cl = 0; cl < CIC; ++cl)
#pragma unroll
                    for (int r = 0; r < RE; ++r)
                        sreg[cl][r] = asm_ld2(gx + ((xb + ((long)(1 * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) as a builtin: it also clears the compiler's own pending-load state, which would
                                                      // otherwise be merged into the loop header and re-waited (vmcnt(1)) in every iteration
            }
        } else {
            for (int wm = 0; wm < 2; ++wm) for (int q = 0; q < 4; ++q) a[wm][q] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
        }
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            const float* xr = xs + buf * BUF + rd_base;
            const f32x4* wn = wq + (long)((c + 1) % nchunks) * 8 * 64;
            if (MODE == 0) {
                if (!(ABL & 1)) load_chunk((c + 1) % nchunks);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    float bv[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kp = nb * 4 + i;
                        const int off = (kp >> 2) * 4 * U + ((kp & 3) >> 1) * 2 * U + (kp & 1);   // (ci, phase pair, q): immediates
                        if (ABL & 4) { bv[i][0] = sreg[0][0][0] + off; bv[i][1] = sreg[0][0][1]; }
                        else { bv[i][0] = xr[off]; bv[i][1] = xr[off + 32]; }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kp = nb * 4 + i;
#pragma unroll
                        for (int wm = 0; wm < 2; ++wm) {
                            const f32x4 q4 = a[wm][kp >> 2];
                            const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                            for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i][wn_], acc[wm][wn_], 0, 0, 0);
                        }
                        if ((kp & 3) == 3 && !(ABL & 8)) {
#pragma unroll
                            for (int wm = 0; wm < 2; ++wm) a[wm][kp >> 2] = wn[(wm * 4 + (kp >> 2)) * 64];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 2)) write_chunk(buf ^ 1);
                __builtin_amdgcn_wave_barrier();
            } else if (MODE == 1) {
                // ring of B fragments, 4 k pairs deep
                float bv[KPC][2];
                auto rd = [&](int kp) {
                    const int off = (kp >> 2) * 4 * U + ((kp & 3) >> 1) * 2 * U + (kp & 1);
                    if (ABL & 4) { bv[kp][0] = sreg[0][0][0] + off; bv[kp][1] = sreg[0][0][1]; return; }
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
                        const f32x4 q4 = a[wm][kp >> 2];
                        const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kp][wn_], acc[wm][wn_], 0, 0, 0);
                    }
                    if (kp < CIC * RE) {   // one staged element pair per k pair: write chunk c+1, then fetch chunk c+2 into the same register
                        const int cl = kp / RE, r = kp % RE;
                        float* d = xs + (buf ^ 1) * BUF + cl * 4 * U + wr_off[r];
                        if (!(ABL & 2)) {
                            d[0] = sreg[cl][r][0];
                            d[U] = sreg[cl][r][1];
                        }
                        if (!(ABL & 1)) sreg[cl][r] = *reinterpret_cast<const f32x2*>(gx + ((xb + ((long)(((c + 2) % nchunks) * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
                    }
                    if ((kp & 3) == 3 && !(ABL & 8)) {
#pragma unroll
                        for (int wm = 0; wm < 2; ++wm) a[wm][kp >> 2] = wn[(wm * 4 + (kp >> 2)) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_wave_barrier();
            } else if (MODE == 5) {
                // mode 4 with the input loads TWO chunks ahead of their LDS write (two register sets; the loop is unrolled by two so that
                // every asm load has ONE fixed destination): 20 loads are issued per chunk, so the load a write depends on has 39 younger ones
                if (c & 1) continue;
                auto body = [&](int cc, f32x2 (&sr)[CIC][RE]) __attribute__((always_inline)) {
                    const int bufc = cc & 1;
                    const float* xrc = xs + bufc * BUF + rd_base;
                    const f32x4* wnc = wq + (long)((cc + 1) % nchunks) * 8 * 64;
                    float bv[KPC][2];
                    auto rd = [&](int kp) {
                        const int off = (kp >> 2) * 4 * U + ((kp & 3) >> 1) * 2 * U + (kp & 1);
                        bv[kp][0] = xrc[off];
                        bv[kp][1] = xrc[off + 32];
                    };
#pragma unroll
                    for (int kp = 0; kp < 4; ++kp) rd(kp);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kp = 0; kp < KPC; ++kp) {
                        if (kp + 4 < KPC) rd(kp + 4);
                        if ((kp & 3) == 0) {
                            if (kp == 12) WAIT_VM(18, "+v"(a[0][3]), "+v"(a[1][3]));
                            else if (kp == 0) WAIT_VM(14, "+v"(a[0][0]), "+v"(a[1][0]));
                            else if (kp == 4) WAIT_VM(14, "+v"(a[0][1]), "+v"(a[1][1]));
                            else WAIT_VM(14, "+v"(a[0][2]), "+v"(a[1][2]));
                        }
#pragma unroll
                        for (int wm = 0; wm < 2; ++wm) {
                            const f32x4 q4 = a[wm][kp >> 2];
                            const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                            for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kp][wn_], acc[wm][wn_], 0, 0, 0);
                        }
                        if (kp < CIC * RE) {
                            const int cl = kp / RE, r = kp % RE;
                            float* d = xs + (bufc ^ 1) * BUF + cl * 4 * U + wr_off[r];
                            WAIT_VM(39, "+v"(sr[cl][r]));
                            d[0] = sr[cl][r][0];
                            d[U] = sr[cl][r][1];
                            sr[cl][r] = asm_ld2(gx + ((xb + ((long)(((cc + 3) % nchunks) * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
                        }
                        if ((kp & 3) == 3) {
#pragma unroll
                            for (int wm = 0; wm < 2; ++wm) a[wm][kp >> 2] = asm_ld4(wnc + (wm * 4 + (kp >> 2)) * 64);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_wave_barrier();
                };
                body(c, sreg);
                body(c + 1, sreg2);
            } else if (MODE == 4) {
                // interleaved, loads in asm with hand-counted waits: 20 loads are issued per chunk (12 x, 8 weight quads)
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
                    if ((kp & 3) == 0) {   // first use of weight quad kp/4, fetched one chunk ago
                        if (kp == 12) WAIT_VM(18, "+v"(a[0][3]), "+v"(a[1][3]));
                        else if (kp == 0) WAIT_VM(14, "+v"(a[0][0]), "+v"(a[1][0]));
                        else if (kp == 4) WAIT_VM(14, "+v"(a[0][1]), "+v"(a[1][1]));
                        else WAIT_VM(14, "+v"(a[0][2]), "+v"(a[1][2]));
                    }
#pragma unroll
                    for (int wm = 0; wm < 2; ++wm) {
                        const f32x4 q4 = a[wm][kp >> 2];
                        const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kp][wn_], acc[wm][wn_], 0, 0, 0);
                    }
                    if (kp < CIC * RE) {
                        const int cl = kp / RE, r = kp % RE;
                        float* d = xs + (buf ^ 1) * BUF + cl * 4 * U + wr_off[r];
                        WAIT_VM(19, "+v"(sreg[cl][r]));
                        d[0] = sreg[cl][r][0];
                        d[U] = sreg[cl][r][1];
                        sreg[cl][r] = asm_ld2(gx + ((xb + ((long)(((c + 2) % nchunks) * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
                    }
                    if ((kp & 3) == 3) {
#pragma unroll
                        for (int wm = 0; wm < 2; ++wm) a[wm][kp >> 2] = asm_ld4(wn + (wm * 4 + (kp >> 2)) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int kp = 0; kp < KPC; ++kp)
#pragma unroll
                    for (int wm = 0; wm < 2; ++wm) {
                        const f32x4 q4 = a[wm][kp >> 2];
                        const float av = (kp & 3) == 0 ? q4.x : (kp & 3) == 1 ? q4.y : (kp & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int wn_ = 0; wn_ < 2; ++wn_) acc[wm][wn_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, (float)kp, acc[wm][wn_], 0, 0, 0);
                    }
            }
        }
        if (MODE == 4 || MODE == 5) {
            // loads issued by the last chunks are never consumed: their destination registers must stay reserved until they have
            // landed (the compiler does not know they are in flight and would hand the registers out again)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(sreg[0][0]), "+v"(sreg[0][1]), "+v"(sreg[0][2]), "+v"(sreg[1][0]), "+v"(sreg[1][1]), "+v"(sreg[1][2]),
                         "+v"(sreg[2][0]), "+v"(sreg[2][1]), "+v"(sreg[2][2]), "+v"(sreg[3][0]), "+v"(sreg[3][1]), "+v"(sreg[3][2]) : : "memory");
            asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
            if (MODE == 5)
                asm volatile("" : "+v"(sreg2[0][0]), "+v"(sreg2[0][1]), "+v"(sreg2[0][2]), "+v"(sreg2[1][0]), "+v"(sreg2[1][1]), "+v"(sreg2[1][2]),
                             "+v"(sreg2[2][0]), "+v"(sreg2[2][1]), "+v"(sreg2[2][2]), "+v"(sreg2[3][0]), "+v"(sreg2[3][1]), "+v"(sreg2[3][2]));
        }
        // epilogue stand-in: one store per lane per tile, depending on every accumulator
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        out[((long)tile * 256 + threadIdx.x) & 0xFFFFF] = s;
    }
}


// mode 3: the phased loop with the two waves of a SIMD taking turns on the matrix pipe (workgroup of 8 waves, an LDS turn word per SIMD):
// a wave runs its MFMA block only while it holds the turn, so the partner's loads / LDS writes fall into that block instead of
// both waves computing together and then staging together.  LOCK = 0: same 8-wave workgroup without the turn.
template <int LOCK>
__global__ __launch_bounds__(512, 2) void k_lock(const float* __restrict__ gx, const f32x4* __restrict__ gw, float* __restrict__ out,
                                                  int nchunks, int ntiles, long xmask, int* __restrict__ diag) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    __shared__ int simd_of[8];
    __shared__ volatile int turn[4];
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID
    const int simd = (hw >> 4) & 3;
    if (lane == 0) simd_of[wave] = simd;
    if (threadIdx.x < 4) turn[threadIdx.x] = 0;
    __syncthreads();
    int id = 0, mates = 0;
    for (int w = 0; w < 8; ++w) {
        if (simd_of[w] == simd) { ++mates; if (w < wave) ++id; }
    }
    const bool paired = __builtin_amdgcn_readfirstlane(mates == 2);
    if (diag && lane == 0 && blockIdx.x == 0) diag[wave] = simd | (id << 8) | (mates << 16);
    float* xs = lds_all + wave * 2 * BUF;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int rd_base = half * U + (lane & 31) + 1;
    int wr_off[RE];
    for (int r = 0; r < RE; ++r) {
        const int e = 2 * (lane + 64 * r);
        const int slot = e / 4, p = e & 3;
        wr_off[r] = e < 264 ? p * U + slot : 4 * 4 * U;
    }
    auto lane_el = [&](int r) { const int e = lane + 64 * r; return e < 132 ? e : 128; };   // 264 elements per channel window
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long xb = (((long)(tile >> 1) * 8 + wave) * 65536) & xmask;
        const f32x4* wq = gw + (long)(tile & 3) * nchunks * 8 * 64 + lane;
        f32x2 sreg[CIC][RE];
        f32x4 a[2][4];
        auto load_chunk = [&](int c) {
#pragma unroll
            for (int cl = 0; cl < CIC; ++cl)
#pragma unroll
                for (int r = 0; r < RE; ++r)
                    sreg[cl][r] = *reinterpret_cast<const f32x2*>(gx + ((xb + ((long)(c * CIC + cl) * 1024) + 2 * lane_el(r)) & xmask));
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
        load_chunk(0);
#pragma unroll
        for (int wm = 0; wm < 2; ++wm)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[wm][q] = wq[(wm * 4 + q) * 64];
        write_chunk(0);
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            const float* xr = xs + buf * BUF + rd_base;
            const f32x4* wn = wq + (long)((c + 1) % nchunks) * 8 * 64;
            load_chunk((c + 1) % nchunks);
            __builtin_amdgcn_sched_barrier(0);
            if (LOCK && paired) {
                while (turn[simd] != id) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float bv[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kp = nb * 4 + i;
                    const int off = (kp >> 2) * 4 * U + ((kp & 3) >> 1) * 2 * U + (kp & 1);
                    bv[i][0] = xr[off]; bv[i][1] = xr[off + 32];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kp = nb * 4 + i;
#pragma unroll
                    for (int wm = 0; wm < 2; ++wm) {
                        const f32x4 q4 = a[wm][kp >> 2];
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
            if (LOCK && paired) {
                if (lane == 0) turn[simd] = id ^ 1;
            }
            __builtin_amdgcn_sched_barrier(0);
            write_chunk(buf ^ 1);
            __builtin_amdgcn_wave_barrier();
        }
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        out[((long)tile * 512 + threadIdx.x) & 0xFFFFF] = s;
    }
}
template <int LOCK>
static void run_lock(int nchunks, const char* tag, const float* gx, const f32x4* gw, float* out, long xmask) {
    const int grid = 256;
    const int ntiles = grid * 24;
    const size_t lds = 8 * 2 * BUF * 4;
    hipFuncSetAttribute((const void*)k_lock<LOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int* diag; hipMalloc(&diag, 64); hipMemset(diag, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_lock<LOCK><<<grid, 512, lds>>>(gx, gw, out, nchunks, grid, xmask, diag);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_lock<LOCK><<<grid, 512, lds>>>(gx, gw, out, nchunks, ntiles, xmask, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int h[8]; hipMemcpy(h, diag, 32, hipMemcpyDeviceToHost);
    const double flops = (double)ntiles * 8 * nchunks * 64 * 4096.0;
    printf("%-12s 8-wave workgroups chunks=%d : %6.1f TFLOP/s = %.3f of 157.3 (%.3f ms)  [wave: simd/id/mates", tag, nchunks, flops / ms / 1e9, flops / ms / 1e9 / 157.3, ms);
    for (int w = 0; w < 8; ++w) printf(" %d/%d/%d", h[w] & 255, (h[w] >> 8) & 255, h[w] >> 16);
    printf("]\n");
}

template <int MODE, int ABL = 0>
static void run(int wg_per_cu, int nchunks, const char* tag, const float* gx, const f32x4* gw, float* out, long xmask) {
    const int grid = 256 * wg_per_cu;
    const int ntiles = grid * 24;            // 24 wave tiles per wave: prologue-free steady state
    const size_t lds = 4 * 2 * BUF * 4;
    hipFuncSetAttribute((const void*)k<MODE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, ABL><<<grid, 256, lds>>>(gx, gw, out, nchunks, grid, xmask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, ABL><<<grid, 256, lds>>>(gx, gw, out, nchunks, ntiles, xmask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)ntiles * 4 * nchunks * 64 * 4096.0;
    printf("%-12s waves/SIMD=%d chunks=%d : %6.1f TFLOP/s = %.3f of 157.3 (%.3f ms)\n", tag, wg_per_cu, nchunks, flops / ms / 1e9, flops / ms / 1e9 / 157.3, ms);
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool only4 = argc > 1 && argv[1][0] == '4', skip4 = argc > 1 && argv[1][0] == 'n';
    const long xfloats = 1L << 27;   // 512 MB of input: streamed
    float* gx; f32x4* gw; float* out;
    hipMalloc(&gx, xfloats * 4 + 65536); hipMemset(gx, 0, xfloats * 4 + 65536);
    hipMalloc(&gw, 4 * 64 * 8 * 64 * 16); hipMemset(gw, 0, 4 * 64 * 8 * 64 * 16);
    hipMalloc(&out, (1 << 20) * 4 + 4096);
    const long xmask = xfloats - 1;
    if (only4) {
        run<5>(1, 16, "il asm 2-ahead", gx, gw, out, xmask);
        run<5>(2, 16, "il asm 2-ahead", gx, gw, out, xmask);
        run<5>(1, 64, "il asm 2-ahead", gx, gw, out, xmask);
        run<5>(2, 64, "il asm 2-ahead", gx, gw, out, xmask);
        run<4>(1, 16, "il asm-wait", gx, gw, out, xmask);
        run<4>(2, 16, "il asm-wait", gx, gw, out, xmask);
        run<4>(1, 64, "il asm-wait", gx, gw, out, xmask);
        run<4>(2, 64, "il asm-wait", gx, gw, out, xmask);
        return 0;
    }
    for (int w = 1; w <= 2; ++w) {
        run<2>(w, 16, "mfma-only", gx, gw, out, xmask);
        run<0>(w, 16, "phased", gx, gw, out, xmask);
        run<1>(w, 16, "interleaved", gx, gw, out, xmask);
    }
    for (int w = 1; w <= 2; ++w) {
        run<0, 1>(w, 16, "ph -xload", gx, gw, out, xmask);
        run<0, 2>(w, 16, "ph -ldswrite", gx, gw, out, xmask);
        run<0, 4>(w, 16, "ph -ldsread", gx, gw, out, xmask);
        run<0, 8>(w, 16, "ph -aload", gx, gw, out, xmask);
        run<0, 3>(w, 16, "ph -x-w", gx, gw, out, xmask);
        run<0, 7>(w, 16, "ph -x-w-r", gx, gw, out, xmask);
        run<0, 15>(w, 16, "ph -all", gx, gw, out, xmask);
    }
    for (int w = 1; w <= 2; ++w) {
        run<1, 1>(w, 16, "il -xload", gx, gw, out, xmask);
        run<1, 2>(w, 16, "il -ldswrite", gx, gw, out, xmask);
        run<1, 4>(w, 16, "il -ldsread", gx, gw, out, xmask);
        run<1, 8>(w, 16, "il -aload", gx, gw, out, xmask);
        run<1, 3>(w, 16, "il -x-w", gx, gw, out, xmask);
        run<1, 11>(w, 16, "il -x-w-a", gx, gw, out, xmask);
        run<1, 15>(w, 16, "il -all", gx, gw, out, xmask);
    }
    if (!skip4) {
        run<4>(1, 16, "il asm-wait", gx, gw, out, xmask);
        run<4>(2, 16, "il asm-wait", gx, gw, out, xmask);
    }
    run_lock<0>(16, "8w no turn", gx, gw, out, xmask);
    run_lock<1>(16, "8w turns", gx, gw, out, xmask);
    run_lock<0>(64, "8w no turn", gx, gw, out, xmask);
    run_lock<1>(64, "8w turns", gx, gw, out, xmask);
    run<0>(2, 64, "phased", gx, gw, out, xmask);
    run<1>(2, 64, "interleaved", gx, gw, out, xmask);
    return 0;
}
