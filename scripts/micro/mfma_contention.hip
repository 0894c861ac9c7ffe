// Micro-benchmark: how much matrix-pipe time does ONE other instruction cost when it is issued into a stream of f32 MFMAs?
// A group is 4 independent v_mfma_f32_32x32x2_f32 (256 pipe cycles) followed by NPER copies of the probed instruction; the loss is
// reported in pipe cycles per probed instruction, relative to the same stream without it, at 1 and 2 waves per SIMD, with the
// accumulators in VGPRs and in AGPRs.  (What the conv kernel's chunk loop is made of: ds_read2_b32, ds_write2_b32, buffer loads.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { NONE, SALU, VMOV, VFMA, DSR32, DSR2_32, DSR128, DSW32, DSW2_32, DSW128, GLD32, GLD128, NKIND };
static const char* kname[] = {"none", "s_add_u32", "v_mov_b32", "v_fma_f32", "ds_read_b32", "ds_read2_b32", "ds_read_b128",
                              "ds_write_b32", "ds_write2_b32", "ds_write_b128", "global_load_dword", "global_load_dwordx4"};

template <int KIND>
__device__ __forceinline__ void probe(unsigned lds_addr, const float* gp, float& v0, f32x4& v4, f32x2& v2) {
    if (KIND == SALU) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
    if (KIND == VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "v"(v2[0]));
    if (KIND == VFMA) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v0) : "v"(v2[0]), "v"(v2[1]), "v"(v4[0]));
    if (KIND == DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(lds_addr));
    if (KIND == DSR2_32) asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(v2) : "v"(lds_addr));
    if (KIND == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(lds_addr));
    if (KIND == DSW32) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr), "v"(v0));
    if (KIND == DSW2_32) asm volatile("ds_write2_b32 %0, %1, %2 offset1:72" ::"v"(lds_addr), "v"(v2[0]), "v"(v2[1]));
    if (KIND == DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v4));
    if (KIND == GLD32) asm volatile("global_load_dword %0, %1, %2" : "=v"(v0) : "v"(lds_addr), "s"(gp));
    if (KIND == GLD128) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v4) : "v"(lds_addr), "s"(gp));
}

template <int KIND, int NPER, int AGPR>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ gp, float* __restrict__ out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a0 + i;
    __syncthreads();
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const float a = a0 + threadIdx.x, b = a0 * 2.f;
    // per-lane 16-byte-aligned LDS byte address (conflict-free for every width), per-wave region
    const unsigned lds_addr = (unsigned)(size_t)(lds) + (threadIdx.x >> 6) * 8192 + (threadIdx.x & 63) * 16;
    const unsigned g_off = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    float v0 = a;
    f32x4 v4 = {a, b, a, b};
    f32x2 v2 = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int p = 0; p < NPER; ++p) probe<KIND>(KIND >= GLD32 ? g_off : lds_addr, gp, v0, v4, v2);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = v0 + v4[0] + v4[1] + v4[2] + v4[3] + v2[0] + v2[1];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_in; static float* g_out;
template <int KIND, int NPER, int AGPR>
static double run(int wg_per_cu) {
    const int iters = 1500, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NPER, AGPR><<<grid, 256>>>(g_in, g_out, 10, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND, NPER, AGPR><<<grid, 256>>>(g_in, g_out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / wg_per_cu;   // time per wave-round
}
template <int KIND, int NPER>
static void row(const double base[2][2]) {
    printf("%-20s x%d :", kname[KIND], NPER);
    for (int ag = 0; ag < 2; ++ag)
        for (int w = 1; w <= 2; ++w) {
            const double t = ag ? run<KIND, NPER, 1>(w) : run<KIND, NPER, 0>(w);
            // pipe cycles lost per probed instruction (a group of 4 MFMAs = 256 cycles at the no-probe rate)
            printf("  %s w%d %6.1f", ag ? "agpr" : "vgpr", w, (t / base[ag][w - 1] - 1.0) * 256.0 / NPER);
        }
    printf("\n");
}
int main() {
    hipMalloc(&g_in, 1 << 20); hipMemset(g_in, 0, 1 << 20);
    hipMalloc(&g_out, 512 * 256 * 4);
    double base[2][2];
    for (int w = 1; w <= 2; ++w) { base[0][w - 1] = run<NONE, 1, 0>(w); base[1][w - 1] = run<NONE, 1, 1>(w); }
    const double fl = 256.0 * 4 * 1500 * 8 * 4 * 4096.0;
    printf("base: vgpr w1 %.1f TF, w2 %.1f TF; agpr w1 %.1f TF, w2 %.1f TF   (columns below: pipe cycles lost per probed instruction)\n",
           fl / base[0][0] / 1e9, fl / base[0][1] / 1e9, fl / base[1][0] / 1e9, fl / base[1][1] / 1e9);
    row<SALU, 2>(base); row<VMOV, 1>(base); row<VMOV, 4>(base); row<VFMA, 2>(base);
    row<DSR32, 1>(base); row<DSR32, 4>(base); row<DSR2_32, 2>(base); row<DSR128, 1>(base); row<DSR128, 2>(base);
    row<DSW32, 2>(base); row<DSW2_32, 2>(base); row<DSW128, 1>(base); row<DSW128, 2>(base);
    row<GLD32, 2>(base); row<GLD128, 1>(base); row<GLD128, 2>(base);
    return 0;
}
