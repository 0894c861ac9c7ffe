// Micro-benchmark: does the VGPR bank of the A / B operands of v_mfma_f32_32x32x2_f32 matter?  (bank = register index mod 4)
// Streams of 4 independent MFMAs with hard-coded operand registers: A and B in the same bank, in different banks, B shared by all.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, A, B) "v_mfma_f32_32x32x2_f32 %" #acc ", " A ", " B ", %" #acc "\n"
template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0) {
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    asm volatile("v_mov_b32 v200, %0\n v_mov_b32 v201, %0\n v_mov_b32 v202, %0\n v_mov_b32 v203, %0\n v_mov_b32 v204, %0\n v_mov_b32 v205, %0\n v_mov_b32 v206, %0\n v_mov_b32 v207, %0\n"
                 "v_mov_b32 v208, %0\n v_mov_b32 v209, %0\n v_mov_b32 v210, %0\n v_mov_b32 v211, %0\n s_nop 4\n"
                 :: "v"(a0 + threadIdx.x) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (V == 0)        // A, B in different banks, every MFMA its own pair
                asm volatile(MF(0, "v200", "v205") MF(1, "v201", "v206") MF(2, "v202", "v207") MF(3, "v203", "v204")
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            else if (V == 1)   // A, B in the same bank
                asm volatile(MF(0, "v200", "v204") MF(1, "v201", "v205") MF(2, "v202", "v206") MF(3, "v203", "v207")
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            else if (V == 2)   // one B for all, A rotating through the banks
                asm volatile(MF(0, "v200", "v204") MF(1, "v201", "v204") MF(2, "v202", "v204") MF(3, "v203", "v204")
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            else if (V == 3)   // the conv kernel's shape: 2 A x 2 B per k pair, A from one quad (consecutive registers), B from a pair
                asm volatile(MF(0, "v200", "v208") MF(1, "v200", "v209") MF(2, "v204", "v208") MF(3, "v204", "v209")
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            else               // same shape, A and B chosen in distinct banks
                asm volatile(MF(0, "v200", "v209") MF(1, "v200", "v210") MF(2, "v204", "v209") MF(3, "v204", "v210")
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0;
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V>
static void run(int wg_per_cu, const char* tag, float* out) {
    const int iters = 1500, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V><<<grid, 256>>>(out, 10, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V><<<grid, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 8 * 4 * 4096.0;
    printf("%-44s waves/SIMD=%d : %6.1f TFLOP/s\n", tag, wg_per_cu, fl / ms / 1e9);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; (void)hipMalloc(&out, 512 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 1; w <= 2; ++w) {
            run<0>(w, "A,B different banks", out);
            run<1>(w, "A,B same bank", out);
            run<2>(w, "one B register for all", out);
            run<3>(w, "2x2 tile, A v200/v204 B v208/v209", out);
            run<4>(w, "2x2 tile, A v200/v204 B v209/v210", out);
        }
    return 0;
}
