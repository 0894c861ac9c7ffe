// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this chip, by waves/SIMD and accumulators?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int LDSR>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = b0 + i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const float* p = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float bb = b;
            if (LDSR) bb = p[(u * 64 + it) & 4032];
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int LDSR>
void run(int wg_per_cu, const char* tag) {
    float* out; hipMalloc(&out, 256 * 64 * 256 * 4);
    int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDSR><<<grid, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDSR><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    printf("%s nacc=%d lds=%d waves/SIMD=%d : %.1f TFLOP/s (%.2f ms)\n", tag, NACC, LDSR, wg_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    run<1, 0>(1, "pure"); run<4, 0>(1, "pure"); run<4, 0>(2, "pure"); run<4, 0>(4, "pure");
    run<4, 1>(1, "lds "); run<4, 1>(2, "lds "); run<2, 1>(2, "lds "); run<2, 1>(4, "lds ");
    return 0;
}
