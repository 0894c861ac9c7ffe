// Microbenchmark: what does alternating between different kernels cost on MI355X?  Chains of 96 dependent kernels in
// a hipGraph; each kernel runs STRAIGHT-LINE code of N unrolled FMAs (N*8 bytes of code, executed once per wave).
// "same" repeats one kernel (instruction cache warm), "alt6" cycles through six distinct instantiations.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/icache_switch.bin scripts/micro/icache_switch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ID, int N>
__global__ void __launch_bounds__(256) k_code(float* p) {
    float a = p[threadIdx.x], b = (float)ID + 0.5f;
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_fmaf(a, b, (float)(i * 7 + ID));   // distinct literals: no code sharing
    p[blockIdx.x * 256 + threadIdx.x] = a;
}
template <int N> void enqueue(int id, int grid, float* p, hipStream_t st) {
    switch (id) {
        case 0: k_code<0, N><<<grid, 256, 0, st>>>(p); break;
        case 1: k_code<1, N><<<grid, 256, 0, st>>>(p); break;
        case 2: k_code<2, N><<<grid, 256, 0, st>>>(p); break;
        case 3: k_code<3, N><<<grid, 256, 0, st>>>(p); break;
        case 4: k_code<4, N><<<grid, 256, 0, st>>>(p); break;
        default: k_code<5, N><<<grid, 256, 0, st>>>(p); break;
    }
}
template <int N> void run(int grid, bool alt, float* p, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 96; ++i) enqueue<N>(alt ? i % 6 : 0, grid, p, st);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("code %5d B  grid %4d  %-5s %.2f us per kernel\n", N * 8, grid, alt ? "alt6" : "same", ms * 1e3 / (20 * 96));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}
int main() {
    float* p; CK(hipMalloc(&p, 4096 * 256 * 4)); CK(hipMemset(p, 0, 4096 * 256 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {64, 512}) {
        for (int alt = 0; alt < 2; ++alt) {
            run<256>(grid, alt, p, st, e0, e1);
            run<1024>(grid, alt, p, st, e0, e1);
            run<4096>(grid, alt, p, st, e0, e1);
        }
    }
    return 0;
}
