// Microbenchmark: does the shape of a hipGraph change the per-kernel cost of a dependent kernel chain on MI355X?
//  A: 96 kernels only            B: pinned H2D memcpy node + 96 kernels + pinned D2H memcpy node (the LM step graph)
//  C: like B but the kernels alternate between two functions with different kernarg sizes
//  D: kernels read the pinned host buffer directly (no memcpy nodes)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/graph_fence.bin scripts/micro/graph_fence.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct Big { const float* a; const float* b; const float* c; int n[24]; };
__global__ void __launch_bounds__(256) k_touch(const int* __restrict__ state, float* p) {
    float v = __builtin_nontemporal_load(&p[((blockIdx.x * 7 + 3) % gridDim.x) * 256 + threadIdx.x]);
    p[blockIdx.x * 256 + threadIdx.x] = v + (float)state[0];
}
__global__ void __launch_bounds__(64) k_small(const int* __restrict__ state, float* p, Big big) {
    if (state[0] > big.n[3] + 1000) return;
    p[blockIdx.x * 64 + threadIdx.x] += 1.0f;
}
int main() {
    float* p; CK(hipMalloc(&p, 4096 * 256 * 4)); CK(hipMemset(p, 0, 4096 * 256 * 4));
    int* dstate; CK(hipMalloc(&dstate, 256)); CK(hipMemset(dstate, 0, 256));
    int* hstate; CK(hipHostMalloc(&hstate, 256, hipHostMallocDefault)); hstate[0] = 1;
    int* hstate_dev; CK(hipHostGetDevicePointer((void**)&hstate_dev, hstate, 0));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Big big{}; 
    const char* names[] = {"A kernels only", "B pinned H2D + kernels + pinned D2H", "C like B, two alternating kernels", "D kernels read pinned host memory, no memcpy nodes", "E like B, sync after every graph launch"};
    for (int mode = 0; mode < 5; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (mode == 1 || mode == 2 || mode == 4) CK(hipMemcpyAsync(dstate, hstate, 136, hipMemcpyHostToDevice, st));
        for (int i = 0; i < 96; ++i) {
            const int* s = mode == 3 ? hstate_dev : dstate;
            if (mode == 2 && (i & 1)) k_small<<<64, 64, 0, st>>>(s, p, big);
            else k_touch<<<512, 256, 0, st>>>(s, p);
        }
        if (mode == 1 || mode == 2 || mode == 4) CK(hipMemcpyAsync(hstate + 32, dstate + 32, 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        const int R = 20;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < R; ++r) { CK(hipGraphLaunch(ge, st)); if (mode == 4) CK(hipStreamSynchronize(st)); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s %.2f us per kernel (%.1f us per graph)\n", names[mode], ms * 1e3 / (R * 96), ms * 1e3 / R);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
