// Microbenchmark: per-kernel cost of a chain of dependent kernels replayed from a hipGraph on MI355X
// (the floor under a decode step built from N small kernels).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/launch_floor.bin scripts/micro/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_empty(float* p) { if (p == nullptr) __builtin_trap(); }
__global__ void __launch_bounds__(256) k_touch(float* p) {   // read one line written by the previous kernel, write one
    float v = __builtin_nontemporal_load(&p[((blockIdx.x * 7 + 3) % gridDim.x) * 256 + threadIdx.x]);
    p[blockIdx.x * 256 + threadIdx.x] = v + 1.0f;
}
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream(const f4* w, float* p, int n_per_wg) {   // stream n_per_wg float4 per thread
    const f4* q = w + (long)blockIdx.x * n_per_wg * 256 + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < n_per_wg; ++i) { f4 v = __builtin_nontemporal_load(&q[(long)i * 256]); acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) p[0] = acc;
}

// chain of HOPS dependent loads through data the previous kernel wrote (pointer chase), then one store
template <int HOPS>
__global__ void __launch_bounds__(64) k_chase(int* idx, float* p) {
    int i = blockIdx.x * 64 + threadIdx.x;
#pragma unroll
    for (int h = 0; h < HOPS; ++h) i = idx[i];
    idx[blockIdx.x * 64 + threadIdx.x] = i;   // identity permutation: rewrites the same value, keeps the line dirty
    if (i < 0) p[0] = 1.0f;
}
// scalar (s_load) read of a small state struct, then a dependent vector load, then a store
__global__ void __launch_bounds__(64) k_state(const int* __restrict__ state, const float* __restrict__ in, float* __restrict__ out) {
    const int m = state[0];
    out[blockIdx.x * 64 + threadIdx.x] = in[(blockIdx.x * 64 + threadIdx.x) * m] + 1.0f;
}

int main() {
    float* p; CK(hipMalloc(&p, 4096 * 256 * 4)); CK(hipMemset(p, 0, 4096 * 256 * 4));
    f4* w; const long wbytes = 2048L << 20; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0, wbytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 96;
    int* idx; CK(hipMalloc(&idx, 64 * 64 * 4));
    { int h[64 * 64]; for (int i = 0; i < 64 * 64; ++i) h[i] = i; CK(hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice)); }
    int* state; CK(hipMalloc(&state, 64)); { int one = 1; CK(hipMemcpy(state, &one, 4, hipMemcpyHostToDevice)); }
    for (int mode = 0; mode < 11; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) {
            switch (mode) {
                case 0: k_empty<<<1, 64, 0, st>>>(p); break;
                case 1: k_empty<<<512, 256, 0, st>>>(p); break;
                case 2: k_touch<<<512, 256, 0, st>>>(p); break;
                case 3: k_stream<<<512, 256, 0, st>>>(w + (long)(i % 16) * (8 << 20) / 16 * 16, p, 4); break;     // 8 MB per kernel
                case 4: k_stream<<<512, 256, 0, st>>>(w + (long)(i % 16) * (64 << 20) / 16, p, 32); break;        // 64 MB per kernel, 1 GB rotation
                case 5: k_stream<<<1024, 256, 0, st>>>(w + (long)(i % 16) * (64 << 20) / 16, p, 16); break;       // 64 MB, 1024 WGs
                case 6: k_chase<1><<<64, 64, 0, st>>>(idx, p); break;
                case 7: k_chase<2><<<64, 64, 0, st>>>(idx, p); break;
                case 8: k_chase<4><<<64, 64, 0, st>>>(idx, p); break;
                case 9: k_state<<<64, 64, 0, st>>>(state, p + (i & 1) * 8192, p + ((i + 1) & 1) * 8192); break;
                case 10: k_chase<1><<<2, 64, 0, st>>>(idx, p); break;
            }
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int R = 20;
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* names[] = {"empty 1x64", "empty 512x256", "touch 512x256 (dependent line)", "stream 8 MB 512x256", "stream 64 MB 512x256", "stream 64 MB 1024x256", "chase 1 hop 64x64", "chase 2 hops 64x64", "chase 4 hops 64x64", "s_load state + dependent load 64x64", "chase 1 hop 2x64"};
        printf("%-34s %.2f us per kernel\n", names[mode], ms * 1e3 / (R * N));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
