// Microbenchmark: cost of a software grid barrier (one atomic counter, agent scope) on MI355X for a persistent
// kernel of G workgroups x 256 threads, with a data exchange between phases to check visibility.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier scripts/micro/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* abort_flag) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default for global atomics
        long spins = 0;
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000L) { *abort_flag = 1; ok = false; break; }   // exit condition every wave reaches
        }
    }
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(256) bar_kernel(unsigned* counter, unsigned* abort_flag, float* buf0, float* buf1, int iters, int exchange, unsigned* errs) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    float* src = buf0; float* dst = buf1;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (exchange) {
            // phase: every WG writes 256 floats; after the barrier reads what WG (wg+it+1)%G wrote
            dst[wg * 256 + tid] = (float)(it * 7 + wg);
        }
        if (!grid_barrier(counter, (unsigned)(it + 1) * G, abort_flag)) return;
        if (exchange) {
            const int o = (wg + it + 1) % G;
            float v = __builtin_nontemporal_load(&dst[o * 256 + tid]);
            if (v != (float)(it * 7 + o)) ++bad;
            float* t = src; src = dst; dst = t;
        }
    }
    if (bad) atomicAdd(errs, bad);
}

int main(int argc, char** argv) {
    int iters = 2000;
    unsigned *counter, *abort_flag, *errs; float *b0, *b1;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&errs, 4));
    CK(hipMalloc(&b0, 1024 * 256 * 4)); CK(hipMalloc(&b1, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int exchange = 0; exchange < 2; ++exchange)
        for (int G : {64, 128, 256, 512}) {
            float best = 1e9;
            unsigned herr = 0, habort = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(counter, 0, 4)); CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(errs, 0, 4));
                CK(hipEventRecord(e0));
                bar_kernel<<<G, 256>>>(counter, abort_flag, b0, b1, iters, exchange, errs);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                CK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&habort, abort_flag, 4, hipMemcpyDeviceToHost));
                if (habort) break;
            }
            printf("G=%d exchange=%d: %.3f us per barrier  (errors %u, abort %u)\n", G, exchange, best * 1e3 / iters, herr, habort);
            if (habort) return 2;
        }
    return 0;
}
