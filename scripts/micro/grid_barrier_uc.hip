// Microbenchmark: grid barrier for a persistent decode kernel on MI355X with the exchanged data and the flags in
// UNCACHED device memory (hipDeviceMallocUncached): stores complete at memory, loads bypass the per-XCD L2, so no
// agent-scope release/acquire (L2 write-back + invalidate) is needed -- a workgroup-scope fence (s_waitcnt) orders the
// stores before a RELAXED agent-scope atomic.  Variants: one counter; per-XCD counters + one global (arrival tree).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/grid_barrier_uc.bin scripts/micro/grid_barrier_uc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_relaxed(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// cnt[0] global, cnt[16 + 16 * x] per-XCD (separate lines)
template <int TREE>
__device__ __forceinline__ bool barrier(unsigned* cnt, unsigned epoch, int G, unsigned* abort_flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores have been issued and acknowledged (vmcnt 0)
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (TREE) {
            const int x = blockIdx.x & 7;
            const unsigned per = (unsigned)((G + 7 - x) / 8);               // workgroups with this id mod 8
            const unsigned old = add_relaxed(cnt + 16 + 16 * x, 1u);
            if (old + 1u == epoch * per) add_relaxed(cnt, 1u);               // last arrival of the XCD group
            long spins = 0;
            while (ld_relaxed(cnt) < epoch * 8u) { __builtin_amdgcn_s_sleep(1); if (++spins > 20000000L) { *abort_flag = 1; ok = false; break; } }
        } else {
            add_relaxed(cnt, 1u);
            long spins = 0;
            while (ld_relaxed(cnt) < epoch * (unsigned)G) { __builtin_amdgcn_s_sleep(1); if (++spins > 20000000L) { *abort_flag = 1; ok = false; break; } }
        }
    }
    __syncthreads();
    return ok;
}

template <int TREE>
__global__ void __launch_bounds__(256) bar_kernel(unsigned* cnt, unsigned* abort_flag, float* buf0, float* buf1, int iters, unsigned* errs) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    float* dst = buf1; float* src = buf0;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        dst[wg * 256 + tid] = (float)(it * 7 + wg);                           // uncached store
        if (!barrier<TREE>(cnt, (unsigned)(it + 1), G, abort_flag)) return;
        const int o = (wg + it + 1) % G;
        const float v = dst[o * 256 + tid];                                    // uncached load: written by another workgroup / XCD
        if (v != (float)(it * 7 + o)) ++bad;
        float* t = src; src = dst; dst = t;
    }
    if (bad) atomicAdd(errs, bad);
}

int main() {
    const int iters = 2000;
    unsigned *cnt, *abort_flag, *errs; float *b0, *b1;
    CK(hipExtMallocWithFlags((void**)&cnt, 4096, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&b0, 1024 * 256 * 4, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&b1, 1024 * 256 * 4, hipDeviceMallocUncached));
    CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&errs, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int tree = 0; tree < 2; ++tree)
        for (int G : {128, 256, 512}) {
            float best = 1e9; unsigned herr = 0, habort = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(errs, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (tree) bar_kernel<1><<<G, 256>>>(cnt, abort_flag, b0, b1, iters, errs);
                else bar_kernel<0><<<G, 256>>>(cnt, abort_flag, b0, b1, iters, errs);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                CK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&habort, abort_flag, 4, hipMemcpyDeviceToHost));
                if (habort) break;
            }
            printf("uncached, %s, G=%d: %.3f us per barrier + exchange (errors %u, abort %u)\n", tree ? "per-XCD tree" : "one counter ", G, best * 1e3 / iters, herr, habort);
            if (habort) return 2;
        }
    return 0;
}
