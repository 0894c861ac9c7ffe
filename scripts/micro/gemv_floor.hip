// Where does the fixed cost of a decode GEMV launch go?  Chains of the SAME kernel (o_proj shape: N = 2048, K = 2048, M = 2, EPI 3) in
// a hipGraph, timed with events:
//   stream   every launch reads a different 8.4 MB weight matrix (32 matrices, 268 MB: past the 256 MB Infinity Cache)
//   resident every launch reads the same matrix (L2 / Infinity Cache resident)
//   tiny     N = 64 rows (16 workgroups): launch + prologue + tail only
//   empty    an empty kernel on the same grid
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o /tmp/gemv_floor scripts/micro/gemv_floor.hip
#include "../../realtime_codec_agent_amd/csrc/rca_lm.hip"
#include <cstdio>
#include <vector>
namespace rca { thread_local char g_err[512] = {0}; }
__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 0; }
template <typename F> static float time_graph(hipStream_t st, int reps, F&& enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    enqueue();
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / reps;
}
int main() {
    const int N = 2048, K = 2048, NM = 32, CH = 64;
    bf16_t* w; float *x, *y; LmDevState* stt;
    hipMalloc(&w, (size_t)NM * N * K * 2); hipMemset(w, 0x3c, (size_t)NM * N * K * 2);
    hipMalloc(&x, 2 * K * 4); hipMemset(x, 0, 2 * K * 4); hipMalloc(&y, 2 * N * 4); hipMemset(y, 0, 2 * N * 4);
    hipMalloc(&stt, sizeof(LmDevState)); hipMemset(stt, 0, sizeof(LmDevState));
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const GemvPro nopro{nullptr, nullptr, 0.0f, 0};
    const GemvPro pro{x, x, 1e-5f, 0};
    const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    const GemvQ8 noq{nullptr, nullptr};
    auto chain = [&](int mode) {
        for (int i = 0; i < CH; ++i) {
            const bf16_t* wi = w + (size_t)(mode == 0 ? i % NM : 0) * N * K;
            if (mode == 3) empty_kernel<<<N / 4, 256, 0, st>>>(y);
            else if (mode == 2) lm_gemv_kernel<2, 1, 4, 0, 3, 0><<<64 / 4, 256, 0, st>>>(stt, wi, x, y, 64, K, 1, N, nopro, norope, noq);
            else if (mode == 4) lm_gemv_kernel<2, 1, 4, 1, 3, 0><<<N / 4, 256, 0, st>>>(stt, wi, x, y, N, K, 1, N, pro, norope, noq);
            else if (mode == 5) lm_gemv_kernel<2, 1, 16, 0, 3, 0><<<N / 16, 256, 0, st>>>(stt, wi, x, y, N, K, 1, N, nopro, norope, noq);
            else lm_gemv_kernel<2, 1, 4, 0, 3, 0><<<N / 4, 256, 0, st>>>(stt, wi, x, y, N, K, 1, N, nopro, norope, noq);
        }
    };
    const char* names[] = {"stream   (8.4 MB from HBM each launch, R=4, 512 WGs)", "resident (same 8.4 MB every launch)", "tiny     (N = 64: 16 workgroups)",
                           "empty    (empty kernel, 512 WGs)", "stream + RMSNorm prologue (PRO=1)", "stream, R=16 (128 WGs)"};
    for (int mode : {0, 1, 2, 3, 4, 5}) {
        const float us = time_graph(st, 20, [&] { chain(mode == 4 || mode == 5 ? 0 : mode); if (false) chain(0); });
        (void)us;
    }
    for (int mode = 0; mode < 6; ++mode) {
        float us;
        if (mode == 4) us = time_graph(st, 20, [&] { for (int i = 0; i < CH; ++i) lm_gemv_kernel<2, 1, 4, 1, 3, 0><<<N / 4, 256, 0, st>>>(stt, w + (size_t)(i % NM) * N * K, x, y, N, K, 1, N, pro, norope, noq); });
        else if (mode == 5) us = time_graph(st, 20, [&] { for (int i = 0; i < CH; ++i) lm_gemv_kernel<2, 1, 16, 0, 3, 0><<<N / 16, 256, 0, st>>>(stt, w + (size_t)(i % NM) * N * K, x, y, N, K, 1, N, nopro, norope, noq); });
        else us = time_graph(st, 20, [&] { chain(mode); });
        printf("%-58s %6.2f us per launch\n", names[mode], us / CH);
    }
    return 0;
}
