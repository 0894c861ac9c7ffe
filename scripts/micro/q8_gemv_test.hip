// Stand-alone check of the packed q8_0 GEMV against a host computation (debug helper).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o /tmp/q8t scripts/micro/q8_gemv_test.hip
#include "../../realtime_codec_agent_amd/csrc/rca_lm.hip"
#include <cstdio>
#include <vector>
namespace rca { thread_local char g_err[512] = {0}; }
int main() {
    const int N = 64, K = 512;
    std::vector<bf16_t> w((size_t)N * K);
    std::vector<float> x(K), wf((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (size_t i = 0; i < w.size(); ++i) { float v = 0.1f * rnd(); unsigned u; memcpy(&u, &v, 4); w[i] = (bf16_t)(u >> 16); unsigned uu = (unsigned)w[i] << 16; memcpy(&wf[i], &uu, 4); }
    for (int k = 0; k < K; ++k) x[k] = rnd();
    bf16_t* dw; float *dx, *dy; signed char* dq; f16_t* dd; u32x4* qs; unsigned* sc; LmDevState* stt;
    hipMalloc(&dw, w.size() * 2); hipMalloc(&dx, K * 4); hipMalloc(&dy, N * 4); hipMalloc(&dq, w.size()); hipMalloc(&dd, w.size() / 32 * 2);
    hipMalloc(&qs, (size_t)(N / 2) * (K / 8) * 16); hipMalloc(&sc, (size_t)(K / 32) * (N / 2) * 4 + 256); hipMalloc(&stt, sizeof(LmDevState));
    hipMemset(stt, 0, sizeof(LmDevState)); hipMemset(sc, 0, (size_t)(K / 32) * (N / 2) * 4 + 256);
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), K * 4, hipMemcpyHostToDevice);
    lm_q8_quantize_kernel<<<64, 256>>>(dw, (long)w.size() / 32, dq, dd);
    lm_q8_pack_kernel<<<64, 256>>>(dq, dd, N, K, 0, qs, sc);
    std::vector<signed char> hq(w.size()); std::vector<unsigned short> hd(w.size() / 32);
    hipMemcpy(hq.data(), dq, hq.size(), hipMemcpyDeviceToHost); hipMemcpy(hd.data(), dd, hd.size() * 2, hipMemcpyDeviceToHost);
    const GemvPro nopro{nullptr, nullptr, 0.0f, 0};
    const GemvRope norope{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    for (int pass = 0; pass < 3; ++pass) {
        hipMemset(dy, 0, N * 4);
        if (pass == 0) lm_gemv_kernel<1, 1, 4, 0, 0, 0><<<N / 4, 256>>>(stt, dw, dx, dy, N, K, 1, N, nopro, norope, GemvQ8{nullptr, nullptr});
        if (pass == 1) lm_gemv_kernel<1, 1, 4, 0, 0, 1><<<N / 4, 256>>>(stt, dw, dx, dy, N, K, 1, N, nopro, norope, GemvQ8{qs, sc});
        if (pass == 2) lm_gemv_kernel<1, 1, 8, 0, 0, 1><<<N / 8, 256>>>(stt, dw, dx, dy, N, K, 1, N, nopro, norope, GemvQ8{qs, sc});
        std::vector<float> y(N);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(y.data(), dy, N * 4, hipMemcpyDeviceToHost);
        double maxd = 0;
        for (int n = 0; n < N; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) {
                float wv = wf[(size_t)n * K + k];
                if (pass) { _Float16 dh; memcpy(&dh, &hd[((size_t)n * K + k) / 32], 2); wv = (float)dh * (float)hq[(size_t)n * K + k]; }
                ref += (double)wv * x[k];
            }
            maxd = std::max(maxd, std::fabs(ref - y[n]));
            if (n < 3) printf("  pass %d row %d: got %g want %g\n", pass, n, y[n], ref);
        }
        printf("pass %d (%s): max|d| = %g  (%s)\n", pass, pass ? "q8_0" : "bf16", maxd, hipGetErrorString(e));
    }
    printf("q[0..7] = %d %d %d %d %d %d %d %d  d[0] bits = %04x\n", hq[0], hq[1], hq[2], hq[3], hq[4], hq[5], hq[6], hq[7], hd[0]);
    return 0;
}
