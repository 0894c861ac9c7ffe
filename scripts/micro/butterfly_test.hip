// checks the permlane-swap + DPP butterfly against a __shfl_xor reference, bit for bit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ void k(const float* in, float* out) {
    float v = in[threadIdx.x];
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[0 * 64 + threadIdx.x] = a;
    out[1 * 64 + threadIdx.x] = b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[2 * 64 + threadIdx.x] = a;
    out[3 * 64 + threadIdx.x] = b;
    out[4 * 64 + threadIdx.x] = dpp_mov<0x128>(v);
    out[5 * 64 + threadIdx.x] = dpp_mov<0x124>(v);
    out[6 * 64 + threadIdx.x] = dpp_mov<0x4E>(v);
    out[7 * 64 + threadIdx.x] = dpp_mov<0xB1>(v);
    out[8 * 64 + threadIdx.x] = dpp_mov<0x141>(v);
}
int main() {
    float h[64], *din, *dout, ho[9 * 64];
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    (void)hipMalloc(&din, 256); (void)hipMalloc(&dout, sizeof(ho)); (void)hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout); (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    const char* names[] = {"swap32.x", "swap32.y", "swap16.x", "swap16.y", "ror8", "ror4", "xor2", "xor1", "half_mirror"};
    for (int r = 0; r < 9; ++r) { printf("%-12s", names[r]); for (int i = 0; i < 64; ++i) printf("%d ", (int)ho[r * 64 + i]); printf("\n"); }
    return 0;
}
