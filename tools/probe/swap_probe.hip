#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__global__ void k(float* out) {
    int lane = threadIdx.x;
    float a = (float)lane, b = 100.0f + lane;
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out[lane] = __builtin_bit_cast(float, r[0]);
    out[64 + lane] = __builtin_bit_cast(float, r[1]);
    auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out[128 + lane] = __builtin_bit_cast(float, q[0]);
    out[192 + lane] = __builtin_bit_cast(float, q[1]);
    out[256 + lane] = dpp_f<0x128>(a);
    out[320 + lane] = dpp_f<0x124>(a);
}
int main() {
    float* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"p32.r0", "p32.r1", "p16.r0", "p16.r1", "ror8", "ror4"};
    for (int s = 0; s < 6; ++s) { printf("%s:", names[s]); for (int i = 0; i < 64; ++i) printf(" %g", h[s * 64 + i]); printf("\n"); }
    return 0;
}
