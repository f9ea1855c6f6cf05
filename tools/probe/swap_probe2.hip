#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__global__ void k(float* out) {
    int lane = threadIdx.x;
    float a = (float)lane, b = 100.0f + lane;
    float a2 = a, b2 = b;
    swap32(a2, b2);
    out[lane] = a2; out[64 + lane] = b2;
    a2 = a; b2 = b;
    swap16(a2, b2);
    out[128 + lane] = a2; out[192 + lane] = b2;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"p32.a", "p32.b", "p16.a", "p16.b"};
    for (int s = 0; s < 4; ++s) { printf("%s:", names[s]); for (int i = 0; i < 64; ++i) printf(" %g", h[s * 64 + i]); printf("\n"); }
    return 0;
}
