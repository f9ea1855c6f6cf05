// rocPRIM radix_sort_pairs timing: 64-bit vs 32-bit keys, 1 M and 8 M items (level-1 depth sort sizes).
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <vector>
template <typename K>
static void run(size_t n, unsigned bits, const char* name) {
    K *a, *b; int *va, *vb;
    hipMalloc(&a, n * sizeof(K)); hipMalloc(&b, n * sizeof(K)); hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4);
    std::vector<K> h(n); unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (K)(x & ((bits >= 64 ? ~0ull : ((1ull << bits) - 1)))); }
    hipMemcpy(a, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
    size_t tmp = 0; rocprim::radix_sort_pairs(nullptr, tmp, a, b, va, vb, n, 0u, bits, 0);
    void* t; hipMalloc(&t, tmp);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rocprim::radix_sort_pairs(t, tmp, a, b, va, vb, n, 0u, bits, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) rocprim::radix_sort_pairs(t, tmp, a, b, va, vb, n, 0u, bits, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s n=%8zu bits=%2u  %.3f ms\n", name, n, bits, ms / 10);
    hipFree(a); hipFree(b); hipFree(va); hipFree(vb); hipFree(t);
}
int main() {
    run<unsigned>(26400000, 16, "u32 keys"); run<unsigned short>(26400000, 16, "u16 keys");
    run<unsigned>(3300000, 13, "u32 keys"); run<unsigned short>(3300000, 13, "u16 keys");
    run<unsigned>(8000000, 32, "u32 keys"); run<unsigned>(8000000, 24, "u32 keys");
    return 0;
}
