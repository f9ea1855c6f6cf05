// rocPRIM radix_sort_pairs with explicit onesweep configurations (radix bits per pass, items per thread) on the two
// sort shapes of the train step: 8 M pairs with 32-bit keys (camera | depth) and 26 M pairs with 16-bit tile keys.
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <vector>
template <typename Cfg>
static void run(size_t n, unsigned bits, const char* name) {
    unsigned *a, *b; int *va, *vb;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4);
    std::vector<unsigned> h(n); unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (unsigned)(x & ((1ull << bits) - 1)); }
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp = 0; rocprim::radix_sort_pairs<Cfg>(nullptr, tmp, a, b, va, vb, n, 0u, bits, 0);
    void* t; hipMalloc(&t, tmp);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rocprim::radix_sort_pairs<Cfg>(t, tmp, a, b, va, vb, n, 0u, bits, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) rocprim::radix_sort_pairs<Cfg>(t, tmp, a, b, va, vb, n, 0u, bits, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s n=%8zu bits=%2u  %.3f ms\n", name, n, bits, ms / 10);
    hipFree(a); hipFree(b); hipFree(va); hipFree(vb); hipFree(t);
}
template <unsigned RB, unsigned BS, unsigned IPT>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>,
                                                                           rocprim::kernel_config<BS, IPT>, RB,
                                                                           rocprim::block_radix_rank_algorithm::match>, 0>;
int main() {
    run<rocprim::default_config>(8000000, 32, "default");
    run<cfg<8, 512, 12>>(8000000, 32, "8 bits 512x12");
    run<cfg<8, 1024, 8>>(8000000, 32, "8 bits 1024x8");
    run<cfg<11, 512, 12>>(8000000, 32, "11 bits 512x12");
    run<cfg<11, 1024, 8>>(8000000, 32, "11 bits 1024x8");
    run<cfg<11, 256, 16>>(8000000, 32, "11 bits 256x16");
    run<rocprim::default_config>(26000000, 16, "default");
    run<cfg<8, 512, 12>>(26000000, 16, "8 bits 512x12");
    run<cfg<8, 1024, 8>>(26000000, 16, "8 bits 1024x8");
    run<cfg<8, 512, 18>>(26000000, 16, "8 bits 512x18");
    run<cfg<8, 256, 22>>(26000000, 16, "8 bits 256x22");
    run<cfg<6, 512, 12>>(26000000, 17, "6 bits 512x12 (3 passes)");
    run<rocprim::default_config>(1048577, 32, "default 1M");
    run<cfg<8, 512, 12>>(1000000, 32, "8 bits 512x12 1M (no merge sort)");
    run<cfg<11, 512, 12>>(1000000, 32, "11 bits 512x12 1M");
    return 0;
}
