// Validates radix_sort.hip against std::stable_sort and times it next to rocPRIM on the sort shapes of the train step.
//   cd tools/probe && hipcc --offload-arch=gfx950 -O3 -std=c++17 rsort_probe.hip -L../../starst3r_amd -lst3r_hip \
//         -Wl,-rpath,'$ORIGIN/../../starst3r_amd' -o rsort_probe
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include <string.h>
#include "../../include/st3r.h"

template <typename K>
static int sort_call(st3r_ctx* c, int64_t n, int eb, const K* a, const int32_t* va, K* b, int32_t* vb) {
    return st3r_radix_sort_pairs(c, nullptr, (int)sizeof(K), n, 0, eb, a, va, b, vb);
}

template <typename K>
static bool run(st3r_ctx* ctx, size_t n, int bits, int mode, bool check, bool time_rocprim) {
    K *a, *b; int *va, *vb;
    hipMalloc(&a, n * sizeof(K)); hipMalloc(&b, n * sizeof(K)); hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4);
    std::vector<K> h(n); std::vector<int> hv(n);
    unsigned long long x = 88172645463325252ull + n;
    const K mask = bits >= (int)sizeof(K) * 8 ? ~K(0) : ((K(1) << bits) - 1);
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        K k = (K)x & mask;
        if (mode == 1) k = (K)((i / 37) % 65280) & mask;         // tile-key like: long runs of close values
        if (mode == 2) k = (K)(x % 7) & mask;                    // very few distinct keys
        h[i] = k; hv[i] = (int)i;
    }
    hipMemcpy(a, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
    hipMemcpy(va, hv.data(), n * 4, hipMemcpyHostToDevice);
    int rc = sort_call<K>(ctx, (int64_t)n, bits, a, va, b, vb);
    if (rc) { printf("sort failed: %s\n", st3r_last_error()); return false; }
    hipDeviceSynchronize();
    bool ok = true;
    if (check) {
        std::vector<K> o(n); std::vector<int> ov(n);
        hipMemcpy(o.data(), b, n * sizeof(K), hipMemcpyDeviceToHost);
        hipMemcpy(ov.data(), vb, n * 4, hipMemcpyDeviceToHost);
        std::vector<int> idx(n); std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int p, int q) { return h[p] < h[q]; });
        for (size_t i = 0; i < n; ++i)
            if (o[i] != h[idx[i]] || ov[i] != idx[i]) { printf("  MISMATCH at %zu\n", i); ok = false; break; }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, ms2 = 0;
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) sort_call<K>(ctx, (int64_t)n, bits, a, va, b, vb);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (time_rocprim) {
        size_t tmp = 0; rocprim::radix_sort_pairs(nullptr, tmp, a, b, va, vb, n, 0u, (unsigned)bits, 0);
        void* t; hipMalloc(&t, tmp);
        rocprim::radix_sort_pairs(t, tmp, a, b, va, vb, n, 0u, (unsigned)bits, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) rocprim::radix_sort_pairs(t, tmp, a, b, va, vb, n, 0u, (unsigned)bits, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        hipFree(t);
    }
    const double gb = (double)n * (sizeof(K) + 4) * 2 * ((bits + 7) / 8) / 1e9;
    printf("K=%zu n=%9zu bits=%2d mode=%d  %s  own %.3f ms (%.2f TB/s)  rocprim %.3f ms\n", sizeof(K), n, bits, mode,
           check ? (ok ? "OK  " : "FAIL") : "    ", ms / 10, gb / (ms / 10) , ms2 / 10);
    hipFree(a); hipFree(b); hipFree(va); hipFree(vb);
    return ok;
}

int main(int argc, char** argv) {
    st3r_ctx* ctx; if (st3r_ctx_create(0, &ctx)) { printf("ctx: %s\n", st3r_last_error()); return 1; }
    bool ok = true;
    if (argc > 1) {   // rsort_probe <abl>: only the two big shapes, kernel ablation flags (timing only)
        const int abl = atoi(argv[1]);
        st3r_ctx_set_debug(ctx, abl << 8);
        run<uint32_t>(ctx, 8000000, 32, 0, false, false);
        run<uint32_t>(ctx, 26000000, 16, 1, false, false);
        return 0;
    }
    for (size_t n : {1ul, 63ul, 64ul, 65ul, 4095ul, 4096ul, 4097ul, 8192ul, 8193ul, 100000ul, 1000003ul})
        for (int mode = 0; mode < 3; ++mode) {
            ok &= run<uint32_t>(ctx, n, 32, mode, true, false);
            ok &= run<uint32_t>(ctx, n, 17, mode, true, false);
            ok &= run<uint64_t>(ctx, n, 49, mode, true, false);
        }
    ok &= run<uint32_t>(ctx, 8000000, 32, 0, true, true);
    ok &= run<uint32_t>(ctx, 26000000, 16, 0, true, true);
    ok &= run<uint32_t>(ctx, 26000000, 16, 1, true, true);
    ok &= run<uint64_t>(ctx, 8000000, 36, 0, true, true);
    ok &= run<uint64_t>(ctx, 26000000, 49, 0, false, true);
    printf(ok ? "ALL OK\n" : "FAILURES\n");
    return ok ? 0 : 1;
}
