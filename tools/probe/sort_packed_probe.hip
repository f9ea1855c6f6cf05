// Would the two sorts of the train step be faster on PACKED items -- one u64 (key << 32 | value), sorted keys-only on the key's
// bits -- than on separate u32 key and int32 value arrays?  Same bytes, half the memory instructions and one LDS regroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 sort_packed_probe.hip -L../../starst3r_amd -lst3r_hip \
//         -Wl,-rpath,'$ORIGIN/../../starst3r_amd' -o sort_packed_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../../include/st3r.h"

static float time_it(st3r_ctx* c, int kb, int64_t n, int b0, int b1, const void* a, const int32_t* va, void* b, int32_t* vb) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) st3r_radix_sort_pairs(c, nullptr, kb, n, b0, b1, a, va, b, vb);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) st3r_radix_sort_pairs(c, nullptr, kb, n, b0, b1, a, va, b, vb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}

int main() {
    st3r_ctx* ctx; if (st3r_ctx_create(0, &ctx)) return 1;
    struct Shape { int64_t n; int bits; int mode; } shapes[] = {{8000000, 32, 0}, {26000000, 16, 1}};
    for (auto sh : shapes) {
        std::vector<uint32_t> k(sh.n); std::vector<int32_t> v(sh.n); std::vector<uint64_t> p(sh.n);
        unsigned long long x = 88172645463325252ull;
        for (int64_t i = 0; i < sh.n; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t key = sh.mode ? (uint32_t)((i / 37) % 65280) : (uint32_t)x;
            if (sh.bits < 32) key &= (1u << sh.bits) - 1u;
            k[i] = key; v[i] = (int32_t)i; p[i] = ((uint64_t)key << 32) | (uint32_t)i;
        }
        uint32_t *ka, *kb_; int32_t *va, *vb; uint64_t *pa, *pb;
        hipMalloc(&ka, sh.n * 4); hipMalloc(&kb_, sh.n * 4); hipMalloc(&va, sh.n * 4); hipMalloc(&vb, sh.n * 4);
        hipMalloc(&pa, sh.n * 8); hipMalloc(&pb, sh.n * 8);
        hipMemcpy(ka, k.data(), sh.n * 4, hipMemcpyHostToDevice); hipMemcpy(va, v.data(), sh.n * 4, hipMemcpyHostToDevice);
        hipMemcpy(pa, p.data(), sh.n * 8, hipMemcpyHostToDevice);
        const float t_pairs = time_it(ctx, 4, sh.n, 0, sh.bits, ka, va, kb_, vb);
        const float t_packed = time_it(ctx, 8, sh.n, 32, 32 + sh.bits, pa, nullptr, pb, nullptr);
        // check: same order
        std::vector<int32_t> ov(sh.n); std::vector<uint64_t> op(sh.n);
        hipMemcpy(ov.data(), vb, sh.n * 4, hipMemcpyDeviceToHost); hipMemcpy(op.data(), pb, sh.n * 8, hipMemcpyDeviceToHost);
        bool same = true;
        for (int64_t i = 0; i < sh.n && same; ++i) same = (uint32_t)op[i] == (uint32_t)ov[i];
        printf("n=%lld bits=%d: u32 key + i32 value %.3f ms | packed u64 keys-only %.3f ms | same order: %s\n", (long long)sh.n,
               sh.bits, t_pairs, t_packed, same ? "yes" : "NO");
        hipFree(ka); hipFree(kb_); hipFree(va); hipFree(vb); hipFree(pa); hipFree(pb);
    }
    return 0;
}
