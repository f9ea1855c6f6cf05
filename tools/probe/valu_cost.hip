// Issue cost of single VALU instructions on gfx950, relative to v_mul_f32 (independent chains, 8 waves/SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/probe/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define ITERS 4000

#define KERNEL(name, body)                                                      \
    __global__ __launch_bounds__(256) void name(float* out, float seed) {       \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;   \
        float b0 = 1.0001f, b1 = 0.9999f;                                       \
        for (int i = 0; i < ITERS; ++i) { asm volatile(REP8(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1)); } \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;               \
    }

// each body = 4 independent instructions (one per chain)
KERNEL(k_mul, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %5\n")
KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %5, %4\n")
__global__ __launch_bounds__(256) void k_pkmul(float* out, float seed) {
    float a0 = seed + threadIdx.x;
    asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n"
                 "v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n"
                 :: "v"(a0) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21");
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(REP8("v_pk_mul_f32 v[10:11], v[10:11], v[20:21]\n v_pk_mul_f32 v[12:13], v[12:13], v[20:21]\n"
                          "v_pk_mul_f32 v[14:15], v[14:15], v[20:21]\n v_pk_mul_f32 v[16:17], v[16:17], v[20:21]\n")
                     ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    }
    asm volatile("v_mov_b32 %0, v10" : "=v"(a0) :: "v10");
    out[blockIdx.x * 256 + threadIdx.x] = a0;
}
KERNEL(k_fmac, "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %5, %4\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %5, %4\n")
KERNEL(k_add, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(k_sub, "v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %5, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %5, %3\n")
KERNEL(k_max3, "v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %5, %4\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %5, %4\n")
KERNEL(k_cmpx, "v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %5\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %5\n")
__global__ __launch_bounds__(256) void k_pkfma(float* out, float seed) {
    float a0 = seed + threadIdx.x;
    asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n"
                 "v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n"
                 "v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
                 :: "v"(a0) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23");
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(REP8("v_pk_fma_f32 v[10:11], v[10:11], v[20:21], v[22:23]\n v_pk_fma_f32 v[12:13], v[12:13], v[20:21], v[22:23]\n"
                          "v_pk_fma_f32 v[14:15], v[14:15], v[20:21], v[22:23]\n v_pk_fma_f32 v[16:17], v[16:17], v[20:21], v[22:23]\n")
                     ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    }
    asm volatile("v_mov_b32 %0, v10" : "=v"(a0) :: "v10");
    out[blockIdx.x * 256 + threadIdx.x] = a0;
}
KERNEL(k_cnd64, "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %5, s[22:23]\n")
// compare + select pairs as compiled code has them (two instructions per chain step: the per-instruction figure is half)
KERNEL(k_cmpcnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %5\n v_cndmask_b32 %1, %1, %4, vcc\n")
KERNEL(k_cmpcnd64, "v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %1, %5\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_dpp, "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %5\n")
KERNEL(k_cnd, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n")
KERNEL(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n")
KERNEL(k_min, "v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %5\n")
KERNEL(k_readlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n")

template <typename K>
static double run(K k, float* out, const char* name, double base) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)blocks * 4 / 1024.0 * ITERS * 32;
    const double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-10s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (x%.2f of v_mul)\n", name, ms, ns_per_instr,
           base > 0 ? ns_per_instr / base : 1.0);
    return ns_per_instr;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    double b = run(k_mul, out, "v_mul", 0);
    run(k_fma, out, "v_fma", b); run(k_pkmul, out, "v_pk_mul", b); run(k_exp, out, "v_exp", b); run(k_rcp, out, "v_rcp", b);
    run(k_dpp, out, "add_dpp", b); run(k_swap32, out, "swap32", b); run(k_swap16, out, "swap16", b);
    run(k_cmp, out, "v_cmp", b); run(k_cnd, out, "cndmask", b); run(k_mov, out, "v_mov", b); run(k_min, out, "v_min", b);
    run(k_readlane, out, "readfirst", b);
    run(k_fmac, out, "v_fmac", b); run(k_add, out, "v_add", b); run(k_sub, out, "v_sub", b); run(k_max3, out, "v_max3", b);
    run(k_cmpx, out, "v_cmp_e64", b); run(k_pkfma, out, "v_pk_fma", b);
    run(k_cnd64, out, "cndmask_e64", b); run(k_cmpcnd, out, "cmp+cnd vcc", b); run(k_cmpcnd64, out, "cmp+cnd s[]", b);
    return 0;
}
