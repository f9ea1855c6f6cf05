// What decides whether a wave64 VALU instruction costs two or four cycles of its SIMD on gfx950?
// (tools/probe/valu_cost.hip: pure streams of one opcode run at ~1.05 ns per instruction and SIMD; the counters of the
// product kernels say four cycles per instruction.)  Streams of 32 instructions x ITERS, fixed registers, W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/probe/valu_issue.hip -o tools/probe/valu_issue && tools/probe/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITERS 2000
#define R4(x) x x x x
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","s20","s21","s22","s23","vcc","scc"

#define KERNEL(name, body)                                                                     \
    __global__ __launch_bounds__(256) void name(float* out, float seed) {                      \
        asm volatile("s_mov_b32 s20, 0x3f800000\n s_mov_b32 s21, 0x3f800000\n s_mov_b32 s22, 0x3f800000\n s_mov_b32 s23, 0x3f800000\n" ::: CLOB); \
        for (int i = 0; i < ITERS; ++i) asm volatile(R4(body) ::: CLOB);                       \
        float r; asm volatile("v_mov_b32 %0, v8" : "=v"(r));                                   \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                               \
    }

// 8 instructions per body
KERNEL(k_indep, "v_fmac_f32 v8, v16, v24\n v_fmac_f32 v9, v17, v25\n v_fmac_f32 v10, v18, v26\n v_fmac_f32 v11, v19, v27\n v_fmac_f32 v12, v20, v28\n v_fmac_f32 v13, v21, v29\n v_fmac_f32 v14, v22, v30\n v_fmac_f32 v15, v23, v31\n")
KERNEL(k_sgpr, "v_fmac_f32 v8, s20, v24\n v_fmac_f32 v9, s21, v25\n v_fmac_f32 v10, s22, v26\n v_fmac_f32 v11, s23, v27\n v_fmac_f32 v12, s20, v28\n v_fmac_f32 v13, s21, v29\n v_fmac_f32 v14, s22, v30\n v_fmac_f32 v15, s23, v31\n")
KERNEL(k_bank, "v_fmac_f32 v8, v16, v24\n v_fmac_f32 v12, v20, v28\n v_fmac_f32 v8, v16, v24\n v_fmac_f32 v12, v20, v28\n v_fmac_f32 v8, v16, v24\n v_fmac_f32 v12, v20, v28\n v_fmac_f32 v8, v16, v24\n v_fmac_f32 v12, v20, v28\n")
KERNEL(k_chain, "v_fmac_f32 v8, v16, v24\n v_fmac_f32 v8, v17, v25\n v_fmac_f32 v8, v18, v26\n v_fmac_f32 v8, v19, v27\n v_fmac_f32 v8, v20, v28\n v_fmac_f32 v8, v21, v29\n v_fmac_f32 v8, v22, v30\n v_fmac_f32 v8, v23, v31\n")
KERNEL(k_chain_sgpr, "v_fmac_f32 v8, s20, v24\n v_fmac_f32 v8, s21, v25\n v_fmac_f32 v8, s22, v26\n v_fmac_f32 v8, s23, v27\n v_fmac_f32 v8, s20, v28\n v_fmac_f32 v8, s21, v29\n v_fmac_f32 v8, s22, v30\n v_fmac_f32 v8, s23, v31\n")
KERNEL(k_mixed, "v_fmac_f32 v8, v16, v24\n v_mul_f32 v9, v17, v25\n v_add_f32 v10, v18, v26\n v_sub_f32 v11, v19, v27\n v_fmac_f32 v12, v20, v28\n v_mul_f32 v13, v21, v29\n v_add_f32 v14, v22, v30\n v_sub_f32 v15, v23, v31\n")
KERNEL(k_fma3, "v_fma_f32 v8, v16, v24, v32\n v_fma_f32 v9, v17, v25, v33\n v_fma_f32 v10, v18, v26, v34\n v_fma_f32 v11, v19, v27, v35\n v_fma_f32 v12, v20, v28, v36\n v_fma_f32 v13, v21, v29, v37\n v_fma_f32 v14, v22, v30, v38\n v_fma_f32 v15, v23, v31, v39\n")
// five accumulators of one convolution tap, as the loss kernel issues them: products depend on the muls before them
KERNEL(k_tap, "v_mul_f32 v16, s20, v24\n v_mul_f32 v17, s20, v25\n v_add_f32 v8, v8, v16\n v_add_f32 v9, v9, v17\n v_fmac_f32 v10, v16, v24\n v_fmac_f32 v11, v17, v25\n v_fmac_f32 v12, v16, v25\n v_fmac_f32 v13, s21, v26\n")
// VGPR banks (register number mod 4): which operands have to differ?
KERNEL(k_bank_ab, "v_fmac_f32 v9, v16, v24\n v_fmac_f32 v10, v17, v25\n v_fmac_f32 v11, v18, v26\n v_fmac_f32 v8, v19, v27\n v_fmac_f32 v13, v20, v28\n v_fmac_f32 v14, v21, v29\n v_fmac_f32 v15, v22, v30\n v_fmac_f32 v12, v23, v31\n")
KERNEL(k_bank_da, "v_fmac_f32 v8, v16, v25\n v_fmac_f32 v9, v17, v26\n v_fmac_f32 v10, v18, v27\n v_fmac_f32 v11, v19, v24\n v_fmac_f32 v12, v20, v29\n v_fmac_f32 v13, v21, v30\n v_fmac_f32 v14, v22, v31\n v_fmac_f32 v15, v23, v28\n")
KERNEL(k_bank_none, "v_fmac_f32 v8, v17, v26\n v_fmac_f32 v9, v18, v27\n v_fmac_f32 v10, v19, v24\n v_fmac_f32 v11, v16, v25\n v_fmac_f32 v12, v21, v30\n v_fmac_f32 v13, v22, v31\n v_fmac_f32 v14, v23, v28\n v_fmac_f32 v15, v20, v29\n")
KERNEL(k_mul_ab, "v_mul_f32 v9, v16, v24\n v_mul_f32 v10, v17, v25\n v_mul_f32 v11, v18, v26\n v_mul_f32 v8, v19, v27\n v_mul_f32 v13, v20, v28\n v_mul_f32 v14, v21, v29\n v_mul_f32 v15, v22, v30\n v_mul_f32 v12, v23, v31\n")
KERNEL(k_mul_none, "v_mul_f32 v8, v17, v26\n v_mul_f32 v9, v18, v27\n v_mul_f32 v10, v19, v24\n v_mul_f32 v11, v16, v25\n v_mul_f32 v12, v21, v30\n v_mul_f32 v13, v22, v31\n v_mul_f32 v14, v23, v28\n v_mul_f32 v15, v20, v29\n")
KERNEL(k_literal, "v_mul_f32 v8, 0x3f7fbe77, v26\n v_mul_f32 v9, 0x3f7fbe77, v27\n v_mul_f32 v10, 0x3f7fbe77, v24\n v_mul_f32 v11, 0x3f7fbe77, v25\n v_mul_f32 v12, 0x3f7fbe77, v30\n v_mul_f32 v13, 0x3f7fbe77, v31\n v_mul_f32 v14, 0x3f7fbe77, v28\n v_mul_f32 v15, 0x3f7fbe77, v29\n")
KERNEL(k_inline, "v_mul_f32 v8, 2.0, v26\n v_mul_f32 v9, 2.0, v27\n v_mul_f32 v10, 2.0, v24\n v_mul_f32 v11, 2.0, v25\n v_mul_f32 v12, 2.0, v30\n v_mul_f32 v13, 2.0, v31\n v_mul_f32 v14, 2.0, v28\n v_mul_f32 v15, 2.0, v29\n")
KERNEL(k_half_sgpr, "v_fmac_f32 v8, s20, v26\n v_fmac_f32 v9, v18, v27\n v_fmac_f32 v10, s21, v24\n v_fmac_f32 v11, v16, v25\n v_fmac_f32 v12, s22, v30\n v_fmac_f32 v13, v22, v31\n v_fmac_f32 v14, s23, v28\n v_fmac_f32 v15, v20, v29\n")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, v17, v26\n v_cmp_lt_f32 vcc, v18, v27\n v_cmp_lt_f32 vcc, v19, v24\n v_cmp_lt_f32 vcc, v16, v25\n v_cmp_lt_f32 vcc, v21, v30\n v_cmp_lt_f32 vcc, v22, v31\n v_cmp_lt_f32 vcc, v23, v28\n v_cmp_lt_f32 vcc, v20, v29\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 v8, v17, v26, vcc\n v_cndmask_b32 v9, v18, v27, vcc\n v_cndmask_b32 v10, v19, v24, vcc\n v_cndmask_b32 v11, v16, v25, vcc\n v_cndmask_b32 v12, v21, v30, vcc\n v_cndmask_b32 v13, v22, v31, vcc\n v_cndmask_b32 v14, v23, v28, vcc\n v_cndmask_b32 v15, v20, v29, vcc\n")
KERNEL(k_max, "v_max_f32 v8, v17, v26\n v_max_f32 v9, v18, v27\n v_max_f32 v10, v19, v24\n v_max_f32 v11, v16, v25\n v_max_f32 v12, v21, v30\n v_max_f32 v13, v22, v31\n v_max_f32 v14, v23, v28\n v_max_f32 v15, v20, v29\n")
KERNEL(k_salu_mix, "v_fmac_f32 v8, v17, v26\n s_add_u32 s20, s20, 1\n v_fmac_f32 v9, v18, v27\n s_add_u32 s21, s21, 1\n v_fmac_f32 v10, v19, v24\n s_add_u32 s22, s22, 1\n v_fmac_f32 v11, v16, v25\n s_add_u32 s23, s23, 1\n")
KERNEL(k_fma_clamp, "v_fma_f32 v8, v17, v26, v35 clamp\n v_fma_f32 v9, v18, v27, v36 clamp\n v_fma_f32 v10, v19, v24, v37 clamp\n v_fma_f32 v11, v16, v25, v38 clamp\n v_fma_f32 v12, v21, v30, v39 clamp\n v_fma_f32 v13, v22, v31, v32 clamp\n v_fma_f32 v14, v23, v28, v33 clamp\n v_fma_f32 v15, v20, v29, v34 clamp\n ")
KERNEL(k_mul_clamp, "v_mul_f32_e64 v8, v17, v26 clamp\n v_mul_f32_e64 v9, v18, v27 clamp\n v_mul_f32_e64 v10, v19, v24 clamp\n v_mul_f32_e64 v11, v16, v25 clamp\n v_mul_f32_e64 v12, v21, v30 clamp\n v_mul_f32_e64 v13, v22, v31 clamp\n v_mul_f32_e64 v14, v23, v28 clamp\n v_mul_f32_e64 v15, v20, v29 clamp\n ")
KERNEL(k_or_inl, "v_or_b32_e32 v8, 4, v17\n v_or_b32_e32 v9, 4, v18\n v_or_b32_e32 v10, 4, v19\n v_or_b32_e32 v11, 4, v16\n v_or_b32_e32 v12, 4, v21\n v_or_b32_e32 v13, 4, v22\n v_or_b32_e32 v14, 4, v23\n v_or_b32_e32 v15, 4, v20\n ")
KERNEL(k_alignbit, "v_alignbit_b32 v8, v17, v26, 4\n v_alignbit_b32 v9, v18, v27, 4\n v_alignbit_b32 v10, v19, v24, 4\n v_alignbit_b32 v11, v16, v25, 4\n v_alignbit_b32 v12, v21, v30, 4\n v_alignbit_b32 v13, v22, v31, 4\n v_alignbit_b32 v14, v23, v28, 4\n v_alignbit_b32 v15, v20, v29, 4\n ")
KERNEL(k_lshlrev, "v_lshlrev_b32_e32 v8, 4, v17\n v_lshlrev_b32_e32 v9, 4, v18\n v_lshlrev_b32_e32 v10, 4, v19\n v_lshlrev_b32_e32 v11, 4, v16\n v_lshlrev_b32_e32 v12, 4, v21\n v_lshlrev_b32_e32 v13, 4, v22\n v_lshlrev_b32_e32 v14, 4, v23\n v_lshlrev_b32_e32 v15, 4, v20\n ")
KERNEL(k_and, "v_and_b32_e32 v8, v17, v26\n v_and_b32_e32 v9, v18, v27\n v_and_b32_e32 v10, v19, v24\n v_and_b32_e32 v11, v16, v25\n v_and_b32_e32 v12, v21, v30\n v_and_b32_e32 v13, v22, v31\n v_and_b32_e32 v14, v23, v28\n v_and_b32_e32 v15, v20, v29\n ")
KERNEL(k_mad24, "v_mad_u32_u24 v8, v17, 48, v26\n v_mad_u32_u24 v9, v18, 48, v27\n v_mad_u32_u24 v10, v19, 48, v24\n v_mad_u32_u24 v11, v16, 48, v25\n v_mad_u32_u24 v12, v21, 48, v30\n v_mad_u32_u24 v13, v22, 48, v31\n v_mad_u32_u24 v14, v23, 48, v28\n v_mad_u32_u24 v15, v20, 48, v29\n ")
KERNEL(k_lshl_add, "v_lshl_add_u32 v8, v17, 4, v26\n v_lshl_add_u32 v9, v18, 4, v27\n v_lshl_add_u32 v10, v19, 4, v24\n v_lshl_add_u32 v11, v16, 4, v25\n v_lshl_add_u32 v12, v21, 4, v30\n v_lshl_add_u32 v13, v22, 4, v31\n v_lshl_add_u32 v14, v23, 4, v28\n v_lshl_add_u32 v15, v20, 4, v29\n ")
KERNEL(k_ldexp, "v_ldexp_f32 v8, v17, 64\n v_ldexp_f32 v9, v18, 64\n v_ldexp_f32 v10, v19, 64\n v_ldexp_f32 v11, v16, 64\n v_ldexp_f32 v12, v21, 64\n v_ldexp_f32 v13, v22, 64\n v_ldexp_f32 v14, v23, 64\n v_ldexp_f32 v15, v20, 64\n ")
KERNEL(k_addu, "v_add_u32_e32 v8, v17, v26\n v_add_u32_e32 v9, v18, v27\n v_add_u32_e32 v10, v19, v24\n v_add_u32_e32 v11, v16, v25\n v_add_u32_e32 v12, v21, v30\n v_add_u32_e32 v13, v22, v31\n v_add_u32_e32 v14, v23, v28\n v_add_u32_e32 v15, v20, v29\n ")
KERNEL(k_lshl_or_s, "v_lshl_or_b32 v8, s20, 1, v17\n v_lshl_or_b32 v9, s20, 1, v18\n v_lshl_or_b32 v10, s20, 1, v19\n v_lshl_or_b32 v11, s20, 1, v16\n v_lshl_or_b32 v12, s20, 1, v21\n v_lshl_or_b32 v13, s20, 1, v22\n v_lshl_or_b32 v14, s20, 1, v23\n v_lshl_or_b32 v15, s20, 1, v20\n ")
KERNEL(k_subrev_lit, "v_subrev_f32_e32 v8, 0x3b808080, v17\n v_subrev_f32_e32 v9, 0x3b808080, v18\n v_subrev_f32_e32 v10, 0x3b808080, v19\n v_subrev_f32_e32 v11, 0x3b808080, v16\n v_subrev_f32_e32 v12, 0x3b808080, v21\n v_subrev_f32_e32 v13, 0x3b808080, v22\n v_subrev_f32_e32 v14, 0x3b808080, v23\n v_subrev_f32_e32 v15, 0x3b808080, v20\n ")
KERNEL(k_cvt, "v_cvt_f32_i32_e32 v8, v17\n v_cvt_f32_i32_e32 v9, v18\n v_cvt_f32_i32_e32 v10, v19\n v_cvt_f32_i32_e32 v11, v16\n v_cvt_f32_i32_e32 v12, v21\n v_cvt_f32_i32_e32 v13, v22\n v_cvt_f32_i32_e32 v14, v23\n v_cvt_f32_i32_e32 v15, v20\n ")
KERNEL(k_fmac_lit, "v_fmac_f32_e32 v8, 0x40400000, v17\n v_fmac_f32_e32 v9, 0x40400000, v18\n v_fmac_f32_e32 v10, 0x40400000, v19\n v_fmac_f32_e32 v11, 0x40400000, v16\n v_fmac_f32_e32 v12, 0x40400000, v21\n v_fmac_f32_e32 v13, 0x40400000, v22\n v_fmac_f32_e32 v14, 0x40400000, v23\n v_fmac_f32_e32 v15, 0x40400000, v20\n ")
KERNEL(k_salu_only, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n")
KERNEL(k_salu_3to1, "v_fmac_f32 v8, v17, v26\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n v_fmac_f32 v10, v19, v24\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
KERNEL(k_salu_1to3, "v_fmac_f32 v8, v17, v26\n v_fmac_f32 v9, v18, v27\n v_fmac_f32 v10, v19, v24\n s_add_u32 s20, s20, 1\n v_fmac_f32 v11, v16, v25\n v_fmac_f32 v12, v21, v30\n v_fmac_f32 v13, v22, v31\n s_add_u32 s21, s21, 1\n")
KERNEL(k_salu_bitscan, "s_flbit_i32_b64 s20, s[22:23]\n s_xor_b32 s20, s20, 63\n s_lshl_b64 s[20:21], 1, s20\n s_andn2_b64 s[22:23], s[22:23], s[20:21]\n v_fmac_f32 v8, v17, v26\n v_fmac_f32 v9, v18, v27\n v_fmac_f32 v10, v19, v24\n v_fmac_f32 v11, v16, v25\n")
KERNEL(k_mov, "v_mov_b32 v8, v16\n v_mov_b32 v9, v17\n v_mov_b32 v10, v18\n v_mov_b32 v11, v19\n v_mov_b32 v12, v20\n v_mov_b32 v13, v21\n v_mov_b32 v14, v22\n v_mov_b32 v15, v23\n")

template <typename K>
static void run(K k, float* out, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-12s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;   // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)wps * ITERS * 32;
        printf("  %d w/SIMD: %6.3f ns/instr", wps, ms * 1e6 / per_simd);
    }
    printf("\n");
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run(k_indep, out, "indep"); run(k_sgpr, out, "sgpr-src"); run(k_bank, out, "same-bank"); run(k_chain, out, "chain");
    run(k_chain_sgpr, out, "chain+sgpr"); run(k_mixed, out, "mixed-op"); run(k_fma3, out, "fma 3-src"); run(k_tap, out, "conv tap");
    run(k_mov, out, "v_mov");
    run(k_bank_none, out, "fmac banks-");  run(k_bank_ab, out, "fmac A=B"); run(k_bank_da, out, "fmac D=A");
    run(k_mul_none, out, "mul banks-"); run(k_mul_ab, out, "mul A=B");
    run(k_literal, out, "literal"); run(k_inline, out, "inline 2.0"); run(k_half_sgpr, out, "half sgpr");
    run(k_cmp_vcc, out, "cmp->vcc"); run(k_cnd_vcc, out, "cndmask vcc"); run(k_max, out, "v_max");
    run(k_salu_mix, out, "fmac+s_add");
    run(k_salu_only, out, "salu only"); run(k_salu_3to1, out, "1 fmac:3 salu"); run(k_salu_1to3, out, "3 fmac:1 salu"); run(k_salu_bitscan, out, "4 salu+4 fmac");
    run(k_fma_clamp, out, "fma clamp"); run(k_mul_clamp, out, "mul clamp"); run(k_or_inl, out, "or inline"); run(k_alignbit, out, "alignbit");
    run(k_lshlrev, out, "lshlrev"); run(k_and, out, "v_and"); run(k_mad24, out, "mad_u32_u24"); run(k_lshl_add, out, "lshl_add");
    run(k_ldexp, out, "ldexp"); run(k_addu, out, "add_u32"); run(k_lshl_or_s, out, "lshl_or sgpr"); run(k_subrev_lit, out, "subrev lit");
    run(k_cvt, out, "cvt_f32_i32"); run(k_fmac_lit, out, "fmac lit");
    return 0;
}
