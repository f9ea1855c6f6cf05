// K9: fused Adam over the 23 active scalars of every Gaussian in one launch.
// Replaces the 6 torch.optim.Adam instances of starster/gs.py:37,159-161 (lr 1e-3,
// betas (0.9, 0.999), eps 1e-8, no weight decay, bias corrected).  sh0 and SH rows 4..23
// never receive a gradient in the reference (App. B-2/B-3 of SURVEY.md): their Adam update
// is exactly 0, so they are simply not touched here.
//
// grads / m / v use the block layout means[3N] quats[4N] scales[3N] opacities[N] sh4[12N];
// parameters are updated in place in the caller's tensors (sh with row stride sh_stride).
// Arithmetic mirrors torch.optim.Adam's single-tensor path (see oracle/gs_oracle.c gso_adam).
#include "common.h"

struct AdamK {
    float step_size, bc2_sqrt, w1, w2, b2, eps;
};

#define ADAM_SKIP(count_dev, count_cap, status_dev) \
    (((count_dev) && (uint32_t)(count_dev)[0] > (count_cap)) || ((status_dev) && (status_dev)[0] != 0))

// The update covers either a range [i0, i1) of the 23N scalars in buffer order (reduce-scatter exchange: a rank owns
// one contiguous piece of the gradient buffer) or the 23 scalars of the Gaussians [g0, g1) (range-wise exchange); the
// whole buffer is the range [0, 23N).  pstage != NULL: the new parameter value is also left in pstage[i] (buffer
// order), the payload of the parameter all-gather that follows.
__device__ __forceinline__ float* adam_param(int64_t i, int64_t N, float* means, float* quats, float* scales,
                                             float* opacities, float* sh, int sh_stride) {
    if (i < 3 * N) return means + i;
    if (i < 7 * N) return quats + (i - 3 * N);
    if (i < 10 * N) return scales + (i - 7 * N);
    if (i < 11 * N) return opacities + (i - 10 * N);
    const int64_t j = i - 11 * N;
    const int64_t g = j / 12;
    return sh + g * sh_stride + (j - g * 12);
}

__global__ __launch_bounds__(256) void k_adam(int64_t N, float* __restrict__ means, float* __restrict__ quats,
                                              float* __restrict__ scales, float* __restrict__ opacities,
                                              float* __restrict__ sh, int sh_stride,
                                              const float* __restrict__ grads, float* __restrict__ m,
                                              float* __restrict__ v, AdamK k, const int32_t* __restrict__ count_dev,
                                              uint32_t count_cap, const int32_t* __restrict__ status_dev, int64_t i0,
                                              int64_t i1, int64_t g0, int64_t g1,
                                              float* __restrict__ pstage, const float* __restrict__ gstage,
                                              float* __restrict__ grads_out) {
    // Asynchronous training steps keep the record count on the device; a step whose count outgrew the capacity of its
    // buffers dropped records, so its gradients are incomplete: the update is skipped HERE, on the device, and the next
    // call reports ST3R_ERR_CAPACITY -- the caller repeats the iteration with nothing to undo (a count above 2^31 wraps
    // negative: the unsigned compare catches it).
    // status_dev: the max-reduced status word of an exchanged step (comm.hip): a step that failed on any rank is applied
    // on none.  ONLY st3r_gs_train_step passes it -- the word is rewritten by the next exchanged step and by nothing else,
    // so the stand-alone entry points (st3r_adam_step, _range, st3r_params_from_stage) must not look at it.
    if (ADAM_SKIP(count_dev, count_cap, status_dev)) return;
    const bool by_gaussian = g1 - g0 < N || gstage;
    const int64_t n = g1 - g0;
    const int64_t total = by_gaussian ? 23 * n : i1 - i0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        int64_t i;
        if (by_gaussian) {   // local index -> block (means 3, quats 4, scales 3, opacities 1, sh 12) -> buffer index
            if (j < 3 * n) i = 3 * g0 + j;
            else if (j < 7 * n) i = 3 * N + 4 * g0 + (j - 3 * n);
            else if (j < 10 * n) i = 7 * N + 3 * g0 + (j - 7 * n);
            else if (j < 11 * n) i = 10 * N + g0 + (j - 10 * n);
            else i = 11 * N + 12 * g0 + (j - 11 * n);
        } else {
            i = i0 + j;
        }
        float* p = adam_param(i, N, means, quats, scales, opacities, sh, sh_stride);
        // gstage (range-wise exchange): the range's gradients sit contiguously at 23 g0 in the range's own block layout,
        // i.e. at the local index j; they are handed on to the caller's buffer in its layout
        float gi;
        if (gstage) { gi = gstage[23 * g0 + j]; grads_out[i] = gi; }
        else gi = grads[i];
        const float mi = fmaf(k.w1, gi - m[i], m[i]);
        const float vi = v[i] * k.b2 + (k.w2 * gi) * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / k.bc2_sqrt + k.eps;
        const float pn = *p - k.step_size * (mi / denom);
        *p = pn;
        if (pstage) pstage[i] = pn;
    }
}

// Four scalars per thread for the whole-buffer / contiguous-range update when everything is 16-byte aligned (N % 4 == 0:
// the five parameter blocks then start on multiples of four, and a group of four never leaves its Gaussian's 12 SH
// scalars): the same arithmetic per scalar, a quarter of the memory instructions.
__global__ __launch_bounds__(256) void k_adam4(int64_t N, float* __restrict__ means, float* __restrict__ quats,
                                               float* __restrict__ scales, float* __restrict__ opacities,
                                               float* __restrict__ sh, int sh_stride,
                                               const float* __restrict__ grads, float* __restrict__ m,
                                               float* __restrict__ v, AdamK k, const int32_t* __restrict__ count_dev,
                                               uint32_t count_cap, const int32_t* __restrict__ status_dev, int64_t i0,
                                               int64_t i1, float* __restrict__ pstage) {
    if (ADAM_SKIP(count_dev, count_cap, status_dev)) return;
    const int64_t total4 = (i1 - i0) >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total4; j += stride) {
        const int64_t i = i0 + 4 * j;
        float4* p = reinterpret_cast<float4*>(adam_param(i, N, means, quats, scales, opacities, sh, sh_stride));
        const float4 g4 = *reinterpret_cast<const float4*>(grads + i);
        float4 m4 = *reinterpret_cast<float4*>(m + i), v4 = *reinterpret_cast<float4*>(v + i), p4 = *p;
        const float gi[4] = {g4.x, g4.y, g4.z, g4.w};
        float mi[4] = {m4.x, m4.y, m4.z, m4.w}, vi[4] = {v4.x, v4.y, v4.z, v4.w}, pi[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mi[q] = fmaf(k.w1, gi[q] - mi[q], mi[q]);
            vi[q] = vi[q] * k.b2 + (k.w2 * gi[q]) * gi[q];
            const float denom = sqrtf(vi[q]) / k.bc2_sqrt + k.eps;
            pi[q] = pi[q] - k.step_size * (mi[q] / denom);
        }
        *reinterpret_cast<float4*>(m + i) = make_float4(mi[0], mi[1], mi[2], mi[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vi[0], vi[1], vi[2], vi[3]);
        const float4 pn = make_float4(pi[0], pi[1], pi[2], pi[3]);
        *p = pn;
        if (pstage) *reinterpret_cast<float4*>(pstage + i) = pn;
    }
}

// parameters of the scalars OUTSIDE [i0, i1) <- pstage (what the other ranks computed and the all-gather delivered)
__global__ __launch_bounds__(256) void k_params_from_stage(int64_t N, float* __restrict__ means, float* __restrict__ quats,
                                                           float* __restrict__ scales, float* __restrict__ opacities,
                                                           float* __restrict__ sh, int sh_stride,
                                                           const float* __restrict__ pstage, int64_t i0, int64_t i1,
                                                           int64_t lim, const int32_t* __restrict__ count_dev,
                                                           uint32_t count_cap, const int32_t* __restrict__ status_dev) {
    if (ADAM_SKIP(count_dev, count_cap, status_dev)) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lim; i += stride) {
        if (i >= i0 && i < i1) continue;
        *adam_param(i, N, means, quats, scales, opacities, sh, sh_stride) = pstage[i];
    }
}

// the same from the owners' staging buffers directly (direct exchange, comm.hip): piece p of the buffer -- scalars
// [p q, (p + 1) q) -- is read from tab[p], a peer's exported staging buffer mapped through HIP IPC; own piece skipped
__global__ __launch_bounds__(256) void k_params_from_peers(int64_t N, float* __restrict__ means, float* __restrict__ quats,
                                                           float* __restrict__ scales, float* __restrict__ opacities,
                                                           float* __restrict__ sh, int sh_stride,
                                                           const float* const* __restrict__ tab, int r, int64_t q,
                                                           int64_t lim, const int32_t* __restrict__ status_dev) {
    if (status_dev && status_dev[0] != 0) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lim; i += stride) {
        const int p = (int)(i / q);
        if (p == r) continue;
        // (another device's memory: a system-scope load, past whatever this device cached of it during the last step)
        *adam_param(i, N, means, quats, scales, opacities, sh, sh_stride) =
            __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(tab[p] + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
}

int st3r_params_from_peers_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities,
                                float* sh, int sh_stride, const float* const* tab, int r, int64_t q, int64_t lim,
                                const int32_t* status_dev) {
    if (lim <= 0 || q <= 0) return ST3R_OK;
    int blocks = ceil_div(lim, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_params_from_peers, dim3(blocks), dim3(256), 0, s, (int64_t)N, means, quats, scales, opacities, sh,
                       sh_stride, tab, r, q, lim, status_dev);
    LAUNCH_CHECK();
    return ST3R_OK;
}

static AdamK adam_constants(double lr, double b1, double b2, double eps, int step) {
    AdamK k;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    k.step_size = (float)(lr / bc1);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.w1 = (float)(1.0 - b1); k.w2 = (float)(1.0 - b2); k.b2 = (float)b2; k.eps = (float)eps;
    return k;
}

// i0 < 0: the whole buffer.  [g0, g1) a proper sub-range of the Gaussians: that range instead of [i0, i1).
int st3r_adam_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities, float* sh,
                   int sh_stride, const float* grads, float* m, float* v, double lr, double b1, double b2,
                   double eps, int step, const int32_t* count_dev, uint32_t count_cap, const int32_t* status_dev, int64_t i0,
                   int64_t i1, int64_t g0, int64_t g1, float* pstage, const float* gstage, float* grads_out) {
    if (N == 0) return ST3R_OK;
    if (i0 < 0) { i0 = 0; i1 = 23 * (int64_t)N; }
    if (g1 < 0) { g0 = 0; g1 = N; }
    const AdamK k = adam_constants(lr, b1, b2, eps, step);
    const int64_t total = (g1 - g0 < N) ? 23 * (g1 - g0) : i1 - i0;
    if (total <= 0) return ST3R_OK;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool contiguous = !(g1 - g0 < N) && !gstage;
#ifndef ADAM_NO_VEC
    if (contiguous && N % 4 == 0 && i0 % 4 == 0 && (i1 - i0) % 4 == 0 && sh_stride % 4 == 0 && al16(means) && al16(quats) &&
        al16(scales) && al16(opacities) && al16(sh) && al16(grads) && al16(m) && al16(v) && (!pstage || al16(pstage))) {
        int blocks4 = ceil_div(total / 4, 256);
        if (blocks4 > 256 * 16) blocks4 = 256 * 16;
        hipLaunchKernelGGL(k_adam4, dim3(blocks4), dim3(256), 0, s, (int64_t)N, means, quats, scales, opacities, sh,
                           sh_stride, grads, m, v, k, count_dev, count_cap, status_dev, i0, i1, pstage);
        LAUNCH_CHECK();
        return ST3R_OK;
    }
#endif
    int blocks = ceil_div(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, (int64_t)N, means, quats, scales, opacities, sh,
                       sh_stride, grads, m, v, k, count_dev, count_cap, status_dev, i0, i1, g0, g1, pstage, gstage, grads_out);
    LAUNCH_CHECK();
    return ST3R_OK;
}

int st3r_params_from_stage_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities,
                                float* sh, int sh_stride, const float* pstage, int64_t i0, int64_t i1, int64_t lim,
                                const int32_t* count_dev, uint32_t count_cap, const int32_t* status_dev) {
    if (lim <= 0) return ST3R_OK;
    int blocks = ceil_div(lim, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_params_from_stage, dim3(blocks), dim3(256), 0, s, (int64_t)N, means, quats, scales, opacities, sh,
                       sh_stride, pstage, i0, i1, lim, count_dev, count_cap, status_dev);
    LAUNCH_CHECK();
    return ST3R_OK;
}

// the device-side guard of an asynchronous step that is still in flight (see k_adam)
// (word 0 of the counts buffer: the record count of the step, compared with its capacity.  Word 4 of the same buffer is
// the status word of an exchanged step; it is NOT part of this guard: st3r_gs_train_step hands it to the kernels itself.)
void st3r_adam_guard(st3r_ctx* ctx, const int32_t** count_dev, uint32_t* count_cap) {
    const bool guard = ctx->slot_ptr[SLOT_COUNTS] != nullptr && ctx->count_pending;
    *count_dev = guard ? (const int32_t*)ctx->slot_ptr[SLOT_COUNTS] : nullptr;
    *count_cap = guard ? (uint32_t)ctx->count_cap : 0xFFFFFFFFu;
}

ST3R_EXPORT int st3r_adam_step(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                               float* opacities, float* sh, int sh_stride, const float* grads, float* m, float* v,
                               double lr, double beta1, double beta2, double eps, int step) {
    ARG_CHECK(ctx && N >= 0 && step >= 1 && sh_stride >= 12);
    ARG_CHECK(means && quats && scales && opacities && sh && grads && m && v);
    st3r_prof_begin(ctx, (hipStream_t)stream, STG_ADAM);
    // an asynchronous step is in flight and its count not yet settled: guard the update with it (see k_adam)
    const int32_t* count_dev; uint32_t count_cap;
    st3r_adam_guard(ctx, &count_dev, &count_cap);
    int rc = st3r_adam_impl((hipStream_t)stream, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr,
                            beta1, beta2, eps, step, count_dev, count_cap, nullptr, -1, -1, 0, -1, nullptr, nullptr, nullptr);
    st3r_prof_end(ctx, (hipStream_t)stream, STG_ADAM);
    return rc;
}

ST3R_EXPORT int st3r_adam_step_range(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                                     float* opacities, float* sh, int sh_stride, const float* grads, float* m, float* v,
                                     double lr, double beta1, double beta2, double eps, int step, int64_t i0, int64_t i1,
                                     float* param_stage) {
    ARG_CHECK(ctx && N >= 0 && step >= 1 && sh_stride >= 12 && i0 >= 0 && i1 >= i0 && i1 <= (int64_t)23 * N);
    ARG_CHECK(means && quats && scales && opacities && sh && grads && m && v);
    const int32_t* count_dev; uint32_t count_cap;
    st3r_adam_guard(ctx, &count_dev, &count_cap);
    return st3r_adam_impl((hipStream_t)stream, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr, beta1,
                          beta2, eps, step, count_dev, count_cap, nullptr, i0, i1, 0, -1, param_stage, nullptr, nullptr);
}

ST3R_EXPORT int st3r_params_from_stage(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                                       float* opacities, float* sh, int sh_stride, const float* param_stage, int64_t i0,
                                       int64_t i1, int64_t limit) {
    ARG_CHECK(ctx && N >= 0 && sh_stride >= 12 && i0 >= 0 && i1 >= i0 && limit >= 0 && limit <= (int64_t)23 * N);
    ARG_CHECK(means && quats && scales && opacities && sh && param_stage);
    const int32_t* count_dev; uint32_t count_cap;
    st3r_adam_guard(ctx, &count_dev, &count_cap);
    return st3r_params_from_stage_impl((hipStream_t)stream, N, means, quats, scales, opacities, sh, sh_stride, param_stage,
                                       i0, i1, limit, count_dev, count_cap, nullptr);
}
