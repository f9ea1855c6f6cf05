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

__global__ __launch_bounds__(256) void k_adam(int64_t N, float* __restrict__ means, float* __restrict__ quats,
                                              float* __restrict__ scales, float* __restrict__ opacities,
                                              float* __restrict__ sh, int sh_stride,
                                              const float* __restrict__ grads, float* __restrict__ m,
                                              float* __restrict__ v, AdamK k) {
    const int64_t total = 23 * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float* p;
        if (i < 3 * N) p = means + i;
        else if (i < 7 * N) p = quats + (i - 3 * N);
        else if (i < 10 * N) p = scales + (i - 7 * N);
        else if (i < 11 * N) p = opacities + (i - 10 * N);
        else {
            const int64_t j = i - 11 * N;
            const int64_t g = j / 12;
            p = sh + g * sh_stride + (j - g * 12);
        }
        const float gi = grads[i];
        const float mi = fmaf(k.w1, gi - m[i], m[i]);
        const float vi = v[i] * k.b2 + (k.w2 * gi) * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / k.bc2_sqrt + k.eps;
        *p = *p - k.step_size * (mi / denom);
    }
}

int st3r_adam_impl(hipStream_t s, int N, float* means, float* quats, float* scales, float* opacities, float* sh,
                   int sh_stride, const float* grads, float* m, float* v, double lr, double b1, double b2,
                   double eps, int step) {
    if (N == 0) return ST3R_OK;
    AdamK k;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    k.step_size = (float)(lr / bc1);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.w1 = (float)(1.0 - b1); k.w2 = (float)(1.0 - b2); k.b2 = (float)b2; k.eps = (float)eps;
    const int64_t total = 23 * (int64_t)N;
    int blocks = ceil_div(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, (int64_t)N, means, quats, scales, opacities, sh,
                       sh_stride, grads, m, v, k);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_adam_step(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                               float* opacities, float* sh, int sh_stride, const float* grads, float* m, float* v,
                               double lr, double beta1, double beta2, double eps, int step) {
    ARG_CHECK(ctx && N >= 0 && step >= 1 && sh_stride >= 12);
    ARG_CHECK(means && quats && scales && opacities && sh && grads && m && v);
    st3r_prof_begin(ctx, (hipStream_t)stream, STG_ADAM);
    int rc = st3r_adam_impl((hipStream_t)stream, N, means, quats, scales, opacities, sh, sh_stride, grads, m, v, lr,
                            beta1, beta2, eps, step);
    st3r_prof_end(ctx, (hipStream_t)stream, STG_ADAM);
    return rc;
}
