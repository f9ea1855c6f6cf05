// SURVEY 8(f) row 3 -- dense point extraction between the alignment (path B) and the 3DGS seeding (path C):
// what starster/scene.py:148 gets from `scene.get_dense_pts3d(clean_depth=True)` (Mast3r SparseGA [U]: every
// pixel of every view is unprojected like an anchor of that view's optimised core depthmap, then
// dust3r clean_pointcloud lowers the confidence of points that float in front of another view's surface).
//
//   k_dense_unproject  one thread per dense pixel: depth = (A + B core[idx]) * (1 + (offset - 1) base_focal / f),
//                      camera point = depth * K^-1 (x, y, 1), world point = cam2w * camera point
//                      (App. A.5 make_pts3d; same arithmetic as k_align_points for the anchors)
//   k_dense_clean      view i against every other view j, in the reference's loop order (i outer, j inner,
//                      confidences updated in place): a point of i that projects inside view j, lies in front of
//                      j's depth by more than `tol` and is less confident than j's pixel gets conf = min(conf, bad_conf)
//
// The confidence threshold and the colour gather of scene.py:150-155 stay boolean-mask indexing on the host side.
#include "common.h"

#define CAM_STRIDE 24  // per-view row of st3r_align_run's cam_out: R(9) T(3) f cx cy A B ...

__device__ __forceinline__ int view_of(const int32_t* __restrict__ view_start, int C, int i) {
    int lo = 0, hi = C;  // last v with view_start[v] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (view_start[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_dense_unproject(int C, int G, int n, const int32_t* __restrict__ view_start,
                                                         const float* __restrict__ pixels,
                                                         const int32_t* __restrict__ idxs,
                                                         const float* __restrict__ offsets,
                                                         const float* __restrict__ core, const float* __restrict__ cam,
                                                         const float* __restrict__ base_focals,
                                                         float* __restrict__ pts, float* __restrict__ zcam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = view_of(view_start, C, i);
    const float* c = cam + v * CAM_STRIDE;
    const float f = c[12], cx = c[13], cy = c[14];
    const float depth = c[15] + c[16] * core[(int64_t)v * G + idxs[i]];
    const float offp = 1.0f + (offsets[i] - 1.0f) * (base_focals[v] / f);
    const float z = depth * offp;
    const float x = (pixels[2 * i] - cx) / f * z, y = (pixels[2 * i + 1] - cy) / f * z;
    pts[3 * i + 0] = (c[0] * x + c[1] * y + c[2] * z) + c[9];
    pts[3 * i + 1] = (c[3] * x + c[4] * y + c[5] * z) + c[10];
    pts[3 * i + 2] = (c[6] * x + c[7] * y + c[8] * z) + c[11];
    zcam[i] = z;
}

__global__ __launch_bounds__(256) void k_dense_clean(int C, int vi, const int32_t* __restrict__ view_start,
                                                     const int32_t* __restrict__ sizes, const float* __restrict__ cam,
                                                     const float* __restrict__ pts, const float* __restrict__ zcam,
                                                     float tol, float bad_conf, float* __restrict__ conf) {
    extern __shared__ float sc[];  // camera rows
    for (int k = threadIdx.x; k < C * CAM_STRIDE; k += blockDim.x) sc[k] = cam[k];
    __syncthreads();
    const int base = view_start[vi];
    const int i = base + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= view_start[vi + 1]) return;
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    float ci = conf[i];
    for (int j = 0; j < C; ++j) {
        if (j == vi) continue;
        const float* c = sc + j * CAM_STRIDE;
        // world -> camera j: R^T (p - T)
        const float dx = px - c[9], dy = py - c[10], dz = pz - c[11];
        const float x = c[0] * dx + c[3] * dy + c[6] * dz;
        const float y = c[1] * dx + c[4] * dy + c[7] * dz;
        const float z = c[2] * dx + c[5] * dy + c[8] * dz;
        if (!(z > 0.0f)) continue;
        const float f = c[12];
        const float uf = rintf((f * x + c[13] * z) / z), vf = rintf((f * y + c[14] * z) / z);  // round half to even
        const int Hj = sizes[2 * j], Wj = sizes[2 * j + 1];
        if (!(uf >= 0.0f && uf < (float)Wj && vf >= 0.0f && vf < (float)Hj)) continue;
        const int k = view_start[j] + (int)vf * Wj + (int)uf;
        const bool bad = (z < (1.0f - tol) * zcam[k]) && (ci < conf[k]);
        if (bad) ci = fminf(ci, bad_conf);
    }
    conf[i] = ci;
}

ST3R_EXPORT int st3r_dense_unproject(st3r_ctx* ctx, void* stream, int C, int G, int n, const int32_t* view_start,
                                     const float* pixels, const int32_t* idxs, const float* offsets,
                                     const float* core_depth, const float* cam, const float* base_focals, float* pts_out,
                                     float* zcam_out) {
    ARG_CHECK(ctx && C > 0 && G > 0 && n >= 0 && view_start && cam && base_focals && core_depth);
    if (n == 0) return ST3R_OK;
    ARG_CHECK(pixels && idxs && offsets && pts_out && zcam_out);
    hipLaunchKernelGGL(k_dense_unproject, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, C, G, n, view_start,
                       pixels, idxs, offsets, core_depth, cam, base_focals, pts_out, zcam_out);
    LAUNCH_CHECK();
    return ST3R_OK;
}

ST3R_EXPORT int st3r_dense_clean(st3r_ctx* ctx, void* stream, int C, int max_view_pixels, const int32_t* view_start,
                                 const int32_t* sizes_hw, const float* cam, const float* pts, const float* zcam,
                                 float tol, float bad_conf, float* conf) {
    ARG_CHECK(ctx && C > 0 && C <= 512 && max_view_pixels >= 0 && view_start && sizes_hw && cam);
    if (max_view_pixels == 0) return ST3R_OK;
    ARG_CHECK(pts && zcam && conf);
    const size_t shmem = sizeof(float) * CAM_STRIDE * (size_t)C;
    for (int vi = 0; vi < C; ++vi) {  // sequential over i: view i reads the already cleaned confidences of j < i
        hipLaunchKernelGGL(k_dense_clean, dim3(ceil_div(max_view_pixels, 256)), dim3(256), shmem, (hipStream_t)stream, C,
                           vi, view_start, sizes_hw, cam, pts, zcam, tol, bad_conf, conf);
    }
    LAUNCH_CHECK();
    return ST3R_OK;
}
