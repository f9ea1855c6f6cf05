/*
 * oracle/gs_oracle.c -- CPU restatement of the 3DGS train-step arithmetic (path C).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under starst3r_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: "parity unpinned vs upstream".  The arithmetic of path C lives in
 * third-party packages that are absent from /root/reference (PyPI `gsplat`, unpinned
 * at reference requirements.txt:1, 1.4.x line at the reference date; PyPI
 * `torchmetrics`, unpinned, requirements.txt:5).  This file restates their published
 * algorithm as reached from the reference call sites:
 *   starster/gs.py:76-87   gsplat.rasterization(means, quats, scales, opacities,
 *                          colors=shN, viewmats, Ks, width, height, sh_degree=1)
 *   starster/gs.py:126-136 compute_loss (L1 + SSIM + two regularisers)
 *   starster/gs.py:37,159-161  torch.optim.Adam (this part IS pinned: run against
 *                          torch.optim.Adam itself, tests/test_oracle_gs.py::test_adam_matches_torch_optim)
 * It is pinned by its own known-answer tests and by fp64 autograd of an independent
 * dense torch restatement (oracle/gs_torch_ref.py) -- see tests/test_oracle_gs.py.
 *
 * Floating-point contract (so integer outputs can be bit-exact against the HIP path):
 *   - all arithmetic is IEEE binary32, compiled with -ffp-contract=off, no fast-math;
 *   - sums of three products are evaluated left to right: (a*b + c*d) + e*f;
 *   - 1/x and sqrt are the correctly rounded operations.
 * Gradient accumulators are double (the reference accumulates with unordered float
 * atomics; the double sum is the value those approximate).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GSO_API __attribute__((visibility("default")))

static inline float dot3f(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

/* quaternion (w,x,y,z), not necessarily unit -> rotation matrix (row major).
 * gsplat quat_to_rotmat [U]; normalisation restated as 1/sqrt (correctly rounded). */
static void quat_to_rotmat(const float* q, float* R, float* qn_out, float* inv_norm_out) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float n2 = ((w * w + x * x) + y * y) + z * z;
    float inv = 1.0f / sqrtf(n2);
    w *= inv; x *= inv; y *= inv; z *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.0f - 2.0f * (y2 + z2); R[1] = 2.0f * (xy - wz);        R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);        R[4] = 1.0f - 2.0f * (x2 + z2); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);        R[7] = 2.0f * (yz + wx);        R[8] = 1.0f - 2.0f * (x2 + y2);
    if (qn_out) { qn_out[0] = w; qn_out[1] = x; qn_out[2] = y; qn_out[3] = z; }
    if (inv_norm_out) *inv_norm_out = inv;
}

/* world covariance (6 unique, row-major upper: 00 01 02 11 12 22) from quat+scale */
static void quat_scale_to_covar(const float* q, const float* s, float* cov6, float* Rq, float* M) {
    float R[9];
    quat_to_rotmat(q, R, NULL, NULL);
    float m[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i * 3 + j] = R[i * 3 + j] * s[j];
    cov6[0] = dot3f(m[0], m[1], m[2], m[0], m[1], m[2]);
    cov6[1] = dot3f(m[0], m[1], m[2], m[3], m[4], m[5]);
    cov6[2] = dot3f(m[0], m[1], m[2], m[6], m[7], m[8]);
    cov6[3] = dot3f(m[3], m[4], m[5], m[3], m[4], m[5]);
    cov6[4] = dot3f(m[3], m[4], m[5], m[6], m[7], m[8]);
    cov6[5] = dot3f(m[6], m[7], m[8], m[6], m[7], m[8]);
    if (Rq) memcpy(Rq, R, sizeof(R));
    if (M) memcpy(M, m, sizeof(m));
}

typedef struct {
    int valid;
    float pc[3];       /* camera-space mean */
    float covc[6];     /* camera-space covariance, upper */
    float J[4];        /* J00, J02, J11, J12 */
    float cov2d[3];    /* after blur: c00 c01 c11 */
    float det;
    float conic[3];
    float mean2d[2];
    int radius;
    int x_clamped, y_clamped;
    float tx, ty, rz;
} proj_t;

/* One (camera, gaussian) projection.  gsplat fully_fused_projection_packed_fwd [U]
 * as reached from starster/gs.py:76; eps2d/near/far/radius_clip are gsplat defaults. */
static void project_one(const float* mean, const float* q, const float* s, const float* V /*4x4 row major*/,
                        const float* K /*3x3*/, int W, int H, float eps2d, float near_plane, float far_plane,
                        float radius_clip, proj_t* o) {
    memset(o, 0, sizeof(*o));
    const float R00 = V[0], R01 = V[1], R02 = V[2], t0 = V[3];
    const float R10 = V[4], R11 = V[5], R12 = V[6], t1 = V[7];
    const float R20 = V[8], R21 = V[9], R22 = V[10], t2 = V[11];
    float x = dot3f(R00, R01, R02, mean[0], mean[1], mean[2]) + t0;
    float y = dot3f(R10, R11, R12, mean[0], mean[1], mean[2]) + t1;
    float z = dot3f(R20, R21, R22, mean[0], mean[1], mean[2]) + t2;
    o->pc[0] = x; o->pc[1] = y; o->pc[2] = z;
    if (z < near_plane || z > far_plane) return;

    float cov[6];
    quat_scale_to_covar(q, s, cov, NULL, NULL);
    /* T = R * cov (full 3x3), covc = T * R^T (upper) */
    const float Rm[9] = {R00, R01, R02, R10, R11, R12, R20, R21, R22};
    const float C[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
    float T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i * 3 + j] = dot3f(Rm[i * 3 + 0], Rm[i * 3 + 1], Rm[i * 3 + 2], C[0 * 3 + j], C[1 * 3 + j], C[2 * 3 + j]);
    float* cc = o->covc;
    cc[0] = dot3f(T[0], T[1], T[2], Rm[0], Rm[1], Rm[2]);
    cc[1] = dot3f(T[0], T[1], T[2], Rm[3], Rm[4], Rm[5]);
    cc[2] = dot3f(T[0], T[1], T[2], Rm[6], Rm[7], Rm[8]);
    cc[3] = dot3f(T[3], T[4], T[5], Rm[3], Rm[4], Rm[5]);
    cc[4] = dot3f(T[3], T[4], T[5], Rm[6], Rm[7], Rm[8]);
    cc[5] = dot3f(T[6], T[7], T[8], Rm[6], Rm[7], Rm[8]);

    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    float tan_fovx = 0.5f * (float)W / fx;
    float tan_fovy = 0.5f * (float)H / fy;
    float lim_x_pos = ((float)W - cx) / fx + 0.3f * tan_fovx;
    float lim_x_neg = cx / fx + 0.3f * tan_fovx;
    float lim_y_pos = ((float)H - cy) / fy + 0.3f * tan_fovy;
    float lim_y_neg = cy / fy + 0.3f * tan_fovy;
    float rz = 1.0f / z;
    float rz2 = rz * rz;
    float xr = x * rz, yr = y * rz;
    float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
    float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
    o->x_clamped = !(xr <= lim_x_pos && xr >= -lim_x_neg);
    o->y_clamped = !(yr <= lim_y_pos && yr >= -lim_y_neg);
    o->tx = tx; o->ty = ty; o->rz = rz;
    float a = fx * rz, c = -(fx * tx) * rz2;
    float b = fy * rz, d = -(fy * ty) * rz2;
    o->J[0] = a; o->J[1] = c; o->J[2] = b; o->J[3] = d;
    float t0x = a * cc[0] + c * cc[2];
    float t0y = a * cc[1] + c * cc[4];
    float t0z = a * cc[2] + c * cc[5];
    float t1y = b * cc[3] + d * cc[4];
    float t1z = b * cc[4] + d * cc[5];
    float c00 = t0x * a + t0z * c;
    float c01 = t0y * b + t0z * d;
    float c11 = t1y * b + t1z * d;
    o->mean2d[0] = (fx * x) * rz + cx;
    o->mean2d[1] = (fy * y) * rz + cy;
    c00 += eps2d; c11 += eps2d;
    float det = c00 * c11 - c01 * c01;
    o->cov2d[0] = c00; o->cov2d[1] = c01; o->cov2d[2] = c11; o->det = det;
    if (det <= 0.0f) return;
    float inv_det = 1.0f / det;
    o->conic[0] = c11 * inv_det;
    o->conic[1] = -c01 * inv_det;
    o->conic[2] = c00 * inv_det;
    float bb = 0.5f * (c00 + c11);
    float v1 = bb + sqrtf(fmaxf(0.01f, bb * bb - det));
    float radius = ceilf(3.0f * sqrtf(v1));
    if (radius <= radius_clip) return;
    if (o->mean2d[0] + radius <= 0.0f || o->mean2d[0] - radius >= (float)W ||
        o->mean2d[1] + radius <= 0.0f || o->mean2d[1] - radius >= (float)H) return;
    o->radius = (int)radius;
    o->valid = 1;
}

/* Packed projection: outputs are flat over visible pairs ordered by camera then gaussian.
 * Returns nnz.  Output arrays must have capacity C*N. */
GSO_API int64_t gso_project_packed(int N, int C, const float* means, const float* quats, const float* scales,
                                   const float* viewmats, const float* Ks, int W, int H, float eps2d,
                                   float near_plane, float far_plane, float radius_clip, int32_t* camera_ids,
                                   int32_t* gaussian_ids, int32_t* radii, float* means2d, float* depths,
                                   float* conics) {
    int64_t n = 0;
    proj_t p;
    for (int c = 0; c < C; ++c) {
        for (int g = 0; g < N; ++g) {
            project_one(means + 3 * g, quats + 4 * g, scales + 3 * g, viewmats + 16 * c, Ks + 9 * c, W, H, eps2d,
                        near_plane, far_plane, radius_clip, &p);
            if (!p.valid) continue;
            camera_ids[n] = c; gaussian_ids[n] = g; radii[n] = p.radius;
            means2d[2 * n] = p.mean2d[0]; means2d[2 * n + 1] = p.mean2d[1];
            depths[n] = p.pc[2];
            conics[3 * n] = p.conic[0]; conics[3 * n + 1] = p.conic[1]; conics[3 * n + 2] = p.conic[2];
            ++n;
        }
    }
    return n;
}

#define SH_C0 0.2820947917738781f
#define SH_C1 0.48860251190292f

/* degree-1 SH colour + clamp_min(c + 0.5, 0).  gsplat spherical_harmonics [U];
 * only rows 0..3 of `sh` are read (sh_degree=1, starster/gs.py:86).
 * campos = inverse(viewmats)[:, :3, 3], computed by the caller. */
GSO_API void gso_sh_colors(int64_t nnz, const int32_t* camera_ids, const int32_t* gaussian_ids, const float* means,
                           const float* campos, const float* sh, int sh_stride /*floats per gaussian*/,
                           float* colors) {
    for (int64_t i = 0; i < nnz; ++i) {
        int c = camera_ids[i], g = gaussian_ids[i];
        float dx = means[3 * g] - campos[3 * c];
        float dy = means[3 * g + 1] - campos[3 * c + 1];
        float dz = means[3 * g + 2] - campos[3 * c + 2];
        float inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
        dx *= inorm; dy *= inorm; dz *= inorm;
        const float* k = sh + (int64_t)g * sh_stride;
        for (int ch = 0; ch < 3; ++ch) {
            float r = SH_C0 * k[ch];
            r = r + SH_C1 * ((-dy * k[3 + ch] + dz * k[6 + ch]) - dx * k[9 + ch]);
            r = r + 0.5f;
            colors[3 * i + ch] = r < 0.0f ? 0.0f : r;
        }
    }
}

static inline int tile_clampi(float v, int hi) {
    /* CUDA (uint32_t)float saturates negatives to 0; then min(max(0,.), hi) */
    if (!(v > 0.0f)) return 0;
    if (v >= (float)hi) return hi;
    return (int)v;
}

static int bit_length_u32(uint32_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

/* gsplat isect_tiles [U].  Pass isect_ids == NULL to only count. Returns n_isects. */
GSO_API int64_t gso_isect_tiles(int64_t nnz, const float* means2d, const int32_t* radii, const float* depths,
                                const int32_t* camera_ids, int tile_size, int tile_w, int tile_h,
                                int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids) {
    int tile_n_bits = bit_length_u32((uint32_t)(tile_w * tile_h));
    int64_t cur = 0;
    for (int64_t i = 0; i < nnz; ++i) {
        if (radii[i] <= 0) { if (tiles_per_gauss) tiles_per_gauss[i] = 0; continue; }
        float tile_radius = (float)radii[i] / (float)tile_size;
        float tile_x = means2d[2 * i] / (float)tile_size;
        float tile_y = means2d[2 * i + 1] / (float)tile_size;
        int x0 = tile_clampi(floorf(tile_x - tile_radius), tile_w);
        int y0 = tile_clampi(floorf(tile_y - tile_radius), tile_h);
        int x1 = tile_clampi(ceilf(tile_x + tile_radius), tile_w);
        int y1 = tile_clampi(ceilf(tile_y + tile_radius), tile_h);
        int cnt = (y1 - y0) * (x1 - x0);
        if (tiles_per_gauss) tiles_per_gauss[i] = cnt;
        if (isect_ids) {
            int64_t cid_enc = (int64_t)camera_ids[i] << (32 + tile_n_bits);
            int32_t dbits; memcpy(&dbits, &depths[i], 4);
            int64_t depth_enc = (int64_t)(uint32_t)dbits; /* positive depths: sign bit clear */
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    int64_t tile_id = (int64_t)ty * tile_w + tx;
                    isect_ids[cur] = cid_enc | (tile_id << 32) | depth_enc;
                    flatten_ids[cur] = (int32_t)i;
                    ++cur;
                }
        } else cur += cnt;
    }
    return cur;
}

/* stable sort of (key,val) by key ascending -- LSD radix, 8-bit digits (cub SortPairs [U]) */
GSO_API void gso_sort_pairs(int64_t n, int64_t* keys, int32_t* vals) {
    if (n <= 1) return;
    int64_t* k2 = (int64_t*)malloc(sizeof(int64_t) * n);
    int32_t* v2 = (int32_t*)malloc(sizeof(int32_t) * n);
    int64_t *ka = keys, *kb = k2; int32_t *va = vals, *vb = v2;
    for (int pass = 0; pass < 8; ++pass) {
        int64_t hist[257]; memset(hist, 0, sizeof(hist));
        int shift = pass * 8;
        for (int64_t i = 0; i < n; ++i) hist[((uint64_t)ka[i] >> shift & 0xff) + 1]++;
        if (hist[1] == n) continue; /* all digits zero: nothing moves */
        for (int d = 0; d < 256; ++d) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < n; ++i) {
            int64_t p = hist[(uint64_t)ka[i] >> shift & 0xff]++;
            kb[p] = ka[i]; vb[p] = va[i];
        }
        int64_t* tk = ka; ka = kb; kb = tk; int32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys) { memcpy(keys, ka, sizeof(int64_t) * n); memcpy(vals, va, sizeof(int32_t) * n); }
    free(k2); free(v2);
}

/* gsplat isect_offset_encode [U]: offsets[c*T + t] = first sorted position with (cam,tile) >= (c,t) */
GSO_API void gso_isect_offsets(int64_t n_isects, const int64_t* sorted_ids, int C, int tile_w, int tile_h,
                               int32_t* offsets) {
    int n_tiles = tile_w * tile_h;
    int tile_n_bits = bit_length_u32((uint32_t)n_tiles);
    int64_t total = (int64_t)C * n_tiles;
    int64_t next = 0; /* next (cam,tile) id whose offset is still unset */
    for (int64_t i = 0; i < n_isects; ++i) {
        int64_t hi = sorted_ids[i] >> 32;
        int64_t cid = hi >> tile_n_bits;
        int64_t tid = hi & (((int64_t)1 << tile_n_bits) - 1);
        int64_t id = cid * n_tiles + tid;
        while (next <= id) offsets[next++] = (int32_t)i;
    }
    while (next < total) offsets[next++] = (int32_t)n_isects;
}

/* gsplat rasterize_to_pixels fwd [U].  margin (optional, may be NULL): per pixel, how far the pixel is from being
 * undetermined in float32 arithmetic -- tests exclude pixels with margin <= 1e-4 and require the rest to agree to 1e-4:
 *   - the smallest relative distance of any skip/stop decision to its threshold (a 1-ulp exp difference may flip it);
 *   - sigma = (a dx^2 + c dy^2)/2 + b dx dy cancels catastrophically for strongly anisotropic Gaussians far from their
 *     mean: two float evaluations (this one, gsplat's fma-contracted one, the q-form of the HIP kernel) differ by
 *     ~2 ulp of the largest term, es = 2^-22 (|a dx^2| + |c dy^2| + 2 |b dx dy|), i.e. alpha is only known to a
 *     relative es.  Decisions are judged against es (distance / max(1, es / 2.5e-5)), and a pixel whose accumulated
 *     value uncertainty sum(vis * es) exceeds 2e-5 gets margin 0. */
GSO_API void gso_blend_fwd(int C, int W, int H, int tile_size, int tile_w, int tile_h, const float* means2d,
                           const float* conics, const float* colors, const float* opacities,
                           const int32_t* offsets, const int32_t* flatten_ids, int64_t n_isects, float* out_rgb,
                           float* out_alpha, int32_t* last_ids, float* margin) {
    int n_tiles = tile_w * tile_h;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                int tile_id = (i / tile_size) * tile_w + (j / tile_size);
                int64_t gt = (int64_t)c * n_tiles + tile_id;
                int64_t start = offsets[gt];
                int64_t end = (gt == (int64_t)C * n_tiles - 1) ? n_isects : offsets[gt + 1];
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T = 1.0f, r = 0.f, g = 0.f, b = 0.f;
                int32_t cur = 0;
                float mg = 1e30f, uT = 0.f, uval = 0.f;
                for (int64_t k = start; k < end; ++k) {
                    int32_t id = flatten_ids[k];
                    float dx = means2d[2 * id] - px, dy = means2d[2 * id + 1] - py;
                    float ca = conics[3 * id], cb = conics[3 * id + 1], cc = conics[3 * id + 2];
                    float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    float alpha = fminf(0.999f, opacities[id] * expf(-sigma));
                    float es = 0.f;
                    if (margin) {
                        es = 5.9604644775390625e-08f * (fabsf(ca * dx * dx) + fabsf(cc * dy * dy) + 2.f * fabsf(cb * dx * dy));
                        float loosen = fmaxf(1.f, es / 2.5e-5f);
                        float m1 = fabsf(alpha - (1.f / 255.f)) * 255.f / loosen;
                        if (m1 < mg) mg = m1;
                        float m0 = fabsf(sigma); /* sigma<0 decision; absolute */
                        if (m0 < fmaxf(1e-6f, es) && m0 < mg) mg = m0;
                    }
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    float nT = T * (1.0f - alpha);
                    if (margin) {
                        uT += alpha < 0.999f ? alpha * es / (1.0f - alpha) : 0.f;   /* relative uncertainty of T */
                        float m2 = fabsf(nT - 1e-4f) * 1e4f / fmaxf(1.f, uT / 2.5e-5f);
                        if (m2 < mg) mg = m2;
                        uval += alpha * T * (es + uT);
                    }
                    if (nT <= 1e-4f) break;
                    float vis = alpha * T;
                    r += colors[3 * id] * vis; g += colors[3 * id + 1] * vis; b += colors[3 * id + 2] * vis;
                    cur = (int32_t)k;
                    T = nT;
                }
                int64_t p = ((int64_t)c * H + i) * W + j;
                out_rgb[3 * p] = r; out_rgb[3 * p + 1] = g; out_rgb[3 * p + 2] = b;
                out_alpha[p] = 1.0f - T;
                last_ids[p] = cur;
                if (margin) margin[p] = uval > 5e-5f ? 0.f : mg;
            }
}

/* gsplat rasterize_to_pixels bwd [U].  v_alpha may be NULL (== 0: starster/gs.py:126
 * never reads render_alpha).  Outputs are double accumulators over the packed arrays. */
GSO_API void gso_blend_bwd(int C, int W, int H, int tile_size, int tile_w, int tile_h, const float* means2d,
                           const float* conics, const float* colors, const float* opacities,
                           const int32_t* offsets, const int32_t* flatten_ids, int64_t n_isects,
                           const float* out_alpha, const int32_t* last_ids, const float* v_rgb,
                           const float* v_alpha, double* v_means2d, double* v_conics, double* v_colors,
                           double* v_opacities) {
    int n_tiles = tile_w * tile_h;
    (void)n_isects;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                int tile_id = (i / tile_size) * tile_w + (j / tile_size);
                int64_t gt = (int64_t)c * n_tiles + tile_id;
                int64_t start = offsets[gt];
                int64_t p = ((int64_t)c * H + i) * W + j;
                int64_t bin_final = last_ids[p];
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T_final = 1.0f - out_alpha[p];
                float T = T_final;
                float buf[3] = {0.f, 0.f, 0.f};
                const float vr[3] = {v_rgb[3 * p], v_rgb[3 * p + 1], v_rgb[3 * p + 2]};
                float va = v_alpha ? v_alpha[p] : 0.f;
                for (int64_t k = bin_final; k >= start; --k) {
                    int32_t id = flatten_ids[k];
                    float dx = means2d[2 * id] - px, dy = means2d[2 * id + 1] - py;
                    float ca = conics[3 * id], cb = conics[3 * id + 1], cc = conics[3 * id + 2];
                    float opac = opacities[id];
                    float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    float vis = expf(-sigma);
                    float alpha = fminf(0.999f, opac * vis);
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    float fac = alpha * T;
                    float v_al = 0.f;
                    for (int ch = 0; ch < 3; ++ch) {
                        v_colors[3 * (int64_t)id + ch] += (double)(fac * vr[ch]);
                        v_al += (colors[3 * id + ch] * T - buf[ch] * ra) * vr[ch];
                    }
                    v_al += T_final * ra * va;
                    if (opac * vis <= 0.999f) {
                        float v_sigma = -opac * vis * v_al;
                        v_conics[3 * (int64_t)id] += (double)(0.5f * v_sigma * dx * dx);
                        v_conics[3 * (int64_t)id + 1] += (double)(v_sigma * dx * dy);
                        v_conics[3 * (int64_t)id + 2] += (double)(0.5f * v_sigma * dy * dy);
                        v_means2d[2 * (int64_t)id] += (double)(v_sigma * (ca * dx + cb * dy));
                        v_means2d[2 * (int64_t)id + 1] += (double)(v_sigma * (cb * dx + cc * dy));
                        v_opacities[id] += (double)(vis * v_al);
                    }
                    for (int ch = 0; ch < 3; ++ch) buf[ch] += colors[3 * id + ch] * fac;
                }
            }
}

/* Backward of SH colour + projection for every packed pair, accumulated per gaussian.
 * Restates gsplat spherical_harmonics bwd (with v_dirs: dirs depends on means) and
 * fully_fused_projection_packed_bwd [U]; viewmats receive no gradient (not parameters,
 * starster/gs.py:33-34).  Outputs: v_means[N,3] v_quats[N,4] v_scales[N,3] v_opac[N]
 * v_sh[N,4,3] (rows 0..3; rows 4..23 are identically zero) -- all double. */
GSO_API void gso_project_sh_bwd(int64_t nnz, int N, const int32_t* camera_ids, const int32_t* gaussian_ids,
                                const float* means, const float* quats, const float* scales, const float* sh,
                                int sh_stride, const float* viewmats, const float* Ks, const float* campos, int W,
                                int H, float eps2d, const double* v_means2d, const double* v_conics,
                                const double* v_colors, const double* v_opac_packed, double* v_means,
                                double* v_quats, double* v_scales, double* v_opac, double* v_sh) {
    (void)N;
    for (int64_t i = 0; i < nnz; ++i) {
        int c = camera_ids[i], g = gaussian_ids[i];
        const float* V = viewmats + 16 * c;
        const float* K = Ks + 9 * c;
        const float* mean = means + 3 * g;
        /* ---- SH backward ---- */
        {
            double dx = mean[0] - campos[3 * c], dy = mean[1] - campos[3 * c + 1], dz = mean[2] - campos[3 * c + 2];
            double nrm = sqrt(dx * dx + dy * dy + dz * dz);
            double ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
            const float* k = sh + (int64_t)g * sh_stride;
            double vdn[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
                /* forward value in float to reproduce the clamp mask */
                float fdx = mean[0] - campos[3 * c], fdy = mean[1] - campos[3 * c + 1], fdz = mean[2] - campos[3 * c + 2];
                float inorm = 1.0f / sqrtf((fdx * fdx + fdy * fdy) + fdz * fdz);
                fdx *= inorm; fdy *= inorm; fdz *= inorm;
                float r = SH_C0 * k[ch];
                r = r + SH_C1 * ((-fdy * k[3 + ch] + fdz * k[6 + ch]) - fdx * k[9 + ch]);
                r = r + 0.5f;
                double vc = (r >= 0.0f) ? v_colors[3 * i + ch] : 0.0; /* clamp_min passes grad where input >= min */
                v_sh[(int64_t)g * 12 + 0 + ch] += (double)SH_C0 * vc;
                v_sh[(int64_t)g * 12 + 3 + ch] += -(double)SH_C1 * uy * vc;
                v_sh[(int64_t)g * 12 + 6 + ch] += (double)SH_C1 * uz * vc;
                v_sh[(int64_t)g * 12 + 9 + ch] += -(double)SH_C1 * ux * vc;
                vdn[0] += -(double)SH_C1 * k[9 + ch] * vc;
                vdn[1] += -(double)SH_C1 * k[3 + ch] * vc;
                vdn[2] += (double)SH_C1 * k[6 + ch] * vc;
            }
            double dotp = vdn[0] * ux + vdn[1] * uy + vdn[2] * uz;
            v_means[3 * (int64_t)g + 0] += (vdn[0] - dotp * ux) / nrm;
            v_means[3 * (int64_t)g + 1] += (vdn[1] - dotp * uy) / nrm;
            v_means[3 * (int64_t)g + 2] += (vdn[2] - dotp * uz) / nrm;
        }
        v_opac[g] += v_opac_packed[i];
        /* ---- projection backward (double math on the float forward state) ---- */
        proj_t p;
        project_one(mean, quats + 4 * g, scales + 3 * g, V, K, W, H, eps2d, 0.0f, 3.0e38f, -1.0f, &p);
        const double fx = K[0], fy = K[4];
        /* conic -> cov2d: v_Sigma = -Sinv * Vc * Sinv */
        double A = p.conic[0], B = p.conic[1], Cc = p.conic[2];
        double vA = v_conics[3 * i], vB = 0.5 * v_conics[3 * i + 1], vC = v_conics[3 * i + 2];
        /* X = Sinv*Vc */
        double X00 = A * vA + B * vB, X01 = A * vB + B * vC;
        double X10 = B * vA + Cc * vB, X11 = B * vB + Cc * vC;
        double G00 = -(X00 * A + X01 * B), G01 = -(X00 * B + X01 * Cc);
        double G10 = -(X10 * A + X11 * B), G11 = -(X10 * B + X11 * Cc);
        double g01 = 0.5 * (G01 + G10); /* symmetric part: gradient wrt each off-diagonal entry */
        /* cov2d = J S J^T ; J = [[a,0,c],[0,b,d]] */
        double a = p.J[0], cj = p.J[1], b = p.J[2], d = p.J[3];
        const float* cc = p.covc;
        double S[3][3] = {{cc[0], cc[1], cc[2]}, {cc[1], cc[3], cc[4]}, {cc[2], cc[4], cc[5]}};
        double Jm[2][3] = {{a, 0, cj}, {0, b, d}};
        double Gm[2][2] = {{G00, g01}, {g01, G11}};
        /* v_S = J^T G J */
        double vS[3][3];
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) {
                double acc = 0;
                for (int u = 0; u < 2; ++u)
                    for (int v = 0; v < 2; ++v) acc += Jm[u][r] * Gm[u][v] * Jm[v][s];
                vS[r][s] = acc;
            }
        /* v_J = 2 G J S  (G, S symmetric) */
        double vJ[2][3];
        for (int u = 0; u < 2; ++u)
            for (int s = 0; s < 3; ++s) {
                double acc = 0;
                for (int v = 0; v < 2; ++v)
                    for (int r = 0; r < 3; ++r) acc += Gm[u][v] * Jm[v][r] * S[r][s];
                vJ[u][s] = 2.0 * acc;
            }
        double x = p.pc[0], y = p.pc[1], z = p.pc[2];
        double rz = 1.0 / z, rz2 = rz * rz, rz3 = rz2 * rz;
        double vm2x = v_means2d[2 * i], vm2y = v_means2d[2 * i + 1];
        double vpc[3];
        vpc[0] = fx * rz * vm2x;
        vpc[1] = fy * rz * vm2y;
        vpc[2] = -(fx * x * vm2x + fy * y * vm2y) * rz2;
        /* J00 = fx/z, J11 = fy/z, J02 = -fx*tx/z^2, J12 = -fy*ty/z^2 */
        vpc[2] += -fx * rz2 * vJ[0][0] - fy * rz2 * vJ[1][1];
        if (!p.x_clamped) { vpc[0] += -fx * rz2 * vJ[0][2]; vpc[2] += 2.0 * fx * x * rz3 * vJ[0][2]; }
        else { vpc[2] += fx * (double)p.tx * rz3 * vJ[0][2]; }
        if (!p.y_clamped) { vpc[1] += -fy * rz2 * vJ[1][2]; vpc[2] += 2.0 * fy * y * rz3 * vJ[1][2]; }
        else { vpc[2] += fy * (double)p.ty * rz3 * vJ[1][2]; }
        /* world: p_c = R m + t ; S_c = R S_w R^T */
        double R[3][3] = {{V[0], V[1], V[2]}, {V[4], V[5], V[6]}, {V[8], V[9], V[10]}};
        for (int r = 0; r < 3; ++r) {
            double acc = 0;
            for (int u = 0; u < 3; ++u) acc += R[u][r] * vpc[u];
            v_means[3 * (int64_t)g + r] += acc;
        }
        double vSw[3][3];
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) {
                double acc = 0;
                for (int u = 0; u < 3; ++u)
                    for (int v = 0; v < 3; ++v) acc += R[u][r] * vS[u][v] * R[v][s];
                vSw[r][s] = acc;
            }
        /* S_w = M M^T, M = Rq diag(s): v_M = (vSw + vSw^T) M */
        float Rqf[9], Mf[9], cov6[6], qn[4], inv_norm;
        quat_scale_to_covar(quats + 4 * g, scales + 3 * g, cov6, Rqf, Mf);
        quat_to_rotmat(quats + 4 * g, Rqf, qn, &inv_norm);
        double vM[3][3];
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) {
                double acc = 0;
                for (int u = 0; u < 3; ++u) acc += (vSw[r][u] + vSw[u][r]) * Mf[u * 3 + s];
                vM[r][s] = acc;
            }
        const float* sc = scales + 3 * g;
        double vR[3][3];
        for (int s = 0; s < 3; ++s) {
            double acc = 0;
            for (int r = 0; r < 3; ++r) { acc += Rqf[r * 3 + s] * vM[r][s]; vR[r][s] = vM[r][s] * sc[s]; }
            v_scales[3 * (int64_t)g + s] += acc;
        }
        double w = qn[0], qx = qn[1], qy = qn[2], qz = qn[3];
        double vq[4];
        vq[0] = 2.0 * (qx * (vR[2][1] - vR[1][2]) + qy * (vR[0][2] - vR[2][0]) + qz * (vR[1][0] - vR[0][1]));
        vq[1] = 2.0 * (-2.0 * qx * (vR[1][1] + vR[2][2]) + qy * (vR[1][0] + vR[0][1]) + qz * (vR[2][0] + vR[0][2]) +
                       w * (vR[2][1] - vR[1][2]));
        vq[2] = 2.0 * (qx * (vR[1][0] + vR[0][1]) - 2.0 * qy * (vR[0][0] + vR[2][2]) + qz * (vR[2][1] + vR[1][2]) +
                       w * (vR[0][2] - vR[2][0]));
        vq[3] = 2.0 * (qx * (vR[2][0] + vR[0][2]) + qy * (vR[2][1] + vR[1][2]) - 2.0 * qz * (vR[0][0] + vR[1][1]) +
                       w * (vR[1][0] - vR[0][1]));
        double dq = vq[0] * w + vq[1] * qx + vq[2] * qy + vq[3] * qz;
        v_quats[4 * (int64_t)g + 0] += (vq[0] - dq * w) * inv_norm;
        v_quats[4 * (int64_t)g + 1] += (vq[1] - dq * qx) * inv_norm;
        v_quats[4 * (int64_t)g + 2] += (vq[2] - dq * qy) * inv_norm;
        v_quats[4 * (int64_t)g + 3] += (vq[3] - dq * qz) * inv_norm;
    }
}

/* ---------------------------------------------------------------------------------
 * L1 + SSIM of one view.  starster/gs.py:126-130 with torchmetrics
 * StructuralSimilarityIndexMeasure(data_range=1) defaults [U]: gaussian 11x11 sigma 1.5,
 * k1=.01 k2=.03, reflect-pad 5 then crop 5 => mean over the interior (H-10)x(W-10) of
 * the un-padded "valid" convolution (padding never reaches the kept region).
 * img layout [H,W,3].  Returns l1 and ssim means; if v_render != NULL also writes
 * d( w_l1*l1 + w_ssim*(1-ssim) ) / d render.
 * --------------------------------------------------------------------------------- */
GSO_API void gso_l1_ssim(int H, int W, const float* render, const float* gt, double w_l1, double w_ssim,
                         double* out_l1, double* out_ssim, float* v_render) {
    const int KS = 11, HALF = 5;
    double g1[11], gs = 0;
    for (int i = 0; i < KS; ++i) { double d = (i - HALF) / 1.5; g1[i] = exp(-0.5 * d * d); gs += g1[i]; }
    for (int i = 0; i < KS; ++i) g1[i] /= gs;
    float gw[11]; /* torchmetrics builds the window in float32 */
    for (int i = 0; i < KS; ++i) gw[i] = (float)g1[i];
    const double c1 = 0.01 * 0.01, c2 = 0.03 * 0.03;
    int64_t npx = (int64_t)H * W;
    double l1 = 0;
    for (int64_t i = 0; i < npx * 3; ++i) l1 += fabs((double)gt[i] - (double)render[i]);
    l1 /= (double)(npx * 3);
    int Hi = H - 2 * HALF, Wi = W - 2 * HALF;
    double ssim_sum = 0;
    int64_t cnt = (Hi > 0 && Wi > 0) ? (int64_t)Hi * Wi * 3 : 0;
    double *dA = NULL, *dB = NULL, *dC = NULL;
    if (v_render) {
        dA = (double*)calloc(npx * 3, sizeof(double));
        dB = (double*)calloc(npx * 3, sizeof(double));
        dC = (double*)calloc(npx * 3, sizeof(double));
    }
    /* separable: horizontal pass into temp rows, then vertical */
    double* tmp = (double*)malloc(sizeof(double) * 5 * (size_t)H * (Wi > 0 ? Wi : 1));
    for (int ch = 0; ch < 3 && cnt; ++ch) {
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < Wi; ++j) {
                double s[5] = {0, 0, 0, 0, 0};
                for (int k = 0; k < KS; ++k) {
                    double x = render[((int64_t)i * W + j + k) * 3 + ch], y = gt[((int64_t)i * W + j + k) * 3 + ch];
                    double w = gw[k];
                    s[0] += w * x; s[1] += w * y; s[2] += w * x * x; s[3] += w * y * y; s[4] += w * x * y;
                }
                for (int m = 0; m < 5; ++m) tmp[((size_t)m * H + i) * Wi + j] = s[m];
            }
        for (int i = 0; i < Hi; ++i)
            for (int j = 0; j < Wi; ++j) {
                double s[5] = {0, 0, 0, 0, 0};
                for (int k = 0; k < KS; ++k)
                    for (int m = 0; m < 5; ++m) s[m] += gw[k] * tmp[((size_t)m * H + i + k) * Wi + j];
                double mx = s[0], my = s[1];
                /* torchmetrics clamps both variances at 0 (torch.clamp(E[x^2] - mu^2, min=0)) [U] */
                double sxx_raw = s[2] - mx * mx;
                double sxx = sxx_raw < 0 ? 0 : sxx_raw, syy = s[3] - my * my, sxy = s[4] - mx * my;
                if (syy < 0) syy = 0;
                double n1 = 2 * mx * my + c1, n2 = 2 * sxy + c2, d1 = mx * mx + my * my + c1, d2 = sxx + syy + c2;
                double ssim = (n1 * n2) / (d1 * d2);
                ssim_sum += ssim;
                if (v_render) {
                    /* d ssim / d(mx), d(Exx), d(Exy) with Exx = E[x^2], Exy = E[xy]; y constant */
                    double dn1 = n2 / (d1 * d2), dn2 = n1 / (d1 * d2);
                    double dd1 = -ssim / d1, dd2 = sxx_raw < 0 ? 0.0 : -ssim / d2;   /* clamped: no gradient */
                    /* n1 = 2 mx my + c1 ; n2 = 2(Exy - mx my) + c2 ; d1 = mx^2+my^2+c1 ; d2 = Exx - mx^2 + Eyy - my^2 + c2 */
                    double d_mx = dn1 * 2 * my + dn2 * (-2 * my) + dd1 * 2 * mx + dd2 * (-2 * mx);
                    double d_exx = dd2;
                    double d_exy = dn2 * 2;
                    int64_t q = ((int64_t)(i + HALF) * W + (j + HALF)) * 3 + ch;
                    dA[q] = d_mx; dB[q] = d_exx; dC[q] = d_exy;
                }
            }
    }
    free(tmp);
    double ssim = cnt ? ssim_sum / (double)cnt : 0.0;
    *out_l1 = l1; *out_ssim = ssim;
    if (v_render) {
        double kl1 = w_l1 / (double)(npx * 3);
        double kss = cnt ? -w_ssim / (double)cnt : 0.0; /* loss has (1 - ssim) */
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                for (int ch = 0; ch < 3; ++ch) {
                    int64_t p = ((int64_t)i * W + j) * 3 + ch;
                    double x = render[p], y = gt[p];
                    double acc = 0;
                    /* correlation with the (symmetric) window over interior outputs q */
                    for (int di = -HALF; di <= HALF; ++di) {
                        int qi = i + di; if (qi < HALF || qi >= H - HALF) continue;
                        for (int dj = -HALF; dj <= HALF; ++dj) {
                            int qj = j + dj; if (qj < HALF || qj >= W - HALF) continue;
                            double w = (double)gw[di + HALF] * (double)gw[dj + HALF];
                            int64_t q = ((int64_t)qi * W + qj) * 3 + ch;
                            acc += w * (dA[q] + 2 * x * dB[q] + y * dC[q]);
                        }
                    }
                    double sgn = (x > y) ? 1.0 : ((x < y) ? -1.0 : 0.0);
                    v_render[p] = (float)(kl1 * sgn + kss * acc);
                }
        free(dA); free(dB); free(dC);
    }
}

/* torch.optim.Adam single-tensor update (lr, betas, eps; no weight decay, no amsgrad),
 * starster/gs.py:37,159-161.  `step` is the 1-based step count AFTER increment.
 * Scalars are doubles exactly as Python hands them to torch: `1 - beta` is formed in
 * double and only then rounded to float (torch casts the scalar to the tensor dtype);
 * exp_avg uses lerp_ == fma(w, g - m, m) (ATen lerp, weight < 0.5 branch). */
GSO_API void gso_adam(int64_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2,
                      double eps, int step) {
    double bc1 = 1.0 - pow(b1, (double)step);
    double bc2 = 1.0 - pow(b2, (double)step);
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    float w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2), fb2 = (float)b2, feps = (float)eps;
    for (int64_t i = 0; i < n; ++i) {
        float gi = g[i];
        m[i] = fmaf(w1, gi - m[i], m[i]);                 /* lerp_ */
        float vv = v[i] * fb2;                            /* mul_ */
        v[i] = vv + (w2 * gi) * gi;                       /* addcmul_ */
        float denom = sqrtf(v[i]) / bc2_sqrt + feps;
        p[i] = p[i] - step_size * (m[i] / denom);         /* addcdiv_ */
    }
}
