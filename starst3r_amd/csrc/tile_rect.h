// Tile rectangles of a projected Gaussian -- shared by the kernels that must agree on them exactly
// (tile count in k_project_sh_fwd, emission in k_isect_emit*, slot index in k_blend_bwd).
#pragma once
#include <hip/hip_runtime.h>

struct TileRect { int x0, y0, x1, y1; };  // [x0,x1) x [y0,y1) in tile units

__device__ __forceinline__ int tile_clampi(float v, int hi) {
    // CUDA's float -> uint32 conversion saturates negatives to 0; then min(max(0, .), hi)
    if (!(v > 0.0f)) return 0;
    if (v >= (float)hi) return hi;
    return (int)v;
}

// gsplat isect_tiles: tiles overlapped by the square [mean2d - radius, mean2d + radius] (tile size 16)
__device__ __forceinline__ TileRect ref_tile_rect(float x, float y, float radius, int tile_size, int tile_w,
                                                  int tile_h) {
    const float tile_radius = radius / (float)tile_size;
    const float tile_x = x / (float)tile_size, tile_y = y / (float)tile_size;
    TileRect r;
    r.x0 = tile_clampi(floorf(tile_x - tile_radius), tile_w);
    r.y0 = tile_clampi(floorf(tile_y - tile_radius), tile_h);
    r.x1 = tile_clampi(ceilf(tile_x + tile_radius), tile_w);
    r.y1 = tile_clampi(ceilf(tile_y + tile_radius), tile_h);
    return r;
}

// Half extents of the axis-aligned box around {p : opacity * exp(-sigma(p)) >= 1/255}, slightly inflated
// (tau by 2e-4 relative + 1e-4, extents by `slack` pixels) so that every pixel OUTSIDE the box fails the
// reference's alpha test in float arithmetic too.  Returns false when the Gaussian can never pass the test.
__device__ __forceinline__ bool influence_extent(float opac, float ca, float cb, float cc, float slack, float* ex,
                                                 float* ey) {
#pragma clang fp contract(off)
    const float o255 = 255.0f * opac;
    if (!(o255 > 1.0f)) return false;
    // single hardware instructions (v_log_f32, v_rcp_f32, v_sqrt_f32: ~1 ulp, identical in every kernel that
    // includes this header); their error is orders of magnitude below the inflation
    const float det = ca * cc - cb * cb;
    const float rdet = __builtin_amdgcn_rcpf(det);
    // sigma cancels catastrophically for strongly correlated conics: on the contour sigma = tau its terms reach
    // tau * ca*cc/det, so two float evaluations of sigma differ by ~1e-6 of that -- the threshold is widened by as much
    const float tau = __logf(o255) * (1.0002f + 4e-6f * (ca * cc) * rdet) + 1e-4f;
    const float k = 2.0f * tau * rdet;
    *ex = __builtin_amdgcn_sqrtf(k * cc) * 1.0001f + slack;
    *ey = __builtin_amdgcn_sqrtf(k * ca) * 1.0001f + slack;
    return true;
}

// Reference rectangle intersected with the tiles whose pixel centres the influence box can reach (16-pixel
// tiles: centres 16*t + 0.5 ... 16*t + 15.5).  Used only by the fused train path: the dropped (record, tile)
// pairs fail the alpha test on all 256 pixels, so images and gradients are unchanged.  The slack (0.05 px) is
// larger than the one of the per-quadrant test in k_blend_* (0.02 px): a superset of what blending needs.
__device__ __forceinline__ TileRect tight_tile_rect(TileRect r, float x, float y, float opac, float ca, float cb,
                                                    float cc) {
#pragma clang fp contract(off)
    float ex, ey;
    if (!influence_extent(opac, ca, cb, cc, 0.05f, &ex, &ey)) { r.x1 = r.x0; r.y1 = r.y0; return r; }
    const int tx_lo = (int)fmaxf(ceilf((x - ex - 15.5f) * 0.0625f), -1.0f);
    const int tx_hi = (int)fminf(floorf((x + ex - 0.5f) * 0.0625f), 1.0e6f);
    const int ty_lo = (int)fmaxf(ceilf((y - ey - 15.5f) * 0.0625f), -1.0f);
    const int ty_hi = (int)fminf(floorf((y + ey - 0.5f) * 0.0625f), 1.0e6f);
    r.x0 = max(r.x0, tx_lo); r.x1 = min(r.x1, tx_hi + 1);
    r.y0 = max(r.y0, ty_lo); r.y1 = min(r.y1, ty_hi + 1);
    if (r.x1 < r.x0) r.x1 = r.x0;
    if (r.y1 < r.y0) r.y1 = r.y0;
    return r;
}

// x0 | y0 << 16 | width << 32 | height << 48 (all < 65536 tiles); 0 for an empty rectangle
__device__ __forceinline__ uint64_t pack_rect(TileRect r) {
    const uint64_t w = (uint64_t)(r.x1 - r.x0), h = (uint64_t)(r.y1 - r.y0);
    if (w == 0 || h == 0) return 0;
    return (uint64_t)r.x0 | ((uint64_t)r.y0 << 16) | (w << 32) | (h << 48);
}

// The fused path keeps one packed rectangle per (camera, Gaussian) pair and gathers it by pair id (emit kernel): half
// the bytes per entry is half the cache footprint of that gather, so tile grids of up to 255 x 255 tiles (images up
// to 4080 pixels a side: every BASELINE configuration) use a 32-bit form, `is32` = 1:
//   x0 | y0 << 8 | w << 16 | h << 24
// (Round 5 also carried a 9-bit mask of the tiles the exact ellipse test keeps in rectangles of at most 3 x 3 tiles:
// 6 % fewer records on SYNTH-1M, measured not faster -- tools/experiments/README.md; removed in round 6.)
__device__ __forceinline__ uint32_t pack_rect32(TileRect r) {
    const uint32_t w = (uint32_t)(r.x1 - r.x0), h = (uint32_t)(r.y1 - r.y0);
    if (w == 0 || h == 0) return 0u;
    return (uint32_t)r.x0 | ((uint32_t)r.y0 << 8) | (w << 16) | (h << 24);
}
__device__ __forceinline__ void rect_store(void* rects, int is32, int64_t i, TileRect r) {
    if (is32) reinterpret_cast<uint32_t*>(rects)[i] = pack_rect32(r);
    else reinterpret_cast<uint64_t*>(rects)[i] = pack_rect(r);
}
// entry i of either form: origin x0 | y0 << 16, width, height (is32 is uniform over the launch)
__device__ __forceinline__ void rect_load(const void* rects, int is32, int64_t i, uint32_t* org, uint32_t* w, uint32_t* h) {
    if (is32) {
        const uint32_t r = reinterpret_cast<const uint32_t*>(rects)[i];
        *org = (r & 0xFFu) | ((r & 0xFF00u) << 8); *w = (r >> 16) & 0xFFu; *h = r >> 24;
    } else {
        const uint64_t r = reinterpret_cast<const uint64_t*>(rects)[i];
        *org = (uint32_t)(r & 0xFFFFFFFFull); *w = (uint32_t)((r >> 32) & 0xFFFF); *h = (uint32_t)(r >> 48);
    }
}

// ---- exact culling: does the ellipse {sigma(p - mean) <= tau} reach a square of pixel centres? ----
// sigma(d) = (A dx^2 + C dy^2)/2 + B dx dy with the conic (A, B, C).  sigma is convex, so when the mean lies
// outside the square its minimum over the square sits on one of the four edges; on an edge it is a 1-D quadratic
// whose minimiser is clamped to the edge.  tau = ln(255 opacity) inflated by 2e-4 relative + 2e-4: a square
// declared unreachable fails the alpha test of the blend loop on every pixel in float arithmetic as well.
// (Used for the per-quadrant relevance of the blend kernels.  Culling whole tiles with it as well was tried:
// 10 % fewer records, but the count / emission / slot-ordinal loops cost more than that saved.)
struct EllipseTest { float A, B, C, kyx, kxy, tau; };

__device__ __forceinline__ bool ellipse_prepare(float opac, float ca, float cb, float cc, EllipseTest* e) {
#pragma clang fp contract(off)
    const float o255 = 255.0f * opac;
    if (!(o255 > 1.0f)) return false;
    // (the anisotropy term: see influence_extent)
    e->tau = __logf(o255) * (1.0002f + 4e-6f * (ca * cc) * __builtin_amdgcn_rcpf(ca * cc - cb * cb)) + 2e-4f;
    e->A = ca; e->B = cb; e->C = cc;
    e->kyx = -cb * __builtin_amdgcn_rcpf(cc);
    e->kxy = -cb * __builtin_amdgcn_rcpf(ca);
    return true;
}

// square [x0, x1] x [y0, y1] given relative to the mean: dx0 = x0 - mean_x, ...
__device__ __forceinline__ bool ellipse_hits_square(const EllipseTest& e, float dx0, float dx1, float dy0, float dy1) {
#pragma clang fp contract(off)
    bool hit = (dx0 <= 0.0f) && (dx1 >= 0.0f) && (dy0 <= 0.0f) && (dy1 >= 0.0f);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float dx = k ? dx1 : dx0;  // vertical edge
        const float dy = fminf(fmaxf(e.kyx * dx, dy0), dy1);
        hit |= (0.5f * (e.A * dx * dx + e.C * dy * dy) + e.B * dx * dy) <= e.tau;
        const float ey = k ? dy1 : dy0;  // horizontal edge
        const float ex = fminf(fmaxf(e.kxy * ey, dx0), dx1);
        hit |= (0.5f * (e.A * ex * ex + e.C * ey * ey) + e.B * ex * ey) <= e.tau;
    }
    return hit;
}

// ---- exact culling at cell granularity: which of the sixteen 4x4-pixel cells of a tile can the ellipse reach? ----
// Bit cy * 4 + cx of the result is set when some pixel centre of cell (cx, cy) -- centres tx0 + 4 cx + 0.5 ... + 3.5,
// rows alike -- may pass the alpha >= 1/255 test; a clear bit means all sixteen pixels fail it (same inflated tau as
// ellipse_prepare, plus 0.01 px on every extent: the closed forms below round differently from the blend loop's
// polynomial).  By row strips: for dy in [d0, d1] the ellipse A dx^2 + 2 B dx dy + C dy^2 <= 2 tau spans
//   dx in [ -(B/A) dy - sqrt(2 tau A - det dy^2) / A ,  -(B/A) dy + sqrt(2 tau A - det dy^2) / A ],   |dy| <= ey,
// whose upper end is concave in dy with its maximum at dy = -B k and whose lower end is convex with its minimum at
// dy = +B k, k = sqrt(2 tau / (det C)): the x extent over the strip is attained at those points clamped to the strip.
__device__ __forceinline__ unsigned cell_mask16(float mx, float my, float opac, float A, float B, float C, int tx0,
                                                int ty0) {
    // (a conservative test: contraction is welcome here, unlike in the bit-exact tile rectangles above)
    const float o255 = 255.0f * opac;
    if (!(o255 > 1.0f)) return 0u;
    const float det = A * C - B * B;
    const float rdet = __builtin_amdgcn_rcpf(det);
    const float tau2 = 2.0f * (__logf(o255) * (1.0002f + 4e-6f * (A * C) * rdet) + 2e-4f);
    const float rA = __builtin_amdgcn_rcpf(A);
    const float t2A = tau2 * A;
    const float ey = __builtin_amdgcn_sqrtf(t2A * rdet) * 1.0001f + 0.01f;
    const float dys = B * __builtin_amdgcn_sqrtf(tau2 * rdet * __builtin_amdgcn_rcpf(C));
    const float bA = B * rA, hA = rA * 1.0001f;
    const float rx = ((float)tx0 + 0.5f) - mx;   // first pixel-centre column / row of the tile relative to the mean
    const float ry = ((float)ty0 + 0.5f) - my;
    const float kh = -0.25f * rx, kl = -0.25f * (rx + 3.0f);
    unsigned m = 0u;
#pragma unroll
    for (int cy = 0; cy < 4; ++cy) {
        const float e0 = fmaxf(ry + (float)(4 * cy), -ey), e1 = fminf(ry + (float)(4 * cy + 3), ey);
        // (for e0 > e1 the strip misses the ellipse and the result is discarded below)
        const float yu = __builtin_amdgcn_fmed3f(-dys, e0, e1), yl = __builtin_amdgcn_fmed3f(dys, e0, e1);
        // Round 5 (issue classes, tools/probe/valu_issue.hip: min / max / floor / convert / shift are four-cycle instructions,
        // and one evaluation of this test costs the forward ~0.1 ms): |.| under the root instead of max(., 0) -- a slightly
        // negative radicand at the very tips of the ellipse becomes a slightly positive one, i.e. a superset --, floor and
        // convert in one instruction (v_cvt_flr_i32_f32; ceil(x) = -floor(-x)), the run of cell bits from v_bfm_b32.
        const float hu = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fabsf(__builtin_fmaf(-det * yu, yu, t2A))), hA, 0.01f);
        const float hl = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fabsf(__builtin_fmaf(-det * yl, yl, t2A))), hA, 0.01f);
        const float xhi = __builtin_fmaf(-bA, yu, hu), nxlo = __builtin_fmaf(bA, yl, hl);   // nxlo = -xlo
        // cells with  rx + 4 cx <= xhi  and  rx + 4 cx + 3 >= xlo:  cx in [ceil(xlo / 4 + kl), floor(xhi / 4 + kh)] within 0 .. 3
        int ih, nil;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ih) : "v"(__builtin_fmaf(xhi, 0.25f, kh)));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(nil) : "v"(__builtin_fmaf(nxlo, 0.25f, -kl)));   // -ceil(xlo / 4 + kl)
        const int il = max(-nil, 0);
        int n = min(ih, 3) - il + 1;
        n = (e0 <= e1) ? n : 0;
        unsigned bits;
        asm("v_bfm_b32 %0, %1, %2" : "=v"(bits) : "v"(max(n, 0)), "v"(il + 4 * cy));   // ((1 << n) - 1) << (il + 4 cy)
        m |= bits;
    }
    return m;
}
