// How fast are 26 M device-scope atomics on 65 k counters (tile counts / tile cursors of a binning rasterizer front end)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/atomic_probe.hip -o tools/probe/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// mode 0: non-returning add; mode 1: returning add + 8-byte store at the returned position
template <int MODE>
__global__ void k_atomics(int n_pairs, int n_tiles, int rect_w, int rect_h, int tile_w, unsigned* counters, const unsigned* offsets, uint2* list) {
    int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= n_pairs) return;
    uint32_t h = hash32(pid);
    int cam = pid / (n_pairs / 8);
    int tiles_per_cam = n_tiles / 8;
    int tx0 = h % (tile_w - rect_w), ty0 = (h >> 12) % (tiles_per_cam / tile_w - rect_h);
    for (int y = 0; y < rect_h; ++y)
        for (int x = 0; x < rect_w; ++x) {
            int t = cam * tiles_per_cam + (ty0 + y) * tile_w + tx0 + x;
            if (MODE == 0) atomicAdd(&counters[t], 1u);
            else { unsigned pos = atomicAdd(&counters[t], 1u); list[offsets[t] + pos] = make_uint2(h, pid); }
        }
}
int main() {
    const int n_pairs = 8000000, n_tiles = 65280, tile_w = 120;
    unsigned *cnt, *off; uint2* list;
    hipMalloc(&cnt, n_tiles * 4); hipMalloc(&off, n_tiles * 4); hipMalloc(&list, (size_t)n_pairs * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rw = 1; rw <= 2; ++rw) for (int rh = 1; rh <= 2; ++rh) {
        // exact counts for the offsets
        hipMemset(cnt, 0, n_tiles * 4);
        k_atomics<0><<<(n_pairs + 255) / 256, 256>>>(n_pairs, n_tiles, rw, rh, tile_w, cnt, off, list);
        unsigned* h = new unsigned[n_tiles]; hipMemcpy(h, cnt, n_tiles * 4, hipMemcpyDeviceToHost);
        unsigned run = 0, mx = 0; for (int i = 0; i < n_tiles; ++i) { unsigned c = h[i]; h[i] = run; run += c; if (c > mx) mx = c; }
        hipMemcpy(off, h, n_tiles * 4, hipMemcpyHostToDevice); delete[] h;
        float ms0 = 0, ms1 = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(cnt, 0, n_tiles * 4);
            hipEventRecord(e0); k_atomics<0><<<(n_pairs + 255) / 256, 256>>>(n_pairs, n_tiles, rw, rh, tile_w, cnt, off, list);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms0, e0, e1);
            hipMemset(cnt, 0, n_tiles * 4);
            hipEventRecord(e0); k_atomics<1><<<(n_pairs + 255) / 256, 256>>>(n_pairs, n_tiles, rw, rh, tile_w, cnt, off, list);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        }
        printf("rect %dx%d: %.1f M atomics (max %u per tile): count %.3f ms, returning + 8-byte scatter %.3f ms\n", rw, rh, n_pairs * rw * rh / 1e6, mx, ms0, ms1);
    }
    return 0;
}
