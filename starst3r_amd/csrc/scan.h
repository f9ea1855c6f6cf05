// Device-wide prefix sums of 32- and 64-bit unsigned integers, hand-written (three phases: per-tile reduce, one
// workgroup scanning the tile sums, per-tile downsweep with wave64 prefix scans) -- the MCMC sampler's weight prefix
// sums and dead-Gaussian ranks (gsplat relocate / sample_add reach torch.multinomial + cumsum here [U]).  Integer
// arithmetic: the result does not depend on the grouping.  In-place operation is allowed (out == in).
#pragma once
#include "common.h"

namespace st3r_scan {

constexpr int THREADS = 256;
constexpr int ITEMS = 8;
constexpr int TILE = THREADS * ITEMS;

template <typename T>
__device__ __forceinline__ T wave_incl(T v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// inclusive scan of one value per thread over the workgroup; *total = workgroup sum
template <typename T>
__device__ __forceinline__ T block_incl(T v, T* wsum, T* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T inc = wave_incl(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    T base = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) base += i < w ? wsum[i] : T(0);
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return inc + base;
}

template <typename T>
__global__ __launch_bounds__(THREADS) void k_reduce(const T* __restrict__ in, int64_t n, T* __restrict__ tile_sums) {
    __shared__ T wsum[THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
    T s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (base + i < n) s += in[base + i];
    T total;
    block_incl(s, wsum, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// exclusive scan of the tile sums in place (one workgroup)
template <typename T>
__global__ __launch_bounds__(THREADS) void k_tile_sums(T* __restrict__ tile_sums, int n_tiles) {
    __shared__ T wsum[THREADS / 64];
    T carry = 0;
    for (int b = 0; b < n_tiles; b += THREADS) {
        const int idx = b + threadIdx.x;
        const T v = idx < n_tiles ? tile_sums[idx] : T(0);
        T total;
        const T inc = block_incl(v, wsum, &total);
        if (idx < n_tiles) tile_sums[idx] = carry + inc - v;
        carry += total;
    }
}

template <typename T, bool EXCLUSIVE>
__global__ __launch_bounds__(THREADS) void k_down(const T* in, int64_t n, const T* __restrict__ tile_sums, T* out) {
    __shared__ T wsum[THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
    T v[ITEMS];
    T s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { v[i] = base + i < n ? in[base + i] : T(0); s += v[i]; }
    T total;
    const T inc = block_incl(s, wsum, &total);
    T run = tile_sums[blockIdx.x] + inc - s;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (EXCLUSIVE) { if (base + i < n) out[base + i] = run; run += v[i]; }
        else { run += v[i]; if (base + i < n) out[base + i] = run; }
    }
}

// scratch: `tmp` must hold ceil(n / TILE) values of T
template <typename T>
inline size_t scratch_bytes(int64_t n) { return sizeof(T) * (size_t)((n + TILE - 1) / TILE + 1); }

template <typename T, bool EXCLUSIVE>
inline void scan(hipStream_t s, const T* in, T* out, int64_t n, void* tmp) {
    if (n <= 0) return;
    const int n_tiles = (int)((n + TILE - 1) / TILE);
    T* sums = (T*)tmp;
    hipLaunchKernelGGL(k_reduce<T>, dim3(n_tiles), dim3(THREADS), 0, s, in, n, sums);
    hipLaunchKernelGGL(k_tile_sums<T>, dim3(1), dim3(THREADS), 0, s, sums, n_tiles);
    hipLaunchKernelGGL((k_down<T, EXCLUSIVE>), dim3(n_tiles), dim3(THREADS), 0, s, in, n, sums, out);
}

}  // namespace st3r_scan
