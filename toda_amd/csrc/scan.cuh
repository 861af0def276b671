// Device-wide int32 exclusive scan (reduce / scan-partials / apply), templated on an accessor so
// the same three kernels serve plain int arrays and the {bits, prefix} cells of the grid index.
#pragma once
#include "common.h"

namespace toda {

struct PlainAccess {
    int32_t* data;
    __device__ __forceinline__ int load(long long i) const { return data[i]; }
    __device__ __forceinline__ void store(long long i, int v) const { data[i] = v; }
};

// cell.x = 32 occupancy bits, cell.y = number of set bits in all earlier cells
struct CellAccess {
    uint2* cells;
    __device__ __forceinline__ int load(long long i) const { return __popc(cells[i].x); }
    __device__ __forceinline__ void store(long long i, int v) const { cells[i].y = (unsigned)v; }
};

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive prefix of v within a 256-thread block; *block_total = block sum
__device__ __forceinline__ int block_exclusive_scan(int v, int* block_total) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_inclusive_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_BLOCK / 64; ++i) {
        int s = wsum[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *block_total = tot;
    return base + inc - v;
}

template <class Acc>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_kernel(Acc acc, long long n, int32_t* __restrict__ partials) {
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += acc.load(base + i);
    int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// single block: exclusive scan of partials[0..nb) in place, total to *total_dev
// (a template only so that every translation unit carries its own copy)
template <int UNUSED>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_partials_kernel(int32_t* __restrict__ partials, int nb,
                                                                   int32_t* __restrict__ total_dev) {
    int carry = 0;
    for (int start = 0; start < nb; start += SCAN_BLOCK) {
        int i = start + threadIdx.x;
        int v = i < nb ? partials[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot);
        if (i < nb) partials[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_dev) *total_dev = carry;
}

template <class Acc>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(Acc acc, long long n, const int32_t* __restrict__ partials) {
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? acc.load(base + i) : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot) + partials[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) acc.store(base + i, ex);
        ex += v[i];
    }
}

// in-place exclusive scan through `acc`; *total_dev (nullable) receives the grand total
template <class Acc>
static int exclusive_scan(Acc acc, long long n, int32_t* partials, int32_t* total_dev, hipStream_t s) {
    if (n <= 0) {
        if (total_dev) TODA_HIP(hipMemsetAsync(total_dev, 0, sizeof(int32_t), s));
        return TODA_OK;
    }
    const int nb = cdiv(n, SCAN_TILE);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_reduce_kernel<Acc>), dim3(nb), dim3(SCAN_BLOCK), 0, s, acc, n, partials);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_partials_kernel<0>), dim3(1), dim3(SCAN_BLOCK), 0, s, partials, nb, total_dev);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_apply_kernel<Acc>), dim3(nb), dim3(SCAN_BLOCK), 0, s, acc, n, partials);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

}  // namespace toda
