// Halo plans for the submanifold gather-GEMM (spconv.hip: gather_gemm_halo_kernel).
//
// A SubM layer's output rows are visited in blocks of R rows that are COMPACT IN SPACE (Morton order over (y, x), z innermost):
// the K = 27 neighbour sets of such a block overlap almost completely, so the block's unique input rows are few
// (C3 stride-4 level, R = 128: 223 on average, 333 at most, against 750 for 128 consecutive canonical rows and 2048 pair
// references) and fit in LDS.  The plan names, per block,
//   order[R]      the output rows of the block (Morton order; -1 pads the last block),
//   urows[UMAX]   the block's unique neighbour rows in ascending order (-1 pads), at most UMAX of them,
//   lids[K][R]    per (offset, row) the 16-bit position of the neighbour in urows: HALO_NONE = no neighbour, HALO_SPILL = the
//                 neighbour exists but did not fit (the kernel then takes that (tile, offset) from the global table).
// One plan serves the forward AND the data gradient of every SubM layer on the table (the dgrad reads the same table with the
// offsets reversed in its packed weights).
//
// Order: no sort.  The rows are marked in a bitmap laid out in Morton order and ranked by the library's popcount scan - the
// same mark -> scan -> rank machinery that builds every generated index set (rulebook.hip).
// Dedupe: one workgroup per block, bitonic sort of the K * R neighbour ids in LDS, adjacent-difference + scan, binary search
// for the local ids.  Deterministic (no hash, no atomics on the result).
#include <stdlib.h>

#include "scan.cuh"

namespace toda {

constexpr int HP_BLOCK = 256;
constexpr unsigned short HALO_NONE = 0xFFFFu, HALO_SPILL = 0xFFFEu;

// geometry shared with spconv.hip (keep in sync with halo_geom there)
struct HaloGeom {
    int R, UMAX;
};
static inline bool halo_geom(int c_gather, HaloGeom* g) {
    if (c_gather == 64 || c_gather == 32) {
        *g = HaloGeom{128, 320};
        return true;
    }
    return false;
}

__device__ __forceinline__ unsigned part1by1(unsigned v) {      // 16 bits -> even bit positions
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

struct MortonDims {
    int B, D, H, W;
    long long s2;     // S * S, S = next power of two >= max(H, W)
};

__device__ __forceinline__ long long morton_lin(const int4 c, const MortonDims& g) {
    const unsigned m = part1by1((unsigned)c.w) | (part1by1((unsigned)c.z) << 1);       // x in the even bits, y in the odd ones
    return ((long long)c.x * g.s2 + m) * g.D + c.y;
}

__device__ __forceinline__ bool morton_inside(const int4 c, const MortonDims& g) {
    return (unsigned)c.x < (unsigned)g.B && (unsigned)c.y < (unsigned)g.D && (unsigned)c.z < (unsigned)g.H && (unsigned)c.w < (unsigned)g.W;
}

__global__ void __launch_bounds__(HP_BLOCK)
halo_mark_kernel(const int4* __restrict__ idx, int n, MortonDims g, uint2* __restrict__ cells) {
    const int i = blockIdx.x * HP_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = idx[i];
    if (!morton_inside(c, g)) return;
    const long long lin = morton_lin(c, g);
    atomicOr(&cells[lin >> 5].x, 1u << (lin & 31));
}

// order[rank] = row, ranks from the Morton bitmap.  Duplicate coordinate rows (hand-built tensors; the library's own sets have none) share a
// bit: the lowest row index claims the rank (atomicMin), the other duplicates and the rows outside the lattice (no neighbours,
// nobody's neighbour) are appended behind the ranked rows - every row is produced by exactly one block.
__global__ void __launch_bounds__(HP_BLOCK)
halo_order_claim_kernel(const int4* __restrict__ idx, int n, MortonDims g, const uint2* __restrict__ cells, int32_t* __restrict__ claim) {
    const int i = blockIdx.x * HP_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = idx[i];
    if (!morton_inside(c, g)) return;
    const long long lin = morton_lin(c, g);
    const uint2 cell = cells[lin >> 5];
    const unsigned bit = 1u << (lin & 31);
    atomicMin(&claim[(int)cell.y + __popc(cell.x & (bit - 1))], i);      // lowest row index owns the rank
}

__global__ void __launch_bounds__(HP_BLOCK)
halo_order_place_kernel(const int4* __restrict__ idx, int n, MortonDims g, const uint2* __restrict__ cells, const int32_t* __restrict__ total,
                        const int32_t* __restrict__ claim, int32_t* __restrict__ stray, int32_t* __restrict__ order) {
    const int i = blockIdx.x * HP_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = idx[i];
    bool ranked = false;
    if (morton_inside(c, g)) {
        const long long lin = morton_lin(c, g);
        const uint2 cell = cells[lin >> 5];
        const unsigned bit = 1u << (lin & 31);
        const int r = (int)cell.y + __popc(cell.x & (bit - 1));
        if (claim[r] == i) {
            order[r] = i;
            ranked = true;
        }
    }
    if (!ranked) {      // outside the lattice, or a duplicate of a lower row: appended behind the ranked rows (any order: disjoint slots)
        const int pos = *total + atomicAdd(stray, 1);
        if (pos < n) order[pos] = i;
    }
}

// One workgroup per block of R rows: in-block row order, unique neighbour rows, local ids.
//   1. the K * R neighbour ids of the block's rows (registers, <= ceil(K R / 256) per thread); every row's K-bit neighbour mask;
//      the ids go into an LDS hash set (linear probing; only membership is used, so the racy slot layout does not reach the result)
//   2. rows are re-ordered INSIDE the block by their neighbour mask (rank by counting, R^2 comparisons): the 16-row tiles of the
//      kernel then share offsets and skip the others wave-uniformly - in plain Morton order EVERY (tile, offset) has a neighbour
//      (C3 stride-4 level: executed tile fraction 1.00 against 0.72 mask-sorted and 0.705 for the canonical x-runs; rows of a block
//      sit in LDS, so their order costs nothing in locality)
//   3. occupied hash slots are compacted and sorted (bitonic over the next power of two >= the unique count: 256 or 512 keys, not
//      the 4096 of a sort-based dedupe), the ranks go back into the table, every (offset, row) looks its local id up
template <int R, int HS>      // HS = hash slots, power of two > 27 * R
__global__ void __launch_bounds__(HP_BLOCK)
halo_plan_kernel(const int32_t* __restrict__ nbr, int n, int K, int umax, int32_t* __restrict__ order_all, int32_t* __restrict__ urows_all,
                 unsigned short* __restrict__ lids_all) {
    constexpr int PER = (27 * R + HP_BLOCK - 1) / HP_BLOCK;
    __shared__ int hkey[HS];
    __shared__ int hval[HS];
    __shared__ int uq[HS];
    __shared__ int rows[R];
    __shared__ unsigned mask[R];
    __shared__ int pos[R];
    __shared__ int s_nu;
    const int t = threadIdx.x, blk = blockIdx.x;
    int32_t* order = order_all + (size_t)blk * R;
    for (int i = t; i < R; i += HP_BLOCK) {
        const long long p = (long long)blk * R + i;
        rows[i] = p < n ? order[i] : -1;
        mask[i] = 0u;
    }
    for (int e = t; e < HS; e += HP_BLOCK) hkey[e] = -1;
    __syncthreads();
    const int total = K * R;
    int ids[PER];
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int e = t + m * HP_BLOCK;
        int id = -1;
        if (e < total) {
            const int k = e / R, i = e - k * R;
            const int row = rows[i];
            if (row >= 0) id = nbr[(size_t)k * n + row];
        }
        ids[m] = id;
    }
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int e = t + m * HP_BLOCK, id = ids[m];
        if (id < 0) continue;
        const int k = e / R, i = e - k * R;
        atomicOr(&mask[i], 1u << k);
        unsigned h = ((unsigned)id * 2654435761u) >> 7 & (HS - 1);
        while (true) {
            const int prev = atomicCAS(&hkey[h], -1, id);
            if (prev == -1 || prev == id) break;
            h = (h + 1) & (HS - 1);
        }
    }
    __syncthreads();
    // in-block order: position = number of rows with a smaller (mask, index); padding rows (mask 0, row -1) are given the largest key
    for (int i = t; i < R; i += HP_BLOCK) {
        const unsigned long long mine = rows[i] >= 0 ? (((unsigned long long)mask[i] << 16) | (unsigned)i) : (0xFFFFFFFFFFFF0000ull | (unsigned)i);
        int c = 0;
        for (int j = 0; j < R; ++j) {
            const unsigned long long other = rows[j] >= 0 ? (((unsigned long long)mask[j] << 16) | (unsigned)j) : (0xFFFFFFFFFFFF0000ull | (unsigned)j);
            c += other < mine;
        }
        pos[i] = c;
    }
    // compaction of the occupied slots (slot order; sorted below)
    constexpr int SL = HS / HP_BLOCK;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < SL; ++q) cnt += hkey[t * SL + q] >= 0;
    int tot;
    int base = block_exclusive_scan(cnt, &tot);
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int v = hkey[t * SL + q];
        if (v >= 0) uq[base++] = v;
    }
    if (t == 0) s_nu = tot;
    __syncthreads();
    const int nu = s_nu;
    int P = 1;
    while (P < nu) P <<= 1;
    for (int e = nu + t; e < P; e += HP_BLOCK) uq[e] = 0x7FFFFFFF;
    __syncthreads();
    for (int span = 2; span <= P; span <<= 1) {
        for (int j = span >> 1; j > 0; j >>= 1) {
            for (int p2 = t; p2 < P / 2; p2 += HP_BLOCK) {
                const int i = 2 * p2 - (p2 & (j - 1));
                const int l = i + j;
                const bool up = (i & span) == 0;
                const int a = uq[i], b = uq[l];
                if ((a > b) == up) {
                    uq[i] = b;
                    uq[l] = a;
                }
            }
            __syncthreads();
        }
    }
    // ranks back into the table
    for (int j = t; j < nu; j += HP_BLOCK) {
        const int id = uq[j];
        unsigned h = ((unsigned)id * 2654435761u) >> 7 & (HS - 1);
        while (hkey[h] != id) h = (h + 1) & (HS - 1);
        hval[h] = j;
    }
    __syncthreads();
    int32_t* urows = urows_all + (size_t)blk * umax;
    for (int j = t; j < umax; j += HP_BLOCK) urows[j] = j < nu ? uq[j] : -1;
    for (int i = t; i < R; i += HP_BLOCK) order[pos[i]] = rows[i];
    unsigned short* lids = lids_all + (size_t)blk * K * R;
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int e = t + m * HP_BLOCK;
        if (e >= total) continue;
        const int k = e / R, i = e - k * R, id = ids[m];
        unsigned short lid = HALO_NONE;
        if (id >= 0) {
            unsigned h = ((unsigned)id * 2654435761u) >> 7 & (HS - 1);
            while (hkey[h] != id) h = (h + 1) & (HS - 1);
            const int rk = hval[h];
            lid = rk < umax ? (unsigned short)rk : HALO_SPILL;
        }
        lids[(size_t)k * R + pos[i]] = lid;
    }
}

struct HaloWs {
    long long cells;
    size_t o_cells, o_part, o_total, o_stray, o_claim, bytes;
};
static HaloWs halo_ws(int n, int batch, const int32_t* shape) {
    HaloWs w;
    int s = 1;
    while (s < shape[1] || s < shape[2]) s <<= 1;
    const long long bits = (long long)batch * s * s * shape[0];
    w.cells = (bits + 31) / 32;
    size_t o = 0;
    w.o_cells = o;
    o += align_up((size_t)w.cells * sizeof(uint2), 256);
    w.o_part = o;
    o += scan_partials_bytes(w.cells);
    w.o_total = o;
    o += 256;
    w.o_stray = o;
    o += 256;
    w.o_claim = o;
    o += align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256);
    w.bytes = o;
    return w;
}

}  // namespace toda

using namespace toda;

extern "C" int toda_halo_supported(int c_gather, int c_produce, int k_vol) {
    HaloGeom g;
    return (c_gather == c_produce && k_vol >= 2 && k_vol <= 27 && halo_geom(c_gather, &g)) ? 1 : 0;
}

extern "C" size_t toda_halo_plan_bytes(int n, int k_vol, int c_gather) {
    HaloGeom g;
    if (!halo_geom(c_gather, &g) || n <= 0) return 0;
    const size_t nb = (size_t)cdiv(n, g.R);
    // order | urows | lids (each part 256-byte aligned)
    return align_up(nb * g.R * 4, 256) + align_up(nb * g.UMAX * 4, 256) + align_up(nb * k_vol * g.R * 2, 256);
}

extern "C" size_t toda_halo_plan_workspace_bytes(int n, int batch, const int32_t* shape_host) {
    return halo_ws(n, batch, shape_host).bytes;
}

extern "C" int toda_halo_plan_build(const int32_t* indices, int n, int batch, const int32_t* shape_host, const int32_t* nbr, int k_vol,
                                    int c_gather, void* plan, size_t plan_bytes, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HaloGeom hg;
    TODA_CHECK_ARG(halo_geom(c_gather, &hg), "halo_plan_build: no halo geometry for %d channels", c_gather);
    TODA_CHECK_ARG(k_vol >= 2 && k_vol <= 27, "halo_plan_build: kernel volume %d outside [2, 27]", k_vol);
    TODA_CHECK_ARG(batch >= 1 && shape_host && shape_host[0] >= 1 && shape_host[1] >= 1 && shape_host[2] >= 1 && shape_host[1] <= 65536 &&
                       shape_host[2] <= 65536, "halo_plan_build: bad batch / shape");
    TODA_CHECK_ARG(n >= 0, "halo_plan_build: n < 0");
    if (n == 0) return TODA_OK;
    TODA_CHECK_ARG(indices && nbr && plan && ws, "halo_plan_build: null pointer");
    const size_t need = toda_halo_plan_bytes(n, k_vol, c_gather);
    const HaloWs w = halo_ws(n, batch, shape_host);
    TODA_CHECK_ARG((long long)w.cells < (1LL << 31), "halo_plan_build: lattice too large for the Morton bitmap");
    if (plan_bytes < need || ws_bytes < w.bytes) {
        set_error("halo_plan_build: plan %zu < %zu or workspace %zu < %zu bytes", plan_bytes, need, ws_bytes, w.bytes);
        return TODA_EWORKSPACE;
    }
    char* b = (char*)ws;
    uint2* cells = (uint2*)(b + w.o_cells);
    int32_t* total = (int32_t*)(b + w.o_total);
    int32_t* stray = (int32_t*)(b + w.o_stray);
    int32_t* claim = (int32_t*)(b + w.o_claim);
    MortonDims g;
    g.B = batch, g.D = shape_host[0], g.H = shape_host[1], g.W = shape_host[2];
    int sdim = 1;
    while (sdim < g.H || sdim < g.W) sdim <<= 1;
    g.s2 = (long long)sdim * sdim;
    const size_t nb = (size_t)cdiv(n, hg.R);
    int32_t* order = (int32_t*)plan;
    int32_t* urows = (int32_t*)((char*)plan + align_up(nb * hg.R * 4, 256));
    unsigned short* lids = (unsigned short*)((char*)urows + align_up(nb * hg.UMAX * 4, 256));
    TODA_HIP(hipMemsetAsync(cells, 0, (size_t)w.cells * sizeof(uint2), s));
    TODA_HIP(hipMemsetAsync(stray, 0, sizeof(int32_t), s));
    TODA_HIP(hipMemsetAsync(claim, 0x7F, (size_t)n * sizeof(int32_t), s));
    TODA_HIP(hipMemsetAsync(order, 0xFF, nb * hg.R * sizeof(int32_t), s));
    const dim3 grid(cdiv(n, HP_BLOCK)), block(HP_BLOCK);
    hipLaunchKernelGGL(halo_mark_kernel, grid, block, 0, s, (const int4*)indices, n, g, cells);
    int rc = exclusive_scan(CellAccess{cells}, w.cells, (int32_t*)(b + w.o_part), total, s);
    if (rc) return rc;
    hipLaunchKernelGGL(halo_order_claim_kernel, grid, block, 0, s, (const int4*)indices, n, g, cells, claim);
    hipLaunchKernelGGL(halo_order_place_kernel, grid, block, 0, s, (const int4*)indices, n, g, cells, total, claim, stray, order);
    if (hg.R == 128)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_plan_kernel<128, 4096>), dim3((unsigned)nb), block, 0, s, nbr, n, k_vol, hg.UMAX, order, urows, lids);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_plan_kernel<256, 8192>), dim3((unsigned)nb), block, 0, s, nbr, n, k_vol, hg.UMAX, order, urows, lids);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
