// Hard voxelisation + MeanVFE for gfx950.
//
// The reference voxelises sequentially on the CPU (spconv Point2VoxelCPU3d, called from
// pcdet/datasets/processor/data_processor.py:44-60), one sample at a time in DataLoader workers, and collate_batch
// (pcdet/datasets/dataset.py:161-178) concatenates the samples and prepends the batch column.  The sequential result
// (voxel ids in first-appearance order, first P points kept, voxels past the cap dropped) is reproduced bit-exactly for a
// WHOLE BATCH by three launches of an order-free formulation (round 4; rounds 1-3: sixteen launches per sample):
//   1. vox_insert: every point pushes itself on the list of its cell.  One 64-bit word per hash slot = {cell key, list
//      head}; claiming an empty slot and pushing on an occupied one are the same compare-and-swap, so the common case (one
//      point per voxel) is ONE device atomic per point (before: CAS + atomicMin + atomicAdd; scattered device atomics run at
//      ~20 G/s chip-wide on MI355X - they execute at the memory side - and were the kernel's whole time).
//   2. vox_walk: every point walks its cell's list: r = #points of the cell with a smaller index, f = the smallest index
//      (the FOUNDER), len.  A point is kept iff r < P; a founder (r == 0) takes an in-block exclusive rank `loc`; the
//      founder count of each 1024-point block goes to `partial`.
//   3. vox_emit: every block re-derives from `partial` (a few hundred ints) the founders before it = the first-appearance
//      voxel id of its first founder, the per-sample totals and the row offset of its sample in the batch output (samples
//      are concatenated, each clipped at the cap: spconv's `continue` at the cap).  The founders of a block own CONSECUTIVE
//      voxel ids, so their rows (point 0 + zero padding), coordinates (b, z, y, x) and counts are one contiguous,
//      coalesced region written cooperatively; the few non-founders (r = 1 .. P-1) scatter their 20-byte row.  Every point
//      finally puts its slot back to EMPTY: the table is clean again when the call returns (no memset per call).
// Launch-bound index work: no FLOPs, ~40 B per point.
#include "scan.cuh"

namespace toda {

constexpr int VOX_BLOCK = 256;
constexpr int VOX_WALK = 1024;          // points per block of vox_walk / vox_emit
constexpr int VOX_MAXB = 32;            // samples per launch
constexpr int VOX_MAX_NBLK = 4096;      // 1024-point blocks per sample (4 M points)
constexpr unsigned long long VOX_EMPTY = ~0ull;

struct VoxGeom {
    float r0[3];   // range min xyz
    float vs[3];   // voxel size xyz
    int grid[3];   // cells xyz
};

struct VoxBatch {
    const float* pts[VOX_MAXB];   // first feature column of each sample's first point
    int n[VOX_MAXB];
};

__device__ __forceinline__ unsigned hash_u32(unsigned k) {
    k ^= k >> 16;
    k *= 0x7feb352dU;
    k ^= k >> 15;
    k *= 0x846ca68bU;
    k ^= k >> 16;
    return k;
}

// the sequential voxeliser's fp32 expression floor((p - lo) / size) per axis; false = point outside the range (or NaN)
__device__ __forceinline__ bool vox_cell(const float* __restrict__ p, const VoxGeom& g, int* cc) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float f = floorf((p[j] - g.r0[j]) / g.vs[j]);
        ok = ok && (f >= 0.0f) && (f < (float)g.grid[j]);  // NaN fails both
        cc[j] = (int)f;
    }
    return ok;
}

// 1. push every point on its cell's list.  slot word = {key (high), head point index (low)}; nxt[i] = previous head or -1.
__global__ void __launch_bounds__(VOX_BLOCK)
vox_insert_kernel(VoxBatch vb, int row_stride, VoxGeom g, unsigned long long* __restrict__ table, unsigned mask, int n_pad,
                  int* __restrict__ slot_of, int* __restrict__ nxt, int max_pts) {
    __shared__ unsigned s_key[VOX_BLOCK / 64][64];
    __shared__ int s_cnt[VOX_BLOCK / 64][64], s_n[VOX_BLOCK / 64];
    const int b = blockIdx.y;
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    const bool live = i < vb.n[b];
    const size_t gi = (size_t)b * n_pad + i;
    int cc[3] = {0, 0, 0};
    const bool inside = live && vox_cell(vb.pts[b] + (size_t)i * row_stride, g, cc);
    const unsigned key = (unsigned)((cc[2] * g.grid[1] + cc[1]) * g.grid[0] + cc[0]);
    // Hot cells (a zero-padded cloud, tens of thousands of returns at the origin): k pushes on ONE word cost ~k^2 / 2 serialised
    // compare-and-swaps and the walk k^2 dependent loads - seconds instead of microseconds.  A point with max_pts or more same-cell points
    // of smaller index INSIDE ITS OWN WORKGROUP (= 256 consecutive indices) can be neither kept nor a founder, and every later point of the
    // cell has rank >= max_pts with or without it (those max_pts block-mates are smaller than both), so it is not pushed at all: at most
    // max_pts pushes per workgroup and cell; same voxels, same kept points, same counts (min(len, max_pts)).
    //   wave level: one pass per distinct cell of the wave (~64 on a shuffled cloud: ~1.5 k cycles per wave, 3-4 us per 360 k points);
    //   workgroup level: only when some wave of the workgroup found two points in one cell (rare on real clouds).
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int rank_w = 0, dup = 0;
    {
        unsigned long long todo = __ballot(inside);
        int t = 0;
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const unsigned lk = (unsigned)__shfl((int)key, lead, 64);
            const bool same_cell = inside && key == lk;
            const unsigned long long same = __ballot(same_cell);
            const int cnt = __popcll(same);
            if (same_cell) rank_w = __popcll(same & ((1ull << lane) - 1ull));
            if (lane == lead) {
                s_key[wv][t] = lk;
                s_cnt[wv][t] = cnt < max_pts ? cnt : max_pts;
            }
            dup |= cnt > 1;
            ++t;
            todo &= ~same;
        }
        if (lane == 0) s_n[wv] = t;
    }
    bool push = inside && rank_w < max_pts;
    if (__syncthreads_or(dup)) {
        if (push) {
            int before = rank_w;
            for (int w = 0; w < wv && before < max_pts; ++w)
                for (int t = 0; t < s_n[w]; ++t)
                    if (s_key[w][t] == key) before += s_cnt[w][t];
            push = before < max_pts;
        }
    }
    if (!live) return;
    if (!push) {
        slot_of[gi] = -1;      // outside the range, or certainly not among its cell's first max_pts points: no part in the lists
        return;
    }
    unsigned long long* tab = table + (size_t)b * ((size_t)mask + 1);
    unsigned s = hash_u32(key) & mask;
    const unsigned long long mine = ((unsigned long long)key << 32) | (unsigned)i;
    unsigned long long seen = tab[s];          // may be stale: only ever "EMPTY although taken", which the CAS below corrects
    int prev_head = -1;
    while (true) {
        if (seen == VOX_EMPTY || (unsigned)(seen >> 32) == key) {
            const unsigned long long old = atomicCAS(&tab[s], seen, mine);
            if (old == seen) {
                prev_head = seen == VOX_EMPTY ? -1 : (int)(unsigned)seen;
                break;
            }
            seen = old;                         // somebody else got there first: look at what is there now
            continue;
        }
        s = (s + 1) & mask;
        seen = tab[s];
    }
    slot_of[gi] = (int)s;
    nxt[gi] = prev_head;
}

__device__ __forceinline__ int block_sum_1024(int v, int* s_w) {     // all 1024 threads; s_w: 16 ints of LDS
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < VOX_WALK / 64; ++w) t += s_w[w];
    return t;
}

// 2. rank inside the cell, founder, founder's in-block rank, founders per block
__global__ void __launch_bounds__(VOX_WALK)
vox_walk_kernel(VoxBatch vb, const unsigned long long* __restrict__ table, unsigned mask, int n_pad, int nblk,
                const int* __restrict__ slot_of, const int* __restrict__ nxt, int* __restrict__ rnk, int* __restrict__ aux,
                int* __restrict__ len_of, int* __restrict__ partial) {
    __shared__ int s_w[VOX_WALK / 64];
    const int b = blockIdx.y;
    const int i = blockIdx.x * VOX_WALK + threadIdx.x;
    const size_t base = (size_t)b * n_pad;
    int r = -1, f = -1, len = 0;
    if (i < vb.n[b]) {
        const int s = slot_of[base + i];
        if (s >= 0) {
            int j = (int)(unsigned)table[(size_t)b * ((size_t)mask + 1) + s];
            r = 0;
            f = i;
            const int nb = vb.n[b];
            while ((unsigned)j < (unsigned)nb && len < nb) {      // (bounded: a table that was not clean must not hang the GPU)
                ++len;
                r += j < i;
                f = j < f ? j : f;
                j = nxt[base + j];
            }
        }
    }
    const bool founder = r == 0;
    const unsigned long long vote = __ballot(founder);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = __popcll(vote);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < VOX_WALK / 64; ++k) {
        const int c = s_w[k];
        before += k < w ? c : 0;
        total += c;
    }
    const int loc = before + __popcll(vote & ((1ull << lane) - 1ull));
    if (i < vb.n[b]) {
        rnk[base + i] = r;
        aux[base + i] = founder ? loc : f;
        len_of[base + i] = len;
    }
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = total;
}

// 3. emit.  coord_cols = 4: rows (b, z, y, x) (the collated batch); 3: rows (z, y, x) (one sample, spconv's own layout)
__global__ void __launch_bounds__(VOX_WALK)
vox_emit_kernel(VoxBatch vb, int batch, int c, int row_stride, VoxGeom g, int max_pts, int max_voxels,
                unsigned long long* __restrict__ table, unsigned mask, int n_pad, int nblk, const int* __restrict__ slot_of,
                const int* __restrict__ rnk, const int* __restrict__ aux, const int* __restrict__ len_of,
                const int* __restrict__ partial, float* __restrict__ voxels, int* __restrict__ coords, int coord_cols,
                int* __restrict__ num_pts, int* __restrict__ counts_dev) {
    __shared__ int s_pref[VOX_MAX_NBLK];      // founders of this sample before each of its blocks
    __shared__ int s_tot[VOX_MAXB];
    __shared__ int s_w[VOX_WALK / 64];
    __shared__ int s_fi[VOX_WALK], s_flen[VOX_WALK];
    __shared__ int s_base;
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    // founders per sample: 32 lanes per sample
    {
        const int j = tid >> 5, l = tid & 31;
        int t = 0;
        if (j < batch)
            for (int k = l; k < nblk; k += 32) t += partial[(size_t)j * nblk + k];
        for (int d = 16; d > 0; d >>= 1) t += __shfl_down(t, d, 32);
        if (l == 0 && j < VOX_MAXB) s_tot[j] = j < batch ? t : 0;
    }
    // exclusive scan of this sample's block counts
    int carry = 0;
    for (int k0 = 0; k0 < nblk; k0 += VOX_WALK) {
        const int k = k0 + tid;
        const int v = k < nblk ? partial[(size_t)b * nblk + k] : 0;
        int inc = v;
        const int lane = tid & 63, w = tid >> 6;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < VOX_WALK / 64; ++q) {
            const int cq = s_w[q];
            before += q < w ? cq : 0;
            total += cq;
        }
        if (k < nblk) s_pref[k] = carry + before + inc - v;
        carry += total;
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0, all = 0;
        for (int j = 0; j < batch; ++j) {
            const int m = s_tot[j] < max_voxels ? s_tot[j] : max_voxels;
            if (j < b) o += m;
            all += m;
            if (b == 0 && blk == 0) counts_dev[j] = m;
        }
        if (b == 0 && blk == 0) counts_dev[batch] = all;
        s_base = o;
    }
    __syncthreads();
    const int pre = s_pref[blk];                      // first-appearance id of this block's first founder
    const int nf = partial[(size_t)b * nblk + blk];
    const int room = max_voxels - pre;
    const int nf_keep = room <= 0 ? 0 : (nf < room ? nf : room);
    const size_t row0 = (size_t)s_base + pre;         // output row of that founder
    const size_t base = (size_t)b * n_pad;
    const int i = blk * VOX_WALK + tid;
    const bool live = i < vb.n[b];
    const float* pts = vb.pts[b];
    int r = -1, a = 0, s = -1;
    if (live) {
        r = rnk[base + i];
        a = aux[base + i];
        s = slot_of[base + i];
    }
    if (r == 0) {
        s_fi[a] = i;
        s_flen[a] = len_of[base + i];
    } else if (r > 0 && r < max_pts) {
        // a later point of a cell founded by point a (possibly in another block)
        const int vid = s_pref[a / VOX_WALK] + aux[base + a];
        if (vid < max_voxels) {
            float* dst = voxels + (((size_t)s_base + vid) * max_pts + r) * c;
            const float* src = pts + (size_t)i * row_stride;
            for (int j = 0; j < c; ++j) dst[j] = src[j];
        }
    }
    if (s >= 0) table[(size_t)b * ((size_t)mask + 1) + s] = VOX_EMPTY;      // leave the table clean
    __syncthreads();
    if (tid < nf_keep) {
        const int fi = s_fi[tid];
        int cc[3];
        vox_cell(pts + (size_t)fi * row_stride, g, cc);
        if (coord_cols == 4) {
            reinterpret_cast<int4*>(coords)[row0 + tid] = make_int4(b, cc[2], cc[1], cc[0]);
        } else {
            int* d = coords + 3 * (row0 + tid);
            d[0] = cc[2];
            d[1] = cc[1];
            d[2] = cc[0];
        }
        const int len = s_flen[tid];
        num_pts[row0 + tid] = len < max_pts ? len : max_pts;
    }
    // rows of the kept founders' voxels: point 0 and the zero padding behind the voxel's last point, contiguous in memory
    const int rows = nf_keep * max_pts;
    float* out = voxels + row0 * max_pts * c;
    for (int e = tid; e < rows; e += VOX_WALK) {
        const int q = e / max_pts, rr = e - q * max_pts;
        float* dst = out + (size_t)e * c;
        if (rr == 0) {
            const float* src = pts + (size_t)s_fi[q] * row_stride;
            for (int j = 0; j < c; ++j) dst[j] = src[j];
        } else if (rr >= s_flen[q]) {
            for (int j = 0; j < c; ++j) dst[j] = 0.0f;
        }
    }
}

// MeanVFE forward: one thread per (voxel, channel)
__global__ void __launch_bounds__(VOX_BLOCK)
mean_vfe_fwd_kernel(const float* __restrict__ voxels, const float* __restrict__ num_pts, int m, int p, int c,
                    float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (t >= (long long)m * c) return;
    const int v = (int)(t / c), j = (int)(t % c);
    float s = 0.0f;
    for (int q = 0; q < p; ++q) s += voxels[((size_t)v * p + q) * c + j];
    const float nrm = fmaxf(num_pts[v], 1.0f);
    out[t] = s / nrm;
}

__global__ void __launch_bounds__(VOX_BLOCK)
mean_vfe_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ num_pts, int m, int p, int c,
                    float* __restrict__ gvox) {
    const long long t = (long long)blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (t >= (long long)m * p * c) return;
    const int j = (int)(t % c);
    const int v = (int)(t / ((long long)p * c));
    gvox[t] = gout[(size_t)v * c + j] / fmaxf(num_pts[v], 1.0f);
}

struct VoxWs {
    size_t slots;  // hash slots per sample (power of two)
    int n_pad, nblk;
    size_t o_table, o_slot, o_nxt, o_rnk, o_aux, o_len, o_part, bytes;
};

static VoxWs vox_layout(int batch, int n_cap) {
    VoxWs w;
    size_t slots = 1024;
    while (slots < (size_t)2 * (size_t)(n_cap > 0 ? n_cap : 1)) slots <<= 1;
    w.slots = slots;
    w.nblk = cdiv(n_cap > 0 ? n_cap : 1, VOX_WALK);
    w.n_pad = w.nblk * VOX_WALK;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    const size_t np = (size_t)batch * w.n_pad;
    w.o_table = take((size_t)batch * slots * 8);
    w.o_slot = take(np * 4);
    w.o_nxt = take(np * 4);
    w.o_rnk = take(np * 4);
    w.o_aux = take(np * 4);
    w.o_len = take(np * 4);
    w.o_part = take((size_t)batch * w.nblk * 4);
    w.bytes = o;
    return w;
}

static int vox_run(const float* const* pts_host, const int32_t* n_host, int batch, int n_cap, int c, int row_stride,
                   const float* range_host, const float* vsize_host, const int32_t* grid_host, int max_pts, int max_voxels,
                   float* voxels, int32_t* coords, int coord_cols, int32_t* num_pts, int32_t* counts_dev, void* ws,
                   size_t ws_bytes, int ws_clean, hipStream_t s) {
    TODA_CHECK_ARG(batch >= 1 && batch <= VOX_MAXB, "voxelize: batch must be in [1,%d] (got %d)", VOX_MAXB, batch);
    TODA_CHECK_ARG(c >= 3 && row_stride >= c, "voxelize: need c >= 3 and row_stride >= c (got c=%d stride=%d)", c, row_stride);
    TODA_CHECK_ARG(max_pts >= 1 && max_pts <= 64, "voxelize: max_pts must be in [1,64] (got %d)", max_pts);
    TODA_CHECK_ARG(max_voxels >= 1, "voxelize: max_voxels must be >= 1");
    TODA_CHECK_ARG((long long)grid_host[0] * grid_host[1] * grid_host[2] < (1LL << 31),
                   "voxelize: grid of one sample must have < 2^31 cells");
    TODA_CHECK_ARG(n_cap >= 0 && (long long)n_cap <= (long long)VOX_MAX_NBLK * VOX_WALK, "voxelize: at most %d points per sample",
                   VOX_MAX_NBLK * VOX_WALK);
    VoxBatch vb;
    int n_max = 0;
    for (int b = 0; b < VOX_MAXB; ++b) {
        vb.pts[b] = b < batch ? pts_host[b] : nullptr;
        vb.n[b] = b < batch ? n_host[b] : 0;
        if (b < batch) {
            TODA_CHECK_ARG(n_host[b] >= 0 && n_host[b] <= n_cap, "voxelize: sample %d has %d points, workspace laid out for %d", b,
                           n_host[b], n_cap);
            n_max = n_host[b] > n_max ? n_host[b] : n_max;
        }
    }
    const VoxWs w = vox_layout(batch, n_cap);
    if (ws_bytes < w.bytes) {
        set_error("voxelize: workspace %zu < required %zu", ws_bytes, w.bytes);
        return TODA_EWORKSPACE;
    }
    char* p = (char*)ws;
    unsigned long long* table = (unsigned long long*)(p + w.o_table);
    if (!ws_clean) TODA_HIP(hipMemsetAsync(table, 0xFF, (size_t)batch * w.slots * 8, s));
    VoxGeom g;
    for (int j = 0; j < 3; ++j) {
        g.r0[j] = range_host[j];
        g.vs[j] = vsize_host[j];
        g.grid[j] = grid_host[j];
    }
    int* slot_of = (int*)(p + w.o_slot);
    int* nxt = (int*)(p + w.o_nxt);
    int* rnk = (int*)(p + w.o_rnk);
    int* aux = (int*)(p + w.o_aux);
    int* len_of = (int*)(p + w.o_len);
    int* partial = (int*)(p + w.o_part);
    const unsigned mask = (unsigned)(w.slots - 1);
    if (n_max > 0)
        hipLaunchKernelGGL(vox_insert_kernel, dim3(cdiv(n_max, VOX_BLOCK), batch), dim3(VOX_BLOCK), 0, s, vb, row_stride, g, table,
                           mask, w.n_pad, slot_of, nxt, max_pts);
    // the walk and emit grids cover every block of the layout: blocks past a sample's points publish / consume zero founders
    hipLaunchKernelGGL(vox_walk_kernel, dim3(w.nblk, batch), dim3(VOX_WALK), 0, s, vb, table, mask, w.n_pad, w.nblk, slot_of, nxt,
                       rnk, aux, len_of, partial);
    hipLaunchKernelGGL(vox_emit_kernel, dim3(w.nblk, batch), dim3(VOX_WALK), 0, s, vb, batch, c, row_stride, g, max_pts,
                       max_voxels, table, mask, w.n_pad, w.nblk, slot_of, rnk, aux, len_of, partial, voxels, coords, coord_cols,
                       num_pts, counts_dev);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

extern "C" size_t toda_voxelize_workspace_bytes(int n_points, int /*max_voxels*/) { return vox_layout(1, n_points).bytes + 256; }

extern "C" int toda_voxelize_hard(const float* points, int n, int c, const float* range_host, const float* vsize_host,
                                  const int32_t* grid_host, int max_pts, int max_voxels, float* voxels,
                                  int32_t* coords_zyx, int32_t* num_pts, int32_t* m_dev, void* ws, size_t ws_bytes,
                                  void* stream) {
    TODA_CHECK_ARG(n >= 0 && c >= 3, "voxelize: need n >= 0 and c >= 3 (got n=%d c=%d)", n, c);
    const VoxWs w = vox_layout(1, n);
    if (ws_bytes < w.bytes + 256) {
        set_error("voxelize: workspace %zu < required %zu", ws_bytes, w.bytes + 256);
        return TODA_EWORKSPACE;
    }
    // counts {m, m} land behind the layout; the caller's m_dev gets the first
    int32_t* counts = (int32_t*)((char*)ws + w.bytes);
    const float* pts[1] = {points};
    const int32_t ns[1] = {n};
    int rc = vox_run(pts, ns, 1, n, c, c, range_host, vsize_host, grid_host, max_pts, max_voxels, voxels, coords_zyx, 3, num_pts,
                     counts, ws, w.bytes, 0, (hipStream_t)stream);
    if (rc) return rc;
    TODA_HIP(hipMemcpyAsync(m_dev, counts, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return TODA_OK;
}

extern "C" size_t toda_voxelize_batch_workspace_bytes(int batch, int n_cap) { return vox_layout(batch, n_cap).bytes; }

extern "C" int toda_voxelize_batch(const float* const* points_host, const int32_t* n_points_host, int batch, int n_cap, int c,
                                   int row_stride, const float* range_host, const float* vsize_host, const int32_t* grid_host,
                                   int max_pts, int max_voxels, float* voxels, int32_t* coords_bzyx, int32_t* num_pts,
                                   int32_t* counts_dev, void* ws, size_t ws_bytes, int ws_clean, void* stream) {
    return vox_run(points_host, n_points_host, batch, n_cap, c, row_stride, range_host, vsize_host, grid_host, max_pts, max_voxels,
                   voxels, coords_bzyx, 4, num_pts, counts_dev, ws, ws_bytes, ws_clean, (hipStream_t)stream);
}

extern "C" int toda_mean_vfe_fwd(const float* voxels, const float* num_pts, int m, int p, int c, float* out,
                                 void* stream) {
    TODA_CHECK_ARG(m >= 0 && p >= 1 && c >= 1, "mean_vfe_fwd: bad shape m=%d p=%d c=%d", m, p, c);
    if (m == 0) return TODA_OK;
    hipLaunchKernelGGL(mean_vfe_fwd_kernel, dim3(cdiv((long long)m * c, VOX_BLOCK)), dim3(VOX_BLOCK), 0,
                       (hipStream_t)stream, voxels, num_pts, m, p, c, out);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_mean_vfe_bwd(const float* grad_out, const float* num_pts, int m, int p, int c, float* grad_voxels,
                                 void* stream) {
    TODA_CHECK_ARG(m >= 0 && p >= 1 && c >= 1, "mean_vfe_bwd: bad shape m=%d p=%d c=%d", m, p, c);
    if (m == 0) return TODA_OK;
    hipLaunchKernelGGL(mean_vfe_bwd_kernel, dim3(cdiv((long long)m * p * c, VOX_BLOCK)), dim3(VOX_BLOCK), 0,
                       (hipStream_t)stream, grad_out, num_pts, m, p, c, grad_voxels);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
