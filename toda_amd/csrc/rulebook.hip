// Grid index (bitmap + popcount rank) and rulebook (neighbour table) builders for gfx950.
//
// spconv builds its indice pairs with a GPU hash table; on MI355X the lattice of one sparse level
// fits comfortably in HBM as a bitmap (Waymo stride-1 level, batch 2: 185 M cells = 23 MB incl.
// prefix words), so membership + row id become one 8-byte load: cell = {32 occupancy bits,
// popcount of all earlier cells}; rank = cell.y + popc(cell.x & below).  Ranks enumerate sites in
// ascending ((b*D+z)*H+y)*W+x order, which is the canonical row order of every generated set, so
// strided-conv outputs need no sort and no hash probing.  Pure 4/8-byte index traffic: HBM / L2
// latency bound.
#include "scan.cuh"

namespace toda {

constexpr int RB_BLOCK = 256;

struct GridDims {
    int B, D, H, W;
};

struct GiLayout {
    long long cells;   // number of 32-bit words
    long long rows;    // B * D * H lattice rows (x runs)
    size_t o_cells, o_part, o_total, o_rows, bytes;
};

static GiLayout gi_layout(int batch, const int32_t* shape) {
    GiLayout l;
    long long bits = (long long)batch * shape[0] * shape[1] * shape[2];
    l.cells = (bits + 31) / 32;
    size_t o = 0;
    l.o_cells = o;
    o += align_up((size_t)l.cells * sizeof(uint2), 256);
    l.o_part = o;
    o += align_up((size_t)(cdiv(l.cells, RB_BLOCK) + 1) * sizeof(int32_t), 256);      // 256-word tiles (>= the scan's 2048-word ones)
    l.o_total = o;
    o += 256;
    // one byte per lattice row (b, z, y): "this row holds a site".  Kept by the O(sites) builder of the voxel level (mark / clear);
    // toda_gridindex_from_bitmap skips the input rows whose byte is zero (most of them at the voxel level: 123 k rows, 46 MB of words)
    l.rows = (long long)batch * shape[0] * shape[1];
    l.o_rows = o;
    o += align_up((size_t)l.rows, 256);
    l.bytes = o;
    return l;
}

__device__ __forceinline__ long long lin_index(int b, int z, int y, int x, const GridDims& g) {
    return (((long long)b * g.D + z) * g.H + y) * g.W + x;
}

// A coordinate row outside [0,B) x [0,D) x [0,H) x [0,W) (wrong batch_size, coordinates made for another grid, a
// hand-built SparseConvTensor) must never become a bitmap address: such rows are ignored by every builder - they set
// no bit, get no neighbours and are nobody's neighbour.
__device__ __forceinline__ bool in_grid(const int4 c, const GridDims& g) {
    return (unsigned)c.x < (unsigned)g.B && (unsigned)c.y < (unsigned)g.D && (unsigned)c.z < (unsigned)g.H &&
           (unsigned)c.w < (unsigned)g.W;
}

__device__ __forceinline__ int gi_rank(const uint2* __restrict__ cells, long long lin) {
    const uint2 c = cells[lin >> 5];
    const unsigned bit = 1u << (lin & 31);
    if (!(c.x & bit)) return -1;
    return (int)c.y + __popc(c.x & (bit - 1));
}

__global__ void __launch_bounds__(RB_BLOCK)
gi_mark_coords_kernel(const int4* __restrict__ idx, int n, const int32_t* __restrict__ n_dev, GridDims g,
                      uint2* __restrict__ cells) {
    n = eff_n(n, n_dev);
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = idx[i];  // (b, z, y, x)
    if (!in_grid(c, g)) return;
    const long long lin = lin_index(c.x, c.y, c.z, c.w, g);
    atomicOr(&cells[lin >> 5].x, 1u << (lin & 31));
}

__global__ void __launch_bounds__(RB_BLOCK)
gi_rowof_kernel(const int4* __restrict__ idx, int n, const int32_t* __restrict__ n_dev, GridDims g,
                const uint2* __restrict__ cells, int* __restrict__ rowof) {
    n = eff_n(n, n_dev);
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = idx[i];
    if (!in_grid(c, g)) return;
    const int r = gi_rank(cells, lin_index(c.x, c.y, c.z, c.w, g));
    rowof[r] = i;
}

struct ConvGeom {
    int ks[3], st[3], pd[3];
    GridDims out;
};

// output coordinate reached from input coordinate `i` through tap `k`, or -1
__device__ __forceinline__ int out_coord(int i, int k, int s, int p, int out_dim) {
    int t = i + p - k;
    if (t < 0) return -1;
    if (s != 1) {
        if (t % s) return -1;
        t /= s;
    }
    return t < out_dim ? t : -1;
}

__global__ void __launch_bounds__(RB_BLOCK)
gi_mark_conv_kernel(const int4* __restrict__ idx, int n, const int32_t* __restrict__ n_dev, ConvGeom cg,
                    uint2* __restrict__ cells) {
    n = eff_n(n, n_dev);
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = idx[i];
    if ((unsigned)c.x >= (unsigned)cg.out.B) return;      // the spatial axes are clamped by out_coord
    for (int kz = 0; kz < cg.ks[0]; ++kz) {
        const int zo = out_coord(c.y, kz, cg.st[0], cg.pd[0], cg.out.D);
        if (zo < 0) continue;
        for (int ky = 0; ky < cg.ks[1]; ++ky) {
            const int yo = out_coord(c.z, ky, cg.st[1], cg.pd[1], cg.out.H);
            if (yo < 0) continue;
            for (int kx = 0; kx < cg.ks[2]; ++kx) {
                const int xo = out_coord(c.w, kx, cg.st[2], cg.pd[2], cg.out.W);
                if (xo < 0) continue;
                const long long lin = lin_index(c.x, zo, yo, xo, cg.out);
                const unsigned bit = 1u << (lin & 31);
                unsigned* word = &cells[lin >> 5].x;
                if (!(*word & bit)) atomicOr(word, bit);  // plain pre-test: most bits are already set
            }
        }
    }
}

// one thread per cell: write the coordinates of its set bits at their ranks
__global__ void __launch_bounds__(RB_BLOCK)
gi_decode_kernel(const uint2* __restrict__ cells, long long n_cells, GridDims g, int4* __restrict__ idx_out, int cap,
                 const int32_t* __restrict__ total_dev, int32_t* __restrict__ n_out_dev) {
    const long long w = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    if (w == 0 && n_out_dev) *n_out_dev = *total_dev;
    if (w >= n_cells) return;
    const uint2 c = cells[w];
    unsigned bits = c.x;
    int r = (int)c.y;
    while (bits) {
        const int t = __ffs(bits) - 1;
        bits &= bits - 1;
        long long lin = (w << 5) + t;
        if (r < cap) {
            int x = (int)(lin % g.W);
            lin /= g.W;
            int y = (int)(lin % g.H);
            lin /= g.H;
            int z = (int)(lin % g.D);
            int b = (int)(lin / g.D);
            idx_out[r] = make_int4(b, z, y, x);
        }
        ++r;
    }
}

// ---- round 4: O(sites) index of an UNORDERED coordinate list (the voxel level) ------------------------------------------
// The voxel level is the big lattice (Waymo, batch 2: 185 M cells = 46 MB of {bits, prefix} words for 0.3 M sites) and its
// rows are in voxel order anyway (rowof maps rank -> row), so its ranks need not be canonical: instead of a popcount scan
// over every word, the site that set the LOWEST bit of a word allocates popc(word) consecutive ranks for it from a counter
// (one atomic per wave).  mark -> alloc -> rowof touch only the occupied words; gi_clear puts them back to zero afterwards,
// so neither a memset nor a scan ever sweeps the lattice.  cell.y = first rank of the word, exactly what gi_rank reads.
__global__ void __launch_bounds__(RB_BLOCK)
gi_mark_first_kernel(const int4* __restrict__ idx, int n, const int32_t* __restrict__ n_dev, GridDims g,
                     uint2* __restrict__ cells, unsigned char* __restrict__ rowmask, int* __restrict__ first,
                     int32_t* __restrict__ counter) {
    n = eff_n(n, n_dev);
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (i == 0) *counter = 0;
    if (i >= n) return;
    const int4 c = idx[i];
    int mine = 0;
    if (in_grid(c, g)) {
        const long long lin = lin_index(c.x, c.y, c.z, c.w, g);
        const unsigned bit = 1u << (lin & 31);
        mine = (atomicOr(&cells[lin >> 5].x, bit) & bit) ? 0 : 1;      // exactly one of several rows with the same coordinate
        rowmask[((long long)c.x * g.D + c.y) * g.H + c.z] = 1;          // plain store: every writer stores the same value
    }
    first[i] = mine;
}

constexpr int GA_BLOCK = 1024;
__global__ void __launch_bounds__(GA_BLOCK)
gi_alloc_kernel(const int4* __restrict__ idx, int n, const int32_t* __restrict__ n_dev, GridDims g, uint2* __restrict__ cells,
                const int* __restrict__ first, int32_t* __restrict__ counter) {
    __shared__ int s_w[GA_BLOCK / 64];
    __shared__ int s_base;
    n = eff_n(n, n_dev);
    const int i = blockIdx.x * GA_BLOCK + threadIdx.x;
    int want = 0;
    long long w = 0;
    if (i < n && first[i]) {
        const int4 c = idx[i];
        const long long lin = lin_index(c.x, c.y, c.z, c.w, g);
        w = lin >> 5;
        const unsigned bits = cells[w].x, bit = 1u << (lin & 31);
        if (!(bits & (bit - 1))) want = __popc(bits);       // this row set the word's lowest bit: it allocates for the word
    }
    // ONE atomic per 1024 rows: adds to one address serialise at ~12 ns each at the memory side (one per wave: 58-120 us for a
    // 300 k-row level, measured; one per block: ~4 us)
    int inc = want;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < GA_BLOCK / 64; ++k) {
        const int c = s_w[k];
        before += k < wv ? c : 0;
        total += c;
    }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(counter, total) : 0;
    __syncthreads();
    if (want) cells[w].y = (unsigned)(s_base + before + inc - want);
}

struct ClearLevel {
    const int4* idx;
    const int32_t* n_dev;
    uint2* cells;
    unsigned char* rowmask;
    GridDims g;
    int n;
};
constexpr int CLEAR_MAX = 8;
struct ClearArgs {
    ClearLevel lv[CLEAR_MAX];
};

// un-mark: the words of the listed sites back to {0, 0} (plain stores; rows sharing a word store the same value)
__global__ void __launch_bounds__(RB_BLOCK) gi_clear_kernel(ClearArgs a) {
    const ClearLevel& l = a.lv[blockIdx.y];
    const int n = eff_n(l.n, l.n_dev);
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 c = l.idx[i];
    if (!in_grid(c, l.g)) return;
    l.cells[lin_index(c.x, c.y, c.z, c.w, l.g) >> 5] = make_uint2(0u, 0u);
    l.rowmask[((long long)c.x * l.g.D + c.y) * l.g.H + c.z] = 0;
}

// ---- round 4: output set of a strided convolution from the input BITMAP, no atomics ---------------------------------------
// gi_mark_conv_kernel sets up to prod(ceil(k / s)) output bits per input site with device atomics (1 M for the voxel level
// of a Waymo batch; they execute at the memory side at ~20 G/s: 55 us per level on average, the most expensive index kernel
// of round 3).  Output-stationary instead: one thread per output WORD ORs the input rows (kz, ky) that reach it - read as
// unaligned bit windows of the input bitmap - and compacts the x axis with the stride.  Every word of the output bitmap is
// written (zero or not): it needs no clearing, and the block's popcount goes straight to the scan's partial sums.
__device__ __forceinline__ unsigned bits32_at(const uint2* __restrict__ cells, long long p, long long lo, long long hi) {
    // 32 bitmap bits starting at linear position p (any alignment); positions outside [lo, hi) read as zero
    const long long a = p > lo ? p : lo, e = (p + 32) < hi ? (p + 32) : hi;
    if (a >= e) return 0u;
    const long long w0 = a >> 5;
    const unsigned lo_word = cells[w0].x;
    const unsigned hi_word = ((e - 1) >> 5) > w0 ? cells[w0 + 1].x : 0u;
    const unsigned long long v = (((unsigned long long)hi_word << 32) | lo_word) >> (a & 31);
    const unsigned cnt = (unsigned)(e - a);
    const unsigned m = cnt >= 32u ? ~0u : ((1u << cnt) - 1u);
    return ((unsigned)v & m) << (unsigned)(a - p);
}

constexpr int CONV_WIN = 4;      // 32-bit windows per input row segment: (count - 1) * sx + KX <= 128 bits

__global__ void __launch_bounds__(RB_BLOCK)
gi_conv_bits_kernel(const uint2* __restrict__ cells_in, const unsigned char* __restrict__ rowmask_in, GridDims gin, ConvGeom cg,
                    uint2* __restrict__ cells_out, long long n_cells_out, int32_t* __restrict__ partials) {
    const long long w = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    const GridDims go = cg.out;
    const long long total_bits = (long long)go.B * go.D * go.H * go.W;
    unsigned word = 0u;
    if (w < n_cells_out) {
        long long pos = w << 5;
        const long long end = (pos + 32) < total_bits ? (pos + 32) : total_bits;
        while (pos < end) {
            // the run of this word inside one output row (b, zo, yo)
            int xo, yo, zo, b;
            if (total_bits < (1LL << 31)) {      // (wave uniform; 64-bit divisions are ~100 instructions each)
                unsigned t = (unsigned)pos;
                xo = (int)(t % (unsigned)go.W);
                t /= (unsigned)go.W;
                yo = (int)(t % (unsigned)go.H);
                t /= (unsigned)go.H;
                zo = (int)(t % (unsigned)go.D);
                b = (int)(t / (unsigned)go.D);
            } else {
                long long t = pos;
                xo = (int)(t % go.W);
                t /= go.W;
                yo = (int)(t % go.H);
                t /= go.H;
                zo = (int)(t % go.D);
                b = (int)(t / go.D);
            }
            const int room = go.W - xo;
            const int count = (int)((end - pos) < room ? (end - pos) : room);
            const int xi0 = xo * cg.st[2] - cg.pd[2];
            unsigned win[CONV_WIN] = {0u, 0u, 0u, 0u};
            const int nwin = ((count - 1) * cg.st[2] + cg.ks[2] + 31) >> 5;
            for (int kz = 0; kz < cg.ks[0]; ++kz) {
                const int zi = zo * cg.st[0] - cg.pd[0] + kz;
                if ((unsigned)zi >= (unsigned)gin.D) continue;
                for (int ky = 0; ky < cg.ks[1]; ++ky) {
                    const int yi = yo * cg.st[1] - cg.pd[1] + ky;
                    if ((unsigned)yi >= (unsigned)gin.H) continue;
                    const long long rix = ((long long)b * gin.D + zi) * gin.H + yi;
                    if (rowmask_in && !rowmask_in[rix]) continue;       // an input row without a site (one byte instead of up to 8 words)
                    const long long row = rix * gin.W;
#pragma unroll
                    for (int q = 0; q < CONV_WIN; ++q)
                        if (q < nwin) win[q] |= bits32_at(cells_in, row + xi0 + 32 * q, row, row + gin.W);
                }
            }
            // output bit j = OR over kx of window bit j * sx + kx
            unsigned seg = 0u;
            const unsigned long long lo64 = ((unsigned long long)win[1] << 32) | win[0], hi64 = ((unsigned long long)win[3] << 32) | win[2];
            if ((lo64 | hi64) && cg.st[2] <= 2 && cg.ks[2] <= 8) {
                // stride 1 / 2 (every layer of the reference's backbones): shift-OR over the taps, then keep every stride-th bit
                unsigned long long u = 0ull;
                for (int kx = 0; kx < cg.ks[2]; ++kx) u |= kx ? ((lo64 >> kx) | (hi64 << (64 - kx))) : lo64;
                if (cg.st[2] == 2) {
                    u &= 0x5555555555555555ull;
                    u = (u | (u >> 1)) & 0x3333333333333333ull;
                    u = (u | (u >> 2)) & 0x0f0f0f0f0f0f0f0full;
                    u = (u | (u >> 4)) & 0x00ff00ff00ff00ffull;
                    u = (u | (u >> 8)) & 0x0000ffff0000ffffull;
                    u = (u | (u >> 16)) & 0x00000000ffffffffull;
                }
                seg = (unsigned)u & (count >= 32 ? ~0u : ((1u << count) - 1u));
            } else if (lo64 | hi64) {
                for (int j = 0; j < count; ++j) {
                    unsigned hit = 0u;
                    for (int kx = 0; kx < cg.ks[2]; ++kx) {
                        const int bp = j * cg.st[2] + kx;
                        hit |= (win[bp >> 5] >> (bp & 31)) & 1u;
                    }
                    seg |= hit << j;
                }
            }
            word |= seg << (unsigned)(pos - (w << 5));
            pos += count;
        }
        cells_out[w] = make_uint2(word, 0u);
    }
    int tot;
    block_exclusive_scan(__popc(word), &tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// ranks of a bitmap whose per-256-word popcounts are in `partials`: every block sums the partials before it, scans its own
// words, stores the prefix words and decodes its sites in canonical order; the last thread publishes the total
__global__ void __launch_bounds__(RB_BLOCK)
gi_scan_decode_kernel(uint2* __restrict__ cells, long long n_cells, const int32_t* __restrict__ partials, GridDims g,
                      int4* __restrict__ idx_out, int cap, int32_t* __restrict__ total_dev, int32_t* __restrict__ n_out_dev) {
    int s = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += RB_BLOCK) s += partials[k];
    int before;
    block_exclusive_scan(s, &before);
    const long long w = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    unsigned bits = w < n_cells ? cells[w].x : 0u;
    int tot;
    int r = before + block_exclusive_scan(__popc(bits), &tot);
    if (w < n_cells) cells[w].y = (unsigned)r;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == RB_BLOCK - 1) {
        const int all = r + __popc(bits);
        *total_dev = all;
        if (n_out_dev) *n_out_dev = all;
    }
    if (bits) {
        // coordinates of the word's first cell once (32-bit arithmetic when the lattice allows), then carries per set bit
        int x, y, z, b;
        const long long lin0 = w << 5;
        if (n_cells < (1LL << 26)) {
            unsigned t = (unsigned)lin0;
            x = (int)(t % (unsigned)g.W);
            t /= (unsigned)g.W;
            y = (int)(t % (unsigned)g.H);
            t /= (unsigned)g.H;
            z = (int)(t % (unsigned)g.D);
            b = (int)(t / (unsigned)g.D);
        } else {
            long long t = lin0;
            x = (int)(t % g.W);
            t /= g.W;
            y = (int)(t % g.H);
            t /= g.H;
            z = (int)(t % g.D);
            b = (int)(t / g.D);
        }
        int at = 0;
        while (bits) {
            const int t = __ffs(bits) - 1;
            bits &= bits - 1;
            x += t - at;
            at = t;
            while (x >= g.W) {          // the word runs over the end of a row (W < 32 or no multiple of 32)
                x -= g.W;
                if (++y == g.H) {
                    y = 0;
                    if (++z == g.D) {
                        z = 0;
                        ++b;
                    }
                }
            }
            if (r < cap) idx_out[r] = make_int4(b, z, y, x);
            ++r;
        }
    }
}

// per-block per-offset pair counters: wave ballots -> LDS -> one global atomic per block and offset
__device__ __forceinline__ void count_pairs(bool valid, int k, int* s_cnt) {
    const unsigned long long vote = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && vote) atomicAdd(&s_cnt[k], __popcll(vote));
}

// SubM rulebook, one thread per output site.  The K lookups of a site are independent, so they
// are issued as K back-to-back 8-byte cell loads (then K rowof loads) before anything is consumed:
// one round trip to L2/MALL per phase instead of one per offset.  k-major table => the stores of
// a wave are 256 contiguous bytes per offset.
template <int KZ, int KY, int KX>
__global__ void __launch_bounds__(RB_BLOCK)
rb_subm_kernel(const int4* __restrict__ idx, int n, GridDims g, int dz, int dy, int dx,
               const uint2* __restrict__ cells, const int* __restrict__ rowof, int* __restrict__ nbr,
               int* __restrict__ pair_cnt) {
    constexpr int K = KZ * KY * KX;
    __shared__ int s_cnt[K];
    if (threadIdx.x < K) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int o = blockIdx.x * RB_BLOCK + threadIdx.x;
    const bool live = o < n;
    int4 c = make_int4(0, 0, 0, 0);
    if (live) c = idx[o];
    const bool inside = live && in_grid(c, g);
    uint2 cell[K];
    unsigned bit[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int kz = k / (KY * KX), ky = (k / KX) % KY, kx = k % KX;
        const int z = c.y + (kz - KZ / 2) * dz, y = c.z + (ky - KY / 2) * dy, x = c.w + (kx - KX / 2) * dx;
        const bool ok = inside && z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W;
        cell[k] = make_uint2(0u, 0u);
        bit[k] = 0u;
        if (ok) {
            const long long lin = lin_index(c.x, z, y, x, g);
            cell[k] = cells[lin >> 5];
            bit[k] = 1u << (lin & 31);
        }
    }
    int r[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        r[k] = (cell[k].x & bit[k]) ? (int)cell[k].y + __popc(cell[k].x & (bit[k] - 1)) : -1;
    if (rowof) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (r[k] >= 0) r[k] = rowof[r[k]];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (live) nbr[(size_t)k * n + o] = r[k];
        count_pairs(r[k] >= 0, k, s_cnt);
    }
    __syncthreads();
    if (threadIdx.x < K && s_cnt[threadIdx.x]) atomicAdd(&pair_cnt[threadIdx.x], s_cnt[threadIdx.x]);
}

// generic kernel sizes (K <= 64): same structure, runtime loops
__global__ void __launch_bounds__(RB_BLOCK)
rb_subm_generic_kernel(const int4* __restrict__ idx, int n, GridDims g, int kz_n, int ky_n, int kx_n, int dz, int dy,
                       int dx, const uint2* __restrict__ cells, const int* __restrict__ rowof, int* __restrict__ nbr,
                       int* __restrict__ pair_cnt) {
    __shared__ int s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int o = blockIdx.x * RB_BLOCK + threadIdx.x;
    const bool live = o < n;
    int4 c = make_int4(0, 0, 0, 0);
    if (live) c = idx[o];
    int k = 0;
    for (int kz = 0; kz < kz_n; ++kz)
        for (int ky = 0; ky < ky_n; ++ky)
            for (int kx = 0; kx < kx_n; ++kx, ++k) {
                int r = -1;
                if (live) {
                    const int z = c.y + (kz - kz_n / 2) * dz, y = c.z + (ky - ky_n / 2) * dy,
                              x = c.w + (kx - kx_n / 2) * dx;
                    if (in_grid(c, g) && z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W) {
                        r = gi_rank(cells, lin_index(c.x, z, y, x, g));
                        if (r >= 0 && rowof) r = rowof[r];
                    }
                    nbr[(size_t)k * n + o] = r;
                }
                count_pairs(r >= 0, k, s_cnt);
            }
    __syncthreads();
    if (threadIdx.x < k && s_cnt[threadIdx.x]) atomicAdd(&pair_cnt[threadIdx.x], s_cnt[threadIdx.x]);
}

// strided conv rulebook, one thread per input site; same issue-all-loads-first structure.
// o2i == nullptr: only the input -> output table and the pair counts (the output -> input table then comes from
// rb_conv_o2i_kernel, output stationary, with complete coalesced stores instead of a 0xFF fill + scattered 4-byte stores)
template <int KZ, int KY, int KX>
__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_kernel(const int4* __restrict__ idx, int n_in, ConvGeom cg, const uint2* __restrict__ cells, int n_out,
               int* __restrict__ o2i, int* __restrict__ i2o, int* __restrict__ pair_cnt) {
    constexpr int K = KZ * KY * KX;
    __shared__ int s_cnt[K];
    if (threadIdx.x < K) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    const bool live = i < n_in;
    int4 c = make_int4(0, 0, 0, 0);
    if (live) c = idx[i];
    int zo[KZ], yo[KY], xo[KX];
#pragma unroll
    for (int a = 0; a < KZ; ++a) zo[a] = out_coord(c.y, a, cg.st[0], cg.pd[0], cg.out.D);
#pragma unroll
    for (int a = 0; a < KY; ++a) yo[a] = out_coord(c.z, a, cg.st[1], cg.pd[1], cg.out.H);
#pragma unroll
    for (int a = 0; a < KX; ++a) xo[a] = out_coord(c.w, a, cg.st[2], cg.pd[2], cg.out.W);
    uint2 cell[K];
    unsigned bit[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int kz = k / (KY * KX), ky = (k / KX) % KY, kx = k % KX;
        cell[k] = make_uint2(0u, 0u);
        bit[k] = 0u;
        if (live && (unsigned)c.x < (unsigned)cg.out.B && zo[kz] >= 0 && yo[ky] >= 0 && xo[kx] >= 0) {
            const long long lin = lin_index(c.x, zo[kz], yo[ky], xo[kx], cg.out);
            cell[k] = cells[lin >> 5];
            bit[k] = 1u << (lin & 31);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int o = (cell[k].x & bit[k]) ? (int)cell[k].y + __popc(cell[k].x & (bit[k] - 1)) : -1;
        if (o >= n_out) o = -1;
        if (o >= 0 && o2i) o2i[(size_t)k * n_out + o] = i;
        if (live) i2o[(size_t)k * n_in + i] = o;
        count_pairs(o >= 0, k, s_cnt);
    }
    __syncthreads();
    if (threadIdx.x < K && s_cnt[threadIdx.x]) atomicAdd(&pair_cnt[threadIdx.x], s_cnt[threadIdx.x]);
}

// output -> input table of a strided convolution, one thread per OUTPUT site: the input coordinate behind tap k is
// o * stride - pad + k; its row comes from the INPUT level's grid index (rank, then rowof for the voxel level).
template <int KZ, int KY, int KX>
__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_o2i_kernel(const int4* __restrict__ idx_out, int n_out, ConvGeom cg, GridDims gin, const uint2* __restrict__ cells_in,
                   const int* __restrict__ rowof_in, int n_in, int* __restrict__ o2i) {
    constexpr int K = KZ * KY * KX;
    const int o = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (o >= n_out) return;
    const int4 c = idx_out[o];
    uint2 cell[K];
    unsigned bit[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int kz = k / (KY * KX), ky = (k / KX) % KY, kx = k % KX;
        const int z = c.y * cg.st[0] - cg.pd[0] + kz, y = c.z * cg.st[1] - cg.pd[1] + ky, x = c.w * cg.st[2] - cg.pd[2] + kx;
        cell[k] = make_uint2(0u, 0u);
        bit[k] = 0u;
        if ((unsigned)c.x < (unsigned)gin.B && (unsigned)z < (unsigned)gin.D && (unsigned)y < (unsigned)gin.H &&
            (unsigned)x < (unsigned)gin.W) {
            const long long lin = lin_index(c.x, z, y, x, gin);
            cell[k] = cells_in[lin >> 5];
            bit[k] = 1u << (lin & 31);
        }
    }
    int r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        r[k] = (cell[k].x & bit[k]) ? (int)cell[k].y + __popc(cell[k].x & (bit[k] - 1)) : -1;
        if (r[k] >= n_in) r[k] = -1;
    }
    if (rowof_in) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (r[k] >= 0) r[k] = rowof_in[r[k]];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) o2i[(size_t)k * n_out + o] = r[k];
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_generic_kernel(const int4* __restrict__ idx, int n_in, ConvGeom cg, const uint2* __restrict__ cells, int n_out,
                       int* __restrict__ o2i, int* __restrict__ i2o, int* __restrict__ pair_cnt) {
    __shared__ int s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * RB_BLOCK + threadIdx.x;
    const bool live = i < n_in;
    int4 c = make_int4(0, 0, 0, 0);
    if (live) c = idx[i];
    int k = 0;
    for (int kz = 0; kz < cg.ks[0]; ++kz) {
        const int zo = out_coord(c.y, kz, cg.st[0], cg.pd[0], cg.out.D);
        for (int ky = 0; ky < cg.ks[1]; ++ky) {
            const int yo = out_coord(c.z, ky, cg.st[1], cg.pd[1], cg.out.H);
            for (int kx = 0; kx < cg.ks[2]; ++kx, ++k) {
                const int xo = out_coord(c.w, kx, cg.st[2], cg.pd[2], cg.out.W);
                int o = -1;
                if (live) {
                    if ((unsigned)c.x < (unsigned)cg.out.B && zo >= 0 && yo >= 0 && xo >= 0) {
                        o = gi_rank(cells, lin_index(c.x, zo, yo, xo, cg.out));
                        if (o >= n_out) o = -1;
                        if (o >= 0) o2i[(size_t)k * n_out + o] = i;
                    }
                    i2o[(size_t)k * n_in + i] = o;
                }
                count_pairs(o >= 0, k, s_cnt);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < k && s_cnt[threadIdx.x]) atomicAdd(&pair_cnt[threadIdx.x], s_cnt[threadIdx.x]);
}

static int check_geom(const char* who, int batch, const int32_t* shape) {
    TODA_CHECK_ARG(batch >= 1 && shape[0] >= 1 && shape[1] >= 1 && shape[2] >= 1, "%s: bad batch/shape", who);
    TODA_CHECK_ARG((long long)batch * shape[0] * shape[1] * shape[2] < (1LL << 36), "%s: lattice too large", who);
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

extern "C" size_t toda_gridindex_bytes(int batch, const int32_t* shape_host) {
    return gi_layout(batch, shape_host).bytes;
}

extern "C" int toda_gridindex_from_coords(const int32_t* idx, int n, const int32_t* n_dev, int batch,
                                          const int32_t* shape_host, void* gi, int32_t* rowof, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_geom("gridindex_from_coords", batch, shape_host);
    if (rc) return rc;
    TODA_CHECK_ARG(n >= 0, "gridindex_from_coords: n < 0");
    const GiLayout l = gi_layout(batch, shape_host);
    char* b = (char*)gi;
    uint2* cells = (uint2*)(b + l.o_cells);
    TODA_HIP(hipMemsetAsync(cells, 0, (size_t)l.cells * sizeof(uint2), s));
    const GridDims g{batch, shape_host[0], shape_host[1], shape_host[2]};
    if (n > 0)
        hipLaunchKernelGGL(gi_mark_coords_kernel, dim3(cdiv(n, RB_BLOCK)), dim3(RB_BLOCK), 0, s, (const int4*)idx, n,
                           n_dev, g, cells);
    rc = exclusive_scan(CellAccess{cells}, l.cells, (int32_t*)(b + l.o_part), (int32_t*)(b + l.o_total), s);
    if (rc) return rc;
    if (n > 0 && rowof)
        hipLaunchKernelGGL(gi_rowof_kernel, dim3(cdiv(n, RB_BLOCK)), dim3(RB_BLOCK), 0, s, (const int4*)idx, n, n_dev, g,
                           cells, rowof);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_gridindex_from_coords_unordered(const int32_t* idx, int n, const int32_t* n_dev, int batch,
                                                    const int32_t* shape_host, void* gi, int32_t* rowof, int gi_clean,
                                                    void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_geom("gridindex_from_coords_unordered", batch, shape_host);
    if (rc) return rc;
    TODA_CHECK_ARG(n >= 0 && rowof, "gridindex_from_coords_unordered: n < 0 or no rowof");
    const GiLayout l = gi_layout(batch, shape_host);
    char* b = (char*)gi;
    uint2* cells = (uint2*)(b + l.o_cells);
    int32_t* counter = (int32_t*)(b + l.o_total);
    unsigned char* rowmask = (unsigned char*)(b + l.o_rows);
    if (!gi_clean) {
        TODA_HIP(hipMemsetAsync(cells, 0, (size_t)l.cells * sizeof(uint2), s));
        TODA_HIP(hipMemsetAsync(rowmask, 0, (size_t)l.rows, s));
    }
    if (n == 0) return TODA_OK;
    const GridDims g{batch, shape_host[0], shape_host[1], shape_host[2]};
    const dim3 grid(cdiv(n, RB_BLOCK)), block(RB_BLOCK);
    // rowof doubles as the "this row set its bit first" flags between mark and alloc (rowof[rank] = row is written last)
    hipLaunchKernelGGL(gi_mark_first_kernel, grid, block, 0, s, (const int4*)idx, n, n_dev, g, cells, rowmask, rowof, counter);
    hipLaunchKernelGGL(gi_alloc_kernel, dim3(cdiv(n, GA_BLOCK)), dim3(GA_BLOCK), 0, s, (const int4*)idx, n, n_dev, g, cells, (const int*)rowof,
                       counter);
    hipLaunchKernelGGL(gi_rowof_kernel, grid, block, 0, s, (const int4*)idx, n, n_dev, g, cells, rowof);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_gridindex_clear(const int32_t* idx, int n, const int32_t* n_dev, int batch, const int32_t* shape_host,
                                    void* gi, void* stream) {
    int rc = check_geom("gridindex_clear", batch, shape_host);
    if (rc) return rc;
    TODA_CHECK_ARG(n >= 0, "gridindex_clear: n < 0");
    if (n == 0) return TODA_OK;
    const GiLayout l = gi_layout(batch, shape_host);
    ClearArgs a;
    a.lv[0].idx = (const int4*)idx;
    a.lv[0].n_dev = n_dev;
    a.lv[0].cells = (uint2*)((char*)gi + l.o_cells);
    a.lv[0].rowmask = (unsigned char*)gi + l.o_rows;
    a.lv[0].g = GridDims{batch, shape_host[0], shape_host[1], shape_host[2]};
    a.lv[0].n = n;
    hipLaunchKernelGGL(gi_clear_kernel, dim3(cdiv(n, RB_BLOCK), 1), dim3(RB_BLOCK), 0, (hipStream_t)stream, a);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

static int fill_conv_geom(const char* who, int batch, const int32_t* shape_in, const int32_t* ks, const int32_t* st,
                          const int32_t* pd, const int32_t* shape_out, ConvGeom* cg) {
    for (int a = 0; a < 3; ++a) {
        TODA_CHECK_ARG(ks[a] >= 1 && st[a] >= 1 && pd[a] >= 0, "%s: bad kernel/stride/pad on axis %d", who, a);
        const int expect = (shape_in[a] + 2 * pd[a] - ks[a]) / st[a] + 1;
        TODA_CHECK_ARG(shape_out[a] == expect, "%s: shape_out[%d]=%d but (in+2p-k)/s+1=%d", who, a, shape_out[a], expect);
        cg->ks[a] = ks[a];
        cg->st[a] = st[a];
        cg->pd[a] = pd[a];
    }
    TODA_CHECK_ARG(ks[0] * ks[1] * ks[2] <= 64, "%s: kernel volume > 64 unsupported", who);
    cg->out = GridDims{batch, shape_out[0], shape_out[1], shape_out[2]};
    return TODA_OK;
}

extern "C" int toda_gridindex_from_conv(const int32_t* idx_in, int n_in, const int32_t* n_in_dev, int batch,
                                        const int32_t* shape_in_host, const int32_t* ksize_host,
                                        const int32_t* stride_host, const int32_t* pad_host,
                                        const int32_t* shape_out_host, void* gi_out, int32_t* idx_out,
                                        int32_t* n_out_dev, int out_cap, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_geom("gridindex_from_conv", batch, shape_out_host);
    if (rc) return rc;
    ConvGeom cg;
    rc = fill_conv_geom("gridindex_from_conv", batch, shape_in_host, ksize_host, stride_host, pad_host, shape_out_host, &cg);
    if (rc) return rc;
    TODA_CHECK_ARG(n_in >= 0 && out_cap >= 0, "gridindex_from_conv: negative size");
    const GiLayout l = gi_layout(batch, shape_out_host);
    char* b = (char*)gi_out;
    uint2* cells = (uint2*)(b + l.o_cells);
    int32_t* total = (int32_t*)(b + l.o_total);
    TODA_HIP(hipMemsetAsync(cells, 0, (size_t)l.cells * sizeof(uint2), s));
    if (n_in > 0)
        hipLaunchKernelGGL(gi_mark_conv_kernel, dim3(cdiv(n_in, RB_BLOCK)), dim3(RB_BLOCK), 0, s, (const int4*)idx_in,
                           n_in, n_in_dev, cg, cells);
    rc = exclusive_scan(CellAccess{cells}, l.cells, (int32_t*)(b + l.o_part), total, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gi_decode_kernel, dim3(cdiv(l.cells, RB_BLOCK)), dim3(RB_BLOCK), 0, s, cells, l.cells, cg.out,
                       (int4*)idx_out, out_cap, total, n_out_dev);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_gridindex_from_bitmap(const void* gi_in, int batch, const int32_t* shape_in_host,
                                          const int32_t* ksize_host, const int32_t* stride_host, const int32_t* pad_host,
                                          const int32_t* shape_out_host, void* gi_out, int32_t* idx_out, int32_t* n_out_dev,
                                          int out_cap, int in_rows_marked, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_geom("gridindex_from_bitmap", batch, shape_out_host);
    if (rc) return rc;
    rc = check_geom("gridindex_from_bitmap", batch, shape_in_host);
    if (rc) return rc;
    ConvGeom cg;
    rc = fill_conv_geom("gridindex_from_bitmap", batch, shape_in_host, ksize_host, stride_host, pad_host, shape_out_host, &cg);
    if (rc) return rc;
    TODA_CHECK_ARG(out_cap >= 0, "gridindex_from_bitmap: negative size");
    TODA_CHECK_ARG(31 * stride_host[2] + ksize_host[2] <= 32 * CONV_WIN, "gridindex_from_bitmap: x stride %d / kernel %d too wide",
                   stride_host[2], ksize_host[2]);
    const GiLayout li = gi_layout(batch, shape_in_host), lo = gi_layout(batch, shape_out_host);
    const uint2* cells_in = (const uint2*)((const char*)gi_in + li.o_cells);
    char* b = (char*)gi_out;
    uint2* cells = (uint2*)(b + lo.o_cells);
    int32_t* part = (int32_t*)(b + lo.o_part);
    int32_t* total = (int32_t*)(b + lo.o_total);
    const GridDims gin{batch, shape_in_host[0], shape_in_host[1], shape_in_host[2]};
    const int nb = cdiv(lo.cells, RB_BLOCK);
    const unsigned char* rowmask_in = in_rows_marked ? (const unsigned char*)gi_in + li.o_rows : nullptr;
    hipLaunchKernelGGL(gi_conv_bits_kernel, dim3(nb), dim3(RB_BLOCK), 0, s, cells_in, rowmask_in, gin, cg, cells, lo.cells, part);
    hipLaunchKernelGGL(gi_scan_decode_kernel, dim3(nb), dim3(RB_BLOCK), 0, s, cells, lo.cells, (const int32_t*)part, cg.out,
                       (int4*)idx_out, out_cap, total, n_out_dev);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_rulebook_subm(const int32_t* idx, int n, int batch, const int32_t* shape_host,
                                  const int32_t* ksize_host, const int32_t* dilation_host, const void* gi,
                                  const int32_t* rowof, int32_t* nbr, int32_t* pair_cnt, int cnt_zeroed, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_geom("rulebook_subm", batch, shape_host);
    if (rc) return rc;
    const int K = ksize_host[0] * ksize_host[1] * ksize_host[2];
    TODA_CHECK_ARG(K >= 1 && K <= 64, "rulebook_subm: kernel volume %d unsupported", K);
    for (int a = 0; a < 3; ++a)
        TODA_CHECK_ARG(ksize_host[a] % 2 == 1 && dilation_host[a] >= 1, "rulebook_subm: kernel must be odd, dilation >= 1");
    if (!cnt_zeroed) TODA_HIP(hipMemsetAsync(pair_cnt, 0, K * sizeof(int32_t), s));
    if (n == 0) return TODA_OK;
    const GiLayout l = gi_layout(batch, shape_host);
    const uint2* cells = (const uint2*)((const char*)gi + l.o_cells);
    const GridDims g{batch, shape_host[0], shape_host[1], shape_host[2]};
    const dim3 grid(cdiv(n, RB_BLOCK)), block(RB_BLOCK);
    if (ksize_host[0] == 3 && ksize_host[1] == 3 && ksize_host[2] == 3)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_subm_kernel<3, 3, 3>), grid, block, 0, s, (const int4*)idx, n, g,
                           dilation_host[0], dilation_host[1], dilation_host[2], cells, rowof, nbr, pair_cnt);
    else
        hipLaunchKernelGGL(rb_subm_generic_kernel, grid, block, 0, s, (const int4*)idx, n, g, ksize_host[0],
                           ksize_host[1], ksize_host[2], dilation_host[0], dilation_host[1], dilation_host[2], cells,
                           rowof, nbr, pair_cnt);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_rulebook_conv(const int32_t* idx_in, int n_in, int batch, const int32_t* shape_in_host,
                                  const int32_t* ksize_host, const int32_t* stride_host, const int32_t* pad_host,
                                  const int32_t* shape_out_host, const void* gi_out, int n_out, int32_t* nbr_o2i,
                                  int32_t* nbr_i2o, int32_t* pair_cnt, const int32_t* idx_out, const void* gi_in,
                                  const int32_t* rowof_in, int cnt_zeroed, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_geom("rulebook_conv", batch, shape_out_host);
    if (rc) return rc;
    ConvGeom cg;
    rc = fill_conv_geom("rulebook_conv", batch, shape_in_host, ksize_host, stride_host, pad_host, shape_out_host, &cg);
    if (rc) return rc;
    const int K = ksize_host[0] * ksize_host[1] * ksize_host[2];
    const bool k333 = ksize_host[0] == 3 && ksize_host[1] == 3 && ksize_host[2] == 3;
    const bool k311 = ksize_host[0] == 3 && ksize_host[1] == 1 && ksize_host[2] == 1;
    // output -> input table by its own output-stationary kernel when the caller has the input level's index and the output
    // coordinates at hand (the index plan does); otherwise 0xFF fill + scattered stores from the input side
    const bool by_output = idx_out && gi_in && (k333 || k311);
    if (!cnt_zeroed) TODA_HIP(hipMemsetAsync(pair_cnt, 0, K * sizeof(int32_t), s));
    if (n_out > 0 && !by_output) TODA_HIP(hipMemsetAsync(nbr_o2i, 0xFF, (size_t)K * n_out * sizeof(int32_t), s));
    const GiLayout l = gi_layout(batch, shape_out_host);
    const uint2* cells = (const uint2*)((const char*)gi_out + l.o_cells);
    const dim3 block(RB_BLOCK);
    if (by_output && n_out > 0) {
        rc = check_geom("rulebook_conv", batch, shape_in_host);
        if (rc) return rc;
        const GiLayout li = gi_layout(batch, shape_in_host);
        const uint2* cells_in = (const uint2*)((const char*)gi_in + li.o_cells);
        const GridDims gin{batch, shape_in_host[0], shape_in_host[1], shape_in_host[2]};
        const dim3 grid(cdiv(n_out, RB_BLOCK));
        if (k333)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_conv_o2i_kernel<3, 3, 3>), grid, block, 0, s, (const int4*)idx_out, n_out, cg, gin,
                               cells_in, rowof_in, n_in, nbr_o2i);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_conv_o2i_kernel<3, 1, 1>), grid, block, 0, s, (const int4*)idx_out, n_out, cg, gin,
                               cells_in, rowof_in, n_in, nbr_o2i);
    }
    if (n_in == 0) {
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    int32_t* o2i = by_output ? nullptr : nbr_o2i;
    const dim3 grid(cdiv(n_in, RB_BLOCK));
    if (k333)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_conv_kernel<3, 3, 3>), grid, block, 0, s, (const int4*)idx_in, n_in, cg,
                           cells, n_out, o2i, nbr_i2o, pair_cnt);
    else if (k311)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_conv_kernel<3, 1, 1>), grid, block, 0, s, (const int4*)idx_in, n_in, cg,
                           cells, n_out, o2i, nbr_i2o, pair_cnt);
    else
        hipLaunchKernelGGL(rb_conv_generic_kernel, grid, block, 0, s, (const int4*)idx_in, n_in, cg, cells, n_out,
                           nbr_o2i, nbr_i2o, pair_cnt);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}
