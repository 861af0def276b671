// Measured-and-rejected kernel families of the sparse convolution, kept as OPT-IN variants: compiled only with
// `make VARIANTS=1` (-DTODA_VARIANTS=1), textually included by spconv.hip behind its helpers.  The default libtoda_hip.so - the one
// bench.py times - contains none of this; the C ABI entry points that belong to these families then answer "built without
// TODA_VARIANTS" (variants_stub section of spconv.hip), and toda_variants_built() says which build is loaded.
//   gather_gemm_wres_kernel   TODA_GG_WRES=1     all K weight slices resident in LDS (<= 32 x 32 channels)
//   gather_gemm_ws_kernel     TODA_GG_WS=1       producer / consumer wave specialisation with LDS-DMA gathers (64 -> 64)
//   gather_gemm_stage_kernel  TODA_GG_STAGE=1    three offsets per LDS stage, 512-thread workgroups (64 -> 64)
//   gather_gemm_line_kernel   TODA_GG_LINE=1     x-run operand reuse over submanifold tables (round 4; 64 -> 64, 32 -> 32)
//   gather_gemm_wide_kernel   TODA_GG_LDS88=5    128 -> 128 with half slices by LDS-DMA, double buffered (round 4)
//   gather_gemm_halo_kernel   TODA_HALO=1        LDS-staged halo tiles (+ halo_plan.hip)
//   wgrad_tile_kernel         TODA_WG_TILE=1     dout-stationary sparse wgrad
//   row_order_kernel          TODA_ROW_ORDER=1   mask-sorted row order
//   instantiations of gather_gemm_lds_kernel / gather_gemm_kernel behind TODA_GG_LDS_PF, TODA_GG_BLK512, TODA_GG_RT, TODA_GG_PF,
//   TODA_GG_LDS88 in {1, 2, 4}
// Each family's numbers are in DESIGN.md section 7 and profiles/.
#pragma once
namespace toda {

// Weights-resident variant for the narrow layers (<= 32 x 32 channels: conv_input, conv1, conv2.*, spconv2 and their
// dgrads).  With 4 KiB or less of weights per offset a workgroup's MFMA work per offset is tiny (32 MFMAs per wave at
// 32 -> 32), so the per-offset barrier and the unpipelined gather of gather_gemm_lds_kernel set the pace (0.05-0.32 of the
// roof in round 1).  Here the packed weights of ALL K offsets (<= 108 KiB) are staged in LDS once per workgroup; after that
// single barrier the 16 waves of a workgroup are independent: each walks its row tiles with the neighbour ids of offset k+2
// and the gathered rows of offset k+1 in flight under the MFMAs of offset k, reading B fragments with ds_read_b128.  One
// 1024-thread workgroup per CU (4 waves per SIMD), persistent over its tiles.
constexpr int WR_BLOCK = 1024;
template <int Q, int NT, int RT, int KMAX>
__global__ void __launch_bounds__(WR_BLOCK)
gather_gemm_wres_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp, const int* __restrict__ nbr,
                        int n_out, int K, int cp, const float* __restrict__ bias, float* __restrict__ out) {
    __shared__ f32x4 wl[KMAX * Q * NT * 64];
    {
        const f32x4* __restrict__ wp4 = reinterpret_cast<const f32x4*>(wp);
        const int total = K * Q * NT * 64;
        for (int e = threadIdx.x; e < total; e += WR_BLOCK) wl[e] = wp4[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
    const int n_tiles = (n_out + 16 * RT - 1) / (16 * RT);
    const int wave0 = blockIdx.x * (WR_BLOCK / 64) + (threadIdx.x >> 6), wave_stride = gridDim.x * (WR_BLOCK / 64);
    float bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bv[n] = (bias && NT * r + n < cp) ? bias[NT * r + n] : 0.0f;

    for (int tile = wave0; tile < n_tiles; tile += wave_stride) {
        const int row0 = tile * (16 * RT);
        f32x4 acc[RT][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][n] = f32x4{bv[n], bv[n], bv[n], bv[n]};
        int rows[RT];
        bool live[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            live[rt] = row0 + rt * 16 + r < n_out;
            rows[rt] = live[rt] ? row0 + rt * 16 + r : n_out - 1;
        }
        auto load_ids = [&](int k, int (&dst)[RT]) {
            const int kk = k < K ? k : K - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int v = nbr[(size_t)kk * n_out + rows[rt]];
                dst[rt] = (k < K && live[rt]) ? v : -1;
            }
        };
        int s0[RT], s1[RT], s2[RT];
        f32x4 a0[RT][Q], a1[RT][Q];
        load_ids(0, s0);
        load_ids(1, s1);
        gather_rows<Q, RT, true>(in_rsrc, cg, g, s0, a0);
        for (int k = 0; k < K; ++k) {
            load_ids(k + 2, s2);
            gather_rows<Q, RT, true>(in_rsrc, cg, g, s1, a1);      // rows of offset k + 1, in flight during the MFMAs below
            bool hit[RT];
            bool any = false;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                hit[rt] = __any(s0[rt] >= 0);
                any = any || hit[rt];
            }
            if (any) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    f32x4 b[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = wl[((k * Q + q) * NT + n) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                if (hit[rt]) acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[rt][q][j], b[n][j], acc[rt][n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                s0[rt] = s1[rt];
                s1[rt] = s2[rt];
#pragma unroll
                for (int q = 0; q < Q; ++q) a0[rt][q] = a1[rt][q];
            }
        }
        const bool full = cp == 16 * NT;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = row0 + rt * 16 + 4 * g + reg;
                if (row >= n_out) continue;
                float* dst = out + (size_t)row * cp + NT * r;
                if (full) {
                    if constexpr (NT == 1) dst[0] = acc[rt][0][reg];
                    else *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][reg], acc[rt][1][reg]);
                } else {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if (NT * r + n < cp) dst[n] = acc[rt][n][reg];
                }
            }
        }
    }
}

// Wave-specialised variant for the 64 -> 64 layers (the dominant launches of the step; forward and dgrad).
// gather_gemm_lds_kernel keeps the matrix pipe 71 % busy: every wave gathers its own A fragments (dependent id -> row loads)
// between its MFMA bursts and meets the other waves at a barrier per offset.  Here a 512-thread workgroup owns 128 rows:
//   waves 4-7  PRODUCERS: per offset, the 128 gathered rows go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 64 rows x
//              one 16-byte piece per instruction; "no neighbour" = out-of-range offset = hardware zero; the neighbour ids
//              are fetched one offset earlier), laid out [piece = 16 q + 4 g][row] so that the consumers' ds_read_b128 are
//              conflict free; the offset's 16 KiB weight slice follows the same way; a ballot of the ids gives the per-tile
//              "any neighbour" flags.  Two stages in flight (96 KiB of LDS), ONE barrier per offset.
//   waves 0-3  CONSUMERS: 32 rows x 64 produced channels each, nothing but ds_read_b128 + MFMA (tiles without a neighbour at
//              this offset skipped wave-uniformly), output rows stored at the end of the tile.
// Workgroups are persistent (one per CU) over the 128-row tiles, so the producers run ahead across tile boundaries.
constexpr int GW_BLOCK = 512, GW_ROWS = 128, GW_STAGES = 3;
#define GW_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int Q, int NT>
__global__ void __launch_bounds__(GW_BLOCK, 2)
gather_gemm_ws_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp, const int* __restrict__ nbr,
                      int n_out, int K, int cp, const float* __restrict__ bias, float* __restrict__ out, int n_tiles) {
    constexpr int A_F4 = Q * 4 * GW_ROWS, B_F4 = Q * NT * 64, STAGE = A_F4 + B_F4;
    constexpr int PER_STEP = 1 + 2 * Q + B_F4 / 256;     // LDS-DMA instructions a producer wave issues per step: ids, rows, weights
    __shared__ f32x4 lds[GW_STAGES * STAGE];
    __shared__ int ids[GW_STAGES][4][64];                // neighbour ids of the step, one private copy per producer wave
    __shared__ int hits[GW_STAGES][GW_ROWS / 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int total = my_tiles * K;          // (tile, offset) steps of this workgroup = barriers every wave passes
    if (total == 0) return;

    if (wave < 4) {
        // ---------------------------------------------------------------- consumers
        const int r = lane & 15, g = lane >> 4;
        float bv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = (bias && NT * r + n < cp) ? bias[NT * r + n] : 0.0f;
        int step = 0, buf = 0;
        for (int t = 0; t < my_tiles; ++t) {
            const int row0 = (blockIdx.x + t * gridDim.x) * GW_ROWS + wave * 32;
            f32x4 acc[2][NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[0][n] = acc[1][n] = f32x4{bv[n], bv[n], bv[n], bv[n]};
            for (int k = 0; k < K; ++k, ++step, buf = (buf + 1 == GW_STAGES ? 0 : buf + 1)) {
                __syncthreads();                         // stage of this step published (producers waited for its DMA)
                const f32x4* const A = lds + buf * STAGE + (wave * 2 * 16 + r) * 4 + g;      // [q][16-row block][r][g]
                const f32x4* const B = lds + buf * STAGE + A_F4 + lane;
                const bool hit0 = hits[buf][wave * 2] != 0, hit1 = hits[buf][wave * 2 + 1] != 0;     // wave-uniform
                if (!(hit0 || hit1)) continue;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const f32x4 a0 = A[q * (GW_ROWS * 4)], a1 = A[q * (GW_ROWS * 4) + 64];
                    f32x4 b[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = B[(q * NT + n) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if (hit0) acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b[n][j], acc[0][n], 0, 0, 0);
                            if (hit1) acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b[n][j], acc[1][n], 0, 0, 0);
                        }
                    }
                }
            }
            const bool full = cp == 16 * NT;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = row0 + rt * 16 + 4 * g + reg;
                    if (row >= n_out) continue;
                    float* dst = out + (size_t)row * cp + NT * r;
                    if (full) {
#pragma unroll
                        for (int n = 0; n < NT; n += 4)
                            *reinterpret_cast<f32x4*>(dst + n) = f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
                    } else {
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if (NT * r + n < cp) dst[n] = acc[rt][n][reg];
                    }
                }
            }
        }
    } else {
        // ---------------------------------------------------------------- producers
        // Every vector-memory instruction of a producer is an LDS-DMA and every step issues exactly PER_STEP of them (out of
        // range = writes zeros, touches no memory), so "the DMAs of the stage that is consumed next have landed" is the
        // counted wait vmcnt(PER_STEP): two steps of memory latency are covered instead of one.
        const int p = wave - 4;                  // owns rows 32 p .. 32 p + 31 of the tile (= consumer p's rows): 16-row blocks 2 p, 2 p + 1
        const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
        const __amdgpu_buffer_rsrc_t w_rsrc = table_rsrc(wp, (unsigned)K * B_F4 * 16u);
        const __amdgpu_buffer_rsrc_t id_rsrc = table_rsrc(reinterpret_cast<const float*>(nbr), (unsigned)((size_t)K * n_out * 4u));
        auto dma_ids = [&](int s) {              // ids of rows 32 p + lane (lanes 0..31) of step s -> ids[s % 3][p][lane]
            unsigned off = OOB;
            if (s < total && lane < 32) {
                const int t = s / K, k = s - t * K;
                const int row = (blockIdx.x + t * gridDim.x) * GW_ROWS + p * 32 + lane;
                if (row < n_out) off = (unsigned)(((size_t)k * n_out + row) * 4u);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(id_rsrc, GW_LDS_PTR(&ids[s % GW_STAGES][p][0]), 4, off, 0, 0, 0);
        };
        auto dma_stage = [&](int s) {            // rows + weights of step s; its ids have landed
            const int st = s % GW_STAGES;
            // one DMA instruction = 16 rows x the four 16-byte pieces g of channel group q: a row's 64 bytes are fetched by 4
            // adjacent lanes (whole sectors; one row per lane quadrupled the fill traffic).  LDS image [q][block][r][g].
            // ids through inline asm: hipcc drains every outstanding LDS-DMA (vmcnt(0)) in front of an LDS read it can see, which
            // would put the row gathers of the previous step back on the critical path; these words were written by the DMA
            // this wave waited for at the end of the previous step
            int id[2];
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"        // one statement: the results exist only after the wait
                         : "=&v"(id[0]), "=&v"(id[1])
                         : "v"((unsigned)(size_t)GW_LDS_PTR(&ids[st][p][lane >> 2])), "v"((unsigned)(size_t)GW_LDS_PTR(&ids[st][p][16 + (lane >> 2)]))
                         : "memory");
            const int t = s < total ? s / K : 0;
            const int row_base = (blockIdx.x + t * gridDim.x) * GW_ROWS + p * 32 + (lane >> 2);
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
                if (s >= total || row_base + 16 * b2 >= n_out) id[b2] = -1;      // past the end the LDS word holds a zero, not "no neighbour"
            f32x4* const stage = lds + st * STAGE;
            const bool any0 = __any(id[0] >= 0), any1 = __any(id[1] >= 0);
            if (lane < 2)
                asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)GW_LDS_PTR(&hits[st][2 * p + lane])), "v"((int)(lane ? any1 : any0)) : "memory");
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const unsigned row_off = (unsigned)id[b2] * (unsigned)cg * 4u + (unsigned)(lane & 3) * 16u;
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, GW_LDS_PTR(stage + ((q * 8 + 2 * p + b2) * 16) * 4), 16,
                                                             id[b2] >= 0 ? row_off + (unsigned)q * 64u : OOB, 0, 0, 0);
            }
            const int k = s < total ? s % K : 0;
#pragma unroll
            for (int it = 0; it < B_F4 / 256; ++it) {
                const int blk = p * (B_F4 / 256) + it;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, GW_LDS_PTR(stage + A_F4 + blk * 64), 16,
                                                         s < total ? ((unsigned)k * B_F4 + blk * 64 + lane) * 16u : OOB, 0, 0, 0);
            }
        };
        dma_ids(0);
        dma_ids(1);
        dma_ids(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dma_stage(0);
        dma_stage(1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PER_STEP - 1) : "memory");     // barrier 0: stage 0 landed
        for (int s = 0; s + 1 < total; ++s) {
            dma_ids(s + 3);
            dma_stage(s + 2);
            // all but the 12 row / weight DMAs just issued have landed: stage s + 1 AND the ids of step s + 3 (first DMA of this step)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PER_STEP - 1) : "memory");     // barrier s + 1
        }
    }
}

// ---- round 4: x-run operand reuse for submanifold tables ("line" kernel) -----------------------------------------------------
// The 27 offsets of a 3x3x3 submanifold stencil are 9 LINES (dz, dy) of three offsets dx = -1, 0, +1.  If the site one cell to the
// left of output row i is itself an output row j (nbr[(0,0,-1)][i] == j), then the input row that i reads through (dz, dy, -1) is the
// one j reads through (dz, dy, 0): both are the site at (z + dz, y + dy, x - 1).  Rows come in x-runs (canonical order), so j is
// usually row i - 1, i.e. LANE r - 1 of the same 16-row MFMA tile: the A fragment of offset (dz, dy, -1) is the fragment of
// (dz, dy, 0) moved one lane up inside each 16-lane row (DPP row_shr:1), and (dz, dy, +1) one lane down.  Per line the wave
// therefore gathers the centre offset's rows completely and, for the two outer offsets, only the rows whose x-neighbour is not
// the adjacent lane (run ends, tile edges) - those lanes' loads carry an out-of-range offset and cost no memory traffic.  The
// per-offset kernel gathers 3 x 16 rows per line and tile; this one 16 + the run ends (25-30 on the C3 levels).  Same operands,
// same MFMA order: bit-identical to gather_gemm_lds_kernel.  Everything else (LDS weight slices, barrier per offset, tile
// skipping, statistics epilogue, XCD chunking) is that kernel's.
__device__ __forceinline__ float dpp_row_shr1(float v) {      // lane r <- lane r - 1 within its row of 16 (lane 0 of a row: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_shl1(float v) {      // lane r <- lane r + 1 within its row of 16 (lane 15: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

template <int Q, int NT, int RT>
__global__ void __launch_bounds__(SC_BLOCK, (Q * NT <= 4) ? 4 : 3)
gather_gemm_line_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp, const int* __restrict__ nbr,
                        int n_out, int cp, const float* __restrict__ bias, float* __restrict__ out, double* __restrict__ stats) {
    constexpr int BLK = SC_BLOCK;
    constexpr int SLICE = Q * NT * 64;                    // float4 per offset
    constexpr int PER_THREAD = (SLICE + BLK - 1) / BLK;
    static_assert(SLICE % BLK == 0, "the slice is staged in whole 1 KiB wave pieces");
    __shared__ f32x4 wl[2][SLICE];
    const int lane = threadIdx.x & 63;
    const int blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int wave = blk * (BLK / 64) + (threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row0 = wave * (16 * RT);
    const f32x4* __restrict__ wp4 = reinterpret_cast<const f32x4*>(wp);

    f32x4 acc[RT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float b = 0.0f;
        if (bias && NT * r + n < cp) b = bias[NT * r + n];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][n] = f32x4{b, b, b, b};
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
    int rows[RT];
    bool live[RT], adjL[RT], adjR[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        live[rt] = row0 + rt * 16 + r < n_out;
        rows[rt] = live[rt] ? row0 + rt * 16 + r : n_out - 1;
        // x-neighbours that are the adjacent rows of this tile (offsets 12 = (0, 0, -1) and 14 = (0, 0, +1))
        const int left = nbr[(size_t)12 * n_out + rows[rt]], right = nbr[(size_t)14 * n_out + rows[rt]];
        adjL[rt] = live[rt] && r > 0 && left == rows[rt] - 1;
        adjR[rt] = live[rt] && r < 15 && right == rows[rt] + 1;
    }
#pragma unroll
    for (int t = 0; t < PER_THREAD; ++t) {
        const int e = t * BLK + threadIdx.x;
        if (e < SLICE) wl[0][e] = wp4[e];
    }
    int idn[3][RT];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) idn[d][rt] = __builtin_nontemporal_load(nbr + (size_t)d * n_out + rows[rt]);
    __syncthreads();

    // next offset's weight slice: global -> LDS by LDS-DMA (16 B per lane, lane-linear: exactly the slice's layout), no staging
    // registers - this kernel holds two operand sets (centre + run ends) where the per-offset kernel holds one
    const int wave_base = threadIdx.x & ~63;
    auto stage_dma = [&](int kn, int buf) {
#pragma unroll
        for (int t = 0; t < PER_THREAD; ++t) {
            const int e = t * BLK + threadIdx.x;
            if (SLICE % BLK == 0 || e < SLICE)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wp4 + (size_t)kn * SLICE + e),
                                                 reinterpret_cast<float*>(&wl[buf][t * BLK + wave_base]), 16, 0, 0);
        }
    };
    auto mma = [&](int buf, const f32x4 (&a)[RT][Q], const bool (&hit)[RT]) {
        // per accumulator the products arrive in the per-offset kernel's order (q, then j): same bits.  n outside j keeps one B
        // fragment live at a time; consecutive MFMAs still alternate between the two row tiles' accumulators.
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 b = wl[buf][(q * NT + n) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        if (hit[rt]) acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q][j], b[j], acc[rt][n], 0, 0, 0);
            }
        }
    };

#pragma unroll 1
    for (int line = 0; line < 9; ++line) {
        const int k0 = 3 * line;
        int s0[RT], s1[RT], s2[RT], f0[RT], f2[RT];
        bool hit0[RT], hit1[RT], hit2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            s0[rt] = live[rt] ? idn[0][rt] : -1;
            s1[rt] = live[rt] ? idn[1][rt] : -1;
            s2[rt] = live[rt] ? idn[2][rt] : -1;
            hit0[rt] = __any(s0[rt] >= 0);
            hit1[rt] = __any(s1[rt] >= 0);
            hit2[rt] = __any(s2[rt] >= 0);
            f0[rt] = adjL[rt] ? -1 : s0[rt];      // rows whose left neighbour is not the lane next door gather for themselves
            f2[rt] = adjR[rt] ? -1 : s2[rt];
        }
        f32x4 a1[RT][Q], f[RT][Q];
        // ---- dx = -1
        {
            gather_rows<Q, RT, true>(in_rsrc, cg, g, s1, a1);      // the centre offset's rows first: both outer offsets take most lanes from them
            gather_rows<Q, RT, true>(in_rsrc, cg, g, f0, f);
            stage_dma(k0 + 1, (k0 & 1) ^ 1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sh = dpp_row_shr1(a1[rt][q][j]);
                        f[rt][q][j] = adjL[rt] ? sh : f[rt][q][j];
                    }
            mma(k0 & 1, f, hit0);
            __syncthreads();
        }
        // ---- dx = 0 (the run ends of dx = +1 are requested first and arrive under this offset's matrix work)
        {
            gather_rows<Q, RT, true>(in_rsrc, cg, g, f2, f);
            stage_dma(k0 + 2, (k0 + 2) & 1);
            mma((k0 + 1) & 1, a1, hit1);
            __syncthreads();
        }
        // ---- dx = +1
        {
            if (line < 8) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) idn[d][rt] = __builtin_nontemporal_load(nbr + (size_t)(k0 + 3 + d) * n_out + rows[rt]);
                stage_dma(k0 + 3, (k0 + 3) & 1);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sh = dpp_row_shl1(a1[rt][q][j]);
                        f[rt][q][j] = adjR[rt] ? sh : f[rt][q][j];
                    }
            mma((k0 + 2) & 1, f, hit2);
            __syncthreads();
        }
    }
    if (row0 >= n_out && !stats) return;

    if (stats) {      // BatchNorm moments of the output rows (see gather_gemm_lds_kernel)
        __shared__ float st_sh[BLK / 64][2][16 * NT];
        float sm[NT], sq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            sm[n] = sq[n] = 0.0f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (row0 + rt * 16 + 4 * g + reg < n_out) {
                        const float v = acc[rt][n][reg];
                        sm[n] += v;
                        sq[n] += v * v;
                    }
            sm[n] += __shfl_xor(sm[n], 16, 64);
            sq[n] += __shfl_xor(sq[n], 16, 64);
            sm[n] += __shfl_xor(sm[n], 32, 64);
            sq[n] += __shfl_xor(sq[n], 32, 64);
            if (g == 0) {
                st_sh[threadIdx.x >> 6][0][NT * r + n] = sm[n];
                st_sh[threadIdx.x >> 6][1][NT * r + n] = sq[n];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * cp) {
            const int qq = threadIdx.x / cp, ch = threadIdx.x - qq * cp;
            double a2 = 0.0;
#pragma unroll
            for (int w = 0; w < BLK / 64; ++w) a2 += (double)st_sh[w][qq][ch];
            stats[2 * cp + (size_t)(qq * cp + ch) * gridDim.x + blk] = a2;
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + rt * 16 + 4 * g + reg;
            if (row >= n_out) continue;
            float* dst = out + (size_t)row * cp + NT * r;
            if constexpr (NT == 2) {
                *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][reg], acc[rt][1][reg]);
            } else {
#pragma unroll
                for (int n = 0; n < NT; n += 4)
                    *reinterpret_cast<f32x4*>(dst + n) = f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
            }
        }
    }
}

// ---- round 4: 128 -> 128 (VoxelResBackBone8x stride-8 level): half slices by LDS-DMA, double buffered --------------------------
// The 64 KiB weight slice of an offset does not fit twice into a workgroup's LDS share, so gather_gemm_lds_kernel<8, 8, 1, .., DB =
// false, 512> keeps ONE buffer: per offset every thread holds 128 bytes of the next slice in registers (32 VGPRs), and between
// two barriers all 8 waves stop multiplying while the slice is copied registers -> LDS.  Here the slice is cut into its two
// halves of 64 produced channels (n tiles 0-3 / 4-7 are contiguous 4 KiB pieces of the packed operand): two 32 KiB buffers, the
// NEXT half always in flight by LDS-DMA (global_load_lds, no registers) while the current one is multiplied - one barrier per
// half, no copy phase, 32 registers fewer.  A wave gathers its 16 rows once per offset and uses them for both halves; per
// accumulator the products arrive in the per-offset kernel's order: bit-identical.
template <int BLK>
__global__ void __launch_bounds__(BLK, 4)
gather_gemm_wide_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp, const int* __restrict__ nbr,
                        int n_out, int K, int cp, const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int Q = 8, NT = 8, HN = 4;                 // 128 gathered, 128 produced channels; 4 n tiles per half
    constexpr int HALF = Q * HN * 64;                    // float4 per half slice (32 KiB)
    constexpr int PER_THREAD = HALF / BLK;
    static_assert(HALF % BLK == 0 && 256 % 64 == 0, "a wave's 1 KiB DMA piece stays inside one (q, half) run of the packed operand");
    __shared__ f32x4 wl[2][HALF];
    const int lane = threadIdx.x & 63;
    const int blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int wave = blk * (BLK / 64) + (threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row0 = wave * 16;
    const f32x4* __restrict__ wp4 = reinterpret_cast<const f32x4*>(wp);
    const int wave_base = threadIdx.x & ~63;

    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float b = 0.0f;
        if (bias && NT * r + n < cp) b = bias[NT * r + n];
        acc[n] = f32x4{b, b, b, b};
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
    const bool live = row0 + r < n_out;
    const int row = live ? row0 + r : n_out - 1;

    // half h of offset k: for every q the 256 float4 at ((k * Q + q) * NT + 4 h) * 64 of the packed operand -> wl[buf][q * 256 ..]
    auto dma_half = [&](int k, int h, int buf) {
#pragma unroll
        for (int t = 0; t < PER_THREAD; ++t) {
            const int e = t * BLK + threadIdx.x;          // 0 .. HALF: q = e / 256
            const size_t src = ((size_t)(k * Q + (e >> 8)) * NT + HN * h) * 64 + (e & 255);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wp4 + src), reinterpret_cast<float*>(&wl[buf][t * BLK + wave_base]), 16,
                                             0, 0);
        }
    };
    auto mma_half = [&](int buf, int h, const f32x4 (&a)[Q]) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            f32x4 b[HN];
#pragma unroll
            for (int n = 0; n < HN; ++n) b[n] = wl[buf][(q * HN + n) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < HN; ++n)
                    acc[HN * h + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[n][j], acc[HN * h + n], 0, 0, 0);
        }
    };

    dma_half(0, 0, 0);
    int id_next = __builtin_nontemporal_load(nbr + row);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const int src = live ? id_next : -1;
        const int kn = k + 1 < K ? k + 1 : K - 1;
        id_next = __builtin_nontemporal_load(nbr + (size_t)kn * n_out + row);
        const bool hit = __any(src >= 0);
        f32x4 a[Q];
        {
            const int s1[1] = {src};
            f32x4 a1[1][Q];
            gather_rows<Q, 1, true>(in_rsrc, cg, g, s1, a1);
#pragma unroll
            for (int q = 0; q < Q; ++q) a[q] = a1[0][q];
        }
        dma_half(k, 1, 1);                    // behind the gathers in issue order: the wait for the rows leaves it in flight
        if (hit) mma_half(0, 0, a);
        __syncthreads();                      // half 1 has landed; everybody is done with half 0
        if (k + 1 < K) dma_half(k + 1, 0, 0);
        if (hit) mma_half(1, 1, a);
        __syncthreads();
    }
    if (row0 >= n_out) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int orow = row0 + 4 * g + reg;
        if (orow >= n_out) continue;
        float* dst = out + (size_t)orow * cp + NT * r;
#pragma unroll
        for (int n = 0; n < NT; n += 4) *reinterpret_cast<f32x4*>(dst + n) = f32x4{acc[n][reg], acc[n + 1][reg], acc[n + 2][reg], acc[n + 3][reg]};
    }
}

// Staged variant of gather_gemm_lds_kernel for the 64 -> 64 layers: the weight slices of KS consecutive offsets share one LDS
// stage and the waves of a workgroup meet only at the stage boundaries (two barriers per KS offsets instead of one per offset).
// Why: with a barrier per offset every wave waits for the busiest wave OF THAT OFFSET.  A wave's 32 rows have a neighbour at a
// given offset in 0, 1 or 2 of its two 16-row tiles (72 % of the (tile, offset) pairs are non-empty on the stride-4 level), so
// the expected maximum over the 4 waves of a block is ~1.35x the mean - exactly the 70 % matrix-pipe utilisation the PMC pass
// shows.  Over 3 offsets the waves' sums differ much less.  512-thread workgroups (8 waves share a stage, 2 workgroups = 16
// waves per CU as before), 48 KiB of LDS per workgroup.
template <int Q, int NT, int KS, int BLK>
__global__ void __launch_bounds__(BLK, 4)
gather_gemm_stage_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp, const int* __restrict__ nbr,
                         int n_out, int K, int cp, const float* __restrict__ bias, float* __restrict__ out, double* __restrict__ stats) {
    constexpr int RT = 2;
    constexpr int SLICE = Q * NT * 64;                    // float4 per offset
    constexpr int STG = KS * SLICE;
    constexpr int PER_THREAD = STG / BLK;
    static_assert(STG % BLK == 0, "stage must divide over the workgroup");
    __shared__ f32x4 wl[STG];
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row0 = wave * (16 * RT);
    const f32x4* __restrict__ wp4 = reinterpret_cast<const f32x4*>(wp);

    f32x4 acc[RT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float b = 0.0f;
        if (bias && NT * r + n < cp) b = bias[NT * r + n];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][n] = f32x4{b, b, b, b};
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
    int rows[RT];
    bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        live[rt] = row0 + rt * 16 + r < n_out;
        rows[rt] = live[rt] ? row0 + rt * 16 + r : n_out - 1;
    }
    const int total_f4 = K * SLICE;
    for (int k0 = 0; k0 < K; k0 += KS) {
        f32x4 stage[PER_THREAD];
#pragma unroll
        for (int t = 0; t < PER_THREAD; ++t) {
            const int e = k0 * SLICE + t * BLK + threadIdx.x;
            stage[t] = e < total_f4 ? wp4[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();          // every wave has finished the previous stage
#pragma unroll
        for (int t = 0; t < PER_THREAD; ++t) wl[t * BLK + threadIdx.x] = stage[t];
        __syncthreads();
#pragma unroll 1
        for (int kk = 0; kk < KS; ++kk) {
            const int k = k0 + kk;
            if (k >= K) break;
            int src[RT];
            bool hit[RT];
            bool any = false;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int v = nbr[(size_t)k * n_out + rows[rt]];
                src[rt] = live[rt] ? v : -1;
                hit[rt] = __any(src[rt] >= 0);
                any = any || hit[rt];
            }
            if (!any) continue;
            f32x4 a[RT][Q];
            gather_rows<Q, RT, true>(in_rsrc, cg, g, src, a);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                f32x4 b[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n] = wl[kk * SLICE + (q * NT + n) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            if (hit[rt]) acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q][j], b[n][j], acc[rt][n], 0, 0, 0);
            }
        }
    }

    if (stats) {      // BatchNorm moments of the output rows (see gather_gemm_lds_kernel)
        __shared__ float st_sh[BLK / 64][2][16 * NT];
        float sm[NT], sq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            sm[n] = sq[n] = 0.0f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (row0 + rt * 16 + 4 * g + reg < n_out) {
                        const float v = acc[rt][n][reg];
                        sm[n] += v;
                        sq[n] += v * v;
                    }
            sm[n] += __shfl_xor(sm[n], 16, 64);
            sq[n] += __shfl_xor(sq[n], 16, 64);
            sm[n] += __shfl_xor(sm[n], 32, 64);
            sq[n] += __shfl_xor(sq[n], 32, 64);
            if (g == 0) {
                st_sh[threadIdx.x >> 6][0][NT * r + n] = sm[n];
                st_sh[threadIdx.x >> 6][1][NT * r + n] = sq[n];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * cp) {
            const int qq = threadIdx.x / cp, ch = threadIdx.x - qq * cp;
            double a2 = 0.0;
#pragma unroll
            for (int w = 0; w < BLK / 64; ++w) a2 += (double)st_sh[w][qq][ch];
            stats[2 * cp + (size_t)(qq * cp + ch) * gridDim.x + blockIdx.x] = a2;
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + rt * 16 + 4 * g + reg;
            if (row >= n_out) continue;
            float* dst = out + (size_t)row * cp + NT * r;
#pragma unroll
            for (int n = 0; n < NT; n += 4)
                *reinterpret_cast<f32x4*>(dst + n) = f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
        }
    }
}

// ---- LDS-staged halo tiles (submanifold layers) ----------------------------------------------------------------------------
// Every variant above re-gathers a block's 256-byte input rows from L1 / L2 once PER OFFSET (27 times per row block; the waves
// wait on vector memory in half of their cycles, ~750 cycles per gather instruction - profiles/r02_pmc_stall_gather_gemm.json).
// Here a workgroup owns R = 128 output rows that are compact in space (Morton order, halo_plan.hip): the union of their 27
// neighbour sets is 220 rows on average on the C3 stride-4 level (128 x 16 pair references), so the block's unique input rows are
// loaded ONCE into LDS (<= UMAX rows; rows swizzled by their local id so that the fragment reads spread over the banks) and all 27
// offsets' A fragments come from ds_read_b128 through 16-bit local ids - the offset loop issues no vector-memory instruction
// except the next weight slice.  8 waves x 16 rows; weight slices in a 3-deep LDS ring (the slice of offset k + 2 is fetched
// into registers at the start of offset k and written after its MFMAs: the slice a wave reads first after a barrier has been
// in LDS for a whole step); local ids two offsets ahead, A fragments one offset ahead (two register sets, loop unrolled by two:
// no copies).  A (tile, offset) with a neighbour that did not fit (HALO_SPILL) takes that offset from the global table exactly
// as gather_gemm_lds_kernel does: same operands, same MFMA order, same bits.
constexpr unsigned short HALO_NONE = 0xFFFFu, HALO_SPILL = 0xFFFEu;
struct HaloGeomK {
    int R, UMAX;
};
static inline bool halo_geom_k(int c_gather, HaloGeomK* g) {      // keep in sync with halo_plan.hip
    if (c_gather == 64) {
        *g = HaloGeomK{128, 320};
        return true;
    }
    if (c_gather == 32) {
        *g = HaloGeomK{128, 320};
        return true;
    }
    return false;
}

// Two measured dead ends shaped this kernel (389.5 k-row 64 -> 64 level, per-offset kernel 0.58-0.60 ms):
//   * whole 256-byte rows in LDS + the weight slices in an LDS ring with a barrier per offset = ONE 512-thread workgroup per CU:
//     0.75 ms.  A wave that waits at a barrier has no other workgroup's waves to give its SIMD to.
//   * the same without barriers, every wave streaming its B fragments from L1 / L2: 0.83-0.89 ms - 16 KiB of weights per
//     (16-row tile, offset) is 7 GB per launch through the CUs' 64 B/clk L1 path.
// So: weights stay in LDS, and the workgroup is made small enough for TWO per CU by staging HALF rows.  The gathered channels are
// worked off in passes of QP = 2 channel groups (32 channels = 128 bytes per row): pass p stages channels [32 p, 32 p + 32) of
// the block's unique rows (41 KiB), walks the K offsets with the matching half of each weight slice (8 KiB, 3-deep ring) and
// leaves its sums in the accumulators; the next pass refills the rows with the other half.  77 KiB per workgroup, two
// workgroups = 16 waves per CU with independent barriers.
template <int Q, int NT, int QP, int R, int UMAX, int KMAX>
__global__ void __launch_bounds__(R * 4, 4)
gather_gemm_halo_kernel(const float* __restrict__ in, int n, int cg, const float* __restrict__ wp, const int* __restrict__ nbr, int K, int cp,
                        const float* __restrict__ bias, float* __restrict__ out, const int* __restrict__ order_all,
                        const int* __restrict__ urows_all, const unsigned short* __restrict__ lids_all, double* __restrict__ stats, const int ablate) {
    constexpr int WAVES = R / 16, BLK = WAVES * 64;
    constexpr int PASSES = Q / QP;
    constexpr int ROW4 = QP * 4;                        // float4 per staged (partial) row
    constexpr int SLICE = QP * NT * 64;                 // float4 of weights per (offset, pass)
    constexpr int W_PER = (SLICE + BLK - 1) / BLK;
    constexpr int ROWS_PER_INSTR = 64 / ROW4;           // rows one wave instruction moves (16-byte pieces)
    constexpr int FILL_ITERS = UMAX / (WAVES * ROWS_PER_INSTR);
    static_assert(Q % QP == 0 && UMAX % (WAVES * ROWS_PER_INSTR) == 0, "UMAX must be a whole number of fill rounds");
    static_assert((KMAX * R) % 8 == 0, "local ids are copied 16 bytes at a time");
    __shared__ f32x4 halo[(UMAX + 1) * ROW4];           // row UMAX = zeros ("no neighbour")
    __shared__ f32x4 wl[3][SLICE];
    __shared__ __attribute__((aligned(16))) unsigned short lid_s[KMAX * R];
    __shared__ int s_spill;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_spill = 0;
    __syncthreads();
    const int r = lane & 15, g = lane >> 4;
    const int blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const f32x4* __restrict__ wp4 = reinterpret_cast<const f32x4*>(wp);
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n * (unsigned)cg * 4u);
    const int total = PASSES * K;                       // (pass, offset) steps = barriers every wave passes
    const int* __restrict__ urows = urows_all + (size_t)blk * UMAX;
    bool saw_spill = false;
    auto fill = [&](int p) {
        const int sub = lane / ROW4, c = lane % ROW4;
        f32x4 v[FILL_ITERS];
        int jj[FILL_ITERS];
#pragma unroll
        for (int it = 0; it < FILL_ITERS; ++it) {
            const int j = (it * WAVES + wave) * ROWS_PER_INSTR + sub;
            jj[it] = j;
            const int u = urows[j];
            v[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  in_rsrc, u >= 0 ? (unsigned)u * (unsigned)cg * 4u + (unsigned)(p * ROW4 + c) * 16u : OOB, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < FILL_ITERS; ++it) halo[jj[it] * ROW4 + (c ^ (jj[it] & (ROW4 - 1)))] = v[it];
    };
    {
        // (integer vectors: with the ids moved as float4 and inspected through bit casts hipcc tested only the first dword)
        const u32x4* __restrict__ l4 = reinterpret_cast<const u32x4*>(lids_all + (size_t)blk * K * R);
        u32x4* ls4 = reinterpret_cast<u32x4*>(lid_s);
        for (int e = tid; e < K * R / 8; e += BLK) {
            const u32x4 v = l4[e];
            ls4[e] = v;
            const unsigned lo_hit = (unsigned)((v.x & 0xFFFFu) == HALO_SPILL) | (unsigned)((v.y & 0xFFFFu) == HALO_SPILL) |
                                    (unsigned)((v.z & 0xFFFFu) == HALO_SPILL) | (unsigned)((v.w & 0xFFFFu) == HALO_SPILL);
            const unsigned hi_hit = (unsigned)((v.x >> 16) == HALO_SPILL) | (unsigned)((v.y >> 16) == HALO_SPILL) |
                                    (unsigned)((v.z >> 16) == HALO_SPILL) | (unsigned)((v.w >> 16) == HALO_SPILL);
            saw_spill = saw_spill || (lo_hit | hi_hit) != 0u;
        }
        if (tid < ROW4) halo[UMAX * ROW4 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int myrow = order_all[(size_t)blk * R + 16 * wave + r];        // -1 on the padding of the last block

    f32x4 acc[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        float b = 0.0f;
        if (bias && NT * r + nn < cp) b = bias[NT * r + nn];
        acc[nn] = f32x4{b, b, b, b};
    }
    const unsigned short* my_lids = lid_s + 16 * wave + r;
    auto lid_at = [&](int k) -> unsigned { return myrow < 0 ? (unsigned)HALO_NONE : (unsigned)my_lids[(k < K ? k : K - 1) * R]; };
    // A block with a neighbour that did not fit (more than UMAX unique rows: dense, deep lattices - 57 ids of 10.5 M on the C3
    // stride-4 level) is worked off from the global table and the global weights, in the SAME (pass, offset, channel group) order,
    // so a row's bits do not depend on which path its block took.  A separate loop on purpose: merged into the LDS loop, the
    // buffer loads of this path and the LDS reads of that one share destination registers and hipcc then waits for every
    // outstanding vector-memory load (the weight slice in flight) in front of each fragment read.
    if (__any(saw_spill) && lane == 0) s_spill = 1;      // (benign race: every writer stores the same value)
    __syncthreads();                                      // publishes the flag and the local ids
    const bool block_spills = s_spill != 0 || (ablate & 32);      // (ablate: timing experiments only)
    if (block_spills) {
        for (int p = 0; p < PASSES; ++p)
            for (int k = 0; k < K; ++k) {
                const int src = myrow >= 0 ? nbr[(size_t)k * n + myrow] : -1;
                if (!__any(src >= 0)) continue;
                f32x4 a[QP];
#pragma unroll
                for (int q = 0; q < QP; ++q)
                    a[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         in_rsrc, src >= 0 ? (unsigned)src * (unsigned)cg * 4u + (unsigned)(16 * (p * QP + q) + 4 * g) * 4u : OOB, 0, 0));
                const f32x4* __restrict__ wk = wp4 + ((size_t)k * Q + (size_t)p * QP) * NT * 64 + lane;
#pragma unroll
                for (int q = 0; q < QP; ++q) {
                    f32x4 b[NT];
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) b[nn] = wk[(q * NT + nn) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[nn][j], acc[nn], 0, 0, 0);
                }
            }
    } else {
        // Weight ring, 3 LDS buffers + 2 register stages: at the START of step s (every wave is past the barrier of step s - 1, so
        // the buffer step s - 1 read is free) the slice of step s + 2 goes from its register stage to LDS and the slice of step
        // s + 4 is requested into that stage - two whole steps of cover for the L2 round trip (requested and written inside one
        // step it was exposed: -0.06 ms in the ablation).  Stages alternate with the step parity (K odd: their contents are swapped between
        // passes).  No division or modulo in the loop: cursors.
        int buf = 0, s = 0;                                 // buffer holding the slice of step s
        const size_t k_stride = (size_t)Q * NT * 64;
        int ld_k = 0, ld_p = 0;                             // (offset, pass) of the slice requested next
        const f32x4* ld_ptr = wp4;
        auto ld_advance = [&]() {
            if (ld_k + 1 < K) {
                ++ld_k;
                ld_ptr += k_stride;
            } else if (ld_p + 1 < PASSES) {
                ld_k = 0;
                ++ld_p;
                ld_ptr = wp4 + (size_t)ld_p * QP * NT * 64;
            }                                               // past the last slice: keep re-requesting it (never written)
        };
        f32x4 stA[W_PER], stB[W_PER];
        {   // slices 0 and 1 straight to LDS, 2 and 3 into the stages
#pragma unroll
            for (int t = 0; t < W_PER; ++t) {
                const int e = t * BLK + tid;
                if (SLICE % BLK == 0 || e < SLICE) wl[0][e] = ld_ptr[e];
            }
            ld_advance();
#pragma unroll
            for (int t = 0; t < W_PER; ++t) {
                const int e = t * BLK + tid;
                if (SLICE % BLK == 0 || e < SLICE) wl[1][e] = ld_ptr[e];
            }
            ld_advance();
#pragma unroll
            for (int t = 0; t < W_PER; ++t) {
                const int e = t * BLK + tid;
                if (SLICE % BLK == 0 || e < SLICE) stA[t] = ld_ptr[e];
            }
            ld_advance();
#pragma unroll
            for (int t = 0; t < W_PER; ++t) {
                const int e = t * BLK + tid;
                if (SLICE % BLK == 0 || e < SLICE) stB[t] = ld_ptr[e];
            }
            ld_advance();
        }
        fill(0);
        __syncthreads();
        auto load_a = [&](unsigned lid, f32x4 (&a)[QP]) {
            const unsigned row = lid == HALO_NONE ? (unsigned)UMAX : lid;
            const f32x4* base = halo + row * ROW4;
            const unsigned sw = row & (ROW4 - 1);
#pragma unroll
            for (int q = 0; q < QP; ++q) a[q] = base[(unsigned)(4 * q + g) ^ sw];
        };
        // one step: cur = operands of offset k (already loaded), nxt = the set the fragments of offset k + 1 are loaded into,
        // st = the register stage of this step's parity
        // B fragments of the NEXT step's first channel group are read before the barrier that ends a step (their slice has been in
        // LDS since the step before), so that the MFMAs behind a barrier start at once instead of behind an LDS round trip that
        // all waves of the workgroup would take together.
        f32x4 b_first[NT];
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) b_first[nn] = wl[0][nn * 64 + lane];
        auto step = [&](bool more, unsigned lid_cur, const f32x4 (&a_cur)[QP], unsigned lid_nxt, f32x4 (&a_nxt)[QP], f32x4 (&st)[W_PER]) {
            const bool hit = __any(lid_cur != HALO_NONE);      // wave-uniform: some row of the tile has a neighbour at this offset
            f32x4 b[NT];
            if (hit) {
                if (QP > 1) {
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) b[nn] = wl[buf][(NT + nn) * 64 + lane];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0][j], b_first[nn][j], acc[nn], 0, 0, 0);
            }
            {
                const int wb = buf == 0 ? 2 : buf - 1;      // buffer of step s + 2 = the one step s - 1 read
                if (s + 2 < total) {
#pragma unroll
                    for (int t = 0; t < W_PER; ++t) {
                        const int e = t * BLK + tid;
                        if (SLICE % BLK == 0 || e < SLICE) wl[wb][e] = st[t];
                    }
                }
#pragma unroll
                for (int t = 0; t < W_PER; ++t) {
                    const int e = t * BLK + tid;
                    if (SLICE % BLK == 0 || e < SLICE) st[t] = ld_ptr[e];
                }
                ld_advance();
            }
            if (more) load_a(lid_nxt, a_nxt);
            if (hit) {
#pragma unroll
                for (int q = 1; q < QP; ++q) {
                    if (q > 1) {
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) b[nn] = wl[buf][(q * NT + nn) * 64 + lane];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[q][j], b[nn][j], acc[nn], 0, 0, 0);
                }
            }
            buf = buf == 2 ? 0 : buf + 1;
            ++s;
            if (s < total) {
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) b_first[nn] = wl[buf][nn * 64 + lane];
            }
            __syncthreads();
        };
        auto run_pass = [&](int p, f32x4 (&stE)[W_PER], f32x4 (&stO)[W_PER]) {      // stE / stO: stages of the even / odd offsets of this pass
            f32x4 aA[QP], aB[QP];
            unsigned l0 = lid_at(0), l1 = lid_at(1), l2, l3;
            load_a(l0, aA);
            int k = 0;
            for (; k + 3 < K; k += 2) {            // all but the last one or two offsets
                l2 = lid_at(k + 2);
                step(true, l0, aA, l1, aB, stE);
                l3 = lid_at(k + 3);
                step(true, l1, aB, l2, aA, stO);
                l0 = l2;
                l1 = l3;
            }
            if (k + 2 < K) {                        // three offsets left (K odd)
                l2 = lid_at(k + 2);
                step(true, l0, aA, l1, aB, stE);
                step(true, l1, aB, l2, aA, stO);
                step(false, l2, aA, l2, aB, stE);
            } else if (k + 1 < K) {                 // two left (K even)
                step(true, l0, aA, l1, aB, stE);
                step(false, l1, aB, l1, aA, stO);
            } else {
                step(false, l0, aA, l0, aB, stE);
            }
            if (p + 1 < PASSES) {                   // (every wave is past the barrier of the pass's last offset: nobody reads the rows)
                fill(p + 1);
                __syncthreads();
            }
        };
        for (int p = 0; p < PASSES; ++p) {
            run_pass(p, stA, stB);
            if ((K & 1) && p + 1 < PASSES) {       // odd K: the next pass starts on the other parity - swap the stages' contents
#pragma unroll
                for (int t = 0; t < W_PER; ++t) {
                    const f32x4 tmp = stA[t];
                    stA[t] = stB[t];
                    stB[t] = tmp;
                }
            }
        }
    }

    // BatchNorm moments of the block's rows (see gather_gemm_lds_kernel): partial sums [2 cp][gridDim.x] behind the results
    if (stats) {
        __shared__ float st_sh[WAVES][2][16 * NT];
        float sm[NT], sq[NT];
        int prow[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) prow[reg] = __shfl(myrow, 4 * g + reg, 64);
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            sm[nn] = sq[nn] = 0.0f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (prow[reg] >= 0) {
                    const float v = acc[nn][reg];
                    sm[nn] += v;
                    sq[nn] += v * v;
                }
            sm[nn] += __shfl_xor(sm[nn], 16, 64);
            sq[nn] += __shfl_xor(sq[nn], 16, 64);
            sm[nn] += __shfl_xor(sm[nn], 32, 64);
            sq[nn] += __shfl_xor(sq[nn], 32, 64);
            if (g == 0) {
                st_sh[wave][0][NT * r + nn] = sm[nn];
                st_sh[wave][1][NT * r + nn] = sq[nn];
            }
        }
        __syncthreads();
        if (tid < 2 * cp) {
            const int qq = tid / cp, ch = tid - qq * cp;
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) a += (double)st_sh[w][qq][ch];
            stats[2 * cp + (size_t)(qq * cp + ch) * gridDim.x + blk] = a;
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = __shfl(myrow, 4 * g + reg, 64);
        if (row < 0) continue;
        float* dst = out + (size_t)row * cp + NT * r;
        if constexpr (NT == 2) {
            *reinterpret_cast<float2*>(dst) = make_float2(acc[0][reg], acc[1][reg]);
        } else {
#pragma unroll
            for (int nn = 0; nn < NT; nn += 4)
                *reinterpret_cast<f32x4*>(dst + nn) = f32x4{acc[nn][reg], acc[nn + 1][reg], acc[nn + 2][reg], acc[nn + 3][reg]};
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad, dout-stationary form for the wide K = 27 layers (32 -> 32, 64 -> 64, 32 -> 64).  In wgrad_kernel a
// workgroup is one (row chunk, offset): the 27 offset-blocks of a chunk each fetch the chunk's dout rows again, and every
// (in, out) pair costs one row of each table through L2 - (cin + cout) * 4 bytes for 2 * cin * cout FLOP, 8 (32 -> 32) or 16
// (64 -> 64) FLOP per L2 byte: PMC 1.03 GB from HBM / 2.4 GB through L2 per launch of the 32 -> 32 layer at 682 k rows against
// 0.25 GB algorithmic, MFMA pipe 50 % busy, 70 % of the wave cycles in s_waitcnt.  Here a workgroup owns whole 128-row tiles for
// ALL 27 offsets: the tile's dout rows are staged in LDS once (one contiguous 16 / 32 KiB copy) and serve every pair of the
// tile as the B operand; only the gathered input rows still come through L2, half of the former traffic.  8 waves, wave w
// owns three or four fixed offsets (WT_OFF: dealt by pair density so that the waves of a tile finish together) and keeps their
// MTB x NTB accumulator tiles for the whole launch.  Per tile a wave first compacts the valid pairs of all its offsets into
// per-(wave, offset) LDS queues (ballot + prefix popcount), then runs the full 16-pair rounds of all its queues as ONE stream in
// which the input-row gathers of round r + 1 are issued before the MFMAs of round r (in wgrad_kernel a round is load -> wait ->
// multiply, and only other waves cover the wait).  A queue's tail (< 16 pairs) is carried into the next tile - its B rows stay valid
// because the tiles rotate through THREE LDS buffers (the one being filled for t + 1, the current one, the previous one); a
// tail that has seen no full round for a whole tile is flushed as a partial round before its buffer can be refilled.  64 input
// channels are split over two workgroups (contiguous 128-byte half rows each): 4 offsets x 2 x 4 tiles = 128 accumulator registers.
// Tiles are dealt so that each XCD walks one contiguous row range with all its workgroups side by side (neighbouring tiles
// gather overlapping input rows: they meet in that XCD's L2).  Slabs + wgrad_reduce_kernel as before: deterministic.
// ------------------------------------------------------------------------------------------------------------------
#ifndef TODA_WT_DEPTH
#define TODA_WT_DEPTH 2
#endif
constexpr int WT_DEPTH = TODA_WT_DEPTH;      // rounds of input-row gathers in flight per wave
constexpr int WT_WAVES = 8, WT_SLOTS = 4, WT_BLOCK = WT_WAVES * 64, WT_R = 128, WT_QCAP = WT_R + 16, WT_K = 27;
// offsets of a wave (-1: empty slot): longest-processing-time deal of the per-offset pair counts of the C3 levels (centre 1.0;
// level 2: dz = 0 ring 0.5-0.6, dz = +-1 0.3-0.5; levels 3, 4: dz = 0 ring 0.85, dz = +-1 0.45): heaviest wave 1.10x the mean
__device__ __constant__ signed char WT_OFF[2][WT_WAVES][WT_SLOTS] = {
    {{13, 0, 2, -1}, {10, 1, 3, 6}, {12, 5, 7, 8}, {14, 19, 21, 18}, {16, 23, 25, -1}, {4, 9, 20, -1}, {22, 11, 24, -1}, {15, 17, 26, -1}},
    {{13, 19, 8, -1}, {10, 1, 21, 24}, {12, 3, 23, 26}, {14, 5, 25, -1}, {16, 7, 0, 20}, {9, 17, 18, -1}, {11, 4, 2, -1}, {15, 22, 6, -1}}};

template <int MTB, int NTB>
__global__ void __launch_bounds__(WT_BLOCK, NTB <= 2 ? 2 : 1)
wgrad_tile_kernel(const float* __restrict__ in, int n_in, int cin, const float* __restrict__ dout, const int* __restrict__ nbr,
                  int n_out, int n_tiles, int G, int nsub, int profile, float* __restrict__ slab) {
    constexpr int CO = 16 * NTB, TILE_F = WT_R * CO;
    constexpr int FILL = (TILE_F / 4 + WT_BLOCK - 1) / WT_BLOCK;     // 16-byte pieces per thread and tile
    __shared__ float tile[3 * TILE_F];
    // queue entry: (input row << 9) | row of the 3 x 128-row ring (n_in < 2^23, checked by the host)
    __shared__ unsigned q_all[WT_WAVES][WT_SLOTS][WT_QCAP];
    __shared__ unsigned rl[WT_WAVES][64];     // padded round stream of the wave: index of the round's first queue entry in q_all[wave], ~0: empty round
    static_assert(MTB == 2, "8-byte half-row gathers");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ii = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int sub = l % nsub, wl = l / nsub, gx = G >> 3;
    const int gid = xcd * gx + wl;                                   // slab of this workgroup (both channel halves share it)
    const int t8 = (n_tiles + 7) >> 3;
    const int t_end = min((xcd + 1) * t8, n_tiles);
    const int ci_base = sub * 16 * MTB;
    int koff[WT_SLOTS];
#pragma unroll
    for (int j = 0; j < WT_SLOTS; ++j) koff[j] = __builtin_amdgcn_readfirstlane((int)WT_OFF[profile][wv][j]);

    f32x4 acc[WT_SLOTS][MTB][NTB];
#pragma unroll
    for (int j = 0; j < WT_SLOTS; ++j)
#pragma unroll
        for (int m = 0; m < MTB; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n) acc[j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int qn[WT_SLOTS] = {0, 0, 0, 0};

    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cin * 4u);
    const __amdgpu_buffer_rsrc_t dout_rsrc = table_rsrc(dout, (unsigned)n_out * (unsigned)CO * 4u);
    const __amdgpu_buffer_rsrc_t id_rsrc = table_rsrc(reinterpret_cast<const float*>(nbr), (unsigned)((size_t)WT_K * n_out * 4u));
    // the same descriptor as in_rsrc, as four scalar words for the inline-asm loads of the round stream (GFX9 layout: base[47:0],
    // stride 0, num_records in bytes, word 3 as table_rsrc)
    const unsigned long long in_addr = (unsigned long long)in;
    const u32x4 in_desc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)in_addr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(in_addr >> 32) & 0xFFFFu)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)n_in * (unsigned)cin * 4u)), 0x00020000u};

    // A operands of one round: queue entries [d, d + 16), 4 pairs per MFMA step; entries >= limit: zeros
    auto gather = [&](const unsigned* q, int d, int limit, float (&a)[4][MTB]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = d + 4 * t + g;
            const bool ok = p < limit;
            const unsigned e = q[ok ? p : d];
            const unsigned ia = ((e >> 9) * (unsigned)cin + (unsigned)(ci_base + MTB * ii)) * 4u;
            if constexpr (MTB == 2) {
                const f32x2w v = __builtin_bit_cast(f32x2w, __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, ok ? ia : OOB, 0, 0));
                a[t][0] = v[0], a[t][1] = v[1];
            } else {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? ia : OOB, 0, 0));
#pragma unroll
                for (int m = 0; m < 4; ++m) a[t][m] = v[m];
            }
        }
    };
    // B operands from the LDS ring + the MFMAs of the round
    auto multiply = [&](const unsigned* q, int d, int limit, const float (&a)[4][MTB], f32x4 (&ac)[MTB][NTB]) {
        float b[4][NTB];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = d + 4 * t + g;
            const bool ok = p < limit;
            const float* bp = tile + (q[ok ? p : d] & 511u) * CO + NTB * ii;
            if constexpr (NTB == 2) {
                const f32x2w v = *reinterpret_cast<const f32x2w*>(bp);
                b[t][0] = ok ? v[0] : 0.f, b[t][1] = ok ? v[1] : 0.f;
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
                for (int n = 0; n < 4; ++n) b[t][n] = ok ? v[n] : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MTB; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n) ac[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][m], b[t][n], ac[m][n], 0, 0, 0);
    };

    // dout rows of tile t -> registers (out-of-range pieces of the last tile: zeros) / registers -> LDS buffer
    f32x4 stage[FILL];
    auto fetch_tile = [&](int t) {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int piece = tid + f * WT_BLOCK;
            const unsigned off = ((unsigned)t * (unsigned)TILE_F + (unsigned)piece * 4u) * 4u;
            stage[f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dout_rsrc, (piece < TILE_F / 4 && t < t_end) ? off : OOB, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int piece = tid + f * WT_BLOCK;
            if (piece < TILE_F / 4) *reinterpret_cast<f32x4*>(tile + buf * TILE_F + piece * 4) = stage[f];
        }
    };
    // ids of tile t for this wave's offsets, two batches of 64 rows each (-1: no neighbour / past the end / empty slot)
    int ids[WT_SLOTS][2];
    auto fetch_ids = [&](int t) {
#pragma unroll
        for (int j = 0; j < WT_SLOTS; ++j)
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const int o = t * WT_R + bt * 64 + lane;
                const bool live = koff[j] >= 0 && t < t_end && o < n_out;
                const int v = __builtin_bit_cast(int, __builtin_amdgcn_raw_buffer_load_b32(
                    id_rsrc, live ? (unsigned)(((size_t)koff[j] * n_out + o) * 4u) : OOB, 0, 0));
                ids[j][bt] = live ? v : -1;
            }
    };

    int t = xcd * t8 + wl;
    if (t < t_end) {
        fetch_tile(t);
        fetch_ids(t);
        store_tile(0);
    }
    int buf = 0;
    for (; t < t_end; t += gx) {
        __syncthreads();                 // buffer `buf` complete; every wave has left the previous tile (its carry-overs included)
        // 1. compact this tile's pairs behind the carried tails
        int carried[WT_SLOTS];
#pragma unroll
        for (int j = 0; j < WT_SLOTS; ++j) {
            carried[j] = qn[j];
            unsigned* q = q_all[wv][j];
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const int i = ids[j][bt];
                const unsigned long long vote = __ballot(i >= 0);
                if (i >= 0) q[qn[j] + __popcll(vote & ((1ull << lane) - 1))] = ((unsigned)i << 9) | (unsigned)(buf * WT_R + bt * 64 + lane);
                qn[j] += __popcll(vote);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int tn = t + gx;
        constexpr bool EARLY = false;     // the next tile's dout rows and ids in flight under this tile's rounds: 16 + 8 registers that
                                          // the 64-channel form (128 accumulators + the ring) does not have
        if constexpr (EARLY) {
            fetch_tile(tn);
            fetch_ids(tn);
        }
        // 2. the full 16-pair rounds of all slots as ONE stream r = 0 .. r_tot - 1 (slot-major).  The input-row gathers of round
        //    r + WT_DEPTH are issued when round r has been multiplied, WT_DEPTH - 1 rounds of loads stay in flight behind the one a
        //    round waits for (vmcnt is counted by hand: the loads are inline asm, hipcc does not see them and cannot merge their
        //    waits with anything; past the end of the stream the same four loads go to an out-of-range offset - no memory access -
        //    so that the count is the same on every path).  Ring positions are static (the stream is walked WT_DEPTH rounds per
        //    loop iteration), the slot of a round is a wave-uniform switch around the MFMAs.
        //    Each slot's rounds are padded to a multiple of WT_DEPTH with empty rounds (dummy loads, no MFMAs), so a round's ring
        //    position is its index modulo WT_DEPTH in every slot and both the ring position and the accumulator set are static.
        int rj[WT_SLOTS], base[WT_SLOTS + 1];
        base[0] = 0;
#pragma unroll
        for (int j = 0; j < WT_SLOTS; ++j) {
            rj[j] = qn[j] >> 4;
            base[j + 1] = base[j] + (rj[j] + WT_DEPTH - 1) / WT_DEPTH * WT_DEPTH;
        }
        const int s_tot = base[WT_SLOTS];            // padded stream length (<= 4 x 12)
        {
            int jj = 0;
#pragma unroll
            for (int j = 1; j < WT_SLOTS; ++j)
                if (lane >= base[j]) jj = j;
            int idx = lane - base[0];
#pragma unroll
            for (int j = 1; j < WT_SLOTS; ++j)
                if (jj == j) idx = lane - base[j];
            int rjj = rj[0];
#pragma unroll
            for (int j = 1; j < WT_SLOTS; ++j)
                if (jj == j) rjj = rj[j];
            rl[wv][lane] = (lane < s_tot && idx < rjj) ? ((unsigned)(jj * WT_QCAP + (idx << 4))) : 0xFFFFFFFFu;
        }
        __builtin_amdgcn_wave_barrier();
        f32x2w ring_a[WT_DEPTH][4];
        unsigned ring_e[WT_DEPTH][4];
        auto issue = [&](int sidx, f32x2w (&ra)[4], unsigned (&re)[4]) {
            const unsigned ent = sidx < 64 ? rl[wv][sidx] : 0xFFFFFFFFu;
            const bool live = ent != 0xFFFFFFFFu;
            const unsigned* q = &q_all[wv][0][0] + (live ? ent : 0u);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const unsigned e = q[4 * tt + g];
                re[tt] = e;
                const unsigned ia = live ? ((e >> 9) * (unsigned)cin + (unsigned)(ci_base + 2 * ii)) * 4u : OOB;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(ra[tt]) : "v"(ia), "s"(in_desc) : "memory");
            }
        };
        auto mult = [&](f32x2w (&ra)[4], const unsigned (&re)[4], f32x4 (&ac)[MTB][NTB]) {
            float bb[4][NTB];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const float* bp = tile + (re[tt] & 511u) * CO + NTB * ii;
                if constexpr (NTB == 2) {
                    const f32x2w v = *reinterpret_cast<const f32x2w*>(bp);
                    bb[tt][0] = v[0], bb[tt][1] = v[1];
                } else {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
                    for (int n = 0; n < 4; ++n) bb[tt][n] = v[n];
                }
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int m = 0; m < MTB; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n) ac[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[tt][m], bb[tt][n], ac[m][n], 0, 0, 0);
        };
        if (s_tot > 0) {
#pragma unroll
            for (int pp = 0; pp < WT_DEPTH; ++pp) issue(pp, ring_a[pp], ring_e[pp]);
#pragma unroll
            for (int j = 0; j < WT_SLOTS; ++j) {
                for (int i0 = 0; i0 < rj[j]; i0 += WT_DEPTH) {
#pragma unroll
                    for (int pp = 0; pp < WT_DEPTH; ++pp) {
                        // the oldest round of the ring has landed when at most (WT_DEPTH - 1) x 4 younger loads are outstanding
                        asm volatile("s_waitcnt vmcnt(%4)"
                                     : "+v"(ring_a[pp][0]), "+v"(ring_a[pp][1]), "+v"(ring_a[pp][2]), "+v"(ring_a[pp][3])
                                     : "n"((WT_DEPTH - 1) * 4));
                        if (i0 + pp < rj[j]) mult(ring_a[pp], ring_e[pp], acc[j]);
                        issue(base[j] + i0 + pp + WT_DEPTH, ring_a[pp], ring_e[pp]);
                    }
                }
            }
            // the dummy loads behind the end of the stream: their destinations stay live up to this wait (a register the compiler
            // considers dead would be handed to another value and overwritten when the load returns)
#pragma unroll
            for (int pp = 0; pp < WT_DEPTH; ++pp)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring_a[pp][0]), "+v"(ring_a[pp][1]), "+v"(ring_a[pp][2]), "+v"(ring_a[pp][3])::"memory");
        }
        if constexpr (!EARLY) {
            fetch_tile(tn);
            fetch_ids(tn);
        }
        // 3. tails
#pragma unroll
        for (int j = 0; j < WT_SLOTS; ++j) {
            unsigned* q = q_all[wv][j];
            const int done = rj[j] << 4;
            const int left = qn[j] - done;
            if (done == 0 && carried[j] > 0) {
                // the tail has met no full round for a whole tile and holds pairs of the PREVIOUS tile: flush it as a partial round
                float a_fl[4][MTB];
                gather(q, 0, left, a_fl);
                multiply(q, 0, left, a_fl, acc[j]);
                qn[j] = 0;
            } else {
                if (done > 0 && left > 0) {      // move the tail (< 16 entries) to the front of the queue
                    unsigned te = 0;
                    if (lane < left) te = q[done + lane];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < left) q[lane] = te;
                }
                qn[j] = left;
            }
            __builtin_amdgcn_wave_barrier();
        }
        buf = buf == 2 ? 0 : buf + 1;
        store_tile(buf);                 // tile t + gx -> the buffer last read two tiles ago
    }
#pragma unroll
    for (int j = 0; j < WT_SLOTS; ++j)
        if (qn[j] > 0) {
            float a_fl[4][MTB];
            gather(q_all[wv][j], 0, qn[j], a_fl);
            multiply(q_all[wv][j], 0, qn[j], a_fl, acc[j]);
        }

    // D: col = lane & 15 -> cout, row = 4 g + reg -> cin (channel interleave MTB / NTB inside this workgroup's channel block)
    float* dst = slab + (size_t)gid * CO * WT_K * cin;
#pragma unroll
    for (int j = 0; j < WT_SLOTS; ++j) {
        if (koff[j] < 0) continue;
#pragma unroll
        for (int m = 0; m < MTB; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int ci = ci_base + MTB * (4 * g + reg) + m;
                    const int co = NTB * ii + n;
                    dst[((size_t)co * WT_K + koff[j]) * cin + ci] = acc[j][m][n][reg];
                }
    }
}

// workgroups (= slabs) per channel block of the dout-stationary wgrad: two (32-channel dout) or one per CU
static int wgrad_tile_groups(int n_out, int cin, int cout) {
    static const int env_g = getenv("TODA_WG_TILE_GROUPS") ? atoi(getenv("TODA_WG_TILE_GROUPS")) : 0;
    const int nsub = cin / (cout == 64 ? 32 : cin);   // 64 output channels: input channels in blocks of 32
    int G = env_g > 0 ? env_g : (cout == 32 ? 512 : 256) / nsub;
    const int tiles = (n_out + WT_R - 1) / WT_R;
    while (G > 8 && G > tiles) G -= 8;
    return (G + 7) / 8 * 8;
}

// Row chunks per offset: every (chunk, offset) pair is a workgroup and a slab the reduce kernel reads back.  Measured on the
// C3 / C5 levels: the >= 64-channel kernels (4 waves / SIMD resident) run best with at most 48 chunks (64x64 @ 389k rows 0.606 ->
// 0.587 ms, @ 227k 0.373 -> 0.338 ms), the 32-channel ones (7 waves / SIMD) with up to 144 (32x32 @ 682k 0.418 -> 0.388 ms).
// The grid is (chunks, K, sub-blocks): with few offsets (conv_out's 3 x 1 x 1 kernel) 48 chunks leave most CUs without a block
// (48 x 3 = 144 blocks); measured on the K = 3 layers (64 -> 128 @ 111 k rows / 128 -> 128 @ 91 k rows): 48 chunks 0.137 / 0.271 ms,
// 96 0.103 / 0.165, 144 0.118 / 0.168, 216 0.141 / 0.190, 432 0.226 / 0.214 (the slab fold grows with the chunk count).
}  // namespace toda


namespace toda {

// Dispatch of the opt-in kernels: called by gather_gemm_impl before its default path; *handled = false leaves the launch to it.
static int variant_gather_gemm(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr, int n_out, int k_vol,
                               int c_produce, const float* bias, float* out, const int32_t* order, double* stats, hipStream_t s,
                               const unsigned char* cls_sorted, int Q, int NT, bool vec_ok, int env_lds, bool subm_table, bool* handled) {
    *handled = true;
    static const int env_rt = getenv("TODA_GG_RT") ? atoi(getenv("TODA_GG_RT")) : 0;      // row tiles per wave: 1 / 2 / 4 (0 = built-in choice)
    const int env_lds88 = getenv("TODA_GG_LDS88") ? atoi(getenv("TODA_GG_LDS88")) : 3;    // (read per call: tests and A/B runs flip it inside one process)
    // 128 -> 128: 1 = RT 1 / 256 threads (0.67 ms), 2 = RT 2 / 256 (0.76), 4 = RT 2 / 512, 5 = half slices by LDS-DMA (gather_gemm_wide_kernel);
    // 3 = the default kernel
    if (env_lds88 != 3 && env_lds88 != 0 && vec_ok && Q == 8 && NT == 8 && !cls_sorted) {
        if (env_lds88 == 2)
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<8, 8, 2, true, false>), dim3(cdiv(cdiv(n_out, 32), SC_BLOCK / 64)),
                      dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        else if (env_lds88 == 5 && c_gather == 128 && c_produce == 128 && order == nullptr && !stats)
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_wide_kernel<512>), dim3(cdiv(cdiv(n_out, 16), 8)), dim3(512), 0, s, in, n_in, c_gather, wp, nbr,
                      n_out, k_vol, c_produce, bias, out);
        else if (env_lds88 == 4)
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<8, 8, 2, true, false, 512>), dim3(cdiv(cdiv(n_out, 32), 8)),
                      dim3(512), 0, s, in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        else
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<8, 8, 1, true, false>), dim3(cdiv(cdiv(n_out, 16), SC_BLOCK / 64)),
                      dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    // 64 -> 64: 3 offsets per LDS stage, 512-thread workgroups (fewer, better balanced barriers)
    // measured 0.502 ms against 0.498 ms for the per-offset barrier kernel on the 389.5k-row level - the barrier imbalance it removes is
    // not what holds the matrix pipe at 70 %.
    static const int env_stage = getenv("TODA_GG_STAGE") ? atoi(getenv("TODA_GG_STAGE")) : 0;
    if (env_stage && !cls_sorted && vec_ok && order == nullptr && Q == 4 && NT == 4 && c_gather == 64 && c_produce == 64 && n_out >= 8192) {
        const int blocks = cdiv(cdiv(n_out, 32), 8);
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_stage_kernel<4, 4, 3, 512>), dim3(blocks), dim3(512), 0, s, in, n_in, c_gather, wp, nbr, n_out,
                  k_vol, c_produce, bias, out, stats);
        TODA_LAUNCH_CHECK();
        if (stats) {
            fold_or_defer(stats, blocks, c_produce, s);
            TODA_LAUNCH_CHECK();
        }
        return TODA_OK;
    }
    // 64 -> 64: wave-specialised producer / consumer kernel (persistent, one 512-thread workgroup per CU)
    // measured 0.653 ms against 0.539 ms for gather_gemm_lds_kernel on the 389.5k-row stride-4 level (0.245 vs 0.207 ms at 117k rows).
    // The consumers' MFMA stream is clean (about 2,950 cycles per offset and 128 rows), but the producers' LDS-DMA row gather runs at
    // 16-30 GB/s per CU out of a 100 MB feature table (MI355X_MICROARCH.md "Indexed rows: gather into LDS"), i.e. about 4,800 cycles
    // for the 32 KiB of an offset: the design is gather-bound, three stages deep or not.
    static const int env_ws = getenv("TODA_GG_WS") ? atoi(getenv("TODA_GG_WS")) : 0;
    if (env_ws && !cls_sorted && !stats && vec_ok && order == nullptr && Q == 4 && NT == 4 && c_gather == 64 && n_out >= 8192) {
        static int n_cu_ws = 0;
        if (!n_cu_ws) {
            int dev = 0;
            hipDeviceProp_t prop;
            TODA_HIP(hipGetDevice(&dev));
            TODA_HIP(hipGetDeviceProperties(&prop, dev));
            n_cu_ws = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        const int n_tiles = cdiv(n_out, GW_ROWS);
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_ws_kernel<4, 4>), dim3(n_tiles < n_cu_ws ? n_tiles : n_cu_ws), dim3(GW_BLOCK), 0, s, in, n_in,
                  c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, n_tiles);
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    // narrow layers: all K offsets of the packed weights resident in LDS (<= 108 KiB), barrier-free offset loop
    // measured slower than the per-offset LDS slices (32->32 @ 682k rows 0.356 vs 0.321 ms, 16->32 0.234 vs 0.168)
    static const int env_wres = getenv("TODA_GG_WRES") ? atoi(getenv("TODA_GG_WRES")) : 0;
    if (env_wres && !cls_sorted && !stats && vec_ok && order == nullptr && Q <= 2 && NT <= 2 && k_vol <= 27 && n_out >= 4096) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            TODA_HIP(hipGetDevice(&dev));
            TODA_HIP(hipGetDeviceProperties(&prop, dev));
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        const int tiles = cdiv(n_out, 32);
        const int per_cu = Q * NT <= 2 ? 2 : 1;      // <= 54 KiB of weights: two workgroups (8 waves / SIMD) per CU
        const int grid = cdiv(tiles, WR_BLOCK / 64) < n_cu * per_cu ? cdiv(tiles, WR_BLOCK / 64) : n_cu * per_cu;
#define WR(QQ, NN)                                                                                                     \
    GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_wres_kernel<QQ, NN, 2, 27>), dim3(grid), dim3(WR_BLOCK), 0, s, in, n_in, c_gather, wp, \
              nbr, n_out, k_vol, c_produce, bias, out)
        if (Q == 1 && NT == 1) WR(1, 1);
        else if (Q == 1) WR(1, 2);
        else if (NT == 1) WR(2, 1);
        else WR(2, 2);
#undef WR
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    // submanifold tables (toda_spconv_gather_gemm_subm): x-run operand reuse.  TODA_GG_LINE=1 (read by the caller's routing: the hint
    // arrives only when ops._line_route took it)
    if (subm_table && k_vol == 27 && vec_ok && !cls_sorted && order == nullptr && Q == NT && (Q == 2 || Q == 4) && c_gather == 16 * Q &&
        c_produce == 16 * NT && n_out >= 64) {
        const int blocks = cdiv(cdiv(n_out, 32), SC_BLOCK / 64);
        if (Q == 4)
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_line_kernel<4, 4, 2>), dim3(blocks), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out,
                      c_produce, bias, out, stats);
        else
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_line_kernel<2, 2, 2>), dim3(blocks), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out,
                      c_produce, bias, out, stats);
        TODA_LAUNCH_CHECK();
        if (stats) {
            fold_or_defer(stats, blocks, c_produce, s);
            TODA_LAUNCH_CHECK();
        }
        return TODA_OK;
    }
    // instantiations of the default LDS kernel: register-pipelined gathers (TODA_GG_LDS_PF: 154 VGPRs = 3 waves per SIMD, 0.603 vs 0.595 ms),
    // 8 waves per weight slice (TODA_GG_BLK512: 0.618 vs 0.581 ms, round 4), 1 or 4 row tiles per wave (TODA_GG_RT)
    static const int env_pfl = getenv("TODA_GG_LDS_PF") ? atoi(getenv("TODA_GG_LDS_PF")) : 0;
    static const int env_blk512 = getenv("TODA_GG_BLK512") ? atoi(getenv("TODA_GG_BLK512")) : 0;
    if ((env_lds || stats) && !cls_sorted && vec_ok && Q * NT <= 32 && Q <= 4 && NT <= 4) {
        int blocks = 0;
        if (env_blk512 && Q == 4 && NT == 4) {
            blocks = cdiv(cdiv(n_out, 32), 8);
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<4, 4, 2, true, true, 512, false, true>), dim3(blocks), dim3(512), 0, s, in, n_in,
                      c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        } else if (env_pfl && Q == 4 && NT == 4) {
            blocks = cdiv(cdiv(n_out, 32), SC_BLOCK / 64);
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<4, 4, 2, true, true, SC_BLOCK, true>), dim3(blocks), dim3(SC_BLOCK), 0, s, in,
                      n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        } else if (env_rt == 1 || env_rt == 4) {
            blocks = cdiv(cdiv(n_out, 16 * env_rt), SC_BLOCK / 64);
#define GLV(QQ, NN)                                                                                                                  \
    if (env_rt == 1)                                                                                                                 \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<QQ, NN, 1, true, true>), dim3(blocks), dim3(SC_BLOCK), 0, s, in, n_in,       \
                  c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);                                              \
    else                                                                                                                             \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<QQ, NN, 4, true, true>), dim3(blocks), dim3(SC_BLOCK), 0, s, in, n_in,       \
                  c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats)
#define GLV_ROW(QQ)                  \
    switch (NT) {                    \
        case 1: GLV(QQ, 1); break;   \
        case 2: GLV(QQ, 2); break;   \
        default: GLV(QQ, 4); break;  \
    }
            switch (Q) {
                case 1: GLV_ROW(1); break;
                case 2: GLV_ROW(2); break;
                default: GLV_ROW(4); break;
            }
#undef GLV_ROW
#undef GLV
        }
        if (blocks) {
            TODA_LAUNCH_CHECK();
            if (stats) {
                fold_or_defer(stats, blocks, c_produce, s);
                TODA_LAUNCH_CHECK();
            }
            return TODA_OK;
        }
    }
    *handled = false;
    return TODA_OK;
}

}  // namespace toda

using namespace toda;

// ---- mask-sorted processing order -----------------------------------------------------------
// Inside blocks of ORD_B consecutive (canonical, i.e. spatially adjacent) rows, rows are visited in
// ascending order of their K-bit neighbour mask, so that the 32 rows of a wave share offsets and
// whole (tile, offset) pairs drop out of the gather + MFMA loop (measured on the C3 tables: non-empty
// fraction 0.74 -> 0.61 at the stride-2 SubM level, 0.29 -> 0.13 for the strided convs' dgrad).
// Results are unchanged: each output row is still produced by one wave from its own K inputs.
namespace toda {
constexpr int ORD_B = 2048;
__global__ void __launch_bounds__(256)
row_order_kernel(const int* __restrict__ nbr, int n, int K, int* __restrict__ order) {
    __shared__ unsigned long long key[ORD_B];
    const int base = blockIdx.x * ORD_B;
    for (int i = threadIdx.x; i < ORD_B; i += 256) {
        const int row = base + i;
        unsigned m = 0xFFFFFFFFu;  // padding sorts last
        if (row < n) {
            m = 0;
            for (int k = 0; k < K; ++k) m |= (unsigned)(nbr[(size_t)k * n + row] >= 0) << k;
        }
        key[i] = ((unsigned long long)m << 32) | (unsigned)i;
    }
    __syncthreads();
    for (int span = 2; span <= ORD_B; span <<= 1) {
        for (int j = span >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < ORD_B / 2; t += 256) {
                const int i = 2 * t - (t & (j - 1));  // element whose bit j is clear
                const int l = i + j;
                const bool up = (i & span) == 0;
                const unsigned long long a = key[i], b = key[l];
                if ((a > b) == up) {
                    key[i] = b;
                    key[l] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < ORD_B; i += 256)
        if (base + i < n) order[base + i] = base + (int)(key[i] & 0xFFFFFFFFull);
}
}  // namespace toda

extern "C" int toda_rulebook_row_order(const int32_t* nbr, int n_out, int k_vol, int32_t* order, void* stream) {
    TODA_CHECK_ARG(n_out >= 0 && k_vol >= 1 && k_vol <= 31, "rulebook_row_order: needs 1 <= K <= 31 offsets");
    if (n_out == 0) return TODA_OK;
    hipLaunchKernelGGL(row_order_kernel, dim3(cdiv(n_out, ORD_B)), dim3(256), 0, (hipStream_t)stream, nbr, n_out, k_vol, order);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}


extern "C" int toda_spconv_gather_gemm_halo(const float* in, int n, int c_gather, const float* wp, const int32_t* nbr, int k_vol, int c_produce,
                                            const float* bias, float* out, const void* plan, size_t plan_bytes, double* sums, size_t sums_doubles,
                                            void* stream) {
    HaloGeomK hg;
    TODA_CHECK_ARG(c_gather == c_produce && halo_geom_k(c_gather, &hg) && k_vol >= 2 && k_vol <= 27,
                   "gather_gemm_halo: unsupported shape (%d -> %d channels, %d offsets)", c_gather, c_produce, k_vol);
    TODA_CHECK_ARG(n >= 0, "gather_gemm_halo: n < 0");
    if (n == 0) return TODA_OK;
    TODA_CHECK_ARG(in && wp && nbr && out && plan, "gather_gemm_halo: null pointer");
    TODA_CHECK_ARG((unsigned long long)n * c_gather * 4ull < 0xFFFFFFF0ull, "gather_gemm_halo: feature table must be < 4 GiB");
    const size_t nb = (size_t)cdiv(n, hg.R);
    const size_t o_urows = align_up(nb * hg.R * 4, 256), o_lids = o_urows + align_up(nb * hg.UMAX * 4, 256);
    TODA_CHECK_ARG(plan_bytes >= o_lids + nb * k_vol * hg.R * 2, "gather_gemm_halo: plan buffer too small");
    TODA_CHECK_ARG(sums == nullptr || sums_doubles >= (size_t)2 * c_produce * (1 + nb), "gather_gemm_halo: statistics buffer too small");
    const int32_t* order = (const int32_t*)plan;
    const int32_t* urows = (const int32_t*)((const char*)plan + o_urows);
    const unsigned short* lids = (const unsigned short*)((const char*)plan + o_lids);
    hipStream_t s = (hipStream_t)stream;
    static const int ablate = getenv("TODA_HALO_ABLATE") ? atoi(getenv("TODA_HALO_ABLATE")) : 0;      // timing experiments only (wrong results)
    if (c_gather == 64)
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_halo_kernel<4, 4, 2, 128, 320, 27>), dim3((unsigned)nb), dim3(512), 0, s, in, n, c_gather, wp, nbr, k_vol,
                  c_produce, bias, out, order, urows, lids, sums, ablate);
    else
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_halo_kernel<2, 2, 2, 128, 320, 27>), dim3((unsigned)nb), dim3(512), 0, s, in, n, c_gather, wp, nbr, k_vol,
                  c_produce, bias, out, order, urows, lids, sums, ablate);
    TODA_LAUNCH_CHECK();
    if (sums) {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(2 * c_produce), dim3(256), 0, s, sums, (int)nb, 2 * c_produce);
        TODA_LAUNCH_CHECK();
    }
    return TODA_OK;
}

// dout-stationary wgrad (wgrad_tile_kernel): K = 27, 32 or 64 channels on both sides with cin <= cout, input rows < 2^23
extern "C" int toda_spconv_wgrad_tiled_supported(int n_in, int n_out, int k_vol, int cin, int cout) {
    return k_vol == WT_K && n_out > 0 && n_in > 0 && n_in < (1 << 23) && (cin == 32 || cin == 64) && (cout == 32 || cout == 64) && cin <= cout &&
           (unsigned long long)n_in * cin * 4ull < 0xFFFFFFF0ull && (unsigned long long)n_out * cout * 4ull < 0xFFFFFFF0ull &&
           (unsigned long long)n_out * WT_K * 4ull < 0xFFFFFFF0ull;
}

extern "C" size_t toda_spconv_wgrad_tiled_workspace_bytes(int n_out, int cin, int cout) {
    if (!((cin == 32 || cin == 64) && (cout == 32 || cout == 64) && cin <= cout)) return 0;
    return align_up((size_t)wgrad_tile_groups(n_out, cin, cout) * WT_K * cin * cout * sizeof(float), 256);
}

extern "C" int toda_spconv_wgrad_tiled(const float* in, int n_in, const float* dout, const int32_t* nbr, int n_out, int k_vol,
                                       int cin, int cout, float* dw, void* ws, size_t ws_bytes, void* stream) {
    TODA_CHECK_ARG(toda_spconv_wgrad_tiled_supported(n_in, n_out, k_vol, cin, cout),
                   "wgrad_tiled: needs K = 27, channels in {32, 64} with cin <= cout, 0 < rows < 2^23 (got K %d, %d -> %d, %d / %d rows)", k_vol, cin,
                   cout, n_in, n_out);
    hipStream_t s = (hipStream_t)stream;
    const long long elems = (long long)cout * k_vol * cin;
    const int groups = wgrad_tile_groups(n_out, cin, cout);
    const size_t need = (size_t)groups * elems * sizeof(float);
    if (ws_bytes < need) {
        set_error("wgrad_tiled: workspace %zu < required %zu", ws_bytes, need);
        return TODA_EWORKSPACE;
    }
    const int n_tiles = (n_out + WT_R - 1) / WT_R;
    const int nsub = cin / (cout == 64 ? 32 : cin);
    float* const slabs = (float*)ws;
    const dim3 grid(groups * nsub);
    static const int env_prof = getenv("TODA_WG_TILE_PROFILE") ? atoi(getenv("TODA_WG_TILE_PROFILE")) : -1;
    const int profile = env_prof >= 0 ? (env_prof & 1) : (cin == 32 ? 0 : 1);
    if (cout == 32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_tile_kernel<2, 2>), grid, dim3(WT_BLOCK), 0, s, in, n_in, cin, dout, nbr, n_out, n_tiles,
                           groups, nsub, profile, slabs);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_tile_kernel<2, 4>), grid, dim3(WT_BLOCK), 0, s, in, n_in, cin, dout, nbr, n_out, n_tiles,
                           groups, nsub, profile, slabs);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(elems, SC_BLOCK)), dim3(SC_BLOCK), 0, s, slabs, groups, elems, dw);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

