// Sparse convolution arithmetic for gfx950: weight packing, gather -> fp32-MFMA GEMM -> store
// (forward and dgrad share one kernel), and wgrad.
//
// Formulation: OUTPUT STATIONARY.  A wave owns 16*RT output rows and all produced channels; for
// each kernel offset k it gathers the neighbour rows named by the k-major table nbr[k][row]
// straight from HBM/L2 into MFMA A-fragments (16-byte loads, no LDS round trip), multiplies by
// the offset's weight slice (pre-packed in fragment order, 1 KiB coalesced loads that every wave
// shares through L1/L2) with v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD) and keeps the
// sums in registers until one vectorised store.  No atomics, deterministic, every output row is
// written once; offsets none of whose 16*RT rows has a neighbour are skipped wave-uniformly.
//
// MFMA operand maps (cdna_hip_programming.md §3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D: col = lane&15, row = 4*(lane>>4) + reg.
// Two layout tricks remove all shuffles:
//   * k-permutation: lane (r, g) loads 4 CONSECUTIVE gathered channels 16q+4g..+3 as one float4;
//     MFMA step (q, j) therefore contracts over channel 16q+4g+j in lane group g, and the packed
//     weights are laid out with the same permutation.
//   * channel-interleaved N tiles: column c of tile n is produced channel NT*c + n, so a lane's
//     NT accumulators for one row are NT consecutive channels -> one 16/32-byte store.
#include <stdlib.h>

#include <type_traits>

#include <hip/hip_ext.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace toda {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SC_BLOCK = 256;

static int tiles_pow2(int channels) {
    int t = (channels + 15) / 16, p = 1;
    while (p < t) p <<= 1;
    return p;
}

// wp[(((k*Q + q)*NT + n)*64 + lane)*4 + j]
__global__ void __launch_bounds__(SC_BLOCK)
pack_weight_kernel(const float* __restrict__ w, int cout, int K, int cin, int transpose, int flip_k, int Q, int NT,
                   float* __restrict__ wp) {
    const long long e = (long long)blockIdx.x * SC_BLOCK + threadIdx.x;
    const long long total = (long long)K * Q * NT * 256;
    if (e >= total) return;
    const int j = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    long long t = e >> 8;
    const int n = (int)(t % NT);
    t /= NT;
    const int q = (int)(t % Q);
    const int k = (int)(t / Q);
    const int c = lane & 15, g = lane >> 4;
    const int gch = 16 * q + 4 * g + j;  // gathered channel
    const int pch = NT * c + n;          // produced channel
    const int kk = flip_k ? K - 1 - k : k;
    float v = 0.0f;
    if (!transpose) {
        if (gch < cin && pch < cout) v = w[((size_t)pch * K + kk) * cin + gch];
    } else {
        if (gch < cout && pch < cin) v = w[((size_t)gch * K + kk) * cin + pch];
    }
    wp[e] = v;
}

// All packs of a training step (forward operand and dgrad operand of every sparse conv of a backbone) in ONE launch:
// blockIdx.y = segment.  A step otherwise spends 46 launches of ~4 us (and as many host calls) on them.
constexpr int PACK_MAX_SEG = 48;
struct PackBatch {
    const float* w[PACK_MAX_SEG];
    float* wp[PACK_MAX_SEG];
    int cout[PACK_MAX_SEG], K[PACK_MAX_SEG], cin[PACK_MAX_SEG];
    unsigned char transpose[PACK_MAX_SEG], flip[PACK_MAX_SEG], Q[PACK_MAX_SEG], NT[PACK_MAX_SEG];
};

__global__ void __launch_bounds__(SC_BLOCK)
pack_weight_batch_kernel(const PackBatch b) {
    const int sg = blockIdx.y;
    const int K = b.K[sg], Q = b.Q[sg], NT = b.NT[sg], cin = b.cin[sg], cout = b.cout[sg];
    const long long e = (long long)blockIdx.x * SC_BLOCK + threadIdx.x;
    if (e >= (long long)K * Q * NT * 256) return;
    const int j = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    long long t = e >> 8;
    const int n = (int)(t % NT);
    t /= NT;
    const int q = (int)(t % Q);
    const int k = (int)(t / Q);
    const int c = lane & 15, g = lane >> 4;
    const int gch = 16 * q + 4 * g + j, pch = NT * c + n;
    const int kk = b.flip[sg] ? K - 1 - k : k;
    const float* __restrict__ w = b.w[sg];
    float v = 0.0f;
    if (!b.transpose[sg]) {
        if (gch < cin && pch < cout) v = w[((size_t)pch * K + kk) * cin + gch];
    } else {
        if (gch < cout && pch < cin) v = w[((size_t)gch * K + kk) * cin + pch];
    }
    b.wp[sg][e] = v;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Buffer resource over a whole fp32 table.  Reads through it are bounds-checked by the hardware:
// an offset >= bytes returns 0 and touches no memory, which is how "no neighbour" (-1) rows are
// gathered as zeros WITHOUT a branch (branches around loads make hipcc serialise them behind
// s_waitcnt vmcnt(0), cdna_hip_programming.md §5 trap (c)).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
constexpr unsigned OOB = 0xFFFFFFF0u;

// Gather one wave's A fragments for one offset: lane (r, g) of row tile rt reads the 4 channels
// 16q+4g..+3 of input row src[rt] (zeros when src < 0).  Channels >= cg (only when cg is not a
// multiple of 16) meet zero weights in the packed operand and are zeroed here as well.
template <int Q, int RT, bool VEC>
__device__ __forceinline__ void gather_rows(__amdgpu_buffer_rsrc_t rsrc, int cg, int g, const int (&src)[RT],
                                            f32x4 (&a)[RT][Q]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const unsigned row_off = (unsigned)src[rt] * (unsigned)cg * 4u;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int col = 16 * q + 4 * g;
            const bool ok = src[rt] >= 0 && col < cg;
            const unsigned off = row_off + (unsigned)col * 4u;
            f32x4 v;
            if constexpr (VEC) {
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off : OOB, 0, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         rsrc, (ok && col + j < cg) ? off + 4u * j : OOB, 0, 0));
            }
            a[rt][q] = v;
        }
    }
}

// Data gradient of a STRIDED convolution.  Input site i reaches output o = (i + pad - k) / stride only for the kernel offsets k with
// k = i + pad (mod stride) on every axis: a fine voxel has 1, 2, 4 or 8 CANDIDATE offsets out of 27 (27 / 8 on average - exactly
// the pair density of these tables), fixed by the residues of its coordinates, and every candidate exists (the output set holds
// every cell an input reaches).  Rows grouped by residue class (toda_rulebook_class_order) therefore form tiles whose work is a
// short list of fully populated offsets; a wave whose 16 RT rows share one class walks only that list (classes and lists in
// GatherClasses, passed by value), any other wave walks all K offsets as before.  Same sums in the same order: bit-identical.
struct GatherClasses {
    unsigned char count[8];       // candidate offsets per class (0 = no class information)
    unsigned char k[8][27];       // ascending kernel-offset indices
};

// PF = software pipeline depth: with PF the neighbour ids of offset k+2 and the gathered rows of
// offset k+1 are requested before the MFMAs of offset k issue, so a wave's HBM/L2 round trips run
// under its own matrix work instead of relying on other waves to cover them.
template <int Q, int NT, int RT, bool PF, bool VEC, bool CLS = false>
__global__ void __launch_bounds__(SC_BLOCK)
gather_gemm_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp,
                   const int* __restrict__ nbr, int n_out, int K, int cp, const float* __restrict__ bias,
                   float* __restrict__ out, int xcd_order, const int* __restrict__ order,
                   const unsigned char* __restrict__ cls_sorted, const GatherClasses classes) {
    const int lane = threadIdx.x & 63;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so give
    // each XCD one contiguous range of row tiles - canonical rows are spatial neighbours and gather
    // overlapping input rows, which then hit in that XCD's L2 instead of being fetched 8 times.
    // (speed only; any placement is correct)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile_block = xcd_order ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3)
                                     : (int)blockIdx.x;
    const int wave = tile_block * (SC_BLOCK / 64) + (threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row0 = wave * (16 * RT);
    if (row0 >= n_out) return;  // wave-uniform

    f32x4 acc[RT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float b = 0.0f;
        if (bias && NT * r + n < cp) b = bias[NT * r + n];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][n] = f32x4{b, b, b, b};
    }
    const f32x4* __restrict__ wp4 = reinterpret_cast<const f32x4*>(wp);
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
    int rows[RT];
    bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        live[rt] = row0 + rt * 16 + r < n_out;
        rows[rt] = live[rt] ? (order ? order[row0 + rt * 16 + r] : row0 + rt * 16 + r) : n_out - 1;  // clamped: loads stay unconditional
    }

    auto load_ids = [&](int k, int (&dst)[RT]) {
        const int kk = k < K ? k : K - 1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int v = nbr[(size_t)kk * n_out + rows[rt]];
            dst[rt] = (k < K && live[rt]) ? v : -1;
        }
    };
    // MASK = compile-time set of row tiles that take part: the wave-uniform "does this tile have a
    // neighbour at offset k" test is made ONCE per offset (3 specialised bodies for RT = 2), not
    // around every MFMA, so the matrix instructions issue back to back.
    auto mma_masked = [&](auto mask_tag, int k, const f32x4 (&a)[RT][Q]) {
        constexpr unsigned MASK = decltype(mask_tag)::value;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            f32x4 b[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = wp4[(((size_t)k * Q + q) * NT + n) * 64 + lane];
            // j outermost: consecutive MFMAs hit different accumulators (dependent latency 40 > issue 32)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if ((MASK >> rt) & 1u)  // folds after unrolling
                            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q][j], b[n][j], acc[rt][n], 0, 0, 0);
                    }
                }
            }
        }
    };
    auto mma = [&](int k, const f32x4 (&a)[RT][Q], const bool (&hit)[RT]) {
        if constexpr (RT == 1) {
            mma_masked(std::integral_constant<unsigned, 1u>{}, k, a);
        } else if constexpr (RT == 2) {
            if (hit[0] && hit[1]) mma_masked(std::integral_constant<unsigned, 3u>{}, k, a);
            else if (hit[0]) mma_masked(std::integral_constant<unsigned, 1u>{}, k, a);
            else mma_masked(std::integral_constant<unsigned, 2u>{}, k, a);
        } else {
            mma_masked(std::integral_constant<unsigned, (1u << RT) - 1u>{}, k, a);  // RT = 4: no per-tile skip
        }
    };

    if constexpr (PF) {
        int s0[RT], s1[RT], s2[RT];
        f32x4 a0[RT][Q], a1[RT][Q];
        load_ids(0, s0);
        load_ids(1, s1);
        gather_rows<Q, RT, VEC>(in_rsrc, cg, g, s0, a0);
        for (int k = 0; k < K; ++k) {
            load_ids(k + 2, s2);
            gather_rows<Q, RT, VEC>(in_rsrc, cg, g, s1, a1);  // rows of offset k+1, in flight during the MFMAs below
            bool hit[RT];
            bool any = false;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                hit[rt] = __any(s0[rt] >= 0);
                any = any || hit[rt];
            }
            if (any) mma(k, a0, hit);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                s0[rt] = s1[rt];
                s1[rt] = s2[rt];
#pragma unroll
                for (int q = 0; q < Q; ++q) a0[rt][q] = a1[rt][q];
            }
        }
    } else {
        auto offset = [&](int k) {
            int src[RT];
            load_ids(k, src);
            bool hit[RT];
            bool any = false;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                hit[rt] = __any(src[rt] >= 0);
                any = any || hit[rt];
            }
            if (!any) return;  // wave-uniform skip of an empty offset
            f32x4 a[RT][Q];
            gather_rows<Q, RT, VEC>(in_rsrc, cg, g, src, a);
            mma(k, a, hit);
        };
        // CLS (separate instantiation, so that the plain kernel compiles as it always did): class of this wave's rows (positions
        // row0 .. row0 + 16 RT - 1 of the class-sorted order); if they all agree the wave walks that class's offset list only
        const unsigned char* klist = nullptr;
        int n_k = 0;
        if constexpr (CLS) {
            const int p = row0 + (lane & (16 * RT - 1) & 63);
            const int c = cls_sorted[p < n_out ? p : n_out - 1];
            const int c0 = __builtin_amdgcn_readfirstlane(c);
            if (__all(c == c0 || p >= n_out) && classes.count[c0 & 7] > 0) {
                klist = classes.k[c0 & 7];
                n_k = classes.count[c0 & 7];
            }
        }
        if (CLS && klist) {
            for (int t = 0; t < n_k; ++t) offset(klist[t]);
        } else {
            for (int k = 0; k < K; ++k) offset(k);
        }
    }

    const bool full = cp == 16 * NT;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pos = row0 + rt * 16 + 4 * g + reg;
            const int row = __shfl(rows[rt], 4 * g + reg, 64);  // lane r' holds the (possibly permuted) row of tile position r'
            if (pos >= n_out) continue;
            float* dst = out + (size_t)row * cp + NT * r;
            if (full) {
                if constexpr (NT == 1) {
                    dst[0] = acc[rt][0][reg];
                } else if constexpr (NT == 2) {
                    *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][reg], acc[rt][1][reg]);
                } else {
#pragma unroll
                    for (int n = 0; n < NT; n += 4)
                        *reinterpret_cast<f32x4*>(dst + n) =
                            f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (NT * r + n < cp) dst[n] = acc[rt][n][reg];
            }
        }
    }
}

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

// Row blocks -> XCDs.  The hardware deals workgroup ids round-robin over the 8 XCDs, each with a private 4 MiB L2: with the
// identity mapping every XCD walks the whole (sorted) row range, so all 8 L2s fetch the whole gathered table from HBM (PMC: 690 MB
// fetched per launch of the 64 -> 64 kernel on a 100 MB table).  Handing each XCD one contiguous eighth balances badly (z-slabs
// differ in density: measured 6-8 % slower).  In between: chunks of GG_XCD_CHUNK consecutive row blocks are dealt round-robin, XCD x
// works on chunks x, x + 8, ...: neighbours in x / y of a row stay inside its chunk's L2, every XCD still samples the whole scene.
// Measured on the 389.5k-row 64 -> 64 layer (FETCH_SIZE per launch, kernel ms): identity 337.8 MiB-units / 0.595, chunk 16
// 191.1 / 0.593, 32 181.0 / 0.590, 64 185.3 / 0.606, 256 - / 0.629: HBM traffic 3.2x -> 1.9x the algorithmic bytes at equal speed.
#ifndef GG_XCD_CHUNK
#define GG_XCD_CHUNK 32
#endif
__device__ int d_xcd_chunk = GG_XCD_CHUNK;      // TODA_GG_XCD_CHUNK overrides it (experiments: the chunk sweep per launch shape)
__device__ __forceinline__ int xcd_chunked_block(int b, int nblk) {
    const int C = d_xcd_chunk;
    if (C <= 0) return b;
    const int per = 8 * C, full = (nblk / per) * per;
    if (b >= full) return b;
    const int xcd = b & 7, local = b >> 3;
    return ((local / C) * 8 + xcd) * C + local % C;
}

#ifndef GG_LDS_WAVES_WIDE
#define GG_LDS_WAVES_WIDE 4   // 512-thread blocks of the 128-channel variant: 2 blocks x 8 waves per CU
#endif
#ifndef GG_LDS_WAVES_NARROW
#define GG_LDS_WAVES_NARROW 4   // <= 32 x 32 channels: few MFMAs per offset, so more resident waves hide the dependent id -> row loads
#endif
#ifndef GG_LDS_WAVES
#define GG_LDS_WAVES 4   // waves per SIMD asked of the compiler for the <= 64-channel LDS variants (97+32 registers otherwise: 3)
#endif
// LDS-staged variant: the offset's weight slice (Q*NT KiB) is loaded ONCE per workgroup and offset
// into a double-buffered LDS image and read by the 4 waves with ds_read_b128, instead of every wave
// streaming it through L1 (4x less vector-memory traffic: with per-wave weight loads the CU's
// 64 B/clk L1 path, not the MFMA pipe, sets the pace - measured 59 % matrix-pipe utilisation).
// One barrier per offset; waves still skip the MFMAs of offsets without a neighbour in their rows.
template <int Q, int NT, int RT, bool VEC, bool DB = true, int BLK = SC_BLOCK, bool PFL = false, bool IDPF = false>
__global__ void __launch_bounds__(BLK, PFL ? 3 : (Q * NT <= 4 && RT <= 2) ? GG_LDS_WAVES_NARROW : (Q * NT * RT <= 32 && Q * NT <= 16) ? GG_LDS_WAVES : (BLK > SC_BLOCK ? GG_LDS_WAVES_WIDE : 1))
gather_gemm_lds_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ wp,
                       const int* __restrict__ nbr, int n_out, int K, int cp, const float* __restrict__ bias,
                       float* __restrict__ out, const int* __restrict__ order, double* __restrict__ stats) {
    constexpr int SLICE = Q * NT * 64;                    // float4 per offset
    constexpr int PER_THREAD = (SLICE + BLK - 1) / BLK;
    __shared__ f32x4 wl[DB ? 2 : 1][SLICE];  // DB = false: one 64 KiB buffer (128-channel layers), two barriers per offset
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
    bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        live[rt] = row0 + rt * 16 + r < n_out;
        rows[rt] = live[rt] ? (order ? order[row0 + rt * 16 + r] : row0 + rt * 16 + r) : n_out - 1;
    }

    // stage offset 0
#pragma unroll
    for (int t = 0; t < PER_THREAD; ++t) {
        const int e = t * BLK + threadIdx.x;
        if (e < SLICE) wl[0][e] = wp4[e];
    }
    __syncthreads();

    if constexpr (PFL) {
        // software-pipelined variant (experiment): neighbour ids two offsets ahead, gathered rows one offset ahead, in two
        // register sets that alternate (loop body written twice: no register copies - vector moves cost matrix time here)
        static_assert(DB, "pipelined variant needs the double-buffered weight slices");
        auto load_ids = [&](int k, int (&dst)[RT]) {
            const int kk = k < K ? k : K - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int v = nbr[(size_t)kk * n_out + rows[rt]];
                dst[rt] = (k < K && live[rt]) ? v : -1;
            }
        };
        auto body = [&](int k, const int (&s_cur)[RT], const f32x4 (&a_cur)[RT][Q], int (&s_ids)[RT], const int (&s_next)[RT], f32x4 (&a_next)[RT][Q]) {
            const int cur = k & 1;
            f32x4 stage[PER_THREAD];
            if (k + 1 < K) {
#pragma unroll
                for (int t = 0; t < PER_THREAD; ++t) {
                    const int e = t * BLK + threadIdx.x;
                    if (e < SLICE) stage[t] = wp4[(size_t)(k + 1) * SLICE + e];
                }
            }
            load_ids(k + 2, s_ids);                                   // overwrites the ids of offset k - 1 (dead)
            gather_rows<Q, RT, VEC>(in_rsrc, cg, g, s_next, a_next);   // rows of offset k + 1, in flight under the MFMAs below
            bool hit[RT];
            bool any = false;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                hit[rt] = __any(s_cur[rt] >= 0);
                any = any || hit[rt];
            }
            if (any) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    f32x4 b[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = wl[cur][(q * NT + n) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                if (hit[rt]) acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[rt][q][j], b[n][j], acc[rt][n], 0, 0, 0);
                }
            }
            if (k + 1 < K) {
#pragma unroll
                for (int t = 0; t < PER_THREAD; ++t) {
                    const int e = t * BLK + threadIdx.x;
                    if (e < SLICE) wl[cur ^ 1][e] = stage[t];
                }
            }
            __syncthreads();
        };
        int sA[RT], sB[RT], sC[RT];
        f32x4 aA[RT][Q], aB[RT][Q];
        load_ids(0, sA);
        load_ids(1, sB);
        gather_rows<Q, RT, VEC>(in_rsrc, cg, g, sA, aA);
        // ids rotate through three sets (period 3), rows through two (period 2): six offsets per trip
        int k = 0;
        while (true) {
            body(k, sA, aA, sC, sB, aB); if (++k >= K) break;      // cur ids A, next ids B, k+2 -> C
            body(k, sB, aB, sA, sC, aA); if (++k >= K) break;
            body(k, sC, aA, sB, sA, aB); if (++k >= K) break;
            body(k, sA, aB, sC, sB, aA); if (++k >= K) break;
            body(k, sB, aA, sA, sC, aB); if (++k >= K) break;
            body(k, sC, aB, sB, sA, aA); if (++k >= K) break;
        }
    } else {
    int id_next[RT];
    if constexpr (IDPF) {
    #pragma unroll
        for (int rt = 0; rt < RT; ++rt) id_next[rt] = __builtin_nontemporal_load(nbr + rows[rt]);
    }
    for (int k = 0; k < K; ++k) {
            const int cur = DB ? (k & 1) : 0;
            // this offset's neighbour ids FIRST in program order: vmcnt counts in issue order, so a wait for ids that were issued
            // behind the weight loads below would also wait for those (they are not needed before the end of the offset)
            int src[RT];
            if constexpr (IDPF) {
                // ids one offset ahead (<= 64-channel variants with two row tiles per wave): the id -> row -> MFMA chain of an offset
                // loses its first memory round trip.  Worth 1-4 % (32 -> 32 @ 682k rows 0.307 -> 0.303 ms, 32 -> 64 0.299 -> 0.288,
                // 64 -> 64 0.589 -> 0.584): the chain is not what holds the matrix pipe at 62-69 %
                const int kn = k + 1 < K ? k + 1 : K - 1;
    #pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    src[rt] = live[rt] ? id_next[rt] : -1;
                    id_next[rt] = __builtin_nontemporal_load(nbr + (size_t)kn * n_out + rows[rt]);
                }
            } else {
    #pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int v = __builtin_nontemporal_load(nbr + (size_t)k * n_out + rows[rt]);
                src[rt] = live[rt] ? v : -1;
            }
            }
            asm volatile("" ::: "memory");      // keep the id loads in front of the weight loads
            // next offset's weights: global -> registers now, registers -> LDS after this offset's math
            // (unconditional: on the last offset the slice of offset K - 1 is fetched again and dropped - a branch around the loads
            // makes the compiler's wait for the ids a wait for everything, see the note on wgrad_kernel's EXACT)
            f32x4 stage[PER_THREAD];
            {
                const int kn = k + 1 < K ? k + 1 : K - 1;
    #pragma unroll
                for (int t = 0; t < PER_THREAD; ++t) {
                    const int e = t * BLK + threadIdx.x;
                    if (SLICE % BLK == 0 || e < SLICE) stage[t] = wp4[(size_t)kn * SLICE + e];
                }
            }
            bool hit[RT];
            bool any = false;
    #pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                hit[rt] = __any(src[rt] >= 0);
                any = any || hit[rt];
            }
            if (any) {
                f32x4 a[RT][Q];
                gather_rows<Q, RT, VEC>(in_rsrc, cg, g, src, a);
    #pragma unroll
                for (int q = 0; q < Q; ++q) {
                    f32x4 b[NT];
    #pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = wl[cur][(q * NT + n) * 64 + lane];
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
    #pragma unroll
                        for (int n = 0; n < NT; ++n) {
    #pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                if (hit[rt])
                                    acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q][j], b[n][j], acc[rt][n], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (!DB) __syncthreads();  // everyone is done reading the single buffer
            if (k + 1 < K) {
    #pragma unroll
                for (int t = 0; t < PER_THREAD; ++t) {
                    const int e = t * BLK + threadIdx.x;
                    if (e < SLICE) wl[DB ? (cur ^ 1) : 0][e] = stage[t];
                }
            }
            __syncthreads();
        }
}
    if (row0 >= n_out && !stats) return;

    // BatchNorm statistics of the layer's output, taken from the accumulators (reference spconv_backbone.py:21-25: every conv
    // of post_act_block / SparseBasicBlock is followed by BatchNorm1d): per-channel sum and sum of squares of this workgroup's
    // rows -> stats scratch [2 cp][gridDim.x] behind the 2 cp results, folded in fixed order by fold_partials_kernel.  fp32
    // over the <= 8 values of a lane and the 4 lane groups, fp64 across waves and workgroups (as toda_rows_moments).
    if (stats) {
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
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < BLK / 64; ++w) a += (double)st_sh[w][qq][ch];
            stats[2 * cp + (size_t)(qq * cp + ch) * gridDim.x + blk] = a;
        }
    }

    const bool full = cp == 16 * NT;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pos = row0 + rt * 16 + 4 * g + reg;
            const int row = __shfl(rows[rt], 4 * g + reg, 64);  // lane r' holds the (possibly permuted) row of tile position r'
            if (pos >= n_out) continue;
            float* dst = out + (size_t)row * cp + NT * r;
            if (full) {
                if constexpr (NT == 1) {
                    dst[0] = acc[rt][0][reg];
                } else if constexpr (NT == 2) {
                    *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][reg], acc[rt][1][reg]);
                } else {
#pragma unroll
                    for (int n = 0; n < NT; n += 4)
                        *reinterpret_cast<f32x4*>(dst + n) =
                            f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (NT * r + n < cp) dst[n] = acc[rt][n][reg];
            }
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

// wgrad: dW[co][k][ci] = sum_o in[nbr[k][o]][ci] * dout[o][co].  Grid (row chunk, offset, channel
// sub-block).  The contraction runs over PAIRS, not rows: every wave reads 64 neighbour ids at a
// time (one coalesced 256-byte load), compacts the valid (in,out) pairs into a small LDS queue
// (ballot + prefix popcount) and feeds the MFMAs 16 pairs per round (4 per MFMA step, one per lane
// group), so no matrix work is spent on rows without a neighbour and each round has 4 steps of
// loads in flight.  The channel interleave of the forward kernel makes the operand loads
// MTB/NTB-wide vector loads.  Partial sums go to a slab per chunk (plain stores) and a second
// kernel adds the slabs in fixed order: deterministic, no float atomics.
#ifndef WG_WAVES
#define WG_WAVES 4
#endif
// COOP (128-channel sides): the 4 waves of a block walk the SAME pairs and each owns one (MTB x NTB)-tile quarter of
// the 8 x 8-tile weight block - operands of the quarters that share a row half hit in L1, accumulators stay at 64
// registers (4 waves / SIMD instead of 1 for a single 8 x 8 wave) and no cross-wave fold is needed.
typedef float f32x2w __attribute__((ext_vector_type(2)));
template <int MTB, int NTB, bool COOP = false, bool EXACT = false>
__global__ void __launch_bounds__(SC_BLOCK, (MTB * NTB <= 16) ? WG_WAVES : 1)
wgrad_kernel(const float* __restrict__ in, int n_in, int cin, const float* __restrict__ dout, int cout,
             const int* __restrict__ nbr, int n_out, int K, int rows_per_chunk, int MT, int NT, int nsub_n,
             float* __restrict__ slab, int xcd_chunks) {
    constexpr int QCAP = 64 + 16;
    __shared__ float red[MTB * NTB * 4 * 64];
    __shared__ int q_in[SC_BLOCK / 64][QCAP], q_out[SC_BLOCK / 64][QCAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ii = lane & 15, g = lane >> 4;
    // 1-D grid over (row chunk, offset, channel sub-block).  The K offset-blocks of a chunk read the same dout rows and neighbouring
    // input rows: they are given consecutive slots of ONE XCD (workgroup ids are dealt round-robin over the 8 XCDs, so XCD x owns
    // ids x, x + 8, ...) and run side by side out of that XCD's L2.  With the offset as the slow grid axis (round 1) the blocks
    // of a chunk were on one XCD too, but a whole grid pass apart in time: PMC 1.50 GB from HBM per launch of the 32 -> 32
    // layer at 682 k rows against 0.27 GB algorithmic.  xcd_chunks < 0 (TODA_WG_XCD=0): plain order.  The launch pads the chunk count to a multiple of 8.
    const int n_chunks = xcd_chunks < 0 ? -xcd_chunks : xcd_chunks;
    const int nsub_blk = (int)gridDim.x / (n_chunks * K);
    int chunk, k, sub_blk;
    {
        const int b = blockIdx.x;
        if (xcd_chunks > 0) {
            const int xcd = b & 7, t = (b >> 3) / K;
            k = (b >> 3) - t * K;
            sub_blk = t % nsub_blk;
            chunk = (t / nsub_blk) * 8 + xcd;
        } else {
            chunk = b % n_chunks;
            k = (b / n_chunks) % K;
            sub_blk = b / (n_chunks * K);
        }
    }
    const int sub = COOP ? wv : sub_blk;
    const int m0 = (sub / nsub_n) * MTB, n0 = (sub % nsub_n) * NTB;
    const int row_begin = chunk * rows_per_chunk;
    if (row_begin >= n_out) return;      // a padding chunk of the XCD-ordered grid (whole block, before any barrier)
    const int row_end = min(n_out, row_begin + rows_per_chunk);
    const bool exact_a = cin == 16 * MT, exact_b = cout == 16 * NT;
    int* qi = q_in[wv];
    int* qo = q_out[wv];

    f32x4 acc[MTB][NTB];
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cin * 4u);
    const __amdgpu_buffer_rsrc_t dout_rsrc = table_rsrc(dout, (unsigned)n_out * (unsigned)cout * 4u);
    // one round: 4 MFMA steps over queue entries [d, d+16); entries >= limit contribute zeros.
    // Operands come through bounds-checked buffer loads (out-of-range offset -> 0), so there is no
    // branch around any load and all 8 loads of a round are in flight together.
    // full_tag: the round holds 16 pairs (every round but a chunk's last): no per-step test around the MFMAs.  With the test the
    // compiler sinks the loads of steps 1-3 into the conditional blocks, next to their use: load -> wait -> 16 MFMAs four times per
    // round instead of eight loads in flight and 64 MFMAs behind them.
    auto round16 = [&](int d, int limit, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float a[4][MTB], b[4][NTB];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = d + 4 * t + g;
            const bool ok = FULL || p < limit;
            const int pc = ok ? p : d;  // any valid queue slot
            const unsigned ia = ((unsigned)qi[pc] * (unsigned)cin + (unsigned)(MT * ii + m0)) * 4u;
            const unsigned ib = ((unsigned)qo[pc] * (unsigned)cout + (unsigned)(NT * ii + n0)) * 4u;
            // EXACT (channel counts = 16 x tiles: every layer but conv_input): the load form of each side is chosen at
            // compile time (16-byte loads when the block holds >= 4 tiles of that side).  As a run-time branch the two load forms share destination registers and hipcc puts an
            // s_waitcnt vmcnt(3) in front of every 16-byte load: four loads in flight per round instead of eight.
            if constexpr (EXACT && MTB == 2) {          // 32 channels: the lane's two consecutive channels as one 8-byte load
                const f32x2w v = __builtin_bit_cast(f32x2w, __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, ok ? ia : OOB, 0, 0));
                a[t][0] = v[0], a[t][1] = v[1];
            } else if ((EXACT || exact_a) && MTB % 4 == 0) {
#pragma unroll
                for (int m4 = 0; m4 < MTB; m4 += 4) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? ia + 4u * m4 : OOB, 0, 0));
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[t][m4 + m] = v[m];
                }
            } else {
#pragma unroll
                for (int m = 0; m < MTB; ++m)
                    a[t][m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                  in_rsrc, (ok && (EXACT || MT * ii + m0 + m < cin)) ? ia + 4u * m : OOB, 0, 0));
            }
            if constexpr (EXACT && NTB == 2) {
                const f32x2w v = __builtin_bit_cast(f32x2w, __builtin_amdgcn_raw_buffer_load_b64(dout_rsrc, ok ? ib : OOB, 0, 0));
                b[t][0] = v[0], b[t][1] = v[1];
            } else if ((EXACT || exact_b) && NTB % 4 == 0) {
#pragma unroll
                for (int n4 = 0; n4 < NTB; n4 += 4) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dout_rsrc, ok ? ib + 4u * n4 : OOB, 0, 0));
#pragma unroll
                    for (int n = 0; n < 4; ++n) b[t][n4 + n] = v[n];
                }
            } else {
#pragma unroll
                for (int n = 0; n < NTB; ++n)
                    b[t][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                  dout_rsrc, (ok && (EXACT || NT * ii + n0 + n < cout)) ? ib + 4u * n : OOB, 0, 0));
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (FULL || d + 4 * t < limit) {  // wave-uniform
#pragma unroll
                for (int m = 0; m < MTB; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][m], b[t][n], acc[m][n], 0, 0, 0);
            }
        }
    };

    int qn = 0;  // wave-uniform queue length (< 16 between batches)
    // the neighbour ids of the NEXT 64 rows are requested before this batch's rounds run, so their latency is covered by those rounds
    constexpr int BSTEP = COOP ? 64 : SC_BLOCK;
    const int base0 = row_begin + (COOP ? 0 : wv * 64);
    int iv_next = nbr[(size_t)k * n_out + min(base0 + lane, n_out - 1)];  // clamped, unconditional
    for (int base = base0; base < row_end; base += BSTEP) {
        const int o = base + lane;
        const int iv = iv_next;
        iv_next = nbr[(size_t)k * n_out + min(o + BSTEP, n_out - 1)];
        const int i = o < row_end ? iv : -1;
        const unsigned long long vote = __ballot(i >= 0);
        if (vote == 0) continue;
        if (i >= 0) {
            const int pos = qn + __popcll(vote & ((1ull << lane) - 1));
            qi[pos] = i;
            qo[pos] = o;
        }
        qn += __popcll(vote);
        __builtin_amdgcn_wave_barrier();
        int done = 0;
        while (qn - done >= 16) {
            round16(done, qn, std::bool_constant<!COOP>{});     // (the cooperative 128 x 128 kernel measured 8 % slower without the per-step test)
            done += 16;
        }
        const int left = qn - done;
        if (done > 0 && left > 0) {  // move the tail (< 16 entries) to the front of the queue
            int ti = 0, to = 0;
            if (lane < left) {
                ti = qi[done + lane];
                to = qo[done + lane];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < left) {
                qi[lane] = ti;
                qo[lane] = to;
            }
        }
        qn = left;
        __builtin_amdgcn_wave_barrier();
    }
    if (qn > 0) round16(0, qn, std::false_type{});

    // fold the 4 waves of the block in fixed order 0+1+2+3
    for (int src = 1; !COOP && src < SC_BLOCK / 64; ++src) {
        if (wv == src) {
#pragma unroll
            for (int m = 0; m < MTB; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) red[((m * NTB + n) * 4 + reg) * 64 + lane] = acc[m][n][reg];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int m = 0; m < MTB; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[m][n][reg] += red[((m * NTB + n) * 4 + reg) * 64 + lane];
        }
        __syncthreads();
    }
    if (!COOP && wv != 0) return;
    // D: col = lane&15 -> cout tile column, row = 4*(lane>>4)+reg -> cin tile row
    float* dst = slab + (size_t)chunk * cout * K * cin;
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = MT * (4 * g + reg) + m0 + m;
                const int co = NT * ii + n0 + n;
                if (ci < cin && co < cout) dst[((size_t)co * K + k) * cin + ci] = acc[m][n][reg];
            }
}

__global__ void __launch_bounds__(SC_BLOCK)
wgrad_reduce_kernel(const float* __restrict__ slab, int chunks, long long elems, float* __restrict__ dw) {
    // the adds stay in chunk order (fixed summation order); the loads of eight chunks are in flight together - as one dependent
    // load per add this fold read 21 MB at 0.8 TB/s (26 us per 64 -> 64 layer)
    const long long e = (long long)blockIdx.x * SC_BLOCK + threadIdx.x;
    if (e >= elems) return;
    const float* p = slab + e;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(p + (size_t)(c + i) * elems);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; c < chunks; ++c) s += p[(size_t)c * elems];
    dw[e] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// Gather-GEMM for the NARROW K = 27 layers (<= 32 gathered channels, 16 produced channels per workgroup):
// conv_input 5 -> 16, SubM 16 -> 16 and its dgrad, the strided 16 -> 32 and its dgrad 32 -> 16.  On these levels a row has 1.7 -
// 3.9 of its 27 neighbours, so almost every (16-row tile, offset) of the output-stationary kernel holds at least one pair and is
// gathered and multiplied at 6 - 14 % occupancy (their time follows the number of vector-memory instructions, not the bytes).
// Here a wave owns 64 output rows and COMPACTS the rows that have a neighbour at the offset (ballot + prefix popcount into a
// small LDS queue): one 16-row tile per offset instead of four, every gathered row a real one; one coalesced id load per offset
// and 64 rows (all 27 issued up front).  The compacted tile's product is added to the wave's rows of an LDS accumulator
// (ds_add_f32; only the owning wave touches a row and its offsets come in order: deterministic).  All 27 weight slices of the
// workgroup's 16 output channels sit in LDS (27 / 54 KiB), filled once from the plain [cout][K][cin] tensor, so the offsets need no
// barrier; the gathers of offset k + CG_DEPTH are issued before the multiplies of offset k (the loop is fully unrolled: static
// ring positions, the compiler counts vmcnt).  blockIdx.y = block of 16 produced channels (contiguous: 64-byte row segments).
// ------------------------------------------------------------------------------------------------------------------
#ifndef TODA_CG_WAVES
#define TODA_CG_WAVES 4
#endif
#ifndef TODA_CG_DEPTH
#define TODA_CG_DEPTH 3
#endif
constexpr int CG_WAVES = TODA_CG_WAVES, CG_BLOCK = CG_WAVES * 64, CG_ROWS = CG_BLOCK, CG_K = 27, CG_DEPTH = TODA_CG_DEPTH;

template <int Q, bool VEC>
__global__ void __launch_bounds__(CG_BLOCK)
gather_gemm_compact_kernel(const float* __restrict__ in, int n_in, int cg, const float* __restrict__ w, int w_cout, int w_cin, int transpose,
                           int flip_k, const int* __restrict__ nbr, int n_out, int cp, const float* __restrict__ bias,
                           float* __restrict__ out, double* __restrict__ stats) {
    __shared__ f32x4 w_lds[CG_K * Q * 64];            // [k][q][lane]: B fragments of the 4 MFMA steps of channel group q
    __shared__ float acc_lds[CG_WAVES * 65 * 16];     // per wave 64 rows + one row that absorbs the empty queue slots
    __shared__ int q_id[CG_WAVES][CG_DEPTH + 1][64];
    __shared__ int q_row[CG_WAVES][CG_DEPTH + 1][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int co0 = blockIdx.y * 16;
    const int row0 = blockIdx.x * CG_ROWS + wv * 64;

    // ids of this lane's row for all 27 offsets: in flight while the weights are staged
    const __amdgpu_buffer_rsrc_t id_rsrc = table_rsrc(reinterpret_cast<const float*>(nbr), (unsigned)((size_t)CG_K * n_out * 4u));
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)cg * 4u);
    int ids[CG_K];
    const bool row_ok = row0 + lane < n_out;
#pragma unroll
    for (int k = 0; k < CG_K; ++k)
        ids[k] = __builtin_bit_cast(int, __builtin_amdgcn_raw_buffer_load_b32(id_rsrc, row_ok ? (unsigned)(((size_t)k * n_out + row0 + lane) * 4u) : OOB, 0, 0));

    // weights of the 16 produced channels co0 .. co0 + 15 -> LDS (one 16-byte fragment per (k, q, lane); all loads of a thread
    // are independent: in flight together); accumulator rows start at the bias
    {
        constexpr int FR = CG_K * Q * 64, IT = (FR + CG_BLOCK - 1) / CG_BLOCK;
        f32x4 frag[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = tid + it * CG_BLOCK;
            const int ln = e & 63, kq = e >> 6;
            const int q = kq % Q, k = kq / Q;
            const int g0 = 16 * q + 4 * (ln >> 4), pch = co0 + (ln & 15);
            const int kk = flip_k ? CG_K - 1 - k : k;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < FR) {
                if (!transpose) {
                    if (pch < w_cout) {
                        const float* src = w + ((size_t)pch * CG_K + kk) * w_cin + g0;
                        if (VEC && g0 + 3 < w_cin) {
                            v = *reinterpret_cast<const f32x4*>(src);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (g0 + j < w_cin) v[j] = src[j];
                        }
                    }
                } else if (pch < w_cin) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (g0 + j < w_cout) v[j] = w[((size_t)(g0 + j) * CG_K + kk) * w_cin + pch];
                }
            }
            frag[it] = v;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it)
            if (tid + it * CG_BLOCK < FR) w_lds[tid + it * CG_BLOCK] = frag[it];
    }
    {
        const float bv = (bias && co0 + (tid & 15) < cp) ? bias[co0 + (tid & 15)] : 0.f;
        for (int e = tid; e < CG_WAVES * 65 * 16; e += CG_BLOCK) acc_lds[e] = bv;      // CG_BLOCK is a multiple of 16: e & 15 == tid & 15
    }
    __syncthreads();

    // compaction of offset k into queue slot k % (CG_DEPTH + 1) and the gathers of its first tile.  The ring's gathers are inline
    // asm with hand-counted vmcnt (every prepare issues exactly LOADS of them, out of range = no memory access): left to hipcc, the
    // loop over a rare offset's further tiles makes every offset wait for vmcnt(0) and nothing is in flight across offsets.
    struct Stage {
        f32x4 a[Q];          // VEC: the asm loads' destinations
        float s[Q][4];       // !VEC: one destination per 4-byte load (an element of a vector register would be a copy made at issue)
        int cnt;
    };
    const unsigned long long in_addr = (unsigned long long)in;
    const u32x4 in_desc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)in_addr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(in_addr >> 32) & 0xFFFFu)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)n_in * (unsigned)cg * 4u)), 0x00020000u};
    constexpr int LOADS = VEC ? Q : 4 * Q;
    auto prepare = [&](int k, Stage& st) {
        // every lane writes one queue entry: its pair behind the pairs of the lower lanes, or an empty entry (id -1, the wave's
        // spare accumulator row) behind all pairs - no branch, and the slots past the count never hold an older offset's pairs
        const int i = row_ok ? ids[k] : -1;
        const unsigned long long vote = __ballot(i >= 0);
        int* qi = q_id[wv][k % (CG_DEPTH + 1)];
        int* qr = q_row[wv][k % (CG_DEPTH + 1)];
        st.cnt = __popcll(vote);
        const unsigned long long below = (1ull << lane) - 1;
        const int pos = i >= 0 ? __popcll(vote & below) : st.cnt + __popcll(~vote & below);
        qi[pos] = i;
        qr[pos] = i >= 0 ? lane : 64;
        __builtin_amdgcn_wave_barrier();
        const int src = qi[c];
        const bool ok = src >= 0;
        const unsigned base = (unsigned)src * (unsigned)cg * 4u;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if constexpr (VEC) {
                const unsigned off = (ok && 16 * q + 4 * g < cg) ? base + (unsigned)(16 * q + 4 * g) * 4u : OOB;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st.a[q]) : "v"(off), "s"(in_desc) : "memory");
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = (ok && 16 * q + 4 * g + j < cg) ? base + (unsigned)(16 * q + 4 * g + j) * 4u : OOB;
                    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(st.s[q][j]) : "v"(off), "s"(in_desc) : "memory");
                }
            }
        }
    };
    // product of one compacted tile (rows = queue slots 16 tt .. 16 tt + 15) added to the wave's accumulator rows
    auto multiply_add = [&](int k, int tt, int cnt, const f32x4 (&a)[Q]) {
        f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const f32x4 b = w_lds[(k * Q + q) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) r = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[j], r, 0, 0, 0);
        }
        // D: row 4 g + reg = queue slot, column c.  Plain read-modify-write: only this wave touches its rows, the rows of a tile are
        // distinct (an LDS float atomic is a 64-cycle instruction here: 120 of them per wave set the pace of the first version); the
        // empty slots all point at the spare row
        const i32x4 rows = *reinterpret_cast<const i32x4*>(q_row[wv][k % (CG_DEPTH + 1)] + 16 * tt + 4 * g);
        float* const acc_w = acc_lds + wv * 65 * 16 + c;
        float old[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) old[reg] = acc_w[rows[reg] * 16];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc_w[rows[reg] * 16] = old[reg] + r[reg];
        __builtin_amdgcn_wave_barrier();
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the 27 ids (the compiler has seen its own wait at their first use)
    float dummy = 0.f;
    Stage st[CG_DEPTH];
#pragma unroll
    for (int k = 0; k < CG_DEPTH; ++k) prepare(k, st[k]);
#pragma unroll
    for (int k = 0; k < CG_K; ++k) {
        Stage& ring = st[k % CG_DEPTH];
        // offset k's gathers have landed when at most the (CG_DEPTH - 1) x LOADS younger ones are outstanding (fewer near the end)
        if constexpr (VEC) {
            if constexpr (Q == 1)
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring.a[0]) : "n"((CG_DEPTH - 1) * LOADS));
            else
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ring.a[0]), "+v"(ring.a[1]) : "n"((CG_DEPTH - 1) * LOADS));
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring.s[q][0]), "+v"(ring.s[q][1]), "+v"(ring.s[q][2]), "+v"(ring.s[q][3]) : "n"((CG_DEPTH - 1) * LOADS));
        }
        Stage cur = ring;
        if constexpr (!VEC) {
#pragma unroll
            for (int q = 0; q < Q; ++q) cur.a[q] = f32x4{ring.s[q][0], ring.s[q][1], ring.s[q][2], ring.s[q][3]};
        }
        if (k + CG_DEPTH < CG_K) {
            prepare(k + CG_DEPTH, ring);
        } else {      // keep the count: dummy loads behind the last offset, all into ONE register that stays live up to the final wait
                      // (a destination the compiler considers dead would be handed to another value and overwritten when the load returns)
#pragma unroll
            for (int d = 0; d < LOADS; ++d)
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(dummy) : "v"(OOB), "s"(in_desc) : "memory");
        }
        multiply_add(k, 0, cur.cnt, cur.a);
        // more than 16 pairs at this offset (always the centre of a SubM table): the remaining tiles, not pipelined
        for (int tt = 1; 16 * tt < cur.cnt; ++tt) {
            const int src = q_id[wv][k % (CG_DEPTH + 1)][16 * tt + c];
            const bool ok = src >= 0;
            const unsigned base = (unsigned)src * (unsigned)cg * 4u;
            f32x4 a[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if constexpr (VEC) {
                    a[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (ok && 16 * q + 4 * g < cg) ? base + (unsigned)(16 * q + 4 * g) * 4u : OOB, 0, 0));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        a[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (ok && 16 * q + 4 * g + j < cg) ? base + (unsigned)(16 * q + 4 * g + j) * 4u : OOB, 0, 0));
                }
            }
            multiply_add(k, tt, cur.cnt, a);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory");      // the dummy loads
    __builtin_amdgcn_wave_barrier();
    if (stats) {
        // BatchNorm moments of the layer's output (reference spconv_backbone.py:21-25: the conv is followed by BatchNorm1d), taken
        // from the LDS accumulator: thread (row group rg, channel ch) sums 16 rows in fp32, 32 threads add the 16 groups in fp64 and
        // store the workgroup's partial sums -> stats scratch [2 cp][gridDim.x] behind the 2 cp results (fold: fold_partials_kernel or
        // toda_bn_finalize_partials).  Fixed order: deterministic.
        __shared__ float st_sh[2][16][16];
        __syncthreads();                              // every wave's rows are final
        static_assert(CG_BLOCK % 256 == 0, "one (row group, channel) pair per thread of each 256-thread slice");
        for (int base = 0; base < CG_ROWS; base += 256) {
            float sm = 0.f, sq = 0.f;
            if (tid < 256) {
                const int ch = tid & 15, rg = tid >> 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int R = base + rg * 16 + i;              // row of the workgroup
                    if (blockIdx.x * CG_ROWS + R < n_out) {
                        const float v = acc_lds[((R >> 6) * 65 + (R & 63)) * 16 + ch];
                        sm += v;
                        sq += v * v;
                    }
                }
                st_sh[0][rg][ch] = sm;
                st_sh[1][rg][ch] = sq;
            }
            __syncthreads();
            if (tid < 32) {
                const int qq = tid >> 4, ch = tid & 15;
                double a = 0.0;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) a += (double)st_sh[qq][rg][ch];
                if (co0 + ch < cp) {
                    double* dst = stats + 2 * cp + (size_t)(qq * cp + co0 + ch) * gridDim.x + blockIdx.x;
                    *dst = base == 0 ? a : *dst + a;
                }
            }
            __syncthreads();
        }
    }
    // the wave's 64 rows x 16 channels: LDS -> out (64-byte row segments at column co0)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 16 + (lane >> 2), c4 = (lane & 3) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&acc_lds[(wv * 65 + r) * 16 + c4]);
        const int row = row0 + r;
        if (row < n_out) {
            float* dst = out + (size_t)row * cp + co0 + c4;
            if (co0 + c4 + 3 < cp) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co0 + c4 + j < cp) dst[j] = v[j];
            }
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
static void wgrad_plan(int n_out, int k_vol, int cin, int cout, int* chunks, int* rows_per_chunk) {
    static const int env_chunks = getenv("TODA_WG_CHUNKS") ? atoi(getenv("TODA_WG_CHUNKS")) : 0;
    int max_chunks = env_chunks > 0 ? env_chunks : ((tiles_pow2(cin) * tiles_pow2(cout) >= 16) ? 48 : 144);
    if (env_chunks <= 0 && k_vol <= 9) max_chunks *= 2;
    int ch = k_vol <= 9 ? n_out / 1024 : n_out / 2048;
    if (ch < 1) ch = 1;
    if (ch > max_chunks) ch = max_chunks;
    // with the chunk's K offset-blocks side by side on one XCD (wgrad_kernel's grid order) the big levels take smaller chunks:
    // about 4 k rows each, at most 96 (64 -> 64 @ 389 k rows: 48 chunks 0.531 ms, 96 0.505, 192 0.509, 288 0.52; @ 117 k rows 48
    // chunks 0.187, 96 0.196)
    if (env_chunks <= 0 && k_vol > 9 && max_chunks == 48 && n_out / 4096 > 48) ch = n_out / 4096 < 96 ? n_out / 4096 : 96;
    if (ch >= 8) ch = (ch + 7) / 8 * 8;      // whole rounds of the 8 XCDs (55 chunks = 56 launched with one XCD a chunk short: 0.313 ms, 48: 0.304, 96: 0.296 on 64 -> 64 @ 227 k rows)
    int rpc = (n_out + ch - 1) / ch;
    rpc = (rpc + 15) / 16 * 16;
    if (rpc < 16) rpc = 16;
    *rows_per_chunk = rpc;
    *chunks = n_out > 0 ? (n_out + rpc - 1) / rpc : 1;
}

}  // namespace toda

using namespace toda;

extern "C" size_t toda_spconv_packed_weight_floats(int k_vol, int c_gather, int c_produce) {
    return (size_t)k_vol * tiles_pow2(c_gather) * tiles_pow2(c_produce) * 256;
}

extern "C" int toda_spconv_pack_weight(const float* w, int cout, int k_vol, int cin, int transpose, int flip_k,
                                       float* wp, void* stream) {
    TODA_CHECK_ARG(cout >= 1 && cin >= 1 && k_vol >= 1, "pack_weight: bad shape");
    const int cgather = transpose ? cout : cin, cprod = transpose ? cin : cout;
    TODA_CHECK_ARG(cgather <= 128 && cprod <= 128, "pack_weight: channels > 128 unsupported (gather %d, produce %d)",
                   cgather, cprod);
    const int Q = tiles_pow2(cgather), NT = tiles_pow2(cprod);
    const long long total = (long long)k_vol * Q * NT * 256;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cdiv(total, SC_BLOCK)), dim3(SC_BLOCK), 0, (hipStream_t)stream, w, cout,
                       k_vol, cin, transpose, flip_k, Q, NT, wp);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_spconv_pack_weights(int n, const float* const* w, const int32_t* cout, const int32_t* k_vol, const int32_t* cin,
                                        const int32_t* transpose, const int32_t* flip_k, float* const* wp, void* stream) {
    TODA_CHECK_ARG(n >= 0 && w && cout && k_vol && cin && transpose && flip_k && wp, "pack_weights: null argument");
    for (int base = 0; base < n; base += PACK_MAX_SEG) {
        const int m = n - base < PACK_MAX_SEG ? n - base : PACK_MAX_SEG;
        PackBatch b = {};
        long long most = 0;
        for (int i = 0; i < m; ++i) {
            const int s = base + i;
            TODA_CHECK_ARG(w[s] && wp[s] && cout[s] >= 1 && cin[s] >= 1 && k_vol[s] >= 1, "pack_weights: bad segment %d", s);
            const int cgather = transpose[s] ? cout[s] : cin[s], cprod = transpose[s] ? cin[s] : cout[s];
            TODA_CHECK_ARG(cgather <= 128 && cprod <= 128, "pack_weights: channels > 128 unsupported (segment %d)", s);
            b.w[i] = w[s], b.wp[i] = wp[s], b.cout[i] = cout[s], b.K[i] = k_vol[s], b.cin[i] = cin[s];
            b.transpose[i] = transpose[s] != 0, b.flip[i] = flip_k[s] != 0;
            b.Q[i] = (unsigned char)tiles_pow2(cgather), b.NT[i] = (unsigned char)tiles_pow2(cprod);
            const long long total = (long long)k_vol[s] * b.Q[i] * b.NT[i] * 256;
            if (total > most) most = total;
        }
        hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(cdiv(most, SC_BLOCK), m), dim3(SC_BLOCK), 0, (hipStream_t)stream, b);
        TODA_LAUNCH_CHECK();
    }
    return TODA_OK;
}

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

namespace toda {
__global__ void fold_partials_kernel(double* __restrict__ sums, int blocks, int cols);   // dense.hip
}
// The per-workgroup BatchNorm partials of a statistics launch are folded right away (fold_partials_kernel), or - for
// toda_spconv_gather_gemm_stats_partials - left for toda_bn_finalize_partials, which folds and finalises in one launch; the entry
// point then reports the number of partials per column through this (thread-local: the call is synchronous on the host) slot.
static thread_local int* g_stats_blocks_out = nullptr;
static void fold_or_defer(double* stats, int blocks, int c_produce, hipStream_t s) {
    if (g_stats_blocks_out) {
        *g_stats_blocks_out = blocks;
        return;
    }
    hipLaunchKernelGGL(toda::fold_partials_kernel, dim3(2 * c_produce), dim3(256), 0, s, stats, blocks, 2 * c_produce);
}

static int gather_gemm_impl(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr, int n_out, int k_vol,
                            int c_produce, const float* bias, float* out, const int32_t* order, double* stats, void* stream,
                            const unsigned char* cls_sorted = nullptr, const toda::GatherClasses* cls_table = nullptr);

// ---- residue-class row order for the data gradient of strided convolutions (see GatherClasses) ------------------------------
// Inside blocks of CLS_ROWS consecutive (spatially adjacent) rows the rows are regrouped by class with a stable counting sort:
// one workgroup per block, a thread owns CLS_ROWS / 256 consecutive rows.
namespace toda {
constexpr int CLS_ROWS = 8192, CLS_PER = CLS_ROWS / 256;
struct ClassGeom {
    int stride[3], pad[3];
};
__global__ void __launch_bounds__(256)
class_order_kernel(const int4* __restrict__ coords, int n, const ClassGeom gm, int32_t* __restrict__ order, unsigned char* __restrict__ cls_sorted) {
    __shared__ int cnt[8][256 + 1];
    const int base = blockIdx.x * CLS_ROWS, t = threadIdx.x;
    unsigned char c[CLS_PER];
    int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < CLS_PER; ++i) {
        const int r = base + t * CLS_PER + i;
        int cl = 0;
        if (r < n) {
            const int4 v = coords[r];            // (batch, z, y, x)
            cl = (((v.y + gm.pad[0]) % gm.stride[0]) * gm.stride[1] + (v.z + gm.pad[1]) % gm.stride[1]) * gm.stride[2] + (v.w + gm.pad[2]) % gm.stride[2];
            ++mine[cl & 7];
        }
        c[i] = (unsigned char)cl;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt[k][t] = mine[k];
    __syncthreads();
    if (t < 8) {          // exclusive scan of class t over the 256 threads (serial: 256 adds, once per 8192 rows)
        int run = 0;
        for (int j = 0; j < 256; ++j) {
            const int v = cnt[t][j];
            cnt[t][j] = run;
            run += v;
        }
        cnt[t][256] = run;
    }
    __syncthreads();
    int cls_base[8], run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cls_base[k] = run + cnt[k][t];
        run += cnt[k][256];
    }
#pragma unroll
    for (int i = 0; i < CLS_PER; ++i) {
        const int r = base + t * CLS_PER + i;
        if (r >= n) break;
        const int k = c[i] & 7;
        int pos = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q == k) pos = cls_base[q]++;
        order[base + pos] = r;
        cls_sorted[base + pos] = c[i];
    }
}

// candidate lists of every class, ascending in the table's offset index (kz, ky, kx row-major)
static int class_table(const int32_t* ksize, const int32_t* stride, const int32_t* pad, GatherClasses* out) {
    const int n_class = stride[0] * stride[1] * stride[2];
    if (n_class < 1 || n_class > 8 || ksize[0] * ksize[1] * ksize[2] > 27) return -1;
    *out = GatherClasses{};
    for (int rz = 0; rz < stride[0]; ++rz)
        for (int ry = 0; ry < stride[1]; ++ry)
            for (int rx = 0; rx < stride[2]; ++rx) {
                const int c = (rz * stride[1] + ry) * stride[2] + rx;
                int m = 0;
                for (int kz = 0; kz < ksize[0]; ++kz)
                    for (int ky = 0; ky < ksize[1]; ++ky)
                        for (int kx = 0; kx < ksize[2]; ++kx)
                            if (kz % stride[0] == rz && ky % stride[1] == ry && kx % stride[2] == rx)
                                out->k[c][m++] = (unsigned char)((kz * ksize[1] + ky) * ksize[2] + kx);
                out->count[c] = (unsigned char)m;
            }
    return 0;
}
}  // namespace toda

extern "C" int toda_rulebook_class_order(const int32_t* in_coords, int n_in, const int32_t* stride, const int32_t* padding, int32_t* order,
                                         unsigned char* cls_sorted, void* stream) {
    TODA_CHECK_ARG(in_coords && stride && padding && order && cls_sorted && n_in >= 0, "rulebook_class_order: null argument");
    ClassGeom gm;
    for (int a = 0; a < 3; ++a) {
        TODA_CHECK_ARG(stride[a] >= 1 && stride[a] <= 2 && padding[a] >= 0, "rulebook_class_order: stride 1 or 2 per axis (got %d)", stride[a]);
        gm.stride[a] = stride[a], gm.pad[a] = padding[a];
    }
    if (n_in == 0) return TODA_OK;
    hipLaunchKernelGGL(class_order_kernel, dim3(cdiv(n_in, CLS_ROWS)), dim3(256), 0, (hipStream_t)stream, (const int4*)in_coords, n_in, gm, order,
                       cls_sorted);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_spconv_gather_gemm_classed(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr, int n_out, int k_vol,
                                               int c_produce, const float* bias, float* out, const int32_t* order,
                                               const unsigned char* cls_sorted, const int32_t* ksize, const int32_t* stride,
                                               const int32_t* padding, void* stream) {
    TODA_CHECK_ARG(order && cls_sorted && ksize && stride && padding, "gather_gemm_classed: null argument");
    TODA_CHECK_ARG(ksize[0] * ksize[1] * ksize[2] == k_vol, "gather_gemm_classed: kernel size does not match the table's %d offsets", k_vol);
    GatherClasses table;
    TODA_CHECK_ARG(class_table(ksize, stride, padding, &table) == 0, "gather_gemm_classed: needs <= 8 residue classes and <= 27 offsets");
    return gather_gemm_impl(in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, nullptr, stream, cls_sorted, &table);
}

// ---- optional per-launch timestamps of the gather-GEMM kernels -------------------------------
// hipExtLaunchKernelGGL stamps a start / stop event pair on the kernel dispatch itself, so the
// elapsed time is the kernel's own duration (what rocprofv3 --kernel-trace reports).  Events
// recorded around a launch with hipEventRecord also time the cache write-back their release
// fence triggers (measured +60 us on the 64->64 level).  Used by bench.py for `roofline`.
namespace toda {
struct LaunchTimer {
    std::vector<hipEvent_t> ev;   // start0, stop0, start1, stop1, ...
    int used = 0;
    bool on = false;
};
// The one piece of process-wide state of the library (bench instrumentation only; off unless toda_timing_begin was called):
// guarded by a mutex so that launches from several host threads each get their own event pair.  Durations are only
// meaningful for launches that went to ONE stream between begin and end (include/toda.h says so).
static LaunchTimer g_timer;
static std::mutex g_timer_mu;
static inline void timer_next(hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!g_timer.on) return;      // unlocked fast path: the flag only changes in toda_timing_begin / _end
    std::lock_guard<std::mutex> lock(g_timer_mu);
    if (!g_timer.on || 2 * (g_timer.used + 1) > (int)g_timer.ev.size()) return;
    *start = g_timer.ev[2 * g_timer.used];
    *stop = g_timer.ev[2 * g_timer.used + 1];
    g_timer.used++;
}
}  // namespace toda

#define GG_LAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
    do {                                                                                               \
        hipEvent_t t0_, t1_;                                                                           \
        toda::timer_next(&t0_, &t1_);                                                                  \
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, t0_, t1_, 0, __VA_ARGS__);           \
    } while (0)

extern "C" int toda_timing_begin(int capacity) {
    TODA_CHECK_ARG(capacity > 0 && capacity <= (1 << 20), "timing_begin: capacity in (0, 2^20]");
    std::lock_guard<std::mutex> lock(toda::g_timer_mu);
    for (hipEvent_t e : g_timer.ev) (void)hipEventDestroy(e);
    g_timer.ev.assign((size_t)2 * capacity, nullptr);
    for (auto& e : g_timer.ev) TODA_HIP(hipEventCreate(&e));
    g_timer.used = 0;
    g_timer.on = true;
    return TODA_OK;
}

extern "C" int toda_timing_end(float* ms_out, int cap, int* n_out) {
    std::lock_guard<std::mutex> lock(toda::g_timer_mu);
    g_timer.on = false;
    const int n = g_timer.used < cap ? g_timer.used : cap;
    for (int i = 0; i < n; ++i) {
        TODA_HIP(hipEventSynchronize(g_timer.ev[2 * i + 1]));
        TODA_HIP(hipEventElapsedTime(&ms_out[i], g_timer.ev[2 * i], g_timer.ev[2 * i + 1]));
    }
    if (n_out) *n_out = g_timer.used;
    for (hipEvent_t e : g_timer.ev) (void)hipEventDestroy(e);
    g_timer.ev.clear();
    g_timer.used = 0;
    return TODA_OK;
}

namespace toda {
__global__ void fold_partials_kernel(double* __restrict__ sums, int blocks, int cols);   // dense.hip
static bool gg_stats_supported(int c_gather, int c_produce) {
    const int Q = tiles_pow2(c_gather), NT = tiles_pow2(c_produce);
    return (c_gather & 3) == 0 && Q >= 2 && NT >= Q && Q * NT <= 32 && !(Q == 8 && NT == 8) && c_produce % 4 == 0 && c_produce == 16 * NT;
}
}  // namespace toda



extern "C" int toda_spconv_gather_gemm_ordered(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr,
                                               int n_out, int k_vol, int c_produce, const float* bias, float* out,
                                               const int32_t* order, void* stream) {
    return gather_gemm_impl(in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, nullptr, stream);
}

extern "C" int toda_spconv_gather_gemm_stats_supported(int c_gather, int c_produce) {
    return gg_stats_supported(c_gather, c_produce) ? 1 : 0;
}

extern "C" size_t toda_spconv_gather_gemm_stats_doubles(int n_out, int c_produce) {
    return (size_t)2 * c_produce * (1 + (size_t)cdiv(n_out > 0 ? n_out : 1, 64));     // 2 c results + [2 c][workgroups] scratch
}

extern "C" int toda_spconv_gather_gemm_stats(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr,
                                             int n_out, int k_vol, int c_produce, const float* bias, float* out, double* sums,
                                             size_t sums_doubles, void* stream) {
    TODA_CHECK_ARG(sums != nullptr && gg_stats_supported(c_gather, c_produce) && n_out > 0 && n_in > 0,
                   "gather_gemm_stats: unsupported channel pair (%d -> %d) or empty table", c_gather, c_produce);
    TODA_CHECK_ARG(sums_doubles >= toda_spconv_gather_gemm_stats_doubles(n_out, c_produce), "gather_gemm_stats: statistics buffer too small");
    return gather_gemm_impl(in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, nullptr, sums, stream);
}

// The same launch without the fold: *blocks_out (host) = partial sums per column in the scratch behind the 2 c result slots, for
// toda_bn_finalize_partials (one launch folds them in the same fixed order and finalises the statistics).
extern "C" int toda_spconv_gather_gemm_stats_partials(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr,
                                                      int n_out, int k_vol, int c_produce, const float* bias, float* out, double* sums,
                                                      size_t sums_doubles, int* blocks_out, void* stream) {
    TODA_CHECK_ARG(sums != nullptr && blocks_out != nullptr && gg_stats_supported(c_gather, c_produce) && n_out > 0 && n_in > 0,
                   "gather_gemm_stats_partials: unsupported channel pair (%d -> %d), empty table or null argument", c_gather, c_produce);
    TODA_CHECK_ARG(sums_doubles >= toda_spconv_gather_gemm_stats_doubles(n_out, c_produce), "gather_gemm_stats_partials: statistics buffer too small");
    *blocks_out = 0;
    g_stats_blocks_out = blocks_out;
    const int rc = gather_gemm_impl(in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, nullptr, sums, stream);
    g_stats_blocks_out = nullptr;
    if (rc == TODA_OK && *blocks_out <= 0) {
        set_error("gather_gemm_stats_partials: the launch took no statistics");
        return TODA_EINVAL;
    }
    return rc;
}

static thread_local bool g_subm_table = false;      // set by toda_spconv_gather_gemm_subm around its launch

static int gather_gemm_impl(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr, int n_out, int k_vol,
                            int c_produce, const float* bias, float* out, const int32_t* order, double* stats, void* stream,
                            const unsigned char* cls_sorted, const toda::GatherClasses* cls_table) {
    toda::GatherClasses classes = {};
    if (cls_sorted && cls_table) classes = *cls_table;
    else cls_sorted = nullptr;
    TODA_CHECK_ARG(c_gather >= 1 && c_gather <= 128 && c_produce >= 1 && c_produce <= 128,
                   "gather_gemm: channels must be in [1,128] (gather %d, produce %d)", c_gather, c_produce);
    TODA_CHECK_ARG(n_out >= 0 && n_in >= 0 && k_vol >= 1, "gather_gemm: bad sizes");
    TODA_CHECK_ARG((unsigned long long)n_in * c_gather * 4ull < 0xFFFFFFF0ull, "gather_gemm: gathered table must be < 4 GiB");
    if (n_out == 0) return TODA_OK;
    if (n_in == 0) {  // nothing to gather: bias only
        TODA_CHECK_ARG(bias == nullptr, "gather_gemm: empty input with bias is unsupported");
        TODA_HIP(hipMemsetAsync(out, 0, (size_t)n_out * c_produce * sizeof(float), (hipStream_t)stream));
        return TODA_OK;
    }
    const int Q = tiles_pow2(c_gather), NT = tiles_pow2(c_produce);
    hipStream_t s = (hipStream_t)stream;
    static const bool chunk_set = [] {
        if (const char* e = getenv("TODA_GG_XCD_CHUNK")) {
            const int v = atoi(e);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(toda::d_xcd_chunk), &v, sizeof(int));
        }
        return true;
    }();
    (void)chunk_set;
    // tuning knobs for experiments: TODA_GG_RT in {1,2,4} (0 = built-in choice), TODA_GG_PF in {0,1}
    static const int env_rt = getenv("TODA_GG_RT") ? atoi(getenv("TODA_GG_RT")) : 0;
    static const int env_pf = getenv("TODA_GG_PF") ? atoi(getenv("TODA_GG_PF")) : 0;
    static const int env_xcd = getenv("TODA_GG_XCD") ? atoi(getenv("TODA_GG_XCD")) : 0;  // measured 6-8 % slower: z-slabs differ in density, round-robin balances better
    // weight staging: 1 = auto (LDS-shared slice when it measured faster: Q >= 2 and NT >= Q), 0 = never, 2 = always
    static const int env_lds_raw = getenv("TODA_GG_LDS") ? atoi(getenv("TODA_GG_LDS")) : 1;
    const int env_lds = env_lds_raw == 2 ? 1 : (env_lds_raw == 1 ? (Q >= 2 && NT >= Q) : 0);
    const bool vec_ok = (c_gather & 3) == 0;
    const int env_lds88 = getenv("TODA_GG_LDS88") ? atoi(getenv("TODA_GG_LDS88")) : 3;      // (read per call: tests and A/B runs flip it inside one process)  // 1: RT=1 single-buffer LDS (0.67 ms), 2: RT=2 (0.76), 0: registers-only RT=2 (0.70) on 97.5k x 27 x 128 x 128
    if (env_lds88 && vec_ok && Q == 8 && NT == 8 && !cls_sorted) {
        if (env_lds88 == 2)
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<8, 8, 2, true, false>), dim3(cdiv(cdiv(n_out, 32), SC_BLOCK / 64)),
                               dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        else if (env_lds88 == 3)
            GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<8, 8, 1, true, false, 512>), dim3(cdiv(cdiv(n_out, 16), 8)),
                               dim3(512), 0, s, in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
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
    // OFF by default: measured 0.502 ms against 0.498 ms for the per-offset barrier kernel on the 389.5k-row level - the
    // barrier imbalance it removes is not what holds the matrix pipe at 70 %.
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
    // OFF by default: measured 0.653 ms against 0.539 ms for gather_gemm_lds_kernel on the 389.5k-row stride-4 level (0.245 vs
    // 0.207 ms at 117k rows).  The consumers' MFMA stream is clean (about 2,950 cycles per offset and 128 rows), but the
    // producers' LDS-DMA row gather runs at 16-30 GB/s per CU out of a 100 MB feature table (MI355X_MICROARCH.md "Indexed
    // rows: gather into LDS"), i.e. about 4,800 cycles for the 32 KiB of an offset: the design is gather-bound, three stages
    // deep or not.  The register gathers of the 16 resident waves of gather_gemm_lds_kernel re-hit L1 for the rows that
    // neighbouring offsets share and keep more requests in flight.
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
    static const int env_wres = getenv("TODA_GG_WRES") ? atoi(getenv("TODA_GG_WRES")) : 0;   // measured slower than the per-offset LDS slices (32->32 @ 682k rows 0.356 vs 0.321 ms, 16->32 0.234 vs 0.168): off
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
    // submanifold tables (toda_spconv_gather_gemm_subm): x-run operand reuse
    if (g_subm_table && k_vol == 27 && vec_ok && !cls_sorted && order == nullptr && Q == NT && (Q == 2 || Q == 4) && c_gather == 16 * Q &&
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
    static const int env_pfl = getenv("TODA_GG_LDS_PF") ? atoi(getenv("TODA_GG_LDS_PF")) : 0;   // 64 -> 64: register-pipelined gathers (experiment)
    static const int env_blk512 = getenv("TODA_GG_BLK512") ? atoi(getenv("TODA_GG_BLK512")) : 0;   // 64 -> 64: 8 waves share a weight slice (experiment)
    if ((env_lds || stats) && !cls_sorted && vec_ok && Q * NT <= 32) {  // weight slice <= 32 KiB per buffer
#define GL(QQ, NN, RR)                                                                                                   \
    if (env_blk512 && QQ == 4 && NN == 4 && RR == 2)                                                                     \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<4, 4, 2, true, true, 512, false, true>),                        \
                  dim3(cdiv(cdiv(n_out, 16 * RR), 8)), dim3(512), 0, s, in, n_in, c_gather, wp, nbr, n_out,               \
                  k_vol, c_produce, bias, out, order, stats);                                                            \
    else if (env_pfl && QQ == 4 && NN == 4 && RR == 2)                                                                   \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<4, 4, 2, true, true, SC_BLOCK, true>),                          \
                  dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out, \
                  k_vol, c_produce, bias, out, order, stats);                                                            \
    else                                                                                                                 \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<QQ, NN, RR, true, true, SC_BLOCK, false, (QQ <= 4 && NN <= 4 && RR == 2)>), \
                       dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out, \
                       k_vol, c_produce, bias, out, order, stats)
#define GL_RT(QQ, NN)                          \
    if (env_rt == 4 && QQ <= 4 && NN <= 4) {   \
        GL(QQ, NN, 4);                         \
    } else if (env_rt == 1) {                  \
        GL(QQ, NN, 1);                         \
    } else {                                   \
        GL(QQ, NN, 2);                         \
    }
#define GL_ROW(QQ)                 \
    switch (NT) {                  \
        case 1: GL_RT(QQ, 1); break; \
        case 2: GL_RT(QQ, 2); break; \
        case 4: GL_RT(QQ, 4); break; \
        default: GL(QQ, 8, 1); break; \
    }
        switch (Q) {
            case 1: GL_ROW(1); break;
            case 2: GL_ROW(2); break;
            case 4: GL_ROW(4); break;
            default: GL_ROW(8); break;
        }
#undef GL_ROW
#undef GL_RT
#undef GL
        TODA_LAUNCH_CHECK();
        if (stats) {      // fold the per-workgroup partial sums of the launch above (same grid arithmetic as GL_RT / GL_ROW)
            const int rr = NT >= 8 ? 1 : ((env_rt == 4 && Q <= 4 && NT <= 4) ? 4 : (env_rt == 1 ? 1 : 2));
            const int blocks = cdiv(cdiv(n_out, 16 * rr), (env_blk512 && Q == 4 && NT == 4 && rr == 2) ? 8 : SC_BLOCK / 64);
            fold_or_defer(stats, blocks, c_produce, s);
            TODA_LAUNCH_CHECK();
        }
        return TODA_OK;
    }
    TODA_CHECK_ARG(stats == nullptr, "gather_gemm: statistics requested on a launch shape without the fused epilogue");
    int rt_sel = env_rt ? env_rt : ((NT >= 8 && Q < 8) ? 1 : 2);  // 128->128: RT = 2 measured 16 % faster than 1
    if (NT >= 8 && rt_sel > 2) rt_sel = 2;
    if (Q >= 8 && rt_sel > 2) rt_sel = 2;
#define GG(QQ, NN, RR, PP)                                                                                            \
    GGV(QQ, NN, RR, PP, true)
#define GGV(QQ, NN, RR, PP, VV)                                                                                       \
    if (cls_sorted && !(PP) && (VV))                                                                                  \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_kernel<QQ, NN, RR, false, true, true>),                                 \
                  dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, \
                  n_out, k_vol, c_produce, bias, out, env_xcd, order, cls_sorted, classes);                           \
    else                                                                                                              \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_kernel<QQ, NN, RR, PP, VV>),                                       \
                       dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, \
                       n_out, k_vol, c_produce, bias, out, env_xcd, order, nullptr, classes)
#define GG_PF(QQ, NN, RR)        \
    if (env_pf) {                \
        GG(QQ, NN, RR, true);    \
    } else {                     \
        GG(QQ, NN, RR, false);   \
    }
#define GG_RT(QQ, NN)                                  \
    if (rt_sel == 1) {                                 \
        GG_PF(QQ, NN, 1)                               \
    } else if (rt_sel == 2 || QQ >= 8 || NN >= 8) {    \
        GG_PF(QQ, NN, 2)                               \
    } else {                                           \
        GG_PF(QQ, NN, 4)                               \
    }
#define GG_ROW(QQ)                 \
    switch (NT) {                  \
        case 1: GG_RT(QQ, 1); break; \
        case 2: GG_RT(QQ, 2); break; \
        case 4: GG_RT(QQ, 4); break; \
        default: GG_RT(QQ, 8); break; \
    }
    if (!vec_ok) {  // rows not 16-byte aligned (e.g. 5 point features): dword gathers, plain variant
        TODA_CHECK_ARG(Q == 1, "gather_gemm: gathered channel counts above 16 must be multiples of 4 (got %d)", c_gather);
        switch (NT) {
            case 1: GGV(1, 1, 2, false, false); break;
            case 2: GGV(1, 2, 2, false, false); break;
            case 4: GGV(1, 4, 2, false, false); break;
            default: GGV(1, 8, 1, false, false); break;
        }
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    switch (Q) {
        case 1: GG_ROW(1); break;
        case 2: GG_ROW(2); break;
        case 4: GG_ROW(4); break;
        default: GG_ROW(8); break;
    }
#undef GG_ROW
#undef GG_RT
#undef GG_PF
#undef GGV
#undef GG
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

// Gather-GEMM over the table of a SUBMANIFOLD convolution (toda_rulebook_subm, 3 x 3 x 3: nbr[12] / nbr[14] are a row's x-neighbours):
// the 64 -> 64 and 32 -> 32 layers take gather_gemm_line_kernel, every other shape the kernels of toda_spconv_gather_gemm.  sums
// (nullable): BatchNorm moments as toda_spconv_gather_gemm_stats; blocks_out (nullable, host): leave the partial sums unfolded as
// toda_spconv_gather_gemm_stats_partials.  The data gradient of a submanifold layer runs over the same table (reversed weights).
extern "C" int toda_spconv_gather_gemm_subm(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr, int n_out,
                                            int k_vol, int c_produce, const float* bias, float* out, double* sums, size_t sums_doubles,
                                            int* blocks_out, void* stream) {
    if (sums) {
        TODA_CHECK_ARG(gg_stats_supported(c_gather, c_produce) && n_out > 0 && n_in > 0,
                       "gather_gemm_subm: statistics on an unsupported channel pair (%d -> %d) or an empty table", c_gather, c_produce);
        TODA_CHECK_ARG(sums_doubles >= toda_spconv_gather_gemm_stats_doubles(n_out, c_produce), "gather_gemm_subm: statistics buffer too small");
    }
    if (blocks_out) *blocks_out = 0;
    g_subm_table = true;
    g_stats_blocks_out = sums ? blocks_out : nullptr;
    const int rc = gather_gemm_impl(in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, nullptr, sums, stream);
    g_stats_blocks_out = nullptr;
    g_subm_table = false;
    return rc;
}

extern "C" int toda_spconv_gather_gemm(const float* in, int n_in, int c_gather, const float* wp, const int32_t* nbr,
                                       int n_out, int k_vol, int c_produce, const float* bias, float* out,
                                       void* stream) {
    return toda_spconv_gather_gemm_ordered(in, n_in, c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, nullptr, stream);
}

// Narrow K = 27 layers by per-offset compaction (gather_gemm_compact_kernel).  w is the PLAIN weight [w_cout][27][w_cin];
// transpose / flip_k select the data-gradient operand exactly as toda_spconv_pack_weight would pack it.
extern "C" int toda_spconv_gather_gemm_compact_supported(int c_gather, int c_produce, int k_vol) {
    return k_vol == CG_K && c_gather >= 1 && c_gather <= 32 && c_produce >= 1 && c_produce <= 32 && c_produce % 4 == 0;
}

static int gather_gemm_compact_impl(const float* in, int n_in, int c_gather, const float* w, int w_cout, int w_cin, int transpose,
                                    int flip_k, const int32_t* nbr, int n_out, int k_vol, int c_produce, const float* bias,
                                    float* out, double* stats, void* stream) {
    TODA_CHECK_ARG(toda_spconv_gather_gemm_compact_supported(c_gather, c_produce, k_vol),
                   "gather_gemm_compact: needs K = 27, <= 32 gathered and <= 32 produced channels (a multiple of 4) (got K %d, %d -> %d)", k_vol, c_gather,
                   c_produce);
    TODA_CHECK_ARG((transpose ? w_cout : w_cin) == c_gather && (transpose ? w_cin : w_cout) == c_produce,
                   "gather_gemm_compact: weight [%d][27][%d] (transpose %d) does not map %d -> %d channels", w_cout, w_cin, transpose, c_gather, c_produce);
    TODA_CHECK_ARG(n_out >= 0 && n_in >= 0, "gather_gemm_compact: bad sizes");
    TODA_CHECK_ARG((unsigned long long)n_in * c_gather * 4ull < 0xFFFFFFF0ull && (unsigned long long)n_out * CG_K * 4ull < 0xFFFFFFF0ull,
                   "gather_gemm_compact: gathered table and neighbour table must be < 4 GiB each");
    if (n_out == 0) return TODA_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(n_out, CG_ROWS), cdiv(c_produce, 16));
    const bool vec = c_gather % 4 == 0;
#define CG_LAUNCH(QQ, VV)                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gather_gemm_compact_kernel<QQ, VV>), grid, dim3(CG_BLOCK), 0, s, in, n_in, c_gather, w, w_cout, w_cin, \
                       transpose, flip_k, nbr, n_out, c_produce, bias, out, stats)
    if (c_gather <= 16) {
        if (vec) CG_LAUNCH(1, true);
        else CG_LAUNCH(1, false);
    } else {
        if (vec) CG_LAUNCH(2, true);
        else CG_LAUNCH(2, false);
    }
#undef CG_LAUNCH
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

extern "C" int toda_spconv_gather_gemm_compact(const float* in, int n_in, int c_gather, const float* w, int w_cout, int w_cin, int transpose,
                                               int flip_k, const int32_t* nbr, int n_out, int k_vol, int c_produce, const float* bias,
                                               float* out, void* stream) {
    return gather_gemm_compact_impl(in, n_in, c_gather, w, w_cout, w_cin, transpose, flip_k, nbr, n_out, k_vol, c_produce, bias, out, nullptr, stream);
}

// The same launch with the BatchNorm moments of its output from the epilogue (per-channel sum and sum of squares, fp64, in the layout of
// toda_spconv_gather_gemm_stats: 2 c results + [2 c][workgroups] scratch).  blocks_out == NULL: folded right away; else the partials
// stay unfolded and *blocks_out (host) = partials per column, for toda_bn_finalize_partials.
extern "C" size_t toda_spconv_gather_gemm_compact_stats_doubles(int n_out, int c_produce) {
    return (size_t)2 * c_produce * (1 + (size_t)cdiv(n_out > 0 ? n_out : 1, CG_ROWS));
}

extern "C" int toda_spconv_gather_gemm_compact_stats(const float* in, int n_in, int c_gather, const float* w, int w_cout, int w_cin,
                                                     int transpose, int flip_k, const int32_t* nbr, int n_out, int k_vol, int c_produce,
                                                     const float* bias, float* out, double* sums, size_t sums_doubles, int* blocks_out,
                                                     void* stream) {
    TODA_CHECK_ARG(sums != nullptr && n_out > 0, "gather_gemm_compact_stats: null statistics buffer or no output rows");
    TODA_CHECK_ARG(sums_doubles >= toda_spconv_gather_gemm_compact_stats_doubles(n_out, c_produce), "gather_gemm_compact_stats: statistics buffer too small");
    const int rc = gather_gemm_compact_impl(in, n_in, c_gather, w, w_cout, w_cin, transpose, flip_k, nbr, n_out, k_vol, c_produce, bias, out, sums, stream);
    if (rc != TODA_OK) return rc;
    const int blocks = cdiv(n_out, CG_ROWS);
    if (blocks_out) {
        *blocks_out = blocks;
        return TODA_OK;
    }
    hipLaunchKernelGGL(toda::fold_partials_kernel, dim3(2 * c_produce), dim3(256), 0, (hipStream_t)stream, sums, blocks, 2 * c_produce);
    TODA_LAUNCH_CHECK();
    return TODA_OK;
}

// SubM gather-GEMM over a halo plan (toda_halo_plan_build): forward, and - with the transposed / offset-reversed packed operand - the
// data gradient of the same table.  sums (nullable): BatchNorm moments from the epilogue, as toda_spconv_gather_gemm_stats.
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

extern "C" size_t toda_spconv_wgrad_workspace_bytes(int n_out, int k_vol, int cin, int cout) {
    int chunks, rpc;
    wgrad_plan(n_out, k_vol, cin, cout, &chunks, &rpc);
    return align_up((size_t)chunks * k_vol * cin * cout * sizeof(float), 256);
}

extern "C" int toda_spconv_wgrad(const float* in, int n_in, const float* dout, const int32_t* nbr, int n_out, int k_vol,
                                 int cin, int cout, float* dw, void* ws, size_t ws_bytes, void* stream) {
    TODA_CHECK_ARG(cin >= 1 && cin <= 128 && cout >= 1 && cout <= 128, "wgrad: channels must be in [1,128]");
    TODA_CHECK_ARG(n_out >= 0 && n_in >= 0 && k_vol >= 1, "wgrad: bad sizes");
    TODA_CHECK_ARG((unsigned long long)n_in * cin * 4ull < 0xFFFFFFF0ull && (unsigned long long)n_out * cout * 4ull < 0xFFFFFFF0ull,
                   "wgrad: feature tables must be < 4 GiB");
    hipStream_t s = (hipStream_t)stream;
    const long long elems = (long long)cout * k_vol * cin;
    if (n_out == 0 || n_in == 0) {
        TODA_HIP(hipMemsetAsync(dw, 0, elems * sizeof(float), s));
        return TODA_OK;
    }
    int chunks, rpc;
    wgrad_plan(n_out, k_vol, cin, cout, &chunks, &rpc);
    const size_t need = (size_t)chunks * elems * sizeof(float);
    if (ws_bytes < need) {
        set_error("wgrad: workspace %zu < required %zu", ws_bytes, need);
        return TODA_EWORKSPACE;
    }
    const int MT = tiles_pow2(cin), NT = tiles_pow2(cout);
    static const int env_sub = getenv("TODA_WG_SUB") ? atoi(getenv("TODA_WG_SUB")) : 7;  // 128-channel sides take 8 tiles per block: 1.08 -> 0.76 ms on 97.5k x 27 x 128 x 128
    int mtb = MT < 4 ? MT : 4, ntb = NT < 4 ? NT : 4;
    if (MT == 8 && (env_sub & 1)) mtb = 8;
    if (NT == 8 && (env_sub & 2)) ntb = 8;
    const int nsub_m = MT / mtb, nsub_n = NT / ntb;
    float* slab = (float*)ws;
    static const int env_wg_xcd = getenv("TODA_WG_XCD") ? atoi(getenv("TODA_WG_XCD")) : 1;
    const int chunks_launch = env_wg_xcd ? (chunks + 7) / 8 * 8 : chunks;      // padded to whole rounds of the 8 XCDs: blocks of the padding chunks leave at once
    const int xcd_chunks = env_wg_xcd ? chunks_launch : -chunks;
    if (MT == 8 && NT == 8 && (env_sub & 4)) {   // cooperative quarters: 0.75 -> 0.68 ms on 97.5k x 27 x 128 x 128
        if (cin == 128 && cout == 128)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<4, 4, true, true>), dim3(chunks_launch * k_vol), dim3(SC_BLOCK), 0, s, in, n_in, cin, dout,
                               cout, nbr, n_out, k_vol, rpc, MT, NT, 2, slab, xcd_chunks);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<4, 4, true>), dim3(chunks_launch * k_vol), dim3(SC_BLOCK), 0, s, in, n_in, cin, dout,
                               cout, nbr, n_out, k_vol, rpc, MT, NT, 2, slab, xcd_chunks);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(elems, SC_BLOCK)), dim3(SC_BLOCK), 0, s, slab, chunks, elems, dw);
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    const dim3 grid(chunks_launch * k_vol * nsub_m * nsub_n);
    const bool exact = cin == 16 * MT && cout == 16 * NT;
#define WG(MM, NN)                                                                                                  \
    if (exact)                                                                                                      \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<MM, NN, false, true>), grid, dim3(SC_BLOCK), 0, s, in, n_in, cin, dout, cout, nbr, \
                           n_out, k_vol, rpc, MT, NT, nsub_n, slab, xcd_chunks);                                    \
    else                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<MM, NN>), grid, dim3(SC_BLOCK), 0, s, in, n_in, cin, dout, cout, nbr, \
                       n_out, k_vol, rpc, MT, NT, nsub_n, slab, xcd_chunks)
#define WG_ROW(MM)             \
    switch (ntb) {             \
        case 1: WG(MM, 1); break; \
        case 2: WG(MM, 2); break; \
        case 4: WG(MM, 4); break; \
        default: WG(MM, 8); break; \
    }
    switch (mtb) {
        case 1: WG_ROW(1); break;
        case 2: WG_ROW(2); break;
        case 4: WG_ROW(4); break;
        default: WG_ROW(8); break;
    }
#undef WG_ROW
#undef WG
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(elems, SC_BLOCK)), dim3(SC_BLOCK), 0, s, slab, chunks, elems, dw);
    TODA_LAUNCH_CHECK();
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
