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
// (Round 4 swept the chunk at run time on both workloads once more - C3: 16 / 32 / 64 = 118.2 / 118.3 / 117.9 samples per second;
// C5, 128 -> 128 @ 97.5 k rows: 0 / 8 / 16 / 32 / 64 = 0.616 / 0.615 / 0.629 / 0.614 / 0.670 ms - and left it where it was.)
__device__ __forceinline__ int xcd_chunked_block(int b, int nblk) {
    constexpr int C = GG_XCD_CHUNK;
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
    // a side of 16 or 32 channels (one or two tiles = the whole side: the launch gives blocks of fewer than four tiles only then) is staged
    constexpr bool STAGE_A = EXACT && !COOP && MTB < 4, STAGE_B = EXACT && !COOP && NTB < 4;
    __shared__ float red[MTB * NTB * 4 * 64];
    __shared__ __attribute__((aligned(16))) float stage[(STAGE_A || STAGE_B) ? SC_BLOCK / 64 : 1][(STAGE_A || STAGE_B) ? 1024 : 4];      // per wave: 16 rows of both operands
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
    const unsigned lane_a = (unsigned)(MT * ii + m0) * 4u, lane_b = (unsigned)(NT * ii + n0) * 4u;
    const unsigned row_a = (unsigned)cin * 4u, row_b = (unsigned)cout * 4u;
    auto round16 = [&](int d, int limit, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float a[4][MTB], b[4][NTB];
        if constexpr (STAGE_A || STAGE_B) {
            // Sides of 16 or 32 channels: a lane's one or two channels of a pair are 4 or 8 bytes, and the address unit takes a
            // wave-instruction's 64 addresses at the same pace whatever their width (16 cycles): eight 8-byte loads per 16 MFMAs kept it
            // busy all the time (16 waves per CU; 32 -> 32 at 682 k rows: 0.273 ms).  Such a side comes in as WHOLE ROWS, 16 bytes per
            // lane (the 16 rows of a round in one or two loads instead of four), and takes the turn into the MFMA layout through a
            // wave-private LDS tile (contiguous writes, 256- or 512-byte reads: no conflicts): 0.241 ms.
            float* const st = stage[wv];
            constexpr int LA = 4 * MTB, LB = 4 * NTB;          // lanes per row (16 bytes each)
            constexpr int IA = STAGE_A ? LA / 4 : 0, IB = STAGE_B ? LB / 4 : 0;      // loads per round: 64 lanes cover 64 / L rows
            f32x4 ra[STAGE_A ? IA : 1], rb[STAGE_B ? IB : 1];
            if constexpr (STAGE_A) {
#pragma unroll
                for (int j = 0; j < IA; ++j) {
                    const int p = d + lane / LA + (64 / LA) * j;
                    const bool ok = FULL || p < limit;
                    ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? (unsigned)qi[ok ? p : d] + (unsigned)(lane % LA) * 16u : OOB, 0, 0));
                }
            }
            if constexpr (STAGE_B) {
#pragma unroll
                for (int j = 0; j < IB; ++j) {
                    const int p = d + lane / LB + (64 / LB) * j;
                    const bool ok = FULL || p < limit;
                    rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dout_rsrc, ok ? (unsigned)qo[ok ? p : d] + (unsigned)(lane % LB) * 16u : OOB, 0, 0));
                }
            }
            // the side that is not staged: 16-byte loads straight into the MFMA layout, as below
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int p = d + 4 * t + g;
                const bool ok = FULL || p < limit;
                const int pc = ok ? p : d;
                if constexpr (!STAGE_A) {
#pragma unroll
                    for (int m4 = 0; m4 < MTB; m4 += 4) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? (unsigned)qi[pc] + lane_a + 4u * m4 : OOB, 0, 0));
#pragma unroll
                        for (int m = 0; m < 4; ++m) a[t][m4 + m] = v[m];
                    }
                }
                if constexpr (!STAGE_B) {
#pragma unroll
                    for (int n4 = 0; n4 < NTB; n4 += 4) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dout_rsrc, ok ? (unsigned)qo[pc] + lane_b + 4u * n4 : OOB, 0, 0));
#pragma unroll
                        for (int n = 0; n < 4; ++n) b[t][n4 + n] = v[n];
                    }
                }
            }
            if constexpr (STAGE_A) {
#pragma unroll
                for (int j = 0; j < IA; ++j) *reinterpret_cast<f32x4*>(st + (lane / LA + (64 / LA) * j) * (16 * MTB) + (lane % LA) * 4) = ra[j];
            }
            if constexpr (STAGE_B) {
#pragma unroll
                for (int j = 0; j < IB; ++j) *reinterpret_cast<f32x4*>(st + 512 + (lane / LB + (64 / LB) * j) * (16 * NTB) + (lane % LB) * 4) = rb[j];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (STAGE_A) {
#pragma unroll
                    for (int m = 0; m < MTB; ++m) a[t][m] = st[(4 * t + g) * (16 * MTB) + MTB * ii + m];
                }
                if constexpr (STAGE_B) {
#pragma unroll
                    for (int n = 0; n < NTB; ++n) b[t][n] = st[512 + (4 * t + g) * (16 * NTB) + NTB * ii + n];
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = d + 4 * t + g;
            const bool ok = FULL || p < limit;
            const int pc = ok ? p : d;  // any valid queue slot
            const unsigned ia = (unsigned)qi[pc] + lane_a;       // the queue holds BYTE offsets of rows (multiplied once, when a pair is queued:
            const unsigned ib = (unsigned)qo[pc] + lane_b;       // a 32-bit multiply is a quarter-rate instruction on the lanes the MFMAs use)
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
            qi[pos] = (int)((unsigned)i * row_a);
            qo[pos] = (int)((unsigned)o * row_b);
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

#include "spconv_split.cuh"

using namespace toda;

// ---- matrix path (include/toda.h): 0 = native fp32 MFMA, 1 = exact bf16 hi/mid/lo split for the channel pairs split_shape_ok names.
// Process-wide like the kernels' environment knobs; initial value from TODA_MM (native | split, default split).  The packed operand of a supported
// channel pair is written in the format of the path that is current at pack time and must be multiplied under the same path.
#include <atomic>
static std::atomic<int> g_matrix_path{-1};
static int matrix_path() {
    int m = g_matrix_path.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("TODA_MM");
        m = (e && (e[0] == 'n' || e[0] == '0')) ? 0 : 1;      // default: split (it holds the gates of tests/test_gpu_split.py)
        g_matrix_path.store(m, std::memory_order_relaxed);
    }
    return m;
}
static inline bool use_split(int c_gather, int c_produce) { return matrix_path() == 1 && toda::split_shape_ok(c_gather, c_produce); }

extern "C" int toda_matrix_path(void) { return matrix_path(); }
extern "C" int toda_set_matrix_path(int mode) {
    TODA_CHECK_ARG(mode == 0 || mode == 1, "set_matrix_path: mode must be 0 (native) or 1 (split)");
    g_matrix_path.store(mode, std::memory_order_relaxed);
    return TODA_OK;
}
extern "C" int toda_spconv_split_supported(int c_gather, int c_produce) { return toda::split_shape_ok(c_gather, c_produce) ? 1 : 0; }

extern "C" size_t toda_spconv_packed_weight_floats(int k_vol, int c_gather, int c_produce) {
    const size_t native = (size_t)k_vol * tiles_pow2(c_gather) * tiles_pow2(c_produce) * 256;
    if (!toda::split_shape_ok(c_gather, c_produce)) return native;
    const size_t split = toda::split_packed_bytes(k_vol, c_gather, c_produce) / 4;      // the larger of the two formats, whatever the path
    return split > native ? split : native;
}

static void split_pack_fill(toda::SplitPackBatch& b, int i, const float* w, int cout, int k_vol, int cin, int transpose, int flip_k, float* wp) {
    const int cgather = transpose ? cout : cin, cprod = transpose ? cin : cout;
    b.w[i] = w, b.wps[i] = reinterpret_cast<toda::u32x4*>(wp), b.cout[i] = cout, b.K[i] = k_vol, b.cin[i] = cin;
    b.transpose[i] = transpose != 0, b.flip[i] = flip_k != 0;
    b.KC[i] = (unsigned char)(cgather / 32), b.NT[i] = (unsigned char)(cprod / 16);
}

extern "C" int toda_spconv_pack_weight(const float* w, int cout, int k_vol, int cin, int transpose, int flip_k,
                                       float* wp, void* stream) {
    TODA_CHECK_ARG(cout >= 1 && cin >= 1 && k_vol >= 1, "pack_weight: bad shape");
    const int cgather = transpose ? cout : cin, cprod = transpose ? cin : cout;
    TODA_CHECK_ARG(cgather <= 128 && cprod <= 128, "pack_weight: channels > 128 unsupported (gather %d, produce %d)",
                   cgather, cprod);
    if (use_split(cgather, cprod)) {
        toda::SplitPackBatch b = {};
        split_pack_fill(b, 0, w, cout, k_vol, cin, transpose, flip_k, wp);
        hipLaunchKernelGGL(split_pack_batch_kernel, dim3(cdiv((long long)k_vol * b.KC[0] * b.NT[0] * 192, SC_BLOCK), 1), dim3(SC_BLOCK), 0,
                           (hipStream_t)stream, b);
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
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
    // two launches at most per PACK_MAX_SEG operands: the fp32 fragment format and (matrix path "split") the three-plane bf16 format
    PackBatch b = {};
    toda::SplitPackBatch sb = {};
    int m = 0, sm = 0;
    long long most = 0, smost = 0;
    auto flush = [&]() -> int {
        if (m > 0) {
            hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(cdiv(most, SC_BLOCK), m), dim3(SC_BLOCK), 0, (hipStream_t)stream, b);
            TODA_LAUNCH_CHECK();
        }
        m = 0, most = 0;
        return TODA_OK;
    };
    auto sflush = [&]() -> int {
        if (sm > 0) {
            hipLaunchKernelGGL(split_pack_batch_kernel, dim3(cdiv(smost, SC_BLOCK), sm), dim3(SC_BLOCK), 0, (hipStream_t)stream, sb);
            TODA_LAUNCH_CHECK();
        }
        sm = 0, smost = 0;
        return TODA_OK;
    };
    for (int s = 0; s < n; ++s) {
        TODA_CHECK_ARG(w[s] && wp[s] && cout[s] >= 1 && cin[s] >= 1 && k_vol[s] >= 1, "pack_weights: bad segment %d", s);
        const int cgather = transpose[s] ? cout[s] : cin[s], cprod = transpose[s] ? cin[s] : cout[s];
        TODA_CHECK_ARG(cgather <= 128 && cprod <= 128, "pack_weights: channels > 128 unsupported (segment %d)", s);
        if (use_split(cgather, cprod)) {
            split_pack_fill(sb, sm, w[s], cout[s], k_vol[s], cin[s], transpose[s], flip_k[s], wp[s]);
            const long long total = (long long)k_vol[s] * sb.KC[sm] * sb.NT[sm] * 192;
            if (total > smost) smost = total;
            if (++sm == toda::SPLIT_PACK_MAX_SEG) {
                const int rc = sflush();
                if (rc != TODA_OK) return rc;
            }
            continue;
        }
        const int i = m;
        b.w[i] = w[s], b.wp[i] = wp[s], b.cout[i] = cout[s], b.K[i] = k_vol[s], b.cin[i] = cin[s];
        b.transpose[i] = transpose[s] != 0, b.flip[i] = flip_k[s] != 0;
        b.Q[i] = (unsigned char)tiles_pow2(cgather), b.NT[i] = (unsigned char)tiles_pow2(cprod);
        const long long total = (long long)k_vol[s] * b.Q[i] * b.NT[i] * 256;
        if (total > most) most = total;
        if (++m == PACK_MAX_SEG) {
            const int rc = flush();
            if (rc != TODA_OK) return rc;
        }
    }
    {
        const int rc = flush();
        if (rc != TODA_OK) return rc;
    }
    return sflush();
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
    const bool vec_ok = (c_gather & 3) == 0;
    // weight staging: the LDS-shared slice when it measured faster (Q >= 2 and NT >= Q); TODA_GG_LDS = 0 never, 2 always
    static const int env_lds_raw = getenv("TODA_GG_LDS") ? atoi(getenv("TODA_GG_LDS")) : 1;
    const int env_lds = env_lds_raw == 2 ? 1 : (env_lds_raw == 1 ? (Q >= 2 && NT >= Q) : 0);
    if (use_split(c_gather, c_produce)) {     // matrix path "split": wp holds the three-plane bf16 operand (spconv_split.cuh)
        TODA_CHECK_ARG(k_vol >= 1 && !(cls_sorted && stats), "gather_gemm (split path): no statistics on a class-sorted launch");
        const toda::u32x4* wps = reinterpret_cast<const toda::u32x4*>(wp);
#define SPL(KK, SS, NN, BB, WW)                                                                                                            \
    GG_LAUNCH(HIP_KERNEL_NAME(gg_split_kernel<KK, SS, NN, 2, BB, WW, (NN == 2)>), dim3(cdiv(cdiv(n_out, 32), BB / 64)), dim3(BB), 0, s, in, n_in, wps, \
              nbr, n_out, k_vol, c_produce, bias, out, order, stats, cls_sorted, classes)
        // one 32-channel chunk per stage, 256-thread workgroups; 32 produced channels: loads issued between the matrix groups (spconv_split.cuh has the numbers)
        const int blk = 256;
        const int kc = c_gather / 32, nt = c_produce / 16;
        if (0) {}
        else if (kc == 1 && nt == 2) SPL(1, 1, 2, 256, 4);
        else if (kc == 1 && nt == 4) SPL(1, 1, 4, 256, 4);
        else if (kc == 1 && nt == 8) SPL(1, 1, 8, 256, 3);
        else if (kc == 2 && nt == 2) SPL(2, 1, 2, 256, 4);
        else if (kc == 2 && nt == 4) SPL(2, 1, 4, 256, 4);
        else if (kc == 2 && nt == 8) SPL(2, 1, 8, 256, 3);
        else if (kc == 4 && nt == 2) SPL(4, 1, 2, 256, 4);
        else if (kc == 4 && nt == 4) SPL(4, 1, 4, 256, 4);
        else SPL(4, 1, 8, 256, 3);
#undef SPL
        TODA_LAUNCH_CHECK();
        if (stats) {
            fold_or_defer(stats, cdiv(cdiv(n_out, 32), blk / 64), c_produce, s);
            TODA_LAUNCH_CHECK();
        }
        return TODA_OK;
    }
    // 128 -> 128: one 64 KiB slice shared by a 512-thread workgroup, 16 rows per wave (0.67 ms against 0.76 for two row tiles per
    // wave and 0.70 for register-only weights on 97.5k x 27 x 128 x 128)
    if (vec_ok && Q == 8 && NT == 8 && !cls_sorted) {
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<8, 8, 1, true, false, 512>), dim3(cdiv(cdiv(n_out, 16), 8)), dim3(512), 0, s, in, n_in,
                  c_gather, wp, nbr, n_out, k_vol, c_produce, bias, out, order, stats);
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
    if ((env_lds || stats) && !cls_sorted && vec_ok && Q * NT <= 32) {  // weight slice <= 32 KiB per buffer
#define GL(QQ, NN, RR)                                                                                                   \
    GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_lds_kernel<QQ, NN, RR, true, true, SC_BLOCK, false, (QQ <= 4 && NN <= 4 && RR == 2)>), \
              dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, n_out, k_vol,  \
              c_produce, bias, out, order, stats)
#define GL_ROW(QQ)                 \
    switch (NT) {                  \
        case 1: GL(QQ, 1, 2); break; \
        case 2: GL(QQ, 2, 2); break; \
        case 4: GL(QQ, 4, 2); break; \
        default: GL(QQ, 8, 1); break; \
    }
        switch (Q) {
            case 1: GL_ROW(1); break;
            case 2: GL_ROW(2); break;
            case 4: GL_ROW(4); break;
            default: GL_ROW(8); break;
        }
#undef GL_ROW
#undef GL
        TODA_LAUNCH_CHECK();
        if (stats) {      // fold the per-workgroup partial sums of the launch above (same grid arithmetic as GL / GL_ROW)
            const int rr = NT >= 8 ? 1 : 2;
            fold_or_defer(stats, cdiv(cdiv(n_out, 16 * rr), SC_BLOCK / 64), c_produce, s);
            TODA_LAUNCH_CHECK();
        }
        return TODA_OK;
    }
    TODA_CHECK_ARG(stats == nullptr, "gather_gemm: statistics requested on a launch shape without the fused epilogue");
    // register-resident weights (gather_gemm_kernel): the narrow, the "more gathered than produced" and the class-sorted launches
#define GGV(QQ, NN, RR, VV)                                                                                           \
    if (cls_sorted && (VV))                                                                                           \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_kernel<QQ, NN, RR, false, true, true>),                                 \
                  dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, \
                  n_out, k_vol, c_produce, bias, out, 0, order, cls_sorted, classes);                                 \
    else                                                                                                              \
        GG_LAUNCH(HIP_KERNEL_NAME(gather_gemm_kernel<QQ, NN, RR, false, VV>),                                         \
                  dim3(cdiv(cdiv(n_out, 16 * RR), SC_BLOCK / 64)), dim3(SC_BLOCK), 0, s, in, n_in, c_gather, wp, nbr, \
                  n_out, k_vol, c_produce, bias, out, 0, order, nullptr, classes)
#define GG_ROW(QQ)                                                                      \
    switch (NT) {                                                                       \
        case 1: GGV(QQ, 1, 2, true); break;                                             \
        case 2: GGV(QQ, 2, 2, true); break;                                             \
        case 4: GGV(QQ, 4, 2, true); break;                                             \
        default:                                                                        \
            if (QQ < 8) { GGV(QQ, 8, 1, true); } else { GGV(QQ, 8, 2, true); }           \
            break;      /* 128 -> 128: two row tiles measured 16 % faster than one */    \
    }
    if (!vec_ok) {  // rows not 16-byte aligned (e.g. 5 point features): dword gathers
        TODA_CHECK_ARG(Q == 1, "gather_gemm: gathered channel counts above 16 must be multiples of 4 (got %d)", c_gather);
        switch (NT) {
            case 1: GGV(1, 1, 2, false); break;
            case 2: GGV(1, 2, 2, false); break;
            case 4: GGV(1, 4, 2, false); break;
            default: GGV(1, 8, 1, false); break;
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
#undef GGV
    TODA_LAUNCH_CHECK();
    return TODA_OK;
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
    if (matrix_path() == 1 && toda::wgrad_split_shape_ok(cin, cout)) {      // matrix path "split": spconv_split.cuh
        const dim3 g(chunks_launch * k_vol);
#define WGS(MM, NN)                                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_split_kernel<MM, NN>), g, dim3(SC_BLOCK), 0, s, in, n_in, dout, nbr, n_out, k_vol, rpc, slab, \
                       xcd_chunks)
        if (cin == 128 && cout == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_split_kernel<4, 4, 2, 2>), g, dim3(SC_BLOCK), 0, s, in, n_in, dout, nbr, n_out, k_vol, rpc, slab, xcd_chunks);
        else if (cin == 64 && cout == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_split_kernel<4, 4, 1, 2>), g, dim3(SC_BLOCK), 0, s, in, n_in, dout, nbr, n_out, k_vol, rpc, slab, xcd_chunks);
        else if (cin == 128 && cout == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_split_kernel<4, 4, 2, 1>), g, dim3(SC_BLOCK), 0, s, in, n_in, dout, nbr, n_out, k_vol, rpc, slab, xcd_chunks);
        else if (cin == 32 && cout == 32) WGS(2, 2);
        else if (cin == 32) WGS(2, 4);
        else if (cout == 32) WGS(4, 2);
        else WGS(4, 4);
#undef WGS
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(elems, SC_BLOCK)), dim3(SC_BLOCK), 0, s, slab, chunks, elems, dw);
        TODA_LAUNCH_CHECK();
        return TODA_OK;
    }
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

