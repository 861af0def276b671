// Exact fp32 gather-GEMM on the BF16 matrix pipe (the "split" matrix path; included by spconv.hip).
//
// gfx950 multiplies fp32 operands at 64 FLOP/clk/SIMD (v_mfma_f32_16x16x4_f32) and bf16 operands at 1024
// (v_mfma_f32_16x16x32_bf16): 16 x.  An fp32 value is EXACTLY the sum of three bf16 values,
//     x = hi + mid + lo,   hi = x with its low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid
// (24 significant bits = 8 + 8 + 8; both differences are exact in fp32, and lo has at most 8 significant bits, so it is a bf16
// number), and a product of two bf16 values is exact in fp32.  So
//     a . b = ah.bh + (ah.bm + am.bh) + (ah.bl + am.bm + al.bh) + [am.bl + al.bm + al.bl]
// where the bracket is below 2^-24 of |a . b| - beneath the rounding of the fp32 product itself - and is dropped: six bf16
// matrix instructions with fp32 accumulation stand for sixteen fp32 ones (K = 32 against K = 4 per instruction at half the
// cycles: 6 / 16 of the matrix cycles).  Every (tile, 32-channel chunk) adds its six terms smallest first.
//
// Structure = the per-offset output-stationary kernel of spconv.hip (gather_gemm_lds_kernel) with these differences:
//   * weights are split ONCE per step by the pack kernel into three bf16 planes in fragment order (6 bytes per weight; the slice of an
//     offset goes global -> LDS by LDS-DMA, no staging registers, double buffered);
//   * gathered rows arrive as fp32 (same 16-byte bounds-checked buffer loads, "no neighbour" = out-of-range offset = zeros) and are
//     split in registers (5.5 vector instructions per value) - the vector pipe issues beside the matrix pipe of the other waves of the SIMD;
//   * the rows of offset k + 1 are requested before the matrix work of offset k (one offset of look-ahead per wave) and land under it.
// MFMA operand maps (cdna_hip_programming.md section 3): A[i = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][lane & 15],
// D: col = lane & 15, row = 4 (lane >> 4) + reg (the map of the fp32 instruction: the epilogue is the one of gather_gemm_lds_kernel).
// k-permutation: lane (r, g) contracts element j of chunk kc over the gathered channel 32 kc + 16 (j >> 2) + 4 g + (j & 3): its two
// 16-byte loads per chunk are the pieces 4 g .. 4 g + 3 of the 64-byte segments 2 kc and 2 kc + 1 of its row, so ONE load instruction
// of the wave reads 16 rows x one whole 64-byte segment (a lane reading 32 KC contiguous bytes makes every instruction touch all
// segments of all 16 rows for a quarter of their bytes: 4 x the cache-line accesses).  The packed planes follow the same map.
// Produced channel of column c of tile n: NT c + n (vector stores).
#pragma once

namespace toda {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// {bf16(x0) in the low half, bf16(x1) in the high half}, both truncated (upper 16 bits of the fp32 patterns)
__device__ __forceinline__ unsigned sp_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ float sp_trunc(float x) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);
}
#ifndef SP_STAMPS
#define SP_STAMPS 0      // diagnostic build only: per-wave cycle sums of the loop's segments through the statistics pointer (no statistics)
#endif
#if SP_STAMPS
#define SP_STAMP(i)                                                                               \
    do {                                                                                          \
        unsigned long long t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        seg[i] += t_ - t_prev;                                                                    \
        t_prev = t_;                                                                              \
    } while (0)
#else
#define SP_STAMP(i)
#endif
#ifndef SP_ABLATE
#define SP_ABLATE 0      // measurement builds only (wrong numbers): 1 = no operand split, 2 = no matrix instructions
#endif
// 8 consecutive fp32 values -> the three bf16 planes of an A fragment
__device__ __forceinline__ void sp_split8(const f32x4& v0, const f32x4& v1, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = p < 2 ? v0[2 * p] : v1[2 * p - 4], x1 = p < 2 ? v0[2 * p + 1] : v1[2 * p - 3];
        h[p] = sp_pack_hi(x0, x1);
#if SP_ABLATE & 1
        m[p] = h[p] ^ 0x00010001u, l[p] = h[p] ^ 0x00020002u;
        continue;
#endif
        const float r0 = x0 - sp_trunc(x0), r1 = x1 - sp_trunc(x1);
        m[p] = sp_pack_hi(r0, r1);
        const float l0 = r0 - sp_trunc(r0), l1 = r1 - sp_trunc(r1);
        l[p] = sp_pack_hi(l0, l1);
    }
}
__device__ __forceinline__ unsigned short sp_plane_of(float v, int plane) {
    const float h = sp_trunc(v), r = v - h, m = sp_trunc(r), l = r - m;
    const float pick = plane == 0 ? h : (plane == 1 ? m : l);
    return (unsigned short)(__builtin_bit_cast(unsigned, pick) >> 16);
}

// ---- packed operand: u32x4 units  wps[((((k KC + kc) NT + n) 3 + plane) 64 + lane)] = 8 bf16, j = 0 .. 7 ---------------------------
constexpr int SPLIT_PACK_MAX_SEG = 48;
struct SplitPackBatch {
    const float* w[SPLIT_PACK_MAX_SEG];
    u32x4* wps[SPLIT_PACK_MAX_SEG];
    int cout[SPLIT_PACK_MAX_SEG], K[SPLIT_PACK_MAX_SEG], cin[SPLIT_PACK_MAX_SEG];
    unsigned char transpose[SPLIT_PACK_MAX_SEG], flip[SPLIT_PACK_MAX_SEG], KC[SPLIT_PACK_MAX_SEG], NT[SPLIT_PACK_MAX_SEG];
};
// one thread = one lane's 16 bytes of one plane; blockIdx.y = segment (all split operands of a step in one launch)
__global__ void __launch_bounds__(SC_BLOCK)
split_pack_batch_kernel(const SplitPackBatch b) {
    const int sg = blockIdx.y;
    const int K = b.K[sg], KC = b.KC[sg], NT = b.NT[sg], cin = b.cin[sg], cout = b.cout[sg];
    const long long e = (long long)blockIdx.x * SC_BLOCK + threadIdx.x;
    if (e >= (long long)K * KC * NT * 192) return;
    const int lane = (int)(e & 63);
    long long t = e >> 6;
    const int plane = (int)(t % 3);
    t /= 3;
    const int n = (int)(t % NT);
    t /= NT;
    const int kc = (int)(t % KC);
    const int k = (int)(t / KC);
    const int c = lane & 15, g = lane >> 4;
    const int pch = NT * c + n;
    const int kk = b.flip[sg] ? K - 1 - k : k;
    const float* __restrict__ w = b.w[sg];
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gch = 32 * kc + 16 * (j >> 2) + 4 * g + (j & 3);
        float x = 0.0f;
        if (!b.transpose[sg]) {
            if (gch < cin && pch < cout) x = w[((size_t)pch * K + kk) * cin + gch];
        } else {
            if (gch < cout && pch < cin) x = w[((size_t)gch * K + kk) * cin + pch];
        }
        v[j] = sp_plane_of(x, plane);
    }
    b.wps[sg][e] = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
                         (unsigned)v[6] | ((unsigned)v[7] << 16)};
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------------
// BLK threads = BLK / 64 waves share one double-buffered weight slice: every workgroup re-reads the whole packed operand (24 KiB per
// offset at 64 -> 64), so with 256-thread workgroups the slices are HALF of the kernel's vector-memory traffic (2.0 GB beside 1.9 GB of
// gathered rows at 389 k rows) and the waves spend 39 % of their time pushing loads into a full memory pipeline (in-kernel stamps,
// profiles/r05_split_stamps.txt).  Three waves per SIMD either way (168 registers): 256 threads x 3 workgroups or 768 x 1 per CU.
template <int KC, int NT, int RT, int BLK>
__global__ void __launch_bounds__(BLK, 3)
gg_split_kernel(const float* __restrict__ in, int n_in, const u32x4* __restrict__ wps, const int* __restrict__ nbr, int n_out, int K, int cp,
                const float* __restrict__ bias, float* __restrict__ out, const int* __restrict__ order, double* __restrict__ stats) {
    constexpr int CG = 32 * KC;
    constexpr int UNITS = KC * NT * 3;      // 1 KiB wave-instruction images per offset
    constexpr int SLICE = UNITS * 64;       // u32x4 per offset
    static_assert(2 * SLICE * 16 >= (BLK / 64) * 2 * 16 * NT * 4, "the statistics scratch aliases the weight buffers");
    __shared__ u32x4 wl[2 * SLICE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int wave = blk * (BLK / 64) + wv;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = wave * (16 * RT);

    f32x4 acc[RT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float b = 0.0f;
        if (bias && NT * r + n < cp) b = bias[NT * r + n];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][n] = f32x4{b, b, b, b};
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = table_rsrc(in, (unsigned)n_in * (unsigned)CG * 4u);
    int rows[RT];
    bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        live[rt] = row0 + rt * 16 + r < n_out;       // rows past the end read the ids of the last row and are never stored
        rows[rt] = live[rt] ? (order ? order[row0 + rt * 16 + r] : row0 + rt * 16 + r) : n_out - 1;
    }
    auto load_ids = [&](int k, int (&dst)[RT]) {
        const int kk = k < K ? k : K - 1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            dst[rt] = __builtin_nontemporal_load(nbr + (size_t)kk * n_out + rows[rt]);      // past the last offset: its ids again, never used
        }
    };
    auto gather = [&](const int (&src)[RT], bool valid, f32x4 (&raw)[RT][2 * KC]) {      // valid (wave-uniform): the offset exists
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bool ok = src[rt] >= 0 && valid;
            const unsigned base = (unsigned)src[rt] * (unsigned)(CG * 4) + (unsigned)(16 * g);
#pragma unroll
            for (int i = 0; i < 2 * KC; ++i)
                raw[rt][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? base + 64u * i : OOB, 0, 0));
        }
    };
    auto stage = [&](int k, int buf) {       // the slice of offset k: global -> LDS, 1 KiB per wave-instruction, no registers
        const int kk = k < K ? k : K - 1;
#pragma unroll
        for (int t = 0; t < (UNITS + BLK / 64 - 1) / (BLK / 64); ++t) {
            const int u = t * (BLK / 64) + wv;
            if (UNITS % (BLK / 64) == 0 || u < UNITS)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wps + (size_t)kk * SLICE + u * 64 + lane),
                                                 reinterpret_cast<float*>(&wl[buf * SLICE + u * 64]), 16, 0, 0);
        }
    };

    // ids live in three register sets that rotate by NAME (the loop body is written three times): a copy of a set whose load is
    // still in flight, or the "row exists" select applied at load time, would be a wait for that load at the end of every offset
    int idA[RT], idB[RT], idC[RT];
    f32x4 raw[RT][2 * KC];
    load_ids(0, idA);
    load_ids(1, idB);
    stage(0, 0);
    gather(idA, true, raw);
    __syncthreads();

#if SP_STAMPS
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, t_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
    auto body = [&](int k, const int (&ids_cur)[RT], const int (&ids_nxt)[RT], int (&ids_new)[RT]) {
        const int cur = k & 1;
        SP_STAMP(5);
        bool any = false;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) any = any || __any(ids_cur[rt] >= 0);
        // one straight-line body per (wave, offset) that has a neighbour in any of its rows: on the submanifold tables 0.83 of the
        // executed tile rows are pairs this way against 0.85 with a test per 16-row tile (C3, 389 k rows) - not worth three bodies
        // (hipcc joins them with 32 accumulator copies per offset)
        u32x4 ah[RT][KC], am[RT][KC], al[RT][KC];
        if (any) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) sp_split8(raw[rt][2 * kc], raw[rt][2 * kc + 1], ah[rt][kc], am[rt][kc], al[rt][kc]);
        }
        SP_STAMP(0);
        // issue order = the order the waits below and at the top of the next offset retire them in: the slice of offset k + 1 (needed
        // by every wave behind the barrier), the rows of offset k + 1, the ids of offset k + 2
        stage(k + 1, cur ^ 1);
        asm volatile("" ::: "memory");
        gather(ids_nxt, k + 1 < K, raw);          // rows of offset k + 1: in flight under the matrix work below
        asm volatile("" ::: "memory");
        load_ids(k + 2, ids_new);
        asm volatile("" ::: "memory");
        SP_STAMP(1);
        if (any) {
            const u32x4* __restrict__ wb = wl + cur * SLICE + lane;
            constexpr int S = KC * NT;           // steps: (kc, n), the three planes of a step one step ahead of its 6 RT instructions
            u32x4 bq[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[0][pl] = wb[pl * 64];
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int st = 0; st < S; ++st) {
                const int kc = st / NT, n = st % NT;
                if (st + 1 < S) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[(st + 1) & 1][pl] = wb[((st + 1) * 3 + pl) * 64];
                }
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[st & 1][0]), bm = __builtin_bit_cast(bf16x8, bq[st & 1][1]),
                             bl = __builtin_bit_cast(bf16x8, bq[st & 1][2]);
#if SP_ABLATE & 2
#define SP_TERM(AA, BB)                                                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                                \
        acc[rt][n][0] += __builtin_bit_cast(float, AA[rt][kc][0] ^ __builtin_bit_cast(u32x4, BB)[0])
#else
#define SP_TERM(AA, BB)                                                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                                \
        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AA[rt][kc]), BB, acc[rt][n], 0, 0, 0)
#endif
                SP_TERM(al, bh);
                SP_TERM(ah, bl);
                SP_TERM(am, bm);
                SP_TERM(am, bh);
                SP_TERM(ah, bm);
                SP_TERM(ah, bh);
#undef SP_TERM
                // keep the order written here: the next step's three fragment reads, then this step's matrix instructions
                if (st + 1 < S) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * RT, 0);
            }
        }
        // The barrier only hands over the LDS slice: wait for this wave's share of it (the oldest of the loads above) and leave the
        // rows and ids in flight across the barrier - __syncthreads() would drain them all (vmcnt(0): an LDS-DMA is a pending LDS write),
        // which put one full memory round trip, the far-ahead id loads included, into every offset (1.6 us of the 3.7 us per offset)
        SP_STAMP(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RT * 2 * KC + RT) : "memory");
        SP_STAMP(3);
        __builtin_amdgcn_s_barrier();
        SP_STAMP(4);
    };
    for (int k = 0;;) {
        body(k, idA, idB, idC);
        if (++k >= K) break;
        body(k, idB, idC, idA);
        if (++k >= K) break;
        body(k, idC, idA, idB);
        if (++k >= K) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (row0 >= n_out && !stats) return;

#if SP_STAMPS
    if (stats) {
        if (lane == 0)
            for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned long long*>(stats)[(size_t)wave * 6 + i] = seg[i];
        stats = nullptr;
    }
#endif
    // BatchNorm statistics of the layer's output from the accumulators - as gather_gemm_lds_kernel (same scratch layout and fold)
    if (stats) {
        float (*st_sh)[2][16 * NT] = reinterpret_cast<float (*)[2][16 * NT]>(wl);      // every wave is past its last slice read (barrier above)
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
                st_sh[wv][0][NT * r + n] = sm[n];
                st_sh[wv][1][NT * r + n] = sq[n];
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
            const int row = __shfl(rows[rt], 4 * g + reg, 64);
            if (pos >= n_out) continue;
            float* dst = out + (size_t)row * cp + NT * r;
            if (full) {
                if constexpr (NT == 2) {
                    *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][reg], acc[rt][1][reg]);
                } else {
#pragma unroll
                    for (int n = 0; n < NT; n += 4)
                        *reinterpret_cast<f32x4*>(dst + n) = f32x4{acc[rt][n][reg], acc[rt][n + 1][reg], acc[rt][n + 2][reg], acc[rt][n + 3][reg]};
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (NT * r + n < cp) dst[n] = acc[rt][n][reg];
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
static inline bool split_shape_ok(int c_gather, int c_produce) {
    return (c_gather == 32 || c_gather == 64) && (c_produce == 32 || c_produce == 64);
}
static inline size_t split_packed_bytes(int k_vol, int c_gather, int c_produce) {
    return (size_t)k_vol * (c_gather / 32) * (c_produce / 16) * 3 * 1024;
}

}  // namespace toda
