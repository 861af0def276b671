// Shared host/device helpers for libtoda_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/toda.h"

namespace toda {

void set_error(const char* fmt, ...);

#define TODA_CHECK_ARG(cond, ...)           \
    do {                                    \
        if (!(cond)) {                      \
            toda::set_error(__VA_ARGS__);   \
            return TODA_EINVAL;             \
        }                                   \
    } while (0)

#define TODA_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            toda::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                    \
            return TODA_ELAUNCH;                                                    \
        }                                                                           \
    } while (0)

#define TODA_LAUNCH_CHECK() TODA_HIP(hipGetLastError())

// device fault word (common.hip): device-visible pointer to a host-mapped word, nullptr when pinned memory is unavailable;
// fault_take() returns and clears it.  Kernels raise bits with fault_raise().
unsigned* fault_word_dev();
unsigned fault_take();
constexpr unsigned FAULT_SPIN_LIMIT = 1u << 21;      // polls of ~1 us: about two seconds
__device__ __forceinline__ void fault_raise(unsigned* word, unsigned bit) {
    if (word) __hip_atomic_fetch_or(word, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// effective row count: min(n, *n_dev) when a device-side count is supplied
__device__ __forceinline__ int eff_n(int n, const int32_t* n_dev) {
    if (n_dev) {
        int d = *n_dev;
        return d < n ? d : n;
    }
    return n;
}

// ---- device-wide exclusive scan over int32 (reduce / scan-sums / apply) -------------------
// SCAN_TILE elements per block.  `partials` needs cdiv(n, SCAN_TILE) + 1 ints.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

static inline size_t scan_partials_bytes(long long n) {
    return align_up((size_t)(cdiv(n, SCAN_TILE) + 1) * sizeof(int32_t), 256);
}

}  // namespace toda
