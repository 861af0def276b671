// Error plumbing shared by every translation unit of libtoda_hip.so.
#include <stdarg.h>

#include <mutex>

#include "common.h"

namespace toda {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- device fault word ------------------------------------------------------------------------------------------------
// The two kernels whose workgroups wait for each other inside one launch (bn2d_*_split_kernel, wino_fwd_ws_kernel) bound
// their spins: a wait that outlives FAULT_SPIN_LIMIT polls raises a word in host-mapped pinned memory and falls through
// (the launch then finishes with wrong numbers instead of hanging the GPU).  The host reads that word without any device
// synchronisation at the next call of the same family and through toda_device_fault().
static std::mutex g_fault_mu;
static unsigned* g_fault_host = nullptr;
static unsigned* g_fault_dev = nullptr;
static bool g_fault_tried = false;

unsigned* fault_word_dev() {
    std::lock_guard<std::mutex> lock(g_fault_mu);
    if (!g_fault_tried) {
        g_fault_tried = true;
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && h) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                g_fault_host = (unsigned*)h;
                g_fault_dev = (unsigned*)d;
                for (int i = 0; i < 16; ++i) g_fault_host[i] = 0u;
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    return g_fault_dev;
}

unsigned fault_take() {
    std::lock_guard<std::mutex> lock(g_fault_mu);
    if (!g_fault_host) return 0u;
    volatile unsigned* w = g_fault_host;
    const unsigned v = w[0];
    if (v) w[0] = 0u;
    return v;
}

}  // namespace toda

extern "C" const char* toda_last_error(void) { return toda::g_err; }
extern "C" int toda_abi_version(void) { return 3; }

extern "C" int toda_device_fault(void) {
    const unsigned v = toda::fault_take();
    if (!v) return TODA_OK;
    toda::set_error("device fault word 0x%x: a bounded inter-workgroup wait gave up (%s%s); the results of that launch are invalid",
                    v, (v & TODA_FAULT_BN2D) ? "bn2d split kernel " : "", (v & TODA_FAULT_WINO) ? "wino_fwd_ws_kernel" : "");
    return TODA_EFAULT;
}
