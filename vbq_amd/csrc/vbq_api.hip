// C-ABI plumbing: error string, ABI version, device queries.
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vbq_common.h"

namespace vbq {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        dev = 0;
    }
    if (cached[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;                                             // MI355X; only reached when the query itself fails
        }
        cached[dev] = n;
    }
    return cached[dev];
}
// Workgroup slots the resident grids leave free (for a collective's kernel running beside them): process-wide, default 0;
// VBQ_RESERVED_WORKGROUPS presets it (read once).
static std::atomic<int> g_reserved{-1};
int reserved_workgroups() {
    int v = g_reserved.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("VBQ_RESERVED_WORKGROUPS");
        v = e ? atoi(e) : 0;
        if (v < 0) v = 0;
        g_reserved.store(v, std::memory_order_relaxed);
    }
    return v;
}
int64_t resident_slots(int per_cu) {
    const int64_t all = (int64_t)num_cus() * per_cu;
    const int64_t left = all - reserved_workgroups();
    return left < num_cus() ? num_cus() : left;                  // never below one workgroup per CU
}
}  // namespace vbq

extern "C" int vbq_set_reserved_workgroups(int32_t n) {
    if (n < 0) {
        vbq::set_error("vbq_set_reserved_workgroups: n=%d is negative", n);
        return VBQ_ERR_INVALID_ARGUMENT;
    }
    vbq::g_reserved.store(n, std::memory_order_relaxed);
    return VBQ_OK;
}

extern "C" int vbq_abi_version(void) { return VBQ_ABI_VERSION; }

extern "C" const char *vbq_last_error(void) { return vbq::g_err; }

extern "C" int vbq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int vbq_device_name(int dev, char *buf, size_t buflen) {
    if (!buf || buflen == 0) {
        vbq::set_error("vbq_device_name: empty buffer");
        return VBQ_ERR_INVALID_ARGUMENT;
    }
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        vbq::set_error("vbq_device_name(%d): %s", dev, hipGetErrorString(e));
        return VBQ_ERR_INVALID_ARGUMENT;
    }
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return VBQ_OK;
}
