// C-ABI plumbing: error string, ABI version, device queries.
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
// Workgroup slots a resident grid leaves free for a kernel of another stream: a PER-CALL argument of the entry points that launch
// resident grids (vbq_quantize_rows_f32, vbq_level_counts_f32, vbq_build_entropy_models_f32).  A negative argument takes the
// default, which VBQ_RESERVED_WORKGROUPS presets (read once, never written afterwards: no mutable process state).
int default_reserved_workgroups() {
    static const int v = [] {
        const char *e = getenv("VBQ_RESERVED_WORKGROUPS");
        const int n = e ? atoi(e) : 0;
        return n < 0 ? 0 : n;
    }();
    return v;
}
int64_t resident_slots(int per_cu, int reserved) {
    const int64_t all = (int64_t)num_cus() * per_cu;
    const int64_t left = all - (reserved > 0 ? reserved : 0);
    return left < num_cus() ? num_cus() : left;                  // never below one workgroup per CU
}
}  // namespace vbq

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
