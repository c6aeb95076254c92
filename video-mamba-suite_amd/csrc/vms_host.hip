// vms_host.hip -- host-side plumbing of libvms_hip.so shared by every entry point: the thread-local error / dispatch
// messages and the immutable per-device cache (vms_hip.h "Conventions": no environment reads, no mutable state).
#include <stdarg.h>
#include <stdio.h>

#include "vms_common.h"

namespace vms {

namespace {
thread_local char g_err[512] = "";
thread_local const char* g_kernel = "";
std::once_flag g_cu_once[kMaxDevices];
int g_cu[kMaxDevices];
int query_cu_count(int dev) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;  // MI355X
    return n;
}
}  // namespace

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }
void set_last_kernel(const char* name) { g_kernel = name; }

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return query_cu_count(0);
    std::call_once(g_cu_once[dev], [dev] { g_cu[dev] = query_cu_count(dev); });
    return g_cu[dev];
}

}  // namespace vms

extern "C" const char* vms_last_error(void) { return vms::last_error(); }
extern "C" const char* vms_last_kernel(void) { return vms::g_kernel; }
extern "C" int vms_abi_version(void) { return VMS_ABI_VERSION; }
extern "C" int vms_build_flags(void) { return 0; }   // (bit VMS_BUILD_EXPERIMENTAL: the kernel generations of rounds 1-3, removed in round 5)
extern "C" int vms_sizeof_scan_fwd_params(void) { return (int)sizeof(vms_scan_fwd_params); }
extern "C" int vms_sizeof_scan_bwd_params(void) { return (int)sizeof(vms_scan_bwd_params); }
extern "C" int vms_sizeof_conv_fwd_params(void) { return (int)sizeof(vms_conv_fwd_params); }
extern "C" int vms_sizeof_conv_bwd_params(void) { return (int)sizeof(vms_conv_bwd_params); }
