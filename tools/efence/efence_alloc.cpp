// An "electric fence" device allocator for PyTorch (torch.cuda.memory.CUDAPluggableAllocator): every allocation is its own mapping of
// whole granules inside a reserved address range that leaves an UNMAPPED granule on the guarded side, and the tensor is placed
// flush against that side -- a kernel that reads or writes past the tensor's end (EF_MODE=end, default) or before its start
// (EF_MODE=start) takes a GPU memory fault instead of silently touching a neighbour.  A measurement helper (tools/efence_run.py);
// allocations are slow (a reserve + map each), at least one granule (4 KiB here) plus two guard granules of address space, and never returned.
//   hipcc -O2 -fPIC -shared tools/efence/efence_alloc.cpp -o tools/efence/libefence.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
struct Rec { void* va; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; void* map_at; };
std::unordered_map<void*, Rec> g_live;
std::mutex g_mu;
size_t g_gran = 0;
bool g_start_mode = false;
long g_count = 0;

#define EF_CHECK(x)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "efence: %s failed: %s\n", #x, hipGetErrorString(e_));                    \
            abort();                                                                                  \
        }                                                                                             \
    } while (0)
}  // namespace

extern "C" void* ef_malloc(ssize_t size, int device, hipStream_t) {
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (!g_gran) {
        EF_CHECK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
        const char* m = getenv("EF_MODE");
        g_start_mode = m && !strcmp(m, "start");
        fprintf(stderr, "efence: granularity %zu bytes, guarding the %s of every allocation\n", g_gran, g_start_mode ? "start" : "end");
    }
    const size_t sz = ((size_t)size + 255) & ~(size_t)255;          // tensors keep the 256-byte alignment kernels may assume
    const size_t mapped = (sz + g_gran - 1) / g_gran * g_gran;
    Rec r;
    r.reserved = mapped + 2 * g_gran;                                // an unmapped granule on either side of the mapping
    r.mapped = mapped;
    EF_CHECK(hipMemAddressReserve(&r.va, r.reserved, 0, nullptr, 0));
    r.map_at = static_cast<char*>(r.va) + g_gran;
    EF_CHECK(hipMemCreate(&r.h, mapped, &prop, 0));
    EF_CHECK(hipMemMap(r.map_at, mapped, 0, r.h, 0));
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    EF_CHECK(hipMemSetAccess(r.map_at, mapped, &acc, 1));
    void* p = g_start_mode ? r.map_at : static_cast<char*>(r.map_at) + (mapped - sz);
    g_live[p] = r;
    ++g_count;
    return p;
}

extern "C" void ef_free(void* ptr, ssize_t, int, hipStream_t) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) {
        fprintf(stderr, "efence: free of an unknown pointer %p\n", ptr);
        return;
    }
    const Rec r = it->second;
    g_live.erase(it);
    // Default: never unmap (short runs; no address is ever reused).  EF_UNMAP=1 unmaps on free -- on ROCm 7.0 a later mapping that got the
    // same address range back then showed stale data to kernels (torch's own conv1d came out wrong too), so it is not the default.
    static const bool unmap = getenv("EF_UNMAP") != nullptr;
    if (!unmap) return;
    EF_CHECK(hipDeviceSynchronize());                               // nothing may still be running on the mapping
    EF_CHECK(hipMemUnmap(r.map_at, r.mapped));
    EF_CHECK(hipMemRelease(r.h));
    EF_CHECK(hipMemAddressFree(r.va, r.reserved));
}
