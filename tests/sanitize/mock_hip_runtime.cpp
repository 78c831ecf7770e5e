// tests/sanitize/mock_hip_runtime.cpp — TEST INFRASTRUCTURE for tools/sanitize_host.sh, never linked into the product.
//
// A host-memory stand-in for the few dozen HIP runtime entry points libgpvecchia_hip's HOST code calls, so that this code
// — plan construction, Morton ordering, the 128-bit threaded hash, the plan cache, level scheduling and symbolic fill of
// the posterior pass, graph capture bookkeeping, the staged copies — can run under AddressSanitizer / UBSan /
// ThreadSanitizer on a machine without a GPU (sanitizers are CPU-only on this pool).  "Device" memory is calloc'ed host
// memory, copies are memcpy, streams / events / graphs are opaque tokens, and a kernel launch does NOTHING but count:
// device buffers therefore hold zeros, numeric results are meaningless, and the driver checks statuses and bounds, not values.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
std::atomic<long> g_launches{0}, g_allocs{0}, g_frees{0}, g_graph_launches{0};
std::mutex g_mu;
std::unordered_map<void *, size_t> g_live;                      // device + pinned allocations still alive
thread_local int t_capturing = 0;
thread_local dim3 t_grid, t_block;
thread_local size_t t_shmem = 0;
thread_local hipStream_t t_stream = nullptr;
struct Token { int kind; };
void *track(size_t bytes)
{
    void *p = calloc(bytes ? bytes : 1, 1);
    if (!p) return nullptr;
    std::lock_guard<std::mutex> g(g_mu);
    g_live[p] = bytes;
    ++g_allocs;
    return p;
}
hipError_t untrack(void *p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) return hipErrorInvalidValue;     // double free / foreign pointer: reported, ASan shows the rest
        g_live.erase(it);
        ++g_frees;
    }
    free(p);
    return hipSuccess;
}
}  // namespace

extern "C" {
long mockhip_launches(void) { return g_launches.load(); }
long mockhip_graph_launches(void) { return g_graph_launches.load(); }
long mockhip_live_allocations(void) { std::lock_guard<std::mutex> g(g_mu); return (long)g_live.size(); }

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->name, "mock gfx950");
    std::strcpy(p->gcnArchName, "gfx950");
    p->multiProcessorCount = 256;
    p->warpSize = 64;
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 160 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->maxThreadsPerBlock = 1024;
    p->clockRate = 2400000;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "mock"; }
const char *hipGetErrorName(hipError_t) { return "mock"; }

hipError_t hipMalloc(void **p, size_t bytes) { *p = track(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { return untrack(p); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = track(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { return untrack(p); }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
    if (n && !t_capturing) std::memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n && !t_capturing) std::memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(new Token{1}); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<Token *>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { t_capturing = 1; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g)
{
    t_capturing = 0;
    *g = reinterpret_cast<hipGraph_t>(new Token{2});
    return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st)
{
    *st = t_capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t)
{
    *e = reinterpret_cast<hipGraphExec_t>(new Token{3});
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete reinterpret_cast<Token *>(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete reinterpret_cast<Token *>(e); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { ++g_graph_launches; return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new Token{4}); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Token *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.125f; return hipSuccess; }

hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s)
{
    t_grid = g; t_block = b; t_shmem = sh; t_stream = s;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *sh, hipStream_t *s)
{
    *g = t_grid; *b = t_block; *sh = t_shmem; *s = t_stream;
    return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
}
