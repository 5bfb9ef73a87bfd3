// TEST INFRASTRUCTURE (not product code): a fake HIP runtime for the CPU-only sanitizer build of the product's HOST code.
//
// conicip.jl_amd/csrc/*.hip are compiled with `hipcc --cuda-host-only -fsanitize=...` (the host half of every translation unit:
// handle / arena management, the lock-step driver, the thread pool of the batch entry points, the native interior-point loop, the
// launch sequences of the factorisation) and linked against THIS file instead of libamdhip64: "device" memory is host memory that
// ALWAYS READS ZERO (a copy into it is carried out -- so that the sanitizer checks both ranges -- and then zeroed again: no kernel
// runs, so nothing a kernel would have overwritten may survive in a recycled arena), pinned host memory is ordinary memory, copies
// are memmove, streams are in order and synchronous, events are always complete, and a kernel launch is a NO-OP --
// except the two one-wave kernels whose only job is to hand a word to a polling host (k_publish_info of api.hip, k_lg_pubflag of
// sdp_large.hip), which are emulated so that the host's polls end the way they do on the GPU.  With every "device" result zero the
// interior-point loop sees zero residuals and stops at its first convergence test: what runs under the sanitizers is the control
// plane, not the arithmetic (tests/hostsan/drive.cpp says what is driven through it).
//
// Nothing here is used by, linked into or shipped with the product library (tests/test_host_sanitizers.py is the only user).
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
std::mutex g_mu;
std::map<const void *, std::string> &kernels() { static std::map<const void *, std::string> m; return m; }
std::map<void *, size_t> &allocs() { static std::map<void *, size_t> m; return m; }       // every live allocation
std::map<char *, size_t> &device_ranges() { static std::map<char *, size_t> m; return m; }   // the hipMalloc'ed ones
std::atomic<long> g_launches{0}, g_emulated{0}, g_bytes_live{0};
thread_local hipError_t tl_last = hipSuccess;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> tl_cfg;
struct FakeStream { int tag; };
struct FakeEvent { int tag; };

void *dev_alloc(size_t bytes, bool device) {
    void *p = calloc(bytes ? bytes : 1, 1);
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    allocs()[p] = bytes;
    if (device) device_ranges()[(char *)p] = bytes;
    g_bytes_live += (long)bytes;
    return p;
}
bool in_device(const void *q) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = device_ranges().upper_bound((char *)q);
    if (it == device_ranges().begin()) return false;
    --it;
    return (const char *)q < it->first + it->second;
}
void copy(void *dst, const void *src, size_t n) {
    if (!n) return;
    memmove(dst, src, n);                      // (the sanitizer sees both ranges)
    if (in_device(dst)) memset(dst, 0, n);     // "device" memory always reads zero
}
hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = allocs().find(p);
        if (it == allocs().end()) { fprintf(stderr, "fake_hip: free of a pointer that was never allocated: %p\n", p); abort(); }
        g_bytes_live -= (long)it->second;
        allocs().erase(it);
        device_ranges().erase((char *)p);
    }
    free(p);
    return hipSuccess;
}
}  // namespace

extern "C" {
// ---- registration (what the compiler-generated module constructor of every translation unit calls)
void **__hipRegisterFatBinary(const void *) { static void *h[1] = {nullptr}; return h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFun, char *, const char *deviceName, unsigned, void *, void *, void *, void *, int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    kernels()[hostFun] = deviceName ? deviceName : "";
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s) { tl_cfg.push_back({grid, block, shmem, s}); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *s) {
    if (tl_cfg.empty()) { fprintf(stderr, "fake_hip: pop of an empty launch configuration stack\n"); abort(); }
    const CallCfg c = tl_cfg.back(); tl_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *s = c.stream;
    return hipSuccess;
}

// ---- launches: no-ops, but the configuration is checked the way the hardware would refuse it, and the two publishing kernels run
hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t) {
    g_launches += 1;
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 ||
        shmem > 160 * 1024 || grid.y > 65535 || grid.z > 65535) {
        fprintf(stderr, "fake_hip: invalid launch configuration grid (%u, %u, %u) block (%u, %u, %u) shmem %zu\n", grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
        abort();
    }
    std::string name;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = kernels().find(fn);
        if (it == kernels().end()) { fprintf(stderr, "fake_hip: launch of an unregistered kernel\n"); abort(); }
        name = it->second;
    }
    if (name.find("k_publish_info") != std::string::npos) {              // (const int *info, int *host, int seq)
        const int *info = *(const int **)args[0]; int *host = *(int **)args[1]; const int seq = *(int *)args[2];
        for (int q = 0; q < 4; ++q) host[q] = info[q];
        __atomic_store_n(host + 4, seq, __ATOMIC_RELEASE);
        g_emulated += 1;
    } else if (name.find("k_lg_pubflag") != std::string::npos) {         // (const unsigned *sweepflag, const int *info_a, const int *info_b, int *host, int seq)
        const unsigned *sf = *(const unsigned **)args[0]; const int *ia = *(const int **)args[1], *ib = *(const int **)args[2];
        int *host = *(int **)args[3]; const int seq = *(int *)args[4];
        host[0] = (int)(*sf != 0u); host[1] = (ia[0] != 0 || ib[0] != 0) ? 1 : 0;
        __atomic_store_n(host + 2, seq, __ATOMIC_RELEASE);
        g_emulated += 1;
    }
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// ---- memory
hipError_t hipMalloc(void **p, size_t bytes) { *p = dev_alloc(bytes, true); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { return dev_free(p); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = dev_alloc(bytes, false); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { return dev_free(p); }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { copy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { copy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) copy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) { if (n) memset(p, in_device(p) ? 0 : v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { if (n) memset(p, in_device(p) ? 0 : v, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void *dst, const void *, size_t n, size_t, hipMemcpyKind) { if (n) memset(dst, 0, n); return hipSuccess; }

// ---- device, streams, events: one device, in-order synchronous streams, events that are complete the moment they are recorded
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) { *v = (a == hipDeviceAttributeMultiprocessorCount) ? 256 : 0; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t) new FakeStream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t) new FakeStream{2}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete (FakeStream *)s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t) new FakeEvent{1}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t) new FakeEvent{2}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete (FakeEvent *)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipGetLastError(void) { const hipError_t e = tl_last; tl_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }

// ---- graphs: not offered (the library's replay of small systems is opt-in and falls back to plain launches)
hipError_t hipGraphCreate(hipGraph_t *, unsigned) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphAddKernelNode(hipGraphNode_t *, hipGraph_t, const hipGraphNode_t *, size_t, const hipKernelNodeParams *) { return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, hipGraphNode_t *, char *, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

// ---- for the driver program
void fake_hip_stats(long *launches, long *emulated, long *live_bytes, long *live_allocs) {
    *launches = g_launches.load(); *emulated = g_emulated.load(); *live_bytes = g_bytes_live.load();
    std::lock_guard<std::mutex> lk(g_mu);
    *live_allocs = (long)allocs().size();
}
}  // extern "C"
