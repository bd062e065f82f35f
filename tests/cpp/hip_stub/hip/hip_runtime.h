// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets the HOST orchestration of libgarden_vis (gv_context.cpp, gv_mirror.cpp,
// gv_exchange.cpp — 2.8 k lines that are otherwise only ever compiled as HIP) be built as plain C++ and run under
// AddressSanitizer / UndefinedBehaviorSanitizer on a box without a GPU (tests/cpp/host_orchestration_test.cpp). "Device"
// memory is zero-filled host memory, copies are memcpy (so a copy that overruns either side is an ASan report), streams
// and events do nothing, kernels are no-ops (kernel_stubs.cpp, generated from the launch headers). Never part of the
// product: the real library is built by garden_amd/csrc/Makefile against the real runtime and returns GV_E_NODEVICE
// without a gfx950 device.
#pragma once
#define GV_HIP_STUB 1
#define __host__
#define __device__
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorPeerAccessAlreadyEnabled = 704 };
typedef struct HipStubStream* hipStream_t;
typedef struct HipStubEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipHostMallocDefault = 0, hipHostRegisterDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };

struct hipDeviceProp_t { char gcnArchName[256]; };

static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::strcpy(p->gcnArchName, "gfx950:stub"); return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }

// Fault injection (tests/cpp/host_orchestration_test.cpp::allocation_failures): when the countdown reaches 1 the allocation it
// lands on fails with hipErrorOutOfMemory (once). 0: never. One counter for every translation unit of the stub build.
// (atomics: the exchange tests allocate from several rank threads at once; the countdown itself is only armed by single-threaded tests)
struct GvStubCounter {
    std::atomic<long> value{0};
    GvStubCounter& operator=(long v) { value.store(v, std::memory_order_relaxed); return *this; }
    operator long() const { return value.load(std::memory_order_relaxed); }
    long operator++(int) { return value.fetch_add(1, std::memory_order_relaxed); }
};
inline GvStubCounter& gv_stub_fail_countdown() { static GvStubCounter countdown; return countdown; }
inline GvStubCounter& gv_stub_allocations() { static GvStubCounter count; return count; }
static inline bool gv_stub_allocation_fails()
{
    gv_stub_allocations()++;
    std::atomic<long>& c = gv_stub_fail_countdown().value;
    return c.load(std::memory_order_relaxed) > 0 && c.fetch_sub(1, std::memory_order_relaxed) == 1;
}
static inline hipError_t hipMalloc(void** p, size_t n)
{
    *p = nullptr;
    if (gv_stub_allocation_fails())
        return hipErrorOutOfMemory;
    *p = std::calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    *p = nullptr;
    if (gv_stub_allocation_fails())
        return hipErrorOutOfMemory;
    *p = std::calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostRegister(void* p, size_t n, unsigned)
{   // touch both ends: a range the caller does not own is a sanitizer report here, where the real call would pin it
    if (n) { volatile char* c = static_cast<volatile char*>(p); (void)c[0]; (void)c[n - 1]; }
    return hipSuccess;
}
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { if (n) std::memmove(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) { if (n) std::memset(dst, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* dst, int v, size_t n) { if (n) std::memset(dst, v, n); return hipSuccess; }
typedef void* hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t words, hipStream_t)
{
    for (size_t k = 0; k < words; k++)
        std::memcpy(static_cast<char*>(dst) + 4 * k, &v, 4);
    return hipSuccess;
}

static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// "Work that never finishes" for the bounded waits of the exchange (host_orchestration_test.cpp::exchange_bounded_waits): bit 0 — no
// stream ever drains, bit 1 — no event ever completes. 0: everything has always run already (the stub executes at enqueue).
inline GvStubCounter& gv_stub_never_ready() { static GvStubCounter bits; return bits; }
static inline hipError_t hipStreamQuery(hipStream_t) { return ((long)gv_stub_never_ready() & 1) ? hipErrorNotReady : hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return ((long)gv_stub_never_ready() & 2) ? hipErrorNotReady : hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(1)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
