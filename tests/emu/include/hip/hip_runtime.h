// TEST INFRASTRUCTURE -- not part of the product, never loaded by galah_amd.
//
// A stand-in for <hip/hip_runtime.h> that lets the library's own sources (galah_amd/csrc/*.hip, *.cpp) be compiled for
// the host CPU and run under a wave64 emulator (tests/emu/hipemu.cpp): every work-item of a workgroup is a fibre, the 64
// lanes of a wavefront meet at every cross-lane operation (__shfl*, __ballot, readfirstlane, wave_barrier), the waves
// of a workgroup meet at __syncthreads, workgroups run on a pool of OS threads (so global atomics are real atomics).
// Purpose: the container the code is written in has no GPU; this runs the kernels' LOGIC (indexing, LDS layout,
// barriers, cross-lane data flow, atomics protocols, host orchestration) bit-exactly against the oracle on the CPU.
// What it cannot show: timing, occupancy, register pressure, memory-model races between waves that x86 ordering hides.
#pragma once
#define HIPEMU 1
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>

// ------------------------------------------------------------------ host runtime API (the subset the library uses)
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
struct hipemuStream;
struct hipemuEvent;
typedef hipemuStream *hipStream_t;
typedef hipemuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    size_t maxSharedMemoryPerMultiProcessor;
    int multiProcessorCount;
    int warpSize;
    int maxThreadsPerBlock;
    int clockRate;
    int l2CacheSize;
};

extern "C" {
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceCanAccessPeer(int *can, int d, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpyPeerAsync(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t s);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t hipMemset(void *dst, int value, size_t bytes);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipFuncSetAttribute(const void *fn, hipFuncAttribute attr, int value);
}
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void *dst, const void *sym, size_t bytes, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
    memcpy(dst, (const char *)sym + off, bytes);
    return hipSuccess;
}
static inline hipError_t hipMemcpyToSymbol(void *sym, const void *src, size_t bytes, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
    memcpy((char *)sym + off, src, bytes);
    return hipSuccess;
}

// ------------------------------------------------------------------ the emulator's device side
namespace hipemu {

struct ThreadCtx {   // one work-item
    dim3 tid, bid, bdim, gdim;
    uint32_t lane, wave, flat;
    void *worker;   // the emulator's per-OS-thread state
    // the cross-lane operation this lane waits in
    uint64_t op_val, op_res;
    const void *op_site;
    int op_kind, op_arg, op_width;
};
// A fibre's stack is HIPEMU_STACK_BYTES long and aligned to that: the work-item's context pointer sits in its lowest word
// and is found from the stack pointer (no thread-local lookup at every threadIdx; a stack overflow lands on it first).
constexpr uintptr_t STACK_BYTES = 256 * 1024;
inline ThreadCtx *cur() {
    uintptr_t sp;
    asm("mov %%rsp, %0" : "=r"(sp));
    return *reinterpret_cast<ThreadCtx **>(sp & ~(STACK_BYTES - 1));
}

// a kernel compiled for at most `bound` work-items per workgroup that is launched with more: the GPU refuses the launch
[[noreturn]] void launch_bound_exceeded(const char *kernel, unsigned threads, unsigned bound);
inline void check_launch_bound(unsigned bound, const char *kernel) {
    const dim3 &b = cur()->bdim;
    const unsigned t = b.x * b.y * b.z;
    if (__builtin_expect(t > bound, 0)) launch_bound_exceeded(kernel, t, bound);
}

enum { OP_SHFL = 1, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_BALLOT, OP_FIRST, OP_WAVE_BARRIER };
// (convergent: what keeps a GPU compiler from duplicating or sinking a cross-lane operation into divergent branches keeps the
// host compiler from giving one operation two call sites, which the emulator would match separately)
__attribute__((convergent)) uint64_t wave_op(int kind, uint64_t val, int arg, int width, const void *site);
__attribute__((convergent)) void block_barrier();
void yield();   // s_sleep inside a spin loop: lets the other workgroups run
unsigned char *dyn_lds();
void launch(const char *name, const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t stream, std::function<void()> body);

template <class T> inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "shuffle of a type wider than 64 bits");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T> inline T from_bits(uint64_t b) {
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}
}  // namespace hipemu

#define threadIdx (hipemu::cur()->tid)
#define blockIdx (hipemu::cur()->bid)
#define blockDim (hipemu::cur()->bdim)
#define gridDim (hipemu::cur()->gdim)
static constexpr int warpSize = 64;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline, convergent))
#define __launch_bounds__(...)   // (gen_sources.py turns the bound into a check at the top of the kernel: hipemu::check_launch_bound)
#define __shared__ static thread_local   // one OS thread runs one workgroup at a time: its statics are the group's LDS
#define __constant__

// (kernel) may be a parenthesised template-id; the arguments are evaluated once, converted at the call like a real launch
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                            \
        auto hipemu_args_ = std::make_tuple(__VA_ARGS__);                                                           \
        hipemu::launch(#kernel, (const void *)(kernel), (grid), (block), (size_t)(lds), (stream),                   \
                       [=]() { std::apply(kernel, hipemu_args_); });                                                \
    } while (0)

// ------------------------------------------------------------------ vector types
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(4) ushort2 { unsigned short x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }

// ------------------------------------------------------------------ barriers, fences
__attribute__((convergent)) static inline void __syncthreads() { hipemu::block_barrier(); }
// Every fence is a full host fence; what the race detector build (wavesan.cpp) additionally learns is the SCOPE the kernel named: a
// fence or atomic of workgroup / wavefront scope orders nothing another CU can observe, so it does not count as the release or
// acquire of a hand-over between workgroups (MI355X: per-XCD L2s, a CU's L1 never refreshed by another CU's stores).
namespace hipemu { void wavesan_scope(bool sub_agent); }
static inline void hipemu_fence(const char *scope) {
    hipemu::wavesan_scope(scope[0] == 'w');   // "workgroup", "wavefront"; "agent" and "" (system) are seen by every CU
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
}
static inline void __threadfence() { hipemu_fence("agent"); }
static inline void __threadfence_block() { hipemu_fence("workgroup"); }
#define __builtin_amdgcn_fence(order, scope) hipemu_fence(scope)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_sleep(n) hipemu::yield()
__attribute__((noinline, convergent)) static void hipemu_wave_barrier() { (void)hipemu::wave_op(hipemu::OP_WAVE_BARRIER, 0, 0, 64, __builtin_return_address(0)); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()

// ------------------------------------------------------------------ cross-lane operations (wave64)
// `site` tells apart two operations that lanes of one wave wait in at the same time (divergent code): the emulator lets
// the group at the lower code address go first -- an if-body or a loop body before the code behind it.
template <class T> __attribute__((noinline, convergent)) T __shfl(T v, int src, int width = 64) {
    return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL, hipemu::to_bits(v), src, width, __builtin_return_address(0)));
}
template <class T> __attribute__((noinline, convergent)) T __shfl_up(T v, unsigned delta, int width = 64) {
    return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL_UP, hipemu::to_bits(v), (int)delta, width, __builtin_return_address(0)));
}
template <class T> __attribute__((noinline, convergent)) T __shfl_down(T v, unsigned delta, int width = 64) {
    return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL_DOWN, hipemu::to_bits(v), (int)delta, width, __builtin_return_address(0)));
}
template <class T> __attribute__((noinline, convergent)) T __shfl_xor(T v, int mask, int width = 64) {
    return hipemu::from_bits<T>(hipemu::wave_op(hipemu::OP_SHFL_XOR, hipemu::to_bits(v), mask, width, __builtin_return_address(0)));
}
__attribute__((noinline, convergent)) static unsigned long long __ballot(int pred) {
    return hipemu::wave_op(hipemu::OP_BALLOT, pred != 0, 0, 64, __builtin_return_address(0));
}
__attribute__((noinline, convergent)) static unsigned long long hipemu_ballot_w64(bool pred) {
    return hipemu::wave_op(hipemu::OP_BALLOT, pred, 0, 64, __builtin_return_address(0));
}
__attribute__((noinline, convergent)) static int __any(int pred) { return hipemu::wave_op(hipemu::OP_BALLOT, pred != 0, 0, 64, __builtin_return_address(0)) != 0; }
__attribute__((noinline, convergent)) static int __all(int pred) { return hipemu::wave_op(hipemu::OP_BALLOT, pred == 0, 0, 64, __builtin_return_address(0)) == 0; }
__attribute__((noinline, convergent)) static uint32_t hipemu_readfirstlane(uint32_t v) {
    return (uint32_t)hipemu::wave_op(hipemu::OP_FIRST, v, 0, 64, __builtin_return_address(0));
}
__attribute__((noinline, convergent)) static uint32_t hipemu_readlane(uint32_t v, uint32_t lane) {   // v_readlane_b32: lane is uniform
    return (uint32_t)hipemu::wave_op(hipemu::OP_SHFL, v, (int)(lane & 63u), 64, __builtin_return_address(0));
}
#define __builtin_amdgcn_readlane(v, l) hipemu_readlane((v), (l))
#define __builtin_amdgcn_ballot_w64(p) hipemu_ballot_w64(p)
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
static inline uint32_t hipemu_mbcnt_lo(uint32_t mask, uint32_t base) {
    const uint32_t l = hipemu::cur()->lane;
    return base + (uint32_t)__builtin_popcount(mask & (l >= 32 ? 0xffffffffu : ((1u << l) - 1u)));
}
static inline uint32_t hipemu_mbcnt_hi(uint32_t mask, uint32_t base) {
    const uint32_t l = hipemu::cur()->lane;
    return base + (uint32_t)__builtin_popcount(mask & (l <= 32 ? 0u : ((1u << (l - 32)) - 1u)));
}
#define __builtin_amdgcn_mbcnt_lo(m, b) hipemu_mbcnt_lo((m), (b))
#define __builtin_amdgcn_mbcnt_hi(m, b) hipemu_mbcnt_hi((m), (b))
static inline uint32_t __lane_id() { return hipemu::cur()->lane; }

// ------------------------------------------------------------------ gfx950 integer builtins the kernels name directly
static inline uint32_t hipemu_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {   // v_alignbit_b32: ({hi, lo} >> sh[4:0])[31:0]
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
}
static inline uint32_t hipemu_perm(uint32_t a, uint32_t b, uint32_t sel) {   // v_perm_b32: bytes 0-3 of b, 4-7 of a
    const uint64_t src = ((uint64_t)a << 32) | b;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xffu;
        uint32_t byte;
        if (s <= 7) byte = (uint32_t)(src >> (8 * s)) & 0xffu;
        else if (s == 8) byte = (b >> 15) & 1u ? 0xffu : 0u;     // sign of b[15:0] etc.: the kernels do not use these, kept for completeness
        else if (s == 9) byte = (b >> 31) & 1u ? 0xffu : 0u;
        else if (s == 10) byte = (a >> 15) & 1u ? 0xffu : 0u;
        else if (s == 11) byte = (a >> 31) & 1u ? 0xffu : 0u;
        else if (s == 12) byte = 0u;
        else byte = 0xffu;
        r |= byte << (8 * i);
    }
    return r;
}
static inline uint32_t hipemu_ubfe(uint32_t v, uint32_t off, uint32_t width) {   // v_bfe_u32
    off &= 31u; width &= 31u;
    return width == 0 ? 0u : (v >> off) & ((1u << width) - 1u);
}
#define __builtin_amdgcn_alignbit(h, l, s) hipemu_alignbit((h), (l), (s))
#define __builtin_amdgcn_perm(a, b, s) hipemu_perm((a), (b), (s))
#define __builtin_amdgcn_ubfe(v, o, w) hipemu_ubfe((v), (o), (w))

static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline uint32_t __brev(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(v);
}
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
static inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

// min / max of the integer types device code mixes (HIP declares these in the global namespace)
#define HIPEMU_MINMAX(T)                                        \
    static inline T min(T a, T b) { return b < a ? b : a; }     \
    static inline T max(T a, T b) { return a < b ? b : a; }
HIPEMU_MINMAX(int)
HIPEMU_MINMAX(unsigned int)
HIPEMU_MINMAX(long)
HIPEMU_MINMAX(unsigned long)
HIPEMU_MINMAX(long long)
HIPEMU_MINMAX(unsigned long long)
HIPEMU_MINMAX(float)
HIPEMU_MINMAX(double)
#undef HIPEMU_MINMAX
static inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
static inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
static inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
static inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
static inline unsigned long min(unsigned long a, unsigned int b) { return min(a, (unsigned long)b); }
static inline unsigned long min(unsigned int a, unsigned long b) { return min((unsigned long)a, b); }
static inline unsigned long max(unsigned long a, unsigned int b) { return max(a, (unsigned long)b); }
static inline unsigned long max(unsigned int a, unsigned long b) { return max((unsigned long)a, b); }
static inline unsigned long long min(unsigned long long a, unsigned int b) { return min(a, (unsigned long long)b); }
static inline unsigned long long min(unsigned int a, unsigned long long b) { return min((unsigned long long)a, b); }
static inline unsigned long long max(unsigned long long a, unsigned int b) { return max(a, (unsigned long long)b); }
static inline unsigned long long max(unsigned int a, unsigned long long b) { return max((unsigned long long)a, b); }

// ------------------------------------------------------------------ atomics (workgroups run on several OS threads)
template <class T, class U> static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicSub(T *p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicXor(T *p, U v) { return __atomic_fetch_xor(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U, class V> static inline T atomicCAS(T *p, U cmp, V v) {
    T expected = (T)cmp;
    __atomic_compare_exchange_n(p, &expected, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected;
}
template <class T, class U> static inline T atomicMax(T *p, U v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T, class U> static inline T atomicMin(T *p, U v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v < old && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define HIPEMU_SCOPE(scope) hipemu::wavesan_scope((scope) < __HIP_MEMORY_SCOPE_AGENT)
#define __hip_atomic_load(p, order, scope) (HIPEMU_SCOPE(scope), __atomic_load_n((p), (order)))
#define __hip_atomic_store(p, v, order, scope) (HIPEMU_SCOPE(scope), __atomic_store_n((p), (v), (order)))
#define __hip_atomic_fetch_add(p, v, order, scope) (HIPEMU_SCOPE(scope), __atomic_fetch_add((p), (v), (order)))
#define __hip_atomic_fetch_or(p, v, order, scope) (HIPEMU_SCOPE(scope), __atomic_fetch_or((p), (v), (order)))
#define __hip_atomic_exchange(p, v, order, scope) (HIPEMU_SCOPE(scope), __atomic_exchange_n((p), (v), (order)))
#define __hip_atomic_compare_exchange_strong(p, expected, desired, success, failure, scope) \
    (HIPEMU_SCOPE(scope), __atomic_compare_exchange_n((p), (expected), (desired), false, (success), (failure)))
