// TEST INFRASTRUCTURE -- proves that the wave race detector (make SAN=wavesan, tests/emu/wavesan.cpp) reports what it is there
// to report and nothing else: a __syncthreads missing behind a producer wave (in LDS and in global memory), a flag handed
// between workgroups without release / acquire fences -- and stays silent on the same kernels written correctly.
// usage: wavesan_selftest <mode>; prints the detector's counts (tests/test_emu.py::test_wave_race_detector_reports_what_it_should).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

extern "C" void hipemu_wavesan_counts(uint64_t out[5]);

// wave 0 produces, wave 1 consumes
__global__ void lds_handover(uint32_t *out, int barrier) {
    __shared__ uint32_t buf[64];
    if (threadIdx.x < 64) buf[threadIdx.x] = threadIdx.x * 3u;
    if (barrier) __syncthreads();
    if (threadIdx.x >= 64) out[blockIdx.x * 64 + threadIdx.x - 64] = buf[threadIdx.x - 64];
}
__global__ void global_handover(uint32_t *scratch, uint32_t *out, int barrier) {
    uint32_t *mine = scratch + blockIdx.x * 64;
    if (threadIdx.x < 64) mine[threadIdx.x] = threadIdx.x * 5u;
    if (barrier) __syncthreads();
    if (threadIdx.x >= 64) out[blockIdx.x * 64 + threadIdx.x - 64] = mine[127 - threadIdx.x];
}
// neighbouring lanes of one wave exchange through LDS: right with a wavefront fence + wave_barrier between store and load; without,
// the compiler may emit the load first (lds[t] and lds[t ^ 1] never alias for one work-item) -- the opt-in lane rule (WAVESAN_LANES=1)
__global__ void lane_exchange(uint32_t *out, int sync) {
    __shared__ uint32_t buf[64];
    buf[threadIdx.x] = threadIdx.x * 7u;
    if (sync) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x ^ 1u];
}
// every wave writes its own part of LDS, atomics on a shared counter: nothing to report
__global__ void disjoint_and_atomic(uint32_t *out) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t count;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    part[threadIdx.x] = threadIdx.x;
    atomicAdd(&count, 1u);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = count + part[255];
}
// workgroup 0 publishes a value for workgroup 1 through a flag
// fences: 0 none, 1 agent scope (right), 2 workgroup scope (a full fence on x86, nothing another CU observes on the GPU)
__global__ void flag_handover(uint32_t *data, uint32_t *flag, uint32_t *out, int fences) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            data[0] = 42u;
            if (fences == 1) __threadfence();
            if (fences == 2) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            atomicAdd(flag, 1u);
        }
    } else if (threadIdx.x == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
        if (fences == 1) __threadfence();
        if (fences == 2) __threadfence_block();
        out[0] = data[0];
    }
}

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "ok";
    auto is = [&](const char *m) { return !strcmp(mode, m); };
    uint32_t *a = nullptr, *b = nullptr, *f = nullptr;
    hipMalloc((void **)&a, 4096);
    hipMalloc((void **)&b, 4096);
    hipMalloc((void **)&f, 64);
    hipMemset(f, 0, 64);
    hipLaunchKernelGGL(lds_handover, dim3(3), dim3(128), 0, nullptr, b, is("lds") ? 0 : 1);
    hipLaunchKernelGGL(global_handover, dim3(3), dim3(128), 0, nullptr, a, b, is("global") ? 0 : 1);
    hipLaunchKernelGGL(lane_exchange, dim3(2), dim3(64), 0, nullptr, b, is("lanes") ? 0 : 1);
    hipLaunchKernelGGL(disjoint_and_atomic, dim3(3), dim3(256), 0, nullptr, b);
    hipLaunchKernelGGL(flag_handover, dim3(2), dim3(64), 0, nullptr, a, f, b, is("flag") ? 0 : is("scope") ? 2 : 1);
    hipDeviceSynchronize();
    uint64_t c[5];
    hipemu_wavesan_counts(c);
    printf("wavesan_selftest %s: write-write %llu read-write %llu inter-block %llu lanes %llu (accesses checked: %llu)\n", mode, (unsigned long long)c[0],
           (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[4], (unsigned long long)c[3]);
    return 0;
}
