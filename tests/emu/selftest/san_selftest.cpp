// TEST INFRASTRUCTURE -- proves that the sanitizer build of the emulator (make SAN=asan) reports what it is there to report:
// a kernel's store past the end of a device buffer, a store past the dynamic LDS the launch asked for, a read of a freed
// device buffer, a misaligned vector access, a shift by the type's width -- and (the plain emulator's own check) a launch with more
// work-items per workgroup than the kernel's __launch_bounds__.  usage: san_selftest <mode>; "ok" does the same
// kinds of access inside the bounds and must exit 0 (tests/test_emu.py::test_sanitizer_build_reports_what_it_should).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void global_store(uint32_t *p, uint32_t at) {
    if (threadIdx.x == 0) p[at] = 1u;
}
__global__ void lds_store(uint32_t *out, uint32_t at) {
    unsigned char *const dyn = hipemu::dyn_lds();
    uint32_t *w = reinterpret_cast<uint32_t *>(dyn);
    w[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) { w[at] = 7u; out[0] = w[at] + w[63]; }
}
__global__ void global_load(const uint32_t *p, uint32_t *out) {
    if (threadIdx.x == 0) out[0] = p[0];
}
__global__ void vector_load(const unsigned char *p, uint32_t off, uint32_t *out) {
    if (threadIdx.x == 0) out[0] = reinterpret_cast<const uint4 *>(p + off)->x;
}
__global__ void shift(uint32_t v, uint32_t by, uint32_t *out) {
    if (threadIdx.x == 0) out[0] = v << by;
}

// what tests/emu/gen_sources.py puts at the top of a kernel that declares __launch_bounds__(64)
__global__ void bounded(uint32_t *out) {
    hipemu::check_launch_bound(64, "bounded");
    if (threadIdx.x == 0) out[0] = 1u;
}

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "ok";
    auto is = [&](const char *m) { return !strcmp(mode, m); };
    uint32_t *buf = nullptr, *out = nullptr;
    hipMalloc((void **)&buf, 256 * sizeof(uint32_t));
    hipMalloc((void **)&out, 256);
    hipLaunchKernelGGL(global_store, dim3(2), dim3(64), 0, nullptr, buf, is("heap") ? 256u : 255u);
    hipLaunchKernelGGL(lds_store, dim3(2), dim3(64), 64 * sizeof(uint32_t), nullptr, out, is("lds") ? 64u : 62u);
    hipLaunchKernelGGL(vector_load, dim3(1), dim3(64), 0, nullptr, (const unsigned char *)buf, is("align") ? 4u : 16u, out);
    hipLaunchKernelGGL(shift, dim3(1), dim3(64), 0, nullptr, 5u, is("shift") ? 32u : 31u, out);
    hipLaunchKernelGGL(bounded, dim3(1), dim3(is("bounds") ? 128 : 64), 0, nullptr, out);
    if (is("freed")) {
        hipFree(buf);
        hipLaunchKernelGGL(global_load, dim3(1), dim3(64), 0, nullptr, (const uint32_t *)buf, out);
        buf = nullptr;
    }
    hipDeviceSynchronize();
    hipFree(buf);
    hipFree(out);
    printf("san_selftest %s: nothing reported\n", mode);
    return 0;
}
