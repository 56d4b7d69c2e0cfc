// Host-side HIP costs that shape the ingest pipeline: pinned allocation, stream/event creation, H2D rates.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0); hipFree(0);
    for (size_t mb : {1, 8, 64, 128, 640}) {
        void *p; double t = now(); hipHostMalloc(&p, mb << 20, hipHostMallocDefault); double a = now() - t;
        t = now(); memset(p, 1, mb << 20); double m = now() - t;
        t = now(); hipHostFree(p); double f = now() - t;
        printf("hipHostMalloc %4zu MB: alloc %.2f ms, first touch %.2f ms, free %.2f ms\n", mb, a * 1e3, m * 1e3, f * 1e3);
    }
    { double t = now(); std::vector<hipStream_t> s(64); for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
      double c = now() - t; t = now(); for (auto &x : s) hipStreamDestroy(x); printf("64 streams: create %.2f ms destroy %.2f ms\n", c * 1e3, (now() - t) * 1e3); }
    { double t = now(); std::vector<hipEvent_t> e(64); for (auto &x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
      printf("64 events: create %.2f ms\n", (now() - t) * 1e3); }
    void *d; hipMalloc(&d, 1 << 30);
    void *pin; hipHostMalloc(&pin, 256 << 20, hipHostMallocDefault); memset(pin, 1, 256 << 20);
    void *pg = malloc(256 << 20); memset(pg, 1, 256 << 20);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpyAsync(d, pin, 256 << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        printf("H2D pinned 256 MB: %.1f GB/s\n", 0.268 / (now() - t));
        t = now(); hipMemcpyAsync(d, pg, 256 << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        printf("H2D pageable 256 MB: %.1f GB/s\n", 0.268 / (now() - t));
        t = now(); for (int i = 0; i < 256; i++) hipMemcpyAsync((char *)d + ((size_t)i << 20), (char *)pin + ((size_t)i << 20), 1 << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        printf("H2D pinned 256 x 1 MB: %.1f GB/s\n", 0.268 / (now() - t));
    }
    double t = now(); memcpy(pin, pg, 256 << 20); printf("memcpy pageable->pinned 256 MB: %.1f GB/s\n", 0.268 / (now() - t));
    return 0;
}
