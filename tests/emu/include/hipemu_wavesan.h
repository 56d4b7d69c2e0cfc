// TEST INFRASTRUCTURE -- what the emulator's scheduler (hipemu.cpp) tells the wave race detector (wavesan.cpp) about the code
// that is running on this OS thread.  Always maintained (a few stores per fibre switch); only the SAN=wavesan build reads it.
#pragma once
#include <cstdint>

namespace hipemu {
struct WaveSanState {
    bool in_kernel = false;        // a work-item's fibre is running (not the scheduler, not host code)
    bool sub_agent = false;        // the atomic / fence about to execute names a scope below "agent" (workgroup, wavefront): nothing another CU can observe
    uint32_t suppress = 0;         // > 0: the emulator's own code runs inside the work-item (the interpreter of the inline assembly)
    uint32_t launch = 0;           // serial number of the launch
    uint32_t block = 0;            // serial number of the workgroup (unique over the process: never 0)
    uint32_t wave = 0, lane = 0;   // of the running work-item
    uint32_t epoch = 0;            // how many __syncthreads this workgroup has passed
    uint32_t wave_epoch = 0;       // how many wave_barriers the running work-item's wave has passed (in this workgroup)
    const char *kernel = "";
    uintptr_t stack_lo = 0, stack_len = 0;   // the fibre stacks of this thread: lane-private memory
    uintptr_t ctx_lo = 0, ctx_len = 0;       // the work-items' own records (threadIdx, blockIdx, ...): registers on the GPU
    // this workgroup's LDS: the dynamic buffer, and the OS thread's block of the library's thread_local statics (= `__shared__`).
    // The next workgroup on this thread gets the same addresses -- and a different LDS on the GPU: not memory two workgroups share.
    uintptr_t dyn_lds_lo = 0, dyn_lds_len = 0, static_lds_lo = 0, static_lds_len = 0;
};
extern thread_local WaveSanState wavesan_state;
void wavesan_suppress(int delta);   // (hipemu.cpp: not instrumented)
void wavesan_scope(bool sub_agent); // the scope of the next atomic / fence of this work-item (consumed by it)
struct WaveSanSuppress {   // for the duration of a scope: what runs is the emulator's, its memory not the kernel's
    WaveSanSuppress() { wavesan_suppress(1); }
    ~WaveSanSuppress() { wavesan_suppress(-1); }
};
}  // namespace hipemu
