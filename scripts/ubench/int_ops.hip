// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer ops the
// sketch / seed kernels are made of.  Each kernel runs 8 independent dependency chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../galah_amd/csrc/murmur21_asm.h"
#define REP 4096
#define CHAINS 8

#define KERNEL(NAME, DECL, BODY, SINK)                                                          \
    __global__ void NAME(uint64_t *out, uint32_t seed) {                                        \
        DECL;                                                                                    \
        for (int i = 0; i < REP; i++) { BODY }                                                   \
        SINK;                                                                                    \
    }

KERNEL(k_add32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mul_lo, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mul_hi, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mul_u24, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mad64, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[c]) : "v"(seed) : "vcc");,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mad64_sgpr, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u64_u32 %0, s[40:41], %1, %1, %0" : "+v"(a[c]) : "v"(seed) : "s40", "s41");,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mad64_rot, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       asm volatile("v_mad_u64_u32 %0, s[40:41], %4, %4, %0\n v_mad_u64_u32 %1, s[42:43], %4, %4, %1\n v_mad_u64_u32 %2, s[44:45], %4, %4, %2\n v_mad_u64_u32 %3, s[46:47], %4, %4, %3"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(seed) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
       asm volatile("v_mad_u64_u32 %0, s[40:41], %4, %4, %0\n v_mad_u64_u32 %1, s[42:43], %4, %4, %1\n v_mad_u64_u32 %2, s[44:45], %4, %4, %2\n v_mad_u64_u32 %3, s[46:47], %4, %4, %3"
                    : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mad64_s0, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"((uint32_t)a[c]), "s"(seed) : "vcc"); a[c] = t; },
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mad64_s0_sgpr, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { uint64_t t; asm volatile("v_mad_u64_u32 %0, s[40:41], %1, %2, 0" : "=v"(t) : "v"((uint32_t)a[c]), "s"(seed) : "s40", "s41"); a[c] = t; },
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mul_lo_s, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "s"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mad_u32_u24, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
// does the cost of a cheap op depend on its neighbours?  8 alignbits and 8 xors per iteration, alternating or grouped
KERNEL(k_mix_alt, uint32_t a[CHAINS]; uint32_t b[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %2, 7\n v_xor_b32_e32 %1, %1, %2" : "+v"(a[c]), "+v"(b[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mix_grp, uint32_t a[CHAINS]; uint32_t b[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(seed));
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(b[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
// ... and on being independent of it?  one chain: alignbit feeds the xor feeds the next alignbit
KERNEL(k_mix_dep, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_xor_b32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mix_dep1, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_xor_b32_e32 %0, %0, %1" : "+v"(a[0]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
// pairs of cheap ops between expensive ones: independent, or the second using the first's result
KERNEL(k_mix_ecc, uint32_t a[CHAINS]; uint32_t b[CHAINS]; uint32_t d[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; d[c] = a[c] * 5; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %3, 7\n v_xor_b32_e32 %1, %1, %3\n v_xor_b32_e32 %2, %2, %3" : "+v"(a[c]), "+v"(b[c]), "+v"(d[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c] + d[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mix_ecc_dep, uint32_t a[CHAINS]; uint32_t b[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %2, 7\n v_lshrrev_b32_e32 %1, 1, %0\n v_xor_b32_e32 %0, %0, %1" : "+v"(a[c]), "+v"(b[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mix_ecc_dep1, uint32_t a[CHAINS]; uint32_t b[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %2, 7\n v_lshrrev_b32_e32 %1, 1, %0\n v_xor_b32_e32 %0, %0, %1" : "+v"(a[0]), "+v"(b[0]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mix_eecc1, uint32_t a[CHAINS]; uint32_t b[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %2, 7\n v_alignbit_b32 %1, %1, %2, 9\n v_xor_b32_e32 %0, %0, %2\n v_xor_b32_e32 %1, %1, %2" : "+v"(a[0]), "+v"(b[0]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mix_ecec1, uint32_t a[CHAINS]; uint32_t b[CHAINS]; for (int c = 0; c < CHAINS; c++) { a[c] = seed + c + threadIdx.x; b[c] = a[c] * 3; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %2, 7\n v_xor_b32_e32 %0, %0, %2\n v_alignbit_b32 %1, %1, %2, 9\n v_xor_b32_e32 %1, %1, %2" : "+v"(a[0]), "+v"(b[0]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c] + b[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_lshl_add64, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(a[c]));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_lshl64, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[c]));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_lshr64, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = ~0ull - seed - c - threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a[c]));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_alignbit, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_addc_pair, uint32_t lo[CHAINS]; uint32_t hi[CHAINS]; for (int c = 0; c < CHAINS; c++) { lo[c] = seed + c + threadIdx.x; hi[c] = c; },
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo[c]), "+v"(hi[c]) : "v"(seed) : "vcc");,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += lo[c] + hi[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_cmp64, uint64_t a[CHAINS]; uint32_t cnt = 0; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(a[c]), "v"(a[(c + 1) % CHAINS]) : "vcc");,
       out[blockIdx.x * blockDim.x + threadIdx.x] = cnt)
KERNEL(k_perm, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)

KERNEL(k_add32_e64, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_xor_e32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_xor_e64, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_xor_lit, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_xor_b32_e32 %0, 0x12345678, %0" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_lshr_e32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_and_e32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_cndmask, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(seed) : "vcc");,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_bfe, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_lshl_or, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_and_or, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_add3, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_bitop3, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x48" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_sdwa, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_or_sdwa, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_cmp32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(a[c]) : "v"(seed) : "vcc");,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_sub_e32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_min, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)
KERNEL(k_mov, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x,
       _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[c]) : "v"(seed));,
       uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s)

template <typename K>
static void run(const char *name, K kern, uint64_t *d_out, int insts_per_iter) {
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * (threads / 64) * REP * CHAINS * insts_per_iter;
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;  // 1024 SIMDs, nominal 2.4 GHz (DVFS makes this an upper bound)
    printf("%-14s %8.3f ms  %6.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, ms, simd_cycles / wave_insts);
}


// --- composite blocks of the sketch kernel -------------------------------------------------
#define REPB 1024
__global__ void k_hash_block(uint64_t *out, uint32_t seed) {   // the 55-instruction asm hash, dependent chain per lane
    uint32_t a0 = seed + threadIdx.x, a1 = seed * 3 + threadIdx.x, b0 = a0 ^ 0x1234567, b1 = a1 + 99, t0 = a0 * 7, t1 = a1 * 11;
    uint64_t acc = 0;
    for (int i = 0; i < REPB; i++) {
        uint64_t h = murmur21_core<true>(a0, a1, b0, b1, t0, t1, 0);
        acc += h; a0 = (uint32_t)h; a1 ^= (uint32_t)(h >> 32);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_filter_block(uint64_t *out, uint32_t seed) {  // what the hot loop runs per position: the filter form only
    uint32_t a0 = seed + threadIdx.x, a1 = seed * 3 + threadIdx.x, b0 = a0 ^ 0x1234567, b1 = a1 + 99, t0 = a0 * 7, t1 = a1 * 11;
    uint32_t acc = 0;
    for (int i = 0; i < REPB; i++) {
        uint64_t A, B;
        const uint32_t s1 = murmur21_filter<true>(a0, a1, b0, b1, t0, t1, 0, A, B);
        acc += s1; a0 = (uint32_t)A; a1 ^= (uint32_t)B;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_cmp_cnd(uint64_t *out, uint32_t seed) {      // v_cmp_lt_u64 + 2 v_cndmask (canonical select)
    uint64_t a = seed + threadIdx.x, b = seed * 77ull + threadIdx.x * 3, acc = 0;
    for (int i = 0; i < REPB * 8; i++) {
        uint32_t m, alo = (uint32_t)a, blo = (uint32_t)b;
        asm volatile("v_cmp_lt_u64 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %3, %4, vcc\n" : "=v"(m) : "v"(a), "v"(b), "v"(alo), "v"(blo) : "vcc");
        acc += m; a += acc; b ^= a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_select_c(uint64_t *out, uint32_t seed) {     // what hipcc makes of min(a,b) on u64
    uint64_t a = seed + threadIdx.x, b = seed * 77ull + threadIdx.x * 3, acc = 0;
#pragma unroll 8
    for (int i = 0; i < REPB * 8; i++) {
        uint64_t m = a < b ? a : b;
        acc += m; a += acc * 3; b ^= a >> 7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_select_base(uint64_t *out, uint32_t seed) {  // same loop without the select
    uint64_t a = seed + threadIdx.x, b = seed * 77ull + threadIdx.x * 3, acc = 0;
#pragma unroll 8
    for (int i = 0; i < REPB * 8; i++) {
        uint64_t m = a ^ b;
        acc += m; a += acc * 3; b ^= a >> 7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_lds_lut(uint64_t *out, uint32_t seed) {      // 3 ds_read_b64 + 2 ds_read_b32 at hashed indices per iteration
    __shared__ uint64_t lut[1536];
    for (uint32_t i = threadIdx.x; i < 1536; i += blockDim.x) lut[i] = i * 0x9e3779b97f4a7c15ull;
    __syncthreads();
    uint32_t x = seed + threadIdx.x * 2654435761u; uint64_t acc = 0;
#pragma unroll 4
    for (int i = 0; i < REPB * 4; i++) {
        uint64_t A = lut[x & 0xff], B = lut[256 + ((x >> 16) & 0xff)], T = lut[512 + ((x >> 9) & 0x3ff)];
        uint32_t a1 = (uint32_t)lut[(x >> 8) & 0xff], b1 = (uint32_t)lut[256 + (x >> 24)];
        acc += A ^ B ^ T ^ a1 ^ ((uint64_t)b1 << 32);
        x = x * 1664525u + 1013904223u + (uint32_t)acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// same, with the number of resident waves per SIMD limited by a dynamic LDS allocation
template <typename K>
static void run_block_occ(const char *name, K kern, uint64_t *d_out, double iters, int waves_per_simd) {
    const int threads = 256, blocks = 256 * waves_per_simd * 4;  // 4 rounds of fully occupied CUs
    const size_t lds = (160 * 1024 / waves_per_simd) & ~(size_t)255;  // one 4-wave block per SIMD-wave slot
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_iters = (double)blocks * (threads / 64) * iters;
    printf("%-14s waves/SIMD %d  %8.3f ms  %7.1f SIMD-cycles per wave-iteration (at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_iters);
}

template <typename K>
static void run_block(const char *name, K kern, uint64_t *d_out, double iters) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_iters = (double)blocks * (threads / 64) * iters;
    printf("%-14s %8.3f ms  %7.1f SIMD-cycles per wave-iteration (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_iters);
}

int main() {
    uint64_t *d_out;
    hipMalloc(&d_out, (size_t)256 * 8 * 4 * 256 * 8);
    run("v_add_u32", k_add32, d_out, 1);
    run("v_mul_lo_u32", k_mul_lo, d_out, 1);
    run("v_mul_hi_u32", k_mul_hi, d_out, 1);
    run("v_mul_u32_u24", k_mul_u24, d_out, 1);
    run("v_mad_u64_u32", k_mad64, d_out, 1);
    run("mad64 sdst=s[40:41]", k_mad64_sgpr, d_out, 1);
    run("mad64 sdst rotating", k_mad64_rot, d_out, 1);
    run("mad64 v*s+0 vcc", k_mad64_s0, d_out, 1);
    run("mad64 v*s+0 sgpr", k_mad64_s0_sgpr, d_out, 1);
    run("v_mul_lo_u32 v*s", k_mul_lo_s, d_out, 1);
    run("v_mad_u32_u24", k_mad_u32_u24, d_out, 1);
    run("alignbit,xor alternating", k_mix_alt, d_out, 2);
    run("alignbit x8 then xor x8", k_mix_grp, d_out, 2);
    run("alignbit->xor dependent, 8 chains", k_mix_dep, d_out, 2);
    run("alignbit->xor one chain", k_mix_dep1, d_out, 2);
    run("E C C independent regs", k_mix_ecc, d_out, 3);
    run("E lshr xor dependent, 8 chains", k_mix_ecc_dep, d_out, 3);
    run("E lshr xor dependent, one chain", k_mix_ecc_dep1, d_out, 3);
    run("E E C C two chains (a,b)", k_mix_eecc1, d_out, 4);
    run("E C E C two chains (a,b)", k_mix_ecec1, d_out, 4);
    run("v_lshl_add_u64", k_lshl_add64, d_out, 1);
    run("v_lshlrev_b64", k_lshl64, d_out, 1);
    run("v_lshrrev_b64", k_lshr64, d_out, 1);
    run("v_alignbit_b32", k_alignbit, d_out, 1);
    run("add_co+addc", k_addc_pair, d_out, 2);
    run("cmp_lt_u64+addc", k_cmp64, d_out, 2);
    run("v_perm_b32", k_perm, d_out, 1);
    run("v_add_u32_e64", k_add32_e64, d_out, 1);
    run("v_xor_b32_e32", k_xor_e32, d_out, 1);
    run("v_xor_b32_e64", k_xor_e64, d_out, 1);
    run("v_xor_b32 literal", k_xor_lit, d_out, 1);
    run("v_lshrrev_b32_e32", k_lshr_e32, d_out, 1);
    run("v_and_b32_e32", k_and_e32, d_out, 1);
    run("v_cndmask_b32_e32", k_cndmask, d_out, 1);
    run("v_bfe_u32", k_bfe, d_out, 1);
    run("v_lshl_or_b32", k_lshl_or, d_out, 1);
    run("v_and_or_b32", k_and_or, d_out, 1);
    run("v_add3_u32", k_add3, d_out, 1);
    run("v_bitop3_b32", k_bitop3, d_out, 1);
    run("v_lshlrev_b32_sdwa", k_sdwa, d_out, 1);
    run("v_or_b32_sdwa", k_or_sdwa, d_out, 1);
    run("cmp_lt_u32+addc", k_cmp32, d_out, 2);
    run("v_sub_u32_e32", k_sub_e32, d_out, 1);
    run("v_min_u32_e32", k_min, d_out, 1);
    run("v_mov_b32_e32", k_mov, d_out, 1);
    run_block("hash_block", k_hash_block, d_out, REPB);
    run_block("filter_block", k_filter_block, d_out, REPB);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run_block_occ("hash_block", k_hash_block, d_out, REPB, w);
    run_block("cmp64+cndmask", k_cmp_cnd, d_out, REPB * 8);
    run_block("select_c", k_select_c, d_out, REPB * 8);
    run_block("select_base", k_select_base, d_out, REPB * 8);
    run_block("lds_lut5", k_lds_lut, d_out, REPB * 4);
    return 0;
}
