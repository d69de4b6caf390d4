// ldsdma.hip.h -- LDS-DMA (global_load_lds_dwordx4) as inline asm, shared by the kernels that prefetch their next tile straight
// into LDS (fft_fir.hip, chan.hip).  Inline asm on purpose: the request must stay out of hipcc's own vmcnt bookkeeping, which would
// otherwise drain it -- together with every store issued after it -- at the next barrier or load use; the kernels count it
// themselves (`s_waitcnt vmcnt(N)` with N = the vector-memory instructions they have issued since).  M0 (the LDS base of the
// request) is compiler-reserved: it is saved and restored inside the statement (cdna_hip_programming.md 5.7).
// Each active lane moves 16 bytes from sbase + voff (+ nothing else) to LDS byte lds_byte + 16 * lane.
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

__device__ __forceinline__ void dma16_to_lds(const void* sbase, unsigned voff, unsigned lds_byte) {   // sbase, lds_byte wave-uniform
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma16_to_lds_nt(const void* sbase, unsigned voff, unsigned lds_byte) {   // the same, non-temporal
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

}  // namespace qk
