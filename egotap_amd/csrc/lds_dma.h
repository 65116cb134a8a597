// [r5] The address of an LDS DMA (global_load_lds_*) as a wave-uniform 64-bit base in scalar registers + a 32-bit byte offset per lane (the
// instruction's s[base] form).  Measured against a 64-bit pointer per lane (the `off` form) on every kernel that was moved over: fp32 headline GEMM
// +1.6 % frames/s, weight-gradient GEMM's gathers -8 ... -13 %, decoder convolutions -4 %, attention -1 %.
//
// Every DMA statement of the library is  s_mov_b32 m0, <LDS address> ; s_nop 0 ; global_load_lds_* : M0 (the DMA's LDS base) is written in the statement
// that reads it (hipcc keeps nothing in M0 across statements), and the SALU write of M0 -> LDS-DMA read takes one wait state, which nothing pads inside
// an asm string.  (Rounds 1-4 ran without the s_nop and bit-exact against the oracle; the pad costs one cycle per KiB moved and removes the question.)
#pragma once
#include <hip/hip_runtime.h>

// A value the whole wave agrees on, moved to scalar registers.  v_readfirstlane is a VALU write of an SGPR; a VMEM instruction that reads that SGPR as
// its base needs 5 wait states after it, and hipcc pads nothing for the operands of an asm statement (cdna_hip_programming.md 5.7, item 2): the s_nop
// sits between the two, bound to both through its operands.  (__builtin_amdgcn_readfirstlane returns int: the halves go through `unsigned` --
// a sign-extended low half ORed into the high one is an address 4 GB below the canonical hole.)
__device__ __forceinline__ unsigned long long lds_dma_base(unsigned long long v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long lds_dma_base(const void* p) { return lds_dma_base((unsigned long long)(size_t)p); }
