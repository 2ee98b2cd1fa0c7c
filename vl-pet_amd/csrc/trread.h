// LDS transpose reads (ds_read_b64_tr_b16) and counted waits as inline asm, shared by the streaming kernels.
// hipcc orders every LDS read it can see behind ALL outstanding global_load_lds of the wave (s_waitcnt vmcnt(0) before the
// first ds_read of a step -- which would wait for the stages just requested and serialise the ring), so the reads of a kernel
// that keeps LDS-DMA in flight are inline asm: the counted vmcnt of a stage is issued by hand, and so is the lgkmcnt before the
// MFMAs (a fence / tie carries the operand registers as in-out operands so that no MFMA can be scheduled above the wait).
#pragma once
#include "common.h"

typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
struct TrOp { v2i32 lo, hi; };
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ void tr_read(TrOp& o, uint32_t lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(o.lo) : "v"(lds_addr), "n"(OFF_LO) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(o.hi) : "v"(lds_addr), "n"(OFF_HI) : "memory");
}
__device__ __forceinline__ void tr_fence(TrOp& a) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi) :: "memory");
}
__device__ __forceinline__ void tr_tie(TrOp& a) {            // no instruction: the operand is only usable after the preceding fence
    asm volatile("" : "+v"(a.lo), "+v"(a.hi) :: "memory");
}
__device__ __forceinline__ bf16x8 tr_val(const TrOp& a) {
    const v4i32 r = {a.lo[0], a.lo[1], a.hi[0], a.hi[1]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int N> __device__ __forceinline__ void wgs_wait_n() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter on gfx950");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
// at most `pending` later stages (NI instructions each) may still be in flight
template <int NI, int MAXP>
__device__ __forceinline__ void wgs_wait(int pending) {
    if (pending <= 0) wgs_wait_n<0>();
    else if (pending == 1 || MAXP == 1) wgs_wait_n<NI>();
    else if (pending == 2 || MAXP == 2) wgs_wait_n<2 * NI>();
    else wgs_wait_n<3 * NI>();
}

