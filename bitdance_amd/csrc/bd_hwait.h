// The consumer side of the sequence-parallel hand-off, shared by the GEMM prologue (bd_gemm_kernel.h) and the construction-time
// self-test of the hand-off (bd_sp.hip sp_test_check_kernel): the SAME device function polls, invalidates and releases both.
#pragma once
#include "bd_common.h"
#include "bd_kernels.h"

// The consumer side of the sequence-parallel hand-off (bd_sp.hip): the weights do not depend on the peers, so the first R stages are
// requested BEFORE this; the operand rows were written into cacheable local memory by sc0 sc1 write-through stores from other GPUs
// (or, in the one-GPU tests, other XCDs), so after the flags this CU's L1 and this XCD's L2 may still hold lines of the PREVIOUS
// operand: one `buffer_inv sc0 sc1` (system-scope invalidate of non-coherent lines) ahead of the barrier, then plain loads.
BD_DEV void gemm_hwait(const BdHWait& w, int tid, int nthreads) {
    const int e = bd_sp_epoch_of(__hip_atomic_load(w.rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), w.seq);
    const long long t0 = wall_clock64();
    for (int i = tid; i < w.n; i += nthreads) {
        while (bd_epoch_before(__hip_atomic_load(w.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e)) {
            if (__hip_atomic_load(w.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;      // a dead exchange: run on (garbage in, the host raises)
            if (wall_clock64() - t0 > w.timeout_ticks) { __hip_atomic_fetch_or(w.err, 1 << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (w.inv == 0 && tid < 64) asm volatile("buffer_inv sc0 sc1" ::: "memory");
    __syncthreads();
    if (w.inv == 1) asm volatile("buffer_inv sc0 sc1" ::: "memory");
}

