// Synchronisation primitives of the wave-specialised SP kernels (conv_mfma_sp.hip, upfuse_sp.hip): LDS counters between
// mover and consumer waves of one workgroup, counted vector-memory waits, raw barriers.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void sp_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void sp_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform)
__device__ __forceinline__ void sp_wait_vm(int n) {
#define DRS_SP_CASE(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
  switch (n) {
    DRS_SP_CASE(1) DRS_SP_CASE(2) DRS_SP_CASE(3) DRS_SP_CASE(4) DRS_SP_CASE(5) DRS_SP_CASE(6) DRS_SP_CASE(7)
    DRS_SP_CASE(8) DRS_SP_CASE(9) DRS_SP_CASE(10) DRS_SP_CASE(11) DRS_SP_CASE(12) DRS_SP_CASE(13) DRS_SP_CASE(14)
    DRS_SP_CASE(15) DRS_SP_CASE(16) DRS_SP_CASE(17) DRS_SP_CASE(18) DRS_SP_CASE(19) DRS_SP_CASE(20) DRS_SP_CASE(21)
    DRS_SP_CASE(22) DRS_SP_CASE(23) DRS_SP_CASE(24) DRS_SP_CASE(25) DRS_SP_CASE(26) DRS_SP_CASE(27) DRS_SP_CASE(28)
    DRS_SP_CASE(29) DRS_SP_CASE(30) DRS_SP_CASE(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef DRS_SP_CASE
}

typedef __attribute__((address_space(3))) unsigned* sp_flag_ptr;

// Spin until the LDS counter *f reaches `target`.  A protocol error must never leave waves spinning on the GPU and must
// not fault the process either: after 2^24 polls (seconds of wall time; a legitimate wait is one K-step, microseconds,
// whatever a profiler or the clock governor does to it) the wave records the timeout in the caller's fault word
// (TapConv::fault, read back by drs_unet_check_faults) and ENDS.  Every other wave of the block then runs into the same
// bound on the counters this wave no longer advances, so the launch drains by itself with its output incomplete.
__device__ __forceinline__ void sp_poll(sp_flag_ptr f, unsigned target, unsigned* fault) {
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 24)) {
      if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_endpgm();
    }
  }
}
// The same wait for data that lives in LDS only (window / weight-ring images written by ds_write, read by ds_read): the
// LDS executes one wave's operations in order and is coherent inside the workgroup, so a relaxed poll followed by a
// compiler barrier orders the fragment reads behind it.  The acquire form above makes hipcc drain the wave's VECTOR-memory
// counter as well (s_waitcnt vmcnt(0)): a consumer would wait for the previous item's epilogue stores to reach memory
// before it may touch the next item's first window.
__device__ __forceinline__ void sp_poll_lds(sp_flag_ptr f, unsigned target, unsigned* fault) {
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 24)) {
      if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_endpgm();
    }
  }
  asm volatile("" ::: "memory");
}
// Release without draining the LDS queue: the LDS executes one wave's operations in order, so an add issued after the
// fragment reads (or the staging writes) of a buffer is PERFORMED after them, whatever is still in flight towards the
// registers.  One lane adds; the compiler barrier keeps the memory operations of the source on their side.
__device__ __forceinline__ void sp_release(sp_flag_ptr f, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void sp_bump(sp_flag_ptr f) {
  __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
