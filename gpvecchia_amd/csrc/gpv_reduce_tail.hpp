// gpv_reduce_tail.hpp — the final reduction of the per-workgroup partial sums, done by the conditioning-set kernel itself.
//
// Every workgroup stores its kNSums partials to block_sums[blockIdx.x][.], publishes them (agent-scope release) and takes a
// ticket; the workgroup whose ticket is the last one re-reads ALL partials and sums them in a fixed order, so the result
// does not depend on which workgroup happens to finish last: bitwise reproducible for a given grid, no float atomics, and
// no second launch (the separate one-workgroup reduction kernel cost 6-8 us of a 180 us evaluation at the per-rank load of
// an 8-GPU run).  Hand-off form: plain stores -> s_waitcnt vmcnt(0) -> agent release -> s_waitcnt -> relaxed agent ticket;
// the last arriver: agent acquire -> s_waitcnt -> workgroup barrier -> loads (MI355X_MICROARCH.md, "Valid forms").
#pragma once
#include "gpv_internal.h"

namespace gpv {

// Hand-off of a value this thread has just stored to HOST memory: its sequence number follows it with a system-scope release,
// so a host thread that reads the number (and then an acquire fence) reads the value.  The host spins on the number instead
// of waiting for the stream: no completion signal, no interrupt, no runtime call between the last store and the reader.
__device__ __forceinline__ void publish_seq(unsigned long long *cell, unsigned long long seq)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(cell, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Call from every thread of the workgroup; threads 0..kNSums-1 (wave 0) hold this workgroup's partial sums in `mine`.
// NT = threads per workgroup (a multiple of 64, at most 512).  s_part (NT doubles) and s_last_p (one int) are LDS the
// workgroup no longer uses (the caller has passed a workgroup barrier since their last use): the kernels run at the edge of
// the 160 KiB of a CU and a declaration of their own would cost a resident workgroup.
// AP: `const SetArgs *` or a pointer to the arguments in the kernarg segment (kargs_now() of gpv_sets_kernel.hpp): the fields are
// read where they are used, after the task loop, instead of sitting in SGPRs through it.
template <int NT, class AP>
__device__ __forceinline__ void reduce_tail(AP Ap, double mine, double *s_part, int *s_last_p)
{
    const auto &A = *Ap;
    static_assert(NT % 64 == 0 && NT <= 512, "workgroup size");
    if (threadIdx.x < kNSums) A.block_sums[(int64_t)blockIdx.x * kNSums + threadIdx.x] = mine;
    if (threadIdx.x < 64) {                              // the wave that stored
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (hipcc may drop the wait behind buffer_wbl2: keep it in asm)
            const unsigned old = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == gridDim.x - 1u) ? 1 : 0;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *s_last_p = last;
        }
    }
    __syncthreads();
    if (!*s_last_p) return;
    // thread = q + 8 * part: NT/8 strided partial sums per quantity, four independent accumulators each (loads of a
    // burst in flight together), then a fixed-order combine
    constexpr int NP = NT / kNSums;
    const int q = threadIdx.x & (kNSums - 1), part = threadIdx.x / kNSums;
    const int nblocks = (int)gridDim.x;
    auto ld = [&](int b) {
        return __hip_atomic_load(A.block_sums + (int64_t)b * kNSums + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = part;
    for (; b + 3 * NP < nblocks; b += 4 * NP) {
        s0 += ld(b);
        s1 += ld(b + NP);
        s2 += ld(b + 2 * NP);
        s3 += ld(b + 3 * NP);
    }
    for (; b < nblocks; b += NP) s0 += ld(b);
    s_part[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double t = 0.0;
#pragma unroll 4
        for (int p2 = 0; p2 < NP; ++p2) t += s_part[p2 * kNSums + threadIdx.x];
        A.sums[threadIdx.x] = t;
        if (A.sums_copy != nullptr) A.sums_copy[threadIdx.x] = t;
        if (A.seq_cells != nullptr) publish_seq(A.seq_cells + threadIdx.x, A.seq);
        if (threadIdx.x == 0) {
            *A.ticket = 0u;                              // for the next launch (ordered behind this one by its stream)
            if (A.nug_cell != nullptr) *A.nug_cell = A.nug_scalar;
        }
    }
}

}  // namespace gpv
