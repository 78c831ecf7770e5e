// Developer tool (round 6): can the posterior pass run as ONE dependency-driven launch?  The protocol under test is the
// data-tagged granule of MI355X_MICROARCH.md: a column's (B, R) pairs are 16-byte granules whose R half carries "pending" as a
// NaN payload until the column's owner overwrites it with one sc1 (write-through) store; a reader gathers with sc1 loads and
// repeats the gather while any R it needs is still pending.  No flag, no fence, no ordering between stores.
//
// The workload mimics the schedule: `levels` levels of `width` columns, a column = a block of 12 pairs (192 bytes on a
// 64-byte boundary: two neighbouring blocks share a 128-byte line, as in the plan), a column of level l gathers the first
// 6 pairs of 8 random columns of level l - 1 and writes pair j = sum of the gathered R's + j + 1 (exact in doubles).
//   mode 0: one launch per level, plain loads and stores (what the HIP graph of the pass does today)
//   mode 1: ONE launch, a wavefront per column in level order (wave w takes columns w, w + W, ...), sc1 + pending tags
//   mode 2: mode 1, but every wavefront first reads the lines of ALL its future sources with plain loads (pending values
//           into its XCD's L2 and its CU's L1 on purpose: the stale-line hazard) before it starts
// Every value is checked against the host's; polls and give-ups are counted (a bounded spin: no hang, the run reports it).
//   hipcc --offload-arch=gfx950 -O3 -o flow_handoff flow_handoff.hip && ./flow_handoff 256 60 && ./flow_handoff 30000 12
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kPairs = 12, kUse = 6, kSrc = 8;
constexpr unsigned long long kPending = 0x7FF8DEAD00000001ull;

__device__ __forceinline__ v2d load_sc1(const v2d *p)
{
    v4i r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return __builtin_bit_cast(v2d, r);
}
__device__ __forceinline__ void store_sc1(v2d *p, v2d v)
{
    const v4i r = __builtin_bit_cast(v4i, v);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ bool pending(double x) { return __builtin_bit_cast(unsigned long long, x) == kPending; }

__global__ void init_kernel(v2d *C, long npairs, int width)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const long col = i / kPairs;
    // level 0 is final from the start (R = column + pair index), the rest pending
    v2d v;
    v.x = (double)(i % 97);
    v.y = (col < width) ? (double)(col % 1000 + (i % kPairs) + 1) : __builtin_bit_cast(double, kPending);
    C[i + (col / 1) * 0] = v;
}

// one column: lanes 0..47 = 8 sources x 6 pairs
__device__ int g_sleep = 2;
template <bool FLOW>
__device__ __forceinline__ void do_column(v2d *C, const int *src, long col, int lane, unsigned long long *stats)
{
    const int s = lane / kUse, e = lane % kUse;
    const bool act = lane < kSrc * kUse;
    const long sc = src[col * kSrc + (act ? s : 0)];
    const v2d *p = C + sc * kPairs + (act ? e : 0);
    v2d v;
    if (FLOW) {
        unsigned spins = 0;
        for (;;) {
            v = load_sc1(p);
            const bool pend = act && pending(v.y);
            if (__builtin_amdgcn_ballot_w64(pend) == 0) break;
            if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(&stats[1], 1ull); break; }   // give up: reported, never a hang
            for (int q = 0; q < g_sleep; ++q) __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0 && spins) atomicAdd(&stats[0], (unsigned long long)spins);
    } else {
        v = *p;
    }
    double sum = act ? v.y : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < kPairs) {
        v2d out;
        out.x = (double)((col * kPairs + lane) % 97);
        out.y = (sum - 1048576.0 * floor(sum * (1.0 / 1048576.0))) + (double)(lane + 1);      // (stays exact in doubles)
        if (FLOW) store_sc1(C + col * kPairs + lane, out);
        else C[col * kPairs + lane] = out;
    }
}

__global__ void __launch_bounds__(256) level_kernel(v2d *C, const int *src, long first, int count, unsigned long long *stats)
{
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= count) return;
    do_column<false>(C, src, first + w, threadIdx.x & 63, stats);
}

__global__ void __launch_bounds__(256) flow_kernel(v2d *C, const int *src, long first, long ncols, int prewarm, double *sink,
                                                   unsigned long long *stats)
{
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prewarm) {                                     // pull the (pending) lines of every future source through L1 and L2
        double acc = 0.0;
        for (long c = first + w0; c < ncols; c += W) {
            const int s = lane / kUse, e = lane % kUse;
            if (lane < kSrc * kUse) acc += C[(long)src[c * kSrc + s] * kPairs + e].x;
        }
        if (acc == 1.2345) sink[0] = acc;
        __syncthreads();
    }
    for (long c = first + w0; c < ncols; c += W) do_column<true>(C, src, c, lane, stats);
}

int main(int argc, char **argv)
{
    const int width = argc > 1 ? atoi(argv[1]) : 256, levels = argc > 2 ? atoi(argv[2]) : 60, rep = 5;
    const int sleep_units = argc > 3 ? atoi(argv[3]) : 2;              // x 8 x 64 clocks between two polls
    hipMemcpyToSymbol(HIP_SYMBOL(g_sleep), &sleep_units, sizeof(int));
    const long ncols = (long)width * levels, npairs = ncols * kPairs;
    std::vector<int> src((size_t)ncols * kSrc, 0);
    std::vector<double> expect((size_t)npairs, 0.0);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (long c = 0; c < width; ++c)
        for (int j = 0; j < kPairs; ++j) expect[(size_t)(c * kPairs + j)] = (double)(c % 1000 + j + 1);
    for (long c = width; c < ncols; ++c) {
        const long base = (c / width - 1) * width;
        double sum = 0.0;
        for (int s = 0; s < kSrc; ++s) {
            const long sc = base + (long)(rnd() % (uint64_t)width);
            src[(size_t)(c * kSrc + s)] = (int)sc;
            for (int e = 0; e < kUse; ++e) sum += expect[(size_t)(sc * kPairs + e)];
        }
        sum = sum - 1048576.0 * std::floor(sum * (1.0 / 1048576.0));
        for (int j = 0; j < kPairs; ++j) expect[(size_t)(c * kPairs + j)] = sum + (double)(j + 1);
    }
    v2d *C; int *d_src; unsigned long long *d_stats; double *d_sink;
    hipMalloc(&C, sizeof(v2d) * npairs);
    hipMalloc(&d_src, sizeof(int) * src.size());
    hipMalloc(&d_stats, 16);
    hipMalloc(&d_sink, 8);
    hipMemcpy(d_src, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid_flow = prop.multiProcessorCount * 4;              // 4 blocks of 4 waves per CU: all resident
    std::vector<v2d> host((size_t)npairs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("width %d, levels %d (%ld columns), flow grid %d x 256, %d x 512 clocks between polls\n", width, levels, ncols, grid_flow, sleep_units);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        unsigned long long stats[2] = {0, 0};
        long bad = 0;
        for (int r = 0; r < rep; ++r) {
            hipLaunchKernelGGL(init_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, 0, C, npairs, width);
            hipMemset(d_stats, 0, 16);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            if (mode == 0) {
                for (int l = 1; l < levels; ++l)
                    hipLaunchKernelGGL(level_kernel, dim3((width + 3) / 4), dim3(256), 0, 0, C, d_src, (long)l * width, width, d_stats);
            } else {
                hipLaunchKernelGGL(flow_kernel, dim3(grid_flow), dim3(256), 0, 0, C, d_src, (long)width, ncols, mode == 2 ? 1 : 0, d_sink, d_stats);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            hipMemcpy(host.data(), C, sizeof(v2d) * npairs, hipMemcpyDeviceToHost);
            for (long i = 0; i < npairs; ++i)
                if (host[(size_t)i].y != expect[(size_t)i]) ++bad;
            unsigned long long s2[2];
            hipMemcpy(s2, d_stats, 16, hipMemcpyDeviceToHost);
            stats[0] += s2[0]; stats[1] += s2[1];
        }
        printf("  mode %d (%s): best %.1f us = %.2f us per level; wrong values over %d runs: %ld; re-polls %llu, give-ups %llu\n", mode,
               mode == 0 ? "one launch per level, plain" : (mode == 1 ? "one launch, sc1 + pending tags" : "one launch, sc1 + tags, L1/L2 pre-warmed with pending lines"),
               best * 1e3f, best * 1e3f / (levels - 1), rep, bad, stats[0], stats[1]);
    }
    return 0;
}
