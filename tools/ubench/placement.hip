// Developer tool: where does the dispatcher put the workgroups of a launch shaped like the conditioning-set kernel's?
// Every wave records (XCC_ID, HW_ID) and its start / end s_memrealtime; the host prints, per workgroup, XCD / SE / CU / SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o placement placement.hip && ./placement [grid=512] [lds_kb=80] [spin_us=20] [waves_per_wg=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

__global__ void __launch_bounds__(256) probe(unsigned *out, unsigned long long *tm, int spin)
{
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t1 = wall_clock64();
    const unsigned long long tstart = t1;
    lds[threadIdx.x] = (double)hw;
    while (wall_clock64() - tstart < (unsigned long long)spin * 100ull) { lds[threadIdx.x] += 1.0; }   // 100 MHz
    t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
        tm[2 * w] = tstart;
        tm[2 * w + 1] = t1;
    }
    if (lds[threadIdx.x] < 0) out[0] = (unsigned)t0;
}

int main(int argc, char **argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 512, ldskb = argc > 2 ? atoi(argv[2]) : 80, spin = argc > 3 ? atoi(argv[3]) : 20;
    const int wpw = argc > 4 ? atoi(argv[4]) : 4;
    unsigned *d; unsigned long long *t;
    hipMalloc(&d, sizeof(unsigned) * grid * 8);
    hipMalloc(&t, sizeof(unsigned long long) * grid * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64 * wpw), ldskb * 1024, 0, d, t, spin);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(grid * 8);
    std::vector<unsigned long long> ht(grid * 8);
    hipMemcpy(h.data(), d, sizeof(unsigned) * grid * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int w = 0; w < grid * wpw; ++w) tmin = std::min(tmin, ht[2 * w]);
    std::map<unsigned, std::vector<int>> percu;       // (xcc, se, sh, cu) -> workgroups
    printf("# wg wave xcc se sh cu simd waveid start_us end_us   (hw_id raw)\n");
    std::map<unsigned, int> persimd;                  // (xcc, se, sh, cu, simd) -> wavefronts
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < wpw; ++w) {
            const unsigned hw = h[2 * (b * wpw + w)], xcc = h[2 * (b * wpw + w) + 1] & 0xf;
            const unsigned waveid = hw & 0xf, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            if ((ht[2 * (b * wpw + w)] - tmin) < 500) persimd[(xcc << 20) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;   // first wave of residents
            if (b < 24 || b % 64 == 0)
                printf("%4d %d  %u %u %u %2u  %u %2u  %8.2f %8.2f  (0x%08x)\n", b, w, xcc, se, sh, cu, simd, waveid,
                       (ht[2 * (b * wpw + w)] - tmin) / 100.0, (ht[2 * (b * wpw + w) + 1] - tmin) / 100.0, hw);
            if (w == 0) percu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
        }
    printf("# %zu distinct CUs used by %d workgroups\n", percu.size(), grid);
    int shown = 0;
    std::map<int, int> hist;
    for (auto &kv : percu) {
        hist[(int)kv.second.size()]++;
        if (shown++ < 40) {
            printf("cu xcc=%u se=%u sh=%u cu=%2u:", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf);
            for (int b : kv.second) printf(" %d", b);
            printf("\n");
        }
    }
    for (auto &kv : hist) printf("# CUs with %d workgroups: %d\n", kv.first, kv.second);
    {
        std::map<int, int> h2;
        for (auto &kv : persimd) h2[kv.second]++;
        for (auto &kv : h2) printf("# SIMDs holding %d wavefronts of the first resident set: %d\n", kv.first, kv.second);
        printf("# SIMDs touched: %zu of %zu\n", persimd.size(), percu.size() * 4);
    }
    // do wave w of the two workgroups of a CU share SIMD w?
    int same = 0, tot = 0;
    for (auto &kv : percu) {
        if (kv.second.size() != 2) continue;
        for (int w = 0; w < wpw; ++w) {
            const unsigned s0 = (h[2 * (kv.second[0] * wpw + w)] >> 4) & 3, s1 = (h[2 * (kv.second[1] * wpw + w)] >> 4) & 3;
            same += (s0 == s1); tot++;
        }
    }
    printf("# wave w of both workgroups of a CU on the same SIMD: %d of %d; wave w on SIMD w: ", same, tot);
    int onw = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < wpw; ++w) onw += (((h[2 * (b * wpw + w)] >> 4) & 3) == (unsigned)w);
    printf("%d of %d\n", onw, grid * wpw);
    return 0;
}
