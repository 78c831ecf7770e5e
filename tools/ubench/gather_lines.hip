// Developer tool: what a scattered 16-byte gather costs on MI355X, and what FETCH_SIZE says about it.
// The wide levels of the posterior pass (gpv_posterior.hip) are nothing but such gathers: every lane reads one (B, R)
// pair = 16 bytes of another column's compact block.  Questions answered here, each as a kernel of its own so that
// rocprofv3 --pmc FETCH_SIZE can be read per kernel beside the times printed below:
//   g16_line128   one 16-byte load per lane, every lane a different random 128-byte line
//   g16_pair      lanes 2i and 2i+1 read the two 64-byte halves of ONE random 128-byte line (16 bytes each)
//   g16_sector64  4 consecutive lanes read one random 64-byte sector (4 x 16 bytes)
//   g16_line128f  8 consecutive lanes read one random 128-byte line completely
//   g16_run96     6 consecutive lanes read a 96-byte run starting at a random 64-byte boundary (a column-block prefix)
// on a buffer of `mb` megabytes (argument 1; 128 = resident in the 256 MB Infinity Cache after the first pass, 2048 = not),
// `rep` passes (argument 2) of n = 64 M lane-loads... each kernel loads the SAME number of 16-byte elements.
//   hipcc --offload-arch=gfx950 -O3 -o gather_lines gather_lines.hip && ./gather_lines 2048 && ./gather_lines 128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long mix(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// GROUP consecutive lanes share one random base (aligned to ALIGN bytes) and read 16 bytes each at base + 16 * (lane % GROUP)
// (+ HALF: lane pairs take the two 64-byte halves of a 128-byte line)
template <int GROUP, int ALIGN, bool HALF>
__global__ void __launch_bounds__(256) gather(const char *buf, unsigned long long nlines, int loads, double *out, unsigned seed)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long grp = gid / GROUP;
    const int l = (int)(gid % GROUP);
    double acc = 0.0;
#pragma unroll 4
    for (int it = 0; it < loads; ++it) {
        const unsigned long long r = mix(grp * 0x9E3779B97F4A7C15ull + (unsigned long long)it * 0xD6E8FEB86659FD93ull + seed) % nlines;
        const char *p = buf + r * ALIGN + (HALF ? (l * 64) : (l * 16));
        const v2d v = *reinterpret_cast<const v2d *>(p);
        acc += v.x + v.y;
    }
    if (acc == 123.456) out[0] = acc;
}

template <int GROUP, int ALIGN, bool HALF>
static void run(const char *name, const char *buf, size_t bytes, int loads, double *out, int rep)
{
    const unsigned long long nlines = bytes / ALIGN - 2;
    const int threads = 256, grid = 256 * 32 * 4;                     // 8.4 M lanes in flight over the launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<GROUP, ALIGN, HALF>), dim3(grid), dim3(threads), 0, 0, buf, nlines, loads, out, 1u);   // warm (and fill the cache)
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < rep; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gather<GROUP, ALIGN, HALF>), dim3(grid), dim3(threads), 0, 0, buf, nlines, loads, out, 2u + r);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double nl = (double)grid * threads * loads;
    printf("%-14s %8.3f ms  %7.1f G lane-loads/s  %7.1f GB/s useful (16 B per lane)  distinct %d-byte units per ns: %.2f\n", name, best,
           nl / best * 1e-6, nl * 16 / best * 1e-6, ALIGN, nl / GROUP * (HALF ? 1 : 1) / best * 1e-6);
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 2048;
    const int rep = argc > 2 ? atoi(argv[2]) : 3;
    const size_t bytes = mb << 20;
    char *buf; double *out;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    printf("buffer %zu MB\n", mb);
    const int loads = 8;
    run<1, 128, false>("g16_line128", buf, bytes, loads, out, rep);
    run<2, 128, true>("g16_pair", buf, bytes, loads, out, rep);
    run<4, 64, false>("g16_sector64", buf, bytes, loads, out, rep);
    run<8, 128, false>("g16_line128f", buf, bytes, loads, out, rep);
    run<6, 64, false>("g16_run96", buf, bytes, loads, out, rep);
    return 0;
}
