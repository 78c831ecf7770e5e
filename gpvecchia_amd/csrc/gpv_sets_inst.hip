// One translation unit per compiled row length: hipcc ... -DGPV_INST_P=31 -c gpv_sets_inst.hip -o sets_p31.o
// Long rows (the fully unrolled sweep of P >= 41 takes minutes per instantiation) are split further, one TU per
// spatial dimension:  -DGPV_INST_P=61 -DGPV_INST_DIM=1|2|3|0  (0 = run-time dimension > 3 and the dense-covariance
// variant of U_NZentries_mat) plus  -DGPV_INST_P=61 -DGPV_INST_DISPATCH  for the function that picks among them.
#include "gpv_sets_kernel.hpp"
#ifndef GPV_INST_P
#error "compile with -DGPV_INST_P=<row length>"
#endif
#define GPV_CAT2(a, b) a##b
#define GPV_CAT(a, b) GPV_CAT2(a, b)
#define GPV_PART(d) GPV_CAT(GPV_CAT(GPV_CAT(launch_sets_p, GPV_INST_P), _d), d)
namespace gpv {
#if defined(GPV_INST_DISPATCH)
hipError_t GPV_PART(0)(const SetArgs &, int, int *, hipStream_t);
hipError_t GPV_PART(1)(const SetArgs &, int, int *, hipStream_t);
hipError_t GPV_PART(2)(const SetArgs &, int, int *, hipStream_t);
hipError_t GPV_PART(3)(const SetArgs &, int, int *, hipStream_t);
hipError_t GPV_CAT(launch_sets_p, GPV_INST_P)(const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    if (a.cov == COV_DENSE) return GPV_PART(0)(a, cus, grid_out, stream);
    switch (a.dim) {
        case 1: return GPV_PART(1)(a, cus, grid_out, stream);
        case 2: return GPV_PART(2)(a, cus, grid_out, stream);
        case 3: return GPV_PART(3)(a, cus, grid_out, stream);
        default: return GPV_PART(0)(a, cus, grid_out, stream);
    }
}
#elif defined(GPV_INST_DIM)
hipError_t GPV_PART(GPV_INST_DIM)(const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
#if GPV_INST_DIM == 0
    if (a.cov == COV_DENSE) return launch_sets_PDC<GPV_INST_P, 1, COV_DENSE>(a, cus, grid_out, stream);   // U_NZentries_mat: no coordinates
#endif
    return launch_sets_PD<GPV_INST_P, GPV_INST_DIM>(a, cus, grid_out, stream);
}
#else
hipError_t GPV_CAT(launch_sets_p, GPV_INST_P)(const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    return launch_sets_P<GPV_INST_P>(a, cus, grid_out, stream);
}
#endif
}  // namespace gpv
