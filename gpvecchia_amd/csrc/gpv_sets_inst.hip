// One translation unit per compiled row length: hipcc ... -DGPV_INST_P=31 -c gpv_sets_inst.hip -o sets_p31.o
#include "gpv_sets_kernel.hpp"
#ifndef GPV_INST_P
#error "compile with -DGPV_INST_P=<row length>"
#endif
#define GPV_CAT2(a, b) a##b
#define GPV_CAT(a, b) GPV_CAT2(a, b)
namespace gpv {
hipError_t GPV_CAT(launch_sets_p, GPV_INST_P)(const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    return launch_sets_P<GPV_INST_P>(a, cus, grid_out, stream);
}
}  // namespace gpv
