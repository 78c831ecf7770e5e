// gpv_generic.h — the workgroup-per-set kernel for row lengths in (64, 192] and dimensions above 8 (gpv_sets_generic.hip)
#pragma once
#include "gpv_internal.h"

namespace gpv {
int generic_max_P();
hipError_t launch_sets_generic(int P, const SetArgs &a, int cus, int *grid_out, hipStream_t stream);
}  // namespace gpv
