// gpv_posterior_ext.h — single-workgroup kernels for the NARROW levels of the posterior pass (gpv_posterior.hip).
// A level with a handful of columns costs 5-7 us as a launch of its own (dispatch, kernel-argument load, cold dependent
// loads, end-of-kernel cache actions); consecutive narrow levels therefore run inside ONE workgroup, separated by
// workgroup barriers instead of kernel boundaries.
#pragma once
#include "gpv_internal.h"

namespace gpv {
// posterior mean sweep R^T u = t (R/vecchia_prediction.R:124-125): levels [0, nlev) of the ascending schedule, each at
// most kMeanHeadMax columns wide; levptr2: device copy of the level offsets into order2
constexpr int kMeanHeadMax = 32;
hipError_t launch_mean_head(const PostArgs &a, const int32_t *order2, double *u, const int32_t *levptr2, int nlev, hipStream_t s);
}  // namespace gpv
