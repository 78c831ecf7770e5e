// gpv_posterior_ext.h — the kernels that take the NARROW ends of the posterior pass's two schedules out of the per-level
// launches (gpv_posterior.hip).  A level with a handful of columns costs 5-8 us as a launch of its own (dispatch, cold
// dependent loads, end-of-kernel cache actions, and the trip through memory between consecutive levels).
#pragma once
#include "gpv_internal.h"

namespace gpv {
// posterior mean sweep R^T u = t (R/vecchia_prediction.R:124-125): levels [0, nlev) of the ascending schedule, each at
// most kMeanHeadMax columns wide; levptr2: device copy of the level offsets into order2
constexpr int kMeanHeadMax = 32;
hipError_t launch_mean_head(const PostArgs &a, const int32_t *order2, double *u, const int32_t *levptr2, int nlev, hipStream_t s);
// Factor pass: the dense top block, columns 0 .. K-1 (K <= kTopMax) of the ordering, which the plan keeps out of the level
// schedule (gpv_posterior.hip, gpv_posterior_top_kernel).  Their column records sit at positions [first, first + K) of
// colrec, ascending, with colrec[..][1].y = the end of the row-list entries that are top columns themselves; tpart: [K][66].
constexpr int kTopMax = 64;
hipError_t launch_posterior_top(const PostArgs &a, int first, int K, double *tpart, hipStream_t s);
}  // namespace gpv
