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
// Factor pass: the dense top block (gpv_posterior.hip, gpv_posterior_top_kernel / gpv_posterior_top2_kernel): a set T of K <=
// kTopMax columns the plan keeps out of the level schedule: the first kTopBlock columns of the ordering and, in the two-block
// form, the columns of the schedule's highest levels (T is closed: the rows of its columns are in T, and no column outside T
// waits for one inside).  Their column records sit at positions [first, first + K) of colrec, ascending; rowrec[q].w flags
// the row-list pairs whose column is in T; tpart: [K][66]; topinfo[j] = {column, its block's offset in C}; toprows[j][e] = the
// index INSIDE the block of entry e's row (0xFF: no such entry).
constexpr int kTopBlock = 64, kTopMax = 2 * kTopBlock;
// rr0_off: the block's first-round records (stride kTopRr0Stride per column) in PostArgs::rr0
constexpr int kTopRr0Stride = 256;
hipError_t launch_posterior_top(const PostArgs &a, int first, int K, double *tpart, const int2 *topinfo, const uint8_t *toprows,
                                int64_t rr0_off, hipStream_t s);
// Mean sweep, the columns of T: the plan's ascending schedule (order2 / levptr2 / meanrec) holds the OTHER columns only
hipError_t launch_mean_top(const PostArgs &a, double *u, int K, const int2 *topinfo, const uint8_t *toprows, hipStream_t s);
}  // namespace gpv
