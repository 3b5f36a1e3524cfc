// sscan_n.h — general d_state (N > 1) selective-scan kernels (sscan_n.hip), called from sscan.hip's launch plan.
#pragma once
#include "common.h"

namespace vmasr {

// split: 0 = one wave walks its rows' whole segment, 1 = tile-parallel 3-phase (aggregates -> carry -> apply);
// tiles_per_task / nseg: segment geometry of the plan (sscan.hip: make_plan); rows: requested rows per wave of the
// backward (<= 0: automatic).
int sscan_n_fwd(const vmasr_sscan_params &p, int split, int tiles_per_task, int nseg, bool vec, hipStream_t st);
int sscan_n_bwd(const vmasr_sscan_bwd_params &q, int split, int tiles_per_task, int nseg, int rows, bool vec, hipStream_t st);

// sscan.hip: in-place scan of the per-tile aggregates (forward: inclusive; reverse: exclusive from the right)
void sscan_launch_carry(bool reverse, float *x, int nseq, int n_chunks, int N, double bytes, hipStream_t st);

}  // namespace vmasr
