// sscan_n.h — general d_state (N > 1) selective-scan kernels (sscan_n.hip), called from sscan.hip.
#pragma once
#include "common.h"

namespace vmasr {

// split_req: -1 automatic, 0 = the workgroups walk whole rows, 1 = 3-phase split along L (aggregates -> carry -> apply)
int sscan_n_fwd(const vmasr_sscan_params &p, int split_req, bool vec, hipStream_t st);
int sscan_n_bwd(const vmasr_sscan_bwd_params &q, int split_req, bool vec, hipStream_t st);
size_t sscan_n_bwd_ws_floats(const vmasr_sscan_params &p, int split_req);   // workspace of the backward under the same plan

// sscan.hip: in-place scan of the per-tile aggregates (forward: inclusive; reverse: exclusive from the right)
void sscan_launch_carry(bool reverse, float *x, int nseq, int n_chunks, int N, double bytes, hipStream_t st);

}  // namespace vmasr
