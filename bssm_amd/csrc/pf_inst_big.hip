// pf_inst_big.hip -- instantiations of the matrix-core filters of 17 ... 64 state dimensions (pf_big.h)
#include "pf_filter_impl.h"

namespace bssm {

template <> hipError_t launch_wg_mp<32>(const FilterBatch& b, hipStream_t stream) { return launch_big_wg<2>(b, stream); }
template <> hipError_t launch_wg_mp<64>(const FilterBatch& b, hipStream_t stream) { return launch_big_wg<4>(b, stream); }
// (the step shape of these states is the row family: a workspace carved by row_workspace_carve)
template <> hipError_t launch_grid_mp<32>(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  return ws.chunk == CANON_ROW ? launch_big_row<2>(b, ws, stream, timing) : hipErrorInvalidValue;
}
template <> hipError_t launch_grid_mp<64>(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  return ws.chunk == CANON_ROW ? launch_big_row<4>(b, ws, stream, timing) : hipErrorInvalidValue;
}

}  // namespace bssm
