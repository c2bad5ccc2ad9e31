// pf_inst_mp8.hip -- instantiations of the filter kernels for padded state dimension 8
#include "pf_filter_impl.h"

namespace bssm {

template <> hipError_t launch_wg_mp<8>(const FilterBatch& b, hipStream_t stream) {
  switch (b.KP) {
    case 2: return launch_wg_mk<8, 2>(b, stream);
    case 4: return launch_wg_mk<8, 4>(b, stream);
    case 8: return launch_wg_mk<8, 8>(b, stream);
    default: return hipErrorInvalidValue;
  }
}
template <> hipError_t launch_grid_mp<8>(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  switch (b.KP) {
    case 2: return launch_grid_mk<8, 2>(b, ws, stream, timing);
    case 4: return launch_grid_mk<8, 4>(b, ws, stream, timing);
    case 8: return launch_grid_mk<8, 8>(b, ws, stream, timing);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bssm
