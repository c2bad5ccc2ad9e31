// pf_inst_mp2.hip -- instantiations of the filter kernels for padded state dimension 2
#include "pf_filter_impl.h"

namespace bssm {

template <> hipError_t launch_wg_mp<2>(const FilterBatch& b, hipStream_t stream) {
  switch (b.KP) {
    case 1: return launch_wg_mk<2, 1>(b, stream);
    case 2: return launch_wg_mk<2, 2>(b, stream);
    default: return hipErrorInvalidValue;
  }
}
template <> hipError_t launch_grid_mp<2>(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  switch (b.KP) {
    case 1: return launch_grid_mk<2, 1>(b, ws, stream, timing);
    case 2: return launch_grid_mk<2, 2>(b, ws, stream, timing);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bssm
