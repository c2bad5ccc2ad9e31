// pf_inst_mp16.hip -- instantiations of the filter kernels for padded state dimension 16
#include "pf_filter_impl.h"

namespace bssm {

template <> hipError_t launch_wg_mp<16>(const FilterBatch& b, hipStream_t stream) {
  switch (b.KP) {
    case 2: return launch_wg_mk<16, 2>(b, stream);   // (k <= 2: the WIDE stream format is chosen by the real number of noise terms)
    case 4: return launch_wg_mk<16, 4>(b, stream);
    case 16: return launch_wg_mk<16, 16>(b, stream);
    default: return hipErrorInvalidValue;
  }
}
template <> hipError_t launch_grid_mp<16>(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  switch (b.KP) {
    case 2: return launch_grid_mk<16, 2>(b, ws, stream, timing);
    case 4: return launch_grid_mk<16, 4>(b, ws, stream, timing);
    case 16: return launch_grid_mk<16, 16>(b, ws, stream, timing);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bssm
