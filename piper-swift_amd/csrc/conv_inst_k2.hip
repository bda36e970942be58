// conv_inst_k2.hip — instantiates the MFMA conv kernels for 2-tap filters (see conv_kernels.hpp).
#include "conv_kernels.hpp"

namespace ph {
namespace detail {
template bool launch_k<2, 32>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_k<2, 16>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_tile_k<2>(hipStream_t, const ConvArgs&, int, int, int);
}  // namespace detail
}  // namespace ph
namespace ph { namespace { PH_WARM(conv_inst_k2, (detail::conv_stream_kernel<2, 1, false, 0, 256, 16>)); } }
