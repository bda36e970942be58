// conv_inst_k7.hip — instantiates the MFMA conv kernels for 7-tap filters (see conv_kernels.hpp).
#include "conv_kernels.hpp"

namespace ph {
namespace detail {
template bool launch_k<7, 32>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_k<7, 16>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_tile_k<7>(hipStream_t, const ConvArgs&, int, int, int);
}  // namespace detail
}  // namespace ph
namespace ph { namespace { PH_WARM(conv_inst_k7, (detail::conv_stream_kernel<7, 1, false, 0, 256, 16>)); } }
