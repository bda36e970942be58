// conv_inst_k3.hip — instantiates the MFMA conv kernels for 3-tap filters (see conv_kernels.hpp).
#include "conv_kernels.hpp"

namespace ph {
namespace detail {
template bool launch_k<3, 32>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_k<3, 16>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_tile_k<3>(hipStream_t, const ConvArgs&, int, int, int);
}  // namespace detail
}  // namespace ph
namespace ph { namespace { PH_WARM(conv_inst_k3, (detail::conv_stream_kernel<3, 1, false, 0, 256, 16>)); } }
