// conv_inst_k1.hip — instantiates the MFMA conv kernels for 1-tap filters (see conv_kernels.hpp).
#include "conv_kernels.hpp"

namespace ph {
namespace detail {
template bool launch_k<1, 32>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_k<1, 16>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_tile_k<1>(hipStream_t, const ConvArgs&, int, int, int);
}  // namespace detail
}  // namespace ph
namespace ph { namespace { PH_WARM(conv_inst_k1, (detail::conv_stream_kernel<1, 1, false, 0, 256, 16>)); } }
