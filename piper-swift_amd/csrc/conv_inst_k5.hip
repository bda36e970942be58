// conv_inst_k5.hip — instantiates the MFMA conv kernels for 5-tap filters (see conv_kernels.hpp).
#include "conv_kernels.hpp"

namespace ph {
namespace detail {
template bool launch_k<5, 32>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_k<5, 16>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);
template bool launch_tile_k<5>(hipStream_t, const ConvArgs&, int, int, int);
}  // namespace detail
}  // namespace ph
namespace ph { namespace { PH_WARM(conv_inst_k5, (detail::conv_stream_kernel<5, 1, false, 0, 256, 16>)); } }
