// explicit instantiation of the fused detection head (conv_igemm_impl.hpp): YMI_F16, anchor padding 128 rows, single launch
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_head_decode<YMI_F16, 4>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
}
