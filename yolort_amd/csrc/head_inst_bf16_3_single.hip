// explicit instantiation of the fused detection head (conv_igemm_impl.hpp): YMI_BF16, anchor padding 96 rows, single launch
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_head_decode<YMI_BF16, 3>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
}
