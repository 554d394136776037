// explicit instantiation of the fused detection head (conv_igemm_impl.hpp): YMI_BF16, anchor padding 32 rows, single launch
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_head_decode<YMI_BF16, 1>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
}
