// explicit instantiation of the fused detection head (conv_igemm_impl.hpp): YMI_BF16, anchor padding 96 / 128 rows
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_head_decode<YMI_BF16, 3>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
template int launch_head_group<YMI_BF16, 3>(const HeadGroupArgs&, hipStream_t);
template int launch_head_decode<YMI_BF16, 4>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
template int launch_head_group<YMI_BF16, 4>(const HeadGroupArgs&, hipStream_t);
}
