// explicit instantiation of the fused detection head (conv_igemm_impl.hpp): YMI_BF16, anchor padding 64 rows, group launch
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_head_group<YMI_BF16, 2>(const HeadGroupArgs&, hipStream_t);
}
