// explicit instantiation of the fused detection head (conv_igemm_impl.hpp): YMI_F16, anchor padding 32 rows, group launch
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_head_group<YMI_F16, 1>(const HeadGroupArgs&, hipStream_t);
}
