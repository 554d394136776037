// explicit instantiation of one slice of the convolution tile space (see conv_igemm_impl.hpp): tile group 4 (row-transposed stores), YMI_BF16 -> YMI_BF16
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_tile_group<4, YMI_BF16, YMI_BF16>(const ConvArgs&, bool, int, hipStream_t);
}
