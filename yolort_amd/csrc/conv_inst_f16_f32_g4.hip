// explicit instantiation of one slice of the convolution tile space (see conv_igemm_impl.hpp): tile group 4 (row-transposed stores), YMI_F16 -> YMI_F32
#include "conv_igemm_impl.hpp"
namespace ymi {
template int launch_tile_group<4, YMI_F16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
}
