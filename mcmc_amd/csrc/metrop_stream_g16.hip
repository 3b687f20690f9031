// metrop_stream_g16.hip -- metrop_stream_kernel<16, *> (see metrop_stream_impl.h)
#include "metrop_stream_impl.h"

int launch_stream_g16(const MetropDev &p, const StreamLayout &lay, dim3 grid, hipStream_t s)
{
    return launch_stream_group<16>(p, lay, grid, s);
}
