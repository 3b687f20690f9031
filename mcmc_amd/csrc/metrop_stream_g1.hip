// metrop_stream_g1.hip -- metrop_stream_kernel<1, *> (see metrop_stream_impl.h)
#include "metrop_stream_impl.h"

int launch_stream_g1(const MetropDev &p, const StreamLayout &lay, dim3 grid, hipStream_t s)
{
    return launch_stream_group<1>(p, lay, grid, s);
}
