// metrop_stream_g64.hip -- metrop_stream_kernel<64, *> (see metrop_stream_impl.h)
#include "metrop_stream_impl.h"

int launch_stream_g64(const MetropDev &p, const StreamLayout &lay, dim3 grid, hipStream_t s)
{
    return launch_stream_group<64>(p, lay, grid, s);
}
