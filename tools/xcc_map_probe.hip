// xcc_map_probe.hip -- which XCD (accelerator die) does workgroup i of a grid run on?  Each workgroup records the XCC_ID
// hardware register.  Used to check the round-robin assumption behind an XCD-aware tile mapping.
// build: hipcc --offload-arch=gfx950 -O3 tools/xcc_map_probe.hip -o tools/xcc_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int *xcc)
{
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        xcc[blockIdx.y * gridDim.x + blockIdx.x] = (int)(v & 15);
    }
}
static void run(dim3 grid, int threads, size_t lds)
{
    int n = grid.x * grid.y, *d, h[4096];
    hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(probe, grid, dim3(threads), lds, 0, d);
    hipMemcpy(h, d, n * sizeof(int), hipMemcpyDeviceToHost);
    printf("grid %ux%u, %d threads, %zu B LDS:", grid.x, grid.y, threads, lds);
    int rr = 0;
    for (int i = 0; i < n; i++) rr += h[i] == (i % 8);
    for (int i = 0; i < 40 && i < n; i++) printf(" %d", h[i]);
    printf(" ...  xcc == id %% 8 for %d of %d\n", rr, n);
    hipFree(d);
}
int main()
{
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    run(dim3(256, 1), 512, 0);
    run(dim3(32, 8), 512, 0);
    run(dim3(32, 8), 512, 131072);
    run(dim3(32, 32), 512, 131072);
    return 0;
}
