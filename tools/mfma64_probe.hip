// mfma64_probe.hip -- measures on the box (a) the issue rate of v_mfma_f64_16x16x4_f64,
// (b) the fp64 VALU FMA rate, (c) whether the two overlap when issued from one wave and from
// two waves per SIMD.  Used to fix the roofline denominator for the fp64 MFMA path.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma64_probe.hip -o tools/mfma64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>   // per loop iteration: NM MFMAs (4 independent accumulators), NV dependent-free FMAs
__global__ __launch_bounds__(256) void probe(double *out, int iters, double seed)
{
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (d4){seed, seed, seed, seed};
    double a = seed + threadIdx.x, b = seed * 0.5;
    double v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < NM; m++) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; q++) v[q & 7] = __builtin_fma(v[q & 7], 0.999999, 1e-9);
    }
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV> static void run(const char *name, int blocks_per_cu, int ncu)
{
    double *out;
    int blocks = ncu * blocks_per_cu, iters = 4000;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NM, NV><<<blocks, 256>>>(out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NM, NV><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * 4;
    double mfma_flops = waves * iters * NM * 2048.0, valu_flops = waves * iters * NV * 128.0;
    printf("%-28s blocks/CU=%d  %.3f ms  MFMA %.1f TF  VALU %.1f TF  (per wave-iter %.0f ns)\n", name, blocks_per_cu, ms,
           mfma_flops / ms * 1e-9, valu_flops / ms * 1e-9, ms * 1e6 / iters);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", p.name, ncu, p.clockRate / 1000);
    for (int b = 1; b <= 2; b++) {
        run<16, 0>("mfma only (16/iter)", b, ncu);
        run<0, 64>("valu fma only (64/iter)", b, ncu);
        run<16, 64>("16 mfma + 64 fma", b, ncu);
        run<16, 128>("16 mfma + 128 fma", b, ncu);
        run<16, 224>("16 mfma + 224 fma", b, ncu);
    }
    return 0;
}
