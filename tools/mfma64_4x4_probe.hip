// mfma64_4x4_probe.hip -- (a) operand/result lane layout of v_mfma_f64_4x4x4_f64 (4 blocks of 4x4x4) found
// by one-hot inputs, (b) its issue rate against v_mfma_f64_16x16x4_f64.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma64_4x4_probe.hip -o tools/mfma64_4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void onehot(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

template <int KIND> __global__ __launch_bounds__(256) void rate(double *out, int iters, double seed)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    double a = seed + threadIdx.x, b = seed * 0.5, s = 0;
    if (KIND == 0) {
        double acc[16];
        for (int i = 0; i < 16; i++) acc[i] = seed;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int m = 0; m < 16; m++) acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m], 0, 0, 0);
        for (int i = 0; i < 16; i++) s += acc[i];
    } else {
        d4 acc[4];
        for (int i = 0; i < 4; i++) acc[i] = (d4){seed, seed, seed, seed};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int m = 0; m < 16; m++) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
        for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    std::vector<double> ha(64), hb(64), hd(64);
    // A one-hot at lane la, B all ones: D lanes that light up share (block, i) with la -> rows of A
    printf("A lane -> D lanes lit (B = 1):\n");
    for (int la = 0; la < 64; la++) {
        for (int i = 0; i < 64; i++) { ha[i] = i == la; hb[i] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        onehot<<<1, 64>>>(a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("A%02d:", la);
        for (int i = 0; i < 64; i++) if (hd[i] != 0) printf(" %d", i);
        printf("\n");
    }
    printf("B lane -> D lanes lit (A = 1):\n");
    for (int lb = 0; lb < 64; lb++) {
        for (int i = 0; i < 64; i++) { hb[i] = i == lb; ha[i] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        onehot<<<1, 64>>>(a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("B%02d:", lb);
        for (int i = 0; i < 64; i++) if (hd[i] != 0) printf(" %d", i);
        printf("\n");
    }
    // which (A lane, B lane) pairs meet (same k): A one-hot la=0..3 x B one-hot each lane
    printf("pairs (A lane, B lane) with a nonzero product, A lanes 0..15:\n");
    for (int la = 0; la < 16; la++) {
        printf("A%02d meets B:", la);
        for (int lb = 0; lb < 64; lb++) {
            for (int i = 0; i < 64; i++) { ha[i] = i == la; hb[i] = i == lb; }
            hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
            onehot<<<1, 64>>>(a, b, d);
            hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
            for (int i = 0; i < 64; i++) if (hd[i] != 0) printf(" %d(D%d)", lb, i);
        }
        printf("\n");
    }
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount, iters = 120000;     // >= 15 ms per kernel: shorter runs read the clock ramp (profiles/r2_fp64_sustain_probe.txt)
    double *out;
    hipMalloc(&out, (size_t)ncu * 2 * 256 * 8);
    for (int kind = 0; kind < 2; kind++)
        for (int bpc = 1; bpc <= 2; bpc++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            if (kind == 0) rate<0><<<ncu * bpc, 256>>>(out, 10, 1.0); else rate<1><<<ncu * bpc, 256>>>(out, 10, 1.0);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (kind == 0) rate<0><<<ncu * bpc, 256>>>(out, iters, 1.0); else rate<1><<<ncu * bpc, 256>>>(out, iters, 1.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)ncu * bpc * 4 * iters * 16 * (kind == 0 ? 512.0 : 2048.0);
            printf("%s blocks/CU=%d: %.3f ms  %.1f TF\n", kind == 0 ? "4x4x4 " : "16x16x4", bpc, ms, flops / ms * 1e-9);
        }
    return 0;
}
