// fp64_sustain_probe.hip -- why does a pure stream of v_mfma_f64_16x16x4_f64 (or of v_fma_f64) reach ~67 TF on the
// box and not the nominal 78.6 TF (256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz)?  Two readings: (a) an issue gap per
// instruction that more wavefronts per SIMD would close, (b) the chip lowers its rate under an all-CU fp64 load
// (power management), which nothing in a kernel can change.  This probe separates them: the same instruction stream
// with 1, 2 and 4 wavefronts per SIMD, on all CUs and on a few CUs only, with the shader clock measured from inside
// the kernel (cycle counter against the constant 100 MHz wall clock).
// build: hipcc --offload-arch=gfx950 -O3 tools/fp64_sustain_probe.hip -o tools/fp64_sustain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

// KIND 0: 16 MFMA 16x16x4 per iteration (4 independent accumulators), 1: 128 independent v_fma_f64,
// 2/3/4: 16 MFMA + 64 v_fma_f64 in 8 / 2 / 1 dependent chains, interleaved one MFMA : four FMAs (does the vector FMA
// issue under the matrix instruction or after it, and what does a dependent chain -- a Horner evaluation -- cost?)
// 10 + g: the same 16 MFMA + 64 FMA (8 chains) in groups of g MFMAs followed by 4g FMAs (g = 2, 4, 8, 16): is there a
// cost per switch between the matrix and the vector instruction?
template <int KIND>
__global__ __launch_bounds__(1024) void probe(double *out, long long *clk, int iters, double seed)
{
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (d4){seed, seed, seed, seed};
    double a = seed + threadIdx.x, b = seed * 0.5;
    double v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int m = 0; m < 16; m++) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
        } else if (KIND >= 41 && KIND <= 44) {      // 16 MFMA round-robin over KIND - 40 accumulators (dependent every n-th)
#pragma unroll
            for (int m = 0; m < 16; m++) acc[m % (KIND - 40)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m % (KIND - 40)], 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int q = 0; q < 128; q++) v[q & 7] = __builtin_fma(v[q & 7], 0.999999, 1e-9);
        } else if (KIND > 10) {
            constexpr int G = KIND - 10;
#pragma unroll
            for (int m = 0; m < 16; m++) {
                acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) v[(4 * m + q) % 8] = __builtin_fma(v[(4 * m + q) % 8], 0.999999, 1e-9);
            }
#pragma unroll
            for (int m = 0; m < 16 / G; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4 * G, 0);
            }
        } else {
            constexpr int NCH = KIND == 2 ? 8 : (KIND == 3 ? 2 : 1);
#pragma unroll
            for (int m = 0; m < 16; m++) {
                acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) v[(4 * m + q) % NCH] = __builtin_fma(v[(4 * m + q) % NCH], 0.999999, 1e-9);
            }
#pragma unroll
            for (int m = 0; m < 16; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int KIND> static void run(const char *name, int blocks, int threads, int iters, int wall_khz)
{
    double *out;
    long long *clk, *h = (long long *)malloc((size_t)blocks * 16);
    hipMalloc(&out, (size_t)blocks * threads * 8);
    hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, threads>>>(out, clk, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<blocks, threads>>>(out, clk, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    cyc /= blocks; wall /= blocks;
    const double waves = (double)blocks * threads / 64;
    const double flop = waves * iters * (KIND == 1 ? 128 * 128.0 : 16 * 2048.0);     // mixed kinds: the MFMA flops
    const double secs = wall / (wall_khz * 1e3);
    // per-SIMD instruction interval in wall ns and in counted cycles (waves per SIMD share the pipe)
    const double wps = threads / 256.0, ninst = (double)iters * (KIND == 1 ? 128 : 16) * wps;
    printf("%-14s blocks=%4d waves/SIMD=%d  %8.3f ms  %6.2f TF  per-CU %6.1f GF  counter/wall = %.3f GHz  per instr %.2f ns = %.2f counted cycles\n",
           name, blocks, (int)wps, ms, flop / ms * 1e-9, flop / ms * 1e-6 / blocks, cyc / secs * 1e-9, secs * 1e9 / ninst, cyc / ninst);
    hipFree(out); hipFree(clk); free(h);
}

int main(int argc, char **argv)
{
    const int mult = argc > 1 ? atoi(argv[1]) : 1;      // run-length multiplier: does the rate hold over 100+ ms?
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount, wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, clock %d MHz, wall clock %d kHz\n", p.name, ncu, p.clockRate / 1000, wall_khz);
    const int used[] = {ncu, 8};
    for (int u = 0; u < 2; u++)
        for (int t = 256; t <= 1024; t *= 2) {
            run<0>("mfma", used[u], t, mult * 40000 * 256 / t, wall_khz);
            run<1>("fma", used[u], t, mult * 20000 * 256 / t, wall_khz);
            if (u == 0 && mult == 1) {
                run<41>("mfma 1 acc", used[u], t, 40000 * 256 / t, wall_khz);
                run<42>("mfma 2 acc", used[u], t, 40000 * 256 / t, wall_khz);
                run<43>("mfma 3 acc", used[u], t, 40000 * 256 / t, wall_khz);
                run<2>("mfma+4fma/8ch", used[u], t, 40000 * 256 / t, wall_khz);
                run<3>("mfma+4fma/2ch", used[u], t, 40000 * 256 / t, wall_khz);
                run<4>("mfma+4fma/1ch", used[u], t, 40000 * 256 / t, wall_khz);
                run<12>("2mfma+8fma", used[u], t, 40000 * 256 / t, wall_khz);
                run<14>("4mfma+16fma", used[u], t, 40000 * 256 / t, wall_khz);
                run<18>("8mfma+32fma", used[u], t, 40000 * 256 / t, wall_khz);
                run<26>("16mfma+64fma", used[u], t, 40000 * 256 / t, wall_khz);
            }
        }
    return 0;
}
