// hbm_granule_probe.hip -- read bandwidth of the olbm input pattern as a function of the contiguous run per column.
// x is p = 256 columns of n doubles (column-major).  256 workgroups x 8 waves; workgroup g walks the time segment
// [g*seg, (g+1)*seg) of ALL columns, wave w owning 32 of them, in steps of G rows, reading two streams (rows t and t+len,
// as olbm's x[t] and x[t+len]); each step a wave issues all its loads (8-byte lanes, min(64, G) lanes per column), then
// adds them up.  G = 16 is olbm_mfma_kernel's pattern (128-byte runs); larger G = longer runs per column.
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_granule_probe.hip -o tools/hbm_granule_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int G, int DEPTH>    // DEPTH: steps in flight (register double / triple buffering)
__global__ __launch_bounds__(512) void probe(const double *__restrict__ x, long n, long len, long seg, double *out)
{
    constexpr int LPC = G < 64 ? G : 64;          // lanes per column
    constexpr int CPI = 64 / LPC;                 // columns per instruction
    constexpr int RPC = G / LPC;                  // instructions per column (G > 64)
    constexpr int NI = 32 / CPI * RPC;            // instructions per stream and step
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long t0 = (long)blockIdx.x * seg;
    const double *base = x + (size_t)(wave * 32 + lane / LPC) * n + t0 + lane % LPC;
    double v[DEPTH][2][NI];
    double s = 0.0;
    auto issue = [&](int slot, long t) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const double *p = base + (size_t)((i / RPC) * CPI) * n + t + (i % RPC) * 64;
            v[slot][0][i] = p[0];
            v[slot][1][i] = p[len];
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) issue(d, (long)d * G);
    long t = 0;
    for (; t + (DEPTH - 1) * G < seg; t += (long)DEPTH * G) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            issue((d + DEPTH - 1) % DEPTH, t + (long)(d + DEPTH - 1) * G);
#pragma unroll
            for (int i = 0; i < NI; i++) s += v[d][0][i] - v[d][1][i];
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int G, int DEPTH> static void run(const double *x, long n, long len, double *out)
{
    const long seg = ((n - len - 4096) / 256 / (DEPTH * G)) * (DEPTH * G);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<G, DEPTH><<<256, 512>>>(x, n, len, seg / 16, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<G, DEPTH><<<256, 512>>>(x, n, len, seg, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * 256 * 8 * (double)seg * 256;
    printf("run per column %5d B  steps in flight %d  %8.3f ms  %6.2f TB/s requested (two streams; the second stream re-reads the first %ld rows later)\n",
           G * 8, DEPTH - 1, ms, bytes / ms * 1e-9, len);
}

int main()
{
    const long n = 20000256, len = 10000;
    double *x, *out;
    if (hipMalloc(&x, (size_t)n * 256 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 256 * 512 * 8);
    hipMemset(x, 0, (size_t)n * 256 * 8);
    run<16, 2>(x, n, len, out);
    run<16, 3>(x, n, len, out);
    run<32, 2>(x, n, len, out);
    run<64, 2>(x, n, len, out);
    run<128, 2>(x, n, len, out);
    return 0;
}
