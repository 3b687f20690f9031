// common.h -- shared host-side context, error plumbing and small device helpers of libmcmchip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/mcmchip.h"
#include "softplus.h"

#define MCMCHIP_MAX_DEVICES 16
#define MCMCHIP_MFMA_MAXD 256
#define MCMCHIP_NOALLOC_SCRATCH_DOUBLES (4L << 20)   // 32 MiB per GPU (include/mcmchip.h: MCMCHIP_NOALLOC_SCRATCH_BYTES)

struct DeviceCtx {
    bool inited = false;
    int device = -1;                     // HIP device this context lives on
    hipStream_t stream = nullptr;
    SoftplusTables *d_tables = nullptr;  // softplus tables in HBM
    int *d_flag = nullptr;               // error flag word shared by the kernels
    int num_cu = 256;
    // scratch of the MFMA path (state / proposal / partial sums / accumulators), kept between calls: a call used to pay
    // seven hipMalloc/hipFree pairs (~3 ms).  Grows on demand; released after a call when it exceeds MCMCHIP_WS_KEEP.
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // two side streams for launches that may overlap (initseq: HBM-bound short-lag items next to MFMA-bound long ones)
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[2] = {nullptr, nullptr};
    // device address of this GPU's copy of the file-scope scratch array of mcmchip_logdens_noalloc (metrop_mfma.hip)
    double *noalloc_scratch = nullptr;
};
#define MCMCHIP_WS_KEEP ((size_t)2 << 30)

// timing of the dominant kernel of a call (HIP events on the stream it runs on)
struct TimingState {
    std::vector<hipEvent_t> ev_start, ev_stop;
    int ev_used = 0;
    char ev_name[64] = {0};
    int device = -1;                     // device the events were created on
};

// per-process state (one process per GPU in the multi-GPU launch, but any device may be selected)
struct LibCtx {
    int device = -1;
    DeviceCtx dev[MCMCHIP_MAX_DEVICES];
    TimingState timing;                  // of calls made on the caller's thread
    TimingState *last_timing = nullptr;  // what mcmchip_last_kernel_time reports (a *_multi call: its first worker's)
    // persistent sampler kernels are cut into launches of about launch_budget_s seconds (max_iters_per_launch > 0
    // fixes the iterations per launch instead); between launches the host calls `poll` (R_CheckUserInterrupt's
    // place, metrop.c:269 / temper.c:848): a non-zero return abandons the call with MCMCHIP_EINTR
    int (*poll)(void *) = nullptr;
    void *poll_ctx = nullptr;
    long max_iters_per_launch = 0;
    double launch_budget_s = 0.25;
};

LibCtx &lib_ctx();
DeviceCtx *cur_dev();  // nullptr (and error set) when mcmchip_init has not succeeded
int cur_device_index();            // device of cur_dev() (-1: none)
int init_device_ctx(int device, DeviceCtx &D);      // streams, softplus tables, flag words on `device` (made current)

// A worker thread of a *_multi call (multi.hip) runs the ordinary entry points against a context of its own: its
// device context (own streams, flag words, MFMA workspace), its kernel timing, and an abort word in place of the
// caller's interrupt callback, which only the calling thread may run (R_CheckUserInterrupt is not thread safe).
struct WorkerScope {
    WorkerScope(DeviceCtx *dc, TimingState *timing, volatile int *abort_word);
    ~WorkerScope();
};

int set_error(int code, const char *fmt, ...);
void clear_error();

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return set_error(MCMCHIP_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                             __FILE__, __LINE__);                                              \
    } while (0)

// device buffer freed at scope exit (outputs staged through HBM when device_ptrs == 0)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n)
    {
        bytes = n;
        return hipMalloc(&p, n ? n : 8);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() { return (T *)p; }
};

// hostcopy.hip: pageable host memory <-> HBM; large copies are pipelined through pinned chunks and host threads
hipError_t copy_to_pageable_host(void *host, const void *dev, size_t bytes, hipStream_t s);
hipError_t copy_from_pageable_host(void *dev, const void *host, size_t bytes, hipStream_t s);
hipError_t copy_rows_to_pageable_host(void *host, const void *dev, size_t nrows, size_t row_bytes, size_t pitch, hipStream_t s);

// bring an input into HBM: either the caller's device pointer as-is, or a staged copy
struct InBuf {
    DevBuf own;
    const void *ptr = nullptr;
    hipError_t set(const void *src, size_t bytes, int device_ptrs, hipStream_t s)
    {
        if (!src || bytes == 0) { ptr = nullptr; return hipSuccess; }
        if (device_ptrs) { ptr = src; return hipSuccess; }
        hipError_t e = own.alloc(bytes);
        if (e != hipSuccess) return e;
        ptr = own.p;
        return copy_from_pageable_host(own.p, src, bytes, s);
    }
    template <class T> const T *as() const { return (const T *)ptr; }
};

// an output: the caller's device pointer, or a device staging buffer copied back at the end
struct OutBuf {
    DevBuf own;
    void *ptr = nullptr;
    void *host = nullptr;
    size_t bytes = 0;
    hipError_t set(void *dst, size_t n, int device_ptrs)
    {
        bytes = n;
        if (!dst) { ptr = nullptr; return hipSuccess; }
        if (device_ptrs) { ptr = dst; return hipSuccess; }
        host = dst;
        hipError_t e = own.alloc(n);
        ptr = own.p;
        return e;
    }
    bool went_home = false;      // the run has copied it back already, piece by piece (metrop_stream.hip)
    hipError_t finish(hipStream_t s)
    {
        if (went_home) return hipSuccess;
        if (host && ptr && bytes) return copy_to_pageable_host(host, ptr, bytes, s);
        return hipSuccess;
    }
    template <class T> T *as() { return (T *)ptr; }
};

// Separately rounded multiply / add.  HIP's __dmul_rn/__dadd_rn are plain `*`/`+` and get
// contracted into v_fma_f64 under hipcc's default -ffp-contract=fast-honor-pragmas; the reference
// is built without FMA contraction (x86-64 R), and bitwise-equal proposals need the two roundings.
__device__ __forceinline__ double mul_nofma(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double add_nofma(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}

// error flag bits written by kernels
#define FLAG_NEGINF_INIT 1  // log unnormalized density -Inf at initial state (metrop.c:196-197)
#define FLAG_POSINF 2       // logh: func returned +Inf (metrop.c:317-318)
#define FLAG_NAN 4          // logh: func returned NA or NaN (metrop.c:319-320)
#define FLAG_HASTINGS 8     // temper: log hastings ratio +Inf or NaN (temper.c:578-580)

struct mcmchip_target {
    int kind = 0, d = 0;
    int64_t n = 0;
    double prior_c = 0;
    double *x = nullptr;    // n x d row-major (logistic kinds)
    double *y = nullptr;    // n
    double *par = nullptr;
    int64_t npar = 0;
    std::vector<double> host_par;
    // morph (R/morph.R): flags, shape parameters, centre (d doubles in HBM or null)
    int morph_flags = 0;
    double morph_b = 1, morph_r = 0, morph_p = 3;
    double *morph_center = nullptr;
    // fp64 MFMA layout (logistic kind only): sign-folded, zero-padded, fragment-major tiles
    int dpad = 0, ksteps = 0;
    int64_t ntiles = 0;     // 16-observation tiles (padded up to a multiple of MFMA_TILES_PER_CHUNK)
    int64_t npad_rows = 0;  // zero rows appended (each contributes exactly -L(0) = -ln 2)
    double *xt = nullptr;
    double *colsum = nullptr;  // [dpad] column sums of the sign-folded X (sum_i eta_i = colsum . theta)
    double *wt = nullptr;      // row weights, tile-major (MCMCHIP_T_BINOMIAL_LOGIT: the totals; null: unweighted)
    int device = 0;
    // copies in the HBM of other devices, made on demand by the *_multi entry points and released with the original
    std::vector<mcmchip_target *> replicas;
};

// kernel timing hooks (metrop_mfma.hip uses them around the log-density launches)
void timing_reset(const char *name);
void timing_begin(hipStream_t s);
void timing_end(hipStream_t s);
double timing_last_ms();          // waits for the launch timed last and returns its duration (-1: not available)

// launch chunking of the persistent kernels: iterations for the next launch, given the iterations and duration of the
// previous one (0 iterations: first launch, a short calibration launch); rounded up to a multiple of `align`
long next_launch_iters(long remaining, long last_iters, double last_ms, long align);
// host poll point between launches: MCMCHIP_OK, or MCMCHIP_EINTR (error text set) when the caller's callback asks to stop
int poll_interrupt();
