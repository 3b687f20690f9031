// hostcopy.hip -- pageable host memory <-> HBM for the large arrays of a call: results back to the caller's buffers,
// host inputs (a series, a design matrix, injected streams) up, and a host batch into HBM once for several diagnostics
// (mcmchip_series_upload / _centre / _free).
//
// The reference hands its results to R in freshly allocated R vectors (metrop.c:224-226, temper.c:614-640); through the
// C-ABI with device_ptrs == 0 that is a device-to-host copy into pageable memory nobody has touched yet.  One
// hipMemcpy of that kind runs at ~8 GB/s on the box (one runtime thread copies out of its staging buffer and takes
// every first-touch page fault), which for C2 (1.6 GB of batch means behind a 33 ms kernel) is six sevenths of the
// call.  Here the copy is pipelined: the DMA engine fills a ring of pinned chunks at PCIe rate while a few host threads
// move the chunk before it into the destination, page faults included.  No HIP call is made off the calling thread.
// Uploads run the same ring the other way round (host threads fill a chunk, the DMA engine empties the one before):
// 27 -> 50 GB/s.  MCMCHIP_COPY_THREADS sets the thread count (default 8), 0 takes the plain hipMemcpy.
#include "common.h"
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {
constexpr size_t HC_CHUNK = (size_t)16 << 20;     // bytes per pinned chunk
constexpr int HC_NBUF = 4;                        // chunks in the ring
constexpr size_t HC_MIN_BYTES = (size_t)64 << 20; // smaller copies go straight through hipMemcpyAsync

struct Ring {
    void *buf[HC_NBUF] = {nullptr, nullptr, nullptr, nullptr};
    bool tried = false, ok = false;
    int device = -1;                                // the device that was current when the ring was pinned
};
std::mutex g_ring_mutex;
Ring g_ring;                                       // pinned memory is per process, kept until it ends

bool ring_ready()
{
    if (g_ring.tried) return g_ring.ok;
    g_ring.tried = true;
    (void)hipGetDevice(&g_ring.device);
    for (int i = 0; i < HC_NBUF; i++)
        if (hipHostMalloc(&g_ring.buf[i], HC_CHUNK, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < i; j++) (void)hipHostFree(g_ring.buf[j]);
            return false;                          // no pinned memory: the plain copy still works
        }
    g_ring.ok = true;
    return true;
}

int copy_threads()
{
    const char *e = getenv("MCMCHIP_COPY_THREADS");
    int t = e ? atoi(e) : 8;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (unsigned)t > hw) t = (int)hw;
    return t < 1 ? 1 : (t > 32 ? 32 : t);
}
}  // namespace

// A chunk from HBM into the pinned ring.  Under the system's HIP runtime (no torch in the process) hipMemcpyAsync to pinned
// memory goes through the DMA engines at ~27 GB/s, a pitched hipMemcpy2DAsync -- a copy kernel -- at ~49 GB/s, the PCIe 5
// x16 rate: so the contiguous chunks go out by a copy kernel too, 16 bytes per lane straight into the mapped pinned
// buffer (MCMCHIP_COPY_DMA=1: the DMA engines).
__global__ __launch_bounds__(256) void ring_store_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n16,
                                                         const char *__restrict__ src_tail, char *__restrict__ dst_tail, int ntail)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
static hipError_t chunk_to_ring(void *ring, const void *dev, size_t len, hipStream_t s)
{
    static const bool dma = getenv("MCMCHIP_COPY_DMA") != nullptr;
    int cur = -1;
    (void)hipGetDevice(&cur);
    // a kernel stores into the ring only on the device the ring was pinned under (another device of a *_multi call takes the
    // DMA engines, which need no mapping)
    if (dma || cur != g_ring.device || ((uintptr_t)dev & 15)) return hipMemcpyAsync(ring, dev, len, hipMemcpyDeviceToHost, s);
    const size_t n16 = len / 16;
    hipLaunchKernelGGL(ring_store_kernel, dim3(256), dim3(256), 0, s, (const double2 *)dev, (double2 *)ring, n16,
                       (const char *)dev + n16 * 16, (char *)ring + n16 * 16, (int)(len - n16 * 16));
    return hipGetLastError();
}

// Everything enqueued on s before the call is complete when it returns (the copies are ordered behind it on s).
hipError_t copy_to_pageable_host(void *host, const void *dev, size_t bytes, hipStream_t s)
{
    const char *env = getenv("MCMCHIP_COPY_THREADS");                   // "0": the plain copy (for A/B measurements)
    if (bytes < HC_MIN_BYTES || (env && atoi(env) == 0)) return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    if (!ring_ready()) return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    hipEvent_t ev[HC_NBUF];
    for (int i = 0; i < HC_NBUF; i++) {
        hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipEventDestroy(ev[j]);
            return e;
        }
    }
    const long nchunks = (long)((bytes + HC_CHUNK - 1) / HC_CHUNK);
    const int T = copy_threads();
    std::atomic<long> ready{0};                    // chunks [0, ready) have landed in the ring
    std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nchunks]);
    for (long c = 0; c < nchunks; c++) done[c].store(0, std::memory_order_relaxed);
    auto chunk_len = [&](long c) { return c + 1 < nchunks ? HC_CHUNK : bytes - (size_t)c * HC_CHUNK; };
    auto worker = [&](int t) {
        for (long c = 0; c < nchunks; c++) {
            while (ready.load(std::memory_order_acquire) <= c) {
                if (stop.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            // slices on 4 KiB boundaries: a page of the destination is first touched by one thread
            const size_t len = chunk_len(c), per = (((len + T - 1) / T + 4095) / 4096) * 4096;   // ceil(len / T): T * per >= len
            const size_t a = (size_t)t * per, b = a + per < len ? a + per : len;
            if (a < len) memcpy((char *)host + (size_t)c * HC_CHUNK + a, (const char *)g_ring.buf[c % HC_NBUF] + a, b - a);
            done[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(T);
    for (int t = 0; t < T; t++) pool.emplace_back(worker, t);
    hipError_t rc = hipSuccess;
    // the calling thread: issue chunk c, then wait for chunk c - 1 (so one chunk is always queued behind the one in
    // flight) and hand it to the workers; a ring slot is reused once every worker is done with the chunk it held
    for (long c = 0; c <= nchunks && rc == hipSuccess; c++) {
        if (c < nchunks) {
            if (c >= HC_NBUF)
                while (done[c - HC_NBUF].load(std::memory_order_acquire) < T) std::this_thread::yield();
            rc = chunk_to_ring(g_ring.buf[c % HC_NBUF], (const char *)dev + (size_t)c * HC_CHUNK, chunk_len(c), s);
            if (rc == hipSuccess) rc = hipEventRecord(ev[c % HC_NBUF], s);
        }
        if (c >= 1 && rc == hipSuccess) {
            rc = hipEventSynchronize(ev[(c - 1) % HC_NBUF]);
            if (rc == hipSuccess) ready.store(c, std::memory_order_release);
        }
    }
    if (rc != hipSuccess) stop.store(true, std::memory_order_relaxed);
    for (auto &th : pool) th.join();
    for (int i = 0; i < HC_NBUF; i++) (void)hipEventDestroy(ev[i]);
    return rc;
}

// Rows of a pitched device array into the same rows of a pageable host array of the same pitch: row r = row_bytes bytes at
// offset r * pitch on both sides.  What a launch of a run leaves behind when the results are laid out [chain][batch][nout]
// and the launch produced batches [b0, b1) of every chain: the part of the result that can go home while the next
// launch runs (metrop_stream.hip).  A ring chunk takes as many whole rows as fit, packed, from one 2-d DMA; the host
// threads deal the rows of a chunk out among themselves.  Blocks until the rows have landed; s is idle on return.
hipError_t copy_rows_to_pageable_host(void *host, const void *dev, size_t nrows, size_t row_bytes, size_t pitch, hipStream_t s)
{
    if (!nrows || !row_bytes) return hipSuccess;
    const char *env = getenv("MCMCHIP_COPY_THREADS");
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    if (row_bytes > HC_CHUNK || (env && atoi(env) == 0) || !ring_ready()) {
        hipError_t e = hipMemcpy2DAsync(host, pitch, dev, pitch, row_bytes, nrows, hipMemcpyDeviceToHost, s);
        return e != hipSuccess ? e : hipStreamSynchronize(s);
    }
    hipEvent_t ev[HC_NBUF];
    for (int i = 0; i < HC_NBUF; i++) {
        hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipEventDestroy(ev[j]);
            return e;
        }
    }
    const size_t R = HC_CHUNK / row_bytes;                  // rows per chunk
    const long nchunks = (long)((nrows + R - 1) / R);
    const int T = copy_threads();
    std::atomic<long> ready{0};
    std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nchunks]);
    for (long c = 0; c < nchunks; c++) done[c].store(0, std::memory_order_relaxed);
    auto chunk_rows = [&](long c) { return (size_t)(c + 1) * R <= nrows ? R : nrows - (size_t)c * R; };
    auto worker = [&](int t) {
        for (long c = 0; c < nchunks; c++) {
            while (ready.load(std::memory_order_acquire) <= c) {
                if (stop.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const size_t rows = chunk_rows(c);
            // a thread takes neighbouring rows: the pages (2 MB ones, where the kernel hands them out) of its part of the
            // destination are first touched by it alone
            const size_t per = (rows + (size_t)T - 1) / (size_t)T, i0 = (size_t)t * per, i1 = i0 + per < rows ? i0 + per : rows;
            for (size_t i = i0; i < i1; i++)
                memcpy((char *)host + ((size_t)c * R + i) * pitch, (const char *)g_ring.buf[c % HC_NBUF] + i * row_bytes, row_bytes);
            done[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(T);
    for (int t = 0; t < T; t++) pool.emplace_back(worker, t);
    hipError_t rc = hipSuccess;
    for (long c = 0; c <= nchunks && rc == hipSuccess; c++) {
        if (c < nchunks) {
            if (c >= HC_NBUF)
                while (done[c - HC_NBUF].load(std::memory_order_acquire) < T) std::this_thread::yield();
            rc = hipMemcpy2DAsync(g_ring.buf[c % HC_NBUF], row_bytes, (const char *)dev + (size_t)c * R * pitch, pitch, row_bytes,
                                  chunk_rows(c), hipMemcpyDeviceToHost, s);
            if (rc == hipSuccess) rc = hipEventRecord(ev[c % HC_NBUF], s);
        }
        if (c >= 1 && rc == hipSuccess) {
            rc = hipEventSynchronize(ev[(c - 1) % HC_NBUF]);
            if (rc == hipSuccess) ready.store(c, std::memory_order_release);
        }
    }
    if (rc != hipSuccess) stop.store(true, std::memory_order_relaxed);
    for (auto &th : pool) th.join();
    for (int i = 0; i < HC_NBUF; i++) (void)hipEventDestroy(ev[i]);
    return rc;
}

// The other direction: pageable host memory into HBM.  One hipMemcpy runs at ~27 GB/s on the box (the runtime stages
// through its own pinned buffer on one thread); here host threads fill the pinned ring while the DMA engine empties the
// chunk before.  The source must stay valid until the call returns; on return the copies are enqueued AND complete.
hipError_t copy_from_pageable_host(void *dev, const void *host, size_t bytes, hipStream_t s)
{
    const char *env = getenv("MCMCHIP_COPY_THREADS");
    if (bytes < HC_MIN_BYTES || (env && atoi(env) == 0)) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    if (!ring_ready()) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
    hipEvent_t ev[HC_NBUF];
    for (int i = 0; i < HC_NBUF; i++) {
        hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipEventDestroy(ev[j]);
            return e;
        }
    }
    const long nchunks = (long)((bytes + HC_CHUNK - 1) / HC_CHUNK);
    const int T = copy_threads();
    std::atomic<long> freed{0};                    // chunks [0, freed) have left the ring: slot c % HC_NBUF may take chunk c when c < freed + HC_NBUF
    std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<int>[]> filled(new std::atomic<int>[nchunks]);
    for (long c = 0; c < nchunks; c++) filled[c].store(0, std::memory_order_relaxed);
    auto chunk_len = [&](long c) { return c + 1 < nchunks ? HC_CHUNK : bytes - (size_t)c * HC_CHUNK; };
    auto worker = [&](int t) {
        for (long c = 0; c < nchunks; c++) {
            while (freed.load(std::memory_order_acquire) + HC_NBUF <= c) {
                if (stop.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const size_t len = chunk_len(c), per = (((len + T - 1) / T + 4095) / 4096) * 4096;   // ceil(len / T): T * per >= len
            const size_t a = (size_t)t * per, b = a + per < len ? a + per : len;
            if (a < len) memcpy((char *)g_ring.buf[c % HC_NBUF] + a, (const char *)host + (size_t)c * HC_CHUNK + a, b - a);
            filled[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(T);
    for (int t = 0; t < T; t++) pool.emplace_back(worker, t);
    hipError_t rc = hipSuccess;
    long retired = 0;                              // chunks whose DMA is known complete
    for (long c = 0; c < nchunks && rc == hipSuccess; c++) {
        // while the workers fill chunk c, retire the DMAs that have finished (their slots go back to the workers)
        while (filled[c].load(std::memory_order_acquire) < T) {
            if (retired < c && hipEventQuery(ev[retired % HC_NBUF]) == hipSuccess) freed.store(++retired, std::memory_order_release);
            else std::this_thread::yield();
        }
        rc = hipMemcpyAsync((char *)dev + (size_t)c * HC_CHUNK, g_ring.buf[c % HC_NBUF], chunk_len(c), hipMemcpyHostToDevice, s);
        if (rc == hipSuccess) rc = hipEventRecord(ev[c % HC_NBUF], s);
        // a slot is needed for chunk c + 1 .. : if the ring is full, wait for the oldest DMA
        if (rc == hipSuccess && c + 1 - retired >= HC_NBUF) {
            rc = hipEventSynchronize(ev[retired % HC_NBUF]);
            if (rc == hipSuccess) freed.store(++retired, std::memory_order_release);
        }
    }
    if (rc != hipSuccess) stop.store(true, std::memory_order_relaxed);
    for (auto &th : pool) th.join();
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);      // the ring is free again when the lock is released
    (void)hipGetLastError();                                 // hipEventQuery's hipErrorNotReady is not an error of the call
    for (int i = 0; i < HC_NBUF; i++) (void)hipEventDestroy(ev[i]);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// mcmchip_series_upload / mcmchip_series_free (include/mcmchip.h): a host matrix into HBM in the column-major layout the
// diagnostics read, transposed on the device when the host holds it in C order.
// 64 x 64 tiles through LDS: rows of the input are read along p (contiguous), columns of the output written along n.
__global__ __launch_bounds__(256) void transpose_rows_kernel(const double *__restrict__ in, long n, long p,
                                                             double *__restrict__ out)
{
    __shared__ double tile[64][65];
    const long r0 = (long)blockIdx.x * 64, c0 = (long)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < n && c0 + tx < p) tile[i][tx] = in[(size_t)(r0 + i) * p + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 64; j += 4)
        if (c0 + j < p && r0 + tx < n) out[(size_t)(c0 + j) * n + r0 + tx] = tile[tx][j];
}

extern "C" int mcmchip_series_upload(const double *x, int64_t n, int64_t p, int row_major, double **dev)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!x || !dev || n <= 0 || p <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_series_upload");
    if ((n + 63) / 64 > 0x7fffffffL || (p + 63) / 64 > 65535) return set_error(MCMCHIP_EINVAL, "series too large to tile");
    *dev = nullptr;
    const size_t bytes = (size_t)n * (size_t)p * 8;
    DevBuf out, raw;
    HIP_TRY(out.alloc(bytes));
    hipStream_t s = dc->stream;
    if (!row_major || p == 1) {
        HIP_TRY(copy_from_pageable_host(out.p, x, bytes, s));
    } else {
        HIP_TRY(raw.alloc(bytes));
        HIP_TRY(copy_from_pageable_host(raw.p, x, bytes, s));
        hipLaunchKernelGGL(transpose_rows_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((p + 63) / 64)), dim3(256), 0, s,
                           raw.as<double>(), (long)n, (long)p, out.as<double>());
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s));            // raw is released on return
    *dev = out.as<double>();
    out.p = nullptr;                             // ownership passes to the caller
    return MCMCHIP_OK;
}

extern "C" int mcmchip_series_free(double *dev)
{
    clear_error();
    if (dev) HIP_TRY(hipFree(dev));
    return MCMCHIP_OK;
}
