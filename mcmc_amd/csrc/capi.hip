// capi.hip -- the extern "C" entry points of include/mcmchip.h: argument validation with the
// reference's error texts, staging of host buffers through HBM, path selection, error flags.
#include "kernels.h"
#include <math.h>
#include <new>

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { g_err[0] = 0; }

LibCtx &lib_ctx()
{
    static LibCtx c;
    return c;
}

static thread_local DeviceCtx *tl_dc = nullptr;           // set in the worker threads of a *_multi call
static thread_local TimingState *tl_timing = nullptr;
static thread_local volatile int *tl_abort = nullptr;

WorkerScope::WorkerScope(DeviceCtx *dc, TimingState *timing, volatile int *abort_word)
{
    tl_dc = dc;
    tl_timing = timing;
    tl_abort = abort_word;
}
WorkerScope::~WorkerScope()
{
    tl_dc = nullptr;
    tl_timing = nullptr;
    tl_abort = nullptr;
}

static TimingState &cur_timing()
{
    if (tl_timing) return *tl_timing;
    LibCtx &L = lib_ctx();
    L.last_timing = &L.timing;
    return L.timing;
}

int cur_device_index() { return tl_dc ? tl_dc->device : lib_ctx().device; }

DeviceCtx *cur_dev()
{
    if (tl_dc) {
        if (hipSetDevice(tl_dc->device) != hipSuccess) {
            set_error(MCMCHIP_EHIP, "hipSetDevice(%d) failed", tl_dc->device);
            return nullptr;
        }
        return tl_dc;
    }
    LibCtx &L = lib_ctx();
    if (L.device < 0 || !L.dev[L.device].inited) {
        set_error(MCMCHIP_EHIP, "mcmchip_init() has not been called (no HIP device selected)");
        return nullptr;
    }
    if (hipSetDevice(L.device) != hipSuccess) {
        set_error(MCMCHIP_EHIP, "hipSetDevice(%d) failed", L.device);
        return nullptr;
    }
    return &L.dev[L.device];
}

void timing_reset(const char *name)
{
    TimingState &T = cur_timing();
    T.ev_used = 0;
    snprintf(T.ev_name, sizeof T.ev_name, "%s", name);
}
// one event pair per timed launch; the pool grows with the longest call (EV_MAX only bounds a runaway loop)
static const int EV_MAX = 1 << 20;
void timing_begin(hipStream_t s)
{
    TimingState &T = cur_timing();
    if (T.ev_used >= EV_MAX) return;
    if ((int)T.ev_start.size() <= T.ev_used) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        T.ev_start.push_back(a);
        T.ev_stop.push_back(b);
        T.device = cur_device_index();
    }
    (void)hipEventRecord(T.ev_start[T.ev_used], s);
}
void timing_end(hipStream_t s)
{
    TimingState &T = cur_timing();
    if (T.ev_used >= EV_MAX || (int)T.ev_stop.size() <= T.ev_used) return;
    (void)hipEventRecord(T.ev_stop[T.ev_used], s);
    T.ev_used++;
}

// Inputs handed over as device pointers are validated on the device: the reference refuses a non-finite `initial`
// (metrop.c:72-73, temper.c:129-131) or series (initseq.c:15) and, for serial tempering, a first element that is
// not an integer in 1..k (temper.c:55-63) -- the kernels index LDS with it.
#define VAL_NONFINITE 1
#define VAL_NOTINT 2
#define VAL_RANGE 4
__global__ void validate_inputs_kernel(const double *__restrict__ v, long n, long stride, int k, int *__restrict__ flag)
{
    const long step = (long)gridDim.x * blockDim.x;
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const double x = v[i];
        if (!isfinite(x)) bad |= VAL_NONFINITE;
        else if (k > 0 && i % stride == 0) {
            if (x != trunc(x)) bad |= VAL_NOTINT;
            else if (x < 1.0 || x > (double)k) bad |= VAL_RANGE;
        }
    }
    if (bad) atomicOr(flag, bad);
}

// runs the validation kernel over v[0..n) and returns the VAL_* bits in *bits (k > 0: every stride-th element is a
// serial-tempering component index)
int device_validate(DeviceCtx *dc, const double *v, long n, long stride, int k, int *bits)
{
    int *f = dc->d_flag + 1;          // word 0 is the samplers' error flag
    HIP_TRY(hipMemsetAsync(f, 0, sizeof(int), dc->stream));
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(validate_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, dc->stream, v, n, stride, k, f);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bits, f, sizeof(int), hipMemcpyDeviceToHost, dc->stream));
    HIP_TRY(hipStreamSynchronize(dc->stream));
    return MCMCHIP_OK;
}

double timing_last_ms()
{
    TimingState &T = cur_timing();
    if (T.ev_used <= 0 || (int)T.ev_stop.size() < T.ev_used) return -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(T.ev_stop[T.ev_used - 1]) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, T.ev_start[T.ev_used - 1], T.ev_stop[T.ev_used - 1]) != hipSuccess) return -1.0;
    return (double)ms;
}

long next_launch_iters(long remaining, long last_iters, double last_ms, long align)
{
    LibCtx &L = lib_ctx();
    long c;
    if (L.max_iters_per_launch > 0) c = L.max_iters_per_launch;
    else if (last_iters <= 0) c = 256;                      // calibration launch
    else if (last_ms > 0.0) {
        const double per_iter_ms = last_ms / (double)last_iters;
        const double want = L.launch_budget_s * 1e3 / per_iter_ms;
        c = want > 4e18 ? remaining : (long)want;
    } else c = last_iters * 4;
    if (c < 1) c = 1;
    if (align > 1) c = ((c + align - 1) / align) * align;
    return c < remaining ? c : remaining;
}

int poll_interrupt()
{
    if (tl_abort)        // a worker of a *_multi call: the calling thread runs the callback and raises this word
        return *tl_abort ? set_error(MCMCHIP_EINTR, "interrupted between two kernel launches") : MCMCHIP_OK;
    LibCtx &L = lib_ctx();
    if (L.poll && L.poll(L.poll_ctx)) return set_error(MCMCHIP_EINTR, "interrupted between two kernel launches");
    return MCMCHIP_OK;
}

int init_device_ctx(int device, DeviceCtx &D)
{
    HIP_TRY(hipSetDevice(device));
    if (D.inited) return MCMCHIP_OK;
    D.device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    D.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipStreamCreateWithFlags(&D.side[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&D.join_ev[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&D.fork_ev, hipEventDisableTiming));
    SoftplusTables host;
    softplus_tables_init(&host);
    HIP_TRY(hipMalloc((void **)&D.d_tables, sizeof host));
    HIP_TRY(hipMemcpy(D.d_tables, &host, sizeof host, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&D.d_flag, 256));
    HIP_TRY(hipMemset(D.d_flag, 0, 256));
    if (int rc = mfma_scratch_setup(&D)) return rc;      // this device's address of the no-alloc scratch array
    D.inited = true;
    return MCMCHIP_OK;
}

extern "C" {

int mcmchip_set_interrupt_poll(int (*poll)(void *), void *ctx)
{
    lib_ctx().poll = poll;
    lib_ctx().poll_ctx = ctx;
    return MCMCHIP_OK;
}

int mcmchip_set_launch_limit(int64_t max_iters, double budget_seconds)
{
    clear_error();
    if (max_iters < 0) return set_error(MCMCHIP_EINVAL, "max_iters must be >= 0");
    lib_ctx().max_iters_per_launch = (long)max_iters;
    if (budget_seconds > 0.0) lib_ctx().launch_budget_s = budget_seconds;
    return MCMCHIP_OK;
}

const char *mcmchip_last_error(void) { return g_err; }
const char *mcmchip_version(void) { return "mcmchip 0.1 (gfx950)"; }

int mcmchip_init(int device)
{
    clear_error();
    if (device < 0 || device >= MCMCHIP_MAX_DEVICES) return set_error(MCMCHIP_EINVAL, "bad device index %d", device);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device >= ndev) return set_error(MCMCHIP_EHIP, "device %d requested but only %d visible", device, ndev);
    LibCtx &L = lib_ctx();
    if (int rc = init_device_ctx(device, L.dev[device])) return rc;
    L.device = device;
    return MCMCHIP_OK;
}

int mcmchip_sync(void)
{
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    HIP_TRY(hipStreamSynchronize(dc->stream));
    return MCMCHIP_OK;
}

int mcmchip_last_kernel_time(double *avg_ms, int64_t *launches, char *name, int name_cap)
{
    LibCtx &L = lib_ctx();
    TimingState &T = L.last_timing ? *L.last_timing : L.timing;
    int back = -1;
    if (T.device >= 0 && hipGetDevice(&back) == hipSuccess && back != T.device) (void)hipSetDevice(T.device);
    else back = -1;
    double tot = 0.0;
    int n = 0;
    for (int i = 0; i < T.ev_used; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(T.ev_stop[i]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, T.ev_start[i], T.ev_stop[i]) != hipSuccess) continue;
        tot += ms;
        n++;
    }
    if (back >= 0) (void)hipSetDevice(back);
    if (avg_ms) *avg_ms = n ? tot / n : 0.0;
    if (launches) *launches = n;
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", T.ev_name);
    return MCMCHIP_OK;
}

// ---------------------------------------------------------------------------------------------
static bool is_logistic(int kind)      // kinds that carry a design matrix x (n x d) and a response y (n)
{
    return kind == MCMCHIP_T_LOGISTIC || kind == MCMCHIP_T_LOGISTIC_NAIVE || kind == MCMCHIP_T_LOGISTIC_MODELS ||
           kind == MCMCHIP_T_BINOMIAL_LOGIT;
}
static bool is_temper_kind(int kind) { return kind == MCMCHIP_T_WITCH || kind == MCMCHIP_T_TEMPERED_NORMMIX || kind == MCMCHIP_T_LOGISTIC_MODELS; }

int mcmchip_target_create(const mcmchip_target_desc *desc, int desc_device_ptrs, mcmchip_target **out)
{
    clear_error();
    if (!desc || !out) return set_error(MCMCHIP_EINVAL, "null argument");
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (desc->kind < MCMCHIP_T_LOGISTIC || desc->kind > MCMCHIP_T_BINOMIAL_LOGIT)
        return set_error(MCMCHIP_EINVAL, "unknown log density descriptor kind %d", desc->kind);
    if (desc->d <= 0) return set_error(MCMCHIP_EINVAL, "descriptor dimension must be positive");
    mcmchip_target *t = new (std::nothrow) mcmchip_target();
    if (!t) return set_error(MCMCHIP_EHIP, "out of host memory");
    t->kind = desc->kind; t->d = desc->d; t->prior_c = desc->prior_c; t->device = dc->device;
    hipMemcpyKind kind = desc_device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    int rc = MCMCHIP_OK;
    try {
    do {
        if (is_logistic(desc->kind)) {
            if (desc->n <= 0 || !desc->x || !desc->y) { rc = set_error(MCMCHIP_EINVAL, "logistic descriptor needs x (n x d) and y (n)"); break; }
            t->n = desc->n;
            size_t xb = (size_t)desc->n * desc->d * 8, yb = (size_t)desc->n * 8;
            if (hipMalloc((void **)&t->x, xb) != hipSuccess || hipMalloc((void **)&t->y, yb) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "hipMalloc of the design matrix failed"); break; }
            // (a host design matrix of 64 MB and more goes up through the pinned ring of hostcopy.hip)
            if ((desc_device_ptrs ? hipMemcpyAsync(t->x, desc->x, xb, kind, dc->stream)
                                  : copy_from_pageable_host(t->x, desc->x, xb, dc->stream)) != hipSuccess ||
                hipMemcpyAsync(t->y, desc->y, yb, kind, dc->stream) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "copy of the design matrix failed"); break; }
            // fp64 MFMA layout: the logistic posterior, and (for mcmchip_logdens) the model-selection posterior, whose
            // masks are folded into the states; the binomial regression needs its totals first (below)
            if ((desc->kind == MCMCHIP_T_LOGISTIC || desc->kind == MCMCHIP_T_LOGISTIC_MODELS) && desc->d <= MCMCHIP_MFMA_MAXD) {
                rc = build_mfma_tiles(t, dc->stream);
                if (rc) break;
            }
        }
        if (desc->npar > 0 && desc->par) {
            t->npar = desc->npar;
            t->host_par.resize((size_t)desc->npar);
            if (desc_device_ptrs) {
                if (hipMemcpy(t->host_par.data(), desc->par, (size_t)desc->npar * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "copy of descriptor parameters failed"); break; }
            } else {
                memcpy(t->host_par.data(), desc->par, (size_t)desc->npar * 8);
            }
            {
                // length of the parameter block per kind (include/mcmchip.h): the kernels take K and k from par[0] and
                // index the block with them, so a short block would be read out of bounds
                const double *hp = t->host_par.data();
                const long np_ = (long)desc->npar, dd = desc->d;
                auto whole = [](double v, long lo) { return v >= (double)lo && v <= 1e9 && v == (double)(long)v; };
                bool ok = true;
                switch (desc->kind) {
                case MCMCHIP_T_MVNORM: ok = np_ == dd + dd * dd; break;
                case MCMCHIP_T_NORMMIX: ok = whole(hp[0], 1) && np_ == 1 + (long)hp[0] * (2 + dd); break;
                case MCMCHIP_T_TEMPERED_NORMMIX:
                    ok = whole(hp[0], 2) && np_ > 1 + (long)hp[0] && whole(hp[1 + (long)hp[0]], 1) &&
                         np_ == 1 + (long)hp[0] + 1 + (long)hp[1 + (long)hp[0]] * (2 + dd);
                    break;
                case MCMCHIP_T_WITCH: ok = np_ >= 2; break;
                default: break;       // SIMPLEX / STUDENT_T: one value (np_ >= 1 holds here); logistic kinds checked below
                }
                if (!ok) { rc = set_error(MCMCHIP_EINVAL, "descriptor kind %d: parameter block of %ld values does not match its layout", desc->kind, np_); break; }
            }
            if (desc->kind == MCMCHIP_T_WITCH) {
                // the witch's hat takes only two values per component (tests/temp-par-witch.R:24-40): evaluate
                // them once, on the host, with the same libm expressions as the reference's closure, and ship
                // them behind the bounds: par = bnd[k], inside[k], corner[k]
                const int64_t k = desc->npar;
                const int p = desc->d;
                for (int64_t i = 0; i < k; i++) {
                    const double bnd = t->host_par[(size_t)i];
                    if (!(bnd > 0.0 && bnd < 1.0)) { rc = set_error(MCMCHIP_EINVAL, "witch.which[%d] must be in (0, 1)", (int)i + 1); break; }
                }
                if (rc) break;
                for (int64_t i = 0; i < k; i++) t->host_par.push_back(-(p + 1) * log(2) - log1p(-pow(1 - t->host_par[(size_t)i], p)));
                for (int64_t i = 0; i < k; i++) t->host_par.push_back(-(p + 1) * log(2) - p * log(1 - t->host_par[(size_t)i]));
            }
            if (hipMalloc((void **)&t->par, t->host_par.size() * 8) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "hipMalloc failed"); break; }
            if (hipMemcpyAsync(t->par, t->host_par.data(), t->host_par.size() * 8, hipMemcpyHostToDevice, dc->stream) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "copy of descriptor parameters failed"); break; }
        } else if ((!is_logistic(desc->kind) || desc->kind == MCMCHIP_T_LOGISTIC_MODELS || desc->kind == MCMCHIP_T_BINOMIAL_LOGIT) && desc->kind != MCMCHIP_T_BALL) {
            rc = set_error(MCMCHIP_EINVAL, "descriptor kind %d needs a parameter block", desc->kind);
            break;
        }
        if (desc->kind == MCMCHIP_T_LOGISTIC_MODELS) {
            const long k = (long)t->host_par[0];
            if (k < 2 || desc->npar != 1 + k * desc->d + k + 1) { rc = set_error(MCMCHIP_EINVAL, "logistic-models descriptor: par must be k, k x d masks, k log pseudo-priors, constant"); break; }
        }
        if (desc->kind == MCMCHIP_T_BINOMIAL_LOGIT && desc->npar != desc->n) { rc = set_error(MCMCHIP_EINVAL, "binomial descriptor needs one total per row of x"); break; }
        if (desc->kind == MCMCHIP_T_BINOMIAL_LOGIT && desc->d <= MCMCHIP_MFMA_MAXD) {
            rc = build_mfma_tiles(t, dc->stream);      // row weights = the totals (par, in HBM by now)
            if (rc) break;
        }
        if (desc->kind == MCMCHIP_T_STUDENT_T && !(t->host_par[0] > 0)) { rc = set_error(MCMCHIP_EINVAL, "Student t descriptor needs df > 0"); break; }
        if (desc->morph_flags) {
            // morph() argument rules (R/morph.R:51,75-76)
            if (desc->morph_flags & ~(MCMCHIP_MORPH_EXPO | MCMCHIP_MORPH_SUBEXPO | MCMCHIP_MORPH_CENTER)) { rc = set_error(MCMCHIP_EINVAL, "unknown morph flags %d", desc->morph_flags); break; }
            if (is_temper_kind(desc->kind)) { rc = set_error(MCMCHIP_EINVAL, "morph applies to metrop targets only"); break; }
            if ((desc->morph_flags & MCMCHIP_MORPH_SUBEXPO) && !(desc->morph_b > 0)) { rc = set_error(MCMCHIP_EINVAL, "b > 0 is not TRUE"); break; }
            if ((desc->morph_flags & MCMCHIP_MORPH_EXPO) && !(desc->morph_p > 2)) { rc = set_error(MCMCHIP_EINVAL, "p > 2 is not TRUE"); break; }
            if ((desc->morph_flags & MCMCHIP_MORPH_EXPO) && !(desc->morph_r >= 0)) { rc = set_error(MCMCHIP_EINVAL, "r >= 0 is not TRUE"); break; }
            t->morph_flags = desc->morph_flags;
            t->morph_b = desc->morph_b; t->morph_r = desc->morph_r; t->morph_p = desc->morph_p;
            if (desc->morph_flags & MCMCHIP_MORPH_CENTER) {
                if (!desc->morph_center) { rc = set_error(MCMCHIP_EINVAL, "morph center flagged but null"); break; }
                if (hipMalloc((void **)&t->morph_center, (size_t)desc->d * 8) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "hipMalloc failed"); break; }
                if (hipMemcpyAsync(t->morph_center, desc->morph_center, (size_t)desc->d * 8, kind, dc->stream) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "copy of the morph centre failed"); break; }
            }
        }
        if (hipStreamSynchronize(dc->stream) != hipSuccess) { rc = set_error(MCMCHIP_EHIP, "stream synchronize failed"); break; }
    } while (0);
    } catch (const std::bad_alloc &) { rc = set_error(MCMCHIP_EHIP, "out of host memory"); }
    if (rc) { mcmchip_target_destroy(t); return rc; }
    *out = t;
    return MCMCHIP_OK;
}

void mcmchip_target_destroy(mcmchip_target *t)
{
    if (!t) return;
    for (mcmchip_target *r : t->replicas) mcmchip_target_destroy(r);      // hipFree takes a pointer of any device
    if (t->x) (void)hipFree(t->x);
    if (t->y) (void)hipFree(t->y);
    if (t->par) (void)hipFree(t->par);
    if (t->xt) (void)hipFree(t->xt);
    if (t->colsum) (void)hipFree(t->colsum);
    if (t->wt) (void)hipFree(t->wt);
    if (t->morph_center) (void)hipFree(t->morph_center);
    delete t;
}

static int check_flag(DeviceCtx *dc, bool temper)
{
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, dc->d_flag, sizeof flag, hipMemcpyDeviceToHost, dc->stream));
    HIP_TRY(hipStreamSynchronize(dc->stream));
    if (!flag) return MCMCHIP_OK;
    if (flag & FLAG_NEGINF_INIT) return set_error(MCMCHIP_EDOMAIN, "log unnormalized density -Inf at initial state");
    if (flag & FLAG_POSINF) return set_error(MCMCHIP_EDOMAIN, temper ? "log unnormalized density function returned +Inf" : "logh: func returned +Inf");
    if (flag & FLAG_NAN) return set_error(MCMCHIP_EDOMAIN, temper ? "log unnormalized density function returned NA or NaN" : "logh: func returned NA or NaN");
    return set_error(MCMCHIP_EDOMAIN, "Can't happen: log hastings ratio +Inf or NaN\n");
}

// AUTO rule between the two metrop paths, from tools/path_crossover.py (profiles/r1_path_crossover.txt): the MFMA path
// pays two launches and a 130 KB table load per workgroup and iteration, so it needs enough contraction work
// (n x d) or enough chains; below that the persistent stream kernel is faster.
// a descriptor lives in the HBM of the device that was current when it was created
static int check_target_device(const mcmchip_target *t)
{
    if (t->device != cur_device_index())
        return set_error(MCMCHIP_EINVAL, "descriptor was created on device %d, the current device is %d", t->device, cur_device_index());
    return MCMCHIP_OK;
}

static bool mfma_pays(const mcmchip_target *t, long nchain)
{
    if (nchain < 64 || t->n < 1024) return false;
    return (long)t->n * t->d >= 65536 || (t->n >= 4096 && nchain >= 4096);
}

int mcmchip_logdens(const mcmchip_target *t, const double *states, int64_t nchain, int path,
                    int device_ptrs, double *out)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!t || !states || !out || nchain <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_logdens");
    if (int rc = check_target_device(t)) return rc;
    int w = is_temper_kind(t->kind) ? t->d + 1 : t->d;
    InBuf in;
    OutBuf ob;
    HIP_TRY(in.set(states, (size_t)nchain * w * 8, device_ptrs, dc->stream));
    HIP_TRY(ob.set(out, (size_t)nchain * 8, device_ptrs));
    bool mfma = (path == MCMCHIP_PATH_MFMA) || (path == MCMCHIP_PATH_AUTO && t->xt && mfma_pays(t, nchain));
    int rc;
    if (mfma) rc = launch_logdens_mfma(t, in.as<double>(), nchain, ob.as<double>(), dc);
    else rc = launch_logdens_stream(t, in.as<double>(), nchain, ob.as<double>(), dc->stream);
    if (rc) return rc;
    HIP_TRY(ob.finish(dc->stream));
    HIP_TRY(hipStreamSynchronize(dc->stream));
    return MCMCHIP_OK;
}

// No counterpart in the reference: the MFMA log density for callers that cannot allocate.  Device pointers in and
// out, scratch in the per-device static array (metrop_mfma.hip), no hipMalloc / hipFree / event creation on this path.
int mcmchip_logdens_noalloc(const mcmchip_target *t, const double *states, int64_t nchain, double *out)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!t || !states || !out || nchain <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_logdens_noalloc");
    if (int rc = check_target_device(t)) return rc;
    if (t->morph_flags) return set_error(MCMCHIP_EINVAL, "mcmchip_logdens_noalloc does not take a morphed descriptor (use mcmchip_logdens)");
    if (is_temper_kind(t->kind)) return set_error(MCMCHIP_EINVAL, "mcmchip_logdens_noalloc takes metrop descriptors (use mcmchip_logdens)");
    return launch_logdens_mfma_noalloc(t, states, nchain, out, dc);
}

// ---------------------------------------------------------------------------------------------
int mcmchip_metrop(const mcmchip_target *t, const mcmchip_metrop_args *a, mcmchip_metrop_out *o)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!t || !a || !o) return set_error(MCMCHIP_EINVAL, "null argument");
    if (is_temper_kind(t->kind)) return set_error(MCMCHIP_EINVAL, "descriptor kind %d is a tempering target", t->kind);
    if (int rc = check_target_device(t)) return rc;
    // argument checks in the reference's order and words (metrop.c:40-74)
    if (!a->initial) return set_error(MCMCHIP_EINVAL, "argument \"initial\" must be numeric");
    if (!a->scale) return set_error(MCMCHIP_EINVAL, "argument \"scale\" must be numeric");
    if (a->nbatch <= 0) return set_error(MCMCHIP_EINVAL, "argument \"nbatch\" must be positive");
    if (a->blen <= 0) return set_error(MCMCHIP_EINVAL, "argument \"blen\" must be positive");
    if (a->nspac <= 0) return set_error(MCMCHIP_EINVAL, "argument \"nspac\" must be positive");
    if (a->nchain <= 0) return set_error(MCMCHIP_EINVAL, "nchain must be positive");
    const int d = t->d;
    const long nchain = a->nchain;
    const long niter = (long)a->nbatch * a->blen * a->nspac;
    size_t slen;
    switch (a->scale_kind) {
    case MCMCHIP_SCALE_CONSTANT: slen = 1; break;
    case MCMCHIP_SCALE_DIAGONAL: slen = d; break;
    case MCMCHIP_SCALE_FULL: slen = (size_t)d * d; break;
    default: return set_error(MCMCHIP_EINVAL, "length of \"scale\" vector not d or 1");
    }
    if (!a->device_ptrs) {
        for (size_t i = 0; i < (size_t)nchain * d; i++)
            if (!isfinite(a->initial[i])) return set_error(MCMCHIP_EINVAL, "all elements of \"state\" must be finite");
    } else {
        int bits = 0;
        if (int rc = device_validate(dc, a->initial, nchain * d, 1, 0, &bits)) return rc;
        if (bits & VAL_NONFINITE) return set_error(MCMCHIP_EINVAL, "all elements of \"state\" must be finite");
    }
    // output functional -> index map (metrop.c:416-503 already resolved to a mask by the caller)
    std::vector<int> map;
    int sq_from;
    if (a->out_kind == MCMCHIP_OUT_IDENTITY) {
        for (int j = 0; j < d; j++) map.push_back(j);
        sq_from = d;
    } else if (a->out_kind == MCMCHIP_OUT_MOMENTS2) {
        for (int j = 0; j < 2 * d; j++) map.push_back(j % d);
        sq_from = d;
    } else if (a->out_kind == MCMCHIP_OUT_INDEX) {
        if (!a->out_index) return set_error(MCMCHIP_EINVAL, "outfun must be NULL, a function, a numeric vector, or a logical vector");
        for (int j = 0; j < d; j++) if (a->out_index[j]) map.push_back(j);
        sq_from = (int)map.size();
    } else {
        return set_error(MCMCHIP_EINVAL, "outfun must be NULL, a function, a numeric vector, or a logical vector");
    }
    const int nout = (int)map.size();
    if (nout == 0) { map.push_back(0); }
    if (a->rng_mode == MCMCHIP_RNG_INJECTED) {
        if (!a->z || !a->u) return set_error(MCMCHIP_EINVAL, "injected random streams z and u are required in parity mode");
    } else if (a->rng_mode != MCMCHIP_RNG_PHILOX) {
        return set_error(MCMCHIP_EINVAL, "unknown rng_mode %d", a->rng_mode);
    }
    if (a->debug && (!o->current || !o->proposal || !o->z || !o->log_green || !o->u || !o->debug_accept))
        return set_error(MCMCHIP_EINVAL, "debug requested but trace outputs are NULL");
    if (!o->accept || (!o->batch && nout > 0) || !o->final || !o->accept_batch) return set_error(MCMCHIP_EINVAL, "null output");

    bool mfma_ok = t->xt && !a->debug;
    bool mfma = (a->path == MCMCHIP_PATH_MFMA) || (a->path == MCMCHIP_PATH_AUTO && mfma_ok && mfma_pays(t, nchain));
    if (mfma && !mfma_ok) return set_error(MCMCHIP_EINVAL, "MFMA path needs a logistic or binomial-regression descriptor with d <= %d and debug = FALSE", MCMCHIP_MFMA_MAXD);

    hipStream_t s = dc->stream;
    const int dp = a->device_ptrs;
    InBuf init, scale, outmap, z, u;
    OutBuf accept, batch, final_, accb, dcur, dprop, dz, dlg, du, dacc;
    HIP_TRY(init.set(a->initial, (size_t)nchain * d * 8, dp, s));
    HIP_TRY(scale.set(a->scale, slen * 8, 0, s));            // scale and the index map are tiny host arrays
    HIP_TRY(outmap.set(map.data(), map.size() * sizeof(int), 0, s));
    if (a->rng_mode == MCMCHIP_RNG_INJECTED) {
        HIP_TRY(z.set(a->z, (size_t)nchain * niter * d * 8, dp, s));
        HIP_TRY(u.set(a->u, (size_t)nchain * niter * 8, dp, s));
    }
    HIP_TRY(accept.set(o->accept, (size_t)nchain * 8, dp));
    // nout == 0 is legal (an all-FALSE logical outfun: the reference returns an nbatch x 0 batch): nothing is staged or copied back
    HIP_TRY(batch.set(o->batch, (size_t)nchain * a->nbatch * nout * 8, dp));
    HIP_TRY(final_.set(o->final, (size_t)nchain * d * 8, dp));
    HIP_TRY(accb.set(o->accept_batch, (size_t)nchain * a->nbatch * 8, dp));
    if (a->debug) {
        HIP_TRY(dcur.set(o->current, (size_t)nchain * niter * d * 8, dp));
        HIP_TRY(dprop.set(o->proposal, (size_t)nchain * niter * d * 8, dp));
        HIP_TRY(dz.set(o->z, (size_t)nchain * niter * d * 8, dp));
        HIP_TRY(dlg.set(o->log_green, (size_t)nchain * niter * 8, dp));
        HIP_TRY(du.set(o->u, (size_t)nchain * niter * 8, dp));
        HIP_TRY(dacc.set(o->debug_accept, (size_t)nchain * niter * 4, dp));
    }
    HIP_TRY(hipMemsetAsync(dc->d_flag, 0, sizeof(int), s));

    MetropDev p;
    memset(&p, 0, sizeof p);
    p.kind = t->kind; p.d = d; p.n = t->n; p.x = t->x; p.y = t->y; p.par = t->par; p.npar = t->npar;
    p.prior_c = t->prior_c;
    p.morph_flags = t->morph_flags; p.morph_b = t->morph_b; p.morph_r = t->morph_r; p.morph_p = t->morph_p;
    p.morph_center = t->morph_center;
    p.sp_tables = dc->d_tables;
    { const char *e = getenv("MCMCHIP_NO_SOFTPLUS_TABLE"); p.no_table = e && e[0] == '1'; }
    p.nchain = nchain; p.nbatch = a->nbatch; p.blen = a->blen; p.nspac = a->nspac;
    p.scale_kind = a->scale_kind; p.scale = scale.as<double>();
    p.nout = nout; p.sq_from = sq_from; p.out_map = outmap.as<int>();
    p.rng_mode = a->rng_mode; p.z = z.as<double>(); p.u = u.as<double>();
    p.seed = a->seed; p.chain_offset = a->chain_offset; p.iter_offset = a->iter_offset;
    p.debug = a->debug;
    p.initial = init.as<double>();
    p.final_ = final_.as<double>(); p.accept = accept.as<double>(); p.batch = batch.as<double>();
    p.accept_batch = accb.as<double>();
    p.dbg_current = dcur.as<double>(); p.dbg_proposal = dprop.as<double>(); p.dbg_z = dz.as<double>();
    p.dbg_lg = dlg.as<double>(); p.dbg_u = du.as<double>(); p.dbg_accept = dacc.as<int>();
    p.flag = dc->d_flag;

    int rc;
    if (mfma) {
        rc = run_metrop_mfma(t, p, dc);
    } else {
        timing_reset("metrop_stream_kernel");
        rc = launch_metrop_stream(p, s, batch.host, dc->side[0], &batch.went_home);         // one or more timed launches
    }
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    rc = check_flag(dc, false);     // synchronises; device buffers are released by the destructors
    if (rc) return rc;
    HIP_TRY(accept.finish(s));
    HIP_TRY(batch.finish(s));
    HIP_TRY(final_.finish(s));
    HIP_TRY(accb.finish(s));
    if (a->debug) {
        HIP_TRY(dcur.finish(s)); HIP_TRY(dprop.finish(s)); HIP_TRY(dz.finish(s));
        HIP_TRY(dlg.finish(s)); HIP_TRY(du.finish(s)); HIP_TRY(dacc.finish(s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    o->nout = nout;
    o->iter_offset_out = a->iter_offset + (uint64_t)niter;
    o->path_used = mfma ? MCMCHIP_PATH_MFMA : MCMCHIP_PATH_STREAM;
    return MCMCHIP_OK;
}

// ---------------------------------------------------------------------------------------------
int mcmchip_temper(const mcmchip_target *t, const mcmchip_temper_args *a, mcmchip_temper_out *o)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!t || !a || !o) return set_error(MCMCHIP_EINVAL, "null argument");
    if (!is_temper_kind(t->kind)) return set_error(MCMCHIP_EINVAL, "descriptor kind %d is not a tempering target", t->kind);
    if (int rc = check_target_device(t)) return rc;
    const int k = a->ncomp, p = t->d;
    // temper.c:29-95 checks
    if (!a->neighbors) return set_error(MCMCHIP_EINVAL, "argument \"neighbors\" must be logical");
    if (k <= 1) return set_error(MCMCHIP_EINVAL, "must have at least two components");
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++)
            if ((a->neighbors[i + k * j] != 0) != (a->neighbors[j + k * i] != 0))
                return set_error(MCMCHIP_EINVAL, "argument \"neighbors\" must be symmetric matrix");
    // neighbour counts (temper.c:497-502).  A TRUE diagonal is legal input (only symmetry is checked), so a row can hold
    // up to k neighbours; a row with none makes the reference index the_neighbors[i][-1] (temper.c:699,795): refused.
    int maxnn = 0;
    for (int i = 0; i < k; i++) {
        int deg = 0;
        for (int j = 0; j < k; j++) deg += a->neighbors[i + k * j] != 0;
        if (deg == 0) return set_error(MCMCHIP_EINVAL, "component %d of \"neighbors\" has no neighbors", i + 1);
        if (deg > maxnn) maxnn = deg;
    }
    if (!a->initial) return set_error(MCMCHIP_EINVAL, "argument \"initial\" must be real");
    if (a->nens <= 0) return set_error(MCMCHIP_EINVAL, "nens must be positive");
    if (a->nbatch <= 0) return set_error(MCMCHIP_EINVAL, "argument \"nbatch\" must be positive");
    if (a->blen <= 0) return set_error(MCMCHIP_EINVAL, "argument \"blen\" must be positive");
    if (a->nspac <= 0) return set_error(MCMCHIP_EINVAL, "argument \"nspac\" must be positive");
    if (t->kind == MCMCHIP_T_WITCH && t->npar != k) return set_error(MCMCHIP_EINVAL, "witch descriptor needs one bound per component");
    if (t->kind == MCMCHIP_T_LOGISTIC_MODELS && (int)t->host_par[0] != k) return set_error(MCMCHIP_EINVAL, "logistic-models descriptor has %d components, neighbors has %d", (int)t->host_par[0], k);
    const long nens = a->nens;
    const int nstate = a->parallel ? k * p : p + 1;
    const long niter = (long)a->nbatch * a->blen * a->nspac;
    if (!a->device_ptrs) {
        for (size_t i = 0; i < (size_t)nens * nstate; i++)
            if (!isfinite(a->initial[i])) return set_error(MCMCHIP_EINVAL, "argument \"initial\" must have all elements finite");
        if (!a->parallel)
            for (long e = 0; e < nens; e++) {
                double ri = a->initial[(size_t)e * nstate];
                int i = (int)ri;
                if (i != ri) return set_error(MCMCHIP_EINVAL, "1st elem of argument \"initial\" must be integer in serial case");
                if (i <= 0 || i > k) return set_error(MCMCHIP_EINVAL, "1st elem of argument \"initial\" must be in 1, ..., k in serial case");
            }
    } else {
        int bits = 0;
        if (int rc = device_validate(dc, a->initial, nens * nstate, nstate, a->parallel ? 0 : k, &bits)) return rc;
        if (bits & VAL_NONFINITE) return set_error(MCMCHIP_EINVAL, "argument \"initial\" must have all elements finite");
        if (bits & VAL_NOTINT) return set_error(MCMCHIP_EINVAL, "1st elem of argument \"initial\" must be integer in serial case");
        if (bits & VAL_RANGE) return set_error(MCMCHIP_EINVAL, "1st elem of argument \"initial\" must be in 1, ..., k in serial case");
    }
    size_t slen1;
    switch (a->scale_kind & 7) {
    case MCMCHIP_SCALE_CONSTANT: slen1 = 1; break;
    case MCMCHIP_SCALE_DIAGONAL: slen1 = p; break;
    case MCMCHIP_SCALE_FULL: slen1 = (size_t)p * p; break;
    default: return set_error(MCMCHIP_EINVAL, "scale not matrix, scalar, or vector of length k");
    }
    size_t slen = (a->scale_kind & MCMCHIP_SCALE_LIST) ? slen1 * k : slen1;
    if (!a->scale) return set_error(MCMCHIP_EINVAL, "scale not type double");
    for (size_t i = 0; i < slen; i++)
        if (!isfinite(a->scale[i])) return set_error(MCMCHIP_EINVAL, "scale has non-finite element");
    if (a->rng_mode == MCMCHIP_RNG_INJECTED) {
        if (!a->unif || !a->norm) return set_error(MCMCHIP_EINVAL, "injected random streams unif and norm are required in parity mode");
    } else if (a->rng_mode != MCMCHIP_RNG_PHILOX) {
        return set_error(MCMCHIP_EINVAL, "unknown rng_mode %d", a->rng_mode);
    }
    if (!o->batch || !o->acceptx || !o->accepti || !o->final || (!a->parallel && !o->ibatch))
        return set_error(MCMCHIP_EINVAL, "null output");
    if (a->debug && (!o->which || !o->unif_which || !o->state || !o->log_hastings || !o->unif_hastings || !o->proposal ||
                     !o->acceptd || !o->norm || !o->unif_choose || (a->parallel && !o->coproposal)))
        return set_error(MCMCHIP_EINVAL, "debug requested but trace outputs are NULL");

    hipStream_t s = dc->stream;
    const int dp = a->device_ptrs;
    if (a->out_kind < MCMCHIP_TOUT_IDENTITY || a->out_kind > MCMCHIP_TOUT_MOMENTS2)
        return set_error(MCMCHIP_EINVAL, "argument \"outfun\" must be function or NULL");     // temper.c:68-69
    // index vector / moments over the identity output v (k*p state entries, or x): g_i = v[map[i]] (^2 from sq_from on)
    std::vector<int> tmap;
    int tsq_from = 0;
    {
        const int nid = a->parallel ? k * p : p;
        if (a->out_kind == MCMCHIP_TOUT_INDEX) {
            if (!a->out_index) return set_error(MCMCHIP_EINVAL, "outfun index vector missing");
            for (int j = 0; j < nid; j++) if (a->out_index[j]) tmap.push_back(j);
            tsq_from = (int)tmap.size();
        } else if (a->out_kind == MCMCHIP_TOUT_MOMENTS2) {
            for (int j = 0; j < 2 * nid; j++) tmap.push_back(j % nid);
            tsq_from = nid;
        }
    }
    if (a->out_kind == MCMCHIP_TOUT_WITCH_CORNER && t->kind != MCMCHIP_T_WITCH)
        return set_error(MCMCHIP_EINVAL, "the corner-indicator output functional belongs to the witch's hat descriptor");
    if (a->path != MCMCHIP_PATH_AUTO && a->path != MCMCHIP_PATH_LDS && a->path != MCMCHIP_PATH_RUN && a->path != MCMCHIP_PATH_TMFMA)
        return set_error(MCMCHIP_EINVAL, "unknown temper path %d", a->path);
    const bool tmapped = a->out_kind == MCMCHIP_TOUT_INDEX || a->out_kind == MCMCHIP_TOUT_MOMENTS2;
    const int nout = tmapped ? (int)tmap.size()
                             : (a->out_kind == MCMCHIP_TOUT_WITCH_CORNER ? (a->parallel ? k : 2 * k) : (a->parallel ? k * p : p));
    InBuf init, scale, nbrs, unif, norm, tmapbuf;
    if (tmapped && !tmap.empty()) HIP_TRY(tmapbuf.set(tmap.data(), tmap.size() * sizeof(int), 0, s));
    OutBuf batch, ibatch, acceptx, accepti, final_;
    HIP_TRY(init.set(a->initial, (size_t)nens * nstate * 8, dp, s));
    HIP_TRY(scale.set(a->scale, slen * 8, 0, s));
    HIP_TRY(nbrs.set(a->neighbors, (size_t)k * k * sizeof(int), 0, s));
    if (a->rng_mode == MCMCHIP_RNG_INJECTED) {
        HIP_TRY(unif.set(a->unif, (size_t)nens * niter * 4 * 8, dp, s));
        HIP_TRY(norm.set(a->norm, (size_t)nens * niter * p * 8, dp, s));
    }
    HIP_TRY(batch.set(o->batch, (size_t)nens * a->nbatch * nout * 8, dp));
    if (!a->parallel) HIP_TRY(ibatch.set(o->ibatch, (size_t)nens * a->nbatch * k * 8, dp));
    HIP_TRY(acceptx.set(o->acceptx, (size_t)nens * k * 8, dp));
    HIP_TRY(accepti.set(o->accepti, (size_t)nens * k * k * 8, dp));
    HIP_TRY(final_.set(o->final, (size_t)nens * nstate * 8, dp));
    OutBuf dwhich, duw, dstate, dlh, duh, dprop, dacc, dnorm, duc, dcop;
    if (a->debug) {
        const size_t T = (size_t)nens * niter;
        HIP_TRY(dwhich.set(o->which, T * 4, dp));
        HIP_TRY(duw.set(o->unif_which, T * 8, dp));
        HIP_TRY(dstate.set(o->state, T * nstate * 8, dp));
        HIP_TRY(dlh.set(o->log_hastings, T * 8, dp));
        HIP_TRY(duh.set(o->unif_hastings, T * 8, dp));
        HIP_TRY(dprop.set(o->proposal, T * (p + 1) * 8, dp));
        HIP_TRY(dacc.set(o->acceptd, T * 4, dp));
        HIP_TRY(dnorm.set(o->norm, T * p * 8, dp));
        HIP_TRY(duc.set(o->unif_choose, T * (a->parallel ? 2 : 1) * 8, dp));
        if (a->parallel) HIP_TRY(dcop.set(o->coproposal, T * (p + 1) * 8, dp));
    }
    HIP_TRY(hipMemsetAsync(dc->d_flag, 0, sizeof(int), s));

    TemperDev q;
    memset(&q, 0, sizeof q);
    q.kind = t->kind; q.p = p; q.par = t->par; q.npar = t->npar;
    q.x = t->x; q.y = t->y; q.n = t->n; q.prior_c = t->prior_c;
    q.nens = nens; q.k = k; q.parallel = a->parallel; q.nbatch = a->nbatch; q.blen = a->blen; q.nspac = a->nspac;
    q.neighbors = nbrs.as<int>(); q.maxnn = maxnn; q.initial = init.as<double>();
    q.scale_kind = a->scale_kind; q.scale = scale.as<double>();
    q.rng_mode = a->rng_mode; q.unif = unif.as<double>(); q.norm = norm.as<double>();
    q.seed = a->seed; q.ens_offset = a->ens_offset; q.iter_offset = a->iter_offset;
    q.batch = batch.as<double>(); q.ibatch = ibatch.as<double>(); q.acceptx = acceptx.as<double>();
    q.accepti = accepti.as<double>(); q.final_ = final_.as<double>();
    q.debug = a->debug;
    q.out_kind = a->out_kind;
    q.out_map = tmapbuf.as<int>(); q.nout_map = (int)tmap.size(); q.sq_from = tsq_from;
    q.path = a->path;
    q.dbg_which = dwhich.as<int>(); q.dbg_acceptd = dacc.as<int>(); q.dbg_unif_which = duw.as<double>();
    q.dbg_state = dstate.as<double>(); q.dbg_lh = dlh.as<double>(); q.dbg_uh = duh.as<double>();
    q.dbg_proposal = dprop.as<double>(); q.dbg_norm = dnorm.as<double>(); q.dbg_unif_choose = duc.as<double>();
    q.dbg_coproposal = dcop.as<double>();
    q.flag = dc->d_flag;
    timing_reset("temper_kernel");
    // model-selection posterior over a large data set: lockstep ensembles, densities on the matrix cores
    // (temper_mfma.hip); one wavefront per ensemble otherwise
    const bool tmfma = a->path == MCMCHIP_PATH_TMFMA ||
                       (a->path == MCMCHIP_PATH_AUTO && temper_mfma_applies(t, q) && t->n >= MCMCHIP_TMFMA_MIN_N);
    int rc = tmfma ? run_temper_mfma(t, q, dc) : launch_temper(q, s);            // one or more timed launches
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    rc = check_flag(dc, true);
    if (rc) return rc;
    HIP_TRY(batch.finish(s));
    if (!a->parallel) HIP_TRY(ibatch.finish(s));
    HIP_TRY(acceptx.finish(s));
    HIP_TRY(accepti.finish(s));
    HIP_TRY(final_.finish(s));
    if (a->debug) {
        HIP_TRY(dwhich.finish(s)); HIP_TRY(duw.finish(s)); HIP_TRY(dstate.finish(s)); HIP_TRY(dlh.finish(s));
        HIP_TRY(duh.finish(s)); HIP_TRY(dprop.finish(s)); HIP_TRY(dacc.finish(s)); HIP_TRY(dnorm.finish(s));
        HIP_TRY(duc.finish(s));
        if (a->parallel) HIP_TRY(dcop.finish(s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    o->iter_offset_out = a->iter_offset + (uint64_t)niter;
    return MCMCHIP_OK;
}

// ---------------------------------------------------------------------------------------------
int mcmchip_initseq(const double *x, int64_t len, int64_t ncol, int device_ptrs, int64_t cap,
                    int64_t *n_out, double *gamma0, double *Gamma_pos, double *Gamma_dec,
                    double *Gamma_con, double *var_pos, double *var_dec, double *var_con)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!x) return set_error(MCMCHIP_EINVAL, "argument must be numeric");
    if (len <= 0 || ncol <= 0 || cap <= 0 || !n_out || !gamma0 || !Gamma_pos || !Gamma_dec || !Gamma_con ||
        !var_pos || !var_dec || !var_con)
        return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_initseq");
    if (!device_ptrs)
        for (size_t i = 0; i < (size_t)len * ncol; i++)
            if (!isfinite(x[i])) return set_error(MCMCHIP_EINVAL, "all elements of argument must be finite");
    InBuf in;
    HIP_TRY(in.set(x, (size_t)len * ncol * 8, device_ptrs, dc->stream));
    return run_initseq(in.as<double>(), len, ncol, cap, (long *)n_out, gamma0, Gamma_pos, Gamma_dec,
                       Gamma_con, var_pos, var_dec, var_con, dc);
}

int mcmchip_initseq_uncentred(const double *x, int64_t len, int64_t ncol, int device_ptrs, int64_t cap,
                              int64_t *n_out, double *gamma0, double *Gamma_pos, double *Gamma_dec,
                              double *Gamma_con, double *var_pos, double *var_dec, double *var_con, double *mean)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!x) return set_error(MCMCHIP_EINVAL, "argument must be numeric");
    if (len <= 0 || ncol <= 0 || cap <= 0 || !n_out || !gamma0 || !Gamma_pos || !Gamma_dec || !Gamma_con ||
        !var_pos || !var_dec || !var_con)
        return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_initseq_uncentred");
    if (!device_ptrs)
        for (size_t i = 0; i < (size_t)len * ncol; i++)
            if (!isfinite(x[i])) return set_error(MCMCHIP_EINVAL, "all elements of argument must be finite");
    InBuf in;
    HIP_TRY(in.set(x, (size_t)len * ncol * 8, device_ptrs, dc->stream));
    return run_initseq(in.as<double>(), len, ncol, cap, (long *)n_out, gamma0, Gamma_pos, Gamma_dec,
                       Gamma_con, var_pos, var_dec, var_con, dc, true, mean);
}

int mcmchip_lagsums(const double *x, int64_t len, int64_t ncol, int64_t nlags, int device_ptrs, double *sums)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!x || !sums || len <= 0 || ncol <= 0 || nlags <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_lagsums");
    InBuf in;
    HIP_TRY(in.set(x, (size_t)len * ncol * 8, device_ptrs, dc->stream));
    return run_lagsums(in.as<double>(), len, ncol, nlags, sums, dc);
}

int mcmchip_series_centre(double *dev, int64_t n, int64_t p, double *mean)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!dev || n <= 0 || p <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_series_centre");
    return run_series_centre(dev, n, p, mean, dc);
}

int mcmchip_initseq_from_lagsums(const double *sums, int64_t len, int64_t ncol, int64_t nlags, int64_t cap, int64_t *n_out,
                                 double *gamma0, double *Gamma_pos, double *Gamma_dec, double *Gamma_con,
                                 double *var_pos, double *var_dec, double *var_con)
{
    clear_error();                   // host only: no device is touched
    if (!sums || len <= 0 || ncol <= 0 || nlags <= 0 || cap <= 0 || !n_out || !gamma0 || !Gamma_pos || !Gamma_dec ||
        !Gamma_con || !var_pos || !var_dec || !var_con)
        return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_initseq_from_lagsums");
    return run_initseq_from_lagsums(sums, len, ncol, nlags, cap, (long *)n_out, gamma0, Gamma_pos, Gamma_dec, Gamma_con,
                                    var_pos, var_dec, var_con);
}

int mcmchip_olbm(const double *x, int64_t n, int32_t p, int64_t len, double *mean, double *var,
                 int32_t nocalc, int device_ptrs)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!x || !mean || !var || n <= 0 || p <= 0 || len <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_olbm");
    if (len > n) return set_error(MCMCHIP_EINVAL, "len > n\n");
    InBuf in;
    HIP_TRY(in.set(x, (size_t)n * p * 8, device_ptrs, dc->stream));
    return run_olbm(in.as<double>(), n, p, len, mean, var, nocalc, dc);
}

// ---------------------------------------------------------------------------------------------
int mcmchip_chain_summary(const double *batch, int64_t nchain, int32_t nbatch, int32_t nout, int device_ptrs,
                          mcmchip_chain_summary_out *o)
{
    clear_error();
    DeviceCtx *dc = cur_dev();
    if (!dc) return MCMCHIP_EHIP;
    if (!batch || !o || nchain <= 0 || nbatch <= 0 || nout <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_chain_summary");
    const bool want_is = o->mcse_initseq != nullptr;
    if (want_is && (nchain > 65535 || (double)nchain * nout * nbatch > (double)(1L << 28)))
        return set_error(MCMCHIP_EINVAL, "mcse_initseq is for a few long chains: nchain <= 65535 and nchain * nout * nbatch <= 2^28");
    hipStream_t s = dc->stream;
    InBuf in;
    HIP_TRY(in.set(batch, (size_t)nchain * nbatch * nout * 8, device_ptrs, s));
    OutBuf cmean, cvar;
    HIP_TRY(cmean.set(o->chain_mean, (size_t)nchain * nout * 8, device_ptrs));
    HIP_TRY(cvar.set(o->chain_var, (size_t)nchain * nout * 8, device_ptrs));
    std::vector<double> stats((size_t)5 * nout);
    if (int rc = run_chain_summary(in.as<double>(), nchain, nbatch, nout, want_is, cmean.as<double>(), cvar.as<double>(), stats.data(), dc))
        return rc;
    HIP_TRY(cmean.finish(s));
    HIP_TRY(cvar.finish(s));
    HIP_TRY(hipStreamSynchronize(s));
    if (o->stats) memcpy(o->stats, stats.data(), stats.size() * 8);
    chain_summary_vectors(stats.data(), nbatch, nout, o, want_is);
    return MCMCHIP_OK;
}

int mcmchip_chain_summary_combine(const double *stats, int32_t nranks, int32_t nbatch, int32_t nout, mcmchip_chain_summary_out *o)
{
    clear_error();                   // host only: no device is touched
    if (!stats || !o || nranks <= 0 || nbatch <= 0 || nout <= 0) return set_error(MCMCHIP_EINVAL, "bad argument to mcmchip_chain_summary_combine");
    std::vector<double> total((size_t)5 * nout);
    combine_chain_stats(stats, nranks, nout, total.data());
    if (o->stats) memcpy(o->stats, total.data(), total.size() * 8);
    chain_summary_vectors(total.data(), nbatch, nout, o, true);
    return MCMCHIP_OK;
}

}  // extern "C"
