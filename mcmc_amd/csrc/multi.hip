// multi.hip -- mcmchip_metrop_multi / mcmchip_temper_multi: ONE call that uses several GPUs of the node.
//
// The reference's front ends make one native call per run (R/metrop.R:48-51, R/temper.R:66-68); an R session is one
// process, so the process-per-GPU launch of mcmc_amd/dist.py is out of its reach.  Here the chains (ensembles) of a
// call are split into contiguous blocks by shard_range's rule (blocks differ by at most one unit), one host thread per
// entry of the device list runs the ordinary entry point on its block -- descriptor replicated into that device's HBM
// once (hipMemcpyPeer) and kept with the original, Philox keyed by the GLOBAL chain id, results written straight into
// the caller's arrays at the block's offset.  No collective, no second process.  A device may be listed more than
// once (two blocks side by side on one GPU: what the one-GPU test box can check).
// The interrupt callback runs on the calling thread only (R_CheckUserInterrupt is not thread safe): it polls while the
// workers run and raises a word they look at between launches.
#include "kernels.h"
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Worker {
    DeviceCtx dc;
    TimingState timing;
};
std::mutex g_multi_mutex;                              // one *_multi call at a time
std::vector<std::unique_ptr<Worker>> g_workers;        // kept until the process ends, like the contexts of mcmchip_init
std::vector<std::pair<int, int>> g_worker_key;         // (device, slot on that device)

int get_worker(int device, int slot, Worker **out)
{
    for (size_t i = 0; i < g_workers.size(); i++)
        if (g_worker_key[i] == std::make_pair(device, slot)) { *out = g_workers[i].get(); return MCMCHIP_OK; }
    std::unique_ptr<Worker> w(new Worker());
    if (int rc = init_device_ctx(device, w->dc)) return rc;
    *out = w.get();
    g_workers.push_back(std::move(w));
    g_worker_key.push_back(std::make_pair(device, slot));
    return MCMCHIP_OK;
}

int peer_copy(void **dst, int dst_dev, const void *src, int src_dev, size_t bytes)
{
    *dst = nullptr;
    if (!src || !bytes) return MCMCHIP_OK;
    HIP_TRY(hipMalloc(dst, bytes));                                          // on the current device = dst_dev
    HIP_TRY(hipMemcpyPeer(*dst, dst_dev, src, src_dev, bytes));
    return MCMCHIP_OK;
}

// the descriptor in the HBM of `device`: the original, a replica made earlier, or a new one (kept with the original)
int target_on_device(const mcmchip_target *t, int device, const mcmchip_target **out)
{
    if (t->device == device) { *out = t; return MCMCHIP_OK; }
    for (const mcmchip_target *r : t->replicas)
        if (r->device == device) { *out = r; return MCMCHIP_OK; }
    HIP_TRY(hipSetDevice(device));
    mcmchip_target *r = new mcmchip_target();
    r->kind = t->kind; r->d = t->d; r->n = t->n; r->prior_c = t->prior_c; r->npar = t->npar; r->host_par = t->host_par;
    r->morph_flags = t->morph_flags; r->morph_b = t->morph_b; r->morph_r = t->morph_r; r->morph_p = t->morph_p;
    r->dpad = t->dpad; r->ksteps = t->ksteps; r->ntiles = t->ntiles; r->npad_rows = t->npad_rows;
    r->device = device;
    int rc = MCMCHIP_OK;
    if (!rc) rc = peer_copy((void **)&r->x, device, t->x, t->device, (size_t)t->n * t->d * 8);
    if (!rc) rc = peer_copy((void **)&r->y, device, t->y, t->device, (size_t)t->n * 8);
    if (!rc) rc = peer_copy((void **)&r->par, device, t->par, t->device, t->host_par.size() * 8);
    if (!rc) rc = peer_copy((void **)&r->xt, device, t->xt, t->device, (size_t)t->ntiles * t->ksteps * 64 * 8);
    if (!rc) rc = peer_copy((void **)&r->colsum, device, t->colsum, t->device, (size_t)t->dpad * 8);
    if (!rc) rc = peer_copy((void **)&r->wt, device, t->wt, t->device, (size_t)t->ntiles * 16 * 8);
    if (!rc) rc = peer_copy((void **)&r->morph_center, device, t->morph_center, t->device, (size_t)t->d * 8);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = set_error(MCMCHIP_EHIP, "replicating the descriptor onto device %d failed", device);
    if (rc) { mcmchip_target_destroy(r); return rc; }
    const_cast<mcmchip_target *>(t)->replicas.push_back(r);
    *out = r;
    return MCMCHIP_OK;
}

struct Block { long lo, hi; };
Block shard(long n, int rank, int ws)                 // mcmc_amd/dist.py::shard_range
{
    const long base = n / ws, extra = n % ws;
    const long lo = rank * base + (rank < extra ? rank : extra);
    return Block{lo, lo + base + (rank < extra ? 1 : 0)};
}

struct Job {
    Worker *w = nullptr;
    const mcmchip_target *t = nullptr;
    Block b{0, 0};
    int rc = MCMCHIP_OK;
    std::string err;
};

// common part: contexts and replicas on the calling thread, then the worker threads and the interrupt poll
template <class Run> int run_jobs(const mcmchip_target *t, long nunits, const int32_t *devices, int ndev, Run run, std::vector<Job> &jobs)
{
    if (!devices || ndev <= 0 || ndev > 64) return set_error(MCMCHIP_EINVAL, "device list must hold 1 to 64 devices");
    int visible = 0, back = -1;
    HIP_TRY(hipGetDeviceCount(&visible));
    (void)hipGetDevice(&back);
    jobs.assign((size_t)ndev, Job());
    int rc = MCMCHIP_OK;
    for (int r = 0; r < ndev && !rc; r++) {
        const int dev = devices[r];
        if (dev < 0 || dev >= visible || dev >= MCMCHIP_MAX_DEVICES) { rc = set_error(MCMCHIP_EINVAL, "device %d in the list, %d visible", dev, visible); break; }
        int slot = 0;
        for (int q = 0; q < r; q++) slot += devices[q] == dev;
        jobs[(size_t)r].b = shard(nunits, r, ndev);
        if (jobs[(size_t)r].b.hi == jobs[(size_t)r].b.lo) continue;           // more devices than units
        rc = get_worker(dev, slot, &jobs[(size_t)r].w);
        if (!rc) rc = target_on_device(t, dev, &jobs[(size_t)r].t);
    }
    if (back >= 0) (void)hipSetDevice(back);
    if (rc) return rc;
    volatile int abort_word = 0;
    std::vector<std::thread> pool;
    std::vector<int> finished((size_t)ndev, 0);
    std::mutex done_mutex;
    int ndone = 0, nstarted = 0;
    for (int r = 0; r < ndev; r++) {
        Job &j = jobs[(size_t)r];
        if (!j.w) continue;
        nstarted++;
        pool.emplace_back([&, r]() {
            Job &me = jobs[(size_t)r];
            {
                WorkerScope scope(&me.w->dc, &me.w->timing, &abort_word);
                me.rc = run(me, r);
                if (me.rc) me.err = mcmchip_last_error();
            }
            std::lock_guard<std::mutex> lock(done_mutex);
            ndone++;
        });
    }
    LibCtx &L = lib_ctx();
    for (;;) {
        {
            std::lock_guard<std::mutex> lock(done_mutex);
            if (ndone == nstarted) break;
        }
        if (L.poll && !abort_word && L.poll(L.poll_ctx)) abort_word = 1;
        std::this_thread::sleep_for(std::chrono::microseconds(L.poll ? 2000 : 200));
    }
    for (auto &th : pool) th.join();
    for (int r = 0; r < ndev; r++)
        if (jobs[(size_t)r].w) { L.last_timing = &jobs[(size_t)r].w->timing; break; }
    for (int r = 0; r < ndev; r++)
        if (jobs[(size_t)r].rc) return set_error(jobs[(size_t)r].rc, "%s", jobs[(size_t)r].err.c_str());
    return MCMCHIP_OK;
}

template <class T> T *at(T *p, long off) { return p ? p + off : nullptr; }
}  // namespace

extern "C" {

int mcmchip_metrop_multi(const mcmchip_target *t, const mcmchip_metrop_args *a, mcmchip_metrop_out *o, const int32_t *devices,
                         int32_t ndev)
{
    clear_error();
    if (!t || !a || !o) return set_error(MCMCHIP_EINVAL, "null argument");
    if (a->nchain <= 0) return set_error(MCMCHIP_EINVAL, "nchain must be positive");
    if (a->nbatch <= 0 || a->blen <= 0 || a->nspac <= 0) return set_error(MCMCHIP_EINVAL, "argument \"nbatch\" must be positive");
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    const long d = t->d, niter = (long)a->nbatch * a->blen * a->nspac, nbatch = a->nbatch;
    long nout = d;                                        // dimension of g(x): what mcmchip_metrop will report
    if (a->out_kind == MCMCHIP_OUT_MOMENTS2) nout = 2 * d;
    else if (a->out_kind == MCMCHIP_OUT_INDEX) {
        if (!a->out_index) return set_error(MCMCHIP_EINVAL, "outfun must be NULL, a function, a numeric vector, or a logical vector");
        nout = 0;
        for (long j = 0; j < d; j++) nout += a->out_index[j] != 0;
    }
    // AUTO looks at the chain count: decide on the whole call, so that a split run takes the kernel the unsplit one takes
    int path = a->path;
    if (path == MCMCHIP_PATH_AUTO) {
        const bool mfma_ok = t->xt && !a->debug;
        const bool pays = a->nchain >= 64 && t->n >= 1024 && ((long)t->n * t->d >= 65536 || (t->n >= 4096 && a->nchain >= 4096));
        path = mfma_ok && pays ? MCMCHIP_PATH_MFMA : MCMCHIP_PATH_STREAM;
    }
    std::vector<Job> jobs;
    std::vector<mcmchip_metrop_out> outs((size_t)(ndev > 0 ? ndev : 1));
    int rc = run_jobs(t, a->nchain, devices, ndev, [&](Job &j, int r) {
        const long lo = j.b.lo;
        mcmchip_metrop_args sa = *a;
        sa.nchain = j.b.hi - lo;
        sa.initial = at(a->initial, lo * d);
        sa.z = at(a->z, lo * niter * d);
        sa.u = at(a->u, lo * niter);
        sa.chain_offset = a->chain_offset + (uint64_t)lo;
        sa.path = path;
        mcmchip_metrop_out so = *o;
        so.accept = at(o->accept, lo);
        so.batch = at(o->batch, lo * nbatch * nout);
        so.final = at(o->final, lo * d);
        so.accept_batch = at(o->accept_batch, lo * nbatch);
        so.current = at(o->current, lo * niter * d);
        so.proposal = at(o->proposal, lo * niter * d);
        so.z = at(o->z, lo * niter * d);
        so.log_green = at(o->log_green, lo * niter);
        so.u = at(o->u, lo * niter);
        so.debug_accept = at(o->debug_accept, lo * niter);
        const int wrc = mcmchip_metrop(j.t, &sa, &so);
        outs[(size_t)r] = so;
        return wrc;
    }, jobs);
    if (rc) return rc;
    for (int r = 0; r < ndev; r++)
        if (jobs[(size_t)r].w) {
            o->nout = outs[(size_t)r].nout;
            o->path_used = outs[(size_t)r].path_used;
            o->iter_offset_out = outs[(size_t)r].iter_offset_out;
            break;
        }
    return MCMCHIP_OK;
}

int mcmchip_temper_multi(const mcmchip_target *t, const mcmchip_temper_args *a, mcmchip_temper_out *o, const int32_t *devices,
                         int32_t ndev)
{
    clear_error();
    if (!t || !a || !o) return set_error(MCMCHIP_EINVAL, "null argument");
    if (a->nens <= 0) return set_error(MCMCHIP_EINVAL, "nens must be positive");
    if (a->nbatch <= 0 || a->blen <= 0 || a->nspac <= 0 || a->ncomp < 2) return set_error(MCMCHIP_EINVAL, "bad run shape");
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    const long p = t->d, k = a->ncomp, niter = (long)a->nbatch * a->blen * a->nspac, nbatch = a->nbatch;
    const long nstate = a->parallel ? k * p : 1 + p;
    const long nident = a->parallel ? k * p : p;          // the identity output (temper.c:851-864)
    long nout = nident;
    if (a->out_kind == MCMCHIP_TOUT_WITCH_CORNER) nout = a->parallel ? k : 2 * k;
    else if (a->out_kind == MCMCHIP_TOUT_MOMENTS2) nout = 2 * nident;
    else if (a->out_kind == MCMCHIP_TOUT_INDEX) {
        if (!a->out_index) return set_error(MCMCHIP_EINVAL, "out_index is required for the index output");
        nout = 0;
        for (long j = 0; j < nident; j++) nout += a->out_index[j] != 0;
    }
    std::vector<Job> jobs;
    std::vector<mcmchip_temper_out> outs((size_t)(ndev > 0 ? ndev : 1));
    int rc = run_jobs(t, a->nens, devices, ndev, [&](Job &j, int r) {
        const long lo = j.b.lo;
        mcmchip_temper_args sa = *a;
        sa.nens = j.b.hi - lo;
        sa.initial = at(a->initial, lo * nstate);
        sa.unif = at(a->unif, lo * niter * 4);
        sa.norm = at(a->norm, lo * niter * p);
        sa.ens_offset = a->ens_offset + (uint64_t)lo;
        mcmchip_temper_out so = *o;
        so.batch = at(o->batch, lo * nbatch * nout);
        so.ibatch = at(o->ibatch, lo * nbatch * k);
        so.acceptx = at(o->acceptx, lo * k);
        so.accepti = at(o->accepti, lo * k * k);
        so.final = at(o->final, lo * nstate);
        so.which = at(o->which, lo * niter);
        so.unif_which = at(o->unif_which, lo * niter);
        so.state = at(o->state, lo * niter * nstate);
        so.log_hastings = at(o->log_hastings, lo * niter);
        so.unif_hastings = at(o->unif_hastings, lo * niter);
        so.proposal = at(o->proposal, lo * niter * (p + 1));
        so.acceptd = at(o->acceptd, lo * niter);
        so.norm = at(o->norm, lo * niter * p);
        so.unif_choose = at(o->unif_choose, lo * niter * (a->parallel ? 2 : 1));
        so.coproposal = at(o->coproposal, lo * niter * (p + 1));
        const int wrc = mcmchip_temper(j.t, &sa, &so);
        outs[(size_t)r] = so;
        return wrc;
    }, jobs);
    if (rc) return rc;
    for (int r = 0; r < ndev; r++)
        if (jobs[(size_t)r].w) { o->iter_offset_out = outs[(size_t)r].iter_offset_out; break; }
    return MCMCHIP_OK;
}

}  // extern "C"
