#!/usr/bin/env python3
"""bench.py -- chain-iterations/sec of batched metrop on the synthetic logistic posterior
(BASELINE.json config 3: n = 1e6 observations, d = 64, 65 536 chains, fp64, device Philox).

One "step" = one lockstep Metropolis iteration of every chain on this rank: Philox normals ->
proposal -> log density (fp64 MFMA X.Theta + softplus) -> accept/reject -> batch means.
Inputs (X tiles, y, initial states) are resident in HBM before the timed region; outputs stay
in HBM.  N > 1: one process per GPU (torchrun), chains sharded, no per-iteration communication,
one small RCCL all-gather of per-rank summaries at the end.  --scaling strong (the default for
N > 1: BASELINE config 3 is "65 536 chains sharded across 1/2/4/8 GPUs") gives every GPU
65 536 / N chains; --scaling weak keeps 65 536 chains per GPU.  At N = 1 the two coincide.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X vendor peak, fp64 matrix (= vector) -- see DESIGN.md "Roofline"


def synthetic(n, d, seed=42):
    """SURVEY.md 8(d) C3 inputs"""
    rng = np.random.default_rng(seed)
    X = np.empty((n, d))
    X[:, 0] = 1.0
    X[:, 1:] = rng.standard_normal((n, d - 1)) / np.sqrt(d)
    beta = 0.5 * rng.standard_normal(d)
    y = (rng.random(n) < 1.0 / (1.0 + np.exp(-(X @ beta)))).astype(np.float64)
    return X, y, beta


def host_cores():
    """(threads to use, cores the scheduler lets this process run on, cores of the machine, cgroup CPU quota or None)"""
    affinity = len(os.sched_getaffinity(0))
    total = os.cpu_count() or affinity
    quota = None
    try:                                              # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    use = affinity if quota is None else max(1, min(affinity, int(round(quota))))
    return use, affinity, total, quota


def cpu_baseline(X, y, beta, scale, budget_s=12.0):
    """the oracle (CPU restatement of the reference loop + the same descriptor), one metrop chain per thread on ALL
    host cores this process may use, on a bounded sample of the same workload"""
    import oracle
    from mcmc_amd import abi
    oracle.build()
    threads, affinity, total, quota = host_cores()
    tg = oracle.Target(abi.T_LOGISTIC, X.shape[1], x=X, y=y, prior_c=0.125)
    d = X.shape[1]
    init = beta + 1e-3 * np.random.default_rng(1).standard_normal((threads, d))
    t0 = time.perf_counter()
    oracle.metrop(tg, init, 1, scale=scale, seed=1, nthreads=threads)      # 2 density evaluations per chain
    per_eval = (time.perf_counter() - t0) / 2.0
    iters = int(max(2, min(1000, budget_s / max(per_eval, 1e-6) - 1)))
    t0 = time.perf_counter()
    oracle.metrop(tg, init, iters, scale=scale, seed=1, nthreads=threads)
    el = time.perf_counter() - t0
    return {"value": threads * iters / el, "unit": "chain-iterations/s", "cores": threads, "cores_total": total,
            "cores_affinity": affinity, "cpu_quota": quota, "kind": "port",
            "sample": "%d chains (one per thread, every core this process may use) x %d iterations of the same "
                      "n=%d, d=%d posterior, %.1f s; %.2f chain-iterations/s per core (all threads stream the same "
                      "512 MB X: shared memory bandwidth, not the core count, sets the rate beyond a few dozen threads)"
                      % (threads, iters, X.shape[0], d, el, iters / el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--chains", type=int, default=65536, help="chains of the whole job (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="strong: --chains split over the GPUs (default for --gpus > 1); weak: --chains per GPU")
    ap.add_argument("--scale", type=float, default=4.0e-3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (BASELINE configs 2, 4 and 5 on the device; N = 1 only)")
    ap.add_argument("--secondary", default="c2,c4,c5,temper_models", help="which secondary configs to run")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a box with ONE GPU: every rank uses cuda:0 and the collectives go through "
                         "gloo on host copies (RCCL refuses two ranks on one device); exercises the sharding, not the fabric")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    dev_index = 0 if a.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # launched by torch.distributed.run
    if use_dist:
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm

    from mcmc_amd import abi, api
    api.init(dev_index)
    L = api.load_library()

    scaling = a.scaling or ("strong" if world > 1 else "weak")
    X, y, beta = synthetic(a.n, a.d)
    tg = api.logistic_target(X, y)                      # X staged and pre-tiled in HBM once
    d = a.d
    if scaling == "strong":                             # contiguous blocks of the job's chains (mcmc_amd.dist.shard_range)
        from mcmc_amd.dist import shard_range
        chain0, chain1 = shard_range(a.chains, rank, world)
    else:
        chain0, chain1 = rank * a.chains, (rank + 1) * a.chains
    Cn = chain1 - chain0                                # global chain ids key the Philox streams: results independent of N
    total_chains = a.chains if scaling == "strong" else a.chains * world
    # initial states by GLOBAL chain id, so that the job's numbers do not depend on how it is sharded
    init_h = beta + 1e-3 * np.random.default_rng(1000).standard_normal((total_chains, d))[chain0:chain1]
    init = torch.from_numpy(np.ascontiguousarray(init_h)).to(dev)
    nb_max = max(a.steps, a.warmup, 1)
    out_accept = torch.empty(Cn, dtype=torch.float64, device=dev)
    out_batch_buf = torch.empty((Cn * nb_max * d,), dtype=torch.float64, device=dev)
    out_final = [torch.empty((Cn, d), dtype=torch.float64, device=dev) for _ in range(2)]
    out_ab = torch.empty((Cn * nb_max,), dtype=torch.float64, device=dev)
    scale = np.array([a.scale])

    def run(nbatch, initial, iter_offset, slot):
        """one C-ABI call = nbatch lockstep iterations of this rank's chains, then the path's one collective: the
        all-gather of the per-rank summaries (acceptance rate, mean of the batch means)"""
        out_batch = out_batch_buf[: Cn * nbatch * d].view(Cn, nbatch, d)
        args = abi.MetropArgs(Cn, nbatch, 1, 1, initial.data_ptr(), abi.SCALE_CONSTANT, scale.ctypes.data,
                              abi.OUT_IDENTITY, None, abi.RNG_PHILOX, None, None, 20240607, chain0, iter_offset,
                              0, 1, abi.PATH_MFMA)
        out = abi.MetropOut(out_accept.data_ptr(), out_batch.data_ptr(), out_final[slot].data_ptr(), out_ab.data_ptr(),
                            None, None, None, None, None, None, 0, 0, 0)
        torch.cuda.synchronize()
        rc = L.mcmchip_metrop(tg.handle, C.addressof(args), C.addressof(out))
        if rc:
            raise RuntimeError(L.mcmchip_last_error().decode())
        summ = torch.cat([out_accept.sum().reshape(1), out_batch.sum(dim=(0, 1)) / nbatch])     # sums over local chains
        if use_dist:
            src = summ.cpu() if a.rehearse_on_one_gpu else summ
            allsum = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(allsum, src)
            summ = torch.stack(allsum).sum(dim=0).to(dev)
        return summ / total_chains, out_final[slot], int(out.iter_offset_out)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:                                        # create the RCCL communicator outside the timed region
        cdev = "cpu" if a.rehearse_on_one_gpu else dev
        warm = [torch.empty(1 + d, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(warm, torch.zeros(1 + d, dtype=torch.float64, device=cdev))
    # warmup (untimed): the same call path as the timed region, collective and summary kernels included
    state, off = init, 0
    if a.warmup > 0:
        _, state, off = run(a.warmup, init, 0, 0)

    barrier()
    t0 = time.perf_counter()
    summ, state, off = run(a.steps, state, off, 1)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if a.rehearse_on_one_gpu else dev))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kname, kms, klaunch = api.last_kernel_time()
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the recorded
    # rocprofv3 measurement of the same workload is reported -- only while it is a measurement of THIS kernel: the record
    # carries a hash of the kernel's sources (tools/kernel_source_hash.py) and a change of theirs turns it into null
    traffic, traffic_src = None, None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_source_hash import logdens_kernel_hash
        for name in ("r3_pmc_traffic.json", "r2_pmc_traffic.json"):
            f = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(f):
                continue
            rec = json.load(open(f))
            if rec["workload"] == {"n": a.n, "d": a.d, "chains_per_gpu": Cn} and rec.get("kernel_source_sha256") == logdens_kernel_hash(ROOT):
                traffic, traffic_src = rec["hbm_bytes_per_launch"], "profiles/" + name
                break
    except Exception:
        pass
    if rank == 0:
        flops_per_launch = 2.0 * a.n * a.d * Cn
        achieved = flops_per_launch / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        res = {
            "metric": "chain-iterations/sec per node, batched metrop on logistic posterior",
            "value": total_chains * a.steps / elapsed,
            "unit": "chain-iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3: synthetic logistic posterior n=%d d=%d, %d chains in all (%d on rank 0), "
                                   "scalar scale %.2g, identity outfun, device Philox, blen=nspac=1"
                                   % (a.n, a.d, total_chains, Cn, a.scale),
                       "chains_total": total_chains, "chains_per_gpu": Cn, "n": a.n, "d": a.d,
                       "parallelism": "chains sharded x%d (%s scaling)" % (world, scaling),
                       "accept_rate": float(summ[0].item()),
                       "note": "timed call = %d iterations + the call's initial-state density evaluation" % a.steps},
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of this workload and these "
                                         "kernel sources: %s)" % (traffic_src or "no record matches: null"),
                         "launch_ms": kms, "launches_timed": klaunch,
                         "flops_per_launch": flops_per_launch},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(X, y, beta, a.scale)
            except Exception as e:          # the baseline is a reported extra, never the product path
                res["cpu_baseline"] = {"value": None, "error": str(e)}
        if world == 1 and not a.no_secondary:
            # BASELINE configs 2, 4 and 5 beside the headline (config 3); device path only, after the timed region
            del out_batch_buf, out_ab
            torch.cuda.empty_cache()
            import bench_secondary
            res["secondary"] = bench_secondary.run(api, torch, dev, tuple(w for w in a.secondary.split(",") if w))
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
