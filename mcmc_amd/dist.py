"""Multi-GPU sharding of the batched samplers: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by independent units -- chains (metrop) or ensembles (temper): rank r owns a
contiguous block, X/y/scale are replicated, and Philox streams are keyed by the GLOBAL chain id
(args.chain_offset), so every chain draws the same random numbers whatever the number of GPUs.
In temper and on metrop's MFMA path (which sums the observations in units that depend on the data alone)
the whole trajectory of a chain is then identical for any sharding.  Two things look at the chain count of
a call: the AUTO rule between the two metrop kernels and, on the stream path, the number of lanes that
share a chain's observations (its cross-lane sum rounds differently for 16 and 64 lanes); log densities
then agree to rounding (<= 1e-12 relative) and a decision could differ only at such a tie.  Callers that
need bit-identical results across shardings pin `path` to the MFMA kernel.  There is NO
per-iteration communication; the only collective is one all-gather of the batch means and
acceptance rates at the end.  The reference has no counterpart (it runs one
chain in one process).
"""
import numpy as np


def shard_range(n_units, rank, world_size):
    """contiguous block [lo, hi) of rank `rank`; blocks differ in size by at most one unit"""
    base, extra = divmod(int(n_units), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local, group=None, device=None):
    """all-gather a per-unit array (first axis = local units, possibly ragged across ranks) into
    the global array on every rank.  `local` is a numpy array; the transport is a torch tensor
    on `device` (a cuda device for RCCL, None/cpu for gloo)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(local)
    ws = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(ws)]
    dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device), group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).cpu().numpy()


def metrop_sharded(run_local, initial, group=None, device=None):
    """run_local(initial_block, chain_offset) -> result dict of mcmc_amd.metrop for this rank's
    block; returns the local result with the batch means / acceptance rates of ALL chains
    gathered under "batch.all", "accept.all", "accept.batch.all"."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    initial = np.asarray(initial)
    lo, hi = shard_range(initial.shape[0], rank, ws)
    out = run_local(initial[lo:hi], lo)
    out["shard"] = (lo, hi)
    out["batch.all"] = all_gather_rows(out["batch"], group, device)
    out["accept.all"] = all_gather_rows(np.atleast_1d(out["accept"]), group, device)
    out["accept.batch.all"] = all_gather_rows(out["accept.batch"], group, device)
    return out


def all_reduce_sum(local, group=None, device=None):
    """sum a small numpy array over the ranks (RCCL all-reduce / gloo)"""
    import torch
    import torch.distributed as dist

    a = np.ascontiguousarray(local, dtype=np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return a
    t = torch.from_numpy(a.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def temper_sharded(run_local, initial, group=None, device=None):
    """ensembles are independent (the k components of one ensemble stay on one GPU): rank r runs
    run_local(initial_block, ens_offset) -> result dict of mcmc_amd.temper; batch means and the acceptance
    tables of ALL ensembles are gathered under "<name>.all" (SURVEY.md 8(e), row temper)."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    initial = np.asarray(initial)
    lo, hi = shard_range(initial.shape[0], rank, ws)
    out = run_local(initial[lo:hi], lo)
    out["shard"] = (lo, hi)
    for key in ("batch", "acceptx", "accepti", "ibatch"):
        if key in out and out[key] is not None:
            out[key + ".all"] = all_gather_rows(out[key], group, device)
    return out


def initseq_sharded(run_local, ncol, group=None, device=None):
    """initseq by functional (column): rank r computes run_local(col_lo, col_hi) -> list of mcmc_amd.initseq results
    for its columns; the scalar outputs of all columns are gathered as arrays, the Gamma sequences as NaN-padded
    matrices with their lengths (SURVEY.md 8(e), row initseq: no communication but this gather)."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(ncol, rank, ws)
    res = run_local(lo, hi) if hi > lo else []
    out = {"shard": (lo, hi), "local": res}
    for key in ("gamma0", "var.pos", "var.dec", "var.con"):
        out[key] = all_gather_rows(np.array([r[key] for r in res], dtype=np.float64).reshape(-1), group, device)
    lens = all_gather_rows(np.array([len(r["Gamma.pos"]) for r in res], dtype=np.float64).reshape(-1), group, device)
    out["length"] = lens.astype(np.int64)
    width = int(all_reduce_max(max([len(r["Gamma.pos"]) for r in res], default=0), group, device))
    for key in ("Gamma.pos", "Gamma.dec", "Gamma.con"):
        pad = np.full((len(res), width), np.nan)
        for i, r in enumerate(res):
            pad[i, : len(r[key])] = r[key]
        out[key] = all_gather_rows(pad, group, device)
    return out


def all_reduce_max(value, group=None, device=None):
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def olbm_time_shard(n, batch_length, rank, world_size):
    """rows [row_lo, row_hi) of the n x p series that rank `rank` needs for its share of the n - len + 1 windows
    (its windows plus a halo of len - 1 rows), and how many of them it owns for the column means"""
    nwin = int(n) - int(batch_length) + 1
    w_lo, w_hi = shard_range(nwin, rank, world_size)
    row_hi = w_hi + int(batch_length) - 1 if w_hi > w_lo else w_lo
    own = (int(n) - w_lo) if rank == world_size - 1 else (w_hi - w_lo)
    return w_lo, row_hi, own


def _is_device_tensor(x):
    return hasattr(x, "is_cuda") and x.is_cuda


def olbm_sharded(olbm_local, x_local, n, batch_length, group=None, device=None):
    """olbm by time (SURVEY.md 8(e), row olbm): every rank holds rows olbm_time_shard(...) of the series.
    Two exchange steps, both tiny: an all-reduce of the column sums (p doubles) for the global means, and an
    all-reduce of the un-normalised p x p sum of outer products.  olbm_local(x, len, mean) must return the
    `nocalc` result of mcmc_amd.olbm's entry point for that block (var normalised by the block's own
    nbatch * n * len, olbm.c:28,74), which is undone here before the reduction.
    x_local: numpy rows x p, or -- the device route, nothing through host memory but p and p x p doubles -- a
    contiguous float64 CUDA tensor of shape (p, rows) (column-major); olbm_local then receives that tensor
    (mcmc_amd.api.olbm_device takes it)."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi, own = olbm_time_shard(n, batch_length, rank, ws)
    on_device = _is_device_tensor(x_local)
    if not on_device:
        x_local = np.asarray(x_local, dtype=np.float64)
    rows, p = (x_local.shape[1], x_local.shape[0]) if on_device else x_local.shape
    if rows != hi - lo:
        raise ValueError("rank %d must hold rows [%d, %d) of the series" % (rank, lo, hi))
    colsum = x_local[:, :own].sum(dim=1).cpu().numpy() if on_device else x_local[:own].sum(axis=0)
    mean = all_reduce_sum(colsum, group, device) / float(n)
    n_loc = hi - lo
    nwin_loc = n_loc - int(batch_length) + 1
    if nwin_loc > 0:
        var_loc = np.asarray(olbm_local(x_local, int(batch_length), mean))
        part = var_loc * (float(nwin_loc) * float(n_loc) * float(batch_length))
    else:
        part = np.zeros((p, p))
    total = all_reduce_sum(part, group, device)
    nwin = int(n) - int(batch_length) + 1
    return total / (float(nwin) * float(n) * float(batch_length)), mean


def chain_summary_sharded(summary_local, combine, batch_local, group=None, device=None, **kw):
    """Cross-chain diagnostics of a run sharded by chains (metrop row; north_star: RCCL "for the final
    all-gather of batch means and cross-chain diagnostics"): every rank reduces ITS chains on its GPU
    (summary_local = mcmc_amd.api.chain_summary: count, mean, M2, sum of within-chain variances, sum of var.con per
    functional), one all-gather of those 5 nout doubles per rank, and every rank merges them in rank order
    (combine = mcmc_amd.api.chain_summary_combine, host only): the same bits on every rank, whatever the collective's
    internal order.  Returns the summary of ALL chains plus "local" (this rank's own)."""
    local = summary_local(batch_local, **kw)
    stats = all_gather_rows(np.asarray(local["stats"])[None], group, device)         # nranks x 5 x nout
    out = combine(stats, local["nbatch"])
    out["local"] = local
    return out


def initseq_time_shard(n, halo, rank, world_size):
    """rows [row_lo, row_hi) of the series that rank `rank` holds for a time-sharded initseq: its own block of base
    positions plus a halo of `halo` rows (the largest lag it can serve); returns (row_lo, row_hi, own)"""
    lo, hi = shard_range(int(n), rank, world_size)
    return lo, min(hi + int(halo), int(n)), hi - lo


def initseq_time_sharded(lagsums, finish, x_local, n, halo, group=None, device=None, first_lags=64):
    """initseq by time (SURVEY.md 8(e), row initseq, second way): rank r holds rows initseq_time_shard(n, halo, r, ws) of
    the series (numpy rows x ncol, or a float64 CUDA tensor (ncol, rows) for the device route).
    Exchange: an all-reduce of the column sums (centring, R/initseq.R:4), then per round an all-reduce of the raw lag
    sums (ncol x nlags doubles).  A rank's share of a lag sum is over ITS base positions: the sums of its whole block
    minus the sums of the halo alone (the halo's base positions belong to the next rank).  Rounds quadruple the lags
    until every column has reached its stopping lag (initseq.c:42-46); a stopping lag beyond the halo is an error.
    lagsums(x, nlags) -> ncol x nlags and finish(total_sums, n) -> list of result dicts (None: not stopped yet) are
    mcmc_amd.api.lagsums / initseq_from_lagsums in production."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi, own = initseq_time_shard(n, halo, rank, ws)
    on_device = _is_device_tensor(x_local)
    if not on_device:
        x_local = np.asarray(x_local, dtype=np.float64)
    rows = x_local.shape[1] if on_device else x_local.shape[0]
    if rows != hi - lo:
        raise ValueError("rank %d must hold rows [%d, %d) of the series" % (rank, lo, hi))
    colsum = x_local[:, :own].sum(dim=1).cpu().numpy() if on_device else x_local[:own].sum(axis=0)
    mean = all_reduce_sum(colsum, group, device) / float(n)
    if on_device:
        import torch
        xc = (x_local - torch.from_numpy(mean).to(x_local.device)[:, None]).contiguous()
        tail = xc[:, own:].contiguous()
    else:
        xc = x_local - mean
        tail = xc[own:]
    if ws > 1 and int(halo) < 2:
        raise ValueError("initseq_time_sharded: the halo must hold at least 2 rows (one lag pair)")
    # no round may ask for more lags than the halo serves: a rank's sums for lags beyond its halo would silently miss
    # the terms whose partner row lies past its block
    nl = min(int(first_lags), int(halo)) if ws > 1 else int(first_lags)
    while True:
        nl_eff = min(nl, int(n))
        part = np.asarray(lagsums(xc, nl_eff))
        if rows > own:                                   # the halo's own base positions belong to the next rank
            part = part - np.asarray(lagsums(tail, nl_eff))
        total = all_reduce_sum(part, group, device)
        res = finish(total, int(n))
        if all(r is not None for r in res):
            return res
        if nl_eff >= int(n):
            raise RuntimeError("initseq_time_sharded: no stopping lag within the series length")
        if nl >= int(halo) and ws > 1:
            raise RuntimeError("initseq_time_sharded: stopping lag beyond the halo of %d rows" % int(halo))
        nl = min(4 * nl, int(halo)) if ws > 1 else 4 * nl
