"""Multi-GPU sharding of the batched samplers: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by independent units -- chains (metrop) or ensembles (temper): rank r owns a
contiguous block, X/y/scale are replicated, and Philox streams are keyed by the GLOBAL chain id
(args.chain_offset), so the numbers do not depend on the number of GPUs.  There is NO
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
