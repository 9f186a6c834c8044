"""
Multi-GPU sharding of the (walker x order) units.

The units of a log-likelihood batch are independent, so the path shards with NO data-path collective:
every rank (one process per GPU) evaluates a contiguous slice of the unit index on its own device and
the B results are gathered on the host.  ``torch.distributed`` is used only for that tiny host gather
(any backend; ``gloo`` in the CPU tests, ``nccl`` == RCCL on the GPU node).
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Contiguous [lo, hi) slice of ``n_units`` owned by ``rank``; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(int(n_units), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_host(local, n_units, group=None):
    """All ranks contribute their slice (numpy array, leading dim = slice length) and every rank
    receives the concatenation in unit order.  Host-side object gather: B doubles, not a hot path."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local)
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, np.asarray(local), group=group)
    out = np.concatenate([np.asarray(p) for p in parts], axis=0)
    if out.shape[0] != n_units:
        raise RuntimeError(f"gathered {out.shape[0]} units, expected {n_units}")
    return out


def sharded_batch(evaluate, P, group=None):
    """Evaluate ``evaluate(P_slice) -> array`` on this rank's slice of the rows of ``P`` and return
    the full result on every rank.  ``evaluate`` is typically ``model.log_likelihood_batch`` of a model
    whose buffers live on this rank's GPU."""
    import torch.distributed as dist

    P = np.atleast_2d(np.asarray(P, dtype=np.float64))
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(P.shape[0], rank, world)
    local = np.asarray(evaluate(P[lo:hi])) if hi > lo else np.zeros((0,))
    return gather_host(local, P.shape[0], group)


def order_major_units(n_orders, n_walkers):
    """Flattened (order, walker) unit list, order-major so per-order static data stays resident on the
    rank that owns the slice (cfg 4)."""
    o, w = np.divmod(np.arange(n_orders * n_walkers), n_walkers)
    return np.stack([o, w], axis=1)


def order_major_slices(n_orders, n_walkers, lo, hi):
    """The units [lo, hi) of the order-major list as ``[(order, w_lo, w_hi), ...]``: whole orders in the middle,
    at most one partial order at either end -- what a rank hands to ``MultiPlan`` (cfg 4: per-order static
    data is only needed on the ranks that own walkers of that order)."""
    out = []
    for o in range(int(n_orders)):
        w_lo, w_hi = max(lo, o * n_walkers) - o * n_walkers, min(hi, (o + 1) * n_walkers) - o * n_walkers
        if w_hi > w_lo:
            out.append((o, int(w_lo), int(w_hi)))
    return out
