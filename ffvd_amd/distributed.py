"""Multi-GPU data parallelism for the ELBO: one process per GPU.

The nll is a sum of per-(chain, latent-dim) terms plus cheap shared terms, so it shards with NO data-path
collective; the only exchange is one all-reduce(sum) of the 8-double partial-sum vector of ffvd_abi.h
(`torch.distributed`, backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).

  mode "chains": rank r evaluates chains [s_begin, s_begin + s_count) for all latent dims (BASELINE configs 2-4)
  mode "dims"  : rank r evaluates latent dims [d_begin, d_begin + d_count) for all chains; only rank 0 adds
                 the shared terms (likelihood, prior_Z, prior_x_0, hyper prior)          (BASELINE config 5)
"""
from __future__ import annotations

import numpy as np


def shard_range(n, world, rank):
    """Contiguous balanced split of range(n): returns (begin, count); the first n % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(int(n), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def plan(meta, world, rank, mode="chains"):
    """Engine keyword arguments + the slice of X this rank owns."""
    S, D = meta["S"], meta["D"]
    if mode == "chains":
        if S < world:
            raise ValueError(f"cannot shard {S} chains over {world} ranks; use mode='dims'")
        s_begin, s_count = shard_range(S, world, rank)
        return dict(s_begin=s_begin, s_count=s_count, d_begin=0, d_count=D, shared_terms=True)
    if mode == "dims":
        if D < world:
            raise ValueError(f"cannot shard {D} latent dims over {world} ranks")
        d_begin, d_count = shard_range(D, world, rank)
        return dict(s_begin=0, s_count=S, d_begin=d_begin, d_count=d_count, shared_terms=(rank == 0))
    raise ValueError("mode must be 'chains' or 'dims'")


def finish(sums8):
    """Mean terms from the (all-reduced) partial-sum vector: sums8[0:7] / sums8[7]."""
    sums8 = np.asarray(sums8, dtype=np.float64)
    if sums8.shape != (8,) or not sums8[7] > 0:
        raise ValueError("bad partial-sum vector")
    from ._lib import TERM_NAMES
    return {n: float(sums8[i] / sums8[7]) for i, n in enumerate(TERM_NAMES)}


def all_reduce_sums(tensor, group=None):
    """In-place all-reduce(sum) of the 8-double partial-sum tensor (device tensor under RCCL, CPU under gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


class ShardedElbo:
    """One rank's share of the ELBO on its own GPU + the scalar all-reduce.

    The 8 partial sums are written by the finalize kernel straight into a torch CUDA tensor (its
    `data_ptr()` crosses the C ABI as a plain device pointer), which RCCL then all-reduces in place."""

    def __init__(self, params, Y, control_inputs, meta, rank=0, world=1, mode="chains", device=0, **engine_kw):
        import torch
        from .engine import ElboEngine
        self.torch = torch
        self.meta, self.rank, self.world, self.mode = meta, rank, world, mode
        self.plan = plan(meta, world, rank, mode)
        pl = self.plan
        self.engine = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], pl["s_count"], Ydim=meta["Ydim"],
                                 kernel_type=meta["kernel_type"], U_collapse=meta["U_collapse"], device=device,
                                 d_begin=pl["d_begin"], d_count=pl["d_count"], shared_terms=pl["shared_terms"],
                                 **engine_kw)
        self.engine.set_data(Y, control_inputs)
        local = dict(params)
        local["X"] = np.ascontiguousarray(params["X"][pl["s_begin"]: pl["s_begin"] + pl["s_count"]])
        self.engine.set_params(local)
        self.sums = torch.zeros(8, dtype=torch.float64, device=f"cuda:{device}")

    def step(self):
        """One ELBO iteration: local kernels -> 8 partial sums in HBM -> all-reduce -> host."""
        self.engine.elbo_async(self.sums.data_ptr())
        self.engine.sync()
        all_reduce_sums(self.sums)
        return self.sums.cpu().numpy()

    def nll_terms(self):
        return finish(self.step())
