"""Multi-GPU data parallelism for the ELBO: one process per GPU.

The nll is a sum of per-(chain, latent-dim) terms plus cheap shared terms, so it shards with NO data-path
collective; the only exchange is one all-reduce(sum) of the 8-double partial-sum vector of ffvd_abi.h
(`torch.distributed`, backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).

  mode "chains": rank r evaluates chains [s_begin, s_begin + s_count) for all latent dims (BASELINE configs 2-4)
  mode "dims"  : rank r evaluates latent dims [d_begin, d_begin + d_count) for all chains; only rank 0 adds
                 the shared terms (likelihood, prior_Z, prior_x_0, hyper prior)          (BASELINE config 5)
"""
from __future__ import annotations

import os

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


GRAD_KEYS = ("Z", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


def all_reduce_grads(grads, mode="chains", device=None, group=None):
    """Sum the per-rank gradient dicts of `ElboEngine.nll_and_grad(S_total=...)` over the ranks.

    The shared parameters (GRAD_KEYS) are packed into ONE flat fp64 buffer and all-reduced in a single call
    (a few KB: latency-bound, so one collective, not seven).  X gradients: with mode "chains" every rank owns
    its chains' rows and nothing is exchanged; with mode "dims" every rank holds a partial sum over its latent
    dims for all chains, so dX joins the same buffer.  Returns a new dict."""
    import torch
    import torch.distributed as dist
    keys = list(GRAD_KEYS) + (["U"] if "U" in grads else []) + (["X"] if mode == "dims" else [])
    flat = np.concatenate([np.asarray(grads[k], dtype=np.float64).ravel() for k in keys])
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    flat = t.cpu().numpy()
    out, off = dict(grads), 0
    for k in keys:
        n = int(np.asarray(grads[k]).size)
        out[k] = flat[off: off + n].reshape(np.asarray(grads[k]).shape).copy()
        off += n
    return out


class ShardedElbo:
    """One rank's share of the ELBO on its own GPU + the scalar all-reduce.

    The 8 partial sums are written by the finalize kernel straight into a torch CUDA tensor (its
    `data_ptr()` crosses the C ABI as a plain device pointer), which RCCL then all-reduces in place."""

    def __init__(self, params, Y, control_inputs, meta, rank=0, world=1, mode="chains", device=0, always_reduce=False,
                 **engine_kw):
        import torch
        from .engine import ElboEngine
        self.torch = torch
        self.meta, self.rank, self.world, self.mode = meta, rank, world, mode
        self.always_reduce = bool(always_reduce)      # run the collective path even with one rank (tests)
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
        self.ext_stream = torch.cuda.ExternalStream(self.engine.stream_handle(), device=f"cuda:{device}")
        self._sync_step = False          # set when the stream-ordered step failed on its FIRST use (see step)
        self._stream_step_ok = False

    def step(self):
        """One ELBO iteration: local kernels -> 8 partial sums in HBM -> all-reduce -> host."""
        if self.world == 1 and not self.always_reduce:
            # nothing to reduce: the engine's own pinned-host copy of the 8 sums (one synchronisation, no torch hop)
            return self.engine.elbo_sums()
        if os.environ.get("FFVD_SYNC_STEP") or self._sync_step:   # conservative variant: host sync, collective on torch's stream
            self.engine.elbo_async(self.sums.data_ptr())
            self.engine.sync()
            all_reduce_sums(self.sums)
            return self.sums.cpu().numpy()
        # stream-ordered: the finalize kernel, the RCCL all-reduce and the device-to-host copy all follow the engine's
        # stream (torch sees it as an external stream), so the only host synchronisation is the final copy
        try:
            with self.torch.cuda.stream(self.ext_stream):
                self.engine.elbo_async(self.sums.data_ptr())
                all_reduce_sums(self.sums)
                out = self.sums.cpu().numpy()
        except RuntimeError as exc:
            # a collective backend that cannot run on an external stream: say so once and keep going the conservative
            # way (same kernels, same numbers, one more host synchronisation per step)
            if self._stream_step_ok:
                raise
            import warnings
            warnings.warn(f"stream-ordered collective step failed ({exc}); falling back to the synchronous step")
            self._sync_step = True
            return self.step()
        self._stream_step_ok = True
        if not np.all(np.isfinite(out)):
            self.engine.sync()          # a failed factorisation poisons the sums: fetch the info flags, raise LinAlgError
        return out

    def nll_terms(self):
        return finish(self.step())

    def nll_and_grad(self):
        """Whole-job nll terms and gradient (the engine must have been built with grad=True, route="gram"):
        local backward pass scaled by 1/S_total, then one all-reduce of the 8 sums and one of the packed
        shared-parameter gradients."""
        terms, g = self.engine.nll_and_grad(S_total=self.meta["S"])
        self.sums.copy_(self.torch.from_numpy(terms["sums8"]))
        all_reduce_sums(self.sums)
        g = all_reduce_grads(g, self.mode, device=self.sums.device)
        return finish(self.sums.cpu().numpy()), g
