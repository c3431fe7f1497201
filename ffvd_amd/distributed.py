"""Multi-GPU data parallelism for the ELBO: one process per GPU.

The nll is a sum of per-(chain, latent-dim) terms plus cheap shared terms, so it shards with NO data-path
collective; the only exchange is one all-reduce(sum) of the 8-double partial-sum vector of ffvd_abi.h
(`torch.distributed`, backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).

  mode "chains": rank r evaluates chains [s_begin, s_begin + s_count) for all latent dims (BASELINE configs 2-4)
  mode "dims"  : rank r evaluates latent dims [d_begin, d_begin + d_count) for all chains; only rank 0 adds
                 the shared terms (likelihood, prior_Z, prior_x_0, hyper prior)          (BASELINE config 5)
  mode "time"  : fallback when chains x dims < ranks (SURVEY 8e last bullet): rank r evaluates transitions
                 [t_begin, t_begin + t_count) of every (chain, dim) unit; the exchange is ONE all-reduce of the raw
                 Gram tiles K_uf K_fu + the per-chain sums (MiB-sized, the only link-bandwidth-bound collective of
                 this code base), after which every rank finishes the same factorisations
`plan(..., mode="auto")` picks chains, then dims, then time.
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
    if mode == "auto":
        mode = "chains" if S >= world else ("dims" if D >= world else "time")
    if mode == "time":
        T = meta["T"]
        if T < world:
            raise ValueError(f"cannot shard {T} transitions over {world} ranks")
        t_begin, t_count = shard_range(T, world, rank)
        return dict(s_begin=0, s_count=S, d_begin=0, d_count=D, shared_terms=True, t_begin=t_begin, t_count=t_count, mode="time")
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
    raise ValueError("mode must be 'chains', 'dims', 'time' or 'auto'")


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


_RENDEZVOUS_GENERATION = [0]      # communicators this process has formed through a rendezvous directory
import time as _time
_PROCESS_SEEN = _time.time()      # (file rendezvous: ids written before this process started belong to an earlier run)


def exchange_unique_id(make_id, rank, world, rendezvous_dir=None, timeout_s=120.0, tag=None, cleanup_s=None):
    """Hand rank 0's 128-byte RCCL id (ffvd_comm_unique_id) to every rank.  Host-side plumbing only:
    with `rendezvous_dir` through a file (written atomically by rank 0, polled by the others), otherwise through the
    initialised torch.distributed group (any backend; gloo in the launchers of this repo).

    A failure of `make_id` on rank 0 (librccl cannot be bound) is DELIVERED to every rank -- rank 0 still takes part in the
    exchange and sends the error text instead of the id -- so that all ranks raise the same RuntimeError at the same
    point and a caller's next collective (bench.py's "did every rank get a communicator" all-reduce) lines up.
    File rendezvous: the file name carries `tag` (default: the count of communicators this process has formed, which is
    the same on every rank of an SPMD program), so an id left behind by an earlier communicator or an earlier run with
    another tag is never read; rank 0 removes its file's predecessor before writing."""
    if world == 1:
        return make_id()

    def guarded():
        try:
            return bytes(make_id()), None
        except Exception as exc:        # noqa: BLE001 -- forwarded to every rank below
            return None, f"{type(exc).__name__}: {exc}"

    if rendezvous_dir:
        import time
        if tag is None:
            tag = _RENDEZVOUS_GENERATION[0]
            _RENDEZVOUS_GENERATION[0] += 1
        nonce = os.environ.get("FFVD_RENDEZVOUS_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or ""
        nonce = "".join(ch for ch in nonce if ch.isalnum() or ch in "-_")[:48]
        path = os.path.join(rendezvous_dir, f"rccl_unique_id.{nonce + '.' if nonce else ''}{tag}")
        if rank == 0:
            blob, err = guarded()
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            with open(path + ".tmp", "wb") as f:
                f.write(b"OK" + blob if err is None else b"ER" + err.encode())
            os.replace(path + ".tmp", path)
            if cleanup_s is not None:
                import threading

                def _remove():
                    try:
                        os.unlink(path)
                    except OSError:
                        pass
                timer = threading.Timer(float(cleanup_s), _remove)
                timer.daemon = True
                timer.start()
            if err is not None:
                raise RuntimeError(f"RCCL unique id could not be created on rank 0: {err}")
            return blob
        t0 = time.monotonic()
        born = _PROCESS_SEEN - float(os.environ.get("FFVD_RENDEZVOUS_SLACK_S", "2"))
        while True:
            # (consecutive runs with the same nonce -- MASTER_PORT 29500, run id 'none' -- share the file name; a file left by
            #  an EARLIER run was written before this process existed: a reader ignores anything older than the moment this module
            #  was imported (minus FFVD_RENDEZVOUS_SLACK_S, default 2 s, for ranks that start a little apart) and keeps polling for
            #  the file rank 0 of ITS job writes)
            try:
                if os.path.getmtime(path) >= born:
                    break
            except OSError:
                pass
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no fresh RCCL id at {path} after {timeout_s:.0f} s")
            time.sleep(0.01)
        with open(path, "rb") as f:
            data = f.read()
        if data[:2] != b"OK":
            raise RuntimeError(f"RCCL unique id could not be created on rank 0: {data[2:].decode(errors='replace')}")
        return data[2:]
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("exchange_unique_id: pass rendezvous_dir or initialise torch.distributed first")
    box = [guarded() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    blob, err = box[0]
    if err is not None:
        raise RuntimeError(f"RCCL unique id could not be created on rank 0: {err}")
    return blob


def tshard_nll_and_grad(engine, meta, t_begin, reduce_host, native=False):
    """nll terms + gradient of a T-sharded job from one shard's engine (include/ffvd_abi.h "Gradient of a T-sharded job").
    `engine` offers tshard_local / tshard_finish_grad / tshard_grad_fetch (or, native=True, elbo_tshard_grad with both exchange
    steps inside the library); `reduce_host(array) -> summed flat array` is the all-reduce over the shards.  Three exchange steps:
    raw tiles + chain sums; the gradient block (8 term sums of the first shard + every shared-parameter gradient); and dX, whose rows
    are placed at their global position first, so that the sum also adds the two parts of the rows neighbouring shards share."""
    S = meta["S"]
    if native:
        sums, g = engine.elbo_tshard_grad(S_total=S)
    else:
        t = np.asarray(reduce_host(engine.tshard_local()))
        # (a finish that fails on THIS rank -- a second stall, a HIP error -- must still take part in the reduce the other
        #  ranks are entering, with a NaN-headed block like the native path's, and raise afterwards: they then fail too instead of waiting)
        failure = None
        try:
            block = engine.tshard_finish_grad(t, S_total=S)
        except Exception as exc:        # noqa: BLE001 -- re-raised below, after the collective
            failure = exc
            block = np.full(int(engine.lib.ffvd_train_exchange_count(engine._h)), np.nan) if hasattr(engine, "lib") else None
            if block is None:
                raise
        block = np.asarray(reduce_host(block))
        if failure is not None:
            raise failure               # (the other ranks raise in tshard_grad_fetch on the NaN-headed block: nobody enters the dX exchange)
        sums, g = engine.tshard_grad_fetch(block)
    full = np.zeros((S, meta["T"] + 1, meta["D"]))
    full[:, t_begin: t_begin + g["X"].shape[1]] = g["X"]
    g = dict(g, X=np.asarray(reduce_host(full)).reshape(full.shape))
    return finish(sums), g


def tshard_boundary_rows(gX, rank, world, reduce_host):
    """Complete the rows of dX that neighbouring T-shards share.  `gX` = this shard's S x (tc + 1) x D rows (first / last row: this
    shard's part only).  One all-reduce of a (world - 1) x S x D buffer -- boundary b sits between shards b and b + 1; each shard adds
    its part of the (at most two) boundaries it touches -- instead of the whole trajectory's dX: 2 rows per shard travel, not T."""
    gX = np.array(gX, dtype=np.float64, copy=True)
    if world <= 1:
        return gX
    S, _, D = gX.shape
    B = np.zeros((world - 1, S, D))
    if rank > 0:
        B[rank - 1] = gX[:, 0, :]
    if rank < world - 1:
        B[rank] = gX[:, -1, :]
    B = np.asarray(reduce_host(B)).reshape(world - 1, S, D)
    if rank > 0:
        gX[:, 0, :] = B[rank - 1]
    if rank < world - 1:
        gX[:, -1, :] = B[rank]
    return gX


def tshard_adam_step(engine, meta, rank, world, reduce_host, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None, native=False):
    """One Adam step of a T-sharded job (dgp_model.py:303-305 across T-shards): the two exchange steps of the gradient (raw tiles +
    chain sums; the gradient block), the boundary rows of dX, then every shard updates its own rows of X and its copy of the shared
    parameters.  Returns the job's terms before the update."""
    S = meta["S"]
    if native:
        sums, g = engine.elbo_tshard_grad(S_total=S)
    else:
        t = np.asarray(reduce_host(engine.tshard_local()))
        sums, g = engine.tshard_grad_fetch(np.asarray(reduce_host(engine.tshard_finish_grad(t, S_total=S))))
    rows = tshard_boundary_rows(g["X"], rank, world, reduce_host)
    return finish(engine.tshard_adam_apply(rows, lr, beta1, beta2, eps, train))


def tshard_sghmc_step(engine, meta, reduce_host, noise, epsilon=0.01, mdecay=0.05, burn_in=True, native=False):
    """One burn_in_op / sample_op of a T-sharded job: the two exchange steps of the gradient, then every shard applies the update of
    the shared parameters with the same noise (X is never an SG-HMC variable: no rows travel)."""
    S = meta["S"]
    if native:
        engine.elbo_tshard_grad(S_total=S)
    else:
        t = np.asarray(reduce_host(engine.tshard_local()))
        engine.tshard_grad_fetch(np.asarray(reduce_host(engine.tshard_finish_grad(t, S_total=S))))
    return finish(engine.tshard_sghmc_apply(noise, epsilon, mdecay, burn_in))


class ShardedElbo:
    """One rank's share of the ELBO on its own GPU + the scalar all-reduce.

    collective="rccl" (default): the native path of include/ffvd_abi.h -- `ffvd_comm_init` creates the rank's
    ncclComm_t, `ffvd_elbo_allreduce` runs kernels -> finalize -> ncclAllReduce(8 doubles) -> one copy back on the
    engine's stream; no torch tensor or torch stream is involved (torch.distributed, if initialised, only carries the
    128-byte rendezvous id).
    collective="torch": the 8 partial sums are written into a torch CUDA tensor and reduced by
    `torch.distributed.all_reduce` -- for groups RCCL cannot form (two test ranks sharing ONE GPU over gloo)."""

    def __init__(self, params, Y, control_inputs, meta, rank=0, world=1, mode="chains", device=0, always_reduce=False,
                 collective="rccl", rendezvous_dir=None, rendezvous_tag=None, **engine_kw):
        from .engine import ElboEngine
        if collective not in ("rccl", "torch"):
            raise ValueError("collective must be 'rccl' or 'torch'")
        self.meta, self.rank, self.world, self.mode = meta, rank, world, mode
        self.collective = collective
        self.always_reduce = bool(always_reduce)      # run the collective path even with one rank (tests)
        self.plan = plan(meta, world, rank, mode)
        pl = self.plan
        self.time_shard = pl.get("mode") == "time"
        if self.time_shard:
            t0, tc = pl["t_begin"], pl["t_count"]
            engine_kw = dict(engine_kw, route="gram", t_shard=(t0, meta["T"]))
            self.engine = ElboEngine(tc, meta["D"], meta["C"], meta["M"], pl["s_count"], Ydim=meta["Ydim"],
                                     kernel_type=meta["kernel_type"], U_collapse=meta["U_collapse"], device=device,
                                     **engine_kw)
            ci = np.asarray(control_inputs, dtype=np.float64)[t0: t0 + tc] if meta["C"] > 0 else None
            self.engine.set_data(np.asarray(Y, dtype=np.float64)[t0: t0 + tc], ci)
            local = dict(params)
            local["X"] = np.ascontiguousarray(np.asarray(params["X"])[:, t0: t0 + tc + 1])
            self.engine.set_params(local)
        else:
            self.engine = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], pl["s_count"], Ydim=meta["Ydim"],
                                     kernel_type=meta["kernel_type"], U_collapse=meta["U_collapse"], device=device,
                                     d_begin=pl["d_begin"], d_count=pl["d_count"], shared_terms=pl["shared_terms"],
                                     **engine_kw)
            self.engine.set_data(Y, control_inputs)
            local = dict(params)
            local["X"] = np.ascontiguousarray(params["X"][pl["s_begin"]: pl["s_begin"] + pl["s_count"]])
            self.engine.set_params(local)
        self.reduces = world > 1 or self.always_reduce or self.time_shard
        self.engine.shard_of = world          # plain ElboEngine.adam_step would train on this rank's share only
        try:
            if self.reduces and collective == "rccl":
                blob = exchange_unique_id(self.engine.comm_unique_id, rank, world, rendezvous_dir, tag=rendezvous_tag)
                self.engine.comm_init(world, rank, blob)
            elif self.reduces and not self.time_shard:
                import torch
                self.torch = torch
                self.sums = torch.zeros(8, dtype=torch.float64, device=f"cuda:{device}")
                self.ext_stream = torch.cuda.ExternalStream(self.engine.stream_handle(), device=f"cuda:{device}")
        except BaseException:
            self.engine.close()           # a half-built shard must not keep its (multi-GB) workspaces alive through a traceback
            raise

    def close(self):
        self.engine.close()

    def step(self):
        """One ELBO iteration: local kernels -> 8 partial sums in HBM -> all-reduce -> host.  Errors are not masked:
        a failing collective or HIP call raises; a failed factorisation on any rank raises LinAlgError."""
        if self.time_shard:
            if self.collective == "rccl":
                return self.engine.elbo_tshard()
            # host-carried exchange (test groups RCCL cannot form): partial sums -> all-reduce on the CPU -> finish
            import torch
            t = torch.from_numpy(self.engine.tshard_local())
            all_reduce_sums(t)
            return self.engine.tshard_finish(t.numpy())
        if not self.reduces:
            # nothing to reduce: the engine's own pinned-host copy of the 8 sums (one synchronisation)
            return self.engine.elbo_sums()
        if self.collective == "rccl":
            return self.engine.elbo_allreduce()
        if os.environ.get("FFVD_SYNC_STEP"):   # explicit opt-in, conservative variant: host sync, collective on torch's stream
            self.engine.elbo_async(self.sums.data_ptr())
            self.engine.sync()
            all_reduce_sums(self.sums)
            return self.sums.cpu().numpy()
        # stream-ordered: the finalize kernel, the all-reduce and the device-to-host copy all follow the engine's
        # stream (torch sees it as an external stream), so the only host synchronisation is the final copy
        with self.torch.cuda.stream(self.ext_stream):
            self.engine.elbo_async(self.sums.data_ptr())
            all_reduce_sums(self.sums)
            out = self.sums.cpu().numpy()
        self.engine.sync()              # reports this rank's Cholesky info flags (LinAlgError)
        if not np.all(np.isfinite(out)):
            raise np.linalg.LinAlgError("non-finite partial sums after the all-reduce: a factorisation failed on another rank")
        return out

    def nll_terms(self):
        return finish(self.step())

    def nll_and_grad(self):
        """Whole-job nll terms and gradient (the engine must have been built with grad=True, route="gram"):
        local backward pass scaled by 1/S_total, then one all-reduce of the 8 sums and one of the packed
        shared-parameter gradients.  T-shards: two exchange steps (raw tiles + chain sums, then the gradient block); dX comes
        back for the WHOLE trajectory -- every shard's rows placed at their global position and summed, which also adds the two
        parts of the rows neighbouring shards share."""
        if self.time_shard:
            if self.collective == "rccl":
                return tshard_nll_and_grad(self.engine, self.meta, self.plan["t_begin"], self.engine.allreduce_host, native=True)
            return tshard_nll_and_grad(self.engine, self.meta, self.plan["t_begin"],
                                       lambda a: self._host_reduce(np.ascontiguousarray(a, dtype=np.float64).ravel().copy()))
        terms, g = self.engine.nll_and_grad(S_total=self.meta["S"])
        if not self.reduces:
            return finish(terms["sums8"]), g
        if self.collective == "rccl":
            keys = list(GRAD_KEYS) + (["U"] if "U" in g else []) + (["X"] if self.mode == "dims" else [])
            flat = np.concatenate([terms["sums8"]] + [np.asarray(g[k], dtype=np.float64).ravel() for k in keys])
            flat = self.engine.allreduce_host(flat)          # ONE ncclAllReduce: 8 sums + packed shared gradients
            out, off = dict(g), 8
            for k in keys:
                n = int(np.asarray(g[k]).size)
                out[k] = flat[off: off + n].reshape(np.asarray(g[k]).shape).copy()
                off += n
            return finish(flat[:8]), out
        self.sums.copy_(self.torch.from_numpy(terms["sums8"]))
        all_reduce_sums(self.sums)
        g = all_reduce_grads(g, self.mode, device=self.sums.device)
        return finish(self.sums.cpu().numpy()), g

    # -- device-resident sharded training (dgp_model.py:303-305, base_model.py:944-950 / :143-179 across ranks) -----------
    def _host_reduce(self, block):
        import torch
        t = torch.from_numpy(block)
        all_reduce_sums(t)
        return t.numpy()

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None):
        """One Adam step of the WHOLE job: every rank's forward + backward, ONE all-reduce of the gradient block
        [8 sums | shared-parameter gradients (| dX for latent-dim shards)], fused update on the device.  With the native
        collective nothing but the 8 sums reaches the host; groups RCCL cannot form (collective="torch": two test ranks on
        one GPU) carry the block through torch.distributed.  Returns the whole-job terms before the update."""
        if self.time_shard:
            # (T-shards: two exchange steps of the gradient + the boundary rows of dX, tshard_adam_step)
            if self.collective == "rccl":
                return tshard_adam_step(self.engine, self.meta, self.rank, self.world, self.engine.allreduce_host, lr, beta1, beta2,
                                        eps, train, native=True)
            return tshard_adam_step(self.engine, self.meta, self.rank, self.world,
                                    lambda a: self._host_reduce(np.ascontiguousarray(a, dtype=np.float64).ravel().copy()),
                                    lr, beta1, beta2, eps, train)
        S = self.meta["S"]
        if not self.reduces:
            self.engine.shard_of = 1
            try:
                t = self.engine.adam_step(lr, beta1, beta2, eps, train)
            finally:
                self.engine.shard_of = self.world
            return {k: v for k, v in t.items()}
        if self.collective == "rccl":
            return finish(self.engine.adam_step_allreduce(S, lr, beta1, beta2, eps, train))
        block = self._host_reduce(self.engine.train_local(S))
        return finish(self.engine.adam_apply(block, lr, beta1, beta2, eps, train))

    def sghmc_step(self, noise, epsilon=0.01, mdecay=0.05, burn_in=True):
        """One burn_in_op / sample_op of the whole job; `noise` must be identical on every rank."""
        if self.time_shard:
            if self.collective == "rccl":
                return tshard_sghmc_step(self.engine, self.meta, self.engine.allreduce_host, noise, epsilon, mdecay, burn_in, native=True)
            return tshard_sghmc_step(self.engine, self.meta,
                                     lambda a: self._host_reduce(np.ascontiguousarray(a, dtype=np.float64).ravel().copy()),
                                     noise, epsilon, mdecay, burn_in)
        S = self.meta["S"]
        if not self.reduces:
            self.engine.shard_of = 1
            try:
                return self.engine.sghmc_step(noise, epsilon, mdecay, burn_in)
            finally:
                self.engine.shard_of = self.world
        if self.collective == "rccl":
            return finish(self.engine.sghmc_step_allreduce(S, noise, epsilon, mdecay, burn_in))
        block = self._host_reduce(self.engine.train_local(S))
        return finish(self.engine.sghmc_apply(block, noise, epsilon, mdecay, burn_in))
