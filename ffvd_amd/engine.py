"""ElboEngine: thin Python owner of one `ffvd_handle` (one GPU, one HIP stream, resident buffers).

This is plumbing over include/ffvd_abi.h: every number it returns is computed by the HIP kernels.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib

PARAM_KEYS = ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


class ElboEngine:
    """Evaluates `DGPSSM.nll` (dgp_model.py:248-297) for S latent trajectories on one MI355X.

    Shapes follow SURVEY.md: X (S, T+1, D); Z (M, D+C); U (M, D); logvariance (D,);
    loglengthscales (D, D+C); log_Q (D,); CC (D, Ydim); DD (Ydim,); log_Rchols (Ydim, Ydim).

    route (collapsed branch only): "reference" forms F = K_fu L^-T and H = F^T F / Q + I in the reference's
    op order; "gram" evaluates the same bound as log|K_uu + K_uf K_fu / Q| - log|K_uu| (about half the flops,
    agrees to ~1e-9 relative; see include/ffvd_abi.h FFVD_ROUTE_*).

    dtype: "f64" (the reference's) or "f32c" = K_fu and the two T x M x M contractions in fp32 on the matrix cores,
    everything M x M and every accumulation that feeds the scalar in fp64 (include/ffvd_abi.h FFVD_F32C).
    """

    def __init__(self, T, D, C, M, S, Ydim=1, kernel_type="SquaredExponential", U_collapse=True,
                 prior_type="normal", device=0, d_begin=0, d_count=0, shared_terms=True,
                 chains_per_pass=0, jitter=1e-5, route="reference", grad=False, dtype="f64", t_shard=None):
        if kernel_type not in _lib.KERNEL_KIND:
            raise ValueError("Invalid kernel type")
        if prior_type not in _lib.PRIOR_TYPE:
            raise ValueError("Invalid prior type")           # models.py:41
        if route not in _lib.ROUTE:
            raise ValueError("route must be 'reference' or 'gram'")
        if dtype not in _lib.DTYPE:
            raise ValueError("dtype must be 'f64' or 'f32c'")
        self.route = route
        self.dtype = dtype
        # t_shard = (t_begin, T_total): this engine holds rows [t_begin, t_begin + T) of a job with T_total transitions
        # (include/ffvd_abi.h "T-shard"); X then has T + 1 rows starting at global row t_begin
        self.t_shard = None if t_shard is None else (int(t_shard[0]), int(t_shard[1]))
        self.grad = bool(grad)
        self.lib = _lib.load()
        self.T, self.D, self.C, self.M, self.S, self.Ydim = int(T), int(D), int(C), int(M), int(S), int(Ydim)
        self.P = self.D + self.C
        self.kernel_type, self.U_collapse = kernel_type, bool(U_collapse)
        self.d_begin, self.d_count = int(d_begin), int(d_count) or int(D)
        self.shared_terms = bool(shared_terms)
        cfg = _lib.FfvdConfig(
            T=self.T, D=self.D, C=self.C, M=self.M, S_local=self.S, Ydim=self.Ydim, d_begin=self.d_begin,
            d_count=self.d_count, shared_terms=int(self.shared_terms), dtype=_lib.DTYPE[dtype],
            kernel_kind=_lib.KERNEL_KIND[kernel_type], branch=_lib.BRANCH_B if U_collapse else _lib.BRANCH_A,
            prior_type=_lib.PRIOR_TYPE[prior_type], device_id=int(device), chains_per_pass=int(chains_per_pass),
            route=_lib.ROUTE[route], grad=int(bool(grad)), T_total=self.t_shard[1] if self.t_shard else 0,
            t_begin=self.t_shard[0] if self.t_shard else 0, reserved=0, jitter=float(jitter))
        self._h = ct.c_void_p()
        _lib.check(self.lib.ffvd_create(ct.byref(cfg), ct.byref(self._h)), None, "ffvd_create")
        self._keep = {}

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ffvd_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def workspace_bytes(self):
        return int(self.lib.ffvd_workspace_bytes(self._h))

    # -- inputs -----------------------------------------------------------------------------
    def set_data(self, Y, control_inputs):
        """Y: (T, Ydim); control_inputs: (>=T, C), the first T rows are used (dgp_model.py:255)."""
        Y = _lib.as_f64(Y, (self.T, self.Ydim), "Y")
        if self.C > 0:
            c = np.asarray(control_inputs, dtype=np.float64)
            if c.ndim != 2 or c.shape[1] != self.C or c.shape[0] < self.T:
                raise ValueError(f"control_inputs: expected (>= {self.T}, {self.C}), got {c.shape}")
            c = np.ascontiguousarray(c[: self.T])
            cp = c.ctypes.data
        else:
            c, cp = None, None
        _lib.check(self.lib.ffvd_set_data(self._h, Y.ctypes.data, cp, 0), self._h, "ffvd_set_data")

    def _pack(self, params):
        X = np.asarray(params["X"], dtype=np.float64)
        if X.ndim == 2:
            X = X[None]
        arrs = {
            "X": _lib.as_f64(X, (self.S, self.T + 1, self.D), "X"),
            "Z": _lib.as_f64(params["Z"], (self.M, self.P), "Z"),
            "logvariance": _lib.as_f64(params["logvariance"], (self.D,), "logvariance"),
            "log_Q": _lib.as_f64(params["log_Q"], (self.D,), "log_Q"),
            "CC": _lib.as_f64(params["CC"], (self.D, self.Ydim), "CC"),
            "DD": _lib.as_f64(params["DD"], (self.Ydim,), "DD"),
            "log_Rchols": _lib.as_f64(params["log_Rchols"], (self.Ydim, self.Ydim), "log_Rchols"),
        }
        if params.get("U") is not None:
            arrs["U"] = _lib.as_f64(params["U"], (self.M, self.D), "U")
        elif not self.U_collapse:
            raise ValueError("U is required in the explicit-U branch")
        if self.kernel_type == "SquaredExponential":
            arrs["loglengthscales"] = _lib.as_f64(params["loglengthscales"], (self.D, self.P), "loglengthscales")
        p = _lib.FfvdParams()
        for k in PARAM_KEYS:
            setattr(p, k, arrs[k].ctypes.data if k in arrs else None)
        return p, arrs

    def set_params(self, params):
        """Upload the parameters into the handle's resident device buffers."""
        p, arrs = self._pack(params)
        _lib.check(self.lib.ffvd_set_params(self._h, ct.byref(p), 0), self._h, "ffvd_set_params")

    # -- the hot path -------------------------------------------------------------------------
    def elbo_sums(self, params=None):
        """One ELBO iteration.  Returns the 8-vector of ffvd_abi.h: sums over local chains + chain count."""
        out = np.zeros(8)
        nll = ct.c_double()
        if params is not None:
            p, arrs = self._pack(params)
            rc = self.lib.ffvd_elbo(self._h, ct.byref(p), 0, _lib.dptr(out), ct.byref(nll))
        else:
            rc = self.lib.ffvd_elbo(self._h, None, 0, _lib.dptr(out), ct.byref(nll))
        _lib.check(rc, self._h, "ffvd_elbo")
        return out

    def nll_terms(self, params=None):
        """Mean over the local chains of nll and its named component terms (reference names)."""
        sums = self.elbo_sums(params)
        names = _lib.TERM_NAMES if self.U_collapse else _lib.TERM_NAMES[:4] + ("nll",)
        idx = {n: i for i, n in enumerate(_lib.TERM_NAMES)}
        out = {n: float(sums[idx[n]] / self.S) for n in names}
        out["nll_per_chain"] = self.chain_nll()
        return out

    def nll(self, params=None):
        return self.nll_terms(params)["nll"]

    def chain_nll(self):
        out = np.zeros(self.S)
        _lib.check(self.lib.ffvd_chain_nll(self._h, _lib.dptr(out)), self._h, "ffvd_chain_nll")
        return out

    def nll_and_grad(self, params=None, S_total=None):
        """nll terms and the gradient of the mean-over-chains nll w.r.t. every parameter
        (tf.gradients(nll, vars), base_model.py:148).  Needs grad=True; both kernels, both branches, both routes (fp32 contractions:
        SE kernel only).

        Returns (terms dict, grads dict with keys X, Z, logvariance, loglengthscales, log_Q, CC, DD, log_Rchols).
        With chains sharded over ranks pass S_total = chains of the whole job and sum the shared-parameter
        gradients over the ranks (X gradients are per rank)."""
        if not self.grad:
            raise ValueError("engine was created without grad=True")
        S_total = int(S_total or self.S)
        g = {
            "X": np.zeros((self.S, self.T + 1, self.D)), "Z": np.zeros((self.M, self.P)),
            "logvariance": np.zeros(self.D), "loglengthscales": np.zeros((self.D, self.P)),
            "log_Q": np.zeros(self.D), "CC": np.zeros((self.D, self.Ydim)), "DD": np.zeros(self.Ydim),
            "log_Rchols": np.zeros((self.Ydim, self.Ydim)), "U": np.zeros((self.M, self.D)),
        }
        gs = _lib.FfvdGrads(**{k: v.ctypes.data for k, v in g.items()})
        out = np.zeros(8)
        nll = ct.c_double()
        if params is not None:
            p, arrs = self._pack(params)
            rc = self.lib.ffvd_elbo_grad(self._h, ct.byref(p), 0, S_total, _lib.dptr(out), ct.byref(nll), ct.byref(gs))
        else:
            rc = self.lib.ffvd_elbo_grad(self._h, None, 0, S_total, _lib.dptr(out), ct.byref(nll), ct.byref(gs))
        _lib.check(rc, self._h, "ffvd_elbo_grad")
        idx = {n: i for i, n in enumerate(_lib.TERM_NAMES)}
        terms = {n: float(out[idx[n]] / self.S) for n in _lib.TERM_NAMES}
        terms["nll_per_chain"] = self.chain_nll()
        terms["sums8"] = out          # raw partial sums + chain count, for the all-reduce of a sharded job
        return terms, g

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None):
        """One `train_hypers` iteration (base_model.py:944-950) on the device: forward + backward on the resident
        parameters, then the Adam update (TF defaults) of the arrays named in `train` (default: all).  Returns the
        nll terms of the parameters BEFORE the update.  Needs grad=True and set_params() first."""
        if not self.grad:
            raise ValueError("engine was created without grad=True")
        if getattr(self, "shard_of", 1) > 1:
            raise ValueError("adam_step: this engine holds one shard of a multi-rank job; use ShardedElbo.adam_step")
        mask = _lib.TRAIN_ALL if train is None else sum(_lib.TRAIN_BITS[k] for k in train)
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_adam_step(self._h, float(lr), float(beta1), float(beta2), float(eps), int(mask),
                                           _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_adam_step")
        idx = {n: i for i, n in enumerate(_lib.TERM_NAMES)}
        return {n: float(out[idx[n]] / self.S) for n in _lib.TERM_NAMES}

    def sghmc_step(self, noise, epsilon=0.01, mdecay=0.05, burn_in=True):
        """One burn_in_op / sample_op (base_model.py:143-179) on the device for the arrays that are keys of `noise`
        (name -> standard-normal array of the parameter's shape).  Returns the nll terms before the update."""
        if getattr(self, "shard_of", 1) > 1:
            raise ValueError("sghmc_step: this engine holds one shard of a multi-rank job; use ShardedElbo.sghmc_step")
        if not self.grad:
            raise ValueError("engine was created without grad=True")
        shapes = {"Z": (self.M, self.P), "logvariance": (self.D,), "loglengthscales": (self.D, self.P),
                  "log_Q": (self.D,), "CC": (self.D, self.Ydim), "DD": (self.Ydim,), "log_Rchols": (self.Ydim, self.Ydim),
                  "U": (self.M, self.D)}
        arrs, mask = {}, 0
        for k, v in noise.items():
            if k not in shapes:
                raise ValueError(f"sghmc_step: '{k}' cannot be an SG-HMC variable")
            arrs[k] = _lib.as_f64(v, shapes[k], f"noise[{k}]")
            mask |= _lib.TRAIN_BITS[k]
        ps = _lib.FfvdParams(**{k: v.ctypes.data for k, v in arrs.items()})
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_sghmc_step(self._h, float(epsilon), float(mdecay), int(mask), int(bool(burn_in)),
                                            ct.byref(ps), _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_sghmc_step")
        idx = {n: i for i, n in enumerate(_lib.TERM_NAMES)}
        return {n: float(out[idx[n]] / self.S) for n in _lib.TERM_NAMES}

    # -- sharded, device-resident training steps (include/ffvd_abi.h "sharded training step") -------------------
    def _noise_struct(self, noise, who):
        shapes = {"Z": (self.M, self.P), "logvariance": (self.D,), "loglengthscales": (self.D, self.P),
                  "log_Q": (self.D,), "CC": (self.D, self.Ydim), "DD": (self.Ydim,), "log_Rchols": (self.Ydim, self.Ydim),
                  "U": (self.M, self.D)}
        arrs, mask = {}, 0
        for k, v in noise.items():
            if k not in shapes:
                raise ValueError(f"{who}: '{k}' cannot be an SG-HMC variable")
            arrs[k] = _lib.as_f64(v, shapes[k], f"noise[{k}]")
            mask |= _lib.TRAIN_BITS[k]
        return _lib.FfvdParams(**{k: v.ctypes.data for k, v in arrs.items()}), mask, arrs

    def adam_step_allreduce(self, S_total, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None, comm=None):
        """One sharded `train_hypers` iteration: this rank's forward + backward (divisor S_total), ONE ncclAllReduce of the
        gradient block in HBM, fused Adam update from the reduced block.  Returns the whole-job 8 sums (before the update)."""
        mask = _lib.TRAIN_ALL if train is None else sum(_lib.TRAIN_BITS[k] for k in train)
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_adam_step_allreduce(self._h, comm, int(S_total), float(lr), float(beta1), float(beta2), float(eps),
                                                     int(mask), _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_adam_step_allreduce")
        return out

    def sghmc_step_allreduce(self, S_total, noise, epsilon=0.01, mdecay=0.05, burn_in=True, comm=None):
        """One sharded burn_in_op / sample_op; `noise` must be the same on every rank.  Returns the whole-job 8 sums."""
        ps, mask, keep = self._noise_struct(noise, "sghmc_step_allreduce")
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_sghmc_step_allreduce(self._h, comm, int(S_total), float(epsilon), float(mdecay), int(mask),
                                                      int(bool(burn_in)), ct.byref(ps), _lib.dptr(out), ct.byref(nll)),
                   self._h, "ffvd_sghmc_step_allreduce")
        del keep
        return out

    def train_local(self, S_total):
        """Three-step form, step 1: forward + backward (divisor S_total); returns the exchange block as a host array."""
        _lib.check(self.lib.ffvd_train_local(self._h, int(S_total)), self._h, "ffvd_train_local")
        buf = np.zeros(int(self.lib.ffvd_train_exchange_count(self._h)))
        _lib.check(self.lib.ffvd_train_exchange_get(self._h, _lib.dptr(buf)), self._h, "ffvd_train_exchange_get")
        return buf

    def _train_set(self, reduced):
        r = _lib.as_f64(reduced, (int(self.lib.ffvd_train_exchange_count(self._h)),), "reduced")
        _lib.check(self.lib.ffvd_train_exchange_set(self._h, _lib.dptr(r)), self._h, "ffvd_train_exchange_set")

    def adam_apply(self, reduced, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None):
        """Three-step form, steps 2-3: upload the all-reduced block, then the Adam update from it.  Returns the 8 sums."""
        self._train_set(reduced)
        mask = _lib.TRAIN_ALL if train is None else sum(_lib.TRAIN_BITS[k] for k in train)
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_adam_apply(self._h, float(lr), float(beta1), float(beta2), float(eps), int(mask),
                                            _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_adam_apply")
        return out

    def sghmc_apply(self, reduced, noise, epsilon=0.01, mdecay=0.05, burn_in=True):
        self._train_set(reduced)
        ps, mask, keep = self._noise_struct(noise, "sghmc_apply")
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_sghmc_apply(self._h, float(epsilon), float(mdecay), int(mask), int(bool(burn_in)), ct.byref(ps),
                                             _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_sghmc_apply")
        del keep
        return out

    def update_params(self, arrays):
        """Overwrite some resident parameter arrays (dict name -> array); the others keep their device values."""
        shapes = {"X": (self.S, self.T + 1, self.D), "Z": (self.M, self.P), "U": (self.M, self.D),
                  "logvariance": (self.D,), "loglengthscales": (self.D, self.P), "log_Q": (self.D,),
                  "CC": (self.D, self.Ydim), "DD": (self.Ydim,), "log_Rchols": (self.Ydim, self.Ydim)}
        arrs = {k: _lib.as_f64(v, shapes[k], k) for k, v in arrays.items()}
        ps = _lib.FfvdParams(**{k: v.ctypes.data for k, v in arrs.items()})
        _lib.check(self.lib.ffvd_update_params(self._h, ct.byref(ps)), self._h, "ffvd_update_params")

    def reset_optimizer(self):
        _lib.check(self.lib.ffvd_optimizer_reset(self._h), self._h, "ffvd_optimizer_reset")

    def get_params(self):
        """Host copies of the resident parameters (after optimiser steps)."""
        g = {
            "X": np.zeros((self.S, self.T + 1, self.D)), "Z": np.zeros((self.M, self.P)), "U": np.zeros((self.M, self.D)),
            "logvariance": np.zeros(self.D), "loglengthscales": np.zeros((self.D, self.P)),
            "log_Q": np.zeros(self.D), "CC": np.zeros((self.D, self.Ydim)), "DD": np.zeros(self.Ydim),
            "log_Rchols": np.zeros((self.Ydim, self.Ydim)),
        }
        ps = _lib.FfvdParams(**{k: v.ctypes.data for k, v in g.items()})
        _lib.check(self.lib.ffvd_get_params(self._h, ct.byref(ps)), self._h, "ffvd_get_params")
        return g

    def elbo_async(self, out_dev_ptr=None):
        """Enqueue one iteration; the 8 partial sums land in device memory `out_dev_ptr` (int address)."""
        _lib.check(self.lib.ffvd_elbo_async(self._h, out_dev_ptr), self._h, "ffvd_elbo_async")

    # -- native RCCL collectives (include/ffvd_abi.h "multi-GPU") -------------------------------------
    def comm_unique_id(self):
        """The 128-byte RCCL rendezvous id (rank 0 calls this and hands the bytes to the other ranks)."""
        buf = ct.create_string_buffer(128)
        _lib.check(self.lib.ffvd_comm_unique_id(buf), None, "ffvd_comm_unique_id")
        return bytes(buf.raw)

    def comm_init(self, world, rank, unique_id):
        """ncclCommInitRank on this engine's device (collective over all ranks); the handle owns the communicator."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        buf = ct.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self.lib.ffvd_comm_init(self._h, int(world), int(rank), buf), self._h, "ffvd_comm_init")

    def elbo_allreduce(self, comm=None):
        """One iteration of this rank's shard + ncclAllReduce of the 8 partial sums; returns the whole-job sums."""
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_elbo_allreduce(self._h, comm, _lib.dptr(out), ct.byref(nll)), self._h,
                   "ffvd_elbo_allreduce")
        return out

    # -- T-shard fallback -----------------------------------------------------------------------
    def elbo_tshard(self, comm=None):
        """Local rows -> ncclAllReduce of the raw Gram tiles + chain sums -> finish; returns the whole-job 8 sums."""
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_elbo_tshard(self._h, comm, _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_elbo_tshard")
        return out

    def tshard_local(self):
        """Enqueue this shard's partial sums and return the exchange buffer as a host array (three-step form)."""
        _lib.check(self.lib.ffvd_tshard_local(self._h), self._h, "ffvd_tshard_local")
        buf = np.zeros(int(self.lib.ffvd_tshard_count(self._h)))
        _lib.check(self.lib.ffvd_tshard_get(self._h, _lib.dptr(buf)), self._h, "ffvd_tshard_get")
        return buf

    def tshard_finish(self, reduced):
        """Upload the all-reduced exchange buffer and finish: returns the whole-job 8 sums."""
        r = _lib.as_f64(reduced, (int(self.lib.ffvd_tshard_count(self._h)),), "reduced")
        _lib.check(self.lib.ffvd_tshard_set(self._h, _lib.dptr(r)), self._h, "ffvd_tshard_set")
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_tshard_finish(self._h, _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_tshard_finish")
        return out

    # gradient of a T-sharded job (include/ffvd_abi.h "Gradient of a T-sharded job")
    def _grad_arrays(self):
        g = {
            "X": np.zeros((self.S, self.T + 1, self.D)), "Z": np.zeros((self.M, self.P)),
            "logvariance": np.zeros(self.D), "loglengthscales": np.zeros((self.D, self.P)),
            "log_Q": np.zeros(self.D), "CC": np.zeros((self.D, self.Ydim)), "DD": np.zeros(self.Ydim),
            "log_Rchols": np.zeros((self.Ydim, self.Ydim)), "U": np.zeros((self.M, self.D)),
        }
        return g, _lib.FfvdGrads(**{k: v.ctypes.data for k, v in g.items()})

    def elbo_tshard_grad(self, S_total=None, comm=None):
        """Local rows -> ncclAllReduce (tiles + chain sums) -> finish + backward pass -> ncclAllReduce (gradient block).
        Returns (whole-job 8 sums, grads dict); grads["X"] holds this shard's own T + 1 rows (boundary rows: add the neighbour's)."""
        g, gs = self._grad_arrays()
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_elbo_tshard_grad(self._h, comm, int(S_total or self.S), _lib.dptr(out), ct.byref(nll), ct.byref(gs)),
                   self._h, "ffvd_elbo_tshard_grad")
        return out, g

    def tshard_finish_grad(self, reduced, S_total=None):
        """Three-step form: upload the all-reduced exchange buffer, finish + backward pass; returns this shard's gradient block
        (to be summed over the shards) as a host array."""
        r = _lib.as_f64(reduced, (int(self.lib.ffvd_tshard_count(self._h)),), "reduced")
        _lib.check(self.lib.ffvd_tshard_set(self._h, _lib.dptr(r)), self._h, "ffvd_tshard_set")
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_tshard_finish_grad(self._h, int(S_total or self.S), _lib.dptr(out), ct.byref(nll)), self._h,
                   "ffvd_tshard_finish_grad")
        buf = np.zeros(int(self.lib.ffvd_train_exchange_count(self._h)))
        _lib.check(self.lib.ffvd_train_exchange_get(self._h, _lib.dptr(buf)), self._h, "ffvd_train_exchange_get")
        return buf

    def tshard_grad_fetch(self, reduced_block):
        """... and the summed block back: returns (whole-job 8 sums, grads dict)."""
        self._train_set(reduced_block)
        g, gs = self._grad_arrays()
        out = np.zeros(8)
        _lib.check(self.lib.ffvd_tshard_grad_fetch(self._h, _lib.dptr(out), ct.byref(gs)), self._h, "ffvd_tshard_grad_fetch")
        return out, g

    def tshard_adam_apply(self, dX_rows, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None):
        """Optimiser step of a T-sharded job, last part: `dX_rows` = this shard's S x (T + 1) x D rows of dX with the neighbours'
        parts of the first and last row added (distributed.tshard_adam_step); the shared-parameter gradients are the ones the
        preceding tshard_grad_fetch / elbo_tshard_grad left on the device.  Returns the job's 8 sums (before the update)."""
        rows = _lib.as_f64(dX_rows, (self.S, self.T + 1, self.D), "dX_rows")
        mask = _lib.TRAIN_ALL if train is None else sum(_lib.TRAIN_BITS[k] for k in train)
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_tshard_adam_apply(self._h, _lib.dptr(rows), float(lr), float(beta1), float(beta2), float(eps),
                                                   int(mask), _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_tshard_adam_apply")
        return out

    def tshard_sghmc_apply(self, noise, epsilon=0.01, mdecay=0.05, burn_in=True):
        """SG-HMC update of a T-sharded job from the block the preceding tshard_grad_fetch / elbo_tshard_grad left on the device."""
        ps, mask, keep = self._noise_struct(noise, "tshard_sghmc_apply")
        out = np.zeros(8)
        nll = ct.c_double()
        _lib.check(self.lib.ffvd_tshard_sghmc_apply(self._h, float(epsilon), float(mdecay), int(mask), int(bool(burn_in)), ct.byref(ps),
                                                    _lib.dptr(out), ct.byref(nll)), self._h, "ffvd_tshard_sghmc_apply")
        del keep
        return out

    def allreduce_host(self, array, comm=None):
        """all-reduce(sum) of a small host fp64 array through the handle's device staging buffer on the engine's stream
        (8 sums + shared-parameter gradients of a sharded training step: a few KB).  Returns a new flat array."""
        a = np.ascontiguousarray(np.asarray(array, dtype=np.float64)).ravel().copy()
        _lib.check(self.lib.ffvd_allreduce_sum(self._h, comm, _lib.dptr(a), a.size), self._h, "ffvd_allreduce_sum")
        return a

    def stream_handle(self):
        """The engine's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        return int(self.lib.ffvd_get_stream(self._h) or 0)

    def sync(self):
        _lib.check(self.lib.ffvd_sync(self._h), self._h, "ffvd_sync")

    def time_elbo(self, iters):
        """Total milliseconds (HIP events on the handle's stream) of `iters` back-to-back iterations."""
        ms = ct.c_float()
        _lib.check(self.lib.ffvd_time_elbo(self._h, int(iters), ct.byref(ms)), self._h, "ffvd_time_elbo")
        return float(ms.value)

    STAGES = ("kuu_chol_inverse", "project_F", "gram_H", "chol_H_solve", "reduce_finalize")

    def stage_timing(self, enable=True):
        """Record HIP events around every stage of subsequent iterations (live, on the handle's stream)."""
        _lib.check(self.lib.ffvd_stage_timing(self._h, int(bool(enable))), self._h, "ffvd_stage_timing")

    def stage_times(self):
        """{stage: (total ms, timed launch groups)} accumulated since the last read."""
        ms = np.zeros(8)
        n = np.zeros(8, dtype=np.int32)
        _lib.check(self.lib.ffvd_stage_times(self._h, _lib.dptr(ms), n.ctypes.data_as(ct.POINTER(ct.c_int32))),
                   self._h, "ffvd_stage_times")
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(self.STAGES)}

    def profile_stages(self):
        ms = (ct.c_float * 8)()
        _lib.check(self.lib.ffvd_profile_stages(self._h, ms), self._h, "ffvd_profile_stages")
        names = ("kuu_chol_inverse", "project_F", "gram_H", "chol_H_solve", "reduce_finalize")
        return {n: float(ms[i]) for i, n in enumerate(names)}
