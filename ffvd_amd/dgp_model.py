"""DGPSSM: the model object whose `nll` is the hot path -- counterpart of vfegpssm/dgp_model.py:160-324.

The reference builds a TensorFlow graph whose root tensor is `self.nll` and evaluates it with
`session.run` (base_model.py:952-989).  Here the same quantities are methods backed by one ElboEngine
(hand-written HIP kernels behind the C ABI); parameters live in NumPy arrays on the object with the
reference's attribute names (layers[-1].X / .U / .Z, kernels, log_Q, likelihood.CC / .DD / .log_Rchols).
"""
from __future__ import annotations

import numpy as np

from .engine import ElboEngine
from .kernels import stack_hypers


class Layer:
    """Layer.__init__ (dgp_model.py:46-94): variables X (T+1, D), U (M, D), Z (M, P) and the kernel list."""

    def __init__(self, ZZ, U_ini, X_0_ini, X_train_ini, kern, outputs, n_inducing, x_dims_l, Y_len,
                 prior_type="uniform", kernel_type="SquaredExponential"):
        self.inputs, self.outputs, self.kernel, self.kernel_type = kern[0].input_dim, outputs, kern, kernel_type
        self.M = n_inducing
        self.prior_type = prior_type
        X_ini_val = np.zeros((Y_len + 1, x_dims_l))            # dgp_model.py:56
        X_ini_val[0] = X_0_ini                                 # :57
        X_ini_val[1:] = X_train_ini                            # :58
        self.X = X_ini_val
        self.U = np.array(U_ini, dtype=np.float64)             # :66
        self.Z = np.array(ZZ, dtype=np.float64)                # :67


class DGPSSM:
    """One-layer GP state-space model; `nll()` / `nll_terms()` evaluate dgp_model.py:248-297 on the GPU.

    Constructor arguments keep the reference's names (dgp_model.py:160-166).  `num_chains` > 1 evaluates
    S latent trajectories X_s in one call (`set_X`), which is what BASELINE.json's metric measures."""

    def __init__(self, Y, x_dims, n_inducing, kernels, likelihood, minibatch_size=None, window_size=64,
                 output_dim=None, prior_type="uniform", full_cov=False, epsilon=0.01, mdecay=0.05, QQ_chol=None,
                 ZZ=None, variance=None, lengthscales=None, control_inputs=None, kernel_type="SquaredExponential",
                 kernel_train_flag=True, U_ini=None, X_0_ini=None, X_train_ini=None, X_PG=False, PG_particles=100,
                 hyperparameter_sampling=False, kernel_optimization=False, U_optimization=False, U_collapse=False,
                 Z_optimization=False, case_val=1, num_chains=1, device=0, **engine_kw):
        if full_cov:
            raise NotImplementedError("full_cov=True is not used by the GP-SSM path (FFVD_Main.py:266)")
        if len(kernels) != 1:
            raise NotImplementedError("n_layers != 1 (FFVD_Main.py:360 default 1; the nll uses layers[-1] only)")
        Y = np.asarray(Y, dtype=np.float64)
        if Y.ndim == 1:
            Y = Y[:, None]                                      # models.py:44-45
        self.Y = Y
        self.x_dims, self.n_inducing, self.kernels, self.likelihood = x_dims, n_inducing, kernels, likelihood
        self.minibatch_size, self.window_size = minibatch_size, window_size
        self.output_dim = output_dim or x_dims[-1]
        self.kernel_type, self.U_collapse, self.case_val = kernel_type, bool(U_collapse), case_val
        self.prior_type = prior_type
        self.log_Q = np.array(2.0 * np.log(np.asarray(QQ_chol, dtype=np.float64)))     # dgp_model.py:182
        kern = kernels[-1]
        D = self.output_dim
        if len(kern) != D:
            raise ValueError(f"expected one kernel per latent dim ({D}), got {len(kern)}")
        self.layers = [Layer(ZZ, U_ini, X_0_ini, X_train_ini, kern, D, n_inducing, x_dims[-1], Y.shape[0],
                             prior_type=prior_type, kernel_type=kernel_type)]
        self.X_N = self.layers[-1].X.shape[0]
        T = self.X_N - 1
        if control_inputs is None or np.asarray(control_inputs).shape[0] == 0:
            self.control_inputs = np.zeros((T, 0))
        else:
            self.control_inputs = np.asarray(control_inputs, dtype=np.float64)
        C = self.control_inputs.shape[1]
        self.num_chains = int(num_chains)
        self._X_chains = np.repeat(self.layers[-1].X[None], self.num_chains, axis=0)
        self.engine = ElboEngine(T, D, C, n_inducing, self.num_chains, Ydim=Y.shape[1], kernel_type=kernel_type,
                                 U_collapse=self.U_collapse, prior_type=prior_type, device=device, **engine_kw)
        self.engine.set_data(Y, self.control_inputs)
        self._last = None
        self.epsilon, self.mdecay = epsilon, mdecay
        self.X_PG, self.PG_particles = bool(X_PG), int(PG_particles)
        self.PG_mode = "intent"        # "reference" reproduces the reference's no-op (see gp_x_sampling)
        # which variables SG-HMC samples and which Adam trains (dgp_model.py:213-244):
        # case 4 (collapsed U, kernel/Z optimisation on: what FFVD_Main.py:300-305 sets) leaves `vars` empty
        self.vars = []
        if not kernel_optimization and kernel_train_flag:                       # :224-232
            self.vars += ["logvariance", "loglengthscales"] if kernel_type == "SquaredExponential" else ["logvariance"]
        if not U_optimization and not self.U_collapse:                           # :234-236
            self.vars += ["U"]
        if not Z_optimization:                                                   # :240-242
            self.vars += ["Z"]
        if hyperparameter_sampling:                                              # :244-246
            self.vars += ["log_Q", "CC", "DD", "log_Rchols"]
        # what AdamOptimizer.minimize(nll) (dgp_model.py:303-305) updates = the variables created with trainable=True:
        # X unless X_PG or case 7 (:62-66); U iff U_optimization (:68), Z iff Z_optimization (:69); the kernel
        # hyper-parameters iff kernel_optimization (kernels_multi_output.py:156,160); log_Q unless hyperparameter_sampling
        # or case 7 (:176-184); CC, DD, log_Rchols iff likelihood_traning and not hyperparameter_sampling
        # (likelihoods.py:14-24,47-55).  A variable that is neither trainable nor in `vars` simply stays put.
        train = []
        if not (self.X_PG or case_val == 7):
            train.append("X")
        if Z_optimization:
            train.append("Z")
        if kernel_optimization:
            train += ["logvariance", "loglengthscales"]
        if not hyperparameter_sampling and case_val != 7:
            train.append("log_Q")
        if getattr(likelihood, "trainable", True):
            train += ["CC", "DD", "log_Rchols"]
        if U_optimization and not self.U_collapse:
            train.append("U")
        self._adam_train = tuple(k for k in train if k not in self.vars)
        self._rng = np.random.default_rng()
        self.window = []
        self._resident = False      # device copy of the parameters is current (set by train_hypers)
        self._host_stale = False    # ... and newer than the NumPy attributes

    @property
    def Q(self):
        return np.exp(self.log_Q)                               # dgp_model.py:186

    def set_X(self, X_chains):
        """Latent trajectories to evaluate: (S, T+1, D) (or (T+1, D) for one chain)."""
        if self._host_stale:
            self.pull_parameters()
        self._resident = False
        X = np.asarray(X_chains, dtype=np.float64)
        if X.ndim == 2:
            X = X[None]
        if X.shape != self._X_chains.shape:
            raise ValueError(f"X: expected {self._X_chains.shape}, got {X.shape}")
        self._X_chains = X
        self.layers[-1].X = X[0]

    def parameters(self):
        """The parameter dictionary the engine consumes, gathered from the reference-named attributes."""
        lay = self.layers[-1]
        _, _, logvar, loglen = stack_hypers(lay.kernel)
        return dict(X=self._X_chains, Z=lay.Z, U=lay.U, logvariance=logvar, loglengthscales=loglen,
                    log_Q=self.log_Q, CC=self.likelihood.CC, DD=self.likelihood.DD,
                    log_Rchols=self.likelihood.log_Rchols)

    def nll_terms(self):
        """nll and the named component tensors of dgp_model.py:264-297 (mean over chains)."""
        self._last = self.engine.nll_terms(None if self._resident else self.parameters())
        return self._last

    def nll(self):
        return self.nll_terms()["nll"]

    def print_sample_performance(self):
        """Counterpart of BaseModel.print_sample_performance (base_model.py:952-989): the component breakdown."""
        t = self.nll_terms()
        for k, v in t.items():
            if k != "nll_per_chain":
                print(f"{k}: {v}")
        return t

    # ---- training loop pieces (models.py:142-182 drives these) -----------------------------------------
    def get_minibatch(self, global_step=1):
        """BaseModel.get_minibatch (base_model.py:188-194): full batch, lr = 0.003 * 0.95^(global_step/1000)."""
        return [0, self.X_N], 0.003 * (0.95 ** (global_step / 1000))

    def nll_and_grad(self):
        """nll terms + d nll / d variables (what tf.gradients(nll, vars), base_model.py:148, hands the optimisers)."""
        return self.engine.nll_and_grad(None if self._resident else self.parameters())

    def seed(self, seed):
        """Seed the generator behind the SG-HMC noise draws and the window choice of train_hypers (the reference
        draws both unseeded, base_model.py:169,948)."""
        self._rng = np.random.default_rng(seed)

    def _noise(self):
        shapes = {"Z": self.layers[-1].Z.shape, "logvariance": (self.output_dim,),
                  "loglengthscales": (self.output_dim, self.engine.P), "log_Q": (self.output_dim,),
                  "CC": self.likelihood.CC.shape, "DD": self.likelihood.DD.shape,
                  "log_Rchols": self.likelihood.log_Rchols.shape, "U": self.layers[-1].U.shape}
        return {k: self._rng.standard_normal(shapes[k]) for k in self.vars}

    def _ensure_resident(self):
        if not self._resident:
            self.engine.set_params(self.parameters())
            self._resident = True

    def sghmc_step(self):
        """BaseModel.sghmc_step (base_model.py:915-933): burn_in_op, then 10 x (burn_in_op, sample_op) on
        `self.vars` -- each one forward + backward + SG-HMC update on the device -- and the current values of the
        variables join the window.  With an empty variable list (cases 1, 4, 6) the ops are no-ops; cases 2 and 3 sample
        the kernel hyper-parameters, U (and Z), case 5 the kernel hyper-parameters."""
        if not self.vars:
            self.window.append({})
        else:
            self._ensure_resident()
            self.engine.sghmc_step(self._noise(), self.epsilon, self.mdecay, burn_in=True)            # :919
            for _ in range(10):                                                                       # :920-925
                self.engine.sghmc_step(self._noise(), self.epsilon, self.mdecay, burn_in=True)
                self.engine.sghmc_step(self._noise(), self.epsilon, self.mdecay, burn_in=False)
            self._host_stale = True
            g = self.engine.get_params()
            self.window.append({k: g[k].copy() for k in self.vars})                                   # :927-930
        if len(self.window) > self.window_size:
            self.window = self.window[-self.window_size:]                                             # :931-933

    def train_hypers(self):
        """BaseModel.train_hypers (base_model.py:944-950): one Adam step on nll w.r.t. every variable that SG-HMC
        does not sample, with the sampled ones fed from a random window entry for this step only (:948-949);
        forward + backward + update on the device.  Returns the nll terms before the update."""
        self._ensure_resident()
        _, lr = self.get_minibatch()
        fed = self.window[int(self._rng.integers(len(self.window)))] if (self.vars and self.window) else {}
        if fed:
            cur = self.engine.get_params()
            self.engine.update_params(fed)
        t = self.engine.adam_step(lr, train=self._adam_train)
        if fed:
            self.engine.update_params({k: cur[k] for k in fed})        # feed_dict does not assign: restore the chain state
        self._host_stale = True
        return t

    def gp_x_sampling(self, mode=None):
        """`gp_x_sampling()` of the training loop (models.py:156-158) = `pg_x_sampling_op = PG_for_X_speedup(...)`
        (dgp_model.py:309, base_model.py:78-138).

        mode "reference": exactly what the reference's op does -- nothing: its TensorArray writes are discarded (:115)
        and the final assign is never run (:137), X stays as it is.
        mode "intent" (default): one particle-Gibbs sweep per chain on the device (`ffvd_op_pg_sweep`):
        PG_particles - 1 free particles from N(0, I) (:79) advanced with the explicit-U conditional (:93-101),
        weighted against Y (:105-109), resampled together with the current trajectory as the conditioned particle
        (:111-115); a uniformly drawn slot replaces X unless it is the conditioned one (:135-137).
        Returns the number of chains whose trajectory was replaced."""
        mode = mode or self.PG_mode
        if mode == "reference":
            return 0
        if mode != "intent":
            raise ValueError("PG mode must be 'reference' or 'intent'")
        if self.U_collapse:
            raise ValueError("PG_for_X_speedup conditions on the explicit U (base_model.py:96): U_collapse must be False")
        from . import conditionals_multi_output as cmo
        from .prediction import pg_sweep
        if self._host_stale:
            self.pull_parameters()
        lay = self.layers[-1]
        n1 = self.PG_particles - 1
        if n1 < 1:
            raise ValueError("PG_particles must be at least 2")
        T, D = self.X_N - 1, self.output_dim
        Lm = cmo.kernel_pre_cal(lay.Z, lay.kernel)                                             # :81
        Rch = np.exp(np.asarray(self.likelihood.log_Rchols, dtype=np.float64))                  # likelihood.Rchols
        Rch = np.tril(Rch) if Rch.shape[0] > 1 else Rch
        Q = np.exp(self.log_Q)
        replaced = 0
        Xc = np.array(self._X_chains, dtype=np.float64, copy=True)
        for s_ in range(self.num_chains):
            x0 = self._rng.standard_normal((n1, D))                                           # :79
            eps = self._rng.standard_normal((T, n1, D))                                       # :101
            unif = self._rng.random((T, n1))                                                  # :113
            parts, _ = pg_sweep(Lm, lay.Z, lay.kernel, lay.U, Xc[s_], self.Y, self.control_inputs, self.likelihood.CC,
                                self.likelihood.DD, Rch, Q, x0, eps, unif)
            final_index = int(self._rng.integers(self.PG_particles))                          # :135
            if final_index < n1:                                                              # :136-137
                Xc[s_] = parts[:, final_index]
                replaced += 1
        if replaced:
            self._X_chains = Xc
            lay.X = Xc[0]
            if self._resident:
                self.engine.update_params({"X": Xc})
        return replaced

    def pull_parameters(self):
        """Copy the trained parameters from the device back into the reference-named attributes."""
        g = self.engine.get_params()
        lay = self.layers[-1]
        self._X_chains = g["X"]
        lay.X, lay.Z = g["X"][0], g["Z"]
        if not self.U_collapse:
            lay.U = g["U"]
        for d, k in enumerate(lay.kernel):
            k.logvariance = np.float64(g["logvariance"][d])
            if hasattr(k, "loglengthscales"):
                k.loglengthscales = g["loglengthscales"][d].copy()
        self.log_Q = g["log_Q"]
        self.likelihood.CC, self.likelihood.DD, self.likelihood.log_Rchols = g["CC"], g["DD"], g["log_Rchols"]
        self._host_stale = False
        return g

    def sample_op(self, noise=None):
        """One `sample_op` of generate_update_step (base_model.py:171-179) on `self.vars`: forward + backward + SG-HMC
        update of theta and p on the device.  `noise`: name -> standard-normal array (default: the model's generator)."""
        self._ensure_resident()
        t = self.engine.sghmc_step(self._noise() if noise is None else noise, self.epsilon, self.mdecay, burn_in=False)
        self._host_stale = True
        return t

    def collect_samples_formal(self, num, spacing, control_inputs, test_len, sghmc_var_len=0, U_collapse=None,
                               Y_test=None, Y_train_std=1.0, save_path_file=None, Y_train=None, case="C1", ll_seq=(0.0,),
                               running_time_seq=(0.0,), PG_num=None, synthetic_data_function_plot=False, data_uu=None,
                               eps=None, seed=None, rollout_mode="reference"):
        """BaseModel.collect_samples_formal (base_model.py:197-517), called as FFVD_Main.py:345-349 does: K_uu factors,
        posterior U (collapsed branch), `num` rollouts of `test_len` steps from X[-1], predictive y mean / variance, the
        RMSE over the first 30 test points and, with `save_path_file`, the `_results.npz` of :512-517.

        sghmc_var_len > 0 (cases 2, 3, 5: `len(model.vars)`): before rollout num_i the reference runs `spacing` x
        `sample_op` (:223-231) and records the variables (:239-240).  rollout_mode:
          "reference" -- what the reference's graph computes: its rollout tensors are only EVALUATED by the one
                         session.run at :326-327, after the last sample_op, so every rollout sees the final variable
                         values; the `num` rollouts differ by their noise only;
          "intent"    -- rollout num_i uses the variables as they are after its own sample_ops (K_uu factors and
                         posterior U recomputed per sample), which is what the loop sets out to do.
        `eps` (test_len, num, D) injects the standard-normal draws of :306 (else numpy's default_rng(seed)); the
        SG-HMC noise comes from the model's generator (`seed()`).  Returns a dict and sets the reference's attributes
        (fit_x, predict_y, predict_y_var, fit_y, RMSE_val)."""
        from . import conditionals_multi_output as cmo
        from .prediction import predict_y_summary, rollout
        if synthetic_data_function_plot:
            raise NotImplementedError("synthetic_data_function_plot: plotting aid of the kink toy problem, not on the GP-SSM path")
        if rollout_mode not in ("reference", "intent"):
            raise ValueError("rollout_mode must be 'reference' or 'intent'")
        if sghmc_var_len and sghmc_var_len != len(self.vars):
            raise ValueError(f"sghmc_var_len = {sghmc_var_len} but the model samples {len(self.vars)} variables")
        if self._host_stale:
            self.pull_parameters()
        U_collapse = self.U_collapse if U_collapse is None else bool(U_collapse)
        D = self.output_dim
        ci = self.control_inputs if control_inputs is None else np.asarray(control_inputs, dtype=np.float64)
        T = self.X_N - 1
        self.fit_x = self.layers[-1].X.copy()                                                  # :203
        if eps is None:
            eps = np.random.default_rng(seed).standard_normal((test_len, num, D))
        eps = np.asarray(eps, dtype=np.float64)
        n_train = self.Y.shape[0] if Y_train is None else np.asarray(Y_train).shape[0]

        def posterior():
            lay = self.layers[-1]
            Lm = cmo.kernel_pre_cal(lay.Z, lay.kernel)                                          # :207 / :234
            if U_collapse:
                xc = np.concatenate((lay.X[:T], ci[:T]), axis=1) if ci.shape[1] > 0 else lay.X[:-1]    # :243-246
                U_val, U_chol = cmo.collapse_u_mean_after_kernel_precalculation(Lm, xc, lay.X, lay.Z, lay.kernel, self.Q)  # :251
            else:
                U_val, U_chol = lay.U, None                                                     # :255-256
            return Lm, U_val, U_chol

        mc = [[] for _ in self.vars]
        px_parts, pv_parts = [], []
        if sghmc_var_len:
            for num_i in range(num):
                for _ in range(spacing):                                                        # :225-231
                    self.sample_op()
                self.pull_parameters()
                cur = self.parameters()
                for i, k in enumerate(self.vars):
                    mc[i].append(np.array(cur[k], copy=True))                                   # :239-240
                if rollout_mode == "intent":
                    Lm, U_val, U_chol = posterior()
                    lay = self.layers[-1]
                    px, pv = rollout(Lm, lay.Z, lay.kernel, U_val, U_chol, lay.X[-1], ci, n_train, test_len, self.Q,
                                     eps[:, num_i:num_i + 1])
                    px_parts.append(px)
                    pv_parts.append(pv)
        if px_parts:
            px, pv = np.concatenate(px_parts, axis=0), np.concatenate(pv_parts, axis=0)
            U_val = U_val if U_collapse else self.layers[-1].U
        else:
            Lm, U_val, U_chol = posterior()
            lay = self.layers[-1]
            px, pv = rollout(Lm, lay.Z, lay.kernel, U_val, U_chol, lay.X[-1], ci, n_train, test_len, self.Q, eps)
        lay = self.layers[-1]
        out = predict_y_summary(px, pv, self.likelihood.CC, self.likelihood.DD, self.likelihood.log_Rchols, Y_test,
                                Y_train_std)
        self.predict_y, self.predict_y_var = out["predict_y"], out["predict_y_var"]
        self.fit_y = (np.asarray(self.fit_x)[1:] @ self.likelihood.CC + self.likelihood.DD).reshape(-1)   # :344
        if "RMSE" in out:
            self.RMSE_val = out["RMSE"]
            print("RMSE: ", self.RMSE_val)                                                      # :350
        out.update(predict_x=px, predict_x_var=pv, U_val=U_val, fit_y=self.fit_y,
                   mc_posterior_samples={k: np.stack(v) for k, v in zip(self.vars, mc) if v})
        if save_path_file is not None:                                                         # :486-517
            from .data_io import save_results
            out["results_file"] = save_results(save_path_file, self, Y_test, Y_train, Y_train_std, case=case, ll_seq=ll_seq,
                                               running_time_seq=running_time_seq, PG_num=PG_num,
                                               mc_posterior_samples=out["mc_posterior_samples"])
        return out
