"""Model / RegressionModel facade -- counterpart of vfegpssm/models.py (driver-facing surface).

`RegressionModel(prior_type)` exposes the class-attribute bag `ARGS` that FFVD_Main.py:236-340 fills and a
`fit(Y_train, ...)` that builds the kernels (models.py:57-62), the Gaussian likelihood (models.py:320) and the
DGPSSM (models.py:66-74), then runs the loop of models.py:142-182: `sghmc_step` (a no-op on the empty SG-HMC
variable list of the default collapsed case 4) and `train_hypers` (one Adam step on nll, forward + backward +
update on the device), `ARGS.iterations` x 2 times as the reference does (models.py:142) -- `fit(Y_train, ...)`
called exactly as FFVD_Main.py:343 does trains.  `fit(..., iterations=k)` runs k rounds instead; `iterations=0` only
builds the model and records the initial nll.  The engine's route and backward-pass workspace are chosen here from
`U_collapse` and from whether any round will run; callers never pass them.
"""
from __future__ import annotations

import numpy as np

from .dgp_model import DGPSSM
from .kernels import LinearK, SquaredExponential
from .likelihoods import Gaussian


class Model:
    def __init__(self, prior_type, output_dim=None):
        class ARGS:                                  # models.py:21-32
            num_inducing = 100
            iterations = 10000
            minibatch_size = 10000
            window_size = 64
            num_posterior_samples = 100
            posterior_sample_spacing = 50
            full_cov = False
            n_layers = 1
            prior_type = None
        ARGS.prior_type = prior_type
        self.ARGS = ARGS
        self.model = None
        self.output_dim = output_dim
        if prior_type not in ("determinantal", "normal", "strauss", "uniform"):
            raise Exception("Invalid prior type")    # models.py:35-41

    def _fit(self, Y_train, lik, kernel_type, kernel_train_flag, iterations=None, epsilon=0.01, **kwargs):
        Y_train = np.asarray(Y_train, dtype=np.float64)
        if Y_train.ndim == 1:
            Y_train = Y_train[:, None]
        A = self.ARGS
        n_iter = 2 * A.iterations if iterations is None else int(iterations)     # models.py:142 `2*self.ARGS.iterations`
        if not self.model:
            if n_iter > 0:
                # training needs the backward-pass workspace; the collapsed bound then runs in its Gram form, the one the
                # closed-form gradient is written in
                kwargs.setdefault("grad", True)
                if A.U_collapse:
                    kwargs.setdefault("route", "gram")
            control = np.asarray(A.control_inputs, dtype=np.float64)
            D = A.x_dims[-1]
            Z_dim = control.shape[1] + D                                        # models.py:51
            if kernel_type == "SquaredExponential":
                kern = [SquaredExponential(Z_dim, ARD=True, variance=A.variance[kk], lengthscales=A.lengthscales[kk],
                                           kernel_optimization=A.kernel_optimization) for kk in range(D)]   # :57-59
            elif kernel_type == "LinearK":
                # the reference appends ONE LinearK object, which cannot run;
                # here: D LinearK kernels with variance[kk] each through the list path.
                var = np.broadcast_to(np.asarray(1.0 if A.variance is None else A.variance, dtype=np.float64), (D,))
                kern = [LinearK(Z_dim, ARD=False, variance=float(var[kk])) for kk in range(D)]
            else:
                raise ValueError("Invalid kernel type")
            self.model = DGPSSM(Y_train, A.x_dims, A.num_inducing, [kern], lik, minibatch_size=A.minibatch_size,
                                window_size=A.window_size, full_cov=A.full_cov, prior_type=A.prior_type,
                                output_dim=self.output_dim, QQ_chol=A.QQ_chol, ZZ=A.ZZ, variance=A.variance,
                                lengthscales=A.lengthscales, control_inputs=control, kernel_type=kernel_type,
                                kernel_train_flag=kernel_train_flag, U_ini=A.UU_ini, X_0_ini=A.XX_0_ini,
                                X_train_ini=A.x_initialization, X_PG=getattr(A, "X_PG", False), PG_particles=getattr(A, "PG_particles", 100),
                                hyperparameter_sampling=getattr(A, "hyperparameter_sampling", False),
                                kernel_optimization=getattr(A, "kernel_optimization", True),
                                U_optimization=getattr(A, "U_optimization", False),
                                Z_optimization=getattr(A, "Z_optimization", True),
                                U_collapse=A.U_collapse, case_val=getattr(A, "case_val", 4), epsilon=epsilon,
                                **kwargs)                                                         # models.py:73-74
        elif n_iter > 0 and not self.model.engine.grad:
            raise ValueError("this model was built without the backward-pass workspace (fit(..., iterations=0)); "
                             "build a new RegressionModel to train")
        self.nll_seq, self.rmse_seq, self.ll_seq, self.running_time_seq = [], [], [], []   # models.py:89-92
        self.nll_seq.append(self.model.nll())
        self.global_step = 0
        for it in range(n_iter):
            self.global_step += 1
            self.model.sghmc_step()                                              # models.py:150
            if getattr(A, "X_PG", False):
                self.model.PG_mode = getattr(A, "PG_mode", self.model.PG_mode)
                self.model.gp_x_sampling()                                       # models.py:156-158
            t = self.model.train_hypers()                                        # models.py:168
            self.nll_seq.append(t["nll"])
        if n_iter:
            self.model.pull_parameters()
        return self


class RegressionModel(Model):
    def __init__(self, prior_type, output_dim=None):
        super().__init__(prior_type, output_dim)

    def fit(self, Y_train, Y_test=None, tensorboard_savepath="", dataname="", fileid="",
            kernel_type="SquaredExponential", kernel_train_flag=True, likelihood_traning=True, X_train=None,
            X_test=None, Ystd=None, data_uu=None, epsilon=0.01, iterations=None, **kwargs):
        """models.py:319-322.  `iterations`: number of (sghmc_step, [gp_x_sampling,] train_hypers) rounds; the default
        None = 2 * ARGS.iterations as models.py:142, so the reference's call (FFVD_Main.py:343) trains; 0 builds the model
        and evaluates the initial nll only."""
        Y_train = np.asarray(Y_train, dtype=np.float64)
        if Y_train.ndim == 1:
            Y_train = Y_train[:, None]
        A = self.ARGS
        lik = Gaussian(Y_train.shape[1], A.x_dims[-1], CC=A.CC, DD=A.DD, RR_chol=A.RR_chol,
                       hyperparameter_sampling=getattr(A, "hyperparameter_sampling", False),
                       likelihood_traning=likelihood_traning)                     # models.py:320
        return self._fit(Y_train, lik, kernel_type, kernel_train_flag, iterations=iterations, epsilon=epsilon, **kwargs)
