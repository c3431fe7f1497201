"""CPU oracle for the FFVD per-iteration ELBO (`nll`) hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy fp64 restatement, op by op and in the reference's own
order, of the arithmetic that the reference (xuhuifan/FFVD) delegates to
TensorFlow on the path named by BASELINE.json `north_star`.
Citations are `file:line` into the reference tree (`vfegpssm/...`).

Status of the pin: **parity unpinned**.  The reference ships no tests, no golden
vectors and no known-answer fixtures for this path, and it cannot be executed
here or on the GPU box (TensorFlow / tensorflow_probability are not installed
and cannot be fetched; `vfegpssm/quadrature.py:16` also fails to import on
Python >= 3.10).  The oracle is therefore pinned only by (i) line-cited
restatement, (ii) agreement with an independently written torch-CPU-fp64
restatement (`oracle/ffvd_oracle_torch.py`) and (iii) the anchor values the
survey recorded for the actuator fixture.

Who may import this module: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the checker / reported baseline, never
as the product.  Nothing under `ffvd_amd/` imports it.

TensorFlow op -> NumPy/SciPy mapping used throughout:
  tf.matmul                 -> `@`
  tf.linalg.cholesky        -> np.linalg.cholesky (lower)
  tf.linalg.triangular_solve-> scipy.linalg.solve_triangular
  tf.linalg.solve           -> np.linalg.solve (LU, partial pivoting)
  tf.linalg.logdet          -> 2*sum(log(diag(cholesky(.))))  (TF's definition)
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_triangular

JITTER_MULTI_OUTPUT = 1e-5   # conditionals_multi_output.py:108,159
JITTER_SINGLE = 1e-7         # conditionals.py:101


# --------------------------------------------------------------------------
# L1 primitives: kernels
# --------------------------------------------------------------------------
class SquaredExponential:
    """SE/ARD kernel, log-parameterised (kernels_multi_output.py:140-161, 246-247)."""

    def __init__(self, logvariance, loglengthscales):
        self.logvariance = np.float64(logvariance)
        self.loglengthscales = np.asarray(loglengthscales, dtype=np.float64)
        self.variance = np.exp(self.logvariance)              # :157
        self.lengthscales = np.exp(self.loglengthscales)      # :161

    def scaled_square_dist(self, X, X2=None):
        """Expanded-form squared distance (kernels_multi_output.py:163-182).

        dist = -2 X X2^T + |X|^2 + |X2|^2^T after dividing by lengthscales; it is
        NOT clamped at zero for the SE kernel.
        """
        X = X / self.lengthscales                              # :170
        Xs = np.sum(np.square(X), axis=-1, keepdims=True)      # :171
        if X2 is None:
            dist = -2.0 * (X @ X.T)                            # :174
            dist = dist + (Xs + Xs.T)                          # :175
            return dist
        X2 = X2 / self.lengthscales                            # :178
        X2s = np.sum(np.square(X2), axis=-1, keepdims=True)    # :179
        dist = -2.0 * (X @ X2.T)                               # :180
        dist = dist + (Xs + X2s.T)                             # :181
        return dist

    def K(self, X, X2=None):
        """K = variance * exp(-r2/2) (kernels_multi_output.py:202-214, 246-247)."""
        return self.variance * np.exp(-self.scaled_square_dist(X, X2) / 2.0)

    def Kdiag(self, X):
        """fill(N, variance) (kernels_multi_output.py:199-200)."""
        return np.full(X.shape[:-1], self.variance, dtype=np.float64)


class LinearK:
    """Linear kernel with one scalar variance (kernels.py:250-281, ARD=False)."""

    def __init__(self, logvariance):
        self.logvariance = np.float64(logvariance)
        self.variance = np.exp(self.logvariance)               # kernels.py:265

    def K(self, X, X2=None):
        if X2 is None:                                         # kernels.py:271-274
            return (X * self.variance) @ X.T
        return (X * self.variance) @ X2.T                      # kernels.py:276

    def Kdiag(self, X):
        return np.sum(np.square(X) * self.variance, axis=1)    # kernels.py:278-281


# --------------------------------------------------------------------------
# L1 primitives: likelihood pieces and the MC draw
# --------------------------------------------------------------------------
def predict_mean(X_end, CC, DD):
    """Gaussian.predict_mean: X_end @ C + d (likelihoods.py:76-79)."""
    return X_end @ CC + DD


def logdensity_norm_diag_nonvec(y, ymean, Rchols):
    """Elementwise -0.5((y-ymean)/R)^2 - log R (likelihoods.py:89-93). No log(2 pi)."""
    exp_term = -0.5 * (((y - ymean) / Rchols[None, :]) ** 2)
    log_R = -np.log(Rchols)[None, :]
    return exp_term + log_R


def logdensity_norm_diag(y, ymean, Rchols):
    """Per-row -0.5 sum_j ((y-ymean)/R_j)^2 - sum_j log R_j (likelihoods.py:96-111)."""
    exp_term = -0.5 * np.sum(((y - ymean) / Rchols[None, :]) ** 2, axis=1)
    log_R = -np.sum(np.log(Rchols))
    return exp_term + log_R


def logdensity_norm(y, ymean, Rchols):
    """Full-Cholesky Gaussian log density without constants (likelihoods.py:114-127)."""
    alphav = solve_triangular(Rchols, (y - ymean).T, lower=True)
    exp_term = -0.5 * np.sum(np.square(alphav), axis=0)
    logdet_R = -np.sum(np.log(np.diag(Rchols)))
    return exp_term + logdet_R


def get_rand(mean, var, eps):
    """Reparameterised draw mean + eps*sqrt(var) (utils.py:11); eps injected."""
    return mean + eps * np.sqrt(var)


# --------------------------------------------------------------------------
# L2 GP operators (conditionals_multi_output.py)
# --------------------------------------------------------------------------
def base_conditional(Kmn, Kmm, Knn, f, *, white=True):
    """conditionals_multi_output.py:6-70, full_cov=False, q_sqrt=None.

    Returns fmean (N x R) and fvar (N x R)."""
    num_func = f.shape[1]
    Lm = np.linalg.cholesky(Kmm)                               # :28
    A = solve_triangular(Lm, Kmn, lower=True)                  # :34
    fvar = Knn - np.sum(np.square(A), axis=0)                  # :41
    fvar = np.tile(fvar[None, :], (num_func, 1))               # :42
    if not white:
        A = solve_triangular(Lm.T, A, lower=False)             # :46
    fmean = A.T @ f                                            # :48
    return fmean, fvar.T                                       # :65


def conditional(Xnew, X, kern, f, *, white=True, jitter=JITTER_MULTI_OUTPUT):
    """conditionals_multi_output.py:73-120 (full_cov=False, q_sqrt=None).

    One kernel per output column of f; returns mean, var each N x D."""
    num_data = X.shape[0]
    f_mu, f_var = [], []
    for kk in range(len(kern)):
        Kmm = kern[kk].K(X) + np.eye(num_data) * jitter        # :108
        Kmn = kern[kk].K(X, Xnew)                              # :109
        Knn = kern[kk].Kdiag(Xnew)                             # :113
        mu_k, var_k = base_conditional(Kmn, Kmm, Knn, f[:, kk][:, None], white=white)
        f_mu.append(mu_k)
        f_var.append(var_k)
    return np.asarray(f_mu)[:, :, 0].T, np.asarray(f_var)[:, :, 0].T   # :120


def kernel_pre_cal(X, kern, jitter=JITTER_MULTI_OUTPUT):
    """Per dim: L = chol(K(Z)+jitter I); returns L^{-T} (conditionals_multi_output.py:124-169)."""
    num_data = X.shape[0]
    out = []
    for kk in range(len(kern)):
        Kmm = kern[kk].K(X) + np.eye(num_data) * jitter        # :159
        Lm = np.linalg.cholesky(Kmm)                           # :162
        out.append(solve_triangular(Lm.T, np.eye(num_data), lower=False))   # :166
    return out


def collapse_after_kernel_precalculation(Lm_inverse_seq, X_combine, X, Z, kern, Q, batch_size, Y_N):
    """Collapsed-U ELBO terms (conditionals_multi_output.py:230-257)."""
    term1 = 0.0
    term2 = 0.0
    trace_Q_inverse_B = 0.0
    M = Z.shape[0]
    for dd in range(len(kern)):
        Knm = kern[dd].K(X_combine, Z)                                          # :240
        tilde_F = Knm @ Lm_inverse_seq[dd]                                      # :242
        Knn_diag = kern[dd].Kdiag(X_combine)                                    # :244
        H = (tilde_F.T @ tilde_F) / (batch_size * Q[dd]) * Y_N + np.eye(M)      # :246
        X_t = (X[1:, dd] - X[:-1, dd])[None, :]                                 # :247
        b = (X_t @ tilde_F) / (batch_size * Q[dd]) * Y_N                        # :248
        logdet = 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(H))))           # tf.linalg.logdet
        term1 += -0.5 * logdet                                                  # :253
        term2 += 0.5 * (b @ np.linalg.solve(H, b.T))[0, 0]                      # :254
        trace_Q_inverse_B += -0.5 * np.sum((Knn_diag - np.sum(tilde_F ** 2, axis=1)) / Q[dd])   # :255
    return -term1 / Y_N, -term2 / Y_N, -trace_Q_inverse_B / Y_N                 # :257


def collapse_u_mean_after_kernel_precalculation(Lm_inverse_seq, X_combine, X, Z, kern, Q):
    """Posterior mean of whitened U and H^{-1/2} stack (conditionals_multi_output.py:206-227)."""
    M = Z.shape[0]
    U_mean, Linv_seq = [], []
    for dd in range(len(kern)):
        Knm = kern[dd].K(X_combine, Z)                                          # :212
        tilde_F = Knm @ Lm_inverse_seq[dd]                                      # :213
        H = (tilde_F.T @ tilde_F) / Q[dd] + np.eye(M)                           # :215
        X_t = (X[1:, dd] - X[:-1, dd])[None, :]                                 # :216
        b = (X_t @ tilde_F) / Q[dd]                                             # :217
        U_mean.append(np.linalg.solve(H, b.T))                                  # :219
        Lm_dd = np.linalg.cholesky(H)                                           # :221
        Linv_seq.append(solve_triangular(Lm_dd.T, np.eye(M), lower=False))      # :222
    return np.stack(U_mean)[:, :, 0].T, np.stack(Linv_seq)                      # :227 (M x D after transpose)


def base_conditional_after_kernel_precalculation(Kmn, Lm_inverse_kk, Knn, f, *, q_sqrt=None, white=True):
    """conditionals_multi_output.py:324-387 (full_cov=False)."""
    num_func = f.shape[1]
    A = Lm_inverse_kk.T @ Kmn                                   # :349
    fvar = Knn - np.sum(np.square(A), axis=0)                   # :356
    fvar = np.tile(fvar[None, :], (num_func, 1))                # :357
    if not white:
        A = Lm_inverse_kk.T @ A                                 # :362
    fmean = A.T @ f                                             # :365
    if q_sqrt is not None:
        if q_sqrt.ndim == 2:
            LTA = A * q_sqrt.T[:, :, None]                      # :369
        else:
            A_tiled = np.tile(A[None, :, :], (num_func, 1, 1))  # :372
            LTA = np.swapaxes(q_sqrt, -1, -2) @ A_tiled         # :373  (R x M x N)
        fvar = fvar + np.sum(np.square(LTA), axis=1)            # :380
    return fmean, fvar.T                                        # :383


def conditional_after_kernel_precalculation(Lm_inverse_seq, Xnew, Z, kern, f, *, q_sqrt=None, white=True):
    """conditionals_multi_output.py:306-322.

    Quirk preserved: when q_sqrt is a D x M x M stack the whole
    stack is handed to every dim (:317) and `[:, :, 0]` (:322) then keeps the
    d=0 slice, so all dims get the d=0 posterior covariance inflation."""
    f_mu, f_var = [], []
    for kk in range(len(kern)):
        Kmn = kern[kk].K(Z, Xnew)                               # :311
        Knn = kern[kk].Kdiag(Xnew)                              # :315
        mu_k, var_k = base_conditional_after_kernel_precalculation(
            Kmn, Lm_inverse_seq[kk], Knn, f[:, kk][:, None], q_sqrt=q_sqrt, white=white)
        f_mu.append(mu_k)
        f_var.append(var_k)
    return np.asarray(f_mu)[:, :, 0].T, np.asarray(f_var)[:, :, 0].T


def rollout(Lm_inverse_seq, Z, kern, U_val, q_sqrt, x_last, control_inputs, ctrl_offset, steps, Q, eps):
    """The prediction loop of collect_samples_formal (base_model.py:288-314) for R = eps.shape[1] posterior
    rollouts advanced side by side (the reference runs them one after another; they only differ by the noise).

    x_last: (D,) = layers[-1].X[-1] (:226, pre_index = 1); control_inputs: (n, C) with row ctrl_offset + test_i fed
    at step test_i (:293, ctrl_offset = Y_train.shape[0]); eps: (steps, R, D) replaces tf.random.normal (:304).
    Returns predict_x (R, steps, D) and predict_x_var (R, steps, D) = f_var + Q (:311)."""
    steps, R, D = eps.shape
    x_t = np.repeat(np.asarray(x_last, dtype=np.float64)[None, :], R, axis=0)
    px, pv = np.zeros((R, steps, D)), np.zeros((R, steps, D))
    has_c = control_inputs is not None and control_inputs.shape[1] > 0
    for ti in range(steps):
        if has_c:                                                                       # :292-295
            xc = np.concatenate((x_t, np.repeat(control_inputs[ctrl_offset + ti][None, :], R, axis=0)), axis=1)
        else:
            xc = x_t
        f_mu, f_var = conditional_after_kernel_precalculation(Lm_inverse_seq, xc, Z, kern, U_val, q_sqrt=q_sqrt)   # :300
        f_mu = f_mu + x_t                                                               # :304 identity mean function
        x_next = f_mu + eps[ti] * np.sqrt(f_var + Q[None, :])                           # :306
        px[:, ti], pv[:, ti] = x_next, f_var + Q[None, :]                               # :313-314
        x_t = x_next
    return px, pv


def predict_y_summary(predict_x, predict_x_var, CC, DD, log_Rchols, Y_test=None, Y_train_std=1.0):
    """base_model.py:330-347: predictive mean / variance of y from the stacked rollouts, RMSE over the first 30."""
    predict_y = (np.mean(np.einsum("ijk,kl->ijl", predict_x, CC), axis=0) + DD[None, :]).reshape(-1)          # :341
    predict_y_var = np.mean(np.einsum("ijk,kl->ijl", predict_x_var, CC ** 2), axis=0).reshape(-1) + np.exp(2 * log_Rchols)   # :342
    out = {"predict_y": predict_y, "predict_y_var": np.asarray(predict_y_var).reshape(-1)}
    if Y_test is not None:
        y30, p30 = np.asarray(Y_test)[:30].reshape(-1), predict_y[:30]                                        # :346-347
        out["RMSE"] = float(np.sqrt(np.mean((y30 - p30) ** 2)) * Y_train_std)                                 # :348
    return out


# --------------------------------------------------------------------------
# L3: priors + nll assembly (dgp_model.py)
# --------------------------------------------------------------------------
def make_kernels(params, kernel_type="SquaredExponential"):
    """D kernels from log-parameters (models.py:57-59); LinearK list per SURVEY Appendix B item 2."""
    D = params["logvariance"].shape[0]
    if kernel_type == "SquaredExponential":
        return [SquaredExponential(params["logvariance"][d], params["loglengthscales"][d]) for d in range(D)]
    if kernel_type == "LinearK":
        return [LinearK(params["logvariance"][d]) for d in range(D)]
    raise ValueError("Invalid kernel type")


def prior_Z(Z, prior_type="normal"):
    """Layer.prior_Z (dgp_model.py:105-121): 'uniform' -> 0, 'normal' -> -|Z|^2/2."""
    if prior_type == "uniform":
        return 0.0
    if prior_type == "normal":
        return -np.sum(np.square(Z)) / 2.0
    raise ValueError("Invalid prior type")


# dgp_model.py:127 centres the SquaredExponential prior at tf.cast(tf.math.log(0.05), tf.float64): the logarithm of a Python
# float is taken in float32 (TF's default dtype for it) and only then widened -- -2.995732307434082, not the fp64
# -2.995732273553991.  The LinearK line (:130) uses np.log(0.05), i.e. the fp64 value.
LOG_PRIOR_VARIANCE_SE = float(np.float64(np.log(np.float32(0.05))))
LOG_PRIOR_VARIANCE_LIN = float(np.log(0.05))


def prior_hyper(kern, kernel_type="SquaredExponential"):
    """Layer.prior_hyper (dgp_model.py:123-130).  LinearK: the reference indexes a
    single kernel object (broken wiring); the list form sums the same expression
    over the D kernels."""
    val = 0.0
    if kernel_type == "SquaredExponential":
        for k in kern:
            val += -np.sum(np.square(k.loglengthscales)) / 2.0 \
                   - np.sum(np.square(k.logvariance - LOG_PRIOR_VARIANCE_SE)) / 2.0
        return val
    for k in kern:
        val += -np.sum(np.square(k.logvariance - LOG_PRIOR_VARIANCE_LIN)) / 2.0
    return val


def prior_U(U):
    """Layer.prior_U choice 1 (dgp_model.py:132-135)."""
    return -0.5 * np.sum(np.square(U))


def hyperparameter_prior(log_Q, CC, DD, log_Rchols):
    """DGPSSM.hypaparameter_prior (dgp_model.py:326-334), log_Q_variance = 1."""
    return (-np.sum(np.square(log_Q)) / 2.0 - np.sum(np.square(CC)) / 2.0
            - np.sum(np.square(DD)) / 2.0 - np.sum(np.square(log_Rchols)) / 2.0)


def regularizer(X_batch, control_inputs_batch, Z, kern, U, Q):
    """DGPSSM.regularizer (dgp_model.py:337-359): branch-A transition terms per t."""
    if control_inputs_batch is not None and control_inputs_batch.shape[0] > 0:
        x_comb = np.concatenate((X_batch[:-1], control_inputs_batch), axis=1)   # :340
    else:
        x_comb = X_batch[:-1]
    mean_reg, var_reg = conditional(x_comb, Z, kern, U, white=True)             # :343
    mean_reg = mean_reg + X_batch[:-1]                                          # :346
    reg_trace = -0.5 * np.sum((Q[None, :] ** (-1)) * var_reg, axis=1)           # :348
    reg_x_prior = logdensity_norm_diag(X_batch[1:], mean_reg, Q ** 0.5)         # :351
    return reg_trace, reg_x_prior


TERM_NAMES_B = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q",
                "nll_reg_trace_inverse_Q_B", "later_term1", "later_term2")
TERM_NAMES_A = ("nll_part_prior", "nll_log_likelihood", "x_t_prior_Q",
                "nll_reg_trace_inverse_Q_B")


def nll_terms(params, Y, control_inputs, *, U_collapse=True, kernel_type="SquaredExponential",
              prior_type="normal"):
    """`DGPSSM.nll` and its named component tensors for ONE latent trajectory
    (dgp_model.py:248-297), full batch: batch_placeholder = [0, X_N].

    params: X (T+1,D), Z (M,P), U (M,D), logvariance (D,), loglengthscales (D,P),
            log_Q (D,), CC (D,Ydim), DD (Ydim,), log_Rchols (Ydim,Ydim)
    Y: (T,Ydim); control_inputs: (>=T, C) or None.
    Returns dict with the component names of the reference plus 'nll'."""
    X = params["X"]
    X_N = X.shape[0]
    b0, b1 = 0, X_N                                                     # base_model.py:188-194
    kern = make_kernels(params, kernel_type)
    Q = np.exp(params["log_Q"])                                         # dgp_model.py:186
    Rchols = np.exp(params["log_Rchols"])                               # likelihoods.py:55

    y_mean = predict_mean(X[b0 + 1:b1], params["CC"], params["DD"])     # :248
    log_lik = logdensity_norm_diag(Y[b0:b1 - 1], y_mean, Rchols[0])     # :250
    prior_x_0 = -np.sum(np.square(X[0])) / 2.0                          # :252
    if control_inputs is not None and control_inputs.shape[0] > 0:
        c_batch = control_inputs[b0:b1 - 1]                             # :255
    else:
        c_batch = None
    hyp_prior = hyperparameter_prior(params["log_Q"], params["CC"], params["DD"], params["log_Rchols"])  # :259
    batch_size = float(b1 - b0 - 1)                                     # :261
    Y_N = float(X_N - 1)                                                # :262
    out = {}
    out["nll_log_likelihood"] = -np.sum(log_lik) / batch_size           # :264

    if U_collapse:
        if c_batch is not None:
            x_comb = np.concatenate((X[b0:b1 - 1], c_batch), axis=1)    # :269
        else:
            x_comb = X[b0:b1][:-1]                                      # :271
        Linv_seq = kernel_pre_cal(params["Z"], kern)                    # :273
        t1, t2, tr = collapse_after_kernel_precalculation(
            Linv_seq, x_comb, X[b0:b1], params["Z"], kern, Q, batch_size, Y_N)   # :275-280
        out["later_term1"], out["later_term2"], out["nll_reg_trace_inverse_Q_B"] = t1, t2, tr
        out["x_t_prior_Q"] = -np.sum(logdensity_norm_diag_nonvec(
            X[b0 + 1:b1], X[b0:b1 - 1], Q ** 0.5)) / batch_size         # :283-284
        out["nll_part_prior"] = -(prior_hyper(kern, kernel_type) + prior_Z(params["Z"], prior_type)
                                  + prior_x_0 + hyp_prior) / Y_N        # :286
        out["nll"] = (out["nll_part_prior"] + out["nll_log_likelihood"] + out["x_t_prior_Q"]
                      + out["nll_reg_trace_inverse_Q_B"] + out["later_term1"] + out["later_term2"])  # :288
    else:
        reg_trace, reg_x_prior = regularizer(X[b0:b1], c_batch, params["Z"], kern, params["U"], Q)   # :290
        out["nll_reg_trace_inverse_Q_B"] = -np.sum(reg_trace) / batch_size      # :292
        out["x_t_prior_Q"] = -np.sum(reg_x_prior) / batch_size                  # :294
        prior_layer = prior_U(params["U"]) + prior_hyper(kern, kernel_type) + prior_Z(params["Z"], prior_type)  # :142-143
        out["nll_part_prior"] = -(prior_layer + prior_x_0 + hyp_prior) / Y_N    # :296
        out["nll"] = (out["nll_part_prior"] + out["nll_log_likelihood"] + out["x_t_prior_Q"]
                      + out["nll_reg_trace_inverse_Q_B"])                       # :297
    return {k: float(v) for k, v in out.items()}


def nll_terms_shard(params, Y, control_inputs, d_begin, d_count, shared_terms, *, U_collapse=True,
                    kernel_type="SquaredExponential", prior_type="normal"):
    """One rank's ADDITIVE share of `nll_terms` when the latent dims are sharded (BASELINE config 5): the
    terms tied to the dims [d_begin, d_begin + d_count) -- every kernel, Cholesky and conditional is per dim,
    conditionals_multi_output.py:107,158,238 -- plus, where `shared_terms`, the terms that are not (likelihood,
    prior_Z, prior_x_0, hyper prior).  Summing the shares of a partition of range(D) gives `nll_terms`.
    Test-side stand-in for a rank's GPU engine in the CPU rehearsals of the multi-GPU path."""
    X = params["X"]
    T = X.shape[0] - 1
    sl = slice(d_begin, d_begin + d_count)
    kern = make_kernels(params, kernel_type)[sl]
    Q = np.exp(params["log_Q"])[sl]
    c_batch = control_inputs[:T] if control_inputs is not None and control_inputs.shape[0] > 0 else None
    x_comb = np.concatenate((X[:-1], c_batch), axis=1) if c_batch is not None else X[:-1]
    out = dict.fromkeys(TERM_NAMES_B, 0.0)
    prior = prior_hyper(kern, kernel_type)
    if shared_terms:
        y_mean = predict_mean(X[1:], params["CC"], params["DD"])
        out["nll_log_likelihood"] = -np.sum(logdensity_norm_diag(Y[:T], y_mean, np.exp(params["log_Rchols"])[0])) / T
        prior += (prior_Z(params["Z"], prior_type) - np.sum(np.square(X[0])) / 2.0
                  + hyperparameter_prior(params["log_Q"], params["CC"], params["DD"], params["log_Rchols"]))
    if U_collapse:
        Linv = kernel_pre_cal(params["Z"], kern)
        t1, t2, tr = collapse_after_kernel_precalculation(Linv, x_comb, X[:, sl], params["Z"], kern, Q, float(T), float(T))
        out["later_term1"], out["later_term2"], out["nll_reg_trace_inverse_Q_B"] = t1, t2, tr
        out["x_t_prior_Q"] = -np.sum(logdensity_norm_diag_nonvec(X[1:, sl], X[:-1, sl], Q ** 0.5)) / T
    else:
        U = params["U"][:, sl]
        mean, var = conditional(x_comb, params["Z"], kern, U, white=True)
        out["nll_reg_trace_inverse_Q_B"] = -np.sum(-0.5 * np.sum((Q[None, :] ** (-1)) * var, axis=1)) / T
        out["x_t_prior_Q"] = -np.sum(logdensity_norm_diag(X[1:, sl], mean + X[:-1, sl], Q ** 0.5)) / T
        prior += prior_U(U)
    out["nll_part_prior"] = -prior / T
    out["nll"] = sum(out[k] for k in TERM_NAMES_B)
    return {k: float(v) for k, v in out.items()}


def tshard_partial(params, Y, control_inputs, t_begin, t_count):
    """One rank's share of the collapsed bound when the TRANSITIONS are sharded (SURVEY 8e last bullet, Appendix A Gram
    route): everything that is a sum over t, for t in [t_begin, t_begin + t_count) -- per latent dim the Gram matrix
    K_uf K_fu and the vector K_uf delta, and the likelihood / transition quadratic sums.  A flat vector, so that the
    exchange is one all-reduce(sum).  Test-side stand-in for a rank's GPU engine in the CPU rehearsals."""
    X = params["X"]
    D = X.shape[1]
    M = params["Z"].shape[0]
    kern = make_kernels(params)
    sl = slice(t_begin, t_begin + t_count)
    c = control_inputs[sl] if control_inputs is not None and control_inputs.shape[0] > 0 else None
    x_comb = np.concatenate((X[:-1][sl], c), axis=1) if c is not None else X[:-1][sl]
    out = []
    for d in range(D):
        Kfu = kern[d].K(x_comb, params["Z"])                                   # conditionals_multi_output.py:240
        delta = (X[1:, d] - X[:-1, d])[sl]                                     # :247
        out += [(Kfu.T @ Kfu).ravel(), Kfu.T @ delta]
    y_mean = predict_mean(X[1:][sl], params["CC"], params["DD"])
    R = np.exp(params["log_Rchols"])[0]
    lik_q = np.sum(-0.5 * np.square((Y[sl] - y_mean) / R))                    # likelihoods.py:100 without the log R part
    Q = np.exp(params["log_Q"])
    xq = np.sum(-0.5 * np.square((X[1:][sl] - X[:-1][sl]) / Q ** 0.5))        # likelihoods.py:91
    out.append(np.array([lik_q, xq, float(t_count)]))
    assert out[0].size == M * M
    return np.concatenate(out)


def tshard_finish(params, reduced, prior_type="normal"):
    """The whole-job nll from the all-reduced `tshard_partial` vectors: log|H| = log|K + G/Q| - log|K|,
    b^T H^-1 b = g^T (K + G/Q)^-1 g / Q^2, sum_t |F_t|^2 = tr(K^-1 G)  (SURVEY Appendix A), K = K_uu + 1e-5 I."""
    X = params["X"]
    D = X.shape[1]
    Z = params["Z"]
    M = Z.shape[0]
    kern = make_kernels(params)
    Q = np.exp(params["log_Q"])
    lik_q, xq, T = reduced[-3:]
    t1 = t2 = tr = 0.0
    for d in range(D):
        off = d * (M * M + M)
        G = reduced[off: off + M * M].reshape(M, M)
        g = reduced[off + M * M: off + M * M + M]
        K = kern[d].K(Z) + JITTER_MULTI_OUTPUT * np.eye(M)
        A = K + G / Q[d]
        LA, LK = np.linalg.cholesky(A), np.linalg.cholesky(K)
        logdetH = 2.0 * np.sum(np.log(np.diag(LA))) - 2.0 * np.sum(np.log(np.diag(LK)))
        y = solve_triangular(LA, g / Q[d], lower=True)
        t1 += -0.5 * logdetH                                                   # :253
        t2 += 0.5 * (y @ y)                                                    # :254
        tr += -0.5 * (T * kern[d].variance - np.trace(np.linalg.solve(K, G))) / Q[d]     # :255
    R = np.exp(params["log_Rchols"])[0]
    out = {}
    out["nll_log_likelihood"] = -(lik_q - T * np.sum(np.log(R))) / T
    out["x_t_prior_Q"] = -(xq - T * np.sum(np.log(Q ** 0.5))) / T
    out["later_term1"], out["later_term2"], out["nll_reg_trace_inverse_Q_B"] = -t1 / T, -t2 / T, -tr / T
    prior = (prior_hyper(kern) + prior_Z(Z, prior_type) - np.sum(np.square(X[0])) / 2.0
             + hyperparameter_prior(params["log_Q"], params["CC"], params["DD"], params["log_Rchols"]))
    out["nll_part_prior"] = -prior / T
    out["nll"] = sum(out[k] for k in TERM_NAMES_B)
    return {k: float(v) for k, v in out.items()}


def nll_terms_chains(params, Y, control_inputs, **kw):
    """Benchmark quantity of SURVEY section 8(d): mean over S chains of nll(X_s).

    params['X'] is (S, T+1, D); every other parameter is shared.  Returns the
    per-term means plus the per-chain nll vector."""
    Xs = params["X"]
    acc = None
    per_chain = []
    for s in range(Xs.shape[0]):
        p = dict(params)
        p["X"] = Xs[s]
        t = nll_terms(p, Y, control_inputs, **kw)
        per_chain.append(t["nll"])
        if acc is None:
            acc = dict(t)
        else:
            for k in t:
                acc[k] += t[k]
    out = {k: v / Xs.shape[0] for k, v in acc.items()}
    out["nll_per_chain"] = np.asarray(per_chain)
    return out
