"""Data formats either side of the hot path (SURVEY 8f-4, I/O half) -- host-side NumPy, no TensorFlow.

  create_dataset      the standardisation of FFVD_Main.py:157-171 (control inputs by their own mean/std over the
                      whole series; observations by the mean/std of the TRAINING half; first half train, second test)
  load_init / apply_init   the `Factnonlin_ini/*.npz` keys read at FFVD_Main.py:212-229 and where they land in
                      RegressionModel.ARGS (:245-259)
  save_results        the `_results.npz` keys written by collect_samples_formal (base_model.py:512-517)

File loading of the raw datasets (.dat/.mat/.csv, FFVD_Main.py:138-155) stays with the caller: it needs pandas /
scipy.io and paths the library should not assume.
"""
from __future__ import annotations

import numpy as np

INIT_KEYS = ("qx1_mu_ini", "qx1_cov_chol_ini", "Umu_ini", "Ucov_chol_ini", "Q_sqrt_ini", "kernel_variance",
             "kernel_lengthscales", "C_val", "d_val", "Z_val", "x_samples_training", "R_chol_val")


def create_dataset(xx, observations):
    """(Y_train, Y_test, control_inputs, Y_train_std, Y_train_mean, control_inputs_mean, control_inputs_std) from the
    raw input series `xx` (N, C) and `observations` (N, Ydim), exactly as FFVD_Main.py:157-171."""
    xx = np.asarray(xx, dtype=np.float64)
    observations = np.asarray(observations, dtype=np.float64)
    control_inputs_mean, control_inputs_std = np.mean(xx), np.std(xx)
    control_inputs = (xx - control_inputs_mean) / control_inputs_std                     # :157
    lens = observations.shape[0]                                                          # :160
    Y_train_std = np.std(observations[:int(lens / 2)])                                    # :162
    Y_train_mean = np.mean(observations[:int(lens / 2)])                                  # :163
    observations = (observations - Y_train_mean) / Y_train_std                            # :165
    Y_test = observations[int(lens / 2):]                                                 # :167
    Y_train = observations[:int(lens / 2)]                                                # :168
    return Y_train, Y_test, control_inputs, Y_train_std, Y_train_mean, control_inputs_mean, control_inputs_std


def load_init(path_or_mapping):
    """The arrays FFVD_Main.py:212-229 takes from an initialisation file (a path to an .npz, loaded with
    allow_pickle=False, or any mapping with those keys).  `x_samples_training` is averaged over its sample axis
    (:226); keys the caller does not have may be absent -- only the ones present are returned."""
    src = np.load(path_or_mapping, allow_pickle=False) if isinstance(path_or_mapping, (str, bytes)) else path_or_mapping
    out = {}
    for k in INIT_KEYS:
        if k in src:
            out[k] = np.asarray(src[k], dtype=np.float64)
    if "x_samples_training" in out and out["x_samples_training"].ndim == 3:
        out["x_samples_training_mean"] = np.mean(out["x_samples_training"], axis=1)       # :226
    return out


def apply_init(ARGS, ini, control_inputs, Y_train_std, num_inducing, x_dims):
    """Fill RegressionModel.ARGS from the initialisation arrays the way FFVD_Main.py:245-265 does (NumPy arrays where
    the reference wraps tf.convert_to_tensor)."""
    ARGS.CC = np.asarray(ini["C_val"], dtype=np.float64).T                                # :245
    ARGS.DD = np.asarray(ini["d_val"], dtype=np.float64)                                  # :246
    ARGS.QQ_chol = np.asarray(ini["Q_sqrt_ini"], dtype=np.float64)                        # :247
    ARGS.RR_chol = np.asarray(ini["R_chol_val"], dtype=np.float64)                        # :248
    ARGS.lengthscales = ini["kernel_lengthscales"]                                        # :250
    ARGS.variance = ini["kernel_variance"]                                                # :251
    ARGS.UU_ini = np.asarray(ini["Umu_ini"], dtype=np.float64).T                          # :253
    ARGS.XX_0_ini = ini["qx1_mu_ini"]                                                     # :254
    ARGS.x_initialization = ini.get("x_samples_training_mean", ini.get("x_samples_training"))   # :255
    ARGS.Y_train_std = Y_train_std                                                        # :257
    ARGS.control_inputs = np.asarray(control_inputs, dtype=np.float64)                    # :259
    ARGS.num_inducing = num_inducing
    ARGS.x_dims = list(x_dims)
    ARGS.ZZ = ini["Z_val"]
    return ARGS


def save_results(save_path_file, model, Y_test, Y_train, Y_train_std, case="C1", ll_seq=(0.0,), running_time_seq=(0.0,),
                 PG_num=None, U_val=None, mc_posterior_samples=None):
    """Write `<save_path_file>_results.npz` with the keys of base_model.py:512-517 from a DGPSSM that has run
    collect_samples_formal.  Unlike the reference (FFVD_Main.py:328,348: `results/<dataset>/` must pre-exist) the
    directory is created.  `mc_posterior_samples`: name -> (num, ...) array of the SG-HMC variables recorded per
    rollout (:239-240); stored as numeric arrays `mc_posterior_samples_<name>` (the reference pickles a list of lists
    under `mc_posterior_samples`, which needs allow_pickle to read back).  Returns the file name."""
    import os
    if model._host_stale:
        model.pull_parameters()
    lay = model.layers[-1]
    name = save_path_file + "_results.npz"
    if os.path.dirname(name):
        os.makedirs(os.path.dirname(name), exist_ok=True)
    mc = {f"mc_posterior_samples_{k}": np.asarray(v) for k, v in (mc_posterior_samples or {}).items()}
    np.savez_compressed(
        name, y_train_vfe=model.fit_y, y_test_vfe=model.predict_y, v_test_vfe_var=model.predict_y_var,
        Y_test_data=Y_test, Y_train_data=Y_train, Y_train_std=Y_train_std, CC_val=model.likelihood.CC,
        DD_val=model.likelihood.DD, log_R_cholesky=model.likelihood.log_Rchols, log_QQ=model.log_Q, Z_val=lay.Z,
        U_val=lay.U if U_val is None else U_val, X_val=np.asarray(model.fit_x)[1:],
        k_lengthscales=np.stack([k.loglengthscales for k in lay.kernel]) if hasattr(lay.kernel[0], "loglengthscales")
        else np.zeros(0),
        k_log_variances=np.array([float(k.logvariance) for k in lay.kernel]), case=case, ll_seq=np.asarray(ll_seq),
        running_time_seq=np.asarray(running_time_seq), PG_num=np.asarray(-1 if PG_num is None else PG_num),
        mc_posterior_samples=np.array(sorted((mc_posterior_samples or {}).keys())), **mc)
    return name
