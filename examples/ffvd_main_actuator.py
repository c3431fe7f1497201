#!/usr/bin/env python3
"""Driver-equivalent of the reference's `FFVD_Main.py:192-351` with the import swap
`from vfegpssm.models import RegressionModel` -> `from ffvd_amd.models import RegressionModel`.

Same sequence as `main(file_path, ini_file)`: dataset + initialisation arrays -> `RegressionModel(prior_type)` ->
the `model.ARGS.*` assignments of :236-340 (the `case_val` table :273-324 verbatim) -> `model.fit(...)` called with the
keyword arguments of :343 -> `model.model.collect_samples_formal(...)` called with those of :345-349 (results file
included).  What differs, and why:
  * inputs come from `tests/golden/actuator_slim.npz` (the standardised actuator series of `data/actuator.mat` and the
    arrays FFVD_Main.py:212-229 reads from `Factnonlin_ini/*actuator*` sorted-index 3, already transformed as :245-259
    does) -- the GPU box has no copy of the reference's data files.  With `--data-mat` / `--ini-file` the script runs
    the reference's own loading path (`ffvd_amd.data_io.create_dataset` / `load_init`, FFVD_Main.py:134-171,212-229);
  * `tf.convert_to_tensor(...)` wrappers become NumPy arrays;
  * `--iterations` defaults to 3 here (the reference's 2000 means 4000 training rounds).

    python examples/ffvd_main_actuator.py --case_val 4 --iterations 3
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from datetime import datetime

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ffvd_amd import data_io                       # noqa: E402
from ffvd_amd.models import RegressionModel        # noqa: E402   (the import swap)

logging.basicConfig(level=logging.INFO)            # FFVD_Main.py:12-13
logger = logging.getLogger(__name__)


def load_inputs(args):
    """(Y_train, Y_test, control_inputs, Y_train_std, ini) as FFVD_Main.py:208-229 produces them."""
    if args.data_mat and args.ini_file:
        import scipy.io
        mat = scipy.io.loadmat(args.data_mat)                                      # FFVD_Main.py:143-146
        Y_train, Y_test, control_inputs, Y_train_std, *_ = data_io.create_dataset(mat["u"], mat["p"])
        return Y_train, Y_test, control_inputs, Y_train_std, data_io.load_init(args.ini_file)
    z = np.load(os.path.join(ROOT, "tests", "golden", "actuator_slim.npz"), allow_pickle=False)
    # the fixture stores the arrays in the model's parameterisation; map them back to the init-file keys (:212-229)
    ini = {
        "C_val": z["CC"].T, "d_val": z["DD"], "Q_sqrt_ini": np.exp(0.5 * z["log_Q"]), "R_chol_val": np.exp(z["log_Rchols"]),
        "kernel_lengthscales": np.exp(z["loglengthscales"]), "kernel_variance": np.exp(z["logvariance"]),
        "Umu_ini": z["U"].T, "qx1_mu_ini": z["X"][0], "x_samples_training_mean": z["X"][1:], "Z_val": z["Z"],
    }
    return z["Y"], z["Y_test"], z["control_inputs"], float(z["Y_train_std"]), ini


def main(args, file_path="actuator/"):
    now = datetime.now()
    fileid = now.strftime("%Y_%m_%d_%H_%M_%S_%f")                                  # :198-199
    Y_train, Y_test, control_inputs, Y_train_std, ini = load_inputs(args)

    model = RegressionModel(args.prior_type)                                        # :232

    data_io.apply_init(model.ARGS, ini, control_inputs, Y_train_std, args.num_inducing, args.x_dims)   # :245-259, :262, :269
    model.ARGS.minibatch_size = args.minibatch_size                                 # :263
    model.ARGS.iterations = args.iterations                                         # :264
    model.ARGS.n_layers = args.n_layers
    model.ARGS.num_posterior_samples = args.samples
    model.ARGS.posterior_sample_spacing = args.posterior_sample_spacing
    model.ARGS.prior_type = args.prior_type
    model.ARGS.full_cov = False
    model.ARGS.case_val = args.case_val
    model.ARGS.hyperparameter_sampling = False                                      # :271

    if model.ARGS.case_val == 1:                                                    # :273-324
        model.ARGS.kernel_optimization, model.ARGS.U_optimization, model.ARGS.Z_optimization = True, True, True
        model.ARGS.U_collapse, case, model.ARGS.X_PG = False, "C1", False
    elif model.ARGS.case_val == 2:
        model.ARGS.kernel_optimization, model.ARGS.U_optimization, model.ARGS.Z_optimization = False, False, True
        model.ARGS.U_collapse, case, model.ARGS.X_PG = False, "C2", False
    elif model.ARGS.case_val == 3:
        model.ARGS.kernel_optimization, model.ARGS.U_optimization, model.ARGS.Z_optimization = False, False, False
        model.ARGS.U_collapse, case, model.ARGS.X_PG = False, "C3", False
    elif model.ARGS.case_val == 4:
        model.ARGS.kernel_optimization, model.ARGS.U_optimization, model.ARGS.Z_optimization = True, False, True
        model.ARGS.U_collapse, case, model.ARGS.X_PG = True, "C4", False
    elif model.ARGS.case_val == 5:
        model.ARGS.kernel_optimization, model.ARGS.U_optimization, model.ARGS.Z_optimization = False, False, True
        model.ARGS.U_collapse, case, model.ARGS.X_PG = True, "C5", False
    elif model.ARGS.case_val == 6:
        model.ARGS.kernel_optimization, model.ARGS.U_optimization, model.ARGS.Z_optimization = True, True, True
        model.ARGS.U_collapse, case, model.ARGS.X_PG = False, "C6", True
    else:
        raise SystemExit("case_val must be 1..6")

    model.ARGS.PG_particles = args.PG_particles                                     # :326 (100 in the reference)
    tensorboard_savepath = args.results_dir                                         # :328 ('results')
    model.ARGS.kink_flag = False
    model.ARGS.posterior_sample_spacing = args.forced_spacing                       # :331 (hard-coded 32 in the reference)
    model.ARGS.kernel_type = args.kernel_type
    model.ARGS.kernel_train_flag = args.kernel_train_flag
    model.ARGS.test_len = len(Y_test) if args.test_len is None else args.test_len   # :334
    fileid += "file_id" + str(args.file_id)                                         # :337

    logger.info("Number of inducing points: %d" % model.ARGS.num_inducing)           # :341
    model.fit(Y_train, Y_test=Y_test, tensorboard_savepath=tensorboard_savepath, dataname=file_path[:-1], fileid=fileid,
              kernel_type=model.ARGS.kernel_type, kernel_train_flag=model.ARGS.kernel_train_flag, epsilon=.01)   # :343

    out = model.model.collect_samples_formal(
        model.ARGS.num_posterior_samples, model.ARGS.posterior_sample_spacing, model.ARGS.control_inputs,
        test_len=model.ARGS.test_len, sghmc_var_len=len(model.model.vars), U_collapse=model.ARGS.U_collapse,
        Y_test=Y_test, Y_train_std=Y_train_std,
        save_path_file=tensorboard_savepath + "/" + file_path[:-1] + "/" + case + "VFE_result_" + file_path[:-1] + "_" + fileid + ".npz",
        Y_train=Y_train, case=case, ll_seq=model.ll_seq, running_time_seq=model.running_time_seq,
        PG_num=model.ARGS.PG_particles)                                             # :345-349
    print("nll:", " ".join(f"{v:.6f}" for v in model.nll_seq))
    print("RMSE:", model.model.RMSE_val, "results:", out.get("results_file"))
    return model, out


def parse(argv=None):
    parser = argparse.ArgumentParser(description="Run FFVD-gpssm experiment (ffvd_amd drop-in)")      # :355-381
    parser.add_argument("--num_inducing", type=int, default=100)
    parser.add_argument("--minibatch_size", type=int, default=1000)
    parser.add_argument("--iterations", type=int, default=3, help="the reference's default is 2000 (x2 training rounds)")
    parser.add_argument("--posterior_sample_spacing", type=int, default=50)
    parser.add_argument("--file_id", type=int, default=3)
    parser.add_argument("--case_val", type=int, default=4)
    parser.add_argument("--x_dims", type=int, nargs="+", default=[4])
    parser.add_argument("--samples", type=int, default=10)
    parser.add_argument("--n_layers", type=int, default=1)
    parser.add_argument("--kernel_type", choices=["SquaredExponential", "LinearK"], default="SquaredExponential")
    parser.add_argument("--kernel_train_flag", type=lambda s: s.lower() not in ("0", "false", "no"), default=True)
    parser.add_argument("--prior_type", choices=["determinantal", "normal", "strauss", "uniform"], default="normal")
    # knobs the reference hard-codes (kept at its values by default)
    parser.add_argument("--PG_particles", type=int, default=100)
    parser.add_argument("--forced_spacing", type=int, default=32)
    parser.add_argument("--test_len", type=int, default=None)
    parser.add_argument("--results_dir", default="results")
    parser.add_argument("--data-mat", default=None, help="path to data/actuator.mat (else the committed fixture)")
    parser.add_argument("--ini-file", default=None, help="path to a Factnonlin_ini/*.npz (else the committed fixture)")
    return parser.parse_args(argv)


if __name__ == "__main__":
    main(parse())
