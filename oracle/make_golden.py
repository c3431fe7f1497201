"""Generate the committed golden fixtures under tests/golden/ -- TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference for the actuator inputs):

    python oracle/make_golden.py

What it writes
  tests/golden/actuator_slim.npz      inputs of BASELINE config 1, reduced to the arrays the
                                      path consumes (data values only; attribution below)
  tests/golden/golden_<name>.npz      oracle outputs (nll + component terms, per chain) for the
                                      actuator fixture and for the seeded synthetic workloads of
                                      ffvd_amd/synthetic.py; inputs are regenerated from the seed.
  tests/golden/ops_small.npz          operator-level vectors (K, Kdiag, kernel_pre_cal, conditional,
                                      collapse terms, posterior U mean) on the 'tiny' workload.

Every value is produced by the NumPy restatement (oracle/ffvd_oracle.py) and accepted only if the
independent torch restatement (oracle/ffvd_oracle_torch.py) agrees to 1e-10 relative; this is NOT
TensorFlow output (parity unpinned -- see the oracle header).

Attribution: `actuator_slim.npz` is derived from xuhuifan/FFVD `data/actuator.mat` (keys u, p) and
`Factnonlin_ini/factnonlin_initialized_10000_actuator_2022_09_06_22_56_20_555556.npz`
(= sorted(glob('*actuator*'))[3]), transformed exactly as FFVD_Main.py:143-168,212-259 and
dgp_model.py:56-58 prescribe.
"""
from __future__ import annotations

import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ffvd_oracle as orc            # noqa: E402
from oracle import ffvd_oracle_torch as orct     # noqa: E402
from ffvd_amd import synthetic                   # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def build_actuator_slim():
    import scipy.io
    mat = scipy.io.loadmat(os.path.join(REF, "data", "actuator.mat"))
    xx = np.asarray(mat["u"], dtype=np.float64)                    # FFVD_Main.py:145
    obs = np.asarray(mat["p"], dtype=np.float64)                   # FFVD_Main.py:146
    control_inputs = (xx - np.mean(xx)) / np.std(xx)               # :157
    lens = obs.shape[0]
    Y_std = np.std(obs[: lens // 2])                               # :162
    Y_mean = np.mean(obs[: lens // 2])                             # :163
    obs = (obs - Y_mean) / Y_std                                   # :165
    Y_train = obs[: lens // 2]                                     # :168
    Y_test = obs[lens // 2:]                                       # :167
    f = sorted(glob.glob(os.path.join(REF, "Factnonlin_ini", "*actuator*")))[3]
    z = np.load(f, allow_pickle=False)
    x_ini = np.mean(z["x_samples_training"], axis=1)               # FFVD_Main.py:226
    T, D = x_ini.shape
    X = np.zeros((T + 1, D))                                       # dgp_model.py:56-58
    X[0] = z["qx1_mu_ini"]
    X[1:] = x_ini
    slim = dict(
        X=X, Z=z["Z_val"], U=z["Umu_ini"].T,                       # FFVD_Main.py:251,339
        logvariance=np.log(z["kernel_variance"]),                  # models.py:59, kernels_multi_output.py:156
        loglengthscales=np.log(z["kernel_lengthscales"]),
        log_Q=2.0 * np.log(z["Q_sqrt_ini"]),                       # dgp_model.py:182
        CC=z["C_val"].T, DD=z["d_val"],                            # FFVD_Main.py:245-246
        log_Rchols=np.log(z["R_chol_val"]),                        # likelihoods.py:54
        Y=Y_train, Y_test=Y_test, control_inputs=control_inputs,
        Y_train_std=Y_std, Y_train_mean=Y_mean,
        source=os.path.basename(f),
    )
    return slim


PARAM_KEYS = ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


def check_pair(name, a, b, rtol=1e-10):
    for k in a:
        if k == "nll_per_chain":
            continue
        va, vb = float(a[k]), float(b[k])
        if abs(va - vb) > rtol * max(1.0, abs(va)):
            raise SystemExit(f"{name}: numpy vs torch oracle disagree on {k}: {va!r} vs {vb!r}")


def torch_terms_chains(params, Y, c, **kw):
    S = params["X"].shape[0]
    acc = None
    for s in range(S):
        p = dict(params)
        p["X"] = params["X"][s]
        tp, tY, tc = orct.to_torch(p, Y, c)
        t = {k: float(v) for k, v in orct.nll_terms(tp, tY, tc, **kw).items()}
        acc = t if acc is None else {k: acc[k] + t[k] for k in t}
    return {k: v / S for k, v in acc.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---- actuator (BASELINE config 1) -----------------------------------
    slim = build_actuator_slim()
    np.savez_compressed(os.path.join(OUT, "actuator_slim.npz"), **slim)
    params = {k: slim[k] for k in PARAM_KEYS}
    gold = {}
    for branch, collapse in (("B", True), ("A", False)):
        t = orc.nll_terms(params, slim["Y"], slim["control_inputs"], U_collapse=collapse)
        tp, tY, tc = orct.to_torch(params, slim["Y"], slim["control_inputs"])
        tt = {k: float(v) for k, v in orct.nll_terms(tp, tY, tc, U_collapse=collapse).items()}
        check_pair(f"actuator/{branch}", t, tt)
        for k, v in t.items():
            gold[f"{branch}_{k}"] = v
        print(f"actuator branch {branch}:", t)
    np.savez(os.path.join(OUT, "golden_actuator.npz"), **gold)

    # ---- seeded synthetic workloads ---------------------------------------
    for name in ("tiny", "small", "ragged", "small_lin"):
        params, Y, c, meta = synthetic.make_named(name)
        kw = dict(U_collapse=meta["U_collapse"], kernel_type=meta["kernel_type"])
        gold = {}
        for branch, collapse in (("B", True), ("A", False)):
            kw["U_collapse"] = collapse
            t = orc.nll_terms_chains(params, Y, c, **kw)
            tt = torch_terms_chains(params, Y, c, **kw)
            check_pair(f"{name}/{branch}", t, tt)
            for k, v in t.items():
                gold[f"{branch}_{k}"] = v
            print(name, branch, {k: v for k, v in t.items() if k != "nll_per_chain"})
        gold["meta_T"], gold["meta_D"], gold["meta_M"], gold["meta_S"] = meta["T"], meta["D"], meta["M"], meta["S"]
        np.savez(os.path.join(OUT, f"golden_{name}.npz"), **gold)

    # ---- operator-level vectors on 'tiny' -----------------------------------
    params, Y, c, meta = synthetic.make_named("tiny")
    kern = orc.make_kernels(params)
    X0 = params["X"][0]
    T = meta["T"]
    xc = np.concatenate((X0[:-1], c[:T]), axis=1)
    Q = np.exp(params["log_Q"])
    Linv = orc.kernel_pre_cal(params["Z"], kern)
    mean, var = orc.conditional(xc, params["Z"], kern, params["U"], white=True)
    t1, t2, tr = orc.collapse_after_kernel_precalculation(Linv, xc, X0, params["Z"], kern, Q, float(T), float(T))
    Um, Hinv = orc.collapse_u_mean_after_kernel_precalculation(Linv, xc, X0, params["Z"], kern, Q)
    mean_pc, var_pc = orc.conditional_after_kernel_precalculation(Linv, xc[:7], params["Z"], kern, Um, q_sqrt=Hinv)
    lin = orc.LinearK(np.log(0.07))
    ops = dict(
        Kuu=np.stack([k.K(params["Z"]) for k in kern]),
        Kfu=np.stack([k.K(xc, params["Z"]) for k in kern]),
        Kdiag=np.stack([k.Kdiag(xc) for k in kern]),
        Klin=lin.K(xc, params["Z"]), Klin_diag=lin.Kdiag(xc),
        Lm_inverse_seq=np.stack(Linv), cond_mean=mean, cond_var=var,
        collapse=np.array([t1, t2, tr]), U_mean=Um, H_inv_sqrt=Hinv,
        precalc_mean=mean_pc, precalc_var=var_pc,
    )
    np.savez_compressed(os.path.join(OUT, "ops_tiny.npz"), **ops)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
