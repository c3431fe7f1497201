"""Host-side data formats (SURVEY 8f-4, I/O half): standardisation, initialisation keys, results file -- no GPU."""
import numpy as np
import pytest

from conftest import GOLDEN
from ffvd_amd import data_io


def test_create_dataset_follows_the_driver():
    """FFVD_Main.py:157-171: inputs standardised over the whole series, outputs by the training half's statistics."""
    rng = np.random.default_rng(0)
    xx, obs = 3.0 + 2.0 * rng.standard_normal((101, 1)), -1.0 + 0.5 * rng.standard_normal((101, 1))
    Ytr, Yte, ci, ystd, ymean, cmean, cstd = data_io.create_dataset(xx, obs)
    assert Ytr.shape == (50, 1) and Yte.shape == (51, 1) and ci.shape == (101, 1)
    assert ymean == pytest.approx(np.mean(obs[:50])) and ystd == pytest.approx(np.std(obs[:50]))
    assert np.mean(Ytr) == pytest.approx(0.0, abs=1e-12) and np.std(Ytr) == pytest.approx(1.0)
    assert np.mean(ci) == pytest.approx(0.0, abs=1e-12) and np.std(ci) == pytest.approx(1.0)
    np.testing.assert_allclose(Yte, (obs[50:] - ymean) / ystd)
    assert (cmean, cstd) == (pytest.approx(np.mean(xx)), pytest.approx(np.std(xx)))


def test_init_keys_land_in_args_like_the_driver():
    """FFVD_Main.py:212-229,245-259 on a mapping with the Factnonlin_ini key names."""
    D, M, T = 4, 7, 20
    rng = np.random.default_rng(1)
    ini_file = {"qx1_mu_ini": rng.standard_normal(D), "Umu_ini": rng.standard_normal((D, M)),
                "Q_sqrt_ini": np.full(D, 0.4), "kernel_variance": np.full(D, 0.5),
                "kernel_lengthscales": np.full((D, D + 1), 2.0), "C_val": rng.standard_normal((1, D)),
                "d_val": np.array([0.05]), "Z_val": rng.standard_normal((M, D + 1)),
                "x_samples_training": rng.standard_normal((T, 9, D)), "R_chol_val": np.array([[0.4]])}
    ini = data_io.load_init(ini_file)
    assert ini["x_samples_training_mean"].shape == (T, D)

    class ARGS:
        pass
    A = data_io.apply_init(ARGS, ini, np.zeros((2 * T, 1)), 1.3, M, [D])
    assert A.CC.shape == (D, 1) and A.UU_ini.shape == (M, D)                 # transposed as at :245,:253
    np.testing.assert_array_equal(A.CC[:, 0], ini_file["C_val"][0])
    np.testing.assert_allclose(A.x_initialization, np.mean(ini_file["x_samples_training"], axis=1))
    assert A.num_inducing == M and A.x_dims == [D] and A.Y_train_std == 1.3


def test_slim_actuator_fixture_is_already_standardised():
    """tests/golden/actuator_slim.npz stores the standardised series: re-standardising them is the identity for the
    control inputs, and the outputs carry the training-half statistics of create_dataset."""
    import os
    z = np.load(os.path.join(GOLDEN, "actuator_slim.npz"), allow_pickle=False)
    ci = z["control_inputs"]
    assert abs(np.mean(ci)) < 0.2 and 0.5 < np.std(ci) < 1.5
