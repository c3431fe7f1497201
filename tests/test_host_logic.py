"""CPU: host-side logic that needs no device -- sharding plans, partial-sum finishing, kernel objects,
and the world_size-2 gloo rehearsal of the multi-GPU reduction."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from ffvd_amd import distributed as dist_mod
from ffvd_amd import synthetic
from ffvd_amd.kernels import LinearK, SquaredExponential, stack_hypers
from ffvd_amd.likelihoods import Gaussian


def test_shard_range_covers_everything():
    for n in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [dist_mod.shard_range(n, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for b, c in spans:
                assert b == pos
                pos += c
    with pytest.raises(ValueError):
        dist_mod.shard_range(4, 2, 2)


def test_plans():
    meta = dict(S=32, D=4)
    p = dist_mod.plan(meta, 8, 3, "chains")
    assert (p["s_begin"], p["s_count"], p["d_count"], p["shared_terms"]) == (12, 4, 4, True)
    meta = dict(S=1, D=16)
    p0, p5 = dist_mod.plan(meta, 8, 0, "dims"), dist_mod.plan(meta, 8, 5, "dims")
    assert p0["shared_terms"] and not p5["shared_terms"]
    assert (p5["d_begin"], p5["d_count"]) == (10, 2)
    with pytest.raises(ValueError):
        dist_mod.plan(dict(S=1, D=16), 8, 0, "chains")


def test_finish():
    sums = np.array([1.0, 2, 3, 4, 5, 6, 21, 2])
    t = dist_mod.finish(sums)
    assert t["nll"] == 10.5 and t["nll_part_prior"] == 0.5
    with pytest.raises(ValueError):
        dist_mod.finish(np.zeros(8))


def test_kernel_objects_follow_reference_parameterisation():
    k = SquaredExponential(5, variance=0.3, lengthscales=np.arange(1.0, 6.0), ARD=True)
    assert k.logvariance == pytest.approx(np.log(0.3))
    np.testing.assert_allclose(k.lengthscales, np.arange(1.0, 6.0))
    kind, name, lv, ll = stack_hypers([k, k])
    assert kind == 0 and name == "SquaredExponential" and lv.shape == (2,) and ll.shape == (2, 5)
    lin = LinearK(5, variance=0.07)
    assert stack_hypers([lin])[3] is None
    with pytest.raises(ValueError):
        stack_hypers([k, lin])
    with pytest.raises(ValueError):
        LinearK(5, variance=np.ones(5))


def test_gaussian_likelihood_parameters():
    lik = Gaussian(1, 4, CC=np.ones((4, 1)) * 0.5, DD=np.array([0.1]), RR_chol=np.array([[0.4]]))
    assert lik.log_Rchols[0, 0] == pytest.approx(np.log(0.4))
    assert lik.Rchols[0, 0] == pytest.approx(0.4)
    assert Gaussian(1, 4).log_Rchols[0, 0] == pytest.approx(np.log(0.1))      # likelihoods.py:52


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ffvd_amd import synthetic, distributed as dm
from oracle import ffvd_oracle as orc
from ffvd_amd._lib import TERM_NAMES
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode, name = sys.argv[1], sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
params, Y, c, meta = synthetic.make_named(name)
pl = dm.plan(meta, world, rank, mode)
# rank-local partial sums; the ORACLE stands in for the GPU engine (test-only fake backend)
sums = np.zeros(8)
kw = dict(U_collapse=meta["U_collapse"], kernel_type=meta["kernel_type"])
for s in range(pl["s_begin"], pl["s_begin"] + pl["s_count"]):
    p = dict(params); p["X"] = params["X"][s]
    if mode == "dims":
        t = orc.nll_terms_shard(p, Y, c, pl["d_begin"], pl["d_count"], pl["shared_terms"], **kw)
    else:
        t = orc.nll_terms(p, Y, c, **kw)
    for i, n in enumerate(TERM_NAMES):
        sums[i] += t.get(n, 0.0)
    sums[7] += 1.0 if pl["shared_terms"] else 0.0
t = torch.from_numpy(sums)
dm.all_reduce_sums(t)
if rank == 0:
    ref = orc.nll_terms_chains(params, Y, c, **kw)
    got = dm.finish(t.numpy())
    for n in TERM_NAMES:
        if n in ref:
            assert abs(got[n] - ref[n]) <= 1e-12 * max(1.0, abs(ref[n])), (n, got[n], ref[n])
    print("OK", got["nll"])
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode,name,port", [("chains", "tiny", "29531"), ("dims", "small_lin", "29535"), ("dims", "tiny", "29536")])
def test_gloo_world2_sharding(tmp_path, mode, name, port):
    """world_size-2 rehearsal of both shard modes: shard -> local partial sums -> all-reduce -> mean.
    'dims' on the LinearK / explicit-U workload is BASELINE config 5's layout, on 'tiny' the collapsed branch's."""
    import subprocess
    world = 2
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), mode, name], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


TIME_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ffvd_amd import synthetic, distributed as dm
from oracle import ffvd_oracle as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
params, Y, c, meta = synthetic.make_named("tiny", S=1, D=1)      # one chain, one latent dim: S * D < world
pl = dm.plan(meta, world, rank, "auto")
assert pl["mode"] == "time", pl
p = dict(params, X=params["X"][0])
t = torch.from_numpy(orc.tshard_partial(p, Y, c, pl["t_begin"], pl["t_count"]))   # the ORACLE stands in for the GPU engine
dm.all_reduce_sums(t)
got = orc.tshard_finish(p, t.numpy())
ref = orc.nll_terms(p, Y, c, U_collapse=True)
for n, v in ref.items():
    assert abs(got[n] - v) <= 1e-9 * max(1.0, abs(v)), (n, got[n], v)
print("OK", got["nll"])
dist.destroy_process_group()
'''


def test_gloo_world2_time_sharding(tmp_path):
    """world_size-2 rehearsal of the T-shard fallback (S * D < ranks): partial Gram sums -> one all-reduce -> every rank
    finishes the factorisations itself and holds the whole-job nll."""
    import subprocess
    script = tmp_path / "tworker.py"
    script.write_text(TIME_WORKER)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0] and "OK" in outs[1]


TIME_SHARD_ENGINE = r'''
import os, sys, math
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ffvd_amd import synthetic, distributed as dm
from oracle import ffvd_oracle_torch as ot
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
params, Y, c, meta = synthetic.make_named("tiny", S=1, D=2)
pl = dm.plan(meta, world, rank, "time")
t0, tc = pl["t_begin"], pl["t_count"]
KEYS = dm.GRAD_KEYS
JIT = 1e-5


class TorchShard:
    # A T-shard "engine" in torch fp64 with the decomposition of the HIP one (include/ffvd_abi.h "Gradient of a T-sharded job"):
    # the shard's rows enter through the raw tiles (K_uf K_fu, K_uf delta, sum_t Kdiag_t, likelihood / transition sums), whose
    # gradient is the shard's ADDITIVE share; everything the finish reads directly (K_uu, hyper-parameters, priors, x_0) is the
    # same on every shard and differentiated on the first only.
    def __init__(self):
        self.first = t0 == 0
        self.th = {k: torch.tensor(np.asarray(params[k]), dtype=torch.float64, requires_grad=True) for k in KEYS}
        self.X = torch.tensor(params["X"][0, t0: t0 + tc + 1], dtype=torch.float64, requires_grad=True)
        self.Y = torch.tensor(Y[t0: t0 + tc]); self.c = torch.tensor(c[t0: t0 + tc])
        self.D, self.M = meta["D"], meta["M"]

    def _local(self):
        th, X = self.th, self.X
        xc = torch.cat((X[:-1], self.c), dim=1)
        Kuf = ot._se_K(th["Z"], xc, th["logvariance"], th["loglengthscales"])          # D, M, tc
        delta = (X[1:] - X[:-1]).T
        G = Kuf @ Kuf.transpose(1, 2)
        g = (Kuf @ delta[:, :, None])[:, :, 0]
        kd = tc * torch.exp(th["logvariance"])
        Rrow = torch.exp(th["log_Rchols"])[0]
        lik_q = (-0.5 * (((self.Y - (X[1:] @ th["CC"] + th["DD"])) / Rrow[None, :]) ** 2)).sum()
        xq = (-0.5 * delta ** 2 / torch.exp(th["log_Q"])[:, None]).sum()
        return torch.cat([G.reshape(-1), g.reshape(-1), kd, torch.stack([lik_q, xq, torch.tensor(float(tc), dtype=torch.float64)])])

    def tshard_local(self):
        self.loc = self._local()
        return self.loc.detach().numpy().copy()

    def tshard_finish_grad(self, total, S_total=1):
        D, M = self.D, self.M
        tin = self.loc + (torch.tensor(total) - self.loc).detach()
        th = self.th if self.first else {k: v.detach() for k, v in self.th.items()}
        G = tin[: D * M * M].reshape(D, M, M); g = tin[D * M * M: D * M * M + D * M].reshape(D, M)
        kd = tin[D * M * M + D * M: D * M * M + D * M + D]; lik_q, xq, T = tin[-3], tin[-2], tin[-1]
        Q = torch.exp(th["log_Q"]); Rrow = torch.exp(th["log_Rchols"])[0]
        K = ot._se_K(th["Z"], th["Z"], th["logvariance"], th["loglengthscales"]) + JIT * torch.eye(M, dtype=torch.float64)
        A = K + G / Q[:, None, None]
        LK, LA = torch.linalg.cholesky(K), torch.linalg.cholesky(A)
        logdet = 2.0 * (torch.log(torch.diagonal(LA, dim1=1, dim2=2)).sum(-1) - torch.log(torch.diagonal(LK, dim1=1, dim2=2)).sum(-1))
        y = torch.linalg.solve_triangular(LA, (g / Q[:, None])[:, :, None], upper=False)[:, :, 0]
        trKG = torch.stack([torch.trace(torch.cholesky_solve(G[d], LK[d])) for d in range(D)])
        c_se = float(torch.log(torch.tensor(0.05, dtype=torch.float32)).double())
        hyp = -0.5 * ((th["log_Q"] ** 2).sum() + (th["CC"] ** 2).sum() + (th["DD"] ** 2).sum() + (th["log_Rchols"] ** 2).sum())
        p_hyper = -0.5 * (th["loglengthscales"] ** 2).sum() - 0.5 * ((th["logvariance"] - c_se) ** 2).sum()
        p_Z = -0.5 * (th["Z"] ** 2).sum()
        p_x0 = -0.5 * (self.X[0] ** 2).sum() if self.first else torch.zeros((), dtype=torch.float64)
        terms = [-(p_hyper + p_Z + p_x0 + hyp) / T, -(lik_q - T * torch.log(Rrow).sum()) / T,
                 -(xq - 0.5 * T * torch.log(Q).sum()) / T, (0.5 * ((kd - trKG) / Q).sum()) / T,
                 (0.5 * logdet).sum() / T, (-0.5 * (y * y).sum(-1)).sum() / T]
        nll = sum(terms)
        leaves = [self.X] + [self.th[k] for k in KEYS]
        grads = torch.autograd.grad(nll, leaves, allow_unused=True)
        self.dX = (grads[0] if grads[0] is not None else torch.zeros_like(self.X)).numpy() / S_total
        shared = [(gk if gk is not None else torch.zeros_like(self.th[k])).numpy().ravel() / S_total for k, gk in zip(KEYS, grads[1:])]
        sums8 = np.array([float(v) for v in terms] + [float(nll), 1.0]) if self.first else np.zeros(8)
        return np.concatenate([sums8] + shared)

    def tshard_adam_apply(self, rows, lr, beta1=0.9, beta2=0.999, eps=1e-8, train=None):
        # ffvd_tshard_adam_apply: the shared gradients are the ones the exchanged block left behind, dX the completed rows
        from oracle import ffvd_optim_oracle as oo
        if not hasattr(self, "adam"):
            self.adam = {k: (np.zeros(np.asarray(params[k]).shape), np.zeros(np.asarray(params[k]).shape)) for k in KEYS}
            self.adam["X"] = (np.zeros(tuple(self.X.shape)), np.zeros(tuple(self.X.shape)))
            self.adam_t = 0
        self.adam_t += 1
        grads = dict(self.last_shared, X=np.asarray(rows)[0])
        for k in list(KEYS) + ["X"]:
            leaf = self.X if k == "X" else self.th[k]
            new, m, v = oo.adam_step(leaf.detach().numpy(), grads[k], self.adam[k][0], self.adam[k][1], self.adam_t, lr, beta1, beta2, eps)
            self.adam[k] = (m, v)
            with torch.no_grad():
                leaf.copy_(torch.from_numpy(new))
        return self.last_sums

    def tshard_grad_fetch(self, block):
        self.last_sums = np.array(block[:8])
        self.last_shared, off0 = {}, 8
        for k in KEYS:
            n = int(np.asarray(params[k]).size)
            self.last_shared[k] = np.array(block[off0: off0 + n]).reshape(np.asarray(params[k]).shape)
            off0 += n
        out, off = {"X": self.dX[None]}, 8
        for k in KEYS:
            n = int(np.asarray(params[k]).size)
            out[k] = block[off: off + n].reshape(np.asarray(params[k]).shape)
            off += n
        return block[:8], out


def reduce_host(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).ravel().copy())
    dm.all_reduce_sums(t)
    return t.numpy()

'''

TIME_GRAD_TAIL = r'''terms, g = dm.tshard_nll_and_grad(TorchShard(), meta, t0, reduce_host)
p1 = dict(params, X=params["X"][0])
ref_t, ref_g = ot.nll_and_grad(p1, Y, c, ["X"] + list(KEYS), U_collapse=True)
assert abs(terms["nll"] - ref_t["nll"]) <= 1e-9 * abs(ref_t["nll"]), (terms["nll"], ref_t["nll"])
for k in ["X"] + list(KEYS):
    got = g[k][0] if k == "X" else g[k]
    err = np.abs(got - ref_g[k]).max() / max(np.abs(ref_g[k]).max(), 1e-12)
    assert err <= 1e-6, (k, err)
print("OK", terms["nll"])
dist.destroy_process_group()
'''


def test_gloo_world2_time_shard_gradient(tmp_path):
    """world_size-2 rehearsal of the T-shard BACKWARD pass: a torch stand-in engine with the HIP engine's decomposition (the shard's
    rows through the raw tiles = its additive share; what the finish reads directly, differentiated on the first shard only) runs
    through the product's own assembly (`distributed.tshard_nll_and_grad`: tiles -> all-reduce -> gradient block -> all-reduce ->
    dX rows at their global position -> all-reduce); every rank must hold the single-process nll and gradient."""
    import subprocess
    script = tmp_path / "tgworker.py"
    script.write_text(TIME_SHARD_ENGINE + TIME_GRAD_TAIL)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29538", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0] and "OK" in outs[1]


def test_plan_auto_picks_the_shard_axis():
    from ffvd_amd.distributed import plan
    assert "t_begin" not in plan(dict(S=32, D=4, T=4096), 8, 3, "auto")
    p = plan(dict(S=1, D=16, T=4096), 8, 3, "auto")
    assert (p["d_begin"], p["d_count"], p["shared_terms"]) == (6, 2, False)
    p = plan(dict(S=1, D=4, T=4096), 8, 7, "auto")
    assert p["mode"] == "time" and (p["t_begin"], p["t_count"]) == (3584, 512)
    covered = sum(plan(dict(S=1, D=1, T=1001), 8, r, "time")["t_count"] for r in range(8))
    assert covered == 1001
    with pytest.raises(ValueError):
        plan(dict(S=1, D=1, T=4), 8, 0, "time")


GRAD_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch.distributed as dist
from ffvd_amd import synthetic, distributed as dm
from oracle import ffvd_grad_oracle as gorc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
params, Y, c, meta = synthetic.make_named("tiny")
S = meta["S"]
pl = dm.plan(meta, world, rank, "chains")
def chain_grad(s):
    p = dict(params); p["X"] = params["X"][s]
    return gorc.nll_grad(p, Y, c)
# rank-local gradient of the mean-over-ALL-chains nll; the closed-form ORACLE stands in for the GPU backward pass
local = {k: np.zeros_like(np.asarray(params[k], dtype=np.float64)) for k in dm.GRAD_KEYS}
local["X"] = np.zeros((pl["s_count"],) + params["X"].shape[1:])
for i, s in enumerate(range(pl["s_begin"], pl["s_begin"] + pl["s_count"])):
    g = chain_grad(s)
    local["X"][i] = g["X"] / S
    for k in dm.GRAD_KEYS:
        local[k] += g[k] / S
got = dm.all_reduce_grads(local, "chains")
ref = {k: sum(chain_grad(s)[k] for s in range(S)) / S for k in dm.GRAD_KEYS}
for k in dm.GRAD_KEYS:
    assert got[k].shape == ref[k].shape and np.allclose(got[k], ref[k], rtol=1e-12, atol=1e-15), k
assert np.array_equal(got["X"], local["X"])              # chain shards keep their own rows
print("OK")
dist.destroy_process_group()
'''


def test_gloo_world2_gradient_all_reduce(tmp_path):
    """world_size-2 rehearsal of the sharded backward pass: one packed all-reduce of the shared-parameter gradients."""
    import subprocess
    script = tmp_path / "gworker.py"
    script.write_text(GRAD_WORKER)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0] and "OK" in outs[1]


ID_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import torch, torch.distributed as dist
from ffvd_amd import distributed as dm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
def broken():
    raise RuntimeError("librccl.so could not be bound")
# (1) make_id fails on rank 0: EVERY rank must raise the same error at the same point ...
err = None
try:
    dm.exchange_unique_id(broken, rank, world)
except RuntimeError as exc:
    err = str(exc)
assert err and "librccl.so could not be bound" in err, err
# ... so that the caller's next collective lines up (bench.py: "did every rank get a communicator")
ok = torch.tensor([0], dtype=torch.int32)
dist.all_reduce(ok)
assert int(ok.item()) == 0
# (2) a working make_id reaches every rank unchanged
blob = dm.exchange_unique_id(lambda: bytes(range(128)), rank, world)
assert blob == bytes(range(128))
print("OK")
dist.destroy_process_group()
'''


def test_gloo_world2_failed_unique_id_reaches_every_rank(tmp_path):
    """when rank 0 cannot create the RCCL id, the other ranks used to stay blocked in the broadcast while rank 0
    moved on to the next collective.  Now the failure travels through the same broadcast."""
    import subprocess
    script = tmp_path / "idworker.py"
    script.write_text(ID_WORKER)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29539", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0] and "OK" in outs[1]


def test_file_rendezvous_ignores_stale_ids_and_forwards_failures(tmp_path, monkeypatch):
    """File rendezvous of the RCCL id: the file name carries a tag (generation), a stale file of another tag is never read,
    and a failure on rank 0 is written into the file so that the polling ranks raise instead of timing out.  With a per-job
    nonce in the environment (launcher run id / MASTER_PORT) the id an EARLIER job left under the same tag is never read
    either, and rank 0 can remove its file once the ranks have had time to read it."""
    import time
    for k in ("FFVD_RENDEZVOUS_NONCE", "TORCHELASTIC_RUN_ID", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    d = str(tmp_path)
    with open(os.path.join(d, "rccl_unique_id"), "wb") as f:        # left behind by an older run / the old file name
        f.write(b"stale")
    with open(os.path.join(d, "rccl_unique_id.run7"), "wb") as f:   # an id of THIS tag from an earlier run: replaced by rank 0
        f.write(b"OKold")
    good = bytes(range(128))
    assert dist_mod.exchange_unique_id(lambda: good, 0, 2, rendezvous_dir=d, tag="run7") == good
    assert dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="run7") == good
    with pytest.raises(TimeoutError):
        dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="other", timeout_s=0.05)

    def broken():
        raise OSError("no librccl")
    with pytest.raises(RuntimeError, match="no librccl"):
        dist_mod.exchange_unique_id(broken, 0, 2, rendezvous_dir=d, tag="bad")
    with pytest.raises(RuntimeError, match="no librccl"):
        dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="bad")
    # default tag: the count of communicators this process has formed through a directory
    g0 = dist_mod._RENDEZVOUS_GENERATION[0]
    dist_mod.exchange_unique_id(lambda: good, 0, 2, rendezvous_dir=d)
    assert os.path.exists(os.path.join(d, f"rccl_unique_id.{g0}")) and dist_mod._RENDEZVOUS_GENERATION[0] == g0 + 1
    # a second job in the same directory, same tag: its nonce keeps it away from the first job's file
    monkeypatch.setenv("MASTER_PORT", "29777")
    other = bytes(reversed(range(128)))
    with pytest.raises(TimeoutError):
        dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="run7", timeout_s=0.05)      # job 1's "run7" is not job 2's
    assert dist_mod.exchange_unique_id(lambda: other, 0, 2, rendezvous_dir=d, tag="run7", cleanup_s=0.2) == other
    assert dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="run7") == other
    assert os.path.exists(os.path.join(d, "rccl_unique_id.29777.run7"))
    time.sleep(0.6)
    assert not os.path.exists(os.path.join(d, "rccl_unique_id.29777.run7"))                          # removed by rank 0's timer
    # two consecutive runs with the SAME nonce (MASTER_PORT 29500 twice) share the file name; an id written before this
    # process existed belongs to the earlier run: a reader ignores it and keeps polling for its own job's file
    stale = os.path.join(d, "rccl_unique_id.29777.again")
    with open(stale, "wb") as f:
        f.write(b"OK" + other)
    old_time = dist_mod._PROCESS_SEEN - 60.0
    os.utime(stale, (old_time, old_time))
    with pytest.raises(TimeoutError, match="fresh"):
        dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="again", timeout_s=0.05)
    assert dist_mod.exchange_unique_id(lambda: good, 0, 2, rendezvous_dir=d, tag="again") == good      # rank 0 of THIS job replaces it
    assert dist_mod.exchange_unique_id(None, 1, 2, rendezvous_dir=d, tag="again") == good


ADAM_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FFVD_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ffvd_amd import synthetic, distributed as dm, optim
from oracle import ffvd_grad_oracle as gorc, ffvd_optim_oracle as oo
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
params, Y, c, meta = synthetic.make_named("tiny")
S = meta["S"]
pl = dm.plan(meta, world, rank, "chains")
lr = optim.decayed_learning_rate()
keys = list(dm.GRAD_KEYS)
def chain_grad(p, s_X):
    q = dict(p); q["X"] = s_X
    return gorc.nll_grad(q, Y, c)
# ---- this rank: its chains only; the closed-form ORACLE stands in for the GPU backward pass ---------------------------
th = {k: np.array(params[k], dtype=np.float64) for k in keys}
thX = np.array(params["X"][pl["s_begin"]: pl["s_begin"] + pl["s_count"]], dtype=np.float64)
m = {k: np.zeros_like(th[k]) for k in keys}; v = {k: np.zeros_like(th[k]) for k in keys}
mX, vX = np.zeros_like(thX), np.zeros_like(thX)
for t in range(1, 5):
    # the exchange block of ffvd_adam_step_allreduce: shared-parameter gradients of this rank's chains, divisor S (whole job)
    local = {k: np.zeros_like(th[k]) for k in keys}
    gX = np.zeros_like(thX)
    for i in range(pl["s_count"]):
        g = chain_grad(dict(params, **th), thX[i])
        gX[i] = g["X"] / S
        for k in keys:
            local[k] += g[k] / S
    block = torch.from_numpy(np.concatenate([local[k].ravel() for k in keys]))
    dm.all_reduce_sums(block)                                       # ONE all-reduce; dX stays on the rank (chain shards)
    off = 0
    for k in keys:
        n = th[k].size
        th[k], m[k], v[k] = oo.adam_step(th[k], block.numpy()[off: off + n].reshape(th[k].shape), m[k], v[k], t, lr)
        off += n
    thX, mX, vX = oo.adam_step(thX, gX, mX, vX, t, lr)
# ---- the single-process trajectory -----------------------------------------------------------------------------------
ref = {k: np.array(params[k], dtype=np.float64) for k in keys + ["X"]}
rm = {k: np.zeros_like(ref[k]) for k in ref}; rv = {k: np.zeros_like(ref[k]) for k in ref}
for t in range(1, 5):
    tot = {k: np.zeros_like(ref[k]) for k in ref}
    for s in range(S):
        g = chain_grad({k: ref[k] for k in keys}, ref["X"][s])
        tot["X"][s] = g["X"] / S
        for k in keys:
            tot[k] += g[k] / S
    for k in ref:
        ref[k], rm[k], rv[k] = oo.adam_step(ref[k], tot[k], rm[k], rv[k], t, lr)
for k in keys:
    assert np.allclose(th[k], ref[k], rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(ref[k])))), k
assert np.allclose(thX, ref["X"][pl["s_begin"]: pl["s_begin"] + pl["s_count"]], rtol=0, atol=1e-9)
print("OK")
dist.destroy_process_group()
'''


def test_gloo_world2_sharded_adam_trajectory(tmp_path):
    """world_size-2 rehearsal of the device-resident sharded training step (ffvd_adam_step_allreduce): per step ONE all-reduce
    of the shared-parameter gradient block, chain shards update their own rows of X, and after 4 steps every rank sits on the
    single-process Adam trajectory (1e-9)."""
    import subprocess
    script = tmp_path / "aworker.py"
    script.write_text(ADAM_WORKER)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29538", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0] and "OK" in outs[1]


TIME_ADAM_TAIL = r'''
from oracle import ffvd_optim_oracle as oo
from ffvd_amd import optim
lr = optim.decayed_learning_rate()
sh = TorchShard()
nll0 = None
for step in range(4):
    t = dm.tshard_adam_step(sh, meta, rank, world, reduce_host, lr)
    nll0 = t["nll"] if nll0 is None else nll0
# ---- the single-process trajectory (torch autograd of the oracle, Adam with TF semantics) --------------------------------------
ref = {k: np.array(params[k], dtype=np.float64) for k in KEYS}
ref["X"] = np.array(params["X"][0], dtype=np.float64)
rm = {k: np.zeros_like(ref[k]) for k in ref}; rv = {k: np.zeros_like(ref[k]) for k in ref}
first = None
for step in range(1, 5):
    rt, rg = ot.nll_and_grad(dict(ref), Y, c, ["X"] + list(KEYS), U_collapse=True)
    first = rt["nll"] if first is None else first
    for k in ref:
        ref[k], rm[k], rv[k] = oo.adam_step(ref[k], rg[k], rm[k], rv[k], step, lr)
assert abs(nll0 - first) <= 1e-9 * abs(first)
for k in KEYS:
    assert np.allclose(sh.th[k].detach().numpy(), ref[k], rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(ref[k])))), k
assert np.allclose(sh.X.detach().numpy(), ref["X"][t0: t0 + tc + 1], rtol=0, atol=1e-9), np.abs(sh.X.detach().numpy() - ref["X"][t0: t0 + tc + 1]).max()
print("OK", nll0)
dist.destroy_process_group()
'''


def test_gloo_world2_time_shard_adam_trajectory(tmp_path):
    """the optimiser step of a T-sharded job (dgp_model.py:303-305 trains everything).  world_size-2 rehearsal of
    `distributed.tshard_adam_step` -- the product's own assembly: tiles -> all-reduce -> gradient block -> all-reduce -> boundary
    rows of dX -> all-reduce of (world - 1) x S x D doubles -> every shard updates its rows of X and its copy of the shared
    parameters -- with the torch stand-in engine of the gradient test; after 4 steps every shard sits on the single-process Adam
    trajectory (1e-9), the row the two shards share included."""
    import subprocess
    script = tmp_path / "taworker.py"
    script.write_text(TIME_SHARD_ENGINE + TIME_ADAM_TAIL)
    env = dict(os.environ, FFVD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29539", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0] and "OK" in outs[1]


def test_tshard_boundary_rows_fake_collective():
    """The boundary exchange on its own, with a fake collective (sum over the ranks' contributions in Python): 3 shards, every
    boundary row ends as the sum of its two parts, interior rows are untouched."""
    from ffvd_amd.distributed import tshard_boundary_rows
    rng = np.random.default_rng(3)
    world, S, D = 3, 2, 3
    parts = [rng.standard_normal((S, 4 + r, D)) for r in range(world)]
    sent = []
    for r in range(world):
        tshard_boundary_rows(parts[r], r, world, lambda a: (sent.append(np.array(a)), a)[1])
    total = np.sum(sent, axis=0)
    for r in range(world):
        out = tshard_boundary_rows(parts[r], r, world, lambda a: total)
        assert np.array_equal(out[:, 1:-1], parts[r][:, 1:-1])
        if r > 0:
            assert np.allclose(out[:, 0], parts[r][:, 0] + parts[r - 1][:, -1])
        else:
            assert np.array_equal(out[:, 0], parts[r][:, 0])
        if r < world - 1:
            assert np.allclose(out[:, -1], parts[r][:, -1] + parts[r + 1][:, 0])
