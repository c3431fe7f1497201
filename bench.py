#!/usr/bin/env python3
"""bench.py -- ELBO iterations/sec on MI355X for BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c4|c5] [--dtype f64|f32c] [--route gram|reference]

`python bench.py --gpus N` run plainly starts its N rank processes ITSELF (fresh children created before anything in
the parent touches the GPU; the parent only waits and forwards the exit code).  Under a launcher that already set
RANK / WORLD_SIZE (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the process is one rank.

Workload (config.workload), default c2 = BASELINE configs[1]: synthetic T=4096, x_dim=4, M=512, S=32, SquaredExponential,
fp64, collapsed-U branch; inputs from ffvd_amd/synthetic.py, resident in HBM before the timed region.
One "step" = one forward evaluation of nll + its component terms for all S chains, scalar result on the host.
--workload c1 = BASELINE configs[0], the reference's own experiment size (actuator fixture: T=512, M=100, x_dim=4, S=10 chains =
the fixture's trajectory + 9 perturbed copies): the whole iteration is ONE kernel launch (ffvd_amd/csrc/tiny.hip); the
line also carries the device-resident training step (forward + backward + Adam, `train_ms_per_step`) and the CPU baseline times
ALL chains, forward and closed-form gradient.
With N > 1 the S chains (c5: the latent dims) are sharded over the ranks, no data-path collective, and the 8 partial
sums are all-reduced by the library's own ncclAllReduce (include/ffvd_abi.h ffvd_elbo_allreduce; torch.distributed/gloo
only carries the 128-byte rendezvous id, the barriers and the max over ranks).  Total work is fixed: scaling = "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel group: algorithmic flops / HIP-event duration on the engine's stream vs the MFMA
                  peak of the contraction dtype
  cpu_baseline -- the NumPy restatement of the reference's CPU path (oracle/, kind "port") timed on this host on a
                  bounded sample of the same workload (N = 1, rank 0 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32c": 157.3}     # MI355X dense matrix peaks (MI355X_MICROARCH.md)
WORKLOAD = "c2"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=WORKLOAD, choices=("c1", "c2", "c4", "c5", "small", "tiny"))
    ap.add_argument("--dtype", default=None, choices=("f64", "f32c"),
                    help="f64 (default; c2 headline) or f32c = fp32 K_fu + fp32 MFMA contractions, fp64 M x M "
                         "factorisations and accumulation (default for c4, BASELINE configs[3])")
    ap.add_argument("--cpu-sample-chains", type=int, default=0, help="0 = sized for about 10-30 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chains-per-pass", type=int, default=0)
    ap.add_argument("--route", choices=("gram", "reference"), default=None,
                    help="gram: log|K_uu + K_uf K_fu/Q| - log|K_uu| form (default for f64, ~half the flops); "
                         "reference: F = K_fu L^-T, H = F^T F/Q + I in the reference's op order (the only route of f32c)")
    args = ap.parse_args(argv)
    if args.dtype is None:
        args.dtype = "f32c" if args.workload == "c4" else "f64"
    if args.route is None:
        args.route = "reference" if (args.dtype == "f32c" or args.workload == "c5") else "gram"
    if args.workload == "c4" and args.steps == 60 and args.warmup == 10:
        args.steps, args.warmup = 5, 1            # one iteration is ~0.5 s
    return args


def spawn_ranks(args):
    """Parent of a plain `python bench.py --gpus N`: start N fresh rank processes, one GPU each, and wait.
    Nothing here imports torch or touches HIP."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait(timeout=3000)))
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def parallelism_text(world, collective, reduces):
    """What the timed step does with the 8 partial sums -- must be TRUE for the run (tests/test_bench_contract.py)."""
    if not reduces:
        return "no collective (single rank: the 8 sums are copied to the host as they are)"
    if collective == "rccl":
        return ("one ncclAllReduce of 8 doubles issued by the library (ffvd_elbo_allreduce)"
                + (", here on a 1-rank communicator so that N=1 times the same sequence as N>1" if world == 1 else ""))
    return "8 partial sums all-reduced through torch.distributed/gloo (rehearsal or RCCL unavailable)"


def load_c1(S=10):
    """BASELINE configs[0]: the actuator fixture (tests/golden/actuator_slim.npz: standardised u, p of data/actuator.mat and the arrays
    of one Factnonlin_ini file the path consumes, FFVD_Main.py:143-168,212-229) with S chains: chain 0 is the fixture's own
    trajectory, chains 1.. are X + 0.1 eps_s with the seeded generator of SURVEY 8(d) (the MC latent-state draw, utils.py:11)."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "actuator_slim.npz"), allow_pickle=False)
    params = {k: np.array(z[k]) for k in ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")}
    Y, c = np.array(z["Y"]), np.array(z["control_inputs"])
    T, D = params["X"].shape[0] - 1, params["X"].shape[1]
    rng = np.random.Generator(np.random.PCG64(20230209))
    eps = rng.standard_normal((S, T + 1, D))
    eps[0] = 0.0
    params["X"] = np.ascontiguousarray(params["X"][None] + 0.1 * eps)
    meta = dict(T=T, D=D, C=c.shape[1], M=params["Z"].shape[0], S=S, P=D + c.shape[1], Ydim=Y.shape[1],
                kernel_type="SquaredExponential", U_collapse=True, seed=20230209)
    return params, Y, c, meta


def cpu_baseline_train(params, Y, c, meta, threads):
    """The CPU counterpart of one training evaluation (nll + its gradient, what tf.gradients(nll, vars) computes per Adam step,
    base_model.py:148, dgp_model.py:303-305): the closed-form gradient restatement on ALL chains."""
    from oracle import ffvd_grad_oracle as gorc     # reported baseline only; never the product path
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    gorc.nll_grad(dict(params, X=params["X"][0]), Y, c)          # untimed: warms the BLAS thread pool
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or time.perf_counter() - t0 < 4.0:
        for s in range(meta["S"]):
            gorc.nll_grad(dict(params, X=params["X"][s]), Y, c)
        reps += 1
    del limiter
    return (time.perf_counter() - t0) / reps


def cpu_baseline(params, Y, c, meta, sample_chains, workload, threads=None):
    """Time the oracle (NumPy restatement, reference op order) on `sample_chains` chains; extrapolate to S.
    At the reference's own experiment size a threaded BLAS is SLOWER than one thread (100 x 100 matrices: 0.56 s against 0.067 s
    per iteration on the build box), so c1 is timed at 1 thread and at the host's share of cores and the faster one is reported."""
    import numpy as np
    from oracle import ffvd_oracle as orc      # reported baseline only; never the product path
    if threads is None and workload == "c1":
        one = cpu_baseline(params, Y, c, meta, sample_chains, workload, threads=1)
        many = cpu_baseline(params, Y, c, meta, sample_chains, workload, threads=min(16, os.cpu_count() or 1))
        best, other = (one, many) if one["value"] >= many["value"] else (many, one)
        best["other_thread_count"] = {"cores": other["cores"], "value": other["value"]}
        return best
    # the GPU box gives one GPU's share of the host (16 cores); more BLAS threads than that only oversubscribes
    threads = threads or min(16, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    kern = orc.make_kernels(params)
    T = meta["T"]
    Q = np.exp(params["log_Q"])

    def one_chain(Linv, s):                            # per-chain part of dgp_model.py:248-288
        X = params["X"][s]
        xc = np.concatenate((X[:-1], c[:T]), axis=1)
        orc.collapse_after_kernel_precalculation(Linv, xc, X, params["Z"], kern, Q, float(T), float(T))
        ym = orc.predict_mean(X[1:], params["CC"], params["DD"])
        orc.logdensity_norm_diag(Y, ym, np.exp(params["log_Rchols"])[0]).sum()
        orc.logdensity_norm_diag_nonvec(X[1:], X[:-1], Q ** 0.5).sum()

    # small workloads (a pass takes well under a second): one untimed pass warms the BLAS thread pool, then whole passes are
    # repeated for about 5 s and averaged; the large ones are timed once
    small = meta["T"] * meta["M"] * meta["M"] * meta["D"] * sample_chains < 4e9
    reps = 0
    if small:
        one_chain(orc.kernel_pre_cal(params["Z"], kern), 0)
    t_shared = t_chain_total = 0.0
    t_begin = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        Linv = orc.kernel_pre_cal(params["Z"], kern)   # shared per-dim part: K_uu, Cholesky, L^-T (:124-169)
        t_shared += time.perf_counter() - t0
        t0 = time.perf_counter()
        for s in range(sample_chains):
            one_chain(Linv, s)
        t_chain_total += time.perf_counter() - t0
        reps += 1
        if not small or time.perf_counter() - t_begin > 5.0:
            break
    t_shared /= reps
    t_chain = t_chain_total / reps / sample_chains
    t_iter = t_shared + meta["S"] * max(t_chain, 0.0)
    del limiter
    return {
        "value": 1.0 / t_iter, "unit": "ELBO iters/sec", "cores": int(threads), "kind": "port",
        "sample": (f"{sample_chains} of {meta['S']} chains of the {workload} workload at full T/M/D, NumPy fp64 "
                   f"(OpenBLAS, {threads} threads), {reps} timed pass(es); per-iteration time = shared K_uu part {t_shared:.4f}s + "
                   f"S x per-chain {max(t_chain, 0.0):.4f}s"),
        "seconds_per_iter": t_iter,
    }


def run_rank(args):
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from ffvd_amd import synthetic
    from ffvd_amd.distributed import ShardedElbo, finish

    # rehearsal on a one-GPU box: FFVD_BENCH_REHEARSAL=1 puts every rank on device 0 and carries the sums over
    # torch/gloo (RCCL wants one GPU per rank); real runs use one GPU per rank and the library's ncclAllReduce
    rehearsal = bool(os.environ.get("FFVD_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = 0
    # stdout of this program is ONE JSON line.  Libraries write there from their C++ side -- gloo announces its connections,
    # RCCL prints a version banner when a communicator forms (seen on the GPU box: "RCCL version : 2.27.7 ...") -- so file
    # descriptor 1 points at stderr for the whole run and the line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1:
        import torch.distributed as dist       # host-side plumbing: rendezvous id, barriers, max over ranks (gloo, CPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()

    params, Y, c, meta = load_c1() if args.workload == "c1" else synthetic.make_named(args.workload)
    mode = "dims" if meta["S"] < world or args.workload == "c5" else "chains"
    eng_kw = dict(route=args.route, dtype=args.dtype)
    if args.chains_per_pass:
        eng_kw["chains_per_pass"] = args.chains_per_pass
    collective = "torch" if rehearsal else "rccl"
    sh, err = None, None
    # N = 1 times what N > 1 times: the single rank forms a 1-rank communicator and every step runs
    # kernels -> finalize -> ncclAllReduce(8 doubles) -> copy back, exactly the sequence of ffvd_elbo_allreduce at N > 1.
    # If no RCCL can be bound on a one-GPU box the step has no collective, and config.parallelism says so.
    always = (world == 1 and not rehearsal)
    try:
        sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode=mode, device=local_rank, collective=collective,
                         always_reduce=always, **eng_kw)
    except Exception as exc:               # noqa: BLE001 -- reported below, never swallowed
        # (only the text: the exception's traceback would pin the half-built shard; ShardedElbo closes its engine itself)
        err = RuntimeError(f"{type(exc).__name__}: {exc}")
    exchange = None                        # what the timed step did with the 8 partial sums (config.parallelism)
    if dist is not None and not rehearsal:
        # The benchmark must say which exchange it timed.  If the library's own RCCL communicator cannot be formed on
        # EVERY rank (e.g. no librccl the process can bind), all ranks agree -- loudly, on stderr and in the JSON line -- to
        # carry the 8 sums through torch.distributed instead; a failure on some ranks only is an error.
        # (exchange_unique_id delivers a failure of rank 0 to every rank, so this all-reduce lines up on all of them.)
        import torch
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        if int(ok.item()) == 0:
            print(f"[bench rank {rank}] native RCCL path unavailable on all ranks ({err}); timing the torch.distributed "
                  "(gloo) exchange instead", file=sys.stderr, flush=True)
            collective = "torch"
            sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode=mode, device=local_rank,
                             collective=collective, **eng_kw)
        elif err is not None or int(ok.item()) != world:
            raise SystemExit(f"rank {rank}: RCCL communicator formed on {int(ok.item())} of {world} ranks only: {err}")
    elif err is not None and always:
        print(f"[bench] 1-rank RCCL communicator unavailable ({err}); the timed step has NO collective",
              file=sys.stderr, flush=True)
        exchange = f"no collective at N=1 (a 1-rank RCCL communicator could not be formed: {err})"
        sh = ShardedElbo(params, Y, c, meta, rank=rank, world=world, mode=mode, device=local_rank, collective=collective,
                         **eng_kw)
    elif err is not None:
        raise err
    if exchange is None:
        exchange = parallelism_text(world, collective, sh.reduces)

    def barrier():
        sh.engine.sync()
        if dist is not None:
            dist.barrier()
        sh.engine.sync()

    def timed(k):
        per = np.zeros(k)
        barrier()
        t0 = time.perf_counter()
        out = None
        for i in range(k):
            t1 = time.perf_counter()
            out = sh.step()
            per[i] = time.perf_counter() - t1
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, per, out

    sums = None
    for _ in range(args.warmup):
        sums = sh.step()
    # (1) the headline region: exactly K steps, nothing but the steps inside
    elapsed, per_step, sums = timed(args.steps)
    # (1b) N = 1 only: the same K steps WITHOUT the 1-rank collective (ffvd_elbo: one copy of the 8 sums), an extra key
    plain_ms = None
    if world == 1 and sh.reduces and not sh.time_shard:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sh.engine.elbo_sums()
        barrier()
        plain_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    # (2) the same K steps again with HIP events recorded around every stage on the engine's stream (roofline numbers)
    sh.engine.stage_timing(True)
    elapsed_ev, _, _ = timed(args.steps)
    stage = sh.engine.stage_times()
    sh.engine.stage_timing(False)
    terms = finish(sums)

    if rank == 0:
        T, D, M, S, P = meta["T"], meta["D"], meta["M"], meta["S"], meta["P"]
        s_local, d_local = sh.plan["s_count"], sh.plan["d_count"]
        collapsed = meta["U_collapse"]
        peak = PEAK_TFLOPS[args.dtype]
        # algorithmic flops per launch group (SURVEY 8d per-(s,d) figures x the (s,d) units one launch processes)
        alg = {
            "project_F": (T * M * M + T * M * (2 * P + 4)),      # reference route: trsm-equivalent T*M^2 + K_fu generation
            "gram_H": (T * M * M),                                # symmetric update T*M^2 (both routes)
        }
        if args.route == "gram":
            alg["project_F"] = T * M * (2 * P + 4)                # K_fu generation only
        dom = max(("project_F", "gram_H"), key=lambda k: stage[k][0])
        ms_tot, launches = stage[dom]
        units_per_launch = s_local * d_local * args.steps / max(launches, 1)     # (s,d) units per launch
        lowrank = (not collapsed and meta["kernel_type"] == "LinearK" and P <= 32
                   and os.environ.get("FFVD_NO_LINEAR_LOWRANK", "") not in ("1", "true", "yes"))
        if lowrank:
            # LinearK through its rank: the projection is T*P^2 work, the dominant launch is the K_uu chain --
            # one dataflow Cholesky of d_local matrices with one 64-row extension block (Z^T -> C), latency-bound by construction
            alg["project_F"] = T * (2 * P * P + 4 * P)
            alg["kuu_chol_inverse"] = M ** 3 / 3 + 64 * M * M
            dom = max(("project_F", "kuu_chol_inverse"), key=lambda k: stage[k][0])
            ms_tot, launches = stage[dom]
            if dom == "kuu_chol_inverse":
                units_per_launch = d_local * args.steps / max(launches, 1)
        single = int(sh.engine.lib.ffvd_single_launch(sh.engine._h))
        if single:
            # the whole iteration is ONE launch in the reference's op order (tiny.hip): per (s,d) unit SURVEY 8(d)'s W_alg terms
            dom = "one_launch_iteration"
            ms_tot, launches = stage["gram_H"]
            alg[dom] = 2 * T * M * M + M ** 3 / 3 + T * M * (2 * P + 4) + 2 * M ** 3 / 3
            units_per_launch = s_local * d_local * args.steps / max(launches, 1)
        dur_s = ms_tot / 1e3 / max(launches, 1)
        achieved = alg[dom] * units_per_launch / dur_s / 1e12 if dur_s > 0 else 0.0
        w_alg = synthetic.algorithmic_flops(**meta)                 # SURVEY 8(d) W_alg (reference formulation)
        if lowrank:     # the formulation actually executed: per d one Cholesky with the Z^T block, per (s,d) the two P x P forms per row
            w_alg = D * (M ** 3 / 3 + 64 * M * M) + S * D * T * (2 * P * P + 4 * P)
        if args.route == "gram" and not single:
            # structure-aware count of the formulation actually executed (SURVEY 8d requires a Gram-route build to
            # say so): per (s,d) T*M^2 syrk + M^3/3 Cholesky of K_uu + K_uf K_fu/Q + K_fu generation; per d
            # Cholesky + inverse factor + K^-1 (M^3/3 + M^3/3 + M^3).  SURVEY's W_gram for comparison: 1.643e11.
            w_alg = S * D * (T * M * M + M ** 3 / 3 + T * M * (2 * P + 4)) + D * (5 * M ** 3 / 3)
        value = args.steps / elapsed
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and os.path.exists(tpath):      # measured on the 1-GPU launch shape only, by a separate --pmc run
            try:
                tj = json.load(open(tpath))
                key = f"{args.workload}/{args.dtype}/{args.route}"
                if key in tj:
                    traffic = tj[key].get(dom)
                    traffic_source = f"profiles/traffic.json[{key}] (rocprofv3 --pmc passes, {tj[key].get('_commit', '?')})"
            except Exception:
                traffic = None
        kname = "SquaredExponential" if meta["kernel_type"] == "SquaredExponential" else "LinearK"
        out = {
            "metric": f"ELBO iters/sec (T={T}, M={M}, x_dim={D}, S={S})", "value": value, "unit": "iters/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "median_ms_per_step": 1e3 * float(np.median(per_step)), "min_ms_per_step": 1e3 * float(per_step.min()),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "actuator fixture + synthetic chains" if args.workload == "c1" else "synthetic",
            "config": {"workload": (f"c1: actuator fixture (tests/golden/actuator_slim.npz) T={T} x_dim={D} C={meta['C']} M={M}, S={S} chains = "
                                    f"the fixture's trajectory + {S - 1} draws X + 0.1 eps (seed {meta['seed']}), {kname} collapsed-U"
                                    if args.workload == "c1" else
                                    f"{args.workload}: synthetic T={T} x_dim={D} C={meta['C']} M={M} S={S} {kname} "
                                    f"{'collapsed-U' if collapsed else 'explicit-U'}, seed {meta['seed']}"),
                       "parallelism": f"{'chains' if mode == 'chains' else 'latent dims'} sharded over {world} GPU(s), " + exchange,
                       "chains_per_gpu": s_local, "dims_per_gpu": d_local,
                       "arithmetic": ("fp64 throughout" if args.dtype == "f64" else
                                      "K_fu and the two T x M x M products in fp32 (v_mfma_f32_32x32x2_f32); every M x M "
                                      "factorisation, the accumulation of H and the trace term in fp64"),
                       "route": ("one launch per iteration (ffvd_amd/csrc/tiny.hip), reference op order F = K_fu L^-T, H = F^T F/Q + I: "
                                 "latency-bound by two 16-tile Cholesky pivot chains, not by the matrix pipe" if single else
                                 "gram: log|K_uu + K_uf K_fu/Q| - log|K_uu| (SURVEY Appendix A), flops counted as W_gram-style"
                                 if args.route == "gram" else
                                 ("explicit-U with LinearK through the kernel's rank: F = sigma^2 X (Z^T L^-T), fmean and sum F^2 as "
                                  "P x P forms per row (DESIGN.md section 5); FFVD_NO_LINEAR_LOWRANK=1 runs the M-wide projection"
                                  if lowrank else "reference: F = K_fu L^-T, H = F^T F/Q + I"))},
            "nll": terms["nll"],
            "ms_per_step_without_collective": plain_ms,
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_ms": 1e3 * dur_s, "launches_timed": launches,
                         "measured_over": f"a second timed pass of the same {args.steps} steps with HIP events on the "
                                          f"engine's stream ({1e3 * elapsed_ev / args.steps:.3f} ms per step with the events)",
                         "whole_iteration": {"W_alg_flops": w_alg, "achieved": w_alg * value / 1e12,
                                             "frac": w_alg * value / 1e12 / peak},
                         "stage_ms_per_step": {k: v[0] / args.steps for k, v in stage.items()}},
        }
        if args.workload == "c1" and world == 1:
            # the device-resident training step of the reference's loop (models.py:142-182: forward + backward + Adam update)
            from ffvd_amd.engine import ElboEngine
            with ElboEngine(T, D, meta["C"], M, S, Ydim=meta["Ydim"], grad=True, device=local_rank) as te:
                te.set_data(Y, c)
                te.set_params(params)
                for _ in range(args.warmup):
                    te.adam_step(1e-9)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    te.adam_step(1e-9)
                out["train_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / args.steps
                out["train_single_launch"] = int(te.lib.ffvd_single_launch(te._h))
        if world == 1 and not args.no_cpu_baseline and collapsed:
            # (c1, c2: ALL chains are timed, nothing is extrapolated -- VERDICT r3 W10; c4: one chain x 2 dims, see cpu_baseline_c4)
            n = args.cpu_sample_chains or {"c1": S, "c2": S, "c4": 1}.get(args.workload, min(S, 4))
            if args.workload == "c4":
                # one (chain, dim) unit costs ~4 s of CPU: time ONE chain of ONE latent dim would not exercise the
                # shared part; instead one chain over 2 of the 8 dims -- see cpu_baseline_c4
                out["cpu_baseline"] = cpu_baseline_c4(params, Y, c, meta)
            else:
                out["cpu_baseline"] = cpu_baseline(params, Y, c, meta, n, args.workload)
            if args.workload == "c1":
                tg = cpu_baseline_train(params, Y, c, meta, out["cpu_baseline"]["cores"])
                out["cpu_baseline"]["train_seconds_per_step"] = tg
                out["cpu_baseline"]["train_sample"] = (f"nll + closed-form gradient (oracle/ffvd_grad_oracle.py) of all {S} chains, NumPy fp64, "
                                                       f"{out['cpu_baseline']['cores']} threads: the CPU counterpart of one train_hypers evaluation")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    sh.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline_c4(params, Y, c, meta):
    """Bounded CPU sample at config 4 (T=16384, M=2048): one chain restricted to its first 2 latent dims (about
    15-25 s of oracle work on 16 threads), extrapolated linearly in dims and chains."""
    import numpy as np
    from oracle import ffvd_oracle as orc
    threads = min(16, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    nd = 2
    kern = orc.make_kernels(params)[:nd]
    T, D, S = meta["T"], meta["D"], meta["S"]
    Q = np.exp(params["log_Q"])[:nd]
    t0 = time.perf_counter()
    Linv = orc.kernel_pre_cal(params["Z"], kern)
    t_shared = (time.perf_counter() - t0) * D / nd
    X = params["X"][0]
    xc = np.concatenate((X[:-1], c[:T]), axis=1)
    t0 = time.perf_counter()
    orc.collapse_after_kernel_precalculation(Linv, xc, X[:, :nd], params["Z"], kern, Q, float(T), float(T))
    t_chain = (time.perf_counter() - t0) * D / nd
    t_iter = t_shared + S * t_chain
    del limiter
    return {
        "value": 1.0 / t_iter, "unit": "ELBO iters/sec", "cores": int(threads), "kind": "port",
        "sample": (f"1 of {S} chains x {nd} of {D} latent dims of the c4 workload at full T/M, NumPy fp64 (OpenBLAS, "
                   f"{threads} threads), scaled linearly: shared K_uu part {t_shared:.1f}s + S x per-chain {t_chain:.1f}s"),
        "seconds_per_iter": t_iter,
    }


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
