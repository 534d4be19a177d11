#!/usr/bin/env python3
"""Headline benchmark: EM iterations/s of the Poisson-GPFA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c1] [--no-cpu-baseline] [--cpu-estimate]
                    [--workload em|online|loo|dual|floor]

A step is one full batch-EM iteration on synthetic spike counts already resident in HBM: warm-started Laplace
E-step over every trial (batched inexact Newton with the shared-preconditioner PCG, posterior covariance blocks by
the low-rank engine) followed by the M-step for C, d (device per-neuron Newton; `--cd-method TNC` runs the reference
engine's scipy driver on the HIP cost/grad kernel instead) and the GP timescales (4-point lockstep root finder on the
batched HIP Gram/Cholesky/trace pass).

Workload (BASELINE.json): config 3 = 200 neurons, 10 latents, 500 bins, 1024 trials per GPU - the
configuration the north-star target is quoted on.  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; every rank owns 1024 trials (weak scaling, config 4's 8192 trials at N = 8), the
M-step sufficient statistics are summed with RCCL all-reduces.  `value` counts EM iterations per
second in units of 1024-trial batches: at N = 1 it is plain EM iterations/s on config 3.

`--gpus N` without a launcher (no RANK in the environment) starts the N ranks itself: the parent - which never touches the
GPU - spawns N children of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays
rank 0's JSON line and exits with the worst child status.  Under torch.distributed.run the ranks exist already.

`--workload online` is BASELINE config 4: stochastic EM ('diag' updates, engine.py:288-448) over 8192 resident trials with
minibatches of 1024; the minibatch is split over the ranks (strong scaling of one EM iteration), every rank holds the packed
count tensor (0.8 GB).

Prints ONE JSON line (rank 0).  The timed region is bracketed by a collective + device sync on both
sides and the reported time is the max over ranks.
"""
import argparse
import json
import os
import sys
import time

if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    # one rank per GPU on one node: keep the ranks' host BLAS/OpenMP pools from oversubscribing the cores
    _share = str(max(1, (os.cpu_count() or 8) // int(os.environ['WORLD_SIZE'])))
    for _var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(_var, _share)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (ROOT, os.path.join(ROOT, 'poisson-gpfa_amd')):
    if path not in sys.path:
        sys.path.insert(0, path)

CONFIGS = {   # name: (neurons, latents, bins, trials per GPU)
    'c1': (30, 3, 100, 20),
    'c2': (100, 5, 200, 256),
    'c3': (200, 10, 500, 1024),
    'c5': (500, 20, 1000, 2048),      # variational E-step variant (BASELINE config 5); only with --workload dual
}
FP64_MATRIX_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix (= FP64 vector) peak, AMD CDNA4 datasheet
HBM_PEAK_GBS = 8000.0              # HBM3E, MI355X_MICROARCH.md
FP32_MATRIX_PEAK_TFLOPS = 157.3    # MI355X FP32 matrix peak (v_mfma_f32_16x16x4_f32: twice the FP64 rate), AMD CDNA4 datasheet


class Shard:
    """Duck-typed experiment (reference engine.py:32-38) holding this rank's trials."""

    def __init__(self, Ys, bin_ms):
        self.data = [{'Y': y} for y in Ys]
        self.ydim, self.T = Ys[0].shape
        self.binSize = bin_ms
        self.trialDur = self.T * bin_ms
        self.numTrials = len(Ys)
        self._pgpfa_local_shard = True


def synth_shard(q, p, T, R, seed, rank):
    """Reference recipe (util.py:705-750): C~U(-.5,.5), d~U(-2,0)-1, tau~U(0,1)+.01 s from the legacy
    global RNG (identical on every rank); trials of this rank drawn per latent through T x T Cholesky
    factors from a Generator seeded with (seed, rank)."""
    np.random.seed(seed)
    C = np.random.rand(q, p) - 0.5
    d = np.random.rand(q) * (-2) - 1.0
    tau = np.abs(np.random.rand(p)) + 0.01
    t = np.arange(T, dtype=np.float64) * 10.0
    dsq = (t[:, None] - t[None, :]) ** 2
    L = np.stack([np.linalg.cholesky(0.999 * np.exp(-0.5 * dsq / (tk * 1000.0) ** 2) + 0.001 * np.eye(T)) for tk in tau])
    rng = np.random.default_rng([seed, rank])
    Ys = []
    for _ in range(R):
        X = np.einsum('kts,ks->kt', L, rng.standard_normal((p, T)))
        Ys.append(rng.poisson(np.exp(C @ X + d[:, None])).astype(np.uint8))
    return {'C': C, 'd': d, 'tau': tau}, Ys


class _Budget(Exception):
    pass


def cpu_baseline(q, p, T, R, true_params, Y0, bin_ms, init, estimate=False, budget_s=240.0):
    """Reference-faithful CPU path (the oracle in 'faithful' mode = the reference's big-matrix arithmetic and scipy
    drivers) on the host cores of this box, scaled to one EM iteration over R trials (the reference loops trials
    sequentially, inference.py:94).

    Default: ONE whole faithful trial, TIMED - scipy Newton-CG on the dense big-matrix callbacks + the dense inverse
    (inference.py:119-131) - and that trial's share of the M-step (learning.py:93-141 on the one-trial posterior; the
    timescale update, learning.py:257-293, once).  Small configurations time several trials.  The Newton-CG run is bounded
    by `budget_s` of wall clock: if the box cannot finish the trial in that time the run is cut after a whole Newton
    iteration and the remaining iterations are priced at the measured cost per Hessian build (labelled as such).
    estimate=True (--cpu-estimate): the round-1/2 extrapolation from two Hessian builds + one inverse."""
    from oracle import pgpfa_oracle as orc
    import scipy.optimize as op
    cores = os.cpu_count()
    n = p * T
    Ys = [Y0.astype(np.float64)]
    t0 = time.time()
    mstep_note = ''
    mstep_s = 0.0
    if n <= 1200:
        # small enough to run whole trials: faithful Laplace E-step on a few trials, then the faithful M-step on them
        ntr = 2 if n > 400 else 8
        Ys = [Y0.astype(np.float64)] * ntr
        res, _, _ = orc.laplace(Ys, init, bin_ms, mode='faithful', return_cov=True)
        per_trial = (time.time() - t0) / ntr
        t1 = time.time()
        orc.update_params(init, Ys, res, bin_ms, cd_method='TNC')
        mstep_s = (time.time() - t1)
        sample = ('MEASURED: %d full trials of the faithful Laplace E-step (scipy Newton-CG on the big-matrix callbacks), %.2f s each, + the faithful '
                  'M-step on them (%.2f s; its (C,d) part scales with the trial count)' % (ntr, per_trial, mstep_s))
        mstep_s = mstep_s * R / ntr
    elif not estimate:
        K = orc.make_K(init['tau'], T, bin_ms)
        C_big, d_big = orc.make_Cd_big(init['C'], init['d'], T)
        K_bigInv = np.linalg.inv(orc.make_K_big(K))
        t_setup = time.time() - t0
        ybar = Ys[0].reshape(-1)
        stat = {'hess': 0, 'hess_s': 0.0, 'nit': 0, 't0': time.time(), 'x': np.zeros(n)}

        def hess(x, *a):
            t1 = time.time()
            H = orc.nlp_big_hess(x, *a)
            stat['hess'] += 1
            stat['hess_s'] += time.time() - t1
            return H

        def cb(xk):
            stat['nit'] += 1
            stat['x'] = np.array(xk)
            if time.time() - stat['t0'] > budget_s:
                raise _Budget()
        complete = True
        try:
            out = op.minimize(orc.nlp_big, np.zeros(n), args=(ybar, C_big, d_big, K_bigInv), method='Newton-CG', jac=orc.nlp_big_grad,
                              hess=hess, callback=cb, options={'disp': False, 'maxiter': 10000})     # inference.py:119-126
            x = out.x
        except _Budget:
            complete = False
            x = stat['x']
        t_newton = time.time() - stat['t0']
        t1 = time.time()
        H = orc.nlp_big_hess(x, ybar, C_big, d_big, K_bigInv)                                         # inference.py:130
        Sigma = np.linalg.inv(H)                                                                      # inference.py:131
        t_inv = time.time() - t1
        if complete:
            per_trial = t_newton + t_inv
            sample = ('MEASURED: 1 whole faithful trial of the Laplace E-step (scipy Newton-CG on the dense big-matrix callbacks, %d Newton iterations, '
                      '%d dense Hessian builds, %.1f s; Hessian at the mode + dense inverse %.1f s; one-off setup of C_big / K_bigInv %.1f s not counted), '
                      'scaled by the trial count (the reference loops trials sequentially)' % (stat['nit'], stat['hess'], t_newton, t_inv, t_setup))
        else:
            per_hess = t_newton / max(stat['hess'], 1)
            per_trial = 15 * per_hess + t_inv
            sample = ('PARTLY MEASURED: the faithful Newton-CG run of 1 trial was cut at the %.0f-s budget after %d Newton iterations / %d dense Hessian '
                      'builds (%.1f s per build incl. its CG steps); per-trial E-step priced as 15 builds (the count measured on the reference, '
                      'BASELINE.md) + the measured Hessian-at-the-mode + dense inverse (%.1f s)' % (budget_s, stat['nit'], stat['hess'], per_hess, t_inv))
        # that trial's share of the M-step: the faithful (C,d) optimisation on the one-trial posterior (cost of an evaluation scales with the
        # number of trials, so R trials cost R times this), the timescale update once
        vsmGP, vsm = orc.marginal_blocks(Sigma, p, T)
        res1 = {'post_mean': [x.reshape(p, T)], 'post_vsm': [vsm], 'post_vsmGP': [vsmGP]}
        t1 = time.time()
        orc.learn_cd(init, Ys, res1, method='TNC')
        t_cd = time.time() - t1
        t1 = time.time()
        orc.learn_tau(init, res1, bin_ms)
        t_tau = time.time() - t1
        mstep_s = t_cd * R + t_tau
        mstep_note = '; M-step: (C,d) by scipy TNC on the one-trial posterior %.2f s (x trial count), timescales by BFGS %.2f s (once)' % (t_cd, t_tau)
        sample += mstep_note
    else:
        # --cpu-estimate: one dense Hessian build (inference.py:50-65) and one dense inverse, scaled by the reference's measured
        # count of 15 Hessian builds per trial at this size
        K = orc.make_K(init['tau'], T, bin_ms)
        C_big, d_big = orc.make_Cd_big(init['C'], init['d'], T)
        K_bigInv = np.linalg.inv(orc.make_K_big(K))
        x = np.zeros(n)
        ybar = Ys[0].reshape(-1)
        t_h = []
        for rep in range(2):                       # two builds at different points: ~13 s of host work
            t1 = time.time()
            H = orc.nlp_big_hess(x, ybar, C_big, d_big, K_bigInv)
            t_h.append(time.time() - t1)
            x = x + 0.05
        t_h = float(np.mean(t_h))
        t1 = time.time()
        np.linalg.inv(H)
        t_i = time.time() - t1
        per_trial = 15 * t_h + t_i
        sample = ('ESTIMATED from 2 dense Hessian builds (%.1f s each) + 1 dense inverse (%.1f s) of one trial; per-trial E-step = '
                  '15 builds + 1 inverse (iteration count measured on the reference, BASELINE.md); M-step not counted' % (t_h, t_i))
    em_iter_s = per_trial * R + mstep_s
    return {'value': 1.0 / em_iter_s, 'unit': 'EM-iterations/s', 'cores': cores, 'kind': 'port', 'sample': sample,
            'estep_s_per_trial': per_trial, 'mstep_s_per_iteration': mstep_s, 'wall_s_spent': time.time() - t0}


def cpu_baseline_loo(params, Y0, bin_ms, y_pred0, searches):
    """cpu_baseline leg of the leave-one-neuron-out workload: the oracle's faithful restatement of
    util.leaveOneOutPrediction (fmin_ncg on the big-matrix callbacks) on the first neurons of trial 0."""
    from oracle import pgpfa_oracle as orc
    import scipy.optimize as op
    C, d = params['C'], params['d']
    q, p = C.shape
    T = Y0.shape[1]
    Y0 = np.asarray(Y0, dtype=np.float64)
    t0 = time.time()
    K_bigInv = np.linalg.inv(orc.make_K_big(orc.make_K(params['tau'], T, bin_ms)))
    worst = 0.0
    for n in range(searches):
        Cw, dw, Yw = np.delete(C, n, 0), np.delete(d, n, 0), np.delete(Y0, n, 0)
        C_big, d_big = orc.make_Cd_big(Cw, dw, T)
        x = op.fmin_ncg(orc.nlp_big, np.zeros(p * T), fprime=orc.nlp_big_grad, fhess=orc.nlp_big_hess,
                        args=(Yw.reshape(-1), C_big, d_big, K_bigInv), disp=False)
        yp = np.exp(C[n] @ x.reshape(p, T) + d[n])
        worst = max(worst, float(np.max(np.abs(yp - y_pred0[n]) / yp)))
    per = (time.time() - t0) / searches
    return {'kind': 'port', 'seconds_per_search': per, 'value': 1.0 / per, 'unit': 'mode searches/s', 'sample': '%d searches of trial 0' % searches,
            'cores': os.cpu_count(), 'max_rel_diff_of_predictions': worst}


def cpu_baseline_dual(q, p, T, params, tau, Y0, bin_ms, budget_s=30.0, evals_per_trial=3800):
    """cpu_baseline leg of the config-5 workload.  The reference cannot run this configuration (its C_big alone is 74.5 GiB, and every dual
    evaluation forms np.diag(lamb), an (m x m) matrix with m = q T = 500 000: 2 TB).  What CAN be measured on the host is the reference's own
    dual cost + gradient (inference.py:187-219, restated operation by operation in the oracle's *_faithful functions) on the first T_s bins of
    trial 0 at the LARGEST T_s whose (m x m) temporaries fit a bounded budget, at two sizes, so that the growth law is measured too
    (2 n m^2 + 2 n^2 m flops: cubic in T at fixed q, p).  The value is that evaluation scaled to T bins by the MEASURED exponent, times the
    evaluations scipy's L-BFGS-B needs per trial under the reference's stopping rule - taken from the device L-BFGS run under the same rule at
    these dimensions (profiles/r03_bench_c5_dual_estep_mixed.json: median 3800) - an EXTRAPOLATION, labelled as such."""
    from oracle import pgpfa_oracle as orc
    C, d = np.asarray(params['C'], dtype=np.float64), np.asarray(params['d'], dtype=np.float64)
    times = {}
    for Ts in (24, 40, 60):
        m = q * Ts
        if 3 * 8.0 * m * m > 24e9:                  # (three m x m temporaries alive at the worst point)
            break
        C_big, d_big = orc.make_Cd_big(C, d, Ts)
        K_big = orc.make_K_big(orc.make_K(tau, Ts, bin_ms))
        Kinv_big = np.linalg.inv(K_big)
        ybar = np.asarray(Y0[:, :Ts], dtype=np.float64).reshape(-1)
        lam = np.zeros(m) + 0.5                    # the reference's start (inference.py:294-297)
        t0 = time.time()
        f = orc.dual_cost_faithful(lam, ybar, C_big, K_big, Kinv_big, d_big)
        g = orc.dual_grad_faithful(lam, ybar, C_big, K_big, Kinv_big, d_big)
        times[Ts] = time.time() - t0
        assert np.isfinite(f) and np.all(np.isfinite(g))
        del C_big, K_big, Kinv_big, g
        if times[Ts] > budget_s / 2:
            break
    sizes = sorted(times)
    Ta, Tb = sizes[-2], sizes[-1]
    expo = float(np.log(times[Tb] / times[Ta]) / np.log(Tb / Ta)) if len(sizes) > 1 and times[Ta] > 0 else 3.0
    t_eval = times[Tb] * (T / Tb) ** expo
    return {'kind': 'port', 'value': 1.0 / (evals_per_trial * t_eval), 'unit': 'trials/s through one whole variational E-step (EXTRAPOLATED)', 'cores': os.cpu_count(),
            'sample': 'one dual cost + gradient of trial 0 by the reference\'s own operations (inference.py:187-219: np.diag(lamb) formed, dense (m x m) products) on its '
                      'first %s bins at %d neurons x %d latents: %s s; scaled to %d bins by the measured exponent %.2f (flop law: 3) = %.3g s per evaluation, times %d '
                      'evaluations per trial (median of the device L-BFGS under the reference\'s stopping rule at these dimensions, round 3); the reference itself '
                      'cannot form its matrices at %d bins (np.diag(lamb) alone: %.1f TB)'
                      % (sizes, q, p, [round(times[k], 2) for k in sizes], T, expo, t_eval, evals_per_trial, T, 8.0 * (q * T) ** 2 / 1e12),
            'seconds_per_evaluation_measured': {str(k): times[k] for k in sizes}, 'measured_exponent': expo, 'seconds_per_evaluation_extrapolated': t_eval,
            'evaluations_per_trial_assumed': evals_per_trial}


def run_loo(args, q, p, T, R):
    """Secondary workload: throughput of leave-one-neuron-out prediction, R*q held-out mode searches
    with the generating parameters.  One JSON line; not the headline metric."""
    import funs
    from funs import _session
    true_params, Ys = synth_shard(q, p, T, R, args.seed, 0)
    exp = Shard(Ys, 10.0)
    params = {k: np.asarray(v, dtype=np.float64) for k, v in true_params.items()}
    sess, _ = _session.session_for(exp, p)
    sess.set_params(params)
    sess.ctx.loo_predict(np.array([0], dtype=np.int32))          # warm-up: workspace, code objects
    t0 = time.time()
    y_pred, err = funs.util.leaveOneOutPrediction(params, exp)
    dt = time.time() - t0
    out = {'metric': 'leave-one-neuron-out mode searches/s', 'value': R * q / dt, 'unit': 'mode searches/s', 'n_gpus': 1,
           'higher_is_better': True, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': '%s leave-one-neuron-out prediction: %d neurons, %d latents, %d bins, %d trials = %d cold mode searches'
                                  % (args.config, q, p, T, R, R * q)},
           'seconds': dt, 'pred_err_mode': err}
    if not args.no_cpu_baseline:
        k = 4 if p * T <= 400 else 2
        out['cpu_baseline'] = cpu_baseline_loo(params, Ys[0], 10.0, y_pred[0], k)
        out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
    print(json.dumps(out))


def spawn_ranks(n, timeout_s=None):
    """`python bench.py --gpus N` with no launcher: start the N ranks as children (this parent never initialises the GPU -
    a process that has must not be replaced by or fork into another program on this pool), relay rank 0's JSON line.
    All children are polled: the first one that exits non-zero (out of memory, bad device) ends the others - which would
    otherwise wait for it in the rendezvous or the first all-reduce for ever - and so does the overall timeout."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    # (default below the 1800-s limit of whoever runs this: a job that cannot finish must end with a message and a non-zero status of its own)
    timeout_s = timeout_s or float(os.environ.get('PGPFA_BENCH_TIMEOUT', '900'))
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + timeout_s
    codes = [None] * n
    why = None
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if failed and why is None:
            why = 'rank %d exited with status %s' % (failed[0], codes[failed[0]])
        if why is None and time.time() > deadline:
            why = 'timeout after %.0f s' % timeout_s
        if why is not None:
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.terminate()
            t_kill = time.time() + 10.0
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=max(0.1, t_kill - time.time()))
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read().decode('utf-8', 'replace'))
    sys.stdout.flush()
    if why is not None:
        sys.stderr.write('bench.py: %s; the other ranks were stopped; exit codes %s\n' % (why, codes))
        sys.exit(1)
    sys.exit(0)


def run_floor(args, q, p, T, R):
    """The other end of the design: the same population with EVERY timescale short (--tau-all, default 30 ms = 3 bins).  The
    pivoted Cholesky of such a Gram matrix has nearly full rank, the low-rank form of the prior buys nothing, and the auto plan must fall back to
    the dense engine (one n x n factorisation + inverse per trial: 0.72 n^3 flops): the FLOOR of the E-step's speed.  EM iterations at fixed
    parameters (E-step warm-started, device M-step) on R trials; also each engine forced, where it can run, to show the plan picked the
    faster one.  One JSON line; not the headline metric."""
    import funs
    from funs import _session
    np.random.seed(args.seed)
    C = np.random.rand(q, p) - 0.5
    d = np.random.rand(q) * (-2) - 1.0
    tau = np.full(p, args.tau_all)
    t = np.arange(T, dtype=np.float64) * 10.0
    L = np.linalg.cholesky(0.999 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / (args.tau_all * 1000.0) ** 2) + 0.001 * np.eye(T))
    rng = np.random.default_rng([args.seed, 0])
    Ys = []
    for _ in range(R):
        X = np.einsum('ts,ks->kt', L, rng.standard_normal((p, T)))
        Ys.append(rng.poisson(np.exp(C @ X + d[:, None])).astype(np.uint8))
    exp = Shard(Ys, 10.0)
    params = {'C': C, 'd': d, 'tau': tau}
    sess, _ = _session.session_for(exp, p)
    res = {}
    for name, mode in (('auto', 0), ('dense', 1), ('lowrank', 2)):
        sess.ctx.set_option('cov_mode', mode)
        par = {k: v.copy() for k, v in params.items()}
        optim, est, mst, plan = None, [], [], None
        try:
            for it in range(args.warmup + args.steps):
                t0 = time.time()
                infRes, nll, optim = funs.inference.laplace(exp, par, prevOptimRes=optim)
                t1 = time.time()
                par, _ = funs.learning.updateParams(par, infRes, exp, CdOptimMethod=args.cd_method)
                t2 = time.time()
                est.append((t1 - t0) * 1e3); mst.append((t2 - t1) * 1e3)
                plan = 'lowrank' if sess.ctx.info('last_cov_lowrank') else 'dense'
            res[name] = {'engine': plan, 'estep_ms': [round(x, 1) for x in est], 'mstep_ms': [round(x, 1) for x in mst], 'lowrank_rtot': sess.ctx.info('lowrank_rtot'),
                         'ms_per_em_iteration': float(np.mean(est[args.warmup:]) + np.mean(mst[args.warmup:])), 'chunk_trials': sess.ctx.info('chunk_trials')}
        except Exception as exc:                                     # (the low-rank engine refuses ranks its buffers cannot hold: that is an answer too)
            res[name] = {'engine': None, 'error': str(exc)[:200]}
    a = res['auto']
    n = p * T
    out = {'metric': 'EM iterations/sec', 'value': 1e3 / a['ms_per_em_iteration'], 'unit': 'EM-iterations/s (%d trials, every timescale %.0f ms: the dense-engine floor)' % (R, args.tau_all * 1e3),
           'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': a['ms_per_em_iteration'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': '%s floor: %d neurons, %d latents, %d bins, %d trials, all timescales %.0f ms (rank %d of n = %d): auto plan -> %s engine'
                                  % (args.config, q, p, T, R, args.tau_all * 1e3, int(a['lowrank_rtot']), n, a['engine'])},
           'engines': res,
           'auto_plan_is_fastest': bool(a['ms_per_em_iteration'] <= 1.1 * min(v['ms_per_em_iteration'] for v in res.values() if v.get('engine'))),
           'dense_engine_tflops': 0.72 * float(n) ** 3 * R / (float(np.mean(a['estep_ms'][args.warmup:])) * 1e-3) / 1e12 if a['engine'] == 'dense' else None}
    print(json.dumps(out))


def pmc_traffic(kernel_substring):
    """HBM bytes per launch of a kernel from the newest committed PMC passes (profiles/rNN_pmc_hbm_traffic.json: separate FETCH_SIZE / WRITE_SIZE
    runs of `bench.py --steps 2 --warmup 1 --lean` under rocprofv3, FETCH doubled per the gfx950 correction - tools/pmc_summary.py).  Counters are not
    collected inside a timed run (--pmc serialises the kernels), and those passes see the first iterations of the fit (the lowest ranks): the figure is
    reported with its source, or None when no such file travels with the tree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))
    if not files:
        return None
    try:
        table = json.load(open(files[-1]))
        for name, row in table.items():
            if kernel_substring in name and 'hbm_bytes_per_launch' in row:
                return {'bytes_per_launch': row['hbm_bytes_per_launch'], 'launches_counted': row.get('calls'),
                        'source': 'profiles/%s (PMC passes over EM iterations 1-3 of this workload: ranks 364-420; not from this run)' % os.path.basename(files[-1])}
    except Exception:
        return None
    return None


def run_online(args, q, p, T, rank, world):
    """BASELINE config 4: stochastic EM (engine.py:288-448, 'diag' updates) - every iteration draws a minibatch from the
    resident trials with the reference's RNG call (util.py:459-473; same stream on every rank), each rank runs the Laplace
    E-step on its slice of the minibatch, the M-step statistics are all-reduced, the prior-regularised M-step
    (learning.py:833-866) runs replicated.  One JSON line."""
    import funs
    from funs import _session
    Rres, batch = args.resident, args.batch
    true_params, Ys = synth_shard(q, p, T, Rres, args.seed, 0)          # same trials on every rank
    exp = Shard(Ys, 10.0)
    exp._pgpfa_local_shard = False                                       # ranks slice each trial list
    sess, _ = _session.session_for(exp, p)
    np.random.seed(0)
    params = funs.util.initializeParams(p, q, exp)
    params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in params.items()}
    np.random.seed(1)
    prior = np.diag(np.ones(q * (p + 1)))
    est, mst, nlls, pcgs, idx_sum = [], [], [], [], []
    cd_method = args.cd_method
    tau_method = [args.tau_method]
    state = {'n': 0}
    nwt = {'ms': [], 'moved': []}
    sess.ctx.set_option('time_newton', 1)

    def step():
        n = state['n']
        sub = funs.util.subsampleTrials(exp, batch)
        t0 = time.time()
        infRes, nll, _ = funs.inference.laplace(sub, params_box[0], prevOptimRes='resident')
        t1 = time.time()
        sz = 1.0 / (n + 1) ** 0.75                                       # engine.py:275-278, stepPow = 0.75
        new, _, pr = funs.learning.updateParamsWithPrior(params_box[0], infRes, sub, cd_method, tau_method[0], sz, sz, prior_box[0], covOpts='useDiag')
        t2 = time.time()
        params_box[0], prior_box[0] = new, pr
        est.append((t1 - t0) * 1e3); mst.append((t2 - t1) * 1e3); nlls.append(float(nll))
        pcgs.append(sess.ctx.info('last_pcg_iterations') / max(1, len(infRes.trial_idx)))
        idx_sum.append(int(np.sum(sub.batchTrIdx)))
        nwt['ms'].append(sess.ctx.info('last_newton_solve_ms')); nwt['moved'].append(sess.ctx.info('last_newton_solve_bytes_moved'))
        state['n'] = n + 1
    params_box, prior_box = [params], [prior]

    def barrier():
        sess.allreduce(np.zeros(1))
    for _ in range(args.warmup):
        step()
    # HIP events around the largest kernel of the step (the fused product + mixing launch of the covariance phase) on every timed step: two events per E-step
    sess.ctx.set_option('profile', 3)
    barrier()
    t_begin = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.time() - t_begin
    ytm_ms, ytm_flops, ytm_launches = (sess.ctx.info('prof_mix_' + k) for k in ('ms', 'flops', 'launches'))
    ytm_fused = sess.ctx.info('last_yt_mix_fused') == 1.0
    sess.ctx.set_option('profile', 0)
    times = np.zeros(world)
    times[rank] = elapsed
    t_max = float(np.max(sess.allreduce(times)))
    # two more (untimed) iterations with the reference engine's default timescale driver (scipy TNC), for the record
    tnc_ms = None
    if world == 1 and not args.lean and args.tau_method != 'TNC':
        tau_method[0] = 'TNC'
        keep = (params_box[0], prior_box[0], state['n'])
        step(); step()
        tnc_ms = mst[-1]
        for lst in (est, mst, nlls, pcgs, idx_sum, nwt['ms'], nwt['moved']):
            del lst[-2:]
        params_box[0], prior_box[0], state['n'] = keep
        tau_method[0] = args.tau_method
    if rank != 0:
        return
    timed = slice(args.warmup, args.warmup + args.steps)
    n_ms, n_moved = float(np.sum(nwt['ms'][timed])), float(np.sum(nwt['moved'][timed]))
    out = {'metric': 'EM iterations/sec', 'value': args.steps / t_max, 'unit': 'stochastic-EM iterations/s (minibatch %d of %d resident trials)' % (batch, Rres),
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t_max / args.steps * 1e3,
           'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': 'c4: %d neurons, %d latents, %d bins, %d resident trials, stochastic EM minibatch=%d split over %d GPU(s), '
                                  "'diag' prior updates ((C,d) by %s, tau by %s)"
                                  % (q, p, T, Rres, batch, world, 'device per-neuron Newton' if cd_method == 'newton' else 'scipy ' + cd_method,
                                     'the 4-point lockstep root finder on the reference gradient' if args.tau_method == 'lockstep' else 'scipy %s as the reference engine' % args.tau_method),
                      'parallelism': 'minibatch-sharded x%d' % world},
           'estep_ms': [round(x, 1) for x in est], 'mstep_ms': [round(x, 1) for x in mst],
           'estep_ms_per_trial': float(np.mean(est[timed])) / (batch / world), 'pcg_iterations_per_trial': [round(x, 1) for x in pcgs],
           'nll': nlls, 'minibatch_index_checksums': idx_sum,
           'lowrank_plan': sess.ctx.info('plan_lowrank'), 'lowrank_rtot': sess.ctx.info('lowrank_rtot'), 'chunk_trials': sess.ctx.info('chunk_trials'),
           # the rate with the reference engine's default timescale driver (tauOptimMethod='TNC': p scipy optimisations, one batched device pass per round)
           'mstep_ms_with_reference_default_tau_TNC': tnc_ms,
           'value_reference_default_tau': None if tnc_ms is None else 1e3 / (float(np.mean(est[timed])) + tnc_ms),
           # dominant kernel of the step, as in the batch workload: product + mixing of the covariance phase (HIP events on the context's stream around it)
           'roofline': None if not (ytm_fused and ytm_ms > 0) else {
               'bound': 'mfma', 'kernel': 'yt_mix_kernel (FP64 16x16x4 MFMA products + FP64 vector mixing in registers; writes the FP32 correction D and post_vsm)',
               'achieved': ytm_flops / (ytm_ms * 1e-3) / 1e12, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
               'frac': ytm_flops / (ytm_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS, 'traffic': None,
               'algorithmic_flops_per_launch': ytm_flops / max(ytm_launches, 1.0), 'launches': ytm_launches, 'avg_launch_ms': ytm_ms / max(ytm_launches, 1.0),
               'kernel_share_of_step': ytm_ms * 1e-3 / t_max},
           'roofline_newton': {'bound': 'hbm', 'bytes_moved': n_moved, 'ms': n_ms, 'achieved': n_moved / (n_ms * 1e-3) / 1e9 if n_ms > 0 else 0.0, 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': n_moved / (n_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if n_ms > 0 else 0.0, 'ms_per_em_iteration': n_ms / args.steps}}
    if not args.no_cpu_baseline and world == 1:
        # the reference's stochastic-EM iteration costs what a batch iteration over the minibatch costs (same E-step loop inference.py:94 over 1024
        # trials, the prior-regularised M-step on their posterior): the faithful CPU trial of the batch workload, scaled to the minibatch
        out['cpu_baseline'] = cpu_baseline(q, p, T, batch, true_params, Ys[0], 10.0, params, estimate=args.cpu_estimate, budget_s=args.cpu_budget)
        out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        out['speedup_vs_cpu_baseline_note'] = 'ratio to the cpu_baseline sample as described there'
    print(json.dumps(out))


def run_dual(args, q, p, T, R, rank, world):
    """BASELINE config 5: the dual-variational E-step (inference.py:259-432) of R trials per GPU at the configuration's dimensions,
    through the low-rank engine with the reference's diagonal jitter - r x r Cholesky, inverse and Yt on the FP32 matrix cores with
    FP64 accumulation ('mixed'), or all FP64.

    Default: WHOLE E-steps - the lockstep device L-BFGS on the dual of every trial run to the reference's stopping rule (scipy
    L-BFGS-B's factr / pgtol), then pgpfa_dual_finalize (posterior means, covariance blocks, the sum of post_vsmGP);
    value = trials/s through the whole E-step over all GPUs, iterations to convergence MEASURED.
    --dual-iters K: K lockstep iterations only (the unit of work: one batched dual cost + gradient per iteration);
    value = trial-evaluations/s.  Trials shard with no data-path collective; the timed region is bracketed by a barrier and the
    reported time is the max over ranks."""
    from funs import _hip, _session
    true_params, Ys = synth_shard(q, p, T, R, args.seed, rank)
    Y = np.stack(Ys)
    rng = np.random.default_rng([args.seed, rank, 5])
    tau = np.linspace(0.1, 0.5, p)
    ctx = _hip.Context(q, p, T, R, 10.0, device=_session.WORLD.device())
    comm = False
    if _session.WORLD.enabled:
        with _session._stdout_to_stderr():
            uid, path = _session.WORLD.exchange_unique_id()
            ctx.comm_init(uid, _session.WORLD.rank, _session.WORLD.size)
            ctx.allreduce_host(np.zeros(1))
        _session.cleanup_rendezvous(path)
        comm = True

    def allreduce(a):
        return ctx.allreduce_host(np.asarray(a, dtype=np.float64)) if comm else np.asarray(a, dtype=np.float64)
    ctx.upload_counts(Y)
    ctx.set_option('cov_mode', 2)
    ctx.set_option('dual_lowrank', 1)
    ctx.set_option('dual_f32', 1 if args.precision == 'mixed' else 0)
    for kv in filter(None, args.opts.split(',')):
        ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    ctx.set_params(true_params['C'], true_params['d'], tau)
    idx = np.arange(R, dtype=np.int32)
    whole = args.dual_iters <= 0
    if whole:
        rho = np.full((R, q * T), np.log(0.5))                    # the reference's start, lambda = 0.5 (inference.py:300-324)
    else:
        rho = np.log(np.exp(true_params['d'])[None, :, None] * (0.5 + rng.random((R, q, T)))).reshape(R, -1)
    fixed_point = whole and args.dual_solver == 'fixedpoint'
    # the evaluations run inside the device drivers (lambda, gradient, modes, offsets and the correction pairs stay resident)
    # Warm-up steps are steps: with the fixed point a warm-up is a WHOLE E-step from the same cold start (lambda = 0.5, modes from zero - nothing
    # of it is reused by the next one), so that the timed E-step is what the second and every later E-step of a fit costs; the first one also
    # pays the context's one-off allocations (2.4 GB of split-sum scratch, the resident dual variables, the single-precision factors: ~150 ms of
    # hipMalloc at these dimensions) - reported as estep_s_first_call.  (Through round 6's first measurement set the warm-up was ONE pass of the
    # fixed point without the finalize call and the timed E-step carried those allocations.)
    first_call_s = None
    for w in range(max(1, args.warmup)):
        if fixed_point:
            tw = time.time()
            st_w = ctx.dual_fixed_point(idx, None, want_rho=False)[3]
            if not np.any(st_w != 0):                     # (a handed-back trial has no resident optimum: the timed step deals with those)
                ctx.dual_finalize(idx, None)
            if w == 0:
                first_call_s = time.time() - tw
        else:
            ctx.dual_lbfgs(idx, rho, max_iter=1)
    allreduce(np.zeros(1))
    t0 = time.time()
    fp_status = None
    if fixed_point:
        # the optimum by the variance fixed point (pgpfa_dual_fixed_point); a trial it hands back would go to L-BFGS, as in
        # inference.dualVariational - counted in the timed region
        # (cold start lambda = 0.5 set on the device; exp / log of the entries run there and the optimum stays resident for the finalize call,
        # as in inference.dualVariational)
        _, fopt, iters, fp_status = ctx.dual_fixed_point(idx, None, want_rho=False)
        bad = np.nonzero(fp_status != 0)[0]
        lam_for_finalize = None
        if len(bad):
            lam_for_finalize = ctx.dual_lambda(idx)
            rho_b, fopt_b, it_b = ctx.dual_lbfgs(idx[bad], np.log(lam_for_finalize[bad]))
            lam_for_finalize[bad], fopt[bad] = np.exp(rho_b), fopt_b
            iters[bad] += it_b
    else:
        rho_opt, fopt, iters = ctx.dual_lbfgs(idx, rho, max_iter=15000 if whole else args.dual_iters)
    t_opt = time.time() - t0
    evals = ctx.info('last_dual_evaluations')
    nlp = None
    if whole:
        nlp = ctx.dual_finalize(idx, lam_for_finalize if fixed_point else np.exp(rho_opt))
    elapsed = time.time() - t0
    allreduce(np.zeros(1))
    times = np.zeros(world)
    times[rank] = elapsed
    t_max = float(np.max(allreduce(times)))
    if rank != 0:
        return
    # roofline of the SAME workload (one more whole E-step, untimed, with HIP events around the tagged launches): the r x r phase of the low-rank
    # engine - factorisation, inverse, Yt, the dual's neuron contractions - on the matrix cores in single precision ('mixed') or FP64, against the
    # dense matrix peak of that type, and the mixing pass over the Yt slab against HBM (bytes: its columns read once, written once)
    roof = None
    if whole and fixed_point and not args.lean:
        ctx.set_option('dual_f32', 1 if args.precision == 'mixed' else 0)
        ctx.set_option('profile', 1)
        ctx.dual_fixed_point(idx, None, want_rho=False)
        ctx.dual_finalize(idx, None)
        shapes = ctx.gemm_shape_report()
        ctx.set_option('profile', 0)
        f32 = args.precision == 'mixed'
        g_ms = g_fl = o_ms = o_fl = 0.0
        for line in shapes.splitlines():
            ms = float(line.split(' ms')[0].split()[-1]); fl = float(line.split(' GFLOP')[0].split()[-1]) * 1e9
            if line.startswith('f32' if f32 else 'f64'):
                g_ms += ms; g_fl += fl
            else:
                o_ms += ms; o_fl += fl
        peak = FP32_MATRIX_PEAK_TFLOPS if f32 else FP64_MATRIX_PEAK_TFLOPS
        mix_ms, mix_by, mix_n = ctx.info('prof_mix_ms'), ctx.info('prof_mix_flops'), ctx.info('prof_mix_launches')
        ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        roof = {'roofline': {'bound': 'mfma', 'kernel': 'gemm_mfma_kernel_t<%s> (r x r factorisation / inverse / Yt products of the low-rank engine%s)'
                                      % ('float' if f32 else 'double', '' if not f32 else '; the FP64 products of the same E-step are listed under other_precision'),
                             'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None, 'ms': g_ms, 'algorithmic_flops': g_fl,
                             'other_precision': {'ms': o_ms, 'algorithmic_flops': o_fl, 'achieved': o_fl / (o_ms * 1e-3) / 1e12 if o_ms > 0 else 0.0,
                                                 'peak': FP64_MATRIX_PEAK_TFLOPS if f32 else FP32_MATRIX_PEAK_TFLOPS},
                             'share_of_estep': g_ms * 1e-3 / t_max},
                'roofline_mixing': {'bound': 'hbm', 'kernel': 'mix_vsm_wide2_kernel<20> (per-bin mixing of the Yt slab: y <- G_t y in place, or the single-precision correction of the split form)',
                                    'achieved': mix_by / (mix_ms * 1e-3) / 1e9 if mix_ms > 0 else 0.0, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                    'frac': mix_by / (mix_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if mix_ms > 0 else 0.0, 'ms': mix_ms, 'bytes': mix_by, 'launches': mix_n},
                'kernel_time_ms_one_untimed_estep': {tag: round(ctx.info('prof_%s_ms' % tag), 1) for tag in ('gemm', 'potrf', 'poisson', 'vsm', 'mix', 'assemble', 'solve')}}
    common = {'n_gpus': world, 'steps': 1 if whole else args.dual_iters, 'warmup': args.warmup, 'higher_is_better': True, 'scaling': 'weak',
              'vs_baseline': None, 'dtype': 'f32 matrix products, f64 accumulation' if args.precision == 'mixed' else 'f64', 'data': 'synthetic',
              'batched_evaluations': evals, 'lbfgs_iterations_max': int(np.max(iters)), 'lbfgs_iterations_median': float(np.median(iters)),
              'lbfgs_iterations_min': int(np.min(iters)), 'dual_objective_mean': float(np.mean(fopt)), 'lowrank_rtot': int(ctx.info('lowrank_rtot')),
              'ms_per_batched_evaluation': t_opt / max(evals, 1) * 1e3,
              'note': 'the reference cannot run this configuration at all (its C_big alone is 74.5 GiB, BASELINE.md)'}
    if whole:
        how = ('variational fixed point on the dual\'s stationarity conditions (Newton-PCG mode search with variance offsets in a loop with the '
               'covariance blocks) to max |dual gradient| <= 1e-8' if fixed_point else
               "lockstep device L-BFGS on the dual to scipy L-BFGS-B's stopping rule (factr 1e7, pgtol 1e-5)")
        if fixed_point:
            # the certificate both solvers are held to: the reference's dual gradient (inference.py:215-219) at the returned lambda, FP64
            ctx.set_option('dual_f32', 0)
            nchk = min(R, 8)
            _, gchk = ctx.dual_costgrad_batch(idx[:nchk], ctx.dual_lambda(idx[:nchk]))
            common = dict(common, solver='fixedpoint', fixed_point_passes_max=int(np.max(iters)), fixed_point_passes_min=int(np.min(iters)),
                          fixed_point_status_counts=[int(v) for v in np.bincount(fp_status, minlength=3)],
                          max_abs_dual_gradient_at_optimum_first_trials=float(np.max(np.abs(gchk))))
            for k in ('lbfgs_iterations_max', 'lbfgs_iterations_median', 'lbfgs_iterations_min', 'ms_per_batched_evaluation'):
                common.pop(k, None)
        out = dict(common, metric='dual-variational E-step trials/sec', value=R * world / t_max, unit='trials/s through one whole variational E-step',
                   ms_per_step=t_max * 1e3, estep_s=t_max, optimiser_s=t_opt, finalize_s=elapsed - t_opt, estep_s_first_call=first_call_s,
                   neg_log_posterior_mean=nlp / R,
                   config={'workload': '%s variational E-step: %d neurons, %d latents, %d bins, %d trials per GPU; %s from lambda = 0.5, then posterior '
                                       'means / covariance blocks; low-rank engine (rank %d) with the reference 1e-6 diagonal jitter'
                                       % (args.config, q, p, T, R, how, int(ctx.info('lowrank_rtot'))),
                           'parallelism': 'trial-sharded x%d' % world})
        if roof:
            out.update(roof)
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline_dual(q, p, T, true_params, tau, Ys[0], 10.0, budget_s=min(args.cpu_budget, 60.0))
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
            out['speedup_vs_cpu_baseline_note'] = 'ratio to an EXTRAPOLATED figure (see cpu_baseline.sample): it says the reference cannot run this configuration, nothing about kernel quality'
    else:
        out = dict(common, metric='dual-variational trial-evaluations/sec', value=evals * R * world / t_max,
                   unit='trial-evaluations/s (dual cost + gradient)', ms_per_step=t_max / max(evals, 1) * 1e3,
                   config={'workload': '%s variational E-step: %d neurons, %d latents, %d bins, %d trials per GPU, %d lockstep L-BFGS iterations of '
                                       'the dual (every iteration = batched dual cost + gradient of all live trials, low-rank engine rank %d)'
                                       % (args.config, q, p, T, R, args.dual_iters, int(ctx.info('lowrank_rtot'))), 'parallelism': 'trial-sharded x%d' % world})
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--precision', default='mixed', choices=['mixed', 'f64'],
                    help="--workload dual: 'mixed' = FP32 matrix cores for the r x r factorisation / inverse / Yt with FP64 accumulation; 'f64'")
    ap.add_argument('--dual-solver', default='fixedpoint', choices=['fixedpoint', 'lbfgs'],
                    help="--workload dual, whole E-steps: the variational fixed point (default since round 4) or the lockstep device L-BFGS")
    ap.add_argument('--trials', type=int, default=0, help='override trials per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cd-method', default='newton', choices=['newton', 'TNC', 'BFGS', 'L-BFGS-B'],
                    help="(C,d) M-step solver: 'newton' = device per-neuron Newton (exact minimiser of the reference's cost); "
                         "'TNC' = the reference engine's default scipy driver on the same device cost/grad")
    ap.add_argument('--tau-method', default='lockstep', choices=['lockstep', 'TNC'],
                    help="--workload online, timescale update with prior: 'lockstep' = the root of the reference's gradient expression for all latents "
                         "together (learning.learnGPparamsWithPrior, opt-in); 'TNC' = the reference engine's default scipy driver")
    ap.add_argument('--seed', type=int, default=12)
    ap.add_argument('--tau-all', type=float, default=0.03, help='--workload floor: every timescale (seconds)')
    ap.add_argument('--workload', default='em', choices=['em', 'online', 'loo', 'dual', 'floor'],
                    help="'em' (default): the headline EM-iterations/s metric (config 3); 'online': config 4, stochastic EM with "
                         "minibatches over a larger resident set; 'loo': leave-one-neuron-out prediction throughput (1 GPU); 'dual': batched "
                         "dual-variational cost + gradient evaluations (the unit of work of config 5's E-step) at --config dimensions")
    ap.add_argument('--cpu-estimate', action='store_true',
                    help='cpu_baseline extrapolated from 2 dense Hessian builds + 1 inverse instead of timing one whole reference-faithful trial')
    ap.add_argument('--cpu-budget', type=float, default=240.0, help='wall-clock bound (s) of the faithful CPU trial')
    ap.add_argument('--dual-iters', type=int, default=0,
                    help='--workload dual: 0 (default) = whole E-steps to convergence; K > 0 = K lockstep L-BFGS iterations (unit of work)')
    ap.add_argument('--resident', type=int, default=8192, help="--workload online: trials resident in HBM (all ranks hold the counts)")
    ap.add_argument('--batch', type=int, default=1024, help="--workload online: minibatch size (split over the ranks)")
    ap.add_argument('--lean', action='store_true', help='skip the extra untimed iterations (per-family event breakdown, TNC M-step, MFMA peak probe): for runs under rocprofv3')
    ap.add_argument('--opts', default='', help='context options key=value,... (experiments)')
    ap.add_argument('--events-every', type=int, default=4, help='HIP events around the GEMM launches on every n-th timed step (1: all of them)')
    ap.add_argument('--dry-run', action='store_true', help='launcher self-test: every rank reports its environment and exits (no GPU work)')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.dry_run:
        # (launcher self-test hooks: a rank that dies at start-up, a rank that never returns)
        if os.environ.get('PGPFA_DRYRUN_DIE') == str(rank):
            sys.exit(3)
        if os.environ.get('PGPFA_DRYRUN_HANG') == str(rank):
            time.sleep(600)
        if os.environ.get('PGPFA_DRYRUN_COMM_HANG') == str(rank):
            # a rank stuck inside pgpfa_comm_init: the same watchdog that guards the real call (funs/_session.py) around a stub that never returns
            sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
            from funs._session import phase_deadline
            with phase_deadline('pgpfa_comm_init (ncclCommInitRank)', float(os.environ.get('PGPFA_COMM_TIMEOUT', '300')), rank):
                time.sleep(600)
        if rank == 0:
            print(json.dumps({'dry_run': True, 'rank': rank, 'world': world, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')),
                              'master': '%s:%s' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')), 'ppid': os.getppid()}))
        return
    q, p, T, R = CONFIGS[args.config]
    if args.trials > 0:
        R = args.trials
    bin_ms = 10.0
    if args.workload == 'loo':
        return run_loo(args, q, p, T, R)
    if args.workload == 'floor':
        return run_floor(args, q, p, T, args.trials if args.trials > 0 else 128)
    if args.workload == 'online':
        return run_online(args, q, p, T, rank, world)
    if args.workload == 'dual':
        return run_dual(args, q, p, T, args.trials if args.trials > 0 else 64, rank, world)
    if args.config == 'c5':
        raise SystemExit("--config c5 is the variational variant: use --workload dual")

    import funs
    from funs import _hip, _session

    true_params, Ys = synth_shard(q, p, T, R, args.seed, rank)
    exp = Shard(Ys, bin_ms)
    sess, _ = _session.session_for(exp, p)
    for kv in filter(None, args.opts.split(',')):
        sess.ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    # Poisson-PCA initialiser on rank 0's shard (reference util.py:505-558), shared with every rank
    np.random.seed(0)
    init = funs.util.initializeParams(p, q, exp)
    if world > 1:
        flat = np.concatenate([init['C'].ravel(), init['d'], init['tau']])
        flat = sess.allreduce(flat if rank == 0 else np.zeros_like(flat))
        init = {'C': flat[:q * p].reshape(q, p), 'd': flat[q * p:q * p + q], 'tau': flat[q * p + q:]}
    init = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in init.items()}

    def barrier():
        sess.allreduce(np.zeros(1))

    params = init
    optim = None
    cd_method = [args.cd_method]
    nll_hist, estep_ms, mstep_ms, facts, solves, pcgs, cdp, nwt_ms, nwt_bytes, ranks = [], [], [], [], [], [], [], [], [], []
    nwt_bytes_survey, nwt_bytes_moved = [], []
    sess.ctx.set_option('time_newton', 1)           # two HIP events per inner solve: the Newton-solve kernels' time, next to their bytes

    def em_step():
        nonlocal params, optim
        t0 = time.time()
        infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
        t1 = time.time()
        params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod=cd_method[0])
        t2 = time.time()
        nll_hist.append(float(nll))
        estep_ms.append((t1 - t0) * 1e3)
        mstep_ms.append((t2 - t1) * 1e3)
        facts.append(sess.ctx.info('last_newton_factorizations'))
        solves.append(sess.ctx.info('last_newton_solves'))
        pcgs.append(sess.ctx.info('last_pcg_iterations'))
        cdp.append(list(getattr(sess, '_cd_passes', (0, 0))))
        nwt_ms.append(sess.ctx.info('last_newton_solve_ms'))
        nwt_bytes.append(sess.ctx.info('last_newton_solve_bytes'))
        nwt_bytes_survey.append(sess.ctx.info('last_newton_solve_bytes_survey'))
        nwt_bytes_moved.append(sess.ctx.info('last_newton_solve_bytes_moved'))
        ranks.append(sess.ctx.info('lowrank_rtot'))

    for _ in range(args.warmup):
        em_step()
    # HIP events (on the context's stream) around the GEMM launches of the timed region - the roofline kernel - on every 4th step: a pair of
    # events costs ~10 us of device time per launch (two barrier packets; tools/probes/launch_probe.hip: a bare launch is 3 us), ~400 launches per
    # step - recorded on every step the measurement itself was 4 % of the number it sits next to
    every = max(1, args.events_every)
    sess.ctx.set_option('profile', 2)
    sess.ctx.set_option('profile_pause', 1)
    event_steps = []
    barrier()
    t_begin = time.time()
    for i in range(args.steps):
        if i % every == 0:
            sess.ctx.set_option('profile_pause', 0)
            event_steps.append(i)
        em_step()
        if i % every == 0:
            sess.ctx.set_option('profile_pause', 1)
    barrier()
    elapsed = time.time() - t_begin
    gemm_ms, gemm_flops, gemm_launches = (sess.ctx.info('prof_gemm_' + k) for k in ('ms', 'flops', 'launches'))
    gemm_max_ms, gemm_max_flops = sess.ctx.info('prof_gemm_max_ms'), sess.ctx.info('prof_gemm_max_flops')
    ytm_ms, ytm_flops, ytm_launches = (sess.ctx.info('prof_mix_' + k) for k in ('ms', 'flops', 'launches'))
    ytm_fused = sess.ctx.info('last_yt_mix_fused') == 1.0
    sess.ctx.set_option('profile', 0)

    def drop_last():
        for lst in (estep_ms, mstep_ms, nll_hist, facts, solves, pcgs, cdp, nwt_ms, nwt_bytes, nwt_bytes_survey, nwt_bytes_moved, ranks):
            lst.pop()

    # one more (untimed) EM iteration with events around every tagged launch: the per-kernel-family breakdown
    prof = {}
    if not args.lean:
        sess.ctx.set_option('profile', 1)
        em_step()
        prof = {tag: sess.ctx.info('prof_%s_ms' % tag) for tag in ('gemm', 'mix', 'cd', 'potrf', 'poisson', 'vsm', 'assemble', 'solve')}
        sess.ctx.set_option('profile', 0)
        drop_last()
    # one more (untimed) EM iteration with the reference engine's default (C,d) driver, for the record
    tnc_ms = None
    if world == 1 and args.cd_method != 'TNC' and not args.lean:
        cd_method[0] = 'TNC'
        em_step()
        tnc_ms = mstep_ms[-1]
        drop_last()
        cd_method[0] = args.cd_method
    # where a fit ends up: the learnt timescales of this synthetic population shorten towards the generating ones over ~45 iterations
    # and the low-rank ranks (so the E-step) grow with them; four (untimed) EM iterations AT the generating parameters give the rate of
    # the settled fit without running it there - the last one is reported (the first two pay for the jump: modes far from the warm
    # start, then an extrapolated start across the jump)
    plateau = None
    if not args.lean and args.config != 'c1':
        keep = (params, optim)
        params = {k: np.asarray(v, dtype=np.float64).copy() for k, v in true_params.items()}
        for _ in range(4):
            em_step()
        plateau = {'estep_ms': estep_ms[-1], 'mstep_ms': mstep_ms[-1], 'lowrank_rtot': ranks[-1], 'pcg_iterations_per_trial': pcgs[-1] / R,
                   'estep_ms_of_the_four': [round(x, 1) for x in estep_ms[-4:]], 'dense_retries': sess.ctx.info('last_dense_retries')}
        for _ in range(4):
            drop_last()
        params, optim = keep
    times = np.zeros(world)
    times[rank] = elapsed
    times = sess.allreduce(times)
    t_max = float(np.max(times))

    if rank != 0:
        return
    total_trials = R * world
    ms_per_step = t_max / args.steps * 1e3
    value = (args.steps * total_trials / 1024.0) / t_max if args.config == 'c3' else args.steps / t_max
    timed = slice(args.warmup, args.warmup + args.steps)
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    warm_e = float(np.mean(estep_ms[timed]))
    n_ms, n_by = float(np.sum(nwt_ms[timed])), float(np.sum(nwt_bytes[timed]))
    n_by_survey = float(np.sum(nwt_bytes_survey[timed]))
    n_by_moved = float(np.sum(nwt_bytes_moved[timed]))
    per_1024 = (total_trials / 1024.0) if args.config == 'c3' else 1.0
    step_ms_on_events = max(1e-9, float(np.sum([estep_ms[args.warmup + i] + mstep_ms[args.warmup + i] for i in event_steps])))
    roof_gemm = {'bound': 'mfma', 'kernel': 'gemm_mfma_kernel (FP64 16x16x4 MFMA: preconditioner applications, prior mat-vecs, factor/inverse/selected products)',
                 'achieved': achieved, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP64_MATRIX_PEAK_TFLOPS,
                 # HBM counters are not collected inside a timed run (rocprofv3 --pmc serialises kernels): null here; the per-round PMC
                 # passes of this command are under profiles/ (rNN_pmc_hbm_traffic.json)
                 'traffic': None,
                 'algorithmic_flops_per_launch': gemm_flops / max(gemm_launches, 1.0), 'launches': gemm_launches, 'avg_launch_ms': gemm_ms / max(gemm_launches, 1.0),
                 # (events on these steps of the timed region only; share = GEMM time of those steps / their wall time)
                 'events_on_steps': event_steps, 'kernel_share_of_step': gemm_ms / step_ms_on_events,
                 # the longest single launch of the timed region
                 'largest_launch': {'ms': gemm_max_ms, 'algorithmic_flops': gemm_max_flops,
                                    'achieved': gemm_max_flops / (gemm_max_ms * 1e-3) / 1e12 if gemm_max_ms > 0 else 0.0,
                                    'frac': gemm_max_flops / (gemm_max_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS if gemm_max_ms > 0 else 0.0}}
    roof_cov = None if not (ytm_fused and ytm_ms > 0) else {
        'bound': 'mfma', 'kernel': 'yt_mix_kernel (FP64 16x16x4 MFMA products + FP64 vector mixing in registers; writes the FP32 correction D and post_vsm)',
        'achieved': ytm_flops / (ytm_ms * 1e-3) / 1e12, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': ytm_flops / (ytm_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
        # HBM bytes per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE): from the committed passes, not from this run - see traffic_source
        'traffic': (pmc_traffic('yt_mix_kernel') or {}).get('bytes_per_launch'), 'traffic_source': (pmc_traffic('yt_mix_kernel') or {}).get('source'),
        'algorithmic_flops_per_launch': ytm_flops / max(ytm_launches, 1.0), 'launches': ytm_launches, 'avg_launch_ms': ytm_ms / max(ytm_launches, 1.0),
        'events_on_steps': event_steps, 'kernel_share_of_step': ytm_ms / step_ms_on_events}
    out = {
        'metric': 'EM iterations/sec',
        'value': value,
        'unit': 'EM-iterations/s (1024-trial batches of 200 neurons x 10 latents x 500 bins)' if args.config == 'c3' else 'EM-iterations/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: %d neurons, %d latents, %d bins, %d trials per GPU, Laplace batch EM (warm-started E-step; M-step: (C,d) by %s, tau by the 4-point lockstep root finder)'
                               % (args.config, q, p, T, R, 'device per-neuron Newton' if args.cd_method == 'newton' else 'scipy ' + args.cd_method), 'trials_total': total_trials, 'parallelism': 'trial-sharded x%d' % world},
        # the rate of a settled fit (see above) and the rate a caller gets with the reference engine's default (C,d) driver (scipy TNC)
        'value_at_plateau': None if plateau is None else per_1024 * 1e3 / (plateau['estep_ms'] + plateau['mstep_ms']),
        'plateau': plateau,
        'value_reference_defaults': None if tnc_ms is None else per_1024 * 1e3 / (warm_e + tnc_ms),
        'estep_ms_per_trial': warm_e / R,
        'estep_ms': [round(x, 1) for x in estep_ms], 'mstep_ms': [round(x, 1) for x in mstep_ms],
        'estep_ms_cold_start': round(estep_ms[0], 1), 'estep_ms_warm_mean': round(warm_e, 1),
        'lowrank_rtot': ranks,
        'factorizations_per_trial': [round(f / R, 2) for f in facts],
        'newton_solves_per_trial': [round(f / R, 2) for f in solves],
        'pcg_iterations_per_trial': [round(f / R, 2) for f in pcgs],
        'cd_newton_passes_full_chord': cdp,
        'nll': nll_hist,
        'mstep_ms_with_reference_default_TNC': tnc_ms,
        'kernel_time_ms_one_untimed_step': {k: round(v, 1) for k, v in prof.items()},
        # The line's `roofline` names the dominant kernel of the step - the top row of the rocprof summary (profiles/rNN_c3_bench_kernel_stats.csv):
        # Yt = F L^-T and its mixing in one launch per E-step (csrc/ytmix.h).  HIP events (on the context's stream) around it on the sampled steps of
        # the timed region; flops = products over the triangular panels + the per-(bin, column) mixing (csrc/cov.hip); traffic: the committed PMC
        # passes (profiles/rNN_pmc_hbm_traffic.json).  The GEMM family (a pool of ~165 launches of a dozen shapes) follows as `roofline_gemm`.
        'roofline': roof_cov if roof_cov is not None else roof_gemm,
        'roofline_gemm': roof_gemm,
        'roofline_covariance': roof_cov,
        # the Newton solve (north star: >= 40 % of the HBM roofline): every inner PCG solve of the timed region between two HIP events.  `frac` prices
        # the bytes the kernels really MOVE (round 6; z, s, p, q, t / y of a solve are stored in single precision - csrc/estep.hip: newton_bytes_moved);
        # the 17-pass FP64 model of rounds 3-5 and SURVEY 8(d)'s B_E = q T + 8 (2 p T + T p^2) bytes per pass per trial are printed next to it so that
        # the fraction cannot move by accounting
        'roofline_newton': {'bound': 'hbm', 'bytes': n_by_moved, 'ms': n_ms, 'achieved': n_by_moved / (n_ms * 1e-3) / 1e9 if n_ms > 0 else 0.0, 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': n_by_moved / (n_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if n_ms > 0 else 0.0,
                            'byte_model': 'bytes moved by the kernels of the step as built (per slot-iteration: n-vector passes at their storage width + packed FP32 curvature; operators once per step)',
                            'bytes_17pass_fp64_model': n_by, 'frac_17pass_fp64_model': n_by / (n_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if n_ms > 0 else 0.0,
                            'bytes_survey_model': n_by_survey, 'frac_survey_model': n_by_survey / (n_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if n_ms > 0 else 0.0,
                            'ms_per_em_iteration': n_ms / args.steps, 'pcg_iterations_per_trial_per_estep': float(np.mean(pcgs[timed])) / R},
    }
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(q, p, T, R, true_params, Ys[0], bin_ms, init, estimate=args.cpu_estimate, budget_s=args.cpu_budget)
        out['speedup_vs_cpu_baseline'] = (args.steps / t_max) / out['cpu_baseline']['value']
        out['speedup_vs_cpu_baseline_note'] = 'ratio to the cpu_baseline sample as described there'
    print(json.dumps(out))


if __name__ == '__main__':
    main()
