#!/usr/bin/env python3
"""Headline benchmark: EM iterations/s of the Poisson-GPFA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c1] [--no-cpu-baseline] [--cpu-trial]
                    [--workload em|online|loo]

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
PMC_SUMMARY = os.path.join(ROOT, 'profiles', 'r02_pmc_hbm_traffic.json')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def pmc_traffic_per_launch(kernel_prefix):
    """HBM bytes per launch of the dominant kernel (all its template instantiations pooled) from the committed
    PMC passes (separate FETCH_SIZE and WRITE_SIZE runs of this same command, FETCH doubled as
    MI355X_MICROARCH.md prescribes); None if absent."""
    try:
        with open(PMC_SUMMARY) as fh:
            d = json.load(fh)
        tot, calls = 0.0, 0
        for k, v in d.items():
            if k.startswith(kernel_prefix):
                tot += v['hbm_bytes_corrected']
                calls += v['calls']
        return tot / calls if calls else None
    except (OSError, ValueError, KeyError):
        return None


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


def cpu_baseline(q, p, T, R, true_params, Y0, bin_ms, init, full_trial=False):
    """Reference-faithful CPU path (the oracle in 'faithful' mode = the reference's big-matrix
    arithmetic and scipy drivers), timed on a bounded sample and scaled to one EM iteration.
    full_trial (--cpu-trial): time ONE whole faithful trial whatever its size (config 3: minutes, ~9 GB)."""
    from oracle import pgpfa_oracle as orc
    import scipy.optimize as op  # noqa: F401
    cores = os.cpu_count()
    n = p * T
    Ys = [Y0.astype(np.float64)]
    t0 = time.time()
    if full_trial and n > 1200:
        orc.laplace(Ys, init, bin_ms, mode='faithful', return_cov=True)
        per_trial = time.time() - t0
        sample = ('MEASURED: 1 full trial of the faithful Laplace E-step (scipy Newton-CG on the big-matrix callbacks + dense inverse, '
                  'inference.py:119-131), %.1f s; scaled by the trial count (the reference loops trials sequentially); M-step not counted' % per_trial)
    elif n <= 1200:
        # small enough to run whole trials: faithful Laplace E-step on a few trials
        ntr = 2 if n > 400 else 8
        Ys = [Y0.astype(np.float64)] * ntr
        orc.laplace(Ys, init, bin_ms, mode='faithful', return_cov=True)
        per_trial = (time.time() - t0) / ntr
        sample = '%d full trials of the faithful Laplace E-step (scipy Newton-CG on the big-matrix callbacks)' % ntr
    else:
        # config 3: one trial is ~4-5 minutes of CPU; time its unit of work instead - one dense Hessian
        # build (inference.py:50-65) and one dense inverse - and scale by the reference's measured count
        # of 15 Hessian builds per trial at this size
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
        sample = ('ESTIMATED from 2 dense Hessian builds (%.1f s each) + 1 dense inverse (%.1f s) of one config-3 trial; per-trial E-step = '
                  '15 builds + 1 inverse (iteration count measured on the reference, BASELINE.md); M-step not counted; '
                  '--cpu-trial times one whole faithful trial instead' % (t_h, t_i))
    em_iter_s = per_trial * R
    return {'value': 1.0 / em_iter_s, 'unit': 'EM-iterations/s', 'cores': cores, 'kind': 'port', 'sample': sample,
            'estep_s_per_trial': per_trial}


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


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: start the N ranks as children (this parent never initialises the GPU -
    a process that has must not be replaced by or fork into another program on this pool), relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [pr.wait() for pr in procs]
    sys.stdout.write(out.decode('utf-8', 'replace'))
    sys.stdout.flush()
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        sys.stderr.write('bench.py: rank exit codes %s\n' % codes)
    sys.exit(1 if worst else 0)


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
    state = {'n': 0}

    def step():
        n = state['n']
        sub = funs.util.subsampleTrials(exp, batch)
        t0 = time.time()
        infRes, nll, _ = funs.inference.laplace(sub, params_box[0], prevOptimRes='resident')
        t1 = time.time()
        sz = 1.0 / (n + 1) ** 0.75                                       # engine.py:275-278, stepPow = 0.75
        new, _, pr = funs.learning.updateParamsWithPrior(params_box[0], infRes, sub, cd_method, 'TNC', sz, sz, prior_box[0], covOpts='useDiag')
        t2 = time.time()
        params_box[0], prior_box[0] = new, pr
        est.append((t1 - t0) * 1e3); mst.append((t2 - t1) * 1e3); nlls.append(float(nll))
        pcgs.append(sess.ctx.info('last_pcg_iterations') / max(1, len(infRes.trial_idx)))
        idx_sum.append(int(np.sum(sub.batchTrIdx)))
        state['n'] = n + 1
    params_box, prior_box = [params], [prior]

    def barrier():
        sess.allreduce(np.zeros(1))
    for _ in range(args.warmup):
        step()
    barrier()
    t_begin = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.time() - t_begin
    times = np.zeros(world)
    times[rank] = elapsed
    t_max = float(np.max(sess.allreduce(times)))
    if rank != 0:
        return
    timed = slice(args.warmup, args.warmup + args.steps)
    out = {'metric': 'EM iterations/sec', 'value': args.steps / t_max, 'unit': 'stochastic-EM iterations/s (minibatch %d of %d resident trials)' % (batch, Rres),
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t_max / args.steps * 1e3,
           'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': 'c4: %d neurons, %d latents, %d bins, %d resident trials, stochastic EM minibatch=%d split over %d GPU(s), '
                                  "'diag' prior updates ((C,d) by %s, tau by scipy TNC as the reference engine)"
                                  % (q, p, T, Rres, batch, world, 'device per-neuron Newton' if cd_method == 'newton' else 'scipy ' + cd_method),
                      'parallelism': 'minibatch-sharded x%d' % world},
           'estep_ms': [round(x, 1) for x in est], 'mstep_ms': [round(x, 1) for x in mst],
           'estep_ms_per_trial': float(np.mean(est[timed])) / (batch / world), 'pcg_iterations_per_trial': [round(x, 1) for x in pcgs],
           'nll': nlls, 'minibatch_index_checksums': idx_sum,
           'lowrank_plan': sess.ctx.info('plan_lowrank'), 'lowrank_rtot': sess.ctx.info('lowrank_rtot'), 'chunk_trials': sess.ctx.info('chunk_trials')}
    print(json.dumps(out))


def run_dual(args, q, p, T, R, rank, world):
    """BASELINE config 5's unit of work: one batched evaluation of the dual objective and its gradient (inference.py:196-219) for R
    trials per GPU at the configuration's dimensions, lambda resident in HBM, through the low-rank engine - r x r Cholesky, inverse
    and Yt on the FP32 matrix cores with FP64 accumulation ('mixed'), or all FP64.  A step = one batched evaluation (what one L-BFGS
    iteration of every trial costs); value = trial-evaluations per second over all GPUs (trials shard, no collective)."""
    from funs import _hip, _session
    true_params, Ys = synth_shard(q, p, T, R, args.seed, rank)
    Y = np.stack(Ys)
    rng = np.random.default_rng([args.seed, rank, 5])
    tau = np.linspace(0.1, 0.5, p)
    ctx = _hip.Context(q, p, T, R, 10.0, device=_session.WORLD.device())
    ctx.upload_counts(Y)
    ctx.set_option('cov_mode', 2)
    ctx.set_option('dual_lowrank', 1)
    ctx.set_option('dual_f32', 1 if args.precision == 'mixed' else 0)
    for kv in filter(None, args.opts.split(',')):
        ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    ctx.set_params(true_params['C'], true_params['d'], tau)
    idx = np.arange(R, dtype=np.int32)
    rho = np.log(np.exp(true_params['d'])[None, :, None] * (0.5 + rng.random((R, q, T)))).reshape(R, -1)
    # the evaluations run inside the device L-BFGS (lambda, gradient and the correction pairs stay resident): time its iterations
    for _ in range(max(1, args.warmup)):
        ctx.dual_lbfgs(idx, rho, max_iter=1)
    t0 = time.time()
    _, fopt, iters = ctx.dual_lbfgs(idx, rho, max_iter=args.steps)
    elapsed = time.time() - t0
    evals = ctx.info('last_dual_evaluations')
    t_max = elapsed                            # (ranks are independent replicas of the shard loop, no communicator: rank 0's clock)
    if rank != 0:
        return
    out = {'metric': 'dual-variational trial-evaluations/sec', 'value': evals * R * world / t_max, 'unit': 'trial-evaluations/s (dual cost + gradient)',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t_max / max(evals, 1) * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 matrix products, f64 accumulation' if args.precision == 'mixed' else 'f64', 'data': 'synthetic',
           'config': {'workload': '%s variational E-step: %d neurons, %d latents, %d bins, %d trials per GPU, %d lockstep L-BFGS iterations of '
                                  'the dual (every iteration = batched dual cost + gradient of all live trials, low-rank engine rank %d)'
                                  % (args.config, q, p, T, R, args.steps, int(ctx.info('lowrank_rtot'))), 'parallelism': 'trial-sharded x%d' % world},
           'batched_evaluations': evals, 'lbfgs_iterations': int(np.max(iters)), 'dual_objective_mean': float(np.mean(fopt)),
           'note': 'the reference cannot run this configuration at all (its C_big alone is 74.5 GiB, BASELINE.md); a full E-step needs '
                   'O(1000) such iterations per trial'}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--precision', default='mixed', choices=['mixed', 'f64'],
                    help="--workload dual: 'mixed' = FP32 matrix cores for the r x r factorisation / inverse / Yt with FP64 accumulation; 'f64'")
    ap.add_argument('--trials', type=int, default=0, help='override trials per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cd-method', default='newton', choices=['newton', 'TNC', 'BFGS', 'L-BFGS-B'],
                    help="(C,d) M-step solver: 'newton' = device per-neuron Newton (exact minimiser of the reference's cost); "
                         "'TNC' = the reference engine's default scipy driver on the same device cost/grad")
    ap.add_argument('--seed', type=int, default=12)
    ap.add_argument('--workload', default='em', choices=['em', 'online', 'loo', 'dual'],
                    help="'em' (default): the headline EM-iterations/s metric (config 3); 'online': config 4, stochastic EM with "
                         "minibatches over a larger resident set; 'loo': leave-one-neuron-out prediction throughput (1 GPU); 'dual': batched "
                         "dual-variational cost + gradient evaluations (the unit of work of config 5's E-step) at --config dimensions")
    ap.add_argument('--cpu-trial', action='store_true',
                    help='cpu_baseline times ONE whole reference-faithful trial (config 3: ~5 minutes, ~9 GB) instead of the bounded sample')
    ap.add_argument('--resident', type=int, default=8192, help="--workload online: trials resident in HBM (all ranks hold the counts)")
    ap.add_argument('--batch', type=int, default=1024, help="--workload online: minibatch size (split over the ranks)")
    ap.add_argument('--lean', action='store_true', help='skip the extra untimed iterations (per-family event breakdown, TNC M-step, MFMA peak probe): for runs under rocprofv3')
    ap.add_argument('--opts', default='', help='context options key=value,... (experiments)')
    ap.add_argument('--dry-run', action='store_true', help='launcher self-test: every rank reports its environment and exits (no GPU work)')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.dry_run:
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
    nll_hist, estep_ms, mstep_ms, facts, solves, pcgs, cdp = [], [], [], [], [], [], []

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

    for _ in range(args.warmup):
        em_step()
    # HIP events (on the context's stream) around every GEMM launch of the timed region - the roofline kernel
    sess.ctx.set_option('profile', 2)
    barrier()
    t_begin = time.time()
    for _ in range(args.steps):
        em_step()
    barrier()
    elapsed = time.time() - t_begin
    gemm_ms, gemm_flops, gemm_launches = (sess.ctx.info('prof_gemm_' + k) for k in ('ms', 'flops', 'launches'))
    gemm_max_ms, gemm_max_flops = sess.ctx.info('prof_gemm_max_ms'), sess.ctx.info('prof_gemm_max_flops')
    sess.ctx.set_option('profile', 0)

    def drop_last():
        estep_ms.pop(); mstep_ms.pop(); nll_hist.pop(); facts.pop(); solves.pop(); pcgs.pop(); cdp.pop()

    # one more (untimed) EM iteration with events around every tagged launch: the per-kernel-family breakdown
    prof = {}
    if not args.lean:
        sess.ctx.set_option('profile', 1)
        em_step()
        prof = {tag: sess.ctx.info('prof_%s_ms' % tag) for tag in ('gemm', 'cd', 'potrf', 'poisson', 'vsm', 'assemble', 'solve')}
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
    sustained = sess.ctx.bench_mfma_peak(20000) if (rank == 0 and not args.lean) else None
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
    out = {
        'metric': 'EM iterations/sec',
        'value': value,
        'unit': 'EM-iterations/s (1024-trial batches of 200 neurons x 10 latents x 500 bins)' if args.config == 'c3' else 'EM-iterations/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: %d neurons, %d latents, %d bins, %d trials per GPU, Laplace batch EM (warm-started E-step; M-step: (C,d) by %s, tau by the 4-point lockstep root finder)'
                               % (args.config, q, p, T, R, 'device per-neuron Newton' if args.cd_method == 'newton' else 'scipy ' + args.cd_method), 'trials_total': total_trials, 'parallelism': 'trial-sharded x%d' % world},
        'estep_ms_per_trial': float(np.mean(estep_ms[timed])) / R,
        'estep_ms': [round(x, 1) for x in estep_ms], 'mstep_ms': [round(x, 1) for x in mstep_ms],
        'estep_ms_cold_start': round(estep_ms[0], 1), 'estep_ms_warm_mean': round(float(np.mean(estep_ms[timed])), 1),
        'factorizations_per_trial': [round(f / R, 2) for f in facts],
        'newton_solves_per_trial': [round(f / R, 2) for f in solves],
        'pcg_iterations_per_trial': [round(f / R, 2) for f in pcgs],
        'cd_newton_passes_full_chord': cdp,
        'nll': nll_hist,
        'mstep_ms_with_reference_default_TNC': tnc_ms,
        'kernel_time_ms_one_untimed_step': {k: round(v, 1) for k, v in prof.items()},
        'roofline': {'bound': 'mfma', 'kernel': 'gemm_mfma_kernel (FP64 16x16x4 MFMA: preconditioner applications, prior mat-vecs, factor/inverse/selected products)',
                     'measured_sustained_mfma_tflops': sustained,
                     'achieved': achieved, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP64_MATRIX_PEAK_TFLOPS,
                     'traffic': pmc_traffic_per_launch('void pgpfa::gemm_mfma_kernel') if args.config == 'c3' else None,
                     'traffic_unit': 'HBM bytes per launch',
                     'traffic_source': 'NOT measured in this run: committed rocprofv3 PMC passes of this command (%s), pooled over the GEMM instantiations' % os.path.relpath(PMC_SUMMARY, ROOT),
                     'algorithmic_flops_per_launch': gemm_flops / max(gemm_launches, 1.0), 'launches': gemm_launches, 'avg_launch_ms': gemm_ms / max(gemm_launches, 1.0),
                     'kernel_share_of_step': gemm_ms / (t_max * 1e3),
                     # the longest single launch of the timed region (config 3: the segmented-K product sum_r Y~ Y~^T)
                     'largest_launch': {'ms': gemm_max_ms, 'algorithmic_flops': gemm_max_flops,
                                        'achieved': gemm_max_flops / (gemm_max_ms * 1e-3) / 1e12 if gemm_max_ms > 0 else 0.0,
                                        'frac': gemm_max_flops / (gemm_max_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS if gemm_max_ms > 0 else 0.0}},
    }
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(q, p, T, R, true_params, Ys[0], bin_ms, init, full_trial=args.cpu_trial)
        out['speedup_vs_cpu_baseline'] = (args.steps / t_max) / out['cpu_baseline']['value']
        out['speedup_vs_cpu_baseline_note'] = 'ratio to the cpu_baseline sample as described there (E-step only on the CPU side)'
    print(json.dumps(out))


if __name__ == '__main__':
    main()
