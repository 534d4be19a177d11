"""E-step entry points with the reference's call surface (funs/inference.py of
mackelab/poisson-gpfa), executed by hand-written HIP kernels on MI355X.

    laplace(experiment, params, prevOptimRes=None, returnOptimRes=True, verbose=False,
            optimMethod='Newton-CG')                         # reference inference.py:67-185
    negLogPosteriorUnNorm / _grad / _hess                     # reference inference.py:12-65

The three callbacks keep the reference's positional signature (xbar, ybar, C_big, d_big,
K_bigInv, xdim, ydim) but are also offered in the structured form the device evaluates
(`laplace_objective`), because the Kronecker "big" matrices are never formed here.
"""
import numpy as np

from . import _hip
from ._session import DeviceInfRes, DeviceOptimRes, session_for

STATUS_TEXT = {0: 'converged', 1: 'iteration limit reached', 2: 'line search failed', 3: 'Hessian not positive definite'}


def _prepare(experiment, params):
    C = np.asarray(params['C'], dtype=np.float64)
    ydim, xdim = C.shape
    # the reference flattens tau in place inside util.makeK_big (util.py:602); keep that side effect
    params['tau'] = np.ndarray.flatten(np.asarray(params['tau'], dtype=np.float64))
    sess, trial_idx = session_for(experiment, xdim)
    if sess.q != ydim:
        raise ValueError("params['C'] has %d rows but the experiment has %d neurons" % (ydim, sess.q))
    sess.set_params(params)
    return sess, trial_idx


def laplace(experiment, params, prevOptimRes=None, returnOptimRes=True, verbose=False, optimMethod='Newton-CG'):
    """Laplace approximation of every trial's latent posterior (reference inference.py:67-185).

    Returns (infRes, -mean objective at the modes[, lapOptimRes]) exactly like the reference.  The
    mode search is a batched inexact Newton iteration on the GPU (shared-preconditioner PCG),
    run until the predicted error of the mode is below 1e-9 (far tighter than the reference's scipy Newton-CG
    stop, xtol=1e-5 on the mean |step|); `optimMethod` is accepted for signature compatibility only.  infRes
    entries are lazy device-backed sequences; 'post_cov' and 'post_vsmGP' are rebuilt on access.
    prevOptimRes: the previous call's lapOptimRes (or host arrays) as in the reference; additionally the string
    'resident' starts every trial from the mode an earlier E-step left on the device, if any (minibatch EM).
    """
    sess, trial_idx = _prepare(experiment, params)
    n_all = len(trial_idx)
    # default: every rank holds the same experiment and takes a contiguous slice of its trials;
    # experiment._pgpfa_local_shard = True says this rank's experiment already IS its shard
    local_shard = bool(getattr(experiment, '_pgpfa_local_shard', False))
    lo, hi = (0, n_all) if local_shard else sess.local_slice(n_all)
    mine = trial_idx[lo:hi]
    warm = False
    if isinstance(prevOptimRes, str):
        if prevOptimRes != 'resident':
            raise ValueError("prevOptimRes: expected a sequence of modes or 'resident'")
        warm = 'resident'
    elif prevOptimRes is not None:
        resident = (isinstance(prevOptimRes, DeviceOptimRes) and prevOptimRes.session is sess
                    and prevOptimRes.stamp == sess.mode_stamp and np.array_equal(prevOptimRes.trial_idx, mine))
        if not resident:
            if len(prevOptimRes) == n_all:
                X = np.stack([np.asarray(prevOptimRes[i], dtype=np.float64).reshape(-1) for i in range(lo, hi)])
            elif len(prevOptimRes) == len(mine):
                X = np.stack([np.asarray(x, dtype=np.float64).reshape(-1) for x in prevOptimRes])
            else:
                raise ValueError('prevOptimRes has %d entries for %d trials' % (len(prevOptimRes), n_all))
            if len(mine):
                sess.ctx.set_modes(mine, X)
        warm = True
    if len(mine):
        obj, iters, status = sess.ctx.estep_laplace(mine, warm_start=warm)
    else:
        obj, iters, status = 0.0, np.zeros(0, np.int32), np.zeros(0, np.int32)
    sess.post_stamp += 1
    sess.mode_stamp += 1
    if verbose:
        for i, (it, st) in enumerate(zip(iters, status)):
            print('laplace inference trajectory of trial %d: %d Newton factorizations, %s' % (lo + i + 1, it, STATUS_TEXT.get(int(st), '?')))
    tot = sess.allreduce(np.array([obj, float(len(mine))]))
    post_lik = tot[0] / tot[1]
    infRes = DeviceInfRes(sess, mine, (lo, hi))
    infRes.newton_iters = iters
    infRes.newton_status = status
    if returnOptimRes:
        return infRes, -post_lik, DeviceOptimRes(sess, mine)
    return infRes, -post_lik


# ------------------------------------------------------------------------------------------------
# callbacks (reference inference.py:12-65) - structured form
# ------------------------------------------------------------------------------------------------
def laplace_objective(experiment, params, X, trials=None, grad=True):
    """negLogPosteriorUnNorm (and _grad) at latent trajectories X[n][xdim][T] for the given trials."""
    sess, trial_idx = _prepare(experiment, params)
    idx = trial_idx if trials is None else trial_idx[np.asarray(trials)]
    return sess.ctx.laplace_eval(idx, X, want_grad=grad)


def laplace_hessian(experiment, params, X, trial=0):
    """negLogPosteriorUnNorm_hess for one trial: dense (xdim*T, xdim*T), latent-major."""
    sess, trial_idx = _prepare(experiment, params)
    return sess.ctx.laplace_hessian(int(trial_idx[trial]), X)


# 'device': the dual optimisations of all trials run as lockstep L-BFGS on the GPU (pgpfa_dual_lbfgs); 'scipy': the
# reference's per-trial scipy L-BFGS-B calls (same options), driven concurrently with batched device evaluations
DUAL_SOLVER = 'device'
DUAL_LOWRANK = False


class _ConcurrentProblems:
    """n independent optimisations, one Python thread each, whose cost/gradient requests are gathered into one batched
    device evaluation per round.  Every optimiser sees exactly the call sequence it would see running alone."""

    def __init__(self, n, evaluate_batch):
        import threading
        self._threading = threading
        self.n = n
        self.evaluate_batch = evaluate_batch          # (positions, [x...]) -> [(cost, grad)...]
        self.cond = threading.Condition()
        self.pending, self.results = {}, {}
        self.active = set(range(n))
        self.generation = 0
        self.error = None
        self.rounds = 0

    def _flush(self):                                 # lock held by the caller
        which = sorted(self.pending)
        try:
            res = self.evaluate_batch(which, [self.pending[i] for i in which])
            self.results = dict(zip(which, res))
        except Exception as exc:                      # hand the failure to every waiting optimiser
            self.error = exc
            self.results = {}
        self.pending = {}
        self.rounds += 1
        self.generation += 1
        self.cond.notify_all()

    def request(self, idx, x):
        with self.cond:
            self.pending[idx] = x
            gen = self.generation
            if len(self.pending) == len(self.active):
                self._flush()
            else:
                while self.generation == gen:
                    self.cond.wait()
            if self.error is not None:
                raise self.error
            return self.results[idx]

    def run(self, solve_one):
        out = [None] * self.n

        def worker(idx):
            try:
                out[idx] = solve_one(idx, lambda x: self.request(idx, x))
            except Exception as exc:
                out[idx] = exc
            finally:
                with self.cond:
                    self.active.discard(idx)
                    if self.pending and len(self.pending) == len(self.active):
                        self._flush()
        import sys
        threads = [self._threading.Thread(target=worker, args=(i,)) for i in range(self.n)]
        # many runnable threads hand the interpreter lock around once per switch interval (5 ms by default): with dozens of
        # optimisers that, not the device, would set the time of a round
        interval = sys.getswitchinterval()
        sys.setswitchinterval(1e-4)
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            sys.setswitchinterval(interval)
        for r in out:
            if isinstance(r, Exception):
                raise r
        return out


def dualVariational(experiment, params, optimizeLogLambda=False, prevOptimRes=None, returnOptimRes=True, verbose=False):
    """Dual variational E-step (reference inference.py:259-432).

    Per trial the dual objective over lambda in R^{ydim*T} (inference.py:196-219; log-lambda variant
    :222-256) is minimised by the same scipy L-BFGS-B calls as the reference (bounds lambda >= 1e-10,
    factr=1e7, start 0.5; or unbounded in rho = log lambda from 0); every cost/gradient evaluation runs on
    the GPU in structured form: W_t = C^T diag(lambda_t) C, one Cholesky of the (xdim*T)^2 precision for the
    log-determinant, its inverse's per-bin blocks for c_n^T Sigma_t c_n - C_big and diag(lambda) are never formed.
    Returns (infRes, -mean negLogPosterior at the VI means, mean dual optimum[, varOptimRes]).
    """
    import scipy.optimize as op
    sess, trial_idx = _prepare(experiment, params)
    n_all = len(trial_idx)
    local_shard = bool(getattr(experiment, '_pgpfa_local_shard', False))
    lo, hi = (0, n_all) if local_shard else sess.local_slice(n_all)
    mine = trial_idx[lo:hi]
    m = sess.q * sess.T
    ctx = sess.ctx
    # DUAL_LOWRANK lets the device solver evaluate the dual through the low-rank covariance engine when that pays (large
    # xdim*T): it is the dual WITHOUT the reference's 1e-6 diagonal jitter (inference.py:190), which on stiff GP priors
    # moves the posterior covariance blocks by up to ~1 % - hence off by default
    ctx.set_option('dual_lowrank', 1 if (DUAL_SOLVER == 'device' and DUAL_LOWRANK) else 0)
    if DUAL_SOLVER == 'device' and len(mine):
        # all trials in lockstep on the device, in rho = log(lambda); same optimum as either of the reference's variants
        if prevOptimRes is None:
            rho0 = np.zeros((len(mine), m)) if optimizeLogLambda else np.full((len(mine), m), np.log(0.5))
        else:
            prev = np.stack([np.asarray(prevOptimRes[j] if len(prevOptimRes) == len(mine) else prevOptimRes[lo + j], dtype=np.float64)
                             for j in range(len(mine))])
            rho0 = prev if optimizeLogLambda else np.log(np.maximum(prev, 1e-300))
        rho, fopt, iters = ctx.dual_lbfgs(mine, rho0)
        lam_all = np.exp(rho)
        optim = list(rho) if optimizeLogLambda else list(lam_all)
        nlp = ctx.dual_finalize(mine, lam_all)
        sess.post_stamp += 1
        sess.mode_stamp += 1
        tot = sess.allreduce(np.array([nlp, float(np.sum(fopt)), float(len(mine))]))
        infRes = DeviceInfRes(sess, mine, (lo, hi))
        infRes.dual_iterations = iters
        if returnOptimRes:
            return infRes, -tot[0] / tot[2], tot[1] / tot[2], optim
        return infRes, -tot[0] / tot[2], tot[1] / tot[2]
    # DUAL_SOLVER == 'scipy':
    # The reference solves the trials one after the other (inference.py:300-397), each with its own scipy L-BFGS-B run.
    # Here every trial still gets exactly that run (same calls, same options, same start), but the runs execute
    # concurrently - one Python thread per trial - and each round of their cost/gradient requests is served by ONE
    # batched device evaluation (pgpfa_dual_costgrad_batch), instead of one dense factorisation at a time.
    starts = []
    for j in range(len(mine)):
        if prevOptimRes is None:
            starts.append(np.zeros(m) if optimizeLogLambda else np.zeros(m) + 0.5)
        else:
            starts.append(np.asarray(prevOptimRes[j] if len(prevOptimRes) == len(mine) else prevOptimRes[lo + j], dtype=np.float64))

    def evaluate_batch(which, xs):
        """which: positions in `mine`; xs: their optimiser variables -> list of (cost, grad in the optimiser's variable)"""
        X = np.stack(xs)
        lam = np.exp(X) if optimizeLogLambda else X
        cost, grad = ctx.dual_costgrad_batch(mine[np.asarray(which)], lam)
        if optimizeLogLambda:
            grad = grad * lam
        return [(float(cost[i]), grad[i]) for i in range(len(which))]

    def solve_one(j, evaluate):
        if verbose:
            print('dual variational inference trajectory of trial %d...' % (lo + j + 1))
        cache = {}

        def cached(x):
            key = x.tobytes()
            if cache.get('k') != key:
                cache['k'], cache['v'] = key, evaluate(np.array(x, dtype=np.float64))
            return cache['v']
        if optimizeLogLambda:
            return op.fmin_l_bfgs_b(func=lambda x: cached(x)[0], x0=starts[j], fprime=lambda x: cached(x)[1], disp=False)
        return op.fmin_l_bfgs_b(func=lambda x: cached(x)[0], x0=starts[j], fprime=lambda x: cached(x)[1], approx_grad=False,
                                bounds=[(1e-10, None)] * m, factr=1e7, disp=False)

    outs = _ConcurrentProblems(len(mine), evaluate_batch).run(solve_one) if len(mine) else []
    optim = [out[0] for out in outs]
    lams = [np.exp(out[0]) if optimizeLogLambda else out[0] for out in outs]
    vlb = float(sum(out[1] for out in outs))
    nlp = ctx.dual_finalize(mine, np.stack(lams)) if len(mine) else 0.0
    sess.post_stamp += 1
    sess.mode_stamp += 1
    tot = sess.allreduce(np.array([nlp, vlb, float(len(mine))]))
    infRes = DeviceInfRes(sess, mine, (lo, hi))
    if returnOptimRes:
        return infRes, -tot[0] / tot[2], tot[1] / tot[2], optim
    return infRes, -tot[0] / tot[2], tot[1] / tot[2]
