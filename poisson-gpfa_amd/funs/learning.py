"""M-step entry points with the reference's call surface (funs/learning.py of
mackelab/poisson-gpfa).  Cost and gradient evaluations run as HIP kernels over the E-step results
that are already resident in HBM; the outer optimisers are the same scipy.optimize drivers, called
with the same methods and options as the reference, so early-stopping behaviour carries over.

    updateParams(oldParams, infRes, experiment, CdOptimMethod='BFGS', CdMaxIter=None,
                 tauMaxIter=None, verbose=False)                       # reference learning.py:295-309
    updateParamsWithPrior(...)  with covOpts='useDiag'                   # reference learning.py:833-866
"""
import math
import os
import numpy as np
import scipy.optimize as op

from . import _hip
from . import util
from ._session import DeviceInfRes, session_for

EPS_NOISE = 0.001      # reference learning.py:286,822


def _resident_session(infRes, experiment, xdim):
    """Make sure the posterior the caller passes is the one resident on the device."""
    if isinstance(infRes, DeviceInfRes):
        sess = infRes.session
        if infRes.stamp != sess.post_stamp:
            raise _hip.HipBackendError('this infRes belongs to a superseded E-step; run the M-step right after '
                                       'the E-step that produced it (device results are overwritten in place)')
        return sess
    # foreign infRes (e.g. produced by the reference): upload it
    sess, trial_idx = session_for(experiment, xdim)
    lo, hi = sess.local_slice(len(trial_idx))
    pm = np.stack([np.asarray(infRes['post_mean'][i]) for i in range(lo, hi)])
    pv = np.stack([np.asarray(infRes['post_vsm'][i]) for i in range(lo, hi)])
    pg = np.stack([np.asarray(infRes['post_vsmGP'][i]) for i in range(lo, hi)])
    sess.ctx.set_posterior(trial_idx[lo:hi], pm, pv, pg)
    sess.mark_written(trial_idx[lo:hi])
    return sess


def _trial_key(infRes):
    """Identity of the trial list an infRes covers (None for a foreign infRes: uploaded whole)."""
    idx = getattr(infRes, 'trial_idx', None)
    return None if idx is None else np.asarray(idx).tobytes()


class _CostGradCache:
    """scipy calls fun and jac separately at the same point: evaluate once on the device."""

    def __init__(self, evaluate):
        self._evaluate = evaluate
        self._x = None
        self._val = None
        self.n_eval = 0

    def _get(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self._x is None or not np.array_equal(x, self._x):
            self._val = self._evaluate(x)
            self._x = x.copy()
            self.n_eval += 1
        return self._val

    def fun(self, x, *args):
        return self._get(x)[0]

    def jac(self, x, *args):
        return self._get(x)[1]


# ------------------------------------------------------------------------------------------------
# (C, d)
# ------------------------------------------------------------------------------------------------
def MStepObservationCost(vecCd, xdim, ydim, experiment, infRes):
    """reference learning.py:20-49"""
    sess = _resident_session(infRes, experiment, xdim)
    return sess.ctx.mstep_cd_costgrad(vecCd)[0]


def MStepObservationCost_grad(vecCd, xdim, ydim, experiment, infRes):
    """reference learning.py:51-91"""
    sess = _resident_session(infRes, experiment, xdim)
    return sess.ctx.mstep_cd_costgrad(vecCd)[1]


CD_EXTRAPOLATE = 2         # start the (C,d) Newton iteration of batch EM one previous displacement ahead (1), continuing the trend of the last two (2); 0: at the parameters handed in (see _newton_cd)


def _newton_cd(sess, x0, prior_center=None, inv_s2=0.0, max_iter=50, xtol=1e-10, verbose=False, hess_key=None, extrapolate=False):
    """Exact minimiser of the (C,d) cost by q independent damped Newton iterations (the cost is separable over
    neurons and convex in each (c_n, d_n)); every iteration is one device pass that returns per-neuron cost,
    step and decrement at the current point.  Each neuron backtracks on its own cost.

    A pass is either FULL (cost, gradient and per-neuron Hessians: ~3.4x the cost of a gradient sweep) or CHORD
    (cost and gradient only; the step uses the Hessians of the last full pass, which stay resident on the device
    - also across EM iterations).  With rho the relative staleness of those Hessians a chord step leaves an
    error ~rho*|step|, a full step ~|step|^2; the driver picks the cheaper pass that still contracts fast and
    stops as soon as the predicted error of the next iterate is below xtol (that last step is taken without a
    confirming pass).  hess_key identifies the trial list the cost sums over: Hessians left by a pass over another
    list (another minibatch) are not trusted as a chord, the first pass is then a full one."""
    ctx = sess.ctx
    if getattr(sess, '_cd_hess_key', None) != hess_key:
        sess._cd_hess_resident = False
    q, D = sess.q, sess.p + 1
    RHO_OLD = 0.1                      # Hessians of the previous EM iteration (other posterior moments)
    x_in = np.array(x0, dtype=np.float64).reshape(-1)
    theta = x_in.reshape(D, q).copy()
    # EM moves (C,d) by nearly the same displacement from one iteration to the next: where the caller hands back exactly the optimum of the
    # previous M-step (same trial list, same cost), the iteration starts one such displacement further on.  The minimiser is the same (the
    # cost is convex per neuron and the stopping rule does not look at the start); the first step is several times shorter.
    track = getattr(sess, '_cd_track', None)
    if (extrapolate and CD_EXTRAPOLATE and track is not None and track['key'] == hess_key and track['x'].shape == x_in.shape
            and np.array_equal(track['x'], x_in) and float(np.max(np.abs(track['step']))) < 0.1):
        ahead = track['step']
        prev = track.get('prev_step')
        if CD_EXTRAPOLATE >= 2 and prev is not None and float(np.max(np.abs(ahead - prev))) < 0.5 * float(np.max(np.abs(ahead))):
            ahead = 2.0 * ahead - prev          # the displacement itself changes smoothly: continue its trend
        theta = theta + ahead.reshape(D, q)
    state = {'n_full': 0, 'n_chord': 0, 'hess_at': None}

    def evaluate(point, want_full):
        """-> cost, delta, dec, rho (staleness of the Hessians the step was built with; 0 = fresh)"""
        have = getattr(sess, '_cd_hess_resident', False)
        beside = getattr(sess, '_beside_cd_pass', None)       # (updateParams: a timescale round rides beside every (C,d) pass)
        if beside is not None:
            beside.kick()
        try:
            if want_full or not have:
                out = ctx.mstep_cd_newton_pass(point.reshape(-1), prior_center, inv_s2)
                sess._cd_hess_resident = True
                sess._cd_hess_key = hess_key
                state['n_full'] += 1
                state['hess_at'] = point.copy()
                return out + (0.0,)
            out = ctx.mstep_cd_chord_pass(point.reshape(-1), prior_center, inv_s2)
        finally:
            if beside is not None:
                beside.reap()
        state['n_chord'] += 1
        rho = RHO_OLD if state['hess_at'] is None else float(np.max(np.abs(point - state['hess_at'])))
        return out + (rho,)

    # the start point is last M-step's optimum: the Hessians left there are a good chord for the first step
    cost, delta, dec, rho = evaluate(theta, want_full=False)
    prev_step = np.inf
    for it in range(max_iter):
        step = delta.reshape(D, q)
        smax = float(np.max(np.abs(step)))
        if max(rho, smax) * smax < xtol:
            # predicted error after this step is below xtol: take it, the cost follows from the quadratic model
            theta = theta + step
            cost = cost - 0.5 * dec
            break
        if rho > 0.0 and smax > 0.2 * prev_step:
            # the chord iteration is not contracting: rebuild the step with fresh Hessians at the same point
            cost, delta, dec, rho = evaluate(theta, want_full=True)
            prev_step = np.inf
            continue
        alpha = np.ones(q)
        trial = theta + step
        # next pass: fresh Hessians unless the ones resident were built at most 1e-2 away (a chord step then gains two
        # digits for 0.3x the cost of a full pass, which is the better rate)
        fresh_near = state['hess_at'] is not None and float(np.max(np.abs(trial - state['hess_at']))) < 1e-2
        c_try, d_try, dec_try, rho_try = evaluate(trial, want_full=not fresh_near)
        slack = 1e-13 * (1.0 + np.abs(cost))
        ok = np.isfinite(c_try) & (c_try <= cost - 1e-4 * alpha * dec + slack)
        ls = 0
        while not np.all(ok) and ls < 40:
            alpha = np.where(ok, alpha, 0.5 * alpha)
            trial = theta + alpha[None, :] * step
            c_bt = ctx.mstep_cd_cost_per_neuron(trial.reshape(-1), prior_center, inv_s2)
            ok_new = np.isfinite(c_bt) & (c_bt <= cost - 1e-4 * alpha * dec + slack)
            c_try = np.where(ok, c_try, c_bt)
            newly = ok_new & ~ok
            ok = ok | ok_new
            ls += 1
            if np.any(newly) and np.all(ok):
                c_try, d_try, dec_try, rho_try = evaluate(trial, want_full=True)          # step data at the accepted point
        if not np.all(ok):                      # neurons whose search failed keep their current value
            alpha = np.where(ok, alpha, 0.0)
            trial = theta + alpha[None, :] * step
            c_try, d_try, dec_try, rho_try = evaluate(trial, want_full=True)
        moved = float(np.max(np.abs(alpha[None, :] * step)))
        theta, cost, delta, dec, rho = trial, c_try, d_try, dec_try, rho_try
        prev_step = moved
        if verbose:
            print('  newton (C,d) it %d: cost %.10g, max step %.3e' % (it + 1, cost.sum(), moved))
    sess._cd_passes = (state['n_full'], state['n_chord'])
    if extrapolate:
        chained = track is not None and track['key'] == hess_key and track['x'].shape == x_in.shape and np.array_equal(track['x'], x_in)
        sess._cd_track = {'key': hess_key, 'x': theta.reshape(-1).copy(), 'step': theta.reshape(-1) - x_in,
                          'prev_step': track['step'] if chained else None}
    return theta.reshape(-1), float(np.sum(cost)), state['n_full'] + state['n_chord']


def learnLTparams(oldParams, infRes, experiment, CdOptimMethod, CdMaxIter=None, verbose=False):
    """reference learning.py:93-141: same scipy call (method, options) on device-evaluated cost/grad.
    CdOptimMethod='newton' (an addition) runs the device Newton solver instead: the exact minimiser of the
    same cost, ~6 passes over the data instead of TNC's ~130 evaluations."""
    ydim, xdim = np.shape(oldParams['C'])
    sess = _resident_session(infRes, experiment, xdim)
    if CdOptimMethod == 'newton':
        x, fun, _ = _newton_cd(sess, util.CdtoVecCd(oldParams['C'], oldParams['d']), verbose=verbose, hess_key=_trial_key(infRes), extrapolate=True)
        newC, newd = util.vecCdtoCd(x, xdim, ydim)
        return newC, newd, fun
    cache = _CostGradCache(lambda v: sess.ctx.mstep_cd_costgrad(v))
    xinit = util.CdtoVecCd(oldParams['C'], oldParams['d'])
    resCd = op.minimize(fun=cache.fun, x0=xinit, jac=cache.jac, method=CdOptimMethod,
                        options={'disp': verbose, 'maxiter': CdMaxIter})
    if verbose:
        print('Cd optimization successful.' if resCd.success else 'Cd optimization unsuccessful.')
    newC, newd = util.vecCdtoCd(resCd.x, xdim, ydim)
    return newC, newd, resCd.fun


def MStepObservationCostWithPrior(vecCd, oldParams, xdim, ydim, experiment, infRes, invPriorCov):
    """reference learning.py:445-486: the (C,d) cost minus 0.5 (v - v_old)^T invPriorCov (v - v_old), invPriorCov a full
    (negative definite) matrix; the data term runs on the device, the quadratic form on the host."""
    sess = _resident_session(infRes, experiment, xdim)
    dv = np.asarray(vecCd, dtype=np.float64) - util.CdtoVecCd(oldParams['C'], oldParams['d'])
    return sess.ctx.mstep_cd_costgrad(vecCd)[0] - 0.5 * dv @ np.asarray(invPriorCov) @ dv


def MStepObservationCostWithPrior_grad(vecCd, oldParams, xdim, ydim, experiment, infRes, invPriorCov):
    """reference learning.py:488-534"""
    sess = _resident_session(infRes, experiment, xdim)
    dv = np.asarray(vecCd, dtype=np.float64) - util.CdtoVecCd(oldParams['C'], oldParams['d'])
    return sess.ctx.mstep_cd_costgrad(vecCd)[1] - np.asarray(invPriorCov) @ dv


def _learn_cd_full_prior(oldParams, infRes, experiment, CdOptimMethod, prevInvPriorCov, hessTol, verbose):
    """covOpts='useHessian' (reference learning.py:546-549, 607-630): the new prior precision is minus the finite-difference
    Jacobian of the regularised gradient at the old parameters (4 q(p+1) device cost/gradient passes), then the same scipy
    call as the other variants with that full matrix."""
    ydim, xdim = np.shape(oldParams['C'])
    sess = _resident_session(infRes, experiment, xdim)
    old = util.CdtoVecCd(oldParams['C'], oldParams['d'])
    prev = np.asarray(prevInvPriorCov, dtype=np.float64)

    def grad_prev(v):
        return sess.ctx.mstep_cd_costgrad(v)[1] - prev @ (v - old)
    invPriorCov = -util.approx_jacobian(old, grad_prev, hessTol)

    def evaluate(v):
        cost, grad = sess.ctx.mstep_cd_costgrad(v)
        dv = v - old
        return cost - 0.5 * dv @ invPriorCov @ dv, grad - invPriorCov @ dv
    cache = _CostGradCache(evaluate)
    kw = dict(fun=cache.fun, x0=old, jac=cache.jac, method=CdOptimMethod, options={'disp': verbose, 'gtol': 1e-10})
    if CdOptimMethod == 'L-BFGS-B':
        kw['bounds'] = [(None, None)] * (xdim * ydim + ydim)
    resCd = op.minimize(**kw)
    newC, newd = util.vecCdtoCd(resCd.x, xdim, ydim)
    return newC, newd, resCd.fun, invPriorCov


def learnLTparamsWithPrior(oldParams, infRes, experiment, CdOptimMethod, regularizer_stepsize_Cd, prevInvPriorCov,
                           covOpts='useDiag', updateCdJointly=True, hessTol=1e-5, verbose=False):
    """reference learning.py:536-676 with joint (C,d) update: 'useDiag' (the engine default, prior on the device) and
    'useHessian' (finite-difference prior precision, full matrix on the host)."""
    if not updateCdJointly:
        raise NotImplementedError('updateCdJointly=False raises in the reference itself (learning.py:393) and is not built')
    if covOpts == 'useHessian':
        if CdOptimMethod == 'newton':
            raise ValueError("CdOptimMethod='newton' needs the diagonal prior of covOpts='useDiag'")
        return _learn_cd_full_prior(oldParams, infRes, experiment, CdOptimMethod, prevInvPriorCov, hessTol, verbose)
    if covOpts != 'useDiag':
        raise ValueError("covOpts must be 'useDiag' or 'useHessian'")
    ydim, xdim = np.shape(oldParams['C'])
    sess = _resident_session(infRes, experiment, xdim)
    old = util.CdtoVecCd(oldParams['C'], oldParams['d'])
    inv_s2 = 1.0 / regularizer_stepsize_Cd ** 2
    invPriorCov = -np.diag(np.ones(xdim * ydim + ydim)) / (regularizer_stepsize_Cd ** 2)       # learning.py:580-581
    if CdOptimMethod == 'newton':
        x, fun, _ = _newton_cd(sess, old, prior_center=old, inv_s2=inv_s2, verbose=verbose, hess_key=_trial_key(infRes))
        newC, newd = util.vecCdtoCd(x, xdim, ydim)
        return newC, newd, fun, invPriorCov
    cache = _CostGradCache(lambda v: sess.ctx.mstep_cd_costgrad(v, old, inv_s2))
    kw = dict(fun=cache.fun, x0=old, jac=cache.jac, method=CdOptimMethod, options={'disp': verbose, 'gtol': 1e-10})
    if CdOptimMethod == 'L-BFGS-B':
        kw['bounds'] = [(None, None)] * (xdim * ydim + ydim)
    resCd = op.minimize(**kw)
    if verbose:
        print('Cd optimization successful.' if resCd.success else 'Cd optimization unsuccessful.')
    newC, newd = util.vecCdtoCd(resCd.x, xdim, ydim)
    return newC, newd, resCd.fun, invPriorCov


# ------------------------------------------------------------------------------------------------
# GP timescales
# ------------------------------------------------------------------------------------------------
class DevicePrecomp(list):
    """makePrecomp result (reference learning.py:145-173) kept on the device; entry k is a small dict
    with the reference's keys, 'PautoSum' fetched lazily."""

    def __init__(self, sess, T):
        super().__init__()
        self.session = sess
        self.numTrials = sess.ctx.mstep_precomp()
        self.stamp = sess.post_stamp
        self._P = None
        for k in range(sess.p):
            self.append({'T': T, 'numTrials': self.numTrials, 'latent': k, 'precomp': self})

    def pautosum(self):
        if self._P is None:
            self._P = self.session.ctx.pautosum()
        return self._P


def makePrecomp(infRes, experiment=None, xdim=None):
    """reference learning.py:145-173 (PautoSum_k = sum_r Sigma_r^kk + m_rk m_rk^T), on device."""
    if isinstance(infRes, DeviceInfRes):
        sess = infRes.session
        if infRes.stamp != sess.post_stamp:
            raise _hip.HipBackendError('this infRes belongs to a superseded E-step')
    else:
        if experiment is None:
            raise ValueError('makePrecomp needs the experiment for a foreign infRes')
        sess = _resident_session(infRes, experiment, xdim if xdim is not None else np.shape(infRes['post_mean'][0])[0])
    return DevicePrecomp(sess, sess.T)


def _tau_eval(precomp_k, p):
    pc = precomp_k['precomp']
    return pc.session.ctx.mstep_tau_costgrad(precomp_k['latent'], float(np.asarray(p).reshape(-1)[0]))


def MStepGPtimescaleCost(p, precomp, epsNoise):
    """reference learning.py:175-214"""
    return _tau_eval(precomp, p)[0]


def MStepGPtimescaleCost_grad(p, precomp, epsNoise):
    """reference learning.py:216-255"""
    return np.array([_tau_eval(precomp, p)[1]])


TAU_SOLVER = 'lockstep'      # 'lockstep' (all latents per device evaluation) or 'scipy' (the reference's BFGS calls)


def _lockstep_minimize(evaluate, p0, gtol=1e-8, xtol=1e-10, max_iter=60, curv0=None):
    """Minimise xdim independent smooth 1-D costs f_k(p_k) together: every iteration is ONE batched device
    evaluation of all (f_k, f_k').  Safeguarded secant iteration on the gradient with a bracketing fallback;
    stops per latent on |f'| <= gtol (scipy BFGS's criterion, learning.py:283-288) or |step| <= xtol.
    curv0: optional per-latent curvature guesses (e.g. from the previous EM iteration) for the first step.
    Returns (p, f, g, nfev, done, curv) with curv the last positive secant curvature seen per latent."""
    p = np.array(p0, dtype=np.float64)
    k = p.size
    f, g = evaluate(p)
    nfev = 1
    done = np.abs(g) <= gtol
    lo = np.where(g < 0, p, -np.inf)          # g < 0: minimum lies to the right
    hi = np.where(g > 0, p, np.inf)
    p_prev, g_prev = p.copy(), g.copy()
    step = -np.sign(g) * 0.25
    curv_seen = np.full(k, np.nan)
    if curv0 is not None:
        c0 = np.asarray(curv0, dtype=np.float64)
        good = np.isfinite(c0) & (c0 > 0)
        with np.errstate(divide='ignore', invalid='ignore'):
            step = np.where(good, np.clip(-g / c0, -1.0, 1.0), step)
        curv_seen = np.where(good, c0, curv_seen)
    for _ in range(max_iter):
        if np.all(done):
            break
        p_new = np.where(done, p, p + step)
        have_lo, have_hi = np.isfinite(lo), np.isfinite(hi)
        both = have_lo & have_hi
        outside = both & ((p_new <= lo) | (p_new >= hi))
        p_new = np.where(outside & ~done, 0.5 * (lo + hi), p_new)
        f_new, g_new = evaluate(p_new)
        nfev += 1
        lo = np.where((g_new < 0) & ~done, np.maximum(lo, p_new), lo)
        hi = np.where((g_new > 0) & ~done, np.minimum(hi, p_new), hi)
        dp = p_new - p
        with np.errstate(divide='ignore', invalid='ignore'):
            curv = (g_new - g) / dp
        sec = np.where((curv > 0) & np.isfinite(curv), -g_new / curv, -np.sign(g_new) * np.minimum(2.0 * np.abs(dp), 1.0))
        curv_seen = np.where((curv > 0) & np.isfinite(curv) & ~done, curv, curv_seen)
        sec = np.clip(sec, -1.0, 1.0)
        newly = (~done) & ((np.abs(g_new) <= gtol) | (np.abs(dp) <= xtol))
        # secant step already below xtol with a trusted (positive) curvature: take it without paying an evaluation to
        # confirm it (the error after it is second order in the step before)
        tiny = (~done) & (~newly) & (curv > 0) & np.isfinite(curv) & (np.abs(sec) <= 10.0 * xtol)
        p_prev, g_prev = np.where(done, p_prev, p), np.where(done, g_prev, g)
        p, f, g = np.where(done, p, p_new), np.where(done, f, f_new), np.where(done, g, g_new)
        p = np.where(tiny, p + sec, p)
        f = np.where(tiny, f - 0.5 * g * g / np.where(tiny, curv, 1.0), f)
        g = np.where(tiny, 0.0, g)
        newly = newly | tiny
        step = np.where(done | newly, 0.0, sec)
        done = done | newly
    return p, f, g, nfev, done, curv_seen


def _newton_poly_root(X, Y, lo, hi, active=None):
    """Roots in (lo, hi) of the polynomials interpolating (X[:, j], Y[:, j]) (columns = independent problems, Y
    changes sign between lo and hi): Newton divided differences, then Newton's iteration on the interpolant kept inside
    the shrinking bracket (a bisection step wherever Newton leaves it) - a handful of vector operations per solve;
    this runs on the host between two device rounds of the timescale M-step."""
    n = X.shape[0]
    coef = Y.copy()
    for lvl in range(1, n):
        coef[lvl:] = (coef[lvl:] - coef[lvl - 1:-1]) / (X[lvl:] - X[:n - lvl])

    def poly(z):
        """value and derivative (Horner on the Newton form)"""
        v = coef[n - 1].copy()
        dv = np.zeros_like(v)
        for i in range(n - 2, -1, -1):
            dv = dv * (z - X[i]) + v
            v = v * (z - X[i]) + coef[i]
        return v, dv
    a, b = lo.copy(), hi.copy()
    z = 0.5 * (a + b)
    idle = np.zeros(z.shape, dtype=bool) if active is None else ~np.asarray(active, dtype=bool)   # columns nobody reads
    for _ in range(80):
        v, dv = poly(z)
        up = v > 0
        b = np.where(up, z, b)
        a = np.where(up, a, z)
        with np.errstate(all='ignore'):
            zn = z - v / dv
        mid = 0.5 * (a + b)
        zn = np.where(np.isfinite(zn) & (zn >= a) & (zn <= b), zn, mid)
        if np.all(idle | (np.abs(zn - z) <= 1e-14 * (1.0 + np.abs(z))) | (b - a <= 1e-14 * (1.0 + np.abs(z)))):
            z = zn
            break
        z = zn
    return z


def _lockstep_multi_np(evaluate_multi, p0, d_hint=None, gtol=1e-8, xtol=1e-10, max_rounds=30, m=4):
    """(The array form of _lockstep_multi below: the statement of the algorithm, and what tests/test_cpu_host.py holds the scalar form to,
    bit for bit.)  Roots of xdim independent increasing functions g_k = f_k' (f_k smooth, one minimum) found together with m
    candidate points per latent per round; a round is ONE batched device pass (its cost is launch latency, nearly
    independent of m).  Round 1 spreads its points along the displacement predicted from the previous EM
    iteration (d_hint), later rounds place them in a shrinking cluster around the root of the cubic through the
    samples next to the sign change; stops per latent on |g| <= gtol or when two successive root predictions
    agree to xtol.  Returns (p, f, g, rounds, done)."""
    p0 = np.array(p0, dtype=np.float64)
    k = p0.size
    if d_hint is not None:
        d = np.asarray(d_hint, dtype=np.float64)
        d = np.where(np.isfinite(d) & (np.abs(d) > 1e-6), np.clip(d, -1.0, 1.0), np.nan)
    else:
        d = np.full(k, np.nan)
    hinted = np.isfinite(d)
    Q = np.where(hinted[None, :], p0[None, :] + np.array([0.0, 0.6, 1.0, 1.5])[:, None] * np.where(hinted, d, 0.0)[None, :],
                 p0[None, :] + 0.25 * np.array([-1.0, -1.0 / 3, 1.0 / 3, 1.0])[:, None])
    Ps, Fs, Gs = np.empty((0, k)), np.empty((0, k)), np.empty((0, k))
    done = np.zeros(k, dtype=bool)
    root, pred_prev = p0.copy(), np.full(k, np.nan)
    rounds = 0
    offs = np.array([-1.5, -0.5, 0.5, 1.5])
    for rounds in range(1, max_rounds + 1):
        F, G = evaluate_multi(Q)
        Ps, Fs, Gs = np.vstack([Ps, Q]), np.vstack([Fs, F]), np.vstack([Gs, G])
        ib = np.argmin(np.abs(Gs), axis=0)
        cols = np.arange(k)
        best_p, best_g = Ps[ib, cols], Gs[ib, cols]
        hit = (~done) & (np.abs(best_g) <= gtol)
        root = np.where(hit, best_p, root)
        done = done | hit
        lo = np.max(np.where(Gs < 0, Ps, -np.inf), axis=0)
        hi = np.min(np.where(Gs > 0, Ps, np.inf), axis=0)
        brack = np.isfinite(lo) & np.isfinite(hi) & (hi > lo)
        Qn = np.tile(root[None, :], (m, 1)) + 1e-7 * offs[:, None]        # finished latents: harmless filler
        work = brack & ~done
        if np.any(work):
            w = np.where(work, hi - lo, 1.0)
            mid = np.where(work, 0.5 * (lo + hi), 0.0)
            order = np.argsort(np.abs(Ps - mid[None, :]), axis=0, kind='stable')[:4]
            X, Y = np.take_along_axis(Ps, order, axis=0), np.take_along_axis(Gs, order, axis=0)
            srt = np.argsort(X, axis=0)
            X, Y = np.take_along_axis(X, srt, axis=0), np.take_along_axis(Y, srt, axis=0)
            distinct = np.all(np.diff(X, axis=0) > 0, axis=0)
            with np.errstate(all='ignore'):
                r_poly = _newton_poly_root(X, Y, np.where(work, lo, 0.0), np.where(work, hi, 1.0), work)
            glo = np.max(np.where((Gs < 0) & (Ps == lo[None, :]), Gs, -np.inf), axis=0)
            ghi = np.min(np.where((Gs > 0) & (Ps == hi[None, :]), Gs, np.inf), axis=0)
            with np.errstate(all='ignore'):
                r_sec = lo - glo * (hi - lo) / (ghi - glo)
            r = np.where(distinct & np.isfinite(r_poly), r_poly, r_sec)
            r = np.where(np.isfinite(r) & (r > lo) & (r < hi), r, mid)
            err = np.abs(r - pred_prev)
            agree = work & np.isfinite(err) & (err <= xtol)
            root = np.where(work, r, root)
            done = done | agree | (work & (w <= 4.0 * xtol))
            delta = np.where(np.isfinite(err), np.clip(2.0 * err, 4.0 * xtol, 0.02 * w), 0.02 * w)
            cluster = r[None, :] + delta[None, :] * offs[:, None]
            Qn = np.where((work & ~done)[None, :], cluster, Qn)
            pred_prev = np.where(work, r, pred_prev)
        loose = (~brack) & (~done)
        if np.any(loose):
            # no sign change yet: step out geometrically beyond the outermost sample on the downhill side
            sgn = -np.sign(np.where(loose, best_g, 1.0))
            far = np.where(sgn > 0, np.max(Ps, axis=0), np.min(Ps, axis=0))
            span = np.maximum(np.max(Ps, axis=0) - np.min(Ps, axis=0), 0.1)
            out = far[None, :] + (sgn * span)[None, :] * np.array([0.5, 1.0, 2.0, 4.0])[:, None]
            Qn = np.where(loose[None, :], out, Qn)
        if np.all(done):
            break
        Q = Qn
    ib = np.argmin(np.abs(Ps - root[None, :]), axis=0)
    cols = np.arange(k)
    return root, Fs[ib, cols], Gs[ib, cols], rounds, done


class _ConcurrentScalarProblems:
    """k independent scipy optimisations of scalar problems, one Python thread each, whose cost/gradient requests are
    gathered into ONE batched device evaluation per round (the evaluation is a latency-bound chain of small launches:
    k problems cost what one does).  Every optimiser sees exactly the call sequence it would see alone."""

    def __init__(self, x_init, evaluate_batch):
        import threading
        self._threading = threading
        self.evaluate_batch = evaluate_batch              # x[k] -> (f[k], g[k])
        self.x = np.array(x_init, dtype=np.float64)
        self.cond = threading.Condition()
        self.pending, self.results = {}, {}
        self.active = set(range(self.x.size))
        self.generation = 0
        self.error = None
        self.rounds = 0

    def _flush(self):                                     # lock held by the caller
        for i, v in self.pending.items():
            self.x[i] = v
        try:
            f, g = self.evaluate_batch(self.x.copy())
            self.results = {i: (float(f[i]), float(g[i])) for i in self.pending}
        except Exception as exc:                          # hand the failure to every waiting optimiser
            self.error = exc
            self.results = {}
        self.pending = {}
        self.rounds += 1
        self.generation += 1
        self.cond.notify_all()

    def request(self, idx, value):
        with self.cond:
            self.pending[idx] = float(value)
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
        """solve_one(idx, evaluate) -> result, with evaluate(v) -> (cost, grad) of problem idx at scalar v."""
        out = [None] * self.x.size

        def worker(idx):
            try:
                out[idx] = solve_one(idx, lambda v: self.request(idx, v))
            except Exception as exc:
                out[idx] = exc
            finally:
                with self.cond:
                    self.active.discard(idx)
                    if self.pending and len(self.pending) == len(self.active):
                        self._flush()
        import sys
        threads = [self._threading.Thread(target=worker, args=(i,)) for i in range(self.x.size)]
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


def _argmin_first(vals):
    """np.argmin of a list of floats: the first NaN if there is one, else the first smallest"""
    best, bi = None, 0
    for i, v in enumerate(vals):
        if v != v:
            return i
        if best is None or v < best:
            best, bi = v, i
    return bi


def _poly_roots_scalar(cols):
    """_newton_poly_root for a list of (X, Y, lo, hi) with plain floats: the same operations in the same order (all problems iterate until all
    have converged, as the array form does), so the same bits - without forty array operations on ten numbers each."""
    st = []
    for X, Y, lo, hi in cols:
        n = len(X)
        coef = list(Y)
        for lvl in range(1, n):
            coef = coef[:lvl] + [(coef[i] - coef[i - 1]) / (X[i] - X[i - lvl]) for i in range(lvl, n)]
        st.append([X, coef, lo, hi, 0.5 * (lo + hi)])
    for _ in range(80):
        conv = True
        for q in st:
            X, coef, a, b, z = q
            n = len(X)
            v, dv = coef[n - 1], 0.0
            for i in range(n - 2, -1, -1):
                dz = z - X[i]
                dv = dv * dz + v
                v = v * dz + coef[i]
            if v > 0:
                b = z
            else:
                a = z
            mid = 0.5 * (a + b)
            if dv != 0.0:
                zn = z - v / dv
                if not (math.isfinite(zn) and a <= zn <= b):
                    zn = mid
            else:
                zn = mid
            tol = 1e-14 * (1.0 + abs(z))
            if not (abs(zn - z) <= tol or b - a <= tol):
                conv = False
            q[2], q[3], q[4] = a, b, zn
        if conv:
            break
    return [q[4] for q in st]


# (round 6) A root is also accepted when the interpolation-error term of the cubic it comes from - the next divided difference, taken with the nearest
# sample outside the four, times prod (r - x_i), over the cubic's slope - is below xtol: the round that would only CONFIRM it (two predictions
# agreeing to xtol) is not run.  At config 3 the finder's rounds are prediction errors of 5e-4, 5e-11 and 0: the third round was that confirmation.
TAU_EARLY_ACCEPT = os.environ.get('PGPFA_TAU_EARLY_ACCEPT', '1') != '0'


def _interp_root_error(X, Y, x4, y4, r):
    """Estimated distance of the cubic's root r (cubic through (X, Y), four distinct points) from the root of the sampled function: the
    Newton-form error term with the fifth sample (x4, y4) over the cubic's derivative at r.  inf when it cannot be formed."""
    xs, c = list(X) + [x4], list(Y) + [y4]
    if any(abs(xs[4] - xs[i]) == 0.0 for i in range(4)):
        return float('inf')
    for lvl in range(1, 5):
        c = c[:lvl] + [(c[i] - c[i - 1]) / (xs[i] - xs[i - lvl]) for i in range(lvl, 5)]
    d0, d1, d2 = r - xs[0], r - xs[1], r - xs[2]
    slope = c[1] + c[2] * (d0 + d1) + c[3] * (d0 * d1 + d0 * d2 + d1 * d2)
    if slope == 0.0 or not math.isfinite(slope):
        return float('inf')
    est = abs(c[4] * d0 * d1 * d2 * (r - xs[3]) / slope)
    return est if math.isfinite(est) else float('inf')


def _lockstep_multi(evaluate_multi, p0, d_hint=None, gtol=1e-8, xtol=1e-10, max_rounds=30, m=4):
    """_lockstep_multi_np with the bookkeeping between two device rounds in plain floats, latent by latent: ten problems of four to twelve
    samples are a few hundred scalar operations, where the array form spends 1.3 ms per M-step in the overhead of some 450 array calls
    (a tenth of the M-step at config 3).  Same samples, same decisions, same results bit for bit (tests/test_cpu_host.py)."""
    if m != 4:
        return _lockstep_multi_np(evaluate_multi, p0, d_hint, gtol, xtol, max_rounds, m)
    rounds = _lockstep_multi_rounds(p0, d_hint, gtol, xtol, max_rounds)
    Q = next(rounds)
    while True:
        try:
            Q = rounds.send(evaluate_multi(Q))
        except StopIteration as stop:
            return stop.value


def _lockstep_multi_rounds(p0, d_hint=None, gtol=1e-8, xtol=1e-10, max_rounds=30):
    """The finder of _lockstep_multi as a generator: yields the 4 x k sample points of a round, is sent (F, G) of that round, returns
    (root, f, g, rounds, done).  The caller decides WHEN a round is evaluated - updateParams starts it on the device's side stream and runs
    the (C,d) passes meanwhile (round 6)."""
    p0 = [float(v) for v in np.asarray(p0, dtype=np.float64).reshape(-1)]
    k = len(p0)
    inf = float('inf')
    if d_hint is not None:
        dh = [float(v) for v in np.asarray(d_hint, dtype=np.float64).reshape(-1)]
        dh = [min(max(v, -1.0), 1.0) if (math.isfinite(v) and abs(v) > 1e-6) else None for v in dh]
    else:
        dh = [None] * k
    first_h, first_u = (0.0, 0.6, 1.0, 1.5), (-1.0, -1.0 / 3, 1.0 / 3, 1.0)
    offs = (-1.5, -0.5, 0.5, 1.5)
    Q = [[(p0[j] + first_h[i] * dh[j]) if dh[j] is not None else (p0[j] + 0.25 * first_u[i]) for j in range(k)] for i in range(4)]
    P = [[] for _ in range(k)]
    Fv = [[] for _ in range(k)]
    Gv = [[] for _ in range(k)]
    done = [False] * k
    root = list(p0)
    pred_prev = [None] * k
    rounds = 0
    for rounds in range(1, max_rounds + 1):
        F, G = yield np.array(Q, dtype=np.float64)
        F, G = np.asarray(F, dtype=np.float64).tolist(), np.asarray(G, dtype=np.float64).tolist()
        work, info = [], {}
        Qn = [[0.0] * k for _ in range(4)]
        for j in range(k):
            Pj, Gj = P[j], Gv[j]
            for i in range(4):
                Pj.append(Q[i][j]); Fv[j].append(F[i][j]); Gj.append(G[i][j])
            ns = len(Pj)
            ib = _argmin_first([abs(g) for g in Gj])
            best_p, best_g = Pj[ib], Gj[ib]
            if not done[j] and abs(best_g) <= gtol:
                root[j] = best_p
                done[j] = True
            lo, hi = -inf, inf
            for i in range(ns):
                if Gj[i] < 0:
                    if Pj[i] > lo:
                        lo = Pj[i]
                elif Gj[i] > 0:
                    if Pj[i] < hi:
                        hi = Pj[i]
            brack = lo > -inf and hi < inf and hi > lo
            for i in range(4):
                Qn[i][j] = root[j] + 1e-7 * offs[i]
            if brack and not done[j]:
                mid = 0.5 * (lo + hi)
                order = sorted(range(ns), key=lambda i: abs(Pj[i] - mid))[:4]
                order.sort(key=lambda i: Pj[i])
                X, Y = [Pj[i] for i in order], [Gj[i] for i in order]
                distinct = all(X[i + 1] - X[i] > 0 for i in range(len(X) - 1))
                glo = max(Gj[i] for i in range(ns) if Gj[i] < 0 and Pj[i] == lo)
                ghi = min(Gj[i] for i in range(ns) if Gj[i] > 0 and Pj[i] == hi)
                info[j] = (lo, hi, mid, glo, ghi, distinct)
                if distinct:
                    work.append((j, X, Y))
            elif not done[j]:
                # no sign change yet: step out geometrically beyond the outermost sample on the downhill side
                sgn = -1.0 if best_g > 0 else (1.0 if best_g < 0 else 0.0)
                pmax, pmin = max(Pj), min(Pj)
                far = pmax if sgn > 0 else pmin
                span = max(pmax - pmin, 0.1)
                for i, f in enumerate((0.5, 1.0, 2.0, 4.0)):
                    Qn[i][j] = far + (sgn * span) * f
        rp = _poly_roots_scalar([(X, Y, info[j][0], info[j][1]) for j, X, Y in work]) if work else []
        r_poly = {j: rp[i] for i, (j, _, _) in enumerate(work)}
        pts = {j: (X, Y) for j, X, Y in work}
        for j, (lo, hi, mid, glo, ghi, distinct) in info.items():
            r = r_poly.get(j)
            from_cubic = r is not None and math.isfinite(r) and lo < r < hi
            if r is None or not math.isfinite(r):
                r = lo - glo * (hi - lo) / (ghi - glo)
            if not (math.isfinite(r) and lo < r < hi):
                r = mid
            w = hi - lo
            err = None if pred_prev[j] is None else abs(r - pred_prev[j])
            agree = err is not None and math.isfinite(err) and err <= xtol
            if TAU_EARLY_ACCEPT and not agree and from_cubic and len(P[j]) >= 5 and len(pts[j][0]) == 4:
                X4, Y4 = pts[j]
                rest = [i for i in range(len(P[j])) if P[j][i] not in X4]
                if rest:
                    i5 = min(rest, key=lambda i: abs(P[j][i] - r))
                    agree = _interp_root_error(X4, Y4, P[j][i5], Gv[j][i5], r) <= xtol
            root[j] = r
            done[j] = agree or (w <= 4.0 * xtol)
            delta = min(max(2.0 * err, 4.0 * xtol), 0.02 * w) if (err is not None and math.isfinite(err)) else 0.02 * w
            if not done[j]:
                for i in range(4):
                    Qn[i][j] = r + delta * offs[i]
            pred_prev[j] = r
        if all(done):
            break
        Q = Qn
    fo, go = [], []
    for j in range(k):
        ib = _argmin_first([abs(v - root[j]) for v in P[j]])
        fo.append(Fv[j][ib]); go.append(Gv[j][ib])
    return np.array(root), np.array(fo), np.array(go), rounds, np.array(done, dtype=bool)


def learnGPparams(oldParams, infRes, experiment):
    """reference learning.py:257-293: minimise each latent's timescale cost from p0 = log(1/tau_bins^2) to
    |grad| <= 1e-8.  The xdim problems are independent and one-dimensional, so by default they are solved in
    lockstep (one batched Gram/Cholesky/trace evaluation of all latents per iteration); TAU_SOLVER='scipy'
    runs the reference's per-latent scipy BFGS calls on the same device callbacks instead."""
    xdim = np.shape(oldParams['C'])[1]
    sess = _resident_session(infRes, experiment, xdim)
    binSize = experiment.binSize
    oldTau = np.asarray(oldParams['tau'], dtype=np.float64) * 1000 / binSize
    if getattr(sess, '_tau_beside', None) is None:           # (updateParams has built PautoSum already when its rounds ride beside the (C,d) passes)
        DevicePrecomp(sess, sess.T)
    initp = np.log(1 / oldTau ** 2)
    details = [[]] * xdim
    if TAU_SOLVER in ('lockstep', 'secant'):
        if TAU_SOLVER == 'lockstep':
            # 4 candidate points per latent per batched pass; the displacement of the previous EM iteration's M-step
            # predicts where this one's optimum lies
            riding = getattr(sess, '_tau_beside', None)
            if riding is not None and riding.key == tuple(initp.tolist()):
                # (updateParams started these rounds beside the (C,d) passes: same generator, same rounds, whatever is left runs here)
                pv, fv, gv, nfev, ok = riding.finish()
            else:
                pv, fv, gv, nfev, ok = _lockstep_multi(sess.ctx.mstep_tau_costgrad_multi, initp, d_hint=getattr(sess, '_tau_step', None))
            sess._tau_step = pv - initp
        else:
            # one point per latent per pass: safeguarded secant, seeded with the curvature seen in the previous M-step
            pv, fv, gv, nfev, ok, curv = _lockstep_minimize(sess.ctx.mstep_tau_costgrad_batch, initp,
                                                            curv0=getattr(sess, '_tau_curv', None))
            sess._tau_curv = curv
        for xd in range(xdim):
            details[xd] = op.OptimizeResult(x=np.array([pv[xd]]), fun=fv[xd], jac=np.array([gv[xd]]), nfev=nfev,
                                            success=bool(ok[xd]), message='lockstep %s' % ('multi-point' if TAU_SOLVER == 'lockstep' else 'secant'))
        tempTau = (1 / np.exp(pv)) ** 0.5
        return tempTau * binSize / 1000, details
    # TAU_SOLVER == 'scipy': the reference's per-latent BFGS calls, run concurrently so that their evaluations batch
    def solve_one(xd, evaluate):
        cache = _CostGradCache(lambda v: evaluate(float(np.asarray(v).reshape(-1)[0])))
        return op.minimize(fun=cache.fun, x0=initp[xd], jac=lambda v, c=cache: np.array([c.jac(v)]),
                           options={'disp': False, 'gtol': 1e-8})
    details = _ConcurrentScalarProblems(initp, sess.ctx.mstep_tau_costgrad_batch).run(solve_one)
    tempTau = np.array([(1 / np.exp(res.x[0])) ** 0.5 for res in details])
    return tempTau * binSize / 1000, details


def learnGPparamsWithPrior(oldParams, infRes, experiment, tauOptimMethod, regularizer_stepsize_tau):
    """reference learning.py:771-830.  The regulariser 0.5*(tau-tau_old)^2/s^2 (tau in seconds) is added to
    the device cost; its gradient is added WITHOUT the chain-rule factor, exactly as the reference does
    (learning.py:733-734,769), because the optimiser trajectory - and so the result - depends on it."""
    xdim = np.shape(oldParams['C'])[1]
    sess = _resident_session(infRes, experiment, xdim)
    binSize = experiment.binSize
    tau_old = np.asarray(oldParams['tau'], dtype=np.float64)
    oldTau = tau_old * 1000 / binSize
    DevicePrecomp(sess, sess.T)
    s = regularizer_stepsize_tau
    initp = np.log(1 / oldTau ** 2)

    if tauOptimMethod == 'lockstep':
        # (round 6, opt-in; the engine's default stays the reference's 'TNC')  The reference hands scipy a cost and a gradient that do not
        # belong together (learning.py:733-734, 769: the regulariser's derivative enters without the chain-rule factor) and takes whatever
        # point the optimiser stops at.  TNC, following that gradient, stops at ITS zero - dcost/dp + (tau(p) - tau_old) / s^2 = 0, a
        # one-dimensional root per latent (where the line search gives up first the difference is the regulariser's share of the gradient:
        # 2e-4 relative in tau at 20 trials, less with every trial more; tests hold 1e-3 against the oracle's TNC call).  That root is found
        # for all latents together by the four-point lockstep finder of the batch M-step - three or four batched device passes instead of
        # ~25 rounds of one scipy evaluation each (config 4: the M-step was half of an iteration).
        def with_prior_multi(Q):
            Q = np.asarray(Q, dtype=np.float64)
            F, G = sess.ctx.mstep_tau_costgrad_multi(Q)
            tau = binSize / 1000 * (1 / np.exp(Q)) ** 0.5
            return np.asarray(F) + 0.5 * (tau - tau_old[None, :]) ** 2 / s ** 2, np.asarray(G) + (tau - tau_old[None, :]) / s ** 2
        pv, fv, gv, nfev, ok = _lockstep_multi(with_prior_multi, initp, d_hint=getattr(sess, '_tau_step_prior', None), gtol=1e-10)
        sess._tau_step_prior = pv - initp
        details = [op.OptimizeResult(x=np.array([pv[xd]]), fun=fv[xd], jac=np.array([gv[xd]]), nfev=nfev, success=bool(ok[xd]),
                                     message='lockstep multi-point root of the reference gradient') for xd in range(xdim)]
        return (1 / np.exp(pv)) ** 0.5 * binSize / 1000, details

    # the xdim scipy optimisations (one per latent, as in the reference) run concurrently: their cost/gradient requests
    # are served by one batched device pass per round
    def solve_one(xd, evaluate):
        def with_prior(v):
            pv = float(np.asarray(v).reshape(-1)[0])
            cost, grad = evaluate(pv)
            tau = binSize / 1000 * (1 / np.exp(pv)) ** 0.5
            return cost + 0.5 * (tau - tau_old[xd]) ** 2 / s ** 2, grad + (tau - tau_old[xd]) / s ** 2
        cache = _CostGradCache(with_prior)
        return op.minimize(fun=cache.fun, x0=initp[xd], jac=lambda v, c=cache: np.array([c.jac(v)]),
                           options={'disp': False, 'gtol': 1e-10}, method=tauOptimMethod)
    details = _ConcurrentScalarProblems(initp, sess.ctx.mstep_tau_costgrad_batch).run(solve_one)
    tempTau = np.array([(1 / np.exp(np.asarray(res.x).reshape(-1)[0])) ** 0.5 for res in details])
    return tempTau * binSize / 1000, details


# ------------------------------------------------------------------------------------------------
# updateParams with CdOptimMethod='newton' and the lockstep timescale finder: timescale rounds on the side stream beside the (C,d) passes.  OFF by
# default (PGPFA_MSTEP_OVERLAP=1 switches it on): same bits either way, and on one MI355X it buys nothing measurable (13.08 against 13.13 EM it/s,
# profiles/r06_bench_c3_mstep_*: a (C,d) pass holds every CU, the timescale round's small launches wait for one to come free)
M_STEP_OVERLAP = os.environ.get('PGPFA_MSTEP_OVERLAP', '0') == '1'


class _TauBeside:
    """The rounds of the lockstep timescale finder, each started on the device's side stream (pgpfa_mstep_tau_costgrad_multi_begin) just before a
    (C,d) pass goes to the main stream and collected right after it (round 6).  The two updates of the reference's M-step (learning.py:93-141 and
    257-293) are independent problems on disjoint inputs - (C,d): counts, post_mean, post_vsm; timescales: PautoSum - and the reference solves
    them one after the other; a timescale round is a latency-bound chain of small launches (1.3 ms with most of the chip idle), a (C,d) pass
    one or two compute-bound launches.  Same generator, same sample points, same device arithmetic as learnGPparams alone: same results."""

    def __init__(self, sess, initp):
        self.ctx = sess.ctx
        self.key = tuple(np.asarray(initp, dtype=np.float64).tolist())
        self.rounds = _lockstep_multi_rounds(initp, getattr(sess, '_tau_step', None))
        self.Q = next(self.rounds)
        self.inflight = False
        self.result = None

    def kick(self):
        if self.result is None and not self.inflight:
            self.ctx.mstep_tau_costgrad_multi_begin(self.Q)
            self.inflight = True

    def reap(self):
        if self.inflight:
            self.inflight = False
            FG = self.ctx.mstep_tau_costgrad_multi_end()
            try:
                self.Q = self.rounds.send(FG)
            except StopIteration as stop:
                self.result = stop.value

    def finish(self):
        while self.result is None:
            self.kick()
            self.reap()
        return self.result

    def abandon(self):
        if self.inflight:
            self.inflight = False
            try:
                self.ctx.mstep_tau_costgrad_multi_end()
            except Exception:
                pass


def updateParams(oldParams, infRes, experiment, CdOptimMethod='BFGS', CdMaxIter=None, tauMaxIter=None, verbose=False):
    """reference learning.py:295-309"""
    if verbose:
        print('Learning C,d...')
    if M_STEP_OVERLAP and CdOptimMethod == 'newton' and TAU_SOLVER == 'lockstep':
        xdim = np.shape(oldParams['C'])[1]
        sess = _resident_session(infRes, experiment, xdim)
        if hasattr(sess.ctx, 'mstep_tau_costgrad_multi_begin'):
            DevicePrecomp(sess, sess.T)                   # PautoSum (and its all-reduce) first: the timescale rounds read it
            oldTau = np.asarray(oldParams['tau'], dtype=np.float64) * 1000 / experiment.binSize
            beside = _TauBeside(sess, np.log(1 / oldTau ** 2))
            sess._beside_cd_pass = beside
            sess._tau_beside = beside
            try:
                newC, newd, obsOptimDetails = learnLTparams(oldParams, infRes, experiment, CdOptimMethod, CdMaxIter, verbose)
                sess._beside_cd_pass = None
                newTau, dynOptimDetails = learnGPparams(oldParams, infRes, experiment)
            finally:
                sess._beside_cd_pass = None
                sess._tau_beside = None
                beside.abandon()
            return {'C': newC, 'd': newd, 'tau': newTau}, {'Cd': obsOptimDetails, 'tau': dynOptimDetails}
    newC, newd, obsOptimDetails = learnLTparams(oldParams, infRes, experiment, CdOptimMethod, CdMaxIter, verbose)
    if verbose:
        print('Learning GP timescale constants')
    newTau, dynOptimDetails = learnGPparams(oldParams, infRes, experiment)
    return {'C': newC, 'd': newd, 'tau': newTau}, {'Cd': obsOptimDetails, 'tau': dynOptimDetails}


def learnLTparamsGradDescent(oldParams, infRes, experiment, stepSize, cumHess, updateCdJointly=True, hessTol=1e-5):
    """reference learning.py:874-907: one damped Newton-like step, vecCd <- vecCd - stepSize * h^-1 g with h the finite-difference
    Jacobian of the gradient of Q = -cost (4 q(p+1) device passes)."""
    if not updateCdJointly:
        raise NotImplementedError('updateCdJointly=False raises in the reference itself (learning.py:393) and is not built')
    ydim, xdim = np.shape(oldParams['C'])
    sess = _resident_session(infRes, experiment, xdim)
    vecCd = util.CdtoVecCd(oldParams['C'], oldParams['d'])

    def q_grad(v):
        return -sess.ctx.mstep_cd_costgrad(v)[1]
    g = q_grad(vecCd)
    h = util.approx_jacobian(vecCd, q_grad, hessTol)
    newC, newd = util.vecCdtoCd(vecCd - stepSize * np.linalg.inv(h) @ g, xdim, ydim)
    return newC, newd, h


def updateParamsWithGradDescent(oldParams, infRes, experiment, stepSize, cumHess, regularizer_stepsize_tau, tauOptimMethod,
                                updateCdJointly=True, verbose=False, hessTol=1e-5):
    """reference learning.py:909-945 ('grad' online mode)"""
    newC, newd, hess = learnLTparamsGradDescent(oldParams, infRes, experiment, stepSize, cumHess, updateCdJointly, hessTol)
    newTau, dynOptimDetails = learnGPparamsWithPrior(oldParams, infRes, experiment, tauOptimMethod, regularizer_stepsize_tau)
    return {'C': newC, 'd': newd, 'tau': newTau}, {'Cd': None, 'tau': dynOptimDetails}, hess


def updateParamsWithPrior(oldParams, infRes, experiment, CdOptimMethod, tauOptimMethod, regularizer_stepsize_Cd,
                          regularizer_stepsize_tau, prevInvPriorCov, covOpts='useHessian', verbose=False,
                          updateCdJointly=True, hessTol=1e-5):
    """reference learning.py:833-866 (covOpts='useDiag' is what the engine's default 'diag' mode passes, 'useHessian' the
    'hess' mode)."""
    if verbose:
        print('Learning C,d...')
    newC, newd, obsOptimDetails, invPriorCov = learnLTparamsWithPrior(
        oldParams, infRes, experiment, CdOptimMethod, regularizer_stepsize_Cd, prevInvPriorCov,
        covOpts=covOpts, updateCdJointly=updateCdJointly, hessTol=hessTol, verbose=verbose)
    if verbose:
        print('Learning GP timescale constants')
    newTau, dynOptimDetails = learnGPparamsWithPrior(oldParams, infRes, experiment, tauOptimMethod, regularizer_stepsize_tau)
    return {'C': newC, 'd': newd, 'tau': newTau}, {'Cd': obsOptimDetails, 'tau': dynOptimDetails}, invPriorCov
