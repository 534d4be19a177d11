"""Elliptical-slice MCMC over one trial's latent trajectory (reference funs/mcmc.py) - an independent
check of the Gaussian posterior approximations.  The random draws stay on the host's legacy NumPy stream, in the
reference's order, so that a seeded run reproduces the reference's chain; every log-density evaluation runs on the
device (the Laplace objective kernel)."""
import math

import numpy as np

from . import inference


def PosteriorMCMC(experiment, params, maxSampleIter, trial):
    """reference mcmc.py:9-36.  Returns the chain, shape (maxSampleIter, xdim*T), latent-major.
    The slice density is minus the FULL Laplace objective (likelihood and GP prior, mcmc.py:26) although the ellipse
    already carries the prior - the reference's choice, kept because the chain depends on it."""
    C = np.asarray(params['C'])
    xdim = C.shape[1]
    T = int(experiment.T)
    sess, trial_idx = inference._prepare(experiment, params)
    tr = trial_idx[np.asarray([trial])]
    # chol(K_big) is block diagonal: one T x T Cholesky factor per latent (mcmc.py:29)
    K = sess.ctx.gram()
    chol = [np.linalg.cholesky(K[k]) for k in range(xdim)]

    def prior_draw():
        z = np.random.normal(size=xdim * T)                       # one draw of length xdim*T, as the reference's np.dot(prior, normal(D))
        return np.concatenate([chol[k] @ z[k * T:(k + 1) * T] for k in range(xdim)])

    def lnpdf(x):
        f, _ = sess.ctx.laplace_eval(tr, x[None, :], want_grad=False)
        return -float(f[0])

    x = np.zeros(xdim * T)
    samples = []
    for _ in range(maxSampleIter):
        x, _ = elliptical_slice(x, prior_draw, lnpdf)
        samples.append(x)
    return np.asarray(samples)


def elliptical_slice(initial_theta, prior_draw, lnpdf, cur_lnpdf=None):
    """One elliptical slice update (Murray, Adams & MacKay 2010; reference mcmc.py:39-105 with its default whole-ellipse
    bracket).  prior_draw() returns a sample of the Gaussian the ellipse is built from.  Random numbers are consumed in the
    reference's order: the prior draw, the slice height, the first angle, then one uniform per shrink."""
    if cur_lnpdf is None:
        cur_lnpdf = lnpdf(initial_theta)
    nu = prior_draw()
    hh = math.log(np.random.uniform()) + cur_lnpdf
    phi = np.random.uniform() * 2.0 * math.pi
    phi_min, phi_max = phi - 2.0 * math.pi, phi
    while True:
        prop = initial_theta * math.cos(phi) + nu * math.sin(phi)
        cur_lnpdf = lnpdf(prop)
        if cur_lnpdf > hh:
            return prop, cur_lnpdf
        if phi > 0:
            phi_max = phi
        elif phi < 0:
            phi_min = phi
        else:
            raise RuntimeError('slice shrunk to the current point and it is still not acceptable')
        phi = np.random.uniform() * (phi_max - phi_min) + phi_min


def PosteriorMCMC_batch(experiment, params, maxSampleIter, trials, seeds):
    """Chains for several trials at once: chain i is exactly what ``np.random.seed(seeds[i]);
    PosteriorMCMC(experiment, params, maxSampleIter, trials[i])`` returns - each chain consumes its own legacy RandomState in
    the reference's order - but the chains advance in lockstep, every round of proposals of ALL chains being ONE batched device
    evaluation of the log-density (pgpfa_laplace_eval over the list of trials) instead of one launch per proposal per trial.
    `trials` must be distinct.  Returns an array (len(trials), maxSampleIter, xdim*T)."""
    C = np.asarray(params['C'])
    xdim = C.shape[1]
    T = int(experiment.T)
    n = xdim * T
    trials = np.asarray(trials, dtype=np.int64)
    if len(set(trials.tolist())) != len(trials):
        raise ValueError('trials must be distinct')
    if len(seeds) != len(trials):
        raise ValueError('one seed per chain')
    sess, trial_idx = inference._prepare(experiment, params)
    dev = trial_idx[trials]
    K = sess.ctx.gram()
    chol = [np.linalg.cholesky(K[k]) for k in range(xdim)]
    rngs = [np.random.RandomState(int(s)) for s in seeds]
    nch = len(trials)

    def lnpdf(rows, X):
        f, _ = sess.ctx.laplace_eval(dev[rows], X, want_grad=False)
        return -f

    def prior_draw(rs):
        z = rs.normal(size=n)
        return np.concatenate([chol[k] @ z[k * T:(k + 1) * T] for k in range(xdim)])

    x = np.zeros((nch, n))
    cur = None                                   # the reference re-evaluates the current point at the start of every update
    out = np.zeros((nch, maxSampleIter, n))
    for it in range(maxSampleIter):
        cur = lnpdf(np.arange(nch), x)
        nu = np.stack([prior_draw(rs) for rs in rngs])
        hh = np.array([math.log(rs.uniform()) for rs in rngs]) + cur
        phi = np.array([rs.uniform() * 2.0 * math.pi for rs in rngs])
        phi_min, phi_max = phi - 2.0 * math.pi, phi.copy()
        pending = np.arange(nch)
        new_x = x.copy()
        while len(pending):
            prop = x[pending] * np.cos(phi[pending])[:, None] + nu[pending] * np.sin(phi[pending])[:, None]
            val = lnpdf(pending, prop)
            ok = val > hh[pending]
            new_x[pending[ok]] = prop[ok]
            rest = pending[~ok]
            for i in rest:
                if phi[i] > 0:
                    phi_max[i] = phi[i]
                elif phi[i] < 0:
                    phi_min[i] = phi[i]
                else:
                    raise RuntimeError('slice shrunk to the current point and it is still not acceptable')
                phi[i] = rngs[i].uniform() * (phi_max[i] - phi_min[i]) + phi_min[i]
            pending = rest
        x = new_x
        out[:, it, :] = x
    return out
