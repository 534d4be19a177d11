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
