"""Host utilities on either side of the hot path, with the names of the reference's funs/util.py:
vec layout helpers, the builders (Gram matrices come from the device kernel), trial subsampling with
the reference's RNG stream, the synthetic population generator and the Poisson-PCA initialiser.
Plotting, CRCNS/Matlab loaders and cross-validation helpers of the reference are outside the hot
path and are not provided."""
import copy
import sys

import numpy as np

from . import _hip


class Printer:
    """One-line progress output (reference util.py:121-128)."""

    def __init__(self, data):
        sys.stdout.write('\r\x1b[K' + str(data))
        sys.stdout.flush()

    @staticmethod
    def stdout(message):
        sys.stdout.write(message)
        sys.stdout.write('\b' * len(message))


# -- vec(C, d) layout (reference util.py:560-592) ----------------------------------------------------
def CdtoVecCd(C, d):
    C = np.asarray(C, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    return np.concatenate((C.T.ravel(), d))


def vecCdtoCd(vecCd, xdim, ydim):
    m = np.asarray(vecCd, dtype=np.float64).reshape(xdim + 1, ydim)
    return m[:xdim].T, m[xdim]


# -- builders (reference util.py:594-619) ---------------------------------------------------------------
def makeCd_big(params, T):
    """Kronecker expansion of (C, d).  Only for small diagnostics: the device path never forms it."""
    C_big = np.kron(params['C'], np.eye(T)).T
    d_big = np.kron(np.ndarray.flatten(np.asarray(params['d'])), np.ones(T)).T
    return C_big, d_big


def makeK_big(params, trialDur, binSize, epsNoise=0.001):
    """Per-latent RBF Gram matrices K (xdim,T,T) from the device kernel, and their block-diagonal K_big."""
    ydim, xdim = np.shape(params['C'])
    params['tau'] = np.ndarray.flatten(np.asarray(params['tau'], dtype=np.float64))
    T = int(trialDur / binSize)
    ctx = _hip.Context(ydim, xdim, T, 1, float(binSize))
    try:
        ctx.set_option('eps_noise', epsNoise)
        ctx.set_params(np.asarray(params['C'], dtype=np.float64), np.zeros(ydim), params['tau'])
        K = ctx.gram()
    finally:
        ctx.close()
    K_big = np.zeros((xdim * T, xdim * T))
    for xd in range(xdim):
        K_big[xd * T:(xd + 1) * T, xd * T:(xd + 1) * T] = K[xd]
    return K_big, K


# -- finite-difference Jacobian (reference util.py:377-434) -------------------------------------------------
def approx_jacobian(x, func, epsilon, *args):
    """Jacobian of the vector function func at x by the reference's fourth-order central differences,
    J[:, i] = (-f(x+2h) + 8 f(x+h) - 8 f(x-h) + f(x-2h)) / (12 h_i), with its step rule: h = epsilon / 2 (a scalar epsilon for
    every coordinate) or, for epsilon None, EPS^(1/3) * max(|x|, 0.1) / 2 (statsmodels' _get_epsilon with s = 3)."""
    x0 = np.atleast_1d(np.asarray(x, dtype=np.float64))
    n = x0.size
    if epsilon is None:
        h = np.finfo(float).eps ** (1.0 / 3.0) * np.maximum(np.abs(x0), 0.1)
    elif np.isscalar(epsilon):
        h = np.full(n, float(epsilon))
    else:
        h = np.asarray(epsilon, dtype=np.float64)
    h = h / 2.0
    f0 = np.asarray(func(x0, *args))
    jac = np.zeros((f0.size, n))
    for i in range(n):
        dx = np.zeros(n)
        dx[i] = h[i]
        jac[:, i] = (-np.asarray(func(x0 + 2 * dx, *args)) + 8 * np.asarray(func(x0 + dx, *args))
                     - 8 * np.asarray(func(x0 - dx, *args)) + np.asarray(func(x0 - 2 * dx, *args))) / (12 * h[i])
    return jac


# -- leave-one-neuron-out prediction (reference util.py:289-334) --------------------------------------------
def leaveOneOutPrediction(params, experiment):
    """For every trial and neuron: posterior mode of the latents given all OTHER neurons, then the held-out neuron's
    predicted rate exp(c_n x + d_n) per bin.  Returns (y_pred_mode[numTrials][ydim][T], pred_err_mode) like the
    reference; the numTrials*ydim mode searches run batched on the device (the reference solves them one by one
    with fmin_ncg from a cold start, to a looser tolerance)."""
    from . import _session
    xdim = np.shape(params['C'])[1]
    sess, trial_idx = _session.session_for(experiment, xdim)
    lo, hi = (0, len(trial_idx)) if getattr(experiment, '_pgpfa_local_shard', False) else sess.local_slice(len(trial_idx))
    sess.set_params(params)
    y_loc, err_loc = sess.ctx.loo_predict(trial_idx[lo:hi])
    unconverged = int(sess.ctx.info('last_loo_unconverged'))
    if unconverged:
        import warnings
        warnings.warn('leaveOneOutPrediction: %d of %d held-out mode searches did not converge; their predictions come from '
                      'the last iterate' % (unconverged, (hi - lo) * sess.q), RuntimeWarning, stacklevel=2)
    if sess.comm_ready and not getattr(experiment, '_pgpfa_local_shard', False):
        y_pred = np.zeros((len(trial_idx), sess.q, sess.T))
        y_pred[lo:hi] = y_loc
        y_pred = sess.allreduce(y_pred)
        err = float(sess.allreduce(np.array([err_loc]))[0])
        return y_pred, err
    return y_loc, float(err_loc)


# -- latent-dimensionality cross-validation (reference util.py:180-275) -------------------------------------
def splitTrainingTestDataset(experiment, numTrainingTrials, numTestTrials):
    """First numTrainingTrials trials / the numTestTrials after them, as shallow copies (reference util.py:263-275)."""
    if numTestTrials + numTrainingTrials > experiment.numTrials:
        print('Error: Number of training trials and test trials must sum to less than the number of available trials.')
    trainingSet, testSet = copy.copy(experiment), copy.copy(experiment)
    trainingSet.data = experiment.data[:numTrainingTrials]
    trainingSet.numTrials = numTrainingTrials
    testSet.data = experiment.data[numTrainingTrials:numTrainingTrials + numTestTrials]
    testSet.numTrials = numTestTrials
    for part in (trainingSet, testSet):                   # the copies are data sets of their own on the device
        part.__dict__.pop('_pgpfa_parent', None)
        part.__dict__.pop('batchTrIdx', None)
    return trainingSet, testSet


class crossValidation:
    """reference util.py:180-249: for xdim = 1..maxXdim fit on the training split and score the leave-one-neuron-out
    prediction error on the test split; optimXdim is the arg-min.  learningMethod: 'batch', 'diag', 'hess' or 'grad'."""

    def __init__(self, experiment, numTrainingTrials=10, numTestTrials=2, maxXdim=6, maxEMiter=3, batchSize=5,
                 inferenceMethod='laplace', learningMethod='batch', quiet=True):
        from . import engine
        if learningMethod not in ('batch', 'diag', 'hess', 'grad'):
            raise ValueError("learningMethod must be 'batch', 'diag', 'hess' or 'grad'")
        trainingSet, testSet = splitTrainingTestDataset(experiment, numTrainingTrials, numTestTrials)
        self.errs, self.fits = [], []
        for xdimFit in range(1, maxXdim + 1):
            initParams = initializeParams(xdimFit, trainingSet.ydim, trainingSet)
            if learningMethod == 'batch':
                fit = engine.PPGPFAfit(experiment=trainingSet, initParams=initParams, inferenceMethod=inferenceMethod,
                                       EMmode='Batch', maxEMiter=maxEMiter, quiet=quiet)
            else:
                fit = engine.PPGPFAfit(experiment=trainingSet, initParams=initParams, inferenceMethod=inferenceMethod,
                                       EMmode='Online', onlineParamUpdateMethod=learningMethod, maxEMiter=maxEMiter,
                                       batchSize=batchSize, quiet=quiet)
            _, predErr = leaveOneOutPrediction(fit.optimParams, testSet)
            self.errs.append(predErr)
            self.fits.append(fit)
        self.inferenceMethod, self.learningMethod = inferenceMethod, learningMethod
        self.optimXdim = int(np.argmin(self.errs)) + 1
        self.maxXdim = maxXdim


# -- minibatches (reference util.py:449-473) ---------------------------------------------------------------
def subsampleTrials(experiment, batchSize):
    """Same draw from the global legacy RNG as the reference (np.random.choice without replacement);
    the returned shallow copy remembers its parent so the device keeps using the resident counts."""
    numTrials = len(experiment.data)
    batchTrIdx = np.random.choice(numTrials, batchSize, replace=False)
    sub = copy.copy(experiment)
    sub.data = [experiment.data[i] for i in batchTrIdx]
    sub.numTrials = batchSize
    sub.batchTrIdx = batchTrIdx
    sub._pgpfa_parent = getattr(experiment, '_pgpfa_parent', experiment)
    if hasattr(experiment, 'batchTrIdx') and getattr(experiment, '_pgpfa_parent', None) is not None:
        sub.batchTrIdx = np.asarray(experiment.batchTrIdx)[batchTrIdx]
    return sub


def seenTrials(experiment, seenIdx):
    idx = np.asarray(seenIdx).flatten()
    seen = copy.copy(experiment)
    seen.data = [experiment.data[i] for i in idx]
    seen.numTrials = len(seen.data)
    return seen


# -- synthetic population (distributions of reference util.py:705-750) ---------------------------------------
class dataset:
    """Trials of population spike counts sampled from  x_k ~ GP(0, K(tau_k)),  y ~ Poisson(exp(Cx+d)).

    Attributes as in the reference: data (list of {'X','Y'}), xdim, ydim, T, trialDur, binSize, numTrials,
    seed, params.  `sampler='reference'` reproduces the reference's global-RNG stream bit for bit: the
    reference calls np.random.multivariate_normal on the (xdim*T)^2 covariance once per trial (util.py:738-741),
    which repeats the same SVD of K_big every time; here the SVD is taken once and each trial is the same
    standard_normal draw times the same sqrt(s) * v matrix - identical bits, without the per-trial
    (xdim*T)^3 (minutes per trial at config 3).  `sampler='cholesky'` draws the same distributions per latent
    through T x T factors from numpy's Generator (no (xdim*T)^2 matrix at all).  `sampler='device'` draws every trial on the
    GPU from a counter-based generator: 1024 config-3 trials in well under a second.
    """

    def __init__(self, trialDur=1000, binSize=10, drawSameX=False, numTrials=20, xdim=3, ydim=30, seed=12, dOffset=-1,
                 fixTau=False, fixedTau=None, params=None, model='pgpfa', sampler='reference', verbose=False):
        if model != 'pgpfa':
            raise NotImplementedError("only model='pgpfa' is on the hot path")
        self.trialDur, self.binSize, self.drawSameX = trialDur, binSize, drawSameX
        self.numTrials, self.xdim, self.ydim, self.seed = numTrials, xdim, ydim, seed
        self.T = int(trialDur / binSize)
        T = self.T
        np.random.seed(seed)
        if params is None:
            params = {'C': np.random.rand(ydim, xdim) - 0.5,
                      'd': np.random.rand(ydim) * (-2) + dOffset,
                      'tau': np.abs(np.random.rand(xdim)) + 0.01}
            if fixTau:
                params['tau'] = np.asarray(fixedTau, dtype=np.float64)
        self.params = params
        tau = np.asarray(params['tau'], dtype=np.float64).reshape(-1)
        t = np.arange(T, dtype=np.float64) * binSize
        dsq = (t[:, None] - t[None, :]) ** 2
        K = np.stack([0.999 * np.exp(-0.5 * (dsq / (tk * 1000.0) ** 2)) + 0.001 * np.eye(T) for tk in tau])
        offset = np.asarray(params['d'], dtype=np.float64)[:, None]
        data = []
        if sampler == 'reference':
            K_big = np.zeros((xdim * T, xdim * T))
            for k in range(xdim):
                K_big[k * T:(k + 1) * T, k * T:(k + 1) * T] = K[k]
            _, sv, vt = np.linalg.svd(K_big)                       # what legacy multivariate_normal factors cov with
            mix = np.sqrt(sv)[:, None] * vt
            mean = np.zeros(T * xdim)

            def draw():
                x = np.dot(np.random.standard_normal((1, T * xdim)).reshape(-1, T * xdim), mix)
                x += mean
                return np.reshape(x, [xdim, T])
            pois = lambda lam: np.random.poisson(lam=lam)
        elif sampler == 'cholesky':
            rng = np.random.default_rng(seed)
            Lk = np.linalg.cholesky(K)
            draw = lambda: np.einsum('kts,ks->kt', Lk, rng.standard_normal((xdim, T)))
            pois = lambda lam: rng.poisson(lam)
        elif sampler == 'device':
            # all trials on the GPU (pgpfa_generate): latents through the resident low-rank form of the Gram matrices, counts by a
            # counter-based generator keyed by `seed` - same distributions, NOT NumPy's stream (no fixture depends on this one)
            if drawSameX:
                raise NotImplementedError("drawSameX needs sampler='reference' or 'cholesky'")
            ctx = _hip.Context(ydim, xdim, T, numTrials, float(binSize))
            try:
                ctx.set_params(np.asarray(params['C'], dtype=np.float64), offset[:, 0], tau)
                Xd, Yd = ctx.generate(seed)
            finally:
                ctx.close()
            self.data = [{'X': Xd[i], 'Y': Yd[i]} for i in range(numTrials)]
            return
        else:
            raise ValueError("sampler must be 'reference', 'cholesky' or 'device'")
        X0 = draw() if drawSameX else None
        for i in range(numTrials):
            X = X0 if drawSameX else draw()
            data.append({'X': X, 'Y': pois(np.exp(params['C'] @ X + offset))})
            if verbose:
                Printer('Sampling trial %d ...' % (i + 1))
        self.data = data

    def getAllRaster(self):
        self.all_raster = np.concatenate([tr['Y'] for tr in self.data], axis=1)
        return self.all_raster

    def getAvgFiringRate(self):
        self.avgFR = float(np.mean(self.getAllRaster()) / self.binSize * 1000.0)
        return self.avgFR


def countMoments(experiment, xdim):
    """Mean and covariance of the spike counts over all (trial, bin) samples - np.mean / np.cov of the concatenated
    raster (reference util.py:523-533, engine.py:487-492) - from the device's exact integer sums of the resident
    count tensor; the raster is never formed on the host.  Returns (mean[q], cov[q][q], per-neuron totals, samples)."""
    from . import _session
    sess, trial_idx = _session.session_for(experiment, xdim)
    local = getattr(experiment, '_pgpfa_local_shard', False)
    lo, hi = (0, len(trial_idx)) if local else sess.local_slice(len(trial_idx))
    s, S, ns = sess.ctx.count_moments(trial_idx[lo:hi])
    s, S, ns = s.astype(np.float64), S.astype(np.float64), float(ns)      # exact: all sums are far below 2^53
    if sess.comm_ready:
        red = sess.allreduce(np.concatenate([s, S.reshape(-1), [ns]]))
        s, S, ns = red[:s.size], red[s.size:-1].reshape(S.shape), float(red[-1])
    mean = s / ns
    cov = (S - np.outer(s, s) / ns) / (ns - 1.0)
    return mean, cov, s, int(ns)


def getMeanCovYfromParams(params, experiment=None):
    """Mean and second moment of the counts implied by the parameters under a unit-variance latent (reference
    util.py:24-39): E[y] = exp(diag(CC^T)/2 + d), E[y_i y_j] = E[y_i]E[y_j]exp(CC^T_ij/2) (+ E[y_i] on the diagonal)."""
    C = np.asarray(params['C'], dtype=np.float64)
    lamb = C @ C.T
    E_y = np.exp(0.5 * np.diag(lamb) + np.asarray(params['d'], dtype=np.float64).reshape(-1))
    E_yy = np.outer(E_y, E_y) * np.exp(0.5 * lamb)
    E_yy[np.diag_indices_from(E_yy)] += E_y
    return E_y, E_yy


def JSLogdetDiv(X, Y):
    """reference util.py:21-22"""
    return np.log(np.linalg.det((X + Y) / 2)) - 0.5 * np.log(np.linalg.det(X.dot(Y)))


def initializeParams(xdim, ydim, experiment=None):
    """Poisson-PCA initialiser (reference util.py:505-558): moment-matched log-rate covariance ->
    leading eigenvectors -> C; d = log mean rate; tau ~ U(0.1, 0.6) s from the global RNG.  The count moments come
    from the device (countMoments); the ydim x ydim eigen-decomposition stays on the host like the reference's."""
    if experiment is None:
        return {'C': np.random.rand(ydim, xdim) * 2 - 1, 'd': np.random.randn(ydim) * 2 - 2, 'tau': np.random.rand(xdim) * 0.5}
    meanY, covY, _, _ = countMoments(experiment, xdim)
    meanY = meanY + 1e-10
    outer = np.outer(meanY, meanY)
    lamb = np.log(np.abs(covY + outer - np.diag(meanY))) - np.log(outer)
    evals, evecs = np.linalg.eig(lamb)
    order = np.argsort(evals)[::-1]
    return {'C': evecs[:, order][:, :xdim], 'd': np.log(meanY), 'tau': np.random.rand(xdim) * 0.5 + 0.1}


def subspaceAngle(F, G):
    """Largest principal angle between the column spaces of F and G (reference util.py:338-367),
    columns scaled by their maximum entry first as the reference does."""
    F = np.array(F, dtype=np.float64)
    G = np.array(G, dtype=np.float64)
    F = F / np.max(F, axis=0)
    G = G / np.max(G, axis=0)
    QF, _ = np.linalg.qr(F)
    QG, _ = np.linalg.qr(G)
    s = np.linalg.svd(QF.T @ QG, compute_uv=False)
    return float(np.max(np.arccos(np.minimum(s, 1.0))))
