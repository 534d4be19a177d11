"""CPU oracle for the Poisson-GPFA EM hot path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``poisson-gpfa_amd/`` imports it, and the product path raises when the
HIP library is missing instead of falling back to anything in here.

It is a from-scratch numpy/scipy restatement of the algorithm of the reference
(mackelab/poisson-gpfa, pure Python; citations are ``file:line`` into the reference
tree).  Two flavours of every E-step quantity are provided:

* *structured* - never forms the Kronecker "big" matrices; works on ``X[p,T]``,
  ``Y[q,T]``, ``K[p,T,T]``.  This is what the HIP kernels are diffed against.
* *faithful* (``*_big`` / ``mode='faithful'``) - forms ``C_big``/``K_big`` exactly as
  the reference does (util.py:594-619) and calls the same ``scipy.optimize`` drivers
  with the same options, so that it reproduces the reference's early-stopped
  answers and its CPU cost.  Used to pin the oracle against the golden vectors and
  as ``bench.py``'s ``cpu_baseline`` (kind "port").

Parity pinning: the reference has no tests of its own, so the
oracle is pinned by ``tests/golden/*.npz``, produced in the build container by
importing the real reference (``tests/golden/make_golden.py``) and compared in
``tests/test_oracle_golden.py``.

Vector layouts (verified against the reference numerically):
  xbar[k*T + t] = X[k, t]   (latent-major; inference.py:129)
  ybar[n*T + t] = Y[n, t]   (inference.py:97)
  vecCd = [C[:,0], ..., C[:,p-1], d]   (util.py:560-574)
"""
import copy
import numpy as np
import scipy.optimize as op

EPS_NOISE = 1e-3   # util.py:599, learning.py:286,822


# --------------------------------------------------------------------------------------
# L0 builders (util.py:560-619)
# --------------------------------------------------------------------------------------
def cd_to_vec(C, d):
    """util.CdtoVecCd, util.py:560-574: columns of C stacked, then d."""
    C = np.asarray(C, dtype=np.float64)
    return np.concatenate([C.T.reshape(-1), np.asarray(d, dtype=np.float64).reshape(-1)])


def vec_to_cd(vec, p, q):
    """util.vecCdtoCd, util.py:576-592."""
    m = np.asarray(vec, dtype=np.float64).reshape(p + 1, q)
    return m[:p].T.copy(), m[p].copy()


def make_K(tau, T, binSize, epsNoise=EPS_NOISE):
    """Per-latent RBF Gram matrices, util.makeK_big util.py:599-619.

    K[k,i,j] = (1-eps) * exp(-0.5 * ((i-j)*binSize)^2 / (1000*tau_k)^2) + eps*[i==j].
    The reference evaluates (T[i]*binSize - T[j]*binSize)**2 / (tau*1000)**2 in that
    order; the same order is kept so entries agree to the last bit.
    """
    tau = np.asarray(tau, dtype=np.float64).reshape(-1)
    t = np.arange(T, dtype=np.float64) * binSize
    num = (t[:, None] - t[None, :]) ** 2
    K = np.empty((tau.size, T, T))
    for k in range(tau.size):
        K[k] = (1.0 - epsNoise) * np.exp(-0.5 * (num / (tau[k] * 1000.0) ** 2))
        K[k] += epsNoise * np.eye(T)
    return K


def make_K_big(K):
    """Block-diagonal (pT x pT) prior covariance, util.py:616-617."""
    p, T, _ = K.shape
    Kb = np.zeros((p * T, p * T))
    for k in range(p):
        Kb[k * T:(k + 1) * T, k * T:(k + 1) * T] = K[k]
    return Kb


def make_Cd_big(C, d, T):
    """util.makeCd_big util.py:594-597: C_big = kron(C, I_T).T (pT x qT), d_big."""
    return np.kron(C, np.eye(T)).T, np.kron(np.asarray(d).reshape(-1), np.ones(T))


# --------------------------------------------------------------------------------------
# Laplace objective / gradient / Hessian
# --------------------------------------------------------------------------------------
def nlp_big(xbar, ybar, C_big, d_big, K_bigInv):
    """inference.negLogPosteriorUnNorm, inference.py:12-32 (faithful big-matrix form)."""
    h = C_big.T @ xbar + d_big
    return np.exp(h).sum() - ybar @ h + 0.5 * xbar @ (K_bigInv @ xbar)


def nlp_big_grad(xbar, ybar, C_big, d_big, K_bigInv):
    """inference.py:34-48."""
    h = C_big.T @ xbar + d_big
    return C_big @ (np.exp(h) - ybar) + K_bigInv @ xbar


def nlp_big_hess(xbar, ybar, C_big, d_big, K_bigInv):
    """inference.py:50-65: C_big diag(exp h) C_big^T + K^-1 through the dense product."""
    h = C_big.T @ xbar + d_big
    return C_big @ (np.exp(h)[:, None] * C_big.T) + K_bigInv


def nlp(X, Y, C, d, Kinv):
    """Structured objective: sum exp(h) - sum y*h + 0.5 sum_k x_k^T Kinv_k x_k, h=CX+d."""
    h = C @ X + d[:, None]
    prior = 0.5 * np.einsum('kt,kts,ks->', X, Kinv, X)
    return np.exp(h).sum() - (Y * h).sum() + prior


def nlp_grad(X, Y, C, d, Kinv):
    """Structured gradient (p,T): C^T(exp(h) - Y) + [Kinv_k x_k]_k  (inference.py:34-48)."""
    h = C @ X + d[:, None]
    return C.T @ (np.exp(h) - Y) + np.einsum('kts,ks->kt', Kinv, X)


def poisson_blocks(X, C, d):
    """W[t] = C^T diag(exp(h[:,t])) C, the (p x p) per-bin likelihood curvature."""
    e = np.exp(C @ X + d[:, None])                      # (q,T)
    return np.einsum('nk,nt,nl->tkl', C, e, C)


def nlp_hess(X, Y, C, d, Kinv):
    """Structured Hessian assembled dense, latent-major (inference.py:50-65):
    H[(k,t),(l,s)] = [t==s] W[t,k,l] + [k==l] Kinv[k,t,s]."""
    p, T = X.shape
    W = poisson_blocks(X, C, d)
    H = np.zeros((p, T, p, T))
    ar = np.arange(T)
    for k in range(p):
        H[k, :, k, :] += Kinv[k]
        for l in range(p):
            H[k, ar, l, ar] += W[:, k, l]
    return H.reshape(p * T, p * T)


def marginal_blocks(Sigma, p, T):
    """inference.py:164-172: post_vsmGP (T,T,p) diagonal blocks, post_vsm (T,p,p)."""
    vsmGP = np.empty((T, T, p))
    for k in range(p):
        vsmGP[:, :, k] = Sigma[k * T:(k + 1) * T, k * T:(k + 1) * T]
    vsm = np.empty((T, p, p))
    for t in range(T):
        vsm[t] = Sigma[t::T, t::T]
    return vsmGP, vsm


def newton_mode(Y, C, d, Kinv, x0=None, xtol=1e-11, max_iter=100):
    """Exact (polished) Laplace mode by damped Newton on the structured objective.

    Not in the reference (it uses scipy Newton-CG, inference.py:119-126); this is the
    tightly converged answer the HIP path is compared with (SURVEY 8c "polished mode").
    Armijo backtracking with a rounding-noise slack (the objective is ~1e3 and its
    evaluation noise ~1e-10, so a pure Armijo test stalls near the mode); stops after
    the first step with max|step| < xtol.  Returns (X, f, iterations).
    """
    q, T = Y.shape
    p = C.shape[1]
    X = np.zeros((p, T)) if x0 is None else np.array(x0, dtype=np.float64).reshape(p, T)
    f = nlp(X, Y, C, d, Kinv)
    it = 0
    for it in range(1, max_iter + 1):
        g = nlp_grad(X, Y, C, d, Kinv).reshape(-1)
        H = nlp_hess(X, Y, C, d, Kinv)
        step = -np.linalg.solve(H, g)
        dec = -(g @ step)                      # Newton decrement squared
        slack = 1e-12 * (1.0 + abs(f))
        a = 1.0
        while True:
            Xn = X + a * step.reshape(p, T)
            fn = nlp(Xn, Y, C, d, Kinv)
            if np.isfinite(fn) and fn <= f - 1e-4 * a * dec + slack:
                break
            a *= 0.5
            if a < 1e-10:
                break
        X, f = Xn, fn
        if a * np.max(np.abs(step)) < xtol:
            break
    return X, f, it


def laplace(Ys, params, binSize, prev=None, mode='faithful', return_cov=True):
    """inference.laplace, inference.py:67-185.

    Ys: sequence of (q,T) count arrays.  mode 'faithful' runs scipy Newton-CG on the
    big-matrix callbacks with the reference's options; mode 'exact' runs the
    structured polished Newton.  Returns (infRes, -mean objective, optimRes) like the
    reference (inference.py:175-185); infRes['post_cov'] is omitted if return_cov is
    False.
    """
    C = np.asarray(params['C'], dtype=np.float64)
    d = np.asarray(params['d'], dtype=np.float64).reshape(-1)
    q, p = C.shape
    T = Ys[0].shape[1]
    K = make_K(params['tau'], T, binSize)
    if mode == 'faithful':
        C_big, d_big = make_Cd_big(C, d, T)
        K_bigInv = np.linalg.inv(make_K_big(K))               # inference.py:82
    else:
        Kinv = np.linalg.inv(K)
    res = {'post_mean': [], 'post_cov': [], 'post_vsm': [], 'post_vsmGP': []}
    optim, total = [], 0.0
    info = {'nit': [], 'nhev': [], 'status': []}
    for r, Y in enumerate(Ys):
        Y = np.asarray(Y, dtype=np.float64)
        x0 = np.zeros(p * T) if prev is None else np.asarray(prev[r]).reshape(-1)
        if mode == 'faithful':
            ybar = Y.reshape(-1)
            out = op.minimize(nlp_big, x0, args=(ybar, C_big, d_big, K_bigInv),
                              method='Newton-CG', jac=nlp_big_grad, hess=nlp_big_hess,
                              options={'disp': False, 'maxiter': 10000})   # inference.py:119-126
            x, f = out.x, out.fun
            H = nlp_big_hess(x, ybar, C_big, d_big, K_bigInv)             # inference.py:130
            info['nit'].append(out.nit); info['nhev'].append(out.nhev); info['status'].append(out.status)
        else:
            X, f, nit = newton_mode(Y, C, d, Kinv, x0)
            x = X.reshape(-1)
            H = nlp_hess(X, Y, C, d, Kinv)
            info['nit'].append(nit); info['nhev'].append(nit); info['status'].append(0)
        Sigma = np.linalg.inv(H)                                            # inference.py:131
        vsmGP, vsm = marginal_blocks(Sigma, p, T)
        optim.append(x.copy())
        total += f
        res['post_mean'].append(x.reshape(p, T).copy())
        if return_cov:
            res['post_cov'].append(Sigma)
        res['post_vsm'].append(vsm)
        res['post_vsmGP'].append(vsmGP)
    if not return_cov:
        del res['post_cov']
    res['_info'] = info
    return res, -total / len(Ys), optim


def leave_one_out_prediction(Ys, params, binSize, mode='faithful'):
    """util.leaveOneOutPrediction util.py:289-334 (= engine.PPGPFAfit.leaveOneOutPrediction, engine.py:599-644).

    For every trial and neuron: the Laplace mode of the latents given all OTHER neurons (cold start at zero),
    the held-out neuron's predicted rate exp(c_n x + d_n) per bin, and the summed squared error against its
    counts.  mode 'faithful' runs scipy's fmin_ncg on the big-matrix callbacks with the reference's (default)
    options; mode 'exact' runs the structured polished Newton.  Returns (y_pred[R][q][T], err)."""
    C = np.asarray(params['C'], dtype=np.float64)
    d = np.asarray(params['d'], dtype=np.float64).reshape(-1)
    q, p = C.shape
    T = Ys[0].shape[1]
    K = make_K(params['tau'], T, binSize)
    Kinv = np.linalg.inv(K)
    K_bigInv = np.linalg.inv(make_K_big(K)) if mode == 'faithful' else None
    pred = np.zeros((len(Ys), q, T))
    err = 0.0
    for r, Y in enumerate(Ys):
        Y = np.asarray(Y, dtype=np.float64)
        for n in range(q):
            Cw, dw, Yw = np.delete(C, n, 0), np.delete(d, n, 0), np.delete(Y, n, 0)       # util.py:301-303,315
            if mode == 'faithful':
                C_big, d_big = make_Cd_big(Cw, dw, T)
                x = op.fmin_ncg(nlp_big, np.zeros(p * T), fprime=nlp_big_grad, fhess=nlp_big_hess,
                                args=(Yw.reshape(-1), C_big, d_big, K_bigInv), disp=False)  # util.py:318-325
                X = x.reshape(p, T)
            else:
                X, _, _ = newton_mode(Yw, Cw, dw, Kinv)
            yp = np.exp(C[n] @ X + d[n])                                                       # util.py:328
            pred[r, n] = yp
            err += float((Y[n] - yp) @ (Y[n] - yp))                                            # util.py:329
    return pred, err


# --------------------------------------------------------------------------------------
# M-step: observation parameters (C, d)
# --------------------------------------------------------------------------------------
def mstep_cd_terms(vecCd, Ys, post_mean, post_vsm, p, q):
    """Shared pass of learning.MStepObservationCost(_grad), learning.py:20-91.

    hh = C m + d ; rho[n,t] = c_n^T V_t c_n ; yhat = exp(hh + rho/2)
    f  = sum(y*hh - yhat)
    dC = (y - yhat) m^T - [sum_t yhat[n,t] V_t c_n]_n ; dd = sum_t (y - yhat)
    Returns (f_sum, dC_sum, dd_sum) summed over trials (not yet divided by R).
    """
    C, d = vec_to_cd(vecCd, p, q)
    f, dC, dd = 0.0, np.zeros((q, p)), np.zeros(q)
    for Y, m, V in zip(Ys, post_mean, post_vsm):
        hh = C @ m + d[:, None]
        VC = np.einsum('tkl,nl->ntk', V, C)                  # V_t c_n
        rho = np.einsum('ntk,nk->nt', VC, C)
        yhat = np.exp(hh + 0.5 * rho)
        f += (Y * hh - yhat).sum()
        dC += (Y - yhat) @ m.T - np.einsum('nt,ntk->nk', yhat, VC)
        dd += (Y - yhat).sum(axis=1)
    return f, dC, dd


def mstep_cd_cost(vecCd, Ys, post_mean, post_vsm, p, q):
    """learning.MStepObservationCost, learning.py:20-49: -f / numTrials."""
    f, _, _ = mstep_cd_terms(vecCd, Ys, post_mean, post_vsm, p, q)
    return -f / len(Ys)


def mstep_cd_grad(vecCd, Ys, post_mean, post_vsm, p, q):
    """learning.MStepObservationCost_grad, learning.py:51-91."""
    _, dC, dd = mstep_cd_terms(vecCd, Ys, post_mean, post_vsm, p, q)
    return -cd_to_vec(dC, dd) / len(Ys)


def mstep_cd_cost_prior(vecCd, old_vec, inv_prior, Ys, post_mean, post_vsm, p, q):
    """learning.MStepObservationCostWithPrior, learning.py:445-485: the reference
    subtracts 0.5*(v-old)^T invPriorCov (v-old) with invPriorCov NEGATIVE definite
    ('useDiag': -I/s^2, learning.py:580-581), i.e. adds |v-old|^2/(2 s^2)."""
    dv = np.asarray(vecCd) - old_vec
    return mstep_cd_cost(vecCd, Ys, post_mean, post_vsm, p, q) - 0.5 * dv @ (inv_prior @ dv)


def mstep_cd_grad_prior(vecCd, old_vec, inv_prior, Ys, post_mean, post_vsm, p, q):
    """learning.MStepObservationCostWithPrior_grad, learning.py:487-534."""
    dv = np.asarray(vecCd) - old_vec
    return mstep_cd_grad(vecCd, Ys, post_mean, post_vsm, p, q) - inv_prior @ dv


def learn_cd(params, Ys, infRes, method='TNC', max_iter=None):
    """learning.learnLTparams, learning.py:93-141 (same scipy call and options)."""
    C0 = np.asarray(params['C'], dtype=np.float64)
    q, p = C0.shape
    args = (Ys, infRes['post_mean'], infRes['post_vsm'], p, q)
    out = op.minimize(mstep_cd_cost, cd_to_vec(C0, params['d']), args=args, jac=mstep_cd_grad,
                      method=method, options={'disp': False, 'maxiter': max_iter})
    C, d = vec_to_cd(out.x, p, q)
    return C, d, out.fun, out


def learn_cd_prior(params, Ys, infRes, method, step_cd):
    """learning.learnLTparamsWithPrior with covOpts='useDiag', updateCdJointly=True,
    learning.py:536-627: invPriorCov = -I/step^2, options {'gtol':1e-10}."""
    C0 = np.asarray(params['C'], dtype=np.float64)
    q, p = C0.shape
    old = cd_to_vec(C0, params['d'])
    inv_prior = -np.eye(old.size) / step_cd ** 2
    args = (old, inv_prior, Ys, infRes['post_mean'], infRes['post_vsm'], p, q)
    kw = dict(args=args, jac=mstep_cd_grad_prior, method=method, options={'disp': False, 'gtol': 1e-10})
    if method == 'L-BFGS-B':
        kw['bounds'] = [(None, None)] * old.size
    out = op.minimize(mstep_cd_cost_prior, old, **kw)
    C, d = vec_to_cd(out.x, p, q)
    return C, d, out.fun, inv_prior, out


# --------------------------------------------------------------------------------------
# M-step: GP timescales
# --------------------------------------------------------------------------------------
def make_precomp(infRes):
    """learning.makePrecomp, learning.py:145-173: PautoSum_k = sum_r Sigma_r^{kk} + m m^T."""
    p, T = infRes['post_mean'][0].shape
    P = np.zeros((p, T, T))
    for m, G in zip(infRes['post_mean'], infRes['post_vsmGP']):
        for k in range(p):
            P[k] += G[:, :, k] + np.outer(m[k], m[k])
    return P, len(infRes['post_mean'])


def _tau_pieces(pv, T, epsNoise):
    idx = np.arange(T, dtype=np.float64)
    dsq = (idx[:, None] - idx[None, :]) ** 2
    g = np.exp(pv)
    S = (1.0 - epsNoise) * np.exp(-0.5 * g * dsq)
    K = S + epsNoise * np.eye(T)
    return K, -0.5 * S * dsq


def tau_cost(pv, Pauto, R, epsNoise=EPS_NOISE):
    """learning.MStepGPtimescaleCost, learning.py:175-214:
    0.5*R*logdet K + 0.5*tr(K^-1 PautoSum), K built from gamma = exp(p) in bins."""
    pv = float(np.asarray(pv).reshape(-1)[0])
    K, _ = _tau_pieces(pv, Pauto.shape[0], epsNoise)
    sign, ld = np.linalg.slogdet(K)
    return 0.5 * R * sign * ld + 0.5 * np.sum(np.linalg.inv(K) * Pauto)


def tau_grad(pv, Pauto, R, epsNoise=EPS_NOISE):
    """learning.MStepGPtimescaleCost_grad, learning.py:216-255:
    -exp(p) * (-0.5 R tr(K^-1 M) + 0.5 tr(K^-1 M K^-1 PautoSum)), M = dK/dgamma.
    The reference evaluates the traces over half the matrix using persymmetry
    (learning.py:231-252); the full traces here are equal up to rounding."""
    pv = float(np.asarray(pv).reshape(-1)[0])
    K, M = _tau_pieces(pv, Pauto.shape[0], epsNoise)
    Ki = np.linalg.inv(K)
    KiM = Ki @ M
    dE = -0.5 * R * np.trace(KiM) + 0.5 * np.sum((KiM @ Ki) * Pauto.T)
    return np.array([-dE * np.exp(pv)])


def learn_tau(params, infRes, binSize):
    """learning.learnGPparams, learning.py:257-293 (default scipy method = BFGS, gtol 1e-8)."""
    P, R = make_precomp(infRes)
    tau_bins = np.asarray(params['tau'], dtype=np.float64) * 1000.0 / binSize
    new, details = np.zeros(P.shape[0]), []
    for k in range(P.shape[0]):
        out = op.minimize(tau_cost, np.log(1.0 / tau_bins[k] ** 2), args=(P[k], R, EPS_NOISE),
                          jac=tau_grad, options={'disp': False, 'gtol': 1e-8})
        details.append(out)
        new[k] = (1.0 / np.exp(out.x[0])) ** 0.5
    return new * binSize / 1000.0, details


def tau_cost_prior(pv, Pauto, R, binSize, old_tau, step, epsNoise=EPS_NOISE):
    """learning.MStepGPtimescaleCostWithPrior, learning.py:681-724: + 0.5 (tau-old)^2/step^2, tau in s."""
    pv0 = float(np.asarray(pv).reshape(-1)[0])
    tau = binSize / 1000.0 * (1.0 / np.exp(pv0)) ** 0.5
    return tau_cost(pv0, Pauto, R, epsNoise) + 0.5 * (tau - old_tau) ** 2 / step ** 2


def tau_grad_prior(pv, Pauto, R, binSize, old_tau, step, epsNoise=EPS_NOISE):
    """learning.MStepGPtimescaleCostWithPrior_grad, learning.py:726-769.  The reference adds
    d(reg)/d(tau) to a d/dp gradient WITHOUT the chain-rule factor (learning.py:733-734,769);
    reproduced as is, because it is part of the reference's results."""
    pv0 = float(np.asarray(pv).reshape(-1)[0])
    tau = binSize / 1000.0 * (1.0 / np.exp(pv0)) ** 0.5
    return tau_grad(pv0, Pauto, R, epsNoise) + (tau - old_tau) / step ** 2


def learn_tau_prior(params, infRes, binSize, method, step):
    """learning.learnGPparamsWithPrior, learning.py:771-830."""
    P, R = make_precomp(infRes)
    tau_old = np.asarray(params['tau'], dtype=np.float64)
    tau_bins = tau_old * 1000.0 / binSize
    new, details = np.zeros(P.shape[0]), []
    for k in range(P.shape[0]):
        out = op.minimize(tau_cost_prior, np.log(1.0 / tau_bins[k] ** 2),
                          args=(P[k], R, binSize, tau_old[k], step), jac=tau_grad_prior,
                          options={'disp': False, 'gtol': 1e-10}, method=method)
        details.append(out)
        new[k] = (1.0 / np.exp(out.x[0])) ** 0.5
    return new * binSize / 1000.0, details


def update_params(params, Ys, infRes, binSize, cd_method='BFGS', cd_max_iter=None):
    """learning.updateParams, learning.py:295-309."""
    C, d, cost, _ = learn_cd(params, Ys, infRes, cd_method, cd_max_iter)
    tau, det = learn_tau(params, infRes, binSize)
    return {'C': C, 'd': d, 'tau': tau}, {'Cd': cost, 'tau': det}


def update_params_prior(params, Ys, infRes, binSize, cd_method, tau_method, step_cd, step_tau):
    """learning.updateParamsWithPrior with covOpts='useDiag', learning.py:833-866."""
    C, d, cost, inv_prior, _ = learn_cd_prior(params, Ys, infRes, cd_method, step_cd)
    tau, det = learn_tau_prior(params, infRes, binSize, tau_method, step_tau)
    return {'C': C, 'd': d, 'tau': tau}, {'Cd': cost, 'tau': det}, inv_prior


# --------------------------------------------------------------------------------------
# Dual variational E-step (inference.py:188-432)
# --------------------------------------------------------------------------------------
def vi_post_cov(Kinv_big, C_big, lam):
    """inference.VIPostCov, inference.py:188-191 (1e-6 relative diagonal jitter inside the inverse)."""
    P = Kinv_big + (C_big * lam[None, :]) @ C_big.T
    return np.linalg.inv(P + 1e-6 * np.diag(np.diag(P))), P


def vi_post_mean(K_big, C_big, ybar, lam):
    """inference.VIPostMean, inference.py:193-194."""
    return -(K_big @ C_big) @ (lam - ybar)


def dual_cost(lam, ybar, C_big, K_big, Kinv_big, d_big):
    """inference.dualProblem, inference.py:196-213."""
    S, _ = vi_post_cov(Kinv_big, C_big, lam)
    lmy = lam - ybar
    v = C_big @ lmy
    sign, ld = np.linalg.slogdet(S)
    return 0.5 * v @ (K_big @ v) - d_big @ lmy + 0.5 * ld + lam @ (np.log(lam) - 1.0)


def dual_grad(lam, ybar, C_big, K_big, Kinv_big, d_big):
    """inference.dualProblem_grad, inference.py:215-219."""
    S, _ = vi_post_cov(Kinv_big, C_big, lam)
    lmy = lam - ybar
    quad = np.einsum('im,ij,jm->m', C_big, S, C_big)
    return C_big.T @ (K_big @ (C_big @ lmy)) - d_big + np.log(lam) - 0.5 * quad


def vi_post_cov_faithful(Kinv_big, C_big, lam):
    """inference.VIPostCov exactly as written (inference.py:187-190): the (m x m) diagonal matrix np.diag(lamb) is FORMED and multiplied
    densely - 2 n m^2 flops and 8 m^2 bytes where the structured form above needs 2 n^2 m.  Used only to TIME the reference's arithmetic
    (bench.py cpu_baseline of the config-5 workload); values equal vi_post_cov's."""
    P = Kinv_big + np.dot(np.dot(C_big, np.diag(lam)), C_big.T)
    return np.linalg.inv(P + 1e-6 * np.diag(np.diag(P))), P


def dual_cost_faithful(lam, ybar, C_big, K_big, Kinv_big, d_big):
    """inference.dualProblem as written, inference.py:196-213 (VIPostMean is evaluated and dropped there too)."""
    _ = vi_post_mean(K_big, C_big, ybar, lam)
    S, _ = vi_post_cov_faithful(Kinv_big, C_big, lam)
    lmy = lam - ybar
    A = 0.5 * np.dot(lmy.T, np.dot(C_big.T, np.dot(K_big, np.dot(C_big, lmy))))
    sign, ld = np.linalg.slogdet(S)
    return A - np.dot(d_big.T, lmy) + 0.5 * ld + np.dot(lam.T, np.log(lam) - np.ones(len(ybar)))


def dual_grad_faithful(lam, ybar, C_big, K_big, Kinv_big, d_big):
    """inference.dualProblem_grad as written, inference.py:215-219: the diagonal of the dense (m x m) product C_big^T Sigma C_big."""
    S, _ = vi_post_cov_faithful(Kinv_big, C_big, lam)
    lmy = lam - ybar
    return np.dot(C_big.T, np.dot(K_big, np.dot(C_big, lmy))) - d_big + np.log(lam) - 0.5 * np.diag(np.dot(C_big.T, np.dot(S, C_big)))


def dual_cost_rho(rho, *a):
    """inference.dualProblemRho, inference.py:222-244."""
    return dual_cost(np.exp(rho), *a)


def dual_grad_rho(rho, *a):
    """inference.dualProblemRho_grad, inference.py:246-256."""
    return dual_grad(np.exp(rho), *a) * np.exp(rho)


def dual_variational(Ys, params, binSize, log_lambda=False, prev=None):
    """inference.dualVariational, inference.py:259-432.
    Returns (infRes, -mean neg-log-posterior at the VI mean, mean dual optimum, optimRes)."""
    C = np.asarray(params['C'], dtype=np.float64)
    d = np.asarray(params['d'], dtype=np.float64).reshape(-1)
    q, p = C.shape
    T = Ys[0].shape[1]
    C_big, d_big = make_Cd_big(C, d, T)
    K_big = make_K_big(make_K(params['tau'], T, binSize))
    Kinv_big = np.linalg.inv(K_big)
    res = {'post_mean': [], 'post_cov': [], 'post_vsm': [], 'post_vsmGP': []}
    optim, lik, vlb = [], 0.0, 0.0
    for r, Y in enumerate(Ys):
        ybar = np.asarray(Y, dtype=np.float64).reshape(-1)
        args = (ybar, C_big, K_big, Kinv_big, d_big)
        if not log_lambda:
            x0 = np.zeros(q * T) + 0.5 if prev is None else prev[r]       # inference.py:294-297
            out = op.fmin_l_bfgs_b(dual_cost, x0, fprime=dual_grad, args=args, approx_grad=False,
                                   bounds=[(1e-10, None)] * (q * T), factr=1e7, disp=False)
            lam = out[0]
        else:
            x0 = np.zeros(q * T) if prev is None else prev[r]             # inference.py:358-361
            out = op.fmin_l_bfgs_b(dual_cost_rho, x0, fprime=dual_grad_rho, args=args, disp=False)
            lam = np.exp(out[0])
        optim.append(out[0])
        vlb += out[1]
        mean = vi_post_mean(K_big, C_big, ybar, lam)
        S, _ = vi_post_cov(Kinv_big, C_big, lam)
        lik += nlp_big(mean, ybar, C_big, d_big, Kinv_big)                # inference.py:333
        vsmGP, vsm = marginal_blocks(S, p, T)
        res['post_mean'].append(mean.reshape(p, T))
        res['post_cov'].append(S)
        res['post_vsm'].append(vsm)
        res['post_vsmGP'].append(vsmGP)
    n = len(Ys)
    return res, -lik / n, vlb / n, optim


# --------------------------------------------------------------------------------------
# Data: synthetic generator, initialiser, minibatch sampler
# --------------------------------------------------------------------------------------
def synth_params(q, p, seed, dOffset=-1.0, fixed_tau=None):
    """Parameter draw of util.dataset.__init__, util.py:705-717 (legacy global RNG, order C,d,tau)."""
    np.random.seed(seed)
    C = np.random.rand(q, p) - 0.5
    d = np.random.rand(q) * (-2) + dOffset
    tau = np.abs(np.random.rand(p)) + 0.01
    if fixed_tau is not None:
        tau = np.asarray(fixed_tau, dtype=np.float64)
    return {'C': C, 'd': d, 'tau': tau}


def synth_dataset(q, p, T, R, seed=12, binSize=10, dOffset=-1.0, fixed_tau=None, exact_reference_stream=False):
    """Synthetic P-GPFA population, distributions of util.dataset util.py:705-750.

    With exact_reference_stream=True the latent draw is np.random.multivariate_normal on
    K_big exactly like the reference (SVD of (pT)^2 per trial: only sensible at config-1
    sizes) and the RNG stream then matches util.dataset bit for bit.  Otherwise X is
    sampled per latent through a T x T Cholesky factor from a Generator seeded with
    `seed` (same distributions, different stream) - SURVEY 8d.
    Returns (params, Ys list of (q,T) float64 counts, Xs).
    """
    params = synth_params(q, p, seed, dOffset, fixed_tau)
    K = make_K(params['tau'], T, binSize)
    Ys, Xs = [], []
    if exact_reference_stream:
        K_big = make_K_big(K)
        for _ in range(R):
            X = np.random.multivariate_normal(np.zeros(p * T), K_big, 1).reshape(p, T)
            lam = np.exp(params['C'] @ X + params['d'][:, None])
            Ys.append(np.random.poisson(lam=lam).astype(np.float64))
            Xs.append(X)
    else:
        rng = np.random.default_rng(seed)
        L = np.linalg.cholesky(K)
        for _ in range(R):
            X = np.einsum('kts,ks->kt', L, rng.standard_normal((p, T)))
            lam = np.exp(params['C'] @ X + params['d'][:, None])
            Ys.append(rng.poisson(lam).astype(np.float64))
            Xs.append(X)
    return params, Ys, Xs


def initialize_params(Ys, p, seed=None):
    """Poisson-PCA initialiser, util.initializeParams util.py:505-558."""
    if seed is not None:
        np.random.seed(seed)
    spikes = np.concatenate(Ys, axis=1)
    meanY = spikes.mean(axis=1) + 1e-10
    covY = np.cov(spikes)
    lamb = np.log(np.abs(covY + np.outer(meanY, meanY) - np.diag(meanY))) - np.log(np.outer(meanY, meanY))
    evals, evecs = np.linalg.eig(lamb)
    order = np.argsort(evals)[::-1]
    return {'C': evecs[:, order][:, :p], 'd': np.log(meanY), 'tau': np.random.rand(p) * 0.5 + 0.1}


def subsample_trials(R, batch):
    """util.subsampleTrials, util.py:459-473: indices from the global legacy RNG."""
    return np.random.choice(R, batch, replace=False)


# --------------------------------------------------------------------------------------
# EM loops (engine.py:180-238 batch, engine.py:288-448 online 'diag')
# --------------------------------------------------------------------------------------
def fit_batch(Ys, init_params, binSize, n_iter, cd_method='TNC', mode='faithful', variational=False,
              log_lambda=False):
    params = copy.deepcopy(init_params)
    nll, vlbs, seq, prev = [], [], [copy.deepcopy(init_params)], None
    for _ in range(n_iter):
        if variational:
            infRes, v, vlb, prev = dual_variational(Ys, params, binSize, log_lambda, prev)
            vlbs.append(vlb)
        else:
            infRes, v, prev = laplace(Ys, params, binSize, prev, mode, return_cov=False)
        nll.append(v)
        params, _ = update_params(params, Ys, infRes, binSize, cd_method)
        seq.append(copy.deepcopy(params))
    return {'nll': nll, 'vlb': vlbs, 'paramSeq': seq, 'infRes': infRes}


def fit_online_diag(Ys, init_params, binSize, n_iter, batch, cd_method='TNC', tau_method='TNC',
                    step_pow=0.75, mode='faithful'):
    params = copy.deepcopy(init_params)
    steps = 1.0 / (np.arange(n_iter) + 1) ** step_pow                     # engine.py:276-277
    nll, seq, idxs = [], [copy.deepcopy(init_params)], []
    for n in range(n_iter):
        idx = subsample_trials(len(Ys), batch)                            # engine.py:293
        idxs.append(idx)
        sub = [Ys[i] for i in idx]
        infRes, v, _ = laplace(sub, params, binSize, None, mode, return_cov=False)   # cold start, engine.py:298-301
        nll.append(v)
        params, _, _ = update_params_prior(params, sub, infRes, binSize, cd_method, tau_method,
                                           steps[n], steps[n])
        seq.append(copy.deepcopy(params))
    return {'nll': nll, 'paramSeq': seq, 'batchTrIdx': idxs}
