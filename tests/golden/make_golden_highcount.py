#!/usr/bin/env python3
"""Golden fixture for spike counts above 255 (the reference keeps counts as float64 of any size, util.py:741,750), captured by
IMPORTING the real reference: the first 3 trials of the config-1 data set with a handful of bins raised to 256...1200 spikes
(a 1-s bin of a 300-Hz unit gives 300), pushed through the reference's callbacks (inference.py:12-65), its Laplace E-step
(inference.py:67-185, modes polished on its own callbacks), its (C,d) M-step cost/gradient (learning.py:20-91), makePrecomp
(learning.py:145-173) and its dual callbacks (inference.py:196-219).  Same accommodations as make_golden.py.

    python tests/golden/make_golden_highcount.py        # writes tests/golden/c1_highcount.npz
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402  (sets up the statsmodels stand-in, cwd, imports)

np, util, inference, learning = mg.np, mg.util, mg.inference, mg.learning

N_TRIALS = 3


def main():
    with mg.quiet():
        ds = util.dataset()                         # config 1 (seed 12)
        np.random.seed(0)
        init = util.initializeParams(3, 30, ds)
    q, p, T = 30, 3, 100
    rng = np.random.RandomState(21)
    data = []
    for r in range(N_TRIALS):
        Y = np.array(ds.data[r]['Y'], dtype=float)
        for _ in range(6):                          # six loud bins per trial, spread over neurons
            Y[rng.randint(q), rng.randint(T)] = float(rng.choice([256, 300, 511, 512, 777, 1200]))
        data.append({'Y': Y, 'X': ds.data[r]['X']})
    data[0]['Y'][4, 17] = 300.0                     # "a 300-count bin"
    sub = types.SimpleNamespace(data=data, numTrials=N_TRIALS, ydim=ds.ydim, T=ds.T, trialDur=ds.trialDur, binSize=ds.binSize)
    Y = np.stack([tr['Y'] for tr in data])
    params = {'C': init['C'].copy(), 'd': init['d'].copy(), 'tau': np.array(init['tau']).copy()}
    K_big, K = util.makeK_big(dict(params), ds.trialDur, ds.binSize)
    C_big, d_big = util.makeCd_big(params, T)
    K_bigInv = np.linalg.inv(K_big)
    xprobe = 0.3 * rng.randn(p * T)
    ybar = Y[0].reshape(-1)
    f = inference.negLogPosteriorUnNorm(xprobe, ybar, C_big, d_big, K_bigInv, p, q)
    g = inference.negLogPosteriorUnNorm_grad(xprobe, ybar, C_big, d_big, K_bigInv, p, q)
    with mg.quiet():
        infRes, nll, lapOpt = inference.laplace(sub, dict(params))
    polished = []
    for r in range(N_TRIALS):
        yb = Y[r].reshape(-1)
        x = lapOpt[r].copy()
        for _ in range(80):
            gg = inference.negLogPosteriorUnNorm_grad(x, yb, C_big, d_big, K_bigInv, p, q)
            HH = inference.negLogPosteriorUnNorm_hess(x, yb, C_big, d_big, K_bigInv, p, q)
            dx = np.linalg.solve(HH, gg)
            x = x - dx
            if np.max(np.abs(dx)) < 1e-13:
                break
        polished.append(x)
    # covariance blocks at the polished modes (the reference's own slicing, inference.py:131-170)
    vsm_pol, fpol = [], 0.0
    for r in range(N_TRIALS):
        yb = Y[r].reshape(-1)
        HH = inference.negLogPosteriorUnNorm_hess(polished[r], yb, C_big, d_big, K_bigInv, p, q)
        S = np.linalg.inv(HH)
        vsm_pol.append(np.stack([S[np.ix_(np.arange(p) * T + t, np.arange(p) * T + t)] for t in range(T)]))
        fpol += inference.negLogPosteriorUnNorm(polished[r], yb, C_big, d_big, K_bigInv, p, q)
    v1 = util.CdtoVecCd(params['C'], params['d']) + 0.05 * rng.randn(q * (p + 1))
    cost1 = learning.MStepObservationCost(v1, p, q, sub, infRes)
    grad1 = learning.MStepObservationCost_grad(v1, p, q, sub, infRes)
    precomp = learning.makePrecomp(infRes)
    lam = 0.1 + rng.rand(q * T)
    dcost = inference.dualProblem(lam, ybar, C_big, K_big, K_bigInv, d_big)
    dgrad = inference.dualProblem_grad(lam, ybar, C_big, K_big, K_bigInv, d_big)
    ras = np.concatenate([tr['Y'] for tr in data], axis=1)
    np.savez_compressed(os.path.join(HERE, 'c1_highcount.npz'), Y=Y.astype(np.uint16), xprobe=xprobe, f=f, g=g, nll=nll,
                        post_mean=np.stack(infRes['post_mean']), post_vsm=np.stack(infRes['post_vsm']),
                        polished=np.stack(polished), post_vsm_polished=np.stack(vsm_pol), nlp_polished_sum=fpol,
                        v1=v1, cost1=cost1, grad1=grad1, PautoSum=np.stack([pc['PautoSum'] for pc in precomp]),
                        lam=lam, dual_cost=dcost, dual_grad=dgrad, raster_sum=ras.sum(axis=1), raster_cross=ras @ ras.T)
    print('c1_highcount.npz: max count', Y.max(), 'nll', nll, 'f', f)


if __name__ == '__main__':
    main()
