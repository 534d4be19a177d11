#!/usr/bin/env python3
"""Golden fixtures for the finite-difference-Hessian online EM variants ('hess', 'grad'; engine.py:354-397,
learning.py:546-549, 874-945), captured by IMPORTING the real reference on config 1 (batchSize 5, 3 iterations,
np.random.seed(1)).  The statsmodels stand-in of make_golden.py is completed here with statsmodels' own step rule for a
scalar epsilon (fill), which util.approx_jacobian relies on.

    python tests/golden/make_golden_online_fd.py        # writes tests/golden/c1_em_online_fd.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402

np, util, engine = mg.np, mg.util, mg.engine


def _get_epsilon(x, s, epsilon, n):     # statsmodels.tools.numdiff._get_epsilon
    if epsilon is None:
        return np.finfo(float).eps ** (1.0 / s) * np.maximum(np.abs(np.asarray(x)), 0.1)
    if np.isscalar(epsilon):
        h = np.empty(n)
        h.fill(epsilon)
        return h
    return np.asarray(epsilon)


util.nd._get_epsilon = _get_epsilon


def main():
    out = {}
    for mode in ('hess', 'grad'):
        with mg.quiet():
            ds = util.dataset()
            np.random.seed(0)
            init = util.initializeParams(3, 30, ds)
            np.random.seed(1)
            fit = engine.PPGPFAfit(ds, initParams=dict(init), inferenceMethod='laplace', EMmode='Online', maxEMiter=3, batchSize=5,
                                   onlineParamUpdateMethod=mode, CdOptimMethod='TNC', tauOptimMethod='TNC')
        out[mode + '_nll'] = np.asarray(fit.posteriorLikelihood)
        out[mode + '_seq_C'] = np.stack([p['C'] for p in fit.paramSeq])
        out[mode + '_seq_d'] = np.stack([p['d'] for p in fit.paramSeq])
        out[mode + '_seq_tau'] = np.stack([np.asarray(p['tau']).reshape(-1) for p in fit.paramSeq])
        if mode == 'hess':
            out['hess_invPriorCov1'] = fit.invPriorCovs[1]
        else:
            out['grad_cumHess1'] = fit.cumHess[1]
        print(mode, 'nll', out[mode + '_nll'])
    np.savez_compressed(os.path.join(HERE, 'c1_em_online_fd.npz'), **out)


if __name__ == '__main__':
    main()
