#!/usr/bin/env python3
"""Golden fixture for the elliptical-slice MCMC chain, captured by IMPORTING the real reference
(funs/mcmc.py PosteriorMCMC) on trial 2 of the config-1 data set with the Poisson-PCA initial parameters, NumPy seed 7.

    python tests/golden/make_golden_mcmc.py        # writes tests/golden/c1_mcmc.npz
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402

np, util = mg.np, mg.util
sys.path.insert(0, os.path.join(mg.REF, 'funs'))
import mcmc                             # noqa: E402  (the reference module imports `inference`, `util` bare)


def main():
    with mg.quiet():
        ds = util.dataset()
        np.random.seed(0)
        init = util.initializeParams(3, 30, ds)
    params = {'C': init['C'].copy(), 'd': init['d'].copy(), 'tau': np.array(init['tau']).copy()}
    np.random.seed(7)
    chain = mcmc.PosteriorMCMC(ds, params, 40, 2)
    np.savez_compressed(os.path.join(HERE, 'c1_mcmc.npz'), chain=chain, trial=2, seed=7, n_samples=40)
    print('c1_mcmc.npz: chain', chain.shape, 'last sample head', chain[-1][:3])


if __name__ == '__main__':
    main()
