#!/usr/bin/env python3
"""Golden fixture for leave-one-neuron-out prediction, captured by IMPORTING the real
reference (util.leaveOneOutPrediction, util.py:289-334) on the first trials of the config-1 data set with the
Poisson-PCA initial parameters.  Same accommodations as make_golden.py (which this script imports for them).

    python tests/golden/make_golden_loo.py        # writes tests/golden/c1_loo.npz
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402  (sets up the statsmodels stand-in, cwd, imports)

np, util = mg.np, mg.util

N_TRIALS = 3


def main():
    with mg.quiet():
        ds = util.dataset()                         # config 1 (seed 12)
        np.random.seed(0)
        init = util.initializeParams(3, 30, ds)
    sub = types.SimpleNamespace(data=ds.data[:N_TRIALS], numTrials=N_TRIALS, ydim=ds.ydim, T=ds.T,
                                trialDur=ds.trialDur, binSize=ds.binSize)
    params = {'C': init['C'].copy(), 'd': init['d'].copy(), 'tau': np.array(init['tau']).copy()}
    with mg.quiet():
        y_pred, err = util.leaveOneOutPrediction(params, sub)
    np.savez_compressed(os.path.join(HERE, 'c1_loo.npz'), n_trials=N_TRIALS, y_pred_mode=y_pred, pred_err_mode=err)
    print('c1_loo.npz: y_pred', y_pred.shape, 'err', err)


if __name__ == '__main__':
    main()
