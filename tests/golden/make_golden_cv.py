#!/usr/bin/env python3
"""Golden fixture for the latent-dimensionality cross-validation driver (util.crossValidation, util.py:180-249), captured
by IMPORTING the real reference on config 1: 10 training / 2 test trials, xdim = 1..3, 2 batch EM iterations, seed 0.

    python tests/golden/make_golden_cv.py        # writes tests/golden/c1_cv.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402

np, util = mg.np, mg.util


def main():
    with mg.quiet():
        ds = util.dataset()
        np.random.seed(0)
        cv = util.crossValidation(ds, numTrainingTrials=10, numTestTrials=2, maxXdim=3, maxEMiter=2, learningMethod='batch')
    np.savez_compressed(os.path.join(HERE, 'c1_cv.npz'), errs=np.asarray(cv.errs), optimXdim=cv.optimXdim,
                        tau_fit3=np.asarray(cv.fits[2].optimParams['tau']).reshape(-1))
    print('errs', cv.errs, 'optimXdim', cv.optimXdim)


if __name__ == '__main__':
    main()
