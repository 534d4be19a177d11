"""Accuracy of the low-rank covariance engine as a function of the pivoted-Cholesky tolerance (option lowrank_tol), on the config-3
fixture captured from the reference (tests/golden/c3_spot.npz): error of mode / post_vsm / post_vsmGP against the polished reference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import numpy as np
from funs import _hip
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'c3_spot.npz'))
q, p, T = 200, 10, 500
rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
for tol in (1e-13, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7):
    ctx = _hip.Context(q, p, T, 1, float(g['binSize']))
    ctx.upload_counts(g['Y']); ctx.set_option('cov_mode', 2); ctx.set_option('keep_trial_vsmgp', 1); ctx.set_option('lowrank_tol', tol)
    ctx.set_params(g['init_C'], g['init_d'], g['init_tau'])
    obj, _, st = ctx.estep_laplace()
    G = ctx.post_vsmgp()[0]
    print('tol %.0e: rank %4d  status %s  mode %.2e  obj %.2e  post_vsm %.2e  vsmGP diag %.2e rows %.2e' % (
        tol, int(ctx.info('lowrank_rtot')), st.tolist(), np.max(np.abs(ctx.post_mean()[0].reshape(-1) - g['polished'])),
        abs(obj - float(g['polished_f'])) / abs(float(g['polished_f'])), rel(ctx.post_vsm()[0], g['polished_vsm']),
        rel(np.stack([np.diag(G[:, :, k]) for k in range(p)]), g['polished_vsmGP_diag']), rel(G[::50], g['polished_vsmGP_rows'])), flush=True)
    ctx.close()
