"""Size of the mixing correction D = eps Wt y of the low-rank covariance engine on the bench workload: max_t eps ||Wt_t||_inf per EM iteration
(option measure_mix).  usage: python tools/mix_norm_probe.py [iterations]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 8
q, p, T, R = 200, 10, 500, 1024
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}
optim = None
for it in range(n_it + 1):
    if it == n_it:
        params = {k: np.asarray(v, dtype=np.float64) for k, v in true.items()}      # the generating parameters
    sess.ctx.set_option('measure_mix', 1)
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    print('iteration %2d%s: max_t eps ||Wt_t||_inf = %.3e   (rank %d)' % (it, ' (generating parameters)' if it == n_it else '',
                                                                         sess.ctx.info('last_eps_wt_norm'), int(sess.ctx.info('lowrank_rtot'))))
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
