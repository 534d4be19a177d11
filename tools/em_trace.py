"""Per-iteration trace of the default bench workload (config 3, 1024 trials on one GPU): E/M wall time, PCG work,
rank plan and the timescales, to see how the cost of an EM iteration moves as the parameters are learned.

usage: python tools/em_trace.py [iterations [neurons latents bins trials]]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 12
q, p, T, R = (int(a) for a in sys.argv[2:6]) if len(sys.argv) > 5 else (200, 10, 500, 1024)
seed = int(os.environ.get('TRACE_SEED', '0'))
true_params, Ys = bench.synth_shard(q, p, T, R, seed, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}
optim = None
c = sess.ctx
for kv in filter(None, os.environ.get('TRACE_OPTS', '').split(',')):      # e.g. TRACE_OPTS=splitk_target=960,pcg_inner=12
    k, v = kv.split('=')
    c.set_option(k, float(v))
print('true tau', np.sort(true_params['tau']).round(3))
for it in range(n_it):
    t0 = time.time()
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    t1 = time.time()
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
    t2 = time.time()
    print('it %2d: E %6.1f ms  M %5.1f ms  pcg/trial %5.1f  newton it max %2d  dense retries %d  rtot %4d  chunk %4d  nll %.2f  tau %s' % (
        it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, c.info('last_pcg_iterations') / R, c.info('last_newton_max_iter'), c.info('last_dense_retries'),
        c.info('lowrank_rtot'), c.info('chunk_trials'), nll, np.sort(params['tau']).round(3)), flush=True)
