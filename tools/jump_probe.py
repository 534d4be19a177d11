"""What an E-step costs right after the parameters JUMP (cross-validation folds, a second fit in one process, a fit evaluated at other parameters).
The bench's fit runs a few EM iterations from the Poisson-PCA start, then the parameters are set to the generating ones (ranks 500 -> 1120) and four
more iterations run there; afterwards every timescale is halved.  Per E-step: wall time, set_params time, inner iterations, dense retries and why,
time spent re-planning the workspace, arena size.
usage: python tools/jump_probe.py [warm iterations] [key=value context options ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session

n_warm = int(sys.argv[1]) if len(sys.argv) > 1 and '=' not in sys.argv[1] else 6
q, p, T, R = 200, 10, 500, 1024
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
for kv in [a for a in sys.argv[1:] if '=' in a]:
    sess.ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}
optim = None
ctx = sess.ctx
keys = ('last_dense_retries', 'last_retry_ms', 'last_fallback_no_descent', 'last_fallback_line_search', 'last_fallback_outer_cap', 'last_param_step',
        'last_param_step_prev', 'plans', 'plan_ms_total', 'last_cold_restarts', 'arena_grow_ms_total')


def step(tag):
    global params, optim
    t0 = time.time()
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    t1 = time.time()
    new, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
    t2 = time.time()
    v = {k: ctx.info(k) for k in keys}
    print('%-14s E %7.1f ms (library %7.1f)  M %5.1f ms  pcg/trial %5.1f  rank %4d  retries %3d (%6.1f ms; no-descent %d, search %d, cap %d)  '
          'cold restarts %d  par step %.3g (prev %.3g)  plans %d (%.0f ms, growth %.0f ms)  arena %.1f GB  nll %.6f'
          % (tag, (t1 - t0) * 1e3, ctx.info('last_estep_ms'), (t2 - t1) * 1e3, ctx.info('last_pcg_iterations') / R, int(ctx.info('lowrank_rtot')),
             v['last_dense_retries'], v['last_retry_ms'], v['last_fallback_no_descent'], v['last_fallback_line_search'], v['last_fallback_outer_cap'],
             v['last_cold_restarts'], v['last_param_step'], v['last_param_step_prev'], v['plans'], v['plan_ms_total'], v['arena_grow_ms_total'],
             ctx.info('arena_bytes') / 1e9, nll), flush=True)
    params = new


for i in range(n_warm):
    step('fit %d' % i)
params = {k: np.asarray(v, dtype=np.float64).copy() for k, v in true.items()}
for i in range(4):
    step('at truth %d' % i)
params = dict(params, tau=params['tau'] * 0.5)
for i in range(3):
    step('tau halved %d' % i)
