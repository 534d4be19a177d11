"""Rounds of the lockstep timescale root finder per EM iteration at config-3 dimensions, for candidate-point offsets given on the command line
(default: the shipped ones).  usage: python tools/tau_rounds_probe.py [trials [iterations ["o1,o2,o3,o4"]]]"""
import os, sys, inspect
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session, learning

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 12
TRACE = os.environ.get('TAU_TRACE')
if TRACE:
    src = inspect.getsource(learning._lockstep_multi_np).replace("            pred_prev = np.where(work, r, pred_prev)", "            _TR.append(np.where(work, r, np.nan).copy())\n            pred_prev = np.where(work, r, pred_prev)")
    ns = {}
    learning.__dict__['_TR'] = []
    exec(src, learning.__dict__, ns)
    learning._lockstep_multi = ns['_lockstep_multi_np']          # (the array form: same results as the scalar one the M-step runs)
if len(sys.argv) > 3:
    src = inspect.getsource(learning._lockstep_multi_np).replace('offs = np.array([-1.5, -0.5, 0.5, 1.5])', 'offs = np.array([%s])' % sys.argv[3])
    ns = {}
    exec(src, learning.__dict__, ns)
    learning._lockstep_multi = ns['_lockstep_multi_np']          # (the array form: same results as the scalar one the M-step runs)
q, p, T = 200, 10, 500
_, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}
optim = None
rounds = []
for it in range(n_it):
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    params, det = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
    rounds.append(det['tau'][0].nfev)
    if TRACE and learning._TR:
        fin = np.array([float(d.x[0]) for d in det['tau']])
        print('      prediction errors per round (max over latents):', ' '.join('%.1e' % np.nanmax(np.abs(r - fin)) for r in learning._TR))
        learning._TR.clear()
    gmax = max(abs(float(d.jac[0])) for d in det['tau'])
    print('it %2d: rounds %d  max|g| %.2e  nll %.6f  tau %s' % (it, rounds[-1], gmax, nll, np.round(params['tau'], 4)), flush=True)
print('mean rounds %.2f' % np.mean(rounds[1:]))
