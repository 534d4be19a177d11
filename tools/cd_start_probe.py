"""How far the extrapolated start of the Newton (C,d) driver (learning._newton_cd, CD_EXTRAPOLATE) is from the optimum it then finds: per EM
iteration at config 3 the displacement of (C,d), what one previous displacement misses of it, what the trend of the last two misses, the
device passes taken, and (verbose) the step of every pass.  usage: python tools/cd_start_probe.py"""
import os, sys
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench, funs
from funs import _session, util, learning
q, p, T, R = 200, 10, 500, 1024
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
np.random.seed(0)
params = util.initializeParams(p, q, exp)
optim = None
for it in range(14):
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    old = util.CdtoVecCd(params['C'], params['d'])
    tr = getattr(sess, '_cd_track', None)
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton', verbose=(it >= 3))
    new = util.CdtoVecCd(params['C'], params['d'])
    if tr is not None:
        ahead = tr['step'] if tr.get('prev_step') is None else 2 * tr['step'] - tr['prev_step']
        print('it %d: |displacement| %.2e  first-order miss %.2e  trend miss %.2e  passes %s' % (it, np.max(np.abs(new - old)), np.max(np.abs(new - old - tr['step'])), np.max(np.abs(new - old - ahead)), sess._cd_passes), flush=True)
