import os, sys, time
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
infRes, nll, optim = funs.inference.laplace(exp, params)
vec = util.CdtoVecCd(params['C'], params['d'])
sess.ctx.mstep_cd_costgrad(vec)
t0 = time.time()
for i in range(50):
    sess.ctx.mstep_cd_costgrad(vec + 1e-6 * i)
print('mstep_cd_costgrad: %.3f ms per call' % ((time.time() - t0) / 50 * 1e3))
n = [0]
orig = sess.ctx.mstep_cd_costgrad
def counted(*a, **k):
    n[0] += 1
    return orig(*a, **k)
sess.ctx.mstep_cd_costgrad = counted
t0 = time.time()
C, d, det = learning.learnLTparams(params, infRes, exp, 'TNC')
print('learnLTparams TNC: %.1f ms, %d cost/grad calls, nfev %s' % ((time.time() - t0) * 1e3, n[0], getattr(det, 'nfev', None)))
