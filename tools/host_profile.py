import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench, funs
from funs import _session, util
q, p, T, R = 200, 10, 500, 1024
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
np.random.seed(0)
params = util.initializeParams(p, q, exp)
optim = None
for it in range(3):
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
pr = cProfile.Profile()
pr.enable()
for it in range(10):
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(18)
