"""GEMM launches of EM iterations at the bench workload's dimensions, grouped by operand shape, with the rate each shape runs at (HIP events
around every launch: option profile = 2).  usage: python tools/gemm_shapes.py [iterations (default 6) [neurons latents bins trials]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
q, p, T, R = (int(a) for a in sys.argv[2:6]) if len(sys.argv) > 5 else (200, 10, 500, 1024)
_, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
for kv in filter(None, os.environ.get('PGPFA_OPTS', '').split(',')):          # context options key=value,... (experiments)
    sess.ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}
optim = None
for it in range(n_it):
    if it == 2:
        sess.ctx.set_option('profile', 2)          # the first two iterations (workspace plan, cold start) are left out
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
print('GEMM launches of EM iterations 3..%d at %d x %d x %d, %d trials (rank %d at the end):' % (n_it, q, p, T, R, int(sess.ctx.info('lowrank_rtot'))))
print(sess.ctx.gemm_shape_report())
