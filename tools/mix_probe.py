"""Mixing pass of the split accumulation: E-step time at the bench dimensions with option mix_slot 1 (thread per bin, workgroup = (slot, 256 bins) walking whole
columns) and 0 (64 bins x 4 columns per workgroup).  usage: python tools/mix_probe.py [trials] [generating|init]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench, funs
from funs import _session, util
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
start = sys.argv[2] if len(sys.argv) > 2 else 'init'
q, p, T = 200, 10, 500
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
if start == 'generating':
    params = {k: np.asarray(v, dtype=np.float64).copy() for k, v in true.items()}
else:
    np.random.seed(0)
    params = util.initializeParams(p, q, exp)
optim = None
for it in range(2):
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
for v in (1, 0, 1):
    sess.ctx.set_option('mix_slot', v)
    sess.ctx.set_option('profile', 1)
    ts = []
    for it in range(3):
        t0 = time.time()
        infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
        ts.append((time.time() - t0) * 1e3)
    print('mix_slot %d: E-step %s ms  vsm-tagged kernels %.2f ms / E-step  (rank %d)' % (v, ' '.join('%.1f' % x for x in ts), sess.ctx.info('prof_vsm_ms') / 3,
                                                                                     int(sess.ctx.info('lowrank_rtot'))), flush=True)
    sess.ctx.set_option('profile', 0)
