"""EM iterations of the bench workload AT the generating parameters (rank 1120: where this population's fit settles), for kernel traces of the
settled regime.  usage: python tools/plateau_probe.py [iterations]   (under rocprofv3 --kernel-trace for the per-kernel split)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
q, p, T, R = 200, 10, 500, 1024
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
params = {k: np.asarray(v, dtype=np.float64).copy() for k, v in true.items()}
optim = None
for it in range(n_it):
    t0 = time.time()
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    t1 = time.time()
    params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
    t2 = time.time()
    print('it %d: E %.1f ms  M %.1f ms  pcg/trial %.1f  rank %d' % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sess.ctx.info('last_pcg_iterations') / R,
                                                                  int(sess.ctx.info('lowrank_rtot'))), flush=True)
