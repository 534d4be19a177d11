"""Config-4-style stochastic EM on one GPU: R trials resident, minibatches of 1024 (the engine's 'Online' / 'diag' loop,
written out so that per-iteration device statistics can be printed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import numpy as np
import bench, funs
from funs import _session
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
q, p, T, _ = bench.CONFIGS['c3']
true_params, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
exp._pgpfa_local_shard = False
np.random.seed(0)
params = funs.util.initializeParams(p, q, exp)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in params.items()}
sess, _ = _session.session_for(exp, p)
np.random.seed(1)
prior = np.diag(np.ones(q * (p + 1)))
for n in range(iters):
    sub = funs.util.subsampleTrials(exp, 1024)
    t0 = time.time()
    infRes, nll, _ = funs.inference.laplace(sub, params, prevOptimRes='resident')
    t1 = time.time()
    step = 1.0 / (n + 1) ** 0.75
    params, det, prior = funs.learning.updateParamsWithPrior(params, infRes, sub, 'newton', 'TNC', step, step, prior, covOpts='useDiag')
    t2 = time.time()
    c = sess.ctx
    print('it %d: E %.3f s (pcg/trial %.1f, dense retries %d, lowrank plan %d, rtot %d, chunk %d)  M %.3f s  nPLL %.3f' % (
        n, t1 - t0, c.info('last_pcg_iterations') / 1024, c.info('last_dense_retries'), c.info('plan_lowrank'), c.info('lowrank_rtot'),
        c.info('chunk_trials'), t2 - t1, nll), flush=True)
