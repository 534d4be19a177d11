"""Inner PCG solve, two-kernel step without the prior mat-vec (pcg_cg_a/b_kernel) against the split kernels of round 3, on the bench workload:
per E-step time, Newton-solve time and bytes, slot-iterations, and the agreement of modes / objective between the two forms.
usage: python tools/pcg_probe.py [trials] [em iterations] [generating|init] [forms, digits of the option's values, e.g. 101] [option, default pcg_form]
(e.g. ... 101 thin_products: thin.h's kernels against the block-sparse GEMMs for the preconditioner's F^T t / F v)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
import funs
from funs import _session, util

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 4
start = sys.argv[3] if len(sys.argv) > 3 else 'init'
q, p, T = 200, 10, 500
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
if start == 'generating':
    p0 = {k: np.asarray(v, dtype=np.float64).copy() for k, v in true.items()}
else:
    np.random.seed(0)
    p0 = util.initializeParams(p, q, exp)
out = {}
forms = [int(ch) for ch in (sys.argv[4] if len(sys.argv) > 4 else '101')]
option = sys.argv[5] if len(sys.argv) > 5 else 'pcg_form'
for onek in forms:
    sess.ctx.set_option(option, onek)
    for kv in filter(None, os.environ.get('PROBE_OPTS', '').split(',')):
        sess.ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    sess.ctx.set_option('time_newton', 1)
    params = {k: np.asarray(v, dtype=np.float64).copy() for k, v in p0.items()}
    optim = None
    nlls = []
    for it in range(n_it):
        t0 = time.time()
        infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim if it else None)
        t1 = time.time()
        nms, nby = sess.ctx.info('last_newton_solve_ms'), sess.ctx.info('last_newton_solve_bytes')
        print(option + ' %d it %d: E %.1f ms  newton %.2f ms  %.2f GB -> %.0f GB/s (%.3f of 8 TB/s)  pcg/trial %.2f  outer max %d  rank %d  nll %.10f' % (
            onek, it, (t1 - t0) * 1e3, nms, nby / 1e9, nby / 1e6 / max(nms, 1e-9), nby / 1e6 / max(nms, 1e-9) / 8000.0,
            sess.ctx.info('last_pcg_iterations') / R, int(sess.ctx.info('last_newton_max_iter')), int(sess.ctx.info('lowrank_rtot')), nll), flush=True)
        nlls.append(nll)
        modes = np.array(sess.ctx.post_mean())
        params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
    out[onek] = (np.array(nlls), modes, {k: np.array(v) for k, v in params.items()})
if 0 not in out or 1 not in out:
    sys.exit(0)
a, b = out[1], out[0]
print('nll difference (%s 1 vs 0):' % option, np.max(np.abs(a[0] - b[0])), ' modes max abs diff after %d iterations: %.3e' % (n_it, np.max(np.abs(a[1] - b[1]))))
print('parameter difference: C %.3e d %.3e tau %.3e' % tuple(np.max(np.abs(a[2][k] - b[2][k])) for k in ('C', 'd', 'tau')))
