"""Where the first E-step of a process goes (config-3 dimensions): context + upload, a one-trial E-step (code objects, small
workspace), the first full E-step (workspace plan for the whole batch: the big allocation), and a warm one.

usage: python tools/cold_start_probe.py [neurons latents bins trials]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
t_import = time.time()
import bench
import funs
from funs import _session
print('imports                %7.1f ms' % ((time.time() - t_import) * 1e3))

q, p, T, R = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (200, 10, 500, 1024)
true_params, Ys = bench.synth_shard(q, p, T, R, 0, 0)
exp = bench.Shard(Ys, 10.0)
t0 = time.time()
sess, _ = _session.session_for(exp, p)
c = sess.ctx
print('context + data upload  %7.1f ms' % ((time.time() - t0) * 1e3))
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}


t0 = time.time()
sess.set_params(params)
print('set_params (Gram, K^-1) %7.1f ms' % ((time.time() - t0) * 1e3))


def estep(idx, label):
    t = time.time()
    c.estep_laplace(idx)
    print('%-24s %8.1f ms   chunk %d' % (label, (time.time() - t) * 1e3, c.info('chunk_trials')), flush=True)


one = np.array([0], dtype=np.int32)
estep(one, 'E-step, 1 trial (cold)')
estep(one, 'E-step, 1 trial (warm)')
estep(None, 'E-step, all (first)')
estep(None, 'E-step, all (second)')
