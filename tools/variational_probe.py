"""Dual-variational E-step at config-2 (default), config-3 or config-5 dimensions (argv: trials, config, [lowrank]): the per-trial scipy L-BFGS-B runs driven concurrently (batched device
evaluations) vs the same runs one trial at a time."""
import os, sys, time, faulthandler
if os.environ.get('PROBE_WATCHDOG'):
    faulthandler.dump_traceback_later(int(os.environ['PROBE_WATCHDOG']), exit=True)      # where is a slow run stuck
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import numpy as np
import scipy.optimize as op
import bench, funs
from funs import _session
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = sys.argv[2] if len(sys.argv) > 2 else 'c2'
funs.inference.DUAL_LOWRANK = (len(sys.argv) > 3 and sys.argv[3] == 'lowrank')
q, p, T, _ = dict(bench.CONFIGS, c5=(500, 20, 1000, 256))[cfg]       # c5: config 5's dimensions (variational, 256 trials per GPU at 8 GPUs)
true_params, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
params = {k: np.asarray(v, dtype=np.float64) for k, v in true_params.items()}
params['tau'] = np.linspace(0.1, 0.5, p)
sess, idx = _session.session_for(exp, p)
sess.set_params(params)
funs.inference.dualVariational(bench.Shard(Ys[:2], 10.0), params)            # warm-up
t0 = time.time()
infRes, nll, vlb, opt = funs.inference.dualVariational(exp, params)
t_dev = time.time() - t0
print('default solver [' + funs.inference.DUAL_SOLVER + '] (%s, low-rank dual %s, plan_lowrank %d): %d trials in %.2f s  (nll %.4f, vlb %.4f, iterations %d..%d)' % (
    cfg, funs.inference.DUAL_LOWRANK, sess.ctx.info('plan_lowrank'), R, t_dev, nll, vlb, infRes.dual_iterations.min(), infRes.dual_iterations.max()))
if os.environ.get('PROBE_SCIPY', '1') == '1':
    funs.inference.DUAL_SOLVER = 'scipy'
    t0 = time.time()
    infRes, nll, vlb, opt = funs.inference.dualVariational(exp, params)
    t_conc = time.time() - t0
    funs.inference.DUAL_SOLVER = 'fixedpoint'
    print('concurrent scipy L-BFGS-B: %d trials in %.2f s  (nll %.4f, vlb %.4f)' % (R, t_conc, nll, vlb))
else:
    t_conc = t_dev
# one trial at a time (what the driver did before): first 4 trials, scaled
m = q * T
ctx = sess.ctx
t0 = time.time()
nev = 0
nser = 4 if cfg not in ('c3', 'c5') else 0
for tr in range(nser):
    def f(x, tr=tr):
        global nev
        nev += 1
        return ctx.dual_costgrad(tr, x)
    op.fmin_l_bfgs_b(func=lambda x: f(x)[0], x0=np.zeros(m) + 0.5, fprime=lambda x: f(x)[1], bounds=[(1e-10, None)] * m, factr=1e7, disp=False)
if nser == 0:
    sys.exit(0)
t_ser = (time.time() - t0) / 4
print('serial scipy: %.2f s per trial (%d callback calls each) -> %.1f s for %d trials; device L-BFGS speed-up %.0fx'
      % (t_ser, nev // 4, t_ser * R, R, t_ser * R / t_dev))
