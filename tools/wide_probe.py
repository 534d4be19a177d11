"""Stage timings of one dual evaluation at config-5-like dimensions (argv: neurons latents bins trials)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import numpy as np
from funs import _hip
q, p, T, R = (int(a) for a in sys.argv[1:5])
synth = len(sys.argv) > 5 and sys.argv[5] == 'synth'      # counts drawn from the model (bench.synth_shard) instead of flat noise
t0 = time.time()
def stamp(msg):
    print('[%7.2f s] %s' % (time.time() - t0, msg), flush=True)
rng = np.random.default_rng(0)
C = rng.random((q, p)) - 0.5
d = -1.0 - 2.0 * rng.random(q)
tau = np.linspace(0.1, 0.5, p)
Y = rng.poisson(0.2, (R, q, T)).astype(np.uint8)
if synth:
    import bench
    _, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    Y = np.stack(Ys)
    stamp('synthetic counts: mean %.3f max %d' % (Y.mean(), Y.max()))
ctx = _hip.Context(q, p, T, R, 10.0)
stamp('context')
ctx.upload_counts(Y)
ctx.set_option('cov_mode', int(os.environ.get('PROBE_COV_MODE', '2')))
ctx.set_option('dual_lowrank', 1)
ctx.set_params(C, d, tau)
stamp('set_params')
idx = np.arange(R, dtype=np.int32)
lam = np.full((R, q * T), 0.5)
for rep in range(3):
    cost, grad = ctx.dual_costgrad_batch(idx, lam)
    stamp('dual_costgrad_batch #%d (plan_lowrank %d, rtot %d, chunk %d)' % (rep, ctx.info('plan_lowrank'), ctx.info('lowrank_rtot'), ctx.info('chunk_trials')))
if os.environ.get('PROBE_EVAL_ONLY'):
    sys.exit(0)
rho, fopt, iters = ctx.dual_lbfgs(idx, np.full((R, q * T), np.log(0.5)))
stamp('dual_lbfgs: iterations %s' % iters)
nlp = ctx.dual_finalize(idx, np.exp(rho))
stamp('dual_finalize (nlp %.3f)' % nlp)
n = ctx.mstep_precomp()
stamp('mstep_precomp (%d trials)' % n)
P = ctx.pautosum()
stamp('pautosum %s' % (P.shape,))
m = ctx.post_mean(idx)
v = ctx.post_vsm(idx)
stamp('post_mean / post_vsm %s %s' % (m.shape, v.shape))
